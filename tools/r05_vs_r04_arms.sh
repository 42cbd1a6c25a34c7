#!/bin/bash
# where the difference between the round's first commit (_r04/) and HEAD sits: HEAD with this round's switches turned back, one box, interleaved
O=gpurun_out/r05_vs_r04_arms; mkdir -p $O
B="bench.py --gpus 1 --steps 30 --warmup 5 --no-f32 --no-cpu-baseline"
OLD="WSMG_PRELAYOUT_FIRST=0 WSMG_RECURRENT_CHAIN=0 WSMG_ROWS_GEMM=0 WSMG_CONV_WIN3_MIXED=0 WSMG_WIN3W_STAGES=3 WSMG_WIN3W_STAGES_222=3 WSMG_WIN3W_STAGES_412=3"
for rep in a b c; do
  (cd _r04 && python3 $B > ../$O/r04_$rep.json 2> ../$O/r04_$rep.err)
  python3 $B --no-other-configs > $O/r05_$rep.json 2> $O/r05_$rep.err
  env $OLD python3 $B --no-other-configs > $O/r05old_$rep.json 2> $O/r05old_$rep.err
  env WSMG_PRELAYOUT_FIRST=0 python3 $B --no-other-configs > $O/r05nopre_$rep.json 2> $O/r05nopre_$rep.err
  env WSMG_RECURRENT_CHAIN=0 WSMG_ROWS_GEMM=0 python3 $B --no-other-configs > $O/r05nochain_$rep.json 2> $O/r05nochain_$rep.err
done
python3 - <<PY
import json
for t in ("r04", "r05", "r05old", "r05nopre", "r05nochain"):
    v = []
    for rep in "abc":
        try:
            d=json.loads(open("$O/%s_%s.json" % (t, rep)).read().strip().splitlines()[-1]); v.append(d["ms_per_step"])
        except Exception as e:
            v.append(None)
    print("%-12s" % t, v)
PY
