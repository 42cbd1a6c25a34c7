#!/bin/bash
# round 5 against the tree of the round's first commit (git archive 1e57fc9 -> _r04/, built there), the driver's command, ONE box, interleaved
O=gpurun_out/r05_vs_r04; mkdir -p $O
B="bench.py --gpus 1 --steps 40 --warmup 5 --no-f32 --no-cpu-baseline"
for rep in a b c d e f; do
  (cd _r04 && python3 $B > ../$O/r04_$rep.json 2> ../$O/r04_$rep.err)
  python3 $B --no-other-configs > $O/r05_$rep.json 2> $O/r05_$rep.err
done
python3 - <<PY
import json
for rep in "abcdef":
    for t in ("r04", "r05"):
        try:
            d=json.loads(open("$O/%s_%s.json" % (t, rep)).read().strip().splitlines()[-1])
            print(t, rep, d["value"], d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"])
        except Exception as e:
            print(t, rep, "FAILED", e, open("$O/%s_%s.err" % (t, rep)).read()[-600:])
PY
