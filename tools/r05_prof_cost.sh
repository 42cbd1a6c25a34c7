#!/bin/bash
# what the live family timing of the bench line costs, one box, interleaved
O=gpurun_out/r05_prof_cost; mkdir -p $O
B="bench.py --gpus 1 --steps 40 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
for rep in a b c; do
  python3 $B > $O/base_$rep.json 2> $O/base_$rep.err
  env WSMG_BENCH_PROF_EVERY=4 python3 $B > $O/every4_$rep.json 2> $O/every4_$rep.err
  env WSMG_BENCH_NOPROF=1 python3 $B > $O/noprof_$rep.json 2> $O/noprof_$rep.err
done
python3 - <<PY
import json
for t in ("base", "every4", "noprof"):
    v = []
    for rep in "abc":
        try:
            d=json.loads(open("$O/%s_%s.json" % (t, rep)).read().strip().splitlines()[-1]); v.append((d["ms_per_step"], d["host_ms_per_update"], (d.get("roofline") or {}).get("achieved")))
        except Exception as e:
            v.append(repr(e)[:80])
    print("%-8s" % t, v)
PY
