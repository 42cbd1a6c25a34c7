#!/bin/bash
# round 5, call 13: weight layout first on the instruction stream (A/B, interleaved), policy tests
O=gpurun_out/r05m; mkdir -p $O
python -m pytest tests/test_gpu_policy.py tests/test_gpu_round4.py -m gpu -q -x 2>&1 | tail -2
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
for rep in a b; do
WSMG_PRELAYOUT_FIRST=0 $B > $O/bench_old_$rep.json 2> $O/bench_old_$rep.err
$B > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
done
for f in old_a new_a old_b new_b; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"], d["loss"])
except Exception as e:
    print("$f", "FAILED", e); print(open("$O/bench_$f.err").read()[-1500:])
PY
done
