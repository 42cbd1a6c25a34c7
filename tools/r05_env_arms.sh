#!/bin/bash
# runtime environment arms, one box, interleaved
O=gpurun_out/r05_env_arms; mkdir -p $O
B="bench.py --gpus 1 --steps 30 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
for rep in a b c; do
  python3 $B > $O/base_$rep.json 2> $O/base_$rep.err
  env HIP_FORCE_DEV_KERNARG=1 python3 $B > $O/devkernarg_$rep.json 2> $O/devkernarg_$rep.err
  env GPU_MAX_HW_QUEUES=8 python3 $B > $O/q8_$rep.json 2> $O/q8_$rep.err
  env GPU_MAX_HW_QUEUES=2 python3 $B > $O/q2_$rep.json 2> $O/q2_$rep.err
  env HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8 python3 $B > $O/both_$rep.json 2> $O/both_$rep.err
done
python3 - <<PY
import json
for t in ("base", "devkernarg", "q8", "q2", "both"):
    v = []
    for rep in "abc":
        try:
            d=json.loads(open("$O/%s_%s.json" % (t, rep)).read().strip().splitlines()[-1]); v.append((d["ms_per_step"], d["host_ms_per_update"]))
        except Exception as e:
            v.append(None)
    print("%-12s" % t, v)
PY
