#!/bin/bash
# round-5 closing call: the driver's command first, then the data-parallel path at one rank, the GPU suite
O=gpurun_out/r05z; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo bench rc $?
WSMG_BENCH_DP_ONE_RANK=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-other-configs > $O/bench_dp1.json 2> $O/bench_dp1.err; echo dp rc $?
python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-f32 --no-other-configs > $O/bench_sustained500.json 2> $O/bench_sustained500.err; echo sustained rc $?
python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids > $O/pytest_gpu_full.txt; grep -B3 -A30 "^E  " $O/pytest_gpu_full.txt | head -80; tail -3 $O/pytest_gpu_full.txt > $O/pytest_gpu_tail.txt; cat $O/pytest_gpu_tail.txt
python3 - <<PY
import json
for f in ("bench_driver_flags","bench_dp1","bench_sustained500"):
    d=json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["windows"]["ms_per_update_by_window"][:6], "host", d["host_ms_per_update"], (d.get("sustained") or {}).get("ms_per_update_second_half"))
PY
