#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -q -k "fp8 or chained or rows_gemm or raw_depth" 2>&1 | tail -40 > $O/pytest_r5.txt; tail -6 $O/pytest_r5.txt
python -m pytest tests/test_gpu_round4.py tests/test_gpu_kernels.py -m gpu -q -x -k "recurrent or gru or rnn" 2>&1 | tail -5 > $O/pytest_rnn.txt; tail -3 $O/pytest_rnn.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
WSMG_RECURRENT_CHAIN=0 $B > $O/bench_chain0.json 2> $O/bench_chain0.err
$B > $O/bench_chain1_c4.json 2> $O/bench_chain1_c4.err
WSMG_RECURRENT_CHUNKS=8 $B > $O/bench_chain1_c8.json 2> $O/bench_chain1_c8.err
WSMG_RECURRENT_CHUNKS=16 $B > $O/bench_chain1_c16.json 2> $O/bench_chain1_c16.err
WSMG_BENCH_DP_ONE_RANK=1 $B > $O/bench_dp.json 2> $O/bench_dp.err
python tools/section_times.py bf16 8 > $O/sections_c4.txt 2>&1
WSMG_RECURRENT_CHUNKS=8 python tools/section_times.py bf16 8 > $O/sections_c8.txt 2>&1
for f in chain0 chain1_c4 chain1_c8 chain1_c16 dp; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"], d["loss"])
except Exception as e:
    print("$f", "FAILED", e); import subprocess; print(open("$O/bench_$f.err").read()[-1500:])
PY
done
grep -v amdgpu $O/sections_c4.txt | sed -n 4,30p; tail -n 1 $O/sections_c8.txt
