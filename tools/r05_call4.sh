#!/bin/bash
# round-5 GPU call 3: rows-GEMM route of the recurrent core (parity + A/B at 4 / 8 chunks), co-residency soak, exchange stream test
O=gpurun_out/r05d; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_r5.txt; cat $O/pytest_r5.txt
python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "recurrent" 2>&1 | tail -4 > $O/pytest_r4_rec.txt; cat $O/pytest_r4_rec.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
WSMG_ROWS_GEMM=0 $B > $O/bench_rg0.json 2> $O/bench_rg0.err
$B > $O/bench_rg1_c4.json 2> $O/bench_rg1_c4.err
WSMG_RECURRENT_CHUNKS=8 $B > $O/bench_rg1_c8.json 2> $O/bench_rg1_c8.err
WSMG_ROWS_GEMM=0 WSMG_RECURRENT_CHUNKS=8 $B > $O/bench_rg0_c8.json 2> $O/bench_rg0_c8.err
WSMG_BENCH_DP_ONE_RANK=1 $B > $O/bench_dp.json 2> $O/bench_dp.err
python tools/section_times.py bf16 8 > $O/sections_c4.txt 2>&1
WSMG_RECURRENT_CHUNKS=8 python tools/section_times.py bf16 8 > $O/sections_c8.txt 2>&1
for f in rg0 rg1_c4 rg1_c8 rg0_c8 dp; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"], d["loss"])
except Exception as e:
    print("$f", "FAILED", e)
PY
done
tail -n 2 $O/sections_c4.txt; tail -n 2 $O/sections_c8.txt
python tools/bf16_vs_f32_long.py 200 8 $O/bf16_vs_f32_200.json > $O/bf16_vs_f32_200.txt 2>&1; tail -8 $O/bf16_vs_f32_200.txt
