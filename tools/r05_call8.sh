#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -q -k "chained or rows_gemm" 2>&1 | tail -30 > $O/pytest_r5.txt; tail -4 $O/pytest_r5.txt
python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k "recurrent" 2>&1 | tail -3 > $O/pytest_rnn.txt; tail -2 $O/pytest_rnn.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
WSMG_RECURRENT_CHAIN=0 WSMG_ROWS_GEMM=0 $B > $O/bench_r4route.json 2> $O/bench_r4route.err
$B > $O/bench_c4.json 2> $O/bench_c4.err
WSMG_RECURRENT_CHUNKS=8 $B > $O/bench_c8.json 2> $O/bench_c8.err
WSMG_RECURRENT_CHUNKS=2 $B > $O/bench_c2.json 2> $O/bench_c2.err
python tools/section_times.py bf16 8 > $O/sections_c4.txt 2>&1
bash tools/runtrace_update.sh > $O/trace.log 2>&1; cp gpurun_out/update_timeline.txt $O/update_timeline.txt
for f in r4route c4 c8 c2; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"], d["loss"])
except Exception as e:
    print("$f", "FAILED", e); print(open("$O/bench_$f.err").read()[-1500:])
PY
done
grep -v amdgpu $O/sections_c4.txt | sed -n 4,24p
grep -n "rows_gemm\|attn_\|gru_" $O/update_timeline.txt | sed -n 1,40p | cut -c1-140
