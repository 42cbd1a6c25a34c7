#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -q -k "chained or rows_gemm" 2>&1 | tail -4 > $O/pytest_r5.txt; tail -2 $O/pytest_r5.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline"
WSMG_RECURRENT_CHAIN=0 WSMG_ROWS_GEMM=0 $B --no-other-configs > $O/bench_r4route.json 2> $O/bench_r4route.err
$B > $O/bench_c4.json 2> $O/bench_c4.err
WSMG_RECURRENT_CHAIN=0 WSMG_ROWS_GEMM=0 $B --no-other-configs > $O/bench_r4route_b.json 2> $O/bench_r4route_b.err
$B --no-other-configs > $O/bench_c4_b.json 2> $O/bench_c4_b.err
WSMG_BENCH_DP_ONE_RANK=1 $B --no-other-configs > $O/bench_dp.json 2> $O/bench_dp.err
python tools/section_times.py bf16 8 > $O/sections_c4.txt 2>&1
for f in r4route c4 r4route_b c4_b dp; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"], d["loss"])
    oc=d.get("other_configs")
    if oc: print(json.dumps(oc["cfg5_attn_fp8"])[:500])
except Exception as e:
    print("$f", "FAILED", e); print(open("$O/bench_$f.err").read()[-1500:])
PY
done
grep -v amdgpu $O/sections_c4.txt | sed -n 4,24p
