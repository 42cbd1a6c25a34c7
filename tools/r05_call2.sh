#!/bin/bash
# round-5 GPU call 2: parity of the mixed-tile window kernel and the BEV diet, their A/B, the data-parallel exchange A/B, sections
O=gpurun_out/r05b; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q -k "conv or bev or map or window or win3 or wgrad or stride" 2>&1 | tail -4 > $O/pytest_conv_bev.txt
cat $O/pytest_conv_bev.txt
for m in 1 0; do WSMG_CONV_WIN3_MIXED=$m python tools/bench_conv.py --dtype bf16 --data relu > $O/conv_mixed$m.txt 2>&1; done
python tools/bench_bev.py > $O/bev.txt 2>&1
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
$B > $O/bench_single.json 2> $O/bench_single.err
WSMG_BENCH_DP_ONE_RANK=1 $B > $O/bench_dp_hook.json 2> $O/bench_dp_hook.err
WSMG_BENCH_DP_ONE_RANK=1 WSMG_DP_EXCHANGE=instruction $B > $O/bench_dp_instr.json 2> $O/bench_dp_instr.err
WSMG_BENCH_DP_ONE_RANK=1 WSMG_DP_EXCHANGE=instruction WSMG_EARLY_DEDUP_DP=1 $B > $O/bench_dp_instr_early.json 2> $O/bench_dp_instr_early.err
WSMG_BENCH_DP_ONE_RANK=1 WSMG_DP_EXCHANGE=instruction WSMG_EARLY_DEDUP_DP=1 GPU_MAX_HW_QUEUES=4 $B > $O/bench_dp_instr_early_q4.json 2> $O/bench_dp_instr_early_q4.err
WSMG_BENCH_DP_ONE_RANK=1 WSMG_DP_EXCHANGE=decoder WSMG_EARLY_DEDUP_DP=1 $B > $O/bench_dp_dec_early.json 2> $O/bench_dp_dec_early.err
python tools/section_times.py bf16 8 > $O/sections.txt 2>&1
for f in single dp_hook dp_instr dp_instr_early dp_instr_early_q4 dp_dec_early; do python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"], (d.get("data_parallel") or {}).get("exposed_allreduce_ms"))
except Exception as e:
    print("$f", "FAILED", e)
PY
done
tail -3 $O/conv_mixed1.txt $O/conv_mixed0.txt; grep -E "cfg4|cfg1" $O/bev.txt | grep -E "scatter\+rotate|ALL fused, own"
tail -2 $O/sections.txt
