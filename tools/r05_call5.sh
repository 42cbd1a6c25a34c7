#!/bin/bash
O=gpurun_out/r05e; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -q 2>&1 | tail -60 > $O/pytest_r5.txt; tail -5 $O/pytest_r5.txt
bash tools/runtrace_update.sh > $O/trace.log 2>&1; cp gpurun_out/update_timeline.txt $O/update_timeline.txt
python tools/bf16_vs_f32_long.py 200 8 $O/bf16_vs_f32_200.json > $O/bf16_vs_f32_200.txt 2>&1; tail -12 $O/bf16_vs_f32_200.txt
