#!/bin/bash
# round 5, call 16: the ring that outlives the epoch — tests, then epoch-to-epoch gaps
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_kernels.py -m gpu -x -q -k "feeder or ring or collate" 2>&1 | tail -4
WSMG_FEEDER_WORKERS=1 WSMG_FEEDER_WORKERS_RAW=8 timeout 900 python tools/bench_feeder.py 2>&1 | grep -v amdgpu | grep "epoch ->\|sparse\|dense " | tail -6
