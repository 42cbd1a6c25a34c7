#!/bin/bash
# round 5, experiment 1: stages of the window weight-gradient kernel (V421), per layer, interleaved
cd "$(dirname "$0")/.."
for st in 5 6; do
WSMG_WIN3W_STAGES=$st python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or weight_grad or conv2d_fwd_bwd" 2>&1 | tail -1
done
for rep in 1 2; do for st in 3 4 5 6; do for l in enc6_k3 encoded_lin_k3 cated_k3; do
  echo -n "stages=$st $l: "; WSMG_WIN3W_STAGES=$st python tools/bench_conv.py --dtype bf16 --data relu --reps 20 --only $l 2>/dev/null | grep "$l" | awk '{print "wgrad", $(NF-1), "ms", $NF, "TF"}'
done; done; done
