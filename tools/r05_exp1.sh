#!/bin/bash
# round 5, experiment 1b: four LDS stages in the small-channel variants of the window weight-gradient kernel, per layer, interleaved
cd "$(dirname "$0")/.."
WSMG_WIN3W_STAGES_222=4 WSMG_WIN3W_STAGES_412=4 WSMG_WIN3W_STAGES_118=4 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_kernels.py -m gpu -q -x -k "wgrad or weight_grad or conv2d_fwd_bwd" 2>&1 | tail -1
for rep in 1 2; do for st in 3 4; do for l in orig0_k3 orig1_k3 orig2_k3 classified_k3 cls_k3_48; do
  echo -n "stages=$st $l: "; WSMG_WIN3W_STAGES_222=$st WSMG_WIN3W_STAGES_412=$st WSMG_WIN3W_STAGES_118=$st python tools/bench_conv.py --dtype bf16 --data relu --reps 20 --only $l 2>/dev/null | grep "$l" | awk '{print "wgrad", $(NF-1), "ms", $NF, "TF"}'
done; done; done
