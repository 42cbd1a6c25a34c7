# A/B of the 3x3 window kernel (tile, experiment variant) against the implicit-GEMM kernel, warm, one process per point
for v in ${TILES:-0 512 256}; do for x in ${VARIANTS:-0}; do echo "WIN3=$v VARIANT=$x"; for l in ${LAYERS:-cated enc6 encoded orig0}; do WSMG_WIN3_VARIANT=$x WSMG_CONV_WIN3=$v python tools/bench_conv.py --dtype bf16 --reps 20 --only $l | sed -n 2p; done; done; done
