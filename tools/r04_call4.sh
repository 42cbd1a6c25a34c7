# GPU-box script (round 4, call 4): per-layer weight-gradient A/B in the product's slab form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call4
mkdir -p $O
for L in orig0_k3 orig1_k3 orig2_k3 cls_k3_48 classified_k3; do
  for S in 0 7; do
    echo "== $L WSMG_WIN3W_SMALL=$S"; WSMG_WIN3W_SMALL=$S timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only $L 2>&1 | grep "^$L"
  done
done | tee $O/conv_ab_slabs.txt
WSMG_WIN3W_SMALL=0 timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > $O/conv_by_layer_slabs_small0.txt 2>&1
timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > $O/conv_by_layer_slabs.txt 2>&1
cat $O/conv_by_layer_slabs.txt
