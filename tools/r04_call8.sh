cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call8
mkdir -p $O
for L in cated_k3 enc6_k3 encoded_lin orig0_k3; do
  echo "== $L base"; timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only $L 2>&1 | grep "^$L"
  echo "== $L half-barriers (TIMING ONLY: results are wrong)"; WSMG_LIB=$GRAFT_REPO_ROOT/tools/bin/libwsmgmap_exp.so timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only $L 2>&1 | grep "^$L"
done | tee $O/half_barrier_timing.txt
