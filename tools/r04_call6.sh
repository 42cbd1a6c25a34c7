cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call6
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_round4.py -k "stride2" -x -q > $O/pytest_wgrad.log 2>&1; echo pytest rc=$?; tail -5 $O/pytest_wgrad.log
for L in enc3_k5s2 stem_k7s2; do
  for S in 0 1; do
    echo "== $L WSMG_WGRAD_S2WIN=$S"; WSMG_WGRAD_S2WIN=$S timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only $L 2>&1 | grep "^$L"
  done
done | tee $O/conv_ab_s2.txt
for G in 40 48; do echo "== enc3 groups $G"; WSMG_S2WIN_GROUPS=$G timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only enc3_k5s2 2>&1 | grep "^enc3"; done | tee -a $O/conv_ab_s2.txt
