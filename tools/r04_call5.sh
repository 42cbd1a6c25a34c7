# GPU-box script (round 4, call 5): stride-2 window weight gradients: parity + per-layer A/B + policy
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call5
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_round4.py -k "stride2 or small_channel" -x -q > $O/pytest_wgrad.log 2>&1; echo pytest rc=$?; tail -15 $O/pytest_wgrad.log
timeout 600 python3 -m pytest tests/test_gpu_round3.py -k "slabs" -x -q > $O/pytest_slabs.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest_slabs.log
for L in enc3_k5s2 stem_k7s2; do
  for S in 0 1; do
    echo "== $L WSMG_WGRAD_S2WIN=$S"; WSMG_WGRAD_S2WIN=$S timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only $L 2>&1 | grep "^$L"
  done
done | tee $O/conv_ab_s2.txt
for G in 24 32 40 56 64; do echo "== enc3 groups $G"; WSMG_S2WIN_GROUPS=$G timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only enc3_k5s2 2>&1 | grep "^enc3"; done | tee -a $O/conv_ab_s2.txt
for G in 8 24 32; do echo "== stem groups $G"; WSMG_S2WIN_GROUPS=$G timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only stem_k7s2 2>&1 | grep "^stem"; done | tee -a $O/conv_ab_s2.txt
B="python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-f32"
for i in 1 2; do
  for cfg in "A=1" "WSMG_WGRAD_S2WIN=0"; do
    echo "== $cfg"; env $cfg timeout 300 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done | tee $O/bench_ab.txt
