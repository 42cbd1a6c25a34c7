# GPU-box script (round 4, call 3): small-channel window weight gradients (parity, per-layer A/B) and the decoder's issue order (A/B in the policy)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call3
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_round4.py -k "small_channel" tests/test_gpu_round2.py -k "small_channel or window_weight_gradient" -x -q > $O/pytest_wgrad.log 2>&1; echo pytest rc=$?; tail -15 $O/pytest_wgrad.log
timeout 600 python3 -m pytest tests/test_gpu_round3.py -k "slabs" -x -q > $O/pytest_slabs.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest_slabs.log
for L in orig0_k3 orig1_k3 orig2_k3 cls_k3_48 classified_k3; do
  for S in 0 7; do
    echo "== $L WSMG_WIN3W_SMALL=$S"; WSMG_WIN3W_SMALL=$S timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 10 --only $L 2>&1 | grep "^$L"
  done
done | tee $O/conv_ab.txt
B="python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-f32"
for i in 1 2; do
  for cfg in "A=1" "WSMG_DECODER_SIDE_LAST=0" "WSMG_WIN3W_SMALL=0"; do
    echo "== $cfg"; env $cfg timeout 300 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done | tee $O/bench_ab.txt
