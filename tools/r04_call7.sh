cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call7
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_round3.py -k "conv or stem or win or slab or wgrad or weight" -x -q > $O/pytest_conv.log 2>&1; echo pytest rc=$?; tail -5 $O/pytest_conv.log
timeout 200 python3 tools/bench_conv.py --dtype bf16 --reps 20 --only enc0 2>&1 | grep "^enc0" | tee $O/enc0.txt
timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > $O/conv_by_layer.txt 2>&1; tail -22 $O/conv_by_layer.txt
B="python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-f32"
for i in 1 2; do
  timeout 300 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done | tee $O/bench.txt
