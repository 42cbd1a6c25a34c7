# GPU-box script (round 4, call 1): the driver's exact bench command as the FIRST GPU process of the lease, then the GPU suite,
# then the per-layer conv table and the section times.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call1
mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo bench rc=$?
cut -c1-1500 $O/bench_driver_flags.json
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -5 $O/pytest_gpu.log
timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > $O/conv_by_layer.txt 2>&1; echo conv rc=$?
timeout 300 python3 tools/section_times.py bf16 8 > $O/update_sections.txt 2>&1; echo sections rc=$?
tail -40 $O/update_sections.txt
