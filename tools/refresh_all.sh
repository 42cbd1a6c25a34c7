# GPU-box script: the round's one-call evidence set.  usage (via gpurun): bash tools/refresh_all.sh <tag>   -> gpurun_out/all_<tag>/
# Order: the driver's exact bench command FIRST (first GPU process of the lease), then the GPU suite, then the profiler passes.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/all_$TAG
mkdir -p $O && rm -rf $O/*
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo driver-flags bench rc=$?
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -3 $O/pytest_gpu.log
timeout 600 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-f32 > $O/bench_sustained500.json 2> $O/bench_sustained500.err; echo sustained rc=$?
bash tools/refresh_profiles.sh ${TAG}x > $O/refresh.log 2>&1; echo refresh rc=$?
cp gpurun_out/refresh_${TAG}x/bench.json $O/bench_default.json; cp gpurun_out/refresh_${TAG}x/kernel_stats.csv gpurun_out/refresh_${TAG}x/hbm_traffic.json gpurun_out/refresh_${TAG}x/mfma_busy.json gpurun_out/refresh_${TAG}x/hbm_traffic.txt gpurun_out/refresh_${TAG}x/mfma_busy.txt $O/
bash tools/runtrace_update.sh > $O/runtrace_update.log 2>&1; cp gpurun_out/update_timeline.txt $O/
bash tools/runtrace.sh > $O/runtrace.log 2>&1; cp gpurun_out/trace_gaps.txt $O/main_queue_breakdown.txt
timeout 300 python3 tools/bench_conv.py --dtype bf16 --reps 10 > $O/conv_by_layer.txt 2>&1; echo conv rc=$?
timeout 300 python3 tools/section_times.py bf16 8 > $O/update_sections.txt 2>&1; echo sections rc=$?
WSMG_BENCH_DP_ONE_RANK=1 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_dp1_rccl_one_rank.json 2> $O/bench_dp1.err; echo dp1 rc=$?
WSMG_FEEDER_WORKERS=1,8,16 timeout 600 python3 tools/bench_feeder.py > $O/feeder.txt 2>&1; echo feeder rc=$?
timeout 200 python3 tools/bench_bev.py > $O/bev.txt 2>&1; echo bev rc=$?
timeout 200 python3 tools/bench_attn_fp8.py > $O/attn_fp8.txt 2>&1; echo attn rc=$?
timeout 300 python3 tools/bench_act.py > $O/act.txt 2>&1; echo act rc=$?
du -sh $O; cut -c1-700 $O/bench_driver_flags.json; echo; tail -3 $O/conv_by_layer.txt; tail -2 $O/update_sections.txt
