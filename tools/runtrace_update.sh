# ordered kernel timeline of one update: bash tools/runtrace_update.sh  -> gpurun_out/update_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > gpurun_out/tr.log 2>&1; echo rc=$?
python3 tools/update_timeline.py $(ls gpurun_out/tr/*/*kernel_trace.csv | head -1) > gpurun_out/update_timeline.txt; tail -1 gpurun_out/update_timeline.txt
rm -rf gpurun_out/tr
