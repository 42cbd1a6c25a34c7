cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --prewarm-s 0 > gpurun_out/tr.log 2>&1; echo rc=$?
f=$(ls gpurun_out/tr/*/*kernel_trace.csv | head -1); python3 tools/trace_gaps.py $f 5 70 > gpurun_out/trace_gaps.txt; tail -75 gpurun_out/trace_gaps.txt
rm -rf gpurun_out/tr
