cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 tools/gpu_only_time.py > gpurun_out/tr.log 2>&1; echo rc=$?; grep "GPU time" gpurun_out/tr.log
f=$(ls gpurun_out/tr/*/*kernel_trace.csv | head -1); for w in 5 6 7; do python3 tools/trace_update_gaps.py $f $w; done
rm -rf gpurun_out/tr
