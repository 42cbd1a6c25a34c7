cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 tools/gpu_only_time.py > gpurun_out/tr.log 2>&1; echo rc=$?; tail -2 gpurun_out/tr.log
f=$(ls gpurun_out/tr/*/*kernel_trace.csv | head -1); python3 tools/rnn_timeline.py $f 7 > gpurun_out/rnn_timeline.txt; grep -v "conv_igemm" gpurun_out/rnn_timeline.txt | head -60; grep -c conv_igemm gpurun_out/rnn_timeline.txt
rm -rf gpurun_out/tr
