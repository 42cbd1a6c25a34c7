# which kernels on other queues run beside <kernel substring>: bash tools/runtrace_overlap.sh <substr>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32 > gpurun_out/tr.log 2>&1; echo rc=$?
f=$(ls gpurun_out/tr/*/*kernel_trace.csv | head -1); for k in "$@"; do python3 tools/overlap_of.py $f "$k"; done
rm -rf gpurun_out/tr
