# GPU-side timeline of one update replayed as a HIP graph: bash tools/runtrace_graphed.sh [tag] -> gpurun_out/graphed_timeline[_tag].txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:+_$1}
rm -rf gpurun_out/trg && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trg -- python3 tools/trace_graphed.py 6 > gpurun_out/trg$TAG.log 2>&1; echo rc=$?
python3 tools/update_timeline.py $(ls gpurun_out/trg/*/*kernel_trace.csv | head -1) > gpurun_out/graphed_timeline$TAG.txt; tail -1 gpurun_out/graphed_timeline$TAG.txt; tail -2 gpurun_out/trg$TAG.log
rm -rf gpurun_out/trg
