# kernels of one update between two anchors: bash tools/runtrace_section.sh <from substr> <occ> <to substr> <occ>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr && timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32 > gpurun_out/tr.log 2>&1; echo rc=$?
f=$(ls gpurun_out/tr/*/*kernel_trace.csv | head -1); python3 tools/section_kernels.py $f 3 "$1" $2 "$3" $4 > gpurun_out/section.txt; head -150 gpurun_out/section.txt
rm -rf gpurun_out/tr
