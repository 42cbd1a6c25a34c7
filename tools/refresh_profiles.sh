# GPU-box script: regenerate the artefacts profiles/ is built from (bench line, kernel stats, HBM traffic + MFMA passes).
# usage (via gpurun): bash tools/refresh_profiles.sh [tag]   -> gpurun_out/refresh_<tag>/
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/refresh_$TAG
mkdir -p $O && rm -rf $O/*
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo bench rc=$?
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > $O/stats.log 2>&1; echo stats rc=$?
# the counter passes serialise the kernels of a process: the chained recurrent core's device-side waits (a kernel spinning until a kernel
# on another stream has produced its chunk) would spin out their bound there — these passes run the chunk-launch route (same conv kernels)
export WSMG_RECURRENT_CHAIN=0
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pb_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > $O/pb_fetch.log 2>&1; echo fetch rc=$?
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pb_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > $O/pb_write.log 2>&1; echo write rc=$?
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pb_mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 --no-other-configs --prewarm-s 0 > $O/pb_mfma.log 2>&1; echo mfma rc=$?
unset WSMG_RECURRENT_CHAIN
python3 tools/pmc_traffic.py $(find $O/pb_fetch -name "*counter_collection.csv") $(find $O/pb_write -name "*counter_collection.csv") $O/hbm_traffic.json > $O/hbm_traffic.txt
python3 tools/pmc_mfma.py $(find $O/pb_mfma -name "*counter_collection.csv") $O/mfma_busy.json > $O/mfma_busy.txt
cp $(find $O/stats -name "*kernel_stats.csv") $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; cat $O/mfma_busy.txt | head -8; head -6 $O/hbm_traffic.txt; cut -c1-400 $O/bench.json
