# GPU-box script: regenerate the artefacts profiles/ is built from (bench line, kernel stats, HBM traffic passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh && rm -rf gpurun_out/refresh/*
timeout 600 python bench.py > gpurun_out/refresh/bench.json 2> gpurun_out/refresh/bench.err; echo bench rc=$?
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/refresh/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32 > gpurun_out/refresh/stats.log 2>&1; echo stats rc=$?
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/refresh/pb_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 > gpurun_out/refresh/pb_fetch.log 2>&1; echo fetch rc=$?
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/refresh/pb_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 > gpurun_out/refresh/pb_write.log 2>&1; echo write rc=$?
find gpurun_out/refresh -name "*kernel_trace.csv" -delete
find gpurun_out/refresh -type f | head -30; du -sh gpurun_out/refresh
cat gpurun_out/refresh/bench.json | cut -c1-600
