cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pb_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 > gpurun_out/pb_fetch.log 2>&1; echo fetch rc=$?
timeout 420 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pb_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-f32 > gpurun_out/pb_write.log 2>&1; echo write rc=$?
ls -la gpurun_out/pb_fetch/*/ gpurun_out/pb_write/*/ 2>/dev/null | head
