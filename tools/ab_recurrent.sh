# A/B of the pipelined recurrent core: bash tools/ab_recurrent.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'])"; }
run WSMG_RECURRENT_CHUNKS=0
run WSMG_RECURRENT_CHUNKS=4
run WSMG_RECURRENT_CHUNKS=4 GPU_MAX_HW_QUEUES=8
run WSMG_RECURRENT_CHUNKS=2 GPU_MAX_HW_QUEUES=8
run WSMG_RECURRENT_CHUNKS=1 GPU_MAX_HW_QUEUES=8
run WSMG_RECURRENT_CHUNKS=1
