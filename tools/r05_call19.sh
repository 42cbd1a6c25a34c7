#!/bin/bash
# round 5, call 19: the chained products' waits as one-workgroup gate launches: recurrent tests x3, then A/B of the update
for i in 1 2 3; do python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_policy.py -m gpu -x -q -k "recurrent or chained or pipelined or update or g3 or occup" 2>&1 | grep -v amdgpu | tail -2 | head -1; done
B="python3 bench.py --gpus 1 --steps 30 --warmup 5 --no-f32 --no-cpu-baseline --no-other-configs"
for rep in a b c; do
  for g in 0 1; do echo -n "gate=$g: "; WSMG_CHAIN_GATE=$g $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['windows']['ms_per_update_by_window'], d['host_ms_per_update'])"; done
done
