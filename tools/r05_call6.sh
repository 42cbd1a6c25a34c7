#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -q -k "fp8 or raw_depth" 2>&1 | tail -40 > $O/pytest_r5_fp8.txt; tail -5 $O/pytest_r5_fp8.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo bench rc $?
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["windows"]["ms_per_update_by_window"], "host", d["host_ms_per_update"])
print(json.dumps(d["other_configs"]["cfg5_attn_fp8"])[:600])
print(json.dumps(d["other_configs"]["cfg4_bev_mapenc"])[:400])
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/pytest_all.txt; cat $O/pytest_all.txt
