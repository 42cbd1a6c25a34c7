#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py -m gpu -q -k "fp8" 2>&1 | tail -12 > $O/pytest_fp8.txt; tail -4 $O/pytest_fp8.txt
python - > $O/legs.json 2> $O/legs.err <<PY
import sys, json; sys.path.insert(0, "ws-mgmap_amd"); sys.path.insert(0, ".")
import torch, bench_legs
print(json.dumps(dict(cfg5=bench_legs.cfg5_attn_fp8(torch.device("cuda:0")))))
PY
python - <<PY
import json
d=json.loads(open("$O/legs.json").read().strip().splitlines()[-1])
c=d["cfg5"]; print({k:c[k] for k in ("us","us_host_paced_loop","launches")}, c["single_query_form"]["us"])
PY
tail -3 $O/legs.err
WSMG_FEEDER_E2E=0 timeout 600 python tools/bench_feeder.py 2>&1 | grep -v amdgpu | tail -4
