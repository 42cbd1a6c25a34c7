#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
python -m pytest tests/test_gpu_round5.py -m gpu -q 2>&1 | tail -30 > $O/pytest_r5.txt; tail -4 $O/pytest_r5.txt
python - > $O/legs.json 2> $O/legs.err <<PY
import sys, json; sys.path.insert(0, "ws-mgmap_amd"); sys.path.insert(0, ".")
import torch, bench_legs
print(json.dumps(bench_legs.other_configs(torch.device("cuda:0"))))
PY
python - <<PY
import json
d=json.loads(open("$O/legs.json").read().strip().splitlines()[-1])
print(json.dumps(d["cfg5_attn_fp8"])[:900]); print({k:(v if not isinstance(v,dict) else v.get("us")) for k,v in d["cfg4_bev_mapenc"].items() if k in ("us","stage_frac_of_8TBs","frac_of_8TBs","stages")}); print(d["cfg1_act_b1"])
PY
WSMG_FEEDER_WORKERS=8 WSMG_FEEDER_WORKERS_RAW=1,8,16 timeout 900 python tools/bench_feeder.py > $O/feeder.txt 2>&1; tail -22 $O/feeder.txt
