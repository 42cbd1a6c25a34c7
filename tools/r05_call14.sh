#!/bin/bash
# round 5, call 14: feeder end to end with the rate window ending at the last batch; other_configs legs with the LDS-tiled retrieval
O=gpurun_out/r05n; mkdir -p $O
WSMG_FEEDER_WORKERS=8 WSMG_FEEDER_WORKERS_RAW=1,8,16 timeout 1200 python tools/bench_feeder.py > $O/feeder.txt 2>&1; grep -v amdgpu $O/feeder.txt | tail -24
python - > $O/legs.json 2> $O/legs.err <<PY
import sys, json; sys.path.insert(0, "ws-mgmap_amd"); sys.path.insert(0, ".")
import torch, bench_legs
print(json.dumps(bench_legs.other_configs(torch.device("cuda:0"))))
PY
python - <<PY
import json
d=json.loads(open("$O/legs.json").read().strip().splitlines()[-1])
print({k:(v if not isinstance(v,dict) else v) for k,v in d["cfg4_bev_mapenc"].items() if k in ("us","stage_frac_of_8TBs","frac_of_8TBs","stages")})
PY
