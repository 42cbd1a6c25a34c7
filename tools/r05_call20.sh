#!/bin/bash
# round 5, call 20: bev_index with per-sample 32-bit indexing: parity, operator bench, the cfg4 leg
python -m pytest tests -m gpu -x -q -k "bev or map or rollout or act" 2>&1 | grep -v amdgpu | tail -2
python tools/bench_bev.py 2>&1 | grep "index\|ALL fused, 5"
python - <<PY
import sys, json; sys.path.insert(0, "ws-mgmap_amd"); sys.path.insert(0, ".")
import torch, bench_legs
for _ in range(3):
    d = bench_legs.cfg4_bev_mapenc(torch.device("cuda:0")); print(d["us"], d["five_launch_frac_of_8TBs"], d["stage_frac_of_8TBs"], {k: v["us"] for k, v in d["stages"].items()})
PY
