#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
WSMG_FP8_TRACE=1 python - 2>&1 <<PY | tail -8
import sys; sys.path.insert(0, "ws-mgmap_amd"); sys.path.insert(0, ".")
import torch
from wsmgmap import ops
B,U,L,C=64,8,160,256
q=torch.randn(B,C,device="cuda"); k=torch.randn(U,L,C,device="cuda"); v=torch.randn(U,L,C,device="cuda")
inv=torch.arange(B,device="cuda")%U; lens=torch.full((U,),L,dtype=torch.int32,device="cuda")
for i in range(6):
    ops.attention_fp8_shared(q,k,v,lens,inv,1/16)
torch.cuda.synchronize()
PY
