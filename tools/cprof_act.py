#!/usr/bin/env python3
"""GPU-box diagnostic: where the HOST time of an eager rollout step goes (cProfile over 100 act() calls, B=1, bf16)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.common.aux_losses import AuxLosses
import bench_act_helpers as hlp
AuxLosses.deactivate()
B = int(os.environ.get("B", "1"))
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
pol = BasePolicy(None, hlp._Box(), default_model_config(num_proc=B, compute_dtype="bf16")).cuda().eval()
obs = hlp.obs_of(B, 256, gen)
h = torch.zeros(2, B, 512, device="cuda"); prev = torch.zeros(B, 2, device="cuda"); masks = torch.ones(B, 1, device="cuda")


def run(n):
    with torch.no_grad():
        for _ in range(n):
            pol.act(dict(obs), h.clone(), prev, masks, deterministic=True)
    torch.cuda.synchronize()


run(5)
pr = cProfile.Profile()
pr.enable()
run(100)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
