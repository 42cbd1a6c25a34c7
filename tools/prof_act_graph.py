"""rocprofv3 target: 30 replays of the rollout-step graph at B=1 (bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.graph import GraphedAct
import bench_act_helpers as hlp
AuxLosses.deactivate()
B = int(os.environ.get("B", "1"))
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
pol = BasePolicy(None, hlp._Box(), default_model_config(num_proc=B, compute_dtype="bf16")).cuda().eval()
obs = hlp.obs_of(B, 256, gen)
h = torch.zeros(2, B, 512, device="cuda"); prev = torch.zeros(B, 2, device="cuda"); masks = torch.ones(B, 1, device="cuda")
ga = GraphedAct(pol)
for _ in range(int(os.environ.get("REPS", "33"))):
    ga(obs, h, prev, masks, deterministic=True)
torch.cuda.synchronize()
import time
for _ in range(4):      # separated replays (tools/act_timeline.py splits the trace at the idle time between them)
    time.sleep(0.01)
    ga(obs, h, prev, masks, deterministic=True)
    torch.cuda.synchronize()
