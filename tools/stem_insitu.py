"""In-situ duration (HIP events, no profiler) of conv launches inside real updates, grouped by entry point and FLOPs.
usage: python tools/stem_insitu.py [min GFLOP, default 600 = the stem only]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops, _abi
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.optim import Adam
T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
opt = Adam(policy.parameters(), lr=2.5e-4)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()
rec = []
MINF = float(sys.argv[1]) * 1e9 if len(sys.argv) > 1 else 6e11
orig = ops.conv._launch      # (the conv family's launches go through the name bound in wsmgmap.ops.conv)


def launch(name, flops, *args, prof_as=None):
    if flops > MINF and rec is not None and launch.on and "conv" in name:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); _abi.call(name, *args); e.record()
        rec.append((f"{name} {flops / 1e9:7.1f} GF", s, e))
    else:
        orig(name, flops, *args, prof_as=prof_as)


launch.on = False
ops.conv._launch = launch


def update():
    opt.zero_grad(set_to_none=True); AuxLosses.clear()
    h0 = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
    o = dict(obs)
    pred, aux = policy(o, h0, prev, masks, weights)
    bench.dagger_loss(pred, aux, o["waypoint"], weights).backward()
    opt.step()


for _ in range(5): update()
launch.on = True
for _ in range(10): update()
torch.cuda.synchronize()
import collections
acc = collections.defaultdict(list)
for n, s, e in rec: acc[n].append(s.elapsed_time(e))
tot = 0.0
for n, v in sorted(acc.items()):
    print(f"{n}: {sum(v) / len(v):.4f} ms x {len(v) // 10} per update (min {min(v):.4f})")
    tot += sum(v) / 10
print(f"sum per update {tot:.3f} ms")
