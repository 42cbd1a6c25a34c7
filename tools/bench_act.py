#!/usr/bin/env python3
"""GPU-box tool: rollout step (BasePolicy.act from raw RGB-D, BASELINE configs[0]: B=1, 256^2, E=100, C=64, L=80) —
latency per env-step and where it goes (frozen RGB ResNet-UNet, BEV operator, the rest of the network)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.common.aux_losses import AuxLosses


class _Box:
    shape = (2,)


def obs_of(B, hw, gen):
    ins = torch.zeros(B, 200, dtype=torch.int64, device="cuda")
    ins[:, :80] = torch.randint(1, 2504, (B, 80), device="cuda", generator=gen)
    return {
        "rgb": torch.randint(0, 256, (B, hw, hw, 3), device="cuda", generator=gen).float(),
        "depth": torch.rand(B, 256, 256, 1, device="cuda", generator=gen),
        "depth_features": torch.randn(B, 128, 4, 4, device="cuda", generator=gen),
        "instruction": ins,
        "gps": (torch.rand(B, 2, device="cuda", generator=gen) - 0.5) * 4,
        "compass": (torch.rand(B, 1, device="cuda", generator=gen) - 0.5) * 6.28,
    }


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


AuxLosses.deactivate()
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
for B, mode in ((1, "f32"), (1, "bf16"), (8, "f32"), (8, "bf16")):
    torch.manual_seed(0)
    pol = BasePolicy(None, _Box(), default_model_config(num_proc=B, compute_dtype=mode)).cuda().eval()
    obs = obs_of(B, 256, gen)
    h = torch.zeros(2, B, 512, device="cuda"); prev = torch.zeros(B, 2, device="cuda"); masks = torch.ones(B, 1, device="cuda")
    with torch.no_grad():
        t_act = timed(lambda: pol.act(dict(obs), h.clone(), prev, masks, deterministic=True))
        t_rgb = timed(lambda: pol.net.rgb_encoder(obs))
        emb, proj = pol.net.rgb_encoder(obs)
        t_bev = timed(lambda: pol.net.rgb_mapping_module(proj, dict(obs), masks))
    from wsmgmap.graph import GraphedAct
    ga = GraphedAct(pol)
    hg = h.clone()
    t_graph = timed(lambda: ga(obs, hg, prev, masks, deterministic=True))
    print(f"B={B} {mode}: act() as one HIP graph (wsmgmap.graph.GraphedAct) {t_graph:.2f} ms per env-step ({B / t_graph * 1e3:.0f} env-steps/s)")
    print(f"B={B} {mode}: act() {t_act:.2f} ms per env-step ({B / t_act * 1e3:.0f} env-steps/s) | frozen RGB ResNet-UNet {t_rgb:.2f} ms | "
          f"BEV operator (index, scatter, rotate, fuse, retrieve, rotate) {t_bev:.3f} ms | rest {t_act - t_rgb - t_bev:.2f} ms")
