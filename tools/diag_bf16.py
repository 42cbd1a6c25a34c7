#!/usr/bin/env python3
"""GPU-box diagnostic: agreement of bf16-mode and f32-mode gradients as a function of batch size
(same weights, same inputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd")):
    sys.path.insert(0, p)
import torch
import bench
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy

def run(mode, T, N, state):
    torch.manual_seed(0)
    pol = BasePolicy(None, bench._Box(), default_model_config(compute_dtype=mode))
    pol.load_state_dict(state)
    pol.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
    pol = pol.cuda(); pol.train(); pol.net.depth_encoder.eval(); pol.net.rgb_encoder.eval()
    obs, prev, masks, weights = bench.synth_batch(T, N, "cuda", 77)
    AuxLosses.activate(); AuxLosses.clear()
    pred, aux = pol(dict(obs), torch.zeros(2, N, 512, device="cuda"), prev, masks, weights)
    loss = bench.dagger_loss(pred, aux, obs["waypoint"], weights)
    loss.backward()
    return pred.detach().float(), float(loss), {n: p.grad.detach().float() for n, p in pol.named_parameters() if p.grad is not None}

torch.manual_seed(0)
state = BasePolicy(None, bench._Box(), default_model_config()).state_dict()
for T, N in [(1, 8), (4, 8), (16, 8), (64, 8)]:
    p32, l32, g32 = run("f32", T, N, state)
    p16, l16, g16 = run("bf16", T, N, state)
    cos = {n: float(torch.nn.functional.cosine_similarity(g32[n].flatten(), g16[n].flatten(), dim=0)) for n in g32 if g32[n].numel() >= 1024}
    worst = sorted(cos.items(), key=lambda t: t[1])[:4]
    allg32 = torch.cat([g.flatten() for g in g32.values()]); allg16 = torch.cat([g16[n].flatten() for n in g32])
    print(f"B={T*N:4d}: logits max diff {float((p32-p16).abs().max()):.2e}  loss {l32:.5f} vs {l16:.5f}  global grad cos {float(torch.nn.functional.cosine_similarity(allg32, allg16, dim=0)):.5f}  worst tensors {[(n[-40:], round(c,4)) for n,c in worst]}")
