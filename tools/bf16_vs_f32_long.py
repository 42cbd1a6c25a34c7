#!/usr/bin/env python3
"""GPU-box tool: K Adam updates on ROTATING synthetic batches from one initial state in the bf16 mode (the bench
mode) and in the float32 parity mode — the reference trains in float32 only (dagger_trainer.py:505-541).  Prints the two loss
curves' gap and, per parameter tensor, the drift |p_bf16 - p_f32| / |p_f32 - p_0|; writes the summary as JSON.
    python tools/bf16_vs_f32_long.py [K=200] [batches=8] [out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
import test_gpu_round2 as r2
from util import NULL_GRAD
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.optim import Adam

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = sys.argv[3] if len(sys.argv) > 3 else None
T, N = 64, 8
state = r2._default_state()
batches = [bench.synth_batch(T, N, "cuda", 500 + i) for i in range(NB)]


def run(mode, snap_at, perturb=0.0):
    st = state
    if perturb:      # the control: the SAME float32 arithmetic from a state that differs in the seventh digit
        g = torch.Generator().manual_seed(99)
        st = {k: (v * (1 + perturb * torch.randn(v.shape, generator=g)) if v.is_floating_point() and "running_" not in k else v) for k, v in state.items()}
    pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype=mode, state=st))
    opt = Adam(pol.parameters(), lr=2.5e-4)
    AuxLosses.activate()
    losses, snaps = [], {}
    for k in range(K):
        obs, prev, masks, weights = batches[k % NB]
        opt.zero_grad(set_to_none=True)
        AuxLosses.clear()
        pred, aux = pol(dict(obs), torch.zeros(2, N, 512, device="cuda"), prev, masks, weights)
        loss = bench.dagger_loss(pred, aux, obs["waypoint"], weights)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
        if k + 1 in snap_at:
            snaps[k + 1] = {n: p.detach().float().cpu() for n, p in pol.named_parameters() if p.requires_grad}
    AuxLosses.deactivate()
    torch.cuda.synchronize()
    return [float(x) for x in losses], snaps


snap_at = sorted({20, 50, 100, K})
l32, s32 = run("f32", snap_at)
torch.cuda.empty_cache()
l16, s16 = run("bf16", snap_at)
p0 = {n: state[n].float().cpu() for n in s32[K]}
torch.cuda.empty_cache()
l32p, s32p = run("f32", snap_at, perturb=1e-6)
res = dict(K=K, batches=NB, loss_f32=[round(v, 5) for v in l32], loss_bf16=[round(v, 5) for v in l16], loss_f32_perturbed=[round(v, 5) for v in l32p])
# loss gap on a smoothed curve (mean over one rotation of the batches) and raw
rel = [abs(a - b) / abs(a) for a, b in zip(l32, l16)]
sm = lambda v: [sum(v[i:i + NB]) / NB for i in range(0, len(v) - NB + 1, NB)]  # noqa: E731
rel_sm = [abs(a - b) / abs(a) for a, b in zip(sm(l32), sm(l16))]
res["max_rel_loss_gap_raw"] = max(rel)
res["max_rel_loss_gap_per_rotation_mean"] = max(rel_sm)
res["drift"] = {}
for k in snap_at:
    moved = float(torch.sqrt(sum(((s32[k][n] - p0[n]) ** 2).sum() for n in p0)))
    drift = float(torch.sqrt(sum(((s32[k][n] - s16[k][n]) ** 2).sum() for n in p0)))
    per = sorted(((float((s32[k][n] - s16[k][n]).norm()) / max(float((s32[k][n] - p0[n]).norm()), 1e-12), n) for n in p0
                  if p0[n].numel() >= 4096 and n not in NULL_GRAD), reverse=True)
    res["drift"][k] = dict(all=round(drift / moved, 4), worst=[(n, round(r, 4)) for r, n in per[:6]],
                           tensors_over_0p35=sum(1 for r, _ in per if r > 0.35), tensors=len(per))
    print(f"after {k:4d} updates: drift over all parameters {drift / moved:.4f}; tensors over 0.35: {res['drift'][k]['tensors_over_0p35']} of {len(per)}; "
          "worst: " + ", ".join(f"{n} {r:.3f}" for r, n in per[:4]))
# the control: float32 against float32 from a state perturbed by 1e-6 (relative) — how far two trajectories of the SAME arithmetic separate
relp_sm = [abs(a - b) / abs(a) for a, b in zip(sm(l32), sm(l32p))]
res["control_f32_vs_f32_perturbed_1e-6"] = {}
for k in snap_at:
    live = [n for n in p0 if float((s32[k][n] - p0[n]).norm()) > 1e-9]       # (parameters that never move differ by the perturbation itself)
    moved = float(torch.sqrt(sum(((s32[k][n] - p0[n]) ** 2).sum() for n in live)))
    drift = float(torch.sqrt(sum(((s32[k][n] - s32p[k][n]) ** 2).sum() for n in live)))
    per = sorted(((float((s32[k][n] - s32p[k][n]).norm()) / max(float((s32[k][n] - p0[n]).norm()), 1e-12), n) for n in p0
                  if p0[n].numel() >= 4096 and n not in NULL_GRAD and float((s32[k][n] - p0[n]).norm()) > 1e-9), reverse=True)
    res["control_f32_vs_f32_perturbed_1e-6"][k] = dict(all=round(drift / moved, 4), worst=[(n, round(r, 4)) for r, n in per[:6]],
                                                        tensors_over_0p35=sum(1 for r, _ in per if r > 0.35))
    print(f"CONTROL f32 vs f32 perturbed 1e-6, after {k:4d} updates: drift {drift / moved:.4f}; tensors over 0.35: "
          f"{res['control_f32_vs_f32_perturbed_1e-6'][k]['tensors_over_0p35']}; worst: " + ", ".join(f"{n} {r:.3f}" for r, n in per[:3]))
res["control_max_rel_loss_gap_per_rotation_mean"] = max(relp_sm)
print(f"CONTROL loss f32 perturbed last rotation mean {sm(l32p)[-1]:.4f}; max per-rotation-mean gap to f32 {max(relp_sm):.3e}")
print(f"loss f32  first/last rotation mean: {sm(l32)[0]:.4f} -> {sm(l32)[-1]:.4f};  bf16: {sm(l16)[0]:.4f} -> {sm(l16)[-1]:.4f}")
print(f"max relative loss gap: raw {max(rel):.3e} (update {rel.index(max(rel))}), per-rotation mean {max(rel_sm):.3e}")
# round 6: the opt-in remedy — the three first-of-chain weight gradients from a 16-mantissa-bit dY
torch.cuda.empty_cache()
l16g, s16g = run("bf16+f32grad", snap_at)
res["loss_bf16_f32grad"] = [round(v, 5) for v in l16g]
res["drift_bf16_f32grad"] = {}
for k in snap_at:
    moved = float(torch.sqrt(sum(((s32[k][n] - p0[n]) ** 2).sum() for n in p0)))
    drift = float(torch.sqrt(sum(((s32[k][n] - s16g[k][n]) ** 2).sum() for n in p0)))
    per = sorted(((float((s32[k][n] - s16g[k][n]).norm()) / max(float((s32[k][n] - p0[n]).norm()), 1e-12), n) for n in p0
                  if p0[n].numel() >= 4096 and n not in NULL_GRAD), reverse=True)
    res["drift_bf16_f32grad"][k] = dict(all=round(drift / moved, 4), worst=[(n, round(r, 4)) for r, n in per[:6]],
                                        tensors_over_0p35=sum(1 for r, _ in per if r > 0.35))
    three = {n: round(r, 4) for r, n in per if n in ("net.map_encoder.cnn.0.weight", "net.map_decoder.base_model.conv1.weight",
                                                    "net.map_decoder.conv_original_size0.0.weight")}
    print(f"bf16+f32grad, after {k:4d} updates: drift {drift / moved:.4f}; tensors over 0.35: {res['drift_bf16_f32grad'][k]['tensors_over_0p35']}; "
          f"the three layers: {three}; worst: " + ", ".join(f"{n} {r:.3f}" for r, n in per[:3]))
print(f"loss bf16+f32grad last rotation mean {sm(l16g)[-1]:.4f} (f32 {sm(l32)[-1]:.4f}, bf16 {sm(l16)[-1]:.4f}, f32 perturbed {sm(l32p)[-1]:.4f}); "
      f"VERDICT's bar: drift <= f32-self + 0.05 = {res['control_f32_vs_f32_perturbed_1e-6'][K]['all'] + 0.05:.3f} at {K} updates, end loss within 1.5 % of f32")
if out:
    json.dump(res, open(out, "w"), indent=1)
