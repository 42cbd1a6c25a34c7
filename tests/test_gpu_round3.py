"""GPU parity, round 3 (VERDICT r02 "Next round" 2, 7 and ADVICE r02 high / medium): the BACKWARD pass at the bench size
against the oracle's `loss.backward()`, a 20-update training trajectory in both numeric modes, checkpoint save / resume on
CUDA policies whose rollout caches hold stale operands, and the version counters that those caches key on after writes
that happen behind autograd's back (the multi-tensor Adam kernel, train-mode BatchNorm statistics, graph replays)."""
import os
import sys

import numpy as np
import pytest
import torch

from wsmgmap.debug import sw as _SW     # the package's A/B switches (read once at import; tests flip attributes)

from oracle import policy_ref
from util import NULL_GRAD

import test_gpu_round2 as r2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _aux_losses_off():
    """The registry is process-global (as in the reference): a test that fails between activate() and deactivate() must not
    leave it on for the rollout tests that follow."""
    from wsmgmap.common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    AuxLosses.clear()
    yield
    AuxLosses.deactivate()
    AuxLosses.clear()


def _oracle_grads(state, obs, prev, masks, weights, N, dtype=torch.float32):
    """The oracle's update step on the host in `dtype`: forward + DAgger loss + backward; -> (pred, loss, {state_dict key: grad})."""
    P = {k: (v.detach().cpu().clone().to(dtype) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in state.items()}
    for k, t in P.items():
        if t.is_floating_point() and not k.startswith(("net.rgb_encoder", "net.instruction_encoder.embedding")) \
                and "running_" not in k and k != "net._scale":
            t.requires_grad_(True)
    ref = policy_ref.PolicyRef(P, num_proc=1)
    ref.aux_active = True
    cast = lambda v: v.cpu().to(dtype) if v.is_floating_point() else v.cpu()   # noqa: E731
    oc = {k: cast(v) for k, v in obs.items()}
    torch.set_default_dtype(dtype)
    try:
        pred, aux, _, _ = ref.forward(oc, torch.zeros(2, N, 512, dtype=dtype), cast(prev), cast(masks), cast(weights))
        loss, _ = policy_ref.dagger_loss(pred, aux, oc["waypoint"], cast(weights))
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return pred.detach(), float(loss.detach()), {k: t.grad for k, t in P.items() if t.requires_grad and t.grad is not None}


# ----------------------------------------------------------------------------- backward at the bench size vs the oracle
def test_bench_workload_f32_backward_vs_oracle_full_size():
    """BASELINE configs[1] exactly as bench.py builds it (T=64 x N=8 = 512 rows), float32 mode: EVERY gradient tensor of
    the update against the oracle's `loss.backward()` evaluated on this box's host cores (models/policy.py:91-103 +
    dagger_trainer.py:526-541 of the reference; the 30-60 s leg bench.py's cpu_baseline also runs).

    Bars.  The whole gradient: cosine >= 0.99999 against the float32 oracle AND the float64 oracle (measured 1.00000000 both), the
    same live set.  Per tensor the asked-for bar is max|g - g_ref| <= 1e-3 max|g_ref| — which float32 arithmetic itself does not
    meet on this workload: through 13 train-mode BatchNorms over 73 728-294 912 elements per channel, the float32 CPU oracle sits up
    to 1.1e-2 of max|grad| from the SAME oracle evaluated in float64 (the truth; ≈4x the float32 time), and only 53 of its 91
    tensors are within 1e-3; this path: up to 1.7e-2, 52 of 91 (measured on MI355X, printed).  So every tensor is held against
    the float64 truth: within 1e-3 of it, or — the rule of test_update_path_gradients_full_tensor_vs_oracle at B = 6 — no farther
    than the float32 CPU oracle by more than a factor (max error: 5x, floor 2e-3; relative L2 error: 4x, floor 5e-4).  The MFMA
    engine adds a reduction sequentially in float32 (chains of 2 304-12 544 products, pixel sums of 73 728 / nsplit), oneDNN in
    blocks: isolated elements are 2-4x farther from the truth, the direction of every tensor is the same to 8 digits."""
    import bench
    from wsmgmap.common.aux_losses import AuxLosses
    Tn, N = 64, 8
    state = r2._default_state()
    obs, prev, masks, weights = bench.synth_batch(Tn, N, "cuda", 77)
    pol = r2._train_mode(r2._policy(num_proc=1, state=state))
    AuxLosses.activate()
    AuxLosses.clear()
    pred, aux = pol(dict(obs), torch.zeros(2, N, 512, device="cuda"), prev, masks, weights)
    loss = bench.dagger_loss(pred, aux, obs["waypoint"], weights)
    loss.backward()
    AuxLosses.deactivate()
    got = {n: p.grad.detach().cpu() for n, p in pol.named_parameters() if p.grad is not None}
    pr, lr, want = _oracle_grads(state, obs, prev, masks, weights, N)
    assert float((pred.detach().cpu() - pr).abs().max()) <= 1e-4 and abs(float(loss.detach()) - lr) <= 1e-4
    _, l64, truth = _oracle_grads(state, obs, prev, masks, weights, N, torch.float64)
    assert abs(float(loss.detach()) - l64) <= 1e-5
    live_ref = {k for k, g in want.items() if float(g.abs().max()) > 0}
    live_got = {k for k, g in got.items() if float(g.abs().max()) > 0}
    missing = sorted(k for k in live_ref - set(got) if k not in NULL_GRAD)
    assert not missing, f"the oracle has gradients the HIP path lacks: {missing[:6]}"
    extra = sorted(k for k in live_got - set(want))
    assert not extra, f"the HIP path has gradients the oracle lacks: {extra[:6]}"
    rows, l2 = [], {}
    for k in sorted(set(got) & set(want)):
        if k in NULL_GRAD:
            continue
        t = truth[k]
        scale = float(t.abs().max())
        if scale == 0.0:
            assert float(got[k].abs().max()) <= 1e-12, k
            continue
        e_hip = float((got[k].double() - t).abs().max()) / scale
        e_f32 = float((want[k].double() - t).abs().max()) / scale
        e_dir = float((got[k] - want[k]).abs().max()) / float(want[k].abs().max())
        l2[k] = (float((got[k].double() - t).norm() / t.norm()), float((want[k].double() - t).norm() / t.norm()))
        rows.append((e_hip, e_f32, e_dir, k))
    a = torch.cat([got[k].flatten() for *_, k in rows]).double()
    b = torch.cat([want[k].flatten() for *_, k in rows]).double()
    c = torch.cat([truth[k].flatten() for *_, k in rows])
    cos32 = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
    cos64 = float(torch.nn.functional.cosine_similarity(a, c, dim=0))
    rows.sort(reverse=True)
    print(f"B=512 f32 backward: {len(rows)} tensors; cosine of the whole gradient vs the float32 oracle {cos32:.8f}, vs the float64 "
          f"oracle {cos64:.8f}; farthest from the float64 truth (HIP / float32 CPU oracle / HIP-vs-f32-oracle, relative to max|grad|): "
          + "; ".join(f"{k} {eh:.1e} / {e3:.1e} / {ed:.1e}" for eh, e3, ed, k in rows[:5]))
    within = sum(1 for eh, *_ in rows if eh <= 1e-3)
    print(f"{within} of {len(rows)} tensors within 1e-3 of the float64 truth; the float32 CPU oracle: {sum(1 for _, e3, *_ in rows if e3 <= 1e-3)}")
    wl2 = sorted(((l2[k][0], l2[k][1], k) for k in l2), reverse=True)
    print("largest relative L2 errors vs the float64 truth (HIP / float32 CPU oracle): " + "; ".join(f"{k} {a_:.1e} / {b_:.1e}" for a_, b_, k in wl2[:5]))
    bad = [(k, f"hip {eh:.2e}", f"f32 oracle {e3:.2e}") for eh, e3, _, k in rows if eh > max(2e-3, 5.0 * e3)]
    assert not bad, f"gradient tensors farther from the float64 oracle than float32 arithmetic explains: {bad[:8]}"
    bad = [(k, f"hip {a_:.2e}", f"f32 oracle {b_:.2e}") for a_, b_, k in wl2 if a_ > max(5e-4, 4.0 * b_)]
    assert not bad, f"relative L2 error vs the float64 oracle beyond what float32 arithmetic explains: {bad[:8]}"
    assert cos32 >= 0.99999 and cos64 >= 0.99999, (cos32, cos64)


def test_bf16_mode_gradient_cosines_full_size_tight():
    """bf16 against float32 at B=512, per tensor.  The round-2 bar (cosine >= 0.95 for every tensor with >= 4096 elements) was a
    guess; measured on MI355X (printed): whole gradient 0.999989, lowest tensors 0.9617-0.973 — the FIRST layers of the three
    backward chains (map_encoder.cnn.0/3/6, the decoder's stem and first block), whose gradients have crossed the most bf16
    roundings; every other tensor >= 0.975.  Bars: whole gradient >= 0.9999, every tensor >= 0.955.  A per-tensor cosine cannot
    tell accumulated rounding noise from a structural error in a small layer (one wrong tap of nine is cosine 0.94), so the
    error is also required to be noise-like: for every convolution weight, the cosine of EACH tap [:, :, ky, kx] must be
    within 0.04 of the tensor's (a wrong tap would sit near 0)."""
    import test_gpu_policy as tp
    state = r2._default_state()
    p32, l32, g32 = tp._bench_like_update("f32", 64, 8, state)
    torch.cuda.empty_cache()
    p16, l16, g16 = tp._bench_like_update("bf16", 64, 8, state)
    torch.cuda.empty_cache()
    cosf = lambda x, y: float(torch.nn.functional.cosine_similarity(x.flatten().double(), y.flatten().double(), dim=0))  # noqa: E731
    rows, taps = [], []
    for n in g32:
        if n in NULL_GRAD or g32[n].numel() < 4096 or float(g32[n].norm()) < 1e-6:
            continue
        c = cosf(g32[n], g16[n])
        rows.append((c, n))
        if g32[n].dim() == 4 and g32[n].shape[2] * g32[n].shape[3] > 1:
            per = [cosf(g32[n][:, :, ky, kx], g16[n][:, :, ky, kx]) for ky in range(g32[n].shape[2]) for kx in range(g32[n].shape[3])]
            taps.append((min(per) - c, n))
    rows.sort()
    taps.sort()
    a = torch.cat([g32[n].flatten() for n in g32]).double()
    b = torch.cat([g16[n].flatten() for n in g32]).double()
    cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
    print(f"bf16 vs f32 at B=512: global cosine {cos:.6f}; lowest per-tensor cosines: " + ", ".join(f"{n} {c:.5f}" for c, n in rows[:6])
          + "; worst tap below its tensor: " + ", ".join(f"{n} {d:+.4f}" for d, n in taps[:3]))
    assert cos >= 0.9999, cos
    low = [(n, round(c, 5)) for c, n in rows if c < 0.955]
    assert not low, low[:8]
    odd = [(n, round(d, 4)) for d, n in taps if d < -0.04]
    assert not odd, f"one tap of a convolution's weight gradient is far worse than the rest: {odd[:6]}"


# ----------------------------------------------------------------------------- a training trajectory in both modes
def _trajectory(mode, state, K, obs, prev, masks, weights, N):
    import bench
    from wsmgmap.common.aux_losses import AuxLosses
    from wsmgmap.optim import Adam
    pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype=mode, state=state))
    opt = Adam(pol.parameters(), lr=2.5e-4)
    AuxLosses.activate()
    losses = []
    for _ in range(K):
        opt.zero_grad(set_to_none=True)
        AuxLosses.clear()
        pred, aux = pol(dict(obs), torch.zeros(2, N, 512, device="cuda"), prev, masks, weights)
        loss = bench.dagger_loss(pred, aux, obs["waypoint"], weights)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    AuxLosses.deactivate()
    params = {n: p.detach().float().cpu() for n, p in pol.named_parameters() if p.requires_grad}
    return losses, params


def test_bf16_training_trajectory_tracks_f32():
    """K = 20 updates (Adam, lr 2.5e-4, the bench inputs at T=64 x N=8) from ONE initial state in the bf16 mode (the headline) and in
    the float32 parity mode (dagger_trainer.py:526-541 around models/policy.py:91-103).  Written bars (measured on MI355X, printed: largest
    gaps 9.6e-4 and 4.0e-3 at steps 18-19 on two boxes, <= 7e-4 through step 14 on both; drift 0.245-0.247): the two loss curves
    within 0.2 % of each other through step 14 and within 1 % to the end (two trajectories that differ in rounding separate
    geometrically: the gap grows 1e-4 -> 4e-3 over the last 7 updates, and the float32 curve itself moved 6e-4 at step 19 when
    only the heads' summation order changed between two commits of this round); both fall; the parameter
    drift between the modes after 20 updates — |p_bf16 - p_f32| relative to the distance |p_f32 - p_0| travelled, over all
    parameters — stays below 0.30 (Adam divides by the gradient's magnitude, so the sign noise of bf16 on near-zero gradient
    elements moves parameters at full step size: the first layers of each chain drift 0.57-0.58)."""
    import bench
    K, Tn, N = 20, 64, 8
    state = r2._default_state()
    obs, prev, masks, weights = bench.synth_batch(Tn, N, "cuda", 77)
    l32, p32 = _trajectory("f32", state, K, obs, prev, masks, weights, N)
    torch.cuda.empty_cache()
    l16, p16 = _trajectory("bf16", state, K, obs, prev, masks, weights, N)
    torch.cuda.empty_cache()
    rel = [abs(a - b) / abs(a) for a, b in zip(l32, l16)]
    p0 = {n: state[n].float().cpu() for n in p32}
    moved = float(torch.sqrt(sum(((p32[n] - p0[n]) ** 2).sum() for n in p32)))
    drift = float(torch.sqrt(sum(((p32[n] - p16[n]) ** 2).sum() for n in p32)))
    per = sorted(((float((p32[n] - p16[n]).norm()) / max(float((p32[n] - p0[n]).norm()), 1e-12), n) for n in p32
                  if p32[n].numel() >= 4096 and n not in NULL_GRAD), reverse=True)
    print("loss f32 : " + " ".join(f"{v:.4f}" for v in l32))
    print("loss bf16: " + " ".join(f"{v:.4f}" for v in l16))
    print(f"max relative loss gap {max(rel):.3e} (step {rel.index(max(rel))}); parameter drift |p16 - p32| / |p32 - p0| = {drift / moved:.4f}; "
          "largest per tensor: " + ", ".join(f"{n} {r:.3f}" for r, n in per[:4]))
    assert all(np.isfinite(l16)) and all(np.isfinite(l32))
    assert l32[-1] < l32[0] and l16[-1] < l16[0], "the loss did not fall over 20 updates"
    assert max(rel[:15]) <= 2e-3, (max(rel[:15]), rel.index(max(rel[:15])))
    assert max(rel) <= 1e-2, (max(rel), rel.index(max(rel)))
    assert drift / moved <= 0.30, drift / moved


# ----------------------------------------------------------------------------- version counters behind raw-pointer writes
def test_rollout_caches_follow_adam_kernel_and_batchnorm_statistics():
    """`wsmgmap.optim.Adam` writes the parameters, and train-mode BatchNorm its running statistics, through
    raw pointers in HIP kernels.  The rollout route's FoldCache / packed LSTM weights key on autograd version counters, so
    those writes must advance them.  The reference's DAgger loop (dagger_trainer.py:648-665) alternates updates with eval() +
    no_grad act(): after real updates (WsmgAdam + train-mode forwards) between rollout steps, the policy's eager act() and
    its GraphedAct replay — both reading caches filled BEFORE the updates — must equal, bit for bit, a twin policy that
    loads the updated state_dict and starts with NO caches."""
    import bench
    from wsmgmap.common.aux_losses import AuxLosses
    from wsmgmap.graph import GraphedAct
    from wsmgmap.optim import Adam
    B, Tn, N = 2, 4, 2
    st0 = r2._default_state()      # (default init: the hash-filled golden weights are deliberately ill-conditioned and do not train)
    pa, pb = r2._policy(num_proc=B, compute_dtype="bf16", state=st0), r2._policy(num_proc=B, compute_dtype="bf16", state=st0)
    ga = GraphedAct(pa, eager_calls=2)
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    obs_u, prev_u, masks_u, weights_u = bench.synth_batch(Tn, N, "cuda", 5)
    opt = Adam(pa.parameters(), lr=2e-3)      # a large step: stale operands must be visible

    def train_updates(k):
        r2._train_mode(pa)
        AuxLosses.activate()
        for _ in range(k):
            opt.zero_grad(set_to_none=True)
            AuxLosses.clear()
            pred, aux = pa(dict(obs_u), torch.zeros(2, N, 512, device="cuda"), prev_u, masks_u, weights_u)
            bench.dagger_loss(pred, aux, obs_u["waypoint"], weights_u).backward()
            opt.step()
        AuxLosses.deactivate()
        AuxLosses.clear()
        pa.eval()

    def twin_follows():      # the twin gets the state through load_state_dict and forgets every derived operand
        pb.load_state_dict(pa.state_dict())
        pb.net._fold = None
        pb.net.instruction_encoder._packed = None

    ha, hb = torch.zeros(2, B, 512, device="cuda"), torch.zeros(2, B, 512, device="cuda")
    prev, masks = torch.zeros(B, 2, device="cuda"), torch.ones(B, 1, device="cuda")
    pa.eval(); pb.eval()
    for k in range(7):
        if k in (3, 5):      # updates between rollout steps, after the caches were filled (k = 0, 1) and the graph captured (k = 2)
            v_before = pa.net.map_encoder.cnn[0].weight._version, pa.net.map_encoder.cnn[1].running_mean._version
            train_updates(2)
            assert pa.net.map_encoder.cnn[0].weight._version > v_before[0], "Adam's kernel write did not advance the version counter"
            assert pa.net.map_encoder.cnn[1].running_mean._version > v_before[1], "BatchNorm's statistics write did not advance it"
            twin_follows()
        obs = r2._rollout_obs(B, gen)
        with torch.no_grad():
            vb, ab, lb, hb = pb.act(dict(obs), hb, prev, masks, deterministic=True)
        va, aa, la, hn = ga(obs, ha, prev, masks, deterministic=True)
        assert torch.isfinite(hn).all() and torch.isfinite(aa).all(), k
        assert torch.equal(ha, hn), "GraphedAct did not leave the new hidden state in the caller's tensor"
        for name, x, y in (("value", va, vb), ("action", aa, ab), ("logp", la, lb), ("h", hn, hb),
                           ("map", pa.net.rgb_mapping_module.full_global_map, pb.net.rgb_mapping_module.full_global_map)):
            assert torch.equal(x, y), (k, name, float((x.float() - y.float()).abs().max()))
        hb = hb.clone()
        prev = ab.clone()
    assert len(ga._graphs) == 1


def test_graphed_update_replays_advance_versions_and_leave_inputs_alone():
    """GraphedUpdate adopted the caller's first batch as the graph's static inputs and later copied
    other batches INTO it.  Now inputs are cloned at capture unless registered as static buffers; a replay advances the
    version counters of everything it rewrote (parameters, BatchNorm statistics); the graph cache is bounded."""
    import bench
    from wsmgmap.graph import GraphedUpdate
    from wsmgmap.optim import Adam
    Tn, N = 4, 2
    pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype="bf16", state=r2._default_state()))
    opt = Adam(pol.parameters(), lr=2.5e-4, capturable=True)
    gu = GraphedUpdate(pol, opt, lambda pred, aux, o, w: bench.dagger_loss(pred, aux, o["waypoint"], w), eager_calls=2)
    from wsmgmap.common.aux_losses import AuxLosses
    AuxLosses.activate()
    batches = [bench.synth_batch(Tn, N, "cuda", 40 + i) for i in range(3)]
    keep = [{k: v.clone() for k, v in b[0].items()} for b in batches]
    w = pol.net.map_encoder.cnn[0].weight
    rm = pol.net.map_encoder.cnn[1].running_mean
    for it in range(8):
        obs, prev, masks, weights = batches[it % 3]
        v0, r0 = w._version, rm._version
        hs = torch.zeros(2, N, 512, device="cuda")
        loss = gu(obs, hs, prev, masks, weights)
        assert np.isfinite(float(loss))
        assert w._version > v0 and rm._version > r0, (it, "a graphed update left the version counters where they were")
    AuxLosses.deactivate()
    for b, k in zip(batches, keep):      # the caller's batches are untouched
        for name in k:
            assert torch.equal(b[0][name], k[name]), name
    assert len(gu._graphs) == 1
    gu.max_graphs = 1
    obs2, prev2, masks2, weights2 = bench.synth_batch(6, N, "cuda", 50)     # another signature: the first graph is evicted
    AuxLosses.activate()
    gu(obs2, torch.zeros(2, N, 512, device="cuda"), prev2, masks2, weights2)
    AuxLosses.deactivate()
    assert len(gu._graphs) == 1


# ----------------------------------------------------------------------------- checkpoint save / resume on CUDA policies
def test_checkpoint_round_trip_on_cuda_policies_with_stale_rollout_caches(tmp_path):
    """common_trainer.py:91-139 on the GPU: save from a CUDA policy after two bf16 updates (`save_checkpoint` through
    `checkpoint.epoch_end`, the trainer's end-of-epoch step, dagger_trainer.py:636-655), `resume_dagger` into a FRESH CUDA
    policy that has already run one eager act() and one GraphedAct replay — so its folded convolution operands, packed LSTM
    weights and captured graph hold the operands of its OWN initial parameters — and require the next eager act() and the
    next replay to be bit-identical to the saving policy's."""
    import bench
    from wsmgmap import checkpoint as ck
    from wsmgmap.common.aux_losses import AuxLosses
    from wsmgmap.graph import GraphedAct
    from wsmgmap.optim import Adam
    B, Tn, N, EPOCHS = 2, 4, 2, 12
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    obs_u, prev_u, masks_u, weights_u = bench.synth_batch(Tn, N, "cuda", 8)
    saver = r2._train_mode(r2._policy(num_proc=B, compute_dtype="bf16", state=r2._default_state()))
    opt = Adam(saver.parameters(), lr=2e-3)
    AuxLosses.activate()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        AuxLosses.clear()
        pred, aux = saver(dict(obs_u), torch.zeros(2, N, 512, device="cuda"), prev_u, masks_u, weights_u)
        bench.dagger_loss(pred, aux, obs_u["waypoint"], weights_u).backward()
        opt.step()
    seen = {}

    def on_epoch_end(policy, ckpt_path):      # the in-train evaluation hook: policy in eval mode, checkpoint on disk
        seen["training"], seen["path"] = policy.training, ckpt_path
        seen["encoders"] = (policy.net.depth_encoder.training, policy.net.rgb_encoder.training)
    path = ck.epoch_end(saver, str(tmp_path), dagger_it=1, epoch=3, epochs=EPOCHS, config={"lr": 1e-2}, local_rank=0, on_epoch_end=on_epoch_end)
    assert os.path.basename(path) == f"ckpt.{1 * EPOCHS + 3}.pth" and seen == {"training": False, "path": path, "encoders": (False, False)}
    assert saver.training and not saver.net.depth_encoder.training and not saver.net.rgb_encoder.training and AuxLosses.is_active()
    AuxLosses.deactivate()
    AuxLosses.clear()

    # the resuming policy: different parameters (hash-filled, then shaken), caches and graph warmed on THOSE
    fresh = r2._policy(num_proc=B, compute_dtype="bf16").eval()
    r2._shake_batchnorm(fresh, 21)
    gf = GraphedAct(fresh, eager_calls=1)
    h = torch.zeros(2, B, 512, device="cuda")
    prev, masks = torch.zeros(B, 2, device="cuda"), torch.ones(B, 1, device="cuda")
    warm = r2._rollout_obs(B, gen)
    gf(warm, h, prev, masks, deterministic=True)           # eager act(): fills FoldCache + packed LSTM weights
    gf(warm, h, prev, masks, deterministic=True)           # capture + first replay
    assert len(gf._graphs) == 1
    d_it, e_it, report = ck.resume_dagger(fresh, str(tmp_path), epochs=EPOCHS)
    assert (d_it, e_it) == (1, 4) and not report.missing_keys and not report.unexpected_keys
    for (k, a), (_, b) in zip(saver.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b) and (not a.is_floating_point() or torch.isfinite(a).all()), k

    saver.eval()
    for pol in (saver, fresh):      # same map state on both sides (the trainer re-creates it, dagger_trainer.py:668-678)
        ck.dagger_iteration_end(pol, B, local_rank=1)
        pol.eval()
    hs, hf, hg = (torch.zeros(2, B, 512, device="cuda") for _ in range(3))
    for step in range(3):
        obs = r2._rollout_obs(B, gen)
        with torch.no_grad():
            out_s = saver.act(dict(obs), hs, prev, masks, deterministic=True)
        map_s = saver.net.rgb_mapping_module.full_global_map.clone()
        if step < 2:       # eager act() on the resumed policy
            with torch.no_grad():
                out_f = fresh.act(dict(obs), hf, prev, masks, deterministic=True)
            hcmp = hf
        else:              # and the captured graph (its map state continues from the eager steps)
            hg.copy_(hf)
            out_f = gf(obs, hg, prev, masks, deterministic=True)
            hcmp = hg
        for name, x, y in (("value", out_f[0], out_s[0]), ("action", out_f[1], out_s[1]), ("logp", out_f[2], out_s[2]), ("h", hcmp, hs),
                           ("map", fresh.net.rgb_mapping_module.full_global_map, map_s)):
            assert torch.equal(x, y), (step, name, float((x.float() - y.float()).abs().max()))
        prev = out_s[1].clone()


# ----------------------------------------------------------------------------- deterministic weight gradients (slab reduce)
@pytest.mark.parametrize("geom", [
    (16, 100, 100, 64, 64, 8, 2, 3, 64),     # the stem: LDS-window kernel, one slab per tile group
    (64, 24, 24, 256, 256, 3, 1, 1, 256),    # 3 x 3 window kernel, one slab per image range
    (32, 50, 50, 64, 128, 5, 2, 1, 64),      # generic kernel, stride 2
    (48, 48, 48, 32, 32, 3, 1, 1, 27),       # 27 real input channels in a 32-channel activation (the padded ones are dropped)
    (8, 24, 24, 192, 64, 3, 1, 1, 192),      # 6 channel chunks, 64 outputs
    (3, 6, 6, 64, 64, 3, 1, 1, 64),          # fewer pixels than one split wants
    (40, 24, 24, 256, 256, 1, 1, 0, 256),    # 1 x 1
], ids=["stem", "win3", "k5s2", "cin27", "c192", "tiny", "1x1"])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_weight_gradient_slabs_are_deterministic_and_match_atomics(geom, dtype, monkeypatch):
    """wsmg_conv2d_bwd_weight[_bf16]_plan / _slabs + wsmg_weight_grad_reduce_oihw (run.py:107-108 of the reference: deterministic
    kernels): the slab form equals the atomic form of the same kernels up to float32 summation order, equals the float64
    torch weight gradient of the same (rounded) operands within the engine's bar, and is BIT-IDENTICAL over repeated launches;
    the plan's workspace contract is enforced."""
    import ctypes
    from wsmgmap import _abi, ops
    B, H, W, Cin, Cout, K, stride, pad, Cin_w = geom
    if dtype == "f32" and B * H * W * Cin > 40e6:
        B = max(2, B // 4)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    sfx = "_bf16" if dtype == "bf16" else ""
    OH, OW = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    g = torch.Generator(device="cuda"); g.manual_seed(B * 7 + Cin + K)
    x = torch.relu(torch.randn(B, H, W, Cin, device="cuda", generator=g)).to(dt)
    if Cin_w < Cin:
        x[..., Cin_w:] = 0
    dy = (torch.randn(B, OH, OW, Cout, device="cuda", generator=g) * 0.1).to(dt)
    dims = (B, H, W, Cin, Cout, K, K, stride, pad, OH, OW)
    fl = 0.0
    outs = [ops._weight_grad(sfx, x, dy, dims, fl, Cin_w) for _ in range(3)]
    torch.cuda.synchronize()
    assert outs[0].shape == (Cout, Cin_w, K, K)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "the slab-reduced weight gradient is not repeatable"
    monkeypatch.setattr(_SW, "wgrad_atomics", True)
    ref_atomic = ops._weight_grad(sfx, x, dy, dims, fl, Cin_w)
    monkeypatch.setattr(_SW, "wgrad_atomics", False)
    xs = x.double().permute(0, 3, 1, 2)[:, :Cin_w]
    dys = dy.double().permute(0, 3, 1, 2)
    want = torch.nn.grad.conv2d_weight(xs, (Cout, Cin_w, K, K), dys, stride=stride, padding=pad)
    scale = float(want.abs().max())
    assert float((outs[0].double() - want).abs().max()) <= 2e-4 * scale, float((outs[0].double() - want).abs().max()) / scale
    assert float((outs[0] - ref_atomic).abs().max()) <= 2e-4 * scale
    # the workspace contract: a plan for another geometry / a short workspace is refused
    ns, fl_ = ctypes.c_int(0), ctypes.c_longlong(0)
    _abi.call("wsmg_conv2d_bwd_weight" + sfx + "_plan", *dims, ctypes.cast(ctypes.byref(ns), ctypes.c_void_p), ctypes.cast(ctypes.byref(fl_), ctypes.c_void_p))
    assert ns.value >= 1 and fl_.value == ns.value * Cout * K * K * Cin
    ws = torch.empty(fl_.value, device="cuda")
    for bad_ns, bad_fl in ((ns.value + 1, fl_.value), (ns.value, fl_.value - 1)):
        rc = getattr(_abi.lib(), "wsmg_conv2d_bwd_weight" + sfx + "_slabs")(ops._p(x), ops._p(dy), ops._p(ws), bad_ns, bad_fl, *dims, ops._stream())
        assert rc == -1, rc      # WSMG_EINVAL


# ----------------------------------------------------------------------------- fused heads / auxiliary reduction / trainer loss
@pytest.mark.parametrize("shape", [(64, 8, 512, 2), (5, 3, 512, 2), (1, 1, 256, 3), (7, 2, 640, 4)], ids=["bench", "ragged", "b1", "a4"])
def test_update_heads_aux_reduce_and_dagger_loss_match_the_reference_lines(shape):
    """csrc/wsmg_heads.hip against the reference's own torch lines (models/policy.py:59,86-88,96-97; common/aux_losses.py:24-35;
    dagger_trainer.py:526-534) evaluated in float64: values within 2e-6, every gradient within 2e-6 of max|grad|; masked rows are
    dropped (a NaN loss on a masked row stays out of the value AND of the gradients); results are bit-identical across runs."""
    import torch.nn as nn
    import torch.nn.functional as F
    from wsmgmap import ops
    T_, N, K, A = shape
    B = T_ * N
    g = torch.Generator(device="cuda"); g.manual_seed(B * 31 + K)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)      # noqa: E731
    feats = rnd(B, K)
    fc, pp = nn.Linear(K, A).cuda(), nn.Linear(K, 1).cuda()
    progress, wp = torch.rand(B, 1, device="cuda", generator=g), rnd(B, A + 1)
    weights = torch.rand(T_, N, device="cuda", generator=g) + 0.1
    if T_ > 2:
        weights[T_ - 2:, 0] = 0.0                 # padded steps of one episode
    ce_rows, kl_rows = torch.rand(B, device="cuda", generator=g), torch.rand(B, device="cuda", generator=g)
    mask = (weights > 0).view(-1)
    if (~mask).any():
        kl_rows = kl_rows.masked_fill(~mask, float("nan"))       # the log of an underflowed attention weight on a padded row
    alphas = (0.1, 0.5, 1.0)

    def run(fused):
        x = feats.clone().requires_grad_(True)
        ce, kl = ce_rows.clone().requires_grad_(True), kl_rows.clone().requires_grad_(True)
        for m in (fc, pp):
            m.zero_grad(set_to_none=True)
        if fused:
            pred, prog, prows = ops.update_heads(x, fc, pp, progress)
            aux = ops.aux_reduce([ce, kl, prows], alphas, mask)
            loss, action = ops.dagger_loss(pred, aux, wp, weights)
        else:
            xd = x.double()
            pred = F.linear(xd, fc.weight.double(), fc.bias.double())
            prog = torch.tanh(F.linear(xd, pp.weight.double(), pp.bias.double()))
            prows = F.mse_loss(prog, progress.double(), reduction="none").mean(-1)
            aux = sum(a * torch.masked_select(l.double(), mask).mean() for a, l in zip(alphas, (ce, kl, prows)))
            logits = torch.tanh(pred).view(T_, N, -1)
            al = F.mse_loss(logits, wp[:, :A].double().view(T_, N, -1), reduction="none").sum(dim=2)
            action = ((weights.double() * al).sum(0) / weights.double().sum(0)).mean()
            loss = action + aux
        loss.backward()
        grads = [x.grad, fc.weight.grad, fc.bias.grad, pp.weight.grad, pp.bias.grad, ce.grad, kl.grad]
        return [t.detach().double().clone() for t in (pred, prog, prows, aux.reshape(1), loss.reshape(1), action.reshape(1))], [t.double().clone() for t in grads]

    v1, g1 = run(True)
    v2, g2 = run(True)
    v0, g0 = run(False)
    for a_, b_ in zip(v1 + g1, v2 + g2):
        assert torch.equal(torch.nan_to_num(a_), torch.nan_to_num(b_)), "the fused heads are not repeatable"
    for name, a_, b_ in zip(("pred", "prog", "prog_rows", "aux", "loss", "action"), v1, v0):
        assert torch.isfinite(a_).all(), name
        assert float((a_ - b_).abs().max()) <= 2e-6 * max(1.0, float(b_.abs().max())), (name, float((a_ - b_).abs().max()))
    for name, a_, b_ in zip(("dx", "dWm", "dbm", "dWp", "dbp", "dce", "dkl"), g1, g0):
        assert torch.isfinite(a_).all(), name
        assert float((a_ - b_).abs().max()) <= 2e-6 * max(float(b_.abs().max()), 1e-3), (name, float((a_ - b_).abs().max()))
    assert float(g1[6][~mask].abs().sum()) == 0.0       # nothing flows into the masked rows


# ----------------------------------------------------------------------------- the fused classifier tail
@pytest.mark.parametrize("geom", [(6, 48, 48, 100, 27), (2, 96, 96, 196, 27), (3, 4, 16, 7, 5), (1, 48, 48, 48, 32)],
                         ids=["bench_geometry", "e196", "one_patch_row", "identity_resize_32_classes"])
@pytest.mark.parametrize("with_loss", [True, False], ids=["ce", "no_ce"])
def test_cls_tail_matches_the_reference_lines_in_float64(geom, with_loss):
    """csrc/wsmg_cls_tail.hip against the reference's lines evaluated in float64 on the same bf16-valued inputs: BatchNorm2d with
    batch statistics + ReLU + Conv2d 1 x 1 (mg_map_policy.py:78-86), F.cross_entropy against `F.interpolate(gt, size)`'s nearest
    resize (policy.py:61-66; the kernel reproduces torch's float32 source-index arithmetic — labels are compared exactly through
    the loss), AvgPool2d(2) (mg_map_policy.py:93-96).  Forward within bf16 rounding of the logits (1e-2 relative), loss rows within 5e-3;
    gradients (incoming activation, 1 x 1 weight and bias) within 3 % of max|grad|, BatchNorm affine within 6 %; running
    statistics updated as nn.BatchNorm2d does; two launches give identical bits."""
    import torch.nn as nn
    import torch.nn.functional as F
    from wsmgmap import ops
    B, H, W, Hg, classes = geom
    g = torch.Generator(device="cuda"); g.manual_seed(B * 1000 + H + classes)
    y2 = (torch.randn(B, H, W, 32, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    bn = nn.BatchNorm2d(32).cuda().train()
    conv = nn.Conv2d(32, classes, 1).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(32, device="cuda", generator=g) + 0.5)
        bn.bias.copy_(torch.randn(32, device="cuda", generator=g) * 0.3)
        conv.weight.copy_(conv.weight.to(torch.bfloat16).float())       # bf16-representable weights: the product itself is then exact
    gt = torch.randint(0, classes, (B, Hg, Hg), device="cuda", generator=g).float() if with_loss else None
    dpooled = (torch.randn(B, H // 2, W // 2, 32, device="cuda", generator=g) * 0.01).to(torch.bfloat16)
    dpooled[..., classes:] = 0       # (the consumer's zero-padded weight sends no gradient into the padded classes)
    g_rows = torch.rand(B, device="cuda", generator=g) + 0.5

    def stats_of(t):
        st = torch.zeros(ops.BN_SLABS, 2, 32, device="cuda", dtype=torch.float64)
        f = t.double().reshape(-1, 32)
        st[3, 0], st[3, 1] = f.sum(0), (f * f).sum(0)
        return st

    def fused():
        for m in (bn, conv):
            m.zero_grad(set_to_none=True)
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        x = y2.clone().requires_grad_(True)
        st = stats_of(y2)
        sem, pooled, ce = ops.cls_tail(x, st, bn, conv, gt)
        assert float(st.abs().max()) == 0.0, "the statistics slabs must come back zeroed"
        loss = (pooled.float() * dpooled.float()).sum() + ((ce * g_rows).sum() if with_loss else 0.0)
        loss.backward()
        return ([sem.float(), pooled.float(), ce if with_loss else torch.zeros(1, device="cuda"), bn.running_mean.clone(), bn.running_var.clone()],
                [x.grad.float(), bn.weight.grad, bn.bias.grad, conv.weight.grad, conv.bias.grad])

    v1, g1 = fused()
    v2, g2 = fused()
    for a_, b_ in zip(v1 + g1, v2 + g2):
        assert torch.equal(a_, b_), "the fused classifier tail is not repeatable"
    # the reference lines, float64, NCHW
    x = y2.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    bnd, convd = nn.BatchNorm2d(32).cuda().double().train(), nn.Conv2d(32, classes, 1).cuda().double()
    with torch.no_grad():
        bnd.weight.copy_(bn.weight); bnd.bias.copy_(bn.bias); convd.weight.copy_(conv.weight); convd.bias.copy_(conv.bias)
    logits = convd(torch.relu(bnd(x)))
    pooled_ref = F.avg_pool2d(logits, 2)
    loss = (pooled_ref * dpooled.double().permute(0, 3, 1, 2)[:, :classes]).sum()
    ce_ref = torch.zeros(1, device="cuda", dtype=torch.float64)
    if with_loss:
        target = F.interpolate(gt.unsqueeze(1), size=(H, W)).squeeze(1).long()
        ce_ref = F.cross_entropy(logits, target, reduction="none").mean([1, 2])
        loss = loss + (ce_ref * g_rows.double()).sum()
    loss.backward()
    sem, pooled, ce, rm, rv = v1
    sc = float(logits.abs().max())
    assert float((sem[..., :classes].double() - logits.detach().permute(0, 2, 3, 1)).abs().max()) <= 1.2e-2 * sc
    assert float(sem[..., classes:].abs().max()) == 0.0 if classes < 32 else True
    assert float((pooled[..., :classes].double() - pooled_ref.detach().permute(0, 2, 3, 1)).abs().max()) <= 1.2e-2 * sc
    if with_loss:
        assert float((ce.double() - ce_ref.detach()).abs().max()) <= 5e-3 * float(ce_ref.abs().max()), (ce, ce_ref)
    assert float((rm.double() - bnd.running_mean).abs().max()) <= 1e-5 and float((rv.double() - bnd.running_var).abs().max()) <= 1e-4
    dx, dgm, dbt, dw, db = g1
    for name, a_, b_ in (("dx", dx.double(), x.grad.permute(0, 2, 3, 1)), ("dgamma", dgm.double(), bnd.weight.grad), ("dbeta", dbt.double(), bnd.bias.grad),
                         ("dW", dw.double(), convd.weight.grad), ("db", db.double(), convd.bias.grad)):
        e = float((a_ - b_).abs().max()) / max(float(b_.abs().max()), 1e-12)
        # (the BatchNorm affine gradients are sums of +- terms of bf16-rounded gradients over every pixel: cancellation leaves
        #  them the least accurate — 4.8 % of max|grad| at 2 x 96 x 96 pixels, under 2 % at the other sizes)
        assert e <= (6e-2 if name in ("dgamma", "dbeta") else 3e-2), (name, e)


def test_cls_tail_in_the_policy_matches_the_unfused_route(monkeypatch):
    """The bf16 update with the fused classifier tail against the same update with WSMG_FUSED_CLS_TAIL=0 (BatchNorm apply, 1 x 1
    convolution, cross-entropy, average pool and their six backward passes as separate launches): logits within 1e-3, loss within
    0.1 %, the whole gradient's cosine >= 0.9999 and the classifier's own tensors within 2 % of max|grad|; T=4 x N=8."""
    import test_gpu_policy as tp
    state = r2._default_state()
    p1, l1, g1 = tp._bench_like_update("bf16", 4, 8, state)
    monkeypatch.setattr(_SW, "fused_cls_tail", False)
    p0, l0, g0 = tp._bench_like_update("bf16", 4, 8, state)
    assert float((p1 - p0).abs().max()) <= 1e-3 and abs(l1 - l0) <= 1e-3 * abs(l0), (float((p1 - p0).abs().max()), l1, l0)
    assert set(g1) == set(g0)
    a = torch.cat([g1[n].flatten() for n in g0]).double()
    b = torch.cat([g0[n].flatten() for n in g0]).double()
    cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
    assert cos >= 0.9999, cos
    for n in g0:
        if "map_classfier" in n and n not in NULL_GRAD:
            e = float((g1[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-12)
            assert e <= 2e-2, (n, e)


# ----------------------------------------------------------------------------- configs[4] on the matrix cores
def test_attention_fp8_mfma_reproduces_the_reference_golden_g5f():
    """BASELINE configs[4] (cross-attention, e4m3 storage, instruction length 160, batch 64) on the fp8 matrix pipe
    (csrc/wsmg_attn_fp8_mfma.hip) against golden g5f: the reference's own `_attn` (mg_map_policy.py:173-178) run on inputs that
    are exactly e4m3 numbers, so quantisation loses nothing and the kernel must reproduce the reference up to float32 summation
    order: attention weights within 3e-5, context within 6e-5 of max|out| (the logits of this case reach 700 before the 1/16 and
    the matrix pipe truncates when it aligns products to its accumulator: measured 3.0e-5 / 5.9e-5 with one 256-deep chain)."""
    from wsmgmap import ops
    from util import golden
    g = golden("g5f_attn_fp8.npz")
    from oracle import cases
    c = cases.attn_fp8_inputs()
    q, k, v = (torch.from_numpy(c[n]).cuda() for n in ("q", "k", "v"))
    out, attn = ops.attention_fp8_shared(q, k, v, torch.from_numpy(c["lengths"]).cuda(), torch.from_numpy(c["inverse"]).cuda(), 1.0 / 16,
                                         scales=(c["q_scale"], c["k_scale"], c["v_scale"]))
    torch.cuda.synchronize()
    ea = float((attn.cpu() - torch.from_numpy(g["attn"])).abs().max())
    eo = float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) / float(np.abs(g["out"]).max())
    print(f"fp8 MFMA attention vs golden g5f: max |d attn| {ea:.2e}, max |d out| / max|out| {eo:.2e}")
    assert ea <= 3e-5 and eo <= 6e-5, (ea, eo)
    lens = c["lengths"][c["inverse"]]
    for b in (1, 9):      # a one-token instruction: weight exactly 1 on it, exactly 0 elsewhere
        assert lens[b] == 1 and float(attn[b, 0]) == 1.0 and float(attn[b, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(64, 8, 160), (512, 8, 80), (37, 5, 200), (3, 3, 33)], ids=["cfg5", "update_rows", "ragged", "tiny"])
def test_attention_fp8_mfma_vs_float64_formula_on_quantised_operands(shape):
    """The same kernel on ordinary inputs (per-tensor scales amax / 448): against the float64 evaluation of the reference formula on
    the DE-QUANTISED operands (oracle/attn_fp8_ref.py's encoder is the quantiser; the HIP quantiser is bit-exact with it:
    test_gpu_kernels.py), weights within 1e-5, context within 3e-5 of max|out|; rows of a set beyond 32 take a second tile."""
    from oracle import attn_fp8_ref as ar
    from wsmgmap import ops
    B, U, L = shape
    rng = np.random.RandomState(B + L)
    q = rng.randn(B, 256).astype(np.float32)
    k = (rng.randn(U, L, 256) * 0.7).astype(np.float32)
    v = rng.randn(U, L, 256).astype(np.float32)
    lengths = rng.randint(1, L + 1, size=U).astype(np.int64)
    inverse = rng.randint(0, U, size=B).astype(np.int64)
    sc = [float(np.abs(t).max()) / 448.0 for t in (q, k, v)]
    out, attn = ops.attention_fp8_shared(*(torch.from_numpy(t).cuda() for t in (q, k, v)), torch.from_numpy(lengths).cuda(),
                                         torch.from_numpy(inverse).cuda(), 1.0 / 16, scales=tuple(sc))
    qd, kd, vd = (ar.dequantize_e4m3(ar.quantize_e4m3(t, s), s) for t, s in zip((q, k, v), sc))
    mask = np.arange(L)[None, :] >= lengths[inverse][:, None]
    lg = (np.einsum("bc,blc->bl", qd, kd[inverse]) - 1e8 * mask) / 16
    p = np.exp(lg - lg.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    want = np.einsum("bl,blc->bc", p, vd[inverse])
    ea = float(np.abs(attn.cpu().numpy() - p).max())
    eo = float(np.abs(out.cpu().numpy() - want).max()) / float(np.abs(want).max())
    assert ea <= 1e-5 and eo <= 3e-5, (ea, eo)
    # scales computed on the device (amax / 448 in float32: the last bit of the scale may differ from the host's, a code here and there with it)
    out2, attn2 = ops.attention_fp8_shared(*(torch.from_numpy(t).cuda() for t in (q, k, v)), torch.from_numpy(lengths).cuda(),
                                           torch.from_numpy(inverse).cuda(), 1.0 / 16)
    assert float((out2 - out).abs().max()) <= 1e-2 * float(out.abs().max()) and float((attn2 - attn).abs().max()) <= 1e-2


@pytest.mark.parametrize("shape", [(64, 8, 160), (4096, 64, 160), (37, 5, 200), (1, 1, 1)], ids=["cfg5", "b4096", "ragged", "one"])
def test_attention_fp8_prep_scales_codes_and_grouping(shape):
    """wsmg_attn_fp8_prep (the operands of the matrix-core attention in two launches) against the stock ops it replaces: scales
    bit-equal to `(x.abs().amax() / 448).clamp_min(1e-30)` in float32, codes bit-equal to wsmg_quantize_e4m3_dev's with those
    scales (which test_gpu_kernels.py pins to the oracle's encoder), `set_start` = the exclusive prefix of bincount(inverse),
    `row_ids` a permutation whose group u holds exactly the rows of set u; caller-supplied scales pass through unchanged."""
    from wsmgmap import ops, _abi
    B, U, L = shape
    C = 256
    g = torch.Generator(device="cuda"); g.manual_seed(B + U + L)
    q = torch.randn(B, C, device="cuda", generator=g) * 3
    k = torch.randn(U, L, C, device="cuda", generator=g) * 0.7
    v = torch.randn(U, L, C, device="cuda", generator=g)
    inverse = torch.randint(0, U, (B,), device="cuda", generator=g)
    for fixed in ((0.0, 0.0, 0.0), (0.0, 0.0125, 0.0)):
        qc = torch.empty(B, C, device="cuda", dtype=torch.uint8); kc = torch.empty(U, L, C, device="cuda", dtype=torch.uint8)
        vc = torch.empty_like(kc); sc = torch.empty(3, device="cuda"); order = torch.empty(B, device="cuda", dtype=torch.int32)
        start = torch.empty(U + 1, device="cuda", dtype=torch.int32); ws = torch.zeros(4, device="cuda", dtype=torch.int32)
        P = ops._p
        _abi.call("wsmg_attn_fp8_prep", P(q), P(k), P(v), P(inverse), B, U, L, C, *fixed, P(qc), P(kc), P(vc), P(sc), P(order), P(start), P(ws),
                  ops._stream())
        for i, (x, codes) in enumerate(((q, qc), (k, kc), (v, vc))):
            want_s = (x.abs().amax() / 448.0).clamp_min(1e-30).reshape(1).float() if fixed[i] == 0.0 else torch.full((1,), fixed[i], device="cuda")
            assert torch.equal(sc[i:i + 1], want_s), (i, float(sc[i]), float(want_s))
            want_c = torch.empty_like(codes)
            _abi.call("wsmg_quantize_e4m3_dev", P(x), x.numel(), P(want_s), P(want_c), ops._stream())
            assert torch.equal(codes, want_c), f"tensor {i}: {int((codes != want_c).sum())} codes differ"
        cnt = torch.bincount(inverse, minlength=U)
        want_start = torch.zeros(U + 1, device="cuda", dtype=torch.int64); want_start[1:] = torch.cumsum(cnt, 0)
        assert torch.equal(start.long(), want_start)
        o = order.long()
        assert torch.equal(torch.sort(o).values, torch.arange(B, device="cuda"))
        group_of_slot = torch.repeat_interleave(torch.arange(U, device="cuda"), cnt)
        assert torch.equal(inverse[o], group_of_slot)
    # a NaN in an operand reaches its scale (torch's amax propagates it), not a silent saturation
    q2 = q.clone(); q2[0, 3] = float("nan")
    ws = torch.zeros(4, device="cuda", dtype=torch.int32)
    _abi.call("wsmg_attn_fp8_prep", P(q2), P(k), P(v), P(inverse), B, U, L, C, 0.0, 0.0, 0.0, P(qc), P(kc), P(vc), P(sc), P(order), P(start), P(ws),
              ops._stream())
    assert bool(torch.isnan(sc[0])) and not bool(torch.isnan(sc[1:]).any())


@pytest.mark.parametrize("shape", [(512, 512, 49), (7, 1), (300, 160), (3, 5, 33)], ids=["rgb_feature", "n1", "n160", "ragged"])
def test_mean_last_matches_torch(shape):
    """ops.mean_last (wsmg_mean_rows): AdaptiveAvgPool1d(1) + Flatten in front of rgb_linear (mg_map_policy.py:90-96) — against
    torch's float64 mean within 1e-6 of max|x|."""
    from wsmgmap import ops
    g = torch.Generator(device="cuda"); g.manual_seed(sum(shape))
    x = torch.randn(*shape, device="cuda", generator=g)
    got = ops.mean_last(x)
    want = x.double().mean(-1)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) <= 1e-6 * max(1.0, float(x.abs().max()))


# ----------------------------------------------------------------------------- BEV: scatter + rotation in one launch
@pytest.mark.parametrize("B,E,C,G", [(3, 100, 64, 240), (2, 200, 40, 480), (2, 33, 8, 64)], ids=["e100_c64", "e200_c40", "e33_c8"])
def test_bev_scatter_rotate_and_plane_fuse_equal_the_separate_launches(B, E, C, G):
    """wsmg_bev_scatter_rotate + wsmg_map_fuse_planes + wsmg_map_retrieve_fused (round 3: what Mapping.project_feat_to_map runs)
    against the launches they replace (wsmg_bev_scatter_max, wsmg_bev_rotate, wsmg_map_fuse, wsmg_map_retrieve;
    rgb_mapping.py:34-70,81-84,206-250), which the oracle tests pin: the rotated map, the global map and the retrieved map bit for bit, over several steps so the max-fuse meets a non-empty map,
    with agents near the border of the global map (window partly outside), a mid-sequence episode reset and every heading sign."""
    from wsmgmap import ops
    Hf = 64
    g = torch.Generator(device="cuda").manual_seed(5)
    gm_a = torch.zeros(B, G, G, C, device="cuda")
    gm_b = torch.zeros(B, G, G, C, device="cuda")
    half = G * 0.12 / 2
    for step in range(4):
        depth = torch.rand(B, 256, 256, device="cuda", generator=g) + 0.05
        depth[:, :8] = 0
        feat = torch.relu(torch.randn(B, 64, Hf, Hf, device="cuda", generator=g))
        lin = ops.bev_index(depth, Hf, Hf, E)
        compass = (torch.rand(B, device="cuda", generator=g) * 6.28 - 3.14) * (1.0 if step != 2 else 0.0)
        gps = (torch.rand(B, 2, device="cuda", generator=g) * 2 - 1) * half * (1.0 if step % 2 else 0.35)   # odd steps reach the border
        masks = torch.ones(B, device="cuda")
        if step == 2:
            masks[0] = 0
        planes = ops.bev_scatter_max(feat, lin, C, E)
        rot = ops.bev_rotate(planes, compass, -1.0)
        rotp = ops.bev_scatter_rotate(feat, lin, compass, -1.0, C, E)
        assert torch.equal(rotp.permute(0, 2, 3, 1), rot), f"step {step}: rotated map differs"
        ops.map_fuse(rot, gm_a, gps, masks, 0.12)
        ops.map_fuse(rotp, gm_b, gps, masks, 0.12, planes=True)
        assert torch.equal(gm_a, gm_b), f"step {step}: global map differs in {int((gm_a != gm_b).sum())} elements"
        one = ops.map_retrieve(gm_a, gps, compass, E, 0.12, fused=True)       # crop + rotation in one launch (registers)
        two = ops.map_retrieve(gm_a, gps, compass, E, 0.12, fused=False)
        assert torch.equal(one, two), f"step {step}: retrieved map differs in {int((one != two).sum())} elements"
        lds = ops.map_retrieve(gm_a, gps, compass, E, 0.12, fused="tiled")    # round 5: one launch through LDS (the default)
        assert torch.equal(lds, two), f"step {step}: LDS-tiled retrieved map differs in {int((lds != two).sum())} elements"
        assert torch.equal(ops.map_retrieve(gm_a, gps, compass, E, 0.12), two)
    assert float(gm_a.abs().max()) > 0


def test_map_sequence_vs_oracle_through_the_mapping_module():
    """The G2 sequence (rotate / paste / translate / max-fuse / retrieve over 4 steps with an episode reset) through
    `Mapping.project_feat_to_map` itself — the one-launch scatter + rotation and the plane-consuming fuse — against the oracle and
    the golden patch, at the tolerance of test_map_sequence_vs_oracle (2e-4 absolute: bilinear weights from float32 grid maths)."""
    from oracle import bev_ref, cases
    from util import T, golden
    from test_gpu_kernels import dev, close
    from wsmgmap import ops
    from wsmgmap.common.rgb_mapping import Mapping
    g = golden("g2_mapseq.npz")
    m = cases.MAP_SEQ
    ref = bev_ref.MapperRef(m["B"])
    mp = Mapping.__new__(Mapping)
    torch.nn.Module.__init__(mp)
    mp.egocentric_map_size, mp.global_map_depth, mp.global_map_size, mp.resolution = m["E"], m["C"], m["G"], 0.12
    gm = torch.zeros(m["B"], m["G"], m["G"], m["C"], device="cuda")
    assert ops.bev_planes_ok(m["C"], m["E"])
    for s in range(m["steps"]):
        c = cases.mapseq_inputs(s)
        ego_ref = ref.step(T(c["feat"]), T(c["depth"]), T(c["gps"]), T(c["compass"]), T(c["masks"]))
        obs = {"depth": dev(c["depth"]), "gps": dev(c["gps"]), "compass": dev(c["compass"])}
        ego, gm = mp.project_feat_to_map(dev(c["feat"]), gm, obs, dev(c["masks"]))
        close(f"s{s}.global", gm, ref.full_global_map, 0, 2e-4)
        close(f"s{s}.ego", ego, ego_ref, 0, 2e-4)
        np.testing.assert_allclose(ego[:, ::16, 40:56, 44:60].cpu().numpy(), g[f"s{s}.ego_patch"], atol=2e-4, rtol=0)


@pytest.mark.parametrize("C,E", [(32, 100), (8, 100), (13, 64), (40, 200), (64, 100)], ids=["c32_win2", "c8_win8", "c13_win5_6", "c40_win2_3", "c64_win1"])
def test_bev_scatter_window_widths_vs_oracle(C, E):
    """The scatter loop fetches the channels of an adaptive-max-pool window together, `WU` at a time (1 - 4, chosen from the widest
    window of the geometry; wider windows finish one channel per trip): every width against the oracle's channel pool + scatter-max
    (rgb_mapping.py:81-84,206-232), bit for bit; the fused scatter + rotation launch (same loop, then a rotation out of LDS) is held
    to the separate scatter and rotation launches on the same inputs."""
    from oracle import bev_ref, cases
    from oracle import detfill as df
    from util import T
    from wsmgmap import ops
    c = cases.bev_inputs("e200_c40_f256")
    B = c["B"]
    feat64 = df.uniform(f"r3.scatterw.{C}.{E}", (B, 64, 256, 256), 4.0)
    pooled = bev_ref.channel_maxpool(T(feat64), C).numpy()
    proj_ref, *_ = bev_ref.project_to_ground(pooled, c["depth"], E)
    depth = torch.from_numpy(c["depth"][..., 0]).cuda()
    lin = ops.bev_index(depth, 256, 256, E)
    feat = torch.from_numpy(feat64).cuda()
    proj = ops.bev_scatter_max(feat, lin, C, E)
    assert np.array_equal(proj.cpu().numpy().view(np.uint32), proj_ref.view(np.uint32)), "scatter-max differs from the oracle"
    heading = torch.tensor(([0.3, -1.1, 2.0, 0.0] * B)[:B], device="cuda")
    rotp = ops.bev_scatter_rotate(feat, lin, heading, -1.0, C, E)      # (any C: only the plane-consuming fuse needs C % 4 == 0)
    assert torch.equal(rotp.permute(0, 2, 3, 1), ops.bev_rotate(proj, heading, -1.0))


def test_fanout3_sums_three_gradients_in_one_pass():
    """ops.fanout3 (the encoded map's three consumers, mg_map_policy.py:78-100 / map_encoder.py:94-112): three aliases forward, and
    backward the float32 sum of the three bf16 gradients rounded once (wsmg_add3_bf16) — within one bf16 rounding of the float64
    sum, at least as close as autograd's two successive bf16 adds; missing gradients and other dtypes take the stock adds."""
    from wsmgmap import ops
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    x = torch.randn(7, 24, 24, 256, device="cuda", generator=g).bfloat16().requires_grad_(True)
    a, b, c = ops.fanout3(x)
    assert a.data_ptr() == x.data_ptr() and b.data_ptr() == x.data_ptr() and c.data_ptr() == x.data_ptr()
    gs = [torch.randn(x.shape, device="cuda", generator=g).bfloat16() for _ in range(3)]
    torch.autograd.backward([a, b, c], gs)
    want = gs[0].double() + gs[1].double() + gs[2].double()
    err = (x.grad.double() - want).abs()
    assert float((err - 2.0 ** -8 * want.abs()).max()) <= 1e-6
    two_adds = ((gs[0] + gs[1]) + gs[2]).double()
    assert float(err.mean()) <= float((two_adds - want).abs().mean())
    # two consumers only / float32: the stock path
    y = torch.randn(5, 8, device="cuda", generator=g, requires_grad=True)
    p, q, r = ops.fanout3(y)
    (p.sum() + 2 * q.sum()).backward()
    assert torch.equal(y.grad, torch.full_like(y, 3.0))
