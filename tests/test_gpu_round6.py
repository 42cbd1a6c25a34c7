"""GPU tests, round 6 (VERDICT r05 "Next round" + ADVICE r05): the in-process recurrent-core fallback of bench.py, the gradient
exchange's bucket hold beside the chained core, the fused fp8 attention's CU-derived barrier bound and reported timeout, the
BatchNorm sums / finalize folded into their neighbours, the new window-kernel shapes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _aux_losses_off():
    from wsmgmap.common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    AuxLosses.clear()
    yield
    AuxLosses.deactivate()
    AuxLosses.clear()


def _bench_line(env_extra, *flags, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-f32",
                        "--no-cpu-baseline", "--no-other-configs", "--prewarm-s", "0", *flags],
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


# ----------------------------------------------------------------------------- bench.py cannot die on a persistent-kernel timeout
@pytest.mark.parametrize("dp", [False, True])
def test_bench_falls_back_in_process_on_an_injected_timeout(dp):
    """VERDICT r05 item 4.  A persistent-kernel timeout bit (injected through wsmg_rnn_debug_inject after the 2nd update, as a kernel
    whose spin ran out would set it) must yield a VALID line, with the fallback named, from the same process: single process, and
    under a one-rank RCCL process group, where the bit travels through GradAllReducer.finish()'s cross-rank error flag
    (reference: the DDP-wrapped update of common_trainer.py:35-38,61-66, dagger_trainer.py:505-543)."""
    env = {"WSMG_BENCH_INJECT_TIMEOUT": "2"}
    if dp:
        env.update(WSMG_BENCH_DP_ONE_RANK="1", MASTER_PORT="29533")
    line, err = _bench_line(env)
    assert line["value"] > 0 and line["steps"] == 4 and np.isfinite(line["loss"])
    rc = line["recurrent_core"]
    assert rc["fallback_level"] == 1 and rc["recurrent_core"].startswith("staged (fallback"), rc
    assert "timed out" in rc["reasons"][0] or "reported an error" in rc["reasons"][0] or "status word" in rc["reasons"][0], rc
    if dp:
        assert line["data_parallel"]["recurrent_core"].startswith("staged (fallback"), line["data_parallel"]
        assert isinstance(line["data_parallel"].get("per_bucket_allreduce_ms"), list), line["data_parallel"]


def test_bench_line_names_the_default_core_without_a_timeout():
    line, _ = _bench_line({})
    assert line["recurrent_core"]["fallback_level"] == 0 and line["recurrent_core"]["recurrent_core"].startswith("chained"), line["recurrent_core"]


# ----------------------------------------------------------------------------- BatchNorm sums / finalize folded into their neighbours
def _update_grads(monkeypatch, T, N, seed=91, **switches):
    import bench
    import test_gpu_round2 as r2
    import test_gpu_round4 as r4
    from wsmgmap import debug, ops
    for k, v in switches.items():
        monkeypatch.setattr(debug.sw, k, v)
    torch.manual_seed(3)
    pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype="bf16", state=r2._default_state()))
    obs, prev, masks, weights = bench.synth_batch(T, N, torch.device("cuda"), seed)
    outs = [r4._one_update(pol, obs, prev, masks, weights, N, 4) for _ in range(2)]      # (two passes: both parities of the slab pairs)
    ops.check_rnn_status()
    stats = {k: v.detach().clone() for k, v in pol.state_dict().items() if "running_" in k}
    return outs, stats


@pytest.mark.parametrize("T,N", [(64, 8), (4, 2)])
def test_batchnorm_sums_in_producer_epilogues_match_the_reduction_pass(monkeypatch, T, N):
    """VERDICT r05 item 1 (map_encoder.py:19-29,94-112, mg_map_policy.py:78-100 of the reference: 13 train-mode BatchNorms).  The
    update with the BatchNorm-backward sums taken in the epilogue of the kernel that produces the gradient (backward-data convolutions
    of all tile shapes incl. the stride-2 parity classes and channel slices, the ConvTranspose's input gradient, the three-way add, the
    upsampling's backward), the two projections writing into the concatenation in place and map_cated_linear's input-gradient kernel
    applying their ReLU masks — against the round-5 route (reduction passes, concatenation and mask passes) on the bench workload and
    a small one: logits, loss, BatchNorm running statistics and every gradient.  The two routes compute the same sums from the same
    bf16 values; only the float64 summation order differs, which can flip a last float32 bit of dgamma / dbeta and with it isolated
    bf16 roundings downstream (the first layer of each backward chain, whose bf16 dy has crossed the most roundings, moves most: 1e-2 of
    its largest element, cosine 0.99997): bars 1e-5 / 0.9999999 for the BatchNorm affine gradients (they ARE the sums), 2e-2 / 0.9999 for
    everything behind them; forward results identical."""
    new, st_new = _update_grads(monkeypatch, T, N)
    old, st_old = _update_grads(monkeypatch, T, N, bn_producer_sums=False, relu_producer_mask=False, conv_into_cat=False)
    for k in st_new:
        assert torch.equal(st_new[k], st_old[k]), k
    for a, b in zip(new, old):
        assert torch.equal(a[0], b[0]) and a[1] == b[1], "the forward pass must not change"
        bad = []
        for k, g in a[4].items():
            h = b[4][k]
            assert (g is None) == (h is None), k
            if g is None:
                continue
            assert torch.isfinite(g).all(), k
            scale = float(h.abs().max())
            d = float((g - h).abs().max()) / max(scale, 1e-20)
            cos = float(torch.nn.functional.cosine_similarity(g.flatten().double(), h.flatten().double(), dim=0)) if scale > 1e-12 else 1.0
            # BatchNorm affine gradients ARE the sums: they may differ in the last float32 bits only; everything behind them in bf16 steps
            tight = k.endswith(("1.weight", "1.bias", "4.weight", "4.bias", "bn1.weight", "bn1.bias", "bn2.weight", "bn2.bias")) and g.dim() == 1
            if d > (1e-5 if tight else 2e-2) or cos < (0.9999999 if tight else 0.9999):
                bad.append((k, d, cos))
        assert not bad, sorted(bad, key=lambda t: -t[1])[:12]
    # bit-reproducible run to run (plain stores of per-workgroup partials, added in block order: no atomics anywhere)
    again, _ = _update_grads(monkeypatch, T, N)
    for a, b in zip(new, again):
        for k, g in a[4].items():
            assert g is None or torch.equal(g, b[4][k]), k


def test_update_launch_counts_after_the_batchnorm_folding(monkeypatch):
    """Done-criterion of VERDICT r05 item 1: per update at most 3 reduction passes over a BatchNorm's gradient (the three BatchNorms whose
    output has more than one consumer or a residual); the other 12 take their sums from the producer's partials; no concatenation
    pass and no ReLU-mask pass around map_cated_linear."""
    import bench
    import test_gpu_round2 as r2
    import test_gpu_round4 as r4
    from wsmgmap import _abi
    pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype="bf16", state=r2._default_state()))
    obs, prev, masks, weights = bench.synth_batch(8, 4, torch.device("cuda"), 5)
    r4._one_update(pol, obs, prev, masks, weights, 4, 4)
    counts = {}
    real = _abi.call

    def counting(name, *a):
        counts[name] = counts.get(name, 0) + 1
        return real(name, *a)
    monkeypatch.setattr(_abi, "call", counting)
    r4._one_update(pol, obs, prev, masks, weights, 4, 4)
    monkeypatch.setattr(_abi, "call", real)
    assert counts.get("wsmg_bn_act_bwd_bf16_parts", 0) == 12, counts
    assert counts.get("wsmg_bn_act_bwd_bf16", 0) + counts.get("wsmg_bn_act_bwd_ld_bf16", 0) == 3, counts
    for gone in ("wsmg_cat_channels", "wsmg_relu_bwd_rows_bf16", "wsmg_relu_bwd_bf16", "wsmg_add3_bf16"):
        assert counts.get(gone, 0) == 0, (gone, counts)


# ----------------------------------------------------------------------------- configs[4] per row, one launch
@pytest.mark.parametrize("B,L", [(64, 160), (3, 37), (17, 224), (300, 80)])
def test_fp8_row_attention_in_one_launch(monkeypatch, B, L):
    """VERDICT r05 item 6 — configs[4] as SURVEY 8d defines it (mg_map_policy.py:173-178: every row its own token set): the query
    fold and the attention of a row in ONE launch (wsmg_attn_fp8_row_fwd) against the float64 evaluation of the reference formula
    on the de-quantised tokens (the bars of the two-launch route: weights 2e-5, context 2e-5 of max|out|) and against the
    two-launch route itself (1e-6: the fold's reduction is split in two halves instead of one fmaf chain); ragged lengths incl.
    1 and L, L up to the kernel's 224, masked tokens carry exactly zero weight; the folded query it hands the backward pass
    equals the fold kernel's to 1e-6."""
    import importlib
    import test_gpu_kernels as tk
    from oracle import attn_fp8_ref as ar
    from util import T
    from wsmgmap import debug, ops
    att = importlib.import_module("wsmgmap.ops.attention")
    q, w, b, x, lengths = tk._cfg5_inputs(B=B, L=L, seed=B + L)
    x_scale = float(np.abs(x).max() / ar.E4M3_MAX)
    codes = ar.quantize_e4m3(x, x_scale)
    out_ref, attn_ref = ar.attn_fp8(q, w, b, codes, x_scale, lengths, 1.0 / 16)
    codes_dev = ops.quantize_e4m3(T(x).cuda(), x_scale)
    args = (T(q).cuda(), T(w).cuda(), T(b).cuda(), codes_dev, x_scale, torch.from_numpy(lengths).cuda(), 1.0 / 16)
    out, attn = ops.attn_fp8_fused(*args)
    assert att.last_fp8_row_launches == 1
    monkeypatch.setattr(debug.sw, "fp8_row_fused", False)
    out2, attn2 = ops.attn_fp8_fused(*args)
    assert att.last_fp8_row_launches == 2
    o, a_ = out.cpu().numpy(), attn.cpu().numpy()
    assert np.abs(a_ - attn_ref).max() <= 2e-5 and np.abs(o - out_ref).max() <= 2e-5 * np.abs(out_ref).max()
    assert float((attn - attn2).abs().max()) <= 1e-6 and float((out - out2).abs().max()) <= 1e-6 * float(out2.abs().max())
    assert all(float(np.abs(a_[i, lengths[i]:]).sum()) == 0.0 for i in range(B))
    out3, attn3 = ops.attn_fp8_fused(*args[:4], x_scale, args[5], 1.0 / 16)
    assert torch.equal(out2, out3) and torch.equal(attn2, attn3)


# ----------------------------------------------------------------------------- BEV: compacted sources
@pytest.mark.parametrize("name", ["e100_c64_f224", "e100_c64_f256", "e200_c40_f256"])
def test_bev_compacted_sources_give_the_same_planes_bit_for_bit(name):
    """VERDICT r05 item 5 (rgb_mapping.py:153-232 of the reference: ComputeSpatialLocs + ProjectToGroundPlane + scatter_max).  The index
    launch that also packs the valid sources: the same linear index / validity as wsmg_bev_index (array equality AND the golden's SHA),
    the packed list = exactly the valid (source, cell) pairs, each once; scatter + rotation from the list bit-identical to the route
    that walks every source, at the three golden geometries incl. the 64 -> 40 channel pool, an all-invalid frame and a
    nearly-all-valid one."""
    import hashlib
    from oracle import cases
    from util import T
    from wsmgmap import ops
    g = np.load(os.path.join(ROOT, "tests", "golden", "g1_bev.npz"))
    c = cases.bev_inputs(name)
    E, C, Hf = c["E"], c["C"], c["Hf"]
    depth = T(c["depth"][..., 0]).cuda()
    lin0 = ops.bev_index(depth, Hf, Hf, E)
    lin, (cl, cnt) = ops.bev_index_compact(depth, Hf, Hf, E)
    assert torch.equal(lin, lin0)
    lh = lin.cpu().numpy()
    inv = lh < 0
    assert hashlib.sha256(np.ascontiguousarray(np.where(inv, 0, lh).astype(np.int32)).tobytes()).hexdigest() == str(g[name + ".lin_idx_sha"])
    clh, cnth = cl.cpu().numpy().view(np.uint32), cnt.cpu().numpy()
    per = Hf * Hf
    for b in range(lh.shape[0]):
        got = np.concatenate([clh[b, k * 8192:k * 8192 + cnth[b, k]] for k in range(cnth.shape[1])])
        src = np.nonzero(~inv[b])[0]
        want = (src.astype(np.uint32) << 16) | lh[b, src].astype(np.uint32)
        assert np.array_equal(np.sort(got), np.sort(want)) and len(got) == len(src)
    feat = T(c["feat"]).cuda()
    if feat.shape[1] == C and C == 40:       # the cfg4 geometry with the 64 -> 40 adaptive max-pool inside the scatter
        gen = torch.Generator(device="cuda"); gen.manual_seed(3)
        feat = torch.rand(feat.shape[0], 64, Hf, Hf, device="cuda", generator=gen) * 4 - 1
    heading = torch.tensor([0.3, -1.2][:feat.shape[0]], device="cuda")
    a = ops.bev_scatter_rotate(feat, lin, heading, -1.0, C, E)
    b_ = ops.bev_scatter_rotate(feat, lin, heading, -1.0, C, E, compact=(cl, cnt))
    assert torch.equal(a, b_)
    for dval in (0.0, 0.45):      # nothing valid; nearly everything in range
        d2 = torch.full_like(depth, dval)
        l0 = ops.bev_index(d2, Hf, Hf, E)
        l1, comp = ops.bev_index_compact(d2, Hf, Hf, E)
        assert torch.equal(l0, l1) and int(comp[1].sum()) == int((l0 >= 0).sum())
        assert torch.equal(ops.bev_scatter_rotate(feat, l0, heading, -1.0, C, E), ops.bev_scatter_rotate(feat, l1, heading, -1.0, C, E, compact=comp))
