"""GPU tests, round 6 (VERDICT r05 "Next round" + ADVICE r05): the in-process recurrent-core fallback of bench.py, the gradient
exchange's bucket hold beside the chained core, the fused fp8 attention's CU-derived barrier bound and reported timeout, the
concatenation written in place with its ReLU masks in the gradient kernel, the one-launch fp8 row attention, the compacted BEV sources."""
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


# ----------------------------------------------------------------------------- the concatenation in front of map_cated_linear, in place
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
    outs = [r4._one_update(pol, obs, prev, masks, weights, N, 4) for _ in range(2)]
    ops.check_rnn_status()
    stats = {k: v.detach().clone() for k, v in pol.state_dict().items() if "running_" in k}
    return outs, stats


@pytest.mark.parametrize("T,N", [(64, 8), (4, 2)])
def test_concatenation_written_in_place_and_masks_in_the_gradient_kernel_change_no_bit(monkeypatch, T, N):
    """VERDICT r05 item 1(c) (mg_map_policy.py:89-100,197,207 of the reference: cat of two Conv2d + ReLU outputs, then Conv2d + ReLU).
    map_encoded_linear / map_classified_linear writing straight into their channel slices of the tensor map_cated_linear reads
    (no concatenation pass), and map_cated_linear's backward-data kernel applying their fused ReLUs' masks and storing the gradient as
    its two contiguous parts (no mask passes): logits, loss, BatchNorm statistics and EVERY gradient bit-identical to the route with
    the three passes, on the bench workload (window kernels, mixed tiles) and a small one (implicit-GEMM kernels); and each switch
    alone."""
    new, st_new = _update_grads(monkeypatch, T, N)
    for sw_off in (dict(relu_producer_mask=False, conv_into_cat=False), dict(relu_producer_mask=False), dict(conv_into_cat=False)):
        old, st_old = _update_grads(monkeypatch, T, N, **{"relu_producer_mask": True, "conv_into_cat": True, **sw_off})
        for k in st_new:
            assert torch.equal(st_new[k], st_old[k]), k
        for a, b in zip(new, old):
            assert torch.equal(a[0], b[0]) and a[1] == b[1]
            for k, g in a[4].items():
                assert (g is None) == (b[4][k] is None) and (g is None or torch.equal(g, b[4][k])), (sw_off, k)


def test_update_has_no_concatenation_or_mask_pass_around_map_cated_linear(monkeypatch):
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
    for gone in ("wsmg_cat_channels", "wsmg_relu_bwd_rows_bf16", "wsmg_relu_bwd_bf16"):
        assert counts.get(gone, 0) == 0, (gone, counts)
    assert counts.get("wsmg_conv2d_bwd_data_bf16_ex", 0) == 1 and counts.get("wsmg_conv2d_fwd_bf16_ex", 0) == 2, counts
    assert counts.get("wsmg_colsum_multi", 0) == 2, counts     # the recurrent core's seven bias gradients; the bi-LSTM's two


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


def test_colsum_multi_matches_float64():
    """The recurrent core's seven bias gradients in one launch (mg_map_policy.py:118-132,147-152: Linear / GRU biases) against
    float64 column sums: 1e-6 of the largest sum; more than 16 tensors, ragged widths; bit-identical twice."""
    from wsmgmap import ops
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    mats = [torch.randn(r, c, device="cuda", generator=g) for r, c in
            [(512, 1536), (512, 1536), (512, 512), (512, 256), (37, 256), (512, 1), (3, 70), (1, 64)] + [(64, 8 * k + 1) for k in range(1, 12)]]
    a = ops.colsum_multi(mats)
    b = ops.colsum_multi(mats)
    for m, x, y in zip(mats, a, b):
        want = m.double().sum(0)
        assert float((x.double() - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())) * max(1, m.shape[0] // 64)
        assert torch.equal(x, y)


# ----------------------------------------------------------------------------- COMPUTE_DTYPE = "bf16+f32grad"
def test_bf16_f32grad_mode_changes_only_the_three_first_of_chain_weight_gradients(monkeypatch):
    """VERDICT r05 item 8 (the reference trains in float32 only: dagger_trainer.py:505-541).  COMPUTE_DTYPE = "bf16+f32grad": the weight
    gradients of map_encoder.cnn.0, map_decoder.base_model.conv1 and conv_original_size0 from a 16-mantissa-bit dY (hi + lo bf16 pair,
    two launches of the bf16 weight-gradient kernel).  On the bench workload: logits, loss and every OTHER gradient bit-identical to
    the bf16 mode; the three tensors at least as close to the float32 mode's gradient as bf16's are (relative L2), and changed."""
    import bench
    import test_gpu_round2 as r2
    import test_gpu_round4 as r4
    T, N = 64, 8
    obs, prev, masks, weights = bench.synth_batch(T, N, torch.device("cuda"), 91)
    out = {}
    for mode in ("f32", "bf16", "bf16+f32grad"):
        torch.manual_seed(3)
        pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype=mode, state=r2._default_state()))
        out[mode] = r4._one_update(pol, obs, prev, masks, weights, N, 4)
        del pol
        torch.cuda.empty_cache()
    three = ("net.map_encoder.cnn.0.weight", "net.map_decoder.base_model.conv1.weight", "net.map_decoder.conv_original_size0.0.weight")
    a, b, ref = out["bf16+f32grad"], out["bf16"], out["f32"]
    assert torch.equal(a[0], b[0]) and a[1] == b[1]
    for k, g in a[4].items():
        if g is None:
            continue
        if k in three:
            r = ref[4][k].double()
            ea = float((g.double() - r).norm() / r.norm())
            eb = float((b[4][k].double() - r).norm() / r.norm())
            assert not torch.equal(g, b[4][k]) and ea <= eb * 1.02, (k, ea, eb)
        else:
            assert torch.equal(g, b[4][k]), k


# ----------------------------------------------------------------------------- 3 x 3 over a 32-channel reduction axis
@pytest.mark.parametrize("B,H,Cin,Cout", [(64, 48, 32, 32), (70, 48, 32, 64), (57, 48, 32, 32), (60, 47, 32, 32), (256, 24, 32, 64), (230, 25, 32, 64), (256, 24, 32, 128)])
def test_k32_window_convolution_is_bit_identical_to_the_general_window_kernel(B, H, Cin, Cout):
    """VERDICT r05 item 2 (the 48 x 48 / 32-channel shapes: the semantic classifier's 3 x 3 layers, mg_map_policy.py:78-86).
    wsmg_conv_win3_k32.hip — weights resident in LDS, one barrier per 256-pixel tile, three workgroups per CU — runs the same MFMA
    sequence per output element as the general window kernel (itself held against the float64 oracle by test_conv2d_fwd_bwd and the
    G3 / full-size policy tests): forward (bias + ReLU, plain, into a channel slice of a wider tensor) and backward-data outputs bit
    for bit, pixel counts that are not a multiple of the tile included, several / uneven / single tiles per workgroup; the BatchNorm sums
    (other partial-sum grouping) to 1e-6."""
    from wsmgmap import _abi, ops
    g = torch.Generator(device="cuda").manual_seed(B + H + Cout)
    x = torch.relu(torch.randn(B, H, H, Cin, device="cuda", generator=g)).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * 0.08).to(torch.bfloat16)
    bias = torch.randn(Cout, device="cuda", generator=g) * 0.1
    dy = torch.randn(B, H, H, Cout, device="cuda", generator=g).to(torch.bfloat16)
    wi = w.permute(3, 1, 2, 0).contiguous()
    args = (B, H, H, Cin, Cout, 3, 3, 1, 1, H, H)
    P, st = ops._p, ops._stream
    nslab = 8
    outs = {}
    for mt in (256, 1):       # 256: the general window kernel's 256-pixel tile, forced; 1: by shape = the k32 kernel
        old = _abi.lib().wsmg_conv_debug_win3_tile(mt)
        try:
            y0 = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
            y1 = torch.empty_like(y0)
            wide = torch.full((B, H, H, Cout + 64), 7.0, device="cuda", dtype=torch.bfloat16)
            dx = torch.empty_like(x)
            stats = torch.zeros(nslab, 2, Cout, device="cuda", dtype=torch.float64)
            _abi.call("wsmg_conv2d_fwd_bf16", P(x), P(w), P(bias), P(y0), 2, *args, st())
            _abi.call("wsmg_conv2d_fwd_bf16_stats", P(x), P(w), None, P(y1), 0, P(stats), nslab, *args, st())
            _abi.call("wsmg_conv2d_fwd_bf16_ex", P(x), P(w), None, wide.data_ptr() + 32 * 2, 0, None, 0, Cout + 64, *args, st())
            _abi.call("wsmg_conv2d_bwd_data_bf16", P(dy), P(wi), P(dx), 0, *args, st())
            torch.cuda.synchronize()
            outs[mt] = (y0, y1, wide, dx, stats.sum(0))
        finally:
            _abi.lib().wsmg_conv_debug_win3_tile(old)
    a, b = outs[1], outs[256]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2], b[2]) and torch.equal(a[2][..., 32:32 + Cout], a[1])
    assert bool((a[2][..., :32] == 7).all()) and bool((a[2][..., 32 + Cout:] == 7).all())
    assert torch.equal(a[3], b[3])
    want = torch.stack([a[1].double().sum((0, 1, 2)), (a[1].double() ** 2).sum((0, 1, 2))])
    assert float((a[4] - want).abs().max() / want.abs().max()) < 1e-6
    assert float((b[4] - want).abs().max() / want.abs().max()) < 1e-6
    # and against plain float64 arithmetic on the bf16 operands (the 32 -> 32 case: small enough for a dense reference here)
    if Cout == 32 and H == 47:
        ref = torch.nn.functional.conv2d(x[:4].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        assert float((a[1][:4].double() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
        refdx = torch.nn.functional.conv_transpose2d(dy[:4].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        # (rows of image 3 next to image 4 are complete: the gradient of a pixel depends on its own image only)
        assert float((a[3][:4].double() - refdx).abs().max()) <= 2 ** -7 * float(refdx.abs().max())


def test_feeder_layout_input_gives_the_same_update_bit_for_bit():
    """bench.py's secondary `feeder_layout_mode` leg hands `rgb_ego_map` over as wsmgmap's feeder does (channels-last bf16 storage behind
    the reference's [B,C,E,E] shape: DeviceCollator(ego_map_nhwc_bf16=True), dagger_trainer.py:336-343 stores the map in float16): the
    update skips its layout pass and is otherwise the same computation — logits, loss and every gradient bit for bit."""
    import bench
    import test_gpu_round2 as r2
    import test_gpu_round4 as r4
    T, N = 16, 4
    obs, prev, masks, weights = bench.synth_batch(T, N, torch.device("cuda"), 17)
    obs["rgb_ego_map"] = obs["rgb_ego_map"].to(torch.bfloat16).float()      # values a bf16 cache can hold
    out = []
    for feeder in (False, True):
        o = dict(obs)
        if feeder:
            o["rgb_ego_map"] = o["rgb_ego_map"].to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        torch.manual_seed(5)
        pol = r2._train_mode(r2._policy(num_proc=1, compute_dtype="bf16", state=r2._default_state()))
        out.append(r4._one_update(pol, o, prev, masks, weights, N, 4))
        del pol
    a, b = out
    assert torch.equal(a[0], b[0]) and a[1] == b[1]
    for k, g in a[4].items():
        assert (g is None and b[4][k] is None) or torch.equal(g, b[4][k]), k


# ----------------------------------------------------------------------------- ConvTranspose2d(64 -> 32, k4, s2, p1) forward
@pytest.mark.parametrize("B,H", [(256, 24), (229, 24), (300, 21)])
def test_convtranspose_k4s2_kernel_is_bit_identical_to_the_implicit_gemm_route(B, H):
    """VERDICT r05 item 2 (the ConvTranspose of the semantic classifier, mg_map_policy.py:79).  wsmg_convt_k4s2.hip — both column
    parities of a row parity per workgroup, weights resident in LDS, full-line stores — runs the implicit-GEMM kernel's MFMA sequence
    per output element: outputs bit for bit (pixel counts that are not a multiple of the tile and a geometry whose tiles cross images
    differently included), BatchNorm sums to 1e-6, and within bf16 rounding of torch's float64 conv_transpose2d on the same operands."""
    from wsmgmap import _abi, ops
    g = torch.Generator(device="cuda").manual_seed(B + H)
    x = torch.relu(torch.randn(B, H, H, 64, device="cuda", generator=g)).to(torch.bfloat16)          # the ConvTranspose's input
    w_t = (torch.randn(64, 32, 4, 4, device="cuda", generator=g) * 0.06).to(torch.bfloat16)          # nn.ConvTranspose2d weight [Cin_t, Cout_t, 4, 4]
    w_ihwo = w_t.permute(1, 2, 3, 0).contiguous()                                                    # adjoint convolution's IHWO: [32][4][4][64]
    dims = (B, 2 * H, 2 * H, 32, 64, 4, 4, 2, 1, H, H)
    P, st = ops._p, ops._stream
    nslab = 8
    outs = {}
    for mt in (256, 1):       # 256: a forced window tile turns the by-shape choice off -> implicit-GEMM kernel; 1: by shape
        old = _abi.lib().wsmg_conv_debug_win3_tile(mt)
        try:
            y0 = torch.empty(B, 2 * H, 2 * H, 32, device="cuda", dtype=torch.bfloat16)
            y1 = torch.empty_like(y0)
            stats = torch.zeros(nslab, 2, 32, device="cuda", dtype=torch.float64)
            _abi.call("wsmg_conv2d_bwd_data_bf16", P(x), P(w_ihwo), P(y0), 0, *dims, st())
            _abi.call("wsmg_conv2d_bwd_data_bf16_stats", P(x), P(w_ihwo), P(y1), 0, P(stats), nslab, *dims, st())
            torch.cuda.synchronize()
            outs[mt] = (y0, y1, stats.sum(0))
        finally:
            _abi.lib().wsmg_conv_debug_win3_tile(old)
    a, b = outs[1], outs[256]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[0], a[1])
    want = torch.stack([a[1].double().sum((0, 1, 2)), (a[1].double() ** 2).sum((0, 1, 2))])
    assert float((a[2] - want).abs().max() / want.abs().max()) < 1e-6
    assert float((b[2] - want).abs().max() / want.abs().max()) < 1e-6
    ref = torch.nn.functional.conv_transpose2d(x[:3].double().permute(0, 3, 1, 2), w_t.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert float((a[0][:3].double() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
    ref = torch.nn.functional.conv_transpose2d(x[-2:].double().permute(0, 3, 1, 2), w_t.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    assert float((a[0][-2:].double() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())


@pytest.mark.parametrize("B,H,W,C", [(2, 2, 2, 8), (3, 3, 5, 16), (2, 5, 4, 64), (4, 24, 24, 64), (2, 12, 12, 128)])
def test_upsample_backward_with_eight_channels_per_thread(B, H, W, C):
    """The bilinear x 2 (align_corners) upsample's backward for bf16 (map_encoder.py:94-110's `nn.Upsample`), 8 channels per thread with
    the row / column weights worked out once per thread: against torch's float64 autograd on the same bf16 operands to the output's own
    bf16 rounding, at sizes where 4, 5 and (H = 2) 6 output rows read one input row."""
    import torch.nn.functional as F
    from wsmgmap import ops
    g = torch.Generator(device="cuda").manual_seed(H * 100 + W * 10 + C)
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    gy = torch.randn(B, 2 * H, 2 * W, C, device="cuda", generator=g).to(torch.bfloat16)
    y = ops.upsample2x(x)
    y.backward(gy)
    xr = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    want = xr.grad.permute(0, 2, 3, 1)
    err = (x.grad.double() - want).abs()
    assert bool((err <= 2 ** -8 * want.abs() + 1e-6 * float(want.abs().max())).all()), float(err.max())
