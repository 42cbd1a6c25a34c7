"""GPU parity, kernel level: every entry point of libwsmgmap.so (called through the C ABI via
wsmgmap.ops) against the oracle / a float64 CPU evaluation of the same operator on seeded
inputs.  Bars: bit-exact for the integer BEV index and the scatter-max; float tolerances are
written at each assert."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import bev_ref, cases, policy_ref
from oracle import detfill as df
from util import T, golden, sha

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from wsmgmap import ops as o
    return o


def dev(a):
    return (a if torch.is_tensor(a) else T(a)).cuda()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(name, got, ref, rtol, atol):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    worst = float((err - bound).max())
    assert worst <= 0, f"{name}: max abs err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e}), exceeds by {worst:.3e}"


# (name, B, Cin, Cout, k, stride, pad, H) — every conv shape of the map stack
CONVS = [
    ("enc0_k8s2", 2, 64, 64, 8, 2, 3, 100),
    ("enc3_k5s2", 3, 64, 128, 5, 2, 1, 50),
    ("enc6_k3", 3, 128, 256, 3, 1, 1, 24),
    ("dec_stem_k7s2", 3, 256, 64, 7, 2, 3, 24),
    ("dec_1x1", 5, 64, 64, 1, 1, 0, 6),
    ("dec_block_k3_6", 5, 64, 64, 3, 1, 1, 6),
    ("dec_up0", 3, 128, 128, 3, 1, 1, 12),
    ("dec_orig2", 2, 192, 64, 3, 1, 1, 24),
    ("cls_k3_48", 2, 32, 32, 3, 1, 1, 48),
    ("cls_1x1_48", 2, 32, 32, 1, 1, 0, 48),
    ("cated_k3", 2, 256, 256, 3, 1, 1, 24),
]


@pytest.mark.parametrize("cfg", CONVS, ids=[c[0] for c in CONVS])
def test_conv2d_fwd_bwd(ops, cfg):
    name, B, Cin, Cout, k, s, p, H = cfg
    x = T(df.uniform(f"conv.{name}.x", (B, Cin, H, H), 2.0))
    w = T(df.uniform(f"conv.{name}.w", (Cout, Cin, k, k), float(np.sqrt(12.0 / (Cin * k * k)))))
    b = T(df.uniform(f"conv.{name}.b", (Cout,), 0.5))
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    gy = T(df.uniform(f"conv.{name}.gy", tuple(yr.shape), 2.0))
    yr.backward(gy.double())

    xg = nhwc(x).cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, s, p)
    y.backward(nhwc(gy).cuda())
    # f32 MFMA = fmaf chain over K = Cin*k*k terms of magnitude <= ~0.5: error ~ K * 2^-24 * |terms|
    close(name + ".y", nchw(y), yr, 2e-5, 2e-5)
    close(name + ".dx", nchw(xg.grad), xr.grad, 2e-5, 2e-5)
    close(name + ".dw", wg.grad, wr.grad, 2e-5, 2e-4 * float(wr.grad.abs().max()) + 1e-6)
    close(name + ".db", bg.grad, br.grad, 2e-5, 2e-4 * float(br.grad.abs().max()) + 1e-6)


def test_conv_transpose2d(ops):
    B, Ci, Co, H = 2, 64, 32, 24
    x = T(df.uniform("convt.x", (B, Ci, H, H), 2.0))
    w = T(df.uniform("convt.w", (Ci, Co, 4, 4), 0.2))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, stride=2, padding=1)
    gy = T(df.uniform("convt.gy", tuple(yr.shape), 2.0))
    yr.backward(gy.double())
    xg = nhwc(x).cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    y = ops.conv_transpose2d(xg, wg, 2, 1)
    assert tuple(y.shape) == (B, 2 * H, 2 * H, Co)
    y.backward(nhwc(gy).cuda())
    close("convt.y", nchw(y), yr, 2e-5, 2e-5)
    close("convt.dx", nchw(xg.grad), xr.grad, 2e-5, 2e-5)
    close("convt.dw", wg.grad, wr.grad, 2e-5, 2e-4 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("C,H,B,res,relu", [(64, 50, 3, False, True), (256, 24, 2, False, True), (64, 6, 5, True, True),
                                            (32, 48, 2, False, True), (128, 12, 3, False, False)])
def test_bn_act_train(ops, C, H, B, res, relu):
    x = T(df.uniform(f"bn.x.{C}.{H}", (B, C, H, H), 3.0)) + T(df.uniform(f"bn.off.{C}", (1, C, 1, 1), 2.0))
    r = T(df.uniform(f"bn.r.{C}.{H}", (B, C, H, H), 2.0)) if res else None
    g = T(df.positive(f"bn.g.{C}", (C,)))
    bt = T(df.uniform(f"bn.b.{C}", (C,), 0.5))
    rm, rv = T(df.uniform(f"bn.rm.{C}", (C,), 0.2)), T(df.positive(f"bn.rv.{C}", (C,)))
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), bt.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, training=True, momentum=0.1, eps=1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    gy = T(df.uniform(f"bn.gy.{C}.{H}", tuple(yr.shape), 2.0))
    yr.backward(gy.double())

    xg = nhwc(x).cuda().requires_grad_(True)
    gg, bg = g.cuda().requires_grad_(True), bt.cuda().requires_grad_(True)
    rg = nhwc(r).cuda().requires_grad_(True) if res else None
    rmg, rvg = rm.cuda(), rv.cuda()
    y = ops.bn_act(xg, gg, bg, rmg, rvg, True, relu, rg, 0.1, 1e-5)
    y.backward(nhwc(gy).cuda())
    close("bn.y", nchw(y), yr, 1e-5, 1e-5)
    close("bn.running_mean", rmg, rmr, 1e-6, 1e-6)
    close("bn.running_var", rvg, rvr, 1e-5, 1e-6)
    close("bn.dx", nchw(xg.grad), xr.grad, 1e-4, 2e-5)
    close("bn.dgamma", gg.grad, gr.grad, 1e-4, 1e-4 * float(gr.grad.abs().max()))
    close("bn.dbeta", bg.grad, br.grad, 1e-4, 1e-4 * float(br.grad.abs().max()))
    if res:
        close("bn.dres", nchw(rg.grad), rr.grad, 0, 0)


def test_bn_eval(ops):
    C, H, B = 64, 12, 2
    x = T(df.uniform("bne.x", (B, C, H, H), 3.0))
    g, bt = T(df.positive("bne.g", (C,))), T(df.uniform("bne.b", (C,), 0.5))
    rm, rv = T(df.uniform("bne.rm", (C,), 0.2)), T(df.positive("bne.rv", (C,)))
    yr = F.relu(F.batch_norm(x.double(), rm.double(), rv.double(), g.double(), bt.double(), training=False, eps=1e-5))
    rmg, rvg = rm.cuda(), rv.cuda()
    y = ops.bn_act(nhwc(x).cuda(), g.cuda(), bt.cuda(), rmg, rvg, False, True, None, 0.1, 1e-5)
    close("bn_eval.y", nchw(y), yr, 1e-5, 1e-5)
    assert torch.equal(rmg.cpu(), rm) and torch.equal(rvg.cpu(), rv)


def test_small_nhwc_ops(ops):
    B, C, H = 3, 64, 12
    # post-ReLU-like input with many exact zeros: exercises max-pool tie breaking
    x = torch.relu(T(df.uniform("pool.x", (B, C, H, H), 2.0)))
    for nm, fn_ref, fn in [
        ("maxpool", lambda t: F.max_pool2d(t, 3, 2, 1), ops.maxpool3x3s2),
        ("upsample", lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True), ops.upsample2x),
        ("avgpool", lambda t: F.avg_pool2d(t, 2, 2), ops.avgpool2),
        ("relu", lambda t: F.relu(t - 0.3), lambda t: ops.relu(t - 0.3)),
    ]:
        xr = x.double().requires_grad_(True)
        yr = fn_ref(xr)
        gy = T(df.uniform(f"pool.gy.{nm}", tuple(yr.shape), 2.0))
        yr.backward(gy.double())
        xg = nhwc(x).cuda().requires_grad_(True)
        y = fn(xg)
        y.backward(nhwc(gy).cuda())
        close(nm + ".y", nchw(y), yr, 1e-6, 1e-6)
        close(nm + ".dx", nchw(xg.grad), xr.grad, 1e-5, 1e-6)


def test_layout_roundtrip_and_padding(ops):
    x = T(df.uniform("lay.x", (3, 27, 48, 48), 2.0)).cuda().requires_grad_(True)
    y = ops.to_nhwc(x, 32)
    assert tuple(y.shape) == (3, 48, 48, 32)
    assert torch.equal(y[..., :27], x.detach().permute(0, 2, 3, 1)) and float(y[..., 27:].abs().max()) == 0
    z = ops.to_nchw(y, 27)
    assert torch.equal(z, x.detach())
    z.backward(torch.ones_like(z))
    assert torch.equal(x.grad, torch.ones_like(x))
    e = T(df.uniform("lay.e", (2, 64, 100, 100), 2.0)).cuda()
    assert torch.equal(ops.to_nhwc(e), e.permute(0, 2, 3, 1))


@pytest.mark.parametrize("masked", [True, False])
def test_attention_fwd_bwd(ops, masked):
    g = golden("g5_attn.npz")
    q, k, v, m = cases.attn_inputs()
    qr, kr, vr = (T(a).double().requires_grad_(True) for a in (q, k, v))
    o_ref, a_ref = policy_ref.attn(qr, kr, vr, T(m) if masked else None)
    go = T(df.uniform("attn.go", tuple(o_ref.shape), 2.0))
    ga = T(df.uniform("attn.ga", tuple(a_ref.shape), 2.0))
    (o_ref * go.double()).sum().backward(retain_graph=True)
    (a_ref * ga.double()).sum().backward()

    qg = T(q).cuda().requires_grad_(True)
    kg = T(k).permute(0, 2, 1).contiguous().cuda().requires_grad_(True)   # token-major [B,I,C]
    vg = T(v).permute(0, 2, 1).contiguous().cuda().requires_grad_(True)
    o, a = ops.attention(qg, kg, vg, T(m).cuda() if masked else None, 1.0 / 16)
    ((o * go.cuda()).sum() + (a * ga.cuda()).sum()).backward()
    close("attn.out", o, o_ref, 1e-5, 1e-6)
    close("attn.attn", a, a_ref, 1e-5, 1e-7)
    np.testing.assert_allclose(o.detach().cpu().numpy(), g["out" if masked else "out_nomask"], rtol=1e-5, atol=1e-6)
    if masked:
        assert float(a[1, 1:].abs().max()) == 0.0
    close("attn.dq", qg.grad, qr.grad, 1e-4, 1e-6)
    close("attn.dk", kg.grad.permute(0, 2, 1), kr.grad, 1e-4, 1e-7)
    close("attn.dv", vg.grad.permute(0, 2, 1), vr.grad, 1e-4, 1e-7)


def test_attention_map_sized(ops):
    """I = 576 map cells, no mask, B = 9 (the map-attention call site)."""
    q = T(df.uniform("attn2.q", (9, 256), 6.0))
    kv = T(df.uniform("attn2.kv", (9, 576, 256), 3.0))
    o_ref, a_ref = policy_ref.attn(q.double(), kv.double().permute(0, 2, 1), kv.double().permute(0, 2, 1))
    o, a = ops.attention(q.cuda(), kv.cuda(), kv.cuda(), None, 1.0 / 16)
    close("attn576.out", o, o_ref, 1e-5, 1e-6)
    close("attn576.attn", a, a_ref, 1e-5, 1e-8)


# ----------------------------------------------------------------------------- BEV: the integer gate
@pytest.mark.parametrize("name", list(cases.BEV_CASES))
def test_bev_index_and_scatter_bit_exact(ops, name):
    g = golden("g1_bev.npz")
    c = cases.bev_inputs(name)
    E, C, Hf = c["E"], c["C"], c["Hf"]
    proj_ref, lin_ref, inv_ref, *_ = bev_ref.project_to_ground(c["feat"], c["depth"], E)
    lin = ops.bev_index(dev(c["depth"][..., 0]), Hf, Hf, E)
    lin_h = lin.cpu().numpy()
    got_inv = lin_h < 0
    assert np.array_equal(got_inv, inv_ref), f"{int((got_inv != inv_ref).sum())} validity flags differ"
    assert np.array_equal(np.where(got_inv, 0, lin_h), lin_ref), "linear cell index differs"
    assert sha(np.where(got_inv, 0, lin_h).astype(np.int32)) == str(g[name + ".lin_idx_sha"])
    assert sha(got_inv.astype(np.uint8)) == str(g[name + ".invalid_sha"])
    proj = ops.bev_scatter_max(dev(c["feat"]), lin, C, E).cpu().numpy()
    assert np.array_equal(proj.view(np.uint32), proj_ref.view(np.uint32)), \
        f"scatter-max differs in {int((proj != proj_ref).sum())} cells"
    assert sha(proj) == str(g[name + ".proj_sha"])


def test_bev_scatter_channel_pool(ops):
    """64 -> 40 channel adaptive max-pool fused into the scatter (cfg4 geometry)."""
    c = cases.bev_inputs("e200_c40_f256")
    feat64 = df.uniform("g1.pool.feat64", (c["B"], 64, 256, 256), 4.0)
    pooled = bev_ref.channel_maxpool(T(feat64), 40).numpy()
    proj_ref, *_ = bev_ref.project_to_ground(pooled, c["depth"], 200)
    lin = ops.bev_index(dev(c["depth"][..., 0]), 256, 256, 200)
    proj = ops.bev_scatter_max(dev(feat64), lin, 40, 200).cpu().numpy()
    assert np.array_equal(proj.view(np.uint32), proj_ref.view(np.uint32))


def test_bev_idempotent_and_empty(ops):
    """size-independent properties: all-invalid depth -> all-zero map; scattering twice is idempotent."""
    B, C, E, Hf = 2, 64, 100, 256
    feat = dev(df.uniform("bev.prop.feat", (B, C, Hf, Hf), 4.0))
    depth = torch.zeros(B, 256, 256, device="cuda")
    lin = ops.bev_index(depth, Hf, Hf, E)
    assert int((lin >= 0).sum()) == 0
    assert float(ops.bev_scatter_max(feat, lin, C, E).abs().max()) == 0.0
    depth = dev(df.uniform("bev.prop.depth", (B, 256, 256)) + np.float32(0.5))
    lin = ops.bev_index(depth, Hf, Hf, E)
    a = ops.bev_scatter_max(feat, lin, C, E)
    b = ops.bev_scatter_max(feat, lin, C, E)
    assert torch.equal(a, b)


def test_map_sequence_vs_oracle(ops):
    """rotate / paste / translate / max-fuse / retrieve over 4 steps with a mid-sequence episode
    reset (G2).  Tolerance: bilinear weights depend on float32 grid coordinates whose last bits
    differ between ATen's CPU kernels and ours (SURVEY §7 'Exactness of K3'): <= 1e-5 * G
    in the weights, times |feature| <= 2 -> 2e-4 absolute."""
    g = golden("g2_mapseq.npz")
    m = cases.MAP_SEQ
    ref = bev_ref.MapperRef(m["B"])
    gm = torch.zeros(m["B"], m["G"], m["G"], m["C"], device="cuda")
    for s in range(m["steps"]):
        c = cases.mapseq_inputs(s)
        ego_ref = ref.step(T(c["feat"]), T(c["depth"]), T(c["gps"]), T(c["compass"]), T(c["masks"]))
        lin = ops.bev_index(dev(c["depth"][..., 0]), m["Hf"], m["Hf"], m["E"])
        planes = ops.bev_scatter_max(dev(c["feat"]), lin, m["C"], m["E"])
        compass = dev(c["compass"]).reshape(-1).contiguous()
        gps = dev(c["gps"])
        rot = ops.bev_rotate(planes, compass, -1.0)
        ops.map_fuse(rot, gm, gps, dev(c["masks"]).reshape(-1).contiguous(), 0.12)
        ego = ops.map_retrieve(gm, gps, compass, m["E"], 0.12).permute(0, 3, 1, 2)
        close(f"s{s}.global", gm, ref.full_global_map, 0, 2e-4)
        close(f"s{s}.ego", ego, ego_ref, 0, 2e-4)
        np.testing.assert_allclose(ego[:, ::16, 40:56, 44:60].cpu().numpy(), g[f"s{s}.ego_patch"], atol=2e-4, rtol=0)


# ----------------------------------------------------------------------------- bf16 storage mode
def bf(x):
    return x.to(torch.bfloat16)


BF16_EPS = 2.0 ** -8  # one bf16 ulp at 1.0 is 2^-7; round-to-nearest error <= 2^-8 relative


@pytest.mark.parametrize("cfg", CONVS, ids=[c[0] for c in CONVS])
def test_conv2d_bf16_fwd_bwd(ops, cfg):
    """bf16 engine against a float64 convolution of the SAME bf16-rounded operands: what is left
    is float32 accumulation order plus one bf16 rounding of each output (<= 2^-8 relative)."""
    name, B, Cin, Cout, k, s, p, H = cfg
    x = bf(T(df.uniform(f"conv.{name}.x", (B, Cin, H, H), 2.0)))
    w = T(df.uniform(f"conv.{name}.w", (Cout, Cin, k, k), float(np.sqrt(12.0 / (Cin * k * k)))))
    b = T(df.uniform(f"conv.{name}.b", (Cout,), 0.5))
    xr = x.double().requires_grad_(True)
    wr = bf(w).double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    gy = bf(T(df.uniform(f"conv.{name}.gy", tuple(yr.shape), 2.0)))
    yr.backward(gy.double())

    xg = nhwc(x).cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, s, p)
    assert y.dtype == torch.bfloat16
    y.backward(nhwc(gy).cuda())
    assert wg.grad.dtype == torch.float32 and xg.grad.dtype == torch.bfloat16
    close(name + ".y", nchw(y.float()), yr, BF16_EPS, 1e-3)
    close(name + ".dx", nchw(xg.grad.float()), xr.grad, BF16_EPS, 1e-3)
    close(name + ".dw", wg.grad, wr.grad, 1e-4, 3e-4 * float(wr.grad.abs().max()) + 1e-6)
    close(name + ".db", bg.grad, br.grad, 1e-4, 3e-4 * float(br.grad.abs().max()) + 1e-6)


def test_conv_transpose2d_bf16(ops):
    B, Ci, Co, H = 2, 64, 32, 24
    x = bf(T(df.uniform("convt.x", (B, Ci, H, H), 2.0)))
    w = T(df.uniform("convt.w", (Ci, Co, 4, 4), 0.2))
    xr, wr = x.double().requires_grad_(True), bf(w).double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, stride=2, padding=1)
    gy = bf(T(df.uniform("convt.gy", tuple(yr.shape), 2.0)))
    yr.backward(gy.double())
    xg = nhwc(x).cuda().requires_grad_(True)
    wg = w.cuda().requires_grad_(True)
    y = ops.conv_transpose2d(xg, wg, 2, 1)
    y.backward(nhwc(gy).cuda())
    close("convt16.y", nchw(y.float()), yr, BF16_EPS, 1e-3)
    close("convt16.dx", nchw(xg.grad.float()), xr.grad, BF16_EPS, 1e-3)
    close("convt16.dw", wg.grad, wr.grad, 1e-4, 3e-4 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("C,H,B,res", [(64, 50, 3, False), (256, 24, 2, False), (64, 6, 5, True)])
def test_bn_act_bf16(ops, C, H, B, res):
    x = bf(T(df.uniform(f"bn.x.{C}.{H}", (B, C, H, H), 3.0)) + T(df.uniform(f"bn.off.{C}", (1, C, 1, 1), 2.0)))
    r = bf(T(df.uniform(f"bn.r.{C}.{H}", (B, C, H, H), 2.0))) if res else None
    g, bt = T(df.positive(f"bn.g.{C}", (C,))), T(df.uniform(f"bn.b.{C}", (C,), 0.5))
    rm, rv = T(df.uniform(f"bn.rm.{C}", (C,), 0.2)), T(df.positive(f"bn.rv.{C}", (C,)))
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), bt.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = F.batch_norm(xr, rmr, rvr, gr, br, training=True, momentum=0.1, eps=1e-5)
    yr = F.relu(yr + rr if res else yr)
    # the kernel masks the gradient with the bf16-ROUNDED output: feed the reference the same mask
    gy = bf(T(df.uniform(f"bn.gy.{C}.{H}", tuple(yr.shape), 2.0)))
    xg = nhwc(x).cuda().requires_grad_(True)
    gg, bg = g.cuda().requires_grad_(True), bt.cuda().requires_grad_(True)
    rg = nhwc(r).cuda().requires_grad_(True) if res else None
    rmg, rvg = rm.cuda(), rv.cuda()
    y = ops.bn_act(xg, gg, bg, rmg, rvg, True, True, rg, 0.1, 1e-5)
    assert y.dtype == torch.bfloat16
    close("bn16.y", nchw(y.float()), yr, BF16_EPS, 1e-6)
    close("bn16.running_mean", rmg, rmr, 1e-6, 1e-6)
    close("bn16.running_var", rvg, rvr, 1e-5, 1e-6)
    mask = (nchw(y.float()).cpu() > 0).double()
    # reference backward with the kernel's own mask (elements that rounded to 0 carry no gradient)
    pre = F.batch_norm(xr, rm.double().clone(), rv.double().clone(), gr, br, training=True, momentum=0.1, eps=1e-5)
    pre = pre + rr if res else pre
    (pre * mask * gy.double()).sum().backward()
    y.backward(nhwc(gy).cuda())
    scale = float(xr.grad.abs().max())
    close("bn16.dx", nchw(xg.grad.float()), xr.grad, BF16_EPS, 2e-3 * scale)
    close("bn16.dgamma", gg.grad, gr.grad, 1e-4, 1e-4 * float(gr.grad.abs().max()))
    close("bn16.dbeta", bg.grad, br.grad, 1e-4, 1e-4 * float(br.grad.abs().max()))


def test_small_ops_and_layout_bf16(ops):
    B, C, H = 3, 64, 12
    x = bf(torch.relu(T(df.uniform("pool.x", (B, C, H, H), 2.0))))
    for nm, fn_ref, fn in [
        ("maxpool", lambda t: F.max_pool2d(t, 3, 2, 1), ops.maxpool3x3s2),
        ("upsample", lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True), ops.upsample2x),
        ("avgpool", lambda t: F.avg_pool2d(t, 2, 2), ops.avgpool2),
        ("relu", lambda t: F.relu(t), ops.relu),
    ]:
        xr = x.double().requires_grad_(True)
        yr = fn_ref(xr)
        gy = bf(T(df.uniform(f"pool.gy.{nm}", tuple(yr.shape), 2.0)))
        yr.backward(gy.double())
        xg = nhwc(x).cuda().requires_grad_(True)
        y = fn(xg)
        assert y.dtype == torch.bfloat16
        y.backward(nhwc(gy).cuda())
        close(nm + "16.y", nchw(y.float()), yr, BF16_EPS, 1e-6)
        close(nm + "16.dx", nchw(xg.grad.float()), xr.grad, BF16_EPS, 1e-6)
    e = T(df.uniform("lay.e", (2, 64, 100, 100), 2.0)).cuda()
    assert torch.equal(ops.to_nhwc(e, dtype=torch.bfloat16), bf(e.permute(0, 2, 3, 1)))
    s27 = bf(T(df.uniform("lay.s", (2, 48, 48, 32), 2.0))).cuda().requires_grad_(True)
    z = ops.to_nchw(s27, 27)
    assert z.dtype == torch.float32 and torch.equal(z, s27.detach().float().permute(0, 3, 1, 2)[:, :27])
    z.backward(torch.ones_like(z))
    assert float(s27.grad[..., :27].float().min()) == 1.0 and float(s27.grad[..., 27:].float().abs().max()) == 0.0


def test_attention_bf16(ops):
    q = T(df.uniform("attn2.q", (9, 256), 6.0))
    kv = bf(T(df.uniform("attn2.kv", (9, 576, 256), 3.0)))
    qr, kr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    o_ref, a_ref = policy_ref.attn(qr, kr.permute(0, 2, 1), kr.permute(0, 2, 1))
    go, ga = T(df.uniform("attn.go16", (9, 256), 2.0)), T(df.uniform("attn.ga16", (9, 576), 2.0))
    ((o_ref * go.double()).sum() + (a_ref * ga.double()).sum()).backward()
    qg = q.cuda().requires_grad_(True)
    kg = kv.cuda().requires_grad_(True)
    o, a = ops.attention(qg, kg, kg, None, 1.0 / 16)
    ((o * go.cuda()).sum() + (a * ga.cuda()).sum()).backward()
    close("attn16.out", o, o_ref, 1e-5, 1e-6)
    close("attn16.attn", a, a_ref, 1e-5, 1e-8)
    close("attn16.dq", qg.grad, qr.grad, 1e-4, 1e-6)
    # k and v are the same tensor here: dk and dv are each rounded to bf16 and then summed in bf16
    close("attn16.dkv", kg.grad.float(), kr.grad, 4 * BF16_EPS, 2 * BF16_EPS * float(kr.grad.abs().max()))


# ----------------------------------------------------------------------------- persistent masked GRU
@pytest.mark.parametrize("Tn,N", [(1, 3), (6, 3), (64, 8), (9, 11)])
def test_masked_gru_persistent_kernel(Tn, N):
    """whole-sequence HIP GRU (forward + BPTT) vs the oracle's per-step masked GRU in float64,
    with episode restarts in the middle of the sequence."""
    from wsmgmap.models.rnn_state_encoder import RNNStateEncoder
    In, Hd = 640, 512
    enc = RNNStateEncoder(In, Hd)
    sd = {k: T(df.uniform(f"gru.{k}", tuple(v.shape), 0.2 if "bias" in k else float(np.sqrt(12.0 / v.shape[1]))))
          for k, v in enc.rnn.state_dict().items()}
    enc.rnn.load_state_dict(sd)
    x = T(df.uniform(f"gru.x.{Tn}.{N}", (Tn * N, In), 2.0))
    h0 = T(df.uniform(f"gru.h0.{N}", (1, N, Hd), 1.0))
    masks = torch.ones(Tn, N)
    masks[0, : max(1, N // 2)] = 0
    if Tn > 3:
        masks[3, 1] = 0
        masks[Tn - 1, 0] = 0
    gy = T(df.uniform(f"gru.gy.{Tn}.{N}", (Tn * N, Hd), 2.0))
    # float64 truth
    P = {"e.rnn." + k: v.double().requires_grad_(True) for k, v in sd.items()}
    xr, hr = x.double().requires_grad_(True), h0.double().requires_grad_(True)
    torch.set_default_dtype(torch.float64)
    try:
        yr, hTr = policy_ref.masked_gru(P, "e", xr, hr, masks.double().view(-1, 1))
    finally:
        torch.set_default_dtype(torch.float32)
    (yr * gy.double()).sum().backward()
    # HIP
    enc = enc.cuda()
    xg, hg = x.cuda().requires_grad_(True), h0.cuda().requires_grad_(True)
    y, hT = enc(xg, hg, masks.view(-1, 1).cuda())
    (y * gy.cuda()).sum().backward()
    close("gru.y", y, yr, 1e-5, 2e-6)
    close("gru.hT", hT, hTr, 1e-5, 2e-6)
    close("gru.dx", xg.grad, xr.grad, 1e-4, 1e-5 * float(xr.grad.abs().max()))
    close("gru.dh0", hg.grad, hr.grad, 1e-4, 1e-5 * float(hr.grad.abs().max()) + 1e-9)
    for k in sd:
        ref = P["e.rnn." + k].grad
        close("gru.d" + k, getattr(enc.rnn, k).grad, ref, 1e-4, 2e-5 * float(ref.abs().max()) + 1e-9)


# ----------------------------------------------------------------------------- persistent packed bi-LSTM
@pytest.mark.parametrize("lens", [[80, 37], [5, 1, 200, 64, 64, 199, 3, 120], [10] * 11])
def test_bilstm_persistent_kernel(lens):
    """HIP packed bi-LSTM (forward + BPTT) behind InstructionEncoder vs the oracle's packed nn.LSTM in
    float64: outputs, pad mask, and the gradients of every LSTM parameter."""
    from util import make_params
    from wsmgmap.config import default_model_config
    from wsmgmap.models.encoders.instruction_encoder import InstructionEncoder
    cfg = default_model_config().INSTRUCTION_ENCODER
    enc = InstructionEncoder(cfg)
    P0 = make_params(grad=False)
    pre = "net.instruction_encoder."
    enc.load_state_dict({k[len(pre):]: v for k, v in P0.items() if k.startswith(pre)})
    instr = T(df.tokens(f"lstm.tok.{len(lens)}", len(lens), lens).astype(np.float32))
    # float64 truth
    P = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in P0.items() if k.startswith(pre)}
    torch.set_default_dtype(torch.float64)
    try:
        hr, mr = policy_ref.instruction_encoder(P, instr)
    finally:
        torch.set_default_dtype(torch.float32)
    gy = T(df.uniform(f"lstm.gy.{len(lens)}", tuple(hr.shape), 2.0))
    (hr * gy.double()).sum().backward()
    enc = enc.cuda()
    hid, mask = enc({"instruction": instr.cuda()})
    assert hid.shape == hr.shape and torch.equal(mask.cpu(), mr)
    (hid * gy.cuda()).sum().backward()
    close("lstm.out", hid, hr, 1e-5, 2e-6)
    for k in ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
              "bias_ih_l0_reverse", "bias_hh_l0_reverse"]:
        ref = P[pre + "encoder_rnn." + k].grad
        close("lstm.d" + k, getattr(enc.encoder_rnn, k).grad, ref, 1e-4, 2e-5 * float(ref.abs().max()) + 1e-9)
    ref = P[pre + "embedding_layer.weight"].grad
    close("lstm.demb", enc.embedding_layer.weight.grad, ref, 1e-4, 2e-5 * float(ref.abs().max()) + 1e-9)


@pytest.mark.parametrize("Tn", [33, 64])
def test_rnn_handoff_under_concurrent_load(ops, Tn):
    """The persistent GRU/LSTM kernels exchange h_t between workgroups on different XCDs inside one
    launch.  Run them repeatedly while a second stream saturates the chip with bf16 MFMA convs, with the
    exchange images poisoned with NaN before every launch: a stale or missed hand-off, or the wrong
    partial sums these kernels produced when a conv workgroup shared their CU (they now own the CU, see
    wsmg_rnn.hip "CU ownership"), surface as a NaN or a bitwise difference from the first, unloaded run."""
    torch.manual_seed(0)
    N, Hd = 8, 512
    gi = torch.randn(Tn, N, 3 * Hd, device="cuda")
    whh = torch.randn(3 * Hd, Hd, device="cuda") * 0.04
    bhh = torch.randn(3 * Hd, device="cuda") * 0.1
    h0 = torch.randn(N, Hd, device="cuda")
    masks = torch.ones(Tn, N, device="cuda")
    masks[0] = 0
    masks[Tn // 2, 3] = 0
    gy = torch.randn(Tn, N, Hd, device="cuda")
    U, L = 8, 60
    lgi = torch.randn(U, L, 2, 512, device="cuda")
    lw = torch.randn(2, 512, 128, device="cuda") * 0.08
    lb = torch.randn(2, 512, device="cuda") * 0.1
    lens = torch.tensor([60, 37, 1, 44, 60, 12, 55, 59], device="cuda", dtype=torch.int32)
    lgy = torch.randn(U, L, 256, device="cuda")

    def run():
        g = gi.clone().requires_grad_(True)
        y = ops.masked_gru(g, whh, bhh, h0, masks)
        (y * gy).sum().backward()
        lg = lgi.clone().requires_grad_(True)
        o = ops.bilstm(lg, lw, lb, lens)
        (o * lgy).sum().backward()
        return [y.detach(), g.grad, o.detach(), lg.grad]

    ref = run()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t).all()) for t in ref)
    from wsmgmap.debug import sw
    old = sw.rnn_poison
    sw.rnn_poison = True
    try:
        side = torch.cuda.Stream()
        x = torch.randn(256, 24, 24, 256, device="cuda").to(torch.bfloat16)
        wconv = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
        for i in range(40):
            with torch.cuda.stream(side):
                for _ in range(4):
                    ops.conv2d(x, wconv, None, 1, 1)
            out = run()
            for name, a, b in zip(["gru.y", "gru.dgi", "lstm.out", "lstm.dgi"], ref, out):
                assert torch.equal(a, b), f"repeat {i}: {name} differs under load (max {float((a - b).abs().max()):.3e})"
        torch.cuda.synchronize()
    finally:
        sw.rnn_poison = old


# ------------------------------------------------------------------ fp8 text attention (configs[4])
def _cfg5_inputs(B=64, L=160, C=256, seed=5):
    rng = np.random.RandomState(seed)
    q = rng.randn(B, C).astype(np.float32)
    w = (rng.randn(C, C) / 16).astype(np.float32)
    b = (rng.randn(C) * 0.1).astype(np.float32)
    x = rng.randn(B, L, C).astype(np.float32)
    lengths = rng.randint(1, L + 1, size=B).astype(np.int32)
    lengths[0], lengths[1] = L, 1
    return q, w, b, x, lengths


@pytest.mark.gpu
def test_quantize_e4m3_bit_exact(ops):
    """The HIP quantiser and the oracle's (torch CPU cast) agree on every code, incl. ties, subnormals, saturation."""
    from oracle import attn_fp8_ref as ar
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.randn(4096) * s for s in (1e-3, 0.02, 1.0, 30.0, 300.0)]
                       + [np.array([0.0, -0.0, 448.0, -448.0, 500.0, -1e9, 2 ** -9, 2 ** -10, 0.0009765625 * 1.5, 17.0, 18.0, 19.0],
                                   dtype=np.float64)]).astype(np.float32)
    x = x[: len(x) // 4 * 4]
    for scale in (1.0, 0.0371):
        want = ar.quantize_e4m3(x, scale)
        got = ops.quantize_e4m3(T(x).cuda(), scale).cpu().numpy()
        assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.gpu
def test_attn_fp8_fused_cfg5(ops):
    """BASELINE configs[4]: B=64, L=160 (ragged lengths incl. 1 and L), e4m3 tokens, float32 accumulate.  The fused
    kernel (key projection folded into the query, x read once) against the float64 evaluation of the reference
    formula on the same de-quantised tokens: attention weights within 2e-5, outputs within 2e-5 * max|out|."""
    from oracle import attn_fp8_ref as ar
    q, w, b, x, lengths = _cfg5_inputs()
    x_scale = float(np.abs(x).max() / ar.E4M3_MAX)
    codes = ar.quantize_e4m3(x, x_scale)
    out_ref, attn_ref = ar.attn_fp8(q, w, b, codes, x_scale, lengths, 1.0 / 16)
    codes_dev = ops.quantize_e4m3(T(x).cuda(), x_scale)
    assert np.array_equal(codes_dev.cpu().numpy(), codes)
    out, attn = ops.attn_fp8_fused(T(q).cuda(), T(w).cuda(), T(b).cuda(), codes_dev, x_scale, torch.from_numpy(lengths).cuda(), 1.0 / 16)
    out, attn = out.cpu().numpy(), attn.cpu().numpy()
    assert np.abs(attn - attn_ref).max() <= 2e-5
    assert np.abs(out - out_ref).max() <= 2e-5 * np.abs(out_ref).max()
    # masked tokens carry exactly zero weight; every row sums to one
    L = x.shape[1]
    assert all(float(np.abs(attn[i, lengths[i]:]).sum()) == 0.0 for i in range(len(lengths)))
    assert np.abs(attn.sum(1) - 1).max() <= 1e-5
    # against the unquantised float32 formula the error is the e4m3 quantisation error (3 mantissa bits)
    k = x @ w.T + b
    lg = (np.einsum("bc,blc->bl", q, k) - 1e8 * (np.arange(L)[None] >= lengths[:, None])) / 16
    a32 = np.exp(lg - lg.max(1, keepdims=True)); a32 /= a32.sum(1, keepdims=True)
    assert np.abs(attn - a32).max() <= 0.08


@pytest.mark.gpu
@pytest.mark.parametrize("B,L", [(64, 160), (3, 37), (300, 80), (17, 200)])
def test_attn_fp8_forward_backward_vs_oracle(ops, B, L):
    """configs[4] forward + backward through the trainable op (device-side amax / 448 scale, e4m3 bytes, row split over
    workgroups, float32-MFMA query fold): out / attn within 2e-5, dq, dW_k, dx within 2e-5 of max|.| of the float64 autograd
    of the reference formula on the same de-quantised tokens; db_k is exactly zero (it cancels in the softmax: the oracle's
    is ~1e-17); ragged lengths incl. 1 and L, B not a multiple of 16, L not a multiple of the chunk."""
    from oracle import attn_fp8_ref as ar
    q, w, b, x, lengths = _cfg5_inputs(B=B, L=L, seed=B + L)
    rng = np.random.RandomState(9)
    dout = rng.randn(B, 256).astype(np.float32)
    dattn = (rng.randn(B, L) * 0.3).astype(np.float32)
    # the op's per-tensor scale: amax / 448 as the device computes it (a reciprocal multiply: 1 ulp from numpy's division)
    x_scale = float(T(x).cuda().abs().amax() / 448.0)
    codes = ar.quantize_e4m3(x, x_scale)
    out_ref, attn_ref = ar.attn_fp8(q, w, b, codes, x_scale, lengths, 1.0 / 16)
    dq_r, dw_r, db_r, dx_r = ar.attn_fp8_grads(q, w, b, codes, x_scale, lengths, 1.0 / 16, dout, dattn)
    qt, wt, bt, xt = [T(a).cuda().requires_grad_(True) for a in (q, w.reshape(256, 256, 1), b, x)]
    out, attn = ops.attention_fp8(qt, wt, bt, xt, torch.from_numpy(lengths).cuda(), 1.0 / 16)
    ((out * T(dout).cuda()).sum() + (attn * T(dattn).cuda()).sum()).backward()
    assert np.abs(attn.detach().cpu().numpy() - attn_ref).max() <= 2e-5
    assert np.abs(out.detach().cpu().numpy() - out_ref).max() <= 2e-5 * np.abs(out_ref).max()
    for got, want, name in ((qt.grad, dq_r, "dq"), (wt.grad.reshape(256, 256), dw_r, "dW_k"), (xt.grad, dx_r, "dx")):
        err = np.abs(got.cpu().numpy() - want).max() / np.abs(want).max()
        assert err <= 2e-5, (name, err)
    assert float(bt.grad.abs().max()) == 0.0 and np.abs(db_r).max() <= 1e-9
    assert all(float(attn[i, lengths[i]:].abs().sum()) == 0.0 for i in range(B))
    # a second launch on the same stream: the ticket words were handed back zeroed
    out2, attn2 = ops.attention_fp8(qt.detach(), wt.detach(), bt.detach(), xt.detach(), torch.from_numpy(lengths).cuda(), 1.0 / 16)
    assert torch.equal(attn2, attn.detach()) or float((attn2 - attn.detach()).abs().max()) <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_attention_with_folded_key_projection(ops, dt, masked):
    """ops.attention_folded == softmax(scale * q.(W x + b)) . x of mg_map_policy.py:126-132,173-178 with the
    Conv1d key projection applied explicitly (plain torch, float64), forward and every gradient; the key
    bias gets an exactly-zero gradient (its logit contribution is constant over the tokens)."""
    torch.manual_seed(3)
    B, I, C = 6, 37, 256
    q = torch.randn(B, C, device="cuda")
    w = torch.randn(C, C, 1, device="cuda") / 16
    b = torch.randn(C, device="cuda") * 0.3
    x = torch.randn(B, I, C, device="cuda")
    if dt == "bf16":
        x = x.bfloat16()
    gout, gattn = torch.randn(B, C, device="cuda"), torch.randn(B, I, device="cuda") * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (q, w, b, x)]
    # (keys are values here: the one-pass / read-pass + write-pass kernels of csrc/wsmg_attn.hip; 37 tokens = 4 full
    # trips of 8 and a ragged one; masked rows keep at least their first token)
    mask = None
    if masked:
        mask = torch.rand(B, I, device="cuda") < 0.4
        mask[:, 0] = False
        mask[2] = False
        mask[3, 1:] = True
    out, attn = ops.attention_folded(leaves[0], leaves[1], leaves[2], leaves[3], mask, 1 / 16)
    ((out * gout).sum() + (attn * gattn).sum()).backward()
    ref = [t.detach().double().requires_grad_(True) for t in (q, w, b, x)]
    k = ref[3] @ ref[1][:, :, 0].t() + ref[2]
    lgt = torch.einsum("bc,bic->bi", ref[0], k)
    if masked:
        lgt = lgt - 1e8 * mask.double()
        assert float(attn[3, 1:].abs().max()) == 0.0 and abs(float(attn[3, 0]) - 1.0) < 1e-6
    a = torch.softmax(lgt / 16, dim=1)
    o = torch.einsum("bi,bic->bc", a, ref[3])
    ((o * gout.double()).sum() + (a * gattn.double()).sum()).backward()
    tol = 1e-5 if dt == "f32" else 2e-2
    assert float((out.double() - o).abs().max()) <= tol and float((attn.double() - a).abs().max()) <= 1e-5
    for name, l, r in zip(("dq", "dw", "db", "dx"), leaves, ref):
        scale = float(r.grad.abs().max()) + 1e-12
        err = float((l.grad.double() - r.grad).abs().max())
        if name == "db":
            assert float(l.grad.abs().max()) == 0.0 and scale < 1e-9
        else:
            assert err <= (2e-5 if dt == "f32" else 2e-2) * scale, (name, err, scale)


# ------------------------------------------------------------------ trajectory-cache device collate
@pytest.mark.gpu
def test_device_collate_matches_reference_golden(ops):
    """DeviceCollator (compact dtypes over PCIe, pad + interleave + float32 conversion in wsmg_collate_pad) produces
    bit-for-bit what the reference's collate_fn + `.float().to(device)` produces (g6: ragged batch; 200-step cap)."""
    import hashlib
    import os
    from oracle import data_cases as dc
    from wsmgmap.data import DeviceCollator
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_g7_data.npz"))
    dsha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    coll = DeviceCollator("cuda")
    for tag, lengths in (("rag", dc.COLLATE_LENGTHS), ("long", dc.LONG_LENGTHS)):
        batch = [dc.episode(100 + i, n) + (torch.ones(n),) for i, n in enumerate(lengths)]
        ob, prev, masks, corr, wts = coll(batch)
        torch.cuda.synchronize()
        for k, v in ob.items():
            assert v.dtype == torch.float32 and list(v.shape) == g[f"g6_{tag}_obs_{k}_shape"].tolist()
            assert dsha(v.cpu().numpy()) == str(g[f"g6_{tag}_obs_{k}_sha"]), (tag, k)
        for name, v in (("prev", prev), ("masks", masks), ("corr", corr), ("wts", wts)):
            assert list(v.shape) == g[f"g6_{tag}_{name}_shape"].tolist()
            assert dsha(v.cpu().numpy()) == str(g[f"g6_{tag}_{name}_sha"]), (tag, name)
    # a second call re-uses the pinned staging buffer
    ob2, *_ = coll([dc.episode(100 + i, n) + (torch.ones(n),) for i, n in enumerate(dc.COLLATE_LENGTHS)])
    assert torch.equal(ob2["progress"].cpu(), torch.from_numpy(g["g6_rag_progress"]))


class _Store:
    """picklable in-memory record store (stands in for the reference's LMDB)"""
    def __init__(self, blobs):
        self.blobs = blobs

    def __call__(self, i):
        return self.blobs[i]


@pytest.mark.gpu
def test_device_feeder_end_to_end(ops):
    """Records -> TrajectoryDataset -> DeviceFeeder (side-stream DeviceCollator, 2 batches ahead) yields the same
    batches as the host path (same `random` seed, in-process decode); with worker processes every record of the
    shard arrives exactly once."""
    import random
    from oracle import data_cases as dc
    from wsmgmap.data import TrajectoryDataset, DeviceFeeder, collate_fn, pack_record
    store = _Store([pack_record(*dc.episode(1000 + i, n)) for i, n in enumerate(dc.DATASET_LENGTHS)])
    mk = lambda: TrajectoryDataset(store, len(store.blobs), batch_size=4, rank=0, world_size=1)  # noqa: E731
    random.seed(21)
    want = []
    items = [x for x in mk()]          # one __iter__ call, like the DataLoader fetcher
    for i in range(0, len(items) - len(items) % 4, 4):
        want.append(collate_fn(items[i:i + 4]))
    random.seed(21)
    got = list(DeviceFeeder(mk(), 4, "cuda", num_workers=0, prefetch=2))
    torch.cuda.synchronize()
    assert len(got) == len(want) == 5
    for (ob, prev, masks, corr, wts), (ob2, prev2, masks2, corr2, wts2) in zip(want, got):
        assert all(torch.equal(ob[k].float(), ob2[k].cpu()) for k in ob)
        assert torch.equal(prev, prev2.cpu()) and torch.equal(masks, masks2.cpu()) and torch.equal(corr, corr2.cpu()) and torch.equal(wts, wts2.cpu())
    firsts = []
    for ob, prev, masks, corr, wts in DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2):
        firsts += prev.view(-1, 2, 2)[0, :, 0].cpu().tolist()       # prev_actions[t=0, n, 0] of the batch's episodes
    # 2 workers x 11 records (floor sharding, 22 of 23); drop_last=True as in the reference (dagger_trainer.py:591)
    # drops each worker's odd record: 20 distinct episodes arrive, each once
    pool = [float(dc.episode(1000 + i, n)[1][0, 0]) for i, n in enumerate(dc.DATASET_LENGTHS[:22])]
    assert len(firsts) == 20 and len(set(firsts)) == 20 and set(firsts) <= set(pool)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_cross_entropy_nhwc(ops, dt):
    """Fused per-pixel CE over NHWC logits == F.cross_entropy on the NCHW view (policy.py:61-66), loss and gradient;
    padded channels 27..31 receive zero gradient."""
    torch.manual_seed(2)
    B, S, C = 3, 10, 27
    logits = torch.randn(B, S, S, 32, device="cuda") * 3
    logits[..., C:] = 0
    if dt == "bf16":
        logits = logits.bfloat16()
    target = torch.randint(0, C, (B, S, S), device="cuda")
    gl = torch.randn(B, S, S, device="cuda")
    x = logits.clone().requires_grad_(True)
    loss = ops.cross_entropy_nhwc(x, target, C)
    (loss * gl).sum().backward()
    xr = logits.double().requires_grad_(True)
    ref = F.cross_entropy(xr[..., :C].permute(0, 3, 1, 2), target, reduction="none")
    (ref * gl.double()).sum().backward()
    assert float((loss.double() - ref).abs().max()) <= 2e-6 * (1 + float(ref.abs().max()))
    tol = 2e-6 if dt == "f32" else 1e-2
    assert float((x.grad.double() - xr.grad).abs().max()) <= tol * float(xr.grad.abs().max())
    assert float(x.grad[..., C:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_attention_over_shared_sets(ops, dt):
    """ops.attention_shared(q, k_sets, v_sets, mask_sets, inverse) == the reference attention (mg_map_policy.py:173-178)
    on per-row gathered copies k_sets[inverse], v_sets[inverse], mask_sets[inverse] (float64 torch), forward and the
    gradients w.r.t. q and the U shared sets (rows of one set accumulate)."""
    torch.manual_seed(6)
    B, U, I, C = 40, 3, 37, 256
    q = torch.randn(B, C, device="cuda")
    k = torch.randn(U, I, C, device="cuda")
    v = torch.randn(U, I, C, device="cuda")
    if dt == "bf16":
        k, v = k.bfloat16(), v.bfloat16()
    lens = torch.tensor([37, 12, 1], device="cuda")
    mask = (torch.arange(I, device="cuda")[None] >= lens[:, None])
    inverse = torch.randint(0, U, (B,), device="cuda")
    inverse[:3] = torch.arange(3, device="cuda")
    gout, gattn = torch.randn(B, C, device="cuda"), torch.randn(B, I, device="cuda") * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (q, k, v)]
    out, attn = ops.attention_shared(leaves[0], leaves[1], leaves[2], mask.to(torch.uint8), inverse, 1 / 16)
    ((out * gout).sum() + (attn * gattn).sum()).backward()
    ref = [t.detach().double().requires_grad_(True) for t in (q, k, v)]
    kk, vv = ref[1][inverse], ref[2][inverse]
    lg = (torch.einsum("bc,bic->bi", ref[0], kk) - 1e8 * mask[inverse].double()) / 16
    a = torch.softmax(lg, dim=1)
    o = torch.einsum("bi,bic->bc", a, vv)
    ((o * gout.double()).sum() + (a * gattn.double()).sum()).backward()
    assert float((attn.double() - a).abs().max()) <= 1e-5 and float((out.double() - o).abs().max()) <= 1e-4
    for name, l, r in zip(("dq", "dk_sets", "dv_sets"), leaves, ref):
        tol = (3e-5 if dt == "f32" else 2e-2) * float(r.grad.abs().max())
        assert float((l.grad.double() - r.grad).abs().max()) <= tol, name


# ------------------------------------------------------------------ size-independent properties at BASELINE's full sizes
@pytest.mark.gpu
def test_fullsize_conv_linearity_bf16_cfg2(ops):
    """BASELINE configs[1] batch (B = T*N = 512), layer enc6 (128 -> 256, k3, 24x24): scaling the input by 2 (exact in
    bf16) doubles y bit for bit and leaves dx bit-identical (the engine has no data-dependent path); dW doubles up
    to the order in which the float32 atomics of the pixel-split reduction land (1e-5 relative)."""
    torch.manual_seed(9)
    B, Cin, Cout, H = 512, 128, 256, 24
    x = torch.randn(B, H, H, Cin, device="cuda").bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03)
    gy = torch.randn(B, H, H, Cout, device="cuda").bfloat16()

    def run(xin):
        xi = xin.clone().requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        y = ops.conv2d(xi, wi, None, 1, 1)
        y.backward(gy)
        return y.detach(), xi.grad, wi.grad
    y1, dx1, dw1 = run(x)
    y2, dx2, dw2 = run(x * 2)
    assert torch.equal(y2.float(), y1.float() * 2)
    assert torch.equal(dx2, dx1)                         # dx does not depend on x
    assert float((dw2 - 2 * dw1).abs().max()) <= 1e-5 * float(dw1.abs().max())
    assert bool(torch.isfinite(dw1).all())


@pytest.mark.gpu
def test_fullsize_map_fuse_idempotent_cfg4(ops):
    """BASELINE configs[3] geometry (B=32, E=200, C=40, 256^2 RGB-D): fusing the same ego map twice leaves the global
    map bit-identical (max is idempotent), retrieval of the fused map is deterministic, and every retrieved value is
    bounded by the maximum of what was scattered."""
    torch.manual_seed(10)
    B, E, C, G = 32, 200, 40, 480
    depth = torch.rand(B, 256, 256, device="cuda")
    depth[:, :8] = 0
    feat = torch.relu(torch.randn(B, 64, 256, 256, device="cuda"))
    gps = (torch.rand(B, 2, device="cuda") - 0.5) * 4
    compass = (torch.rand(B, device="cuda") - 0.5) * 6.28
    masks = torch.ones(B, device="cuda")
    lin = ops.bev_index(depth, 256, 256, E)
    assert int(lin.max()) < E * E and int(lin.min()) >= -1
    planes = ops.bev_scatter_max(feat, lin, C, E)
    assert torch.equal(planes, ops.bev_scatter_max(feat, lin, C, E))
    rot = ops.bev_rotate(planes, compass, -1.0)
    gm = torch.zeros(B, G, G, C, device="cuda")
    ops.map_fuse(rot, gm, gps, masks, 0.12)
    once = gm.clone()
    ops.map_fuse(rot, gm, gps, masks, 0.12)
    assert torch.equal(gm, once)
    e1 = ops.map_retrieve(gm, gps, compass, E, 0.12)
    e2 = ops.map_retrieve(gm, gps, compass, E, 0.12)
    assert torch.equal(e1, e2) and float(e1.min()) >= 0.0
    assert float(e1.max()) <= float(planes.max()) * (1 + 1e-6)


@pytest.mark.gpu
def test_fullsize_gru_sequence_split_cfg2(ops):
    """T=64, N=8 (configs[1]): one 64-step launch == two 32-step launches with the hidden state carried over, bit for
    bit, forward (y) and backward (d gi, d h0) — the recurrence has no dependence on how the sequence is cut."""
    torch.manual_seed(12)
    Tn, N, Hd = 64, 8, 512
    gi = torch.randn(Tn, N, 3 * Hd, device="cuda")
    whh = torch.randn(3 * Hd, Hd, device="cuda") * 0.04
    bhh = torch.randn(3 * Hd, device="cuda") * 0.1
    h0 = torch.randn(N, Hd, device="cuda")
    masks = torch.ones(Tn, N, device="cuda")
    masks[0] = 0
    masks[40, 2] = 0
    gy = torch.randn(Tn, N, Hd, device="cuda")
    g1 = gi.clone().requires_grad_(True)
    h1 = h0.clone().requires_grad_(True)
    y = ops.masked_gru(g1, whh, bhh, h1, masks)
    (y * gy).sum().backward()
    ga, gb = gi[:32].clone().requires_grad_(True), gi[32:].clone().requires_grad_(True)
    h2 = h0.clone().requires_grad_(True)
    ya = ops.masked_gru(ga, whh, bhh, h2, masks[:32].contiguous())
    yb = ops.masked_gru(gb, whh, bhh, ya[-1], masks[32:].contiguous())
    ((ya * gy[:32]).sum() + (yb * gy[32:]).sum()).backward()
    assert torch.equal(torch.cat([ya, yb]), y)
    assert torch.equal(torch.cat([ga.grad, gb.grad]), g1.grad)
    assert torch.equal(h2.grad, h1.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv2d_with_fused_relu(ops, dt):
    """ops.conv2d(..., relu=True) == relu(conv2d(...)) (mg_map_policy.py:89-100 Conv3+ReLU heads), forward and all
    gradients, in both modes (bf16: ReLU in the conv epilogue, gradient masked with the saved output)."""
    torch.manual_seed(8)
    B, Cin, Cout, H = 3, 64, 128, 12
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(Cout, device="cuda") * 0.1
    gy = torch.randn(B, H, H, Cout, device="cuda")
    if dt == "bf16":
        x, gy = x.bfloat16(), gy.bfloat16()
    outs = []
    for fused in (True, False):
        xi, wi, bi = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv2d(xi, wi, bi, 1, 1, relu=True) if fused else ops.relu(ops.conv2d(xi, wi, bi, 1, 1))
        y.backward(gy)
        outs.append((y.detach().float(), xi.grad.float(), wi.grad, bi.grad))
    for a, c, name in zip(outs[0], outs[1], ("y", "dx", "dw", "db")):
        assert float((a - c).abs().max()) <= 1e-5 * (1 + float(c.abs().max())), name
    assert float(outs[0][0].min()) >= 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("relu,pad_last", [(False, 0), (True, 0), (True, 5)])
def test_conv2d_over_concatenated_parts(ops, relu, pad_last):
    """ops.conv2d_cat([a, b], w) == conv2d(torch.cat([a, b], -1), w) (map_encoder.py:104,110 / mg_map_policy.py:99): one
    vectorised concatenation launch and the convolution, the gradients of the parts coming back as channel slices; forward and
    every gradient are held against a float64 evaluation of the concatenated convolution.
    pad_last: the last part carries 5 zero-padded channels beyond the weight's (27 -> 32 logits case)."""
    torch.manual_seed(11)
    B, Ca, Cb, Cout, H = 3, 64, 32, 64, 12
    a = torch.randn(B, H, H, Ca, device="cuda").bfloat16()
    b = torch.randn(B, H, H, Cb, device="cuda").bfloat16()
    if pad_last:
        b[..., Cb - pad_last:] = 0
    w = torch.randn(Cout, Ca + Cb - pad_last, 3, 3, device="cuda") * 0.05
    bias = torch.randn(Cout, device="cuda") * 0.1
    gy = torch.randn(B, H, H, Cout, device="cuda").bfloat16()
    ai, bi = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    wi, biasi = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y = ops.conv2d_cat([ai, bi], wi, biasi, 1, 1, relu=relu)
    assert y.dtype == torch.bfloat16
    y.backward(gy)
    # float64 reference on the same bf16-rounded inputs.  With ReLU the gradient is taken through the mask the engine
    # actually applied (its bf16 output > 0): pre-activations within rounding distance of zero can fall on either side,
    # which is a property of bf16 storage, not of the part-by-part route
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    wr, biasr = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    xcat = torch.cat([ar, br[..., :Cb - pad_last]], dim=-1).permute(0, 3, 1, 2)
    pre = F.conv2d(xcat, wr, biasr, 1, 1)
    yr = F.relu(pre) if relu else pre
    out = pre * (nchw(y.detach().float()) > 0).double() if relu else pre
    out.backward(gy.double().permute(0, 3, 1, 2))
    # y is rounded to bf16 once per part, and the first part's partial sum can be as large as the largest output while
    # the final value is small: the error bound is relative to the output scale, two half-ulps of bf16 (2^-9 each)
    sy = float(yr.detach().abs().max())
    close("cat.y", nchw(y.float()), yr, 2.0 ** -8, 2.0 ** -8 * sy)

    def rel(got, ref):
        return float((got.double() - ref).norm() / (ref.norm() + 1e-30))
    assert rel(ai.grad.float(), ar.grad) < 1e-2
    assert rel(bi.grad.float()[..., :Cb - pad_last], br.grad[..., :Cb - pad_last]) < 1e-2
    assert rel(wi.grad, wr.grad) < 1e-2 and rel(biasi.grad, biasr.grad) < 1e-2
    if pad_last:
        assert tuple(bi.grad.shape) == tuple(b.shape)
    # and the single-tensor route of the same operator agrees to bf16 rounding
    y1 = ops.conv2d(torch.cat([a, b], dim=-1), w, bias, 1, 1, relu=relu)
    assert float((y1.float() - y.float()).abs().max()) <= 2.0 ** -6 * sy


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cat_channels_matches_torch_cat(ops, dt):
    """ops.cat_channels == torch.cat(dim=-1) on NHWC tensors (the reference's torch.cat(dim=1), map_encoder.py:104,110,
    mg_map_policy.py:99), bit for bit, gradients included (they are the channel slices of dy)."""
    torch.manual_seed(5)
    a = torch.randn(3, 12, 12, 64, device="cuda").to(dt).requires_grad_(True)
    b = torch.randn(3, 12, 12, 128, device="cuda").to(dt).requires_grad_(True)
    y = ops.cat_channels(a, b)
    assert torch.equal(y, torch.cat([a, b], dim=-1))
    gy = torch.randn_like(y)
    y.backward(gy)
    assert torch.equal(a.grad, gy[..., :64]) and torch.equal(b.grad, gy[..., 64:])
    # a channel run that is not a multiple of 16 bytes takes the torch.cat route
    c = torch.randn(2, 5, 5, 3, device="cuda").to(dt)
    assert torch.equal(ops.cat_channels(c, c), torch.cat([c, c], dim=-1))


@pytest.mark.gpu
def test_bn_reductions_on_two_streams_do_not_share_scratch(ops):
    """Train-mode BatchNorm launches on two streams at the same time (the decoder's side-stream branch beside the main
    branch) must each get their own reduction scratch: every concurrent result is bit-identical to the result of the same
    call made alone (a scratch buffer shared between streams made the statistics of one launch leak into the other)."""
    torch.manual_seed(9)
    C = 64
    xs = [torch.randn(64, 48, 48, C, device="cuda").bfloat16() * (1 + i) + i for i in range(2)]
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")

    def call(x):
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        return ops.bn_act(x, g, b, rm, rv, True, True, None, 0.1, 1e-5)

    alone = [call(x).clone() for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    big = torch.randn(4096, 4096, device="cuda")
    for rep in range(30):
        outs = []
        for s in streams:       # back both queues up behind ~1 ms of work, so that the BN launches of the two streams
            with torch.cuda.stream(s):   # become runnable together instead of one call after the other
                big @ big
        for s, x in zip(streams, xs):
            with torch.cuda.stream(s):
                outs.append(call(x))
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(outs[i], alone[i]), (rep, i)
