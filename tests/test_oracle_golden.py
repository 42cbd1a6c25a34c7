"""CPU: the oracle (oracle/) against the golden vectors captured from the unmodified
reference (tools/make_goldens.py).  Integer work is compared by hash (bit-exact); float
work with the tolerance written at each assert (the goldens come from torch CPU kernels,
which may differ in the last bits between hosts)."""
import numpy as np
import pytest
import torch

from oracle import bev_ref, cases, policy_ref
from util import NULL_GRAD, T, golden, make_params, sha


@pytest.mark.parametrize("name", list(cases.BEV_CASES))
def test_g1_bev_index_and_scatter_bit_exact(name):
    g = golden("g1_bev.npz")
    c = cases.bev_inputs(name)
    proj, lin, inv, x, y, v = bev_ref.project_to_ground(c["feat"], c["depth"], c["E"])
    assert sha(np.stack([x, y], 1).astype(np.int64)) == str(g[name + ".locs_sha"])
    assert sha(v[:, None].astype(np.uint8)) == str(g[name + ".valid_sha"])
    assert sha(lin.astype(np.int32)) == str(g[name + ".lin_idx_sha"])
    assert sha(inv.astype(np.uint8)) == str(g[name + ".invalid_sha"])
    assert sha(proj.astype(np.float32)) == str(g[name + ".proj_sha"])
    assert int((~inv).sum()) == int(g[name + ".n_valid"])
    np.testing.assert_array_equal(lin.reshape(-1)[::97], g[name + ".lin_idx_sample"])


def test_g1_edge_cases_present():
    """the fixture really contains the edge cases the domain has: an all-invalid image,
    out-of-range cells, and .5 ties in front of round()."""
    c = cases.bev_inputs("e100_c64_f256")
    proj, lin, inv, x, y, v = bev_ref.project_to_ground(c["feat"], c["depth"], c["E"])
    assert inv[-1].all() and np.all(proj[-1] == 0)
    assert ((x < 0) | (x >= 100)).any() and (y < 0).any()
    z = c["depth"][1, :, :, 0] * np.float32(10)
    frac = (-(z / np.float32(0.12)) + np.float32(49.5)) % 1
    assert (frac == 0.5).sum() > 100


def test_g2_map_sequence():
    g = golden("g2_mapseq.npz")
    m = bev_ref.MapperRef(2)
    for s in range(cases.MAP_SEQ["steps"]):
        c = cases.mapseq_inputs(s)
        ego = m.step(T(c["feat"]), T(c["depth"]), T(c["gps"]), T(c["compass"]), T(c["masks"]))
        d = cases.summarize(ego.numpy())
        # tolerance: bilinear weights move by a few ulp of a coordinate ~ 1e-5 * |feature| <= 2
        np.testing.assert_allclose(d["sample"], g[f"s{s}.ego.sample"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(ego[:, ::16, 40:56, 44:60].numpy(), g[f"s{s}.ego_patch"], atol=1e-4, rtol=0)
        assert abs(d["abssum"] - g[f"s{s}.ego.abssum"]) <= 1e-5 * g[f"s{s}.ego.abssum"]
        nnz = int((m.full_global_map != 0).sum())
        assert abs(nnz - int(g[f"s{s}.global_nnz"])) <= 64


def _run_update():
    P = make_params()
    ref = policy_ref.PolicyRef(P, num_proc=2)
    ref.aux_active = True
    obs_np, prev, masks, weights = cases.update_inputs(4, 2)
    obs = {k: T(v) for k, v in obs_np.items()}
    w = T(weights).view(4, 2)
    pred, aux, h, sem = ref.forward(obs, torch.zeros(2, 2, 512), T(prev), T(masks), w)
    loss, al = policy_ref.dagger_loss(pred, aux, obs["waypoint"], w)
    loss.backward()
    return P, ref, pred, aux, loss, h


def test_g3_update_forward_backward():
    g = golden("g3_update.npz")
    P, ref, pred, aux, loss, h = _run_update()
    np.testing.assert_allclose(pred.detach().numpy(), g["pred"], atol=2e-5, rtol=0)  # bar is 1e-4 on logits
    assert abs(float(aux) - float(g["aux_loss"])) < 2e-5
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    for n in ["prediction_monitor", "contrastive_monitor", "progress_monitor"]:
        np.testing.assert_allclose(ref.losses[n][0].detach().numpy(), g["aux." + n], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(ref.att_map_t_m.detach().numpy(), g["att_map_t_m"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(h.detach().numpy(), g["h_out"], atol=2e-5, rtol=0)
    for k in g.files:
        if k.startswith("bn."):
            np.testing.assert_allclose(P[k[3:]].detach().numpy(), g[k], atol=1e-5, rtol=1e-5)
    for i, n in enumerate(g["grad.names"]):
        n = str(n)
        if n in NULL_GRAD:
            continue
        gr = P[n].grad.numpy()
        nr = float(np.sqrt((gr.astype(np.float64) ** 2).sum()))
        assert abs(nr - g["grad.norm"][i]) <= 1e-3 * g["grad.norm"][i] + 1e-9, n
    # the 50 trainable tensors the reference never reaches stay without gradient
    assert {str(s) for s in g["grad.none"]} >= {"critic.fc.weight", "action_distribution.logstd._bias"}


@pytest.mark.parametrize("hw", [224, 256])
def test_g4_rollout_act(hw):
    g = golden("g4_act.npz")
    P = make_params(grad=False)
    ref = policy_ref.PolicyRef(P, num_proc=2)
    ref.train_mode = False
    h, prev = torch.zeros(2, 2, 512), torch.zeros(2, 2)
    with torch.no_grad():
        for step in range(3):
            obs_np, masks = cases.act_inputs(step, rgb_hw=hw)
            obs = {k: T(v) for k, v in obs_np.items()}
            p = f"r{hw}.s{step}"
            if step == 1:
                ref.update_map(obs, T(masks))
            else:
                value, action, logp, h = ref.act(obs, h, prev, T(masks))
                prev = action
                np.testing.assert_allclose(action.numpy(), g[p + ".action"], atol=5e-5, rtol=0)
                np.testing.assert_allclose(value.numpy(), g[p + ".value"], atol=5e-5, rtol=0)
                np.testing.assert_allclose(ref.prog.numpy(), g[p + ".prog"], atol=5e-5, rtol=0)
                np.testing.assert_allclose(logp.numpy(), g[p + ".logp"], atol=1e-5, rtol=0)
            e = cases.summarize(obs["rgb_ego_map"].numpy())
            np.testing.assert_allclose(e["sample"], g[p + ".ego.sample"], atol=1e-4, rtol=0)


def test_g5_attention():
    g = golden("g5_attn.npz")
    q, k, v, m = cases.attn_inputs()
    o, a = policy_ref.attn(T(q), T(k), T(v), T(m))
    np.testing.assert_allclose(o.numpy(), g["out"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(a.numpy(), g["attn"], atol=1e-7, rtol=1e-5)
    assert np.all(a.numpy()[1, 1:] == 0)  # fully masked tail gets exactly 0 weight
    o2, a2 = policy_ref.attn(T(q), T(k), T(v), None)
    np.testing.assert_allclose(o2.numpy(), g["out_nomask"], atol=1e-6, rtol=1e-5)


def test_g5f_attention_on_e4m3_representable_inputs():
    """g5f: the reference's `_attn` at BASELINE configs[4]'s size (B = 64, L = 160) on inputs that are exactly OCP e4m3 numbers
    (oracle/cases.py::attn_fp8_inputs).  The oracle's float32 attention reproduces it; the inputs survive the e4m3 round trip
    bit for bit (which is what makes this golden a pin for the fp8-storage kernels); and the fp8 oracle's float64 formula on the
    codes agrees with it to float32 accuracy."""
    from oracle import attn_fp8_ref as ar
    g = golden("g5f_attn_fp8.npz")
    c = cases.attn_fp8_inputs()
    for name in ("q", "k", "v"):
        sc = c[name + "_scale"]
        assert np.array_equal(ar.dequantize_e4m3(ar.quantize_e4m3(c[name], sc), sc).astype(np.float32), c[name]), name
    inv = c["inverse"]
    k = np.ascontiguousarray(c["k"][inv].transpose(0, 2, 1))
    v = np.ascontiguousarray(c["v"][inv].transpose(0, 2, 1))
    mask = np.arange(k.shape[2])[None, :] >= c["lengths"][inv][:, None]
    o, a = policy_ref.attn(T(c["q"]), T(k), T(v), T(mask))
    np.testing.assert_allclose(o.numpy(), g["out"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(a.numpy(), g["attn"], atol=1e-7, rtol=1e-5)
    # float64 evaluation of the same formula (what the fp8 kernels are held against between goldens)
    lg = (np.einsum("bc,blc->bl", c["q"].astype(np.float64), c["k"][inv].astype(np.float64)) - 1e8 * mask) / 16
    p = np.exp(lg - lg.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
    assert np.abs(p - g["attn"]).max() < 5e-6 and np.abs(np.einsum("bl,blc->bc", p, c["v"][inv].astype(np.float64)) - g["out"]).max() < 5e-6


def test_fp8_oracle_matches_f32_formula_on_representable_inputs():
    """oracle/attn_fp8_ref.py: on inputs that e4m3 represents exactly, the fp8 restatement equals the plain formula
    of g5 (mg_map_policy.py:173-178) and the codes round-trip; quantisation saturates at +-448."""
    from oracle import attn_fp8_ref as ar
    rng = np.random.RandomState(1)
    x = rng.choice(np.array([-2.0, -1.5, -0.5, 0.0, 0.25, 0.875, 1.0, 3.5], dtype=np.float32), size=(3, 11, 256))
    q = rng.randn(3, 256).astype(np.float32)
    w = (rng.randn(256, 256) / 16).astype(np.float32)
    b = rng.randn(256).astype(np.float32)
    lengths = np.array([11, 4, 1])
    codes = ar.quantize_e4m3(x, 1.0)
    assert np.array_equal(ar.dequantize_e4m3(codes, 1.0), x.astype(np.float64))
    out, attn = ar.attn_fp8(q, w, b, codes, 1.0, lengths, 1 / 16)
    k = x.astype(np.float64) @ w.astype(np.float64).T + b
    lg = (np.einsum("bc,blc->bl", q.astype(np.float64), k) - 1e8 * (np.arange(11)[None] >= lengths[:, None])) / 16
    a = np.exp(lg - lg.max(1, keepdims=True)); a /= a.sum(1, keepdims=True)
    assert np.abs(attn - a).max() < 1e-12 and np.abs(out - np.einsum("bl,blc->bc", a, x)).max() < 1e-12
    assert ar.dequantize_e4m3(ar.quantize_e4m3(np.array([1e6, -1e6, 460.0, 0.0], np.float32), 1.0), 1.0).tolist() == [448.0, -448.0, 448.0, 0.0]


# ------------------------------------------------------------------ trajectory cache
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_data_oracle_collate_matches_reference_golden():
    """oracle/data_ref.collate == dagger_trainer.py collate_fn (g6: ragged batch, and the 200-step cap)."""
    from oracle import data_cases as dc, data_ref as dr
    g = np.load(os.path.join(GOLD, "g6_g7_data.npz"))
    for tag, lengths in (("rag", dc.COLLATE_LENGTHS), ("long", dc.LONG_LENGTHS)):
        batch = []
        for i, n in enumerate(lengths):
            obs, prev, oracle = dc.episode(100 + i, n)
            batch.append(({k: torch.from_numpy(v.copy()) for k, v in obs.items()}, torch.from_numpy(prev.copy()),
                          torch.from_numpy(oracle.copy()), torch.ones(n)))
        ob, prev, masks, corr, wts = dr.collate(batch)
        for k, v in ob.items():
            assert list(v.shape) == g[f"g6_{tag}_obs_{k}_shape"].tolist() and str(v.dtype) == str(g[f"g6_{tag}_obs_{k}_dtype"])
            assert _sha(v.float().numpy()) == str(g[f"g6_{tag}_obs_{k}_sha"]), (tag, k)
        for name, v in (("prev", prev), ("masks", masks), ("corr", corr), ("wts", wts)):
            assert _sha(v.float().numpy()) == str(g[f"g6_{tag}_{name}_sha"]), (tag, name)


def test_data_oracle_dataset_order_matches_reference_golden():
    """Sharding, block shuffle and length-sorted preload order == IWTrajectoryDataset (g7), same `random` stream."""
    import random
    from oracle import data_cases as dc, data_ref as dr
    g = np.load(os.path.join(GOLD, "g6_g7_data.npz"))
    for ci, (world, rank, nworkers, wid, bs, seed) in enumerate(dc.DATASET_CASES):
        random.seed(seed)
        order = list(dr.dataset_order(dc.DATASET_LENGTHS, rank, world, nworkers, wid, bs))
        assert [dc.DATASET_LENGTHS[i] for i in order] == g[f"g7_{ci}_yield_lengths"].tolist()
        assert sorted(order) == sorted(g[f"g7_{ci}_order"].tolist())
        first = [float(dc.episode(1000 + i, dc.DATASET_LENGTHS[i])[1][0, 0]) for i in order]
        assert first == g[f"g7_{ci}_first_prev"].tolist()
        assert dr.shard_range(len(dc.DATASET_LENGTHS), rank, world, nworkers, wid)[2] == int(g[f"g7_{ci}_len"])
    random.seed(3)
    assert dr.block_shuffle(list(range(17)), 4) == g["g7_block_shuffle"].tolist()


def test_data_oracle_codec_round_trip_and_wire_format():
    from oracle import data_cases as dc, data_ref as dr
    import msgpack, zlib
    obs, prev, oracle = dc.episode(7, 6)
    blob = dr.pack_record(obs, prev, oracle)
    o2, p2, a2 = dr.unpack_record(blob)
    assert set(o2) == set(obs) and all(o2[k].dtype == obs[k].dtype and np.array_equal(o2[k], obs[k]) for k in obs)
    assert np.array_equal(p2, prev) and np.array_equal(a2, oracle)
    raw = msgpack.unpackb(zlib.decompress(blob), raw=False, strict_map_key=False)   # msgpack_numpy's field names
    e = raw[0]["rgb_ego_map"]
    assert e[b"nd"] is True and e[b"type"] == "<f2" and e[b"shape"] == [6, 4, 5, 5] and len(e[b"data"]) == 6 * 100 * 2
    assert dr.change_data_type({"gt_semantic_map": np.ones((2, 3), np.float32)})["gt_semantic_map"].dtype == np.int64
