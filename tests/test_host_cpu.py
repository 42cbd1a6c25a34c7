"""CPU: host logic and the C-ABI surface (no GPU compute)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import cases, policy_ref
from util import T, golden, make_params, state_dict_values, state_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Box:
    shape = (2,)


def _policy(num_proc=2):
    from wsmgmap.config import default_model_config
    from wsmgmap.models.policy import BasePolicy, CMAPolicy
    assert CMAPolicy is BasePolicy
    return BasePolicy(None, _Box(), default_model_config(num_proc=num_proc))


def test_abi_library_exports_every_declared_symbol():
    from wsmgmap import _abi
    header = open(os.path.join(ROOT, "include", "wsmgmap.h")).read()
    declared = set(re.findall(r"\b(wsmg_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 26
    L = _abi.lib()  # raises if the .so is missing: the HIP library is mandatory
    for name in declared:
        assert hasattr(L, name), f"libwsmgmap.so does not export {name}"
    assert set(_abi.exported_names()) == declared
    assert L.wsmg_abi_version() == 1
    assert b"gfx950" in L.wsmg_build_info()


def test_ops_refuse_cpu_tensors_no_fallback():
    from wsmgmap import _abi, ops
    x = torch.zeros(1, 4, 4, 32)
    w = torch.zeros(32, 32, 1, 1)
    with pytest.raises(_abi.WsmgError):
        ops.conv2d(x, w, None, 1, 0)
    with pytest.raises(_abi.WsmgError):
        ops.attention(torch.zeros(1, 256), torch.zeros(1, 4, 256), torch.zeros(1, 4, 256))
    with pytest.raises(_abi.WsmgError):
        ops.bev_index(torch.zeros(1, 256, 256), 224, 224, 100)


def test_state_dict_contract_matches_reference():
    pol = _policy()
    sd = pol.state_dict()
    spec = state_spec()
    assert set(sd) == set(spec)
    for k, v in spec.items():
        assert list(sd[k].shape) == v["shape"], k
        assert str(sd[k].dtype).replace("torch.", "") == v["dtype"], k
    req = {k: p.requires_grad for k, p in pol.named_parameters(remove_duplicate=False)}
    for k, v in spec.items():
        if v["param"] and k != "net.instruction_encoder.embedding_layer.weight":
            assert req[k] == v["trainable"], k
    pol.load_state_dict(state_dict_values(), strict=True)
    assert sum(p.numel() for p in pol.parameters() if p.requires_grad) == 19762425
    # aliases of the resnet18 stem are one tensor under two names, as in the reference
    assert pol.net.map_decoder.layer0[0].weight is pol.net.map_decoder.base_model.conv1.weight


def test_map_state_follows_module_moves_and_reassignment():
    pol = _policy(num_proc=3)
    m = pol.net.rgb_mapping_module
    assert tuple(m.full_global_map.shape) == (3, 240, 240, 64)
    assert "full_global_map" not in "".join(pol.state_dict().keys())
    pol.double()
    assert m.full_global_map.dtype == torch.float64
    m.full_global_map = torch.zeros(2, 240, 240, 64)
    assert pol.net.rgb_mapping_module.full_global_map.shape[0] == 2


def test_aux_losses_registry_semantics():
    from wsmgmap.common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    AuxLosses.clear()
    with pytest.raises(AssertionError):
        AuxLosses.register_loss("a", torch.ones(3))
    AuxLosses.activate()
    AuxLosses.register_loss("a", torch.tensor([1.0, 2.0, 3.0]), 0.5)
    AuxLosses.register_loss("b", torch.tensor([4.0, 0.0, 2.0]))
    with pytest.raises(AssertionError):
        AuxLosses.register_loss("a", torch.ones(3))
    total = AuxLosses.reduce(torch.tensor([True, False, True]))
    assert abs(float(total) - (0.5 * 2.0 + 3.0)) < 1e-6
    assert torch.equal(AuxLosses.get_loss("b"), torch.tensor([4.0, 0.0, 2.0]))
    AuxLosses.clear()
    AuxLosses.deactivate()


def test_diag_gaussian_matches_oracle_head():
    from wsmgmap.common.distributions import DiagGaussian
    d = DiagGaussian(512, 2)
    assert set(d.state_dict()) == {"fc_mean.weight", "fc_mean.bias", "logstd._bias"}
    x = T(np.random.RandomState(0).randn(4, 512).astype(np.float32))
    dist = d(x)
    assert torch.equal(dist.mode(), dist.mean)
    lp = dist.log_probs(dist.mean)
    ref = torch.distributions.Normal(dist.mean, torch.ones_like(dist.mean)).log_prob(dist.mean).sum(-1)
    assert torch.allclose(lp, ref)


def test_masked_gru_matches_oracle_semantics():
    """stock sequence form (split at restarts) == per-step h*mask form, incl. a mid-sequence restart."""
    from wsmgmap.models.rnn_state_encoder import RNNStateEncoder
    torch.manual_seed(0)
    enc = RNNStateEncoder(16, 8)
    Tn, N = 6, 3
    x = torch.randn(Tn * N, 16)
    masks = torch.ones(Tn, N)
    masks[0] = 0
    masks[3, 1] = 0
    h0 = torch.randn(1, N, 8)
    y, h = enc.forward_stock(x, h0, masks.view(-1, 1))
    P = {"e.rnn." + k: v for k, v in enc.rnn.state_dict().items()}
    yr, hr = policy_ref.masked_gru(P, "e", x, h0, masks.view(-1, 1))
    assert torch.allclose(y, yr, atol=1e-6) and torch.allclose(h, hr, atol=1e-6)
    assert RNNStateEncoder.restart_steps(masks.view(-1, 1), N) == [3]
    y1, h1 = enc.forward_stock(x[:N], h0, masks[0].view(-1, 1))  # single-step form
    assert torch.allclose(y1, y[:N], atol=1e-6)
    from wsmgmap import _abi
    with pytest.raises(_abi.WsmgError):  # the product forward is the HIP kernel: no CPU path
        enc(x, h0, masks.view(-1, 1))


def test_instruction_encoder_dedup_matches_oracle():
    pol = _policy()
    pol.load_state_dict(state_dict_values(), strict=True)
    enc = pol.net.instruction_encoder
    obs_np, *_ = cases.update_inputs(4, 2)
    instr = T(obs_np["instruction"])
    hid, mask = enc({"instruction": instr})
    P = make_params(grad=False)
    hr, mr = policy_ref.instruction_encoder(P, instr)
    assert hid.shape == hr.shape and torch.equal(mask, mr)
    assert torch.allclose(hid, hr, atol=1e-6)
    u, m, inv = enc.encode_unique(instr, stock=True)
    assert u.shape[0] == 2 and inv.shape[0] == 8


# ------------------------------------------------------------------ trajectory cache host logic
def _data_gold():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_g7_data.npz"))


def _dsha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_data_collate_fn_matches_reference_golden():
    from oracle import data_cases as dc
    from wsmgmap.data import collate_fn
    g = _data_gold()
    for tag, lengths in (("rag", dc.COLLATE_LENGTHS), ("long", dc.LONG_LENGTHS)):
        batch = [dc.episode(100 + i, n) + (torch.ones(n),) for i, n in enumerate(lengths)]
        ob, prev, masks, corr, wts = collate_fn(batch)
        for k, v in ob.items():
            assert list(v.shape) == g[f"g6_{tag}_obs_{k}_shape"].tolist() and str(v.dtype) == str(g[f"g6_{tag}_obs_{k}_dtype"])
            assert _dsha(v.float().numpy()) == str(g[f"g6_{tag}_obs_{k}_sha"]), (tag, k)
        for name, v in (("prev", prev), ("masks", masks), ("corr", corr), ("wts", wts)):
            assert _dsha(v.float().numpy()) == str(g[f"g6_{tag}_{name}_sha"]), (tag, name)


def test_data_dataset_order_codec_and_sharding_match_reference_golden():
    """TrajectoryDataset over an in-memory store of packed records reproduces IWTrajectoryDataset's yield order
    (rank / worker shard, block shuffle, length-sorted preload, weights) for the same `random` seed; records written
    by the oracle's restatement of the msgpack_numpy format are readable and vice versa."""
    import random
    import types
    from oracle import data_cases as dc, data_ref as dr
    from wsmgmap.data import TrajectoryDataset, pack_record, unpack_record, block_shuffle, change_data_type
    g = _data_gold()
    store = {}
    for i, n in enumerate(dc.DATASET_LENGTHS):
        obs, prev, oracle = dc.episode(1000 + i, n)
        store[i] = pack_record(obs, prev, oracle) if i % 2 else dr.pack_record(obs, prev, oracle)   # both writers
    for ci, (world, rank, nworkers, wid, bs, seed) in enumerate(dc.DATASET_CASES):
        ds = TrajectoryDataset(store.__getitem__, len(store), use_iw=True, inflection_weight_coef=3.2, batch_size=bs,
                               rank=rank, world_size=world)
        info = None if nworkers == 0 else types.SimpleNamespace(num_workers=nworkers, id=wid)
        old = torch.utils.data.get_worker_info
        torch.utils.data.get_worker_info = lambda info=info: info
        try:
            random.seed(seed)
            lens, wsum, first = [], [], []
            for obs, prev, oracle, w in ds:
                lens.append(len(prev)); wsum.append(float(w.sum())); first.append(float(prev[0, 0]))
            assert lens == g[f"g7_{ci}_yield_lengths"].tolist()
            assert wsum == g[f"g7_{ci}_weight_sums"].tolist() and first == g[f"g7_{ci}_first_prev"].tolist()
            assert ds.loaded_indices == g[f"g7_{ci}_order"].tolist() and len(ds) == int(g[f"g7_{ci}_len"])
        finally:
            torch.utils.data.get_worker_info = old
    random.seed(3)
    assert block_shuffle(list(range(17)), 4) == g["g7_block_shuffle"].tolist()
    obs, prev, oracle = dc.episode(7, 6)
    o2, p2, a2 = dr.unpack_record(pack_record(obs, prev, oracle))
    assert all(o2[k].dtype == obs[k].dtype and np.array_equal(o2[k], obs[k]) for k in obs) and np.array_equal(p2, prev)
    rec = unpack_record(dr.pack_record(dict(obs, ep_id=np.int64(3)), prev, oracle))
    assert "ep_id" not in rec[0] and np.array_equal(rec[2], oracle)
    assert change_data_type({"rgb": np.ones((2, 3), np.float32), "gps": np.ones(2, np.float32)})["rgb"].dtype == np.uint8


# ----------------------------------------------------------------------------- checkpoint / resume contract
def _policy_with_golden_fill():
    from wsmgmap.config import default_model_config
    from wsmgmap.models.policy import BasePolicy

    class Box:
        shape = (2,)
    pol = BasePolicy(None, Box(), default_model_config(num_proc=2))
    pol.load_state_dict(state_dict_values(), strict=True)
    return pol


def test_checkpoint_file_matches_reference_golden(tmp_path):
    """tests/golden/g8_ckpt.json was captured by running the UNMODIFIED reference save_checkpoint / resume_dagger /
    load-for-finetune (common_trainer.py:71-76,91-139) on the hash-filled reference policy: same top-level keys, same
    state_dict keys and values (sha-256 of a sample of tensors), same resume decisions on the same folder layouts."""
    import hashlib
    import json
    import time
    import numpy as np
    from wsmgmap import checkpoint as ck
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_ckpt.json")))
    pol = _policy_with_golden_fill()
    folder = str(tmp_path / "ckpts")
    ck.save_checkpoint(pol, folder, "ckpt.0.pth", config={"any": "object"}, extra_state={"dagger_it": 2})
    time.sleep(0.05)
    path = ck.save_checkpoint(pol, folder, "ckpt.3.pth", config={"any": "object"}, extra_state={"dagger_it": 2})
    d = ck.load_checkpoint(path)
    assert sorted(d.keys()) == g["top_level_keys"] and d["extra_state"] == g["extra_state"]
    assert len(d["state_dict"]) == g["n_state_keys"]
    for k, h in g["state_sha256"].items():
        assert hashlib.sha256(np.ascontiguousarray(d["state_dict"][k].numpy()).tobytes()).hexdigest() == h, k
    fresh = _policy_with_golden_fill()
    with torch.no_grad():
        fresh.prog_pred.bias.add_(1.0)
    it, ep, rep = ck.resume_dagger(fresh, folder, epochs=g["cases"][0]["epochs"])
    assert [it, ep] == g["cases"][0]["result"] and not rep.missing_keys and not rep.unexpected_keys
    assert torch.equal(fresh.prog_pred.bias, pol.prog_pred.bias)
    time.sleep(0.05)
    ck.save_checkpoint(pol, folder, "ckpt.1.pth", extra_state={"dagger_it": 5})     # newest by mtime, not by name
    it, ep, _ = ck.resume_dagger(fresh, folder, epochs=4)
    assert [it, ep] == g["cases"][1]["result"]
    assert ck.resume_dagger(fresh, str(tmp_path / "empty"), epochs=4)[:2] == (0, 0)
    # load-for-finetune through a DDP-style wrapper: 'module.' prefix, strict=False, same report as the reference
    wrapper = torch.nn.Module()
    wrapper.module = fresh
    sd = dict(d["state_dict"])
    sd["not_a_key"] = torch.zeros(1)
    del sd["prog_pred.bias"]
    torch.save({"state_dict": sd, "config": None}, str(tmp_path / "ft.pth"))
    msg = ck.load_pretrained(wrapper, str(tmp_path / "ft.pth"))
    assert list(msg.missing_keys) == g["finetune_missing"] and list(msg.unexpected_keys) == g["finetune_unexpected"]


def test_checkpoint_with_unimportable_config_class_still_loads(tmp_path):
    """The authors' checkpoints pickle a yacs/habitat Config; neither package exists here or on the GPU box."""
    import pickle
    import sys
    import types
    from wsmgmap import checkpoint as ck
    mod = types.ModuleType("fake_yacs_pkg")

    CfgNode = type("CfgNode", (dict,), {"__module__": "fake_yacs_pkg", "__qualname__": "CfgNode"})
    mod.CfgNode = CfgNode
    sys.modules["fake_yacs_pkg"] = mod
    try:
        torch.save({"state_dict": {"w": torch.arange(3.0)}, "config": CfgNode(LR=1e-3), "extra_state": {"dagger_it": 1}},
                   str(tmp_path / "c.pth"))
    finally:
        del sys.modules["fake_yacs_pkg"]
    d = ck.load_checkpoint(str(tmp_path / "c.pth"))
    assert torch.equal(d["state_dict"]["w"], torch.arange(3.0)) and d["extra_state"]["dagger_it"] == 1
    assert d["config"]["LR"] == 1e-3
    # the full unpickler is a caller's choice, and only the safe loader's REJECTION leads to it
    with pytest.raises(pickle.UnpicklingError):
        ck.load_checkpoint(str(tmp_path / "c.pth"), allow_pickle=False)
    (tmp_path / "corrupt.pth").write_bytes(b"this is not a checkpoint")
    with pytest.raises(Exception) as ei:
        ck.load_checkpoint(str(tmp_path / "corrupt.pth"))
    assert not isinstance(ei.value, AttributeError)
    # a stand-in created through REDUCE (called with constructor arguments) keeps them instead of failing obscurely
    o = ck._LenientUnpickler.find_class(ck._LenientUnpickler.__new__(ck._LenientUnpickler), "no_such_pkg_xyz", "Thing")(1, 2, key=3)
    assert o["__args__"] == (1, 2) and o["__kwargs__"] == {"key": 3}


# ----------------------------------------------------------------------------- depth branch from raw depth
def test_depth_encoder_oracle_and_module_vs_golden():
    """g9: the reference's VlnResnetDepthEncoder.forward from raw depth (around the restated third-party backbone).  The
    oracle's functional restatement reproduces it, and so does the product module on its stock-operator (float32) path."""
    from oracle import cases, policy_ref
    from wsmgmap.models.encoders.resnet_encoders import VlnResnetDepthEncoder
    g = golden("g9_depth.npz")
    obs_np, _ = cases.act_inputs(0, B=2, tag="g9")
    P = make_params(grad=False)
    depth = T(obs_np["depth"])
    with torch.no_grad():
        feat = policy_ref.ddppo_resnet50(P, depth)
    assert float((feat - T(g["feat"])).abs().max()) <= 1e-5
    enc = VlnResnetDepthEncoder(None)
    pre = "net.depth_encoder."
    enc.load_state_dict({k[len(pre):]: v for k, v in state_dict_values().items() if k.startswith(pre)}, strict=True)
    assert sum(p.numel() for p in enc.visual_encoder.parameters()) == int(g["n_params"])
    assert not any(p.requires_grad for p in enc.visual_encoder.parameters())
    with torch.no_grad():
        out = enc({"depth": depth})
    assert tuple(out.shape) == (2, 192, 4, 4)
    assert float((out[:, :128] - T(g["feat"])).abs().max()) <= 1e-5
    assert float((out[:, ::7] - T(g["out_sample"])).abs().max()) <= 1e-5


def test_adam_refuses_anything_but_float32_cuda_parameters():
    """wsmgmap.optim.Adam has no CPU path: stepping a CPU parameter raises (the product never falls back), unsupported
    variants are refused at construction, and its state_dict has torch.optim.Adam's layout."""
    import torch
    from wsmgmap import _abi, optim
    p = torch.nn.Parameter(torch.zeros(5))
    opt = optim.Adam([p], lr=1e-3)
    p.grad = torch.ones(5)
    with pytest.raises(_abi.WsmgError):
        opt.step()
    with pytest.raises(ValueError):
        optim.Adam([p], amsgrad=True)
    with pytest.raises(ValueError):
        optim.Adam([p], lr=-1.0)
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(5))], lr=1e-3).state_dict()
    mine = opt.state_dict()["param_groups"][0]
    assert set(mine) >= {"lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"} and set(mine) <= set(ref["param_groups"][0])


def test_action_sample_equals_torch_normal_sample():
    """ActionNormal.sample() (no host read-back, capturable) draws exactly what torch.distributions.Normal.sample() draws
    from the same generator state (the reference samples with the stock class: distributions.py:21-29, policy.py:50-53)."""
    import torch
    from wsmgmap.common.distributions import ActionNormal
    loc, scale = torch.randn(37, 2), torch.rand(37, 2) + 0.1
    torch.manual_seed(123)
    a = ActionNormal(loc, scale, validate_args=False).sample()
    torch.manual_seed(123)
    b = torch.distributions.Normal(loc, scale).sample()
    assert torch.equal(a, b)
    torch.manual_seed(5)
    a3 = ActionNormal(loc, scale, validate_args=False).sample((3,))
    torch.manual_seed(5)
    b3 = torch.distributions.Normal(loc, scale).sample((3,))
    assert torch.equal(a3, b3)


def test_fold_cache_folds_eval_batchnorm_and_refreshes_in_place():
    """encoders.map_encoder.FoldCache (host logic of the rollout route): conv + eval-mode BatchNorm folded into an OHWI bf16
    weight and a float32 bias reproduces bn(conv(x)) to bf16 rounding of the weights; channel padding is zero; a parameter
    or running-statistics change is picked up by refresh() in the SAME storage (a captured graph keeps reading it)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ws-mgmap_amd"))
    import torch.nn as nn
    import torch.nn.functional as F
    from wsmgmap.models.encoders.map_encoder import FoldCache
    torch.manual_seed(0)
    conv, bn = nn.Conv2d(8, 16, 3, padding=1), nn.BatchNorm2d(16).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
    x = torch.randn(2, 8, 10, 10)
    cache = FoldCache()

    def check(w, b):
        with torch.no_grad():
            ref = bn(conv(x))
            got = F.conv2d(x, w[:16, :, :, :8].float().permute(0, 3, 1, 2), b[:16], padding=1)
        assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    w, b = cache.get(None, None, bn, 8, 0, owner=conv)
    assert w.dtype == torch.bfloat16 and tuple(w.shape) == (16, 3, 3, 8) and b.dtype == torch.float32
    check(w, b)
    assert cache.get(None, None, bn, 8, 0, owner=conv)[0] is w and cache.refresh() == 0
    ptr = w.data_ptr()
    with torch.no_grad():
        conv.weight.mul_(1.5)
        bn.running_var.add_(0.25)
    assert cache.refresh() == 1 and cache.refresh() == 0
    w2, b2 = cache.get(None, None, bn, 8, 0, owner=conv)
    assert w2.data_ptr() == ptr
    check(w2, b2)
    wp, bp = cache.get(None, None, bn, 32, 32, owner=conv)          # channel padding of the engine: zeros
    assert tuple(wp.shape) == (32, 3, 3, 32) and tuple(bp.shape) == (32,)
    assert float(wp[16:].abs().max()) == 0 and float(wp[:, :, :, 8:].abs().max()) == 0 and float(bp[16:].abs().max()) == 0
    check(wp, bp)
    lin = nn.Conv2d(8, 4, 1)                                         # no BatchNorm: the bias is a copy, not the parameter
    wl, bl = cache.get(None, None, None, 8, 0, owner=lin)
    assert bl.data_ptr() != lin.bias.data_ptr() and torch.equal(bl, lin.bias.detach())


def test_rnn_kernels_are_built_without_packed_fp32_math():
    """DESIGN §4 "RNN kernels: wrong values beside a co-resident conv wave": the persistent GRU / LSTM kernels returned wrong values
    with `v_pk_fma_f32` in their dependent FMA chains whenever an MFMA-heavy wave shared the SIMD (177 mismatching tensors in 60
    loaded repeats; 0 in 300 without them).  The build turns packed fp32 math off for wsmg_rnn.hip (csrc/Makefile: EXTRA_wsmg_rnn):
    compile that file for gfx950 exactly as the Makefile does and look at the ISA."""
    import re
    import shutil
    import subprocess
    import tempfile
    csrc = os.path.join(ROOT, "ws-mgmap_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    m = re.search(r"^EXTRA_wsmg_rnn\s*:=\s*(.+)$", mk, re.M)
    assert m, "csrc/Makefile no longer sets EXTRA_wsmg_rnn"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rnn.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), *m.group(1).split(),
                        "--cuda-device-only", "-S", os.path.join(csrc, "wsmg_rnn.hip"), "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        isa = open(out).read()
    packed = re.findall(r"\bv_pk_(?:fma|mul|add)_f32", isa)
    assert not packed, f"{len(packed)} packed fp32 instructions in the RNN kernels"
    assert len(re.findall(r"\bv_fmac?_f32", isa)) > 1000      # the scalar FMAs are there instead


def test_feeder_ring_refuses_what_the_host_cannot_hold():
    """A 96-worker ring of 700 MiB batches (212 GB page-locked + 96 processes) took a GPU box down in round 3: the ring is refused up
    front when it exceeds 40 % of min(MemAvailable, control-group limit - usage) — before any process or shared tensor exists."""
    from wsmgmap.data import feeder
    avail = feeder._host_memory_available()
    assert avail is None or (isinstance(avail, int) and avail > 0)
    if avail is None:
        pytest.skip("host memory not readable here")
    fd = feeder.DeviceFeeder(dataset=[], batch_size=2, device="cpu", num_workers=4, slot_bytes=avail // 4, slots_per_worker=2)
    with pytest.raises(RuntimeError, match="host memory|/dev/shm"):
        next(iter(fd))


class _BlobStore:
    def __init__(self, blobs):
        self.blobs = blobs

    def __call__(self, i):
        return self.blobs[i]


def _tiny_dataset(batch_size):
    from oracle import data_cases as dc
    from wsmgmap.data import TrajectoryDataset, pack_record
    store = _BlobStore([pack_record(*dc.episode(1000 + i, n)) for i, n in enumerate(dc.DATASET_LENGTHS)])
    return TrajectoryDataset(store, len(store.blobs), batch_size=batch_size, rank=0, world_size=1)


def test_feeder_probe_bounds_the_slot_and_leaves_the_random_state_alone():
    """the slot size comes from a probe batch — 1.5 x its packed bytes, never above batch_size x 200 steps of the
    probe's per-step bytes — and the probe's block shuffle / tie-breaks must not advance the parent's `random` state."""
    import random
    from wsmgmap.data import feeder
    from wsmgmap.data.collate import plan_batch
    fd = feeder.DeviceFeeder(_tiny_dataset(4), 4, device="cpu", num_workers=2)
    random.seed(77)
    before = random.getstate()
    nbytes = fd._probe_slot_bytes()
    assert random.getstate() == before
    from oracle import data_cases as dc
    # prev [2] f32 + oracle [2] f32 + weights [] f32 = 20 bytes on top of the sensors
    per_step = sum(int(np.prod(shape)) * np.dtype(dt).itemsize for dt, shape in dc.SENSORS.values()) + 20
    assert nbytes <= (per_step * 200 + 16 * (len(dc.SENSORS) + 3)) * 4 + 4096
    ds = _tiny_dataset(4)
    ds._worker_override = (2, 0)
    random.seed(77)
    it = iter(ds)
    _, meta = plan_batch([next(it) for _ in range(4)])
    assert nbytes == int(meta["total"] * 1.5) + 4096


def test_feeder_worker_sends_an_oversize_batch_outside_the_ring_and_waits_for_the_ack():
    """a batch larger than a ring slot used to raise inside the worker and abort the epoch.  Now it travels as its own
    shared block; the worker goes on only after the consumer's "ack", and slot ids that come back meanwhile are kept."""
    import queue
    import threading
    from wsmgmap.data import feeder
    from wsmgmap.data.collate import plan_batch
    ds = _tiny_dataset(2)
    ds._worker_override = (1, 0)
    import random
    random.seed(5)
    sizes = []
    items = [x for x in ds]              # ONE __iter__ call, as in the worker
    for i in range(0, len(items) - 1, 2):
        sizes.append(plan_batch(items[i:i + 2])[1]["total"])
    cap = sorted(sizes)[len(sizes) // 2]                     # about half of the batches do not fit
    slots = [torch.empty(cap, dtype=torch.uint8) for _ in range(2)]
    free_q, ready_q = queue.Queue(), queue.Queue()
    free_q.put(0)
    free_q.put(1)
    nthreads = torch.get_num_threads()       # the worker is a process in production and pins ITS torch to one thread
    th = threading.Thread(target=feeder._ring_worker, args=(_tiny_dataset(2), 2, 0, 1, slots, free_q, ready_q, 5), daemon=True)
    th.start()
    got, big = [], 0
    while True:
        item = ready_q.get(timeout=60)
        if item == feeder._STOP:
            break
        assert item[0] != "__error__", item
        if item[0] == "__big__":
            _, meta, blk = item
            assert meta["total"] > cap and blk.numel() >= meta["total"]
            big += 1
            free_q.put("ack")
        else:
            sid, meta = item
            assert meta["total"] <= cap
            free_q.put(sid)
        got.append(meta["total"])
    th.join(timeout=10)
    torch.set_num_threads(nthreads)
    assert not th.is_alive() and big >= 1 and len(got) == len(sizes) and sorted(got) == sorted(sizes)


def test_feeder_worker_runs_epoch_after_epoch_when_the_ring_is_persistent():
    """Round 5: a persistent ring's decode worker does not end with its epoch — after its _STOP it waits for ("epoch", seed) on the
    queue its slots come back on (keeping the slot ids that arrive meanwhile), runs the next epoch with that seed exactly as a fresh
    worker started with it would, and ends on None."""
    import queue
    import threading
    from wsmgmap.data import feeder

    def run(seeds, persistent):
        slots = [torch.empty(1 << 20, dtype=torch.uint8) for _ in range(2)]
        free_q, ready_q = queue.Queue(), queue.Queue()
        free_q.put(0)
        free_q.put(1)
        th = threading.Thread(target=feeder._ring_worker, args=(_tiny_dataset(3), 2, 0, 1, slots, free_q, ready_q, seeds[0], persistent),
                              daemon=True)
        th.start()
        epochs = []
        for e, _ in enumerate(seeds):
            got = []
            while True:
                item = ready_q.get(timeout=60)
                if item == feeder._STOP:
                    break
                assert item[0] not in ("__error__", "__big__"), item
                sid, meta = item
                got.append((meta["total"], bytes(slots[sid][:64].numpy())))
                free_q.put(sid)
            epochs.append(got)
            if e + 1 < len(seeds):
                free_q.put(("epoch", seeds[e + 1]))
        free_q.put(None)
        th.join(timeout=10)
        assert not th.is_alive()
        return epochs

    nthreads = torch.get_num_threads()
    try:
        three = run([5, 11, 5], True)
        assert len(three) == 3 and all(len(e) > 0 for e in three)
        assert three[0] == three[2], "the same seed must give the same epoch"
        assert three[1] == run([11], False)[0], "an epoch of the standing worker differs from a fresh worker's with the same seed"
    finally:
        torch.set_num_threads(nthreads)


def test_recoded_raw_records_decode_to_the_same_arrays_and_collate_identically():
    """`tools/recode_cache.py` rewrites the reference's zlib(msgpack_numpy) values (dagger_trainer.py:336-343)
    once into an uncompressed layout; a recoded record must give the same arrays (dtype, shape, bytes) and the same batch from
    the reference-semantics collate, so the feeder streams it with no decode."""
    sys_path_tools = os.path.join(ROOT, "tools")
    import sys
    if sys_path_tools not in sys.path:
        sys.path.insert(0, sys_path_tools)
    import recode_cache
    from oracle import data_cases as dc
    from wsmgmap.data import collate_fn, is_raw_record, pack_record, pack_record_raw, recode_record, unpack_record
    eps = [dc.episode(300 + i, n) for i, n in enumerate(dc.COLLATE_LENGTHS + dc.LONG_LENGTHS)]
    blobs = [pack_record(*e) for e in eps]
    store_out = {}
    nin, nout = recode_cache.recode_store(lambda i: blobs[i], len(blobs), store_out.__setitem__, workers=2, chunk=2)
    assert nin == sum(map(len, blobs)) and nout == sum(len(v) for v in store_out.values()) and len(store_out) == len(blobs)
    for i, e in enumerate(eps):
        raw = store_out[i]
        assert is_raw_record(raw) and not is_raw_record(blobs[i])
        assert recode_record(raw) == raw and raw == pack_record_raw(*e)
        a, b = unpack_record(blobs[i]), unpack_record(memoryview(raw))       # (LMDB hands out memoryviews with buffers=True)
        assert list(a[0]) == list(b[0])
        for k in a[0]:
            assert a[0][k].dtype == b[0][k].dtype and a[0][k].shape == b[0][k].shape and a[0][k].tobytes() == b[0][k].tobytes(), k
        assert a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes()
    mk = lambda recs: [(r[0], r[1], r[2], torch.ones(len(r[1]))) for r in recs]    # noqa: E731
    want = collate_fn(mk([unpack_record(x) for x in blobs[:3]]))
    got = collate_fn(mk([unpack_record(store_out[i]) for i in range(3)]))
    assert all(torch.equal(want[0][k], got[0][k]) for k in want[0]) and all(torch.equal(x, y) for x, y in zip(want[1:], got[1:]))
    # the 200-step cap of the collate applies to raw records alike
    long_w = collate_fn(mk([unpack_record(x) for x in blobs[3:]]))
    long_g = collate_fn(mk([unpack_record(store_out[i]) for i in (3, 4)]))
    assert all(torch.equal(long_w[0][k], long_g[0][k]) for k in long_w[0])


def test_truncated_or_corrupt_raw_record_is_refused_with_a_clear_error():
    """the recoded record's stored index is validated against the blob before any zero-copy view is made."""
    import pytest
    from oracle import data_cases as dc
    from wsmgmap.data import pack_record_raw, unpack_record
    raw = pack_record_raw(*dc.episode(400, 5))
    assert len(unpack_record(raw)) == 3
    with pytest.raises(ValueError, match="truncated"):
        unpack_record(raw[:len(raw) // 2])
    with pytest.raises(ValueError, match="truncated"):
        unpack_record(raw[:14])


def test_sparse_ego_map_record_round_trips_bit_for_bit():
    """Round 5: the recoded cache may hold `rgb_ego_map` (dagger_trainer.py:336-343 stores the dense float16 map;
    55-80 % of it is zero) as presence bits + packed non-zeros (codec.sparse_pack_ego).  The expansion is bit-exact (-0.0 included),
    cutting an episode to its first steps works on the packed form, a raw record with the sparse map decodes to the same dense map, and
    the reference-semantics host collate gives the same batch from either record."""
    from wsmgmap.data import collate_fn, pack_record_raw, unpack_record, sparse_pack_ego, sparse_expand_ego, densify, has_sparse_ego
    rng = np.random.RandomState(3)
    eps = []
    for n in (6, 3):
        ego = np.maximum(rng.randn(n, 64, 7, 9), 0.5).astype(np.float16) - np.float16(0.5)
        ego[0, 5, 2, 3] = np.float16(-0.0)
        obs = {"rgb_ego_map": ego, "progress": rng.rand(n, 1).astype(np.float32), "instruction": rng.randint(0, 9, size=(n, 5)).astype(np.int64)}
        eps.append((obs, rng.randn(n, 2).astype(np.float32), rng.randn(n, 2).astype(np.float32)))
    for obs, prev, orc in eps:
        sp = sparse_pack_ego(obs["rgb_ego_map"])
        assert sp["rgb_ego_map__vals"].size < 0.5 * obs["rgb_ego_map"].size
        assert np.array_equal(sparse_expand_ego(sp).view(np.uint16), obs["rgb_ego_map"].view(np.uint16))
        assert np.array_equal(sparse_expand_ego(sp, 2).view(np.uint16), obs["rgb_ego_map"][:2].view(np.uint16))
        raw = pack_record_raw(obs, prev, orc, sparse_ego=True)
        assert len(raw) < 0.75 * len(pack_record_raw(obs, prev, orc))
        o2, p2, a2 = unpack_record(raw)
        assert has_sparse_ego(o2) and "rgb_ego_map" not in o2
        d2 = densify(o2)
        assert set(d2) == set(obs) and all(np.array_equal(np.asarray(d2[k]), obs[k]) for k in obs)
        assert np.array_equal(d2["rgb_ego_map"].view(np.uint16), obs["rgb_ego_map"].view(np.uint16))
    mk = lambda recs: [(r[0], r[1], r[2], torch.ones(len(r[1]))) for r in recs]    # noqa: E731
    want = collate_fn(mk(eps))
    got = collate_fn(mk([unpack_record(pack_record_raw(*e, sparse_ego=True)) for e in eps]))
    assert set(want[0]) == set(got[0]) and all(torch.equal(want[0][k], got[0][k]) for k in want[0])
    with pytest.raises(TypeError):
        sparse_pack_ego(np.zeros((2, 32, 4, 4), np.float16))


def test_every_environment_switch_is_declared():
    """the product reads at most 40 WSMG_* environment names, and every one is declared — the Python side in the one
    table of wsmgmap/debug.py, the library's WSMG_TUNE names in INTEGRATION.md section 3.1 (plus WSMG_LIB, the library path)."""
    import sys
    root = os.path.join(ROOT, "ws-mgmap_amd")
    py_names, lib_names = set(), set()
    for d, _, files in os.walk(root):
        for f in files:
            p = os.path.join(d, f)
            if f.endswith(".py"):
                src = open(p).read()
                if not p.endswith(os.path.join("wsmgmap", "debug.py")):
                    py_names |= set(re.findall(r"environ[^\n]*?[\"'](WSMG_[A-Z0-9_]+)[\"']", src))
            elif f.endswith((".hip", ".h")):
                src = open(p).read()
                lib_names |= set(re.findall(r"WSMG_TUNE\(\"(WSMG_[A-Z0-9_]+)\"", src))
                assert not re.findall(r"getenv\(\"", src), p       # the library reads the environment through WSMG_TUNE only
    sys.path.insert(0, root)
    from wsmgmap import debug
    table = {v for _, v, _, _, _ in debug._TABLE}
    assert py_names <= table | {"WSMG_LIB"}, py_names - table
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in lib_names | {"WSMG_LIB"}:
        assert "`%s`" % n in doc, n
    assert len(table | lib_names | {"WSMG_LIB"}) <= 40
