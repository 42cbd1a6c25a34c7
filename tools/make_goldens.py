#!/usr/bin/env python3
"""Capture golden vectors by running the UNMODIFIED reference (/root/reference) on CPU.

Runs only in the build container (the reference does not exist on the GPU box).  It
imports the reference's own files under the sys.modules stubs of tools/ref_stubs.py,
feeds them the seeded inputs of oracle/cases.py with weights from oracle/detfill.py, and
writes small .npz fixtures (data only: inputs are re-derivable, outputs are digests /
small tensors) to tests/golden/.

    python tools/make_goldens.py [g1 g2 g3 g4 g5]
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_stubs  # noqa: E402

Config = ref_stubs.install()
sys.path.insert(0, "/root/reference")

from oracle import cases, detfill  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_num_threads(8)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def flat_summary(prefix, a, out):
    for k, v in cases.summarize(a.detach().cpu().numpy() if torch.is_tensor(a) else a).items():
        out[f"{prefix}.{k}"] = v


class _TorchProxy:
    """`torch` as seen by rgb_mapping.py, with device() forced to CPU (reference
    hard-codes torch.device('cuda', gpu_id): rgb_mapping.py:14)."""

    def __getattr__(self, k):
        return getattr(torch, k)

    @staticmethod
    def device(*a, **k):
        return torch.device("cpu")


def model_config(E=100, C=64, G=240, num_proc=2):
    return Config(
        INSTRUCTION_ENCODER=Config(vocab_size=2504, embedding_size=50, use_pretrained_embeddings=False,
                                   hidden_size=128, rnn_type="LSTM", final_state_only=False, bidirectional=True),
        RGB_ENCODER=Config(output_size=256, pretrain_model="__synthetic__"),
        DEPTH_ENCODER=Config(output_size=128, ddppo_checkpoint="NONE", backbone="resnet50"),
        MAP_ENCODER=Config(ego_map_size=E, output_size=256),
        STATE_ENCODER=Config(hidden_size=512, rnn_type="GRU", input_type=["rgb", "depth", "map"]),
        PROGRESS_MONITOR=Config(use=True, alpha=1.0),
        CONTRASTIVE_MONITOR=Config(use=True, alpha=1.0, target_tau=0.07),
        PREDICTION_MONITOR=Config(use=True, alpha=0.1),
        RGBMAPPING=Config(map_depth=C, global_map_size=G, egocentric_map_size=E, resolution=0.12, gpu_id=0,
                          num_proc=num_proc),
    )


def build_policy(num_proc=2):
    import vlnce_baselines.common.rgb_mapping as rm
    rm.torch = _TorchProxy()
    import vlnce_baselines.models.encoders.unet_encoder as ue
    from vlnce_baselines.models.policy import BasePolicy
    import gym.spaces as sp

    real_load = torch.load

    def fake_load(path, *a, **k):  # UNet.__init__ loads a segmentation checkpoint (unet_encoder.py:19-22)
        sd = ue.ResNetUNet(3, 27).state_dict()
        return {"models": {"img_segm_model": {"module.model." + k_: v for k_, v in sd.items()}}}

    torch.load = fake_load
    try:
        obs_space = sp.Dict({"depth": sp.Box(shape=(256, 256, 1)), "rgb": sp.Box(shape=(224, 224, 3))})
        pol = BasePolicy(obs_space, sp.Box(shape=(2,)), model_config(num_proc=num_proc))
    finally:
        torch.load = real_load
    sd = pol.state_dict()
    new = {k: T(detfill.state_value(k, tuple(v.shape))).to(v.dtype) for k, v in sd.items()}
    pol.load_state_dict(new, strict=True)
    # reference default: frozen pre-trained word embeddings (instruction_encoder.py:31-35, default.py:85,92)
    pol.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
    return pol


# ============================================================================= state_dict contract
def shapes():
    """Key names / shapes / dtypes / trainability of BasePolicy.state_dict(): the checkpoint
    compatibility contract (common_trainer.py:71-75,99,131)."""
    import json
    pol = build_policy()
    req = {k: bool(p.requires_grad) for k, p in pol.named_parameters(remove_duplicate=False)}
    spec = {k: dict(shape=list(v.shape), dtype=str(v.dtype).replace("torch.", ""), param=k in req,
                    trainable=req.get(k, False)) for k, v in pol.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_spec.json"), "w") as f:
        json.dump(spec, f, indent=0, sort_keys=True)
    print("shapes:", len(spec), "entries,", sum(int(np.prod(v["shape"])) for v in spec.values()), "elements")


# ============================================================================= G1
def g1():
    import vlnce_baselines.common.rgb_mapping as rm
    rm.torch = _TorchProxy()
    out = {}
    for name in cases.BEV_CASES:
        c = cases.bev_inputs(name)
        E = c["E"]
        cmin, cmax = -240 * 0.12 / 2, 240 * 0.12 / 2
        csl = rm.ComputeSpatialLocs(E, 240, torch.device("cpu"), cmin, cmax)
        pgp = rm.ProjectToGroundPlane(E, torch.device("cpu"))
        locs, valid = csl.forward(T(c["depth"]) * 10)
        locs_np = locs.numpy().copy()
        proj = pgp.forward(T(c["feat"]), locs, valid)
        idx = ref_stubs.CAPTURE["index"][:, 0, :].numpy()  # [B, Hf*Wf] linear cell (0 for invalid)
        src = ref_stubs.CAPTURE["src"]
        invalid = (src[:, 0, :] == -1e16).numpy() & (T(c["feat"]).reshape(c["B"], c["C"], -1)[:, 0, :] != -1e16).numpy()
        out[f"{name}.locs_sha"] = sha(locs_np.astype(np.int64))
        out[f"{name}.valid_sha"] = sha(valid.numpy().astype(np.uint8))
        out[f"{name}.lin_idx_sha"] = sha(idx.astype(np.int32))
        out[f"{name}.invalid_sha"] = sha(invalid.astype(np.uint8))
        out[f"{name}.proj_sha"] = sha(proj.numpy().astype(np.float32))
        out[f"{name}.locs_sample"] = locs_np.reshape(-1)[::251].astype(np.int64)
        out[f"{name}.lin_idx_sample"] = idx.reshape(-1)[::97].astype(np.int32)
        out[f"{name}.n_valid"] = np.int64((~invalid).sum())
        flat_summary(f"{name}.proj", proj, out)
        print("g1", name, "valid", int((~invalid).sum()), "of", invalid.size, "proj abssum", float(proj.abs().sum()))
    np.savez_compressed(os.path.join(OUT, "g1_bev.npz"), **out)


# ============================================================================= G2
def g2():
    import vlnce_baselines.common.rgb_mapping as rm
    rm.torch = _TorchProxy()
    m = cases.MAP_SEQ
    mapper = rm.RGBMapping(model_config(num_proc=m["B"]).RGBMAPPING)
    out = {}
    for step in range(m["steps"]):
        c = cases.mapseq_inputs(step)
        obs = {"depth": T(c["depth"]), "gps": T(c["gps"]), "compass": T(c["compass"])}
        ego = mapper.forward(T(c["feat"]), obs, T(c["masks"]))
        flat_summary(f"s{step}.ego", ego, out)
        flat_summary(f"s{step}.global", mapper.full_global_map, out)
        out[f"s{step}.global_nnz"] = np.int64((mapper.full_global_map != 0).sum())
        out[f"s{step}.ego_patch"] = ego[:, ::16, 40:56, 44:60].numpy().copy()
        print("g2 step", step, "ego abssum", float(ego.abs().sum()), "global nnz", int(out[f"s{step}.global_nnz"]))
    np.savez_compressed(os.path.join(OUT, "g2_mapseq.npz"), **out)


# ============================================================================= G3
def g3():
    from vlnce_baselines.common.aux_losses import AuxLosses
    pol = build_policy()
    pol.train()
    pol.net.depth_encoder.eval()
    pol.net.rgb_encoder.eval()
    T_, N = 4, 2
    obs_np, prev, masks, weights = cases.update_inputs(T_, N)
    obs = {k: T(v) for k, v in obs_np.items()}
    AuxLosses.activate()
    AuxLosses.clear()
    h0 = torch.zeros(pol.net.num_recurrent_layers, N, 512)
    pred, aux_loss = pol(obs, h0, T(prev), T(masks), T(weights))
    out = {"pred": pred.detach().numpy(), "aux_loss": aux_loss.detach().numpy()}
    for name in ["prediction_monitor", "contrastive_monitor", "progress_monitor"]:
        out[f"aux.{name}"] = AuxLosses.get_loss(name).detach().numpy()
    out["att_map_t_m"] = pol.net.att_map_t_m.detach().numpy()
    out["prog"] = pol.prog.detach().numpy()
    # loss of dagger_trainer.py:526-533
    logits = torch.tanh(pred).view(T_, N, -1)
    w = T(weights)
    action_loss = torch.nn.functional.mse_loss(logits, obs["waypoint"][:, :2].view(T_, N, -1), reduction="none").sum(2)
    action_loss = ((w * action_loss).sum(0) / w.sum(0)).mean()
    loss = action_loss + aux_loss
    loss.backward()
    out["action_loss"] = action_loss.detach().numpy()
    out["loss"] = loss.detach().numpy()
    out["h_out"] = h0.detach().numpy()  # mutated in place by the net (mg_map_policy.py:220-227,242-249)
    sd = pol.state_dict()
    for k in ["net.map_encoder.cnn.1.running_mean", "net.map_encoder.cnn.1.running_var",
              "net.map_decoder.conv_up0.1.running_mean", "net.map_decoder.conv_up0.1.running_var",
              "net.map_classfier.1.running_mean", "net.map_classfier.4.running_var",
              "net.map_decoder.base_model.layer1.1.bn2.running_var", "net.map_decoder.layer0_1x1.1.running_mean"]:
        out["bn." + k] = sd[k].numpy()
    gn, gs, names, nograd = [], [], [], []
    seen = set()
    for k, p in pol.named_parameters():
        if id(p) in seen:
            continue
        seen.add(id(p))
        if p.grad is None:
            if p.requires_grad:
                nograd.append(k)
            continue
        g = p.grad.detach().numpy().reshape(-1)
        names.append(k)
        gn.append(np.sqrt((g.astype(np.float64) ** 2).sum()))
        s = np.zeros(8, np.float32)
        pick = g[:: max(1, g.size // 8)][:8]
        s[: pick.size] = pick
        gs.append(s)
    out["grad.names"] = np.array(names)
    out["grad.norm"] = np.array(gn, np.float64)
    out["grad.sample"] = np.stack(gs)
    out["grad.none"] = np.array(nograd)
    # pred_sem_map is returned by the net, not kept: re-run the net for its digest (BN stats move again; fine)
    with torch.no_grad():
        AuxLosses.clear()
        h1 = torch.zeros(2, N, 512)
        pol.eval()
        _, _, sem = pol.net(obs, h1, T(prev), T(masks))
        flat_summary("eval.pred_sem_map", sem, out)
        out["eval.pred"] = pol.action_distribution(pol.net(obs, torch.zeros(2, N, 512), T(prev), T(masks))[0]).mean.numpy()
    AuxLosses.deactivate()
    print("g3 pred", out["pred"].ravel()[:4], "aux", float(out["aux_loss"]), "loss", float(out["loss"]),
          "params with grad", len(names), "trainable without grad", len(nograd))
    np.savez_compressed(os.path.join(OUT, "g3_update.npz"), **out)


# ============================================================================= G4
def g4():
    from vlnce_baselines.common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    out = {}
    for rgb_hw in (224, 256):
        pol = build_policy(num_proc=2)
        pol.eval()
        h = torch.zeros(2, 2, 512)
        prev = torch.zeros(2, 2)
        with torch.no_grad():
            for step in range(3):
                obs_np, masks = cases.act_inputs(step, rgb_hw=rgb_hw)
                obs = {k: T(v) for k, v in obs_np.items()}
                if step == 1:  # update_map-only step (dagger_trainer.py:438-439)
                    pol.update_map(obs, T(masks))
                    flat_summary(f"r{rgb_hw}.s{step}.ego", obs["rgb_ego_map"], out)
                    continue
                value, action, logp, h = pol.act(obs, h, prev, T(masks), deterministic=True)
                prev = action
                p = f"r{rgb_hw}.s{step}"
                out[p + ".value"] = value.numpy()
                out[p + ".action"] = action.numpy()
                out[p + ".logp"] = logp.numpy()
                out[p + ".prog"] = pol.prog.numpy()
                flat_summary(p + ".h", h, out)
                flat_summary(p + ".ego", obs["rgb_ego_map"], out)
                flat_summary(p + ".att", pol.net.att_map_t_m, out)
                print("g4", p, "action", action.numpy().ravel(), "value", value.numpy().ravel())
        flat_summary(f"r{rgb_hw}.global", pol.net.rgb_mapping_module.full_global_map, out)
    np.savez_compressed(os.path.join(OUT, "g4_act.npz"), **out)


# ============================================================================= G5
def g5():
    pol = build_policy()
    q, k, v, mask = cases.attn_inputs()
    o, a = pol.net._attn(T(q), T(k), T(v), T(mask))
    o2, a2 = pol.net._attn(T(q), T(k), T(v), None)
    np.savez_compressed(os.path.join(OUT, "g5_attn.npz"), out=o.detach().numpy(), attn=a.detach().numpy(),
                        out_nomask=o2.detach().numpy(), attn_nomask=a2.detach().numpy())
    print("g5 out", o.detach().numpy().ravel()[:4])


def g5f():
    """BASELINE configs[4]: the reference's `_attn` (mg_map_policy.py:173-178) at B = 64, L = 160 on inputs that are exactly
    representable in OCP e4m3 (oracle/cases.py::attn_fp8_inputs) — the pin for the fp8-storage attention kernels."""
    pol = build_policy()
    c = cases.attn_fp8_inputs()
    inv = c["inverse"]
    k = np.ascontiguousarray(c["k"][inv].transpose(0, 2, 1))      # [B, C, L], as the reference's Conv1d output
    v = np.ascontiguousarray(c["v"][inv].transpose(0, 2, 1))
    L = k.shape[2]
    mask = np.arange(L)[None, :] >= c["lengths"][inv][:, None]
    o, a = pol.net._attn(T(c["q"]), T(k), T(v), T(mask))
    np.savez_compressed(os.path.join(OUT, "g5f_attn_fp8.npz"), out=o.detach().numpy(), attn=a.detach().numpy())
    print("g5f out", o.detach().numpy().ravel()[:4], "attn row sums", a.detach().numpy().sum(1)[:3])


# ============================================================================= G9: depth branch from raw depth
def g9():
    """VlnResnetDepthEncoder.forward (resnet_encoders.py:72-102) from RAW depth: the reference's own plumbing (avg-pool
    hand-off, spatial-embedding concat) around the restated third-party backbone of tools/ref_stubs.py."""
    pol = build_policy()
    enc = pol.net.depth_encoder
    obs_np, _ = cases.act_inputs(0, B=2, tag="g9")
    with torch.no_grad():
        feat = enc.visual_encoder({"depth": T(obs_np["depth"])})
        out = enc({"depth": T(obs_np["depth"])})
    assert tuple(feat.shape) == (2, 128, 4, 4) and tuple(out.shape) == (2, 192, 4, 4)
    np.savez_compressed(os.path.join(OUT, "g9_depth.npz"), feat=feat.numpy(), out_sample=out.numpy()[:, ::7, :, :],
                        n_params=np.int64(sum(p.numel() for p in enc.visual_encoder.parameters())))
    print("g9 feat", feat.numpy().ravel()[:4], "params", sum(p.numel() for p in enc.visual_encoder.parameters()))


# ============================================================================= G8: checkpoint / resume contract
class _FakeLogger:
    def warning(self, *a, **k):
        pass

    info = warning


def g8():
    """The reference trainer's own save_checkpoint / resume_dagger / load-for-finetune code
    (common_trainer.py:71-76,91-139), run UNMODIFIED as unbound methods on a stand-in `self` (the trainer class itself
    needs Habitat to be constructed): what a checkpoint file contains, which file a resume picks, and the
    (start_dagger_it, start_epoch_it) it derives."""
    import json
    import tempfile
    import time
    import types
    for name, attrs in {
        "habitat.utils": {}, "habitat.utils.visualizations": {}, "habitat.utils.visualizations.utils": {"append_text_to_image": None},
        "habitat_baselines.common.base_trainer": {"BaseRLTrainer": object},
        "habitat_baselines.common.environments": {"get_env_class": None},
        "habitat_baselines.common.tensorboard_utils": {"TensorboardWriter": None},
        "habitat_extensions": {}, "habitat_extensions.utils": {"observations_to_image": None},
        "vlnce_baselines.common.env_utils": {"construct_envs_auto_reset_false": None},
        "vlnce_baselines.common.utils": {"transform_obs": None}, "tqdm": {},
    }.items():
        m = sys.modules.get(name) or ref_stubs._mod(name)
        for k, v in attrs.items():
            setattr(m, k, v)
    sys.modules["habitat"].logger = _FakeLogger()
    cu = sys.modules["habitat_baselines.common.utils"]
    cu.batch_obs = cu.generate_video = cu.poll_checkpoint_folder = None
    import vlnce_baselines.common_trainer as ct

    pol = build_policy()
    real_load = torch.load
    # the reference was written for torch 1.6, where torch.load had no weights_only default to trip over its pickled config
    torch.load = lambda f, *a, **k: real_load(f, *a, **{**k, "weights_only": False})
    folder = tempfile.mkdtemp(prefix="g8_")
    me = types.SimpleNamespace(actor_critic=types.SimpleNamespace(module=pol),
                               config=Config(CHECKPOINT_FOLDER=folder, RESUME_CKPT=None, DAGGER=Config(EPOCHS=4)))
    me.load_checkpoint = lambda path, *a, **k: ct.CommonTrainer.load_checkpoint(me, path, *a, **k)
    out = {"cases": []}
    ct.CommonTrainer.save_checkpoint(me, "ckpt.0.pth", extra_state={"dagger_it": 2})
    time.sleep(0.05)
    ct.CommonTrainer.save_checkpoint(me, "ckpt.3.pth", extra_state={"dagger_it": 2})
    ck = torch.load(os.path.join(folder, "ckpt.3.pth"), map_location="cpu", weights_only=False)
    out["top_level_keys"] = sorted(ck.keys())
    out["extra_state"] = ck["extra_state"]
    out["n_state_keys"] = len(ck["state_dict"])
    out["state_sha256"] = {k: sha(v.numpy()) for k, v in list(ck["state_dict"].items())[::40]}
    out["cases"].append(dict(files=["ckpt.0.pth", "ckpt.3.pth"], newest="ckpt.3.pth", epochs=4,
                             result=list(ct.CommonTrainer.resume_dagger(me))))
    time.sleep(0.05)
    ct.CommonTrainer.save_checkpoint(me, "ckpt.1.pth", extra_state={"dagger_it": 5})   # newest by mtime, not by name
    out["cases"].append(dict(files=["ckpt.0.pth", "ckpt.3.pth", "ckpt.1.pth"], newest="ckpt.1.pth", epochs=4,
                             result=list(ct.CommonTrainer.resume_dagger(me))))
    # load-for-finetune (:71-76): keys get a 'module.' prefix and go through load_state_dict(strict=False) of the DDP wrapper
    wrapper = torch.nn.Module()
    wrapper.module = pol
    sd = {"module." + k: v for k, v in ck["state_dict"].items()}
    sd["module.not_a_key"] = torch.zeros(1)
    del sd["module.prog_pred.bias"]
    msg = wrapper.load_state_dict(sd, strict=False)
    out["finetune_missing"], out["finetune_unexpected"] = list(msg.missing_keys), list(msg.unexpected_keys)
    torch.load = real_load
    with open(os.path.join(OUT, "g8_ckpt.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("g8", out["top_level_keys"], out["cases"], out["n_state_keys"])


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["shapes", "g1", "g2", "g3", "g4", "g5", "g5f", "g8", "g9"]
    for w in which:
        globals()[w]()
