"""Seeded input builders for the golden cases (TEST INFRASTRUCTURE — part of oracle/).

Every golden under tests/golden/ was produced by tools/make_goldens.py feeding THESE
inputs to the unmodified reference; tests regenerate the same inputs from the formula in
oracle/detfill.py and compare the oracle / the HIP path against the stored outputs.
Shapes follow SURVEY.md §8(c)/(d).
"""
import numpy as np

from . import detfill as df

MAX_TOK = 200
VOCAB = 2504


# ----------------------------------------------------------------------------- G1: BEV index + scatter
BEV_CASES = {
    # name: (E, C_map, Hf, B)
    "e100_c64_f224": (100, 64, 224, 3),
    "e100_c64_f256": (100, 64, 256, 3),
    "e200_c40_f256": (200, 40, 256, 3),
}


def bev_inputs(name):
    """depth_raw [B,256,256,1] in [0,1) (reference multiplies by 10: rgb_mapping.py:37),
    feat [B,C,Hf,Hf] with negative values.  Sample 0: random; sample 1: structured depth
    (zero rows, depths on multiples of the 0.12 m cell => .5 ties before round());
    sample B-1: all-zero depth (every source invalid)."""
    E, C, Hf, B = BEV_CASES[name]
    depth = df.uniform(f"g1.{name}.depth", (B, 256, 256, 1)) + np.float32(0.5)
    depth[:, :8] = 0.0
    k = (np.arange(256, dtype=np.float32) % 97).astype(np.float32)
    depth[1, 128:200, :, 0] = (np.float32(0.012) * k)[None, :]
    depth[1, 200:230, ::3, 0] = 0.0
    depth[1, 230:, :, 0] = np.float32(0.3)
    depth[B - 1] = 0.0
    feat = df.uniform(f"g1.{name}.feat", (B, C, Hf, Hf), 4.0)
    return dict(E=E, C=C, Hf=Hf, B=B, depth=depth, feat=feat)


# ----------------------------------------------------------------------------- G2: map sequence
MAP_SEQ = dict(B=2, E=100, G=240, C=64, Hf=224, steps=4)


def mapseq_inputs(step, E=100, C=64, Hf=224, B=2, tag="g2"):
    depth = df.uniform(f"{tag}.depth.{step}", (B, 256, 256, 1)) + np.float32(0.5)
    depth[:, :8] = 0.0
    feat = np.maximum(df.uniform(f"{tag}.feat.{step}", (B, C, Hf, Hf), 4.0), 0.0)  # post-ReLU features
    gps = df.uniform(f"{tag}.gps.{step}", (B, 2), 4.0) + np.float32(0.35 * step)
    compass = df.uniform(f"{tag}.compass.{step}", (B, 1), 2 * np.pi)
    masks = np.ones((B, 1), np.float32)
    if step == 0:
        masks[:] = 0.0
    if step == 2:
        masks[0] = 0.0  # env 0 starts a new episode at the 3rd step
    return dict(depth=depth, feat=feat, gps=gps, compass=compass, masks=masks)


# ----------------------------------------------------------------------------- G3: update path
def update_inputs(T=4, N=2, n_tok=(80, 37), tag="g3", E=100, C=64):
    """Teacher-forcing batch, time-major rows (row = t*N + n): dagger_trainer.py:93-109."""
    B = T * N
    instr_n = df.tokens(f"{tag}.instruction", N, list(n_tok), MAX_TOK, VOCAB)
    obs = {
        "instruction": np.tile(instr_n, (T, 1)).astype(np.float32),  # cache stores floats; encoder .long()s
        "rgb_features": df.uniform(f"{tag}.rgb_features", (B, 512, 7, 7), 2.0),
        "depth_features": df.uniform(f"{tag}.depth_features", (B, 128, 4, 4), 2.0),
        "rgb_ego_map": np.maximum(df.uniform(f"{tag}.rgb_ego_map", (B, C, E, E), 3.0), 0.0),
        "gt_semantic_map": np.floor((df.uniform(f"{tag}.gt_sem", (B, E, E)) + 0.5) * 27).clip(0, 26).astype(np.float32),
        "gt_path": (df.uniform(f"{tag}.gt_path", (B, E, E)) + np.float32(0.5)) * np.float32(50.0),
        "progress": df.uniform(f"{tag}.progress", (B, 1)) + np.float32(0.5),
        "waypoint": df.uniform(f"{tag}.waypoint", (B, 3), 2.0),
    }
    prev_actions = np.zeros((B, 2), np.float32)
    masks = np.ones((T, N), np.float32)
    masks[0] = 0.0
    weights = np.ones((T, N), np.float32)
    weights[T - 1, N - 1] = 0.0  # one padded step (collate pads weights with 0)
    return obs, prev_actions, masks.reshape(B, 1), weights


# ----------------------------------------------------------------------------- G4: rollout act()
def act_inputs(step, B=2, rgb_hw=224, tag="g4", n_tok=(80, 51)):
    obs = {
        "rgb": np.floor((df.uniform(f"{tag}.rgb.{step}", (B, rgb_hw, rgb_hw, 3)) + 0.5) * 256).clip(0, 255).astype(np.float32),
        "depth": (df.uniform(f"{tag}.depth.{step}", (B, 256, 256, 1)) + np.float32(0.5)),
        "instruction": df.tokens(f"{tag}.instruction", B, list(n_tok), MAX_TOK, VOCAB).astype(np.float32),
        "gps": df.uniform(f"{tag}.gps.{step}", (B, 2), 3.0) + np.float32(0.25 * step),
        "compass": df.uniform(f"{tag}.compass.{step}", (B, 1), 2 * np.pi),
        "depth_features": df.uniform(f"{tag}.depth_features.{step}", (B, 128, 4, 4), 2.0),
    }
    obs["depth"][:, :8] = 0.0
    masks = np.ones((B, 1), np.float32) if step > 0 else np.zeros((B, 1), np.float32)
    return obs, masks


# ----------------------------------------------------------------------------- G5: attention alone
def attn_inputs(B=6, C=256, L=160, tag="g5"):
    q = df.uniform(f"{tag}.q", (B, C), 8.0)
    k = df.uniform(f"{tag}.k", (B, C, L), 4.0)
    v = df.uniform(f"{tag}.v", (B, C, L), 2.0)
    lens = [160, 1, 37, 80, 159, 64][:B]
    mask = np.zeros((B, L), bool)
    for b, n in enumerate(lens):
        mask[b, n:] = True
    return q, k, v, mask


def attn_fp8_inputs(B=64, U=8, C=256, L=160, tag="g5f"):
    """BASELINE configs[4] (cross-attention with fp8 storage, instruction length 160, batch 64) on inputs that ARE e4m3 numbers:
    queries q [B, C], and per unique instruction u (row b uses set b % U, as the update path's T x N rows do) keys k [U, L, C] and
    values v [U, L, C], every value on the OCP e4m3 grid times a power-of-two scale — quantising them to e4m3 with that scale is
    lossless, so the reference's float32 `_attn` on these inputs (golden g5f) is what an fp8 kernel must reproduce, up to
    float32 summation order.  lengths [U]: valid tokens per set (the rest masked).
    -> dict(q, k, v float32; q_scale, k_scale, v_scale; lengths; inverse [B])"""
    from .attn_fp8_ref import dequantize_e4m3, quantize_e4m3
    sq, sk, sv = 2.0 ** -6, 2.0 ** -7, 2.0 ** -8

    def grid(name, shape, amp, scale):
        raw = df.uniform(f"{tag}.{name}", shape, amp)
        return dequantize_e4m3(quantize_e4m3(raw, scale), scale).astype(np.float32)
    q = grid("q", (B, C), 8.0, sq)
    k = grid("k", (U, L, C), 4.0, sk)
    v = grid("v", (U, L, C), 2.0, sv)
    lengths = np.asarray([160, 1, 37, 80, 159, 64, 120, 100][:U], np.int64)
    return dict(q=q, k=k, v=v, q_scale=sq, k_scale=sk, v_scale=sv, lengths=lengths, inverse=np.arange(B, dtype=np.int64) % U)


def summarize(a, n_sample=2048):
    """Small machine-comparable digest of a float array: sums + strided sample."""
    a = np.asarray(a)
    flat = a.reshape(-1).astype(np.float64)
    stride = max(1, flat.size // n_sample)
    return dict(
        shape=np.asarray(a.shape, np.int64),
        sum=np.float64(flat.sum()),
        abssum=np.float64(np.abs(flat).sum()),
        sample=flat[::stride][:n_sample].astype(np.float32),
        stride=np.int64(stride),
    )
