"""ORACLE (test infrastructure, never on the product path): CPU restatement of the
reference's RGB-D -> egocentric BEV operator, vlnce_baselines/common/rgb_mapping.py.

* Integer part (pixel -> metric -> integer BEV cell -> linear index -> scatter-max) is
  restated in NumPy with explicit float32 arithmetic, op for op:
      ComputeSpatialLocs.forward      rgb_mapping.py:153-176
      ProjectToGroundPlane.forward    rgb_mapping.py:184-232
  (torch_scatter 2.0.6 `scatter_max` is third-party and absent: restated as "per-cell max
  over the sources, cells without a source are 0" — the semantics its call site relies on.)
* Float part (rotate / paste / translate / max-fuse / retrieve) restates
      RotateTensor.forward :239-250, get_grid :106-139, to_grid.get_grid_coords :100-103,
      Mapping.project_feat_to_map :32-72, RGBMapping.forward :79-90
  with torch CPU `affine_grid` / `grid_sample` (the operators the reference itself calls).

Pinned against tests/golden/g1_bev.npz (bit-exact hashes) and g2_mapseq.npz (float
digests) captured from the unmodified reference by tools/make_goldens.py.

Note on division: `X / local_scale` with a Python-float divisor is a true IEEE float32
division on the CPU path that produced the goldens; PyTorch's CUDA kernel multiplies by
the float32 reciprocal instead, which can differ in the last bit and flip a round() tie.
Oracle and HIP kernel both use the true division (the golden-pinned behaviour).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


# ----------------------------------------------------------------------------- integer gate
def spatial_locs(depth_m: np.ndarray, ego_size: int, local_scale: float = 28.8 / 240):
    """depth_m [B,H,W] float32 metres (already x10).  Returns x_gp, y_gp int64 [B,H,W], valid bool.
    rgb_mapping.py:153-176 (hfov 90 deg, cx=cy=imh/2, fx=fy=imh/2 / tan(45 deg))."""
    B, H, W = depth_m.shape
    cx, cy = f32(H / 2.0), f32(W / 2.0)
    fx = f32((H / 2.0) / np.tan(np.deg2rad(90 / 2.0)))
    fy = f32((W / 2.0) / np.tan(np.deg2rad(90 / 2.0)))
    x = np.arange(0, W).astype(f32)[None, None, :]
    y = np.arange(H, 0, -1).astype(f32)[None, :, None]
    xx = (x - cx) / fx
    yy = (y - cy) / fy
    Z = depth_m.astype(f32)
    X = xx * Z
    Y = yy * Z
    valid = (Z != 0) & (Y > f32(-1.5)) & (Y < f32(0.1))
    ls = f32(local_scale)
    half = f32((ego_size - 1) / 2)
    with np.errstate(invalid="ignore"):
        x_gp = np.rint(X / ls + half).astype(np.int64)
        y_gp = np.rint(-(Z / ls) + half).astype(np.int64)
    return x_gp, y_gp, valid


def subsample_index(n_feat: int, depth_h: int) -> np.ndarray:
    """(arange(n_feat) * K).long() with K = depth_h / n_feat evaluated in float32
    (int64 tensor * python float -> float32): rgb_mapping.py:188-192."""
    K = f32(depth_h / n_feat)
    return (np.arange(n_feat).astype(f32) * K).astype(np.int64)


def linear_index(x_gp, y_gp, valid, n_feat_h, n_feat_w, ego_size):
    """Sub-sample to the feature grid, flag out-of-range / invalid sources, linearise
    (lin = y * E + x, 0 for invalid): rgb_mapping.py:188-216.  Returns lin int32 [B,Hf*Wf], invalid bool."""
    H = x_gp.shape[-1]
    ih = subsample_index(n_feat_h, H)
    iw = subsample_index(n_feat_w, H)
    xs = x_gp[:, ih[:, None], iw[None, :]]
    ys = y_gp[:, ih[:, None], iw[None, :]]
    vs = valid[:, ih[:, None], iw[None, :]]
    E = ego_size
    invalid = (ys >= E) | (ys < 0) | (xs >= E) | (xs < 0) | ~vs
    lin = np.where(invalid, 0, ys * E + xs)
    B = lin.shape[0]
    return lin.reshape(B, -1).astype(np.int32), invalid.reshape(B, -1)


def scatter_max(feat: np.ndarray, lin: np.ndarray, invalid: np.ndarray, ego_size: int) -> np.ndarray:
    """feat [B,C,Hf,Wf] -> [B,C,E,E]: per-cell max over valid sources, empty cells 0
    (rgb_mapping.py:206-232: invalid sources are forced to -1e16 at cell 0 and -1e16 is
    mapped back to 0, which equals skipping them)."""
    B, C = feat.shape[:2]
    E2 = ego_size * ego_size
    out = np.zeros((B, C, E2), f32)
    src = feat.reshape(B, C, -1)
    for b in range(B):
        keep = ~invalid[b]
        if not keep.any():
            continue
        cells = lin[b, keep].astype(np.int64)
        vals = src[b][:, keep]  # [C, n]
        order = np.argsort(cells, kind="stable")
        cells = cells[order]
        vals = vals[:, order]
        starts = np.flatnonzero(np.r_[True, cells[1:] != cells[:-1]])
        mx = np.maximum.reduceat(vals, starts, axis=1)
        out[b][:, cells[starts]] = mx + f32(0.0)  # +0.0: the reference's fix-up arithmetic never yields -0.0
    return out.reshape(B, C, ego_size, ego_size)


def project_to_ground(feat: np.ndarray, depth_raw: np.ndarray, ego_size: int):
    """feat [B,C,Hf,Wf], depth_raw [B,H,W,1] (sensor units; x10 = metres, rgb_mapping.py:37).
    Returns (proj [B,C,E,E], lin int32, invalid bool, x_gp, y_gp, valid)."""
    depth_m = (depth_raw[..., 0].astype(f32) * f32(10))
    x_gp, y_gp, valid = spatial_locs(depth_m, ego_size)
    lin, invalid = linear_index(x_gp, y_gp, valid, feat.shape[2], feat.shape[3], ego_size)
    proj = scatter_max(feat.astype(f32), lin, invalid, ego_size)
    return proj, lin, invalid, x_gp, y_gp, valid


# ----------------------------------------------------------------------------- float part
def channel_maxpool(feat: torch.Tensor, map_depth: int) -> torch.Tensor:
    """adaptive_max_pool1d over channels, 64 -> map_depth (rgb_mapping.py:81-84)."""
    bs, c, h, w = feat.shape
    x = feat.permute(0, 2, 3, 1).reshape(bs, -1, c)
    x = F.adaptive_max_pool1d(x, map_depth)
    return x.reshape(bs, h, w, -1).permute(0, 3, 1, 2)


def rotate(x: torch.Tensor, heading: torch.Tensor) -> torch.Tensor:
    """RotateTensor.forward (rgb_mapping.py:239-250): A = [[c, s, 0], [-s, c, 0]]."""
    t = heading.reshape(-1)
    A = torch.zeros(x.size(0), 2, 3)
    A[:, 0, 0] = torch.cos(t)
    A[:, 0, 1] = torch.sin(t)
    A[:, 1, 0] = -torch.sin(t)
    A[:, 1, 1] = torch.cos(t)
    grid = F.affine_grid(A, list(x.shape), align_corners=False)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def grid_cell(gps: torch.Tensor, G: int, resolution: float = 0.12):
    """to_grid.get_grid_coords (rgb_mapping.py:100-103)."""
    cmin = -G * resolution / 2
    cmax = G * resolution / 2
    gs = (cmax - cmin) / G
    gx = ((cmax - gps[:, 0]) / gs).round()
    gy = ((gps[:, 1] - cmin) / gs).round()
    return gx, gy


def translate(x: torch.Tensor, tx: torch.Tensor, ty: torch.Tensor) -> torch.Tensor:
    """get_grid's trans_grid + grid_sample (rgb_mapping.py:127-139,52-53)."""
    one, zero = torch.ones_like(tx), torch.zeros_like(tx)
    theta = torch.stack([torch.stack([one, -zero, tx], 1), torch.stack([zero, one, ty], 1)], 1)
    grid = F.affine_grid(theta, list(x.shape), align_corners=False)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


class MapperRef:
    """State + step of Mapping.project_feat_to_map / RGBMapping.forward (rgb_mapping.py:32-90)."""

    def __init__(self, num_proc, G=240, E=100, C=64, resolution=0.12):
        self.G, self.E, self.C, self.res = G, E, C, resolution
        self.full_global_map = torch.zeros(num_proc, G, G, C)

    def step(self, feat, depth_raw, gps, compass, masks):
        """feat [B,Cf,Hf,Wf] torch f32; returns final_retrieval [B,C,E,E]."""
        G, E = self.G, self.E
        bs = feat.shape[0]
        feat = channel_maxpool(feat, self.C)
        gx, gy = grid_cell(gps, G, self.res)
        self.full_global_map[:bs] = self.full_global_map[:bs] * masks.view(bs, 1, 1, 1)
        proj, *_ = project_to_ground(feat.numpy(), depth_raw.numpy(), E)
        proj = rotate(torch.from_numpy(proj), -compass)
        lo, hi = G // 2 - math.floor(E / 2), G // 2 + math.ceil(E / 2)
        agent_view = torch.zeros(bs, self.C, G, G)
        agent_view[:, :, lo:hi, lo:hi] = proj
        half = G // 2
        translated = translate(agent_view, -(gy - half) / half, -(gx - half) / half)
        self.full_global_map[:bs] = torch.maximum(self.full_global_map[:bs], translated.permute(0, 2, 3, 1))
        back = translate(self.full_global_map[:bs].permute(0, 3, 1, 2).contiguous(), (gy - half) / half, (gx - half) / half)
        crop = back[:, :, lo:hi, lo:hi]
        return rotate(crop, compass)
