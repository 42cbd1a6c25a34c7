"""wsmgmap.ops.bev — operator 1: RGB-D -> egocentric BEV index / scatter-max / rotation, global-map fuse and retrieve (rollout only,
no autograd).
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink


# ----------------------------------------------------------------------------- BEV (no autograd: rollout only)
@torch.no_grad()
def bev_index(depth, Hf, Wf, E, depth_scale=10.0, local_scale=0.12):
    """depth [B,Hd,Wd] -> lin_idx int32 [B,Hf*Wf] (-1 = invalid source)."""
    _req(depth)
    _f32(depth)
    B, Hd, Wd = depth.shape
    lin = torch.empty(B, Hf * Wf, device=depth.device, dtype=torch.int32)
    _abi.call("wsmg_bev_index", _p(depth), B, Hd, Wd, float(depth_scale), Hf, Wf, E, float(local_scale), _p(lin), _stream())
    return lin


def bev_compact_ok(Hf, Wf, E, B=None):
    """Shapes the compacted-source route takes (source and cell ids are packed into 16 bits each).  Not below 4 frames: there the
    operator is four launches' latency (cfg1: 40 us), the scatter is not paced by its list, and the compaction adds 3 us to the index."""
    return Hf * Wf <= 65536 and E * E <= 65536 and sw.bev_compact and (B is None or B >= 4)


@torch.no_grad()
def bev_index_compact(depth, Hf, Wf, E, depth_scale=10.0, local_scale=0.12):
    """bev_index + the list of valid sources: -> (lin_idx int32 [B,Hf*Wf], (clist uint32-as-int32 [B,Hf*Wf], cnt int32 [B,nblk]))."""
    _req(depth)
    _f32(depth)
    B, Hd, Wd = depth.shape
    per = Hf * Wf
    lin = torch.empty(B, per, device=depth.device, dtype=torch.int32)
    clist = torch.empty(B, per, device=depth.device, dtype=torch.int32)
    cnt = torch.empty(B, (per + 8191) // 8192, device=depth.device, dtype=torch.int32)
    _abi.call("wsmg_bev_index_compact", _p(depth), B, Hd, Wd, float(depth_scale), Hf, Wf, E, float(local_scale), _p(lin), _p(clist), _p(cnt),
              _stream())
    return lin, (clist, cnt)


@torch.no_grad()
def bev_scatter_max(feat, lin, C, E):
    """feat [B,Cf,Hf,Wf] NCHW -> [B,C,E,E] NCHW planes."""
    _req(feat, lin)
    _f32(feat)
    B, Cf, Hf, Wf = feat.shape
    out = torch.empty(B, C, E, E, device=feat.device, dtype=torch.float32)
    _abi.call("wsmg_bev_scatter_max", _p(feat), _p(lin), B, Cf, Hf, Wf, C, E, _p(out), _stream())
    return out


@torch.no_grad()
def bev_rotate(planes, heading, sign):
    """planes [B,C,E,E] -> rotated NHWC [B,E,E,C]."""
    _req(planes, heading)
    B, C, E, _ = planes.shape
    out = torch.empty(B, E, E, C, device=planes.device, dtype=torch.float32)
    _abi.call("wsmg_bev_rotate", _p(planes), _p(heading), float(sign), B, C, E, _p(out), _stream())
    return out


@torch.no_grad()
def bev_scatter_rotate(feat, lin, heading, sign, C, E, compact=None):
    """bev_scatter_max + bev_rotate in one launch; the rotated map stays in NCHW planes [B,C,E,E] (for map_fuse(..., planes=True)).
    compact: bev_index_compact's (clist, cnt) — the scatter then walks the valid sources only (same planes bit for bit)."""
    _req(feat, lin, heading)
    _f32(feat, heading)
    B, Cf, Hf, Wf = feat.shape
    out = torch.empty(B, C, E, E, device=feat.device, dtype=torch.float32)
    if compact is not None:
        _abi.call("wsmg_bev_scatter_rotate_compact", _p(feat), _p(compact[0]), _p(compact[1]), _p(heading), float(sign), B, Cf, Hf, Wf, C, E,
                  _p(out), _stream())
        return out
    _abi.call("wsmg_bev_scatter_rotate", _p(feat), _p(lin), _p(heading), float(sign), B, Cf, Hf, Wf, C, E, _p(out), _stream())
    return out


def bev_planes_ok(C, E):
    """Shapes the one-launch scatter + rotation and the plane-consuming fuse take."""
    return C % 4 == 0 and C <= 64 and E > 1 and E * E * 4 <= 160 * 1024


@torch.no_grad()
def _check_global_map(global_map, B, C, *f32s):
    """The kernels index global_map[b] for b < B: the reference slices `full_global_map[:bs]` (rgb_mapping.py:43) and
    fails with a shape error when the batch has more rows than num_proc — here that would be an out-of-bounds access."""
    _f32(global_map, *f32s)
    if global_map.dim() != 4 or global_map.shape[1] != global_map.shape[2]:
        raise _abi.WsmgError(f"full_global_map must be [num_proc, G, G, C], got {tuple(global_map.shape)}")
    if global_map.shape[0] < B:
        raise _abi.WsmgError(f"batch of {B} rows but full_global_map holds {global_map.shape[0]} maps (num_proc): "
                             "construct the policy with RGBMAPPING.num_proc >= the rollout batch")
    if global_map.shape[3] != C:
        raise _abi.WsmgError(f"full_global_map has {global_map.shape[3]} channels, the ego map {C}")


def map_fuse(ego_rot, global_map, gps, masks, resolution=0.12, planes=False):
    """planes: ego_rot is [B,C,E,E] (bev_scatter_rotate's output) instead of NHWC [B,E,E,C]; same result bit for bit."""
    _req(ego_rot, global_map, gps, masks)
    if planes:
        B, C, E, _ = ego_rot.shape
    else:
        B, E, _, C = ego_rot.shape
    _check_global_map(global_map, B, C, ego_rot, gps, masks)
    if gps.shape[0] != B or masks.numel() != B:
        raise _abi.WsmgError("map_fuse: gps [B,2] and masks [B] must match the ego maps' batch")
    G = global_map.shape[1]
    _abi.call("wsmg_map_fuse_planes" if planes else "wsmg_map_fuse", _p(ego_rot), _p(global_map), _p(gps), _p(masks), B, C, E, G,
              float(resolution), _stream())


@torch.no_grad()
def map_retrieve(global_map, gps, compass, E, resolution=0.12, fused=None):
    """fused: crop + rotation in one launch, bit-identical to the two.  "tiled" (the default): the LDS-staged launch
    (wsmg_map_retrieve_tiled); True: the register form (16 gathers per item — 11 us at B = 1 but 311 vs 219 us at cfg4);
    False: crop and rotation as two launches through a scratch map."""
    _req(global_map, gps, compass)
    B = gps.shape[0]
    _check_global_map(global_map, B, global_map.shape[3] if global_map.dim() == 4 else -1, gps, compass)
    if compass.numel() != B:
        raise _abi.WsmgError("map_retrieve: compass [B] must match gps [B,2]")
    G, C = global_map.shape[1], global_map.shape[3]
    out = torch.empty(B, E, E, C, device=gps.device, dtype=torch.float32)
    if fused is None:
        fused = "tiled"
    if fused == "tiled":
        _abi.call("wsmg_map_retrieve_tiled", _p(global_map), _p(gps), _p(compass), B, C, E, G, float(resolution), _p(out), _stream())
        return out
    if fused:
        _abi.call("wsmg_map_retrieve_fused", _p(global_map), _p(gps), _p(compass), B, C, E, G, float(resolution), _p(out), _stream())
        return out
    scratch = torch.empty(B, E, E, C, device=gps.device, dtype=torch.float32)
    _abi.call("wsmg_map_retrieve", _p(global_map), _p(gps), _p(compass), B, C, E, G, float(resolution), _p(scratch), _p(out), _stream())
    return out
