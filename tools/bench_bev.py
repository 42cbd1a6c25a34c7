#!/usr/bin/env python3
"""GPU-box tool: time the BEV operator kernels (operator 1, rollout path) and report achieved HBM
GB/s against the algorithmic bytes of SURVEY.md 8(d):
    scatter, per sample : Cf*Hf*Wf*4 (features) + Hd*Wd*4 (depth) + C*E*E*4 (map out)
    fuse + retrieve     : ~4 * C*E*E*4 per sample (window RMW + ego read + ego write)
Configurations: cfg1 (B=1, 256^2 RGB-D, E=100, C=64), the reference-native 224^2 variant, a rollout
batch B=8, and cfg4 (B=32, E=200, C=40 from 64 feature channels).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us

print(f"{'config':26s} {'kernel':19s} {'us':>9s} {'alg MB':>9s} {'GB/s':>9s} {'% of 8 TB/s':>11s}")
for name, B, Hf, E, C, G in [("cfg1 B=1 256^2 E100 C64", 1, 256, 100, 64, 240), ("native B=1 224^2", 1, 224, 100, 64, 240),
                             ("rollout B=8 256^2", 8, 256, 100, 64, 240), ("cfg4 B=32 E200 C40", 32, 256, 200, 40, 480)]:
    depth = (torch.rand(B, 256, 256, device="cuda") + 0.05)
    depth[:, :8] = 0
    feat = torch.relu(torch.randn(B, 64, Hf, Hf, device="cuda"))
    gps = torch.rand(B, 2, device="cuda") * 4 - 2
    compass = torch.rand(B, device="cuda") * 6.28 - 3.14
    masks = torch.ones(B, device="cuda")
    gm = torch.zeros(B, G, G, C, device="cuda")
    lin = ops.bev_index(depth, Hf, Hf, E)
    planes = ops.bev_scatter_max(feat, lin, C, E)
    rot = ops.bev_rotate(planes, compass, -1.0)
    rows = [
        ("index", lambda: ops.bev_index(depth, Hf, Hf, E), B * (256 * 256 * 4 + Hf * Hf * 4)),
        ("scatter_max", lambda: ops.bev_scatter_max(feat, lin, C, E), B * (64 * Hf * Hf * 4 + Hf * Hf * 4 + C * E * E * 4)),
        ("rotate", lambda: ops.bev_rotate(planes, compass, -1.0), B * 2 * C * E * E * 4),
        ("fuse", lambda: ops.map_fuse(rot, gm, gps, masks, 0.12), B * 3 * C * (E + 4) ** 2 * 4),
        ("retrieve", lambda: ops.map_retrieve(gm, gps, compass, E, 0.12, fused=False), B * 4 * C * E * E * 4),
    ]
    tot_us, tot_b = 0.0, 0
    for k, fn, nbytes in rows:
        us = timeit(fn)
        tot_us += us; tot_b += nbytes
        print(f"{name:26s} {k:19s} {us:9.1f} {nbytes / 1e6:9.2f} {nbytes / us / 1e3:9.1f} {nbytes / us / 1e3 / 8000 * 100:10.1f}%")
    print(f"{name:26s} {'ALL (5 launches)':19s} {tot_us:9.1f} {tot_b / 1e6:9.2f} {tot_b / tot_us / 1e3:9.1f} {tot_b / tot_us / 1e3 / 8000 * 100:10.1f}%")
    if ops.bev_planes_ok(C, E):
        # round 3: scatter-max + rotation in one launch, the fuse reads the rotated planes (what Mapping.project_feat_to_map runs).
        # Algorithmic bytes of the fused stages: features + indices in, ONE rotated map out (the unrotated planes never exist);
        # the retrieval reads the window of the global map and writes the ego map (the crop never exists).
        rotp = ops.bev_scatter_rotate(feat, lin, compass, -1.0, C, E)
        by = dict((k, (fn, nb)) for k, fn, nb in rows)
        frows = [
            ("index",) + by["index"],
            ("scatter+rotate", lambda: ops.bev_scatter_rotate(feat, lin, compass, -1.0, C, E), by["scatter_max"][1]),
            ("fuse (planes)", lambda: ops.map_fuse(rotp, gm, gps, masks, 0.12, planes=True), by["fuse"][1]),
            ("retrieve (LDS tiles)", lambda: ops.map_retrieve(gm, gps, compass, E, 0.12), B * 2 * C * E * E * 4),   # ops.map_retrieve's own choice
        ]
        f_us = 0.0
        for k, fn, nbytes in frows:
            us = timeit(fn)
            f_us += us
            if k != "index":
                print(f"{name:26s} {k:19s} {us:9.1f} {nbytes / 1e6:9.2f} {nbytes / us / 1e3:9.1f} {nbytes / us / 1e3 / 8000 * 100:10.1f}%")
        us = timeit(lambda: ops.map_retrieve(gm, gps, compass, E, 0.12, fused=True))
        nbytes = B * 2 * C * E * E * 4
        print(f"{name:26s} {'retrieve (registers)':19s} {us:9.1f} {nbytes / 1e6:9.2f} {nbytes / us / 1e3:9.1f} {nbytes / us / 1e3 / 8000 * 100:10.1f}%   (not in the sum)")
        fb = sum(nb for _, _, nb in frows)
        print(f"{name:26s} {'ALL fused, own B':19s} {f_us:9.1f} {fb / 1e6:9.2f} {fb / f_us / 1e3:9.1f} {fb / f_us / 1e3 / 8000 * 100:10.1f}%")
        print(f"{name:26s} {'ALL fused, 5-stage B':19s} {f_us:9.1f} {tot_b / 1e6:9.2f} {tot_b / f_us / 1e3:9.1f} {tot_b / f_us / 1e3 / 8000 * 100:10.1f}%")
