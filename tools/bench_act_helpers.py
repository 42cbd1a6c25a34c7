"""Synthetic rollout observations shared by tools/bench_act.py and tools/prof_act_graph.py."""
import torch


class _Box:
    shape = (2,)


def obs_of(B, hw, gen):
    ins = torch.zeros(B, 200, dtype=torch.int64, device="cuda")
    ins[:, :80] = torch.randint(1, 2504, (B, 80), device="cuda", generator=gen)
    return {
        "rgb": torch.randint(0, 256, (B, hw, hw, 3), device="cuda", generator=gen).float(),
        "depth": torch.rand(B, 256, 256, 1, device="cuda", generator=gen),
        "depth_features": torch.randn(B, 128, 4, 4, device="cuda", generator=gen),
        "instruction": ins,
        "gps": (torch.rand(B, 2, device="cuda", generator=gen) - 0.5) * 4,
        "compass": (torch.rand(B, 1, device="cuda", generator=gen) - 0.5) * 6.28,
    }


