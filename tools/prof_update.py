#!/usr/bin/env python3
"""GPU-box diagnostic: which Python lines of one teacher-forcing update issue the small fill / copy / cat /
elementwise launches.  torch.profiler with stacks over ONE update (after two warm-up updates), grouped by
(aten op, innermost frame inside this repo).  Usage: python tools/prof_update.py [bf16|f32] [op substrings...]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity


class A:
    T, N = 64, 8


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    want = sys.argv[2:] or ["aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::clone", "aten::cat",
                            "aten::add", "aten::mul", "aten::sum", "aten::empty_strided", "aten::contiguous", "aten::to"]
    from wsmgmap.common.aux_losses import AuxLosses
    from wsmgmap.config import default_model_config
    from wsmgmap.models.policy import BasePolicy
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype=dtype))
    policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
    policy = policy.to(dev)
    policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
    opt = torch.optim.Adam(policy.parameters(), lr=2.5e-4)
    obs, prev, masks, weights = bench.synth_batch(A.T, A.N, dev, 1000)
    AuxLosses.activate()

    def update():
        opt.zero_grad(set_to_none=True)
        AuxLosses.clear()
        h0 = torch.zeros(policy.net.num_recurrent_layers, A.N, 512, device=dev)
        o = dict(obs)
        pred, aux = policy(o, h0, prev, masks, weights)
        loss = bench.dagger_loss(pred, aux, o["waypoint"], weights)
        loss.backward()
        opt.step()

    for _ in range(2):
        update()
    torch.cuda.synchronize()
    if os.environ.get("SITES", "1") == "1":
        # Python call sites of the allocation / copy / reduction entry points (one update, no profiler)
        sites = collections.Counter()
        def wrap(owner, name):
            f = getattr(owner, name)
            def g(*a, **k):
                fr = sys._getframe(1)
                hops = []
                while fr is not None and len(hops) < 2:
                    fn = fr.f_code.co_filename
                    if "/repo/" in fn or "ws-mgmap_amd" in fn:
                        hops.append("%s:%d" % (fn.split("/repo/")[-1].replace("ws-mgmap_amd/wsmgmap/", ""), fr.f_lineno))
                    fr = fr.f_back
                sites[(name, " < ".join(hops) or "[torch]")] += 1
                return f(*a, **k)
            setattr(owner, name, g)
            return f
        saved = []
        for owner, names in ((torch, ["zeros", "zeros_like", "ones", "full", "cat", "stack", "sum", "tensor", "arange", "empty", "empty_like"]),
                             (torch.Tensor, ["zero_", "fill_", "copy_", "clone", "contiguous", "to", "float", "sum", "mean", "cuda", "bfloat16", "masked_fill", "masked_fill_"])):
            for n in names:
                saved.append((owner, n, wrap(owner, n)))
        update()
        torch.cuda.synchronize()
        for owner, n, f in saved:
            setattr(owner, n, f)
        print("%-14s %5s  %s" % ("call", "n", "site"))
        for (n, w), c in sorted(sites.items(), key=lambda t: (t[0][0], -t[1])):
            print("%-14s %5d  %s" % (n, c, w))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        update()
        torch.cuda.synchronize()
    if os.environ.get("TABLE", "0") == "1":
        print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=70, max_name_column_width=60, max_shapes_column_width=90))
    ev = prof.events()
    groups = collections.Counter()
    dur = collections.Counter()
    for e in ev:
        if not any(e.name == w or e.name.startswith(w + ".") for w in want):
            continue
        if e.cpu_parent is not None and any(e.cpu_parent.name == w for w in want):
            continue   # nested aten op of a wanted op
        where = "?"
        for fr in (e.stack or []):
            if "/repo/" in fr and "prof_update" not in fr:
                where = fr.split("/repo/")[-1]
                break
        else:
            if e.stack:
                where = "[autograd/torch] " + e.stack[0][-60:]
        k = sum(1 for kk in e.kernels)
        groups[(e.name, where)] += 1
        dur[(e.name, where)] += sum(kk.duration for kk in e.kernels)
    print("%-22s %5s %9s  %s" % ("op", "calls", "gpu us", "where"))
    for (n, w), c in sorted(groups.items(), key=lambda t: -dur[t[0]])[:90]:
        print("%-22s %5d %9.1f  %s" % (n, c, dur[(n, w)], w))
    if os.environ.get("BIG", "0") == "1":
        # every aten op that directly owns >= 15 us of kernels, with shapes and the enclosing ops
        rows = []
        for e in ev:
            if not e.name.startswith("aten::") or not e.kernels:
                continue
            d = sum(kk.duration for kk in e.kernels)
            if d < 15:
                continue
            names, p = [], e.cpu_parent
            while p is not None and len(names) < 4:
                names.append(p.name[:40]); p = p.cpu_parent
            rows.append((d, e.name, str(e.input_shapes)[:70], " < ".join(names)))
        print("\naten ops owning >= 15 us of kernels:")
        for d, n, sh, ch in sorted(rows, reverse=True):
            print("%7.1f  %-22s %-70s %s" % (d, n, sh, ch))
    # device copies / fills: who launches them (chain of enclosing CPU ops)
    chain = collections.Counter()
    for e in ev:
        for kk in (e.kernels or []):
            nm = kk.name
            if any(t in nm for t in ("copyBuffer", "Memcpy", "Memset", "fillBuffer", "FillFunctor")):
                names, p = [e.name], e.cpu_parent
                while p is not None and len(names) < 5:
                    names.append(p.name); p = p.cpu_parent
                chain[(nm[:28], " < ".join(n[:48] for n in names))] += 1
    print("\ndevice copies / fills by enclosing ops:")
    for (nm, c), n in sorted(chain.items(), key=lambda t: -t[1])[:60]:
        print("%4d  %-28s %s" % (n, nm, c))
    tot = collections.Counter()
    for e in ev:
        if e.device_type is not None and str(e.device_type).endswith("CUDA"):
            tot[e.name[:90]] += 1
    print("\nkernel launches in this update:", sum(tot.values()))


if __name__ == "__main__":
    main()
