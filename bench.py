#!/usr/bin/env python3
"""bench.py — policy steps/sec (fwd+bwd) of WS-MGMap's teacher-forcing update on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" of this script = one policy UPDATE over a batch of T=64 x N=8 synthetic cached-
feature trajectories per GPU (BASELINE.json configs[1] / SURVEY.md §8d cfg2): zero_grad +
BasePolicy.forward + DAgger loss + backward (+ gradient all-reduce over RCCL for N>1) + Adam.
`value` = policy steps (rows) per second over all ranks = T*N*world*K / t.  Inputs are
device-resident before the timed region.  Prints ONE JSON line on rank 0.

The line carries `roofline` for the dominant kernel (live HIP-event timing of the conv-engine
launches inside the timed region) and `cpu_baseline` (the oracle — a PyTorch-CPU port of the
reference's update step — timed on this box's host cores on a bounded sample, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_F32_TFLOPS = 157.3      # MI355X f32 MFMA / vector peak (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF figure is 2:1 sparse)
ALG_GFLOP_PER_STEP = 10.017  # 3.776 fwd + 6.241 bwd per policy step


class _Box:
    shape = (2,)


def synth_batch(T, N, device, seed, E=100, C=64):
    """cfg2 inputs, time-major rows (row = t*N + n)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    B = T * N
    instr = torch.zeros(N, 200)
    instr[:, :80] = torch.randint(1, 2504, (N, 80), generator=g).float()
    obs = {
        "instruction": instr.repeat(T, 1),
        "rgb_features": torch.randn(B, 512, 7, 7, generator=g),
        "depth_features": torch.randn(B, 128, 4, 4, generator=g),
        "rgb_ego_map": torch.relu(torch.randn(B, C, E, E, generator=g)),
        "gt_semantic_map": torch.randint(0, 27, (B, E, E), generator=g).float(),
        "gt_path": torch.rand(B, E, E, generator=g) * 50.0,
        "progress": torch.rand(B, 1, generator=g),
        "waypoint": torch.rand(B, 3, generator=g) * 2 - 1,
    }
    masks = torch.ones(T, N)
    masks[0] = 0
    weights = torch.ones(T, N)
    prev = torch.zeros(B, 2)
    obs = {k: v.to(device) for k, v in obs.items()}
    return obs, prev.to(device), masks.view(B, 1).to(device), weights.to(device)


def dagger_loss(pred, aux_loss, waypoint, weights):
    """The trainer's loss (dagger_trainer.py:526-534).  GPU: wsmgmap.losses.dagger_loss, the same arithmetic as one launch per
    direction (WSMG_FUSED_LOSS=0: the reference's torch lines below, 11 launches each way); CPU tensors: the torch lines."""
    if pred.is_cuda and os.environ.get("WSMG_FUSED_LOSS", "1") != "0":
        from wsmgmap.losses import dagger_loss as fused
        return fused(pred, aux_loss, waypoint, weights)[0]
    T, N = weights.shape
    logits = torch.tanh(pred).view(T, N, -1)
    al = F.mse_loss(logits, waypoint[:, :2].view(T, N, -1), reduction="none").sum(2)
    al = ((weights * al).sum(0) / weights.sum(0)).mean()
    return al + aux_loss


def host_cpu_info():
    """(model string, physical cores, logical CPUs) of this host, from /proc/cpuinfo."""
    model, phys, logical = "unknown", set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None and cid is not None:
                phys.add((pid, cid))
                pid = cid = None
        if pid is not None and cid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return model, (len(phys) or logical or os.cpu_count() or 1), (logical or os.cpu_count() or 1)


def visible_gpus_without_hip():
    """GPUs this process would see, WITHOUT initialising HIP: compute nodes of the amdgpu driver's topology
    (/sys/class/kfd/kfd/topology/nodes/*/properties with simd_count > 0), narrowed by ROCR_VISIBLE_DEVICES /
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None if the topology cannot be read (the ranks check themselves)."""
    import glob
    n = 0
    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        return None
    for f in files:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def cgroup_cpu_quota():
    """CPUs' worth of time the control group of this process may use (cgroup v2 cpu.max, v1 cfs quota), or None if unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(state, T_full, N, budget_s=90.0):
    """The oracle's update step (fwd + loss + bwd + Adam, PyTorch CPU fp32) on the host cores (one thread per
    physical core, >= 1 warm-up + >= 2 timed updates, CPU model and core count in the report).
    Sample: the same workload at the largest T in {T_full, 32, 16, ...} whose TWO timed updates are expected to fit `budget_s`."""
    from oracle import policy_ref
    model, phys, logical = host_cpu_info()
    # one thread per physical core — of those the process may actually use: under a control-group CPU quota (the GPU boxes of this
    # pool: 16 CPUs on a 128-core host) more threads than that only take turns being throttled
    quota = cgroup_cpu_quota()
    nthreads = max(1, phys if quota is None else min(phys, int(quota + 0.999)))
    if os.environ.get("WSMG_CPU_THREADS"):
        nthreads = int(os.environ["WSMG_CPU_THREADS"])
    torch.set_num_threads(nthreads)
    cores = torch.get_num_threads()

    def make_P():
        P = {}
        for k, v in state.items():
            t = v.detach().cpu().clone()
            P[k] = t
        return P

    def run(T):
        P = make_P()
        leaves = {}
        for k, t in P.items():
            if t.is_floating_point() and not k.startswith(("net.rgb_encoder", "net.instruction_encoder.embedding")) \
                    and "running_" not in k and k != "net._scale":
                t.requires_grad_(True)
                leaves[id(t)] = t
        opt = torch.optim.Adam(list(leaves.values()), lr=2.5e-4)
        ref = policy_ref.PolicyRef(P, num_proc=1)
        ref.aux_active = True
        obs, prev, masks, weights = synth_batch(T, N, "cpu", 1234)
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        pred, aux, _, _ = ref.forward(obs, torch.zeros(2, N, 512), prev, masks, weights)
        loss, _ = policy_ref.dagger_loss(pred, aux, obs["waypoint"], weights)
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    run(1)  # warm-up (thread pools, allocator)
    t4 = run(4)
    per_t = t4 / 4
    # SURVEY.md 8d / BASELINE.md 3: the same cfg2 shapes (T=64, B=512) when the host does them within the budget, else
    # T=16 (B=128) scaled linearly, stated in `sample`; smaller only if even that does not fit.  The time per policy step
    # falls with the batch (BatchNorm / conv efficiency), so the estimate from T=4 is an upper bound.
    reps = 2
    T = None
    for cand in (T_full, 32, 16, 8):
        if cand <= T_full and per_t * cand * reps <= budget_s:
            T = cand
            break
    if T is None:
        T = int(max(2, min(T_full, budget_s / reps / max(per_t, 1e-3))))
    dts = [run(T) for _ in range(reps)]
    dt = sum(dts) / len(dts)
    scaled = "" if T == T_full else f" — T={T} instead of {T_full}: policy steps/s assumed linear in T"
    return dict(value=T * N / dt, unit="policy steps/s", cores=cores, kind="port", cpu_model=model, physical_cores=phys,
                logical_cpus=logical, cgroup_cpu_quota=quota, timed_updates=len(dts), seconds_per_update=[round(d, 2) for d in dts],
                sample=f"{len(dts)} updates of T={T} x N={N} ({T * N} policy steps each; {', '.join('%.1f s' % d for d in dts)}; mean) "
                       f"of the oracle (PyTorch-CPU fp32 port of the reference update: fwd+loss+bwd+Adam) after a warm-up, "
                       f"{cores} threads on {model} ({phys} physical cores"
                       + (f", control-group quota {quota:g} CPUs" if quota is not None else "") + f"){scaled}")


def measure(args, dtype, steps, warmup, rank, world, local, dev):
    """Build the policy in `dtype` mode and time `steps` updates; returns (dt, prof, loss, state_cpu)."""
    from wsmgmap import ops
    from wsmgmap.common.aux_losses import AuxLosses
    from wsmgmap.config import default_model_config
    from wsmgmap.models.policy import BasePolicy
    from wsmgmap.parallel import GradAllReducer

    torch.manual_seed(0)
    policy = BasePolicy(None, _Box(), default_model_config(num_proc=1, gpu_id=local, compute_dtype=dtype))
    policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)  # reference default: frozen embeddings
    state_cpu = {k: v.detach().clone() for k, v in policy.state_dict().items()} if rank == 0 else None
    policy = policy.to(dev)
    policy.train()
    policy.net.depth_encoder.eval()
    policy.net.rgb_encoder.eval()
    # the reference's torch.optim.Adam (common_trainer.py:67-69) as one multi-tensor HIP launch per 48 tensors; WSMG_STOCK_ADAM=1: stock
    from wsmgmap.optim import Adam as WsmgAdam
    opt = (torch.optim.Adam if os.environ.get("WSMG_STOCK_ADAM") == "1" else WsmgAdam)(policy.parameters(), lr=2.5e-4)
    reducer = GradAllReducer(policy.parameters(), bucket_bytes=int(float(os.environ.get("WSMG_DP_BUCKET_MB", "8")) * (1 << 20)),
                             single_rank_exchange=True,
                             # round 5: the exchange on the policy's instruction-branch stream, synchronous RCCL collectives (no stream
                             # beyond the policy's own; profiles/r05_dp_exchange_ab.txt); WSMG_DP_EXCHANGE=hook: the round-2 form
                             exchange_stream={"hook": None, "": None}.get(os.environ.get("WSMG_DP_EXCHANGE", "instruction"),
                                                                          os.environ.get("WSMG_DP_EXCHANGE", "instruction"))) if args.dp else None
    if reducer:
        reducer.broadcast_parameters(policy)
    measure.dp_info = None

    T, N = args.T, args.N
    obs, prev, masks, weights = synth_batch(T, N, dev, 1000 + rank)
    if getattr(measure, "ego_layout", None) == "feeder":
        # the secondary `feeder_layout_mode` leg only: the ego map as wsmgmap's own feeder hands it over (DeviceFeeder(...,
        # ego_map_nhwc_bf16=True): channels-last bf16 storage behind the same [B,C,E,E] shape) instead of the reference collate's float32
        # NCHW — the same values after the rounding the layout pass applies anyway, so the loss must come out identical
        obs["rgb_ego_map"] = obs["rgb_ego_map"].to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    # the synthetic batch is resident and complete from here on: say so (what DeviceCollator says of every batch it hands over), so that
    # the forward's one host read-back — the instruction dedup, still launched and read back in every update — need not wait for
    # the previous update to finish on the GPU (wsmgmap/ops/core.py, "input readiness")
    ops.mark_inputs_ready(obs["instruction"])
    AuxLosses.activate()

    def update():
        opt.zero_grad(set_to_none=True)
        AuxLosses.clear()
        h0 = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
        o = dict(obs)  # forward mutates the dict (rgb_ego_map key) like the reference
        pred, aux = policy(o, h0, prev, masks, weights)
        loss = dagger_loss(pred, aux, o["waypoint"], weights)
        loss.backward()
        if reducer:
            reducer.finish()
        opt.step()
        return loss

    # ---- round 6: a persistent-kernel timeout must not cost the run its line --------------------------------
    # The two recurrences and the instruction LSTM are persistent kernels whose workgroups wait for each other with BOUNDED spins;
    # the chained recurrent core adds kernels that spin on device counters.  Beside real RCCL ring kernels (never run from the build
    # environment: its boxes have one GPU) a workgroup that is not resident in time makes a spin run out: the kernel fills its outputs
    # with NaN and sets a bit in a host-mapped status word.  Everything from the probe below to the end of the timed region runs inside
    # `guarded`: if ANY rank saw a timeout (its own status word, or the exchange's cross-rank error flag) ALL ranks together — one
    # 4-byte MAX all-reduce — switch IN PROCESS to the next more conservative recurrent core, restore the state saved before the probe,
    # re-discover the live gradient set and run the phase again.  Level 1: the staged core (one persistent kernel at a time, no
    # device-side chaining, no decoder side stream); level 2: the stock (MIOpen) GRU / LSTM, no persistent kernel at all.  The line
    # says which one ran (`data_parallel.recurrent_core` / `recurrent_core`).  WSMG_BENCH_INJECT_TIMEOUT=n: test hook — the status bit
    # is injected once, after n updates of the first attempt.
    from wsmgmap import _abi, debug
    from wsmgmap.fallback import RecurrentCoreFallback
    fbk = RecurrentCoreFallback(policy, opt, reducer, verbose=(rank == 0))
    guarded = fbk.guarded
    inject = [int(os.environ.get("WSMG_BENCH_INJECT_TIMEOUT", "0") or 0)]
    n_upd = [0]
    _update_inner = update

    def update():     # noqa: F811 — the same update, counted (the injection hook)
        loss_ = _update_inner()
        n_upd[0] += 1
        if inject[0] and n_upd[0] == inject[0] and (rank == 0 or not args.dp):
            inject[0] = 0
            _abi.lib().wsmg_rnn_debug_inject(1)
        return loss_

    def probe():
        for _ in range(3):
            update()
    guarded(probe)

    def phases():
        # N > 1: the decoder's side stream beside RCCL's own streams was never run on the target (no multi-GPU box reachable from the
        # build environment), and on a SHARED GPU a third stream per process once took updates from 49 ms to 4.3 s.  So it is
        # measured before it is trusted: 3 updates with and 3 without it (after one untimed update each), max over ranks; more than
        # 1.3x slower with the stream -> fall back to one stream for the run and say so in the line.
        measure.side_stream = None
        from wsmgmap import debug
        if args.dp and os.environ.get("WSMG_DECODER_STREAMS") is None and fbk.level == 0:      # (a fallback level keeps the side stream off)
            def timed(k):
                update()
                torch.cuda.synchronize()
                dist.barrier()
                t_ = time.perf_counter()
                for _ in range(k):
                    update()
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t_], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item()) / k
            update()                                   # discovery pass of the gradient exchange, allocator warm-up
            debug.sw.decoder_streams = 0
            t_one = timed(3)
            debug.sw.decoder_streams = 1
            t_two = timed(3)
            keep = t_two <= 1.3 * t_one
            if not keep:
                debug.sw.decoder_streams = 0
            measure.side_stream = dict(ms_per_update_one_stream=round(t_one * 1e3, 3), ms_per_update_with_decoder_side_stream=round(t_two * 1e3, 3),
                                       decoder_side_stream_used=bool(keep),
                                       note="3 updates each, max over ranks; the side stream is dropped for the run when it is > 1.3x slower")
            if reducer:
                reducer.stats(reset=True)

        # which conv-engine kernel family dominates is learned on the warm-up updates (all six entry points timed); inside
        # the timed region only that family is bracketed with HIP events — every timed launch costs two event records on a
        # host that is within 10 % of being the bottleneck, and `value` should not pay for the other five
        warm_prof = {}
        for i in range(warmup):
            if i == warmup - 1:
                ops.profile_begin()      # the last warm-up update: first-launch effects are over
            update()
        if warmup > 0:
            warm_prof = ops.profile_end()
        dom_entry = None
        if warm_prof:
            # the family with the most time in the last warm-up update, no tie rule (rounds 2-3 gave ties within 15 % to
            # the backward-weight family); `kernels` carries every family's figures either way
            dom_entry = max(warm_prof.values(), key=lambda r: r["ms_total"])["entry"]
        import gc
        # The host's cyclic garbage collector and the timed region.  The collector stays ON (off, the autograd graphs — reference cycles —
        # give their tensors back late and the allocator grows: 2 / 4 / 23 windows more than 3 % over the median and 11.6 ms per update in
        # the third run of profiles/r04_gc_ab.txt).  But a full (generation-2) collection walks every tracked object of the process — the
        # import-time heap of torch included — and takes 100-170 ms here; it comes once every ~360 updates (profiles/
        # r04_host_stalls_and_collections.txt: update 334-338 of every 500-update run, one window at 31-46 ms).  The heap that exists after
        # the warm-up is therefore moved to the permanent generation (gc.freeze) after one full collection: later collections look at what
        # the updates themselves allocate, and a full one takes about a millisecond.  WSMG_BENCH_GC=plain: no freeze; =0: collector off.
        # (Done HERE, before the pre-timing spin: the full collection takes ~110 ms of host time, the GPU drains and clocks down meanwhile, and
        # right in front of the timed region that showed as a first window 0.3-0.5 ms per update above the rest.)
        gc_mode = os.environ.get("WSMG_BENCH_GC", "freeze")
        if gc_mode == "0":
            gc.collect()
            gc.disable()
        elif gc_mode == "freeze":
            gc.collect()
            gc.freeze()
        # Pre-timing spin of real updates (reported as `prewarm_s` / `prewarm_updates`; `warmup` stays what the caller passed): on a
        # fresh lease the first ~20 updates after 5 warm-up ones ran 14.6 / 11.8 ms against a steady 11.4 (BENCH_r03 `windows`) — the GPU
        # comes out of idle clocks, the allocator is still growing after profile_end(), the per-stream workspaces see first use.
        # The updates are the timed region's own; all ranks agree on when to stop (the exchange is a collective).
        measure.prewarm = None
        prewarm_s = args.prewarm_s if dtype == args.dtype else min(args.prewarm_s, 0.5)   # the extra float32 leg: a short one
        if prewarm_s > 0:
            tp = time.perf_counter()
            n_pw = 0
            while True:
                for _ in range(4):
                    update()
                n_pw += 4
                torch.cuda.synchronize()
                go = 1.0 if time.perf_counter() - tp < prewarm_s else 0.0
                if args.dp:
                    tg = torch.tensor([go], device=dev)
                    dist.all_reduce(tg, op=dist.ReduceOp.MIN)
                    go = float(tg.item())
                if go == 0.0:
                    break
            measure.prewarm = dict(seconds=round(time.perf_counter() - tp, 3), updates=n_pw)
        if args.dp:
            dist.barrier()
        torch.cuda.synchronize()
        if reducer:
            reducer.stats(reset=True)
            if os.environ.get("WSMG_DP_EXCHANGE", "instruction") not in ("hook", ""):
                reducer.time_buckets(True)
        # timed live (HIP events around every launch, inside the timed region): the family with the most time, and the weight-gradient
        # family whichever it is — rounds 1-3 reported that one, and its kernels are what round 4 rebuilt; a line must show both
        WG = "wsmg_conv2d_bwd_weight_bf16" if dtype.startswith("bf16") else "wsmg_conv2d_bwd_weight"
        FW = "wsmg_conv2d_fwd_bf16" if dtype.startswith("bf16") else "wsmg_conv2d_fwd"      # (the *_stats entry points fold into it: ops._prof_key)
        # Every timed launch is two HIP event records on its stream (~1.5 us of GPU time each): with both families timed on every update
        # the line itself cost 0.1-0.2 ms per update (10.58 / 10.67 vs 10.46 / 10.46 ms, profiles/r04_first_window_and_event_cost.txt).
        # So: the largest family on every update, as in rounds 1-3; the weight-gradient family, when it is not the largest, on every
        # 4th update of the timed region (a sample of the same region).  WSMG_BENCH_NOPROF=1: no events at all (diagnostic).
        # round 5: BOTH named families are in every line, whichever of them the warm-up found larger — the larger
        # one timed on every update, the other on every 4th; a third family, should it ever lead, is timed on every update beside them
        both = sorted({dom_entry, WG, FW}) if dom_entry else None
        # round 5, end: the largest family on every update still cost the line 0.12 ms per update (10.42 against 10.30 ms without any
        # event, one box, interleaved: profiles/r05_bench_line_cost.txt) — BOTH families are now timed on every 4th update of the timed
        # region only (10.38); WSMG_BENCH_PROF_EVERY=1 restores the largest family on every update
        prof_every = int(os.environ.get("WSMG_BENCH_PROF_EVERY", "4"))
        if os.environ.get("WSMG_BENCH_NOPROF") != "1":
            ops.profile_begin(only=[dom_entry] if dom_entry else None)
        # an event every WIN updates (no synchronisation): `sustained` for --steps >= 200 (50-update windows), and `windows` for every
        # run (a quarter of the run each), so that a transient stall inside the timed region — one run in ~40 on this pool came out at
        # 14-19 ms per update for no reason the process could see — shows in the line as what it is
        WIN = 50 if steps >= 200 else max(1, (steps + 3) // 4)
        if os.environ.get("WSMG_BENCH_WINDOW"):        # diagnostic: another window length (1 = an event per update)
            WIN = max(1, int(os.environ["WSMG_BENCH_WINDOW"]))
        marks = []
        gc_log, host_each = [], []
        if os.environ.get("WSMG_BENCH_HOSTTIME") == "2":
            def _gc_cb(phase, info, _t=[0.0]):
                if phase == "start":
                    _t[0] = time.perf_counter()
                else:
                    gc_log.append((len(host_each), info.get("generation"), round((time.perf_counter() - _t[0]) * 1e3, 2), info.get("collected")))
            gc.callbacks.append(_gc_cb)
        t0 = time.perf_counter()
        host = 0.0
        for i in range(steps):
            if i % WIN == 0:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            if dom_entry:
                ops.profile_set_only(both if i % 4 == 0 else ([dom_entry] if prof_every == 1 else ["-"]))
            h0 = time.perf_counter()
            loss = update()
            host += time.perf_counter() - h0
            host_each.append(time.perf_counter() - h0)
        if marks:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append(ev)
        torch.cuda.synchronize()
        measure.sustained = measure.windows = None
        measure.host_ms = round(host / steps * 1e3, 3)      # wall time of the update() calls of the timed region, per update
        if len(marks) >= 2:
            spans = [(min(WIN, steps - j * WIN), marks[j].elapsed_time(marks[j + 1])) for j in range(len(marks) - 1)]
            per = [ms / n for n, ms in spans if n > 0]
            measure.windows = dict(window=WIN, ms_per_update_by_window=[round(v, 3) for v in per],
                                   note="HIP-event time of consecutive windows of the timed region (rank 0's stream)")
        if len(marks) >= 3 and steps >= 200:
            tail = per[len(per) // 2:]
            measure.sustained = dict(updates=steps, window=WIN, ms_per_update_by_window=[round(v, 3) for v in per],
                                     ms_per_update_first_window=round(per[0], 3),
                                     ms_per_update_second_half=round(sum(tail) / len(tail), 3),
                                     note="HIP-event time of consecutive 50-update windows inside the same timed region")
        gc.enable()
        gc.unfreeze()
        if gc_log or os.environ.get("WSMG_BENCH_HOSTTIME") == "2":
            gc.callbacks[:] = [c for c in gc.callbacks if getattr(c, "__name__", "") != "_gc_cb"]
            if rank == 0:
                med = sorted(host_each)[len(host_each) // 2]
                slow = [(i, round(v * 1e3, 2)) for i, v in enumerate(host_each) if v > 1.5 * med]
                print("host per update: median %.2f ms; updates over 1.5 x median: %s" % (med * 1e3, slow[:40]), file=sys.stderr)
                print("collections (update index, generation, ms, collected): %s" % [g for g in gc_log if g[1] >= 1 or g[2] > 1.0][:60], file=sys.stderr)
        if os.environ.get("WSMG_BENCH_HOSTTIME") == "1" and rank == 0:   # diagnostic: how long the host needs to ENQUEUE one update
            print("host enqueue time %.3f ms per update" % (host / steps * 1e3), file=sys.stderr)
        if args.dp:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = ops.profile_end()
        if dom_entry:
            for r in prof.values():       # the sampled families: their totals are over every 4th update
                if r.get("entry") in (WG, FW) and (r.get("entry") != dom_entry or prof_every != 1):
                    r["per_steps"] = (steps + 3) // 4
                    r["phase"] = "timed region"
                    r["sampled"] = "every 4th update of the timed region"
        ops.check_rnn_status()        # a persistent-RNN timeout anywhere in the run invalidates it: fail loudly
        if reducer:
            reducer.check()
            st = reducer.stats()
            measure.dp_info = dict(ranks_in_process_group=dist.get_world_size(), backend=dist.get_backend(),
                                   gpu_max_hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"),
                                   exchange=os.environ.get("WSMG_DP_EXCHANGE", "instruction"), early_dedup=bool(debug.sw.early_dedup_dp),
                                   live_gradient_bytes=reducer.live_bytes, buckets=reducer.num_buckets,
                                   devices_visible=torch.cuda.device_count(),
                                   exposed_allreduce_ms=st["exposed_allreduce_ms"], exposed_allreduce_max_ms=st["exposed_allreduce_max_ms"],
                                   host_ms_in_finish=st["host_ms_in_finish"], updates_measured=st["updates"],
                                   per_bucket_allreduce_ms=st.get("per_bucket_allreduce_ms"),
                                   per_bucket_note="HIP-event time of each bucket's pack + all-reduce on the exchange stream, mean over the timed "
                                                   "region (a ring step's time, the peers' skew included); rank 0",
                                   exposed_note="HIP-event time on the compute stream between the end of backward (entry of finish()) and the "
                                                "last averaged bucket: the part of the gradient all-reduce that backward did not hide; rank 0",
                                   side_stream_check=measure.side_stream)
        for name, r in warm_prof.items():     # the other families: per-launch figures from the warm-up updates, labelled so
            if name not in prof:
                prof[name] = dict(r, per_steps=1, phase="last warm-up update")
        if args.dp:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, prof, loss, warm_prof

    dt, prof, loss, warm_prof = guarded(phases)
    measure.fallback = fbk.report()
    final_loss = float(loss.detach())
    # The same update captured into ONE HIP graph and replayed (wsmgmap.graph.GraphedUpdate): reported beside the eager figure,
    # never as `value` — the roofline object is measured with HIP events around eager launches inside the timed region, which
    # a graph replay has none of.  One process only; opt-in (WSMG_BENCH_GRAPH=1): at the bench size the replay takes the same
    # time as the eager update (12.1 ms both; it wins below 256 rows), and a failure in this extra phase must never cost the
    # run its line.
    measure.graphed = None
    if world == 1 and dtype.startswith("bf16") and os.environ.get("WSMG_BENCH_GRAPH", "0") == "1":
        from wsmgmap.graph import GraphedUpdate
        del loss                    # the last eager update's autograd graph (and its AccumulateGrad nodes) must be gone before a capture
        opt.zero_grad(set_to_none=True)
        gopt = WsmgAdam(policy.parameters(), lr=2.5e-4, capturable=True)
        gu = GraphedUpdate(policy, gopt, lambda pred, aux, o, w: dagger_loss(pred, aux, o["waypoint"], w), eager_calls=2)
        gu.register_static_inputs(obs, prev, masks, weights)   # the bench refills nothing: the graph reads the batch in place
        hs = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)

        def gupdate():
            hs.zero_()
            return gu(obs, hs, prev, masks, weights)
        for _ in range(4):          # 2 eager, capture + first replay, one more replay
            gupdate()
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        for _ in range(steps):
            gloss = gupdate()
        torch.cuda.synchronize()
        gdt = time.perf_counter() - g0
        ops.check_rnn_status()
        measure.graphed = dict(ms_per_step=round(gdt / steps * 1e3, 3), value=round(T * N * steps / gdt, 2), unit="policy steps/s", steps=steps,
                               loss=round(float(gloss), 5),
                               note="the same update (zero_grad + forward + loss + backward + Adam) as one captured HIP graph per "
                                    "input signature, replayed; only the instruction dedup stays eager.  The batch is registered as the "
                                    "graph's static inputs: a trainer fed by the feeder pays one more copy of it per update (the "
                                    "cached ego map alone is 1.3 GB), which this figure does not contain")
    if reducer:
        reducer.close()
    del fbk, guarded
    del policy, opt, obs
    torch.cuda.empty_cache()
    return dt, prof, final_loss, state_cpu


def main():
    # The contract is ONE JSON line on rank 0's stdout.  Libraries write there too — RCCL prints a five-line version banner through
    # C stdio when a communicator is created, and being block-buffered on a pipe it lands AFTER Python's own line, at exit — so
    # file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved original descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # Data-parallel runs: RCCL brings its own streams, and HIP multiplexes a process's streams onto 4 hardware queues by default —
    # the update's three compute streams then share queues with the collective library's.  Measured with a ONE-rank communicator
    # (nothing but the process group existing): 11.78 ms per update against 11.45 without it, and 11.45 again with 8 queues; the
    # whole exchange 12.07 -> 11.65 ms.  Must be in the environment before the HIP runtime starts (nothing has touched it yet).
    # Not when ranks SHARE a GPU (the functional mode of a 1-GPU box): two processes with 8 queues each oversubscribe the GPU's
    # queues outright — 609 ms per update instead of 44.
    if ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("WSMG_BENCH_DP_ONE_RANK") == "1")
            and os.environ.get("WSMG_BENCH_SHARE_GPU") != "1"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--N", type=int, default=8)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "bf16+f32grad"],
                    help="storage/MFMA type of the map stack: bf16 = BASELINE configs[1] (default); f32 = parity mode (1e-4 vs reference); "
                         "bf16+f32grad = bf16 with the three first-of-chain weight gradients from a 16-bit-mantissa dY (opt-in, round 6)")
    ap.add_argument("--prewarm-s", type=float, default=float(os.environ.get("WSMG_BENCH_PREWARM_S", "1.5")),
                    help="seconds of untimed updates between the warm-up and the timed region (clock ramp / allocator steady state); 0 = none")
    ap.add_argument("--no-f32", action="store_true", help="skip the extra float32 parity-mode measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the cfg1 / cfg4 / cfg5 legs (BASELINE configs[0], [3], [4]; bench_legs.py, ~2 s after the cfg2 leg)")
    ap.add_argument("--cpu-budget", type=float, default=90.0,
                    help="seconds the host may spend on the TWO timed CPU-baseline updates: T=64 if they fit, else T=32 / 16")
    args = ap.parse_args()

    share = os.environ.get("WSMG_BENCH_SHARE_GPU") == "1"
    if "WORLD_SIZE" not in os.environ:
        if args.gpus is None:
            args.gpus = 1
        if args.gpus > 1:
            # `python bench.py --gpus N` without a launcher: start N ranks (one process per GPU, as the reference does with
            # torch.distributed.launch, README.md:80-84 / common_trainer.py:35-38) as CHILD processes and relay rank 0's
            # line.  This parent makes NO torch.cuda / HIP call at all (on this pool a process that has initialised the GPU
            # must not be the ancestor of an exec; the children are what touches the device): GPUs are counted from the
            # driver's topology files and the *_VISIBLE_DEVICES lists, and every child validates its own device.
            have = visible_gpus_without_hip()
            if have is not None and have < args.gpus and not share:
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
            import socket
            import subprocess
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("OMP_NUM_THREADS", "8")
            raise SystemExit(subprocess.run(cmd, env=env, stdout=real_stdout).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is None:
        args.gpus = world
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only compute path")
    if world > torch.cuda.device_count() and not share:
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPU(s) visible: one process per GPU "
                         "(WSMG_BENCH_SHARE_GPU=1 lets ranks share a GPU for a functional test)")
    local = local % max(1, torch.cuda.device_count()) if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # WSMG_BENCH_DP_ONE_RANK=1: the data-parallel code path (process group, gradient exchange, its statistics, the side-stream check)
    # with ONE rank — a functional run of what `--gpus N` does, over the real backend, on a box with one GPU
    args.dp = world > 1 or os.environ.get("WSMG_BENCH_DP_ONE_RANK") == "1"
    if args.dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("WSMG_BENCH_BACKEND", "nccl")  # "gloo": functional test of the DP path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    T, N = args.T, args.N
    dt, prof, final_loss, state_cpu = measure(args, args.dtype, args.steps, args.warmup, rank, world, local, dev)
    measure.host_ms_main = getattr(measure, "host_ms", None)
    dp_info = measure.dp_info
    fallback = getattr(measure, "fallback", None)
    if dp_info is not None and fallback is not None:
        dp_info["recurrent_core"] = fallback["recurrent_core"]
        dp_info["fallback"] = fallback
    graphed = getattr(measure, "graphed", None)
    sustained = getattr(measure, "sustained", None)
    windows = getattr(measure, "windows", None)
    prewarm = getattr(measure, "prewarm", None)
    parity = None
    if args.dtype.startswith("bf16") and not args.no_f32:
        k32 = max(2, args.steps // 2)
        dt32, _, loss32, _ = measure(args, "f32", k32, 2, rank, world, local, dev)
        parity = dict(dtype="f32", value=round(T * N * world * k32 / dt32, 2), unit="policy steps/s",
                      ms_per_step=round(dt32 / k32 * 1e3, 3), steps=k32, loss=round(loss32, 5),
                      note="same workload in the float32 parity mode (f32 MFMA; logits within 1e-4 of the reference)")
    f32grad = None
    if args.dtype == "bf16" and not args.no_f32 and world == 1:
        kg = max(2, args.steps // 2)
        dtg, _, lossg, _ = measure(args, "bf16+f32grad", kg, 2, rank, world, local, dev)
        f32grad = dict(dtype="bf16+f32grad", value=round(T * N * world * kg / dtg, 2), unit="policy steps/s", ms_per_step=round(dtg / kg * 1e3, 3),
                       steps=kg, loss=round(lossg, 5),
                       note="opt-in COMPUTE_DTYPE: bf16 with the weight gradients of map_encoder.cnn.0, map_decoder.base_model.conv1 and "
                            "conv_original_size0 from a 16-mantissa-bit dY (two bf16 weight-gradient launches each); what it buys over 200 "
                            "updates: profiles/r06_bf16_f32grad_200_updates.txt")

    feeder_layout = None
    if args.dtype == "bf16" and not args.no_f32 and world == 1:
        kf = max(2, args.steps // 2)
        measure.ego_layout = "feeder"
        try:
            dtf, _, lossf, _ = measure(args, "bf16", kf, 2, rank, world, local, dev)
        finally:
            measure.ego_layout = None
        feeder_layout = dict(dtype="bf16", value=round(T * N * world * kf / dtf, 2), unit="policy steps/s", ms_per_step=round(dtf / kf * 1e3, 3),
                             steps=kf, loss=round(lossf, 5),
                             note="NOT the headline: the same update with `rgb_ego_map` handed over the way wsmgmap's own feeder does "
                                  "(DeviceFeeder(ego_map_nhwc_bf16=True): channels-last bf16 storage, same shape and values) — the update then has "
                                  "no layout / conversion pass over the cached ego map (0.37 ms, 2.3 GB).  `value` above takes the reference "
                                  "collate's float32 NCHW tensor")

    # BASELINE configs[0], [3], [4] as short HIP-event-timed legs in the same line:
    # single process only — they are single-GPU measurements — after every cfg2 measurement, outside any timed region
    other = None
    if world == 1 and not args.dp and not args.no_other_configs:
        import bench_legs
        other = bench_legs.other_configs(dev)

    if rank == 0:
        steps_per_s = T * N * world * args.steps / dt
        kernels = {}
        for name, r in prof.items():
            avg_ms = r["ms_total"] / max(r["launches"], 1)
            per = r.get("per_steps", args.steps)
            kernels[name] = dict(launches=r["launches"], avg_ms=avg_ms, ms_per_update=r["ms_total"] / max(per, 1),
                                 tflops=r["flops_total"] / (r["ms_total"] * 1e-3) / 1e12 if r["ms_total"] > 0 else 0.0,
                                 measured_in=r.get("phase", "timed region"))
        timed = [k for k in kernels if kernels[k]["measured_in"] == "timed region"]
        dom = max(timed, key=lambda k: kernels[k]["ms_per_update"]) if timed else None
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        import glob
        tfiles = sorted(glob.glob(os.path.join(pdir, "r*_bench_bf16_hbm_traffic.json")))
        mfiles = sorted(glob.glob(os.path.join(pdir, "r*_bench_bf16_mfma_busy.json")))

        def roofline_of(fam):
            r = prof[fam]
            ach = r["flops_total"] / (r["ms_total"] * 1e-3) / 1e12
            peak = PEAK_BF16_TFLOPS if args.dtype.startswith("bf16") else PEAK_F32_TFLOPS
            o = dict(bound="mfma", kernel=fam, achieved=round(ach, 3), peak=peak, unit="TFLOP/s", timed=r.get("sampled", "every update of the timed region"),
                     frac=round(ach / peak, 4), traffic=None,
                     avg_launch_ms=round(r["ms_total"] / r["launches"], 4), launches=r["launches"],
                     alg_gflop_per_launch=round(r["flops_total"] / r["launches"] / 1e9, 3))
            # HBM bytes per launch and MFMA-pipe utilisation of the same kernel family, from the committed rocprofv3 --pmc
            # passes over this very command (FETCH_SIZE, WRITE_SIZE and the SQ/GRBM counters in separate runs, folded by
            # tools/pmc_traffic.py / tools/pmc_mfma.py; tools/refresh_profiles.sh): PMC collection cannot run inside a timed
            # bench, so the figures are read from profiles/ (newest round), not measured live
            fams = {"wsmg_conv2d_bwd_weight_bf16": ["conv_wgrad_bf16_kernel", "conv_win_wgrad_kernel", "conv_win3_wgrad_kernel", "conv_s2_wgrad_kernel"],
                    # (conv_win3_kernel / conv_win3_mixed_kernel serve forward AND backward-data: the counter fold cannot tell the two
                    #  directions of one kernel symbol apart, so both families' folds include all of its launches)
                    "wsmg_conv2d_fwd_bf16": ["conv_igemm_bf16_kernel<false, *>", "conv_win_fwd_kernel", "conv_win3_kernel", "conv_win3_mixed_kernel"],
                    "wsmg_conv2d_fwd_bf16_stats": ["conv_igemm_bf16_kernel<false, *>", "conv_win_fwd_kernel", "conv_win3_kernel", "conv_win3_mixed_kernel"],
                    "wsmg_conv2d_bwd_data_bf16": ["conv_igemm_bf16_kernel<true, *>", "conv_win3_kernel", "conv_win3_mixed_kernel"]}.get(r.get("entry"), [fam.replace("<*>", "")])
            o["kernels_of_family"] = fams
            if args.dtype.startswith("bf16") and tfiles:
                tk = json.load(open(tfiles[-1])).get("kernels", {})
                hit = [tk[f] for f in fams if f in tk]
                if hit:
                    o["traffic"] = round(sum(h["hbm_bytes_per_update"] for h in hit) / sum(h["launches_per_update"] for h in hit))
                    o["traffic_unit"] = "HBM bytes per launch (1024*(2*FETCH_SIZE+WRITE_SIZE)), mean over the family's launches"
                    o["traffic_source"] = "profiles/" + os.path.basename(tfiles[-1])
            if args.dtype.startswith("bf16") and mfiles:
                mk = json.load(open(mfiles[-1])).get("kernels", {})
                hit = [mk[f] for f in fams if f in mk]
                if hit:
                    w = sum(h["ms_per_update"] for h in hit)
                    o["mfma_busy"] = round(sum(h["mfma_busy"] * h["ms_per_update"] for h in hit) / w, 4)
                    o["eff_clock_ghz"] = round(sum(h["eff_clock_ghz"] * h["ms_per_update"] for h in hit) / w, 3)
                    o["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), time-weighted over the "
                                           "family; achieved/peak = mfma_busy x (clock / 2.4 GHz) x (algorithmic / issued FLOPs)")
                    o["mfma_busy_source"] = "profiles/" + os.path.basename(mfiles[-1])
            return o

        roofline = roofline_of(dom) if dom else None
        # both named families' own objects, unconditionally (same live measurement, same fields; one of them is also `roofline`)
        wg = [k for k in timed if prof[k].get("entry", "").startswith("wsmg_conv2d_bwd_weight")]
        fw = [k for k in timed if prof[k].get("entry", "").startswith("wsmg_conv2d_fwd")]
        roofline_wgrad = roofline_of(wg[0]) if wg else None
        roofline_fwd = roofline_of(fw[0]) if fw else None
        out = {
            "metric": "policy steps/sec (fwd+bwd), CMA batch=8 seq=64",
            "value": round(steps_per_s, 2),
            "unit": "policy steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_s": prewarm["seconds"] if prewarm else 0.0,
            "prewarm_updates": prewarm["updates"] if prewarm else 0,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"teacher-forcing update fwd+bwd+Adam, T={T} x N={N} rows/GPU (B={T * N}), E=100 C=64 "
                                   f"80-token instructions, cached rgb/depth/ego-map features (BASELINE configs[1])",
                       "T": T, "N_per_gpu": N, "parallelism": f"dp{world}"},
            "data_parallel": dp_info,
            "recurrent_core": fallback,
            "whole_update_tflops": round(ALG_GFLOP_PER_STEP * steps_per_s / 1e3 / world, 2),
            "loss": round(final_loss, 5),
            "f32_parity_mode": parity,
            "bf16_f32grad_mode": f32grad,
            "feeder_layout_mode": feeder_layout,
            "graphed_update": graphed,
            "sustained": sustained,
            "windows": windows,
            "host_ms_per_update": getattr(measure, "host_ms_main", None),
            "host_ms_note": "host wall time inside update() per timed update: enqueueing the launches plus whatever the host waits for "
                            "(single process: only the instruction dedup's read-back on the early stream; under a process group the "
                            "read-back waits for the previous update, so the figure is then the update's own time)",
            "roofline": roofline,
            "roofline_weight_gradient_family": roofline_wgrad,
            "roofline_forward_family": roofline_fwd,
            "other_configs": other,
            "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kernels.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(state_cpu, T, N, args.cpu_budget)
            out["gpu_over_cpu"] = round(steps_per_s / out["cpu_baseline"]["value"], 1)
        else:
            out["cpu_baseline"] = None
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if args.dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
