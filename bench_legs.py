"""bench_legs.py — the other BASELINE configs as short, HIP-event-timed legs of the driver's bench line (`other_configs`).

`bench.py` measures BASELINE configs[1] (the teacher-forcing update) as `value`.  SURVEY.md §8d also defines measurements for
  cfg1 = configs[0]: BasePolicy.act, B = 1, 256 x 256 RGB-D, 80-token instruction, E = 100, C = 64 (plumbing / latency config),
  cfg4 = configs[3]: the rollout-side RGBMapping.forward (channel max-pool 64 -> 40 + index + scatter + fuse / retrieve) and the
         MapEncoder at E = 200, C = 40, B = 32 (the decoder is undefined at that size in the reference),
  cfg5 = configs[4]: the text stage of `_attn` with e4m3 storage at B = 64, L = 160,
which rounds 1-4 timed with builder-run tools only (profiles/r0N_bev.txt, r0N_attn_fp8.txt, r0N_act.txt).  Here the same
measurements run inside the driver's own command, after the cfg2 leg and outside its timed region (about 2 s in all), each
with HIP events on the stream the kernels are launched on (torch's current stream: every wsmg_* entry point takes it).

Algorithmic bytes are SURVEY.md §8d's per-unit figures:
  BEV scatter, per sample : Cf*Hf*Wf*4 (features) + Hd*Wd*4 (depth) + C*E*E*4 (map out)
  fuse + retrieve, per sample : 4 * C*E*E*4
  attention, per row (fp8) : 2*256*L token bytes + 256*4 context out
`stage_bytes` is the finer accounting of tools/bench_bev.py (what each launch of the fused route must move), kept so that the
figure can be compared with profiles/r04_bev.txt.  Peak: HBM 8 TB/s (MI355X_MICROARCH.md).
"""
import time

import torch

HBM_PEAK_GBS = 8000.0
PEAK_BF16_TFLOPS = 2500.0


class _Box:
    shape = (2,)


def _events(fn, reps, warm=3):
    """Mean HIP-event time of `fn` in microseconds over `reps` back-to-back calls on the current stream."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


_stall_ab = None


def _events_gpu(fn, reps, warm=3):
    """As _events, but with the stream held busy (a ~10 ms float32 matrix product) while the host enqueues the `reps` calls, so that the
    events bracket kernels that run back to back on the GPU: for operators of a few tens of microseconds the plain loop measures the
    host's enqueue rate (~30 us per Python-level call), not the kernels."""
    global _stall_ab
    if _stall_ab is None:
        _stall_ab = (torch.randn(8192, 8192, device="cuda"), torch.randn(8192, 8192, device="cuda"))
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.mm(_stall_ab[0], _stall_ab[1])
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def _bw(us, nbytes):
    gbs = nbytes / us / 1e3
    return dict(us=round(us, 2), alg_bytes=int(nbytes), gbps=round(gbs, 1), frac_of_8TBs=round(gbs / HBM_PEAK_GBS, 4))


def _rollout_obs(B, hw, gen, dev):
    ins = torch.zeros(B, 200, dtype=torch.int64, device=dev)
    ins[:, :80] = torch.randint(1, 2504, (B, 80), device=dev, generator=gen)
    depth = torch.rand(B, 256, 256, 1, device=dev, generator=gen)
    depth[:, :8] = 0                                   # rows 0-7 = 0
    return {
        "rgb": torch.randint(0, 256, (B, hw, hw, 3), device=dev, generator=gen).float(),
        "depth": depth,
        "depth_features": torch.randn(B, 128, 4, 4, device=dev, generator=gen),
        "instruction": ins,
        "gps": (torch.rand(B, 2, device=dev, generator=gen) - 0.5) * 4,
        "compass": (torch.rand(B, 1, device=dev, generator=gen) - 0.5) * 6.28,
    }


def cfg1_act_b1(dev, reps=30):
    """configs[0]: one rollout step from raw RGB-D at B = 1 (bf16 engine), eager and as one HIP graph; the BEV operator alone."""
    from wsmgmap.common.aux_losses import AuxLosses
    from wsmgmap.config import default_model_config
    from wsmgmap.graph import GraphedAct
    from wsmgmap.models.policy import BasePolicy
    was_active = AuxLosses.is_active()
    AuxLosses.deactivate()
    try:
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        torch.manual_seed(0)
        pol = BasePolicy(None, _Box(), default_model_config(num_proc=1, compute_dtype="bf16")).to(dev).eval()
        obs = _rollout_obs(1, 256, gen, dev)
        h = torch.zeros(2, 1, 512, device=dev)
        prev = torch.zeros(1, 2, device=dev)
        masks = torch.ones(1, 1, device=dev)
        with torch.no_grad():
            t_act = _events(lambda: pol.act(dict(obs), h.clone(), prev, masks, deterministic=True), reps)
            _, proj = pol.net.rgb_encoder(obs)
            t_bev = _events_gpu(lambda: pol.net.rgb_mapping_module(proj, dict(obs), masks), reps)
        ga = GraphedAct(pol)
        hg = h.clone()
        t_graph = _events(lambda: ga(obs, hg, prev, masks, deterministic=True), reps)
        E, C, Hf = 100, 64, 256
        alg = 1 * ((64 * Hf * Hf * 4 + 256 * 256 * 4 + C * E * E * 4) + 4 * C * E * E * 4)
        out = dict(workload="BasePolicy.act, B=1, 256x256 RGB-D, 80-token instruction, E=100 C=64, bf16 engine, deterministic",
                   ms=round(t_act / 1e3, 3), ms_as_one_hip_graph=round(t_graph / 1e3, 3), unit_rate=round(1e6 / t_graph, 1),
                   unit="env-steps/s (graphed)", bev_operator=_bw(t_bev, alg),
                   note="latency-bound at B = 1: the BEV operator is 4 launches (index, scatter+rotate, fuse, retrieve) over 30 MB")
        out.update({k: out["bev_operator"][k] for k in ("alg_bytes", "gbps", "frac_of_8TBs")})
        del pol, ga
        return out
    finally:
        if was_active:
            AuxLosses.activate()


def cfg4_bev_mapenc(dev, reps=10):
    """configs[3]: B = 32, E = 200, C = 40 (from 64 feature channels), 256 x 256 RGB-D: RGBMapping.forward, then MapEncoder."""
    from wsmgmap import ops
    from wsmgmap.common.rgb_mapping import RGBMapping
    from wsmgmap.config import default_model_config
    from wsmgmap.models.encoders.map_encoder import MapEncoder
    B, E, C, G, Hf = 32, 200, 40, 480, 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(12)
    cfg = default_model_config(num_proc=B, ego_map_size=E, map_depth=C, global_map_size=G).RGBMAPPING
    mapper = RGBMapping(cfg).to(dev)
    feat = torch.relu(torch.randn(B, 64, Hf, Hf, device=dev, generator=gen))
    obs = _rollout_obs(B, Hf, gen, dev)
    del obs["rgb"]
    masks = torch.ones(B, 1, device=dev)
    with torch.no_grad():
        t_map = _events_gpu(lambda: mapper(feat, dict(obs), masks), reps)
        # per stage, as the module issues them (fused route)
        depth = obs["depth"].reshape(B, 256, 256).contiguous()
        compass = obs["compass"].reshape(B).contiguous()
        gps = obs["gps"].contiguous()
        gm = mapper.full_global_map
        compact_ok = ops.bev_compact_ok(Hf, Hf, E, B)
        lin, comp = ops.bev_index_compact(depth, Hf, Hf, E) if compact_ok else (ops.bev_index(depth, Hf, Hf, E), None)
        rot = ops.bev_scatter_rotate(feat, lin, compass, -1.0, C, E, compact=comp)
        m1 = masks.reshape(B).contiguous()
        index_fn = (lambda: ops.bev_index_compact(depth, Hf, Hf, E)) if compact_ok else (lambda: ops.bev_index(depth, Hf, Hf, E))
        stages = {
            "index": (_events_gpu(index_fn, reps), B * (256 * 256 * 4 + Hf * Hf * 4)),
            "scatter_rotate": (_events_gpu(lambda: ops.bev_scatter_rotate(feat, lin, compass, -1.0, C, E, compact=comp), reps),
                               B * (64 * Hf * Hf * 4 + Hf * Hf * 4 + C * E * E * 4)),
            "fuse": (_events_gpu(lambda: ops.map_fuse(rot, gm, gps, m1, 0.12, planes=True), reps), B * 3 * C * (E + 4) ** 2 * 4),
            "retrieve": (_events_gpu(lambda: ops.map_retrieve(gm, gps, compass, E, 0.12), reps), B * 2 * C * E * E * 4),
        }
    alg = B * ((64 * Hf * Hf * 4 + 256 * 256 * 4 + C * E * E * 4) + 4 * C * E * E * 4)      # 23.4 + 25.6 MB per sample
    stage_bytes = sum(nb for _, nb in stages.values())
    out = dict(workload="RGBMapping.forward (64->40 channel max-pool + index + scatter-max + rotate + fuse + retrieve), "
                        "B=32, E=200, C=40, G=480, 256x256 RGB-D; then MapEncoder forward on its output (train-mode BN, bf16)")
    out.update(_bw(t_map, alg))
    out["ms"] = round(t_map / 1e3, 4)
    out["unit_rate"] = round(B * 1e6 / t_map, 1)
    out["unit"] = "map updates/s (samples)"
    out["stage_bytes"] = int(stage_bytes)
    out["stage_gbps"] = round(stage_bytes / t_map / 1e3, 1)
    out["stage_frac_of_8TBs"] = round(stage_bytes / t_map / 1e3 / HBM_PEAK_GBS, 4)
    out["valid_source_fraction"] = round(float((lin >= 0).float().mean()), 4)
    out["accounting"] = ("`frac_of_8TBs` is on SURVEY 8d's ALGORITHMIC bytes (23.4 + 25.6 MB per sample): the one headline; `stage_*` counts what "
                         "each launch of the fused route must move (reported, not a headline; the five-launch figure of rounds 3-5 is gone)")
    out["stages"] = {k: _bw(us, nb) for k, (us, nb) in stages.items()}
    # MapEncoder at this geometry: [32, 40 -> 64 (zero-padded), 200, 200] -> [32, 256, 49, 49]
    enc = MapEncoder(E, C, 256).to(dev).train()
    ego = mapper(feat, dict(obs), masks)              # channels-last view of [B, C, E, E], float32
    with torch.no_grad():
        x = ops.to_nhwc(ego.contiguous().float(), 64, dtype=torch.bfloat16)
        t_enc = _events(lambda: enc(x), reps)
    # algorithmic FLOPs on the 40 real input channels: k8 s2 (40 -> 64, 97^2), k5 s2 (64 -> 128, 47^2), k3 (128 -> 256, 49^2 ... the
    # module's own output sizes are read back from the layers)
    flops, hw, cin = 0, E, C
    for i in (0, 3, 6):
        cv = enc.cnn[i]
        k, s, p = cv.kernel_size[0], cv.stride[0], cv.padding[0]
        ohw = (hw + 2 * p - k) // s + 1
        flops += 2 * B * ohw * ohw * cv.out_channels * cin * k * k
        hw, cin = ohw, cv.out_channels
    tf = flops / t_enc / 1e6
    out["map_encoder"] = dict(us=round(t_enc, 1), alg_gflop=round(flops / 1e9, 2), tflops=round(tf, 1),
                              frac_of_bf16_peak=round(tf / PEAK_BF16_TFLOPS, 4), out_hw=hw,
                              note="3 x (conv + train-mode BatchNorm + ReLU) in bf16 storage; input already NHWC bf16")
    del enc, mapper
    return out


def cfg5_attn_fp8(dev, reps=50):
    """configs[4] as SURVEY 8d defines it: the text stage of `_attn` (mg_map_policy.py:173-178) at B = 64 rows, each over its OWN
    160-token set (x = key input = values, ragged valid lengths), e4m3 token storage, float32 query / softmax: 82 KB per row.
    Headline (round 6): ops.attn_fp8_fused — query fold + attention per row in ONE launch.  Secondary: the update path's form, B rows
    over U = 8 SHARED instruction sets from float32 q / k / v (ops.attention_fp8_shared: scales + codes + row grouping + attention)."""
    import importlib
    from wsmgmap import ops
    B, U, L, C = 64, 8, 160, 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(13)
    q = torch.randn(B, C, device=dev, generator=gen)
    k = torch.randn(U, L, C, device=dev, generator=gen)
    v = torch.randn(U, L, C, device=dev, generator=gen)
    inv = torch.arange(B, device=dev) % U
    lens = torch.randint(L // 2, L + 1, (U,), device=dev, generator=gen).to(torch.int32)
    att = importlib.import_module("wsmgmap.ops.attention")
    with torch.no_grad():
        # per-row token sets (x is key input and value), key projection folded into the query inside the launch
        w = torch.randn(C, C, device=dev, generator=gen) / 16
        b = torch.randn(C, device=dev, generator=gen) * 0.1
        x = torch.randn(B, L, C, device=dev, generator=gen)
        xs = float(x.abs().max() / 448.0)
        codes = ops.quantize_e4m3(x, xs)
        xs_t = torch.full((1,), xs, device=dev)
        lr = torch.randint(L // 2, L + 1, (B,), device=dev, generator=gen).to(torch.int32)
        t_row_host = _events(lambda: ops.attn_fp8_fused(q, w, b, codes, xs_t, lr, 1 / 16), reps)
        t_row = _events_gpu(lambda: ops.attn_fp8_fused(q, w, b, codes, xs_t, lr, 1 / 16), reps)
        row_launches = int(att.last_fp8_row_launches)
        t_shared_host = _events(lambda: ops.attention_fp8_shared(q, k, v, lens, inv, 1 / 16), reps)
        t_shared = _events_gpu(lambda: ops.attention_fp8_shared(q, k, v, lens, inv, 1 / 16), reps)
    alg_shared = 2 * U * L * C + B * (C + C * 4 + L * 4)             # key + value bytes of the U sets, B queries in, contexts + weights out
    alg_row = B * (2 * C * L + C * 4)                                 # 82 KB per row at L = 160
    flops = 2.0 * B * L * C * 2
    out = dict(workload="_attn text stage, B=64 rows x their own 160-token set (SURVEY 8d: 82 KB per row), e4m3 token storage, float32 query "
                        "and softmax, ragged lengths (ops.attn_fp8_fused: query fold + logits + softmax + context per row)")
    out.update(_bw(t_row, alg_row))
    out["ms"] = round(t_row / 1e3, 5)
    out["unit_rate"] = round(B * 1e6 / t_row, 1)
    out["unit"] = "attention rows/s"
    out["gflops"] = round((flops + 2.0 * B * C * C) / t_row / 1e3, 2)
    out["launches"] = row_launches
    out["us_host_paced_loop"] = round(t_row_host, 2)
    out["timing_note"] = ("`us`: HIP events around 50 calls enqueued while the stream was held busy, i.e. kernels back to back on the GPU; "
                          "`us_host_paced_loop`: the same 50 calls on an idle stream (the host's ~30 us per Python-level call paces them)")
    out["shared_set_form"] = dict(_bw(t_shared, alg_shared), launches=int(att.last_fp8_shared_launches), us_host_paced_loop=round(t_shared_host, 2),
                                  note="ops.attention_fp8_shared: B = 64 rows over U = 8 shared instruction sets (the update path's form) from "
                                       "float32 q / k / v: scales, e4m3 codes, row grouping, S = QK^T on the fp8 matrix pipe, PV")
    out["note"] = "latency-bound at this size (5.3 MB, 29 MFLOP with the fold): the figure to watch is us, not the fraction"
    return out


def other_configs(dev):
    """-> {leg name: result dict}; a leg that fails reports its error instead of costing the run its line."""
    res = {}
    t0 = time.perf_counter()
    for name, fn in (("cfg1_act_b1", cfg1_act_b1), ("cfg4_bev_mapenc", cfg4_bev_mapenc), ("cfg5_attn_fp8", cfg5_attn_fp8)):
        try:
            res[name] = fn(dev)
        except Exception as e:  # noqa: BLE001 — reported in the line
            res[name] = dict(error=f"{type(e).__name__}: {e}")
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    res["seconds"] = round(time.perf_counter() - t0, 2)
    res["timing"] = ("HIP events on the launch stream around back-to-back calls, mean; outside the cfg2 timed region; kernel-level figures "
                     "(BEV operator and stages, fp8 attention) with the stream held busy while the host enqueues, so that the GPU, not the "
                     "host's enqueue rate, is what is timed; act() latencies include the host")
    global _stall_ab
    _stall_ab = None
    return res
