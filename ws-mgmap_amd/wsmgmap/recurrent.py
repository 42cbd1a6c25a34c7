"""The recurrent core of the update path as ONE autograd node, software-pipelined over time chunks.

Reference: MGMapNet.forward, vlnce_baselines/models/mg_map_policy.py:220-249 —

    state           = state_encoder(state_in, h[0], masks)                       # GRU 1, T serial steps
    text_embedding  = _attn(state_text_q_layer(state), text_k, text_v, text_mask) # per row
    map_embedding   = _attn(text_map_q_layer(text_embedding), map_k, map_v)       # per row
    x               = second_state_compress(cat(state, text_embedding, map_embedding))
    x               = second_state_encoder(x, h[1], masks)                        # GRU 2, T serial steps

Only the two GRUs are recurrent; everything between them is per ROW.  Run stage after stage (rounds 1-3) the update's critical
path holds 2 x T dependent GRU steps forward and 2 x T backward (4 x 64 steps of 5.6-6.0 us = 1.5 ms with 224 of the 256 CUs idle)
plus ~60 dependent small launches around them.  Here the T steps are cut into K chunks: while GRU 1 runs chunk k+1 on one
stream, the attention stage of chunk k runs on a second and GRU 2 of chunk k on a third (the carried hidden state is bit-exact
across split launches: tests/test_gpu_kernels.py::test_fullsize_gru_sequence_split_cfg2), and the reverse order in backward:
T + T/K serial steps each way instead of 2 T.  Weight and bias gradients are NOT part of the chain: the backward chain only
produces activation gradients into full-batch buffers, and every parameter gradient is one full-batch GEMM / column sum on a
"leaf" stream that joins the main stream when the whole backward pass ends (nothing but the optimizer waits for them).

Arithmetic: the same kernels and the same float32 GEMMs as the staged route, row for row; chunking changes no reduction order
inside a row except where the GEMM library picks another kernel for 128 rows than for 512 (float32 rounding differences, <= 1e-6
relative; tests/test_gpu_round4.py holds the block against the staged route)."""
import ctypes

import torch

from . import _abi
from . import ops as _ops
from .ops import core as _core
from .debug import sw as _sw
from .ops import _join_side_at_end, _p, _rnn_launched, _rnn_workspace, _sfx, _stream

MAX_BATCH = 8          # batch slots of the persistent GRU kernels


_owned = {}      # (device index, role, stream handle) -> [workspace, bytes, timeouts seen]: the GRU chunk launches' persistent exchange images


def _owned_ws(dev, role, nbytes):
    """A zeroed workspace that only this role's launches ON THE CURRENT STREAM ever use (wsmg_gru_*_owned: no clear per launch).
    One per (role, stream) — GRU 1 / GRU 2, forward / backward: launches of one role follow each other on one stream, launches of
    different roles overlap; a second policy or a caller on another stream gets an image of its own.  A workspace
    that is outgrown is told which stream may still be using it before it is dropped."""
    sid = torch.cuda.current_stream(dev).cuda_stream
    key = (dev.index, role, sid)
    e = _owned.get(key)
    if e is None or e[1] < nbytes or e[2] != _abi.rnn_timeouts:
        if e is not None:
            e[0].record_stream(torch.cuda.current_stream(dev))
        e = _owned[key] = [torch.zeros((int(nbytes) + 3) // 4, device=dev, dtype=torch.float32), int(nbytes), _abi.rnn_timeouts]
    return e[0]


_hooks_probe = None


def _post_hooks_visible():
    """Does this torch keep `register_post_accumulate_grad_hook` hooks in the private dict the multi-stream backward runs by hand
    (`Tensor._post_accumulate_grad_hooks`)?  Probed once on a scratch tensor; if a torch version moves it, the block stays on the
    one-stream route (gradients through AccumulateGrad) instead of silently skipping the reducer's hooks."""
    global _hooks_probe
    if _hooks_probe is None:
        try:
            t = torch.zeros(1, requires_grad=True)
            seen = []
            h = t.register_post_accumulate_grad_hook(seen.append)
            d = getattr(t, "_post_accumulate_grad_hooks", None)
            _hooks_probe = bool(d) and len(d) == 1
            for f in list((d or {}).values()):
                f(t)
            _hooks_probe = _hooks_probe and len(seen) == 1
            h.remove()
        except Exception:
            _hooks_probe = False
    return _hooks_probe


def _prow(t, row):
    """Pointer to row `row` of a contiguous tensor (a view object per kernel argument costs more host time than the launch)."""
    return ctypes.c_void_p(t.data_ptr() + row * t.stride(0) * t.element_size())


def _rg(a_segs, w, w_is_kn, out_segs, r0, M, bias=None, cin_segs=None, mask=None, relu=False, wait=None, signal=None, fail_bit=1):
    """One launch of wsmg_rows_gemm_f32 on rows r0 .. r0 + M of full-batch row-major tensors: C = epilogue([A0|A1|A2] W^T) (w_is_kn
    False: W an nn.Linear weight [N, K]) or ([A0|A1|A2] W) (True: W [K, N], the backward product dY W).  a_segs / out_segs /
    cin_segs: lists of up to three 2-D tensors (their column counts are the segment widths); mask: the ReLU-backward mask source.
    wait = (counter address, target, gate word address): the product waits for the counter before reading (operands produced by a
    chained GRU launch that is still running), through a one-workgroup gate launch that reports in the gate word — a word of the
    caller's per-pass counter tensor, one per chunk; signal = counter address: every workgroup adds an arrival when its tile is stored."""
    z = (None, 0, 0)
    A = [(_prow(t, r0), t.stride(0), t.shape[1]) for t in a_segs] + [z] * (3 - len(a_segs))
    Cs = [(_prow(t, r0), t.stride(0), t.shape[1]) for t in out_segs] + [z] * (3 - len(out_segs))
    Ci = [(_prow(t, r0), t.stride(0)) for t in (cin_segs or [])] + [(None, 0)] * (3 - len(cin_segs or []))
    _abi.call("wsmg_rows_gemm_f32", *A[0], *A[1], *A[2], _p(w), w.stride(0), int(bool(w_is_kn)), _p(bias),
              None if mask is None else _prow(mask, r0), 0 if mask is None else mask.stride(0), int(bool(relu)),
              *Cs[0], *Cs[1], *Cs[2], *Ci[0], *Ci[1], *Ci[2], int(M),
              None if wait is None else ctypes.c_void_p(wait[0]), 0 if wait is None else int(wait[1]),
              None if signal is None else ctypes.c_void_p(signal), int(fail_bit),
              None if wait is None else ctypes.c_void_p(wait[2]), _stream())


def _fork(main, *streams):
    """Every stream of `streams` waits for what `main` has queued so far — ONE event record on `main` (stream.wait_stream(main)
    records one per call, and a record between two dependent kernels costs the recording stream ~3 us: tools/exp/event_cost.hip)."""
    e = torch.cuda.Event()
    e.record(main)
    for st in streams:
        st.wait_event(e)


def rows_gemm_ok(H, C, in1):
    """Shapes wsmg_rows_gemm_f32 takes for the attention stage's products (hidden size H, attention width C, GRU-2 input): every
    product's reduction length is asked of the library (a multiple of 256 alone admitted K = 1280, which it refuses)."""
    if not (_sw.rows_gemm and H % 256 == 0 and C % 256 == 0 and in1 % 256 == 0):
        return False
    ok = _abi.lib().wsmg_rows_gemm_supported
    # forward: q1 (K = H), q2 / qf (C), compress (H + 2 C), gi2 (in1); backward: dxc (3 H), its split (in1), the queries (C, H)
    return all(ok(int(k)) for k in (H, C, H + 2 * C, in1, 3 * H))


def _roles(streams, main):
    """(attention stream, GRU-2 stream, leaf stream) out of the caller's helper streams.  The block creates NO stream of its own: HIP
    multiplexes a process's streams onto 4 hardware queues, and two streams that share a queue run their work in submission order
    — with three private streams on top of the policy's (instruction branch, decoder branch) the attention stage of chunk k+1
    queued behind GRU 2 of chunk k (measured: 200 us per chunk instead of 117).  The policy's two helper streams are idle while
    this block runs, in both directions, so they are what it uses.  The attention stream is the instruction branch's — the stream
    that branch's backward, the consumer of the shared sets' gradients, runs on — and the leaf work goes behind the attention
    stage's last chunk on the same stream (measured against the decoder branch's stream, three interleaved pairs of 100 updates on
    one box: median 10.52-10.54 vs 10.73-10.74 ms per update)."""
    streams = [s for s in (streams or ()) if s is not None and s.cuda_stream != main.cuda_stream]
    if not streams:
        return main, main, main
    sa = streams[0]
    sg = streams[1] if len(streams) > 1 else streams[0]
    return sa, sg, sa


def usable(state_in, tokens, n_env, text, capturing_ok=True):
    """Can the pipelined block take this call?  Training-size sequences (more than one step) of at most 8 environments on the GPU,
    float32 state / text tensors, map tokens [B, I, 256]."""
    B = state_in.shape[0]
    return (state_in.is_cuda and state_in.dtype == torch.float32 and n_env <= MAX_BATCH and B % n_env == 0 and B // n_env > 1
            and tokens.dim() == 3 and tokens.shape[2] == 256 and all(t.dtype in (torch.float32, torch.bfloat16) for t in text[:2]))


def chunk_plan(T, want):
    """Time steps per chunk: `want` chunks of equal length when T divides, else the largest divisor count below it (chunks of
    unequal length would only complicate the plan: T = 64 in the bench, DAgger batches are padded to their longest episode)."""
    k = max(1, min(int(want), T))
    while T % k:
        k -= 1
    return k, T // k


class _RecurrentBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, state_in, tokens, text_k, text_v, text_mask, inverse, masks, h01, h02,
                w_ih1, b_ih1, w_hh1, b_hh1, wq1, bq1, wq2, bq2, wk, bk, wc, bc, w_ih2, b_ih2, w_hh2, b_hh2):
        N, K, scale, sink, text_ev = cfg["N"], cfg["chunks"], cfg["scale"], cfg["sink"], cfg["text_ready"]
        ctx.streams = cfg["streams"]
        B = state_in.shape[0]
        T = B // N
        H = w_hh1.shape[1]
        dev = state_in.device
        K, Tc = chunk_plan(T, K)
        rows = Tc * N
        main = torch.cuda.current_stream()
        sa, sg, _ = _roles(ctx.streams if K > 1 else None, main)
        if sa is main:
            K, Tc, rows = 1, T, B
        f32 = dict(device=dev, dtype=torch.float32)
        wk2 = wk.reshape(wk.shape[0], -1)
        I, C = tokens.shape[1], tokens.shape[2]
        L = text_k.shape[1]
        lib = _abi.lib()
        # full-batch buffers (main stream's pool; every other stream is joined into main before this function returns, and they
        # live on as saved tensors until the backward pass)
        gi1 = torch.addmm(b_ih1, state_in, w_ih1.t())
        y1 = torch.empty(T, N, H, **f32)
        sv1 = [torch.empty(T, N, H, **f32) for _ in range(4)]
        y2 = torch.empty(T, N, H, **f32)
        sv2 = [torch.empty(T, N, H, **f32) for _ in range(4)]
        # (projection outputs start as their bias rows and are accumulated into: `addmm` with a 1-D bias takes hipBLASLt's bias-epilogue
        #  route, whose heuristic picks ONE 256 x 128 macro-tile for a 128-row chunk — 64 us for a 33-MFLOP product)
        q1 = torch.empty(B, wq1.shape[0], **f32)
        text_emb = torch.empty(B, C, **f32)
        attn_text = torch.empty(B, L, **f32)
        q2 = torch.empty(B, wq2.shape[0], **f32)
        qf = torch.empty(B, C, **f32)
        map_emb = torch.empty(B, C, **f32)
        att_map = torch.empty(B, I, **f32)
        x = None            # (the concatenation is an operand of the rows-GEMM route: the leaf pass rebuilds it for dW)
        if not (rows_gemm_ok(H, C, wc.shape[0]) and wq1.shape[0] % 256 == 0 and wq2.shape[0] % 256 == 0):
            x = torch.empty(B, wc.shape[1], **f32)
        xc = torch.empty(B, wc.shape[0], **f32)
        gi2 = torch.empty(B, 3 * H, **f32)
        m = masks.reshape(T, N)
        h01 = h01.clone()          # the caller overwrites rnn_hidden_states in place (reference contract)
        h02 = h02.clone()
        tsfx = _sfx(text_k)
        ksfx = _sfx(tokens)
        nbytes = lib.wsmg_gru_workspace_bytes(Tc)
        # (wsmgmap.parallel holds its buckets until this block's backward has queued its kernels: see _launch_ready there)
        _core.chain_in_flight = bool(sa is not main and any(ctx.needs_input_grad))
        forked = False
        if sa is not main:
            # the buffers above come from the main stream's pool: whatever used their memory before is queued on main
            # (forked below, once: the chained route after it has zeroed its counters on main)
            if text_ev is not None:
                sa.wait_event(text_ev)
        elif text_ev is not None:
            main.wait_event(text_ev)

        # chunk launches run on workspaces this module owns (no clear in front of every launch); a HIP-graph capture bakes the
        # launch-unique tags in, so replays need the clearing form
        owned = not torch.cuda.is_current_stream_capturing()

        def gru(gi, w_hh, b_hh, h0, y, sv, k, role):
            t0 = k * Tc
            hk = h0 if k == 0 else y[t0 - 1]
            ws = _owned_ws(dev, role, nbytes) if owned else _rnn_workspace(nbytes, dev)
            _abi.call("wsmg_gru_fwd_owned" if owned else "wsmg_gru_fwd", _prow(gi, t0), _p(w_hh), _p(b_hh), _p(hk), _prow(m, t0), Tc, N, H,
                      _prow(y, t0), *[_prow(s_, t0) for s_ in sv], _p(ws), _stream())
            _rnn_launched()

        gi1v, gi2v = gi1.view(T, N, 3 * H), gi2.view(T, N, 3 * H)
        y1r = y1.view(B, H)
        multi = sa is not main
        rg = rows_gemm_ok(H, C, wc.shape[0]) and wq1.shape[0] % 256 == 0 and wq2.shape[0] % 256 == 0
        # round 5 — chained: each recurrence is ONE whole-sequence launch; the attention stage's first product of chunk k waits (on the
        # device) for the first recurrence's arrivals on counter k, its last product signals counter K + k, which the second
        # recurrence waits for at the first step of chunk k (csrc/wsmg_rnn.hip chain_wait).  No chunk prologues (W_hh into registers:
        # ~10 us, 16 times per update), no launch gaps, no events; a waiter is enqueued after its producers.
        chain = multi and rg and owned and _sw.recurrent_chain and K > 1
        cnt = None
        if chain:
            cnt = torch.zeros(3 * K, device=dev, dtype=torch.int32)     # [arrivals in | arrivals out | gate words] per chunk
            nwg = int(lib.wsmg_gru_chain_workgroups())
            _fork(main, sa, sg)          # (the counters are zeroed on main)
            forked = True
            _abi.call("wsmg_gru_fwd_chain", _p(gi1), _p(w_hh1), _p(b_hh1), _p(h01), _p(m), T, N, H, _p(y1), *[_p(s_) for s_ in sv1],
                      _p(_owned_ws(dev, "f1", lib.wsmg_gru_workspace_bytes(T))), Tc, None, 0, _p(cnt), _stream())
            _rnn_launched()
            _ops.mark("f.g1")
        if multi and not forked:
            _fork(main, sa, sg)
        # Enqueue order: stream by stream (all of GRU 1's chunks, then all attention chunks, then all of GRU 2's) — the dependencies are
        # events, so the GPU sees the same pipeline as with a chunk-by-chunk order, and the host changes its current stream twice
        # instead of eight times (the host has < 1.5 ms of lead over the GPU in this part of an update)
        ev1, eva = [], []
        for k in range(0 if chain else K):
            gru(gi1v, w_hh1, b_hh1, h01, y1, sv1, k, "f1")                             # main
            _ops.mark("f.g1.%d" % k)
            if multi:
                e1 = torch.cuda.Event()
                e1.record(main)
                ev1.append(e1)
        with torch.cuda.stream(sa):
            if not rg:
                q1.copy_(bq1.expand_as(q1))
                q2.copy_(bq2.expand_as(q2))
                xc.copy_(bc.expand_as(xc))
                gi2.copy_(b_ih2.expand_as(gi2))
            for k in range(K):
                r0, r1 = k * rows, (k + 1) * rows
                if multi and not chain:
                    sa.wait_event(ev1[k])
                if rg:
                    # round 5: every dense layer of the stage is ONE launch (csrc/wsmg_rows_gemm.hip): bias, the concatenation, the
                    # ReLU in the epilogue / operand segments — 7 launches per chunk instead of 9 + 4 pre-fills, each ~4 us
                    _rg([y1r], wq1, False, [q1], r0, rows, bias=bq1, wait=(cnt.data_ptr() + 4 * k, nwg, cnt.data_ptr() + 4 * (2 * K + k)) if chain else None, fail_bit=1)
                else:
                    q1[r0:r1].addmm_(y1r[r0:r1], wq1.t())
                _abi.call("wsmg_attn_shared_fwd" + tsfx, _prow(q1, r0), _p(text_k), _p(text_v), _p(text_mask), _prow(inverse, r0),
                          scale, rows, L, C, _prow(text_emb, r0), _prow(attn_text, r0), _stream())
                if rg:
                    _rg([text_emb], wq2, False, [q2], r0, rows, bias=bq2)
                    _rg([q2], wk2, True, [qf], r0, rows)
                else:
                    q2[r0:r1].addmm_(text_emb[r0:r1], wq2.t())
                    torch.mm(q2[r0:r1], wk2, out=qf[r0:r1])
                _abi.call("wsmg_attn_fwd" + ksfx, _prow(qf, r0), _prow(tokens, r0), _prow(tokens, r0), None, scale, rows, I, C,
                          _prow(map_emb, r0), _prow(att_map, r0), _stream())
                if rg:
                    _rg([y1r, text_emb, map_emb], wc, False, [xc], r0, rows, bias=bc, relu=True)
                    _rg([xc], w_ih2, False, [gi2], r0, rows, bias=b_ih2, signal=cnt.data_ptr() + 4 * (K + k) if chain else None)
                else:
                    torch.cat([y1r[r0:r1], text_emb[r0:r1], map_emb[r0:r1]], dim=1, out=x[r0:r1])
                    xc[r0:r1].addmm_(x[r0:r1], wc.t())
                    torch.relu_(xc[r0:r1])
                    gi2[r0:r1].addmm_(xc[r0:r1], w_ih2.t())
                _ops.mark("f.at.%d" % k)
                if multi and not chain:
                    ea = torch.cuda.Event()
                    ea.record(sa)
                    eva.append(ea)
        with torch.cuda.stream(sg):
            if chain:
                _abi.call("wsmg_gru_fwd_chain", _p(gi2), _p(w_hh2), _p(b_hh2), _p(h02), _p(m), T, N, H, _p(y2), *[_p(s_) for s_ in sv2],
                          _p(_owned_ws(dev, "f2", lib.wsmg_gru_workspace_bytes(T))), Tc, ctypes.c_void_p(cnt.data_ptr() + 4 * K),
                          int(lib.wsmg_rows_gemm_workgroups(rows, 3 * H)), None, _stream())
                _rnn_launched()
                _ops.mark("f.g2")
            for k in range(0 if chain else K):
                if multi:
                    sg.wait_event(eva[k])
                gru(gi2v, w_hh2, b_hh2, h02, y2, sv2, k, "f2")
                _ops.mark("f.g2.%d" % k)
        if sg is not main:
            main.wait_stream(sg)
            main.wait_stream(sa)
        ctx.save_for_backward(state_in, tokens, text_k, text_v, inverse, m, h01, h02, w_ih1, w_hh1, wq1, wq2, wk2, wc, w_ih2, w_hh2,
                              gi1, y1, *sv1, y2, *sv2, q1, text_emb, attn_text, q2, qf, att_map, map_emb if x is None else x, xc)
        ctx.x_is_parts = x is None
        ctx.chain = chain
        ctx.cfg = (N, K, Tc, scale, sink, tuple(wk.shape), text_mask is not None)
        ctx.params = (w_ih1, b_ih1, w_hh1, b_hh1, wq1, bq1, wq2, bq2, wk, bk, wc, bc, w_ih2, b_ih2, w_hh2, b_hh2)
        h1n, h2n = y1[-1:].clone(), y2[-1:].clone()
        ctx.mark_non_differentiable(h1n, h2n)
        ctx.set_materialize_grads(False)
        return y2.view(B, H), att_map, h1n, h2n

    @staticmethod
    def backward(ctx, dy2, datt, _dh1, _dh2):
        (state_in, tokens, text_k, text_v, inverse, m, h01, h02, w_ih1, w_hh1, wq1, wq2, wk2, wc, w_ih2, w_hh2,
         gi1, y1, sr1, sz1, sn1, sg1, y2, sr2, sz2, sn2, sg2, q1, text_emb, attn_text, q2, qf, att_map, x, xc) = ctx.saved_tensors
        rg = ctx.x_is_parts          # the rows-GEMM route: `x` is map_emb, the concatenation is rebuilt on the leaf stream
        N, K, Tc, scale, sink, wk_shape, _ = ctx.cfg
        T = y1.shape[0]
        H = y1.shape[2]
        B = T * N
        rows = Tc * N
        dev = y1.device
        I, C = tokens.shape[1], tokens.shape[2]
        L = text_k.shape[1]
        U = text_k.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        main = torch.cuda.current_stream()
        params = ctx.params
        # Side streams only when this pass SETS the gradients of leaf parameters that nothing but this block uses: the parameter
        # gradients are produced on the leaf stream, so they must not pass through autograd's AccumulateGrad — it runs on the main
        # stream right after this function returns and, whenever it does not simply keep the tensor (an existing .grad to add to; a
        # tensor it decides to copy), launches a kernel there that reads a gradient the leaf stream has not written yet (seen as
        # one wrong weight gradient in 8 runs of two ranks sharing a GPU).  In that mode this function ASSIGNS p.grad itself, runs
        # the parameters' post-accumulate hooks (wsmgmap.parallel.GradAllReducer's bucket hooks) and returns None for them.
        # A stock DistributedDataParallel wrap hooks the AccumulateGrad nodes instead: there the block stays on one stream.
        own = [p for p in params if p is not None and p.requires_grad]
        # (a parameter with a tensor hook — `register_hook`, which AccumulateGrad's input would have run — keeps the one-stream,
        #  through-autograd form: assigning .grad here would silently skip it; ADVICE r04.  These parameters have no consumer
        #  outside this block — `is_leaf and grad is None` at entry is the whole contract: nothing else can add to them in this pass)
        multi = (ctx.cfg[1] > 1 and _post_hooks_visible()
                 and all(p.is_leaf and p.grad is None and not getattr(p, "_backward_hooks", None) for p in own))
        if multi and torch.distributed.is_available() and torch.distributed.is_initialized():
            multi = all(getattr(p, "_wsmg_reducer", False) for p in own)
        sa, sg, sl = _roles(ctx.streams if multi else None, main)
        multi = sa is not main
        if multi:
            try:
                _join_side_at_end(main, sl, strict=True)      # raises outside a backward pass (torch.autograd.grad of a test)
                if sl not in _ops._leaf_streams:              # (ops.reset_pass_state re-joins it if a pass died before its callbacks)
                    _ops._leaf_streams.append(sl)
            except RuntimeError:
                multi = False
        if not multi:
            sa = sg = sl = main
            K, Tc, rows = 1, T, B
        dy2 = torch.zeros(T, N, H, **f32) if dy2 is None else dy2.contiguous().float().view(T, N, H)
        datt = None if datt is None else datt.contiguous().float()
        lib = _abi.lib()
        nbytes = lib.wsmg_gru_workspace_bytes(Tc)
        tsfx, ksfx = _sfx(text_k), _sfx(tokens)
        # full-batch gradient buffers, allocated on the main stream and used on the others: the allocator must not hand their memory
        # to a later main-stream allocation while a side stream still works on them
        dgi2, dgh2 = torch.empty(T, N, 3 * H, **f32), torch.empty(T, N, 3 * H, **f32)
        dgi1, dgh1 = torch.empty(T, N, 3 * H, **f32), torch.empty(T, N, 3 * H, **f32)
        dxc = torch.empty(B, wc.shape[0], **f32)
        dqf, dq2, dq1 = torch.empty(B, C, **f32), torch.empty(B, wq2.shape[0], **f32), torch.empty(B, wq1.shape[0], **f32)
        dtext = torch.empty(B, C, **f32)
        dmap_all = torch.empty(B, C, **f32) if rg else None
        dl = torch.empty(B, L, **f32)
        dstate = torch.empty(T, N, H, **f32)
        dtokens = torch.empty_like(tokens)
        dh01, dh02 = torch.empty(N, H, **f32), torch.empty(N, H, **f32)
        carry2 = [torch.empty(N, H, **f32) for _ in range(2)]
        carry1 = [torch.empty(N, H, **f32) for _ in range(2)]
        if multi:
            # these are allocated from the main stream's pool and used on the others: they must not be handed to a later main-stream
            # allocation while a side stream still works on them — a reference is held until the pass's final callbacks, which run
            # after the leaf stream has been joined into main (cheaper than ~50 record_stream calls, and legal under graph capture)
            # — and so do the tensors SAVED by the forward pass: autograd drops them the moment this function returns, long before
            # the leaf stream's GEMMs have read x, xc, q2, text_emb, y1, y2, ... (seen as one wrong weight gradient in ~8 runs of two
            # ranks sharing a GPU, where the leaf stream starts late)
            keep = [dgi2, dgh2, dgi1, dgh1, dxc, dqf, dq2, dq1, dtext, dmap_all, dl, dstate, dtokens, dh02, dy2, datt, carry2, carry1,
                    list(ctx.saved_tensors)]
            torch.autograd.Variable._execution_engine.queue_callback(keep.clear)
        xcr = xc
        y1r = y1.view(B, H)
        dstate_r = dstate.view(B, H)
        dgi2r = dgi2.view(B, 3 * H)

        owned = not torch.cuda.is_current_stream_capturing()

        def gru_bwd(dy, w_hh, h0, y, sv, dgi, dgh, dh0_out, carry, k, role):
            t0 = k * Tc
            hk = h0 if k == 0 else y[t0 - 1]
            dhT = None if k == K - 1 else carry[(k + 1) & 1]
            out = dh0_out if k == 0 else carry[k & 1]
            ws = _owned_ws(dev, role, nbytes) if owned else _rnn_workspace(nbytes, dev)
            _abi.call("wsmg_gru_bwd_owned" if owned else "wsmg_gru_bwd", _prow(dy, t0), _p(dhT), _p(w_hh), _p(hk), _prow(m, t0), _prow(y, t0),
                      *[_prow(s_, t0) for s_ in sv], Tc, N, H, _prow(dgi, t0), _prow(dgh, t0), _p(out), _p(ws), _stream())
            _rnn_launched()

        # stream by stream, as in forward: GRU 2's chunks (last chunk first), the attention stage's, GRU 1's
        ev2, eva = {}, {}
        chain = multi and rg and owned and ctx.chain and K > 1
        cnt = None
        if chain:     # (see forward: one launch per recurrence, device-side waits on per-chunk arrival counters)
            cnt = torch.zeros(3 * K, device=dev, dtype=torch.int32)     # [arrivals in | arrivals out | gate words] per chunk
            keep.append(cnt)
            nwg = int(lib.wsmg_gru_chain_workgroups())
        if multi:
            _fork(main, sg, sa)      # once, behind everything the pass has queued on main so far (buffers, the chain's counters)
        with torch.cuda.stream(sg):
            if chain:
                _abi.call("wsmg_gru_bwd_chain", _p(dy2), None, _p(w_hh2), _p(h02), _p(m), _p(y2), _p(sr2), _p(sz2), _p(sn2), _p(sg2), T, N, H,
                          _p(dgi2), _p(dgh2), _p(dh02), _p(_owned_ws(dev, "b2", lib.wsmg_gru_workspace_bytes(T))), Tc, None, 0, _p(cnt),
                          _stream())
                _rnn_launched()
                _ops.mark("b.g2")
            for k in range(K - 1 if not chain else -1, -1, -1):
                gru_bwd(dy2, w_hh2, h02, y2, (sr2, sz2, sn2, sg2), dgi2, dgh2, dh02, carry2, k, "b2")
                _ops.mark("b.g2.%d" % k)
                if multi:
                    ev2[k] = torch.cuda.Event()
                    ev2[k].record(sg)
        with torch.cuda.stream(sa):
            wc_s, wc_t, wc_m = wc[:, :H], wc[:, H:H + C], wc[:, H + C:]
            wk2t = wk2.t()
            for k in range(K - 1, -1, -1):
                r0, r1 = k * rows, (k + 1) * rows
                if multi and not chain:
                    sa.wait_event(ev2[k])
                if rg:
                    # round 5: five launches of wsmg_rows_gemm_f32 around the two attention kernels (was 9 GEMM-library launches + a
                    # threshold_backward): the ReLU mask, the split of d(cat) into its three parts and the two accumulate-intos
                    # (beta = 1, in place) ride in the epilogues
                    _rg([dgi2r], w_ih2, True, [dxc], r0, rows, mask=xcr, wait=(cnt.data_ptr() + 4 * k, nwg, cnt.data_ptr() + 4 * (2 * K + k)) if chain else None, fail_bit=2)
                    _rg([dxc], wc, True, [dstate_r, dtext, dmap_all], r0, rows)
                    _abi.call("wsmg_attn_bwd" + ksfx, _prow(qf, r0), _prow(tokens, r0), _prow(tokens, r0), _prow(att_map, r0), _prow(dmap_all, r0),
                              None if datt is None else _prow(datt, r0), scale, rows, I, C, _prow(dqf, r0), _prow(dtokens, r0),
                              _prow(dtokens, r0), _stream())
                    _rg([dqf], wk2, False, [dq2], r0, rows)
                    _rg([dq2], wq2, True, [dtext], r0, rows, cin_segs=[dtext])
                    _abi.call("wsmg_attn_shared_bwd" + tsfx, _prow(q1, r0), _p(text_k), _p(text_v), _prow(attn_text, r0), _prow(dtext, r0),
                              None, _prow(inverse, r0), scale, rows, L, C, _prow(dq1, r0), _prow(dl, r0), _stream())
                    _rg([dq1], wq1, True, [dstate_r], r0, rows, cin_segs=[dstate_r], signal=cnt.data_ptr() + 4 * (K + k) if chain else None)
                else:
                    # ReLU of second_state_compress: d(pre-activation) = d(xc) where xc > 0
                    dxc_k = dxc[r0:r1]
                    torch.ops.aten.threshold_backward.grad_input(torch.mm(dgi2r[r0:r1], w_ih2), xcr[r0:r1], 0.0, grad_input=dxc_k)
                    dstate_a = torch.mm(dxc_k, wc_s)
                    dtext_a = torch.mm(dxc_k, wc_t)
                    dmap = torch.mm(dxc_k, wc_m)
                    _abi.call("wsmg_attn_bwd" + ksfx, _prow(qf, r0), _prow(tokens, r0), _prow(tokens, r0), _prow(att_map, r0), _p(dmap),
                              None if datt is None else _prow(datt, r0), scale, rows, I, C, _prow(dqf, r0), _prow(dtokens, r0),
                              _prow(dtokens, r0), _stream())
                    dq2_k = dq2[r0:r1]
                    torch.mm(dqf[r0:r1], wk2t, out=dq2_k)
                    torch.addmm(dtext_a, dq2_k, wq2, out=dtext[r0:r1])
                    _abi.call("wsmg_attn_shared_bwd" + tsfx, _prow(q1, r0), _p(text_k), _p(text_v), _prow(attn_text, r0), _prow(dtext, r0),
                              None, _prow(inverse, r0), scale, rows, L, C, _prow(dq1, r0), _prow(dl, r0), _stream())
                    torch.addmm(dstate_a, dq1[r0:r1], wq1, out=dstate_r[r0:r1])
                _ops.mark("b.at.%d" % k)
                if multi and not chain:
                    eva[k] = torch.cuda.Event()
                    eva[k].record(sa)
        if chain:
            _abi.call("wsmg_gru_bwd_chain", _p(dstate), None, _p(w_hh1), _p(h01), _p(m), _p(y1), _p(sr1), _p(sz1), _p(sn1), _p(sg1), T, N, H,
                      _p(dgi1), _p(dgh1), _p(dh01), _p(_owned_ws(dev, "b1", lib.wsmg_gru_workspace_bytes(T))), Tc,
                      ctypes.c_void_p(cnt.data_ptr() + 4 * K), int(lib.wsmg_rows_gemm_workgroups(rows, H)), None, _stream())
            _rnn_launched()
            _ops.mark("b.g1")
        for k in range(K - 1 if not chain else -1, -1, -1):
            if multi:
                main.wait_event(eva[k])
            gru_bwd(dstate, w_hh1, h01, y1, (sr1, sz1, sn1, sg1), dgi1, dgh1, dh01, carry1, k, "b1")      # main
            _ops.mark("b.g1.%d" % k)
        d_state_in = torch.mm(dgi1.view(B, 3 * H), w_ih1) if ctx.needs_input_grad[1] else None
        # gradients of the shared instruction sets (they feed the instruction branch's backward, on ITS stream): after the last
        # attention chunk on the attention stream, joined into main below — by then GRU 1's last chunk has long hidden them
        with torch.cuda.stream(sa):
            member = torch.nn.functional.one_hot(inverse, U).to(torch.float32).t()            # [U, B]
            dk = torch.matmul((member.unsqueeze(2) * dl.unsqueeze(0)).transpose(1, 2), q1).to(text_k.dtype)
            dv = torch.matmul((member.unsqueeze(2) * attn_text.unsqueeze(0)).transpose(1, 2), dtext).to(text_v.dtype)
        if multi:
            ekv = torch.cuda.Event()
            ekv.record(sa)
            sl.wait_stream(main)
            sl.wait_stream(sg)
        # parameter gradients: full-batch GEMMs and column sums, off the chain
        with torch.cuda.stream(sl):
            g2 = dgi2r
            gh2 = dgh2.view(B, 3 * H)
            g1 = dgi1.view(B, 3 * H)
            gh1 = dgh1.view(B, 3 * H)
            # the seven bias gradients (column sums) in one launch (round 6; torch: a reduction kernel + a memset each)
            db_ih2, db_hh2, dbc, dbq2, dbq1, db_hh1, db_ih1 = _ops.colsum_multi([g2, gh2, dxc, dq2, dq1, gh1, g1])
            dw_ih2 = g2.t() @ xc
            hp2 = torch.cat([h02.unsqueeze(0), y2[:-1]], dim=0) * m.unsqueeze(-1)
            dw_hh2 = gh2.t() @ hp2.view(B, H)
            if rg:
                x = torch.cat([y1r, text_emb, x], dim=1)          # (x held map_emb)
            dwc = dxc.t() @ x
            dwk = (q2.t() @ dqf).reshape(wk_shape)
            # (the key projection's bias adds the same number to every token's logit: it cancels in the softmax, gradient exactly 0)
            dbk = torch.zeros(wk_shape[0], **f32) if params[9] is not None else None
            dwq2 = dq2.t() @ text_emb
            dwq1 = dq1.t() @ y1r
            hp1 = torch.cat([h01.unsqueeze(0), y1[:-1]], dim=0) * m.unsqueeze(-1)
            dw_hh1 = gh1.t() @ hp1.view(B, H)
            dw_ih1 = g1.t() @ state_in
        pgrads = [dw_ih1, db_ih1, dw_hh1, db_hh1, dwq1, dbq1, dwq2, dbq2, dwk, dbk, dwc, dbc, dw_ih2, db_ih2, dw_hh2, db_hh2]
        _core.chain_in_flight = False      # every kernel of this block's backward is queued: the gradient exchange may issue buckets again
        if multi:
            main.wait_event(ekv)
            for i, p in enumerate(params):
                if p is None or not p.requires_grad or pgrads[i] is None:
                    continue
                p._wsmg_grad_stream = sl           # wsmgmap.parallel.GradAllReducer packs this gradient behind that stream
                p.grad = pgrads[i]
                for hook in list((getattr(p, "_post_accumulate_grad_hooks", None) or {}).values()):
                    hook(p)
            pgrads = [None] * len(pgrads)
        # the map tokens' gradient: parked for the token mean's backward, which merges its broadcast row and the producing
        # convolution's ReLU mask into it in one pass (ops.TokenGradSink), or returned
        if sink is not None:
            sink.park(dtokens)
            dtokens = None
        return (None, d_state_in, dtokens, dk, dv, None, None, None, dh01, dh02, *pgrads)


def recurrent_block(state_in, tokens, text, masks, h01, h02, net, n_env, chunks=4, sink=None, text_ready=None, streams=None):
    """-> (x [B, H] = GRU 2's outputs, att_map [B, I], h1_n [1, N, H], h2_n [1, N, H]).
    state_in [B, in] (rows time-major, B = T * n_env); tokens [B, I, 256] map tokens (keys == values, the key projection
    `net.text_map_k_layer` folded into the query); text = (keys [U, L, 256], values [U, L, 256], pad mask uint8 [U, L], inverse
    int64 [B]) of the U unique instructions; masks [B, 1]; h01 / h02 [N, H]; net: the MGMapNet whose layers these are;
    text_ready: an event after which `text` is complete (recorded on the instruction branch's stream), or None when the caller's
    stream already waited; streams: up to two helper streams that are idle while the block runs (see _roles), None = one stream."""
    r1, r2 = net.state_encoder.rnn, net.second_state_encoder.rnn
    tk, tv, tm, inv = text
    cfg = dict(N=int(n_env), chunks=int(chunks), scale=float(net._scale_f), sink=sink, text_ready=text_ready, streams=tuple(streams or ()))
    return _RecurrentBlock.apply(
        cfg, state_in.contiguous(), tokens.contiguous(), tk, tv, tm, inv, masks.reshape(-1).float().contiguous(), h01, h02,
        r1.weight_ih_l0, r1.bias_ih_l0, r1.weight_hh_l0, r1.bias_hh_l0,
        net.state_text_q_layer.weight, net.state_text_q_layer.bias, net.text_map_q_layer.weight, net.text_map_q_layer.bias,
        net.text_map_k_layer.weight, net.text_map_k_layer.bias, net.second_state_compress[0].weight, net.second_state_compress[0].bias,
        r2.weight_ih_l0, r2.bias_ih_l0, r2.weight_hh_l0, r2.bias_hh_l0)
