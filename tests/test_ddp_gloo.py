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
