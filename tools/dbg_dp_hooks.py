import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, torch.distributed as dist
import test_gpu_round2 as r2
from oracle import cases, policy_ref
from util import T
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.parallel import GradAllReducer
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
Tn, N = 4, 2
AuxLosses.activate()
pol = r2._train_mode(r2._policy(num_proc=2, compute_dtype="f32"))
names = {id(p): n for n, p in pol.named_parameters()}
red = GradAllReducer(pol.parameters(), bucket_bytes=4 << 20, single_rank_exchange=True)
calls = collections.Counter()
orig = red._on_grad
import traceback
seen_stacks = {}
def spy(p):
    calls[names[id(p)]] += 1
    if names[id(p)] == 'net.text_map_q_layer.weight':
        seen_stacks.setdefault(calls[names[id(p)]], ''.join(traceback.format_stack(limit=6)))
    return orig(p)
for p in red.params:
    for k in list(p._post_accumulate_grad_hooks):
        p._post_accumulate_grad_hooks[k] = spy
for it in range(3):
    calls.clear()
    obs_np, prev, masks, weights = cases.update_inputs(Tn, N, tag="dp0", n_tok=(80, 37))
    obs, w = r2._cuda(obs_np), T(weights).cuda()
    pol.zero_grad(set_to_none=True)
    AuxLosses.clear()
    pred, aux = pol(obs, torch.zeros(2, N, 512, device="cuda"), T(prev).cuda(), T(masks).cuda(), w)
    loss, _ = policy_ref.dagger_loss(pred, aux, obs["waypoint"], w.view(Tn, N))
    loss.backward()
    print("update", it, "hook calls:", sum(calls.values()), "params:", len(calls), "dups:", {k: v for k, v in calls.items() if v > 1})
    if red._buckets:
        print("  bucket pending:", [b["pending"] for b in red._buckets])
        for b in red._buckets:
            if b["pending"]:
                print("   waiting in bucket: ", [names[id(q)] for q in b["params"] if names[id(q)] not in calls])
    if it == 1:
        for k, v in seen_stacks.items():
            print('--- call', k); print(v)
    try:
        red.finish()
    except Exception as e:
        print("finish raised:", str(e)[:300])
    torch.cuda.synchronize()
