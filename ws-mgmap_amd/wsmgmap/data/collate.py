"""Batch assembly for the policy update.

`collate_fn` is the host form with the reference's exact semantics (dagger_trainer.py:40-113).  `DeviceCollator`
produces the tensors the trainer hands to `_update_agent` (:614-625: every observation float32 on the device) but
moves the episodes in their compact on-disk dtypes and pads / interleaves / converts them on the GPU with
`wsmg_collate_pad`: half (float16 sensors) to an eighth (uint8) of the reference's PCIe bytes and no host-side
float32 materialisation."""
import ctypes

import numpy as np
import torch

from .. import _abi

LIMITED_LEN_BY_GPU = 200   # dagger_trainer.py:82
_DT = {np.dtype(np.float16): 0, np.dtype(np.uint8): 1, np.dtype(np.int64): 2, np.dtype(np.float32): 3}
_SPARSE = "rgb_ego_map__"      # the sparse ego map's arrays (codec.sparse_pack_ego): they travel as raw bytes, one kernel expands them


def _pad(t, max_len, fill):
    n = max_len - t.size(0)
    if n <= 0:
        return t[:max_len]
    return torch.cat([t, torch.full_like(t[0:1], fill).expand(n, *t.size()[1:])], dim=0)


def collate_fn(batch):
    """[(obs, prev_actions, oracle_actions, weights)] -> (obs [T*N, ...], prev_actions [T*N, 2], not_done_masks [T*N, 1],
    corrected_actions [T, N, 2], weights [T, N]) on the host, dtypes as stored."""
    from .codec import densify
    as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.array(a))  # noqa: E731
    obs_l = [{k: as_t(v) for k, v in densify(b[0]).items()} for b in batch]
    prev_l, corr_l, w_l = [as_t(b[1]) for b in batch], [as_t(b[2]) for b in batch], [as_t(b[3]) for b in batch]
    T = min(max(p.size(0) for p in prev_l), LIMITED_LEN_BY_GPU)
    obs = {}
    for k in obs_l[0]:
        s = torch.stack([_pad(o[k], T, 1.0) for o in obs_l], dim=1)
        obs[k] = s.view(-1, *s.size()[2:])
    prev = torch.stack([_pad(p, T, 0) for p in prev_l], dim=1)
    corr = torch.stack([_pad(c, T, 0) for c in corr_l], dim=1)
    wts = torch.stack([_pad(w, T, 0) for w in w_l], dim=1)
    masks = torch.ones_like(wts, dtype=torch.float)
    masks[0] = 0
    return obs, prev.view(-1, 2), masks.view(-1, 1), corr, wts


def plan_batch(batch):
    """The staging layout of one batch — all episodes of all sensors back to back, 16-byte aligned — without touching the data:
    -> (plan, meta).  plan = [(name, per-episode arrays cut to T steps, pad value)]; meta is small and picklable: it describes
    the packed bytes to whoever issues the device side (`DeviceCollator.launch`), possibly another process."""
    N = len(batch)
    lengths = [int(len(b[1])) for b in batch]
    T = min(max(lengths), LIMITED_LEN_BY_GPU)
    # episodes longer than T are cut before they travel
    plan = [(k, [np.ascontiguousarray(np.asarray(b[0][k])[:T]) for b in batch], 1.0) for k in batch[0][0] if not k.startswith(_SPARSE)]
    sparse = None
    if _SPARSE + "bits" in batch[0][0]:
        # the sparse ego map (codec.sparse_pack_ego): bits / off cut to T steps, base to T + 1, the values to what those steps hold
        C, H, W = (int(x) for x in np.asarray(batch[0][0][_SPARSE + "shape"]))
        cut = [min(n, T) for n in lengths]
        plan.append((_SPARSE + "bits", [np.ascontiguousarray(np.asarray(b[0][_SPARSE + "bits"])[:c]) for b, c in zip(batch, cut)], 1.0))
        plan.append((_SPARSE + "off", [np.ascontiguousarray(np.asarray(b[0][_SPARSE + "off"])[:c]).view(np.uint8) for b, c in zip(batch, cut)], 1.0))
        plan.append((_SPARSE + "base", [np.ascontiguousarray(np.asarray(b[0][_SPARSE + "base"])[:c + 1]) for b, c in zip(batch, cut)], 1.0))
        plan.append((_SPARSE + "vals", [np.ascontiguousarray(np.asarray(b[0][_SPARSE + "vals"])[:int(np.asarray(b[0][_SPARSE + "base"])[c])])
                                         for b, c in zip(batch, cut)], 1.0))
        sparse = dict(C=C, H=H, W=W)
    plan.append(("__prev", [np.ascontiguousarray(np.asarray(b[1], dtype=np.float32)[:T]) for b in batch], 0.0))
    plan.append(("__corr", [np.ascontiguousarray(np.asarray(b[2], dtype=np.float32)[:T]) for b in batch], 0.0))
    plan.append(("__wts", [np.ascontiguousarray(np.asarray(b[3], dtype=np.float32)[:T]) for b in batch], 0.0))
    offs, total, sensors = [], 0, []
    for name, arrs, pad in plan:
        if arrs[0].dtype not in _DT:
            raise _abi.WsmgError(f"unsupported on-disk dtype {arrs[0].dtype} in the trajectory cache")
        for a in arrs:
            total = (total + 15) & ~15
            offs.append(total)
            total += a.nbytes
        sensors.append((name, tuple(arrs[0].shape[1:]), _DT[arrs[0].dtype], float(pad)))
    meta = dict(N=N, T=T, lengths=[min(n, T) for n in lengths], offsets=offs, total=total + 16, sensors=sensors)
    if sparse is not None:
        meta["sparse_ego"] = sparse
    dd = _host_dedup(plan, meta)
    if dd is not None:
        meta["dedup"] = dd
    return plan, meta


def _host_dedup(plan, meta):
    """The instruction dedup of the padded time-major batch, computed HERE — in the decode worker, from the token arrays that are on the
    host anyway — instead of by a kernel + host read-back in every forward pass (models/encoders/instruction_encoder.py::_dedup_fused
    computes the same thing on the device): distinct rows of the [T*N, L] token matrix (row t*N + n = episode n's step t, or the pad
    value 1 past its end) in order of first appearance, the inverse map, and each distinct row's count of non-zero tokens.  Small and
    picklable: (uniq rows as a list of lists, inverse list, lengths list).  None when the batch has no integer `instruction` sensor."""
    ent = next((e for e in plan if e[0] == "instruction"), None)
    if ent is None or not np.issubdtype(ent[1][0].dtype, np.integer) or ent[1][0].ndim != 2:
        return None
    _, arrs, pad = ent
    T, N, L = meta["T"], meta["N"], arrs[0].shape[1]
    tok = np.full((T, N, L), int(pad), dtype=np.int64)
    for n, a in enumerate(arrs):
        tok[:a.shape[0], n] = a
    rows = tok.reshape(T * N, L)
    seen, uniq, inverse = {}, [], np.empty(T * N, dtype=np.int64)
    for i in range(T * N):
        k = rows[i].tobytes()
        j = seen.get(k)
        if j is None:
            j = seen[k] = len(uniq)
            uniq.append(rows[i])
        inverse[i] = j
    u = np.stack(uniq)
    return dict(uniq=u.tolist(), inverse=inverse.tolist(), lengths=(u != 0).sum(1).tolist())


def pack_batch(plan, meta, host_u8):
    """Copy the planned arrays into `host_u8` (a uint8 numpy view of at least meta['total'] bytes: pinned staging, or a slot of the
    shared-memory ring of the process feeder)."""
    if host_u8.size < meta["total"]:
        raise _abi.WsmgError(f"staging buffer of {host_u8.size} bytes for a batch of {meta['total']}")
    it = iter(meta["offsets"])
    for _, arrs, _ in plan:
        for a in arrs:
            o = next(it)
            host_u8[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)


class DeviceCollator:
    """collate + `.float().to(device)` of the reference, assembled on the GPU.  One pinned staging buffer per call
    holds all episodes of all sensors back to back; it goes to the device in ONE asynchronous copy on `stream`, then
    one `wsmg_collate_pad` launch per sensor writes the padded float32 tensors."""

    def __init__(self, device="cuda", ego_map_nhwc_bf16=False):
        """ego_map_nhwc_bf16 (round 4, for policies built with COMPUTE_DTYPE = bf16): `rgb_ego_map` leaves the collate as a bf16
        tensor of logical shape [T*N, C, E, E] in channels-last memory — what the map encoder's first convolution reads — instead of
        float32 NCHW; the policy then skips its NCHW float32 -> NHWC bf16 pass (0.35 ms of a 10.7 ms update at B = 512).  Same
        values, bit for bit (float16 -> float32 -> bf16 either way).  Off by default: the reference's trainer hands the policy
        float32 NCHW observations."""
        self.device = torch.device(device)
        self._pinned = None
        self.ego_map_nhwc_bf16 = bool(ego_map_nhwc_bf16)
        self.host_dedup = True        # hand the batch's instruction dedup (computed on the host by plan_batch) to the policy

    def _staging(self, nbytes):
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        return self._pinned

    def __call__(self, batch, stream=None):
        plan, meta = plan_batch(batch)
        host = self._staging(meta["total"])
        pack_batch(plan, meta, host.numpy())
        return self.launch(meta, host, stream)

    def launch(self, meta, host, stream=None):
        """The device side of one packed batch: `host` (a uint8 CPU tensor holding the bytes `pack_batch` wrote — pinned for an
        asynchronous copy) -> the tensors the trainer hands to `_update_agent`.  The caller may reuse `host` once `stream` has
        passed this point."""
        stream = stream or torch.cuda.current_stream(self.device)
        N, T, total, offs = meta["N"], meta["T"], meta["total"], meta["offsets"]
        with torch.cuda.stream(stream):
            dev = torch.empty(total, dtype=torch.uint8, device=self.device)
            base = dev.data_ptr()
            # the two small tables go up FIRST and from pinned memory: a `torch.tensor(list, device=...)` is a synchronous copy, and
            # queued behind the batch's 735 MB it made the host wait out the whole transfer on every batch (the consumer loop of
            # the feeder then topped out at 28 batches/s whatever the number of workers)
            lens_dev = torch.tensor(meta["lengths"], dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
            ptrs_dev = torch.tensor([base + o for o in offs], dtype=torch.int64).pin_memory().to(self.device, non_blocking=True)
            dev.copy_(host[:total], non_blocking=True)
            out, row = {}, 0
            sp_rows = {}
            for name, shape, code, pad in meta["sensors"]:
                if name.startswith(_SPARSE):         # raw bytes of the sparse ego map: expanded by one kernel below
                    sp_rows[name[len(_SPARSE):]] = row
                    row += N
                    continue
                elems = int(np.prod(shape, dtype=np.int64))
                if (self.ego_map_nhwc_bf16 and name == "rgb_ego_map" and code == _DT[np.dtype(np.float16)] and len(shape) == 3
                        and shape[0] % 64 == 0 and (shape[1] * shape[2]) % 4 == 0 and T * N <= 65535):
                    C, E1, E2 = shape
                    dst = torch.empty((T, N, E1, E2, C), dtype=torch.bfloat16, device=self.device)
                    _abi.call("wsmg_collate_pad_nhwc_bf16", ctypes.c_void_p(ptrs_dev.data_ptr() + 8 * row), ctypes.c_void_p(lens_dev.data_ptr()),
                              N, T, C, E1 * E2, float(pad), ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
                    row += N
                    out[name] = dst.permute(0, 1, 4, 2, 3)        # [T, N, C, E, E] over channels-last memory
                    continue
                dst = torch.empty((T, N) + tuple(shape), dtype=torch.float32, device=self.device)
                _abi.call("wsmg_collate_pad", ctypes.c_void_p(ptrs_dev.data_ptr() + 8 * row), ctypes.c_void_p(lens_dev.data_ptr()),
                          N, T, max(elems, 1), code, float(pad), ctypes.c_void_p(dst.data_ptr()),
                          ctypes.c_void_p(stream.cuda_stream))
                row += N
                out[name] = dst
            if sp_rows:
                sp = meta["sparse_ego"]
                if not self.ego_map_nhwc_bf16:
                    raise _abi.WsmgError("a trajectory cache with the sparse ego map needs DeviceCollator(ego_map_nhwc_bf16=True) "
                                         "(the bf16 policy's channels-last input); recode it without --sparse-ego for float32 policies")
                C, HW = sp["C"], sp["H"] * sp["W"]
                dst = torch.empty((T, N, sp["H"], sp["W"], C), dtype=torch.bfloat16, device=self.device)
                at = lambda k: ctypes.c_void_p(ptrs_dev.data_ptr() + 8 * sp_rows[k])  # noqa: E731
                _abi.call("wsmg_collate_ego_sparse_nhwc_bf16", at("bits"), at("off"), at("base"), at("vals"), ctypes.c_void_p(lens_dev.data_ptr()),
                          N, T, C, HW, 1.0, ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
                out["rgb_ego_map"] = dst.permute(0, 1, 4, 2, 3)
            dev.record_stream(stream)
            masks = torch.ones(T, N, dtype=torch.float32, device=self.device)
            masks[0] = 0
            ready = torch.cuda.Event()
            ready.record(stream)
        prev, corr, wts = out.pop("__prev"), out.pop("__corr"), out.pop("__wts")
        obs = {k: (v.reshape(-1, *v.shape[2:]) if v.is_contiguous() else v.flatten(0, 1)) for k, v in out.items()}
        if "instruction" in obs:     # the policy's parameter-free preprocessing may start as soon as the collate has run (ops/core.py)
            from ..ops import mark_inputs_ready
            mark_inputs_ready(obs["instruction"], ready)
        dd = meta.get("dedup")
        if dd is not None and self.host_dedup:
            # the dedup came with the batch (plan_batch): the policy finds it beside the token tensor (ops.attach_instruction_dedup) and runs neither
            # the dedup kernel nor its host read-back — a forward pass without a single host synchronisation
            with torch.cuda.stream(stream):
                up = lambda v: torch.tensor(v, dtype=torch.int64).pin_memory().to(self.device, non_blocking=True)   # noqa: E731
                lens = torch.tensor(dd["lengths"], dtype=torch.int64)
                from ..ops import attach_instruction_dedup
                attach_instruction_dedup(obs["instruction"], (up(dd["uniq"]), up(dd["inverse"]), lens, lens.pin_memory().to(self.device, non_blocking=True)))
        return obs, prev.view(-1, 2), masks.view(-1, 1), corr, wts
