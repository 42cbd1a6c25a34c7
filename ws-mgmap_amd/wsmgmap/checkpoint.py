"""Checkpoint save / load / resume with the reference trainer's file contract (vlnce_baselines/common_trainer.py:71-76,
91-139), so that checkpoints written by either side load in the other:

    {"state_dict": policy.state_dict(),     # BasePolicy keys, no 'module.' prefix (:99 saves actor_critic.module)
     "config": <experiment config>,          # whatever the trainer passes (a yacs node in the reference)
     "extra_state": {"dagger_it": int}}      # optional (:101-102)

    file names  ckpt.<epoch>.pth  in CHECKPOINT_FOLDER; resume takes the NEWEST FILE BY MTIME (:126-128), loads it with
    strict=False (:131) and derives (start_dagger_it, start_epoch_it) from extra_state and the file name (:134-137).

The policy may be passed bare, wrapped in DistributedDataParallel, or as any object with a `.module` (the reference always
goes through `.module`).  With `wsmgmap.parallel.GradAllReducer` BatchNorm statistics are per rank: save from rank 0 (as the
reference does), or call `reducer.broadcast_buffers(policy)` first when another rank saves.

`load_checkpoint` also reads files whose pickled `config` refers to classes that are not importable here (yacs / habitat
`Config` in the authors' released checkpoints): unknown classes are materialised as plain attribute dicts instead of
failing the whole load.
"""
import os
import pickle

import torch


def _unwrap(policy):
    return policy.module if hasattr(policy, "module") and isinstance(policy.module, torch.nn.Module) else policy


def save_checkpoint(policy, checkpoint_folder, file_name, config=None, extra_state=None):
    """common_trainer.py:91-103."""
    checkpoint = {"state_dict": _unwrap(policy).state_dict(), "config": config}
    if extra_state is not None:
        checkpoint["extra_state"] = extra_state
    os.makedirs(checkpoint_folder, exist_ok=True)
    path = os.path.join(checkpoint_folder, file_name)
    torch.save(checkpoint, path)
    return path


class _Opaque(dict):
    """Stand-in for a pickled object whose class cannot be imported (e.g. yacs CfgNode without yacs installed)."""

    def __init__(self, *args, **kwargs):
        # (a class that pickle re-creates through REDUCE is CALLED with its constructor arguments: keep them, whatever they are)
        super().__init__()
        if args:
            self["__args__"] = args
        if kwargs:
            self["__kwargs__"] = kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)
        else:
            self["__state__"] = state

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class _LenientUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (_Opaque,), {"__module__": module})


class _LenientPickle:
    """pickle_module for torch.load: the standard unpickler, except that unknown classes do not abort the load."""
    __name__ = "wsmgmap_lenient_pickle"
    Unpickler = _LenientUnpickler
    load = staticmethod(lambda f, **kw: _LenientUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


def load_checkpoint(checkpoint_path, map_location="cpu", allow_pickle=True):
    """common_trainer.py:105-116.  Tensors-only files take torch's safe loader.  Files with a pickled config object (every
    checkpoint the reference writes) are REJECTED by it (pickle.UnpicklingError) and need the full unpickler, which can run
    code from the file: that second attempt is made only for that rejection — a corrupt or unreadable file raises as it is —
    only with allow_pickle=True (the default, because it is what the reference's `torch.load` does: load only checkpoints
    you trust), and says so in a warning."""
    try:
        return torch.load(checkpoint_path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        if not allow_pickle:
            raise
        import warnings
        warnings.warn(f"{checkpoint_path}: not a tensors-only file ({str(e).splitlines()[0][:120]}); loading it with the full "
                      "unpickler, as the reference does — only do this with checkpoints you trust", RuntimeWarning, stacklevel=2)
        return torch.load(checkpoint_path, map_location=map_location, weights_only=False, pickle_module=_LenientPickle)


def load_pretrained(policy, checkpoint_path, map_location="cpu"):
    """Load-for-finetune of `_setup_actor_critic` (:71-76): strict=False; returns the (missing, unexpected) report the
    reference logs.  Works on a bare policy and on a wrapper (keys get the 'module.' prefix exactly when there is one)."""
    ckpt = load_checkpoint(checkpoint_path, map_location=map_location)
    sd = ckpt["state_dict"]
    if policy is not _unwrap(policy):
        sd = {"module." + k: v for k, v in sd.items()}
    return policy.load_state_dict(sd, strict=False)


def newest_checkpoint(checkpoint_folder):
    """The last saved file of a folder BY MODIFICATION TIME (:127-128), or None."""
    if not os.path.isdir(checkpoint_folder):
        return None
    names = os.listdir(checkpoint_folder)
    if not names:
        return None
    names.sort(key=lambda x: os.path.getmtime(os.path.join(checkpoint_folder, x)))
    return os.path.join(checkpoint_folder, names[-1])


def resume_dagger(policy, checkpoint_folder, epochs, resume_ckpt=None, map_location="cpu"):
    """`resume_dagger` (:118-139) -> (start_dagger_it, start_epoch_it, report).  `resume_ckpt` is config.RESUME_CKPT; a
    non-empty folder overrides it (as in the reference); `epochs` is config.DAGGER.EPOCHS.  report = load_state_dict's
    (missing_keys, unexpected_keys), or None when nothing was loaded."""
    start_dagger_it, start_epoch_it, report = 0, 0, None
    ckpt_file = resume_ckpt
    newest = newest_checkpoint(checkpoint_folder)
    if newest is not None:
        ckpt_file = newest
    if ckpt_file is not None:
        previous = load_checkpoint(ckpt_file, map_location=map_location)
        report = _unwrap(policy).load_state_dict(previous["state_dict"], strict=False)
        start_dagger_it = previous["extra_state"]["dagger_it"]
        start_epoch_it = (int(ckpt_file.split("/")[-1].split(".")[1]) + 1) % epochs
        if start_epoch_it == 0:
            start_dagger_it += 1
    return start_dagger_it, start_epoch_it, report


# ----------------------------------------------------------------------------- end-of-epoch / end-of-iteration steps of the trainer
def _frozen_encoders_eval(policy):
    """`.train()` followed by `.net.depth_encoder.eval(); .net.rgb_encoder.eval()` (dagger_trainer.py:650-652,663-665)."""
    pol = _unwrap(policy)
    policy.train()
    pol.net.depth_encoder.eval()
    pol.net.rgb_encoder.eval()


def _barrier():
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()


def epoch_end(policy, checkpoint_folder, dagger_it, epoch, epochs, config=None, local_rank=0, on_epoch_end=None):
    """What the reference's training loop does after the last batch of an epoch (dagger_trainer.py:636-655):

      * rank 0 saves `ckpt.<dagger_it * EPOCHS + epoch>.pth` with `extra_state={'dagger_it': dagger_it}` (:636-640);
      * barrier (:642);
      * if EPOCHS > 10 (:644-655): auxiliary losses off, caches emptied, and on every third epoch rank 0 evaluates in-train:
        `policy.eval()`, the evaluation, `policy.train()` + the two frozen encoders back to eval; barrier; losses on again.

    on_epoch_end(policy, ckpt_path): the in-train evaluation hook — called where the reference calls
    `self._eval_checkpoint('', writer, 0, training=True, training_step=epoch)` (:649), with the policy in eval mode and the
    path of the checkpoint just written (None on ranks that did not write one).  Returns the checkpoint path (rank 0) or None."""
    from .common.aux_losses import AuxLosses
    path = None
    if local_rank == 0:
        path = save_checkpoint(policy, checkpoint_folder, f"ckpt.{dagger_it * epochs + epoch}.pth", config=config,
                               extra_state={"dagger_it": dagger_it})
    _barrier()
    if epochs > 10:
        AuxLosses.deactivate()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        if epoch % 3 == 0 and local_rank == 0 and on_epoch_end is not None:
            policy.eval()
            try:
                on_epoch_end(policy, path)
            finally:
                _frozen_encoders_eval(policy)
        _barrier()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        AuxLosses.activate()
    return path


def dagger_iteration_end(policy, num_processes, local_rank=0, on_iteration_end=None, ckpt_path=None):
    """After the last epoch of a DAgger iteration (dagger_trainer.py:657-678): auxiliary losses off, rank 0 evaluates in
    eval mode (`on_iteration_end(policy, ckpt_path)`, where the reference calls `_eval_checkpoint`), train mode + frozen
    encoders back to eval, barrier, and the map state re-created as zeros for NUM_PROCESSES environments."""
    from .common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    if local_rank == 0 and on_iteration_end is not None:
        policy.eval()
        try:
            on_iteration_end(policy, ckpt_path)
        finally:
            _frozen_encoders_eval(policy)
    _barrier()
    mm = getattr(_unwrap(policy).net, "rgb_mapping_module", None)
    if mm is not None:
        mm.full_global_map = torch.zeros([num_processes] + list(mm.full_global_map.shape[1:]), device=mm.full_global_map.device)
        # (same shape as the reference's zeros; kept as a broadcast view of one element per channel, as RGBMapping does)
        _, C, G1, G2 = mm.agent_view.shape
        mm.agent_view = torch.zeros(num_processes, C, 1, 1, device=mm.agent_view.device).expand(num_processes, C, G1, G2)
