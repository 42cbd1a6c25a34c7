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
