"""wsmgmap — MI355X-native implementation of WS-MGMap's per-step policy hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, autograd tape, torch.distributed).
Compute of the three hot operators: hand-written gfx950 HIP kernels in libwsmgmap.so, bound
through the C ABI of include/wsmgmap.h (see _abi.py).  There is no CPU / eager fallback.
"""
import sys

from . import _abi  # noqa: F401

__all__ = ["_abi", "install", "uninstall", "installed"]

# reference module name -> module of this package that takes its place.  These are the modules of the per-step policy path
# (SURVEY.md 8a/8b): the policy the trainers import (common_trainer.py:24), the auxiliary-loss registry they activate / clear
# (dagger_trainer.py:25) and everything those two import from the reference's own tree (policy.py:10-12, mg_map_policy.py:12-16).
_ALIASES = {
    "vlnce_baselines.models.policy": "wsmgmap.models.policy",
    "vlnce_baselines.models.mg_map_policy": "wsmgmap.models.mg_map_policy",
    "vlnce_baselines.models.encoders.instruction_encoder": "wsmgmap.models.encoders.instruction_encoder",
    "vlnce_baselines.models.encoders.unet_encoder": "wsmgmap.models.encoders.unet_encoder",
    "vlnce_baselines.models.encoders.resnet_encoders": "wsmgmap.models.encoders.resnet_encoders",
    "vlnce_baselines.models.encoders.map_encoder": "wsmgmap.models.encoders.map_encoder",
    "vlnce_baselines.common.aux_losses": "wsmgmap.common.aux_losses",
    "vlnce_baselines.common.distributions": "wsmgmap.common.distributions",
    "vlnce_baselines.common.rgb_mapping": "wsmgmap.common.rgb_mapping",
}
_PARENTS = sorted({n.rsplit(".", k)[0] for n in _ALIASES for k in range(1, n.count(".") + 1)})


class _AliasFinder:
    """First on sys.meta_path: an aliased reference name loads as the product module of _ALIASES.  Going through the import
    system (instead of only seeding sys.modules) keeps every spelling working — `import a.b.c as m` and `a.b.c.X` need the parent
    packages imported and the child bound as their attribute, which the machinery does for a module it loaded itself."""

    @staticmethod
    def find_spec(fullname, path=None, target=None):
        if fullname not in _ALIASES:
            return None
        from importlib.util import spec_from_loader
        return spec_from_loader(fullname, _AliasFinder, origin="wsmgmap.install: " + _ALIASES[fullname])

    @staticmethod
    def create_module(spec):
        import importlib
        return importlib.import_module(_ALIASES[spec.name])    # the existing module object: its own __name__ / __spec__ are kept

    @staticmethod
    def exec_module(module):
        pass


class _ShellFinder:
    """Last on sys.meta_path: when NO reference checkout is importable (tests, tools that only want the policy by its reference
    name), the parent packages of the aliased names exist as empty shells.  With a checkout on the path its own packages win."""

    @staticmethod
    def find_spec(fullname, path=None, target=None):
        if fullname not in _PARENTS:
            return None
        from importlib.util import spec_from_loader
        return spec_from_loader(fullname, _ShellFinder, origin="wsmgmap.install: package shell", is_package=True)

    @staticmethod
    def create_module(spec):
        return None

    @staticmethod
    def exec_module(module):
        module.__path__ = []


def install(strict=True):
    """Zero-edit boundary: make the reference's trainers construct THIS package's policy without touching their sources.

    The trainers do `from vlnce_baselines.models.policy import BasePolicy` (common_trainer.py:24) and
    `from vlnce_baselines.common.aux_losses import AuxLosses` (dagger_trainer.py:25).  `install()` puts a finder in front of
    `sys.meta_path` that loads those names (and the other modules of the path: _ALIASES) as the product's modules — the
    reference's files of the same names are never read — so both resolve to this package's `BasePolicy` and to the ONE
    `AuxLosses` registry the policy registers its losses into (two registries would mean silently empty auxiliary losses).
    Call it before anything imports `vlnce_baselines`: `python -m wsmgmap run.py ...` does, and so does a `sitecustomize.py`
    holding `import wsmgmap; wsmgmap.install()`.

    strict: raise if a reference module of the path was imported BEFORE the call (its classes may already be bound somewhere);
    strict=False replaces it anyway.  Idempotent.  Returns the names that are aliased."""
    early = [n for n in _ALIASES if n in sys.modules and sys.modules[n].__name__ != _ALIASES[n]]
    if early and strict:
        raise ImportError("wsmgmap.install() must run before the reference's policy modules are imported; already loaded: "
                          + ", ".join(early) + " (install(strict=False) replaces them anyway)")
    for n in early:
        import importlib
        mod = importlib.import_module(_ALIASES[n])
        sys.modules[n] = mod
        parent = sys.modules.get(n.rpartition(".")[0])
        if parent is not None:
            setattr(parent, n.rpartition(".")[2], mod)
    if _AliasFinder not in sys.meta_path:
        sys.meta_path.insert(0, _AliasFinder)
    if _ShellFinder not in sys.meta_path:
        sys.meta_path.append(_ShellFinder)
    return list(_ALIASES)


def uninstall():
    """Undo install() (tests): the finders go, and so do the aliased names and package shells they loaded."""
    for f in (_AliasFinder, _ShellFinder):
        while f in sys.meta_path:
            sys.meta_path.remove(f)
    for n in _ALIASES:
        m = sys.modules.get(n)
        if m is not None and m.__name__ == _ALIASES[n]:
            del sys.modules[n]
            parent = sys.modules.get(n.rpartition(".")[0])
            if parent is not None and getattr(parent, n.rpartition(".")[2], None) is m:
                delattr(parent, n.rpartition(".")[2])
    for n in reversed(_PARENTS):
        m = sys.modules.get(n)
        if m is not None and getattr(getattr(m, "__spec__", None), "loader", None) is _ShellFinder:
            del sys.modules[n]


def installed():
    return _AliasFinder in sys.meta_path


def _prefer_rocblas():
    """The policy's dense GEMMs outside the conv engine are small (heads, RNN projections, [256,512] x [512,256] weight
    gradients).  PyTorch-ROCm's default hipBLASLt heuristic picks a 256x256 macro-tile for several of them — ONE
    workgroup, 118 us for a 67-MFLOP product that rocBLAS runs in 7 us (measured on MI355X, tools/bench notes in
    DESIGN.md) — so rocBLAS is selected process-wide when the package is imported.  WSMG_KEEP_BLAS=1 leaves PyTorch's
    choice alone."""
    from .debug import sw
    if sw.keep_blas:
        return
    try:
        import warnings
        import torch
        if torch.version.hip:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.backends.cuda.preferred_blas_library("cublas")   # "cublas" = rocBLAS on ROCm builds
    except Exception:   # pragma: no cover  (older torch: keep the default)
        pass


_prefer_rocblas()
