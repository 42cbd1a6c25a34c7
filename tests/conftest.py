import os
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

warnings.filterwarnings("ignore", message="Default grid_sample and affine_grid")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle runs on the host.  Under a control-group CPU quota (the GPU boxes: 16 CPUs of a 256-thread host) torch's default
    # of one thread per logical CPU only makes the threads take turns being throttled: the full-size backward oracle took 128 s
    # that way.  One thread per CPU the process may actually use.
    try:
        import torch
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            torch.set_num_threads(max(1, min(torch.get_num_threads(), int(float(q) / float(per) + 0.999))))
    except (OSError, ValueError, ImportError):
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _no_stale_rnn_timeouts(request):
    """GPU tests: a persistent-kernel timeout must not outlive the test that caused it — the status word is sticky until read, and a
    bit left set would make the NEXT test's check raise for something it did not do (or hide that this test's kernels timed out).
    A test that provokes a timeout on purpose reads the word itself (ops.check_rnn_status raises and clears)."""
    yield
    if "gpu" not in request.keywords or not _has_gpu():
        return
    import torch
    from wsmgmap import _abi
    torch.cuda.synchronize()
    left = int(_abi.lib().wsmg_rnn_status(1))          # read AND clear, so that one offender does not fail every later test
    assert left == 0, f"this test left persistent-kernel timeout bits {left:#x} set (a kernel of it timed out, unreported)"
