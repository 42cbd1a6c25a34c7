"""GPU tests, round 6 (VERDICT r05 "Next round" + ADVICE r05): the in-process recurrent-core fallback of bench.py, the gradient
exchange's bucket hold beside the chained core, the fused fp8 attention's CU-derived barrier bound and reported timeout, the
BatchNorm sums / finalize folded into their neighbours, the new window-kernel shapes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _aux_losses_off():
    from wsmgmap.common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    AuxLosses.clear()
    yield
    AuxLosses.deactivate()
    AuxLosses.clear()


def _bench_line(env_extra, *flags, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-f32",
                        "--no-cpu-baseline", "--no-other-configs", "--prewarm-s", "0", *flags],
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


# ----------------------------------------------------------------------------- bench.py cannot die on a persistent-kernel timeout
@pytest.mark.parametrize("dp", [False, True])
def test_bench_falls_back_in_process_on_an_injected_timeout(dp):
    """VERDICT r05 item 4.  A persistent-kernel timeout bit (injected through wsmg_rnn_debug_inject after the 2nd update, as a kernel
    whose spin ran out would set it) must yield a VALID line, with the fallback named, from the same process: single process, and
    under a one-rank RCCL process group, where the bit travels through GradAllReducer.finish()'s cross-rank error flag
    (reference: the DDP-wrapped update of common_trainer.py:35-38,61-66, dagger_trainer.py:505-543)."""
    env = {"WSMG_BENCH_INJECT_TIMEOUT": "2"}
    if dp:
        env.update(WSMG_BENCH_DP_ONE_RANK="1", MASTER_PORT="29533")
    line, err = _bench_line(env)
    assert line["value"] > 0 and line["steps"] == 4 and np.isfinite(line["loss"])
    rc = line["recurrent_core"]
    assert rc["fallback_level"] == 1 and rc["recurrent_core"].startswith("staged (fallback"), rc
    assert "timed out" in rc["reasons"][0] or "reported an error" in rc["reasons"][0] or "status word" in rc["reasons"][0], rc
    if dp:
        assert line["data_parallel"]["recurrent_core"].startswith("staged (fallback"), line["data_parallel"]
        assert isinstance(line["data_parallel"].get("per_bucket_allreduce_ms"), list), line["data_parallel"]


def test_bench_line_names_the_default_core_without_a_timeout():
    line, _ = _bench_line({})
    assert line["recurrent_core"]["fallback_level"] == 0 and line["recurrent_core"]["recurrent_core"].startswith("chained"), line["recurrent_core"]
