"""Reads of memory the step did not write: ISA_ARENA_POISON=1 fills every un-zeroed arena buffer, the reduction workspace
and the slab arena with 0xFF (NaN in every float type) each step.  The switch is read at import time, so the checks run
in a child interpreter: the 64x64 training step against the reference's gradients, the graph-replayed step, and the
inference pass all have to stay finite and within their normal bounds."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_suite_subset_with_poisoned_arenas():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, ISA_ARENA_POISON="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_train.py"), os.path.join(ROOT, "tests", "test_gpu_model.py"),
           "-k", "gradients_vs_reference_f64 or graph or inference_vs_reference_golden or bf16"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout and " failed" not in out.stdout
