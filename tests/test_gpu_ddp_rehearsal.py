"""The world_size > 1 branch of the training step (hipGraph for forward+backward on two streams, eager exchange +
update, barrier + max-over-ranks timing) executed on hardware: two ranks share the one visible GPU and exchange over
gloo (ISA_DIST_REHEARSAL=1 in bench.py).  RCCL itself needs one GPU per rank and is covered by the driver's N > 1 runs;
this keeps the rest of that path from being first executed there."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_gpu_bench_line():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, ISA_DIST_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--size", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["scaling"] == "weak"
