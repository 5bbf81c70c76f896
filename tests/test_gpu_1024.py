"""BASELINE configs[4]: 1024x1024 inference.  The reference resizes every image to 256x256 (prediction.py:37) and its
image-global operators (SE pooling utils.py:415, the global softmaxes utils.py:512,649) see whole maps, so "tiled"
inference only has a defined result if tiles reproduce the untiled one.  On a 288 GB part the whole 1024x1024 map is
resident (the kernels tile it internally: 8x32-pixel LDS tiles, 128-pixel GEMM tiles; global reductions are
two-stage - per-workgroup partials, replicated atomics, a finalize), so the parity target is simply the untiled CPU
oracle at 1024x1024.  The fp16 attention path and the LDS tile-size sweep of the same config: tests/test_gpu_byname.py
and scripts/bench_sdp.py (profiles/r02_sdp_tile_sweep.txt)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import reseg_ref as R       # noqa: E402
from test_oracle_golden import assert_index_map    # noqa: E402


def test_1024_inference_vs_untiled_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import isa_amd  # noqa: F401
    from isa_amd.reseg import ReSeg
    x, _, _, _ = R.synth_batch(1, 1024, 1024, seed=9)
    sd = R.synth_state_dict(23, False)
    m = ReSeg(2, False, dtype=torch.float32)
    m.load_state_dict(sd)
    m.eval()
    sem_out, sem_arg = m(False, x)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = R.reseg_forward(sd, x, use_instance_seg=False)
    err = float((sem_out.cpu() - ref["sem_out"]).abs().max() / ref["sem_out"].abs().max())
    assert err < 1e-3, err
    margin = (ref["sem_out"][:, 1] - ref["sem_out"][:, 0]).abs().numpy()
    assert_index_map(ref["sem_argmax"][:, 0].numpy() != 0, sem_arg.cpu().numpy()[:, 0] != 0, margin,
                     float(ref["sem_out"].abs().max()), "sem_argmax 1024", rel=1e-5)
    # hipGraph replay at this size, bf16 storage: same mask wherever the oracle's margin exceeds 10 % of the logit scale
    mb = ReSeg(2, False, dtype=torch.bfloat16)
    mb.load_state_dict(sd)
    mb.eval()
    for _ in range(3):
        so, sa = mb.infer_graphed(x)
    torch.cuda.synchronize()
    l32 = ref["sem_out"]
    keep = (l32[:, 1] - l32[:, 0]).abs() > 0.1 * float(l32.abs().max())
    got = so.cpu()
    agree = ((got[:, 1] > got[:, 0]) == (l32[:, 1] > l32[:, 0]))[keep].float().mean()
    assert float(keep.float().mean()) > 0.3 and float(agree) > 0.999, (float(keep.float().mean()), float(agree))
