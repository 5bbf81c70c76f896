"""isa_sdp_attention under rocprofv3 (kernel trace or one --pmc counter per pass): 40 eager calls over rotating operand sets
(K / V from HBM), bf16 and f32, one query per image, d = 12, 2 interleaved heads, no mask.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 scripts/pmc_sdp.py
       rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 scripts/pmc_sdp.py   (and WRITE_SIZE)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import isa_amd  # noqa
from isa_amd import attention_ops as A

for dtype in (torch.bfloat16, torch.float32):
    for (b, L) in ((16, 65536), (4, 1048576)):
        esz = 2 if dtype == torch.bfloat16 else 4
        nsets = max(2, -(-(300 << 20) // (2 * b * L * 24 * esz)) + 1)
        sets = [(torch.randn(b, 1, 24, device="cuda").to(dtype), torch.randn(b, L, 24, device="cuda").to(dtype),
                 torch.randn(b, L, 24, device="cuda").to(dtype)) for _ in range(nsets)]
        for i in range(40):
            q, k, v = sets[i % nsets]
            A.scaled_dot_product_attention(q, k, v, 12 ** 0.5, None, return_attn=False, heads=2)
        torch.cuda.synchronize()
        print("%s b=%d L=%d: algorithmic bytes per call %.1f MB" % (str(dtype).split(".")[1], b, L, 2 * b * L * 24 * esz / 1e6))
