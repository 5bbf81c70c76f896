#!/usr/bin/env python3
"""pred.py — semantic prediction like the reference's code/pred.py:114-123: prob map -> >0.5 -> x255.
Input: a .npy holding the 21-channel ImageEx tensor [21,H,W] or [B,21,H,W] (the colour-space expansion
itself, lib/utils.py:90-113, is SURVEY §8 f-1 and not part of this round), or --synthetic."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import isa_amd  # noqa: F401,E402
from isa_amd.model import Model  # noqa: E402
from isa_amd.data import synth_batch  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('--input', default='', help='.npy with the 21-channel input tensor')
parser.add_argument('--synthetic', action='store_true')
parser.add_argument('--model', default='', help='path of a state_dict (.pth)')
parser.add_argument('--usegpu', action='store_true', default=True)
parser.add_argument('--output', default='pred_mask.npy')
opt = parser.parse_args()

model = Model('CVPPP', 'ReSeg', 2, 32, use_instance_segmentation=False, load_model_path=opt.model, usegpu=True)
if opt.synthetic or not opt.input:
    x = synth_batch(1, 256, 256, seed=0)[0]
else:
    x = torch.from_numpy(np.load(opt.input)).float()
    if x.dim() == 3:
        x = x.unsqueeze(0)
prob = model.predict(x)                                   # [B,2,H,W] softmax, CPU
mask = (prob[:, 1] > 0.5).numpy().astype(np.uint8) * 255  # pred.py:117-121
np.save(opt.output, mask)
print('wrote', opt.output, mask.shape, 'foreground fraction %.4f' % float((mask > 0).mean()))
