#!/usr/bin/env python3
"""pred.py — the reference's single-image entry point (code/pred.py:12-22,45-50,110-123): image path in, palette PNG
out.  Per lib/prediction.py:33-50,116-124: read RGB, resize to 256x256 (bilinear), ImageEx + standardization (on the
device: isa_image_ex), semantic forward, probability of class 1 up-sampled to the original size with the
cv2.INTER_NEAREST index rule, `> 0.5` -> x255 -> float32 image -> `.convert('P')` -> <name>-fg_mask.png.
softmax(l)[1] > 0.5 is l1 > l0, i.e. the arg-max map the library already returns (isa_chan_argmax), so no probability
map is materialised.  `--synthetic` predicts one random image when no file is at hand (nothing ships with the repo).
The reference's hard-coded checkpoint and image paths (pred.py:25-26,112) are flags here."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import isa_amd  # noqa: F401,E402
from isa_amd.model import Model  # noqa: E402
from pred_list import H, W, nearest_upsample  # noqa: E402


def predict_file(model, image, out_dir, name):
    """image: uint8 RGB [h0,w0,3].  Returns the path of the written mask."""
    from PIL import Image
    from isa_amd.data import resize_bilinear
    x = resize_bilinear(torch.from_numpy(image[None]), (H, W))  # uint8 [1,H,W,3] on the device, bit-identical to PIL's
                                                               # BILINEAR resize (prediction.py:37); ImageEx follows there
    net = model.model
    net.eval()
    with torch.no_grad():
        _, sem_arg = net(False, x)                             # arg-max map == (softmax[:, 1] > 0.5)
    fg = sem_arg[0, 0].cpu().numpy() > 0.5
    full = nearest_upsample(fg, image.shape[0], image.shape[1])            # prediction.py:47-50
    fg_seg_pred_norm = (full * 255).astype(np.float32)                     # pred.py:117
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, name + '-fg_mask.png')
    Image.fromarray(fg_seg_pred_norm).convert('P').save(path)              # pred.py:122-123
    return path


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--image', default='', help='Path of the image')
    parser.add_argument('--model', default='', help='Path of the model (state_dict .pth)')
    parser.add_argument('--usegpu', action='store_true', help='kept for flag compatibility: the HIP path is the only path')
    parser.add_argument('--output', default=os.path.join(ROOT, 'outputs'), help='Path of the output directory')
    parser.add_argument('--n_workers', type=int, default=1, help='accepted for compatibility')
    parser.add_argument('--dataset', type=str, default='CVPPP')
    parser.add_argument('--synthetic', action='store_true', help='predict one random 530x500 image instead of --image')
    opt = parser.parse_args()
    assert opt.dataset in ['CVPPP', ]                          # pred.py:29
    assert opt.image or opt.synthetic, "give --image or --synthetic"
    if opt.synthetic:
        image, name = np.random.default_rng(0).integers(0, 256, (530, 500, 3), dtype=np.uint8), 'synthetic'
    else:
        from PIL import Image
        assert os.path.isfile(opt.image), 'Image : {} does not exists!'.format(opt.image)
        image, name = np.asarray(Image.open(opt.image).convert('RGB')), os.path.splitext(os.path.basename(opt.image))[0]
    model = Model(opt.dataset, 'ReSeg', 2, 32, use_instance_segmentation=False, load_model_path=opt.model, usegpu=True)
    path = predict_file(model, image, opt.output, name)
    print('wrote', path)


if __name__ == '__main__':
    main()
