#!/usr/bin/env python3
"""pred_list.py — batched foreground prediction over a list of images, the working part of the reference's
code/pred_list.py:16-99 (flags --lst/--model/--usegpu/--dataset; outputs <name>.png and <name>-fg_mask.png under
outputs/<dataset>/<model dir>-<model name>/<subset>/).

Per image (lib/prediction.py:33-50,116-124): read RGB, resize to 256x256 (bilinear, `image_resizer`), ImageEx +
standardization (on the device here: isa_image_ex), network forward, softmax > 0.5 (= the arg-max map), nearest-neighbour up-sampling
to the original size (cv2.INTER_NEAREST index rule), x255, PNG.  The instance outputs of the reference
(-ins_mask*.png, -n_objects.npy) come from `Prediction.cluster`, which is dead at HEAD (the
GT-free instance path raises UnboundLocalError, reseg.py:126) and are not produced.
`--synthetic N` runs N random images instead of a list (no files needed)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import isa_amd  # noqa: F401,E402
from isa_amd.model import Model  # noqa: E402
from isa_amd.data import resize_bilinear  # noqa: E402

H = W = 256                                             # data_settings.py: IMAGE_HEIGHT / IMAGE_WIDTH


def nearest_upsample(mask, out_h, out_w):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): src = min(floor(dst * scale), size - 1)."""
    h, w = mask.shape
    ys = np.minimum((np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return mask[ys][:, xs]


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--lst', default='', help='Text file that contains image paths')
    parser.add_argument('--model', default='', help='Path of the model (state_dict .pth)')
    parser.add_argument('--usegpu', action='store_true', help='kept for flag compatibility: the HIP path is the only path')
    parser.add_argument('--dataset', type=str, default='CVPPP')
    parser.add_argument('--output', default='', help='output directory (default: the reference layout under outputs/)')
    parser.add_argument('--batch', type=int, default=16)
    parser.add_argument('--synthetic', type=int, default=0, help='predict N random images instead of --lst')
    opt = parser.parse_args()
    assert opt.dataset in ['CVPPP', ]                    # pred_list.py:26
    assert opt.lst or opt.synthetic, "give --lst or --synthetic N"

    if opt.synthetic:
        rng = np.random.default_rng(0)
        names = ['synthetic_%04d' % i for i in range(opt.synthetic)]
        loaders = [lambda i=i: rng.integers(0, 256, (300 + 7 * (i % 5), 330, 3), dtype=np.uint8) for i in range(opt.synthetic)]
        subset, tag = 'synthetic', 'random'
    else:
        from PIL import Image
        paths = [str(p) for p in np.atleast_1d(np.loadtxt(opt.lst, dtype='str', delimiter=','))]
        names = [os.path.splitext(os.path.basename(p))[0] for p in paths]
        loaders = [lambda p=p: np.asarray(Image.open(p).convert('RGB')) for p in paths]
        subset = os.path.basename(opt.lst).split('_')[0]
        tag = (os.path.basename(os.path.dirname(opt.model)) + '-' + os.path.splitext(os.path.basename(opt.model))[0]) \
            if opt.model else 'random'
    out_dir = opt.output or os.path.join(ROOT, 'outputs', opt.dataset, tag, subset)
    os.makedirs(out_dir, exist_ok=True)

    from PIL import Image
    model = Model(opt.dataset, 'ReSeg', 2, 32, use_instance_segmentation=False, load_model_path=opt.model, usegpu=True)
    net = model.model
    net.eval()
    done = 0
    for s in range(0, len(names), opt.batch):
        imgs = [ld() for ld in loaders[s:s + opt.batch]]
        # resize on the device (isa_resize_bilinear_u8, bit-identical to PIL's BILINEAR): one launch per source size
        x = torch.cat([resize_bilinear(torch.from_numpy(im[None]), (H, W)) for im in imgs])   # uint8 [B,H,W,3]; ImageEx follows
        _, sem_arg = net.infer_graphed(x) if len(imgs) == opt.batch else net(False, x)
        # softmax(l)[1] > 0.5 (pred.py:117-121) is l1 > l0: the arg-max map the library already returns
        fg = (sem_arg[:, 0] > 0.5).to(torch.uint8).cpu().numpy()
        for im, name, m in zip(imgs, names[s:s + opt.batch], fg):
            d = os.path.join(out_dir, name)
            os.makedirs(d, exist_ok=True)
            full = nearest_upsample(m, im.shape[0], im.shape[1]) * 255
            Image.fromarray(im).save(os.path.join(d, name + '.png'))
            Image.fromarray(full.astype(np.uint8)).save(os.path.join(d, name + '-fg_mask.png'))
            done += 1
    print('wrote %d predictions under %s' % (done, out_dir))


if __name__ == '__main__':
    main()
