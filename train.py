#!/usr/bin/env python3
"""train.py — entry point with the reference's flags (code/train.py:18-37) on the MI355X engine.
The reference trains from an LMDB that is not in its repository; this runs the same loop on synthetic
collated batches (data.py) unless a loader is plugged in.  Hyper-parameters: settings/CVPPP/training_settings.py.

Single GPU:   python train.py --batchsize 8
Data parallel (BASELINE configs[3]: bs=64, 512x512, 8 GPUs; one process per GPU, RCCL gradient all-reduce):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
         train.py --batchsize 64 --size 512
--batchsize is the GLOBAL batch (the reference's flag, one process there); each rank takes batchsize / world images of
its own data shard (seeded per rank), only rank 0 logs and writes checkpoints (isa_amd/parallel.py: policy)."""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import isa_amd  # noqa: F401,E402
from isa_amd.model import Model  # noqa: E402
from isa_amd.data import SyntheticLoader  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument('--model', default='', help="Filepath of trained model (to continue training) [Default: '']")
parser.add_argument('--usegpu', action='store_true', default=True, help='Enables the GPU [always on in this build]')
parser.add_argument('--nepochs', type=int, default=800, help='Number of epochs to train for [Default: 800]')
parser.add_argument('--batchsize', type=int, default=2, help='Batch size [Default: 2]')
parser.add_argument('--debug', action='store_true', help='Activates debug mode [Default: False]')
parser.add_argument('--nworkers', type=int, default=2, help='accepted for compatibility (synthetic data needs none)')
parser.add_argument('--dataset', type=str, default='CVPPP', help='Name of the dataset which is "CVPPP"')
parser.add_argument('--iters-per-epoch', type=int, default=8)
parser.add_argument('--size', type=int, default=256, help='image height = width (the reference hard-codes 256, config.py:1)')
parser.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
parser.add_argument('--compact-targets', action='store_true',
                    help='loader yields uint8 targets (sem [B,H,W], ins [B,H,W,32]); expanded on the device')
parser.add_argument('--data', default='', help='directory holding <data>/training-lmdb and <data>/validation-lmdb record '
                    'stores (the reference\'s key schema over a directory: isa_amd/records.py); default: synthetic batches')
parser.add_argument('--out', default=os.path.join(ROOT, 'models', 'CVPPP', 'run'))
opt = parser.parse_args()
assert opt.dataset in ['CVPPP', ]

from isa_amd import parallel  # noqa: E402
world, rank, local_rank = parallel.init_from_env()           # binds the GPU before anything else touches it
assert opt.batchsize % world == 0, "--batchsize is the global batch: it must divide by the number of ranks"
per_rank = opt.batchsize // world
SEED = 23                                                     # training_settings.py:53
random.seed(parallel.rank_seed(SEED, rank)); np.random.seed(parallel.rank_seed(SEED, rank))
torch.manual_seed(parallel.rank_seed(SEED, rank))             # instance order, glimpse points, dropout: per rank
model = Model(opt.dataset, 'ReSeg', 2, 32, use_instance_segmentation=True, load_model_path=opt.model, usegpu=True,
              dtype=torch.bfloat16 if opt.dtype == 'bf16' else torch.float32)
# every rank draws its own shard of each global batch (weights start identical: the model seed is not per rank)
train_loader = SyntheticLoader(opt.iters_per_epoch, per_rank, opt.size, opt.size, seed=parallel.rank_seed(SEED, rank),
                               compact=opt.compact_targets)
test_loader = SyntheticLoader(max(1, opt.iters_per_epoch // 4), per_rank, opt.size, opt.size,
                              seed=parallel.rank_seed(SEED + 7, rank), compact=opt.compact_targets)
if opt.data:                      # the reference's datasets (train.py:87-147): records -> device-side collate
    from isa_amd.records import RecordDataset, RecordLoader
    train_loader = RecordLoader(RecordDataset(os.path.join(opt.data, 'training-lmdb')), per_rank, opt.size, opt.size,
                                mode='training', seed=SEED, rank=rank, world=world)
    test_loader = RecordLoader(RecordDataset(os.path.join(opt.data, 'validation-lmdb')), per_rank, opt.size, opt.size,
                               mode='test', seed=SEED, rank=rank, world=world)
model.fit('Multi', 0.5, 1.5, 2, 1.0, 0.001, 10.0, 0.5, 25, False, 'Adadelta', True, opt.nepochs, None,
          train_loader, test_loader, opt.out, opt.debug)
if world > 1:
    torch.distributed.destroy_process_group()
