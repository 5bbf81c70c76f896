"""Trainer / predictor mirror of the reference's `Model` (code/lib/model.py): same constructor
arguments, `fit(...)` and `predict(images)` semantics, best-validation checkpointing of the
`state_dict` (model.py:437-446), ReduceLROnPlateau on the validation ins_dice_loss (model.py:428-437),
CSV logs (model.py:366-372,458-461).  Differences, all host-side: no visdom plots, no pickles to a
hard-coded path, no debug JPEGs; compute is the HIP engine (Trainer / ReSeg), never torch ops.
"""
import os

import numpy as np
import torch

from . import parallel
from .reseg import ReSeg
from .trainer import Trainer


class Model(object):
    def __init__(self, dataset, model_name, n_classes, max_n_objects, use_instance_segmentation=False,
                 use_coords=False, load_model_path='', load_decoder_model_path='', usegpu=True, use_wae=False,
                 wae_opt=None, dtype=torch.float32):
        self.dataset, self.model_name = dataset, model_name
        self.n_classes, self.max_n_objects = n_classes, max_n_objects
        self.use_instance_segmentation = use_instance_segmentation
        self.use_coords, self.load_model_path, self.usegpu, self.use_wae = use_coords, load_model_path, usegpu, use_wae
        assert self.dataset in ['CVPPP', ]                                    # model.py:39
        assert self.model_name in ['ReSeg', 'StackedRecurrentHourglass']      # model.py:40
        assert self.model_name == 'ReSeg', "only ReSeg is live at HEAD (SURVEY.md §0-2)"
        assert usegpu, "this build has no CPU path"
        # data parallel: one process per GPU under torch.distributed.run; a single process otherwise (parallel.py)
        self.world, self.rank, self.local_rank = parallel.env_world()
        self.model = ReSeg(self.n_classes, self.use_instance_segmentation, pretrained=True,
                           use_coordinates=self.use_coords, use_wae=use_wae, usegpu=True, dtype=dtype)
        self.__load_weights()
        self.trainer = None
        self.lr_scheduler = None

    def __load_weights(self):
        if self.load_model_path != '':
            assert os.path.isfile(self.load_model_path), 'Model : {} does not exists!'.format(self.load_model_path)
            print('Loading model from {}'.format(self.load_model_path))
            # model.py:62-79: update the model's own dict, tolerate partial files.  weights_only: no code runs
            state = self.model.state_dict()
            loaded = torch.load(self.load_model_path, map_location='cpu', weights_only=True)
            state.update({k: v for k, v in loaded.items() if k in state})     # e.g. instance stems -> sem-only net
            self.model.load_state_dict(state)

    # ------------------------------------------------------------------ training
    def __define_optimizer(self, learning_rate, weight_decay, lr_drop_factor, lr_drop_patience, clip_grad_norm,
                           optimizer='Adadelta'):
        assert optimizer in ['RMSprop', 'Adam', 'Adadelta', 'SGD']            # model.py:147
        assert optimizer == 'Adadelta', "the shipped TrainingSettings use Adadelta (training_settings.py:27)"
        self.trainer = Trainer(self.model, world_size=self.world, lr=learning_rate, weight_decay=weight_decay,
                               clip_grad_norm=clip_grad_norm)
        self._plateau = dict(best=float('inf'), bad=0, factor=lr_drop_factor, patience=lr_drop_patience)

    def __plateau_step(self, val):                   # torch ReduceLROnPlateau(mode='min') semantics, rel 1e-4
        p = self._plateau
        if val < p['best'] * (1 - 1e-4):
            p['best'], p['bad'] = val, 0
        else:
            p['bad'] += 1
        if p['bad'] > p['patience']:
            self.trainer.lr *= p['factor']
            p['bad'] = 0
            if self.rank == 0:
                print('Reducing learning rate to {}'.format(self.trainer.lr))

    def __minibatch(self, batch, mode):
        assert mode in ['training', 'test'], 'Mode must be either "training" or "test"'      # model.py:193
        images, sem, ins, n_objects = batch
        if mode == 'training':
            self.model.train()
            # hipGraph replay of the step (ISA_GRAPH=0: eager launch loop)
            step = self.trainer.train_step if os.environ.get('ISA_GRAPH', '1') == '0' else self.trainer.train_step_graphed
            out = step(images, sem, ins, n_objects)
        else:
            self.model.eval()
            with torch.no_grad():
                if self.use_instance_segmentation:
                    sem_out, sem_arg, ins_cost, crit, ce, dice = self.model(False, images, sem, ins, n_objects.unsqueeze(1))
                    row = {'INS Cost': ins_cost, 'Criterion': crit, 'ins_ce_loss': ce, 'ins_dice_loss': dice}
                else:
                    self.model(False, images)
                    row = {}
                if sem.dtype == torch.uint8:              # compact targets: one-hot on the device (isa_collate_targets)
                    sem, _ = self.model.net.collate_targets(sem, ins)
                costs = self.model.sem_costs(sem)         # the reference logs CE / Dice in validation too (model.py:255-269)
                row['CE Cost'], row['Dice Cost'] = costs[0], costs[1]
            return row
        row = {'CE Cost': out['sem'][0].clone(), 'Dice Cost': out['sem'][1].clone()}
        h = out['head']
        if h is not None:                                 # semantic-only models have no instance head (model.py:244)
            row.update({'INS Cost': h[0] + float('nan'), 'Criterion': h[1].clone(), 'ins_ce_loss': h[2].clone(),
                        'ins_dice_loss': h[3].clone()})
        return row

    def fit(self, criterion_type, delta_var, delta_dist, norm, learning_rate, weight_decay, clip_grad_norm,
            lr_drop_factor, lr_drop_patience, optimize_bg, optimizer, train_cnn, n_epochs, class_weights,
            train_loader, test_loader, model_save_path, debug):
        assert criterion_type in ['CE', 'Dice', 'Multi']                       # model.py:364
        main = self.rank == 0                   # only rank 0 writes logs and checkpoints (parallel.py: policy)
        tlog = vlog = None
        if main:
            os.makedirs(model_save_path, exist_ok=True)
            tlog = open(os.path.join(model_save_path, 'training.log'), 'w')
            vlog = open(os.path.join(model_save_path, 'validation.log'), 'w')
            tlog.write('Epoch,Cost\n'); vlog.write('Epoch,Cost\n')
        self.__define_optimizer(learning_rate, weight_decay, lr_drop_factor, lr_drop_patience, clip_grad_norm, optimizer)
        best_val_cost = np.inf
        if os.environ.get('ISA_PREFETCH', '1') != '0':      # batch i+1 uploads on a side stream while step i runs
            from .data import DevicePrefetcher
            train_loader, test_loader = DevicePrefetcher(train_loader), DevicePrefetcher(test_loader)
        for epoch in range(n_epochs):
            tr = [self.__minibatch(b, 'training') for b in train_loader]
            # every rank validates the same model: average the per-rank running estimates first
            parallel.sync_buffers(self.model.store, self.model.head.baseline if self.use_instance_segmentation else None,
                                  self.world)
            if self.world > 1:
                self.model.engine.eval_bn_stale = True      # the running statistics were rewritten: cached eval constants are old
            va = [self.__minibatch(b, 'test') for b in test_loader]
            # an empty loader (a validation set smaller than one batch per rank under a dropping sampler) must not take the
            # epoch down on every rank: its mean is NaN and the plateau scheduler sees the training cost instead
            mean = lambda rows, k: float(torch.stack([r[k].float() for r in rows]).mean()) if rows else float("nan")
            key = 'ins_dice_loss' if self.use_instance_segmentation else 'Dice Cost'
            # rank-averaged costs: the plateau scheduler must take the same decision on every rank
            train_cost = parallel.mean_over_ranks(mean(tr, key), self.world)
            val_cost = parallel.mean_over_ranks(mean(va, key), self.world)
            if val_cost != val_cost:
                val_cost = train_cost
            if main:
                print('Epoch : [{}/{}]  train {} {:.5f} | val {:.5f}'.format(epoch, n_epochs, key, train_cost, val_cost))
            self.__plateau_step(val_cost)
            if val_cost <= best_val_cost:                                       # model.py:439-446
                best_val_cost = val_cost
                if main:
                    torch.save(self.model.state_dict(), os.path.join(
                        model_save_path, 'model_{}_{}_{}.pth'.format(epoch, val_cost, self.trainer.lr)))
            if main:
                tlog.write('{},{}\n'.format(epoch, train_cost)); vlog.write('{},{}\n'.format(epoch, val_cost))
                tlog.flush(); vlog.flush()
        if main:
            tlog.close(); vlog.close()

    # ------------------------------------------------------------------ inference (model.py:466-499)
    def predict(self, images):
        assert len(images.size()) == 4  # b, c, h, w
        self.model.eval()
        m = self.model
        if m.use_instance_seg:
            # the shipped ModelSettings hit an UnboundLocalError here (reseg.py:126); the only working
            # inference mode is the semantic one — same contract, clearer message
            raise RuntimeError("predict() needs a model built with use_instance_segmentation=False")
        m(False, images.contiguous())
        return m.net.softmax_nchw(m._last_sem).cpu()                          # softmax over classes (model.py:486)
