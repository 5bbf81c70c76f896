"""Training step on the HIP engine (reference: code/lib/model.py:190-284 `__minibatch`, :145-166 optimizer).

One step = zero the flat gradient buffer, forward with the backward tape recorded (semantic CE + Dice,
attention-head losses assembled on device), hand-written backward, one RCCL all-reduce of the flat
gradient buffer when world_size > 1, global-norm clip (10) + Adadelta(lr=1,
weight_decay=1e-3) fused in one kernel over the trainable slice.  No host sync inside the step.
"""
import random

import torch

from . import lib as L
from .parallel import allreduce_flat_


class Trainer:
    def __init__(self, model, world_size=1, lr=1.0, weight_decay=1e-3, clip_grad_norm=10.0, rho=0.9, eps=1e-6):
        self.model = model
        self.world = world_size
        self.lr, self.wd, self.clip, self.rho, self.eps = lr, weight_decay, clip_grad_norm, rho, eps
        st = model.store
        n = st.n_train
        self.sq = torch.zeros(n, dtype=torch.float32, device=st.device)       # Adadelta square_avg
        self.acc = torch.zeros(n, dtype=torch.float32, device=st.device)      # Adadelta acc_delta
        self.sqnorm = torch.zeros(4, dtype=torch.float32, device=st.device)
        self.last = None

    def forward_backward(self, x, sem, ins, n_objects, selected_idx=None, injected_s_t=None, capture=None):
        """Forward + backward; gradients land in model.store.grad.  Returns device scalars
        dict(sem=[ce, dice], head=[ins_cost_finite, criterion, ins_ce, ins_dice])."""
        m = self.model
        E, net, st = m.engine, m.net, m.store
        dev = st.device
        x = x.to(device=dev, dtype=torch.float32)
        sem = sem.to(dev).contiguous()
        ins = ins.to(dev).contiguous()
        st.grad[:st.n_train].zero_()
        E.begin(bn_train=m.training, record=True)
        if getattr(m, "_weights_dirty", True) and E.packer.entries:
            E.packer.pack()
        m._weights_dirty = False
        xin = net.to_nhwc(x)
        x_dec, feats = net.unet(xin)
        sem_a = net.sem_head(x_dec)
        sem_scal = net.sem_loss(sem_a, sem)
        head_scal = None
        if m.use_instance_seg:
            n_ins = [int(v) for v in n_objects.reshape(-1).tolist()]
            if selected_idx is None:
                selected_idx = []
                for k in n_ins:
                    order = list(range(k))
                    random.shuffle(order)
                    selected_idx.append(order)
            sem_map = sem.argmax(1).reshape(x.shape[0], -1).float().contiguous()
            rec = m.head.forward(x_dec, feats, sem_map, ins, n_ins, True, selected_idx, injected_s_t, capture)
            m.last_record = rec
            head_scal = rec["scal"]
        E.backward()
        self.last = dict(sem=sem_scal, head=head_scal)
        return self.last

    def apply_update(self):
        st = self.model.store
        n = st.n_train
        gscale = allreduce_flat_(st.grad, n, self.world)     # one flat RCCL all-reduce (sum)
        lib = self.model.engine.lib
        self.sqnorm.zero_()
        if self.clip > 0:
            L.check(lib.isa_sqnorm(L.ptr(st.grad), n, gscale, L.ptr(self.sqnorm), L.stream_ptr()), "isa_sqnorm")
        L.check(lib.isa_adadelta(L.ptr(st.flat), L.ptr(st.grad), L.ptr(self.sq), L.ptr(self.acc), n, self.lr, self.rho,
                                 self.eps, self.wd, L.ptr(self.sqnorm), float(self.clip), gscale, L.stream_ptr()),
                "isa_adadelta")
        self.model.mark_weights_dirty()

    def train_step(self, x, sem, ins, n_objects, selected_idx=None, injected_s_t=None):
        out = self.forward_backward(x, sem, ins, n_objects, selected_idx, injected_s_t)
        self.apply_update()
        return out
