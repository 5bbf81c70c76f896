"""Training step on the HIP engine (reference: code/lib/model.py:190-284 `__minibatch`, :145-166 optimizer).

One step = zero the flat gradient buffer, forward with the backward tape recorded (semantic CE + Dice,
attention-head losses assembled on device), hand-written backward, one RCCL all-reduce of the flat
gradient buffer when world_size > 1, global-norm clip (10) + Adadelta(lr=1,
weight_decay=1e-3) fused in one kernel over the trainable slice.  No host sync inside the step.
"""
import random

import torch

from . import lib as L
from .parallel import exchange_and_update


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
        # the step size lives in a device scalar the optimizer kernel reads at run time: a captured hipGraph follows
        # ReduceLROnPlateau (model.py:164,437) without re-capture
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=st.device)
        self._lr_on_dev = float(lr)
        if world_size > 1:          # identical start on every rank (checkpoints are loaded per process)
            import torch.distributed as dist
            dist.broadcast(st.flat, src=0)
            model.mark_weights_dirty()
        self.last = None
        self._graphs = {}

    def forward_backward(self, x, sem, ins, n_objects, selected_idx=None, injected_s_t=None, capture=None,
                         idx_dev=None, arena_key=None, backward=True):
        """Forward + backward; gradients land in model.store.grad.  Returns device scalars
        dict(sem=[ce, dice], head=[ins_cost_finite, criterion, ins_ce, ins_dice]).  backward=False: the training-mode
        forward alone (batch statistics, sampling, Dropout2d, losses; no tape) - the forward-only benchmark line."""
        m = self.model
        E, net, st = m.engine, m.net, m.store
        dev = st.device
        x = x.to(device=dev) if x.dtype == torch.uint8 else x.to(device=dev, dtype=torch.float32)
        sem = sem.to(dev).contiguous()
        ins = ins.to(dev).contiguous()
        if backward:
            st.grad[:st.n_train].zero_()
        E.begin(bn_train=m.training, record=backward,
                key=arena_key or ("train", tuple(x.shape), x.dtype, tuple(ins.shape), ins.dtype,
                                  injected_s_t is not None))
        if getattr(m, "_weights_dirty", True) and E.packer.entries:
            E.packer.pack()
        m._weights_dirty = False
        if ins.dtype == torch.uint8:         # compact targets (uint8 planes [B,H,W,K] + uint8 map [B,H,W]): expand on device
            sem, ins = net.collate_targets(sem, ins)
        xin = net.input_view(x)
        x_dec, feats = net.unet(xin)
        sem_a = net.sem_head(x_dec)
        sem_scal = net.sem_loss(sem_a, sem)
        head_scal = None
        if m.use_instance_seg:
            n_ins = n_objects if isinstance(n_objects, list) else [int(v) for v in n_objects.reshape(-1).tolist()]
            if selected_idx is None and idx_dev is None:
                selected_idx = []
                for k in n_ins:
                    order = list(range(k))
                    random.shuffle(order)
                    selected_idx.append(order)
            sem_map = net.onehot_map(sem)                    # GT.argmax(1) (reseg.py:118) as the fp32 map the head reads
            rec = m.head.forward(x_dec, feats, sem_map, ins, n_ins, True, selected_idx, injected_s_t, capture,
                                 idx_dev=idx_dev)
            m.last_record = rec
            head_scal = rec["scal"]
        if backward:
            E.backward()
        self.last = dict(sem=sem_scal, head=head_scal)
        return self.last

    def sync_lr(self):
        """Host lr -> device scalar (outside any graph; a no-op unless the scheduler changed it)."""
        if self._lr_on_dev != float(self.lr):
            self.lr_dev.fill_(float(self.lr))
            self._lr_on_dev = float(self.lr)

    def apply_update(self):
        """all-reduce (world > 1) -> norm of the averaged gradient -> clip + Adadelta: parallel.exchange_and_update
        owns the ordering (the CPU test drives the same function with stand-in kernels)."""
        st = self.model.store
        lib = self.model.engine.lib
        self.sqnorm.zero_()

        def sqnorm_fn(grad, n, gscale):
            L.check(lib.isa_sqnorm(L.ptr(grad), n, gscale, L.ptr(self.sqnorm), L.stream_ptr()), "isa_sqnorm")

        def update_fn(grad, n, gscale):
            L.check(lib.isa_adadelta(L.ptr(st.flat), L.ptr(grad), L.ptr(self.sq), L.ptr(self.acc), n, self.lr, self.rho,
                                     self.eps, self.wd, L.ptr(self.sqnorm), float(self.clip), gscale, L.ptr(self.lr_dev),
                                     L.stream_ptr()), "isa_adadelta")

        exchange_and_update(st.grad, st.n_train, self.world, sqnorm_fn, update_fn, self.clip)
        self.model.mark_weights_dirty()

    def train_step(self, x, sem, ins, n_objects, selected_idx=None, injected_s_t=None, arena_key=None):
        self.sync_lr()
        out = self.forward_backward(x, sem, ins, n_objects, selected_idx, injected_s_t, arena_key=arena_key)
        self.apply_update()
        return out

    # ------------------------------------------------------------------ hipGraph-captured step
    def train_step_graphed(self, x, sem, ins, n_objects, selected_idx=None, injected_s_t=None, forward_only=False):
        """Same step, replayed from a hipGraph: the ~880 launches of forward+backward (+ the fused update when
        world_size == 1) are recorded once per (shapes, iteration count) and replayed, so the GPU never waits
        for the Python launch loop.  Inputs are copied into static device buffers; the per-step host decisions
        (instance order) travel through a small staged index tensor; dropout masks come from torch's graph-safe
        generator.  The first call of a configuration runs eagerly (allocations settle), the second captures.
        With world_size > 1 the RCCL all-reduce and the update stay outside the graph.
        forward_only: capture and replay the training-mode forward alone (no gradients, no update)."""
        m = self.model
        st = m.store
        dev = st.device
        n_ins = [int(v) for v in n_objects.reshape(-1).tolist()]
        from .instance_head import MAX_ITER
        max_iter = min(MAX_ITER, min(n_ins)) if m.use_instance_seg else 0
        if m.use_instance_seg and selected_idx is None:
            selected_idx = []
            for k in n_ins:
                order = list(range(k))
                random.shuffle(order)
                selected_idx.append(order)
        key = (tuple(x.shape), tuple(sem.shape), tuple(ins.shape), max_iter, bool(m.training), self.world,
               m.engine.dtype, injected_s_t is not None, x.dtype == torch.uint8, ins.dtype == torch.uint8, bool(forward_only))
        slot = self._graphs.get(key)
        akey = ("train_graph",) + key            # the captured configuration owns its arena (frozen after capture)
        self.sync_lr()
        def eager(arena_key=None):
            if forward_only:
                return self.forward_backward(x, sem, ins, n_objects, selected_idx, injected_s_t, arena_key=arena_key,
                                             backward=False)
            return self.train_step(x, sem, ins, n_objects, selected_idx=selected_idx, injected_s_t=injected_s_t,
                                   arena_key=arena_key)
        if slot is None:                         # first sight: eager step, remember the configuration
            self._graphs[key] = dict(state="warm")
            return eager(akey)
        if slot["state"] == "eager":
            return eager()
        if slot["state"] == "warm":
            slot["x"] = torch.empty(tuple(x.shape), dtype=torch.uint8 if x.dtype == torch.uint8 else torch.float32,
                                    device=dev)
            slot["sem"] = torch.empty(tuple(sem.shape), dtype=sem.dtype, device=dev)
            slot["ins"] = torch.empty(tuple(ins.shape), dtype=ins.dtype, device=dev)
            slot["idx"] = torch.zeros((max(max_iter, 1), x.shape[0]), dtype=torch.int32, device=dev)
            slot["idx_pin"] = torch.zeros((max(max_iter, 1), x.shape[0]), dtype=torch.int32).pin_memory()
            slot["inj"] = None if injected_s_t is None else [torch.zeros_like(t) for t in injected_s_t[:max_iter]]
            self._stage(slot, x, sem, ins, selected_idx, max_iter, injected_s_t)
            before = dict(st.int_buffers)
            g = torch.cuda.CUDAGraph()
            try:
                # thread_local: the RCCL watchdog thread may query events while this thread captures
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = self.forward_backward(slot["x"], slot["sem"], slot["ins"], n_ins, idx_dev=slot["idx"],
                                                injected_s_t=slot["inj"], arena_key=akey, backward=not forward_only)
                    if self.world == 1 and not forward_only:
                        self.apply_update()
                m.engine.freeze_arena()
            except Exception as e:                 # capture refused (driver / collective state): stay eager, loudly
                import sys
                print("[isa_amd] hipGraph capture failed (%s: %s); this configuration runs eagerly" %
                      (type(e).__name__, e), file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                st.int_buffers.update(before)
                slot["state"] = "eager"
                return eager()
            # capture only records: undo its host-side counters, replay() below performs the step
            slot["bumps"] = {k: v - before[k] for k, v in st.int_buffers.items() if v != before[k]}
            st.int_buffers.update(before)
            slot.update(state="ready", graph=g, out=out)
        else:
            self._stage(slot, x, sem, ins, selected_idx, max_iter, injected_s_t)
        slot["graph"].replay()
        for k, v in slot["bumps"].items():
            st.int_buffers[k] += v
        if not forward_only:
            if self.world > 1:
                self.apply_update()
            m.mark_weights_dirty()
        self.last = slot["out"]
        return self.last

    def static_inputs(self):
        """(x, sem, ins) input buffers of the captured graphs, most recent configuration first: a data pipeline can
        write the next batch straight into them (H2D or a device-side producer) and pass these same tensors to
        train_step_graphed, which then replays without any staging copy."""
        return [(s["x"], s["sem"], s["ins"]) for s in reversed(list(self._graphs.values())) if s.get("state") == "ready"]

    def _stage(self, slot, x, sem, ins, selected_idx, max_iter, injected_s_t=None):
        if slot.get("inj") is not None:          # parity runs: glimpse points fixed from outside
            for dst, src in zip(slot["inj"], injected_s_t):
                dst.copy_(src, non_blocking=True)
        # a caller that fills the graph's own input buffers (static_inputs) skips the device-to-device copies
        if x is not slot["x"]:
            slot["x"].copy_(x, non_blocking=True)
        if sem is not slot["sem"]:
            slot["sem"].copy_(sem, non_blocking=True)
        if ins is not slot["ins"]:
            slot["ins"].copy_(ins, non_blocking=True)
        if max_iter > 0:
            n = slot["x"].shape[0]
            slot["idx_pin"].copy_(self.model.head.order_tensor(selected_idx, max_iter, n))
            slot["idx"].copy_(slot["idx_pin"], non_blocking=True)
