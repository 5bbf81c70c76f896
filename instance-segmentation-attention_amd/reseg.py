"""Drop-in `ReSeg` (reference: code/lib/archs/reseg.py:52-130).

Same constructor signature, same `forward(training, *_input)` contract, same `state_dict()` keys,
shapes and order (891 tensors), `.base` attribute, `.parameters()`, `.train()/.eval()`, `.cuda()`.
All compute runs in the HIP library behind include/isa_kernels.h; this class only owns the
parameters (views into one flat fp32 buffer, so DDP needs one all-reduce and the optimizer one
kernel) and sequences launches.  Missing library or missing GPU => hard error, never a fallback.
"""
import torch
import torch.nn as nn

from . import lib as L
from .engine import Engine, ParamStore
from .network import Network
from .instance_head import InstanceHead
from .schema import state_dict_schema


class _Node(nn.Module):
    """Anonymous container reproducing the reference's module tree for state_dict naming."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container node: compute lives in the HIP engine")


class ReSeg(nn.Module):
    def __init__(self, n_classes, use_instance_seg=True, pretrained=True, use_coordinates=False,
                 use_wae=True, usegpu=True, training=True, dtype=torch.float32, device=None):
        super().__init__()
        assert n_classes == 2, "the reference head is built for 2 classes (data_settings.py:19)"
        if not torch.cuda.is_available():
            raise RuntimeError("ReSeg (MI355X build) needs a GPU: there is no CPU fallback")
        L.lib()                                   # fail loudly if the HIP library is missing
        self.backbone = "Unet"
        self.n_classes = n_classes
        self.use_instance_seg = use_instance_seg
        self.use_wae = use_wae
        self.compute_dtype = dtype
        dev = torch.device(device or "cuda")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.store = ParamStore(state_dict_schema(use_instance_seg), dev)
        self._build_tree()
        self.engine = Engine(self.store, dtype, dev)
        self.net = Network(self.engine, use_instance_seg)
        self.head = InstanceHead(self.net)
        self.reset_parameters()
        self.train(training)

    # ------------------------------------------------------------------ module tree
    def _build_tree(self):
        st = self.store
        self._nbt = {}
        for name in st.names:
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Node())
                mod = mod._modules[p]
            leaf = parts[-1]
            if leaf == "num_batches_tracked":
                buf = torch.zeros((), dtype=torch.long)
                mod.register_buffer(leaf, buf)
                self._nbt[name] = (mod, leaf)
            elif leaf in ("running_mean", "running_var"):
                mod.register_buffer(leaf, st.view(name))
            else:
                p_ = nn.Parameter(st.view(name))
                p_.grad = st.gview(name)
                mod.register_parameter(leaf, p_)

    def reset_parameters(self, seed=None):
        """Fresh weights in the spirit of torch defaults (kaiming-uniform convs, BN gamma=1/beta=0,
        maskBN gamma~U(0,1) as utils.py:561).  Host-side plumbing on the flat buffer views."""
        g = torch.Generator(device="cpu")
        g.manual_seed(0 if seed is None else int(seed))
        st = self.store
        bn_prefixes = {n[:-len(".running_mean")] for n in st.names if n.endswith(".running_mean")}
        for name in st.names:
            if name in st.int_buffers:
                continue
            v, shape = st.view(name), st.shapes[name]
            prefix, leaf = name.rsplit(".", 1)
            if leaf == "running_mean":
                v.zero_()
            elif leaf == "running_var":
                v.fill_(1.0)
            elif prefix in bn_prefixes and leaf == "weight":
                v.copy_(torch.rand(shape, generator=g)) if prefix == "decoder.attend.bn" else v.fill_(1.0)
            elif prefix in bn_prefixes and leaf == "bias":
                v.zero_()
            else:
                fan_in = 1
                for d in (shape[1:] if len(shape) > 1 else shape):
                    fan_in *= d
                if leaf == "bias":                       # conv/linear bias: fan-in of its weight
                    wshape = st.shapes.get(prefix + ".weight", shape)
                    fan_in = 1
                    for d in wshape[1:]:
                        fan_in *= d
                bound = (1.0 / max(fan_in, 1)) ** 0.5
                v.copy_((torch.rand(shape, generator=g) * 2 - 1) * bound)

    # ------------------------------------------------------------------ nn.Module plumbing
    def _apply(self, fn, recurse=True):
        # parameters are views of one flat device buffer; moving/casting them individually would
        # break that.  `.cuda()` / `.to(same device)` are accepted as no-ops like the reference's
        # `model.cuda()` call site (model.py:52).
        probe = fn(torch.empty(0, device=self.store.device))
        if probe.device != self.store.device or probe.dtype != torch.float32:
            raise RuntimeError("ReSeg parameters live in one flat fp32 GPU buffer; cannot move/cast")
        return self

    def state_dict(self, *args, **kwargs):
        for name, (mod, leaf) in self._nbt.items():
            mod._buffers[leaf].fill_(self.store.int_buffers[name])
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        for name, (mod, leaf) in self._nbt.items():
            self.store.int_buffers[name] = int(mod._buffers[leaf])
        self.engine.packer.table = None if self.engine.packer.entries else self.engine.packer.table
        self._weights_dirty = True
        self.engine.eval_bn_stale = True
        return out

    def mark_weights_dirty(self):
        """Call after an optimizer step changed the flat parameter buffer (repack on next forward)."""
        self._weights_dirty = True
        self.engine.eval_bn_stale = True

    # ------------------------------------------------------------------ hipGraph-replayed GT-free inference
    def infer_graphed(self, x):
        """(sem_out, sem_argmax) = forward(False, x) replayed from a hipGraph (pred_list-style batched inference:
        ~70 launches per batch are launch-bound at bs=16).  One graph per input shape/dtype; the first call of a
        shape runs eagerly, the second captures.  Returned tensors are the graph's static outputs: consume or
        copy them before the next call."""
        assert not self.training, "infer_graphed is for eval mode (running BatchNorm statistics)"
        dev = self.store.device
        key = (tuple(x.shape), x.dtype)
        graphs = self.__dict__.setdefault("_infer_graphs", {})
        slot = graphs.get(key)
        if slot is None:
            graphs[key] = dict(state="warm")
            with torch.no_grad():
                return self.forward(False, x, _arena_key=("infer_graph",) + key)
        if slot["state"] == "eager":
            with torch.no_grad():
                return self.forward(False, x)
        if slot["state"] == "warm":
            slot["x"] = torch.empty(tuple(x.shape), dtype=x.dtype if x.dtype == torch.uint8 else torch.float32, device=dev)
            slot["x"].copy_(x, non_blocking=True)
            if getattr(self, "_weights_dirty", True) and self.engine.packer.entries:
                self.engine.packer.pack()                  # weights are constant across replays: pack outside
                self._weights_dirty = False
            if self.engine.eval_bn_stale:
                self.engine.refresh_eval_bn()              # ... and so are the eval-mode BN constants
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"), torch.no_grad():
                    out = self.forward(False, slot["x"], _arena_key=("infer_graph",) + key)
                self.engine.freeze_arena()
            except Exception as e:
                import sys
                print("[isa_amd] hipGraph capture failed (%s: %s); inference runs eagerly" % (type(e).__name__, e),
                      file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                slot["state"] = "eager"
                with torch.no_grad():
                    return self.forward(False, x)
            slot.update(state="ready", graph=g, out=out)
        else:
            if getattr(self, "_weights_dirty", False):     # parameters changed since capture: repack, then replay
                self.engine.packer.pack()
                self._weights_dirty = False
            if self.engine.eval_bn_stale:                  # same buffers the graph reads, recomputed in place
                self.engine.refresh_eval_bn()
            slot["x"].copy_(x, non_blocking=True)
        slot["graph"].replay()
        return slot["out"]

    def sem_costs(self, sem_seg_target):
        """Semantic CE + Dice(time=1) of the LAST forward's logits against a one-hot int64 target [B,2,H,W]
        (validation branch of model.py:244-270).  Returns a device tensor [ce, dice]; must be called before the
        next forward (the logits live in that step's arena)."""
        sem = getattr(self, "_last_sem", None)
        assert sem is not None, "sem_costs() follows a forward()"
        t = sem_seg_target.to(self.store.device).contiguous()
        assert t.dtype == torch.int64 and tuple(t.shape) == (sem.n, 2, sem.h, sem.w)
        was = self.engine.record
        self.engine.record = False
        try:
            return self.net.sem_loss(sem, t).clone()
        finally:
            self.engine.record = was

    # ------------------------------------------------------------------ forward
    def forward(self, training, *_input, selected_idx=None, injected_s_t=None, capture=None, _arena_key=None):
        """reseg.py:106-130.  (x) -> (sem_out, sem_argmax);  (x, sem_onehot[B,2,H,W] i64,
        ins[B,32,H,W] i64, N[B,1]) [or the compact uint8 pair sem[B,H,W], ins[B,H,W,32]: expanded on device] -> (sem_out, sem_argmax, ins_cost, criterion, ins_ce_loss,
        ins_dice_loss).  BatchNorm mode follows .train()/.eval() like the reference modules; the
        `training` flag drives sampling, F.dropout2d and the loss branch (attenet2.py:377-399).
        `selected_idx` / `injected_s_t` inject the reference's host RNG choices (random.shuffle,
        torch.multinomial) for parity runs; defaults: random order / argmax-or-sampled on device."""
        E, net = self.engine, self.net
        has_gt = len(_input) == 4
        if has_gt:
            x, sem_seg_target, ins_seg_target, N = _input
        else:
            x = _input[0]
        raw_rgb = x.dtype == torch.uint8           # [B,H,W,3] uint8: ImageEx runs on device (net.image_ex)
        if raw_rgb:
            assert x.dim() == 4 and x.shape[3] == 3, "uint8 input must be RGB [B,H,W,3]"
            assert x.shape[1] % 16 == 0 and x.shape[2] % 16 == 0
        else:
            assert x.dim() == 4 and x.shape[1] == 21, "expects [B,21,H,W] (ImageEx tensor, utils.py:109)"
            assert x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0
        dev = self.store.device
        E.begin(bn_train=self.training, record=False,
                key=_arena_key or ("forward", has_gt, tuple(x.shape), x.dtype, bool(training)))
        if getattr(self, "_weights_dirty", True) and E.packer.entries:
            E.packer.pack()
        self._weights_dirty = False
        if has_gt and ins_seg_target.dtype == torch.uint8:      # compact targets: net.collate_targets (dataset.py:349-379)
            sem_seg_target, ins_seg_target = net.collate_targets(sem_seg_target, ins_seg_target)
        xin = net.input_view(x)
        x_dec, feats = net.unet(xin)
        if capture is not None:                    # UNet.forward's six maps (unet_model.py:36), for parity tests
            capture.update({"unet.x_dec": x_dec, "unet.x1": feats[0], "unet.x2": feats[1], "unet.x3": feats[2],
                            "unet.x4": feats[3], "unet.x5": feats[4]})
        sem = net.sem_head(x_dec)
        self._last_sem = sem                       # logits view in the step's arena (sem_costs)
        sem_out = net.to_nchw(sem)
        if has_gt:
            # (the map lives in the step's arena: the caller gets a copy)
            sem_argmax = net.onehot_map(sem_seg_target.to(dev).contiguous()).view(x.shape[0], 1, sem.h, sem.w).clone()
        else:
            sem_argmax = net.to_nchw(net.argmax_map(sem))
        if not self.use_instance_seg:
            return sem_out, sem_argmax
        if not has_gt:
            # the reference raises UnboundLocalError here (reseg.py:126): the instance head needs
            # ground-truth masks and has no GT-free mode
            raise RuntimeError("instance head needs (x, sem, ins, N); build ReSeg(.., use_instance_seg=False) "
                               "for GT-free inference")
        n_ins = [int(v) for v in N.reshape(-1).tolist()]
        if selected_idx is None:
            import random
            selected_idx = []
            for k in n_ins:                       # attenet2.py:349-355
                order = list(range(k))
                random.shuffle(order)
                selected_idx.append(order)
        sem_map = sem_argmax.reshape(x.shape[0], -1).contiguous()
        ins_dev = ins_seg_target.to(dev).contiguous()
        rec = self.head.forward(x_dec, feats, sem_map, ins_dev, n_ins, bool(training), selected_idx,
                                injected_s_t, capture)
        self.last_record = rec
        scal = rec["scal"].clone()
        ins_cost = scal[0] + float("nan") if training else scal[0]     # attenet2.py:77: H is NaN in training
        return (sem_out, sem_argmax, ins_cost, scal[1], scal[2], scal[3])
