"""Host-side engine: NHWC activation views, static arena, parameter store, op wrappers + tape.

Python orchestrates, the C-ABI library computes.  Everything an op needs lives in buffers that are
allocated once per (shape, mode) and reused every step, so a whole forward/backward/update step
is a fixed sequence of asynchronous launches on one HIP stream and can be captured in a hipGraph
(see trainer.py).  torch is used for: device memory, streams, events, RCCL.  Not for math.

Gradient convention: `Act.grad` is the gradient w.r.t. the TRUE value of the activation (after its
lazy prologue).  Multi-consumer tensors accumulate: the first backward writer overwrites, later
writers add (tracked per buffer and channel range) — no memsets of activation-sized buffers.
"""
import contextlib
import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L


STAT_R = 8      # ISA_STAT_REPLICAS: statistics buffers are [8][2C]


def rup(x, m):
    return (x + m - 1) // m * m


class PendingFin:
    """A train-mode BatchNorm finalize that has not been launched (isa_bn_fin, include/isa_kernels.h): the first consumer
    of the lazy tensor on each stream either runs it inside its own kernel (`inline`: isa_conv_gemm, isa_dwconv3x3,
    isa_affine_act_res) or launches isa_bn_finalize ahead of itself (`resolve`: everything else that reads the scale /
    shift arrays).  Exactly one of them carries the running-statistics update.  Consumers on a stream that has already
    seen one are ordered behind it and read the arrays.  Every consumer recomputes identical values from the same sums,
    so consumers on different streams need no edge between them."""
    __slots__ = ("eng", "stats", "count", "pre", "c", "scale", "shift", "mean", "invstd", "groups", "rep", "running_taken",
                 "done", "keep")

    def __init__(self, eng, stats, count, pre, c, scale, shift, mean, invstd, groups, rep, running_taken):
        self.eng, self.stats, self.count, self.pre, self.c = eng, stats, count, pre, c
        self.scale, self.shift, self.mean, self.invstd, self.groups, self.rep = scale, shift, mean, invstd, groups, rep
        self.running_taken = running_taken
        self.done, self.keep = set(), []

    def _running(self):
        if self.running_taken:
            return None, None
        self.running_taken = True
        P = self.eng.params
        return P.ptr(self.pre + ".running_mean"), P.ptr(self.pre + ".running_var")

    def settled(self):
        E = self.eng
        return E._deferring or E.cur in self.done

    def resolve(self):
        if self.settled():
            return
        E, P = self.eng, self.eng.params
        rm, rv = self._running()
        L.check(E.lib.isa_bn_finalize(L.ptr(self.stats), self.count, P.ptr(self.pre + ".weight"), P.ptr(self.pre + ".bias"),
                                      rm, rv, E.BN_MOMENTUM, E.BN_EPS, L.ptr(self.scale), L.ptr(self.shift),
                                      L.ptr(self.mean), L.ptr(self.invstd), self.c, self.groups, self.rep, E.st()),
                "isa_bn_finalize")
        self.done.add(E.cur)

    def inline(self, pro):
        """isa_pro* carrying this finalize for a consumer with the in-kernel form."""
        E, P = self.eng, self.eng.params
        rm, rv = self._running()
        a = lambda v: v.value if v is not None else None
        d = L.IsaBnFin(self.stats.data_ptr(), a(P.ptr(self.pre + ".weight")), a(P.ptr(self.pre + ".bias")), a(rm), a(rv),
                       self.scale.data_ptr(), self.shift.data_ptr(), self.mean.data_ptr(), self.invstd.data_ptr(),
                       self.count, E.BN_MOMENTUM, E.BN_EPS, self.rep)
        pc = L.IsaPro(L.ptr(pro.scale), L.ptr(pro.shift), L.ptr(pro.bscale), pro.act, C.pointer(d))
        self.keep.append((d, pc))
        self.done.add(E.cur)
        return C.byref(pc)


class Pro:
    """Lazy prologue: value = act(scale*raw + shift) * bscale.  `fin`: the BatchNorm finalize that fills scale / shift is
    still pending (PendingFin) - c() launches it if this stream has not seen it, c_fin() hands it to the consumer."""
    __slots__ = ("scale", "shift", "bscale", "act", "_c", "fin")

    def __init__(self, scale=None, shift=None, act=L.ACT_NONE, bscale=None, fin=None):
        self.scale, self.shift, self.act, self.bscale, self.fin = scale, shift, act, bscale, fin
        self._c = L.IsaPro(L.ptr(scale), L.ptr(shift), L.ptr(bscale), act, None)

    def c(self):
        if self.fin is not None:
            self.fin.resolve()
        return C.byref(self._c)

    def c_fin(self):
        if self.fin is None or self.fin.settled():
            return C.byref(self._c)
        return self.fin.inline(self)


class Act:
    """A channel slice [c0, c0+c) of an NHWC buffer [n,h,w,ld].  groups > 1: the n images are `groups` consecutive
    statistic groups (decoder iterations batched into one pass) with separate BatchNorm statistics - every per-channel
    array that travels with the tensor (prologue scale/shift, statistics, BN-backward constants) is [groups][c]."""
    __slots__ = ("buf", "c0", "c", "pro", "_c", "needs_grad", "bn", "groups")

    def __init__(self, buf: torch.Tensor, c0=0, c=None, pro: Optional[Pro] = None, needs_grad=True, groups=1):
        assert buf.dim() == 4 and buf.is_contiguous()
        self.buf, self.c0 = buf, c0
        self.c = buf.shape[3] - c0 if c is None else c
        assert 0 <= c0 and c0 + self.c <= buf.shape[3]
        self.pro = pro
        self.needs_grad = needs_grad
        self.bn = None
        self.groups = groups
        n, h, w, ld = buf.shape
        assert n % groups == 0
        self._c = L.IsaTensor(buf.data_ptr() + c0 * buf.element_size(), n, h, w, self.c, ld,
                              L.dtype_code(buf.dtype), groups)

    n = property(lambda s: s.buf.shape[0])
    h = property(lambda s: s.buf.shape[1])
    w = property(lambda s: s.buf.shape[2])
    ld = property(lambda s: s.buf.shape[3])

    def d(self):
        return C.byref(self._c)

    def p(self):
        return self.pro.c() if self.pro is not None else None

    def p_fin(self):
        """p() for the entry points that can run a pending BatchNorm finalize themselves."""
        return self.pro.c_fin() if self.pro is not None else None

    def slice(self, c0, c, pro=None):
        return Act(self.buf, self.c0 + c0, c, pro, self.needs_grad, self.groups)

    def with_pro(self, pro):
        a = Act(self.buf, self.c0, self.c, pro, self.needs_grad, self.groups)
        return a

    def images(self, b0, nb):
        """Images [b0, b0+nb) as a view of their own (one statistic group of a batched tensor; tests / captures)."""
        return Act(self.buf[b0:b0 + nb], self.c0, self.c, None, self.needs_grad, 1)

    def nchw(self) -> torch.Tensor:
        """Materialised copy as NCHW float32 (tests / outputs only; applies no prologue)."""
        return self.buf[..., self.c0:self.c0 + self.c].permute(0, 3, 1, 2).float().contiguous()


# ISA_ARENA_POISON=1 (debug / tests): every arena buffer that is handed out un-zeroed, and the slab arena, are filled with
# 0xFF bytes each step, so any read of memory the step itself did not write turns into NaN instead of stale values.
POISON = os.environ.get("ISA_ARENA_POISON", "0") == "1"


class Arena:
    """Bump allocator whose allocations are replayed in the same order every step.  One arena per
    configuration (Engine.begin's key): a captured hipGraph holds raw pointers into its arena, so the
    arena is frozen at capture time - a diverging allocation sequence then raises instead of freeing
    memory the graph still writes to.  The arena also owns the step's zero-initialised fp32 scratch."""

    def __init__(self, device):
        self.device = device
        self.slots: List[torch.Tensor] = []
        self.cursor = 0
        self.frozen = False
        self.stats = None              # flat fp32 scratch zeroed once per step
        self.stats_cursor = 0
        self.keep: List[torch.Tensor] = []   # outgrown scratch buffers a captured graph may still reference

    def reset(self):
        self.cursor = 0
        self.stats_cursor = 0
        if self.stats is not None:
            self.stats.zero_()

    def alloc(self, shape, dtype, zero=False):
        shape = tuple(int(s) for s in shape)
        if self.cursor < len(self.slots):
            t = self.slots[self.cursor]
            if tuple(t.shape) != shape or t.dtype != dtype:
                if self.frozen:
                    raise RuntimeError("arena of a captured hipGraph: allocation %d changed from %s %s to %s %s - "
                                       "this configuration must keep its launch sequence (use another arena key)"
                                       % (self.cursor, tuple(t.shape), t.dtype, shape, dtype))
                # shape changed (new batch/size): drop the tail and start recording again
                del self.slots[self.cursor:]
                t = None
        else:
            if self.frozen:
                raise RuntimeError("arena of a captured hipGraph: the step allocates more buffers than were captured")
            t = None
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.slots.append(t)
        self.cursor += 1
        if zero:
            t.zero_()
        elif POISON:
            t.view(torch.uint8).fill_(0xFF)        # NaN in every float type: a kernel that reads what no kernel wrote shows up
        return t

    def scratch(self, numel) -> torch.Tensor:
        """fp32 scratch that is zero at step start (BN sums, reductions)."""
        numel = rup(numel, 4)
        if self.stats is None or self.stats_cursor + numel > self.stats.numel():
            if self.frozen:
                raise RuntimeError("arena of a captured hipGraph: the step needs more zeroed scratch than was captured")
            # grow: only legal while recording a new configuration (first step)
            new = torch.zeros(max(2 * (self.stats_cursor + numel), 1 << 16), dtype=torch.float32,
                              device=self.device)
            if self.stats is not None:
                new[:self.stats.numel()].copy_(self.stats)
                self.keep.append(self.stats)
            self.stats = new
        t = self.stats[self.stats_cursor:self.stats_cursor + numel]
        self.stats_cursor += numel
        return t

    def bytes(self):
        return sum(t.numel() * t.element_size() for t in self.slots)


class ParamStore:
    """All parameters and float buffers in ONE flat fp32 tensor (reference state_dict layout per
    tensor) and their gradients in another: one RCCL all-reduce, one optimizer kernel."""

    NEVER_GRAD_PREFIXES = ("decoder.pred.", "decoder.attend.l2.", "decoder.embedding.")   # 9 tensors

    def __init__(self, schema: List[Tuple[str, tuple]], device):
        self.device = device
        self.names = [n for n, _ in schema]
        self.shapes = {n: tuple(s) for n, s in schema}
        self.offsets: Dict[str, int] = {}
        self.int_buffers: Dict[str, int] = {}

        def klass(n):
            if n.endswith("num_batches_tracked"):
                return 3
            if n.endswith("running_mean") or n.endswith("running_var"):
                return 2
            return 1 if n.startswith(self.NEVER_GRAD_PREFIXES) else 0

        off = 0
        self.n_train = 0
        for k in (0, 1, 2):          # flat layout: [trainable | never-grad params | float buffers]
            for n, s in schema:
                if klass(n) != k:
                    continue
                self.offsets[n] = off
                numel = 1
                for d in s:
                    numel *= d
                off += rup(numel, 4)          # keep every tensor 16-byte aligned
            if k == 0:
                self.n_train = off
            if k == 1:
                self.buffer_start = off       # float buffers (BN running statistics) start here
        for n, _ in schema:
            if klass(n) == 3:
                self.int_buffers[n] = 0
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)

    def numel(self, name):
        n = 1
        for d in self.shapes[name]:
            n *= d
        return n

    def view(self, name):
        o = self.offsets[name]
        return self.flat[o:o + self.numel(name)].view(self.shapes[name])

    def gview(self, name):
        o = self.offsets[name]
        return self.grad[o:o + self.numel(name)].view(self.shapes[name])

    def ptr(self, name):
        return C.c_void_p(self.flat.data_ptr() + 4 * self.offsets[name])

    def gptr(self, name):
        return C.c_void_p(self.grad.data_ptr() + 4 * self.offsets[name])

    def load_state_dict(self, sd):
        for n in self.names:
            if n in self.int_buffers:
                if n in sd:
                    self.int_buffers[n] = int(sd[n])
                continue
            if n in sd:
                self.view(n).copy_(sd[n].to(torch.float32))

    def state_dict(self):
        out = {}
        for n in self.names:
            if n in self.int_buffers:
                out[n] = torch.tensor(self.int_buffers[n], dtype=torch.long)
            else:
                out[n] = self.view(n).detach().clone()
        return out


class Packer:
    """Table of weight repack jobs (isa_pack_weights).  Keys are (param name, variant)."""

    def __init__(self, params: ParamStore, dtype):
        self.params, self.dtype = params, dtype
        self.entries: List[dict] = []
        self.index: Dict[tuple, dict] = {}
        self.kmaps: List[int] = []
        self.total = 0
        self.buf = None
        self.table = None
        self.kmap_dev = None
        self._keep = []          # outgrown buffers/tables: a captured hipGraph's launches may still reference them

    def add(self, name, variant, kind, n, k, taps, kp, rows, kmap=None):
        key = (name, variant)
        if key in self.index:
            return key
        kmap_off = -1
        if kmap is not None:
            kmap_off = len(self.kmaps)
            self.kmaps.extend(int(v) for v in kmap)
        if kind in (0, 1):
            size = rows * taps * kp
        elif kind == 2:
            size = rows * kp
        elif kind == 3:
            size = rows * 4 * kp
        else:
            size = 9 * rows
        e = dict(name=name, kind=kind, n=n, k=k, taps=taps, kp=kp, rows=rows, kmap_off=kmap_off,
                 dst_off=self.total, size=size, kmap_len=0 if kmap is None else len(kmap))
        self.total += rup(size, 16)
        self.entries.append(e)
        self.index[key] = e
        self.table = None
        return key

    def finalize(self):
        dev = self.params.device
        self._keep.extend(t for t in (self.table, self.kmap_dev) if t is not None)
        if self.buf is None or self.buf.numel() < self.total:
            if self.buf is not None:
                self._keep.append(self.buf)
            self.buf = torch.zeros(max(self.total, 16), dtype=self.dtype, device=dev)
        arr = (L.IsaPackEntry * len(self.entries))()
        for i, e in enumerate(self.entries):
            arr[i] = L.IsaPackEntry(self.params.offsets[e["name"]], e["dst_off"], e["kind"], e["n"],
                                    e["k"], e["taps"], e["kp"], e["kmap_off"], e["rows"])
        raw = bytes(arr)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        km = self.kmaps if self.kmaps else [0]
        self.kmap_dev = torch.tensor(km, dtype=torch.int32, device=dev)

    def pack(self):
        if self.table is None:
            self.finalize()
        L.check(L.lib().isa_pack_weights(L.ptr(self.table), len(self.entries), L.ptr(self.kmap_dev),
                                         L.ptr(self.params.flat), L.ptr(self.buf),
                                         L.dtype_code(self.dtype), L.stream_ptr()), "isa_pack_weights")

    def ptr(self, key):
        e = self.index[key]
        return C.c_void_p(self.buf.data_ptr() + e["dst_off"] * self.buf.element_size())

    def kmap_ptr(self, key):
        e = self.index[key]
        if e["kmap_off"] < 0:
            return None
        return C.c_void_p(self.kmap_dev.data_ptr() + 4 * e["kmap_off"])


class Tape(list):
    """Backward tape.  Entries are (closure, stream id) - a closure runs in the backward pass on the stream its
    forward ran on - or (None, (src, dst)): a forward dependency `dst waits for src`, which the backward pass
    replays reversed (src waits for dst).  See Engine.on / Engine.sync."""

    def __init__(self, eng):
        super().__init__()
        self.eng = eng

    def append(self, fn):
        list.append(self, (fn, self.eng.cur))

    def mark_sync(self, src, dst):
        list.append(self, (None, (src, dst)))

    def last_fn(self):
        """The closure appended last, or a unique object when the tape is empty or ends in a sync edge."""
        return self[-1][0] if (self and self[-1][0] is not None) else Tape


class GradBook:
    """Gradient buffers mirror activation buffers; tracks which channel ranges were written."""

    def __init__(self, eng):
        self.eng = eng
        self.bufs: Dict[int, torch.Tensor] = {}
        self.written: Dict[int, List[Tuple[int, int]]] = {}

    def reset(self):
        self.bufs.clear()
        self.written.clear()
        self.shared: Dict[int, torch.Tensor] = {}     # activation buffers read by several concurrent branches

    def share(self, a: Act):
        """Mark a tensor whose consumers run on different streams (the decoder iterations all read the backbone
        features): each side stream accumulates its gradient in a buffer of its own, merge_shared() adds them."""
        self.shared[a.buf.data_ptr()] = a.buf

    def _key(self, a: Act):
        p = a.buf.data_ptr()
        cur = self.eng.cur
        return (p, cur) if (cur != 0 and p in self.shared) else p

    def grad_of(self, a: Act) -> Act:
        key = self._key(a)
        g = self.bufs.get(key)
        if g is None:
            g = self.eng.arena.alloc(a.buf.shape, a.buf.dtype)
            self.bufs[key] = g
            self.written[key] = []
        return Act(g, a.c0, a.c, groups=a.groups)

    def _covered(self, key, lo, hi):
        """Return (fully_written, fully_unwritten, missing ranges) for [lo,hi)."""
        segs = sorted(self.written[key])
        missing, cur = [], lo
        for s, e in segs:
            if e <= cur:
                continue
            if s >= hi:
                break
            if s > cur:
                missing.append((cur, min(s, hi)))
            cur = max(cur, e)
            if cur >= hi:
                break
        if cur < hi:
            missing.append((cur, hi))
        full = not missing
        none = len(missing) == 1 and missing[0] == (lo, hi)
        return full, none, missing

    def claim(self, a: Act, eng) -> int:
        """Called by a backward writer about to write grad_of(a).  Returns the `accumulate` flag
        and marks the range written; zero-fills gaps when the range is partially written."""
        key = self._key(a)
        if key not in self.bufs:
            self.grad_of(a)
        lo, hi = a.c0, a.c0 + a.c
        full, none, missing = self._covered(key, lo, hi)
        if not full:
            self.written[key].append((lo, hi))
        if full:
            return 1
        if none:
            return 0
        g = self.bufs[key]
        for s, e in missing:                      # rare: partially written -> zero the gaps
            eng.fill_zero(Act(g, s, e - s))
        return 1

    def has(self, a: Act) -> bool:
        key = self._key(a)
        if key not in self.bufs:
            return False
        full, _, _ = self._covered(key, a.c0, a.c0 + a.c)
        return full

    def merge_shared(self, eng):
        """On the main stream, after every side stream has been joined: add the side streams' gradients of the
        shared tensors into the main buffers."""
        assert eng.cur == 0
        for key in [k for k in self.bufs if isinstance(k, tuple)]:
            p, _ = key
            side = self.bufs[key]
            segs, merged = sorted(self.written[key]), []
            for lo, hi in segs:
                if merged and lo <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], hi)
                else:
                    merged.append([lo, hi])
            for lo, hi in merged:
                a = Act(self.shared[p], lo, hi - lo)
                acc = self.claim(a, eng)
                L.check(eng.lib.isa_axpy(Act(side, lo, hi - lo).d(), self.grad_of(a).d(), 1.0, acc, eng.st()),
                        "isa_axpy(merge shared grad)")
            self.written[key] = []


class _ProfiledLib:
    """Pass-through to the ctypes library; in profile mode brackets each launch with events on the
    launch stream so bench.py can report per-kernel-family durations measured live."""

    def __init__(self, lib, eng):
        self._lib, self._eng = lib, eng
        self._cache = {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        eng = self._eng

        def call(*args):
            if not eng.profile:
                return fn(*args)
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            eng.prof_events.append((name, s, e, eng.next_bytes))
            eng.next_bytes = 0
            return rc

        self._cache[name] = call
        setattr(self, name, call)
        return call


class Engine:
    BN_EPS = 1e-5
    BN_MOMENTUM = 0.1

    def __init__(self, params: ParamStore, dtype=torch.float32, device="cuda"):
        self.params = params
        self.dtype = dtype
        self.device = device
        self.packer = Packer(params, dtype)
        self.arenas: Dict[object, Arena] = {}      # one per configuration key (see begin)
        self.arena = self.arenas.setdefault(None, Arena(device))
        self.grads = GradBook(self)
        self.tape = Tape(self)
        self.cur = 0                   # stream id ops are issued on (0 = the step's own stream), see on()
        self.side_streams: Dict[int, torch.cuda.Stream] = {}
        self.main_stream = None
        self.defer_bn_running = False  # finalize leaves the running statistics to flush_bn_running()
        self.bn_running_queue: List[tuple] = []
        self.bn_train = False          # batch statistics + running-stat updates
        self.record = False            # build the backward tape
        self.lib = _ProfiledLib(L.lib(), self)
        self.ws = torch.empty(16 << 20, dtype=torch.float32, device=device)   # 64 MB reduction workspace
        self.profile = False           # when True every launch is bracketed by HIP events
        self.prof_events = []          # (entry point, start event, end event, algorithmic bytes)
        self.next_bytes = 0
        # fused BN-apply + depthwise backward + next BN-reduce (isa_dwconv3x3_bn_backward); ISA_FUSE_DW_BN=0
        # selects the separate kernels (A/B measurements, bisecting)
        self.fuse_dw_bn = os.environ.get("ISA_FUSE_DW_BN", "1") != "0"
        self._dw_out: Dict[tuple, dict] = {}
        # the same for bias-free 1x1 convs with <= 64 channels either side (isa_conv1x1_bn_backward, bf16 only)
        self.fuse_pw_bn = os.environ.get("ISA_FUSE_PW_BN", "1") != "0"
        # eval mode: BatchNorm (a constant affine), activation and residual ride in the conv's output epilogue
        # (isa_conv_gemm_ep) instead of a lazy prologue in the consumer plus a materialising pass per block
        self.fuse_eval = os.environ.get("ISA_FUSE_EVAL", "1") != "0"
        # ... and a whole InvertedV1Residual block in one launch where the shape allows (isa_dwpw_eval)
        self.fuse_block = os.environ.get("ISA_FUSE_BLOCK", "1") != "0"
        # train-mode BatchNorm finalizes run inside the consumer of the lazy tensor (PendingFin); 0: one launch each
        self.inline_fin = os.environ.get("ISA_INLINE_FIN", "1") != "0"
        self._pw_out: Dict[tuple, dict] = {}
        self._pw_in: Dict[tuple, dict] = {}      # conv input -> the same records: residual gradients ride along
        # eval-mode BN constants per layer: persistent buffers (a captured inference graph reads them), recomputed in
        # place by refresh_eval_bn() after the parameters changed
        self.eval_bn_cache: Dict[str, tuple] = {}
        self.eval_bn_stale = False
        # Deferred weight-gradient folds (isa_slab_arena_*): the ~270 second-stage fold launches of a backward pass
        # collapse into ~9 at its end; partial slabs live in their own arena until then.  The same arena gives every
        # weight-gradient call a slab region of its own, which concurrent streams need anyway.
        self.fold_arena = None
        self.slab = None               # isa_slab_arena* handle
        self.fold_stats = (0, 0)       # (folds, arena floats) of the last backward pass
        self._deferring = False
        self.defer_fold = os.environ.get("ISA_DEFER_FOLD", "1") != "0"    # 0: immediate folds (A/B, single stream only)
        self.bn_repeat = 1             # see repeated()
        self.arena_cache = int(os.environ.get("ISA_ARENA_CACHE", "4"))
        self._eval_bn_ready: Dict[str, tuple] = {}    # layer -> (event after its first-fill finalize, stream id it ran on)

    # ------------------------------------------------------------------ step lifecycle
    def begin(self, bn_train: bool, record: bool, key=None):
        """Start a step.  `key` names the configuration (mode, shapes, iteration count ...): each key owns its
        arena, so an eval forward between two replays of a captured training graph cannot free or reshape the
        buffers that graph points to (Arena.frozen turns a diverging sequence into an error)."""
        akey = (bool(bn_train), bool(record), key)
        arena = self.arenas.pop(akey, None)
        if arena is None:
            arena = Arena(self.device)
        self.arenas[akey] = arena                  # dict order = least recently used first
        # Every configuration pins a full activation set; arenas of captured graphs (frozen) must stay, the others are
        # bounded: the least recently used un-captured ones are dropped beyond ISA_ARENA_CACHE (default 4) of them
        loose = [k for k, a in self.arenas.items() if not a.frozen and k != akey and k is not None]
        for k in loose[:max(0, len(loose) - self.arena_cache)]:
            del self.arenas[k]
        self.arena = arena
        arena.reset()
        if POISON:
            self.ws.view(torch.uint8).fill_(0xFF)
        self.grads.reset()
        self.tape = Tape(self)
        self.cur = 0
        self.main_stream = torch.cuda.current_stream()
        self.defer_bn_running = False
        self.bn_running_queue = []
        self._dw_out = {}
        self._pw_out = {}
        self._pw_in = {}
        self.bn_train, self.record = bn_train, record
        if not bn_train and self.eval_bn_stale:
            self.refresh_eval_bn()

    def freeze_arena(self):
        """Called after a hipGraph captured the current configuration."""
        self.arena.frozen = True

    # ------------------------------------------------------------------ concurrent branches (HIP streams)
    def _stream(self, sid):
        if sid == 0:
            return self.main_stream
        st = self.side_streams.get(sid)
        if st is None:
            st = self.side_streams[sid] = torch.cuda.Stream(device=self.device)
        return st

    @contextlib.contextmanager
    def on(self, sid):
        """Issue the enclosed ops on stream `sid` (0 = the step's own stream).  Their backward closures run on the
        same stream.  Data that crosses streams needs an explicit sync() edge."""
        prev = self.cur
        self.cur = sid
        try:
            if sid == 0:
                with torch.cuda.stream(self.main_stream):
                    yield
            else:
                assert self.defer_fold or not self.record, "concurrent branches need the slab arena (ISA_DEFER_FOLD=1)"
                with torch.cuda.stream(self._stream(sid)):
                    yield
        finally:
            self.cur = prev

    def _wait(self, src, dst):
        if src == dst:
            return
        ev = torch.cuda.Event()
        ev.record(self._stream(src))
        self._stream(dst).wait_event(ev)

    def sync(self, src, dst):
        """Forward edge: everything issued so far on stream `src` happens before what follows on `dst`.  Recorded on
        the tape, so the backward pass replays the edge reversed (the gradients flow dst -> src).
        One end must be the step's own stream: ROCm 7.2's hipStreamEndCapture dereferences a null node when a captured
        graph holds side -> side edges (scripts/graph_probe.py) - a process-killing segfault, so it is an error here."""
        if src != 0 and dst != 0 and src != dst:
            raise RuntimeError("Engine.sync(%d, %d): stream edges must have the step's own stream (0) at one end - "
                               "side-to-side edges crash hipGraph capture on ROCm 7.2" % (src, dst))
        self._wait(src, dst)
        if self.record:
            self.tape.mark_sync(src, dst)

    def flush_bn_running(self):
        """Apply the running-statistics updates queued while defer_bn_running was set, in queue order, on the
        current stream (one launch per 64 layers)."""
        q = self.bn_running_queue
        if not q:
            return
        P = self.params
        # isa_bn_running_update runs one workgroup per descriptor, concurrently: two updates of the SAME layer (three or
        # more decoder iterations on side streams) must not share a launch - the queue is cut into runs of distinct layers
        runs, seen = [[]], set()
        for ent in q:
            if ent[2] in seen:
                runs.append([]); seen = set()
            seen.add(ent[2]); runs[-1].append(ent)
        for run in runs:
            arr = (L.IsaBnUpd * len(run))()
            for i, (stats, count, pre, c) in enumerate(run):
                arr[i] = L.IsaBnUpd(stats.data_ptr(), P.ptr(pre + ".running_mean").value, P.ptr(pre + ".running_var").value,
                                    count, c)
            L.check(self.lib.isa_bn_running_update(arr, len(run), self.BN_MOMENTUM, self.st()), "isa_bn_running_update")
        self.bn_running_queue = []

    def _finalize_train(self, stats, count, pre, c, scale, shift, mean, invstd, groups=1):
        """Train-mode finalize of one BatchNorm.  With inline_fin (default) nothing is launched here: the PendingFin that is
        returned travels with the lazy tensor and its first consumer does the work (see PendingFin)."""
        P = self.params
        defer = self.defer_bn_running
        rep = self.bn_repeat
        assert not (defer and (groups > 1 or rep > 1)), "deferred running statistics are a single-group mechanism"
        fin = None
        if self.inline_fin:
            fin = PendingFin(self, stats, count, pre, c, scale, shift, mean, invstd, groups, rep, running_taken=defer)
        else:
            L.check(self.lib.isa_bn_finalize(L.ptr(stats), count, P.ptr(pre + ".weight"), P.ptr(pre + ".bias"),
                                             None if defer else P.ptr(pre + ".running_mean"),
                                             None if defer else P.ptr(pre + ".running_var"), self.BN_MOMENTUM, self.BN_EPS,
                                             L.ptr(scale), L.ptr(shift), L.ptr(mean), L.ptr(invstd), c, groups, rep, self.st()),
                    "isa_bn_finalize")
        if defer:
            self.bn_running_queue.append((stats, count, pre, c))
        P.int_buffers[pre + ".num_batches_tracked"] += groups * rep
        return fin

    def defer_handle(self):
        """isa_slab_arena* for the weight-gradient entry points while a backward pass runs, else NULL."""
        return self.slab if (self._deferring and self.defer_fold) else None

    def _eval_bn_filled(self, pre):
        """The first-fill finalize of an eval-mode BN constant set was just queued on the current stream: remember an event
        so that a consumer on ANOTHER stream (the decoder iterations share weights and may meet the cache entry before
        that finalize ran) waits for it instead of reading unwritten constants."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._eval_bn_ready[pre] = (ev, self.cur)

    def _eval_bn_wait(self, pre):
        ent = self._eval_bn_ready.get(pre)
        if ent is not None and ent[1] != self.cur and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_event(ent[0])

    def refresh_eval_bn(self):
        """Recompute the cached eval-mode BN constants in place (same buffers: captured graphs keep reading them)."""
        P = self.params
        for pre, (scale, shift, mean, invstd) in self.eval_bn_cache.items():
            L.check(self.lib.isa_bn_finalize(None, 1.0, P.ptr(pre + ".weight"), P.ptr(pre + ".bias"),
                                             P.ptr(pre + ".running_mean"), P.ptr(pre + ".running_var"),
                                             self.BN_MOMENTUM, self.BN_EPS, L.ptr(scale), L.ptr(shift), L.ptr(mean),
                                             L.ptr(invstd), scale.numel(), 1, 1, self.st()), "isa_bn_finalize(eval refresh)")
        self.eval_bn_stale = False

    def scratch(self, numel) -> torch.Tensor:
        """fp32 scratch that is zero at step start (BN sums, reductions)."""
        return self.arena.scratch(numel)

    def f32(self, *shape):
        return self.arena.alloc(shape, torch.float32)

    def new_act(self, n, h, w, c, ld=None, dtype=None, groups=1):
        ld = rup(c, 8) if ld is None else ld
        return Act(self.arena.alloc((n, h, w, ld), dtype or self.dtype), 0, c, groups=groups)

    def like(self, x: Act, c, ld=None, n=None):
        """A new activation with x's batch, spatial size and statistic groups."""
        return self.new_act(x.n if n is None else n, x.h, x.w, c, ld, groups=x.groups)

    @contextlib.contextmanager
    def repeated(self, times):
        """The enclosed layers stand for `times` identical evaluations of the reference (same input, same weights, train-mode
        batch statistics: the `cross` block ahead of its Dropout2d in every decoder iteration, utils.py:984): computed
        once, their BatchNorm running statistics and num_batches_tracked advance `times` times."""
        prev = self.bn_repeat
        self.bn_repeat = times
        try:
            yield
        finally:
            self.bn_repeat = prev

    def st(self):
        return L.stream_ptr()

    # ------------------------------------------------------------------ tiny helpers
    def fill_zero(self, a: Act):
        L.check(self.lib.isa_axpy(a.d(), a.d(), 0.0, 0, self.st()), "isa_axpy(zero)")

    def copy(self, src: Act, dst: Act):
        L.check(self.lib.isa_axpy(src.d(), dst.d(), 1.0, 0, self.st()), "isa_axpy(copy)")
        if self.record and src.needs_grad:
            def bwd():
                acc = self.grads.claim(src, self)
                L.check(self.lib.isa_axpy(self.grads.grad_of(dst).d(), self.grads.grad_of(src).d(), 1.0, acc,
                                          self.st()), "isa_axpy(bwd)")
            self.tape.append(bwd)

    # ------------------------------------------------------------------ convolutions
    def reg_conv(self, wname, taps=1, kmap=None, transposed=False):
        """Register pack jobs for a conv weight; returns dict of keys."""
        shape = self.params.shapes[wname]
        if transposed:                       # ConvTranspose2d [K, Co, 2, 2]
            k, co = shape[0], shape[1]
            kphys = len(kmap) if kmap is not None else k
            kp = rup(kphys, 32)
            f = self.packer.add(wname, "fwd", 2, co, k, 4, kp, 4 * co, kmap)
            b = self.packer.add(wname, "dgrad", 3, co, k, 4, rup(co, 32), kphys, kmap)
            return dict(fwd=f, dgrad=b, kp=kp, kp_d=rup(co, 32), n=4 * co, kphys=kphys)
        n, k = shape[0], shape[1]
        kphys = len(kmap) if kmap is not None else k
        kp = rup(kphys, 32)
        f = self.packer.add(wname, "fwd", 0, n, k, taps, kp, n, kmap)
        b = self.packer.add(wname, "dgrad", 1, n, k, taps, rup(n, 32), kphys, kmap)
        return dict(fwd=f, dgrad=b, kp=kp, kp_d=rup(n, 32), n=n, kphys=kphys)

    def eval_fusable(self):
        """True when a forward pass may fold BatchNorm / activation / residual into conv epilogues: eval statistics, no tape."""
        return self.fuse_eval and not self.bn_train and not self.record

    def eval_bn(self, pre, c):
        """(scale, shift, mean, invstd) of an eval-mode BatchNorm: constants of the running statistics, computed once per
        weight version and cached (captured graphs keep reading the same buffers)."""
        cached = self.eval_bn_cache.get(pre)
        if cached is None:
            P = self.params
            cached = tuple(torch.empty(c, dtype=torch.float32, device=self.device) for _ in range(4))
            self.eval_bn_cache[pre] = cached
            L.check(self.lib.isa_bn_finalize(None, 1.0, P.ptr(pre + ".weight"), P.ptr(pre + ".bias"),
                                             P.ptr(pre + ".running_mean"), P.ptr(pre + ".running_var"),
                                             self.BN_MOMENTUM, self.BN_EPS, L.ptr(cached[0]), L.ptr(cached[1]),
                                             L.ptr(cached[2]), L.ptr(cached[3]), c, 1, 1, self.st()), "isa_bn_finalize")
            self._eval_bn_filled(pre)
        else:
            self._eval_bn_wait(pre)
        return cached

    def conv_bn_eval(self, x: Act, wname, out: Act, bn_pre, act, res: Optional[Act] = None, taps=1):
        """Eval mode only: out = act(BN(conv(pro(x)))) (+ res) in ONE launch (isa_conv_gemm_ep); `out` is a plain tensor."""
        assert self.eval_fusable()
        reg = self.reg_conv(wname, taps, None, False)
        if self.packer.table is None:
            self.packer.pack()
        scale, shift, _, _ = self.eval_bn(bn_pre, out.c)
        if self.profile:
            esz = x.buf.element_size()
            self.next_bytes = (x.n * x.h * x.w * x.c + (1 + (res is not None)) * out.n * out.h * out.w * out.c) * esz
        ep = L.IsaConvEp(L.addr(scale), L.addr(shift), act, C.addressof(res._c) if res is not None else None)
        self._keep_ep = (ep, res)                                  # ctypes argument lifetime: until the call returns
        L.check(self.lib.isa_conv_gemm_ep(x.d(), x.p(), self.packer.ptr(reg["fwd"]), reg["kp"], None, out.d(),
                                          L.IN_3X3 if taps == 9 else L.IN_1X1, C.byref(ep), self.st()), "isa_conv_gemm_ep")
        return out

    def block_v1_eval_fusable(self, x: Act, cout):
        """Shapes isa_dwpw_eval is built for (see include/isa_kernels.h); ISA_FUSE_BLOCK=0 selects the two-launch path."""
        return (self.fuse_block and self.eval_fusable() and self.dtype == torch.bfloat16 and x.pro is None and x.c % 32 == 0
                and x.c <= 128 and cout % 16 == 0 and cout <= 64 and x.h % 8 == 0 and x.w % 32 == 0 and x.ld % 8 == 0)

    def block_v1_eval(self, x: Act, pre, out: Act, res: Optional[Act] = None):
        """Eval mode only: the whole InvertedV1Residual (dw3x3 - BN - ReLU6 - pw - BN (+x)) in ONE launch; the depthwise
        output never leaves LDS (isa_dwpw_eval)."""
        rd = self.reg_dw(pre + ".conv.0.weight")
        rc = self.reg_conv(pre + ".conv.3.weight")
        if self.packer.table is None:
            self.packer.pack()
        s1, h1, _, _ = self.eval_bn(pre + ".conv.1", x.c)
        s2, h2, _, _ = self.eval_bn(pre + ".conv.4", out.c)
        if self.profile:
            esz = x.buf.element_size()
            self.next_bytes = (x.n * x.h * x.w * x.c + (1 + (res is not None)) * out.n * out.h * out.w * out.c) * esz
        ep = L.IsaConvEp(L.addr(s2), L.addr(h2), L.ACT_NONE, C.addressof(res._c) if res is not None else None)
        self._keep_ep = (ep, res)
        L.check(self.lib.isa_dwpw_eval(x.d(), self.packer.ptr(rd["fwd"]), L.ptr(s1), L.ptr(h1), self.packer.ptr(rc["fwd"]),
                                       rc["kp"], C.byref(ep), out.d(), self.st()), "isa_dwpw_eval")
        return out

    def conv(self, x: Act, wname, out: Act, *, taps=1, bias=None, stats=False, kmap=None,
             transposed=False, record_bwd=True):
        """out(raw) = conv(pro(x)).  Returns (out, stats tensor or None)."""
        reg = self.reg_conv(wname, taps, kmap, transposed)
        in_mode = L.IN_3X3 if taps == 9 else L.IN_1X1
        out_mode = L.OUT_SHUFFLE2 if transposed else L.OUT_PLAIN
        st = self.scratch(2 * out.c * STAT_R * out.groups) if stats else None
        job = (x, reg, bias, out, in_mode, out_mode, st)
        self._launch_conv(*job)
        self._last_conv = dict(reg=reg, in_mode=in_mode)
        if self.record and record_bwd:
            info = None
            if self.fuse_pw_bn and self.bn_train and self.dtype == torch.bfloat16 and taps == 1 and not transposed \
                    and kmap is None and bias is None and x.needs_grad and out.c <= 64 and x.c <= 64 \
                    and out.c % 8 == 0 and x.c % 8 == 0 and (x.pro is None or x.pro.bscale is None):
                xb = getattr(x, "bn", None)
                ok_x = xb is not None and xb["train"] and self.tape.last_fn() is xb.get("bwd_fn")
                info = dict(xbn=xb if ok_x else None, ybn=None, addend=None, done=False)
                self._pw_out[(out.buf.data_ptr(), out.c0, out.c)] = info
                self._pw_in[(x.buf.data_ptr(), x.c0, x.c)] = info

            def bwd():
                if info is not None:
                    info["done"] = True
                    if info["ybn"] is not None:
                        self._fused_pw_backward(x, wname, info)
                        return
                    if info["addend"] is not None:        # deferred residual gradient, conv ended up unfused
                        acc0 = self.grads.claim(x, self)
                        L.check(self.lib.isa_axpy(info["addend"].d(), self.grads.grad_of(x).d(), 1.0, acc0, self.st()),
                                "isa_axpy(res, deferred)")
                dy = self.grads.grad_of(out)
                pk = self.packer
                def wgrad(ws):
                    if transposed:      # the bias gradient (sum over all output pixels) rides in the four quadrant slabs
                        L.check(self.lib.isa_conv_wgrad(x.d(), x.p(), dy.d(), self.params.gptr(wname),
                                                        self.params.gptr(bias) if bias else None,
                                                        L.IN_1X1, L.OUT_SHUFFLE2, pk.kmap_ptr(reg["fwd"]),
                                                        self.params.shapes[wname][0], L.ptr(ws), ws.numel(),
                                                        self.defer_handle(), self.st()), "isa_conv_wgrad")
                    else:
                        if self.profile:
                            self.next_bytes = (x.n * x.h * x.w * x.c + dy.n * dy.h * dy.w * dy.c) * x.buf.element_size()
                        L.check(self.lib.isa_conv_wgrad(x.d(), x.p(), dy.d(), self.params.gptr(wname),
                                                        self.params.gptr(bias) if bias else None, in_mode,
                                                        L.OUT_PLAIN, pk.kmap_ptr(reg["fwd"]),
                                                        self.params.shapes[wname][1], L.ptr(ws), ws.numel(),
                                                        self.defer_handle(), self.st()), "isa_conv_wgrad")
                wgrad(self.ws)
                if x.needs_grad:
                    acc = self.grads.claim(x, self)
                    dx = self.grads.grad_of(x)
                    if transposed:
                        L.check(self.lib.isa_conv_gemm(dy.d(), None, pk.ptr(reg["dgrad"]), reg["kp_d"], None,
                                                       dx.d(), L.IN_GATHER2, L.OUT_PLAIN, None, acc, self.st()),
                                "isa_conv_gemm(dgradT)")
                    else:
                        if self.profile:
                            self.next_bytes = (dy.n * dy.h * dy.w * dy.c + dx.n * dx.h * dx.w * dx.c) * dy.buf.element_size()
                        L.check(self.lib.isa_conv_gemm(dy.d(), None, pk.ptr(reg["dgrad"]), reg["kp_d"], None,
                                                       dx.d(), in_mode, L.OUT_PLAIN, None, acc, self.st()),
                                "isa_conv_gemm(dgrad)")
            self.tape.append(bwd)
        return out, st

    def _bn_desc(self, b, red=None, out_red=None, with_params=True):
        P = self.params
        return L.IsaBnBwd(L.addr(b["scale"]), L.addr(b["shift"]), L.addr(b["mean"]), L.addr(b["invstd"]),
                          L.addr(red), L.addr(out_red),
                          P.gptr(b["pre"] + ".weight").value if with_params else None,
                          P.gptr(b["pre"] + ".bias").value if with_params else None, b["count"], b["act"])

    def _fused_pw_backward(self, x: Act, wname, info):
        """BN-apply of the conv's output BN + weight gradient + data gradient (+ reduce of the BN that
        produced x) in one pass: isa_conv1x1_bn_backward."""
        yb, yred, g = info["ybn"]
        xb = info["xbn"]
        ydesc = self._bn_desc(yb, red=yred)
        xdesc = None
        if xb is not None:
            xred = self.scratch(2 * x.c * STAT_R * x.groups)
            xdesc = self._bn_desc(xb, out_red=xred, with_params=False)
            xb["red_done"] = xred
        acc = self.grads.claim(x, self)
        if self.profile:
            self.next_bytes = x.n * x.h * x.w * (2 * g.c + 2 * x.c) * x.buf.element_size()
        L.check(self.lib.isa_conv1x1_bn_backward(
            g.d(), yb["raw"].d(), C.byref(ydesc), x.d(), x.p(), C.byref(xdesc) if xdesc is not None else None,
            self.params.ptr(wname), self.params.gptr(wname), self.grads.grad_of(x).d(), acc,
            info["addend"].d() if info["addend"] is not None else None,
            L.ptr(self.ws), self.ws.numel(), self.defer_handle(), self.st()), "isa_conv1x1_bn_backward")

    def _launch_conv(self, x, reg, bias, out, in_mode, out_mode, st):
        if self.packer.table is None:
            self.packer.pack()
        if self.profile:     # algorithmic bytes: input read once + output written once
            esz = x.buf.element_size()
            self.next_bytes = (x.n * x.h * x.w * x.c + out.n * out.h * out.w * out.c) * esz
        L.check(self.lib.isa_conv_gemm(x.d(), x.p_fin(), self.packer.ptr(reg["fwd"]), reg["kp"],
                                       self.params.ptr(bias) if bias else None, out.d(), in_mode, out_mode,
                                       L.ptr(st), 0, self.st()), "isa_conv_gemm")

    def reg_dw(self, wname, kmap=None):
        c = self.params.shapes[wname][0]
        rows = rup(len(kmap) if kmap is not None else c, 8)     # kernel reads rows of rup(C,8)
        f = self.packer.add(wname, "fwd", 4, c, 1, 9, 0, rows, kmap)
        b = self.packer.add(wname, "dgrad", 5, c, 1, 9, 0, rows, kmap)
        return dict(fwd=f, dgrad=b)

    def dwconv(self, x: Act, wname, out: Act, *, bias=None, stats=False, kmap=None):
        reg = self.reg_dw(wname, kmap)
        if self.packer.table is None:
            self.packer.pack()
        st = self.scratch(2 * out.c * STAT_R * out.groups) if stats else None
        if self.profile:
            self.next_bytes = 2 * x.n * x.h * x.w * x.c * x.buf.element_size()
        L.check(self.lib.isa_dwconv3x3(x.d(), x.p_fin(), self.packer.ptr(reg["fwd"]),
                                       self.params.ptr(bias) if bias else None, out.d(), L.ptr(st), self.st()),
                "isa_dwconv3x3")
        if self.record:
            info = None
            if self.fuse_dw_bn and self.bn_train and x.c % 8 == 0 and bias is None and x.needs_grad \
                    and (x.pro is None or x.pro.bscale is None):
                # BN(out)'s backward may leave its "apply" to this conv's backward (see bn()); BN(x)'s
                # "reduce" can be produced here when that BN was recorded immediately before this conv,
                # so every other consumer of x has already accumulated its gradient when we run
                xb = getattr(x, "bn", None)
                ok_x = xb is not None and xb["train"] and self.tape.last_fn() is xb.get("bwd_fn")
                info = dict(xbn=xb if ok_x else None, ybn=None, addend=None, done=False)
                self._dw_out[(out.buf.data_ptr(), out.c0, out.c)] = info
                if x.pro is None and xb is None:                                  # plain tensor: residual gradients may ride along
                    self._pw_in.setdefault((x.buf.data_ptr(), x.c0, x.c), info)

            def bwd():
                dy = self.grads.grad_of(out)
                nb = x.n * x.h * x.w * x.c * x.buf.element_size()
                if info is not None:
                    info["done"] = True
                    if info["ybn"] is None and info["addend"] is not None:    # deferred residual gradient, unfused after all
                        acc0 = self.grads.claim(x, self)
                        L.check(self.lib.isa_axpy(info["addend"].d(), self.grads.grad_of(x).d(), 1.0, acc0, self.st()),
                                "isa_axpy(res, deferred)")
                if info is not None and info["ybn"] is not None:
                    yb, yred = info["ybn"]
                    xb = info["xbn"]
                    ydesc = self._bn_desc(yb, red=yred)
                    xdesc = None
                    if xb is not None:
                        xred = self.scratch(2 * x.c * STAT_R * x.groups)
                        xdesc = self._bn_desc(xb, out_red=xred, with_params=False)
                        xb["red_done"] = xred
                    acc = self.grads.claim(x, self)
                    if self.profile:
                        self.next_bytes = 4 * nb
                    L.check(self.lib.isa_dwconv3x3_bn_backward(
                        dy.d(), yb["raw"].d(), C.byref(ydesc), x.d(), x.p(),
                        C.byref(xdesc) if xdesc is not None else None, self.packer.ptr(reg["dgrad"]),
                        self.params.gptr(wname), self.params.shapes[wname][0], self.grads.grad_of(x).d(), acc,
                        info["addend"].d() if info["addend"] is not None else None,
                        L.ptr(self.ws), self.ws.numel(), self.defer_handle(), self.st()), "isa_dwconv3x3_bn_backward")
                    return
                if self.profile:
                    self.next_bytes = 2 * nb
                L.check(self.lib.isa_dwconv3x3_wgrad(x.d(), x.p(), dy.d(), self.params.gptr(wname),
                                                     self.params.gptr(bias) if bias else None,
                                                     self.params.shapes[wname][0], L.ptr(self.ws), self.ws.numel(),
                                                     self.defer_handle(), self.st()), "isa_dwconv3x3_wgrad")
                if x.needs_grad:
                    acc = self.grads.claim(x, self)
                    if self.profile:
                        self.next_bytes = 2 * nb
                    L.check(self.lib.isa_dwconv3x3_dgrad(dy.d(), self.packer.ptr(reg["dgrad"]),
                                                         self.grads.grad_of(x).d(), acc, self.st()),
                            "isa_dwconv3x3_dgrad")
            self.tape.append(bwd)
        return out, st

    # ------------------------------------------------------------------ batch norm
    def bn(self, raw: Act, stats, pre, act, count=None) -> Act:
        """Lazy BN(+act): returns a view of `raw` whose prologue applies scale/shift/act.
        Gradient w.r.t. the lazy tensor is converted in place to the gradient w.r.t. `raw`."""
        c = raw.c
        G = raw.groups
        count = float(raw.n // G * raw.h * raw.w) if count is None else float(count)     # per statistic group
        P = self.params
        train = self.bn_train
        assert train or G == 1, "statistic groups exist in train mode only (eval statistics are shared)"
        cached = None if train else self.eval_bn_cache.get(pre)
        if cached is not None:
            scale, shift, mean, invstd = cached
        elif train:
            scale, shift, mean, invstd = (self.f32(G * c) for _ in range(4))
        else:       # eval: constants of the running statistics, computed once per weight version
            scale, shift, mean, invstd = (torch.empty(c, dtype=torch.float32, device=self.device) for _ in range(4))
            self.eval_bn_cache[pre] = (scale, shift, mean, invstd)
        fin = None
        if train:
            fin = self._finalize_train(stats, count, pre, c, scale, shift, mean, invstd, G)
        elif cached is None:
            L.check(self.lib.isa_bn_finalize(None, count, P.ptr(pre + ".weight"), P.ptr(pre + ".bias"),
                                             P.ptr(pre + ".running_mean"), P.ptr(pre + ".running_var"),
                                             self.BN_MOMENTUM, self.BN_EPS, L.ptr(scale), L.ptr(shift), L.ptr(mean),
                                             L.ptr(invstd), c, 1, 1, self.st()), "isa_bn_finalize")
            self._eval_bn_filled(pre)
        else:
            self._eval_bn_wait(pre)
        lazy = raw.with_pro(Pro(scale, shift, act, fin=fin))
        lazy.bn = dict(pre=pre, scale=scale, shift=shift, mean=mean, invstd=invstd, act=act, count=count,
                       train=train, raw=raw)
        if self.record:
            key = (raw.buf.data_ptr(), raw.c0, raw.c)
            dw = self._dw_out.get(key) if train else None
            pw = self._pw_out.get(key) if train else None

            def bwd():
                g = self.grads.grad_of(raw)
                if dw is not None:          # raw came from a depthwise conv: reduce here, apply inside its backward
                    red = self._bn_backward(lazy, g, g, None, do_apply=False)
                    dw["ybn"] = (lazy.bn, red)
                    return
                if pw is not None:          # ... or from a small 1x1 conv
                    red = self._bn_backward(lazy, g, g, None, do_apply=False)
                    pw["ybn"] = (lazy.bn, red, g)
                    return
                self._bn_backward(lazy, g, g, None)
            lazy.bn["bwd_fn"] = bwd
            self.tape.append(bwd)
        return lazy

    def _bn_backward(self, lazy: Act, dt: Act, dy: Act, bscale, do_apply=True):
        """reduce (sum g', sum g'*xhat) then apply; `red_done` on the layer means a fused producer
        already wrote the sums; do_apply=False leaves the apply to a fused consumer and returns them."""
        b = lazy.bn
        P = self.params
        red = b.pop("red_done", None)
        nb = lazy.n * lazy.h * lazy.w * lazy.c * lazy.buf.element_size()
        if b["train"] and red is None:
            red = self.scratch(2 * lazy.c * STAT_R * lazy.groups)
            if self.profile:
                self.next_bytes = 2 * nb
            L.check(self.lib.isa_bn_bwd_reduce(dt.d(), b["raw"].d(), L.ptr(b["scale"]), L.ptr(b["shift"]),
                                               L.ptr(b["mean"]), L.ptr(b["invstd"]), b["act"], L.ptr(bscale),
                                               L.ptr(red), self.st()), "isa_bn_bwd_reduce")
        if not do_apply:
            return red
        if self.profile:
            self.next_bytes = 3 * nb
        L.check(self.lib.isa_bn_bwd_apply(dt.d(), b["raw"].d(), L.ptr(b["scale"]), L.ptr(b["shift"]),
                                          L.ptr(b["mean"]), L.ptr(b["invstd"]), b["act"], L.ptr(bscale),
                                          P.ptr(b["pre"] + ".weight"), L.ptr(red), b["count"], 1 if b["train"] else 0,
                                          dy.d(), P.gptr(b["pre"] + ".weight"), P.gptr(b["pre"] + ".bias"),
                                          self.st()), "isa_bn_bwd_apply")
        return red

    def bn_out(self, raw: Act, stats, pre, act, out: Act, res: Optional[Act] = None, bscale=None,
               count=None, res2: Optional[Act] = None, oscale=None) -> Act:
        """Materialising BN: out = (act(BN(raw)) * bscale (+ res) (+ res2)) * oscale.
        Broadcast form (train mode): raw / res hold one statistic group, out has G of them with a per-image oscale each
        (isa_affine_act_res); the gradient of the shared value is the oscale-weighted sum over the groups."""
        c = raw.c
        G = raw.groups
        count = float(raw.n // G * raw.h * raw.w) if count is None else float(count)     # per statistic group
        P = self.params
        train = self.bn_train
        assert train or G == 1, "statistic groups exist in train mode only (eval statistics are shared)"
        cached = None if train else self.eval_bn_cache.get(pre)
        if cached is not None:
            scale, shift, mean, invstd = cached
        elif train:
            scale, shift, mean, invstd = (self.f32(G * c) for _ in range(4))
        else:       # eval: constants of the running statistics, computed once per weight version
            scale, shift, mean, invstd = (torch.empty(c, dtype=torch.float32, device=self.device) for _ in range(4))
            self.eval_bn_cache[pre] = (scale, shift, mean, invstd)
        fin = None
        if train:
            fin = self._finalize_train(stats, count, pre, c, scale, shift, mean, invstd, G)
        elif cached is None:
            L.check(self.lib.isa_bn_finalize(None, count, P.ptr(pre + ".weight"), P.ptr(pre + ".bias"),
                                             P.ptr(pre + ".running_mean"), P.ptr(pre + ".running_var"),
                                             self.BN_MOMENTUM, self.BN_EPS, L.ptr(scale), L.ptr(shift), L.ptr(mean),
                                             L.ptr(invstd), c, 1, 1, self.st()), "isa_bn_finalize")
            self._eval_bn_filled(pre)
        else:
            self._eval_bn_wait(pre)
        lazy = raw.with_pro(Pro(scale, shift, act, bscale, fin=fin))
        lazy.bn = dict(pre=pre, scale=scale, shift=shift, mean=mean, invstd=invstd, act=act, count=count,
                       train=train, raw=raw)
        if self.profile:
            self.next_bytes = (2 + (res is not None) + (res2 is not None)) * raw.n * raw.h * raw.w * raw.c * raw.buf.element_size()
        L.check(self.lib.isa_affine_act_res(lazy.d(), lazy.p_fin(), res.d() if res is not None else None,
                                            res2.d() if res2 is not None else None, L.ptr(oscale), out.d(),
                                            self.st()), "isa_affine_act_res")
        if self.record:
            pw = self._pw_out.get((raw.buf.data_ptr(), raw.c0, raw.c)) if (train and bscale is None) else None

            bcast = out.n != raw.n
            assert not bcast or (oscale is not None and res2 is None and out.n == out.groups * raw.n and raw.groups == 1)

            def bwd():
                dout = self.grads.grad_of(out)
                dsum = dout
                if oscale is not None:           # d(sum) = dout * dropout mask (summed over the groups of a broadcast)
                    dsum = self.like(raw, out.c)
                    L.check(self.lib.isa_scale_bc(dout.d(), L.ptr(oscale), dsum.d(), 0, self.st()), "isa_scale_bc")
                for r in (res, res2):
                    if r is not None and r.needs_grad:
                        pin = self._pw_in.get((r.buf.data_ptr(), r.c0, r.c))
                        if pin is not None and not pin["done"] and pin["addend"] is None:
                            pin["addend"] = dsum          # the consuming 1x1 conv's backward adds it to dx itself
                            continue
                        acc = self.grads.claim(r, self)
                        L.check(self.lib.isa_axpy(dsum.d(), self.grads.grad_of(r).d(), 1.0, acc, self.st()),
                                "isa_axpy(res)")
                if pw is not None:                # the producing 1x1 conv's backward applies this BN on the fly
                    red = self._bn_backward(lazy, dsum, None, None, do_apply=False)
                    pw["ybn"] = (lazy.bn, red, dsum)
                    return
                self.grads.claim(raw, self)       # single consumer: overwrite
                self._bn_backward(lazy, dsum, self.grads.grad_of(raw), bscale)
            self.tape.append(bwd)
        return out

    def act(self, raw: Act, act) -> Act:
        """Lazy activation without BN (tanh / LeakyReLU after a biased conv).  The gradient w.r.t.
        the activated value is converted in place to the gradient w.r.t. `raw`."""
        lazy = raw.with_pro(Pro(act=act))
        if self.record:
            def bwd():
                g = self.grads.grad_of(raw)
                L.check(self.lib.isa_bn_bwd_apply(g.d(), raw.d(), None, None, None, None, act, None, None, None, 1.0,
                                                  0, g.d(), None, None, self.st()), "isa_bn_bwd_apply(act)")
            self.tape.append(bwd)
        return lazy

    def act_out(self, raw: Act, act, out: Act) -> Act:
        """Materialised activation out = act(raw) for consumers that would re-evaluate a lazy prologue many
        times (a 3x3 conv applies it once per tap: tanh ahead of attend_fc cost 9 tanhf per element)."""
        lazy = raw.with_pro(Pro(act=act))
        L.check(self.lib.isa_affine_act_res(lazy.d(), lazy.p(), None, None, None, out.d(), self.st()),
                "isa_affine_act_res(act)")
        if self.record:
            def bwd():
                acc = self.grads.claim(raw, self)
                assert acc == 0, "act_out expects to be the only consumer of its input"
                L.check(self.lib.isa_bn_bwd_apply(self.grads.grad_of(out).d(), raw.d(), None, None, None, None, act,
                                                  None, None, None, 1.0, 0, self.grads.grad_of(raw).d(), None, None,
                                                  self.st()), "isa_bn_bwd_apply(act_out)")
            self.tape.append(bwd)
        return out

    def materialize(self, x: Act, out: Act, res: Optional[Act] = None):
        """out = pro(x) (+res) for a lazy x that is NOT a BN output (bias+act convs)."""
        L.check(self.lib.isa_affine_act_res(x.d(), x.p(), res.d() if res is not None else None, None, None,
                                            out.d(), self.st()), "isa_affine_act_res")
        assert not self.record, "use bn_out / lazy consumers on the training path"
        return out

    # ------------------------------------------------------------------ pooling
    def avgpool2(self, x: Act, out: Act):
        L.check(self.lib.isa_avgpool2(x.d(), out.d(), self.st()), "isa_avgpool2")
        if self.record and x.needs_grad:
            def bwd():
                acc = self.grads.claim(x, self)
                L.check(self.lib.isa_avgpool2_bwd(self.grads.grad_of(out).d(), self.grads.grad_of(x).d(), acc,
                                                  self.st()), "isa_avgpool2_bwd")
            self.tape.append(bwd)
        return out

    def profile_summary(self):
        """{entry point: (calls, total ms, algorithmic bytes)} from the recorded events."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, nbytes in self.prof_events:
            c, t, b = out.get(name, (0, 0.0, 0))
            out[name] = (c + 1, t + s.elapsed_time(e), b + nbytes)
        self.prof_events = []
        return out

    # ------------------------------------------------------------------ backward driver
    def backward(self):
        """Run the tape in reverse.  Closures run on the stream their forward ran on; forward sync edges are
        replayed reversed.  Weight-gradient slabs go to the slab arena, folded once at the end (main stream, after
        every side stream has been joined by the reversed fork edges)."""
        if self.slab is None:
            if self.fold_arena is None:
                self.fold_arena = torch.empty(int(os.environ.get("ISA_FOLD_ARENA_MB", "2048")) << 18,
                                              dtype=torch.float32, device=self.device)
            h = C.c_void_p()
            L.check(self.lib.isa_slab_arena_create(L.ptr(self.fold_arena), self.fold_arena.numel(), C.byref(h)),
                    "isa_slab_arena_create")
            self.slab = h
        L.check(self.lib.isa_slab_arena_begin(self.slab), "isa_slab_arena_begin")
        if POISON:
            self.fold_arena.view(torch.uint8).fill_(0xFF)
        self._deferring = True
        try:
            for fn, tag in reversed(self.tape):
                if fn is None:
                    src, dst = tag
                    self._wait(dst, src)
                else:
                    with self.on(tag):
                        fn()
        finally:
            self._deferring = False
            nf, used = C.c_int32(0), C.c_int64(0)
            L.check(self.lib.isa_slab_arena_flush(self.slab, self.st(), C.byref(nf), C.byref(used)),
                    "isa_slab_arena_flush")
            self.fold_stats = (nf.value, used.value)
        self.tape = Tape(self)
