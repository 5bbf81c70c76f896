"""Fixture encoding shared by oracle/gen_golden.py (writer) and tests/ (reader).  Test infra."""
import numpy as np

MAX_FULL = 4096


def subsample_stride(n, max_full=MAX_FULL):
    s = max(1, n // max_full)
    while s > 1 and any(s % p == 0 for p in (2, 3, 5, 7)):   # avoid aliasing with NCHW strides
        s += 1
    return s


def pack(name, t, store, max_full=MAX_FULL):
    """Store tensor `t`: full if small, else strided subsample; always float64 checksums."""
    a = t.detach().cpu().numpy()
    a64 = a if a.dtype == np.float64 else a.astype(np.float64)
    store[name + "/shape"] = np.array(a.shape, dtype=np.int64)
    store[name + "/sums"] = np.array([a64.sum(), np.abs(a64).sum(), (a64 * a64).sum()])
    flat = a.reshape(-1)
    if flat.size <= max_full:
        store[name + "/full"] = flat.astype(np.float32)
    else:
        store[name + "/sub"] = flat[::subsample_stride(flat.size, max_full)].astype(np.float32)


def pack_bits(name, t, store):
    a = (t.detach().cpu().numpy() != 0)
    store[name + "/shape"] = np.array(a.shape, dtype=np.int64)
    store[name + "/bits"] = np.packbits(a.reshape(-1))


def unpack_bits(z, name):
    shape = tuple(int(v) for v in z[name + "/shape"])
    n = int(np.prod(shape))
    return np.unpackbits(z[name + "/bits"])[:n].reshape(shape).astype(bool)


def compare(z, name, got, max_full=MAX_FULL):
    """Compare array-like `got` (NCHW order) with fixture entry `name`.
    Returns (max_abs_err / max_abs_ref over stored samples, rel err of the abs-sum checksum)."""
    g = np.asarray(got, dtype=np.float64)
    shape = tuple(int(v) for v in z[name + "/shape"])
    assert tuple(g.shape) == shape, (name, g.shape, shape)
    flat = g.reshape(-1)
    if name + "/full" in z:
        ref = z[name + "/full"].astype(np.float64)
        mine = flat
    else:
        ref = z[name + "/sub"].astype(np.float64)
        mine = flat[::subsample_stride(flat.size, max_full)]
    scale = max(np.abs(ref).max(), 1e-30)
    err = np.abs(mine - ref).max() / scale
    sums = z[name + "/sums"]
    cs = abs(np.abs(g).sum() - sums[1]) / max(sums[1], 1e-30)
    return float(err), float(cs)
