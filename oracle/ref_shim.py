"""Harness-side shim that makes the upstream reference network importable on CPU.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/gen_golden.py` (fixture generation, in the build
container where `/root/reference` is mounted) and by the `not gpu` pinning tests when the
reference happens to be present.  Nothing here is imported by the product path, and nothing in
`/root/reference` is edited or copied: we only adjust the *interpreter environment* so that the
reference's own modules import on a CPU-only, modern torch stack:

  (i)   sys.path in the order the reference's own entry script sets it up (train.py:3-6);
  (ii)  `Tensor.cuda` / `Module.cuda` become identity (utils.py:11-12 calls .cuda() at import);
  (iii) empty stub modules for third-party imports that are absent here and unused on the path
        (cv2, visdom, lmdb, skimage.*, torchvision.*);
  (iv)  `Tensor.masked_fill` accepts the uint8 masks torch>=2 rejects (utils.py:507,648).
"""
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("ISA_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "code", "lib", "archs"))


_installed = False


class _Placeholder(object):
    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed third-party symbol %s was called" % self._name)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Placeholder(self._name + "." + item)


def install():
    """Idempotently install the shim; returns the reference `reseg`, `config` modules."""
    global _installed
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if not _installed:
        code = os.path.join(REF_ROOT, "code")
        for p in [code, code + "/lib", code + "/lib/archs", code + "/lib/losses",
                  code + "/lib/archs/modules", code + "/settings/CVPPP"]:
            if p not in sys.path:
                sys.path.append(p)
        # (ii) .cuda() -> identity
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        # (iii) stubs
        names = ["cv2", "visdom", "lmdb", "skimage", "skimage.color", "skimage.transform",
                 "skimage.filters", "skimage.io", "torchvision", "torchvision.models",
                 "torchvision.transforms", "torchvision.datasets"]
        class _Stub(types.ModuleType):
            """Any attribute resolves to an inert placeholder (never called on the path)."""

            def __getattr__(self, item):
                if item.startswith("__"):
                    raise AttributeError(item)
                return _Placeholder(self.__name__ + "." + item)

        for n in names:
            if n not in sys.modules:
                m = _Stub(n)
                m.__path__ = []  # behave as a package
                sys.modules[n] = m
        for n in names:
            if "." in n:
                parent, child = n.rsplit(".", 1)
                setattr(sys.modules[parent], child, sys.modules[n])
        # (iv) byte masks
        _mf = torch.Tensor.masked_fill

        def masked_fill(self, mask, value):
            if mask.dtype == torch.uint8:
                mask = mask.bool()
            return _mf(self, mask, value)

        torch.Tensor.masked_fill = masked_fill
        _installed = True
    import losses  # noqa: F401  (must precede `dice` to dodge the circular import)
    import config
    import reseg
    return reseg, config
