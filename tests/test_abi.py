"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares; product-side schema equals the oracle's (and the reference's, via the golden run)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]


def test_library_exports_every_declared_symbol():
    import isa_amd  # noqa: F401
    from isa_amd import lib as L
    hdr = open(os.path.join(ROOT, "include", "isa_kernels.h")).read()
    declared = set(re.findall(r"\bint\s+(isa_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), name
    sizes = set(re.findall(r"\bint64_t\s+(isa_[a-z0-9_]+)\s*\(", hdr))         # entry points that return a size
    assert sizes == {"isa_resize_bilinear_ws_bytes"}
    for name in sizes:
        assert hasattr(lib, name), name


def test_missing_library_fails_loudly(monkeypatch):
    import isa_amd  # noqa: F401
    from isa_amd import lib as L
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libisa_kernels.so")
    monkeypatch.setattr(L, "_lib", None)
    try:
        L.lib()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected ImportError")


def test_schema_matches_oracle_and_survey_counts():
    import isa_amd  # noqa: F401
    from isa_amd.schema import state_dict_schema as prod
    from reseg_ref import state_dict_schema as orac
    assert prod(True) == orac(True) and prod(False) == orac(False)
    s = prod(True)
    n_el = 0
    for _, shp in s:
        k = 1
        for d in shp:
            k *= d
        n_el += k
    assert len(s) == 891 and n_el == 4824330          # SURVEY.md §8(b) probe of the reference


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "instance-segmentation-attention_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            code = "\n".join(l for l in open(os.path.join(pkg, fn)).read().splitlines()
                             if l.lstrip().startswith(("import ", "from ")))
            assert "reseg_ref" not in code and "oracle" not in code and "golden_io" not in code, fn
