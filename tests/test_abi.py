"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/mlmap_hip.h
declares, fails loudly without a GPU, and the C++ facade compiles against the header."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "mlmap_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mlm_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from mlmapping_amd.mlmap import ABI_SYMBOLS, load_library

    lib = load_library()
    decl = _declared()
    assert len(decl) >= 25
    for s in decl:
        assert hasattr(lib, s), f"{s} declared in include/mlmap_hip.h but not exported"
    assert sorted(ABI_SYMBOLS) == decl, "python binding and header disagree on the ABI surface"
    assert lib.mlm_abi_version() == 6


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly (never route to the oracle or any CPU path)."""
    import ctypes

    from mlmapping_amd.config import S1
    from mlmapping_amd.mlmap import MLMap, MlmError, load_library

    lib = load_library()
    n = ctypes.c_int(0)
    hip = ctypes.CDLL(None)  # the HIP runtime the library is bound to (already in the process, see mlmap.load_library)
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(MlmError):
        MLMap(S1)


def test_product_does_not_import_oracle():
    """Nothing under mlmapping_amd/ or include/ may reference the oracle."""
    bad = []
    for base in ("mlmapping_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if "mlo_" in txt or "libmlmap_oracle" in txt or re.search(r"(from|import)\s+oracle", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_facade_compiles(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "mlmap_facade.hpp"\nint main(){ mlmap_hip::mlmap m; (void)m; return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
