"""C-ABI surface checks that need no GPU: the library loads and exports every symbol include/nct.h declares."""
import ctypes
import os
import re
import pytest

import nct

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "nct.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nct_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert "nct_create" in syms and "nct_patchmatch" in syms and len(syms) >= 10


def test_library_exports_every_declared_symbol():
    l = ctypes.CDLL(nct.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(l, s)]
    assert not missing, f"libnct.so does not export: {missing}"


def test_binding_covers_every_declared_symbol():
    missing = [s for s in _declared_symbols() if s not in nct.SIGNATURES]
    assert not missing, f"python binding lacks: {missing}"


def test_version():
    import re
    hdr = open(os.path.join(os.path.dirname(nct.PKG_ROOT), "include", "nct.h")).read()
    v = int(re.search(r"#define\s+NCT_VERSION\s+(\d+)", hdr).group(1))
    assert nct.lib().nct_version() == v == nct.NCT_VERSION        # header, library and binding agree (the binding refuses a library of another version)


def test_no_cpu_fallback_create_fails_loudly_without_gpu():
    """On a box without a HIP device the product must refuse to run (no fallback to the oracle / CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path is exercised on CPU-only boxes")
    with pytest.raises(nct.NctError) as e:
        nct.Context(0)
    assert e.value.code == -1 and "no CPU fallback" in str(e.value)


def test_product_never_links_oracle():
    """The shipped library must not reference anything under oracle/ (checked on the dynamic symbol table)."""
    import subprocess
    out = subprocess.run(["nm", "-D", nct.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out
    out = subprocess.run(["ldd", nct.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in out
