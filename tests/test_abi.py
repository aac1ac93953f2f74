"""The C-ABI library loads and exports every symbol include/ifx_c_api.h declares (no GPU needed)."""
import os
import re
import subprocess

import instancefusion_amd as ifx


def _declared():
    src = open(ifx.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ifx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    assert os.path.exists(ifx.LIB_PATH), "build libifx.so first: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.check_output(["nm", "-D", "--defined-only", ifx.LIB_PATH]).decode()
    exported = set(re.findall(r"\bT (ifx_[a-z0-9_]+)", out))
    missing = [s for s in _declared() if s not in exported]
    assert not missing, f"declared but not exported: {missing}"


def test_binding_covers_header():
    decl = set(_declared())
    bound = set(ifx.exported_symbols())
    assert decl == bound, (sorted(decl - bound), sorted(bound - decl))
    ifx.lib()  # binds all of them via ctypes; raises on a missing symbol


def test_no_oracle_in_product():
    """The product never links or loads the oracle."""
    out = subprocess.check_output(["ldd", ifx.LIB_PATH]).decode()
    assert "liborc" not in out
    pkg = os.path.dirname(ifx.LIB_PATH)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "liborc" not in txt and "oracle_lib" not in txt, f


def test_create_without_gpu_fails_loudly():
    import ctypes as C

    import torch

    if torch.cuda.is_available():
        return
    L = ifx.lib()
    cfg = ifx.IfxConfig(**ifx.default_config())
    h = C.c_void_p()
    r = L.ifx_create(C.byref(cfg), C.byref(h))
    assert r < 0 and b"no HIP device" in L.ifx_global_error()
