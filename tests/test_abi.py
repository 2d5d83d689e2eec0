"""The C-ABI shared library loads without a GPU and exports every symbol include/clipfs.h declares; the
ctypes signature table covers exactly the header (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "clipfs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clipfs_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_functions():
    fns = _header_functions()
    assert "clipfs_gemm_nt" in fns and "clipfs_tower_fwd" in fns and "clipfs_mta" in fns
    assert len(fns) >= 30


def test_library_exports_every_header_symbol():
    from clipfs import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [f for f in _header_functions() if not hasattr(lib, f)]
    assert not missing, f"declared in clipfs.h but not exported: {missing}"


def test_signature_table_matches_header():
    from clipfs import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()
    lib = _lib.load()
    assert lib.clipfs_abi_version() == _lib.ABI_VERSION == 2


def test_argument_errors_do_not_touch_the_gpu():
    """Bad arguments return CLIPFS_EINVAL with a message and launch nothing (safe without a GPU)."""
    from clipfs import _lib
    lib = _lib.load()
    rc = lib.clipfs_layernorm_fwd(None, 0, None, None, None, None, None, 4, 7, 1e-5, None)
    assert rc == 1
    assert b"width" in lib.clipfs_last_error()
    rc = lib.clipfs_attention_fwd(None, None, None, 1, 5000, 2, 0, None)
    assert rc == 1 and b"seq" in lib.clipfs_last_error()
    assert lib.clipfs_gemm_nt(None, None) == 1
    # ABI v2: a descriptor built against another header (wrong struct_size) is rejected before anything is read from it
    g = _lib.GemmArgs()
    g.M = g.N = g.K = 32
    assert lib.clipfs_gemm_nt(ctypes.byref(g), None) == 1 and b"struct_size" in lib.clipfs_last_error()
    t = _lib.Tower()
    blocks = (_lib.Block * 1)()
    t.blocks = ctypes.cast(blocks, ctypes.POINTER(_lib.Block))
    t.width, t.heads, t.layers, t.seq = 64, 1, 1, 8
    assert lib.clipfs_tower_fwd(ctypes.byref(t), None, 1, None, None, None) == 1
    assert b"struct_size" in lib.clipfs_last_error()
    assert lib.clipfs_tower_scratch_floats(ctypes.byref(t), 1) == 0
    assert lib.clipfs_tower_scratch_floats(ctypes.byref(_fill(_lib.new_tower(), t)), 1) > 0
    with pytest.raises(_lib.ClipfsError):
        _lib.check(rc, "attention_fwd")


def _fill(dst, src):
    for name in ("blocks", "width", "heads", "layers", "seq"):
        setattr(dst, name, getattr(src, name))
    return dst


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import it."""
    pkg = os.path.join(ROOT, "jittor-clip-fewshot_amd")
    bad = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
