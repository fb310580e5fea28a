"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/pistoseg_hip.h declares; argument validation works without a GPU (no compute call is made)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", text))


@pytest.fixture(scope="module")
def lib():
    from pistoseg_amd import _lib, build

    build.build(verbose=False)
    return _lib.load()


def test_exports_match_header(lib):
    from pistoseg_amd import _lib

    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["pistoseg_hip.h", "pistoseg_hip_debug.h"]
    decl = declared_symbols("pistoseg_hip.h")
    assert decl, "no declarations parsed"
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/pistoseg_hip.h but not exported"
    assert decl == set(_lib.PROTOTYPES), (decl ^ set(_lib.PROTOTYPES))
    assert lib.ps_version() == 226


def test_product_library_has_no_debug_switches_and_debug_library_has_all(lib):
    """The product ABI carries no process-global knob: no `ps_debug_*`, no `ps_set_*` (tiles-per-block is a field of ps_conv_geom);
    the hooks of include/pistoseg_hip_debug.h are exported by libpistoseg_hip_debug.so only."""
    from pistoseg_amd import _lib

    dbg_decl = declared_symbols("pistoseg_hip_debug.h")
    assert dbg_decl == set(_lib.DEBUG_PROTOTYPES) and all(n.startswith("ps_debug_") for n in dbg_decl)
    for name in dbg_decl | {"ps_set_tiles_per_block"}:
        assert not hasattr(lib, name), f"product library exports {name}"
    assert not [n for n in _lib.PROTOTYPES if n.startswith(("ps_debug_", "ps_set_"))]
    dbg = C.CDLL(_lib.DEBUG_LIB_PATH)
    for name in dbg_decl | declared_symbols("pistoseg_hip.h"):
        assert hasattr(dbg, name), f"debug library lacks {name}"
    assert any(f[0] == "tiles_per_block" for f in _lib.ConvGeom._fields_)


def test_argument_validation_without_gpu(lib):
    from pistoseg_amd._lib import PS_BF16, PS_F32, ConvGeom

    ok = ConvGeom(PS_BF16, 2, 28, 28, 512, 1024, 3, 1, 4, 512, 1024)
    assert lib.ps_conv_supported(C.byref(ok)) == 1
    bad_k = ConvGeom(PS_BF16, 2, 28, 28, 512, 1024, 5, 1, 1, 512, 1024)
    assert lib.ps_conv_supported(C.byref(bad_k)) == 0 and b"ksize" in lib.ps_last_error()
    bad_c = ConvGeom(PS_F32, 2, 28, 28, 195, 192, 1, 1, 1, 195, 192)
    assert lib.ps_conv_supported(C.byref(bad_c)) == 0 and b"multiples" in lib.ps_last_error()
    # null pointers are rejected before any launch
    assert lib.ps_softmax_ce(None, None, None, None, 1.0, 1, 3, 8, 8, 3, None, None) == -1
    assert lib.ps_fc8_fwd(PS_F32, None, 0, None, None, None, 1, 1, 4096, 4, None) == -1


def test_ops_refuse_cpu_tensors(lib):
    import torch

    from pistoseg_amd import _lib, ops

    x = torch.zeros(1, 4, 4, 64)
    with pytest.raises(_lib.PsError):
        ops.conv2d_fwd(ops.ConvSpec(64, 64, 1), x, torch.zeros(64, 1, 1, 64), out_raw=torch.zeros(1, 4, 4, 64))
