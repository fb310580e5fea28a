"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/pistoseg_hip.h declares; argument validation works without a GPU (no compute call is made)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names.update(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", text))
    return names


@pytest.fixture(scope="module")
def lib():
    from pistoseg_amd import _lib, build

    build.build(verbose=False)
    return _lib.load()


def test_exports_match_header(lib):
    from pistoseg_amd import _lib

    decl = declared_symbols()
    assert decl, "no declarations parsed"
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert decl == set(_lib.PROTOTYPES), (decl ^ set(_lib.PROTOTYPES))
    assert lib.ps_version() == 100


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
