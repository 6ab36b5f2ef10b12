"""C-ABI surface (no GPU): libmfnet_hip.so loads, exports every symbol include/mfnet_hip.h declares, the
ctypes table binds exactly those, and argument validation rejects bad shapes before any launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "mfnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mfnet_[a-z0-9_]+)\s*\(", hdr)))


def test_header_library_and_binding_agree():
    from ssrs_amd import _lib
    names = _declared()
    assert len(names) >= 29
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mfnet_hip.h but not exported"
    assert sorted(_lib.exported_symbols()) == names
    assert _lib.lib().mfnet_version() >= 100


def test_argument_errors_do_not_launch():
    """null pointers / bad shapes return MFNET_ERR_ARG with a message (no device is touched)."""
    from ssrs_amd import _lib
    L = _lib.lib()
    rc = L.mfnet_gemm_bf16(None, 0, None, 0, 4, 16, 64, None, None, 0, None, 0, 0, 1, 0, 1.0, 0, None, 0, None, 0,
                           None, 0, 0, None, 1, 16, 1, None)
    assert rc == -1 and b"null" in L.mfnet_last_error()
    rc = L.mfnet_cast(1, 0, 1, 1, 7, None)
    assert rc == -1 and b"multiple of 8" in L.mfnet_last_error()
    with pytest.raises(_lib.MfnetError):
        _lib.call("mfnet_attn_fwd", 1, 1, 1, 1, 1, 1, 16, 16, 12, 32, 14, 27, 27, 64, 0.125, None)   # head dim 32: only 64 and 80 exist
    rc = L.mfnet_gemm_bf16_grouped(16, 64, 16, 64, 4, 128, 64, 64, None, 0, 64, 2, 0, None, 0, None, 0, None, 0, 16, 0, 64, None)
    assert rc == -1 and b"group strides" in L.mfnet_last_error()
    # weight-gradient GEMM: N / K must be multiples of 8; split count helper is pure host code
    rc = L.mfnet_gemm_tn_bf16(16, 64, 16, 64, 128, 60, 64, 16, 64, 1, None, None)
    assert rc == -1 and b"multiples of 8" in L.mfnet_last_error()
    assert L.mfnet_gemm_tn_splits(65536, 64, 576) >= 2 and L.mfnet_gemm_tn_splits(100, 64, 64) == 1
    # gather: a segment with a null pointer / an accumulate into bf16 is refused
    desc = (ctypes.c_int64 * 17)(0, 16, 1, 1, 1, 4, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0x3f800000)
    assert L.mfnet_gather_segments(ctypes.cast(desc, ctypes.c_void_p), 1, 0, 0, None) == -1 and b"null" in L.mfnet_last_error()
    assert L.mfnet_gather_segments(ctypes.cast(desc, ctypes.c_void_p), 1, 1, 1, None) == -1
    # deterministic column reductions need their workspace
    assert L.mfnet_colreduce_workspace_floats(1, 65536, 64) == 128 * 2 * 64
    rc = L.mfnet_colsum(16, 0, 16, None, 1, 1024, 64, None, None, None)
    assert rc == -1 and b"workspace" in L.mfnet_last_error()


def test_product_rejects_cpu_tensors():
    import torch
    from ssrs_amd import ops
    with pytest.raises(_libErr()):
        ops.layernorm_fwd(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-6)


def _libErr():
    from ssrs_amd._lib import MfnetError
    return MfnetError
