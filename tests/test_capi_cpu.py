"""CPU-only checks of the C-ABI library: it loads, exports every symbol that
include/inferflow_amd.h declares, and its registry matches the oracle's."""
import os
import re

import pytest

import inferflow_amd as ia
from inferflow_amd import _capi, dtypes as dt
import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="inferflow_amd.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ifa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ia.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "missing symbol " + n


def test_ctypes_signatures_cover_header():
    assert sorted(_capi.SIGNATURES) == _declared()


def test_engine_header_is_exported_and_bound():
    names = _declared("inferflow_engine.h")
    assert sorted(_capi.ENGINE_SIGNATURES) == names and len(names) >= 10
    L = ia.lib()
    for n in names:
        assert hasattr(L, n), "missing symbol " + n


def test_registry_matches_oracle():
    L = ia.lib()
    for d in dt.NAMES:
        assert L.ifa_block_capacity(d) == o.block_capacity(d) == dt.block_capacity(d)
        assert L.ifa_block_bytes(d) == o.block_bytes(d) == dt.block_bytes(d)
        for cols in (64, 4096, 11008):
            assert L.ifa_row_bytes(d, cols) == o.row_bytes(d, cols) == dt.row_bytes(d, cols)
    assert L.ifa_block_capacity(99) == 0


@pytest.mark.parametrize("name,expect", [
    ("q4", dt.Q4_B32T1A), ("Q3H", dt.Q3H_B64T1), ("q8", dt.Q8_B32T2), ("q3", dt.Q3_B32T1B),
    ("q2", dt.Q2_B32T1B), ("fp16", dt.F16), ("q6", dt.Q6_B64T1), ("q5", dt.Q5_B64T1),
    ("q4_b32t1", dt.Q4_B32T1A), ("nope", -1)])
def test_dtype_names(name, expect):
    # TensorCommon::InitElementTypeMap, src/tensor/tensor_common.cc:171-205
    assert ia.lib().ifa_dtype_from_name(name.encode()) == expect


def test_errors_are_codes_not_exceptions():
    L = ia.lib()
    # argument validation happens before any HIP call, so this is safe without a GPU
    assert L.ifa_quantize(dt.Q4_B32T1A, None, 1, 32, None, None) == -1
    assert b"null" in L.ifa_last_error()
    assert L.ifa_gemv(dt.F32, 1, 1, 32, dt.F16, 1, None, 1, None) == -1
    assert L.ifa_device_count() >= 0
