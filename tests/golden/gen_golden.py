#!/usr/bin/env python3
"""Generate tests/golden/block_codecs.npz from the REFERENCE's own codecs.

Run in the build container (needs /root/reference, or IFA_REFERENCE=<path>):
    python tests/golden/gen_golden.py
It compiles oracle/_ref/libifa_ref_quant.so from the reference's
src/common/quantization.h (see oracle/ref_quant_wrap.cc -- the reference is
included where it lies, never copied) and records, for seeded and adversarial
input rows, the packed block bytes, the dequantised values (F16 and F32) and
the GetInt4 words the reference produces.  The fixture holds data only.

F1: per-format block fixtures.   F2: Q8_B32T2 host quantizer
(Quantization::QuantizeRow_Q8_B32T2); the device activation quantizer
(tensor_quant.h:44-82) cannot be run here (CUDA) -- its fixture is the oracle's
restatement, cross-checked against the host routine on rows where both agree.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle as o  # noqa: E402


def make_rows(cols, seed):
    rng = np.random.default_rng(seed)
    parts = [rng.normal(0, 0.05, (8, cols)), rng.normal(0, 1.0, (4, cols)),
             np.full((1, cols), 0.125), np.zeros((1, cols))]
    x = rng.normal(0, 0.02, (2, cols)); x[0, 5] = 30.0; x[1, 7] = -65504.0
    parts.append(x)
    parts.append(rng.uniform(-1.4, 1.4, (4, cols)))
    parts.append(rng.normal(0, 1e-4, (2, cols)))
    ramp = np.linspace(-3, 3, cols)[None, :].repeat(2, 0); ramp[1] = ramp[1][::-1]
    parts.append(ramp)
    return np.concatenate(parts).astype(np.float16)


def main():
    assert o.ref_lib() is not None, "reference not available: cannot generate goldens"
    cols = 256
    src = make_rows(cols, 20240611)
    src_q4b16 = np.clip(src.astype(np.float32), -0.9, 1.4).astype(np.float16)
    out = {"cols": np.int32(cols), "src_f16": src.view(np.uint16), "src_q4b16_f16": src_q4b16.view(np.uint16)}
    rng = np.random.default_rng(7)
    f = np.concatenate([rng.normal(0, 1, 4000), rng.normal(0, 1e-5, 4000), rng.uniform(-70000, 70000, 4000),
                        [65504, 65519.9, 65520, 1e-8, 2.0 ** -25, 2.0 ** -24, -0.0, 8.94e-8]]).astype(np.float32)
    out["f2h_in"] = f
    out["f2h_out"] = o.ref_f2h(f).view(np.uint16)
    for dt in o.QUANT_DTYPES:
        name = o.DTYPE_NAMES[dt]
        s = src_q4b16 if dt == o.Q4_B16 else src
        packed = o.ref_quantize(dt, s)
        out["packed_" + name] = packed
        out["deq16_" + name] = o.ref_dequantize(dt, packed, cols).view(np.uint16)
        out["deq32_" + name] = o.ref_dequantize(dt, packed, cols, out_f32=True)
        if dt in o.GETINT4_DTYPES:
            out["int4_" + name] = o.ref_get_int4(dt, packed, cols)
        if dt != o.Q8_B32T2:
            s32 = (s.astype(np.float32) * np.float32(1.0001)).astype(np.float32)
            out["packed32_" + name] = o.ref_quantize(dt, s32)
    path = os.path.join(os.path.dirname(__file__), "block_codecs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
