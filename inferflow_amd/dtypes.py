"""Element types: numeric ids follow the reference's ElementType enum
(src/tensor/tensor_common.h:15-42) and the names its .ini files use
(TensorCommon::InitElementTypeMap, src/tensor/tensor_common.cc:171-205)."""
F32, F16 = 0, 1
Q8_B32T1, Q8_B32T2, Q6_B64T1, Q5_B64T1, Q5_B32T1 = 7, 8, 9, 10, 11
Q4_B16, Q4_B32T1A, Q4_B32T1B, Q4_B64T1, Q3H_B64T1 = 12, 13, 14, 17, 18
Q3_B32T1A, Q3_B32T1B, Q2_B32T1A, Q2_B32T1B = 19, 20, 21, 22

NAMES = {
    F32: "f32", F16: "f16", Q8_B32T1: "q8_b32t1", Q8_B32T2: "q8_b32t2", Q6_B64T1: "q6_b64t1",
    Q5_B64T1: "q5_b64t1", Q5_B32T1: "q5_b32t1", Q4_B16: "q4_b16", Q4_B32T1A: "q4_b32t1a",
    Q4_B32T1B: "q4_b32t1b", Q4_B64T1: "q4_b64t1", Q3H_B64T1: "q3h_b64t1", Q3_B32T1A: "q3_b32t1a",
    Q3_B32T1B: "q3_b32t1b", Q2_B32T1A: "q2_b32t1a", Q2_B32T1B: "q2_b32t1b",
}
def name(dt):
    return NAMES.get(dt, "dtype%d" % dt).upper()


QUANT = [k for k in NAMES if k >= 7]
AX8 = [Q8_B32T2, Q6_B64T1, Q5_B64T1, Q4_B32T1A, Q4_B32T1B, Q4_B64T1, Q3H_B64T1]

_CAP = {F32: 1, F16: 1, Q4_B16: 16, Q6_B64T1: 64, Q5_B64T1: 64, Q4_B64T1: 64, Q3H_B64T1: 64}
_BYTES = {F32: 4, F16: 2, Q8_B32T1: 36, Q8_B32T2: 34, Q6_B64T1: 52, Q5_B64T1: 44, Q5_B32T1: 24, Q4_B16: 10,
          Q4_B32T1A: 20, Q4_B32T1B: 20, Q4_B64T1: 36, Q3H_B64T1: 32, Q3_B32T1A: 16, Q3_B32T1B: 16,
          Q2_B32T1A: 12, Q2_B32T1B: 12}


def block_capacity(dt):
    return _CAP.get(dt, 32)


def block_bytes(dt):
    return _BYTES[dt]


def row_bytes(dt, cols):
    c = block_capacity(dt)
    return (cols + c - 1) // c * block_bytes(dt)


def streamed_row_bytes(dt, cols):
    """Bytes of one weight row in the layout the fused decode kernels STREAM (csrc/ifa_tiled.h): the reference block
    bytes, except Q3H_B64T1 whose pair codes are expanded to nibble pairs at load time (36 instead of 32 per 64 weights)."""
    c = block_capacity(dt)
    per_block = 36 if dt == Q3H_B64T1 else block_bytes(dt)
    return ((cols + c - 1) // c * per_block + 15) // 16 * 16 if dt in AX8 else row_bytes(dt, cols)
