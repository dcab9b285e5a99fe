"""Helpers for the -m gpu tests: torch only carries device memory; every
computation under test goes through the C ABI (inferflow_amd._capi)."""
import ctypes as C

import numpy as np
import torch

import inferflow_amd as ia
from inferflow_amd import dtypes as dt

L = None


def capi():
    global L
    if L is None:
        L = ia.lib()
    return L


def dev(a):
    """numpy -> cuda tensor (float16 arrays travel as int16 bit patterns safely)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        return torch.from_numpy(a.view(np.int16)).cuda().view(torch.float16)
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).cuda()
    return torch.from_numpy(a).cuda()


def host(t):
    t = t.detach().cpu()
    if t.dtype == torch.float16:
        return t.view(torch.int16).numpy().view(np.float16)
    return t.numpy()


_KEEP = []


def p(t):
    """Device pointer of a tensor.  The tensor is kept alive for a while so that
    `p(dev(x))` on a temporary cannot be recycled by torch's caching allocator
    before the (asynchronous) kernel has read it."""
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 256:
        del _KEEP[:128]
    return C.c_void_p(t.data_ptr())


def sync():
    torch.cuda.synchronize()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def empty_u8(*shape):
    return torch.empty(shape, dtype=torch.uint8, device="cuda")


def empty_f16(*shape):
    return torch.empty(shape, dtype=torch.float16, device="cuda")


def quantize(dtype, src):
    rows, cols = src.shape
    out = empty_u8(rows, dt.row_bytes(dtype, cols))
    fn = capi().ifa_quantize_f32 if src.dtype == torch.float32 else capi().ifa_quantize
    ia.check(fn(dtype, p(src), rows, cols, p(out), stream()))
    return out


def dequantize(dtype, packed, cols):
    rows = packed.shape[0]
    out = empty_f16(rows, cols)
    ia.check(capi().ifa_dequantize(dtype, p(packed), rows, cols, p(out), stream()))
    return out


def quantize_act(x):
    rows, cols = x.shape
    out = torch.zeros((rows, (cols + 31) // 32 * 34), dtype=torch.uint8, device="cuda")
    ia.check(capi().ifa_quantize_act_q8(p(x), rows, cols, p(out), stream()))
    return out


def gemv(w_dtype, W, rows, cols, x, x_dtype, bias=None):
    y = empty_f16(rows)
    ia.check(capi().ifa_gemv(w_dtype, p(W), rows, cols, x_dtype, p(x), p(bias), p(y), stream()))
    return y


def repack(dtype, W, rows, cols):
    out = torch.zeros((rows, capi().ifa_tiled_row_bytes(dtype, cols)), dtype=torch.uint8, device="cuda")
    ia.check(capi().ifa_repack_weights(dtype, p(W), rows, cols, p(out), stream()))
    return out


def gemv_tiled(w_dtype, Wt, rows, cols, xq, bias=None):
    y = empty_f16(rows)
    ia.check(capi().ifa_gemv_tiled(w_dtype, p(Wt), rows, cols, p(xq), p(bias), p(y), stream()))
    return y


def half_ulp_diff(a, b):
    """distance in half ulps between two float16 arrays (monotone integer map)."""
    def key(x):
        u = np.ascontiguousarray(x).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - u, u)
    return np.abs(key(a) - key(b))


def gemm(w_dtype, W, rows, cols, x, bias=None):
    tokens = x.shape[0]
    y = empty_f16(tokens, rows)
    ia.check(capi().ifa_gemm(w_dtype, p(W), rows, cols, p(x), tokens, p(bias), p(y), stream()))
    return y
