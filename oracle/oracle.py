"""numpy-facing wrappers for the CPU oracle (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# dtype ids == reference ElementType enum (src/tensor/tensor_common.h:15-42)
F32, F16 = 0, 1
Q8_B32T1, Q8_B32T2, Q6_B64T1, Q5_B64T1, Q5_B32T1 = 7, 8, 9, 10, 11
Q4_B16, Q4_B32T1A, Q4_B32T1B, Q4_B64T1, Q3H_B64T1 = 12, 13, 14, 17, 18
Q3_B32T1A, Q3_B32T1B, Q2_B32T1A, Q2_B32T1B = 19, 20, 21, 22

DTYPE_NAMES = {
    F32: "f32", F16: "f16", Q8_B32T1: "q8_b32t1", Q8_B32T2: "q8_b32t2", Q6_B64T1: "q6_b64t1",
    Q5_B64T1: "q5_b64t1", Q5_B32T1: "q5_b32t1", Q4_B16: "q4_b16", Q4_B32T1A: "q4_b32t1a",
    Q4_B32T1B: "q4_b32t1b", Q4_B64T1: "q4_b64t1", Q3H_B64T1: "q3h_b64t1", Q3_B32T1A: "q3_b32t1a",
    Q3_B32T1B: "q3_b32t1b", Q2_B32T1A: "q2_b32t1a", Q2_B32T1B: "q2_b32t1b",
}
QUANT_DTYPES = [k for k in DTYPE_NAMES if k >= 7]
AX8_DTYPES = [Q8_B32T2, Q6_B64T1, Q5_B64T1, Q4_B32T1A, Q4_B32T1B, Q4_B64T1, Q3H_B64T1]
GETINT4_DTYPES = [Q8_B32T2, Q6_B64T1, Q5_B64T1, Q4_B32T1A, Q4_B32T1B, Q4_B64T1, Q3H_B64T1]

# tensor ids (ifa_oracle.h)
T_EMBD, T_OUT_NORM, T_OUT_NORM_B, T_LM_HEAD = 0, 1, 2, 3
T_ATTN_NORM, T_ATTN_NORM_B, T_WQ, T_WK, T_WV, T_WO = 10, 11, 12, 13, 14, 15
T_FFN_NORM, T_FFN_NORM_B, T_W1, T_W2, T_W3, T_MOE_GATE = 16, 17, 18, 19, 20, 21
T_WQ_B, T_WK_B, T_WV_B, T_WO_B, T_W1_B, T_W2_B, T_W3_B = 22, 23, 24, 25, 26, 27, 28
T_ATTN_POST_NORM, T_ATTN_POST_NORM_B, T_FFN_POST_NORM, T_FFN_POST_NORM_B = 29, 30, 31, 32


def build(force=False):
    """Compile oracle/libifa_oracle.so (and _ref when /root/reference exists)."""
    so = os.path.join(_HERE, "libifa_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ifa_oracle.c", "ifa_oracle_model.c", "ifa_oracle.h")]
    stale = force or not os.path.exists(so) or any(
        os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    ref_so = os.path.join(_HERE, "_ref", "libifa_ref_quant.so")
    ref_src = os.environ.get("IFA_REFERENCE", "/root/reference")
    need_ref = os.path.exists(os.path.join(ref_src, "src/common/quantization.h")) and (
        force or not os.path.exists(ref_so) or not os.path.exists(os.path.join(_HERE, "_ref", "ifa_ref_sampling"))
        or not os.path.exists(os.path.join(_HERE, "_ref", "ifa_ref_moe_rows"))
        or os.path.getmtime(os.path.join(_HERE, "ref_quant_wrap.cc")) > os.path.getmtime(ref_so))
    if stale or need_ref:
        subprocess.check_call(["make", "-C", _HERE, "REF=" + ref_src], stdout=subprocess.DEVNULL)
    return so


def usable_cpus():
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota (a container may see 256
    logical CPUs and own 16 of them; OpenMP's default of one thread per visible CPU then runs 16x oversubscribed)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        # IFA_ORACLE_LIB: another build of the SAME sources (bench.py times an -O3 -march=native build made on the box it runs on;
        # the parity tests always use the committed recipe: -O2, no -march, no FMA contraction)
        so = os.environ.get("IFA_ORACLE_LIB") or build()
        L = C.CDLL(so)
        L.orc_set_num_threads(C.c_int(usable_cpus()))      # the default for every OpenMP loop of the oracle
        L.orc_block_capacity.restype = C.c_int
        L.orc_block_bytes.restype = C.c_int
        L.orc_row_bytes.restype = C.c_size_t
        L.orc_row_bytes.argtypes = [C.c_int, C.c_size_t]
        L.orc_model_create.restype = C.c_void_p
        L.orc_model_last_hidden.restype = C.c_void_p
        L.orc_model_kv_cache.restype = C.c_void_p
        L.orc_model_last_moe_margin.restype = C.c_float
        _lib = L
    return _lib


def ref_lib():
    """The reference's own codecs (None when not built, e.g. no /root/reference)."""
    global _ref
    if _ref is None:
        build()
        p = os.path.join(_HERE, "_ref", "libifa_ref_quant.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f16(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        return a.view(np.uint16)
    assert a.dtype == np.uint16
    return a


def block_capacity(dt):
    return lib().orc_block_capacity(C.c_int(dt))


def block_bytes(dt):
    return lib().orc_block_bytes(C.c_int(dt))


def row_bytes(dt, cols):
    return lib().orc_row_bytes(dt, cols)


def f2h(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, np.uint16)
    lib().orc_f2h_n(_p(x), _p(out), C.c_size_t(x.size))
    return out.view(np.float16)


def h2f(x):
    x = _f16(x)
    out = np.empty(x.shape, np.float32)
    lib().orc_h2f_n(_p(x), _p(out), C.c_size_t(x.size))
    return out


def quantize(dt, src):
    """src: [rows][cols] float16 (or float32) -> uint8 [rows][row_bytes]."""
    src = np.ascontiguousarray(src)
    rows, cols = src.shape
    out = np.zeros((rows, row_bytes(dt, cols)), np.uint8)
    if src.dtype == np.float32:
        rc = lib().orc_quantize_rows_f32(dt, _p(src), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    else:
        rc = lib().orc_quantize_rows(dt, _p(_f16(src)), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    if rc != 0:
        raise ValueError("orc_quantize_rows failed (dtype %d, cols %d)" % (dt, cols))
    return out


def dequantize(dt, packed, cols, out_f32=False):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    rows = packed.shape[0]
    if out_f32:
        out = np.empty((rows, cols), np.float32)
        rc = lib().orc_dequantize_rows_f32(dt, _p(packed), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    else:
        o16 = np.empty((rows, cols), np.uint16)
        rc = lib().orc_dequantize_rows(dt, _p(packed), C.c_size_t(rows), C.c_size_t(cols), _p(o16))
        out = o16.view(np.float16)
    if rc != 0:
        raise ValueError("orc_dequantize_rows failed")
    return out


def unpack_codes(dt, packed, cols):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    rows = packed.shape[0]
    out = np.empty((rows, cols), np.int32)
    rc = lib().orc_unpack_codes(dt, _p(packed), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    if rc != 0:
        raise ValueError("orc_unpack_codes failed")
    return out


def codes_to_int4_words(codes):
    """Pack consecutive 4 codes as int8 lanes of one int32 (what GetInt4 returns)."""
    c = (codes.astype(np.int64) & 0xFF).reshape(codes.shape[0], -1, 4)
    w = c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16) | (c[..., 3] << 24)
    return w.astype(np.uint32).view(np.int32)


def quantize_act_q8(x):
    x = np.ascontiguousarray(x)
    if x.ndim == 1:
        x = x[None, :]
    rows, cols = x.shape
    out = np.zeros((rows, (cols + 31) // 32 * 34), np.uint8)
    lib().orc_quantize_act_q8(_p(_f16(x)), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    return out


def quantize_q8_b32t2_host(x):
    x = np.ascontiguousarray(x)
    rows, cols = x.shape
    out = np.zeros((rows, cols // 32 * 34), np.uint8)
    rc = lib().orc_quantize_q8_b32t2_host(_p(_f16(x)), C.c_size_t(rows), C.c_size_t(cols), _p(out))
    assert rc == 0
    return out


def gemv_ax8(dt, W, rows, cols, xq8, want_f64=False):
    W = np.ascontiguousarray(W, dtype=np.uint8)
    xq8 = np.ascontiguousarray(xq8, dtype=np.uint8)
    y = np.empty(rows, np.uint16)
    y64 = np.empty(rows, np.float64) if want_f64 else None
    rc = lib().orc_gemv_ax8(dt, _p(W), C.c_size_t(rows), C.c_size_t(cols), _p(xq8), _p(y),
                            _p(y64) if want_f64 else None)
    if rc != 0:
        raise ValueError("orc_gemv_ax8 failed")
    return (y.view(np.float16), y64) if want_f64 else y.view(np.float16)


def gemv_f16x(dt, W, rows, cols, x, bias=None, want_f64=False):
    W = np.ascontiguousarray(W)
    x16 = _f16(x)
    y = np.empty(rows, np.uint16)
    y64 = np.empty(rows, np.float64) if want_f64 else None
    b16 = _f16(bias) if bias is not None else None
    rc = lib().orc_gemv_f16x(dt, _p(W), C.c_size_t(rows), C.c_size_t(cols), _p(x16),
                             _p(b16) if b16 is not None else None, _p(y),
                             _p(y64) if want_f64 else None)
    if rc != 0:
        raise ValueError("orc_gemv_f16x failed")
    return (y.view(np.float16), y64) if want_f64 else y.view(np.float16)


def gemm_f16x(dt, W, rows, cols, X, bias=None):
    """Y[t] = gemv_f16x(W, X[t]) for all rows of X at once (each weight row dequantised once; bit-identical to the per-row calls)"""
    W = np.ascontiguousarray(W)
    x16 = _f16(X)
    T = x16.shape[0]
    y = np.empty((T, rows), np.uint16)
    b16 = _f16(bias) if bias is not None else None
    rc = lib().orc_gemm_f16x(dt, _p(W), C.c_size_t(rows), C.c_size_t(cols), _p(x16), C.c_size_t(T), _p(b16) if b16 is not None else None, _p(y))
    if rc != 0:
        raise ValueError("orc_gemm_f16x failed")
    return y.view(np.float16)


def rmsnorm(x, w=None, b=None, multi_base=0.0, eps=1e-5, nthreads_x=128):
    x16 = _f16(x)
    rows, cols = x16.shape
    y = np.empty((rows, cols), np.uint16)
    lib().orc_rmsnorm(_p(x16), C.c_size_t(rows), C.c_size_t(cols),
                      _p(_f16(w)) if w is not None else None, _p(_f16(b)) if b is not None else None,
                      C.c_float(multi_base), C.c_float(eps), C.c_int(nthreads_x), _p(y))
    return y.view(np.float16)


def stdnorm(x, w=None, b=None, eps=1e-5, nthreads_x=128):
    x16 = _f16(x)
    rows, cols = x16.shape
    y = np.empty((rows, cols), np.uint16)
    lib().orc_stdnorm(_p(x16), C.c_size_t(rows), C.c_size_t(cols),
                      _p(_f16(w)) if w is not None else None, _p(_f16(b)) if b is not None else None,
                      C.c_float(eps), C.c_int(nthreads_x), _p(y))
    return y.view(np.float16)


def rope(x, pos0, theta=10000.0, order=2, partial_rotary=1.0):
    """x: [tokens][heads][head_dim] float16; returns rotated copy."""
    x16 = _f16(x).copy()
    tokens, heads, hd = x16.shape
    rd = int(hd * partial_rotary + 0.5)
    lib().orc_rope(_p(x16), hd, heads, tokens, pos0, C.c_float(theta), order, rd, rd)
    return x16.view(np.float16)


def softmax(s, prefix_len=-1, scale=1.0):
    """s: [cz][cy][cx] float16."""
    s16 = _f16(s).copy()
    cz, cy, cx = s16.shape
    lib().orc_softmax(_p(s16), cx, cy, cz, prefix_len, C.c_float(scale))
    return s16.view(np.float16)


def act(x, kind=0, is_glu=False):
    x16 = _f16(x)
    rows, cols = x16.shape
    n = cols // 2 if is_glu else cols
    y = np.empty((rows, n), np.uint16)
    lib().orc_act(_p(x16), C.c_size_t(rows), C.c_size_t(n), kind, 1 if is_glu else 0, _p(y))
    return y.view(np.float16)


def add(a, b, b_period=0):
    a16, b16 = _f16(a), _f16(b)
    c = np.empty(a16.shape, np.uint16)
    lib().orc_add(_p(a16), _p(b16), C.c_size_t(a16.size), C.c_size_t(b_period), _p(c))
    return c.view(np.float16)


def mul(a, b):
    a16, b16 = _f16(a), _f16(b)
    c = np.empty(a16.shape, np.uint16)
    lib().orc_mul(_p(a16), _p(b16), C.c_size_t(a16.size), _p(c))
    return c.view(np.float16)


def scale(a, s):
    a16 = _f16(a)
    c = np.empty(a16.shape, np.uint16)
    lib().orc_scale(_p(a16), C.c_float(s), C.c_size_t(a16.size), _p(c))
    return c.view(np.float16)


def attention(q, kcache, vcache, kv_dtype, n_ctx, prefix_len, heads, kv_heads, head_dim,
              kq_scale=1.0, use_alibi=False, alibi_base_head=0, alibi_total_heads=None):
    """q: [q_tokens][heads][head_dim] f16; caches: f16 [n][kv_dim] or uint8 Q8 rows."""
    q16 = _f16(q)
    qt = q16.shape[0]
    kc = np.ascontiguousarray(kcache)
    vc = np.ascontiguousarray(vcache)
    out = np.empty((qt, heads * head_dim), np.uint16)
    lib().orc_attention(_p(q16), _p(kc), _p(vc), kv_dtype, n_ctx, qt, prefix_len, heads, kv_heads,
                        head_dim, C.c_float(kq_scale), 1 if use_alibi else 0, alibi_base_head,
                        alibi_total_heads or heads, _p(out))
    return out.view(np.float16)


def moe_topk(probs, top_k, norm=True):
    probs = np.ascontiguousarray(probs, np.float32)
    idx = np.zeros(8, np.int32)
    w = np.zeros(8, np.float32)
    n = lib().orc_moe_topk(_p(probs), probs.size, top_k, 1 if norm else 0, _p(idx), _p(w))
    return idx[:n].copy(), w[:n].copy()


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "dim", "layers", "heads", "kv_heads", "head_dim", "ffn", "vocab", "max_ctx",
        "norm_kind", "act_kind", "is_glu", "rope_order", "use_alibi", "parallel_attn",
        "share_input")] + [(n, C.c_float) for n in (
            "rope_theta", "partial_rotary", "kq_scale", "eps")] + [(n, C.c_int) for n in (
                "kv_dtype", "full_quant_gemv", "experts", "moe_top_k", "moe_norm_topk")] + [(n, C.c_float) for n in (
                    "attn_norm_base", "ffn_norm_base", "out_norm_base", "attn_out_scale", "ffn_out_scale", "out_scale", "embd_scale")] + [("unk_id", C.c_int), ("tp_merge", C.c_int)]


class Model:
    """Whole-model oracle (ifa_oracle_model.c)."""

    def __init__(self, **kw):
        cfg = ModelCfg()
        defaults = dict(norm_kind=0, act_kind=0, is_glu=1, rope_order=2, use_alibi=0, parallel_attn=0,
                        share_input=0, rope_theta=10000.0, partial_rotary=1.0, kq_scale=1.0, eps=1e-5,
                        kv_dtype=F16, full_quant_gemv=1, experts=0, moe_top_k=0, moe_norm_topk=1,
                        attn_norm_base=0.0, ffn_norm_base=0.0, out_norm_base=0.0, attn_out_scale=1.0, ffn_out_scale=1.0,
                        out_scale=1.0, embd_scale=0.0, unk_id=-1, tp_merge=1)
        defaults.update(kw)
        for k, v in defaults.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self._keep = []
        self._h = C.c_void_p(lib().orc_model_create(C.byref(cfg)))

    def set_tensor(self, layer, tid, dtype, data, rows, cols, expert=-1):
        data = np.ascontiguousarray(data)
        self._keep.append(data)
        rc = lib().orc_model_set_tensor(self._h, layer, tid, expert, dtype, _p(data),
                                        C.c_size_t(rows), C.c_size_t(cols))
        assert rc == 0, "orc_model_set_tensor failed"

    def reset(self):
        lib().orc_model_reset(self._h)

    def set_attn_post_as_residual(self, on):
        """ModelSpec::is_attn_post_as_residual (model.h:113): the FFN's residual is the attention post-norm's output (default) or its input."""
        lib().orc_model_set_attn_post_as_residual(self._h, int(bool(on)))

    def forward(self, tokens, prefix_len, want_logits=True, nthreads=0):
        toks = np.ascontiguousarray(tokens, np.int32)
        T = toks.size
        logits = np.empty((T, self.cfg.vocab), np.uint16) if want_logits else None
        tok = lib().orc_model_forward(self._h, _p(toks), T, prefix_len,
                                      _p(logits) if want_logits else None, nthreads)
        if tok < 0:
            raise RuntimeError("orc_model_forward failed: %d" % tok)
        return tok, (logits.view(np.float16) if want_logits else None)

    def capture_layers(self, on=True):
        """every forward() records the last row's input of each layer + the last layer's output: layer_io() -> [layers + 1][dim];
        and every layer's router margin (layer_margins(): 2.0 for dense layers)"""
        self._cap = np.zeros((self.cfg.layers + 1, self.cfg.dim), np.uint16) if on else None
        self._mar = np.full(self.cfg.layers, 2.0, np.float32) if on else None
        lib().orc_model_set_capture(self._h, _p(self._cap) if on else None)
        lib().orc_model_set_layer_margins(self._h, _p(self._mar) if on else None)

    def layer_margins(self):
        return self._mar.copy()

    def layer_io(self):
        return self._cap.view(np.float16).copy()

    def kv_rows(self, layer, is_v, n_rows):
        """rows [0, n_rows) of a layer's K (or V) cache as the forwards left them: uint8 [n_rows][row bytes of the cache format]"""
        kvd = self.cfg.kv_heads * self.cfg.head_dim
        rb = row_bytes(Q8_B32T2 if self.cfg.kv_dtype == Q8_B32T2 else F16, kvd)
        p = lib().orc_model_kv_cache(self._h, C.c_int(layer), C.c_int(1 if is_v else 0))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n_rows * rb,)).reshape(n_rows, rb).copy()

    def moe_margin(self):
        """smallest gap between the last selected and the first rejected router probability in the last forward() (2.0: no MoE layer)"""
        return float(lib().orc_model_last_moe_margin(self._h))

    def last_hidden(self):
        p = lib().orc_model_last_hidden(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), (self.cfg.dim,)).copy().view(np.float16)

    def __del__(self):
        try:
            if self._h:
                lib().orc_model_destroy(self._h)
                self._h = None
        except Exception:
            pass


# ---- reference-backed helpers (only when oracle/_ref exists) -------------
def ref_quantize(dt, src):
    R = ref_lib()
    src = np.ascontiguousarray(src)
    rows, cols = src.shape
    rb = row_bytes(dt, cols)
    assert R.ref_block_bytes(dt) == block_bytes(dt)
    out = np.zeros((rows, rb), np.uint8)
    if src.dtype == np.float32:
        rc = R.ref_quantize_rows_f32(dt, _p(src), rows, cols, _p(out), rb)
    else:
        rc = R.ref_quantize_rows_f16(dt, _p(_f16(src)), rows, cols, _p(out), rb)
    assert rc == 0
    return out


def ref_dequantize(dt, packed, cols, out_f32=False):
    R = ref_lib()
    packed = np.ascontiguousarray(packed, np.uint8)
    rows, rb = packed.shape
    if out_f32:
        out = np.empty((rows, cols), np.float32)
        rc = R.ref_dequantize_rows_f32(dt, _p(packed), rows, cols, _p(out), rb)
        assert rc == 0
        return out
    out = np.empty((rows, cols), np.uint16)
    rc = R.ref_dequantize_rows_f16(dt, _p(packed), rows, cols, _p(out), rb)
    assert rc == 0
    return out.view(np.float16)


def ref_get_int4(dt, packed, cols):
    R = ref_lib()
    packed = np.ascontiguousarray(packed, np.uint8)
    rows, rb = packed.shape
    out = np.empty((rows, cols // 4), np.int32)
    rc = R.ref_get_int4_rows(dt, _p(packed), rows, cols, _p(out), rb)
    assert rc == 0
    return out


def ref_f2h(x):
    R = ref_lib()
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint16)
    R.ref_f2h(_p(x), _p(out), x.size)
    return out.view(np.float16)
