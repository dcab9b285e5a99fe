// ifa_engine_decode.hip -- the fused batch-1 decode step (GpuInferenceWorker::ProcessGpuLayer at T = 1, inference_worker.cc:762-981):
// the parameters of every fused launch, the step as a captured graph, ifa_model_decode and the per-kernel timing entry point.
#include "ifa_engine_state.h"

namespace ifae {

// keys split over workgroups past attn_split_ctx; 8 splits per head up to 2K keys, 16 up to 8K, 32 beyond (a head's K / V
// history streams through that many CUs: 8 splits left 16K-key contexts at 2 TB/s).  The captured steps hold the choice.
bool qkv_attn_layer_ok(const ifa_model *m, int l, int *gk_out);

// From how many keys on a head's keys are split over workgroups.  attn_split_ctx >= 0: that many (0: never).  -1 (default): by what
// the one-workgroup attention is -- as the tail of the QKV launch with the heads' workgroups unloaded in the 256-row bucket (attn_unload)
// and the rows past the bucket pulled towards the L2 by the idle waves, it holds up to ~520 keys with one kv head per query head
// (400 / 450 / 500 / 560 keys 689 / 674 / 661 / 635 tok/s against 658 / 655 / 653 / 651 split; Q8 cache 662 / 653 / 643 / 620 against
// 642 / 639 / 638 / 635) and ~650 with grouped queries (Mixtral 560 keys 388 against 381): profiles/r06_attn_unload.log; 320 otherwise
static int attn_split_threshold(const ifa_model *m)
{
    if (m->opt_attn_split_ctx >= 0) return m->opt_attn_split_ctx;
    const ifa_model_config &c = m->cfg;
    bool tail = m->opt_attn_unload && m->opt_fuse_attn && waits_enabled() && dec_attn_smem(c.head_dim, c.max_ctx, 256) <= IFA_LDS_LIMIT;
    int gk = 0;
    for (int l = 0; tail && l < c.layers; l++) if (!qkv_attn_layer_ok(m, l, &gk)) tail = false;
    if (tail && (long long)c.kv_heads * gk > (long long)visible_cus()) tail = false;
    if (!tail) return 320;
    return c.heads > c.kv_heads ? 640 : 512;
}

void choose_attn_split(ifa_model *m, int reach)
{
    // splits per head by the context the call reaches: 8 up to 8192 keys, 16 above (round 6, second session, after the kernels lost their
    // request chains: 2048 / 4096 / 8192 keys 563 / 484 / 404 tok/s with 8 splits against 535 / 483 / 400 with 16; 16384 keys 301 / 303 / 281
    // with 8 / 16 / 32; Q8 cache 4096 keys 487 against 465 -- profiles/r06_long_context_decode.log.  Was 8 / 16 / 32 from 512 / 2048 / 8192.)
    const int thr = attn_split_threshold(m);
    const int want = (thr > 0 && reach > thr) ? (reach > 8192 ? 16 : 8) : 0;
    if (want != m->attn_split) { m->attn_split = want; drop_graphs(m); }
    // rows of the K / V cache the one-workgroup kernel requests before it knows the position: the bucket this call stays
    // inside (a longer context only costs the direct loads of the rows past it)
    const int pb = reach <= 64 ? 64 : (reach <= 128 ? 128 : 256);
    if (pb != m->attn_pb) { m->attn_pb = pb; drop_graphs(m); }
}

// ------------------------------------------------------------------ dispatch
// reads one dword every `stride` bytes: warms the TLB / pulls lines towards L2+MALL
__global__ void __launch_bounds__(256) k_touch(const uint8_t *__restrict__ p, size_t bytes, size_t stride, int *sink)
{
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    int acc = 0;
    for (; i < bytes; i += (size_t)gridDim.x * blockDim.x * stride) acc += *reinterpret_cast<const int *>(p + i);
    if (acc == 0x7FFFFFFF) *sink = acc;
}

static long long *g_trace_ptr = nullptr;   // set by ifa_model_time_kernel when the "trace" option is on

bool fused_ok(const Tensor &t, bool long_rows)
{
    if (!t.present()) return false;
    if (fused_int8(t.dtype)) return t.tiled && (long_rows ? dec_gemv_supported_long(t.dtype, t.cols) : dec_gemv_supported(t.dtype, t.cols));
    return dec_gemv_h_supported(t.dtype, t.cols) && (long_rows || t.cols <= 8192);
}

template <int EPI, int NORM>
static int launch_dec_gemv(int w_dtype, const DecGemvParams &P, int wgs_per_cu_opt, hipStream_t s)
{
    if (!fused_int8(w_dtype) && w_dtype != Q3H_NATIVE) {
        if constexpr (NORM == 2 || epi_is_moe(EPI)) return ifa_fail(IFA_ERR_STATE, "fused GEMV: dtype %d has no kernel for this launch", w_dtype);
        else return dec_gemv_h_launch(w_dtype, EPI, NORM, P, s);
    }
    return dec_gemv_launch(w_dtype, EPI, NORM, P, wgs_per_cu_opt, s, g_trace_ptr);
}

int lmhead_grid(const DecLmHeadParams &P, int wgs_per_cu_opt)
{
    const int nj = (P.cols / 8 + 63) / 64;
    const int R = nj <= 4 ? 2 : 1;
    const int nbatch = (P.rows + R - 1) / R;
    const int per_cu = wgs_per_cu_opt > 0 ? wgs_per_cu_opt : 2;
    return std::max(1, std::min(num_cus() * per_cu, (nbatch + DEC_WAVES - 1) / DEC_WAVES));
}

int launch_lmhead(const DecLmHeadParams &P, int norm, int wgs_per_cu_opt, hipStream_t s, const DecStepTail *Z)
{
    const int chunks = P.cols / 8;
    const int nj = (chunks + 63) / 64;
    if (P.cols % 8 != 0 || nj < 1 || nj > 16) return ifa_fail(IFA_ERR_ARG, "fused lm_head supports cols %% 8 == 0 and <= 8192 (got %d)", P.cols);
    const int R = nj <= 4 ? 2 : 1;
    const int nbatch = (P.rows + R - 1) / R;
    (void)nbatch;
    dim3 grid((unsigned)lmhead_grid(P, wgs_per_cu_opt));
    const size_t smem = (((size_t)P.cols * 2 + 15) & ~(size_t)15) + 132 * 4 + 16;
#define IFA_LM(NJV, RV) \
    case NJV: if (norm) k_dec_lmhead_f16<NJV, RV, 1><<<grid, dim3(DEC_THREADS), smem, s>>>(P); \
              else k_dec_lmhead_f16<NJV, RV, 0><<<grid, dim3(DEC_THREADS), smem, s>>>(P); break;
#define IFA_LMT(NJV, RV) \
    case NJV: if (norm) k_dec_lmhead_tail<NJV, RV, 1><<<grid, dim3(DEC_THREADS), smem, s>>>(P, *Z); \
              else k_dec_lmhead_tail<NJV, RV, 0><<<grid, dim3(DEC_THREADS), smem, s>>>(P, *Z); break;
    if (Z) {
        switch (nj) { IFA_LMT(1, 2) IFA_LMT(2, 2) IFA_LMT(3, 2) IFA_LMT(4, 2) IFA_LMT(5, 1) IFA_LMT(6, 1) IFA_LMT(7, 1) IFA_LMT(8, 1)
                      IFA_LMT(9, 1) IFA_LMT(10, 1) IFA_LMT(11, 1) IFA_LMT(12, 1) IFA_LMT(13, 1) IFA_LMT(14, 1) IFA_LMT(15, 1) IFA_LMT(16, 1) }
    } else
    switch (nj) { IFA_LM(1, 2) IFA_LM(2, 2) IFA_LM(3, 2) IFA_LM(4, 2) IFA_LM(5, 1) IFA_LM(6, 1) IFA_LM(7, 1) IFA_LM(8, 1)
                  IFA_LM(9, 1) IFA_LM(10, 1) IFA_LM(11, 1) IFA_LM(12, 1) IFA_LM(13, 1) IFA_LM(14, 1) IFA_LM(15, 1) IFA_LM(16, 1) }
#undef IFA_LM
#undef IFA_LMT
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// self_attn.post_norm / feed_forward.post_norm (OPT / BERT-style specs): the op-by-op layer only (layer_tail_ops)
bool has_post_norms(const ifa_model *m)
{
    for (const Layer &L : m->layers) if (L.t[T_ATTN_POST_NORM].present() || L.t[T_FFN_POST_NORM].present()) return true;
    return false;
}

// Can the fused decode path run this model?  (otherwise decode falls back to forward())
bool fused_supported(const ifa_model *m, std::string *why)
{
    const ifa_model_config &c = m->cfg;
    auto fail = [&](const char *s) { if (why) *why = s; return false; };
    if (has_post_norms(m)) return fail("post norms (self_attn.post_norm / feed_forward.post_norm) use the op-by-op path");
    if (c.experts > 64 || (c.experts > 0 && (c.moe_top_k < 1 || c.moe_top_k > 8))) return fail("MoE: experts / top_k out of range");
    if ((scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale) || scale_on(c.out_scale)) && c.tp_size > 1)
        return fail("output scales (attn_out_scale / ffn_out_scale / out_scale) on a partitioned model use the op-by-op path");
    if (c.experts > 0 && (c.norm_kind != 0 || c.parallel_attn || c.share_input)) return fail("MoE layers need the sequential RMS-norm wiring");
    if (!c.full_quant_gemv) return fail("full_quant_gemv disabled");
    if (c.head_dim != 32 && c.head_dim != 48 && c.head_dim != 64 && c.head_dim != 80 && c.head_dim != 96 && c.head_dim != 128)
        return fail("fused attention supports head_dim 32/48/64/80/96/128");
    if (c.kv_dtype == Q8_B32T2 && c.head_dim % 32 != 0) return fail("Q8 KV needs head_dim % 32 == 0");
    if (dec_attn_pv_smem(c.head_dim, c.max_ctx, DEC_ATTN_MAX_SPLITS) > IFA_LDS_LIMIT) return fail("max_context_len too large for the fused attention kernels' LDS (decode falls back to the op-by-op path)");
    if (c.dim % 32 != 0 || c.ffn % 32 != 0) return fail("dim/ffn must be multiples of 32");
    if (c.dim > 8192) return fail("fused norm prologue supports dim <= 8192");
    for (const Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();
        if (moe) {
            if ((int)L.experts.size() != c.experts * 3 || !L.moe_table) return fail("MoE: expert tensors missing");
            for (int e = 0; e < c.experts; e++)
                for (int k = 0; k < 3; k++) {
                    const Tensor &t = L.experts[(size_t)e * 3 + k];
                    if (!t.present() || !t.tiled || !(k == 1 ? dec_gemv_supported_long(t.dtype, t.cols) : dec_gemv_supported(t.dtype, t.cols)))
                        return fail("MoE: expert weights must be in an int8-GEMV format");
                    if (!same_fmt(t.dtype, L.experts[(size_t)(k == 1 ? 1 : 0)].dtype) || t.rows != L.experts[(size_t)k].rows) return fail("MoE: experts differ in dtype / shape");
                }
        }
        const int ids_dense[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W2};
        const int ids_moe[] = {T_WQ, T_WK, T_WV, T_WO};
        const int *ids = moe ? ids_moe : ids_dense;
        const int n_ids = moe ? 4 : 6;
        for (int ii = 0; ii < n_ids; ii++) {
            const int id = ids[ii];
            const Tensor &t = L.t[id];
            if (!t.present()) return fail("fused path: a layer's weight matrix is missing");
            const bool plain_input = id == T_WO || id == T_W2;      // neither normalised nor gated: long rows allowed
            if (!fused_ok(t, plain_input)) return fail("fused GEMV: columns out of range for this weight format (or cols % 8 != 0 for fp16 activations)");
        }
        if (L.t[T_W3].present() && (!fused_ok(L.t[T_W3], false) || !same_fmt(L.t[T_W3].dtype, L.t[T_W1].dtype))) return fail("w1/w3 dtype mismatch");
        if (!L.t[T_ATTN_NORM].present()) return fail("pre-norm weights required");
        if (!L.t[T_FFN_NORM].present() && !c.parallel_attn) return fail("ffn pre-norm weights required");
        // (wq / wk / wv of different formats -- grouped-query models under the tensor_quant_threshold rule -- get one launch each)
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    // a pipeline stage (BY_LAYER partition) may hold neither embeddings nor lm_head: checked where they are used
    if (!lm.present()) { /* middle / first stage */ }
    else if (lm.dtype == F16) {
        if (lm.cols > 8192 || lm.cols % 8 != 0) return fail("fused F16 lm_head needs cols <= 8192");
    } else if (!fused_ok(lm, false) || !m->g[T_OUT_NORM].present()) {
        return fail("fused lm_head: columns out of range for its weight format (or no output norm)");
    }
    if (m->g[T_EMBD].present() && m->g[T_EMBD].dtype != F16) return fail("F16 embeddings required");
    return true;
}

// --------------------------------------------------- fused step (enqueue only)
// Std-norm models (Falcon, Bloom, OPT ...): the norm runs as the op-level kernel (same arithmetic as the op path by
// construction) into `dst`, and the GEMV that follows takes it without a norm prologue.
int sep_norm(ifa_model *m, const half_t *x, const Tensor &w, const Tensor &b, half_t *dst)
{
    return ifa_layernorm(m->cfg.norm_kind, x, 1, (size_t)m->cfg.dim, w.data, b.data, 0.0f, m->cfg.eps, dst, (ifa_stream)m->stream);
}

// Can layer l take the attention as the tail of its QKV launch?  (the one-workgroup-per-head attention, RMS-norm wiring, q / k / v
// of one int8-path format with a kernel instance, no tensor-parallel pending sum)
bool qkv_attn_layer_ok(const ifa_model *m, int l, int *gk_out)
{
    const ifa_model_config &c = m->cfg;
    const Layer &L = m->layers[(size_t)l];
    if (c.norm_kind != 0 || c.tp_size > 1) return false;
    const Tensor &wq = L.t[T_WQ], &wk = L.t[T_WK], &wv = L.t[T_WV];
    if (!wq.present() || !wk.present() || !wv.present() || !wq.tiled || !wk.tiled || !wv.tiled) return false;
    if (!fused_int8(wq.dtype) || !same_fmt(wq.dtype, wk.dtype) || !same_fmt(wq.dtype, wv.dtype)) return false;
    if ((int)wq.rows != c.heads * c.head_dim || (int)wk.rows != c.kv_heads * c.head_dim || (int)wv.rows != c.kv_heads * c.head_dim) return false;
    int rw = 0;
    return dec_qkv_attn_supported(wq.dtype, (int)wq.cols, c.heads, c.kv_heads, c.head_dim, num_cus(), gk_out, &rw);
}

// decides qa_on for the next captured step and allocates what the fused launch needs (never under capture)
int qkv_attn_ready(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    int want = m->opt_fuse_attn && waits_enabled() && !m->attn_split && dec_attn_smem(c.head_dim, c.max_ctx, 256) <= IFA_LDS_LIMIT;
    int gk = 0;
    for (int l = 0; want && l < c.layers; l++) if (!qkv_attn_layer_ok(m, l, &gk)) want = 0;
    // a head's attention waits for the gk workgroups of its kv group: the grid (kv_heads * gk workgroups of 512 threads at 256
    // registers, one per CU) must be resident at once on the CUs this process may use (CU mask, partitioned device)
    if (want && (long long)c.kv_heads * gk > (long long)visible_cus()) want = 0;
    if (want && !m->qa_gran) {
        const size_t n = (size_t)c.layers * (size_t)(c.heads + 2 * c.kv_heads) * c.head_dim;
        IFA_HIP_CHECK(hipMalloc((void **)&m->qa_gran, n * 8));
        IFA_HIP_CHECK(hipMemsetAsync(m->qa_gran, 0, n * 8, m->stream));
        if (m->qa_call) { (void)hipFree(m->qa_call); m->qa_call = nullptr; }
        if (m->qa_err) { (void)hipFree(m->qa_err); m->qa_err = nullptr; }
        IFA_HIP_CHECK(hipMalloc((void **)&m->qa_call, 16));
        IFA_HIP_CHECK(hipMemsetAsync(m->qa_call, 0, 16, m->stream));
        IFA_HIP_CHECK(hipMalloc((void **)&m->qa_err, 16));
        IFA_HIP_CHECK(hipMemsetAsync(m->qa_err, 0, 16, m->stream));
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    // the chained FFN launch: dense gated FFN behind an RMS pre-norm, sequential wiring, W1 / W3 / W2 (and Wo) of one format with an
    // instance, one workgroup per CU resident at once
    int want_ch = (m->opt_fuse_ffn && waits_enabled() && c.norm_kind == 0 && c.tp_size <= 1 && c.experts == 0
                   && !c.parallel_attn && !c.share_input && num_cus() <= visible_cus()) ? std::min(m->opt_fuse_ffn, 2) : 0;
    if (want_ch == 2 && (!m->attq || !m->opt_attn_q8 || c.head_dim % 32 != 0)) want_ch = 1;
    for (int l = 0; want_ch && l < c.layers; l++) {
        const Layer &L = m->layers[(size_t)l];
        const Tensor &wo = L.t[T_WO], &w1 = L.t[T_W1], &w3 = L.t[T_W3], &w2 = L.t[T_W2];
        if (!w1.present() || !w1.tiled || !w3.present() || !w3.tiled || !w2.present() || !w2.tiled || !L.t[T_FFN_NORM].present()
            || (int)w1.cols != c.dim || (int)w2.rows != c.dim || w2.cols != w1.rows || w3.rows != w1.rows || !same_fmt(w1.dtype, w3.dtype)
            || !dec_chain_supported(w1.dtype, w2.dtype, w1.dtype, c.dim, (int)w1.rows, false, c.dim, num_cus()))
            want_ch = 0;
        else if (want_ch == 2 && (!wo.present() || !wo.tiled || (int)wo.rows != c.dim
                                  || !dec_chain_supported(w1.dtype, w2.dtype, wo.dtype, c.dim, (int)w1.rows, true, (int)wo.cols, num_cus())))
            want_ch = 1;
    }
    if (want_ch && !m->ch_gran) {
        const size_t n = (size_t)c.layers * (size_t)(c.dim + (int)m->layers[0].t[T_W1].rows);
        IFA_HIP_CHECK(hipMalloc((void **)&m->ch_gran, n * 4));
        IFA_HIP_CHECK(hipMemsetAsync(m->ch_gran, 0, n * 4, m->stream));
        IFA_HIP_CHECK(hipMalloc((void **)&m->ch_flags, (size_t)c.layers * 2 * 1024 * 4));
        IFA_HIP_CHECK(hipMemsetAsync(m->ch_flags, 0, (size_t)c.layers * 2 * 1024 * 4, m->stream));
        if (!m->qa_call) { IFA_HIP_CHECK(hipMalloc((void **)&m->qa_call, 16)); IFA_HIP_CHECK(hipMemsetAsync(m->qa_call, 0, 16, m->stream)); }
        if (!m->qa_err) { IFA_HIP_CHECK(hipMalloc((void **)&m->qa_err, 16)); IFA_HIP_CHECK(hipMemsetAsync(m->qa_err, 0, 16, m->stream)); }
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    }
    if (want != m->qa_on || (want && gk != m->qa_gk) || want_ch != m->ch_on) {
        m->qa_on = want; m->qa_gk = gk; m->ch_on = want_ch; drop_graphs(m);
    }
    return IFA_OK;
}

void qkv_params(ifa_model *m, int l, const half_t *x, DecGemvParams &P)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    memset(&P, 0, sizeof(P));
    P.x = x; P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.norm_b = (const half_t *)L.t[T_ATTN_NORM_B].data;
    P.multi_base = c.attn_norm_base; P.eps = c.eps; P.cols = c.dim; P.nblk = c.dim / 32;
    if (c.parallel_attn) P.xn_out = m->xn;
    const int ids[3] = {T_WQ, T_WK, T_WV}; const int bids[3] = {T_WQ_B, T_WK_B, T_WV_B};
    const size_t QDd = (size_t)c.heads * c.head_dim, KVDd = (size_t)c.kv_heads * c.head_dim;
    half_t *outs[3] = {m->dqkv, m->dqkv + QDd, m->dqkv + QDd + KVDd};
    for (int i = 0; i < 3; i++) {
        P.W0[i] = wbytes(L.t[ids[i]]); P.b0[i] = (const half_t *)L.t[bids[i]].data;
        P.y[i] = outs[i]; P.rows[i] = (int)L.t[ids[i]].rows;
    }
    P.nsets = 3;
}

// QKV GEMVs + the attention of every head in ONE launch (tag_add: distinct tags for the timing loop's repeated launches)
int launch_qkv_attn(ifa_model *m, int l, const half_t *x, unsigned tag_add)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; qkv_params(m, l, x, P);
    DecAttnParams A; attn_params(m, l, A);
    DecQkvAttnExtra E; memset(&E, 0, sizeof(E));
    E.gran = m->qa_gran + (size_t)l * (size_t)(c.heads + 2 * c.kv_heads) * c.head_dim;
    E.epoch = m->qa_call; E.epoch_add = tag_add; E.err = m->qa_err; E.timeout_us = m->opt_fuse_attn_timeout_us; E.gk = m->qa_gk;
    const int pb = (m->attn_pb == 64 || m->attn_pb == 128) ? m->attn_pb : 256;
    const bool kt = m->opt_attn_kt && !A.kv_q8 && dec_attn_smem(c.head_dim, c.max_ctx, pb) <= IFA_LDS_LIMIT;
    E.unload = ((m->opt_attn_unload >= 1 && pb == 256) || (m->opt_attn_unload >= 2 && pb == 128 && (kt || A.kv_q8))) ? 1 : 0;      // (2: the 128-row bucket too -- measurement)
    return dec_qkv_attn_launch(L.t[T_WQ].dtype, 1, A.kv_q8 != 0, pb, kt, P, A, E, c.max_ctx, m->stream);
}

int launch_qkv(ifa_model *m, int l, const half_t *x)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    P.x = x; P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.norm_b = (const half_t *)L.t[T_ATTN_NORM_B].data;
    P.multi_base = c.attn_norm_base; P.eps = c.eps; P.cols = c.dim; P.nblk = c.dim / 32;
    if (c.parallel_attn) P.xn_out = m->xn;         // the normalised input: parallel-attention models feed it to the FFN
    const bool std_norm = c.norm_kind != 0;
    if (std_norm) {
        int rc = sep_norm(m, x, L.t[T_ATTN_NORM], L.t[T_ATTN_NORM_B], m->xn);
        if (rc) return rc;
        P.x = m->xn; P.norm_w = nullptr; P.norm_b = nullptr; P.xn_out = nullptr;
    }
    const int ids[3] = {T_WQ, T_WK, T_WV}; const int bids[3] = {T_WQ_B, T_WK_B, T_WV_B};
    const size_t QDd = (size_t)c.heads * c.head_dim, KVDd = (size_t)c.kv_heads * c.head_dim;
    half_t *outs[3] = {m->dqkv, m->dqkv + QDd, m->dqkv + QDd + KVDd};
    for (int i = 0; i < 3; i++) {
        P.W0[i] = wbytes(L.t[ids[i]]); P.b0[i] = (const half_t *)L.t[bids[i]].data;
        P.y[i] = outs[i]; P.rows[i] = (int)L.t[ids[i]].rows;
    }
    P.nsets = 3;
    if (m->pend.on && !std_norm) {      // x = pend.x + merged product: formed in this kernel's prologue, stored as the new layer input
        P.x = m->pend.x; P.x_add = m->pend.add; P.x_add_bias = m->pend.bias; P.xsum_out = m->pend.out;
        m->pend.on = false;
    }
    auto go = [&](int dtype, const DecGemvParams &Q) {
        if (std_norm) return launch_dec_gemv<EPI_PLAIN, 0>(dtype, Q, m->opt_rpw_qkv, m->stream);
        return launch_dec_gemv<EPI_PLAIN, 1>(dtype, Q, m->opt_rpw_qkv, m->stream);
    };
    if (same_fmt(L.t[T_WQ].dtype, L.t[T_WK].dtype) && same_fmt(L.t[T_WQ].dtype, L.t[T_WV].dtype)) return go(L.t[T_WQ].dtype, P);
    // mixed formats (e.g. wq quantised, wk / wv left F16 by the threshold rule): one launch per matrix; the first one forms
    // a pending sum, the others read the stored result
    for (int i = 0; i < 3; i++) {
        DecGemvParams Q = P;
        Q.nsets = 1; Q.W0[0] = P.W0[i]; Q.b0[0] = P.b0[i]; Q.y[0] = P.y[i]; Q.rows[0] = P.rows[i];
        if (i > 0) {
            Q.xn_out = nullptr;
            if (P.x_add) { Q.x = P.xsum_out; Q.x_add = nullptr; Q.x_add_bias = nullptr; Q.xsum_out = nullptr; }
        }
        int rc = go(L.t[ids[i]].dtype, Q);
        if (rc) return rc;
    }
    return IFA_OK;
}

void attn_params(ifa_model *m, int l, DecAttnParams &A)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const int rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f);
    memset(&A, 0, sizeof(A));
    A.q = m->dqkv; A.k_new = m->dqkv + (size_t)c.heads * c.head_dim; A.v_new = A.k_new + (size_t)c.kv_heads * c.head_dim; A.kcache = (uint8_t *)L.kcache; A.vcache = (uint8_t *)L.vcache;
    A.state = m->state; A.rope_tab = m->rope_tab; A.heads = c.heads; A.kv_heads = c.kv_heads;
    A.kv_q8 = c.kv_dtype == Q8_B32T2; A.kq_scale = c.use_alibi ? 1.0f : c.kq_scale;
    A.rope_order = c.rope_order; A.rope_cols = rope_dims;
    A.alibi = c.use_alibi; A.alibi_base = c.tp_rank * c.heads; A.alibi_total = c.heads * std::max(1, c.tp_size);
    A.out = m->att; A.max_ctx = c.max_ctx; A.xq = (c.head_dim % 32 == 0) ? m->attq : nullptr; A.trace = g_trace_ptr;
}

int launch_attn(ifa_model *m, int l)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const int rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f);
    DecAttnParams A; memset(&A, 0, sizeof(A));
    A.q = m->dqkv; A.k_new = m->dqkv + (size_t)c.heads * c.head_dim; A.v_new = A.k_new + (size_t)c.kv_heads * c.head_dim; A.kcache = (uint8_t *)L.kcache; A.vcache = (uint8_t *)L.vcache;
    A.state = m->state; A.rope_tab = m->rope_tab; A.heads = c.heads; A.kv_heads = c.kv_heads;
    A.kv_q8 = c.kv_dtype == Q8_B32T2; A.kq_scale = c.use_alibi ? 1.0f : c.kq_scale;
    A.rope_order = c.rope_order; A.rope_cols = rope_dims;
    A.alibi = c.use_alibi; A.alibi_base = c.tp_rank * c.heads; A.alibi_total = c.heads * std::max(1, c.tp_size);
    A.out = m->att; A.max_ctx = c.max_ctx; A.xq = (c.head_dim % 32 == 0) ? m->attq : nullptr; A.trace = g_trace_ptr;
    // the one-workgroup kernel keeps a head's score row [max_ctx] in LDS: past the device limit (160 KiB: ~75K tokens of
    // context at head_dim 128) the keys-split-over-workgroups kernels run from position 0 on (scores in global memory)
    const bool lds_split = dec_attn_smem(c.head_dim, c.max_ctx) > IFA_LDS_LIMIT;
    if (m->attn_split || lds_split) {
        // splits per head: 8, or what the decode call chose for the context it will reach (attn_split = 8 / 16 / 32); when only
        // the LDS forces the split (very large max_context_len) as many as keep a split's probabilities inside the LDS
        int nsp = m->attn_split > 1 ? m->attn_split : 8;
        while (nsp < DEC_ATTN_MAX_SPLITS && dec_attn_pv_smem(c.head_dim, c.max_ctx, nsp) > IFA_LDS_LIMIT) nsp *= 2;
        if (m->opt_attn_nsplits > 0) nsp = std::min(m->opt_attn_nsplits, DEC_ATTN_MAX_SPLITS);
        m->attn_ws.nsplits = nsp;
        const dim3 g2((unsigned)c.heads, (unsigned)nsp);
        const size_t psmem = dec_attn_pv_smem(c.head_dim, c.max_ctx, nsp);
        const size_t ssmem = dec_attn_scores_smem(c.head_dim, false);      // staging of 256 F16 key rows
        if (psmem > IFA_LDS_LIMIT) return ifa_fail(IFA_ERR_ARG, "fused attention: max_context_len %d needs %zu bytes of LDS per workgroup", c.max_ctx, psmem);
#define IFA_ATTN_S(HDV) \
    case HDV: if (A.kv_q8) { k_dec_attn_scores<HDV, true><<<g2, dim3(256), 16, m->stream>>>(A, m->attn_ws); \
                             if (psmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_pv<HDV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem)); \
                             k_dec_attn_pv<HDV, true><<<g2, dim3(256), psmem, m->stream>>>(A, m->attn_ws); } \
              else { if (ssmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_scores<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssmem)); \
                     k_dec_attn_scores<HDV, false><<<g2, dim3(256), ssmem, m->stream>>>(A, m->attn_ws); \
                     if (psmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_pv<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem)); \
                     k_dec_attn_pv<HDV, false><<<g2, dim3(256), psmem, m->stream>>>(A, m->attn_ws); } \
              k_dec_attn_combine<HDV><<<dim3((unsigned)c.heads), dim3(HDV), 0, m->stream>>>(m->attn_ws, m->att, A.xq, c.heads); break;
        // head sizes that are not whole Q8 blocks (48, 80) exist with an F16 KV cache only
#define IFA_ATTN_SF(HDV) \
    case HDV: if (ssmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_scores<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssmem)); \
              k_dec_attn_scores<HDV, false><<<g2, dim3(256), ssmem, m->stream>>>(A, m->attn_ws); \
              if (psmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)k_dec_attn_pv<HDV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)psmem)); \
              k_dec_attn_pv<HDV, false><<<g2, dim3(256), psmem, m->stream>>>(A, m->attn_ws); \
              k_dec_attn_combine<HDV><<<dim3((unsigned)c.heads), dim3(HDV), 0, m->stream>>>(m->attn_ws, m->att, A.xq, c.heads); break;
        switch (c.head_dim) {
            IFA_ATTN_S(32) IFA_ATTN_S(64) IFA_ATTN_S(96) IFA_ATTN_S(128) IFA_ATTN_SF(48) IFA_ATTN_SF(80)
        default: return ifa_fail(IFA_ERR_ARG, "fused attention: head_dim %d", c.head_dim);
        }
#undef IFA_ATTN_SF
#undef IFA_ATTN_S
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    const int pb = (m->attn_pb == 64 || m->attn_pb == 128) ? m->attn_pb : 256;
    const bool kt = m->opt_attn_kt && !A.kv_q8 && (c.head_dim == 32 || c.head_dim == 64 || c.head_dim == 128)
        && dec_attn_smem(c.head_dim, c.max_ctx, pb) <= IFA_LDS_LIMIT;
    const size_t asmem = dec_attn_smem(c.head_dim, c.max_ctx, kt ? pb : 0);
    const dim3 grid((unsigned)c.heads), block(256);
    // (attention_lds_ok() routed contexts whose score row does not fit the 160 KiB LDS to the split kernels above)
#define IFA_ATTN_GO(HDV, Q8V, PBV, KTV) do { \
        auto kern = k_dec_attn<HDV, Q8V, false, PBV, KTV>; \
        if (asmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asmem)); \
        kern<<<grid, block, asmem, m->stream>>>(A.q, A.kcache, A.vcache, A.heads, A.kv_heads, A); } while (0)
#define IFA_ATTN_PB(HDV, Q8V, KTV) do { if (pb == 64) IFA_ATTN_GO(HDV, Q8V, 64, KTV); else if (pb == 128) IFA_ATTN_GO(HDV, Q8V, 128, KTV); else IFA_ATTN_GO(HDV, Q8V, 256, KTV); } while (0)
    // head sizes with a power-of-two number of 16-byte pieces take the K rows through the LDS tile (F16 cache)
#define IFA_ATTN(HDV) \
    case HDV: if (A.kv_q8) IFA_ATTN_PB(HDV, true, false); else if (kt) IFA_ATTN_PB(HDV, false, true); else IFA_ATTN_PB(HDV, false, false); break;
#define IFA_ATTN_Q(HDV) \
    case HDV: if (A.kv_q8) IFA_ATTN_GO(HDV, true, 256, false); else IFA_ATTN_GO(HDV, false, 256, false); break;
#define IFA_ATTN_F(HDV) \
    case HDV: IFA_ATTN_GO(HDV, false, 256, false); break;
    switch (c.head_dim) {
        IFA_ATTN(32) IFA_ATTN(64) IFA_ATTN_Q(96) IFA_ATTN(128) IFA_ATTN_F(48) IFA_ATTN_F(80)
    default: return ifa_fail(IFA_ERR_ARG, "fused attention: head_dim %d", c.head_dim);
    }
#undef IFA_ATTN_Q
#undef IFA_ATTN_PB
#undef IFA_ATTN_GO
#undef IFA_ATTN_F
#undef IFA_ATTN
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// partial != nullptr (tensor parallel): write the un-merged product there, no bias, no residual
// option q3h_native: the launch's matrix and kernel format when the tensor has its 32-byte-per-block copy (ensure_q3hn)
static int native_fmt(const ifa_model *m, const Tensor &t, const uint8_t *&w)
{
    if (m->opt_q3h_native && t.q3hn && t.dtype == Q3H_B64T1) { w = (const uint8_t *)t.q3hn; return Q3H_NATIVE; }
    return t.dtype;
}

int launch_wo(ifa_model *m, int l, const half_t *x, half_t *partial)
{
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    P.x = m->att; P.cols = (int)L.t[T_WO].cols; P.nblk = P.cols / 32; P.eps = m->cfg.eps;
    P.W0[0] = wbytes(L.t[T_WO]); P.rows[0] = (int)L.t[T_WO].rows; P.nsets = 1;
    // the attention kernel left its output quantised (XqImage): the GEMV needs no prologue.  Rows longer than a lane's
    // register image (chunked kernel) keep the in-kernel quantiser
    const bool preq = m->attq && m->opt_attn_q8 && m->cfg.head_dim % 32 == 0 && P.cols == m->cfg.heads * m->cfg.head_dim && fused_int8(L.t[T_WO].dtype)
        && dec_gemv_supported(L.t[T_WO].dtype, (size_t)P.cols);
    if (preq) P.x = reinterpret_cast<const half_t *>(m->attq);      // NORM == 2 kernels read the quantised image through P.x
    const int dto = native_fmt(m, L.t[T_WO], P.W0[0]);
    if (partial) {
        P.y[0] = partial;
        return preq ? launch_dec_gemv<EPI_PLAIN, 2>(dto, P, m->opt_rpw_wo, m->stream)
                    : launch_dec_gemv<EPI_PLAIN, 0>(dto, P, m->opt_rpw_wo, m->stream);
    }
    P.b0[0] = (const half_t *)L.t[T_WO_B].data;
    P.y[0] = m->a; P.residual = x;
    if (scale_on(m->cfg.attn_out_scale)) P.pre_scale = m->cfg.attn_out_scale;      // Scale(self_att_out) fused in front of the residual add
    if (m->cfg.parallel_attn || m->cfg.share_input)      // the residual is added once, after the FFN (inference_worker.cc:847-851)
        return preq ? launch_dec_gemv<EPI_PLAIN, 2>(dto, P, m->opt_rpw_wo, m->stream)
                    : launch_dec_gemv<EPI_PLAIN, 0>(dto, P, m->opt_rpw_wo, m->stream);
    return preq ? launch_dec_gemv<EPI_RESIDUAL, 2>(dto, P, m->opt_rpw_wo, m->stream)
                : launch_dec_gemv<EPI_RESIDUAL, 0>(dto, P, m->opt_rpw_wo, m->stream);
}

void moe_params(ifa_model *m, Layer &L, DecGemvParams &P, int slot, int tab_off)
{
    P.w_table = (const uint8_t *const *)L.moe_table;
    P.moe_sel = m->moe_route;
    P.moe_w = reinterpret_cast<const half_t *>(reinterpret_cast<const char *>(m->moe_route) + 32);
    P.moe_acc = m->f;
    P.moe_slot = slot; P.moe_tab_off = tab_off;
}

// moe_slot >= 0: the FFN of the expert the router put in that slot (weights through L.moe_table)
int launch_ffn13(ifa_model *m, int l, int moe_slot, const half_t *x_layer, int moe_nslots)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    P.x = m->a; P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.norm_b = (const half_t *)L.t[T_FFN_NORM_B].data;
    P.eps = c.eps; P.cols = c.dim; P.nblk = c.dim / 32; P.act_kind = c.act_kind; P.multi_base = c.ffn_norm_base;
    if (moe_slot >= 0) {
        const Tensor &e1 = L.experts[0], &e3 = L.experts[2];
        moe_params(m, L, P, moe_slot, 0);
        P.W0[0] = (const uint8_t *)e1.tiled; P.W1 = (const uint8_t *)e3.tiled;   // (replaced by the table lookup)
        // moe_nslots router slots in one launch: set i = the expert of slot moe_slot + i, its product at t1 + i * ffn
        const int ns = std::max(1, std::min(3, moe_nslots));
        for (int i = 0; i < ns; i++) { P.y[i] = m->t1 + (size_t)i * e1.rows; P.rows[i] = (int)e1.rows; P.W0[i] = P.W0[0]; }
        P.nsets = ns;
        if (e3.present()) return launch_dec_gemv<EPI_MOE_GLU, 1>(e1.dtype, P, m->opt_rpw_ffn, m->stream);
        return launch_dec_gemv<EPI_MOE_ACT, 1>(e1.dtype, P, m->opt_rpw_ffn, m->stream);
    }
    P.W0[0] = wbytes(L.t[T_W1]); P.b0[0] = (const half_t *)L.t[T_W1_B].data;
    P.y[0] = m->t1; P.rows[0] = (int)L.t[T_W1].rows; P.nsets = 1;
    // FFN input (inference_worker.cc:853-872): the attention's normalised input (parallel attention), the layer input
    // (shared input) or the attention output + residual; then the FFN pre-norm if the model has one
    const half_t *ff_in = c.parallel_attn ? m->xn : (c.share_input ? x_layer : m->a);
    bool need_norm = L.t[T_FFN_NORM].present();
    P.x = ff_in;
    if (need_norm && c.norm_kind != 0) {
        int rc = sep_norm(m, ff_in, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn);
        if (rc) return rc;
        P.x = m->hn; need_norm = false;
    }
    if (!need_norm) { P.norm_w = nullptr; P.norm_b = nullptr; }
    const bool glu = L.t[T_W3].present();
    if (glu) { P.W1 = wbytes(L.t[T_W3]); P.b1 = (const half_t *)L.t[T_W3_B].data; }
    int dtw = L.t[T_W1].dtype;
    if (m->opt_q3h_native && L.t[T_W1].q3hn && (!glu || L.t[T_W3].q3hn)) {      // both matrices of the pair in the native form, or neither
        dtw = native_fmt(m, L.t[T_W1], P.W0[0]);
        if (glu) (void)native_fmt(m, L.t[T_W3], P.W1);
    }
    if (need_norm && m->pend.on && P.x == m->pend.out) {     // the FFN input is the pending sum
        P.x = m->pend.x; P.x_add = m->pend.add; P.x_add_bias = m->pend.bias; P.xsum_out = m->pend.out;
        m->pend.on = false;
    }
    if (need_norm) return glu ? launch_dec_gemv<EPI_GLU, 1>(dtw, P, m->opt_rpw_ffn, m->stream) : launch_dec_gemv<EPI_ACT, 1>(dtw, P, m->opt_rpw_ffn, m->stream);
    return glu ? launch_dec_gemv<EPI_GLU, 0>(dtw, P, m->opt_rpw_ffn, m->stream) : launch_dec_gemv<EPI_ACT, 0>(dtw, P, m->opt_rpw_ffn, m->stream);
}

int launch_w2(ifa_model *m, int l, half_t *xnext, half_t *partial, int moe_slot, bool moe_last,
                     const half_t *residual2, int moe_t1_slot)
{
    Layer &L = m->layers[(size_t)l];
    DecGemvParams P; memset(&P, 0, sizeof(P));
    if (moe_slot >= 0) {
        const Tensor &e2 = L.experts[1];
        moe_params(m, L, P, moe_slot, 2);
        P.x = m->t1 + (size_t)moe_t1_slot * e2.cols; P.cols = (int)e2.cols; P.eps = m->cfg.eps;      // (the gated product of this slot)
        P.W0[0] = (const uint8_t *)e2.tiled; P.rows[0] = (int)e2.rows; P.nsets = 1;
        if (partial) {      // tensor parallel: accumulate the weighted shard products; merged and finished by the caller
            P.y[0] = partial; P.moe_acc = partial;
            return launch_dec_gemv<EPI_MOE_ACC, 0>(e2.dtype, P, m->opt_rpw_w2, m->stream);
        }
        P.y[0] = moe_last ? xnext : m->f; P.residual = m->a; P.residual2 = residual2;
        if (moe_last && scale_on(m->cfg.ffn_out_scale)) P.pre_scale = m->cfg.ffn_out_scale;
        if (moe_last && l + 1 == m->cfg.layers && scale_on(m->cfg.out_scale)) P.post_scale = m->cfg.out_scale;
        if (moe_last) return launch_dec_gemv<EPI_MOE_LAST, 0>(e2.dtype, P, m->opt_rpw_w2, m->stream);
        return launch_dec_gemv<EPI_MOE_ACC, 0>(e2.dtype, P, m->opt_rpw_w2, m->stream);
    }
    P.x = m->t1; P.cols = (int)L.t[T_W2].cols; P.nblk = P.cols / 32; P.eps = m->cfg.eps;
    P.W0[0] = wbytes(L.t[T_W2]); P.rows[0] = (int)L.t[T_W2].rows; P.nsets = 1;
    const int dt2 = native_fmt(m, L.t[T_W2], P.W0[0]);
    if (partial) {
        P.y[0] = partial;
        return launch_dec_gemv<EPI_PLAIN, 0>(dt2, P, m->opt_rpw_w2, m->stream);
    }
    P.b0[0] = (const half_t *)L.t[T_W2_B].data;
    P.y[0] = xnext; P.residual = m->a; P.residual2 = residual2;     // + layer input for parallel / shared-input models
    if (scale_on(m->cfg.ffn_out_scale)) P.pre_scale = m->cfg.ffn_out_scale;                               // Scale(ff_out)
    if (l + 1 == m->cfg.layers && scale_on(m->cfg.out_scale)) P.post_scale = m->cfg.out_scale;        // Scale(last layer's output)
    return launch_dec_gemv<EPI_RESIDUAL, 0>(dt2, P, m->opt_rpw_w2, m->stream);
}

// [Wo ->] W1 | W3 -> W2 of layer l as ONE launch (ifa_decode_chain.h); x = the layer input (Wo's residual), xnext = the layer output
int launch_chain(ifa_model *m, int l, const half_t *x, half_t *xnext, unsigned tag_add)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const bool wo = m->ch_on == 2;
    DecGemvParams PW; memset(&PW, 0, sizeof(PW));       // launch_wo's EPI_RESIDUAL / NORM 2 parameters
    if (wo) {
        PW.x = reinterpret_cast<const half_t *>(m->attq); PW.cols = (int)L.t[T_WO].cols; PW.eps = c.eps;
        PW.W0[0] = wbytes(L.t[T_WO]); PW.rows[0] = (int)L.t[T_WO].rows;
        PW.b0[0] = (const half_t *)L.t[T_WO_B].data; PW.y[0] = m->a; PW.residual = x;
        if (scale_on(c.attn_out_scale)) PW.pre_scale = c.attn_out_scale;
    }
    DecGemvParams P; memset(&P, 0, sizeof(P));          // launch_ffn13's dense EPI_GLU, NORM 1 parameters
    P.x = m->a; P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.norm_b = (const half_t *)L.t[T_FFN_NORM_B].data;
    P.eps = c.eps; P.cols = c.dim; P.act_kind = c.act_kind; P.multi_base = c.ffn_norm_base;
    P.W0[0] = wbytes(L.t[T_W1]); P.b0[0] = (const half_t *)L.t[T_W1_B].data; P.y[0] = m->t1; P.rows[0] = (int)L.t[T_W1].rows;
    P.W1 = wbytes(L.t[T_W3]); P.b1 = (const half_t *)L.t[T_W3_B].data;
    DecGemvParams Q; memset(&Q, 0, sizeof(Q));          // launch_w2's EPI_RESIDUAL parameters
    Q.x = m->t1; Q.cols = (int)L.t[T_W2].cols; Q.eps = c.eps;
    Q.W0[0] = wbytes(L.t[T_W2]); Q.rows[0] = (int)L.t[T_W2].rows; Q.b0[0] = (const half_t *)L.t[T_W2_B].data;
    Q.y[0] = xnext; Q.residual = m->a;
    if (scale_on(c.ffn_out_scale)) Q.pre_scale = c.ffn_out_scale;
    if (l + 1 == c.layers && scale_on(c.out_scale)) Q.post_scale = c.out_scale;
    DecChainExtra E; memset(&E, 0, sizeof(E));
    const size_t per = (size_t)c.dim + L.t[T_W1].rows;
    E.gran_a = m->ch_gran + (size_t)l * per; E.gran_h = E.gran_a + c.dim;
    E.flags_a = m->ch_flags + (size_t)l * 2048; E.flags_h = E.flags_a + 1024;
    E.state = m->state; E.epoch = m->qa_call; E.epoch_add = tag_add; E.err = m->qa_err; E.timeout_us = m->opt_fuse_attn_timeout_us;
    E.trace = g_trace_ptr; E.late_w2 = m->opt_chain_late_w2;
    return dec_chain_launch(L.t[T_W1].dtype, true, 1, wo, P, Q, wo ? &PW : nullptr, E, num_cus(), m->stream);
}

// can the step end in the one-launch tail?  (F16 lm_head behind the RMS / no final norm, the embedding table on this worker)
bool step_tail_ok(const ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    return m->opt_step_tail && m->g[T_LM_HEAD].present() && m->g[T_LM_HEAD].dtype == F16 && m->g[T_EMBD].present()
        && !(c.norm_kind != 0 && m->g[T_OUT_NORM].present()) && c.dim % 8 == 0 && c.dim <= 8192;
}

DecLmHeadParams lm_params(ifa_model *m, const half_t *x, half_t *logits_out)
{
    const ifa_model_config &c = m->cfg;
    DecLmHeadParams H; memset(&H, 0, sizeof(H));
    H.x = x; H.norm_w = (const half_t *)m->g[T_OUT_NORM].data; H.norm_b = (const half_t *)m->g[T_OUT_NORM_B].data;
    H.eps = c.eps; H.cols = c.dim; H.W = (const half_t *)m->g[T_LM_HEAD].data; H.logits = logits_out ? logits_out : m->logits;
    H.rows = (int)m->g[T_LM_HEAD].rows; H.xn_out = m->xn; H.multi_base = c.out_norm_base;
    return H;
}

// (allocates: not under capture)
int step_tail_ready(ifa_model *m)
{
    const int want = step_tail_ok(m) ? 1 : 0;
    if (want != m->st_on) { m->st_on = want; drop_graphs(m); }
    if (!want) return IFA_OK;
    const int grid = lmhead_grid(lm_params(m, m->x, nullptr), m->opt_rpw_lm);
    if (grid > m->st_keys_n) {
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
        if (m->st_keys) (void)hipFree(m->st_keys);
        m->st_keys = nullptr; m->st_keys_n = 0;
        IFA_HIP_CHECK(hipMalloc((void **)&m->st_keys, sizeof(unsigned long long) * (size_t)grid));
        m->st_keys_n = grid;
    }
    if (!m->st_counter) {
        IFA_HIP_CHECK(hipMalloc((void **)&m->st_counter, 16));
        IFA_HIP_CHECK(hipMemsetAsync(m->st_counter, 0, 16, m->stream));
    }
    return IFA_OK;
}

// the last launch of a captured step: lm_head, argmax, state advance and the next step's gather (k_dec_lmhead_tail)
int launch_lm_tail(ifa_model *m, const half_t *x)
{
    const ifa_model_config &c = m->cfg;
    const DecLmHeadParams H = lm_params(m, x, nullptr);
    DecStepTail Z; memset(&Z, 0, sizeof(Z));
    Z.state = m->state; Z.ring = ifa_model::RING; Z.keys = m->st_keys; Z.counter = m->st_counter;
    Z.embd = (const half_t *)m->g[T_EMBD].data; Z.vocab = (int)m->g[T_EMBD].rows; Z.x_out = m->x;
    Z.rope_tab = c.rope_order ? m->rope_tab : nullptr; Z.head_dim = c.head_dim; Z.theta = c.rope_theta;
    Z.rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f); Z.embd_scale = c.embd_scale;
    return launch_lmhead(H, m->g[T_OUT_NORM].present() ? 1 : 0, m->opt_rpw_lm, m->stream, &Z);
}

int launch_gather(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    // (debug_hidden_in: the layer input is what the caller stored in "x"; only the step's RoPE table is built)
    k_dec_gather<<<dim3(2), dim3(256), 0, m->stream>>>(m->opt_debug_hidden_in ? nullptr : (const half_t *)m->g[T_EMBD].data, m->state, c.dim, (int)m->g[T_EMBD].rows, m->x,
                                                       c.rope_order ? m->rope_tab : nullptr, c.head_dim, c.rope_theta,
                                                       (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int launch_lm(ifa_model *m, const half_t *x, half_t *logits_out)
{
    const ifa_model_config &c = m->cfg;
    const Tensor &lmt = m->g[T_LM_HEAD];
    if (lmt.dtype != F16) {      // quantised lm_head (<= 20-layer models, network_builder.cc:839-844): same fused GEMV as the layers
        DecGemvParams P; memset(&P, 0, sizeof(P));
        P.x = x; P.norm_w = (const half_t *)m->g[T_OUT_NORM].data; P.norm_b = (const half_t *)m->g[T_OUT_NORM_B].data;
        P.eps = c.eps; P.cols = c.dim; P.xn_out = m->xn; P.multi_base = c.out_norm_base;
        P.W0[0] = wbytes(lmt); P.rows[0] = (int)lmt.rows; P.nsets = 1;
        P.y[0] = logits_out ? logits_out : m->logits;
        return launch_dec_gemv<EPI_PLAIN, 1>(lmt.dtype, P, m->opt_rpw_lm, m->stream);
    }
    if (c.norm_kind != 0 && m->g[T_OUT_NORM].present()) {      // std final norm: op-level kernel, then the plain GEMV
        int rc = sep_norm(m, x, m->g[T_OUT_NORM], m->g[T_OUT_NORM_B], m->xn);
        if (rc) return rc;
        DecLmHeadParams H2; memset(&H2, 0, sizeof(H2));
        H2.x = m->xn; H2.eps = c.eps; H2.cols = c.dim; H2.W = (const half_t *)lmt.data; H2.logits = logits_out ? logits_out : m->logits;
        H2.rows = (int)lmt.rows;
        return launch_lmhead(H2, 0, m->opt_rpw_lm, m->stream);
    }
    return launch_lmhead(lm_params(m, x, logits_out), m->g[T_OUT_NORM].present() ? 1 : 0, m->opt_rpw_lm, m->stream);
}

int enqueue_fused_step(ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    hipStream_t s = m->stream;
    int rc;
    // st_on: the previous step's last launch (or ifa_model_decode, for a call's first step) has gathered this step's input
    if (!m->st_on && (rc = launch_gather(m))) return rc;
    half_t *x = m->x, *xnext = m->x2;
    const int l_first = std::min(std::max(m->opt_debug_layer0, 0), c.layers - 1);
    const int n_layers = (m->opt_debug_layers > 0 && l_first + m->opt_debug_layers < c.layers) ? l_first + m->opt_debug_layers : c.layers;
    for (int l = l_first; l < n_layers; l++) {
        if (m->qa_on) {
            if ((rc = launch_qkv_attn(m, l, x))) return rc;
        } else {
            if ((rc = launch_qkv(m, l, x))) return rc;
            if ((rc = launch_attn(m, l))) return rc;
        }
        if (m->ch_on) {      // [Wo ->] W1 | W3 -> W2 as one launch
            if (m->ch_on == 1 && (rc = launch_wo(m, l, x))) return rc;
            if ((rc = launch_chain(m, l, x, xnext))) return rc;
            std::swap(x, xnext);
            continue;
        }
        if ((rc = launch_wo(m, l, x))) return rc;
        Layer &L = m->layers[(size_t)l];
        if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
            if ((rc = launch_moe_router(m, l))) return rc;
            // the gated products of up to three router slots share one launch, then one W2 launch per slot (each accumulates
            // hfma(product, w, acc) in slot order)
            for (int k0 = 0; k0 < c.moe_top_k; k0 += 3) {
                const int ns = std::min(3, c.moe_top_k - k0);
                if ((rc = launch_ffn13(m, l, k0, nullptr, ns))) return rc;
                for (int k = k0; k < k0 + ns; k++)
                    if ((rc = launch_w2(m, l, xnext, nullptr, k, k + 1 == c.moe_top_k, nullptr, k - k0))) return rc;
            }
        } else {
            const bool extra = c.parallel_attn || c.share_input;
            if ((rc = launch_ffn13(m, l, -1, x))) return rc;
            if ((rc = launch_w2(m, l, xnext, nullptr, -1, false, extra ? x : nullptr))) return rc;
        }
        std::swap(x, xnext);
    }
    if (m->st_on) return launch_lm_tail(m, x);
    if ((rc = launch_lm(m, x))) return rc;
    k_dec_argmax_advance<<<dim3(1), dim3(1024), 0, s>>>(m->logits, (int)m->g[T_LM_HEAD].rows, m->state, ifa_model::RING);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // namespace ifae

extern "C" {

int ifa_model_decode(ifa_model *m, int first_token, int start_pos, int n_steps, int *out_tokens_host, float *elapsed_ms)
{
    return decode_impl(m, first_token, start_pos, n_steps, out_tokens_host, elapsed_ms, false);
}

// Everything a decode call of n_steps from start_pos sets up before its first launch -- the attention variant of the contexts it
// reaches, the hand-off arenas, the captured step(s) -- without running a step: a caller that times its first call (bench.py with
// --warmup 0, a service's first request) keeps graph capture / instantiation out of it.  The KV cache and the activations are not
// touched (the token / position words are rewritten by every call anyway).
int ifa_model_decode_prepare(ifa_model *m, int start_pos, int n_steps)
{
    return decode_impl(m, 0, start_pos, n_steps, nullptr, nullptr, true);
}

} // extern "C"

namespace ifae {

int decode_impl(ifa_model *m, int first_token, int start_pos, int n_steps, int *out_tokens_host, float *elapsed_ms, bool prepare_only)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_decode: model not finalized");
    IFA_REQUIRE(n_steps > 0 && n_steps <= ifa_model::RING, "ifa_model_decode: n_steps %d (max %d per call)", n_steps, ifa_model::RING);
    IFA_REQUIRE(start_pos >= 0 && start_pos + n_steps <= m->cfg.max_ctx, "ifa_model_decode: positions [%d,%d) exceed max_ctx %d",
                start_pos, start_pos + n_steps, m->cfg.max_ctx);
    IFA_REQUIRE(m->g[T_EMBD].present() && m->g[T_LM_HEAD].present(), "ifa_model_decode: embeddings / lm_head missing (pipeline stage worker)");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    std::string why;
    if (m->opt_exact_order) {      // order-exact steps, host-driven (a failure is an error, never a silent change of arithmetic)
        if (prepare_only) return IFA_OK;
        int tok = first_token;
        for (int i = 0; i < n_steps; i++) {
            int nt = 0;
            int rc = forward_exact(m, tok, start_pos + i, nullptr, &nt);
            if (rc) return rc;
            if (out_tokens_host) out_tokens_host[i] = nt;
            tok = nt;
        }
        if (elapsed_ms) *elapsed_ms = -1.0f;
        return IFA_OK;
    }
    if (!m->opt_fused || m->opt_perf_stat || !fused_supported(m, &why)) {
        // op-by-op fallback: same semantics, host-driven (also the path of option perf_stat: one launch per reference op to time)
        if (prepare_only) return IFA_OK;
        int tok = first_token;
        for (int i = 0; i < n_steps; i++) {
            int nt = 0;
            int rc = forward_ops(m, &tok, 1, start_pos + i, nullptr, &nt);
            if (rc) return rc;
            if (out_tokens_host) out_tokens_host[i] = nt;
            tok = nt;
        }
        if (elapsed_ms) *elapsed_ms = -1.0f;
        return IFA_OK;
    }
    static const bool trace_host = getenv("IFA_TRACE_DECODE") != nullptr;       // tuning aid: host-side timeline of the call on stderr
    const auto th0 = std::chrono::steady_clock::now();
    auto th = [&](const char *what) {
        if (trace_host) fprintf(stderr, "decode-host %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - th0).count());
    };
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    if (m->opt_q3h_native && (rc = ensure_q3hn(m))) return rc;
    hipStream_t s = m->stream;
    // attention variant of this call: one workgroup per head, or keys split over workgroups once the context the
    // call reaches passes the threshold (the captured step is re-captured when the variant changes)
    choose_attn_split(m, start_pos + n_steps);
    if ((rc = qkv_attn_ready(m))) return rc;
    if ((rc = step_tail_ready(m))) return rc;
    m->host_pinned[0] = first_token; m->host_pinned[1] = start_pos; m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, s));
    if (m->qa_on || m->ch_on) {      // the granule tags of this call: (call counter, position) -- consecutive steps never share one
        m->qa_calls = (m->qa_calls % 4000u) + 1u;
        m->host_pinned[6] = (int)m->qa_calls;
        IFA_HIP_CHECK(hipMemcpyAsync(m->qa_call, m->host_pinned + 6, sizeof(int), hipMemcpyHostToDevice, s));
    }
    if (m->opt_graph && !m->graph_exec) {
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        IFA_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        rc = enqueue_fused_step(m);
        hipGraph_t gph = nullptr;
        hipError_t e = hipStreamEndCapture(s, &gph);
        if (rc) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (e != hipSuccess) return ifa_fail(IFA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        if (m->graph) (void)hipGraphDestroy(m->graph);
        m->graph = gph;
        IFA_HIP_CHECK(hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0));
    }
    const int S = m->opt_graph_steps;
    if (m->opt_graph && S > 1 && n_steps >= S && (!m->graph_exec_n || m->graph_n_steps != S)) {
        if (m->graph_exec_n) { (void)hipGraphExecDestroy(m->graph_exec_n); m->graph_exec_n = nullptr; }
        if (m->graph_n) { (void)hipGraphDestroy(m->graph_n); m->graph_n = nullptr; }
        IFA_HIP_CHECK(hipStreamSynchronize(s));
        IFA_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < S && !rc; i++) rc = enqueue_fused_step(m);
        hipGraph_t gph = nullptr;
        hipError_t e = hipStreamEndCapture(s, &gph);
        if (rc) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (e != hipSuccess) return ifa_fail(IFA_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        m->graph_n = gph; m->graph_n_steps = S;
        IFA_HIP_CHECK(hipGraphInstantiate(&m->graph_exec_n, m->graph_n, nullptr, nullptr, 0));
    }
    if (prepare_only) return IFA_OK;
    th("state copies enqueued");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (elapsed_ms) { IFA_HIP_CHECK(hipEventCreate(&e0)); IFA_HIP_CHECK(hipEventCreate(&e1)); IFA_HIP_CHECK(hipEventRecord(e0, s)); }
    th("events created, first recorded");
    if (m->st_on && (rc = launch_gather(m))) return rc;      // the call's first step: its input is gathered here, every later one by the step before it
    for (int i = 0; i < n_steps;) {
        if (m->opt_graph && m->graph_exec_n && m->graph_n_steps == S && S > 1 && n_steps - i >= S) { IFA_HIP_CHECK(hipGraphLaunch(m->graph_exec_n, s)); i += S; continue; }
        if (m->opt_graph) IFA_HIP_CHECK(hipGraphLaunch(m->graph_exec, s));
        else if ((rc = enqueue_fused_step(m))) return rc;
        i++;
    }
    th("steps enqueued");
    if (elapsed_ms) IFA_HIP_CHECK(hipEventRecord(e1, s));
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->state + 8, sizeof(int) * (size_t)n_steps, hipMemcpyDeviceToHost, s));
    int *qerr = m->host_pinned + 8 + ifa_model::RING;
    qerr[0] = 0;
    if (m->qa_on || m->ch_on) IFA_HIP_CHECK(hipMemcpyAsync(qerr, m->qa_err, 4, hipMemcpyDeviceToHost, s));
    th("copies back enqueued");
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    th("stream synchronised");
    if (qerr[0] != 0) {      // a head's workgroup gave up waiting for its q | k | v rows: the step's results are not valid
        (void)hipMemsetAsync(m->qa_err, 0, 16, s);
        (void)hipStreamSynchronize(s);
        if (elapsed_ms) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
        // the waiting launches go off for this model (and, process-wide, for every model created later): the next call captures the
        // five-launch step, whose kernels wait for nothing
        m->opt_fuse_attn = 0; m->opt_fuse_ffn = 0;
        drop_graphs(m);
        waits_disable("the fused QKV + attention launch timed out waiting for sibling workgroups");
        return ifa_fail(IFA_ERR_STATE, "fused decode launch: a wait for another workgroup's rows timed out (code 0x%x: 0x5_ q | k | v / attention output, 0x6_ Wo output, 0x9_ chained FFN launch); "
                        "the results of this call are not valid -- repeat it: options fuse_attn / fuse_ffn are off now (five-launch step)", (unsigned)qerr[0]);
    }
    if (elapsed_ms) { IFA_HIP_CHECK(hipEventElapsedTime(elapsed_ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
    if (out_tokens_host) memcpy(out_tokens_host, m->host_pinned + 8, sizeof(int) * (size_t)n_steps);
    th("done");
    return IFA_OK;
}

} // namespace ifae

extern "C" {

int ifa_model_time_kernel(ifa_model *m, int which, int iters, float *avg_us)
{
    IFA_REQUIRE(m && m->finalized && avg_us, "ifa_model_time_kernel: bad arguments");
    IFA_REQUIRE(iters > 0 && which >= 0 && which <= 9, "ifa_model_time_kernel: which %d iters %d", which, iters);
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    std::string why;
    if (!fused_supported(m, &why)) return ifa_fail(IFA_ERR_STATE, "fused path unavailable: %s", why.c_str());
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    if (m->opt_q3h_native && (rc = ensure_q3hn(m))) return rc;
    hipStream_t s = m->stream;
    m->host_pinned[0] = 1; m->host_pinned[1] = std::min(m->cfg.max_ctx - 1, 64); m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, s));
    auto touch_layer = [&](int l) {
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            const Tensor &t = m->layers[(size_t)l].t[id];
            if (!t.tiled) continue;
            const size_t bytes = t.rows * ifa_tiled_row_bytes(t.dtype, t.cols);
            k_touch<<<dim3(8), dim3(256), 0, s>>>((const uint8_t *)t.tiled, bytes, (size_t)m->opt_touch_stride, m->state + 7);
        }
    };
    if (which == 7 || which == 9) {      // QKV + attention as one launch; the chained FFN launch
        if ((rc = qkv_attn_ready(m))) return rc;
        if (which == 9 && !m->ch_on) return ifa_fail(IFA_ERR_STATE, "chained FFN launch unavailable for this model / option set");
        if (which == 7 && !m->qa_on) return ifa_fail(IFA_ERR_STATE, "fused QKV + attention launch unavailable for this model / option set");
    }
    auto one = [&](int i) -> int {
        if (which == 7) return launch_qkv_attn(m, i % m->cfg.layers, m->x, (unsigned)(i + 1));
        if (which == 9) return launch_chain(m, i % m->cfg.layers, m->x, m->x2, (unsigned)(i + 1));
        const int l = m->opt_bench_mode == 1 ? 0 : i % m->cfg.layers;     // rotate over layers: distinct weights every launch
        if (m->opt_bench_mode == 2) touch_layer(l);
        switch (which) {
        case 0: return launch_qkv(m, l, m->x);
        case 1: return launch_attn(m, l);
        case 2: return launch_wo(m, l, m->x);
        case 3: return launch_ffn13(m, l, -1, m->x);
        case 4: return launch_w2(m, l, m->x2);
        default: return launch_lm(m, m->x);
        }
    };
    k_dec_gather<<<dim3(2), dim3(256), 0, s>>>((const half_t *)m->g[T_EMBD].data, m->state, m->cfg.dim, (int)m->g[T_EMBD].rows, m->x,
                                               m->cfg.rope_order ? m->rope_tab : nullptr, m->cfg.head_dim, m->cfg.rope_theta,
                                               (int)(m->cfg.head_dim * m->cfg.partial_rotary + 0.5f), m->cfg.embd_scale);
    for (int i = 0; i < 3; i++) if ((rc = one(i))) return rc;
    if (m->opt_trace) {
        if (!m->trace) IFA_HIP_CHECK(hipMalloc((void **)&m->trace, sizeof(long long) * 2048 * 8));
        IFA_HIP_CHECK(hipMemsetAsync(m->trace, 0, sizeof(long long) * 2048 * 8, s));
        g_trace_ptr = m->trace;
    }
    hipEvent_t e0, e1;
    IFA_HIP_CHECK(hipEventCreate(&e0)); IFA_HIP_CHECK(hipEventCreate(&e1));
    IFA_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) if ((rc = one(i))) return rc;
    IFA_HIP_CHECK(hipEventRecord(e1, s));
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    g_trace_ptr = nullptr;
    float ms = 0;
    IFA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = ms * 1000.0f / (float)iters;
    return IFA_OK;
}

} // extern "C"
