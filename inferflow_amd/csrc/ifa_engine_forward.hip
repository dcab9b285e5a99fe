// ifa_engine_forward.hip -- prompts and batched steps: the op-by-op layer through the C-ABI ops the reference worker would call
// (TensorOpr / TensorMul counterparts), the four-launch prompt routes, dynamic batching.
#include "ifa_engine_state.h"

namespace ifae {

__global__ void __launch_bounds__(256) k_gather_rows(const half_t *__restrict__ embd, const int *__restrict__ tokens,
                                                     int T, int dim, int vocab, half_t *__restrict__ x, float embd_scale)
{
    const int t = blockIdx.y;
    int tok = tokens[t];
    tok = min(max(tok, 0), vocab - 1);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < dim; c += gridDim.x * blockDim.x) {
        const half_t e = embd[(size_t)tok * dim + c];
        x[(size_t)t * dim + c] = embd_scale != 0.0f ? f2h(h2f(e) * embd_scale) : e;       // LinearNorm (inference_worker.cc:447-451)
    }
}

// rows src[idx[t]] -> dst[t] (the token embedding, or an expert's rows of a batch); idx clamped to [0, n_src)
int gather_rows(ifa_model *m, const half_t *src, const int *idx_dev, int T, int dim, int n_src, half_t *dst, float scale)
{
    k_gather_rows<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>(src, idx_dev, T, dim, n_src, dst, scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// MatrixMultiplicationEx + MatrixMultiplication dispatch (inference_worker.cc:2337-2432)
int matmul(ifa_model *m, const half_t *A, int T, const Tensor &W, const Tensor &bias, half_t *C)
{
    if (!W.present()) return ifa_fail(IFA_ERR_STATE, "missing weight tensor");
    const void *b = bias.present() ? bias.data : nullptr;
    const size_t K = W.cols, N = W.rows;
    ifa_stream s = m->stream;
    const bool use_gemv = (T == 1) && (K % 32 == 0);
    if (use_gemv && W.dtype != F16 && m->cfg.full_quant_gemv && ax8_eligible(W.dtype)) {
        int rc = ifa_quantize_act_q8(A, 1, K, m->xq, s);
        if (rc) return rc;
        return ifa_gemv(W.dtype, W.data, N, K, Q8_B32T2, m->xq, b, C, s);
    }
    // a handful of rows (dynamic batching, very short prompts): weight-streaming kernel on the tiled layout
    if (T >= 2 && T <= 16 && is_q4(W.dtype) && W.tiled && m->opt_gemm_rows) {
        int rc = ifa_gemm_rows_q4(W.tiled, N, K, A, (size_t)T, b, C, s);
        if (rc != IFA_ERR_STATE) return rc;
    }
    // T > 1: MFMA GEMM with the dequantisation fused in (the reference dequantises the whole
    // tensor and calls cublasGemmEx; same arithmetic: half weights x half activations, fp32 accumulate)
    if (T > 1 && K % 8 == 0) return ifa_gemm(W.dtype, W.data, N, K, A, (size_t)T, b, C, s);
    // T == 1 with ineligible types: weights dequantised to half, fp32 accumulate per row
    for (int t = 0; t < T; t++) {
        int rc = ifa_gemv(W.dtype, W.data, N, K, F16, A + (size_t)t * K, b, C + (size_t)t * N, s);
        if (rc) return rc;
    }
    return IFA_OK;
}

int norm_rows(ifa_model *m, const half_t *x, int T, const Tensor &w, const Tensor &b, half_t *y, float base)
{
    return ifa_layernorm(m->cfg.norm_kind, x, (size_t)T, (size_t)m->cfg.dim, w.present() ? w.data : nullptr,
                         b.present() ? b.data : nullptr, base, m->cfg.eps, y, m->stream);
}

int ffn_dense(ifa_model *m, const half_t *x, int T, const Tensor &w1, const Tensor &b1, const Tensor &w3, const Tensor &b3,
                     const Tensor &w2, const Tensor &b2, half_t *out, int perf_base)
{
    int rc;
    ifa_stream s = (ifa_stream)m->stream;
    // perf_stat keys of ProcessGpuLayer_FeedForward (inference_worker.cc:1790-1880): + 730 w1, + 750 w3, + 740 activation, + 760 Mul (here ONE
    // launch, under + 760), + 780 w2 (with its input quantiser, the reference's + 770)
    const bool st = perf_base > 0;
    { PerfSpan sp(m, perf_base + 730, st); if ((rc = matmul(m, x, T, w1, b1, m->t1))) return rc; }
    if (w3.present()) {
        { PerfSpan sp(m, perf_base + 750, st); if ((rc = matmul(m, x, T, w3, b3, m->t2))) return rc; }
        PerfSpan sp(m, perf_base + 760, st);
        if ((rc = ifa_activation_mul(m->cfg.act_kind, m->t1, m->t2, (size_t)T * w1.rows, m->t1, s))) return rc;
    } else {
        PerfSpan sp(m, perf_base + 740, st);
        if ((rc = ifa_activation(m->cfg.act_kind, 0, m->t1, (size_t)T, w1.rows, m->t1, s))) return rc;
    }
    PerfSpan sp(m, perf_base + 780, st);
    return matmul(m, m->t1, T, w2, b2, out);
}

// Everything of a layer behind the attention product in m->a (bias added / shards merged): TensorOpr::Scale of the attention
// output, the residual wiring, the optional post norms, the FFN (dense or mixture of experts) and the adds in front of what
// follows -- ProcessGpuLayer, inference_worker.cc:841-965.  Shared by the prompt path and the batched step.  x: the layer
// input (on return: the layer output, m->x / m->f exchanged); attn_in: the attention's normalised input (parallel attention feeds
// it to the FFN); xn_ready: m->xn holds the next norm's output already (fused into the last Add).
//   self_attn.post_norm (:857-866): residual = a (+ x); a' = Norm(residual); the FFN reads a'; is_attn_post_as_residual picks a' as
//   the residual too.  feed_forward.post_norm (:954-965): the layer output is Norm(ffn out + residual [+ x]).
int layer_tail_ops(ifa_model *m, int l, int T, half_t *&x, const half_t *attn_in, bool &xn_ready)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    ifa_stream s = m->stream;
    const size_t D = c.dim;
    const Tensor none;
    const bool merging = tp_merging(m), seq_wiring = !c.parallel_attn && !c.share_input;
    const bool a_post = L.t[T_ATTN_POST_NORM].present(), f_post = L.t[T_FFN_POST_NORM].present();
    int rc;
    // perf_stat (layer 0 = the reference's layer_idx_for_study_): + 710 the FFN's pre-norm (with the residual Add where both are one launch),
    // + 700 ProcessGpuLayer_FeedForward as a whole (norm .. w2), + 800 what follows it (Adds, post norm; inference_worker.cc:919-950)
    const int perf_base = m->opt_perf_stat && l == 0 ? (l + 1) * 10000 : 0;
    const bool st = perf_base > 0;
    PerfSpan sp_ffn(m, perf_base + 700, st);
    PerfSpan sp_norm(m, perf_base + 710, st);
    if (scale_on(c.attn_out_scale) && (rc = ifa_scale(m->a, c.attn_out_scale, (size_t)T * D, m->a, s))) return rc;
    const half_t *ff_in = c.parallel_attn ? attn_in : (c.share_input ? x : m->a);
    const half_t *ff_n = ff_in;
    const half_t *residual = m->a;
    if (a_post) {
        if (seq_wiring && (rc = ifa_add(x, m->a, (size_t)T * D, 0, m->a, s))) return rc;
        if ((rc = norm_rows(m, m->a, T, L.t[T_ATTN_POST_NORM], L.t[T_ATTN_POST_NORM_B], m->pn, 0.0f))) return rc;
        if (m->opt_attn_post_as_residual) residual = m->pn;
        if (!c.parallel_attn && !c.share_input) ff_in = m->pn;
        ff_n = ff_in;
        if (L.t[T_FFN_NORM].present()) {
            if ((rc = norm_rows(m, ff_in, T, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn, c.ffn_norm_base))) return rc;
            ff_n = m->hn;
        }
    } else if (seq_wiring && L.t[T_FFN_NORM].present()) {          // Add(x, attn out) + ffn norm
        if ((rc = ifa_add_layernorm(c.norm_kind, x, m->a, (size_t)T, D, L.t[T_FFN_NORM].data, L.t[T_FFN_NORM_B].present() ? L.t[T_FFN_NORM_B].data : nullptr,
                                    c.ffn_norm_base, c.eps, m->a, m->hn, s))) return rc;
        ff_n = m->hn;
    } else {
        if (seq_wiring && (rc = ifa_add(x, m->a, (size_t)T * D, 0, m->a, s))) return rc;
        if (L.t[T_FFN_NORM].present()) {
            if ((rc = norm_rows(m, ff_in, T, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn, c.ffn_norm_base))) return rc;
            ff_n = m->hn;
        }
    }
    sp_norm.done();
    if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
        if ((rc = moe_ffn(m, L, ff_n, T))) return rc;
        if ((rc = tp_merge_rows(m, m->f, T, none))) return rc;          // every expert sliced like the dense FFN: one merge of the weighted sums
    } else {
        if ((rc = ffn_dense(m, ff_n, T, L.t[T_W1], L.t[T_W1_B], L.t[T_W3], L.t[T_W3_B], L.t[T_W2], merging ? none : L.t[T_W2_B], m->f, perf_base))) return rc;
        if ((rc = tp_merge_rows(m, m->f, T, L.t[T_W2_B]))) return rc;
    }
    sp_ffn.done();
    PerfSpan sp_post(m, perf_base + 800, st);
    if (scale_on(c.ffn_out_scale) && (rc = ifa_scale(m->f, c.ffn_out_scale, (size_t)T * D, m->f, s))) return rc;
    // Add(ffn out, residual) + the norm in front of what comes next: the next layer's attention norm, or the output norm
    const bool last_layer = l + 1 == c.layers;
    const Tensor &nw = last_layer ? m->g[T_OUT_NORM] : m->layers[(size_t)l + 1].t[T_ATTN_NORM];
    const Tensor &nb = last_layer ? m->g[T_OUT_NORM_B] : m->layers[(size_t)l + 1].t[T_ATTN_NORM_B];
    xn_ready = false;
    if (!a_post && !f_post && seq_wiring && nw.present() && !(last_layer && scale_on(c.out_scale))) {
        if ((rc = ifa_add_layernorm(c.norm_kind, m->f, m->a, (size_t)T, D, nw.data, nb.present() ? nb.data : nullptr,
                                    last_layer ? c.out_norm_base : c.attn_norm_base, c.eps, m->f, m->xn, s))) return rc;
        xn_ready = true;
    } else {
        if ((rc = ifa_add(m->f, residual, (size_t)T * D, 0, m->f, s))) return rc;
        if (c.parallel_attn || c.share_input)
            if ((rc = ifa_add(m->f, x, (size_t)T * D, 0, m->f, s))) return rc;
        if (f_post) {
            if ((rc = norm_rows(m, m->f, T, L.t[T_FFN_POST_NORM], L.t[T_FFN_POST_NORM_B], m->hn, 0.0f))) return rc;
            std::swap(m->f, m->hn);
        }
    }
    std::swap(m->x, m->f);
    x = m->x;
    return IFA_OK;
}

// no_head: a chunk of a longer prompt that is not its last one -- the layers only (KV cache rows written), no lm_head / argmax / sync
// 33 .. prefill_mid_max tokens through k_gemm_mid (csrc/ifa_gemm_mid.hip): the large-tile route's conditions + an operand-order copy of
// every linear (built here on first use).  From 33 tokens on: 5.3 ms for 33..48 tokens against 5.9-6.6 for two passes of the rows
// GEMM (tools/short_prompt_routes.py, profiles/r06_prefill_mid_parts.log)
bool prefill_mid_ok(ifa_model *m, int T, bool any_length)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_prefill_mid || !m->opt_rows_mo || T <= 32 || (!any_length && T > m->opt_prefill_mid_max) || c.experts != 0 || !prefill_big_ok(m)) return false;
    if (ensure_mo(m) != IFA_OK) return false;
    for (int l = 0; l < c.layers; l++) {
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) { const Tensor &t = m->layers[(size_t)l].t[id]; if (!t.present() || !t.mo || t.cols % 128 != 0 || t.rows % 16 != 0) return false; }
    }
    return true;
}

int forward_ops(ifa_model *m, const int *tokens_host, int T, int prefix_len, void *logits_out, int *next_token, bool no_head)
{
    const ifa_model_config &c = m->cfg;
    if (T <= 0 || prefix_len < 0 || prefix_len + T > c.max_ctx)
        return ifa_fail(IFA_ERR_ARG, "forward: %d tokens at prefix %d exceed max_ctx %d", T, prefix_len, c.max_ctx);
    int rc = ensure_scratch(m, T);
    if (rc) return rc;
    ifa_stream s = m->stream;
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim;
    const ifa_tp_topology *tp = m->topo;            // multi-GPU partition (ifa_model_tp_prefill): merges + stage hand-over
    const bool first_stage = !tp || tp->stage == 0, last_stage = !tp || tp->next_rank < 0 || tp->n_stages == 1;
    const bool merging = tp_merging(m);
    if (first_stage && (!m->g[T_EMBD].present() || m->g[T_EMBD].dtype != F16)) return ifa_fail(IFA_ERR_STATE, "F16 embeddings not set");
    static const bool trace_host = getenv("IFA_TRACE_FORWARD") != nullptr;
    const auto host_t0 = std::chrono::steady_clock::now();
    if (first_stage) {
        PerfSpan sp_embd(m, 1);          // (key 1: the embedding rows, InferenceEngine::Infer_Std, inference_engine.cc:1168-1176)
        IFA_HIP_CHECK(hipMemcpyAsync(m->tokens_dev, tokens_host, sizeof(int) * (size_t)T, hipMemcpyHostToDevice, m->stream));
        k_gather_rows<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, m->tokens_dev, T, (int)D,
                                                                          (int)m->g[T_EMBD].rows, m->x, c.embd_scale);
        IFA_LAUNCH_CHECK();
    } else if ((rc = ifa_recv(tp->world, m->x, (size_t)T * D * 2, tp->prev_rank, s))) return rc;     // the previous group's [T][dim] output
    half_t *x = m->x;
    const Tensor none;
    // small element-wise ops are one launch where the wiring allows it (each keeps its own half rounding): RoPE(q) + RoPE(k)
    // + the F16 cache rows; the residual Add + the norm that follows it (a layer is ~17 launches otherwise, and at short
    // prompts every one of them is a fixed ~5 us)
    const bool seq_wiring = !c.parallel_attn && !c.share_input;
    bool xn_ready = false;           // m->xn already holds the next norm's output (fused into the previous layer's last Add)
    // prompts of 2..16 tokens on a dense Q4 model with the sequential RMS wiring: the linears of a layer as FOUR launches of
    // the rows GEMM (ifa_gemm_rows_mfma.hip) -- norm prologue + wq | wk | wv into q / k / v, wo + residual, norm + w1 / w3 +
    // GLU, w2 + residual -- instead of seven products and four element-wise launches (9..16 tokens: the norms stay launches)
    // Prompts above `prefill_big_min` tokens (47; round 4: 128) take the same four launches per layer from the large-tile GEMM (ifa_gemm.hip, k_gemm_big: the
    // weights dequantised once per workgroup and step into LDS; reference-layout rows), norms as their own launches.
    // the mid-length kernel (33 .. prefill_mid_max tokens): every linear of every layer a 4-bit tensor with its operand-order copy, dense FFN
    const bool pstat = m->opt_perf_stat != 0;       // per-phase times: the op-by-op layer below (one launch per reference op), never the fused routes
    const bool pf_mid = !tp && !pstat && prefill_mid_ok(m, T, false);
    const bool pf_big = pf_mid || (!tp && !pstat && T > std::max(32, m->opt_prefill_big_min) && prefill_big_ok(m));
    bool pf_fused = pf_big || (!tp && !pstat && T >= 2 && T <= 32 && batch_fused_ok(m, T) && c.experts == 0);
    if (pf_fused && !pf_big) {
        if ((rc = ensure_mo(m))) return rc;
        pf_fused = batch_fused_ok(m, T);          // (ensure_mo may have switched the copies off: ask again, see forward_batch)
    }
    if (pf_big && !pf_mid && (rc = ensure_x32(m))) return rc;
    // above prefill_mid_max: wo and w2 (4096 output columns: 128 x 128 tiles whatever the kernel) still through k_gemm_mid -- at 1024 tokens
    // 41 / 102 us per launch against 51 / 143 for the split-K halves of the large-tile kernel, which keeps wq | wk | wv and the gated pair
    // (profiles/r06_prefill_mid_parts.log section 13: 1024 / 1536 / 2048 tokens 4.5 / 3 / 0.8 % faster, 4096 tokens 4.7 % slower: up to 2048)
    const bool pf_res_mid = pf_big && !pf_mid && !tp && T <= m->opt_prefill_res_mid && prefill_mid_ok(m, T, true);
    for (int l = 0; l < c.layers && pf_fused; l++) {
        Layer &L = m->layers[l];
        const size_t F = c.ffn;
        const bool norm_fused = !pf_big && (T <= 8 || rows_mo(m, L.t[T_WQ])) && T <= 16;
        auto wp = [&](int id) { return pf_mid ? (const uint8_t *)L.t[id].mo : (pf_big ? (const uint8_t *)(L.t[id].x32 ? L.t[id].x32 : L.t[id].data) : rows_w(m, L.t[id])); };
        const int mo_flag = pf_mid ? 1 : (pf_big ? 0 : rows_mo(m, L.t[T_WQ]));
        auto lin = [&](const GmArgs &A, int id, int epi, int norm) {
            if (pf_mid && gemm_mid_ok(L.t[id].dtype, A, epi)) return gemm_mid(A, epi, m->stream);
            if (pf_mid) return ifa_fail(IFA_ERR_STATE, "mid-length GEMM declined a product of layer tensor %d", id);
            return pf_big ? gemm_big(L.t[id].x32 ? (int)Q4_B32T1A : L.t[id].dtype, A, epi, m->stream) : gemm_rows_mfma_launch(A, epi, norm, m->stream);
        };
        Tensor nob;
        GmArgs P;
        auto clear = [&]() { memset(&P, 0, sizeof(P)); P.T = T; P.eps = c.eps; P.act_kind = c.act_kind; P.mo = mo_flag; P.no_waits = pf_big ? !m->opt_gemm_splitk : !m->opt_rows_kparts; };
        clear();
        if (!norm_fused && (rc = norm_rows(m, x, T, L.t[T_ATTN_NORM], pf_big ? L.t[T_ATTN_NORM_B] : nob, m->xn, c.attn_norm_base))) return rc;
        P.W[0] = wp(T_WQ); P.W[1] = wp(T_WK); P.W[2] = wp(T_WV);
        P.rows[0] = (int)QD; P.rows[1] = (int)KVD; P.rows[2] = (int)KVD; P.nsets = 3; P.nblk = (int)(D / 32);
        P.X = norm_fused ? x : m->xn; P.ldx = (int)D;
        if (norm_fused) { P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.multi_base = c.attn_norm_base; }
        P.bias[0] = (const half_t *)L.t[T_WQ_B].data; P.bias[1] = (const half_t *)L.t[T_WK_B].data; P.bias[2] = (const half_t *)L.t[T_WV_B].data;
        P.Yset[0] = m->q; P.Yset[1] = m->k; P.Yset[2] = m->v; P.ldyset[0] = (int)QD; P.ldyset[1] = (int)KVD; P.ldyset[2] = (int)KVD;
        if ((rc = lin(P, T_WQ, GM_PLAIN, norm_fused ? 1 : 0))) return rc;
        uint8_t *kdst = (uint8_t *)L.kcache + (size_t)prefix_len * m->kv_row_bytes;
        uint8_t *vdst = (uint8_t *)L.vcache + (size_t)prefix_len * m->kv_row_bytes;
        const bool kv_f16 = c.kv_dtype != Q8_B32T2;
        bool kv_stored = false;
        if (c.rope_order != 0) {
            rc = ifa_rope_qk_store(m->q, m->k, m->v, c.head_dim, c.heads, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary,
                                   kv_f16 ? kdst : nullptr, kv_f16 ? vdst : nullptr, m->kv_row_bytes / 2, s);
            if (rc == IFA_OK) kv_stored = kv_f16;
            else if (rc != IFA_ERR_STATE) return rc;
            else {
                if ((rc = ifa_rope(m->q, c.head_dim, c.heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
                if ((rc = ifa_rope(m->k, c.head_dim, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
            }
        }
        if (!kv_f16) {
            if ((rc = ifa_quantize_act_q8(m->k, T, KVD, kdst, s))) return rc;
            if ((rc = ifa_quantize_act_q8(m->v, T, KVD, vdst, s))) return rc;
        } else if (!kv_stored) {
            IFA_HIP_CHECK(hipMemcpyAsync(kdst, m->k, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
            IFA_HIP_CHECK(hipMemcpyAsync(vdst, m->v, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
        }
        if ((rc = ifa_attention(m->q, L.kcache, L.vcache, c.kv_dtype, prefix_len + T, T, prefix_len, c.heads, c.kv_heads,
                                c.head_dim, c.use_alibi ? 1.0f : c.kq_scale, c.use_alibi, c.tp_rank * c.heads,
                                c.heads * std::max(1, c.tp_size), m->att, s))) return rc;
        clear();
        P.W[0] = wp(T_WO); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(QD / 32);
        P.X = m->att; P.ldx = (int)QD; P.bias[0] = (const half_t *)L.t[T_WO_B].data;
        P.Y = m->a; P.ldy = (int)D; P.res = x; P.ldres = (int)D;
        if (pf_res_mid) { P.W[0] = (const uint8_t *)L.t[T_WO].mo; P.mo = 1; P.no_waits = !m->opt_rows_kparts; }
        if (pf_res_mid && gemm_mid_ok(L.t[T_WO].dtype, P, GM_RESIDUAL)) { if ((rc = gemm_mid(P, GM_RESIDUAL, m->stream))) return rc; }
        else { if (pf_res_mid) { P.W[0] = wp(T_WO); P.mo = mo_flag; } if ((rc = lin(P, T_WO, GM_RESIDUAL, 0))) return rc; }
        clear();
        if (!norm_fused && (rc = norm_rows(m, m->a, T, L.t[T_FFN_NORM], pf_big ? L.t[T_FFN_NORM_B] : nob, m->hn, c.ffn_norm_base))) return rc;
        if (c.experts > 0 && L.t[T_MOE_GATE].present()) {      // (pf_big only) mixture of experts: the attention half fused, the expert FFNs device-routed
            if ((rc = moe_ffn(m, L, m->hn, T))) return rc;
            if ((rc = ifa_add(m->f, m->a, (size_t)T * D, 0, m->f, s))) return rc;
            std::swap(m->x, m->f);
            x = m->x;
            continue;
        }
        P.W[0] = wp(T_W1); P.W1 = wp(T_W3); P.rows[0] = (int)F; P.nsets = 1; P.nblk = (int)(D / 32);
        P.X = norm_fused ? m->a : m->hn; P.ldx = (int)D;
        if (norm_fused) { P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.multi_base = c.ffn_norm_base; }
        P.bias[0] = (const half_t *)L.t[T_W1_B].data; P.bias1 = (const half_t *)L.t[T_W3_B].data;
        P.Y = m->t1; P.ldy = (int)F;
        if ((rc = lin(P, T_W1, GM_GLU, norm_fused ? 1 : 0))) return rc;
        clear();
        P.W[0] = wp(T_W2); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(F / 32);
        P.X = m->t1; P.ldx = (int)F; P.bias[0] = (const half_t *)L.t[T_W2_B].data;
        P.Y = m->f; P.ldy = (int)D; P.res = m->a; P.ldres = (int)D;
        if (pf_res_mid) { P.W[0] = (const uint8_t *)L.t[T_W2].mo; P.mo = 1; P.no_waits = !m->opt_rows_kparts; }
        if (pf_res_mid && gemm_mid_ok(L.t[T_W2].dtype, P, GM_RESIDUAL)) { if ((rc = gemm_mid(P, GM_RESIDUAL, m->stream))) return rc; }
        else { if (pf_res_mid) { P.W[0] = wp(T_W2); P.mo = mo_flag; } if ((rc = lin(P, T_W2, GM_RESIDUAL, 0))) return rc; }
        std::swap(m->x, m->f);
        x = m->x;
    }
    for (int l = pf_fused ? c.layers : 0; l < c.layers; l++) {
        Layer &L = m->layers[l];
        // perf_stat spans in the reference's key space (inference_worker.cc:296-322, 806-950, 1030-1400): the whole layer under + 0 for layers
        // 0..5; for layer 0 (layer_idx_for_study_) + 300 = ProcessGpuLayer_Attention as a whole, + 10 its pre-norm, + 30 the q / k / v products
        // (with their input quantisers: the reference's + 20), + 50 RoPE + cache rows, + 60 scores / softmax / V product, + 90 wo (+ bias, merge)
        const int pbase = (l + 1) * 10000;
        const bool st = pstat && l == 0;
        PerfSpan sp_layer(m, pbase + 0, pstat && l <= 5);
        PerfSpan sp_attn(m, pbase + 300, st);
        const half_t *attn_in = x;
        if (L.t[T_ATTN_NORM].present()) {
            PerfSpan sp(m, pbase + 10, st);
            if (!xn_ready && (rc = norm_rows(m, x, T, L.t[T_ATTN_NORM], L.t[T_ATTN_NORM_B], m->xn, c.attn_norm_base))) return rc;
            attn_in = m->xn;
        }
        xn_ready = false;
        {
            PerfSpan sp(m, pbase + 30, st);
            if ((rc = matmul(m, attn_in, T, L.t[T_WQ], L.t[T_WQ_B], m->q))) return rc;
            if ((rc = matmul(m, attn_in, T, L.t[T_WK], L.t[T_WK_B], m->k))) return rc;
            if ((rc = matmul(m, attn_in, T, L.t[T_WV], L.t[T_WV_B], m->v))) return rc;
        }
        PerfSpan sp_rope(m, pbase + 50, st);
        uint8_t *kdst = (uint8_t *)L.kcache + (size_t)prefix_len * m->kv_row_bytes;
        uint8_t *vdst = (uint8_t *)L.vcache + (size_t)prefix_len * m->kv_row_bytes;
        const bool kv_f16 = c.kv_dtype != Q8_B32T2;
        bool kv_stored = false;
        if (c.rope_order != 0) {
            rc = ifa_rope_qk_store(m->q, m->k, m->v, c.head_dim, c.heads, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary,
                                   kv_f16 ? kdst : nullptr, kv_f16 ? vdst : nullptr, m->kv_row_bytes / 2, s);
            if (rc == IFA_OK) kv_stored = kv_f16;
            else if (rc != IFA_ERR_STATE) return rc;
            else {
                if ((rc = ifa_rope(m->q, c.head_dim, c.heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
                if ((rc = ifa_rope(m->k, c.head_dim, c.kv_heads, T, prefix_len, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
            }
        }
        if (!kv_f16) {
            if ((rc = ifa_quantize_act_q8(m->k, T, KVD, kdst, s))) return rc;
            if ((rc = ifa_quantize_act_q8(m->v, T, KVD, vdst, s))) return rc;
        } else if (!kv_stored) {
            IFA_HIP_CHECK(hipMemcpyAsync(kdst, m->k, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
            IFA_HIP_CHECK(hipMemcpyAsync(vdst, m->v, (size_t)T * m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
        }
        sp_rope.done();
        {
            PerfSpan sp(m, pbase + 60, st);
            if ((rc = ifa_attention(m->q, L.kcache, L.vcache, c.kv_dtype, prefix_len + T, T, prefix_len, c.heads, c.kv_heads,
                                    c.head_dim, c.use_alibi ? 1.0f : c.kq_scale, c.use_alibi, c.tp_rank * c.heads,
                                    c.heads * std::max(1, c.tp_size), m->att, s))) return rc;
        }
        {
            PerfSpan sp(m, pbase + 90, st);
            if ((rc = matmul(m, m->att, T, L.t[T_WO], merging ? none : L.t[T_WO_B], m->a))) return rc;
            if ((rc = tp_merge_rows(m, m->a, T, L.t[T_WO_B]))) return rc;       // BY_TENSOR: sum of the ranks' partial products, bias after
        }
        sp_attn.done();
        if ((rc = layer_tail_ops(m, l, T, x, attn_in, xn_ready))) return rc;
    }
    if (!last_stage) {       // BY_LAYER / HYBRID: hand the [T][dim] output to the next device group, then learn the token
        if ((rc = ifa_send(tp->world, x, (size_t)T * D * 2, tp->next_rank, s))) return rc;
        if ((rc = tp_argmax_scratch(m, 1))) return rc;
        if ((rc = ifa_broadcast(tp->world, m->tp_tok, 4, tp->token_src, s))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->tp_tok, sizeof(int), hipMemcpyDeviceToHost, m->stream));
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
        if (next_token) *next_token = m->host_pinned[0];
        return IFA_OK;
    }
    if (no_head && !logits_out) return IFA_OK;
    PerfSpan sp_out(m, 1000009);         // (ProcessPostLayer: output norm + lm_head, inference_worker.cc:326-336, 673-675)
    if (scale_on(c.out_scale) && (rc = ifa_scale(x, c.out_scale, (size_t)T * D, x, s))) return rc;
    const half_t *hfin = x;
    if (m->g[T_OUT_NORM].present()) {
        if (!xn_ready && (rc = norm_rows(m, x, T, m->g[T_OUT_NORM], m->g[T_OUT_NORM_B], m->xn, c.out_norm_base))) return rc;
        hfin = m->xn;
    } else {
        IFA_HIP_CHECK(hipMemcpyAsync(m->xn, x, (size_t)T * D * 2, hipMemcpyDeviceToDevice, m->stream));
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    const size_t V = lm.rows;                        // (this rank's vocabulary shard under tensor parallelism)
    int t0 = logits_out ? 0 : T - 1;
    if (logits_out) { if ((rc = matmul(m, hfin, T, lm, none, m->logits))) return rc; }
    else if (lm.dtype == F16 && D % 8 == 0 && D <= 8192) {
        // the last row only: the decode step's lm_head kernel on the normalised row (same per-row chain as the op-level GEMV --
        // bit-identical logits -- at 6 TB/s instead of 1.1: 232 -> 45 us per prompt, rocprofv3 r06)
        DecLmHeadParams H2; memset(&H2, 0, sizeof(H2));
        H2.x = hfin + (size_t)t0 * D; H2.eps = c.eps; H2.cols = (int)D; H2.W = (const half_t *)lm.data; H2.logits = m->logits + (size_t)t0 * V; H2.rows = (int)V;
        if ((rc = launch_lmhead(H2, 0, m->opt_rpw_lm, m->stream))) return rc;
    }
    else { if ((rc = matmul(m, hfin + (size_t)t0 * D, 1, lm, none, m->logits + (size_t)t0 * V))) return rc; }
    sp_out.done();
    if (logits_out) IFA_HIP_CHECK(hipMemcpyAsync(logits_out, m->logits, (size_t)T * V * 2, hipMemcpyDeviceToDevice, m->stream));
    if (tp) {                // distributed argmax of the last row over the group's shards (+ announcement to the other groups)
        if ((rc = tp_pick_rows(m, *tp, m->logits + (size_t)(T - 1) * V, V, (int)V, 1))) return rc;
        if (tp->n_stages > 1 && (rc = ifa_broadcast(tp->world, m->tp_tok, 4, tp->token_src, s))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->tp_tok, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    } else {
        if ((rc = ifa_argmax_masked(m->logits + (size_t)(T - 1) * V, V, m->state + 3, m->state, s))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->state, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    }
    const auto host_t1 = std::chrono::steady_clock::now();
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (pstat && (rc = perf_collect(m))) return rc;
    if ((rc = wait_err_check("forward step"))) { drop_graphs(m); return rc; }      // (a split-K / K-parts wait gave up: the step is not valid; those launches are off now)
    if (trace_host)      // how much of a step is the host enqueuing (launch-bound) vs the GPU draining what was enqueued
        fprintf(stderr, "forward T=%d: enqueue %.3f ms, total %.3f ms\n", T, std::chrono::duration<double, std::milli>(host_t1 - host_t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count());
    if (next_token) *next_token = m->host_pinned[0];
    return IFA_OK;
}

// row r of k / v -> position rows[r].n_ctx - 1 of its query's cache (F16 copy or Q8_B32T2 quantisation, 32 lanes per block)
template <bool Q8>
__global__ void __launch_bounds__(256) k_kv_store_rows(const half_t *__restrict__ k, const half_t *__restrict__ v, int kv_dim,
                                                       size_t row_bytes, const AttnRowH *__restrict__ rows)
{
    const int r = blockIdx.x, which = blockIdx.y;
    const half_t *src = (which ? v : k) + (size_t)r * kv_dim;
    uint8_t *dst = (uint8_t *)(which ? rows[r].vc : rows[r].kc) + (size_t)(rows[r].n_ctx - 1) * row_bytes;
    if constexpr (!Q8) {
        for (int c = threadIdx.x; c < kv_dim; c += 256) reinterpret_cast<half_t *>(dst)[c] = src[c];
    } else {
        const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
        for (int b = grp; b < kv_dim / 32; b += 8) {      // Tensor_QuantizeQ8_B32T2_Alg2_Kernel (tensor_quant.h:44-82)
            const float val = h2f(src[b * 32 + lane]);
            float mx = fabsf(val);
#pragma unroll
            for (int m2 = 16; m2 > 0; m2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m2, 32));
            const float sc = mx / 127;
            int qv = sc <= 0.000001f ? 0 : (int)roundf(val / sc);
            qv = min(max(qv, -128), 127);
            uint8_t *blk = dst + (size_t)b * 34;
            blk[2 + lane] = (uint8_t)(int8_t)qv;
            if (lane == 0) *reinterpret_cast<uint16_t *>(blk) = __builtin_bit_cast(uint16_t, f2h(sc));
        }
    }
}

// ---- the batched step as five launches per layer (the structure of the batch-1 step: ifa_gemm_rows_mfma.hip with the norm
// prologue / GLU / residual epilogues, k_dec_attn<.., BATCH>): dense models with the sequential RMS wiring, every linear in
// tiled Q4_B32T1, 2..16 queries.  Everything else takes the op-by-op rows below.
bool batch_fused_ok(const ifa_model *m, int n)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_batch_fused || !m->opt_gemm_rows || !gemm_rows_use_mfma() || n < 2 || n > (m->opt_rows_mo ? 32 : 16) || m->topo) return false;      // (17..32 rows: MO copies only)
    if (has_post_norms(m)) return false;
    if (c.norm_kind != 0 || c.parallel_attn || c.share_input) return false;
    if (scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale) || scale_on(c.out_scale)) return false;
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim, F = c.ffn;
    if (D % 128 || QD % 128 || F % 128 || D > 4096 || KVD % 16 || D % 16 || F % 16) return false;
    if (c.head_dim != 32 && c.head_dim != 48 && c.head_dim != 64 && c.head_dim != 80 && c.head_dim != 96 && c.head_dim != 128) return false;
    if (c.kv_dtype == Q8_B32T2 && c.head_dim % 32 != 0) return false;
    if (dec_attn_smem(c.head_dim, c.max_ctx) > IFA_LDS_LIMIT) return false;
    for (const Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();      // MoE layers: the attention half is fused, the FFN runs moe_ffn
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            if (moe && (id == T_W1 || id == T_W3 || id == T_W2)) continue;
            if (!L.t[id].present() || !L.t[id].tiled) return false;
            if (!is_q4(L.t[id].dtype) && !(m->opt_rows_mo && rows_mo_fmt(L.t[id].dtype) && L.t[id].cols % 128 == 0)) return false;
        }
        if (moe && !moe_device_ok(m, L)) return false;
        if (!L.t[T_ATTN_NORM].present() || !L.t[T_FFN_NORM].present() || L.t[T_ATTN_NORM_B].present() || L.t[T_FFN_NORM_B].present()) return false;
    }
    return true;
}

// prompts above `prefill_big_min` tokens as four launches of the large-tile GEMM per layer (forward_ops, pf_big): dense layers with the
// sequential wiring, every linear a 20-byte-block Q4 tensor (wq / wk / wv of one format), dims in multiples of 64
bool prefill_big_ok(const ifa_model *m)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_prefill_big || m->topo || c.parallel_attn || c.share_input || has_post_norms(m)) return false;
    if (scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale)) return false;
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, F = c.ffn;
    if (D % 64 || QD % 64 || F % 64) return false;
    for (const Layer &L : m->layers) {
        const bool moe = c.experts > 0 && L.t[T_MOE_GATE].present();      // MoE layers: the attention half is fused, the FFN runs moe_ffn
        const int ids[] = {T_WQ, T_WK, T_WV, T_WO, T_W1, T_W3, T_W2};
        for (int id : ids) {
            if (moe && (id == T_W1 || id == T_W3 || id == T_W2)) continue;
            const bool b64 = (L.t[id].dtype == Q4_B64T1 || L.t[id].dtype == Q3H_B64T1) && L.t[id].tiled;      // via their Q4_B32T1A-layout copy (ensure_x32)
            if (!L.t[id].present() || !L.t[id].data || (L.t[id].dtype != Q4_B32T1A && L.t[id].dtype != Q4_B32T1B && !b64)) return false;
        }
        if (L.t[T_WK].dtype != L.t[T_WQ].dtype || L.t[T_WV].dtype != L.t[T_WQ].dtype || (!moe && L.t[T_W3].dtype != L.t[T_W1].dtype)) return false;
        if (!L.t[T_ATTN_NORM].present() || !L.t[T_FFN_NORM].present()) return false;
    }
    return true;
}

int batch_fused_layer(ifa_model *m, int l, int n, const half_t *x, half_t *xnext, const void *rows_l)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    const size_t D = c.dim, QD = (size_t)c.heads * c.head_dim, KVD = (size_t)c.kv_heads * c.head_dim, F = c.ffn;
    int rc;
    GmArgs P;
    auto clear = [&]() { memset(&P, 0, sizeof(P)); P.T = n; P.eps = c.eps; P.act_kind = c.act_kind; P.no_waits = !m->opt_rows_kparts; };
    // 1. RmsNorm -> wq | wk | wv  (one virtual row space, one [n][q | k | v] output)
    clear();
    P.W[0] = rows_w(m, L.t[T_WQ]); P.W[1] = rows_w(m, L.t[T_WK]); P.W[2] = rows_w(m, L.t[T_WV]); P.mo = rows_mo(m, L.t[T_WQ]);
    P.rows[0] = (int)QD; P.rows[1] = (int)KVD; P.rows[2] = (int)KVD; P.nsets = 3; P.nblk = (int)(D / 32);
    // (9..16 queries: the activation rows are staged in chunks of 2048 columns, so the norm runs as its own launch)
    const bool norm_fused = (n <= 8 || rows_mo(m, L.t[T_WQ])) && n <= 16;      // (MO layout: 16 rows x 4096 columns are one chunk too; 17..32 rows: chunked)
    Tensor nob;
    if (!norm_fused && (rc = norm_rows(m, x, n, L.t[T_ATTN_NORM], nob, m->xn, c.attn_norm_base))) return rc;
    P.X = norm_fused ? x : m->xn; P.ldx = (int)D;
    if (norm_fused) { P.norm_w = (const half_t *)L.t[T_ATTN_NORM].data; P.multi_base = c.attn_norm_base; }
    P.bias[0] = (const half_t *)L.t[T_WQ_B].data; P.bias[1] = (const half_t *)L.t[T_WK_B].data; P.bias[2] = (const half_t *)L.t[T_WV_B].data;
    P.Y = m->bqkv; P.ldy = (int)(QD + 2 * KVD);
    if ((rc = gemm_rows_mfma_launch(P, GM_PLAIN, norm_fused ? 1 : 0, m->stream))) return rc;
    // 2. RoPE, KV store, attention of every query on its own cache
    {
        const int rope_dims = (int)(c.head_dim * c.partial_rotary + 0.5f);
        DecAttnParams A; memset(&A, 0, sizeof(A));
        A.q = m->bqkv; A.k_new = m->bqkv + QD; A.v_new = A.k_new + KVD;
        A.state = m->state; A.rope_tab = m->brope; A.heads = c.heads; A.kv_heads = c.kv_heads;
        A.kv_q8 = c.kv_dtype == Q8_B32T2; A.kq_scale = c.use_alibi ? 1.0f : c.kq_scale;
        A.rope_order = c.rope_order; A.rope_cols = rope_dims;
        A.alibi = c.use_alibi; A.alibi_base = c.tp_rank * c.heads; A.alibi_total = c.heads * std::max(1, c.tp_size);
        A.out = m->att; A.max_ctx = c.max_ctx; A.batch_rows = rows_l; A.q_stride = (int)(QD + 2 * KVD);
        const size_t asmem = dec_attn_smem(c.head_dim, c.max_ctx);
        const dim3 grid((unsigned)c.heads, (unsigned)n), block(256);
#define IFA_BATTN(HDV, Q8V) { auto kern = k_dec_attn<HDV, Q8V, true>; \
        if (asmem > 48 * 1024) IFA_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asmem)); \
        kern<<<grid, block, asmem, m->stream>>>(A.q, nullptr, nullptr, A.heads, A.kv_heads, A); }
        switch (c.head_dim) {
        case 32: if (A.kv_q8) IFA_BATTN(32, true) else IFA_BATTN(32, false) break;
        case 64: if (A.kv_q8) IFA_BATTN(64, true) else IFA_BATTN(64, false) break;
        case 96: if (A.kv_q8) IFA_BATTN(96, true) else IFA_BATTN(96, false) break;
        case 128: if (A.kv_q8) IFA_BATTN(128, true) else IFA_BATTN(128, false) break;
        case 48: IFA_BATTN(48, false) break;
        case 80: IFA_BATTN(80, false) break;
        default: return ifa_fail(IFA_ERR_ARG, "fused batched attention: head_dim %d", c.head_dim);
        }
#undef IFA_BATTN
        IFA_LAUNCH_CHECK();
    }
    // 3. wo (+ bias) + residual
    clear();
    P.W[0] = rows_w(m, L.t[T_WO]); P.mo = rows_mo(m, L.t[T_WO]); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(QD / 32);
    P.X = m->att; P.ldx = (int)QD; P.bias[0] = (const half_t *)L.t[T_WO_B].data;
    P.Y = m->a; P.ldy = (int)D; P.res = x; P.ldres = (int)D;
    if ((rc = gemm_rows_mfma_launch(P, GM_RESIDUAL, 0, m->stream))) return rc;
    if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
        // mixture of experts: norm, the device-routed expert FFNs over the n rows (moe_ffn_device), residual
        Tensor none;
        if (moe_device_ok(m, L) && moe_router_rows_ok(m, L, n)) return moe_ffn_device(m, L, m->hn, n, m->a, m->a, xnext);
        {
            if ((rc = norm_rows(m, m->a, n, L.t[T_FFN_NORM], none, m->hn, c.ffn_norm_base))) return rc;
            if ((rc = moe_ffn(m, L, m->hn, n))) return rc;
        }
        return ifa_add(m->f, m->a, (size_t)n * D, 0, xnext, (ifa_stream)m->stream);
    }
    // 4. RmsNorm -> w1, w3 -> act(w1 x) * (w3 x)
    clear();
    P.W[0] = rows_w(m, L.t[T_W1]); P.W1 = rows_w(m, L.t[T_W3]); P.mo = rows_mo(m, L.t[T_W1]); P.rows[0] = (int)F; P.nsets = 1; P.nblk = (int)(D / 32);
    if (!norm_fused && (rc = norm_rows(m, m->a, n, L.t[T_FFN_NORM], nob, m->hn, c.ffn_norm_base))) return rc;
    P.X = norm_fused ? m->a : m->hn; P.ldx = (int)D;
    if (norm_fused) { P.norm_w = (const half_t *)L.t[T_FFN_NORM].data; P.multi_base = c.ffn_norm_base; }
    P.bias[0] = (const half_t *)L.t[T_W1_B].data; P.bias1 = (const half_t *)L.t[T_W3_B].data;
    P.Y = m->t1; P.ldy = (int)F;
    if ((rc = gemm_rows_mfma_launch(P, GM_GLU, norm_fused ? 1 : 0, m->stream))) return rc;
    // 5. w2 (+ bias) + residual -> the next layer's input
    clear();
    P.W[0] = rows_w(m, L.t[T_W2]); P.mo = rows_mo(m, L.t[T_W2]); P.rows[0] = (int)D; P.nsets = 1; P.nblk = (int)(F / 32);
    P.X = m->t1; P.ldx = (int)F; P.bias[0] = (const half_t *)L.t[T_W2_B].data;
    P.Y = xnext; P.ldy = (int)D; P.res = m->a; P.ldres = (int)D;
    return gemm_rows_mfma_launch(P, GM_RESIDUAL, 0, m->stream);
}

int forward_batch(ifa_model *m, int n, const int *tokens_host, const int *pos_host, const int *slot_host, int *next_tokens,
                         void *logits_out)
{
    const ifa_model_config &c = m->cfg;
    const int n_slots = m->slots.empty() ? 1 : (int)m->slots.size();
    int max_ctx = 0;
    for (int r = 0; r < n; r++) {
        if (pos_host[r] < 0 || pos_host[r] >= c.max_ctx) return ifa_fail(IFA_ERR_ARG, "decode_batch: position %d outside max_ctx %d", pos_host[r], c.max_ctx);
        if (slot_host[r] < 0 || slot_host[r] >= n_slots) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d of %d", slot_host[r], n_slots);
        for (int r2 = 0; r2 < r; r2++) if (slot_host[r2] == slot_host[r]) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d used twice", slot_host[r]);
        max_ctx = std::max(max_ctx, pos_host[r] + 1);
    }
    int rc = ensure_scratch(m, n);
    if (rc) return rc;
    ifa_stream s = m->stream;
    const int T = n;
    const size_t D = c.dim, KVD = (size_t)c.kv_heads * c.head_dim, L_ = m->layers.size();
    if (!m->g[T_EMBD].present() || m->g[T_EMBD].dtype != F16) return ifa_fail(IFA_ERR_STATE, "F16 embeddings not set");
    const ifa_tp_topology *tp = m->topo;            // tensor-parallel group (ifa_model_tp_decode_batch): merges + distributed argmax
    if (tp && tp->n_stages > 1) return ifa_fail(IFA_ERR_ARG, "decode_batch: layer groups are not batched (tensor-parallel groups only)");
    const bool merging = tp_merging(m);
    // per-step tables: positions, and for every layer the (k cache, v cache, context) of each row's query
    const size_t tab_bytes = L_ * (size_t)n * sizeof(AttnRowH) + 2 * (size_t)n * sizeof(int);
    if (tab_bytes > m->batch_tab_bytes) {
        drop_graphs(m);                            // the captured steps hold the old table addresses
        if (m->batch_tab_dev) IFA_HIP_CHECK(hipFree(m->batch_tab_dev));
        if (m->batch_tab_pin) IFA_HIP_CHECK(hipHostFree(m->batch_tab_pin));
        IFA_HIP_CHECK(hipMalloc(&m->batch_tab_dev, tab_bytes));
        IFA_HIP_CHECK(hipHostMalloc(&m->batch_tab_pin, tab_bytes, hipHostMallocDefault));
        m->batch_tab_bytes = tab_bytes;
    }
    AttnRowH *rows_h = (AttnRowH *)m->batch_tab_pin;
    int *pos_pin = (int *)(rows_h + L_ * (size_t)n);
    for (size_t l = 0; l < L_; l++)
        for (int r = 0; r < n; r++) {
            AttnRowH &a = rows_h[l * (size_t)n + r];
            a.kc = kv_ptr(m, l, slot_host[r], false); a.vc = kv_ptr(m, l, slot_host[r], true); a.n_ctx = pos_host[r] + 1; a.pad = 0;
        }
    for (int r = 0; r < n; r++) { pos_pin[r] = pos_host[r]; pos_pin[n + r] = tokens_host[r]; }
    const AttnRowH *rows_d = (const AttnRowH *)m->batch_tab_dev;
    const int *pos_d = (const int *)(rows_d + L_ * (size_t)n);
    const int *tok_d = pos_d + n;
    // Everything the device does in a step depends on the step only through the tables above (fixed addresses), so
    // for dense models the whole step -- table upload included -- is captured once per batch size and replayed.
    bool has_moe = false;
    for (const Layer &Lc : m->layers) has_moe = has_moe || (c.experts > 0 && Lc.t[T_MOE_GATE].present());
    // (measured on Llama-2-7B Q4: the batched step is bound by the small-T GEMM kernels, ~6.7 ms with or without the
    //  graph, so replay is opt-in: set_option("batch_graph", 1))
    bool fused = batch_fused_ok(m, n);          // five launches per layer: launch-bound without a graph, so it is replayed
    if (fused) {
        int rcm = ensure_mo(m); if (rcm) return rcm;
        // ensure_mo may have DOWNGRADED the model (the copies did not fit: opt_rows_mo = 0): what batch_fused_ok answered with the
        // copies in view -- up to 32 rows, the 64-weight formats -- no longer holds, so it is asked again before a path or a graph
        // is chosen; a step the tiled kernels do not cover takes the op-by-op rows below
        fused = batch_fused_ok(m, n);
        if (fused && (rcm = gemm_rows_kparts_reserve(m->stream))) return rcm;
    }
    if (has_moe && m->opt_moe_overlap) { int rcs = ensure_side_stream(m); if (rcs) return rcs; }
    // (MoE layers of the fused step route on the device -- no host round trip -- so they are captured too)
    const bool use_graph = (m->opt_batch_graph || fused) && m->opt_graph && (!has_moe || fused) && !logits_out && !tp;
    const int attn_ctx = use_graph ? c.max_ctx : max_ctx;     // LDS sizing of the attention kernel must not depend on the step
    if (use_graph) {
        auto it = m->batch_graphs.find(n);
        if (it != m->batch_graphs.end()) {
            IFA_HIP_CHECK(hipGraphLaunch(it->second, m->stream));
            IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
            if ((rc = wait_err_check("batched decode step"))) { drop_graphs(m); return rc; }      // (the captured steps hold K-parts launches: re-captured without them)
            if (next_tokens) for (int r = 0; r < n; r++) next_tokens[r] = m->host_pinned[8 + r];
            return IFA_OK;
        }
        IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
        IFA_HIP_CHECK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    }
    auto body = [&]() -> int {
    IFA_HIP_CHECK(hipMemcpyAsync(m->batch_tab_dev, m->batch_tab_pin, tab_bytes, hipMemcpyHostToDevice, m->stream));
    if (fused)
        k_dec_batch_gather<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, tok_d, pos_d, (int)D, (int)m->g[T_EMBD].rows,
                                                                              m->x, c.rope_order ? m->brope : nullptr, c.head_dim, c.rope_theta,
                                                                              (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    else
        k_gather_rows<<<dim3(4, (unsigned)T), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, tok_d, T, (int)D,
                                                                          (int)m->g[T_EMBD].rows, m->x, c.embd_scale);
    IFA_LAUNCH_CHECK();
    half_t *x = m->x;
    const Tensor none;
    const bool seq_wiring = !c.parallel_attn && !c.share_input;
    bool xn_ready = false;           // see forward_ops: every residual Add is fused with the norm that follows it
    if (fused) {
        for (int l = 0; l < c.layers; l++) {
            if ((rc = batch_fused_layer(m, l, n, x, m->f, rows_d + (size_t)l * (size_t)n))) return rc;
            std::swap(m->x, m->f);
            x = m->x;
        }
    }
    for (int l = fused ? c.layers : 0; l < c.layers; l++) {
        Layer &L = m->layers[(size_t)l];
        const half_t *attn_in = x;
        if (L.t[T_ATTN_NORM].present()) {
            if (!xn_ready && (rc = norm_rows(m, x, T, L.t[T_ATTN_NORM], L.t[T_ATTN_NORM_B], m->xn, c.attn_norm_base))) return rc;
            attn_in = m->xn;
        }
        xn_ready = false;
        if ((rc = matmul(m, attn_in, T, L.t[T_WQ], L.t[T_WQ_B], m->q))) return rc;
        if ((rc = matmul(m, attn_in, T, L.t[T_WK], L.t[T_WK_B], m->k))) return rc;
        if ((rc = matmul(m, attn_in, T, L.t[T_WV], L.t[T_WV_B], m->v))) return rc;
        if (c.rope_order != 0) {
            if ((rc = ifa_rope_rows(m->q, c.head_dim, c.heads, T, pos_d, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
            if ((rc = ifa_rope_rows(m->k, c.head_dim, c.kv_heads, T, pos_d, c.rope_theta, c.rope_order, c.partial_rotary, s))) return rc;
        }
        const AttnRowH *lr = rows_d + (size_t)l * (size_t)n;
        if (c.kv_dtype == Q8_B32T2) k_kv_store_rows<true><<<dim3((unsigned)n, 2), dim3(256), 0, m->stream>>>(m->k, m->v, (int)KVD, m->kv_row_bytes, lr);
        else k_kv_store_rows<false><<<dim3((unsigned)n, 2), dim3(256), 0, m->stream>>>(m->k, m->v, (int)KVD, m->kv_row_bytes, lr);
        IFA_LAUNCH_CHECK();
        if ((rc = ifa_attention_rows(m->q, lr, c.kv_dtype, n, attn_ctx, c.heads, c.kv_heads, c.head_dim, c.use_alibi ? 1.0f : c.kq_scale,
                                     c.use_alibi, c.tp_rank * c.heads, c.heads * std::max(1, c.tp_size), m->att, s))) return rc;
        if ((rc = matmul(m, m->att, T, L.t[T_WO], merging ? none : L.t[T_WO_B], m->a))) return rc;
        if ((rc = tp_merge_rows(m, m->a, T, L.t[T_WO_B]))) return rc;
        if ((rc = layer_tail_ops(m, l, T, x, attn_in, xn_ready))) return rc;
    }
    if (scale_on(c.out_scale) && (rc = ifa_scale(x, c.out_scale, (size_t)T * D, x, s))) return rc;
    const half_t *hfin = x;
    if (m->g[T_OUT_NORM].present()) {
        if (!xn_ready && (rc = norm_rows(m, x, T, m->g[T_OUT_NORM], m->g[T_OUT_NORM_B], m->xn, c.out_norm_base))) return rc;
        hfin = m->xn;
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    const size_t V = lm.rows;
    if ((rc = matmul(m, hfin, T, lm, none, m->logits))) return rc;
    if (logits_out) IFA_HIP_CHECK(hipMemcpyAsync(logits_out, m->logits, (size_t)T * V * 2, hipMemcpyDeviceToDevice, m->stream));
    if (tp) {                // one distributed argmax per row over the group's vocabulary shards
        if ((rc = tp_pick_rows(m, *tp, m->logits, V, (int)V, n))) return rc;
        IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->tp_tok, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, m->stream));
        return IFA_OK;
    }
    if ((rc = ifa_argmax_rows(m->logits, V, V, (size_t)n, m->state + 8, m->state + 3, s))) return rc;      // one launch for the n rows
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->state + 8, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, m->stream));
    return IFA_OK;
    };
    rc = body();
    if (use_graph) {
        // forward ops swap m->x / m->f per layer: an odd layer count would leave them exchanged between replays
        hipGraph_t gph = nullptr;
        hipError_t e = hipStreamEndCapture(m->stream, &gph);
        if (rc) { if (gph) (void)hipGraphDestroy(gph); return rc; }
        if (e != hipSuccess) return ifa_fail(IFA_ERR_HIP, "hipStreamEndCapture (batched step): %s", hipGetErrorString(e));
        hipGraphExec_t ex = nullptr;
        IFA_HIP_CHECK(hipGraphInstantiate(&ex, gph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(gph);
        m->batch_graphs[n] = ex;
        IFA_HIP_CHECK(hipGraphLaunch(ex, m->stream));
    } else if (rc) return rc;
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if ((rc = wait_err_check("batched decode step"))) { drop_graphs(m); return rc; }
    if (next_tokens) for (int r = 0; r < n; r++) next_tokens[r] = m->host_pinned[8 + r];
    return IFA_OK;
}

} // namespace ifae

extern "C" {

int ifa_model_forward(ifa_model *m, const int *tokens_host, int n_tokens, int prefix_len, void *logits_out_dev,
                      int *next_token_host)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_forward: model not finalized");
    IFA_REQUIRE(tokens_host, "ifa_model_forward: null tokens");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    if (m->opt_exact_order) {      // order-exact: the tokens one by one through the single-row step (the reference's T = 1 branch for every row)
        const size_t V = m->g[T_LM_HEAD].present() ? m->g[T_LM_HEAD].rows : 0;
        for (int t = 0; t < n_tokens; t++) {
            int rc = forward_exact(m, tokens_host[t], prefix_len + t, logits_out_dev ? (char *)logits_out_dev + (size_t)t * V * 2 : nullptr,
                                   t == n_tokens - 1 ? next_token_host : nullptr);
            if (rc) return rc;
        }
        return IFA_OK;
    }
    // Round 5: prompts of 34..48 tokens as TWO passes of the rows GEMM (32 tokens, then 2..16: the weights stream into registers five
    // groups deep) instead of one pass of the op-by-op layer: 40 tokens 7.08 -> 6.35 ms, 48 tokens 7.26 -> 6.67 (profiles/r05_prompt_lengths.log;
    // two passes of 17..32 rows each -- 49..64 tokens -- measured no faster than the tile kernels).  The second pass reads the first
    // one's K / V rows from the cache like any continued prompt; every row goes through the kernels of a prompt of <= 32 tokens.  Never
    // a one-token pass: a single row takes the int8 GEMV (the reference's rule for ONE row), which is not how a prompt's rows are computed.
    if (m->opt_prefill_chunk && !m->topo && n_tokens >= 34 && n_tokens <= 48 && m->cfg.experts == 0 && !prefill_mid_ok(m, n_tokens, false) && batch_fused_ok(m, 32)
        && prefix_len >= 0 && prefix_len + n_tokens <= m->cfg.max_ctx) {
        const int t1 = 32;
        const size_t V = m->g[T_LM_HEAD].rows;
        int rc = forward_ops(m, tokens_host, t1, prefix_len, logits_out_dev, nullptr, true);
        if (rc) return rc;
        return forward_ops(m, tokens_host + t1, n_tokens - t1, prefix_len + t1, logits_out_dev ? (char *)logits_out_dev + (size_t)t1 * V * 2 : nullptr, next_token_host);
    }
    return forward_ops(m, tokens_host, n_tokens, prefix_len, logits_out_dev, next_token_host);
}

int ifa_model_decode_batch(ifa_model *m, int n, const int *tokens_host, const int *positions_host, const int *kv_slots_host,
                           int *next_tokens_host, void *logits_out_dev)
{
    IFA_REQUIRE(m && m->finalized, "ifa_model_decode_batch: model not finalized");
    IFA_REQUIRE(n >= 1 && n <= ifa_model::RING && tokens_host && positions_host && kv_slots_host, "ifa_model_decode_batch: bad arguments");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    if (m->opt_exact_order) {      // order-exact: every query's row through the single-row step on ITS cache set (a batched step of the
        // reference is n independent rows; its T > 1 linear branch has no order-exact form here, so the rows go one by one)
        const int back = m->cur_slot;
        const size_t V = m->g[T_LM_HEAD].present() ? m->g[T_LM_HEAD].rows : 0;
        int rc = IFA_OK;
        for (int i = 0; i < n && rc == IFA_OK; i++) {
            if ((rc = ifa_model_select_kv(m, kv_slots_host[i]))) break;
            int nt = 0;
            rc = forward_exact(m, tokens_host[i], positions_host[i], logits_out_dev ? (char *)logits_out_dev + (size_t)i * V * 2 : nullptr, &nt);
            if (next_tokens_host) next_tokens_host[i] = nt;
        }
        const int rb = ifa_model_select_kv(m, back);
        return rc ? rc : rb;
    }
    // more queries than the fused five-launch step takes (16): balanced chunks of <= 16, each its own step (the queries are
    // independent; 32 queries op-by-op took 6.7 ms against 2 x 3.1 ms for two fused steps)
    const int fused_max = batch_fused_ok(m, 32) ? 32 : 16;
    if (n > fused_max && batch_fused_ok(m, 16)) {
        for (int c0 = 0; c0 < n; c0++) if (kv_slots_host[c0] < 0) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d", kv_slots_host[c0]);
        for (int a = 0; a < n; a++)
            for (int b = 0; b < a; b++)
                if (kv_slots_host[a] == kv_slots_host[b]) return ifa_fail(IFA_ERR_ARG, "decode_batch: KV slot %d used twice", kv_slots_host[a]);
        const int k = (n + fused_max - 1) / fused_max, per = (n + k - 1) / k;
        for (int c0 = 0; c0 < n; c0 += per) {
            const int nc = std::min(per, n - c0);
            int rc = forward_batch(m, nc, tokens_host + c0, positions_host + c0, kv_slots_host + c0, next_tokens_host ? next_tokens_host + c0 : nullptr,
                                   logits_out_dev ? (char *)logits_out_dev + (size_t)c0 * m->g[T_LM_HEAD].rows * 2 : nullptr);
            if (rc) return rc;
        }
        return IFA_OK;
    }
    return forward_batch(m, n, tokens_host, positions_host, kv_slots_host, next_tokens_host, logits_out_dev);
}

} // extern "C"
