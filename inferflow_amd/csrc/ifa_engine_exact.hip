// ifa_engine_exact.hip -- option `exact_order`: single-token steps of the worker through the kernels of csrc/ifa_exact.hip, i.e. in the
// summation order of the reference's CUDA kernels as the oracle restates them.  One token at a time, launch by launch, the layer of
// GpuInferenceWorker::ProcessGpuLayer (inference_worker.cc:762-981) with every op boundary an F16 tensor: all of them, the int8 codes of
// the re-quantised activations and of a Q8 KV cache, and the logits are then bit-identical to oracle.Model's
// (tests/test_gpu_fullsize_oracle.py: 32 layers, configs[1] and configs[2]).  A parity instrument, ~50x slower than the timed path.
#include <cmath>
#include "ifa_engine_state.h"
#include "ifa_exact.h"

namespace ifae {

bool exact_supported(const ifa_model *m, std::string *why)
{
    const ifa_model_config &c = m->cfg;
    auto no = [&](const char *w) { if (why) *why = w; return false; };
    if (m->topo || c.tp_size > 1) return no("partitioned model");
    if (c.norm_kind != 0) return no("Std norm (only the RMS kernel has an order-exact form)");
    if (c.use_alibi) return no("ALiBi");
    if (c.experts > 64 || (c.experts > 0 && (c.moe_top_k < 1 || c.moe_top_k > 8))) return no("more than 64 experts / top-k outside 1..8");
    if (c.parallel_attn || c.share_input) return no("parallel-attention / shared-input wiring");
    if (c.act_kind == 1) return no("GELU (tanhf of the host libm has no device restatement)");
    if (c.rope_order != 0 && (c.partial_rotary < 0.9999f || c.partial_rotary > 1.0001f)) return no("partial rotary embedding");
    if (has_post_norms(m)) return no("post norms");
    if (!m->g[T_EMBD].present() || m->g[T_EMBD].dtype != F16 || !m->g[T_LM_HEAD].present()) return no("embeddings / lm_head");
    return true;
}

// RoPE angles of every position, from the HOST's libm like the CPU side of the comparison (orc_rope): ang = pos * theta_scale^col
// with theta_scale = powf(theta, -2 / head_dim), (cos, sin) per (position, pair)
static int exact_rope_table(ifa_model *m)
{
    if (m->exact_rope_tab) return IFA_OK;
    const ifa_model_config &c = m->cfg;
    const int half_hd = c.head_dim / 2;
    std::vector<float> tab((size_t)c.max_ctx * half_hd * 2);
    const float theta_scale = powf(c.rope_theta, -2.0f / (float)c.head_dim);
    for (int pos = 0; pos < c.max_ctx; pos++)
        for (int col = 0; col < half_hd; col++) {
            float ang = (float)pos;
            if (col > 0) ang *= powf(theta_scale, (float)col);
            tab[((size_t)pos * half_hd + col) * 2] = cosf(ang);
            tab[((size_t)pos * half_hd + col) * 2 + 1] = sinf(ang);
        }
    IFA_HIP_CHECK(hipMalloc((void **)&m->exact_rope_tab, tab.size() * sizeof(float)));
    IFA_HIP_CHECK(hipMemcpy(m->exact_rope_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    return IFA_OK;
}

// MatrixMultiplication at one row (inference_worker.cc:2337-2432): the int8 path for the seven eligible formats, else weights
// dequantised to half x F16 activations
static int exact_matmul(ifa_model *m, const half_t *x, const Tensor &W, const Tensor &bias, half_t *y)
{
    if (!W.present()) return ifa_fail(IFA_ERR_STATE, "missing weight tensor");
    const half_t *b = bias.present() ? (const half_t *)bias.data : nullptr;
    if (W.cols % 32 == 0 && W.dtype != F16 && ax8_eligible(W.dtype)) {
        int rc = ifa_quantize_act_q8(x, 1, W.cols, m->xq, (ifa_stream)m->stream);      // (codes and scales bit-exact: tests/test_gpu_ops.py)
        if (rc) return rc;
        return exact_gemv_ax8(W.dtype, W.data, W.rows, W.cols, m->xq, b, y, m->stream);
    }
    return exact_gemv_f16x(W.dtype, W.data, W.rows, W.cols, x, b, y, m->stream);
}

int forward_exact(ifa_model *m, int token, int pos, void *logits_out, int *next_token)
{
    const ifa_model_config &c = m->cfg;
    std::string why;
    if (!exact_supported(m, &why)) return ifa_fail(IFA_ERR_STATE, "exact_order: this model is outside the order-exact step (%s)", why.c_str());
    if (pos < 0 || pos >= c.max_ctx) return ifa_fail(IFA_ERR_ARG, "exact_order: position %d outside [0, %d)", pos, c.max_ctx);
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    if ((rc = exact_init(m->stream))) return rc;
    if (c.rope_order != 0 && (rc = exact_rope_table(m))) return rc;
    ifa_stream s = (ifa_stream)m->stream;
    const size_t D = c.dim, KVD = (size_t)c.kv_heads * c.head_dim;
    const Tensor none;
    m->host_pinned[0] = token;
    IFA_HIP_CHECK(hipMemcpyAsync(m->tokens_dev, m->host_pinned, sizeof(int), hipMemcpyHostToDevice, m->stream));
    if ((rc = gather_rows(m, (const half_t *)m->g[T_EMBD].data, m->tokens_dev, 1, (int)D, (int)m->g[T_EMBD].rows, m->x, c.embd_scale))) return rc;
    half_t *x = m->x;
    const float alpha = 1.0f / sqrtf((float)c.head_dim) / c.kq_scale;      // orc_attention's alpha, in its operation order
    for (int l = 0; l < c.layers; l++) {
        Layer &L = m->layers[l];
        const half_t *attn_in = x;
        if (L.t[T_ATTN_NORM].present()) {
            if ((rc = exact_rmsnorm(x, 1, (int)D, (const half_t *)L.t[T_ATTN_NORM].data, (const half_t *)L.t[T_ATTN_NORM_B].data, c.attn_norm_base, c.eps, m->xn, m->stream))) return rc;
            attn_in = m->xn;
        }
        if ((rc = exact_matmul(m, attn_in, L.t[T_WQ], L.t[T_WQ_B], m->q))) return rc;
        if ((rc = exact_matmul(m, attn_in, L.t[T_WK], L.t[T_WK_B], m->k))) return rc;
        if ((rc = exact_matmul(m, attn_in, L.t[T_WV], L.t[T_WV_B], m->v))) return rc;
        if (c.rope_order != 0) {
            const float *row = m->exact_rope_tab + (size_t)pos * (c.head_dim / 2) * 2;
            if ((rc = exact_rope(m->q, c.head_dim, c.heads, row, c.rope_order, m->stream))) return rc;
            if ((rc = exact_rope(m->k, c.head_dim, c.kv_heads, row, c.rope_order, m->stream))) return rc;
        }
        uint8_t *kdst = (uint8_t *)L.kcache + (size_t)pos * m->kv_row_bytes, *vdst = (uint8_t *)L.vcache + (size_t)pos * m->kv_row_bytes;
        if (c.kv_dtype == Q8_B32T2) {
            if ((rc = ifa_quantize_act_q8(m->k, 1, KVD, kdst, s))) return rc;
            if ((rc = ifa_quantize_act_q8(m->v, 1, KVD, vdst, s))) return rc;
        } else {
            IFA_HIP_CHECK(hipMemcpyAsync(kdst, m->k, m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
            IFA_HIP_CHECK(hipMemcpyAsync(vdst, m->v, m->kv_row_bytes, hipMemcpyDeviceToDevice, m->stream));
        }
        if ((rc = exact_attention(m->q, L.kcache, L.vcache, c.kv_dtype, m->kv_row_bytes, pos + 1, c.heads, c.kv_heads, c.head_dim, alpha, c.kq_scale,
                                  m->att, m->stream))) return rc;
        if ((rc = exact_matmul(m, m->att, L.t[T_WO], L.t[T_WO_B], m->a))) return rc;
        if (scale_on(c.attn_out_scale) && (rc = ifa_scale(m->a, c.attn_out_scale, D, m->a, s))) return rc;
        if ((rc = ifa_add(x, m->a, D, 0, m->a, s))) return rc;                     // residual (:847-851), a half add
        const half_t *ff_n = m->a;
        if (L.t[T_FFN_NORM].present()) {
            if ((rc = exact_rmsnorm(m->a, 1, (int)D, (const half_t *)L.t[T_FFN_NORM].data, (const half_t *)L.t[T_FFN_NORM_B].data, c.ffn_norm_base, c.eps, m->hn, m->stream))) return rc;
            ff_n = m->hn;
        }
        if (c.experts > 0 && L.t[T_MOE_GATE].present()) {
            // ProcessGpuLayer_Moe (inference_worker.cc:1924-2146) at one row: router GEMV (F16 gate) -> softmax -> top-k with
            // BuildRowsForMoE's rules (ascending expert order, F16 weights: ifa_moe_route_topk, pinned to the reference's own rows) ->
            // the selected experts' FFNs in that order, each added as  f = half(float(out x w + f))
            const int E = c.experts, K = c.moe_top_k;
            if ((rc = exact_matmul(m, ff_n, L.t[T_MOE_GATE], none, m->moe_gate))) return rc;
            if ((rc = exact_softmax_row(m->moe_gate, E, 1.0f, m->stream))) return rc;
            if ((rc = ifa_moe_route_topk(m->moe_gate, 1, E, K, c.moe_norm_topk, m->moe_sel, m->moe_selw, s))) return rc;
            IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 16, m->moe_sel, sizeof(int) * (size_t)K, hipMemcpyDeviceToHost, m->stream));
            IFA_HIP_CHECK(hipMemsetAsync(m->f, 0, D * 2, m->stream));
            IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
            int sel[8];
            for (int j = 0; j < K; j++) sel[j] = m->host_pinned[16 + j];
            for (int j = 0; j < K; j++) {
                const int ex = sel[j];
                if (ex < 0) continue;          // (dropped: probability below 1e-5, or fewer experts than top-k)
                if (ex >= E) return ifa_fail(IFA_ERR_STATE, "exact_order: router selected expert %d of %d", ex, E);
                const Tensor *ew = &L.experts[(size_t)ex * 3];      // w1, w2, w3
                const size_t Fe = ew[0].rows;
                if ((rc = exact_matmul(m, ff_n, ew[0], none, m->t1))) return rc;
                if (ew[2].present()) {
                    if ((rc = exact_matmul(m, ff_n, ew[2], none, m->t2))) return rc;
                    if ((rc = exact_act_mul(c.act_kind, m->t1, m->t2, Fe, m->t1, m->stream))) return rc;
                } else if ((rc = exact_act_mul(c.act_kind, m->t1, nullptr, Fe, m->t1, m->stream))) return rc;
                if ((rc = exact_matmul(m, m->t1, ew[1], none, m->moe_out))) return rc;
                if ((rc = exact_moe_combine(m->f, m->moe_out, m->moe_selw + j, D, m->stream))) return rc;
            }
        } else {
        const size_t F = L.t[T_W1].rows;
        if ((rc = exact_matmul(m, ff_n, L.t[T_W1], L.t[T_W1_B], m->t1))) return rc;
        if (L.t[T_W3].present()) {
            if ((rc = exact_matmul(m, ff_n, L.t[T_W3], L.t[T_W3_B], m->t2))) return rc;
            if ((rc = exact_act_mul(c.act_kind, m->t1, m->t2, F, m->t1, m->stream))) return rc;
        } else if ((rc = exact_act_mul(c.act_kind, m->t1, nullptr, F, m->t1, m->stream))) return rc;
        if ((rc = exact_matmul(m, m->t1, L.t[T_W2], L.t[T_W2_B], m->f))) return rc;
        }
        if (scale_on(c.ffn_out_scale) && (rc = ifa_scale(m->f, c.ffn_out_scale, D, m->f, s))) return rc;
        if ((rc = ifa_add(m->f, m->a, D, 0, m->f, s))) return rc;                  // layer_out = ff_out + residual (:936-947)
        std::swap(m->x, m->f);
        x = m->x;
    }
    if (scale_on(c.out_scale) && (rc = ifa_scale(x, c.out_scale, D, x, s))) return rc;
    const half_t *hfin = x;
    if (m->g[T_OUT_NORM].present()) {
        if ((rc = exact_rmsnorm(x, 1, (int)D, (const half_t *)m->g[T_OUT_NORM].data, (const half_t *)m->g[T_OUT_NORM_B].data, c.out_norm_base, c.eps, m->xn, m->stream))) return rc;
        hfin = m->xn;
    } else {
        IFA_HIP_CHECK(hipMemcpyAsync(m->xn, x, D * 2, hipMemcpyDeviceToDevice, m->stream));
    }
    const Tensor &lm = m->g[T_LM_HEAD];
    if ((rc = exact_matmul(m, hfin, lm, none, m->logits))) return rc;
    if (logits_out) IFA_HIP_CHECK(hipMemcpyAsync(logits_out, m->logits, lm.rows * 2, hipMemcpyDeviceToDevice, m->stream));
    if ((rc = ifa_argmax_masked(m->logits, lm.rows, m->state + 3, m->state, s))) return rc;
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned, m->state, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if (next_token) *next_token = m->host_pinned[0];
    return IFA_OK;
}

} // namespace ifae
