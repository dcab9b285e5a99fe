/*
 * ifa_oracle_model.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Whole-model decoder forward with the reference's GPU-path semantics
 * (GpuInferenceWorker::Run, src/transformer/inference_worker.cc:234-340;
 * per-layer op order: SURVEY.md appendix B).  Every intermediate tensor is
 * F16, activations are re-quantised to Q8_B32T2 before each eligible GEMV
 * (GetUseFullQuantGemv, inference_worker.cc:2707-2730), prefill rows go
 * through "dequantise the weight to half, fp32-accumulate" (MatrixMultiplication,
 * inference_worker.cc:2364-2432 + cublasGemmEx F16 in / F32 accumulate,
 * src/tensor/cublas_engine.cu:420-436).
 *
 * Used (a) as the end-to-end parity oracle for the HIP engine and (b), with
 * OpenMP, as bench.py's "port" cpu_baseline.
 */
#include "ifa_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_TENSOR_ID 36

typedef struct {
    int dtype;
    const void *data;
    size_t rows, cols;
} orc_tensor;

typedef struct {
    orc_tensor t[ORC_MAX_TENSOR_ID];
    orc_tensor *experts; /* [experts][3]: w1,w2,w3 */
    void *kcache, *vcache;
} orc_layer;

struct orc_model {
    orc_model_cfg cfg;
    orc_tensor g[ORC_MAX_TENSOR_ID];
    orc_layer *layers;
    orc_f16 *last_hidden;
    orc_f16 *capture;    /* optional (orc_model_set_capture): [layers + 1][dim] */
    float *layer_margin; /* optional (orc_model_set_layer_margins): [layers], the same gap per layer (2.0 where no MoE layer ran) */
    int attn_post_as_residual; /* ModelSpec::is_attn_post_as_residual (default 1) */
    float moe_margin;    /* smallest gap between the LAST selected and the FIRST rejected router probability over the rows and layers of the last forward */
};

static int is_quant(int dtype) { return dtype != ORC_F16 && dtype != ORC_F32; }

static int full_quant_eligible(int dtype)
{
    switch (dtype) {
    case ORC_Q8_B32T2: case ORC_Q6_B64T1: case ORC_Q5_B64T1: case ORC_Q4_B32T1A:
    case ORC_Q4_B32T1B: case ORC_Q4_B64T1: case ORC_Q3H_B64T1: return 1;
    default: return 0;
    }
}

orc_model *orc_model_create(const orc_model_cfg *cfg)
{
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    m->cfg = *cfg;
    m->attn_post_as_residual = 1;
    if (m->cfg.eps <= 0) m->cfg.eps = 1e-5f;
    if (m->cfg.kq_scale <= 0) m->cfg.kq_scale = 1.0f;
    if (m->cfg.partial_rotary <= 0) m->cfg.partial_rotary = 1.0f;
    m->layers = (orc_layer *)calloc((size_t)cfg->layers, sizeof(orc_layer));
    size_t kv_dim = (size_t)cfg->kv_heads * (size_t)cfg->head_dim;
    size_t rowb = orc_row_bytes(cfg->kv_dtype == ORC_Q8_B32T2 ? ORC_Q8_B32T2 : ORC_F16, kv_dim);
    for (int l = 0; l < cfg->layers; l++) {
        m->layers[l].kcache = calloc((size_t)cfg->max_ctx, rowb);
        m->layers[l].vcache = calloc((size_t)cfg->max_ctx, rowb);
        if (cfg->experts > 0)
            m->layers[l].experts = (orc_tensor *)calloc((size_t)cfg->experts * 3, sizeof(orc_tensor));
    }
    m->last_hidden = (orc_f16 *)calloc((size_t)cfg->dim, sizeof(orc_f16));
    return m;
}

void orc_model_destroy(orc_model *m)
{
    if (!m) return;
    for (int l = 0; l < m->cfg.layers; l++) {
        free(m->layers[l].kcache); free(m->layers[l].vcache); free(m->layers[l].experts);
    }
    free(m->layers); free(m->last_hidden); free(m);
}

int orc_model_set_tensor(orc_model *m, int layer, int tensor_id, int expert, int dtype,
                         const void *data, size_t rows, size_t cols)
{
    if (tensor_id < 0 || tensor_id >= ORC_MAX_TENSOR_ID) return -1;
    orc_tensor t; t.dtype = dtype; t.data = data; t.rows = rows; t.cols = cols;
    if (tensor_id < 10) { m->g[tensor_id] = t; return 0; }
    if (layer < 0 || layer >= m->cfg.layers) return -1;
    if (expert >= 0) {
        if (expert >= m->cfg.experts) return -1;
        int slot = tensor_id == ORC_T_W1 ? 0 : tensor_id == ORC_T_W2 ? 1 : tensor_id == ORC_T_W3 ? 2 : -1;
        if (slot < 0) return -1;
        m->layers[layer].experts[expert * 3 + slot] = t;
        return 0;
    }
    m->layers[layer].t[tensor_id] = t;
    return 0;
}

void orc_model_set_attn_post_as_residual(orc_model *m, int on) { m->attn_post_as_residual = on ? 1 : 0; }

void orc_model_reset(orc_model *m)
{
    size_t kv_dim = (size_t)m->cfg.kv_heads * (size_t)m->cfg.head_dim;
    size_t rowb = orc_row_bytes(m->cfg.kv_dtype == ORC_Q8_B32T2 ? ORC_Q8_B32T2 : ORC_F16, kv_dim);
    for (int l = 0; l < m->cfg.layers; l++) {
        memset(m->layers[l].kcache, 0, (size_t)m->cfg.max_ctx * rowb);
        memset(m->layers[l].vcache, 0, (size_t)m->cfg.max_ctx * rowb);
    }
}

const orc_f16 *orc_model_last_hidden(const orc_model *m) { return m->last_hidden; }

/* Test hooks for the layer-wise (teacher-forced) comparison: every forward copies the LAST row's input of each layer to
 * buf[l * dim ..] and the last layer's output to buf[layers * dim ..] (buf: caller-owned, NULL switches it off); the K / V
 * cache of a layer as the forward left it ([max_ctx] rows of the cache's row format). */
void orc_model_set_capture(orc_model *m, orc_f16 *buf) { m->capture = buf; }
/* Mixture-of-experts routing is a DISCONTINUITY: two correct implementations whose router probabilities differ in the last bit pick
 * different experts when the top-k cut falls on a near tie.  The smallest gap p[k-th] - p[(k+1)-th] (F16 probabilities, as routed)
 * over every row and layer of the last forward tells the tests which rows may legitimately part (2.0: no MoE layer ran). */
float orc_model_last_moe_margin(const orc_model *m) { return m->moe_margin; }
void orc_model_set_layer_margins(orc_model *m, float *buf) { m->layer_margin = buf; }
void *orc_model_kv_cache(orc_model *m, int layer, int is_v)
{
    if (layer < 0 || layer >= m->cfg.layers) return NULL;
    return is_v ? m->layers[layer].vcache : m->layers[layer].kcache;
}

/* C[T][rows(W)] = A[T][cols(W)] x W^T (+bias), reference dispatch rules. */
static int matmul(const orc_model *m, const orc_f16 *A, int T, const orc_tensor *W,
                  const orc_tensor *bias, orc_f16 *C)
{
    if (!W->data) return -1;
    const orc_f16 *bptr = (bias && bias->data) ? (const orc_f16 *)bias->data : NULL;
    size_t K = W->cols, N = W->rows;
    int use_gemv = (T == 1) && (K % 32 == 0);
    if (use_gemv && is_quant(W->dtype) && m->cfg.full_quant_gemv && full_quant_eligible(W->dtype)) {
        uint8_t *xq = (uint8_t *)malloc(orc_row_bytes(ORC_Q8_B32T2, K));
        orc_quantize_act_q8(A, 1, K, xq);
        int rc = orc_gemv_ax8(W->dtype, (const uint8_t *)W->data, N, K, xq, C, NULL);
        free(xq);
        if (rc != 0) return rc;
        if (bptr) orc_add(C, bptr, N, 0, C);
        return 0;
    }
    if (T > 1)      /* (every row of W dequantised once for the T rows of A: the per-token loop below, bit for bit, at a fraction of the time) */
        return orc_gemm_f16x(W->dtype, (const uint8_t *)W->data, N, K, A, (size_t)T, bptr, C);
    for (int t = 0; t < T; t++) {
        int rc = orc_gemv_f16x(W->dtype, (const uint8_t *)W->data, N, K, A + (size_t)t * K, bptr,
                               C + (size_t)t * N, NULL);
        if (rc != 0) return rc;
    }
    return 0;
}

/* The same product as P column-range partials merged in rank order (see orc_model_cfg.tp_merge). */
static int matmul_merged(const orc_model *m, const orc_f16 *A, int T, const orc_tensor *W,
                         const orc_tensor *bias, orc_f16 *C)
{
    const int P = m->cfg.tp_merge;
    if (P <= 1) return matmul(m, A, T, W, bias, C);
    if (!W->data) return -1;
    const size_t K = W->cols, N = W->rows, Kp = K / (size_t)P;
    const size_t cap = is_quant(W->dtype) ? (size_t)orc_block_capacity(W->dtype) : 1;
    if (K % (size_t)P || Kp % cap || (is_quant(W->dtype) && Kp % 32)) return -20;
    const size_t rb_full = orc_row_bytes(W->dtype, K), rb = orc_row_bytes(W->dtype, Kp);
    uint8_t *ws = (uint8_t *)malloc(rb * N);
    orc_f16 *as = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * Kp);
    orc_f16 *part = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * N);
    int rc = 0;
    for (int p = 0; p < P && rc == 0; p++) {
        for (size_t r = 0; r < N; r++) memcpy(ws + r * rb, (const uint8_t *)W->data + r * rb_full + (size_t)p * rb, rb);
        for (int t = 0; t < T; t++) memcpy(as + (size_t)t * Kp, A + (size_t)t * K + (size_t)p * Kp, Kp * sizeof(orc_f16));
        orc_tensor sub = *W;
        sub.data = ws; sub.cols = Kp;
        rc = matmul(m, as, T, &sub, NULL, p == 0 ? C : part);
        if (rc == 0 && p > 0) orc_add(C, part, (size_t)T * N, 0, C);       /* MergeTensors: half add, rank order */
    }
    if (rc == 0 && bias && bias->data) orc_add(C, (const orc_f16 *)bias->data, (size_t)T * N, N, C);   /* bias once, after the merge */
    free(ws); free(as); free(part);
    return rc;
}

static int scale_on(float s) { return s > 0.0f && (s < 0.9999f || s > 1.0001f); }

static void norm_rows(const orc_model *m, const orc_f16 *x, int T, const orc_tensor *w,
                      const orc_tensor *b, orc_f16 *y, float base)
{
    const orc_f16 *wp = w && w->data ? (const orc_f16 *)w->data : NULL;
    const orc_f16 *bp = b && b->data ? (const orc_f16 *)b->data : NULL;
    if (m->cfg.norm_kind == 0)
        orc_rmsnorm(x, (size_t)T, (size_t)m->cfg.dim, wp, bp, base, m->cfg.eps, 128, y);
    else
        orc_stdnorm(x, (size_t)T, (size_t)m->cfg.dim, wp, bp, m->cfg.eps, 128, y);
}

static int ffn_dense(const orc_model *m, const orc_f16 *in, int T, const orc_tensor *w1,
                     const orc_tensor *w1b, const orc_tensor *w3, const orc_tensor *w3b,
                     const orc_tensor *w2, const orc_tensor *w2b, orc_f16 *out)
{
    size_t F = w1->rows, D = w2->rows;
    orc_f16 *t1 = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * F);
    orc_f16 *t2 = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * F);
    int rc = matmul(m, in, T, w1, w1b, t1);
    if (rc == 0) {
        orc_act(t1, (size_t)T, F, m->cfg.act_kind, 0, t1);
        if (w3 && w3->data) {
            rc = matmul(m, in, T, w3, w3b, t2);
            if (rc == 0) orc_mul(t1, t2, (size_t)T * F, t1);
        }
    }
    if (rc == 0) rc = matmul_merged(m, t1, T, w2, w2b, out);
    (void)D;
    free(t1); free(t2);
    return rc;
}

int orc_model_forward(orc_model *m, const int *tokens, int T, int prefix_len,
                      orc_f16 *logits_out, int nthreads)
{
    const orc_model_cfg *c = &m->cfg;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    if (prefix_len + T > c->max_ctx || T <= 0) return -1;
    m->moe_margin = 2.0f;
    const size_t D = (size_t)c->dim;
    const size_t QD = (size_t)c->heads * (size_t)c->head_dim;
    const size_t KVD = (size_t)c->kv_heads * (size_t)c->head_dim;
    const int kvt = c->kv_dtype == ORC_Q8_B32T2 ? ORC_Q8_B32T2 : ORC_F16;
    const size_t kv_rowb = orc_row_bytes(kvt, KVD);
    orc_f16 *x = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
    orc_f16 *xn = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
    orc_f16 *hn = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
    orc_f16 *pn = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
    orc_f16 *q = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * QD);
    orc_f16 *k = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * KVD);
    orc_f16 *v = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * KVD);
    orc_f16 *att = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * QD);
    orc_f16 *a = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
    orc_f16 *f = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
    int rc = 0;

    /* embedding rows (InferenceEngine::GetEmbdTensor, inference_engine.cc:1298-1353) */
    {
        const orc_tensor *e = &m->g[ORC_T_EMBD];
        if (!e->data || e->dtype != ORC_F16) { rc = -2; goto done; }
        for (int t = 0; t < T; t++) {
            if (tokens[t] < 0 || (size_t)tokens[t] >= e->rows) { rc = -3; goto done; }
            memcpy(x + (size_t)t * D, (const orc_f16 *)e->data + (size_t)tokens[t] * D, D * 2);
        }
        if (c->embd_scale != 0.0f)       /* ProcessPreLayer's LinearNorm (inference_worker.cc:447-451): Tensor_Scale_Kernel, F16 out */
            orc_scale(x, c->embd_scale < 0.0f ? sqrtf((float)D) : c->embd_scale, (size_t)T * D, x);
    }
    const int rope_dims = (int)((float)c->head_dim * c->partial_rotary + 0.5f);
    const int rope_cols = rope_dims;
    for (int l = 0; l < c->layers && rc == 0; l++) {
        orc_layer *L = &m->layers[l];
        if (m->capture) memcpy(m->capture + (size_t)l * D, x + (size_t)(T - 1) * D, D * 2);
        if (m->layer_margin) m->layer_margin[l] = 2.0f;
        /* attention pre-norm (inference_worker.cc:1038) */
        const orc_f16 *attn_in = x;
        if (L->t[ORC_T_ATTN_NORM].data) {
            norm_rows(m, x, T, &L->t[ORC_T_ATTN_NORM], &L->t[ORC_T_ATTN_NORM_B], xn, c->attn_norm_base);
            attn_in = xn;
        }
        rc = matmul(m, attn_in, T, &L->t[ORC_T_WQ], &L->t[ORC_T_WQ_B], q); if (rc) break;
        rc = matmul(m, attn_in, T, &L->t[ORC_T_WK], &L->t[ORC_T_WK_B], k); if (rc) break;
        rc = matmul(m, attn_in, T, &L->t[ORC_T_WV], &L->t[ORC_T_WV_B], v); if (rc) break;
        if (c->rope_order != 0) {
            orc_rope(q, c->head_dim, c->heads, T, prefix_len, c->rope_theta, c->rope_order, rope_dims, rope_cols);
            orc_rope(k, c->head_dim, c->kv_heads, T, prefix_len, c->rope_theta, c->rope_order, rope_dims, rope_cols);
        }
        /* KV store at rows [prefix_len, prefix_len+T) (kv_cache.cc:159-249) */
        if (kvt == ORC_Q8_B32T2) {
            orc_quantize_act_q8(k, (size_t)T, KVD, (uint8_t *)L->kcache + (size_t)prefix_len * kv_rowb);
            orc_quantize_act_q8(v, (size_t)T, KVD, (uint8_t *)L->vcache + (size_t)prefix_len * kv_rowb);
        } else {
            memcpy((uint8_t *)L->kcache + (size_t)prefix_len * kv_rowb, k, (size_t)T * kv_rowb);
            memcpy((uint8_t *)L->vcache + (size_t)prefix_len * kv_rowb, v, (size_t)T * kv_rowb);
        }
        orc_attention(q, L->kcache, L->vcache, kvt, prefix_len + T, T, prefix_len, c->heads,
                      c->kv_heads, c->head_dim, c->use_alibi ? 1.0f : c->kq_scale, c->use_alibi,
                      0, c->heads, att);
        rc = matmul_merged(m, att, T, &L->t[ORC_T_WO], &L->t[ORC_T_WO_B], a); if (rc) break;
        if (scale_on(c->attn_out_scale)) orc_scale(a, c->attn_out_scale, (size_t)T * D, a);   /* :842-843 */
        /* residual (inference_worker.cc:847-851) */
        if (!c->parallel_attn && !c->share_input) orc_add(x, a, (size_t)T * D, 0, a);
        /* self_attn.post_norm (:854-866): att_out = Norm(a); the residual follows it when is_attn_post_as_residual */
        const orc_f16 *residual = a, *att_out = a;
        if (L->t[ORC_T_ATTN_POST_NORM].data) {
            norm_rows(m, a, T, &L->t[ORC_T_ATTN_POST_NORM], &L->t[ORC_T_ATTN_POST_NORM_B], pn, 0.0f);
            att_out = pn;
            if (m->attn_post_as_residual) residual = pn;
        }
        const orc_f16 *ff_in = c->parallel_attn ? attn_in : (c->share_input ? x : att_out);
        const orc_f16 *ff_n = ff_in;
        if (L->t[ORC_T_FFN_NORM].data) {
            norm_rows(m, ff_in, T, &L->t[ORC_T_FFN_NORM], &L->t[ORC_T_FFN_NORM_B], hn, c->ffn_norm_base);
            ff_n = hn;
        }
        if (c->experts > 0 && L->t[ORC_T_MOE_GATE].data) {
            /* ProcessGpuLayer_Moe, inference_worker.cc:1924-2146 */
            int E = c->experts;
            orc_f16 *gate = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * (size_t)E);
            rc = matmul(m, ff_n, T, &L->t[ORC_T_MOE_GATE], NULL, gate);
            if (rc == 0) {
                orc_softmax(gate, E, T, 1, -1, 1.0f);
                memset(f, 0, sizeof(orc_f16) * (size_t)T * D);
                /* HostTensorOpr::BuildRowsForMoE (host_tensor_opr.cc:190-244): per expert the list of rows routed to
                 * it (token order) and their weights; then expert by expert (serial loop :2053-2121): gather the
                 * rows, run the FFN on them as ONE matrix (GEMV path for a single row, dequant+GEMM path otherwise:
                 * MatrixMultiplication, :2374-2415), scatter-add w * out (AddByRowIdx_Kernel: B = hfma(A, w, B)). */
                int *rows = (int *)malloc(sizeof(int) * (size_t)T * (size_t)E);
                float *wts = (float *)malloc(sizeof(float) * (size_t)T * (size_t)E);
                int *cnt = (int *)calloc((size_t)E, sizeof(int));
                for (int t = 0; t < T; t++) {
                    float probs[64]; int idx[8]; float w[8];
                    for (int e = 0; e < E; e++) probs[e] = orc_h2f(gate[(size_t)t * (size_t)E + (size_t)e]);
                    if (E > c->moe_top_k) {      /* the margin of the top-k cut (selection sort of a copy: E <= 64) */
                        float sp[64];
                        memcpy(sp, probs, sizeof(float) * (size_t)E);
                        for (int a = 0; a <= c->moe_top_k && a < E; a++)
                            for (int b2 = a + 1; b2 < E; b2++) if (sp[b2] > sp[a]) { float tmpv = sp[a]; sp[a] = sp[b2]; sp[b2] = tmpv; }
                        const float gap = sp[c->moe_top_k - 1] - sp[c->moe_top_k];
                        if (gap < m->moe_margin) m->moe_margin = gap;
                        if (m->layer_margin && gap < m->layer_margin[l]) m->layer_margin[l] = gap;
                    }
                    int n = orc_moe_topk(probs, E, c->moe_top_k, c->moe_norm_topk, idx, w);
                    for (int j = 0; j < n; j++) {
                        int e = idx[j];
                        rows[(size_t)e * T + cnt[e]] = t; wts[(size_t)e * T + cnt[e]] = w[j]; cnt[e]++;
                    }
                }
                orc_f16 *ein = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
                orc_f16 *eo = (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * D);
                for (int e = 0; e < E && rc == 0; e++) {
                    int n = cnt[e];
                    if (n == 0) continue;
                    for (int r = 0; r < n; r++)
                        memcpy(ein + (size_t)r * D, ff_n + (size_t)rows[(size_t)e * T + r] * D, D * sizeof(orc_f16));
                    const orc_tensor *ew = &L->experts[e * 3];
                    rc = ffn_dense(m, ein, n, &ew[0], NULL, &ew[2], NULL, &ew[1], NULL, eo);
                    for (int r = 0; r < n && rc == 0; r++) {
                        size_t t = (size_t)rows[(size_t)e * T + r];
                        orc_f16 wh = orc_f2h(wts[(size_t)e * T + r]);
                        for (size_t d = 0; d < D; d++) {
                            double p = (double)orc_h2f(eo[(size_t)r * D + d]) * (double)orc_h2f(wh) + (double)orc_h2f(f[t * D + d]);
                            f[t * D + d] = orc_f2h((float)p);
                        }
                    }
                }
                free(ein); free(eo); free(rows); free(wts); free(cnt);
            }
            free(gate);
        } else {
            rc = ffn_dense(m, ff_n, T, &L->t[ORC_T_W1], &L->t[ORC_T_W1_B], &L->t[ORC_T_W3],
                           &L->t[ORC_T_W3_B], &L->t[ORC_T_W2], &L->t[ORC_T_W2_B], f);
        }
        if (rc) break;
        if (scale_on(c->ffn_out_scale)) orc_scale(f, c->ffn_out_scale, (size_t)T * D, f);      /* :928-929 */
        /* layer_out = ff_out + residual (+ layer_input) (inference_worker.cc:936-947) */
        orc_add(f, residual, (size_t)T * D, 0, f);
        if (c->parallel_attn || c->share_input) orc_add(f, x, (size_t)T * D, 0, f);
        if (L->t[ORC_T_FFN_POST_NORM].data) {      /* feed_forward.post_norm (:954-965) */
            norm_rows(m, f, T, &L->t[ORC_T_FFN_POST_NORM], &L->t[ORC_T_FFN_POST_NORM_B], pn, 0.0f);
            memcpy(f, pn, sizeof(orc_f16) * (size_t)T * D);
        }
        memcpy(x, f, sizeof(orc_f16) * (size_t)T * D);
    }
    if (rc == 0 && m->capture) memcpy(m->capture + (size_t)c->layers * D, x + (size_t)(T - 1) * D, D * 2);
    if (rc == 0) {
        /* ProcessPostLayer, inference_worker.cc:552-624 */
        if (scale_on(m->cfg.out_scale)) orc_scale(x, m->cfg.out_scale, (size_t)T * D, x);      /* :568-570, in place */
        const orc_f16 *hfin = x;
        if (m->g[ORC_T_OUT_NORM].data) {
            norm_rows(m, x, T, &m->g[ORC_T_OUT_NORM], &m->g[ORC_T_OUT_NORM_B], xn, m->cfg.out_norm_base);
            hfin = xn;
        }
        memcpy(m->last_hidden, hfin + (size_t)(T - 1) * D, D * 2);
        const orc_tensor *lm = &m->g[ORC_T_LM_HEAD];
        size_t V = lm->rows;
        orc_f16 *lg = logits_out ? logits_out : (orc_f16 *)malloc(sizeof(orc_f16) * (size_t)T * V);
        int t0 = logits_out ? 0 : T - 1;
        if (logits_out) rc = matmul(m, hfin, T, lm, NULL, lg);
        else rc = matmul(m, hfin + (size_t)t0 * D, 1, lm, NULL, lg + (size_t)t0 * V);
        if (rc == 0) {
            /* greedy = top-1 (sampling_strategy.cc:372-386); first max wins */
            const orc_f16 *row = lg + (size_t)(T - 1) * V;
            /* greedy = top-1 of GetSortedTopK (sampling_strategy.cc:281-297, 372-386): first maximum over the ids
             * other than the unk id */
            int best = -1; float bv = 0.0f;
            for (size_t i = 0; i < V; i++) {
                if ((int)i == m->cfg.unk_id) continue;
                float vv = orc_h2f(row[i]);
                if (best < 0 || vv > bv) { bv = vv; best = (int)i; }
            }
            rc = best < 0 ? 0 : best;
        } else rc = -10;
        if (!logits_out) free(lg);
    } else {
        rc = -11;
    }
done:
    free(x); free(xn); free(hn); free(pn); free(q); free(k); free(v); free(att); free(a); free(f);
    return rc;
}
