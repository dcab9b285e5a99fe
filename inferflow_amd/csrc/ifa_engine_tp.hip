// ifa_engine_tp.hip -- tensor / layer partitions of the worker (inference_engine.cc:1222-1296, inference_worker.cc:2148-2335):
// per-seam entry points for a caller that all-reduces between them, and the whole multi-GPU step driven from C.
#include "ifa_engine_state.h"

namespace ifae {

// w2 . (act(w1 . x) [* (w3 . x)])   (ProcessGpuLayer_FeedForward, inference_worker.cc:1726-1922)
// ---- distributed greedy argmax over a vocabulary-sharded lm_head (one workgroup per row)
// (value, global id) of the best allowed logit of this rank's shard; first maximum wins
__global__ void __launch_bounds__(1024) k_tp_local_best(const half_t *__restrict__ v_all, size_t row_stride, int n, int vocab_offset,
                                                        const int *__restrict__ excl, float *__restrict__ best_all)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    const half_t *v = v_all + (size_t)blockIdx.x * row_stride;
    float *best_out = best_all + 2 * (size_t)blockIdx.x;
    const int ne = excl ? min(max(excl[0], 0), 3) : 0;
    const int e0 = ne > 0 ? excl[1] : -1, e1 = ne > 1 ? excl[2] : -1, e2 = ne > 2 ? excl[3] : -1;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    argmax_scan(v, (size_t)n, e0, e1, e2, (int)threadIdx.x, (int)blockDim.x, best, besti, vocab_offset);
#pragma unroll
    for (int mk = 32; mk > 0; mk >>= 1) {
        const float ob = __shfl_xor(best, mk); const int oi = __shfl_xor(besti, mk);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { bv[wave] = best; bi[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
        best_out[0] = best;
        reinterpret_cast<int *>(best_out)[1] = besti;
    }
}

// the group's choice per row: highest value, lowest id among equals.  gathered: [rank][row][2]
__global__ void k_tp_pick(const float *__restrict__ gathered, int nranks, int n_rows, int *__restrict__ token)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float best = -INFINITY; int besti = 0x7FFFFFFF;
    for (int k = 0; k < nranks; k++) {
        const float f = gathered[2 * ((size_t)k * n_rows + r)];
        const int gid = reinterpret_cast<const int *>(gathered)[2 * ((size_t)k * n_rows + r) + 1];
        if (f > best || (f == best && gid < besti)) { best = f; besti = gid; }
    }
    token[r] = besti == 0x7FFFFFFF ? 0 : besti;
}

// scratch of the distributed argmax for n_rows rows
int tp_argmax_scratch(ifa_model *m, size_t n_rows)
{
    if (n_rows <= m->tp_rows_cap) return IFA_OK;
    if (m->tp_best) IFA_HIP_CHECK(hipFree(m->tp_best));
    if (m->tp_gather) IFA_HIP_CHECK(hipFree(m->tp_gather));
    if (m->tp_tok) IFA_HIP_CHECK(hipFree(m->tp_tok));
    m->tp_best = nullptr; m->tp_gather = nullptr; m->tp_tok = nullptr;
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_best, 8 * n_rows));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_gather, 8 * 64 * n_rows));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_tok, 4 * n_rows));
    m->tp_rows_cap = n_rows;
    drop_graphs(m);
    return IFA_OK;
}

// tokens[r] (device, m->tp_tok) = the group's greedy choice for row r of this rank's logits shard [n_rows][row_stride]
int tp_pick_rows(ifa_model *m, const ifa_tp_topology &t, const half_t *shard, size_t row_stride, int shard_rows, int n_rows)
{
    int rc = tp_argmax_scratch(m, (size_t)n_rows);
    if (rc) return rc;
    ifa_stream s = (ifa_stream)m->stream;
    const int tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
    const bool merge = t.tp && (tp_size > 1 || t.force_collectives);
    k_tp_local_best<<<dim3((unsigned)n_rows), 1024, 0, m->stream>>>(shard, row_stride, shard_rows, t.vocab_offset, m->state + 3, m->tp_best);
    IFA_LAUNCH_CHECK();
    const float *gathered = m->tp_best;
    int n_g = 1;
    if (merge) {
        if ((rc = ifa_allgather(t.tp, m->tp_best, m->tp_gather, 8 * (size_t)n_rows, s))) return rc;
        gathered = m->tp_gather; n_g = tp_size;
    }
    k_tp_pick<<<dim3((unsigned)((n_rows + 63) / 64)), 64, 0, m->stream>>>(gathered, n_g, n_rows, m->tp_tok);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

// ---- tensor-parallel T > 1 / batched steps: the same op sequence, with the reference's merge
// (DistributeAndMergeTensors, inference_worker.cc:2148-2195) after the two column-sliced products of a layer
bool tp_merging(const ifa_model *m)
{
    const ifa_tp_topology *t = m->topo;
    return t && t->tp && (ifa_comm_size(t->tp) > 1 || t->force_collectives);
}

// buf[T][dim] holds this rank's partial product (computed WITHOUT bias): sum over the group, then the bias once
int tp_merge_rows(ifa_model *m, half_t *buf, int T, const Tensor &bias)
{
    if (!tp_merging(m)) return IFA_OK;
    const size_t D = (size_t)m->cfg.dim;
    int rc = ifa_allreduce_sum_f16(m->topo->tp, buf, buf, (size_t)T * D, (ifa_stream)m->stream);
    if (rc) return rc;
    if (bias.present()) return ifa_add(buf, bias.data, (size_t)T * D, D, buf, (ifa_stream)m->stream);
    return IFA_OK;
}

// ---- tensor-parallel decode, one segment per call (the caller all-reduces between them):
// the reference's DistributeAndMergeTensors sits exactly at these two seams
// (src/transformer/inference_worker.cc:1378-1391, :1882-1895).
__global__ void k_tp_set_state(int *state, int token, int pos)
{
    if (token >= 0) state[0] = token;
    if (pos >= 0) state[1] = pos;
}

// the step's token (chosen across the group) becomes the next input; the position advances on the device
// so that a captured step can be replayed (hipGraph) without the host
__global__ void k_tp_set_token(int *state, const int *token, int ring)
{
    const int t = *token;
    const int step = state[2];
    state[8 + (step % ring)] = t;       // the launch batch's token ring, like k_dec_argmax_advance
    state[0] = t;
    state[1] = state[1] + 1;
    state[2] = step + 1;
}

int tp_ready(ifa_model *m)
{
    IFA_REQUIRE(m && m->finalized, "tensor-parallel step: model not finalized");
    IFA_HIP_CHECK(hipSetDevice(m->cfg.device));
    std::string why;
    if (!fused_supported(m, &why)) return ifa_fail(IFA_ERR_STATE, "fused path unavailable: %s", why.c_str());
    const ifa_model_config &c = m->cfg;      // the fused epilogues apply out_scale after the LOCAL last layer: single-worker models only
    if (scale_on(c.attn_out_scale) || scale_on(c.ffn_out_scale) || scale_on(c.out_scale))
        return ifa_fail(IFA_ERR_STATE, "fused path unavailable: output scales on a partitioned model use the op-by-op path");
    return ensure_scratch(m, 1);
}

} // namespace ifae

extern "C" {

int ifa_model_tp_begin(ifa_model *m, int token, int pos)
{
    int rc = tp_ready(m);
    if (rc) return rc;
    m->pend.on = false;
    IFA_REQUIRE(pos >= -1 && pos < m->cfg.max_ctx, "ifa_model_tp_begin: position %d outside max_ctx %d", pos, m->cfg.max_ctx);
    IFA_REQUIRE(m->g[T_EMBD].present(), "ifa_model_tp_begin: this worker holds no embeddings (use ifa_model_tp_begin_hidden)");
    const ifa_model_config &c = m->cfg;
    k_tp_set_state<<<1, 1, 0, m->stream>>>(m->state, token, pos);     // token < 0: keep the id already on the device
    k_dec_gather<<<dim3(2), dim3(256), 0, m->stream>>>((const half_t *)m->g[T_EMBD].data, m->state, c.dim, (int)m->g[T_EMBD].rows,
                                                       m->x, c.rope_order ? m->rope_tab : nullptr, c.head_dim, c.rope_theta,
                                                       (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_model_tp_begin_hidden(ifa_model *m, const void *x_f16, int pos)
{
    int rc = tp_ready(m);
    if (rc) return rc;
    IFA_REQUIRE(x_f16, "ifa_model_tp_begin_hidden: null input");
    m->pend.on = false;
    IFA_REQUIRE(pos >= -1 && pos < m->cfg.max_ctx, "ifa_model_tp_begin_hidden: position %d outside max_ctx %d", pos, m->cfg.max_ctx);
    const ifa_model_config &c = m->cfg;
    IFA_HIP_CHECK(hipMemcpyAsync(m->x, x_f16, (size_t)c.dim * 2, hipMemcpyDeviceToDevice, m->stream));
    k_tp_set_state<<<1, 1, 0, m->stream>>>(m->state, -1, pos);
    k_dec_gather<<<dim3(1), dim3(256), 0, m->stream>>>(nullptr, m->state, c.dim, 1, m->x, c.rope_order ? m->rope_tab : nullptr,
                                                       c.head_dim, c.rope_theta, (int)(c.head_dim * c.partial_rotary + 0.5f), c.embd_scale);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int ifa_model_tp_hidden(ifa_model *m, void *x_out_f16)
{
    IFA_REQUIRE(m && m->finalized && x_out_f16, "ifa_model_tp_hidden: bad arguments");
    { int rcf = tp_flush_pending(m); if (rcf) return rcf; }
    IFA_HIP_CHECK(hipMemcpyAsync(x_out_f16, m->x, (size_t)m->cfg.dim * 2, hipMemcpyDeviceToDevice, m->stream));
    return IFA_OK;
}

int ifa_model_tp_attn(ifa_model *m, int layer, void *partial_out_f16)
{
    IFA_REQUIRE(m && partial_out_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_attn: bad arguments");
    int rc;
    if ((rc = launch_qkv(m, layer, m->x))) return rc;
    if ((rc = launch_attn(m, layer))) return rc;
    return launch_wo(m, layer, m->x, (half_t *)partial_out_f16);
}

} // extern "C"

namespace ifae {

// form a pending seam sum with the op-level add kernels (when no fused consumer follows)
int tp_flush_pending(ifa_model *m)
{
    if (!m->pend.on) return IFA_OK;
    m->pend.on = false;
    const size_t D = (size_t)m->cfg.dim;
    const half_t *src = m->pend.add;
    int rc;
    if (m->pend.bias) {
        if ((rc = ifa_add(m->pend.add, m->pend.bias, D, 0, m->f, m->stream))) return rc;
        src = m->f;
    }
    return ifa_add(m->pend.x, src, D, 0, m->pend.out, m->stream);
}

} // namespace ifae

extern "C" {

int ifa_model_tp_post_attn(ifa_model *m, int layer, const void *reduced_f16)
{
    IFA_REQUIRE(m && reduced_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_post_attn: bad arguments");
    const size_t D = (size_t)m->cfg.dim;
    const Tensor &b = m->layers[(size_t)layer].t[T_WO_B];
    const Layer &Lp = m->layers[(size_t)layer];
    if (m->cfg.parallel_attn || m->cfg.share_input) {
        // parallel attention / shared MLP input (Falcon, GPT-J/NeoX style; inference_worker.cc:847-851, 941-947): the
        // attention branch does NOT take the residual here -- attention output, FFN output and the layer input are summed
        // once after the FFN (ifa_model_tp_post_ffn).  a = merged product (+ bias once, after the merge :1388-1390)
        if (b.present()) return ifa_add(reduced_f16, b.data, D, 0, m->a, m->stream);
        IFA_HIP_CHECK(hipMemcpyAsync(m->a, reduced_f16, D * 2, hipMemcpyDeviceToDevice, m->stream));
        return IFA_OK;
    }
    // the seam sum can ride in the consumer's prologue only where that prologue exists: RMS-norm models (Std-norm ones
    // run the op-level norm kernel in front of a prologue-free GEMV)
    if (m->opt_tp_fuse_add && m->cfg.norm_kind == 0 && Lp.t[T_FFN_NORM].present() && !(m->cfg.experts > 0 && Lp.t[T_MOE_GATE].present())) {
        // a = x + (reduced + bias): left to the W1/W3 kernel's prologue (ifa_model_tp_ffn)
        m->pend.x = m->x; m->pend.add = (const half_t *)reduced_f16; m->pend.bias = (const half_t *)b.data; m->pend.out = m->a;
        m->pend.on = true;
        return IFA_OK;
    }
    const void *src = reduced_f16;
    int rc;
    if (b.present()) {     // bias once, after the merge (inference_worker.cc:1388-1390)
        if ((rc = ifa_add(reduced_f16, b.data, D, 0, m->a, m->stream))) return rc;
        src = m->a;
    }
    return ifa_add(m->x, src, D, 0, m->a, m->stream);      // Add(out, layer_input, out)
}

int ifa_model_tp_ffn(ifa_model *m, int layer, void *partial_out_f16)
{
    IFA_REQUIRE(m && partial_out_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_ffn: bad arguments");
    int rc;
    Layer &L = m->layers[(size_t)layer];
    if (m->cfg.experts > 0 && L.t[T_MOE_GATE].present()) {     // MoE: every rank routes identically (replicated gate)
        if ((rc = launch_moe_router(m, layer))) return rc;
        for (int k = 0; k < m->cfg.moe_top_k; k++) {
            if ((rc = launch_ffn13(m, layer, k))) return rc;
            if ((rc = launch_w2(m, layer, nullptr, (half_t *)partial_out_f16, k, false))) return rc;
        }
        return IFA_OK;
    }
    if ((rc = launch_ffn13(m, layer, -1, m->x))) return rc;       // (m->x: the layer input, the FFN input of shared-input models)
    return launch_w2(m, layer, nullptr, (half_t *)partial_out_f16);
}

int ifa_model_tp_post_ffn(ifa_model *m, int layer, const void *reduced_f16)
{
    IFA_REQUIRE(m && reduced_f16 && layer >= 0 && layer < m->cfg.layers, "ifa_model_tp_post_ffn: bad arguments");
    const size_t D = (size_t)m->cfg.dim;
    const Tensor &b = m->layers[(size_t)layer].t[T_W2_B];
    if (m->cfg.parallel_attn || m->cfg.share_input) {
        // next layer input = ((merged FFN product + bias) + attention output) + layer input: the order of the fused
        // single-device epilogue (residual, then residual2; inference_worker.cc:936, 941-947)
        const void *src = reduced_f16;
        int rc;
        if (b.present()) {
            if ((rc = ifa_add(reduced_f16, b.data, D, 0, m->f, m->stream))) return rc;
            src = m->f;
        }
        if ((rc = ifa_add(m->a, src, D, 0, m->f, m->stream))) return rc;
        return ifa_add(m->f, m->x, D, 0, m->x, m->stream);
    }
    if (m->opt_tp_fuse_add && m->cfg.norm_kind == 0 && layer + 1 < m->cfg.layers) {
        // next layer input = a + (reduced + bias): left to the next QKV kernel's prologue (ifa_model_tp_attn)
        m->pend.x = m->a; m->pend.add = (const half_t *)reduced_f16; m->pend.bias = (const half_t *)b.data; m->pend.out = m->x;
        m->pend.on = true;
        return IFA_OK;
    }
    const void *src = reduced_f16;
    int rc;
    if (b.present()) {
        if ((rc = ifa_add(reduced_f16, b.data, D, 0, m->f, m->stream))) return rc;
        src = m->f;
    }
    return ifa_add(src, m->a, D, 0, m->x, m->stream);       // Add(layer_out, ff_out, residual)
}

int ifa_model_tp_logits(ifa_model *m, void *logits_shard_out_f16)
{
    IFA_REQUIRE(m && logits_shard_out_f16, "ifa_model_tp_logits: bad arguments");
    IFA_REQUIRE(m->g[T_LM_HEAD].present(), "ifa_model_tp_logits: this worker holds no lm_head (not the last pipeline stage)");
    { int rcf = tp_flush_pending(m); if (rcf) return rcf; }
    return launch_lm(m, m->x, (half_t *)logits_shard_out_f16);
}

int ifa_model_tp_set_token(ifa_model *m, const int *token_dev)
{
    IFA_REQUIRE(m && token_dev, "ifa_model_tp_set_token: bad arguments");
    k_tp_set_token<<<1, 1, 0, m->stream>>>(m->state, token_dev, ifa_model::RING);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // extern "C"

namespace ifae {

// ---- the whole multi-GPU decode step driven from C: worker segments + RCCL collectives (csrc/ifa_comm.hip) on the
// worker's stream, the distributed greedy argmax over the vocabulary-sharded lm_head, token / position fed back in
// device memory; captured once as a hipGraph and replayed per token (tensor-parallel groups; pipelines run eagerly).
int tp_buffers(ifa_model *m)
{
    if (m->tp_a) return IFA_OK;
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim;
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_a, D * 2));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_f, D * 2));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_hid, D * 2));
    IFA_HIP_CHECK(hipMalloc((void **)&m->tp_logits, std::max<size_t>(m->g[T_LM_HEAD].rows, 1) * 2));
    return tp_argmax_scratch(m, 1);
}

// want_token = false (all but the last token of a prompt): the layers run, the lm_head / argmax / token exchange do not
int tp_step(ifa_model *m, const ifa_tp_topology &t, int token, int pos, bool want_token, void *logits_copy)
{
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim;
    ifa_stream s = (ifa_stream)m->stream;
    const int tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
    const bool merge = t.tp && (tp_size > 1 || t.force_collectives);
    int rc;
    if (t.stage == 0) { if ((rc = ifa_model_tp_begin(m, token, pos))) return rc; }
    else {
        if ((rc = ifa_recv(t.world, m->tp_hid, D * 2, t.prev_rank, s))) return rc;
        if ((rc = ifa_model_tp_begin_hidden(m, m->tp_hid, pos))) return rc;
    }
    for (int l = 0; l < c.layers; l++) {
        if ((rc = ifa_model_tp_attn(m, l, m->tp_a))) return rc;
        if (merge && (rc = ifa_allreduce_sum_f16(t.tp, m->tp_a, m->tp_a, D, s))) return rc;
        if ((rc = ifa_model_tp_post_attn(m, l, m->tp_a))) return rc;
        if ((rc = ifa_model_tp_ffn(m, l, m->tp_f))) return rc;
        if (merge && (rc = ifa_allreduce_sum_f16(t.tp, m->tp_f, m->tp_f, D, s))) return rc;
        if ((rc = ifa_model_tp_post_ffn(m, l, m->tp_f))) return rc;
    }
    if (t.n_stages > 1 && t.next_rank >= 0) {      // not the last group: hand the layer output on, then wait for the token
        if ((rc = ifa_model_tp_hidden(m, m->tp_hid))) return rc;
        if ((rc = ifa_send(t.world, m->tp_hid, D * 2, t.next_rank, s))) return rc;
        if (!want_token) return IFA_OK;
        if ((rc = ifa_broadcast(t.world, m->tp_tok, 4, t.token_src, s))) return rc;
        return ifa_model_tp_set_token(m, m->tp_tok);
    }
    if (!want_token && !logits_copy) { m->pend.on = false; return IFA_OK; }     // (the pending seam sum of the last layer is not needed)
    if ((rc = ifa_model_tp_logits(m, m->tp_logits))) return rc;
    if (logits_copy) IFA_HIP_CHECK(hipMemcpyAsync(logits_copy, m->tp_logits, m->g[T_LM_HEAD].rows * 2, hipMemcpyDeviceToDevice, m->stream));
    if (!want_token) return IFA_OK;
    if ((rc = tp_pick_rows(m, t, m->tp_logits, m->g[T_LM_HEAD].rows, (int)m->g[T_LM_HEAD].rows, 1))) return rc;
    if (t.n_stages > 1 && (rc = ifa_broadcast(t.world, m->tp_tok, 4, t.token_src, s))) return rc;
    return ifa_model_tp_set_token(m, m->tp_tok);
}

int tp_check(ifa_model *m, const ifa_tp_topology *topo, const char *who)
{
    IFA_REQUIRE(m && topo, "%s: null pointer", who);
    const ifa_tp_topology &t = *topo;
    const int tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
    IFA_REQUIRE(tp_size <= 64, "%s: group of %d ranks", who, tp_size);
    IFA_REQUIRE(t.n_stages >= 1 && t.stage >= 0 && t.stage < t.n_stages, "%s: stage %d of %d", who, t.stage, t.n_stages);
    IFA_REQUIRE(t.n_stages == 1 || t.world, "%s: layer groups need the job-wide communicator", who);
    int rc = tp_ready(m);
    if (rc) return rc;
    return tp_buffers(m);
}

// One Infer() step of a query over the partition: n_tokens new tokens at positions [start_pos, start_pos + n_tokens),
// fed through the decode path one after the other (the merges are [dim] vectors); the greedy next token of the last
// one comes back on every rank.  logits_shard_out_dev (nullable, last device group): this rank's lm_head rows of every
// token, [n_tokens][shard rows] F16 (return_output_tensors).
// A bounded wait of the one-shot exchange that gave up (a peer that never arrived) left this rank without a sum -- and its epoch
// one behind its peers'.  Every partition entry point checks after its stream sync: the call fails (the engine then aborts the
// group), the captured steps are dropped and this communicator keeps RCCL for every size from now on.
int tp_oneshot_status(ifa_model *m, const ifa_tp_topology &t, const char *who)
{
    if (!t.tp || !ifa_comm_oneshot(t.tp)) return IFA_OK;
    const int st = ifa_comm_status(t.tp);
    if (st == 0) return IFA_OK;
    (void)ifa_comm_set_oneshot(t.tp, 0);
    drop_graphs(m);
    return ifa_fail(IFA_ERR_STATE, "%s: a wait inside the one-shot all-reduce gave up (epoch %d): a peer did not arrive; the exchange is off for this communicator", who, st);
}

} // namespace ifae

extern "C" {

int ifa_model_tp_prefill(ifa_model *m, const ifa_tp_topology *topo, const int *tokens_host, int n_tokens, int start_pos,
                         void *logits_shard_out_dev, int *next_token_host)
{
    IFA_REQUIRE(tokens_host && n_tokens >= 1, "ifa_model_tp_prefill: no tokens");
    int rc = tp_check(m, topo, "ifa_model_tp_prefill");
    if (rc) return rc;
    IFA_REQUIRE(start_pos >= 0 && start_pos + n_tokens <= m->cfg.max_ctx, "ifa_model_tp_prefill: positions %d..%d exceed max_ctx %d",
                start_pos, start_pos + n_tokens, m->cfg.max_ctx);
    if (n_tokens > 1) {
        // T > 1: the op-by-op step over all tokens at once (row-sliced GEMMs on the MFMA kernels, [T][dim] merges after
        // wo and w2, [T][dim] hand-over between layer groups) -- the reference's MatrixMultiplication branch for T > 1
        // (inference_worker.cc:2364-2432) with its merge of token_num x dim values (:2148-2195)
        m->topo = topo;
        rc = forward_ops(m, tokens_host, n_tokens, start_pos, logits_shard_out_dev, next_token_host);
        m->topo = nullptr;
        if (rc == IFA_OK) { (void)hipStreamSynchronize(m->stream); rc = tp_oneshot_status(m, *topo, "ifa_model_tp_prefill"); }
        return rc;
    }
    const size_t shard = m->g[T_LM_HEAD].present() ? m->g[T_LM_HEAD].rows * 2 : 0;
    m->host_pinned[0] = tokens_host[0]; m->host_pinned[1] = start_pos; m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, m->stream));
    for (int i = 0; i < n_tokens; i++) {
        void *lg = (logits_shard_out_dev && shard) ? (char *)logits_shard_out_dev + (size_t)i * shard : nullptr;
        if ((rc = tp_step(m, *topo, tokens_host[i], start_pos + i, i + 1 == n_tokens, lg))) return rc;
    }
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 4, m->tp_tok, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    if ((rc = tp_oneshot_status(m, *topo, "ifa_model_tp_prefill"))) return rc;
    if (next_token_host) *next_token_host = m->host_pinned[4];
    return IFA_OK;
}

// Dynamic batching over a tensor-parallel group: one new token for each of n queries (ifa_model_decode_batch) with the
// two merges per layer over [n][dim] and one distributed argmax per row
int ifa_model_tp_decode_batch(ifa_model *m, const ifa_tp_topology *topo, int n, const int *tokens_host, const int *positions_host,
                              const int *kv_slots_host, int *next_tokens_host, void *logits_shard_out_dev)
{
    IFA_REQUIRE(n >= 1 && n <= ifa_model::RING && tokens_host && positions_host && kv_slots_host, "ifa_model_tp_decode_batch: bad arguments");
    int rc = tp_check(m, topo, "ifa_model_tp_decode_batch");
    if (rc) return rc;
    m->topo = topo;
    rc = forward_batch(m, n, tokens_host, positions_host, kv_slots_host, next_tokens_host, logits_shard_out_dev);
    m->topo = nullptr;
    if (rc == IFA_OK) { (void)hipStreamSynchronize(m->stream); rc = tp_oneshot_status(m, *topo, "ifa_model_tp_decode_batch"); }
    return rc;
}

int ifa_model_tp_decode(ifa_model *m, const ifa_tp_topology *topo, int first_token, int start_pos, int n_steps,
                        int *out_tokens_host, float *elapsed_ms)
{
    IFA_REQUIRE(out_tokens_host, "ifa_model_tp_decode: null pointer");
    IFA_REQUIRE(n_steps >= 1 && n_steps <= ifa_model::RING, "ifa_model_tp_decode: n_steps %d (1..%d)", n_steps, ifa_model::RING);
    int rc = tp_check(m, topo, "ifa_model_tp_decode");
    if (rc) return rc;
    IFA_REQUIRE(start_pos >= 0 && start_pos + n_steps <= m->cfg.max_ctx, "ifa_model_tp_decode: positions %d..%d exceed max_ctx %d",
                start_pos, start_pos + n_steps, m->cfg.max_ctx);
    const ifa_tp_topology &t = *topo;
    hipStream_t s = m->stream;
    choose_attn_split(m, start_pos + n_steps);
    // the step counter restarts: the token ring of this call begins at state[8]
    m->host_pinned[0] = first_token; m->host_pinned[1] = start_pos; m->host_pinned[2] = 0;
    IFA_HIP_CHECK(hipMemcpyAsync(m->state, m->host_pinned, 3 * sizeof(int), hipMemcpyHostToDevice, s));
    const bool use_graph = m->opt_graph && t.n_stages == 1 && (!t.tp || ifa_comm_capturable(t.tp));
    {
        ifa_model::TpKey key;
        key.tp = ifa_comm_serial(t.tp); key.world = ifa_comm_serial(t.world); key.tp_size = t.tp ? ifa_comm_size(t.tp) : 1;
        key.stage = t.stage; key.n_stages = t.n_stages; key.prev = t.prev_rank; key.next = t.next_rank; key.src = t.token_src;
        key.voff = t.vocab_offset; key.force = t.force_collectives; key.fuse = m->opt_tp_fuse_add; key.slot = m->cur_slot;
        key.oneshot = t.tp ? ifa_comm_oneshot(t.tp) : 0;
        if (m->tp_graph_exec && !(key == m->tp_key)) {
            (void)hipGraphExecDestroy(m->tp_graph_exec); m->tp_graph_exec = nullptr;
            if (m->tp_graph) { (void)hipGraphDestroy(m->tp_graph); m->tp_graph = nullptr; }
        }
        m->tp_key = key;
    }
    int done = 0;
    if (!(use_graph && m->tp_graph_exec)) {
        // the first step runs eagerly: it creates whatever the collectives allocate lazily, so that the capture below
        // records pure launches
        if ((rc = tp_step(m, t, first_token, start_pos))) return rc;
        done = 1;
        if (use_graph && done < n_steps) {      // (a one-step call has nothing to replay: no capture, no instantiate)
            IFA_HIP_CHECK(hipStreamSynchronize(s));
            IFA_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            rc = tp_step(m, t, -1, -1);
            hipGraph_t gph = nullptr;
            hipError_t e = hipStreamEndCapture(s, &gph);
            if (rc || e != hipSuccess) {
                if (gph) (void)hipGraphDestroy(gph);
                (void)hipGetLastError();
                m->tp_graph_exec = nullptr;      // eager steps below: correctness does not depend on the graph
            } else {
                m->tp_graph = gph;
                if (hipGraphInstantiate(&m->tp_graph_exec, gph, nullptr, nullptr, 0) != hipSuccess) { m->tp_graph_exec = nullptr; (void)hipGetLastError(); }
            }
        }
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (elapsed_ms) { IFA_HIP_CHECK(hipEventCreate(&e0)); IFA_HIP_CHECK(hipEventCreate(&e1)); IFA_HIP_CHECK(hipEventRecord(e0, s)); }
    for (int i = done; i < n_steps; i++) {
        if (use_graph && m->tp_graph_exec) IFA_HIP_CHECK(hipGraphLaunch(m->tp_graph_exec, s));
        else if ((rc = tp_step(m, t, -1, -1))) return rc;
    }
    if (e1) IFA_HIP_CHECK(hipEventRecord(e1, s));
    IFA_HIP_CHECK(hipMemcpyAsync(m->host_pinned + 8, m->state + 8, sizeof(int) * (size_t)n_steps, hipMemcpyDeviceToHost, s));
    IFA_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < n_steps; i++) out_tokens_host[i] = m->host_pinned[8 + i];
    if (elapsed_ms) {
        float ms = 0.0f;
        IFA_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *elapsed_ms = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    // a bounded wait of the one-shot exchange that gave up (a peer that never arrived) left this rank without a sum: the
    // tokens above are not results -- fail the call (the engine then aborts the group) instead of returning them
    return tp_oneshot_status(m, t, "ifa_model_tp_decode");
}

} // extern "C"
