// ifa_engine_moe.hip -- mixture of experts in the worker (ProcessGpuLayer_Moe, inference_worker.cc:1924-2146): the router as one
// launch (decode step and batched rows), grouped expert GEMMs / GEMVs of a batch on the device, and the host-routed fallback.
#include "ifa_engine_state.h"

namespace ifae {

// Device-side counterpart of the host routing in moe_ffn (HostTensorOpr::BuildRowsForMoE, host_tensor_opr.cc:190-244):
// top-k by repeated first-maximum, probabilities below 1e-5 dropped, optional renormalisation, experts then visited in
// ascending id order.  Unused slots get weight 0 (hfma(y, 0, acc) == acc).  One thread: E <= 64, k <= 8.
// top-k of one row by ONE wave, lane e = expert e with probability p (lanes >= E: -inf): k_moe_topk's rules -- repeated first
// maximum, probabilities below 1e-5 dropped, optional renormalisation in pick order, kept experts in ascending id, unused
// slots expert 0 / weight 0.  (The one-thread form walked local arrays that live in scratch: ~20 us of dependent loads.)
__device__ __forceinline__ void moe_topk_wave(float p, int lane, int E, int top_k, int norm_topk, int *__restrict__ sel, half_t *__restrict__ wout, int unused_id = 0)
{
    bool used = lane >= E;
    int idx[8]; float w[8];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { idx[k] = 0x7FFFFFFF; w[k] = 0.0f; }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= top_k || k >= E) break;
        const float mx = wave_max(used ? -INFINITY : p);
        const unsigned long long cand = __ballot(!used && p == mx);
        if (!cand) break;
        const int best = __ffsll((long long)cand) - 1;
        if (lane == best) used = true;
        const float pb = __shfl(p, best);
        if (pb < 0.00001f) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) if (j == n) { idx[j] = best; w[j] = pb; }
        n++;
    }
    if (norm_topk && n > 0) {
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n) sum = sum + w[j];
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n) w[j] = w[j] / sum;
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j >= n) continue;
            int rank = 0;
#pragma unroll
            for (int j2 = 0; j2 < 8; j2++) rank += (j2 < n && idx[j2] < idx[j]) ? 1 : 0;
            sel[rank] = idx[j]; wout[rank] = f2h(w[j]);
        }
        for (int slot = n; slot < top_k; slot++) { sel[slot] = unused_id; wout[slot] = (half_t)0; }
    }
}

__global__ void __launch_bounds__(64) k_moe_topk(const half_t *__restrict__ probs_h, int E, int top_k, int norm, int *__restrict__ sel, half_t *__restrict__ wout)
{
    const int lane = threadIdx.x & 63;      // launched with one wave
    moe_topk_wave(lane < E ? h2f(probs_h[lane]) : -INFINITY, lane, E, top_k, norm, sel, wout);
}

// The router of a fused decode step in ONE launch (one workgroup of 8 waves): RMS norm of the layer's FFN input, the F16
// gate GEMV, softmax, top-k -- each with the arithmetic of the kernel it replaces (k_layernorm<0>: canonical RMS order;
// k_gemv_f16w: lane l takes chunks l, l + 64, ... as one fp32 fma chain, then the wave butterfly; k_softmax: 32-lane
// max / sum trees, half-rounded exponentials; k_moe_topk), so the routing is bit-identical to the four-launch sequence.
// The normalised input is also written out (hn) for parity checks.  cols % 8 == 0, cols <= 16384, E <= 64.
__global__ void __launch_bounds__(512) k_dec_moe_router(const half_t *__restrict__ x, const half_t *__restrict__ nw, const half_t *__restrict__ nb,
                                                        float multi_base, float eps, int cols, const half_t *__restrict__ gate_w, int E, int top_k,
                                                        int norm_topk, half_t *__restrict__ hn_out, half_t *__restrict__ probs_out,
                                                        int *__restrict__ sel, half_t *__restrict__ wout, int unused_id)
{
    // one workgroup per row (the batched step: blockIdx.x = query; a decode step: one row); rows are `cols` apart, a row's
    // routing top_k slots apart, unused slots carry `unused_id` (0 for the fused decode step: weight 0 makes them no-ops; -1 for
    // the list builder of the batched step, k_moe_route_rows' convention)
    x += (size_t)blockIdx.x * cols;
    if (hn_out) hn_out += (size_t)blockIdx.x * cols;
    if (probs_out) probs_out += (size_t)blockIdx.x * E;
    sel += (size_t)blockIdx.x * top_k; wout += (size_t)blockIdx.x * top_k;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *xs = reinterpret_cast<half_t *>(smem);                                                    // [cols]
    float *part = reinterpret_cast<float *>(smem + (((size_t)cols * 2 + 15) & ~(size_t)15));         // [64] group sums
    half_t *probs = reinterpret_cast<half_t *>(part + 64);                                            // [64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunks = cols >> 3;
    // the first gate row of every wave does not depend on the input: requested now (clamped, unconditional)
    u32x4 w_first[8];
    {
        const u32x4 *wrow = reinterpret_cast<const u32x4 *>(gate_w + (size_t)min(wave, E - 1) * cols);
#pragma unroll
        for (int j = 0; j < 8; j++) w_first[j] = wrow[min(lane + 64 * j, chunks - 1)];
    }
    // ---- RMS norm (or a plain copy when the layer has no FFN norm: nw == nullptr and eps < 0)
    const bool do_norm = eps >= 0.0f;
    for (int k = 0; k * 512 < chunks; k++) {
        const int c = tid + k * 512;
        rms_h8 v8;
#pragma unroll
        for (int e = 0; e < 8; e++) v8[e] = (half_t)0;
        if (c < chunks) { v8 = *reinterpret_cast<const rms_h8 *>(x + (size_t)c * 8); *reinterpret_cast<rms_h8 *>(xs + (size_t)c * 8) = v8; }
        const float pg = wave_sum(rms_chunk_sq(v8));
        if (lane == 0) part[wave + k * 8] = pg;
    }
    __syncthreads();
    if (do_norm) {
        const float scale = rms_scale_of(rms_total(part, (chunks + 63) >> 6), cols, eps);
        for (int c = tid; c < chunks; c += 512) {
            const rms_h8 v8 = *reinterpret_cast<const rms_h8 *>(xs + (size_t)c * 8);
            rms_h8 w8 = v8, b8 = v8;           // (one 16-byte request each: element-wise 2-byte loads behind branches took ~12 us)
            if (nw) w8 = *reinterpret_cast<const rms_h8 *>(nw + (size_t)c * 8);
            if (nb) b8 = *reinterpret_cast<const rms_h8 *>(nb + (size_t)c * 8);
            rms_h8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const half_t we = w8[e], be = b8[e];
                o[e] = f2h(rms_apply((float)v8[e], scale, nw ? &we : nullptr, nb ? &be : nullptr, multi_base));
            }
            *reinterpret_cast<rms_h8 *>(xs + (size_t)c * 8) = o;
            if (hn_out) *reinterpret_cast<rms_h8 *>(hn_out + (size_t)c * 8) = o;
        }
        __syncthreads();
    }
    // ---- gate GEMV: a wave per expert row, eight 16-byte requests in flight per lane (wave w's first expert row was
    // requested at the top of the kernel, before the norm)
    for (int e = wave; e < E; e += 8) {
        const u32x4 *wrow = reinterpret_cast<const u32x4 *>(gate_w + (size_t)e * cols);
        const u32x4 *xv = reinterpret_cast<const u32x4 *>(xs);
        float acc = 0.0f;
        for (int c0 = lane; c0 < chunks; c0 += 512) {
            u32x4 wr[8];
            if (e == wave && c0 == lane) {
#pragma unroll
                for (int j = 0; j < 8; j++) wr[j] = w_first[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) wr[j] = wrow[min(c0 + 64 * j, chunks - 1)];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) if (c0 + 64 * j < chunks) acc = dot8_f16(wr[j], xv[c0 + 64 * j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) probs[e] = f2h(acc);
    }
    __syncthreads();
    // ---- softmax over the E gate values (k_softmax with one row, scale 1, no mask) and the top-k, by the first 32 lanes
    if (tid < 32) {
        float mx = -INFINITY;
        for (int xi = tid; xi < E; xi += 32) mx = fmaxf(mx, 1.0f * h2f(probs[xi]));
#pragma unroll
        for (int mk = 16; mk > 0; mk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, mk, 32));
        float sum = 0.0f;
        for (int xi = tid; xi < E; xi += 32) {
            const float v = 1.0f * h2f(probs[xi]);
            const float ex = expf(v - mx);
            sum = sum + ex;
            probs[xi] = f2h(ex);
        }
#pragma unroll
        for (int mk = 16; mk > 0; mk >>= 1) sum = sum + __shfl_xor(sum, mk, 32);
        const float inv = 1.0f / sum;
        for (int xi = tid; xi < E; xi += 32) { const half_t pr = f2h(h2f(probs[xi]) * inv); probs[xi] = pr; if (probs_out) probs_out[xi] = pr; }
    }
    __syncthreads();
    // ---- top-k by wave 0, lane e = expert e (k_moe_topk's rules: repeated first maximum, probabilities below 1e-5 dropped,
    // optional renormalisation in pick order, kept experts in ascending id, unused slots expert 0 / weight 0).  The
    // one-thread form walks local arrays that live in scratch: ~20 us of dependent scratch loads per layer.
    if (wave == 0) moe_topk_wave(lane < E ? h2f(probs[lane]) : -INFINITY, lane, E, top_k, norm_topk, sel, wout, unused_id);
}

// router of one MoE layer on the device: the same norm / GEMV / softmax kernels the op path runs, then k_moe_topk
int launch_moe_router(ifa_model *m, int l)
{
    const ifa_model_config &c = m->cfg;
    Layer &L = m->layers[(size_t)l];
    int rc;
    Tensor none;
    const Tensor &gw = L.t[T_MOE_GATE];
    if (m->opt_moe_router_fused && c.norm_kind == 0 && gw.dtype == F16 && c.dim % 8 == 0 && c.dim <= 16384 && c.experts <= 64 && (int)gw.cols == c.dim) {
        const bool has_norm = L.t[T_FFN_NORM].present();
        const size_t smem = (((size_t)c.dim * 2 + 15) & ~(size_t)15) + 64 * 4 + 64 * 2;
        k_dec_moe_router<<<1, 512, smem, m->stream>>>(m->a, has_norm ? (const half_t *)L.t[T_FFN_NORM].data : nullptr,
                                                      has_norm ? (const half_t *)L.t[T_FFN_NORM_B].data : nullptr, c.ffn_norm_base, has_norm ? c.eps : -1.0f,
                                                      c.dim, (const half_t *)gw.data, c.experts, c.moe_top_k, c.moe_norm_topk, m->hn, m->moe_gate, m->moe_route,
                                                      reinterpret_cast<half_t *>(reinterpret_cast<char *>(m->moe_route) + 32), 0);
        IFA_LAUNCH_CHECK();
        return IFA_OK;
    }
    const half_t *ff_n = m->a;
    if (L.t[T_FFN_NORM].present()) {
        if ((rc = norm_rows(m, m->a, 1, L.t[T_FFN_NORM], L.t[T_FFN_NORM_B], m->hn, c.ffn_norm_base))) return rc;
        ff_n = m->hn;
    }
    if ((rc = matmul(m, ff_n, 1, L.t[T_MOE_GATE], none, m->moe_gate))) return rc;
    if ((rc = ifa_softmax(m->moe_gate, c.experts, 1, 1, -1, 1.0f, (ifa_stream)m->stream))) return rc;
    k_moe_topk<<<1, 64, 0, m->stream>>>(m->moe_gate, c.experts, c.moe_top_k, c.moe_norm_topk, m->moe_route,
                                        reinterpret_cast<half_t *>(reinterpret_cast<char *>(m->moe_route) + 32));
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int moe_ffn(ifa_model *m, Layer &L, const half_t *ff_n, int T)
{
    if (T > 1 && moe_device_ok(m, L)) return moe_ffn_device(m, L, ff_n, T);
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim; const int E = c.experts;
    ifa_stream s = (ifa_stream)m->stream;
    int rc;
    IFA_REQUIRE(E <= 64 && c.moe_top_k >= 1 && c.moe_top_k <= 8, "MoE: experts %d / top_k %d out of range", E, c.moe_top_k);
    IFA_REQUIRE((int)L.experts.size() == E * 3, "MoE: expert tensors missing");
    Tensor none;
    half_t *gate = m->moe_gate;
    if ((rc = matmul(m, ff_n, T, L.t[T_MOE_GATE], none, gate))) return rc;
    if ((rc = ifa_softmax(gate, E, T, 1, -1, 1.0f, s))) return rc;
    std::vector<uint16_t> probs_h((size_t)T * E);
    IFA_HIP_CHECK(hipMemcpyAsync(probs_h.data(), gate, probs_h.size() * 2, hipMemcpyDeviceToHost, m->stream));
    IFA_HIP_CHECK(hipMemsetAsync(m->f, 0, (size_t)T * D * 2, m->stream));
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));
    // per expert: the rows routed to it (token order) and their weights  (BuildRowsForMoE)
    std::vector<std::vector<int>> rows((size_t)E);
    std::vector<std::vector<uint16_t>> wts((size_t)E);
    for (int t = 0; t < T; t++) {
        float probs[64]; int idx[8]; float w[8]; bool used[64] = {false};
        for (int e = 0; e < E; e++) probs[e] = (float)__builtin_bit_cast(_Float16, probs_h[(size_t)t * E + e]);
        int n = 0;
        for (int k = 0; k < c.moe_top_k && k < E; k++) {          // first maximum wins, like the host sort
            int best = -1;
            for (int e = 0; e < E; e++) if (!used[e] && (best < 0 || probs[e] > probs[best])) best = e;
            if (best < 0) break;
            used[best] = true;
            if (probs[best] < 0.00001f) continue;
            idx[n] = best; w[n] = probs[best]; n++;
        }
        if (c.moe_norm_topk && n > 0) {
            float sum = 0.0f;
            for (int i = 0; i < n; i++) sum = sum + w[i];
            for (int i = 0; i < n; i++) w[i] = w[i] / sum;
        }
        for (int j = 0; j < n; j++) {
            const _Float16 wh = (_Float16)w[j];
            rows[(size_t)idx[j]].push_back(t);
            wts[(size_t)idx[j]].push_back(__builtin_bit_cast(uint16_t, wh));
        }
    }
    // one upload of every (row, weight) list, experts back to back
    const size_t cap = (size_t)T * (size_t)c.moe_top_k;
    int *pin_rows = m->moe_pin; uint16_t *pin_w = reinterpret_cast<uint16_t *>(m->moe_pin + cap);
    size_t off = 0;
    std::vector<size_t> start((size_t)E, 0);
    for (int e = 0; e < E; e++) {
        start[(size_t)e] = off;
        for (size_t r = 0; r < rows[(size_t)e].size(); r++) { pin_rows[off + r] = rows[(size_t)e][r]; pin_w[off + r] = wts[(size_t)e][r]; }
        off += rows[(size_t)e].size();
    }
    IFA_HIP_CHECK(hipMemcpyAsync(m->moe_idx, pin_rows, off * sizeof(int), hipMemcpyHostToDevice, m->stream));
    IFA_HIP_CHECK(hipMemcpyAsync(m->moe_wdev, pin_w, off * 2, hipMemcpyHostToDevice, m->stream));
    // expert by expert (ascending id): gather its rows, FFN on them as one matrix (T = 1 -> GEMV path, else the
    // MFMA GEMM, exactly the split MatrixMultiplication makes), scatter-add weight * output
    for (int e = 0; e < E; e++) {
        const int n = (int)rows[(size_t)e].size();
        if (n == 0) continue;
        const int *idx_dev = m->moe_idx + start[(size_t)e];
        if ((rc = gather_rows(m, ff_n, idx_dev, n, (int)D, T, m->moe_in, 0.0f))) return rc;
        const Tensor *ew = &L.experts[(size_t)e * 3];
        if ((rc = ffn_dense(m, m->moe_in, n, ew[0], none, ew[2], none, ew[1], none, m->moe_out))) return rc;
        if ((rc = ifa_add_by_row_index(m->f, m->moe_out, (size_t)n, D, idx_dev, m->moe_wdev + start[(size_t)e], s))) return rc;
    }
    IFA_HIP_CHECK(hipStreamSynchronize(m->stream));      // the pinned lists are reused by the next MoE layer
    return IFA_OK;
}

// The same layer without the host (T > 1 rows; ifa_moe.h): routing and the per-expert row lists are built on the device,
// the rows of ALL experts are gathered once (experts ascending, token order inside an expert -- the reference's order), and
// each of the three products is ONE grouped launch over the experts with >= 2 rows (MFMA GEMM tiles, the reference's T > 1
// branch: F16 activations on dequantised weights) plus ONE over the single-row experts (the int8-activation GEMV of its
// T = 1 branch, bit-identical to the op-level kernel).  No D2H copy, no stream synchronisation.  Result in m->f.
bool moe_device_ok(const ifa_model *m, const Layer &L)
{
    const ifa_model_config &c = m->cfg;
    if (!m->opt_moe_device || !L.moe_table_aos || (int)L.experts.size() != c.experts * 3) return false;
    const Tensor &w1 = L.experts[0], &w2 = L.experts[1], &w3 = L.experts[2];
    if (!w3.present() || !ax8_eligible(w1.dtype) || !m->cfg.full_quant_gemv || w1.cols % 32 || w2.cols % 32) return false;
    for (int e = 0; e < c.experts; e++)
        for (int k3 = 0; k3 < 3; k3++) {
            const Tensor &t = L.experts[(size_t)e * 3 + k3], &r = L.experts[(size_t)k3];
            if (!t.present() || t.dtype != r.dtype || t.rows != r.rows || t.cols != r.cols) return false;
        }
    return true;
}

int max_smalls_possible(bool rows_kernel, int E, int cap) { return rows_kernel ? std::min(E, cap / 2) : 0; }

// (also called by the batched step before it starts a capture: nothing is created inside one)
int ensure_side_stream(ifa_model *m)
{
    if (m->side_stream) return IFA_OK;
    IFA_HIP_CHECK(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
    IFA_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    IFA_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    return IFA_OK;
}

// can the rows of a batched step be routed by ONE launch (k_dec_moe_router, a workgroup per row: norm, F16 gate GEMV, softmax, top-k)?
bool moe_router_rows_ok(const ifa_model *m, const Layer &L, int T)
{
    const ifa_model_config &c = m->cfg;
    const Tensor &gw = L.t[T_MOE_GATE];
    return m->opt_moe_router_fused && T <= 32 && c.norm_kind == 0 && gw.dtype == F16 && c.dim % 8 == 0 && c.dim <= 16384 && c.experts <= 64
        && (int)gw.cols == c.dim && L.t[T_FFN_NORM].present();
}

// pre_norm: the rows BEFORE the FFN norm (ff_n is then where the normalised rows go): the batched step's router launch does the norm too
// residual / out: the layer's residual Add in the combine launch, result in `out` (default: the FFN output alone in m->f)
int moe_ffn_device(ifa_model *m, Layer &L, const half_t *ff_n, int T, const half_t *pre_norm, const half_t *residual, half_t *out)
{
    const ifa_model_config &c = m->cfg;
    const size_t D = (size_t)c.dim, F = L.experts[0].rows;
    const int E = c.experts, K = c.moe_top_k, cap = T * K;
    ifa_stream s = (ifa_stream)m->stream;
    int rc;
    Tensor none;
    if (pre_norm) {
        // norm + gate + softmax + top-k of every row as one launch instead of four (each with the arithmetic of the decode step's
        // router: the gate product is the F16 GEMV's fp32 chain, not the GEMM tile's): 25 -> 7 us per layer at 8 queries
        const size_t smem = (((size_t)c.dim * 2 + 15) & ~(size_t)15) + 64 * 4 + 64 * 2;
        k_dec_moe_router<<<dim3((unsigned)T), 512, smem, m->stream>>>(pre_norm, (const half_t *)L.t[T_FFN_NORM].data, (const half_t *)L.t[T_FFN_NORM_B].data,
                                                                      c.ffn_norm_base, c.eps, c.dim, (const half_t *)L.t[T_MOE_GATE].data, E, K, c.moe_norm_topk,
                                                                      const_cast<half_t *>(ff_n), m->moe_gate, m->moe_sel, (half_t *)m->moe_selw, -1);
        IFA_LAUNCH_CHECK();
    } else {
        if ((rc = matmul(m, ff_n, T, L.t[T_MOE_GATE], none, m->moe_gate))) return rc;
        if ((rc = ifa_softmax(m->moe_gate, E, T, 1, -1, 1.0f, s))) return rc;
        if ((rc = ifa_moe_route_topk(m->moe_gate, (size_t)T, E, K, c.moe_norm_topk, m->moe_sel, m->moe_selw, s))) return rc;
    }
    // rows per expert on average >= 96: 128-row tiles (each decoded weight block feeds four MFMA tiles); else 64-row split-K tiles
    const int tile_rows = (cap / std::max(1, E) >= 96) ? 128 : 64;
    // a handful of rows per expert (dynamic batching): experts with 2..small_max rows stream their tiled Q4 weights once
    // (ifa_gemm_rows.hip) instead of filling a 64-row MFMA tile with mostly padding
    const int wdt = L.experts[0].dtype;
    const bool rows_mfma = gemm_rows_use_mfma() && gemm_rows_mfma_ok(F, D, 2) && gemm_rows_mfma_ok(D, F, 2);     // matrix-core variant: up to 16 rows
    const bool rows_kernel = is_q4(wdt) && m->opt_gemm_rows && L.moe_table && cap <= 8 * E
        && (rows_mfma || (gemm_rows_q4_grouped_cap(D) > 0 && gemm_rows_q4_grouped_cap(F) > 0));
    const int small_max = rows_kernel ? 8 : 0;
    auto rows_grouped = [&](const MoeSmallGroup &q, size_t rows, size_t cols, const void *X, void *Y, int ng) {
        return rows_mfma ? gemm_rows_mfma_grouped(q, rows, cols, X, Y, ng, small_max, m->stream) : gemm_rows_q4_grouped(q, rows, cols, X, Y, ng, m->stream);
    };
    if ((rc = moe_build_lists(m->moe_sel, m->moe_selw, T, K, E, tile_rows, small_max, m->moe_idx, m->moe_wdev, m->moe_epos, (MoeTile *)m->moe_tiles,
                              (MoeSingle *)m->moe_singles, (MoeTile *)m->moe_smalls, m->moe_counts, m->stream))) return rc;
    MoeSmallGroup sg;
    sg.smalls = (const MoeTile *)m->moe_smalls; sg.counts = m->moe_counts; sg.wtab_tiled = (const uint8_t *const *)L.moe_table; sg.which_tiled = 0;
    // MO copies of the experts (ensure_mo, built by the batched step before its capture): the small groups then take ONE launch
    // for w1 / w3 with the gated product as its output, and one for w2 -- instead of three launches and an element-wise pass
    sg.wtab_mo = (rows_mfma && m->opt_rows_mo) ? (const uint8_t *const *)L.moe_table_mo : nullptr;
    const bool smalls_mo = sg.wtab_mo != nullptr && max_smalls_possible(rows_kernel, E, cap) > 0;
    const int max_smalls = rows_kernel ? std::min(E, cap / 2) : 0;
    if ((rc = moe_gather(ff_n, m->moe_idx, m->moe_counts, cap, (int)D, m->moe_gin, m->stream))) return rc;
    MoeGroup g;
    g.tiles = (const MoeTile *)m->moe_tiles; g.singles = (const MoeSingle *)m->moe_singles; g.counts = m->moe_counts;
    g.wtab = (const uint8_t *const *)L.moe_table_aos; g.on = 1;
    // (T <= small_max: no expert can collect more rows than the small groups take -- the tile list is empty, its launches are skipped)
    const int max_tiles = (small_max > 0 && T <= small_max) ? 0 : cap / tile_rows + E, max_singles = std::min(E, cap);
    // Single-row experts (round 4): when no expert can collect a tile of rows (max_tiles == 0: a batched decode step) and the small
    // groups gate their own products (MO copies), the singles are the only rows the quantiser / element-wise launches below serve --
    // they then take two launches of the decode GEMV's structure on the tiled expert tables (ifa_decode_singles.h) instead of six
    if (m->opt_moe_singles && max_tiles == 0 && (smalls_mo || max_smalls == 0) && L.moe_table && dec_singles_supported(wdt, F, D, true)
        && dec_singles_supported(wdt, D, F, false)) {
        DecSinglesParams S; memset(&S, 0, sizeof(S));
        S.singles = (const MoeSingle *)m->moe_singles; S.counts = m->moe_counts; S.wtab = (const uint8_t *const *)L.moe_table; S.act_kind = c.act_kind;
        // the singles' two launches on the side stream, the small groups' two on the main one: disjoint rows of g1 / gout, joined in
        // front of the combine (inside a capture the fork / join become graph edges)
        hipStream_t ss = m->stream;
        if (m->opt_moe_overlap && max_smalls) {
            if ((rc = ensure_side_stream(m))) return rc;
            ss = m->side_stream;
            IFA_HIP_CHECK(hipEventRecord(m->ev_fork, m->stream));
            IFA_HIP_CHECK(hipStreamWaitEvent(ss, m->ev_fork, 0));
        }
        S.which = 0; S.X = m->moe_gin; S.ldx = (int)D; S.Y = m->moe_g1; S.ldy = (int)F; S.rows = (int)F; S.cols = (int)D; S.nblk = (int)(D / 32);
        if ((rc = dec_singles_launch(wdt, S, true, max_singles, ss))) return rc;                      // act(w1 x) * (w3 x)
        S.which = 2; S.X = m->moe_g1; S.ldx = (int)F; S.Y = m->moe_gout; S.ldy = (int)D; S.rows = (int)D; S.cols = (int)F; S.nblk = (int)(F / 32);
        if ((rc = dec_singles_launch(wdt, S, false, max_singles, ss))) return rc;                     // w2
        if (max_smalls) {
            sg.which_tiled = 0;
            if ((rc = gemm_rows_mo_grouped(sg, F, D, m->moe_gin, m->moe_g1, max_smalls, 1, c.act_kind, m->stream))) return rc;
            sg.which_tiled = 2;
            if ((rc = gemm_rows_mo_grouped(sg, D, F, m->moe_g1, m->moe_gout, max_smalls, 0, c.act_kind, m->stream))) return rc;
        }
        if (ss != m->stream) {
            IFA_HIP_CHECK(hipEventRecord(m->ev_join, ss));
            IFA_HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
        }
        return moe_combine(m->moe_gout, m->moe_epos, m->moe_selw, T, K, (int)D, out ? out : m->f, m->stream, residual);
    }
    // single-row experts take the quantised row (TensorOpr::Quantize in front of Gemv_AX, inference_worker.cc:1772-1774)
    if ((rc = ifa_quantize_act_q8(m->moe_gin, (size_t)cap, D, m->moe_xq_in, s))) return rc;
    g.which = 0;
    if ((rc = gemm_q_grouped(wdt, g, F, D, m->moe_gin, m->moe_g1, max_tiles, tile_rows, m->stream))) return rc;
    if ((rc = gemv_ax8_grouped(wdt, g, F, D, m->moe_xq_in, m->moe_g1, max_singles, m->stream))) return rc;
    if (max_smalls && !smalls_mo && (rc = rows_grouped(sg, F, D, m->moe_gin, m->moe_g1, max_smalls))) return rc;
    g.which = 2; sg.which_tiled = 1;
    if ((rc = gemm_q_grouped(wdt, g, F, D, m->moe_gin, m->moe_g3, max_tiles, tile_rows, m->stream))) return rc;
    if ((rc = gemv_ax8_grouped(wdt, g, F, D, m->moe_xq_in, m->moe_g3, max_singles, m->stream))) return rc;
    if (max_smalls && !smalls_mo && (rc = rows_grouped(sg, F, D, m->moe_gin, m->moe_g3, max_smalls))) return rc;
    if ((rc = ifa_activation_mul(c.act_kind, m->moe_g1, m->moe_g3, (size_t)cap * F, m->moe_g1, s))) return rc;
    if (max_smalls && smalls_mo) {      // (after the element-wise pass over all rows: the small groups' rows of g1 are written here, gated)
        sg.which_tiled = 0;
        if ((rc = gemm_rows_mo_grouped(sg, F, D, m->moe_gin, m->moe_g1, max_smalls, 1, c.act_kind, m->stream))) return rc;
    }
    if ((rc = ifa_quantize_act_q8(m->moe_g1, (size_t)cap, F, m->moe_xq_mid, s))) return rc;
    g.which = 1;
    if ((rc = gemm_q_grouped(wdt, g, D, F, m->moe_g1, m->moe_gout, max_tiles, tile_rows, m->stream))) return rc;
    if ((rc = gemv_ax8_grouped(wdt, g, D, F, m->moe_xq_mid, m->moe_gout, max_singles, m->stream))) return rc;
    sg.which_tiled = 2;
    if (max_smalls && smalls_mo) { if ((rc = gemm_rows_mo_grouped(sg, D, F, m->moe_g1, m->moe_gout, max_smalls, 0, c.act_kind, m->stream))) return rc; }
    else if (max_smalls && (rc = rows_grouped(sg, D, F, m->moe_g1, m->moe_gout, max_smalls))) return rc;
    return moe_combine(m->moe_gout, m->moe_epos, m->moe_selw, T, K, (int)D, out ? out : m->f, m->stream, residual);
}

} // namespace ifae
