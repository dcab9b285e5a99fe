// ifa_moe.hip -- device-side work lists and the combine step of a mixture-of-experts layer over T > 1 rows (ifa_moe.h).
// Replaces the host round trip of ProcessGpuLayer_Moe (src/transformer/inference_worker.cc:1924-2146): router
// probabilities -> D2H -> HostTensorOpr::BuildRowsForMoE (src/tensor/host_tensor_opr.cc:190-244) -> per-expert loops.
#include "ifa_host.h"
#include "ifa_device.h"
#include "ifa_moe.h"

namespace ifa {

// sel / wsel [T][top_k]: the routing of every row (k_moe_route_rows: experts ascending, -1 = unused slot).
// One workgroup.  Output, all in the reference's order (experts ascending, token order inside an expert):
//   idx[pos] = token row of entry pos, wdev[pos] = its weight, epos[t * top_k + j] = pos of row t's j-th expert (-1 none)
//   tiles / singles / counts as described in ifa_moe.h
__global__ void __launch_bounds__(1024) k_moe_build(const int *__restrict__ sel, const half_t *__restrict__ wsel, int T, int top_k, int E, int tile_rows,
                                                    int small_max, int *__restrict__ idx, half_t *__restrict__ wdev, int *__restrict__ epos,
                                                    MoeTile *__restrict__ tiles, MoeSingle *__restrict__ singles, MoeTile *__restrict__ smalls,
                                                    int *__restrict__ counts)
{
    __shared__ int cnt[64], start[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
    const int N = T * top_k;
    if (tid < 64) cnt[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += blockDim.x) {
        const int e = sel[i];
        if (e >= 0 && e < E) atomicAdd(&cnt[e], 1);
        epos[i] = -1;
    }
    __syncthreads();
    if (tid == 0) {
        int off = 0, nt = 0, ns = 0, nm = 0;
        for (int e = 0; e < E; e++) {
            start[e] = off;
            const int c = cnt[e];
            if (c == 1) { singles[ns].expert = e; singles[ns].pos = off; ns++; }
            else if (c >= 2 && c <= small_max) { smalls[nm].expert = e; smalls[nm].row0 = off; smalls[nm].nrows = c; smalls[nm].pad = 0; nm++; }
            else for (int r = 0; r < c; r += tile_rows) { tiles[nt].expert = e; tiles[nt].row0 = off + r; tiles[nt].nrows = min(tile_rows, c - r); tiles[nt].pad = 0; nt++; }
            off += c;
        }
        counts[0] = off; counts[1] = nt; counts[2] = ns; counts[3] = nm;
    }
    __syncthreads();
    // stable fill: a wave per expert walks the entries in order, 64 at a time (ballot + prefix count)
    for (int e = wave; e < E; e += nwaves) {
        int running = start[e];
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            const bool mine = i < N && sel[i] == e;
            const unsigned long long mask = __ballot(mine);
            if (mine) {
                const int pos = running + __popcll(mask & ((1ull << lane) - 1ull));
                idx[pos] = i / top_k; wdev[pos] = wsel[i]; epos[i] = pos;
            }
            running += __popcll(mask);
        }
    }
}

// dst[pos][:] = src[idx[pos]][:] for pos < counts[0]
__global__ void __launch_bounds__(256) k_moe_gather(const half_t *__restrict__ src, const int *__restrict__ idx, const int *__restrict__ counts,
                                                    int dim, half_t *__restrict__ dst)
{
    const int pos = blockIdx.y;
    if (pos >= counts[0]) return;
    const size_t t = (size_t)idx[pos];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < dim / 8; c += gridDim.x * blockDim.x)
        reinterpret_cast<u32x4_t *>(dst + (size_t)pos * dim)[c] = reinterpret_cast<const u32x4_t *>(src + t * dim)[c];
}

// out[t][d] = hfma(y[pos_j][d], w_j, ...) over row t's experts in ascending expert order, starting from 0
// (the memset + AddByRowIdx_Kernel sequence of the reference's expert loop, src/kernels/binary_tensor_opr.h:80-125)
__global__ void __launch_bounds__(256) k_moe_combine(const half_t *__restrict__ y, const int *__restrict__ epos, const half_t *__restrict__ wsel,
                                                     int T, int top_k, int dim, half_t *__restrict__ out, const half_t *__restrict__ residual)
{
    const int t = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T || d >= dim) return;
    half_t acc = (half_t)0;
    for (int j = 0; j < top_k; j++) {
        const int pos = epos[t * top_k + j];
        if (pos >= 0) acc = __builtin_fmaf16(y[(size_t)pos * dim + d], wsel[t * top_k + j], acc);
    }
    // residual: the layer's TensorOpr::Add behind the expert loop (its own half rounding), in the same launch
    if (residual) acc = f2h(h2f(residual[(size_t)t * dim + d]) + h2f(acc));
    out[(size_t)t * dim + d] = acc;
}

int moe_build_lists(const int *sel, const void *wsel, int T, int top_k, int E, int tile_rows, int small_max, int *idx, void *wdev, int *epos,
                    MoeTile *tiles, MoeSingle *singles, MoeTile *smalls, int *counts, hipStream_t s)
{
    k_moe_build<<<1, 1024, 0, s>>>(sel, (const half_t *)wsel, T, top_k, E, tile_rows, small_max, idx, (half_t *)wdev, epos, tiles, singles, smalls, counts);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int moe_gather(const void *src, const int *idx, const int *counts, int max_entries, int dim, void *dst, hipStream_t s)
{
    if (max_entries <= 0) return IFA_OK;
    k_moe_gather<<<dim3(4, (unsigned)max_entries), dim3(256), 0, s>>>((const half_t *)src, idx, counts, dim, (half_t *)dst);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

int moe_combine(const void *y, const int *epos, const void *wsel, int T, int top_k, int dim, void *out, hipStream_t s, const void *residual)
{
    k_moe_combine<<<dim3(ifa_cdiv((size_t)dim, 256), (unsigned)T), dim3(256), 0, s>>>((const half_t *)y, epos, (const half_t *)wsel, T, top_k, dim, (half_t *)out,
                                                                                     (const half_t *)residual);
    IFA_LAUNCH_CHECK();
    return IFA_OK;
}

} // namespace ifa
