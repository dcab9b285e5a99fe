// ifa_decode_gemv.h -- launcher of the fused decode GEMV (k_dec_gemv, ifa_decode_kernels.h).
// One translation unit per weight format instantiates the kernels (ifa_dgemv_*.hip) so the
// formats compile in parallel; ifa_dgemv_q4b32.hip also holds the dtype dispatcher.
#pragma once
#include "ifa_host.h"
#include "ifa_decode_kernels.h"

namespace ifa {

// blocks-per-lane limit of each format's instantiations (cols <= 64 * capacity * MAXNJ)
template <int DT> struct DecGemvLimits { static constexpr int MAXNJ = 4; };
template <> struct DecGemvLimits<Q4_B32T1A> { static constexpr int MAXNJ = 8; };
template <> struct DecGemvLimits<Q8_B32T2> { static constexpr int MAXNJ = 6; };

// true if the fused GEMV can stream a [rows][cols] tensor of this dtype
bool dec_gemv_supported(int w_dtype, size_t cols);
// same for matrices whose input is neither normalised nor gated (wo, w2): up to 4 register chunks / 32768 columns
bool dec_gemv_supported_long(int w_dtype, size_t cols);

// epi: DecEpilogue, norm: 0/1.  P.nblk / P.total_rows are filled in here.
int dec_gemv_launch(int w_dtype, int epi, int norm, const DecGemvParams &P, int wgs_per_cu, hipStream_t s, long long *trace);

// fp16-activation variant (ifa_dgemv_f16x.hip): F16 tensors and the block formats outside the int8 path, weights in the
// reference byte layout; cols % 8 == 0, cols <= 32768
bool dec_gemv_h_supported(int w_dtype, size_t cols);
int dec_gemv_h_launch(int w_dtype, int epi, int norm, const DecGemvParams &P, hipStream_t s);

template <int DT>
int dec_gemv_launch_dt(int epi, int norm, const DecGemvParams &P, int wgs_per_cu, hipStream_t s);

int dec_num_cus();

// Q3H_B64T1 reference-layout rows -> the native 32-byte streaming copy of ifa_decode_formats.h WRowQ3HN (ifa_dgemv_q3hn.hip)
int q3h_native_rows(const void *aos, size_t rows, size_t cols, void *dst, hipStream_t s);

} // namespace ifa
