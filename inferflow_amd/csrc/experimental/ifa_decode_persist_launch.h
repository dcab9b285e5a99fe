// ifa_decode_persist_launch.h -- host interface of the persistent decode layer kernel (ifa_decode_persist.h).
#pragma once
#include "ifa_host.h"
#include "ifa_decode_persist.h"

namespace ifa {

template <int DT> int dec_persist_launch_dt(int nja, int njb, int hd, int kvq8, const PsParams &P, int ncu, size_t smem, hipStream_t s);
template <int DT> bool dec_persist_has_dt(int nja, int njb, int hd);

// true if a kernel exists for this weight format and shape (blocks per lane = ceil(cols / capacity / 64))
bool dec_persist_has(int w_dtype, int nja, int njb, int hd);
// one launch over layers [P.layer_begin, P.layer_end); ncu workgroups (= every CU of the device), smem = ps_lds_bytes()
int dec_persist_launch(int w_dtype, int nja, int njb, int hd, int kvq8, const PsParams &P, int ncu, size_t smem, hipStream_t s);

} // namespace ifa
