/*
 * inferflow_amd.h -- C ABI of the MI355X-native quantized decode path.
 *
 * This is the drop-in boundary for the lower surface that Inferflow's
 * GpuInferenceWorker calls:
 *   TensorOpr::*      src/tensor/tensor_opr.h:33-131
 *   TensorMul::*      src/tensor/tensor_mul.h:14-57
 *   CublasEngine      src/tensor/cublas_engine.h:31-33
 *   LayerKVCache      src/transformer/kv_cache.h:13-34
 *   CudaUtil          src/common/cuda_util.h:40-61
 * and, one level up, for the per-step work of GpuInferenceWorker::Run
 * (src/transformer/inference_worker.cc:234-340) behind InferenceEngine
 * (src/transformer/inference_engine.h:32-129).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - Every function returns 0 on success or a negative IFA_ERR_* code;
 *    ifa_last_error() returns the message of the last failure on this thread
 *    (the reference ops return false + LogError; never throw).
 *  - All data pointers are DEVICE pointers unless the name says host.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  No op
 *    synchronises; the reference's "synchronous on return" behaviour is
 *    obtained by calling ifa_stream_sync() after the op.
 *  - dtype ids are the reference's ElementType enum values
 *    (src/tensor/tensor_common.h:15-42).
 *  - Tensors are row-major with ne0 (columns) contiguous, like DeviceTensor
 *    (src/tensor/device_tensor.h:85-153).  Quantized rows hold
 *    cols/capacity blocks in the reference's byte layout
 *    (src/common/quant_types.h:11-174).
 */
#ifndef INFERFLOW_AMD_H_
#define INFERFLOW_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    IFA_F32 = 0, IFA_F16 = 1,
    IFA_Q8_B32T1 = 7, IFA_Q8_B32T2 = 8, IFA_Q6_B64T1 = 9, IFA_Q5_B64T1 = 10, IFA_Q5_B32T1 = 11,
    IFA_Q4_B16 = 12, IFA_Q4_B32T1A = 13, IFA_Q4_B32T1B = 14, IFA_Q4_B64T1 = 17, IFA_Q3H_B64T1 = 18,
    IFA_Q3_B32T1A = 19, IFA_Q3_B32T1B = 20, IFA_Q2_B32T1A = 21, IFA_Q2_B32T1B = 22
} ifa_dtype;

enum {
    IFA_OK = 0,
    IFA_ERR_ARG = -1,      /* bad shape / null pointer / misaligned size */
    IFA_ERR_DTYPE = -2,    /* dtype not supported by this op */
    IFA_ERR_HIP = -3,      /* HIP runtime error */
    IFA_ERR_STATE = -4,    /* object not initialised / wrong phase */
    IFA_ERR_NOMEM = -5
};

typedef void *ifa_stream;

/* ---- library ------------------------------------------------------------ */
const char *ifa_version(void);
const char *ifa_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int ifa_device_count(void);

/* ---- CudaUtil counterparts (src/common/cuda_util.h:40-61) --------------- */
int ifa_set_device(int device);
int ifa_malloc(void **dptr, size_t bytes);
int ifa_free(void *dptr);
int ifa_memcpy_h2d(void *dst, const void *src_host, size_t bytes, ifa_stream stream);
int ifa_memcpy_d2h(void *dst_host, const void *src, size_t bytes, ifa_stream stream);
int ifa_memcpy_d2d(void *dst, const void *src, size_t bytes, ifa_stream stream);
int ifa_memset(void *dst, int value, size_t bytes, ifa_stream stream);
int ifa_stream_create(ifa_stream *out);
int ifa_stream_destroy(ifa_stream s);
int ifa_stream_sync(ifa_stream s);

/* ---- type registry (src/tensor/tensor_common.cc:6-234) ------------------ */
int ifa_block_capacity(int dtype);               /* 0 if unknown */
int ifa_block_bytes(int dtype);
size_t ifa_row_bytes(int dtype, size_t cols);    /* TensorCommon::ByteCount per row */
/* "q4" -> IFA_Q4_B32T1A etc.; returns -1 if unknown (InitElementTypeMap :171-205) */
int ifa_dtype_from_name(const char *name);

/* ---- TensorOpr::Quantize / Dequantize (src/tensor/tensor_opr.cu:1623, :2229) */
/* src F16 [rows][cols] -> packed reference-layout blocks.  Q8_B32T2 uses the
 * device alg-2 quantizer (src/kernels/tensor_quant.h:44-82) like the reference. */
int ifa_quantize(int dtype, const void *src_f16, size_t rows, size_t cols, void *dst, ifa_stream stream);
int ifa_quantize_f32(int dtype, const void *src_f32, size_t rows, size_t cols, void *dst, ifa_stream stream);
int ifa_dequantize(int dtype, const void *src, size_t rows, size_t cols, void *dst_f16, ifa_stream stream);
/* activation / KV-row quantizer: F16 [rows][cols] -> Q8_B32T2, cols may be ragged */
int ifa_quantize_act_q8(const void *src_f16, size_t rows, size_t cols, void *dst, ifa_stream stream);

/* ---- TensorMul::Gemv_AX (src/tensor/tensor_mul.h:27, tensor_mul.cu:766-846) */
/* y[rows] (F16) = W[rows][cols] . x (+bias).  x_dtype is IFA_Q8_B32T2 (the
 * int8 x intN path, gemv.h:1499-1709; W must be one of the 7 eligible types)
 * or IFA_F16 (gemv.h:469-1497; any W type incl. IFA_F16).  bias may be NULL. */
int ifa_gemv(int w_dtype, const void *W, size_t rows, size_t cols,
             int x_dtype, const void *x, const void *bias_f16, void *y_f16, ifa_stream stream);

/* Re-tile reference-layout rows into the row-local plane layout the fused
 * decode kernels stream (DESIGN.md "HBM layout"); same byte count per row.
 * Supported for the AX8-eligible types; src and dst must not alias. */
int ifa_repack_weights(int dtype, const void *src, size_t rows, size_t cols, void *dst, ifa_stream stream);
/* same GEMV as ifa_gemv(x_dtype = Q8) reading the re-tiled layout */
int ifa_gemv_tiled(int w_dtype, const void *Wt, size_t rows, size_t cols,
                   const void *x_q8, const void *bias_f16, void *y_f16, ifa_stream stream);

/* ---- TensorOpr::LayerNormalization (tensor_opr.cu:458-602) --------------- */
/* kind 0 = RMS (x*rsqrt(mean(x^2)+eps)*(multi_base+w) + b), 1 = STD.  w,b may be NULL. */
int ifa_layernorm(int kind, const void *x_f16, size_t rows, size_t cols, const void *w_f16,
                  const void *b_f16, float multi_base, float eps, void *y_f16, ifa_stream stream);

/* ---- TensorOpr::PositionEmbedding (tensor_opr.cu:693-806) --------------- */
/* x F16 [tokens][heads][head_dim] in place; order 1 = adjacent pairs
 * (PosEmbedding_Rope_Std_Kernel), 2 = (c, c+rope_cols/2) (Rope_Order2). */
int ifa_rope(void *x_f16, int head_dim, int heads, int tokens, int pos0, float theta,
             int order, float partial_rotary_factor, ifa_stream stream);
/* scores F16 [heads][q_tokens][ctx] += col*m_head (PosEmbedding_Alibi_Std_Kernel) */
int ifa_alibi(void *scores_f16, int ctx, int q_tokens, int heads, int base_head, int total_heads,
              ifa_stream stream);

/* ---- TensorOpr::SoftMax (tensor_opr.cu:1189-1225) ------------------------ */
/* s F16 [cz][cy][cx] in place; element xi of row r is masked (-inf) iff
 * prefix_len >= 0 && xi > prefix_len + r; values are multiplied by scale first. */
int ifa_softmax(void *s_f16, int cx, int cy, int cz, int prefix_len, float scale, ifa_stream stream);

/* ---- TensorOpr::Activation / Mul / Add / Scale --------------------------- */
/* kind 0 silu, 1 gelu(tanh), 2 relu; is_glu: input rows are [2*cols], out = act(a)*b */
int ifa_activation(int kind, int is_glu, const void *x_f16, size_t rows, size_t cols, void *y_f16,
                   ifa_stream stream);
int ifa_mul(const void *a_f16, const void *b_f16, size_t n, void *c_f16, ifa_stream stream);
/* c = a + b, b broadcast with period b_period elements (0 = same size) */
int ifa_add(const void *a_f16, const void *b_f16, size_t n, size_t b_period, void *c_f16, ifa_stream stream);
int ifa_scale(const void *a_f16, float s, size_t n, void *c_f16, ifa_stream stream);

/* ---- attention over a KV cache (inference_worker.cc:983-1405, :1639-1724) */
/* q F16 [q_tokens][heads][head_dim]; caches [n_ctx rows][kv_heads*head_dim] in
 * IFA_F16 or IFA_Q8_B32T2 (LayerKVCache, kv_cache.cc:104-249); causal mask with
 * prefix_len; GQA by indexing (no RepeatKV copy); out F16 [q_tokens][heads*head_dim].
 * alibi: 0/1; alibi_base_head/total_heads as in ifa_alibi. */
int ifa_attention(const void *q_f16, const void *kcache, const void *vcache, int kv_dtype,
                  int n_ctx, int q_tokens, int prefix_len, int heads, int kv_heads, int head_dim,
                  float kq_scale, int alibi, int alibi_base_head, int alibi_total_heads,
                  void *out_f16, ifa_stream stream);

/* ---- greedy argmax over F16 logits (SampleTokens top-1) ------------------ */
int ifa_argmax(const void *logits_f16, size_t n, int *out_index_dev, ifa_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* INFERFLOW_AMD_H_ */
