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

/* ---- prefill / batched linear layer (MatrixMultiplication's T>1 branch,
 * src/transformer/inference_worker.cc:2374-2415: TensorOpr::Dequantize + CublasEngine::GemmEx
 * F16xF16->F16 with fp32 accumulate + Transpose).  Y[tokens][rows] (F16) = X[tokens][cols] (F16)
 * . W[rows][cols]^T (+bias); W in any block format (reference layout) or F16.  No F16 copy of W exists: up to 128 tokens
 * every lane dequantises the blocks it needs into its MFMA operand registers (split-K kernels, weight-stream bound);
 * above 128 tokens (formats with blocks of <= 32 values) a workgroup dequantises its weight tile once per K step into
 * LDS and 256 x 256 / 128 x 256 / 128 x 128 output tiles are multiplied from there (v_mfma_f32_32x32x16_f16).
 * Every T runs these in-tree kernels: the library neither links nor loads a vendor GEMM. */
int ifa_gemm(int w_dtype, const void *W, size_t rows, size_t cols, const void *x_f16, size_t tokens,
             const void *bias_f16, void *y_f16, ifa_stream stream);
/* the large-tile kernel for tokens > 128 (default 1; 0: the smaller-tile kernels serve every T).  The tile kernels add the same
 * products in DIFFERENT orders (and the split-K form of the 128 x 128 tiles adds its two halves of K as first + second): results
 * agree within the F16 rounding of one product row, they are not bit-identical.  < 0 only queries; returns the previous setting */
int ifa_gemm_big_tiles(int on);

/* ---- launches that wait for sibling workgroups inside the launch (the fused QKV + attention step, the split-K halves of the
 * large-tile GEMM, the K parts of the 9..32-row GEMM).  They are chosen only when the whole waiting grid can be resident at once:
 * occupancy of the kernel x the CUs this process may use (ROC_GLOBAL_CU_MASK / HSA_CU_MASK are read; IFA_VISIBLE_CUS overrides).
 * Every wait is bounded (0.2 s); one that gives up leaves a code, the synchronising call (ifa_stream_sync, ifa_model_forward /
 * _decode / _decode_batch) returns IFA_ERR_STATE, and the waiting launches stay off for the rest of the process -- the HIP context
 * survives, the non-waiting kernels serve from there on.  IFA_NO_INLAUNCH_WAITS=1 (environment) starts the process that way;
 * per model: ifa_model_set_option "rows_kparts" / "gemm_splitk" / "fuse_attn" = 0.
 * host-only helpers (no device needed): the residency rule and the CU-mask reader behind that choice. */
int ifa_wait_grid_decision(int blocks_per_cu, int visible_cus, long long grid);      /* 1: the grid fits */
int ifa_visible_cus_from_mask(const char *mask_text, int device, int device_cus);    /* CUs the mask leaves for `device` */
int ifa_inlaunch_waits_enabled(void);
/* frees the scratch the prefill kernels keep for this stream on the current device (call before destroying a stream that ran
 * long-prompt attention; ifa_model_destroy / ifa_model_set_stream do it for the worker's) */
int ifa_gemm_release_stream(ifa_stream stream);

/* Re-tile reference-layout rows into the row-local plane layout the fused
 * decode kernels stream (DESIGN.md "HBM layout"): same bytes per row, row stride
 * padded to a multiple of 16 (ifa_tiled_row_bytes; 0 for types without a tiled
 * layout).  Supported for the AX8-eligible types; src and dst must not alias;
 * dst holds rows * ifa_tiled_row_bytes(dtype, cols) bytes. */
size_t ifa_tiled_row_bytes(int dtype, size_t cols);
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
/* TensorOpr::AddByRowIndex (tensor_opr.cu:1519-1548, kernel binary_tensor_opr.h:80-125), the MoE scatter-add:
 * B[row_idx[r]][:] = hfma(A[r][:], weights[r], B[row_idx[r]][:]) in half precision (weights may be NULL = 1). */
int ifa_add_by_row_index(void *b_f16, const void *a_f16, size_t rows, size_t cols, const int *row_idx_dev,
                         const void *weights_f16_dev, ifa_stream stream);

/* ---- attention over a KV cache (inference_worker.cc:983-1405, :1639-1724) */
/* q F16 [q_tokens][heads][head_dim]; caches [n_ctx rows][kv_heads*head_dim] in
 * IFA_F16 or IFA_Q8_B32T2 (LayerKVCache, kv_cache.cc:104-249); causal mask with
 * prefix_len; GQA by indexing (no RepeatKV copy); out F16 [q_tokens][heads*head_dim].
 * alibi: 0/1; alibi_base_head/total_heads as in ifa_alibi. */
int ifa_attention(const void *q_f16, const void *kcache, const void *vcache, int kv_dtype,
                  int n_ctx, int q_tokens, int prefix_len, int heads, int kv_heads, int head_dim,
                  float kq_scale, int alibi, int alibi_base_head, int alibi_total_heads,
                  void *out_f16, ifa_stream stream);
/* ---- parity instruments: single-row ops in the SUMMATION ORDER of the reference's CUDA kernels (csrc/ifa_exact.hip) -- the 32-lane walk +
 * xor butterfly of Gemv_AX8_* (src/kernels/gemv.h:1499-1709; F16 activations: the serial fp32 order of the oracle's restatement of :469-1497),
 * Tensor_RmsNorm_Kernel's 128 chunks (unary_tensor_opr.h:216-289), serial fp32 attention dots + the 32-lane softmax (gemm.h:83-178,
 * unary_tensor_opr.h:480-535), SiLU / ReLU (+ gate) with libm's expf.  Results equal the CPU oracle's bit for bit; not timed kernels.
 * Same argument meaning as ifa_layernorm / ifa_gemv / ifa_attention (one query row, no ALiBi) / ifa_activation_mul (gate may be NULL). */
int ifa_exact_rmsnorm(const void *x_f16, size_t rows, size_t cols, const void *w_f16, const void *b_f16, float multi_base, float eps,
                      void *y_f16, ifa_stream stream);
int ifa_exact_gemv(int w_dtype, const void *W, size_t rows, size_t cols, int x_dtype, const void *x, const void *bias_f16, void *y_f16,
                   ifa_stream stream);
int ifa_exact_attention(const void *q_f16, const void *kcache, const void *vcache, int kv_dtype, int n_ctx, int heads, int kv_heads,
                        int head_dim, float kq_scale, void *out_f16, ifa_stream stream);
int ifa_exact_activation_mul(int kind, const void *a_f16, const void *gate_f16, size_t n, void *y_f16, ifa_stream stream);
/* chunks of at least min_tokens queries (default 64; 0 = never, < 0 only queries) take the two-pass kernels: K / V blocks staged
 * once per query tile, scores recomputed in the second pass instead of a [queries x keys] tile; head_dim 64 / 128; same rounding
 * points as the staged 32-query kernel.  Default variant: 64 queries per workgroup, the key blocks split over wave pairs;
 * on contexts of at least min_keys keys (default: never) chunks of >= 128 queries take the 128-query variant instead.
 * Each returns its previous threshold. */
int ifa_attention_two_pass_min(int min_tokens);
int ifa_attention_two_pass_min_keys(int min_keys);

/* ---- greedy argmax over F16 logits (SampleTokens top-1) ------------------ */
/* MoE routing of `tokens` softmaxed router rows [tokens][experts] on the device -- the per-row part of
 * HostTensorOpr::BuildRowsForMoE (src/tensor/host_tensor_opr.cc:190-244): top_k by repeated first maximum, probabilities
 * below 1e-5 dropped, optional renormalisation over the kept ones.  sel_out [tokens][top_k]: expert ids in ascending
 * order (-1 = unused slot), weights_out [tokens][top_k] F16.  (The reference copies the probabilities to the host.) */
int ifa_moe_route_topk(const void *probs_f16, size_t tokens, int experts, int top_k, int norm_top_k_prob, int *sel_out_dev,
                       void *weights_out_f16_dev, ifa_stream stream);
/* LayerKVCache::SetKRows / SetVRows (src/transformer/kv_cache.cc:159-249): `tokens` F16 rows [kv_dim] into rows
 * [first_row, first_row + tokens) of a K or V cache of dtype F16 (copy) or Q8_B32T2 (the Alg2 quantiser) */
int ifa_kv_store(int kv_dtype, const void *rows_f16, size_t tokens, size_t kv_dim, void *cache, size_t first_row, ifa_stream stream);
int ifa_argmax(const void *logits_f16, size_t n, int *out_index_dev, ifa_stream stream);
/* the same over the ALLOWED ids: excluded_dev = {count (<= 3), id, id, id} in device memory (nullable) -- the ids
 * SamplingStrategy::GetSortedTopK never offers to its queue: the vocabulary's unk id and Invalid-type tokens
 * (src/transformer/sampling_strategy.cc:281-297) */
int ifa_argmax_masked(const void *logits_f16, size_t n, const int *excluded_dev, int *out_index_dev, ifa_stream stream);

/* ======================================================================== */
/* Per-device decode worker: counterpart of GpuInferenceWorker                */
/* (src/transformer/inference_worker.h:23-62, inference_worker.cc:234-340)    */
/* for the hyper-parameters of ModelSpec (src/transformer/model.h:72-151).    */
/* ======================================================================== */
typedef struct ifa_model ifa_model;

typedef struct {
    int dim, layers, heads, kv_heads, head_dim, ffn, vocab, max_ctx;
    int norm_kind;       /* 0 rms, 1 std                      (normalization_function) */
    int act_kind;        /* 0 silu, 1 gelu, 2 relu            (activation_function)    */
    int is_glu;          /* informational: w3 present                                   */
    int rope_order;      /* 0 none, 1 adjacent pairs, 2 half-split (qk_column_order)    */
    int use_alibi;       /* position_embedding == alibi                                 */
    int parallel_attn;   /* is_parallel_attn                                            */
    int share_input;     /* mlp_attn_share_input                                        */
    float rope_theta, partial_rotary, kq_scale, eps;
    int kv_dtype;        /* IFA_F16 or IFA_Q8_B32T2 (device_kv_cache_data_type)         */
    int full_quant_gemv; /* enable_full_quant_gemv (inference_engine.cc:57)             */
    int experts, moe_top_k, moe_norm_topk;
    int tp_rank, tp_size;/* tensor-parallel shard of this worker (heads/kv_heads/ffn are per-shard) */
    int device;          /* HIP device ordinal */
    /* RMS weight = base + w (attn_pre_norm_base / ffn_pre_norm_base / output_norm_base: Gemma) and TensorOpr::Scale of
     * the attention output, the FFN output and the last layer's output (attn_out_scale / ffn_out_scale / out_scale:
     * MiniCPM; inference_worker.cc:568-570, 842-843, 928-929).  Scales <= 0 mean 1. */
    float attn_norm_base, ffn_norm_base, out_norm_base, attn_out_scale, ffn_out_scale, out_scale;
    /* TensorOpr::LinearNorm on the decoder input (has_embedding_linear_norm / embedding_linear_scale: Gemma, MiniCPM;
     * inference_worker.cc:447-451, tensor_opr.cu:482-497): every embedding row is multiplied by this scale;
     * 0 = absent, < 0 = the reference's default sqrt(dim). */
    float embd_scale;
} ifa_model_config;

/* tensor ids for ifa_model_set_tensor (StdDeviceNetwork, src/transformer/model.h:168-276) */
enum {
    IFA_T_EMBD = 0, IFA_T_OUT_NORM = 1, IFA_T_OUT_NORM_B = 2, IFA_T_LM_HEAD = 3,
    IFA_T_ATTN_NORM = 10, IFA_T_ATTN_NORM_B = 11, IFA_T_WQ = 12, IFA_T_WK = 13, IFA_T_WV = 14, IFA_T_WO = 15,
    IFA_T_FFN_NORM = 16, IFA_T_FFN_NORM_B = 17, IFA_T_W1 = 18, IFA_T_W2 = 19, IFA_T_W3 = 20, IFA_T_MOE_GATE = 21,
    IFA_T_WQ_B = 22, IFA_T_WK_B = 23, IFA_T_WV_B = 24, IFA_T_WO_B = 25, IFA_T_W1_B = 26, IFA_T_W2_B = 27, IFA_T_W3_B = 28,
    /* self_attn.post_norm / feed_forward.post_norm (+ biases): StdDeviceNetwork::AttentionLayer::post_norm, FeedForwardLayer::post_norm
     * (src/transformer/model.h:168-276); applied by ProcessGpuLayer (inference_worker.cc:857-866, 954-965).  Models that carry them take
     * the op-by-op layer (the fused decode / prompt launches decline).  ModelSpec::is_attn_post_as_residual (model.h:113, default
     * true) is the model option "attn_post_as_residual". */
    IFA_T_ATTN_POST_NORM = 29, IFA_T_ATTN_POST_NORM_B = 30, IFA_T_FFN_POST_NORM = 31, IFA_T_FFN_POST_NORM_B = 32
};

int ifa_model_create(const ifa_model_config *cfg, ifa_model **out);
int ifa_model_destroy(ifa_model *m);
/* Copy a tensor that is already in its final dtype (reference block layout or F16)
 * from device memory into the worker; eligible weight matrices are also re-tiled. */
int ifa_model_set_tensor(ifa_model *m, int layer, int tensor_id, int expert, int dtype,
                         const void *dev_src, size_t rows, size_t cols);
/* F16 source quantised on the device to target_dtype first
 * (DeviceTensorBuilder::Build_Quant, src/tensor/device_tensor_builder.cu:383-420). */
int ifa_model_set_tensor_f16(ifa_model *m, int layer, int tensor_id, int expert, int target_dtype,
                             const void *dev_src_f16, size_t rows, size_t cols);
/* allocate KV caches (KVCache::Init, kv_cache.cc:278-319) and scratch */
int ifa_model_finalize(ifa_model *m);
int ifa_model_reset(ifa_model *m);
/* options: "fused" (1), "graph" (1), "rpw_qkv|rpw_wo|rpw_ffn|rpw_w2|rpw_lm" (0 = auto), "batch_fused" (1), "rows_mo" (1: the batched
 * step and short prompts stream MFMA-operand-order copies of the weights, built on first use), "attn_split_ctx" (-1 = by model: 512 keys, 640 with grouped queries, 320 without the fused launch; a decode call that reaches more keys than this runs the attention as scores / P.V / combine launches with a head's keys split over 8 or 16 workgroups; 0: never), "fuse_attn" (1: the attention as the
 * tail of the wq | wk | wv launch), "attn_unload" (1: in the 256-row bucket of that launch the heads' workgroups take no weight rows and request their cache rows at once; 2: in the 128-row bucket too, measured slower), "fuse_ffn" (0; 1 / 2: [Wo ->] W1 | W3 -> W2 as one chained launch, csrc/ifa_decode_chain.h: bit-identical,
 * measured slower than the separate launches on MI355X, kept as the measurement harness of that statement), "prefill_mid" (1),
 * "prefill_mid_max" (768), "prefill_res_mid" (2048: up to this many tokens wo / w2 keep the mid-size kernel above prefill_mid_max), "prefill_big_min" (47), "attn_post_as_residual" (1), "exact_order" (0; 1: every single-token step -- and every row
 * of a prompt, one by one -- runs in the summation order of the reference's CUDA kernels, csrc/ifa_exact.hip: a parity instrument whose
 * logits, ids and int8 codes equal the CPU oracle's bit for bit; fails for models outside that step instead of changing arithmetic),
 * "q3h_native" (0; 1: Q3H_B64T1 Wo / W1 / W3 / W2 streamed at 32 bytes per block, pair codes decoded in the kernel: bit-identical, measured slower),
 * "perf_stat" (0; 1: steps and prompts take the op-by-op layer with a HIP event pair around every phase, see ifa_model_perf_stat) */
/* Independent KV caches inside one worker, one per concurrent query -- the reference keeps a LayerKVCache set
 * per query processor (QueryStateTable, src/transformer/query_state_table.h:19-85; KVCache::Init, kv_cache.cc:278-319).
 * ifa_model_kv_slots grows the number of caches to n_slots (slot 0 exists after finalize); ifa_model_select_kv
 * makes one of them the cache that forward()/decode() read and write (each slot keeps its own captured graph). */
int ifa_model_kv_slots(ifa_model *m, int n_slots);
int ifa_model_select_kv(ifa_model *m, int slot);
int ifa_model_set_option(ifa_model *m, const char *name, int value);
/* up to 3 token ids the worker's greedy argmax (forward / decode / decode_batch) never selects: the unk id and
 * Invalid-type tokens GetSortedTopK skips (sampling_strategy.cc:281-297).  None by default at this level; the engine
 * facade sets the model's unk id. */
int ifa_model_set_excluded_tokens(ifa_model *m, const int *ids_host, int n);
/* Per-phase times in the key space of the reference's InferencePerfStat (GpuInferenceWorker::UpdatePerfStat, inference_worker.cc:2670-2697;
 * keys: (layer + 1) * 10000 + phase -- + 0 the whole layer for layers 0..5 (:318-322); for layer 0, the reference's layer_idx_for_study_:
 * + 10 attention pre-norm, + 30 q / k / v products, + 50 RoPE + cache rows, + 60 scores / softmax / V product, + 90 wo, + 300 the attention
 * part as a whole, + 710 FFN pre-norm, + 730 w1, + 750 w3, + 760 activation * gate, + 780 w2, + 700 the FFN as a whole, + 800 what follows it;
 * 1000009 the output stage (:673-675), 1 the embedding rows (inference_engine.cc:1168-1176)).  With option "perf_stat" = 1 every
 * ifa_model_forward / ifa_model_decode step ADDS the device milliseconds of its spans to the keys (the reference adds host-side launch times);
 * this call drains the stream, copies up to cap (key, ms) pairs in ascending key order, stores the number of keys there are in *n_out and
 * clears the map if clear != 0. */
int ifa_model_perf_stat(ifa_model *m, int *keys_out, float *ms_out, int cap, int *n_out, int clear);
/* 1 if the fused batch-1 decode kernels cover this model, else 0 (+ reason) */
int ifa_model_fused_supported(ifa_model *m, char *why, size_t why_len);
/* One Infer() step for one query: n_tokens new tokens at positions
 * [prefix_len, prefix_len+n_tokens); op-by-op; synchronous.  logits_out_dev (F16
 * [n_tokens][vocab], nullable) mirrors return_output_tensors; *next_token_host is
 * the greedy argmax of the last row. */
int ifa_model_forward(ifa_model *m, const int *tokens_host, int n_tokens, int prefix_len,
                      void *logits_out_dev, int *next_token_host);
/* Greedy batch-1 decode of n_steps tokens starting from first_token at start_pos
 * (its KV rows [0,start_pos) must already be cached).  Fused kernels, one graph
 * replay per token, token fed back on the device.  out_tokens_host[n_steps]
 * receives the generated ids; *elapsed_ms (nullable) the HIP-event time of the
 * n_steps replays on the worker's stream. */
int ifa_model_decode(ifa_model *m, int first_token, int start_pos, int n_steps,
                     int *out_tokens_host, float *elapsed_ms);
/* Everything a decode call of n_steps from start_pos sets up before its first launch (the attention variant of the contexts it
 * reaches, the hand-off arenas, the captured step and its multi-step replay) WITHOUT running a step: a caller that times its
 * first call keeps graph capture / instantiation out of it.  The KV cache and the activations are not touched. */
int ifa_model_decode_prepare(ifa_model *m, int start_pos, int n_steps);
/* Dynamic batching: ONE new token for each of n queries in one step (QueryStateTable + Infer_Std over several queries,
 * src/transformer/inference_engine.cc:1054-1220).  Row r is token tokens[r] at position positions[r] of the query whose
 * KV cache is slot kv_slots[r] (ifa_model_kv_slots; slots must be distinct).  The linear layers run once over the n rows
 * (the weights are streamed once for all queries), attention per row on its own cache.  logits_out: optional [n][vocab] F16. */
int ifa_model_decode_batch(ifa_model *m, int n, const int *tokens_host, const int *positions_host, const int *kv_slots_host,
                           int *next_tokens_host, void *logits_out_dev);
/* debugging taps: "logits", "hidden", "kcache", "vcache" (device pointers) */
int ifa_model_get_buffer(ifa_model *m, const char *name, int layer, void **dptr, size_t *bytes);
void *ifa_model_stream(ifa_model *m);
/* run the worker on a caller-owned stream (e.g. the one the caller's RCCL collectives are ordered on) */
int ifa_model_set_stream(ifa_model *m, ifa_stream stream);

/* ======================================================================== */
/* Collectives of the multi-GPU partitions (RCCL over xGMI; csrc/ifa_comm.hip) */
/* -- what GpuInferenceWorker::DistributeAndMergeTensors / MergeTensors /     */
/* DeviceCopy (src/transformer/inference_worker.cc:2148-2335) and             */
/* GpuInfGlobalData (src/transformer/gpu_inf_global_data.cu:25-199) do with   */
/* host-side spin-waits and serial copies.  Enqueue-only, explicit stream,    */
/* capturable; one communicator per (rank, group).                            */
/* ======================================================================== */
typedef struct ifa_comm ifa_comm;
#define IFA_COMM_ID_BYTES 128
/* one process per GPU: rank 0 makes the id, ships the 128 bytes over any host channel, every rank joins */
int ifa_comm_unique_id(void *id_out_128);
int ifa_comm_init_rank(const void *id_128, int nranks, int rank, int device, ifa_comm **out);
/* one process, one host thread per GPU (the engine facade, like inference_engine.cc:1203-1206): all ranks at once */
int ifa_comm_init_all(const int *device_ids, int n, ifa_comm **comms_out);
/* a device list that names ONE device n times makes an in-process loopback group instead (RCCL refuses two ranks on a
 * device): kernels / copies on that device with a host rendezvous between the rank threads -- host-synchronous, not
 * capturable (ifa_comm_capturable() == 0); for exercising the multi-rank paths on a 1-GPU box */
int ifa_comm_capturable(const ifa_comm *c);
/* identity of the communicator OBJECT (a new one at the same address gets a new serial): what a cached, captured step
 * is keyed on */
unsigned long long ifa_comm_serial(const ifa_comm *c);
/* One-shot all-reduce (csrc/ifa_comm.hip): communicators made by ifa_comm_init_all whose devices can map each other's
 * memory exchange vectors of <= 64 KB directly (push into peer inboxes + epoch flags, sum in rank order in half: the
 * arithmetic of MergeTensors, inference_worker.cc:2197-2260) instead of a ring collective -- one launch per rank, no host
 * rendezvous.  ifa_comm_oneshot: 1 if this communicator does; ifa_comm_set_oneshot(c, 0) keeps RCCL for every size;
 * ifa_comm_status: non-zero if a wait inside a one-shot all-reduce of this rank gave up (2 s). */
int ifa_comm_oneshot(const ifa_comm *c);
/* One process per rank (ifa_comm_init_rank): export allocates this rank's inbox / flags and writes their two IPC handles
 * (IFA_ONESHOT_HANDLE_BYTES); the caller gathers all ranks' handles in rank order over the channel that carried the
 * communicator id; import maps the peers' buffers (hipIpcOpenMemHandle) -- ifa_comm_oneshot(c) is 1 afterwards. */
#define IFA_ONESHOT_HANDLE_BYTES 128
int ifa_comm_oneshot_export(ifa_comm *c, void *handle_out_128);
int ifa_comm_oneshot_import(ifa_comm *c, const void *handles_all_ranks);
int ifa_comm_set_oneshot(ifa_comm *c, int on);
int ifa_comm_status(ifa_comm *c);
/* Wake every rank blocked in (or later entering) a collective of this communicator with an error: called by the rank
 * that failed, or by whoever supervises the ranks (ncclCommAbort, inference_worker.cc has no counterpart: the reference
 * deadlocks its worker threads in this case).  The communicator can only be destroyed afterwards. */
int ifa_comm_abort(ifa_comm *c);
int ifa_comm_destroy(ifa_comm *c);
int ifa_comm_rank(const ifa_comm *c);
int ifa_comm_size(const ifa_comm *c);
/* one host thread issuing the calls of several ranks brackets them with group_start / group_end */
int ifa_comm_group_start(void);
int ifa_comm_group_end(void);
/* BY_TENSOR merge: element-wise sum of the ranks' partial [count] F16 vectors (in place allowed) */
int ifa_allreduce_sum_f16(ifa_comm *c, const void *send_f16, void *recv_f16, size_t count, ifa_stream stream);
/* recv = the ranks' `bytes_per_rank` blocks in rank order (distributed argmax over a vocabulary-sharded lm_head) */
int ifa_allgather(ifa_comm *c, const void *send, void *recv, size_t bytes_per_rank, ifa_stream stream);
int ifa_broadcast(ifa_comm *c, void *buf, size_t bytes, int root, ifa_stream stream);
/* BY_LAYER hand-over of the [T][dim] F16 layer output between device groups (DeviceCopy, :2300-2335) */
int ifa_send(ifa_comm *c, const void *buf, size_t bytes, int peer, ifa_stream stream);
int ifa_recv(ifa_comm *c, void *buf, size_t bytes, int peer, ifa_stream stream);

/* ---- tensor-parallel decode (BY_TENSOR partition, src/transformer/network_builder.cc:1594-1686):
 * the worker holds heads/tp_size heads, kv_heads/tp_size KV heads and ffn/tp_size FFN rows; the
 * caller sums the two partial [dim] F16 vectors per layer over the group exactly where the
 * reference calls DistributeAndMergeTensors (inference_worker.cc:1378-1391, :1882-1895).
 * All calls only enqueue work on the worker's stream.
 *   begin(token,pos)            token < 0 / pos < 0: keep the id / position already in the device state
 *   attn(l, partial)            norm + QKV + attention + Wo product (no bias/residual)
 *   post_attn(l, reduced)       + bias, + residual
 *   ffn(l, partial)             norm + W1/W3 + act + W2 product
 *   post_ffn(l, reduced)        + bias, + residual -> next layer input
 *   logits(shard_out)           final norm + this rank's vocabulary rows of lm_head
 *   set_token(dev_ptr)          next token id from device memory (after the distributed argmax); advances the
 *                               device-side position, so begin(-1,-1) ... set_token() is replayable as a hipGraph */
int ifa_model_tp_begin(ifa_model *m, int token, int pos);
/* BY_LAYER / HYBRID partition (MultiGpuStrategy, src/transformer/model.h:61-66; layer ranges per device group:
 * NetworkBuilder::SplitGpuLayers, network_builder.cc:2094-2118): a worker that holds a layer range only starts its
 * step from the previous group's output instead of an embedding row, and hands its last layer's output on.
 *   begin_hidden(x, pos)   layer input from device memory ([dim] F16); pos < 0 keeps the device-side position
 *   hidden(x_out)          copy of the current layer input/output vector (after the last local layer) */
int ifa_model_tp_begin_hidden(ifa_model *m, const void *x_f16, int pos);
int ifa_model_tp_hidden(ifa_model *m, void *x_out_f16);
int ifa_model_tp_attn(ifa_model *m, int layer, void *partial_out_f16);
int ifa_model_tp_post_attn(ifa_model *m, int layer, const void *reduced_f16);
int ifa_model_tp_ffn(ifa_model *m, int layer, void *partial_out_f16);
int ifa_model_tp_post_ffn(ifa_model *m, int layer, const void *reduced_f16);
int ifa_model_tp_logits(ifa_model *m, void *logits_shard_out_f16);
int ifa_model_tp_set_token(ifa_model *m, const int *token_dev);
/* The whole multi-GPU greedy decode from C: the segments above + the collectives of csrc/ifa_comm.hip on the worker's
 * stream, a distributed argmax over the vocabulary-sharded lm_head, token / position fed back in device memory; the step
 * is captured once as a hipGraph (tensor-parallel groups) and replayed per token.  Every rank calls it with the same
 * arguments (one host thread or process per GPU).  out_tokens_host[n_steps]: the generated ids on every rank. */
typedef struct {
    ifa_comm *tp;          /* this worker's tensor-parallel group (NULL / size 1: nothing to merge) */
    ifa_comm *world;       /* all ranks of the job: hand-over between layer groups + token broadcast (n_stages > 1) */
    int stage, n_stages;   /* BY_LAYER / HYBRID: this worker's device group, number of groups (1 = BY_TENSOR only) */
    int prev_rank, next_rank, token_src;   /* ranks in `world`; -1 = none */
    int vocab_offset;      /* first vocabulary row of this rank's lm_head shard */
    int force_collectives; /* issue the collectives even in a group of one (plumbing check on a 1-GPU box) */
} ifa_tp_topology;
int ifa_model_tp_decode(ifa_model *m, const ifa_tp_topology *topo, int first_token, int start_pos, int n_steps,
                        int *out_tokens_host, float *elapsed_ms);
/* One Infer() step of a query over the partition: n_tokens new tokens at [start_pos, start_pos + n_tokens) fed through
 * the partition: n_tokens > 1 as ONE T > 1 step (row-sliced MFMA GEMMs, [T][dim] merges after wo and w2, [T][dim]
 * hand-over between layer groups), a single token through the decode path; *next_token_host = greedy next token of the
 * last one (every rank).
 * logits_shard_out_dev (nullable; last device group): this rank's lm_head rows of every token, [n_tokens][rows] F16. */
int ifa_model_tp_prefill(ifa_model *m, const ifa_tp_topology *topo, const int *tokens_host, int n_tokens, int start_pos,
                         void *logits_shard_out_dev, int *next_token_host);
/* ifa_model_decode_batch over a tensor-parallel group (one new token for each of n queries; merges over [n][dim], one
 * distributed argmax per row); logits_shard_out_dev: optional [n][this rank's lm_head rows] F16 */
int ifa_model_tp_decode_batch(ifa_model *m, const ifa_tp_topology *topo, int n, const int *tokens_host, const int *positions_host,
                              const int *kv_slots_host, int *next_tokens_host, void *logits_shard_out_dev);
/* reference-layout copy of a loaded tensor (device pointer); returns 1 if the tensor is not set */
int ifa_model_get_tensor(ifa_model *m, int layer, int tensor_id, int *dtype, void **dptr, size_t *rows, size_t *cols);
/* the same for W1 / W2 / W3 of expert `expert` of a mixture-of-experts layer */
int ifa_model_get_expert_tensor(ifa_model *m, int layer, int expert, int tensor_id, int *dtype, void **dptr, size_t *rows, size_t *cols);
/* Average duration (HIP events on the worker's stream) of `iters` back-to-back
 * launches of one fused decode kernel, rotating over the layers' weights:
 * which = 0 qkv, 1 attention, 2 wo, 3 ffn w1/w3, 4 w2, 5 lm_head.  For bench.py's roofline. */
int ifa_model_time_kernel(ifa_model *m, int which, int iters, float *avg_us);

#ifdef __cplusplus
}
#endif
#endif /* INFERFLOW_AMD_H_ */
