"""Synthetic models for measurement and parity tests:
fp16 source tensors N(0, 0.02) from seed = 1000 + layer*16 + tensor_id, norm
weights = 1, quantised on the device with the reference rule; tensors below
tensor_quant_threshold stay F16 (src/transformer/network_builder.cc:1557-1562),
lm_head stays F16 (network_builder.cc:825-846)."""
import torch

from . import dtypes as dt
from . import worker as W

SHAPES = {
    # data/models/llama2_7b_chat_hf/model_spec.json + config: d=4096 L=32 H=32 ffn=11008 vocab=32000
    "llama2_7b": dict(dim=4096, layers=32, heads=32, kv_heads=32, head_dim=128, ffn=11008, vocab=32000),
    # bin/llm_inference.tiny.ini shape (stories15M)
    "tiny15m": dict(dim=288, layers=6, heads=6, kv_heads=6, head_dim=48, ffn=768, vocab=32000),
    "test_tiny": dict(dim=288, layers=2, heads=6, kv_heads=6, head_dim=48, ffn=768, vocab=1000),   # stories15M layers, small vocabulary
    # small GQA shapes for parity tests
    "test_gqa": dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000),
    "test_mha": dict(dim=256, layers=3, heads=8, kv_heads=8, head_dim=32, ffn=640, vocab=777),
    # w2 rows longer than one 4096-column chunk of the rows GEMM (chunk loop / whole-row staging)
    "test_longffn": dict(dim=256, layers=2, heads=4, kv_heads=4, head_dim=64, ffn=4352, vocab=500),
    # data/models/yi_34b_chat (configs[3]) and a Llama-2-70B-shaped model: w2 rows of 20480 / 28672 columns
    "yi_34b": dict(dim=7168, layers=60, heads=56, kv_heads=8, head_dim=128, ffn=20480, vocab=64000),
    "llama2_70b": dict(dim=8192, layers=80, heads=64, kv_heads=8, head_dim=128, ffn=28672, vocab=32000),
    # data/models/falcon_40b_instruct (configs[3]): LayerNorm, GELU, plain (non-gated) MLP, attention and MLP both read
    # the layer input through their own norm (mlp_attn_share_input), 128 heads of 64 over 8 KV heads
    "falcon_40b": dict(dim=8192, layers=60, heads=128, kv_heads=8, head_dim=64, ffn=32768, vocab=65024,
                       norm_kind=1, act_kind=1, is_glu=0, share_input=1, rope_order=2),
    # Mixtral-8x7B (data/models/mixtral_8x7b_instruct_v0.1) and a small MoE shape for parity tests
    "mixtral_8x7b": dict(dim=4096, layers=32, heads=32, kv_heads=8, head_dim=128, ffn=14336, vocab=32000, experts=8, moe_top_k=2),
    "test_moe": dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=512, vocab=1000, experts=4, moe_top_k=2),
    "test_moe_longffn": dict(dim=256, layers=2, heads=4, kv_heads=2, head_dim=64, ffn=4352, vocab=600, experts=4, moe_top_k=2),
    # a small Falcon-40B-like wiring (LayerNorm, GELU, plain MLP, shared MLP / attention input, grouped KV heads)
    "test_falcon": dict(dim=256, layers=2, heads=8, kv_heads=2, head_dim=32, ffn=512, vocab=1000,
                        norm_kind=1, act_kind=1, is_glu=0, share_input=1, rope_order=2),
}

MATRICES = [(W.T_WQ, "q"), (W.T_WK, "kv"), (W.T_WV, "kv"), (W.T_WO, "o"), (W.T_W1, "up"), (W.T_W3, "up"), (W.T_W2, "down")]
TENSOR_QUANT_THRESHOLD = 4000000   # ModelSpec::tensor_quant_threshold (src/transformer/model.h:137)


def _shape(kind, s):
    qd, kvd = s["heads"] * s["head_dim"], s["kv_heads"] * s["head_dim"]
    return {"q": (qd, s["dim"]), "kv": (kvd, s["dim"]), "o": (s["dim"], qd),
            "up": (s["ffn"], s["dim"]), "down": (s["dim"], s["ffn"])}[kind]


def gen_f16(shape, seed, std=0.02, device="cuda"):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return (torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std).half()


def build(shape_name, wdtype=dt.Q4_B32T1A, kv_dtype=dt.F16, max_ctx=1024, quant_threshold=TENSOR_QUANT_THRESHOLD,
          std=0.02, keep_host=False, device=0, embd_std=None, tied_lm_head=None, **overrides):
    """Returns (worker, host_tensors or None).  host_tensors: {(layer, tid): (dtype, np array, rows, cols)}.
    embd_std / tied_lm_head = (permutation seed, scale): the "peaky" model of the free-running parity test -- embedding rows of std
    embd_std (the token's own row then carries through the residual stream) and lm_head[v] = scale * embedding[perm[v]]: the logit
    of the one row aligned with the current token's embedding stands several logit-std above the rest, so greedy ids are well
    separated although every layer is random (next token = perm^-1(current): a walk along a cycle of the permutation)."""
    s = dict(SHAPES[shape_name])
    s.update({k: v for k, v in overrides.items() if k in s})
    extra = {k: v for k, v in overrides.items() if k not in s}
    wk = W.DecodeWorker(max_ctx=max_ctx, kv_dtype=kv_dtype, device=device, **s, **extra)
    host = {} if keep_host else None
    dev = "cuda:%d" % device

    def put(layer, tid, target, t16, expert=-1):
        rows, cols = (1, t16.numel()) if t16.dim() == 1 else t16.shape
        wk.set_tensor_f16(layer, tid, target, t16, rows, cols, expert=expert)
        if keep_host:
            key = (layer, tid) if expert < 0 else (layer, tid, expert)
            host[key] = (target, t16.cpu().view(torch.int16).numpy().view("float16").copy(), rows, cols)

    embd = gen_f16((s["vocab"], s["dim"]), 999, std if embd_std is None else embd_std, dev)
    put(-1, W.T_EMBD, dt.F16, embd)
    put(-1, W.T_OUT_NORM, dt.F16, torch.ones(s["dim"], dtype=torch.float16, device=dev))
    if tied_lm_head is None:
        put(-1, W.T_LM_HEAD, dt.F16, gen_f16((s["vocab"], s["dim"]), 998, std, dev))
    else:
        pg = torch.Generator(device="cpu"); pg.manual_seed(int(tied_lm_head[0]))
        perm = torch.randperm(s["vocab"], generator=pg).to(dev)
        put(-1, W.T_LM_HEAD, dt.F16, (embd[perm].float() * float(tied_lm_head[1])).half())
    del embd
    for layer in range(s["layers"]):
        put(layer, W.T_ATTN_NORM, dt.F16, torch.ones(s["dim"], dtype=torch.float16, device=dev))
        put(layer, W.T_FFN_NORM, dt.F16, torch.ones(s["dim"], dtype=torch.float16, device=dev))
        n_exp = s.get("experts", 0)
        for tid, kind in MATRICES:
            if tid == W.T_W3 and not s.get("is_glu", 1):
                continue
            rows, cols = _shape(kind, s)
            target = wdtype if rows * cols >= quant_threshold else dt.F16
            if n_exp and tid in (W.T_W1, W.T_W2, W.T_W3):       # one FFN per expert (ProcessGpuLayer_Moe)
                for e in range(n_exp):
                    put(layer, tid, target, gen_f16((rows, cols), 100000 + (layer * 64 + e) * 16 + tid, std, dev), expert=e)
                continue
            t16 = gen_f16((rows, cols), 1000 + layer * 16 + tid, std, dev)
            put(layer, tid, target, t16)
        if n_exp:   # router: always F16 (far below tensor_quant_threshold)
            put(layer, W.T_MOE_GATE, dt.F16, gen_f16((n_exp, s["dim"]), 1000 + layer * 16 + W.T_MOE_GATE, std, dev))
    wk.finalize()
    return wk, host, s


def weight_bytes(shape_name, wdtype=dt.Q4_B32T1A, quant_threshold=TENSOR_QUANT_THRESHOLD, streamed=False):
    """Algorithmic weight bytes per decoded token, excluding KV.  streamed=True: the bytes of the layout
    the decode kernels read (differs from the reference block bytes for Q3H_B64T1 only: 36 vs 32 per 64 weights)."""
    s = SHAPES[shape_name]
    per_layer = 0
    for tid, kind in MATRICES:
        if tid == W.T_W3 and not s.get("is_glu", 1):
            continue
        rows, cols = _shape(kind, s)
        d = wdtype if rows * cols >= quant_threshold else dt.F16
        n = rows * (dt.streamed_row_bytes(d, cols) if streamed else dt.row_bytes(d, cols))
        if s.get("experts", 0) and tid in (W.T_W1, W.T_W2, W.T_W3):
            n *= s["moe_top_k"]                  # a token streams its top-k experts only
        per_layer += n
    if s.get("experts", 0):
        per_layer += s["experts"] * s["dim"] * 2     # F16 router
    lm_head = s["vocab"] * s["dim"] * 2
    return per_layer * s["layers"] + lm_head


def kv_bytes_per_ctx_row(shape_name, kv_dtype=dt.F16):
    s = SHAPES[shape_name]
    return s["layers"] * 2 * dt.row_bytes(kv_dtype, s["kv_heads"] * s["head_dim"])
