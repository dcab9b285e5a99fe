"""DecodeWorker -- Python handle on the per-device decode worker of the C ABI
(ifa_model_*), the counterpart of the reference's GpuInferenceWorker
(src/transformer/inference_worker.h:23-62).  torch is used only to hold device
memory; all compute happens in libinferflow_amd.so."""
import ctypes as C

import numpy as np

from . import _capi, dtypes as dt
from ._capi import ModelConfig, check, lib

# tensor ids (include/inferflow_amd.h)
T_EMBD, T_OUT_NORM, T_OUT_NORM_B, T_LM_HEAD = 0, 1, 2, 3
T_ATTN_NORM, T_ATTN_NORM_B, T_WQ, T_WK, T_WV, T_WO = 10, 11, 12, 13, 14, 15
T_FFN_NORM, T_FFN_NORM_B, T_W1, T_W2, T_W3, T_MOE_GATE = 16, 17, 18, 19, 20, 21
T_WQ_B, T_WK_B, T_WV_B, T_WO_B, T_W1_B, T_W2_B, T_W3_B = 22, 23, 24, 25, 26, 27, 28
T_ATTN_POST_NORM, T_ATTN_POST_NORM_B, T_FFN_POST_NORM, T_FFN_POST_NORM_B = 29, 30, 31, 32      # self_attn.post_norm / feed_forward.post_norm

_DEFAULTS = dict(norm_kind=0, act_kind=0, is_glu=1, rope_order=2, use_alibi=0, parallel_attn=0, share_input=0,
                 rope_theta=10000.0, partial_rotary=1.0, kq_scale=1.0, eps=1e-5, kv_dtype=dt.F16,
                 full_quant_gemv=1, experts=0, moe_top_k=0, moe_norm_topk=1, tp_rank=0, tp_size=1, device=0,
                 attn_norm_base=0.0, ffn_norm_base=0.0, out_norm_base=0.0, attn_out_scale=1.0, ffn_out_scale=1.0, out_scale=1.0,
                 embd_scale=0.0)


class DecodeWorker:
    def __init__(self, **kw):
        cfg = ModelConfig()
        vals = dict(_DEFAULTS)
        vals.update(kw)
        for k, v in vals.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self._h = C.c_void_p()
        check(lib().ifa_model_create(C.byref(cfg), C.byref(self._h)))

    # ---- weights -----------------------------------------------------------
    def set_tensor(self, layer, tid, dtype, dev_tensor, rows, cols, expert=-1):
        """dev_tensor: torch cuda tensor holding reference-layout blocks (uint8) or F16 values.
        expert >= 0: the W1/W2/W3 of one MoE expert of that layer."""
        check(lib().ifa_model_set_tensor(self._h, layer, tid, expert, dtype, C.c_void_p(dev_tensor.data_ptr()), rows, cols))

    def set_tensor_f16(self, layer, tid, target_dtype, dev_f16, rows=None, cols=None, expert=-1):
        if rows is None:
            rows, cols = (1, dev_f16.numel()) if dev_f16.dim() == 1 else dev_f16.shape
        check(lib().ifa_model_set_tensor_f16(self._h, layer, tid, expert, target_dtype,
                                             C.c_void_p(dev_f16.data_ptr()), rows, cols))

    def finalize(self):
        check(lib().ifa_model_finalize(self._h))

    def reset(self):
        check(lib().ifa_model_reset(self._h))

    def set_option(self, name, value):
        check(lib().ifa_model_set_option(self._h, name.encode(), int(value)))

    def perf_stat(self, clear=True):
        """{key: ms} of option perf_stat in the reference's InferencePerfStat key space (include/inferflow_amd.h: ifa_model_perf_stat)"""
        keys, ms, n = np.zeros(256, np.int32), np.zeros(256, np.float32), C.c_int(0)
        check(lib().ifa_model_perf_stat(self._h, keys.ctypes.data_as(C.c_void_p), ms.ctypes.data_as(C.c_void_p), 256, C.byref(n), 1 if clear else 0))
        return {int(k): float(v) for k, v in zip(keys[:min(n.value, 256)], ms[:min(n.value, 256)])}

    def set_excluded_tokens(self, ids):
        """ids (<= 3) the greedy argmax never selects (unk / Invalid-type tokens, sampling_strategy.cc:281-297)"""
        a = np.ascontiguousarray(ids, np.int32)
        check(lib().ifa_model_set_excluded_tokens(self._h, a.ctypes.data_as(C.c_void_p), a.size))

    def fused_supported(self):
        buf = C.create_string_buffer(256)
        ok = lib().ifa_model_fused_supported(self._h, buf, 256)
        return bool(ok), buf.value.decode()

    # ---- inference -----------------------------------------------------------
    def forward(self, tokens, prefix_len, logits_out=None):
        """One Infer() step (op-by-op).  logits_out: optional torch cuda f16 [T][vocab]."""
        toks = np.ascontiguousarray(tokens, np.int32)
        nxt = C.c_int(0)
        check(lib().ifa_model_forward(self._h, toks.ctypes.data_as(C.c_void_p), toks.size, prefix_len,
                                      C.c_void_p(logits_out.data_ptr()) if logits_out is not None else None,
                                      C.byref(nxt)))
        return nxt.value

    def kv_slots(self, n):
        """Independent KV caches, one per concurrent query (slot 0 exists after finalize)."""
        check(lib().ifa_model_kv_slots(self._h, int(n)))

    def select_kv(self, slot):
        check(lib().ifa_model_select_kv(self._h, int(slot)))

    def decode_batch(self, tokens, positions, slots, logits_out=None):
        """One new token for each of n queries (dynamic batching); returns the n greedy next tokens."""
        toks = np.ascontiguousarray(tokens, np.int32)
        pos = np.ascontiguousarray(positions, np.int32)
        sl = np.ascontiguousarray(slots, np.int32)
        out = np.zeros(toks.size, np.int32)
        check(lib().ifa_model_decode_batch(self._h, toks.size, toks.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                                           sl.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                           C.c_void_p(logits_out.data_ptr()) if logits_out is not None else None))
        return out

    def decode(self, first_token, start_pos, n_steps, timed=True):
        """Greedy batch-1 decode with the fused kernels; returns (tokens, gpu_ms)."""
        out = np.zeros(n_steps, np.int32)
        ms = C.c_float(-1.0)
        check(lib().ifa_model_decode(self._h, int(first_token), int(start_pos), int(n_steps),
                                     out.ctypes.data_as(C.c_void_p), C.byref(ms) if timed else None))
        return out, ms.value

    def decode_prepare(self, start_pos, n_steps):
        """Set up (capture) what decode(., start_pos, n_steps) replays, without running a step."""
        check(lib().ifa_model_decode_prepare(self._h, int(start_pos), int(n_steps)))

    def time_kernel(self, which, iters=200):
        """avg microseconds per launch of one fused kernel (0 qkv, 1 attn, 2 wo, 3 ffn13, 4 w2, 5 lm_head)"""
        us = C.c_float(0)
        check(lib().ifa_model_time_kernel(self._h, which, iters, C.byref(us)))
        return us.value

    def get_tensor_host(self, layer, tid):
        """(dtype, uint8/uint16 numpy copy, rows, cols) of a loaded tensor in reference layout, or None."""
        d, p, r, c = C.c_int(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        rc = lib().ifa_model_get_tensor(self._h, layer, tid, C.byref(d), C.byref(p), C.byref(r), C.byref(c))
        if rc == 1:
            return None
        check(rc)
        nbytes = r.value * dt.row_bytes(d.value, c.value)
        out = np.empty(nbytes, np.uint8)
        check(lib().ifa_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, nbytes, None))
        check(lib().ifa_stream_sync(None))
        if d.value == dt.F16:
            out = out.view(np.uint16)
        return d.value, out, r.value, c.value

    def get_expert_tensor_host(self, layer, expert, tid):
        """(dtype, uint8 copy, rows, cols) of W1 / W2 / W3 of one expert of an MoE layer in reference layout, or None"""
        d, p, r, c = C.c_int(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        rc = lib().ifa_model_get_expert_tensor(self._h, layer, expert, tid, C.byref(d), C.byref(p), C.byref(r), C.byref(c))
        if rc == 1:
            return None
        check(rc)
        nbytes = r.value * dt.row_bytes(d.value, c.value)
        out = np.empty(nbytes, np.uint8)
        check(lib().ifa_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, nbytes, None))
        check(lib().ifa_stream_sync(None))
        return d.value, out, r.value, c.value

    # ---- tensor-parallel segments (enqueue only; the caller all-reduces in between) ----
    def set_stream(self, stream_ptr):
        check(lib().ifa_model_set_stream(self._h, C.c_void_p(stream_ptr)))

    def tp_begin(self, token, pos):
        check(lib().ifa_model_tp_begin(self._h, int(token), int(pos)))

    def tp_begin_hidden(self, x_dev_f16, pos):
        check(lib().ifa_model_tp_begin_hidden(self._h, C.c_void_p(x_dev_f16.data_ptr()), int(pos)))

    def tp_hidden(self, x_out_dev_f16):
        check(lib().ifa_model_tp_hidden(self._h, C.c_void_p(x_out_dev_f16.data_ptr())))

    def tp_attn(self, layer, partial):
        check(lib().ifa_model_tp_attn(self._h, layer, C.c_void_p(partial.data_ptr())))

    def tp_post_attn(self, layer, reduced):
        check(lib().ifa_model_tp_post_attn(self._h, layer, C.c_void_p(reduced.data_ptr())))

    def tp_ffn(self, layer, partial):
        check(lib().ifa_model_tp_ffn(self._h, layer, C.c_void_p(partial.data_ptr())))

    def tp_post_ffn(self, layer, reduced):
        check(lib().ifa_model_tp_post_ffn(self._h, layer, C.c_void_p(reduced.data_ptr())))

    def tp_logits(self, shard_out):
        check(lib().ifa_model_tp_logits(self._h, C.c_void_p(shard_out.data_ptr())))

    def tp_set_token(self, token_dev_int32):
        check(lib().ifa_model_tp_set_token(self._h, C.c_void_p(token_dev_int32.data_ptr())))

    def buffer(self, name, layer=0):
        p, n = C.c_void_p(), C.c_size_t()
        check(lib().ifa_model_get_buffer(self._h, name.encode(), layer, C.byref(p), C.byref(n)))
        return p.value, n.value

    def read_buffer(self, name, layer=0, nbytes=None):
        p, n = self.buffer(name, layer)
        n = n if nbytes is None else nbytes
        out = np.empty(n, np.uint8)
        check(lib().ifa_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(p), n, None))
        check(lib().ifa_stream_sync(None))
        return out

    def write_buffer(self, name, data, layer=0, offset=0):
        """host bytes -> a worker buffer (debug surface of the layer-wise parity tests: the layer input "x", K / V cache rows)"""
        a = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        p, n = self.buffer(name, layer)
        assert offset + a.size <= n, (name, offset, a.size, n)
        check(lib().ifa_memcpy_h2d(C.c_void_p(p + offset), a.ctypes.data_as(C.c_void_p), a.size, None))
        check(lib().ifa_stream_sync(None))

    def close(self):
        if self._h:
            lib().ifa_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TpTopology(C.Structure):
    """ifa_tp_topology (include/inferflow_amd.h)"""
    _fields_ = [("tp", C.c_void_p), ("world", C.c_void_p), ("stage", C.c_int), ("n_stages", C.c_int), ("prev_rank", C.c_int),
                ("next_rank", C.c_int), ("token_src", C.c_int), ("vocab_offset", C.c_int), ("force_collectives", C.c_int)]


class Comm:
    """One rank's communicator of csrc/ifa_comm.hip (RCCL).  unique_id() on rank 0 -> ship the 128 bytes -> Comm(id, ...)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().ifa_comm_unique_id(buf))
        return buf.raw

    def __init__(self, unique_id, nranks, rank, device=0):
        self._h = C.c_void_p()
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        check(lib().ifa_comm_init_rank(self._id, int(nranks), int(rank), int(device), C.byref(self._h)))
        self.rank, self.nranks = rank, nranks

    @classmethod
    def init_all(cls, devices):
        """One communicator per entry of `devices` in THIS process (rank threads); a device named several times makes the
        in-process loopback group (see csrc/ifa_comm.hip)."""
        n = len(devices)
        devs = (C.c_int * n)(*devices)
        hs = (C.c_void_p * n)()
        check(lib().ifa_comm_init_all(devs, n, hs))
        out = []
        for r in range(n):
            c = cls.__new__(cls)
            c._h, c._id, c.rank, c.nranks = C.c_void_p(hs[r]), None, r, n
            out.append(c)
        return out

    def all_reduce_f16(self, t, stream=None):
        check(lib().ifa_allreduce_sum_f16(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(stream)))

    def all_gather(self, src, dst, stream=None):
        check(lib().ifa_allgather(self._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel() * src.element_size(), C.c_void_p(stream)))

    def broadcast(self, t, root, stream=None):
        check(lib().ifa_broadcast(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(root), C.c_void_p(stream)))

    def send(self, t, peer, stream=None):
        check(lib().ifa_send(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(peer), C.c_void_p(stream)))

    def recv(self, t, peer, stream=None):
        check(lib().ifa_recv(self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(peer), C.c_void_p(stream)))

    def oneshot(self):
        """True if small all-reduces of this communicator take the peer-mapped one-shot exchange (csrc/ifa_comm.hip)."""
        return bool(lib().ifa_comm_oneshot(self._h))

    def oneshot_export(self):
        """One process per rank: allocate this rank's one-shot buffers; returns their IPC handles (128 bytes) for the peers."""
        buf = (C.c_ubyte * 128)()
        check(lib().ifa_comm_oneshot_export(self._h, buf))
        return bytes(buf)

    def oneshot_import(self, handles_all_ranks):
        """Map the peers' buffers: handles_all_ranks = the concatenated oneshot_export() results of all ranks, in rank order."""
        b = bytes(handles_all_ranks)
        check(lib().ifa_comm_oneshot_import(self._h, (C.c_ubyte * len(b)).from_buffer_copy(b)))

    def set_oneshot(self, on):
        check(lib().ifa_comm_set_oneshot(self._h, int(bool(on))))

    def size(self):
        """Ranks of the communicator as the library reports them (ncclCommCount for RCCL communicators)."""
        return int(lib().ifa_comm_size(self._h))

    def status(self):
        return int(lib().ifa_comm_status(self._h))

    def abort(self):
        check(lib().ifa_comm_abort(self._h))

    def close(self):
        if self._h:
            lib().ifa_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tp_decode_batch(worker, tokens, positions, slots, tp=None, vocab_offset=0, logits_shard_out=None, force_collectives=False):
    """ifa_model_tp_decode_batch: one new token for each of n queries over a tensor-parallel group; returns the n tokens"""
    topo = TpTopology(tp._h if tp is not None else None, None, 0, 1, -1, -1, 0, vocab_offset, 1 if force_collectives else 0)
    toks = np.ascontiguousarray(tokens, np.int32); pos = np.ascontiguousarray(positions, np.int32); sl = np.ascontiguousarray(slots, np.int32)
    out = np.zeros(toks.size, np.int32)
    check(lib().ifa_model_tp_decode_batch(worker._h, C.byref(topo), toks.size, toks.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                                          sl.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                          C.c_void_p(logits_shard_out.data_ptr()) if logits_shard_out is not None else None))
    return out


def tp_prefill(worker, tokens, start_pos, tp=None, world=None, stage=0, n_stages=1, prev_rank=-1, next_rank=-1, token_src=0,
               vocab_offset=0, logits_shard_out=None, force_collectives=False):
    """ifa_model_tp_prefill: n tokens as one T > 1 step over the partition; returns the greedy next token"""
    topo = TpTopology(tp._h if tp is not None else None, world._h if world is not None else None, stage, n_stages, prev_rank,
                      next_rank, token_src, vocab_offset, 1 if force_collectives else 0)
    toks = np.ascontiguousarray(tokens, np.int32)
    nxt = C.c_int(-1)
    check(lib().ifa_model_tp_prefill(worker._h, C.byref(topo), toks.ctypes.data_as(C.c_void_p), toks.size, int(start_pos),
                                     C.c_void_p(logits_shard_out.data_ptr()) if logits_shard_out is not None else None, C.byref(nxt)))
    return nxt.value


def tp_decode(worker, first_token, start_pos, n_steps, tp=None, world=None, stage=0, n_stages=1, prev_rank=-1, next_rank=-1,
              token_src=0, vocab_offset=0, force_collectives=False):
    """ifa_model_tp_decode: the whole multi-GPU greedy decode driven from C (segments + RCCL collectives + distributed
    argmax, hipGraph per step).  Returns (tokens, gpu_ms of steps 1..n-1)."""
    topo = TpTopology(tp._h if tp is not None else None, world._h if world is not None else None, stage, n_stages, prev_rank,
                      next_rank, token_src, vocab_offset, 1 if force_collectives else 0)
    out = np.zeros(n_steps, np.int32)
    ms = C.c_float(0.0)
    check(lib().ifa_model_tp_decode(worker._h, C.byref(topo), int(first_token), int(start_pos), int(n_steps),
                                    out.ctypes.data_as(C.c_void_p), C.byref(ms)))
    return out, ms.value
