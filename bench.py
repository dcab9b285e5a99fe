#!/usr/bin/env python3
"""Headline benchmark: greedy batch-1 decode of a Llama-2-7B-shaped model with
Q4_B32T1A weights (BASELINE.json configs[1]) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

A "step" is one decoded token (one pass of the fused decode path over the whole
model).  Weights are synthetic N(0,0.02) tensors quantised on the device with
the reference rule (inferflow_amd/synth.py); the prompt is 16 random token ids.
Rank 0 prints ONE JSON line.  Extra keys:
  roofline      -- dominant kernel (fused W1/W3 GEMV): algorithmic bytes / HIP-event time
  cpu_baseline  -- the oracle port of the same quantised decode path timed on the host
                   cores (bounded sample; rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X spec (MI355X_MICROARCH.md)
PROMPT_LEN = 16


def build_fast_oracle():
    """The oracle's sources compiled for THIS host (-O3 -march=native, still without FMA contraction: same results) into a
    temporary directory: the parity object (oracle/libifa_oracle.so: -O2, no -march) understates what the host cores do."""
    import shutil
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle")
    d = tempfile.mkdtemp(prefix="ifa_oracle_fast_")
    so = os.path.join(d, "libifa_oracle_fast.so")
    cmd = [shutil.which("gcc") or "gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-ffp-contract=off", "-fopenmp", "-shared", "-o", so,
           os.path.join(src, "ifa_oracle.c"), os.path.join(src, "ifa_oracle_model.c"), "-lm"]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return so, " ".join(cmd[1:7])


CPU_PROMPT = 4      # tokens fed (one at a time, through the T = 1 path) before the timed CPU decode


def cpu_baseline(wk_host_tensors, shape, n_tokens, threads, kv_dtype):
    """Oracle port (test infrastructure used as the measured CPU baseline, never as the product).  Returns tok/s, seconds and
    the greedy tokens (compared with the GPU stream on the same prompt by the caller)."""
    import numpy as np
    import oracle as o
    from inferflow_amd import dtypes as dt
    max_ctx = PROMPT_LEN + n_tokens + 4
    m = o.Model(dim=shape["dim"], layers=shape["layers"], heads=shape["heads"], kv_heads=shape["kv_heads"],
                head_dim=shape["head_dim"], ffn=shape["ffn"], vocab=shape["vocab"], max_ctx=max_ctx, kv_dtype=kv_dtype)
    for (layer, tid), (dtype, arr, rows, cols) in wk_host_tensors.items():
        m.set_tensor(max(layer, 0), tid, dtype, arr, rows, cols)
    rng = np.random.default_rng(42)
    prompt = rng.integers(3, shape["vocab"], CPU_PROMPT).astype(np.int32)
    tok = 0
    for i, t in enumerate(prompt):       # decode-path prompt feed (T=1), untimed
        tok, _ = m.forward(np.array([t], np.int32), i, want_logits=False, nthreads=threads)
    toks, gaps = [int(tok)], [None]
    t0 = time.perf_counter()
    for i in range(n_tokens):
        tok, lg = m.forward(np.array([tok], np.int32), len(prompt) + i, want_logits=True, nthreads=threads)
        toks.append(int(tok))
        gaps.append(lg)
    dt_s = time.perf_counter() - t0
    # (outside the timed region) the top-2 gap of every step in units of the logits' standard deviation: a greedy id may only
    # differ from the GPU's where this is inside the stated tolerance (tests/test_gpu_fullsize_oracle.py: 0.08 sqrt(layers) x std)
    rel = [None]
    for lg in gaps[1:]:
        row = np.asarray(lg, dtype=np.float32).reshape(-1)
        top2 = np.partition(row, -2)[-2:]
        rel.append(float(abs(top2[1] - top2[0]) / (row.std() + 1e-30)))
    return n_tokens / dt_s, dt_s, prompt, toks, rel


def gpu_tokens_like_cpu(worker, prompt, cpu_toks):
    """The GPU engine on the CPU baseline's prompt, fed the same way (one token at a time through the fused T = 1 step).
    Returns (teacher-forced ids, free-running ids): the GPU's greedy id at every step when it is fed the CPU stream's history
    (a step only differs where the two argmaxes differ on the SAME history: a near-tie of the top two logits), and its own
    free-running stream (which parts from the CPU's for good at the first such step)."""
    worker.reset()
    tok = 0
    for i, t in enumerate(prompt):
        out, _ = worker.decode(int(t), i, 1, timed=False)
        tok = int(out[0])
    forced = [tok]
    for i in range(len(cpu_toks) - 1):          # history = the CPU stream
        out, _ = worker.decode(int(cpu_toks[i]), len(prompt) + i, 1, timed=False)
        forced.append(int(out[0]))
    worker.reset()
    for i, t in enumerate(prompt):
        out, _ = worker.decode(int(t), i, 1, timed=False)
    free, _ = worker.decode(int(out[0]), len(prompt), len(cpu_toks) - 1, timed=False)
    return forced, [int(out[0])] + [int(t) for t in free]


def reference_cpu_baseline(n_tokens=128):
    """Times the reference's CPU inference path (the real thing, not a port) and this engine on one checkpoint."""
    import shutil
    import tempfile
    import numpy as np
    import oracle as o
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_model_fixtures as gmf
    from inferflow_amd.engine import InferenceEngine
    from tests import engine_fixtures as fx
    if not os.path.exists(gmf.DRIVER):
        raise RuntimeError("oracle/_ref/ifa_ref_engine is not built (needs /root/reference at build time)")
    shape = dict(dim=288, layers=6, heads=6, kv_heads=6, head_dim=48, ffn=768, vocab=32000)      # stories15M (llm_inference.tiny.ini)
    threads = min(o.usable_cpus(), 16)
    d = tempfile.mkdtemp(prefix="ifa_ref_cpu_")
    try:
        ini, _ = gmf.write_ref_model_dir(d + "/", shape, seed=15, std=0.05, ctx=256, shared=True)
        txt = open(ini).read().replace("cpu_threads = 4", "cpu_threads = %d" % threads).replace("return_output_tensors = true", "return_output_tensors = false")
        open(ini, "w").write(txt)
        prompt = np.random.default_rng(15).integers(3, shape["vocab"], PROMPT_LEN).astype(np.int32)
        r = gmf.run_reference(ini, prompt, n_tokens + 1, quiet=True)
        ref_tok_s = n_tokens / (r["decode_ms"] / 1e3)
        # this engine on the same file (F16 weights like the tiny .ini; op-by-op path: 48-wide heads are outside the fused kernels)
        gdir = os.path.join(d, "gpu")
        gini, _ = fx.write_model_dir(gdir, fmt="llama2.c", wd="F16", kvd="F16", ctx=256, s=shape, seed=15, std=0.05, shared_classifier=True, ret="false")
        eng = InferenceEngine.from_ini(gini)
        qid = eng.add_query(prompt)
        eng.generate(qid, 8)
        t0 = time.perf_counter()
        gen, _ = eng.generate(qid, n_tokens)
        gpu_tok_s = n_tokens / (time.perf_counter() - t0)
        eng.close()
        return {"value": ref_tok_s, "unit": "tokens/s", "cores": threads, "kind": "reference",
                "sample": "the reference's CPU path (oracle/_ref/ifa_ref_engine) on a stories15M-shaped llama2.c checkpoint (configs[0]: "
                          "d 288, 6 layers, vocab 32000, F32 weights), %d-token prompt + %d greedy tokens, cpu_threads %d; prefill %.1f ms"
                          % (PROMPT_LEN, n_tokens, threads, r["prefill_ms"]),
                "gpu_same_checkpoint_tok_s": gpu_tok_s}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _host_avail_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def _ref_checkpoint_dir(file_gb):
    """Where the F32 checkpoint goes: memory (/dev/shm) when the host has room for the file AND the reference's own copy of it,
    else the temporary directory on disk."""
    import shutil
    import tempfile
    try:
        shm_free = shutil.disk_usage("/dev/shm").free / 2 ** 30
    except OSError:
        shm_free = 0.0
    if shm_free >= file_gb + 2 and _host_avail_gb() >= 2.4 * file_gb + 8:
        return tempfile.mkdtemp(prefix="ifa_ref_cpu7b_", dir="/dev/shm"), "/dev/shm"
    return tempfile.mkdtemp(prefix="ifa_ref_cpu7b_"), "disk"


def pick_reference_layers(avail_gb=None, disk_gb=None):
    """How many of the 32 layers the reference's CPU path is timed on, by what the host can hold: the F32 llama2.c checkpoint
    (0.81 GB per layer + 0.52 GB) is written to a temporary directory AND read into the reference process's heap.  32 layers need
    ~27 GB of file (page cache can drop it) + ~27 GB of heap; 16 and 8 layers are the fallbacks (the line says which)."""
    import shutil
    import tempfile
    if avail_gb is None:
        avail_gb = _host_avail_gb()
    shm_gb = 0.0
    if disk_gb is None:
        disk_gb = shutil.disk_usage(tempfile.gettempdir()).free / 2 ** 30
        try:
            shm_gb = shutil.disk_usage("/dev/shm").free / 2 ** 30         # (_ref_checkpoint_dir writes there when memory allows)
        except OSError:
            shm_gb = 0.0
    for layers in (32, 16, 8):
        file_gb = layers * 0.81 + 0.55
        on_disk = avail_gb >= 1.35 * file_gb + 6 and disk_gb >= file_gb + 4
        in_memory = shm_gb >= file_gb + 2 and avail_gb >= 2.4 * file_gb + 8
        if on_disk or in_memory:
            return layers
    return 0


def reference_cpu_baseline_7b_width(n_tokens=16, layers=32, with_gpu=True):
    """The reference's CPU path on the HEADLINE model (SURVEY 8d "C2-shaped weights, >= 16 tokens"): a Llama-2-7B
    llama2.c checkpoint of `layers` layers (F32 weights: what its CPU side runs; all 32 layers = the headline model, a 27 GB file)
    with this engine on the same file (Q4 weights, F16 KV cache: the headline formats) next to it.  Machine-readable: "layers",
    "model_layers", "full_model", "bytes_per_token"."""
    import shutil
    import tempfile
    import numpy as np
    import oracle as o
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_model_fixtures as gmf
    from tests import engine_fixtures as fx
    if not os.path.exists(gmf.DRIVER):
        raise RuntimeError("oracle/_ref/ifa_ref_engine is not built (needs /root/reference at build time)")
    shape = dict(dim=4096, layers=layers, heads=32, kv_heads=32, head_dim=128, ffn=11008, vocab=32000)
    threads = min(o.usable_cpus(), 16)
    d, where = _ref_checkpoint_dir(layers * 0.81 + 0.55)
    t_start = time.perf_counter()
    try:
        # the checkpoint in the llama2.c layout (model_reader.cc:3248-3430), written tensor by tensor from the fast float32 generator
        # (tests/engine_fixtures.make_weights draws float64: half a minute for 0.33 G values)
        import struct
        rng = np.random.default_rng(16)
        ctx = 128
        # one pool of normal draws, every tensor a window of it at its own offset (drawing 6.7 G values would take longer than
        # the measurement; the values only have to be well-conditioned and different from tensor to tensor)
        biggest = shape["vocab"] * shape["dim"]
        pool = rng.standard_normal(biggest + (1 << 22), dtype=np.float32) * np.float32(0.02)
        with open(os.path.join(d, "model.bin"), "wb") as f:
            f.write(struct.pack("<7i", shape["dim"], shape["ffn"], layers, shape["heads"], shape["kv_heads"], shape["vocab"], ctx))      # vocab > 0: shared classifier
            f.write(memoryview(pool[:biggest]))
            k = 0
            for tid, kind in fx.KINDS:
                for l in range(layers):
                    r_, c_ = fx._shape(kind, shape)
                    k += 1
                    off = (k * 1000003) % (1 << 22)
                    t = pool[off:off + r_ * c_]
                    f.write(memoryview(np.ascontiguousarray((1.0 + t) if kind == "norm" else t)))      # (no tobytes() copy: 26 GB go through here)
            f.write(np.ones((1, shape["dim"]), np.float32).tobytes())
            f.write(b"\0" * (ctx * shape["head_dim"]))
        del pool
        gmf.write_tokenizer(os.path.join(d, "tokenizer.bin"), shape["vocab"])
        spec = json.loads(json.dumps(fx.SPEC)); spec["tokenizer_file"] = "tokenizer.bin"; spec["qkv_format"] = 1
        json.dump(spec, open(os.path.join(d, "model_spec.json"), "w"))
        ini = os.path.join(d, "engine.ini")
        open(ini, "w").write(gmf.REF_INI.format(ctx=ctx).replace("cpu_threads = 4", "cpu_threads = %d" % threads).replace("return_output_tensors = true", "return_output_tensors = false"))
        prompt = np.random.default_rng(16).integers(3, shape["vocab"], 4).astype(np.int32)
        t_written = time.perf_counter()
        r = gmf.run_reference(ini, prompt, n_tokens + 1, quiet=True)
        t_ref = time.perf_counter()
        ref_tok_s = n_tokens / (r["decode_ms"] / 1e3)
        gpu_tok_s = None
        if with_gpu:
            # this engine on the very same model.bin (same directory, its own .ini): Q4 weights quantised at load, F16 cache
            from inferflow_amd.engine import InferenceEngine
            gini = os.path.join(d, "engine_gpu.ini")
            open(gini, "w").write(fx.INI.format(name="ref7b", wd="Q4", kvd="F16", thr=0, ctx=128, ret="false", maxq=2, devices="0", force_partition="false"))
            eng = InferenceEngine.from_ini(gini)
            qid = eng.add_query(prompt)
            (q, tok), = eng.infer()
            eng.commit({qid: tok})
            eng.generate(qid, 8)
            t0 = time.perf_counter()
            eng.generate(qid, 64)
            gpu_tok_s = 64 / (time.perf_counter() - t0)
            eng.close()
        per_layer = (4 * 4096 * 4096 + 3 * 4096 * 11008) * 4
        bytes_tok = layers * per_layer + 32000 * 4096 * 4
        full = layers == 32
        out = {"value": ref_tok_s, "unit": "tokens/s", "cores": threads, "kind": "reference",
               "layers": layers, "model_layers": 32, "full_model": full, "weights": "F32 (the reference's CPU path reads F32 / F16 weights)",
               "bytes_per_token": bytes_tok,
               "sample": "the reference's CPU path (oracle/_ref/ifa_ref_engine) on a %d-layer Llama-2-7B llama2.c checkpoint (d 4096, ffn 11008, "
                         "32 heads, vocab 32000, shared classifier, F32 weights: %.2f GB read per token = %.1f GB/s), 4-token prompt + %d greedy "
                         "tokens, cpu_threads %d; checkpoint written to %s in %.0f s, reference load + prompt + decode %.0f s" % (
                             layers, bytes_tok / 1e9, bytes_tok * ref_tok_s / 1e9, n_tokens, threads, where, t_written - t_start, t_ref - t_written),
               "seconds": time.perf_counter() - t_start,
               "gpu_same_checkpoint_tok_s": gpu_tok_s}
        if not full:
            out["note"] = ("a %d-layer slice of the headline shape (layers + shared classifier), not the headline workload: the 32-layer model "
                           "streams %.1fx these bytes per token" % (layers, (32 * per_layer + 32000 * 4096 * 4) / float(bytes_tok)))
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


class Watchdog:
    """A hang in communicator setup or in a collective must not cost the whole scaling record: every rank watches its own
    progress and, when a phase overruns its limit, rank 0 prints ONE JSON line with an "error" key (the driver's contract)
    and the process exits -- which makes the launcher tear the other ranks down.  Limits: IFA_BENCH_TIMEOUT_INIT /
    IFA_BENCH_TIMEOUT_STEP seconds."""

    def __init__(self, rank, world, args):
        import threading
        self.rank, self.world, self.args = rank, world, args
        self.phase, self.deadline, self.out_fd = "start", None, 1
        self.lock = threading.Lock()
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def arm(self, phase, seconds):
        with self.lock:
            self.phase, self.deadline = phase, time.monotonic() + seconds

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _watch(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                phase = self.phase
            if late:
                msg = "rank %d of %d made no progress in phase '%s' within its limit: aborting the run" % (self.rank, self.world, phase)
                sys.stderr.write("bench.py watchdog: %s\n" % msg)
                sys.stderr.flush()
                if self.rank == 0:
                    line = json.dumps({"metric": "decode tokens/sec (aborted)", "value": None, "unit": "tokens/s", "n_gpus": self.world,
                                       "steps": self.args.steps, "warmup": self.args.warmup, "error": msg, "higher_is_better": True})
                    try:
                        os.write(self.out_fd, (line + "\n").encode())
                    except OSError:
                        pass
                os._exit(3)


def loopback_line(args):
    """R tensor-parallel ranks on device 0 through the product surface (.ini -> InferenceEngine -> MultiGpu -> loopback group)."""
    import tempfile
    import numpy as np
    import torch
    from inferflow_amd import synth
    from inferflow_amd.engine import InferenceEngine
    s = synth.SHAPES[args.shape]
    R = args.loopback
    d = tempfile.mkdtemp(prefix="ifa_loopback_")
    spec = {"config_file": "", "model_files": [], "model_file_format": "synthetic", "tokenizer_file": "", "tokenization_algorithm": "bpe",
            "generation_config": "", "synthetic_std": 0.02,
            "hyper_params": {"vocab_size": s["vocab"], "embd_dims": s["dim"], "hidden_dim": s["ffn"], "decoder_layers": s["layers"],
                             "decoder_heads": s["heads"], "decoder_kv_heads": s["kv_heads"]},
            "network_structure": {"type": "transformer.llama", "normalization_function": "rms", "activation_function": "silu",
                                  "position_embedding": "rope", "qk_column_order": 2, "tensor_name_prefix": "", "tensor_name_mapping": {}}}
    json.dump(spec, open(os.path.join(d, "model_spec.json"), "w"))
    ini = os.path.join(d, "engine.ini")
    open(ini, "w").write("[transformer_engine]\nmodels = bench\ndevices = %s\ndecoder_cpu_layer_count = 0\ncpu_threads = 8\n"
                         "max_concurrent_queries = 2\nreturn_output_tensors = false\n\n[model.bench]\nmodel_dir = ${config_dir}\n"
                         "model_specification_file = model_spec.json\ndevice_weight_data_type = %s\ndevice_kv_cache_data_type = %s\n"
                         "tensor_quant_threshold = 0\nmax_context_len = %d\nprompt_template = {bos}{query}\n" % (
                             "&".join(["0"] * R), {"q4": "Q4", "q3h": "Q3H", "q8": "Q8"}.get(args.wdtype.lower(), "Q4"),
                             {"f16": "F16", "q8": "Q8"}[args.kv_dtype.lower()], PROMPT_LEN + args.warmup + args.steps + 16))
    sys.stdout.flush()
    saved = os.dup(1); os.dup2(2, 1)
    dog = Watchdog(0, 1, args); dog.out_fd = saved
    dog.arm("loopback engine init", float(os.environ.get("IFA_BENCH_TIMEOUT_INIT", "420")))
    eng = InferenceEngine.from_ini(ini)
    qid = eng.add_query(np.random.default_rng(42).integers(3, s["vocab"], PROMPT_LEN).astype(np.int32))
    (q, tok), = eng.infer()
    eng.commit({qid: tok})
    dog.arm("loopback steps", float(os.environ.get("IFA_BENCH_TIMEOUT_STEP", "180")))
    if args.warmup > 0:
        eng.generate(qid, args.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gen, _ = eng.generate(qid, args.steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dog.disarm()
    ranks = eng.model_info("partition_ranks")
    eng.close()
    out = {"metric": "decode tokens/sec, %s batch=1 greedy (whole job)" % args.shape, "value": args.steps / wall, "unit": "tokens/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "i8", "data": "synthetic",
           "config": {"workload": "%s decode through the .ini surface, %d tensor-parallel ranks on ONE device" % (args.shape, R),
                      "parallelism": "tp%d loopback (one process, %d worker threads, in-process group instead of RCCL; partition_ranks = %d)" % (R, R, ranks),
                      "collectives": "loopback group (csrc/ifa_comm.hip LocalGroup): same slicing, merges and step logic as the RCCL path"},
           "note": "functional check of the N > 1 path on a one-GPU box: R ranks share the device, so this is not a scaling number",
           "last_tokens": [int(t) for t in gen[-4:]]}
    os.dup2(saved, 1)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--shape", default="llama2_7b")
    ap.add_argument("--wdtype", default="q4", help="weight format: q4 (Q4_B32T1A, the headline config), q3h, q8, q4_b64, q5, q6")
    ap.add_argument("--kv-dtype", default="f16", help="KV cache: f16 or q8 (configs[2] = --wdtype q3h --kv-dtype q8)")
    ap.add_argument("--groups", type=int, default=1, help="device groups (layer ranges) of the HYBRID partition; "
                    "gpus // groups ranks per group are tensor-parallel (default 1 = pure tensor parallelism)")
    ap.add_argument("--prefill-lens", default="128,1024", help="extra prompt lengths whose prefill rate is reported (N=1 only)")
    ap.add_argument("--batch", type=int, default=8, help="also time dynamic-batching decode: B queries, one new token each per "
                    "step (N=1 only; reported as batch_decode, never as value; 0 = skip)")
    ap.add_argument("--loopback", type=int, default=0, help="R > 1: run the tensor-parallel partition with R ranks on ONE device "
                    "(the C++ InferenceEngine with devices = 0&0..., in-process loopback group instead of RCCL) and print its line; "
                    "a functional check of the N > 1 step logic on a one-GPU box, not a scaling number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=0, help="CPU baseline sample size (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from inferflow_amd import dtypes as dt, synth
    if args.loopback > 1:
        return loopback_line(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    t_init = float(os.environ.get("IFA_BENCH_TIMEOUT_INIT", "420"))
    t_step = float(os.environ.get("IFA_BENCH_TIMEOUT_STEP", "180"))
    dog = Watchdog(rank, world, args)
    dog.arm("process group + weights + communicators", t_init)
    if args.gpus > 1 or world > 1 or os.environ.get("IFA_FORCE_TP"):
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    else:
        torch.cuda.set_device(0)

    from inferflow_amd import parallel
    # the reference's .ini spellings (device_weight_data_type / device_kv_cache_data_type)
    WD = {"q4": dt.Q4_B32T1A, "q3h": dt.Q3H_B64T1, "q8": dt.Q8_B32T2, "q4_b64": dt.Q4_B64T1, "q5": dt.Q5_B64T1, "q6": dt.Q6_B64T1}
    wd = WD[args.wdtype.lower()]
    kvd = {"f16": dt.F16, "q8": dt.Q8_B32T2}[args.kv_dtype.lower()]
    steps, warmup = args.steps, args.warmup
    prefill_lens = [int(x) for x in args.prefill_lens.split(",") if x] if world == 1 else []
    max_ctx = max([PROMPT_LEN + warmup + steps + 8] + [n + 8 for n in prefill_lens])
    # RCCL prints a version banner on stdout when its first communicator is created: stdout carries ONE JSON line, so
    # everything up to the final print goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    dog.out_fd = saved_stdout
    os.dup2(2, 1)
    rng = np.random.default_rng(42)
    prompt = rng.integers(3, synth.SHAPES[args.shape]["vocab"], PROMPT_LEN).astype(np.int32)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- build + prefill + warm-up as a TRIAL.  With more than one rank the C path (RCCL collectives captured into the step's
    # hipGraph) has never run on hardware here: if any rank fails, the failing rank aborts its communicators (peers blocked in a
    # collective return an error), all ranks agree on the outcome over torch.distributed and the next mode is tried -- the same
    # C path with eager (uncaptured) steps, then the round-1 runner (torch.distributed collectives around the worker segments).
    multi = world > 1 or bool(os.environ.get("IFA_FORCE_TP"))
    modes = ["c-oneshot", "c-graph", "c-eager", "torch"] if multi else ["single"]
    collective_modes = None
    if os.environ.get("IFA_TP_BACKEND", "c") == "torch":
        modes = ["torch"]
    inject = [x for x in os.environ.get("IFA_BENCH_FAIL_MODES", "").split(",") if x]      # tests: pretend these modes fail
    fallbacks, runner, mode_used = [], None, None
    for mode in modes:
        ok, err = 1, ""
        try:
            dog.arm("build + prefill + warm-up [%s]" % mode, t_init)
            if mode == "torch":
                os.environ["IFA_TP_BACKEND"] = "torch"
            # c-oneshot: the C path with the cross-process one-shot exchange for the decode-size all-reduces (IPC-mapped inboxes)
            os.environ["IFA_ONESHOT_IPC"] = "1" if mode == "c-oneshot" else "0"
            if mode == "c-oneshot" and (world // max(1, args.groups)) < 2:
                continue                                   # no tensor-parallel group of 2+ ranks: nothing to exchange
            t_build = time.perf_counter()
            runner = parallel.build_runner(args.shape, wd, kvd, max_ctx, world, rank, local_rank, groups=args.groups)
            t_build = time.perf_counter() - t_build
            if mode == "c-eager":
                runner.worker.set_option("graph", 0)
            if mode in inject:
                raise RuntimeError("injected failure of mode %s" % mode)
            barrier()
            t0 = time.perf_counter()
            tok = runner.prefill(prompt)
            barrier()
            prefill_s = time.perf_counter() - t0
            if mode == "c-oneshot":
                # never validated on hardware before this run: a fixed K steps through the exchange and the same K steps again through
                # RCCL (a step rewrites the same cache rows: idempotent) must give the same tokens, and no wait of the exchange may have
                # given up.  Both runs are timed (max over ranks): ONE multi-GPU run records both collective modes side by side.
                if not getattr(runner, "oneshot_ipc", False):
                    raise RuntimeError("one-shot exchange not available")
                K = 8
                def timed_k():
                    barrier(); t_ = time.perf_counter()
                    tk, _ = runner.decode(tok, PROMPT_LEN, K)
                    barrier()
                    return [int(t) for t in tk], (time.perf_counter() - t_) * 1e3 / K
                timed_k()                                                                      # capture + first launches
                toks_o, ms_o = timed_k()
                runner.tp_comm.set_oneshot(0); runner.worker.set_option("graph", 1)       # (any option change drops the captured step)
                timed_k()
                toks_r, ms_r = timed_k()
                if runner.tp_comm.status() != 0 or toks_o != toks_r:
                    raise RuntimeError("one-shot exchange disagrees with RCCL: %r vs %r (status %d)" % (toks_o, toks_r, runner.tp_comm.status()))
                runner.tp_comm.set_oneshot(1); runner.worker.set_option("graph", 1)
                collective_modes = {"oneshot_ms_per_step": ms_o, "rccl_in_graph_ms_per_step": ms_r, "steps": K,
                                    "ranks_seen": int(runner.tp_comm.size()) if hasattr(runner.tp_comm, "size") else world // max(1, args.groups)}
            toks_w, _ = runner.decode(tok, PROMPT_LEN, warmup) if warmup > 0 else ([tok], 0.0)
            tok = int(toks_w[-1])
        except Exception as e:      # noqa: BLE001
            ok, err = 0, repr(e)[:300]
            for cm in (getattr(runner, "tp_comm", None), getattr(runner, "world_comm", None)):
                try:
                    if cm is not None:
                        cm.abort()
                except Exception:      # noqa: BLE001
                    pass
        if multi and dist.is_initialized():
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok_all = int(flag.item())
        else:
            ok_all = ok
        if ok_all:
            mode_used = mode
            break
        fallbacks.append({"mode": mode, "error_on_this_rank": err or "a peer failed"})
        print("bench: mode %s failed on rank %d (%s); trying the next one" % (mode, rank, err or "a peer failed"), file=sys.stderr, flush=True)
        runner = None
    if mode_used is None:
        raise RuntimeError("every multi-GPU mode failed: %r" % (fallbacks,))

    # set-up of the timed call (the captured step and its multi-step replay for the contexts it reaches) happens here, not inside
    # the timed region -- whatever --warmup is; no step runs
    if hasattr(runner, "decode_prepare"):
        runner.decode_prepare(PROMPT_LEN + warmup, steps)
    dog.arm("timed steps", t_step)
    barrier()
    t0 = time.perf_counter()
    toks, gpu_ms = runner.decode(tok, PROMPT_LEN + warmup, steps)
    barrier()
    wall = time.perf_counter() - t0
    dog.arm("reduce of the step times", t_step)
    if world > 1:
        tmax = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())

    dog.disarm()
    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    tok_s = steps / wall
    n_avg = PROMPT_LEN + warmup + steps / 2.0
    # ALGORITHMIC bytes (the reference's own block sizes -- Q3H_B64T1 is 32 B per 64 weights) are what every roofline
    # figure of this line is computed on; the bytes the decode kernels actually stream are reported beside them (they differ for
    # Q3H_B64T1 only: its pair codes are expanded to nibbles at load time, 36 B per 64 weights -- 1.125x wasted traffic by construction)
    w_bytes = synth.weight_bytes(args.shape, wd, streamed=False)
    w_bytes_streamed = synth.weight_bytes(args.shape, wd, streamed=True)
    kv_bytes = synth.kv_bytes_per_ctx_row(args.shape, kvd)
    bytes_per_token = w_bytes + kv_bytes * n_avg
    bytes_per_token_streamed = w_bytes_streamed + kv_bytes * n_avg
    out = {
        "metric": "decode tokens/sec, %s %s batch=1 greedy (whole job)" % (
            {"llama2_7b": "Llama-2-7B", "mixtral_8x7b": "Mixtral-8x7B", "yi_34b": "Yi-34B", "falcon_40b": "Falcon-40B",
             "llama2_70b": "Llama-2-70B"}.get(args.shape, args.shape),
            "Q4" if wd == dt.Q4_B32T1A else dt.name(wd)),
        "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": wall * 1e3 / steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "i8 (Q8 activations x %s weights, i32 dot, f32 scale, f16 I/O)" % dt.name(wd),
        "data": "synthetic",
        "config": {"workload": "%s decode, %s weights + F16 lm_head, %s KV cache, batch 1 greedy, "
                               "%d-token prompt, context %d..%d" % (args.shape, dt.name(wd), dt.name(kvd), PROMPT_LEN,
                                                                  PROMPT_LEN + warmup, PROMPT_LEN + warmup + steps),
                   "parallelism": ("single" if world == 1 else "tp%d" % world if args.groups == 1
                                   else "hybrid: %d layer groups x tp%d" % (args.groups, world // args.groups)),
                   "collectives": (getattr(runner, "backend", "torch.distributed (nccl = RCCL)") + (" [eager steps]" if mode_used == "c-eager" else " [decode-size all-reduces: one-shot exchange over IPC-mapped inboxes, checked against RCCL]" if mode_used == "c-oneshot" else "")) if (world > 1 or os.environ.get("IFA_FORCE_TP")) else None,
                   "fallbacks": fallbacks,
                   "weights_bytes": w_bytes,
                   "bytes_per_token": bytes_per_token,
                   "streamed_bytes_per_token": bytes_per_token_streamed},
        "gpu_event_ms_per_step": gpu_ms / steps if gpu_ms and gpu_ms > 0 else None,
        "collective_modes": collective_modes,
        "token_hbm_GBps": bytes_per_token * tok_s / 1e9,
        "token_roofline_frac": bytes_per_token * tok_s / 1e9 / HBM_PEAK_GBPS,
        "token_roofline_frac_streamed": bytes_per_token_streamed * tok_s / 1e9 / HBM_PEAK_GBPS,
        "prefill_tok_s": PROMPT_LEN / prefill_s,
        "build_s": t_build,
        "last_tokens": [int(t) for t in toks[-4:]],
    }
    # ---- roofline of the dominant kernel, timed live with HIP events on the worker's stream
    is_moe = bool(runner.shape.get("experts", 0))
    if world == 1 and hasattr(runner, "export_host_tensors") and not os.environ.get("IFA_FORCE_TP") and not is_moe:
        s = runner.shape
        ffn_rows, d = s["ffn"], s["dim"]
        rb = dt.row_bytes                          # ALGORITHMIC bytes per row (the reference's block sizes; Q3H_B64T1: 32 per 64 weights)
        ffn13_bytes = (2 if s.get("is_glu", 1) else 1) * ffn_rows * rb(wd, d)
        ffn13_streamed = (2 if s.get("is_glu", 1) else 1) * ffn_rows * dt.streamed_row_bytes(wd, d)
        us = runner.worker.time_kernel(3, 320)
        per_kernel = {}
        names = ["qkv", "attn", "wo", "ffn13", "w2", "lm_head"]
        kb = [(s["heads"] + 2 * s["kv_heads"]) * s["head_dim"] * rb(wd, d), None,
              d * rb(wd, s["heads"] * s["head_dim"]), ffn13_bytes,
              d * rb(wd, ffn_rows), s["vocab"] * d * 2]
        for i, nm in enumerate(names):
            u = runner.worker.time_kernel(i, 160)
            per_kernel[nm] = {"us": u, "GBps": (kb[i] / u / 1e3) if kb[i] else None}
        # HBM bytes per launch of that kernel from the PMC pass (rocprofv3 --pmc FETCH_SIZE, own run: tools/profile_r02.sh;
        # summary and correction in profiles/r02_pmc_traffic.json, produced by tools/pmc_summary.py).  The file names the
        # build it was taken with (hash of the kernel sources): numbers of another build are NOT reported
        traffic, traffic_note, traffic_source = None, None, None
        try:
            from inferflow_amd.build import source_hash
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]        # the newest round's
            if wd == dt.Q3H_B64T1:          # configs[2]: its own PMC pass (tools/profile_r05.sh)
                q = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_q3h_q8_traffic.json")))
                pmc_path = q[-1] if q else pmc_path
            pmc = json.load(open(pmc_path))
            traffic_source = "profiles/" + os.path.basename(pmc_path) + " (rocprofv3 --pmc FETCH_SIZE pass of this build, tools/refresh_pmc.sh; replayed, not measured in this run)"
            if pmc.get("source_hash") != source_hash():
                traffic_note = "profiles/%s was taken with kernel sources %s, this build is %s: stale, not reported" % (
                    os.path.basename(pmc_path), pmc.get("source_hash"), source_hash())
            elif wd == dt.Q4_B32T1A and args.shape == "llama2_7b":
                import re                                                                     # <DT 13, NJ 2, RW any, EPI_GLU 2, NORM 1, ...>
                traffic = [v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items()     # demangled or mangled name
                           if re.match(r"ifa::k_dec_gemv<13, 2, \d+, 2, 1", k) or re.match(r"_ZN3ifa10k_dec_gemvILi13ELi2ELi\d+ELi2ELi1E", k)][0]
            elif wd == dt.Q3H_B64T1 and args.shape == "llama2_7b" and "q3h" in os.path.basename(pmc_path):
                import re                                                                     # <DT 18, NJ 1, RW any, EPI_GLU 2, NORM 1, ...>
                traffic = [v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items()
                           if re.match(r"ifa::k_dec_gemv<18, 1, \d+, 2, 1", k) or re.match(r"_ZN3ifa10k_dec_gemvILi18ELi1ELi\d+ELi2ELi1E", k)][0]
        except Exception as e:
            traffic, traffic_note = None, "no PMC summary: %r" % (e,)
        out["roofline"] = {"bound": "hbm", "kernel": "k_dec_gemv<%s, EPI_GLU> (fused RMSNorm+Q8 quant+W1/W3 GEMV+SiLU*mul)" % dt.name(wd),
                           "achieved": ffn13_bytes / us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": ffn13_bytes / us / 1e3 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note, "traffic_source": traffic_source,
                           "bytes_per_launch": ffn13_bytes, "streamed_bytes_per_launch": ffn13_streamed, "us_per_launch": us}
        out["kernels"] = per_kernel
    # ---- prefill rate at longer prompts (16 / 128 / 1024-token prompts), outside the timed decode region
    if world == 1 and hasattr(runner, "worker") and not os.environ.get("IFA_FORCE_TP"):
        pf = {}
        out["prefill_tok_s_first_call"] = PROMPT_LEN / prefill_s   # cold: code objects, scratch and KV pages touched for the first time
        for n in [PROMPT_LEN] + [n for n in prefill_lens if n != PROMPT_LEN]:
            pr = rng.integers(3, runner.shape["vocab"], n).astype(np.int32)
            runner.worker.forward(pr, 0)                     # warm-up (scratch buffers grow on first use)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            runner.worker.forward(pr, 0)
            torch.cuda.synchronize()
            pf[str(n)] = n / (time.perf_counter() - t0)
        out["prefill_tok_s_by_prompt_len"] = pf
        out["prefill_tok_s"] = pf[str(max([PROMPT_LEN] + prefill_lens))]      # warm, the longest measured prompt (1024 tokens by default)
        flops_per_token = 2.0 * (w_bytes - runner.shape["vocab"] * runner.shape["dim"] * 2) / dt.row_bytes(wd, 32) * 32
        out["prefill_linear_TFLOPs_at_longest"] = pf[str(max([PROMPT_LEN] + prefill_lens))] * flops_per_token / 1e12
        longest = max([PROMPT_LEN] + prefill_lens)
        out["prefill_roofline"] = {"bound": "mfma", "achieved": out["prefill_linear_TFLOPs_at_longest"], "peak": 2500.0, "unit": "TFLOP/s",
                                   "frac": out["prefill_linear_TFLOPs_at_longest"] / 2500.0,
                                   "note": "flops of the layers' linear products (2 x tokens x weights) / WHOLE prefill time of a %d-token prompt "
                                           "(attention, norms, RoPE, cache writes, lm_head of the last row included in the time); dense F16 MFMA peak" % longest}
        out["prefill_gemm_route"] = "in-tree kernels only (csrc/ifa_gemm.hip: large-tile MFMA kernel above 128 tokens, four launches per layer); no vendor GEMM in the library"
    # ---- decode rate at a few hundred keys (secondary key, outside the timed region): the headline line sits at 21..41 keys, a served
    # query at hundreds; a real 16-step call after an n-token prompt (every layer's cache rows read cold, unlike a per-launch timing loop)
    if world == 1 and hasattr(runner, "worker") and not os.environ.get("IFA_FORCE_TP") and not is_moe:
        try:
            by_ctx = {}
            for n in (256, 450):
                if n + 24 > max_ctx:
                    continue
                pr = rng.integers(3, runner.shape["vocab"], n).astype(np.int32)
                tok = runner.worker.forward(pr, 0)
                runner.worker.decode(int(tok), n, 4)
                best = 0.0
                for _ in range(2):
                    _, ms = runner.worker.decode(int(tok), n, 16)
                    best = max(best, 16e3 / ms)
                by_ctx[str(n)] = best
            if by_ctx:
                out["decode_tok_s_by_context"] = by_ctx
        except Exception as e:  # noqa: BLE001 -- a secondary leg must not cost the headline line
            out["decode_tok_s_by_context"] = {"error": str(e)[:200]}
    # ---- dynamic batching: B queries with their own KV caches, one decode step appends one token to each (the reference's
    # InferenceEngine::Infer over several queries, inference_engine.cc:1300-1406); outside the timed headline region
    if world == 1 and args.batch > 1 and hasattr(runner, "worker") and not os.environ.get("IFA_FORCE_TP"):
        try:
            B, wk = args.batch, runner.worker
            wk.kv_slots(B)
            toks_b = []
            for b in range(B):
                wk.select_kv(b)
                toks_b.append(wk.forward(rng.integers(3, runner.shape["vocab"], PROMPT_LEN).astype(np.int32), 0))
            slots = np.arange(B, dtype=np.int32)
            pos = np.full(B, PROMPT_LEN, np.int32)
            cur = np.asarray(toks_b, np.int32)
            nb = max(1, min(steps, max_ctx - PROMPT_LEN - warmup - 2))
            for _ in range(min(warmup, 4)):
                cur = wk.decode_batch(cur, pos, slots); pos += 1
            torch.cuda.synchronize()
            # this leg is driven step by step from Python: keep the interpreter's cyclic collector out of the timed region (a
            # full collection of a process that imported torch is a 35-60 ms pause: it showed up as +15 % on 128 steps)
            import gc
            gc.collect(); gc.disable()
            try:
                t0 = time.perf_counter()
                for _ in range(nb):
                    cur = wk.decode_batch(cur, pos, slots); pos += 1
                torch.cuda.synchronize()
                tb = time.perf_counter() - t0
            finally:
                gc.enable()
            wk.select_kv(0)
            out["batch_decode"] = {"batch": B, "steps": nb, "tok_s": B * nb / tb, "ms_per_step": tb * 1e3 / nb,
                                   "note": "host-driven step (tokens returned to the host every step), greedy"}
        except Exception as e:      # an extra leg: never at the expense of the headline line
            out["batch_decode"] = {"batch": args.batch, "error": repr(e)}
    # ---- CPU baseline (oracle port) on a bounded sample
    if world == 1 and not args.no_cpu_baseline and not os.environ.get("IFA_FORCE_TP") and not is_moe:
        try:
            flags = "-O2 (the parity object)"
            try:                                        # the port built for this host's cores; the parity object when that fails
                fast_so, flags = build_fast_oracle()
                os.environ["IFA_ORACLE_LIB"] = fast_so
            except Exception:      # noqa: BLE001
                pass
            import oracle as _o
            threads = min(_o.usable_cpus(), 128)        # affinity and cgroup quota, not the visible CPU count
            host = runner.export_host_tensors()
            n_cpu = args.cpu_tokens or 8
            v, secs, cpu_prompt, cpu_toks, cpu_gaps = cpu_baseline(host, runner.shape, n_cpu, threads, kvd)
            if secs < 5 and not args.cpu_tokens:      # fast host: take a longer sample (~10-30 s)
                n_cpu = int(min(256, max(8, 15.0 / (secs / n_cpu))))
                v, secs, cpu_prompt, cpu_toks, cpu_gaps = cpu_baseline(host, runner.shape, n_cpu, threads, kvd)
            port = {"value": v, "unit": "tokens/s", "cores": threads, "kind": "port",
                    "sample": "%s %s decode of %d tokens after a %d-token prompt (all %d layers), oracle C port (OpenMP, %s), %.1f s" % (
                        args.shape, dt.name(wd), n_cpu, CPU_PROMPT, runner.shape["layers"], flags, secs)}
            # the same prompt through the GPU engine: the port's greedy stream against the fused decode's, at the headline size
            try:
                forced, free = gpu_tokens_like_cpu(runner.worker, cpu_prompt, cpu_toks)
                agree = sum(1 for a, b in zip(cpu_toks, forced) if a == b)
                first = next((i for i, (a, b) in enumerate(zip(cpu_toks, free)) if a != b), None)
                port["tokens_agree_with_gpu"] = "%d/%d" % (agree, len(cpu_toks))       # same history fed to both (teacher forcing)
                # the depth law of tests/test_gpu_fullsize_oracle.py: |dlogit| <= 0.08 sqrt(layers) x std on the int8 T = 1 path
                tol = 0.08 * (runner.shape["layers"] ** 0.5)
                outside = sum(1 for a, b, gp in zip(cpu_toks, forced, cpu_gaps) if a != b and gp is not None and gp > tol)
                port["token_mismatches_outside_tolerance"] = outside                    # top-2 gap of the port's logits > tol x their std: must be 0
                port["token_tolerance_in_std"] = tol
                port["free_running_common_prefix"] = first if first is not None else len(cpu_toks)
                port["tokens_note"] = ("greedy ids of the fused GPU decode against the port's on the SAME token history, all %d layers, "
                                       "random-init weights (flat logits: a top-2 gap inside the stated logit tolerance flips an id; "
                                       "tests/test_gpu_fullsize_oracle.py holds the bound)" % runner.shape["layers"])
            except Exception as e:      # noqa: BLE001
                port["tokens_agree_with_gpu"] = "not compared: %r" % (e,)
            out["cpu_baseline_port"] = port
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline_port"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port",
                                        "sample": "failed: %r" % (e,)}
    # ---- the REFERENCE's own CPU path (oracle/_ref/ifa_ref_engine: /root/reference sources compiled by oracle/Makefile,
    # travels prebuilt) on configs[0] -- the stories15M-shaped llama2.c checkpoint of bin/llm_inference.tiny.ini, the case
    # the reference itself runs on CPU -- with this engine on the very same checkpoint next to it
    if world == 1 and not args.no_cpu_baseline and not os.environ.get("IFA_FORCE_TP"):
        try:
            out["cpu_baseline_reference"] = reference_cpu_baseline()
        except Exception as e:
            out["cpu_baseline_reference"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "reference", "sample": "failed: %r" % (e,)}
        if args.shape == "llama2_7b":
            # the reference on the HEADLINE model: all 32 layers when the host can hold the 27 GB F32 checkpoint (pick_reference_layers),
            # else the largest slice that fits -- and the 8-layer slice of round 4 as its own line either way
            n_ref = int(os.environ.get("IFA_BENCH_REF_LAYERS", "-1"))
            forced = n_ref >= 0
            if n_ref < 0:
                n_ref = pick_reference_layers()
            budget_s = float(os.environ.get("IFA_BENCH_REF_BUDGET_S", "240"))
            # the 8-layer slice first (round 4's line, kept as its own key): what it took also says what all 32 layers will take on
            # this host (file write + load + decode scale with the layer count) -- the default run must stay inside a few minutes
            t8 = None
            if n_ref >= 8 and not os.environ.get("IFA_BENCH_SKIP_8_LAYERS"):
                try:
                    r8 = reference_cpu_baseline_7b_width(layers=8)
                    out["cpu_baseline_reference_7b_8_layers"] = r8
                    t8 = r8.get("seconds")
                except Exception as e:
                    out["cpu_baseline_reference_7b_8_layers"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "reference", "layers": 8, "sample": "failed: %r" % (e,)}
            why = ""
            if not forced and t8:
                while n_ref > 8 and t8 * (n_ref * 0.81 + 0.55) / (8 * 0.81 + 0.55) > budget_s:
                    why = " (all 32 layers predicted at %.0f s from the 8-layer run, over the %.0f s budget IFA_BENCH_REF_BUDGET_S)" % (t8 * (32 * 0.81 + 0.55) / (8 * 0.81 + 0.55), budget_s)
                    n_ref //= 2
            if n_ref > 8:
                try:
                    out["cpu_baseline_reference_7b"] = reference_cpu_baseline_7b_width(layers=n_ref)
                    if why:
                        out["cpu_baseline_reference_7b"]["note"] = out["cpu_baseline_reference_7b"].get("note", "") + why
                except Exception as e:
                    out["cpu_baseline_reference_7b"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "reference", "layers": n_ref, "sample": "failed: %r" % (e,)}
            elif n_ref == 8:
                out["cpu_baseline_reference_7b"] = dict(out.get("cpu_baseline_reference_7b_8_layers") or {}, note_layers="only the 8-layer slice ran" + (why or
                                                        ": %.0f GB of host memory available, all 32 layers need ~42" % _host_avail_gb()))
            else:
                out["cpu_baseline_reference_7b"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "reference", "layers": 0,
                                                    "sample": "not run: %.0f GB of host memory available, the 8-layer checkpoint needs ~15" % _host_avail_gb()}
    # cpu_baseline (the contract's key) is ALWAYS on the headline workload -- all 32 layers of the model the GPU number is quoted on
    # (ADVICE r4): the REFERENCE's own CPU path when the host could hold the full checkpoint (kind "reference", "full_model": true,
    # F32 weights: what that path runs), else the oracle port of the quantised path on the whole model (kind "port").  A slice of
    # the model is never put under this key; slices keep their own keys with a machine-readable "layers" field.
    r7 = out.get("cpu_baseline_reference_7b") or {}
    if r7.get("value") and r7.get("full_model"):
        out["cpu_baseline"] = r7
    elif "cpu_baseline_port" in out:
        out["cpu_baseline"] = out["cpu_baseline_port"]
    sys.stdout.flush()
    try:                                   # C stdio too: RCCL's banner sits in libc's buffer while fd 1 points at stderr; flushed
        import ctypes                      # after the restore it would land on the real stdout next to the JSON line
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)                          # anything printed during teardown (communicator destruction) goes to stderr again
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
