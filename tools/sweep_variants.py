#!/usr/bin/env python3
"""Tuning sweep over compile-time kernel settings (workgroup size / rows in flight of the four per-layer decode GEMVs).

  build (here, no GPU):   python tools/sweep_variants.py build  [name=FLAGS ...]
        one library per variant under gpurun_out_variants/<name>/libinferflow_amd.so: only csrc/ifa_dgemv_q4b32.hip
        is recompiled with the -DIFA_T_* overrides of ifa_decode_gemv_impl.h, the other objects come from lib/obj
  run   (GPU box):        python tools/sweep_variants.py run [--steps 96]
        bench.py once per variant (IFA_LIB=...), one summary line each -> gpurun_out/sweep.jsonl
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "lib_variants")

DEFAULT = {
    "base": "",
    "wo512": "-DIFA_T_TH_WO=512 -DIFA_T_RW_WO=2",
    "wo256": "-DIFA_T_TH_WO=256 -DIFA_T_RW_WO=4",
    "wo1024r1": "-DIFA_T_TH_WO=1024 -DIFA_T_RW_WO=1",
    "qkv1024": "-DIFA_T_TH_QKV=1024 -DIFA_T_RW_QKV=3",
    "qkv512r3": "-DIFA_T_TH_QKV=512 -DIFA_T_RW_QKV=3",
    "glu512": "-DIFA_T_TH_GLU=512 -DIFA_T_RW_GLU=6",
    "glu1024r2": "-DIFA_T_TH_GLU=1024 -DIFA_T_RW_GLU=2",
    "w2_1024": "-DIFA_T_TH_W2=1024 -DIFA_T_RW_W2=1",
    "w2_512r1": "-DIFA_T_TH_W2=512 -DIFA_T_RW_W2=1",
}


UNITS = os.environ.get("IFA_SWEEP_UNIT", "ifa_dgemv_q4b32,ifa_dqkvattn_q4b32").split(",")   # the translation units the overrides are compiled into
BENCH_ARGS = os.environ.get("IFA_SWEEP_BENCH_ARGS", "").split()       # e.g. "--wdtype q3h --kv-dtype q8"


def build(variants):
    from inferflow_amd import build as b
    b.build_library()
    hipcc = b._hipcc()
    obj_dir = os.path.join(b.LIB_DIR, "obj")
    want = [os.path.basename(x) + ".o" for x in b.sources()] + ["host_" + os.path.basename(x) + ".o" for x in b.host_sources()]     # (not whatever old objects lie in the directory)
    others = [os.path.join(obj_dir, f) for f in want if not any(f.startswith(u + ".") for u in UNITS)]
    procs = []
    for name, flags in variants.items():
        d = os.path.join(VDIR, name)
        os.makedirs(d, exist_ok=True)
        for unit in UNITS:
            src = os.path.join(b.CSRC, unit + ".hip")
            obj = os.path.join(d, unit + ".o")
            hdrs = [os.path.join(b.CSRC, h) for h in os.listdir(b.CSRC) if h.endswith(".h")] + [src]
            ff = os.path.join(d, "flags.txt")
            if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(h) for h in hdrs) \
                    and os.path.exists(ff) and open(ff).read().strip() == flags.strip():
                procs.append((name, obj, None))          # object is current: relink only
                continue
            cmd = [hipcc] + b.HIPCC_FLAGS + flags.split() + ["-I", os.path.join(ROOT, "include"), "-I", b.CSRC, "-c", src, "-o", obj]
            procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            live = [q for _, _, q in procs if q is not None and q.poll() is None]
            while len(live) >= 6:
                live[0].wait()
                live = [q for q in live if q.poll() is None]
    failed = set()
    for name, obj, p in procs:
        out, _ = p.communicate() if p is not None else (b"", None)
        if p is not None and p.returncode != 0:
            print("variant %s failed:\n%s" % (name, out.decode(errors="replace")[-3000:]))
            failed.add(name)
    for name in variants:
        if name in failed:
            continue
        so = os.path.join(VDIR, name, "libinferflow_amd.so")
        objs = [os.path.join(VDIR, name, u + ".o") for u in UNITS]
        subprocess.check_call([hipcc, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", so] + objs + others + ["-L/opt/rocm/lib", "-lrccl"])
        open(os.path.join(VDIR, name, "flags.txt"), "w").write(variants[name] + "\n")
        print("built", so)


def run(steps):
    out_path = os.path.join(ROOT, "gpurun_out", "sweep.jsonl")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "a") as f:
        for name in sorted(os.listdir(VDIR)):
            so = os.path.join(VDIR, name, "libinferflow_amd.so")
            if not os.path.exists(so):
                continue
            env = dict(os.environ, IFA_LIB=so)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "8", "--no-cpu-baseline",
                                "--prefill-lens", ""] + BENCH_ARGS, env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(name, "FAILED", r.stderr[-500:])
                continue
            j = json.loads(line[-1])
            k = j.get("kernels", {})
            rec = {"variant": name, "flags": open(os.path.join(VDIR, name, "flags.txt")).read().strip(), "tok_s": round(j["value"], 1),
                   "us": {n: round(v["us"], 2) for n, v in k.items()}}
            print(json.dumps(rec), flush=True)
            f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        v = dict(DEFAULT)
        extra = dict(a.split("=", 1) for a in sys.argv[2:])
        if extra:
            v = extra
        build(v)
    else:
        run(int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 96)
