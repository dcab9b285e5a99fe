"""Builds inferflow_amd/lib/libinferflow_amd.so (HIP kernels + C ABI) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the tree.
"""
import glob
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libinferflow_amd.so")
ARCH = "gfx950"

HIPCC_FLAGS = [
    "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
    # exact-rounding parity with the host-compiled reference codecs
    "-ffp-contract=off",
    # leading scalar kernel arguments arrive in SGPRs at wave launch (gfx940+), see k_dec_gemv
    "-mllvm", "-amdgpu-kernarg-preload-count=8",
    "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


HOST = os.path.join(_HERE, "host")
BIN_DIR = os.path.join(_HERE, "bin")
CLI_PATH = os.path.join(BIN_DIR, "ifa_llm_inference")
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def host_sources():
    """The C++ InferenceEngine facade (plain g++: it reaches the GPU only through the C ABI)."""
    return sorted(s for s in glob.glob(os.path.join(HOST, "*.cc")) if not s.endswith("_main.cc"))


def source_hash():
    """Identity of the kernel sources a profile was taken with (bench.py refuses PMC numbers of another build)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_INC_RE = None


def local_deps(src, _seen=None):
    """The csrc / include headers a translation unit pulls in (recursive scan of #include "..."), so that a header edit only
    rebuilds the units that see it (the per-format GEMV units take ~2 min each)."""
    global _INC_RE
    import re
    if _INC_RE is None:
        _INC_RE = re.compile(r'^\s*#\s*include\s*"([^"]+)"', re.M)
    seen = _seen if _seen is not None else set()
    try:
        text = open(src, errors="replace").read()
    except OSError:
        return seen
    for name in _INC_RE.findall(text):
        for d in (os.path.dirname(src), CSRC, os.path.join(_HERE, "..", "include")):
            cand = os.path.normpath(os.path.join(d, name))
            if os.path.exists(cand):
                if cand not in seen:
                    seen.add(cand)
                    local_deps(cand, seen)
                break
    return seen


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    if not os.path.exists(CLI_PATH):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))
    deps += glob.glob(os.path.join(HOST, "*.cc")) + glob.glob(os.path.join(HOST, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, jobs=None):
    """Compile every translation unit to an object (in parallel) and link the .so."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libinferflow_amd.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdr_t = max([os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h"))
                 + glob.glob(os.path.join(_HERE, "..", "include", "*.h"))] + [0])
    procs = []
    objs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        dep_t = max([os.path.getmtime(h) for h in local_deps(src)] + [0])
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > dep_t):
            continue
        cmd = [hipcc] + HIPCC_FLAGS + ["-I", os.path.join(_HERE, "..", "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    inc = os.path.join(_HERE, "..", "include")
    host_hdr_t = max([os.path.getmtime(h) for h in glob.glob(os.path.join(HOST, "*.h")) + glob.glob(os.path.join(inc, "*.h"))] + [0])
    cxx = shutil.which("g++") or hipcc
    for src in host_sources():
        obj = os.path.join(obj_dir, "host_" + os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > host_hdr_t):
            continue
        cmd = [cxx] + HOST_FLAGS + ["-I", inc, "-I", HOST, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((src, out.decode(errors="replace")))
        elif verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join("== %s ==\n%s" % f for f in failed))
    link = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if any(os.path.basename(s).startswith("ifa_comm") for s in sources()):
        link += ["-L/opt/rocm/lib", "-lrccl"]
    subprocess.check_call(link)
    # the llm_inference-style driver, linked against the library next to it
    os.makedirs(BIN_DIR, exist_ok=True)
    subprocess.check_call([cxx] + HOST_FLAGS + ["-I", inc, "-I", HOST, os.path.join(HOST, "llm_inference_main.cc"), "-o", CLI_PATH,
                           "-L", LIB_DIR, "-linferflow_amd", "-Wl,-rpath,$ORIGIN/../lib"])
    subprocess.check_call([cxx] + HOST_FLAGS + ["-I", inc, "-I", HOST, os.path.join(HOST, "perplexity_main.cc"), "-o",
                           os.path.join(BIN_DIR, "ifa_perplexity"), "-L", LIB_DIR, "-linferflow_amd", "-Wl,-rpath,$ORIGIN/../lib"])
    subprocess.check_call([cxx] + HOST_FLAGS + ["-I", inc, "-I", HOST, os.path.join(HOST, "inferflow_service_main.cc"), "-o",
                           os.path.join(BIN_DIR, "ifa_service"), "-L", LIB_DIR, "-linferflow_amd", "-Wl,-rpath,$ORIGIN/../lib"])
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
