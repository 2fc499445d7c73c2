"""Build libcountr_hip.so (all HIP kernels + the C ABI) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box.
The .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcountr_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


# per-file flags on top of the common ones (reasons in the file headers)
EXTRA_FLAGS = {"flash_attn_fwd.hip": ["-fno-slp-vectorize", "-fno-honor-nans"]}


def _digest(paths, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


# the library is built twice from the same sources: the 16-bit storage / matrix-operand type is bfloat16 in libcountr_hip.so
# (precision="bf16") and IEEE fp16 in libcountr_hip_f16.so (precision="fp16", -DCOUNTR_HALF_FP16=1: csrc/common.hpp)
LIB_F16 = os.path.join(HERE, "libcountr_hip_f16.so")
VARIANTS = (("", LIB, []), ("f16", LIB_F16, ["-DCOUNTR_HALF_FP16=1"]))


def build(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link libcountr_hip.so + libcountr_hip_f16.so.  An object is reused only if the record
    written when it was compiled (build/<variant>/<src>.o.sha256: content hash of the source, of every shared header and of the flags)
    still matches -- content, not mtime, so a snapshot copy or a checkout cannot make a stale object look fresh.  COUNTR_BUILD_FORCE=1
    (or force=True / --force) recompiles everything.  Prints how many objects were compiled."""
    force = force or os.environ.get("COUNTR_BUILD_FORCE", "0") == "1"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    procs, reused, total = [], 0, 0
    links = []
    jobs = int(os.environ.get("COUNTR_BUILD_JOBS", "0")) or max(2, (os.cpu_count() or 4))
    for tag, lib, extra in VARIANTS:
        bdir = os.path.join(HERE, "build", tag) if tag else os.path.join(HERE, "build")
        os.makedirs(bdir, exist_ok=True)
        objs, fresh = [], False
        for src in sources():
            obj = os.path.join(bdir, os.path.basename(src) + ".o")
            objs.append(obj)
            total += 1
            cmd = ([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + extra
                   + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
            want = _digest([src] + hdrs, " ".join(cmd[1:-3]))
            rec = obj + ".sha256"
            if not force and os.path.exists(obj) and os.path.exists(rec) and open(rec).read().strip() == want:
                reused += 1
                continue
            if os.path.exists(rec):
                os.remove(rec)
            while sum(1 for _s, _r, _w, p_ in procs if p_.poll() is None) >= jobs:      # (30 hipcc processes at once starve an 8-core box)
                import time
                time.sleep(0.2)
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, rec, want, subprocess.Popen(cmd)))
            fresh = True
        links.append((lib, objs, fresh))
    for src, rec, want, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
        open(rec, "w").write(want + "\n")
    for lib, objs, fresh in links:
        lrec = lib + ".sha256"
        lwant = _digest([o + ".sha256" for o in objs])
        if fresh or not os.path.exists(lib) or not os.path.exists(lrec) or open(lrec).read().strip() != lwant:
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            open(lrec, "w").write(lwant + "\n")
    if verbose:
        print("build_mode: %s -- %d of %d objects compiled (2 libraries x %d sources), %d reused after a content-hash check (source + headers + flags)"
              % ("full" if reused == 0 else "incremental", len(procs), total, total // 2, reused), flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
