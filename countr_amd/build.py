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


def build(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link libcountr_hip.so.  An object is reused only if the record written when it was
    compiled (build/<src>.o.sha256: content hash of the source, of every shared header and of the flags) still matches -- content, not
    mtime, so a snapshot copy or a checkout cannot make a stale object look fresh.  COUNTR_BUILD_FORCE=1 (or force=True / --force)
    recompiles everything.  Prints how many objects were compiled."""
    force = force or os.environ.get("COUNTR_BUILD_FORCE", "0") == "1"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    reused = 0
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        want = _digest([src] + hdrs, " ".join(cmd[1:-3]))
        rec = obj + ".sha256"
        if not force and os.path.exists(obj) and os.path.exists(rec) and open(rec).read().strip() == want:
            reused += 1
            continue   # (gemm.hip alone takes ~90 s)
        if os.path.exists(rec):
            os.remove(rec)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, rec, want, subprocess.Popen(cmd)))
    for src, rec, want, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % src)
        open(rec, "w").write(want + "\n")
    lrec = LIB + ".sha256"
    lwant = _digest([o + ".sha256" for o in objs])
    if procs or not os.path.exists(LIB) or not os.path.exists(lrec) or open(lrec).read().strip() != lwant:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        open(lrec, "w").write(lwant + "\n")
    if verbose:
        print("build_mode: %s -- %d of %d objects compiled, %d reused after a content-hash check (source + headers + flags)"
              % ("full" if reused == 0 else "incremental", len(procs), len(objs), reused), flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
