"""Builds libomh.so (the gfx950 HIP kernel library) in-tree with hipcc.

    python omnihuman-1-hack_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  Objects and the shared
library land in ``omnihuman-1-hack_amd/lib/`` (git-ignored, shipped to the GPU
box by gpurun).  A source/flag hash is stored beside the library so rebuilds
only happen when something changed.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libomh.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-Wno-unused-result", f"-I{INCLUDE}"]
# what the digests hash: the flags WITHOUT the absolute include path (the tree is copied to the GPU box and may be
# relocated; a moved tree is not a stale build)
_FLAGS_KEY = " ".join("-I<include>" if f.startswith("-I") else f for f in FLAGS)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the gfx950 kernel library cannot be built")


def _generate():
    """Generated instruction streams: csrc/gen_*.py -> csrc/*_asm.inc (rewritten only when the text changes)."""
    for gname, oname in (("gen_attn_w64.py", "attention_w64_asm.inc"), ("gen_gemm_w64.py", "gemm_w64_asm.inc"),
                         ("gen_conv_w64.py", "conv_w64_asm.inc"), ("gen_gemm_tn_w64.py", "gemm_tn_w64_asm.inc"),
                         ("gen_attn_bwd_w64.py", "attention_bwd2_asm.inc")):
        gen, out = os.path.join(CSRC, gname), os.path.join(CSRC, oname)
        if os.path.exists(gen):
            txt = subprocess.run([sys.executable, gen], check=True, capture_output=True, text=True).stdout
            if not os.path.exists(out) or open(out).read() != txt:
                tmp = out + f".tmp{os.getpid()}"
                with open(tmp, "w") as fh:
                    fh.write(txt)
                os.replace(tmp, out)             # atomic: a concurrent reader never sees a truncated stream


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256(_FLAGS_KEY.encode())
    # (the generators are part of the digest: editing one makes the build stale, and the streams are regenerated
    # under the build lock — not on every import)
    for f in _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                                 if f.endswith((".h", ".inc")) or (f.startswith("gen_") and f.endswith(".py"))) + \
            [os.path.join(INCLUDE, "omh.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())     # not the path: the tree is copied to the GPU box
            h.update(fh.read())
    return h.hexdigest()


def up_to_date() -> bool:
    """True when libomh.so exists and its stamp equals the digest of the sources as they are now."""
    stamp = os.path.join(LIBDIR, "libomh.sha256")
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _digest()


def have_hipcc() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def _obj_digest(src):
    """Digest of ONE translation unit: its source, the shared headers and the flags (so that editing one .hip
    file recompiles one object, not nine)."""
    h = hashlib.sha256(_FLAGS_KEY.encode())
    stem = os.path.basename(src)[:-4]
    for f in [src] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                            if f.endswith(".h") or (f.endswith(".inc") and f.startswith(stem))) + \
            [os.path.join(INCLUDE, "omh.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libomh.sha256")
    if not force and up_to_date():
        return LIB
    # one builder at a time (torchrun starts N ranks that all import the package): the others wait on the lock and
    # then find the library up to date
    import fcntl
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date():
            return LIB
        hipcc = _hipcc()
        _generate()
        dig = _digest()

        def compile_one(src):
            obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
            ostamp = obj + ".sha256"
            od = _obj_digest(src)
            if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read().strip() == od:
                return obj
            cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
            if verbose:
                print("[omh build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            with open(ostamp, "w") as fh:
                fh.write(od)
            return obj

        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            objs = list(ex.map(compile_one, _sources()))
        tmp = LIB + ".tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
        if verbose:
            print("[omh build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp, LIB)                     # atomic: a concurrent dlopen never sees a half-written file
        with open(stamp, "w") as fh:
            fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
