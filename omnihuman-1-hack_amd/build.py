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


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the gfx950 kernel library cannot be built")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in _sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
            [os.path.join(INCLUDE, "omh.h")]:
        with open(f, "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libomh.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print("[omh build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print("[omh build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
