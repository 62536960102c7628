"""Build libreadhip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m read_amd.build [--force] [--debug]

--debug builds read_amd/libreadhip_debug.so with -DREAD_DEBUG_KNOBS: the same library plus the attribution probes whose results
are invalid (read_tuning_set "conv_ablate", "conv_abl").  Only tools/ load it (READ_HIP_DEBUG=1); it is never the product.

hipcc cross-compiles without a GPU; the resulting read_amd/libreadhip.so travels with the
tree to the GPU box (it is git-ignored, never pip-installed).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libreadhip.so")
SOURCES = ["api_common.cpp", "splat.hip", "gather.hip", "conv.hip", "train.hip", "unet.cpp"]
DEBUG_SOURCES = ["probe.hip"]        # measurement probes (read_hip_debug.h): libreadhip_debug.so only, never the product
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-x", "hip",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall", "-Wno-unused-function"]
# the rasteriser's pixel assignment must be bit-exact fp32: no a*b+c contraction
# gather / train: fp32 atomicAdd as the hardware global_atomic_add_f32 (the default lowers it to a compare-and-swap loop)
PER_FILE = {"splat.hip": ["-ffp-contract=off"], "conv.hip": ["-fno-slp-vectorize"], "gather.hip": ["-munsafe-fp-atomics"],
            "train.hip": ["-munsafe-fp-atomics"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, debug: bool = False) -> str:
    OBJ = os.path.join(HERE, "csrc", "_obj_debug" if debug else "_obj")
    LIB = os.path.join(HERE, "libreadhip_debug.so" if debug else "libreadhip.so")
    FLAGS = globals()["FLAGS"] + (["-DREAD_DEBUG_KNOBS"] if debug else []) + [f"-D{d}" for d in os.environ.get("READ_EXTRA_DEFINES", "").split()]
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(ROOT, "include", "read_hip.h"), os.path.join(ROOT, "include", "read_hip_debug.h"), os.path.join(CSRC, "common.h")]
    SOURCES = globals()["SOURCES"] + (DEBUG_SOURCES if debug else [])
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src + ".o")
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + PER_FILE.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
