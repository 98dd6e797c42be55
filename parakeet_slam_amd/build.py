"""Build the HIP shared library in-tree (gfx950 only).

    python -m parakeet_slam_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting ``parakeet_slam_amd/libparakeet_slam.so`` travels to the GPU box with
the snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

# Per-file extra flags.
# The ML kernels are register-bound: with MachineLICM on, the back end hoists constant materialisations and address
# arithmetic out of every loop and keeps them live around it (k_observe_sweep: 168 VGPRs + 96 B of scratch against 129 / none;
# k_step_fused: 121 against 99; k_assoc_grid<768, ...>: 168 + 44-52 B of scratch against 137-141 / none, and its 512-lane
# instances 207 against 121).
_NO_LICM = ["-mllvm", "-disable-machine-licm"]
EXTRA_FLAGS = {"pk_k_observe_ml.hip": _NO_LICM, "pk_k_step_pub.hip": _NO_LICM, "pk_k_cand_entries.hip": _NO_LICM, "pk_k_assoc.hip": _NO_LICM}

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libparakeet_slam.so")
OBJDIR = os.path.join(HERE, "csrc", "_obj")

HIP_SOURCES = ["pk_k_motion.hip", "pk_k_assoc.hip", "pk_k_observe.hip", "pk_k_observe_ml.hip", "pk_k_step_pub.hip", "pk_k_cand_entries.hip", "pk_k_dense.hip", "pk_k_resample.hip", "pk_k_grow.hip", "pk_api.hip"]
CXX_SOURCES = ["pk_rng.cpp"]  # host-only, no FMA contraction: must match NumPy/CPython bit for bit
HEADERS = [
    "pk_math.hpp", "pk_layout.hpp", "pk_kernels.hpp", "pk_philox.hpp", "pk_device.hpp", "pk_pub_math.hpp", "pk_pub_layout.hpp", "pk_k_step_duo.inl",
    os.path.join("..", "..", "include", "parakeet_slam.h"),
]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-Wall", "-Wno-unused-function",
    # a * b + c is fused only inside one expression (by the front end), not wherever the back end finds a
    # multiply feeding an add: the same inlined device function then rounds the same way in every kernel
    # (fused and two-kernel routes are required to agree bit for bit), whatever the surrounding control flow
    "-ffp-contract=on",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


CXX_FLAGS = ["-x", "c++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-Wall"]


def _cxx_objects(hipcc, verbose=True):
    """The host-only sources, compiled with THEIR flags (no FMA contraction: pk_rng.cpp must stay bit-exact with NumPy) --
    for the diagnostic one-call builds too, which used to push them through the HIP flags."""
    os.makedirs(OBJDIR, exist_ok=True)
    objs = []
    for src in CXX_SOURCES:
        path, obj = os.path.join(CSRC, src), os.path.join(OBJDIR, src + ".o")
        if _stale(obj, [path, os.path.abspath(__file__)]):
            cmd = [hipcc] + CXX_FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append("-Wl," + obj)  # (straight to the linker: hipcc would take a bare .o for another HIP source)
    return objs


def build_stamps(verbose=True):
    """Diagnostic build with in-kernel cycle stamps (-DPK_STAMPS) -> libparakeet_slam_stamps.so.
    Never loaded by the package; scripts/gpu_stamps.py uses it to read per-phase shares."""
    hipcc = _hipcc()
    out = os.path.join(HERE, "libparakeet_slam_stamps.so")
    srcs = [os.path.join(CSRC, x) for x in HIP_SOURCES]
    cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS["pk_k_observe_ml.hip"] + ["-DPK_STAMPS", "-shared", "-o", out] + srcs + _cxx_objects(hipcc, verbose)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_variant(name, defines, verbose=True):
    """Diagnostic A/B build: every source in one hipcc call with extra -D flags -> libpk_<name>.so (git-ignored, never loaded
    by the package; bench.py takes it through PK_BENCH_LIB, scripts/gpu_ab_lib.sh)."""
    hipcc = _hipcc()
    out = os.path.join(HERE, "libpk_%s.so" % name)
    srcs = [os.path.join(CSRC, x) for x in HIP_SOURCES]
    cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS["pk_k_observe_ml.hip"] + list(defines) + ["-shared", "-o", out] + srcs + _cxx_objects(hipcc, verbose)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in HIP_SOURCES + CXX_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers + [os.path.abspath(__file__)]):
            if src.endswith(".hip"):
                cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", path, "-o", obj]
            else:
                cmd = [hipcc] + CXX_FLAGS + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps())
    elif "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
