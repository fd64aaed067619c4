"""Build libelphgpu.so in-tree with hipcc for gfx950 (the only target)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("ELPH_LIB") or os.path.join(HERE, "libelphgpu.so")
SOURCES = ["kernels.hip", "cg_fast.hip", "dft.hip", "elph_api.hip", "hmc.hip", "kpm_host.cpp"]
HEADERS = [os.path.join(CSRC, "elph_internal.h"), os.path.join(HERE, "..", "include", "elph_gpu.h")]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libelphgpu.so cannot be built (ROCm toolchain required)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile the HIP kernels + C-ABI into elphdynamics_amd/libelphgpu.so. Returns the path."""
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-x", "hip",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
