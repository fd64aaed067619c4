"""Build libelphgpu.so in-tree with hipcc for gfx950 (the only target)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("ELPH_LIB") or os.path.join(HERE, "libelphgpu.so")
SOURCES = ["kernels.hip", "cg_fast.hip", "cg_fast6.hip", "dft.hip", "dft_mfma.hip", "elph_api.hip", "hmc.hip", "greens.hip", "kpm_host.cpp"]
OBJDIR = os.path.join(HERE, "build")
HEADERS = [os.path.join(CSRC, "elph_internal.h"), os.path.join(CSRC, "cg_fast_impl.inc"), os.path.join(CSRC, "cg_fast_common.h"), os.path.join(CSRC, "host_pool.h"), os.path.join(HERE, "..", "include", "elph_gpu.h")]


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
    """Compile the HIP kernels + C-ABI into elphdynamics_amd/libelphgpu.so. Returns the path.
    One object per source (rebuilt only when it or a header is newer), compiled side by side, then one link."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in HEADERS)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc, *flags, "-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB
