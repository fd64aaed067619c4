"""Build libelphgpu.so in-tree with hipcc for gfx950 (the only target)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("ELPH_LIB") or os.path.join(HERE, "libelphgpu.so")
SOURCES = ["kernels.hip", "cg_fast.hip", "cg_fast6.hip", "cg_wg.hip", "pcg_wg.hip", "shard.hip", "kpm_dev.hip", "dft.hip", "dft_mfma.hip", "dft_big.hip", "elph_api.hip", "hmc.hip", "greens.hip", "kpm_host.cpp"]
OBJDIR = os.path.join(HERE, "build")
HEADERS = [os.path.join(CSRC, "elph_internal.h"), os.path.join(CSRC, "cg_fast_impl.inc"), os.path.join(CSRC, "cg_fast_common.h"), os.path.join(CSRC, "cg_wg_dev.h"), os.path.join(CSRC, "kpm_sq_dev.h"), os.path.join(HERE, "..", "include", "elph_gpu.h")]


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


# A/B build for tests/test_gpu_parity.py::test_lds_sync_build_is_bit_identical: the lane-program kernels order their
# private-LDS traffic with a compiler barrier only (cg_fast_common.h, WAVE_LDS_ORDER); this variant compiles the same two
# translation units with a real s_waitcnt + s_barrier per colour (-DELPH_LDS_SYNC) — a compiler reordering regression
# would show as a difference between the two libraries.  Every other object is shared with the product build.
LIB_LDSSYNC = os.path.join(HERE, "libelphgpu_ldssync.so")
LDSSYNC_SOURCES = ("cg_fast.hip", "cg_fast6.hip", "cg_wg.hip")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, lds_sync_variant=True):
    """Compile the HIP kernels + C-ABI into elphdynamics_amd/libelphgpu.so (and the ELPH_LDS_SYNC A/B variant next to it).
    Returns the path of the product library.  One object per source (rebuilt only when it or a header is newer), compiled
    side by side, then one link per library."""
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    want_main = force or _stale(LIB, deps)
    want_var = lds_sync_variant and (force or _stale(LIB_LDSSYNC, deps))
    if not want_main and not want_var:
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in HEADERS)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    jobs, objs, objs_var = [], [], []

    def compile_if_stale(src, obj, extra=()):
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [hipcc, *flags, *extra, "-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))

    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s + ".o")
        objs.append(obj)
        compile_if_stale(src, obj)
        if lds_sync_variant and s in LDSSYNC_SOURCES:
            obj_v = os.path.join(OBJDIR, s + ".ldssync.o")
            objs_var.append(obj_v)
            compile_if_stale(src, obj_v, ("-DELPH_LDS_SYNC",))
        else:
            objs_var.append(obj)
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for lib, ob, want in ((LIB, objs, True), (LIB_LDSSYNC, objs_var, lds_sync_variant)):
        if not want:
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *ob, "-o", lib]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":      # python -m elphdynamics_amd.build [--force]
    import sys
    print(build_library(force="--force" in sys.argv[1:], verbose="-v" in sys.argv[1:]))
