"""The C ABI called from plain C (tests/abi_c/abi_smoke.c: gcc, links libelphgpu.so, reads a flat binary export of the golden vectors,
makes the calls of julia/ElPhGPU.jl in the order a model's life makes them) — the proof that the boundary is usable from a host
language that is not Python.  CPU: the program builds, links, loads the library and refuses to compute without a GPU.  GPU: every
value against the golden vectors."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "abi_c", "abi_smoke.c")
FIX = os.path.join(ROOT, "tests", "abi_c", "holstein_sq4_L8.bin")


def _build(tmp_path):
    from elphdynamics_amd import _lib
    _lib.load()                                            # (fails loudly when the library has not been built)
    libdir = os.path.join(ROOT, "elphdynamics_amd")
    exe = str(tmp_path / "abi_smoke")
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe, "-L", libdir, "-lelphgpu",
           f"-Wl,-rpath,{libdir}", "-lm"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def test_fixture_binary_is_the_golden_data():
    """tests/abi_c/holstein_sq4_L8.bin is a byte-level export of tests/golden/*.npz (export_fixture.py): spot-check both ends."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "holstein_sq4_L8.npz"))
    k = np.load(os.path.join(ROOT, "tests", "golden", "kpm_sq4_L8.npz"))
    raw = open(FIX, "rb").read()
    assert raw[:8] == b"ELPHFIX1"
    hdr = np.frombuffer(raw, dtype="<i8", count=3, offset=8)
    assert hdr.tolist() == [int(g["N"]), int(g["Ltau"]), g["table"].shape[0]]
    tab = np.frombuffer(raw, dtype="<i8", count=2 * int(hdr[2]), offset=8 + 24 + 64)
    assert np.array_equal(tab, g["table"].reshape(-1))
    ndim = int(hdr[0] * hdr[1])
    tail = np.frombuffer(raw[-8 * ndim:], dtype="<f8")
    assert np.array_equal(tail, k["vout"])


def test_c_program_links_and_refuses_without_a_gpu(tmp_path):
    from elphdynamics_amd import _lib
    if _lib.load().elph_device_count() > 0:
        pytest.skip("a GPU is visible here: the GPU test runs the program in full")
    exe = _build(tmp_path)
    p = subprocess.run([exe, FIX], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "no CPU path" in p.stderr and "libelphgpu abi=2 src=" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-500:])


@pytest.mark.gpu
def test_c_abi_smoke_program(tmp_path):
    """create -> update_model! -> mulM / mulMT / mulMTM -> ldiv! (+ the flag logic) -> KPM create / setup / orders / apply -> preconditioned
    ldiv! -> destroy, from C, against the golden vectors: what a `ccall` wrapper sees."""
    exe = _build(tmp_path)
    p = subprocess.run([exe, FIX], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ABI SMOKE OK"), (p.returncode, p.stdout[-3000:], p.stderr[-2000:])
    assert "MISMATCH" not in p.stdout
