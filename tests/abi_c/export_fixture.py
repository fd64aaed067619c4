"""Writes tests/abi_c/holstein_sq4_L8.bin: the golden vectors of tests/golden/holstein_sq4_L8.npz + kpm_sq4_L8.npz (data produced by
tests/golden/make_golden.py, the independent dense numpy/scipy restatement) as ONE flat little-endian file a C program reads with
fread — no numpy, no zip, no Python in the consumer (tests/abi_c/abi_smoke.c).

layout:  char magic[8] = "ELPHFIX1"; int64 N, Ltau, Nbonds; double dtau, kpm_buf, kpm_c1, kpm_c2, e_min, e_max, lam_lo, lam_hi;
         int64 table[2*Nbonds] (the ABI's 2 x Nbonds column-major, 1-based, checkerboard order); int64 orders[(Ltau+1)/2];
         double cosht[Nbonds], sinht[Nbonds], lam[N], lam2[N], mu[N];
         double x, E, v, Mv, MTv, MTMv, b, xsol, kpm_vin, kpm_vout   (Ndim = N*Ltau each, reference layout: tau fastest)
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(HERE, "..", "golden", "holstein_sq4_L8.npz"))
k = np.load(os.path.join(HERE, "..", "golden", "kpm_sq4_L8.npz"))
N, L, nb = int(g["N"]), int(g["Ltau"]), g["table"].shape[0]
with open(os.path.join(HERE, "holstein_sq4_L8.bin"), "wb") as f:
    f.write(b"ELPHFIX1")
    f.write(struct.pack("<3q", N, L, nb))
    f.write(struct.pack("<8d", float(g["dtau"]), float(k["buf"]), float(k["c1"]), float(k["c2"]), float(k["e_min"]), float(k["e_max"]),
                        float(k["lam_lo"]), float(k["lam_hi"])))
    f.write(np.ascontiguousarray(g["table"], dtype="<i8").tobytes())          # (Nbonds, 2) row-major == 2 x Nbonds column-major
    f.write(np.ascontiguousarray(k["orders"], dtype="<i8").tobytes())
    for name in ("cosht", "sinht", "lam", "lam2", "mu", "x", "E", "v", "Mv", "MTv", "MTMv", "b", "xsol"):
        f.write(np.ascontiguousarray(g[name], dtype="<f8").tobytes())
    for name in ("vin", "vout"):
        f.write(np.ascontiguousarray(k[name], dtype="<f8").tobytes())
print("wrote holstein_sq4_L8.bin:", os.path.getsize(os.path.join(HERE, "holstein_sq4_L8.bin")), "bytes")
