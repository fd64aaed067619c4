#!/usr/bin/env python3
"""Which workgroups of a k_cg_wg launch started where and when (diagnostic build: tools/build_wg_arrive.sh first).
usage: ELPH_LIB=elphdynamics_amd/libelphgpu_arrive.so [ELPH_WG_T=2 ELPH_WG_W=4 ELPH_WG_TIMEOUT_MS=300] python3 tools/diag_wg_arrive.py C 17"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs
from elphdynamics_amd._lib import check
lib = _lib.load()
tag, nr = sys.argv[1], int(sys.argv[2])
m = configs.make_model(tag, tol=1e-5)
_, Bs = configs.rhs(m, nr)
us, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
check(lib.elph_bench_wg_info(m._h, nr, C.byref(us), C.byref(T), C.byref(W), C.byref(G)))
grid = 8 * ((nr + 7) // 8) * G.value
print(f"{tag} nrhs={nr}: T={T.value} W={W.value} G={G.value} grid={grid}")
lib.elph_debug_wg_arrive(None, 0, 1)
ms = C.c_double()
rc = 0
for reps in [int(v) for v in os.environ.get("REPS", "200,1000").split(",")]:
    check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
    rc = lib.elph_bench_run(m._h, 9, nr, reps, 0, C.byref(ms))
    print("run of", reps, "iterations: rc", rc, lib.elph_last_error().decode() if rc else f"{1e3*ms.value/reps:.2f} us/iter", flush=True)
    if rc:
        break
out = (C.c_ulonglong * (4 * grid))()
assert lib.elph_debug_wg_arrive(out, grid, 0) == 0
a = np.array(out[:]).reshape(grid, 4)
arrived = (a[:, 0] > 0) if (a[:, 0] > 0).any() else np.ones(grid, dtype=bool)      # (a build without the start records, ARRIVE_EXTRA=-DELPH_WG_ARRIVE_NOSTART: everything counts as arrived)
t0 = a[arrived, 2].min() if arrived.any() else 0
print("arrived", int(arrived.sum()), "of", grid)
xcc = (a[:, 0] - 1) & 0xF
ok = True
for b in range(grid):
    if arrived[b] and xcc[b] != (b & 7):
        ok = False
print("blockIdx % 8 == XCC_ID for every arrived workgroup:", ok)
for x in range(8):
    idx = [b for b in range(grid) if (b & 7) == x]
    arr = [b for b in idx if arrived[b]]
    miss = [b >> 3 for b in idx if not arrived[b]]
    cu = {}
    for b in arr:
        hw = int(a[b, 1]); key = (hw >> 8) & 0xF, (hw >> 13) & 0x7     # CU_ID[11:8], SE_ID[15:13] (gfx9 HW_ID layout)
        cu[key] = cu.get(key, 0) + 1
    prog = {}
    for b in arr:
        prog.setdefault((b >> 3) // G.value, []).append(int(a[b, 3]))
    print("   polls that gave up, per team [(workgroup in team, iteration, wave mask)]: " +
          "; ".join(f"team {t}: " + str([(i, v >> 8, v & 255) for i, v in enumerate(vs) if v][:8]) for t, vs in sorted(prog.items())))
    late = sorted(((int(a[b, 2]) - int(t0)) / 100.0, b >> 3) for b in arr)[-3:]
    print(f"XCD {x}: arrived {len(arr)} of {len(idx)}; never started (index in XCD): {miss[:24]}; workgroups per CU max {max(cu.values()) if cu else 0} on {len(cu)} CUs; latest starts (us, index) {late}")
m.close()
