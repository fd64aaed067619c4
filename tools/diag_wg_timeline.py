#!/usr/bin/env python3
"""What ONE launch of k_cg_wg with few iterations spends outside them (diagnostic build: tools/build_wg_arrive.sh first): per workgroup the
wall clock at its start, at the top and the end of its first iteration, when it saw `done` and after its last stores; per CU the gap between
one workgroup's end and the next one's start.
usage: ELPH_LIB=elphdynamics_amd/libelphgpu_arrive.so python3 tools/diag_wg_timeline.py C 288 20"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from elphdynamics_amd import _lib, configs
from elphdynamics_amd._lib import check
lib = _lib.load()
tag, nr, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = configs.make_model(tag, tol=1e-5)
_, Bs = configs.rhs(m, nr)
us, T, W, G = C.c_int(), C.c_int(), C.c_int(), C.c_int()
check(lib.elph_bench_wg_info(m._h, nr, C.byref(us), C.byref(T), C.byref(W), C.byref(G)))
grid = 8 * ((nr + 7) // 8) * G.value
assert grid <= 4096
ms = C.c_double()
check(lib.elph_bench_prepare(m._h, 1, nr, _lib.dptr(np.ascontiguousarray(Bs))))
check(lib.elph_bench_run(m._h, 9, nr, 50, 0, C.byref(ms)))
for rep in range(2):
    lib.elph_debug_wg_arrive(None, 0, 1)
    check(lib.elph_bench_prepare(m._h, 1, nr, None))
    check(lib.elph_bench_run(m._h, 9, nr, K, 0, C.byref(ms)))
a = (C.c_ulonglong * (4 * grid))()
t = (C.c_ulonglong * (4 * grid))()
assert lib.elph_debug_wg_arrive(a, grid, 0) == 0 and lib.elph_debug_wg_timeline(t, grid) == 0
a = np.array(a[:], dtype=np.int64).reshape(grid, 4)
t = np.array(t[:], dtype=np.int64).reshape(grid, 4)
start, top, it1, done, end = a[:, 2], t[:, 0], t[:, 1], t[:, 2], t[:, 3]
t0 = start.min()
u = lambda x: x / 100.0        # wall_clock64 ticks (100 MHz) -> us
print(f"{tag} nrhs={nr} K={K}: T={T.value} W={W.value} G={G.value} grid={grid}; launch {1e3 * ms.value:.1f} us by events; first start .. last end {u(end.max() - t0):.1f} us")
print(f"  per workgroup, mean (min .. max) in us:")
for name, v in (("start -> top of iteration 1 (loads, tables)", u(top - start)), ("iteration 1 (incl. waiting for the team)", u(it1 - top)),
                (f"iterations 2 .. {K} each", u(done - it1) / max(1, K - 1)), ("`done` -> after the stores of x", u(end - done)), ("start -> end", u(end - start))):
    print(f"     {name:48s} {v.mean():7.2f}  ({v.min():.2f} .. {v.max():.2f})")
# per CU: order its workgroups by start, gap between end of one and start of the next
key = (a[:, 0] - 1) * (1 << 32) + (a[:, 1] & 0xFFFFFF00)          # XCC_ID, then HW_ID without the wave / SIMD bits
gaps, per_cu = [], {}
for b in range(grid):
    per_cu.setdefault(int(key[b]), []).append(b)
for k, bs in per_cu.items():
    bs.sort(key=lambda b: start[b])
    for x, y in zip(bs[:-1], bs[1:]):
        gaps.append(u(start[y] - end[x]))
gaps = np.array(gaps) if gaps else np.zeros(1)
print(f"  {len(per_cu)} CUs used, {grid / max(1, len(per_cu)):.2f} workgroups each; gap between a workgroup's end and the next one's start on its CU: "
      f"mean {gaps.mean():.2f} us (min {gaps.min():.2f}, max {gaps.max():.2f})")
rounds = np.sort(u(start - t0))
print("  starts (us after the first), deciles:", " ".join(f"{rounds[int(q * (grid - 1) / 10)]:.0f}" for q in range(11)))
team_skew = []
for tq in range(grid // (8 * G.value)):
    for x in range(8):
        bs = [((tq * G.value + g) << 3) | x for g in range(G.value)]
        team_skew.append(u(start[bs].max() - start[bs].min()))
print(f"  start skew within a team: mean {np.mean(team_skew):.2f} us, max {np.max(team_skew):.2f}")
m.close()
