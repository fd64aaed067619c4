"""Oracle (numpy restatement of GreensFunctions.jl setup!/convolve!) against the golden direct-sum correlations."""
import numpy as np
import pytest

from conftest import golden
from oracle.greens import EstimateGreensFunction

NAMES = ["GD0", "GD0_GD0", "GDD_G00", "GD0_G0D"]


def make(g):
    est = EstimateGreensFunction(int(g["Ltau"]), int(g["norb"]), int(g["L1"]), int(g["L2"]), 1, nv=g["R"].shape[0])
    est.R[:], est.MinvR[:] = g["R"], g["MinvR"]
    return est


@pytest.mark.parametrize("name", ["greens_sq4_L8.npz", "greens_hc3_L6.npz"])
def test_oracle_setup_matches_direct_correlations(name):
    g = golden(name)
    est = make(g)
    for (n1, n2) in [(1, 2), (1, 3), (2, 3)]:
        est.setup(n1, n2)
        for nm in NAMES:
            got = getattr(est, nm).reshape(-1, order="F")
            ref = g["%s_%d%d" % (nm, n1, n2)]
            scale = np.abs(ref).max()
            assert np.abs(got.imag).max() < 1e-14 * scale
            assert np.abs(got.real - ref).max() < 1e-13 * scale, (nm, n1, n2)


def test_oracle_measure_indexing_and_symmetries():
    g = golden("greens_hc3_L6.npz")
    est = make(g)
    est.setup(1, 2)
    L = est.L
    G = g["GD0_12"].reshape((2 * L, 2, 2, 3, 3, 1), order="F")
    # measure_GΔ0(l1,l2,l3,o1,o2,τ) = GΔ0[mod1(τ+1,2L), o2, o1, l1+1, l2+1, l3+1]  (GreensFunctions.jl:293-298)
    assert abs(est.measure_GD0(2, 1, 0, 1, 2, 3) - G[3, 1, 0, 2, 1, 0]) < 1e-13
    assert abs(est.measure_GD0(0, 0, 0, 2, 2, 2 * L) - G[0, 1, 1, 0, 0, 0]) < 1e-13
    # antiperiodic in τ → τ+L for G, periodic for the products
    assert np.allclose(est.GD0[L:], -est.GD0[:L], atol=1e-14)
    assert np.allclose(est.GD0_GD0[L:], est.GD0_GD0[:L], atol=1e-14)
    # estimate(): product of one solution element and one noise element (:334-346)
    assert est.estimate(2, 3, 4, 1, 1) == est.MinvR[0][(2 - 1) * L + 3] * est.R[0][(3 - 1) * L + 0]
    assert est.estimate(2, 3, 4, 1, 2) == est.MinvR[1][(2 - 1) * L + 3] * est.R[1][(3 - 1) * L + 0]
