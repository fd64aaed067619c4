"""CPU oracle for the stochastic Green's-function estimator — numpy restatement of
GreensFunctions.jl, statement for statement, with numpy.fft standing in for FFTW's plan_fft/plan_ifft
(forward unnormalised, inverse scaled by 1/n — the same conventions).

TEST INFRASTRUCTURE ONLY (imported by tests/ and __graft_entry__.smoke(); never by elphdynamics_amd/).
PARITY UNPINNED BY THE REFERENCE: it ships no fixture for this path; the pins are tests/golden/greens_*.npz —
direct (FFT-free) cross-correlation sums written by tests/golden/make_golden.py.

Arrays carry Julia's column-major index order: a 5-index array [2L, n_s, L1, L2, L3] is a numpy array of that
shape in Fortran order, so `A.reshape(-1, order="F")` is the reference's memory image.
"""
import numpy as np


def antiperiodic_copy(x, L):
    """GreensFunctions.jl:406-418: y = [x(1..L), -x(1..L)] per site column."""
    N = x.size // L
    xp = x.reshape((L, N), order="F")
    y = np.empty((2 * L, N), order="F")
    y[:L, :] = xp
    y[L:, :] = -xp
    return y


def periodic_product(y, x, L):
    """GreensFunctions.jl:424-440: z = [x.*y, x.*y] per site column."""
    N = x.size // L
    val = y.reshape((L, N), order="F") * x.reshape((L, N), order="F")
    z = np.empty((2 * L, N), order="F")
    z[:L, :] = val
    z[L:, :] = val
    return z


class EstimateGreensFunction:
    """GreensFunctions.jl:23-196 (state) — R, M⁻¹R are (n_v, NL) here (row = Julia column)."""

    def __init__(self, L, norbits, L1, L2, L3, nv=2):
        self.nv = max(2, nv)
        self.L, self.ns, self.L1, self.L2, self.L3 = L, norbits, L1, L2, L3
        self.N = norbits * L1 * L2 * L3
        self.NL = self.N * L
        self.R = np.zeros((self.nv, self.NL))
        self.MinvR = np.zeros((self.nv, self.NL))
        shp = (2 * L, norbits, norbits, L1, L2, L3)
        self.GD0 = np.zeros(shp, dtype=complex, order="F")
        self.GDD_G00 = np.zeros(shp, dtype=complex, order="F")
        self.GD0_GD0 = np.zeros(shp, dtype=complex, order="F")
        self.GD0_G0D = np.zeros(shp, dtype=complex, order="F")

    def _grid(self, y2):
        return y2.reshape((2 * self.L, self.ns, self.L1, self.L2, self.L3), order="F")

    def convolve(self, ab, a, b):
        """GreensFunctions.jl:351-400: ab += ifft( fft(a)[ω,s₂,k] · fft(b)[-ω,s₁,-k] / V )."""
        L, ns, L1, L2, L3 = self.L, self.ns, self.L1, self.L2, self.L3
        ap = np.fft.fftn(self._grid(a), axes=(0, 2, 3, 4))                   # pfft over dims (1,3,4,5)  :364
        bp = np.fft.fftn(self._grid(b), axes=(0, 2, 3, 4))                   # :368
        V = 2 * L * self.N / ns                                              # :371
        nw = (-np.arange(2 * L)) % (2 * L)                                   # mod1(-ω+2, 2L), 0-based  :383
        n1, n2, n3 = (-np.arange(L1)) % L1, (-np.arange(L2)) % L2, (-np.arange(L3)) % L3
        bneg = bp[nw][:, :, n1][:, :, :, n2][:, :, :, :, n3]                 # b′[nω, s₁, nk₁, nk₂, nk₃]
        abp = ap[:, :, None, :, :, :] * bneg[:, None, :, :, :, :] / V        # ab′[ω,s₂,s₁,k]            :384
        ab += np.fft.ifftn(abp, axes=(0, 3, 4, 5))                           # pifft over dims (1,4,5,6) :391-394

    def setup(self, n1, n2):
        """setup!(estimator, n₁, n₂) — GreensFunctions.jl:239-288 (n₁, n₂ 1-based)."""
        L = self.L
        self.n1, self.n2 = n1, n2
        x1, r1 = self.MinvR[n1 - 1], self.R[n1 - 1]
        x2, r2 = self.MinvR[n2 - 1], self.R[n2 - 1]
        self.x1, self.r1, self.x2, self.r2 = x1, r1, x2, r2
        for G in (self.GD0, self.GD0_GD0, self.GD0_G0D, self.GDD_G00):
            G[...] = 0.0
        a = (antiperiodic_copy(x1, L) + antiperiodic_copy(x2, L)) / np.sqrt(2.0)     # :263-265
        b = (antiperiodic_copy(r1, L) + antiperiodic_copy(r2, L)) / np.sqrt(2.0)     # :266-268
        self.convolve(self.GD0, a, b)
        self.convolve(self.GD0_GD0, periodic_product(x1, x2, L), periodic_product(r1, r2, L))   # :272-274
        self.convolve(self.GDD_G00, periodic_product(x2, r2, L), periodic_product(x1, r1, L))   # :277-279
        self.convolve(self.GD0_G0D, periodic_product(x1, r2, L), periodic_product(x2, r1, L))   # :282-284

    def _measure(self, G, l1, l2, l3, o1, o2, tau):
        return G[(tau % (2 * self.L)), o2 - 1, o1 - 1, l1, l2, l3]          # mod1(τ+1,2L), o₂, o₁, l+1  :297

    def measure_GD0(self, l1, l2, l3, o1, o2, tau):
        return self._measure(self.GD0, l1, l2, l3, o1, o2, tau)

    def measure_GD0_GD0(self, l1, l2, l3, o1, o2, tau):
        return self._measure(self.GD0_GD0, l1, l2, l3, o1, o2, tau)

    def measure_GDD_G00(self, l1, l2, l3, o1, o2, tau):
        return self._measure(self.GDD_G00, l1, l2, l3, o1, o2, tau)

    def measure_GD0_G0D(self, l1, l2, l3, o1, o2, tau):
        return self._measure(self.GD0_G0D, l1, l2, l3, o1, o2, tau)

    def estimate(self, i, j, tau2, tau1, sigma):
        """GreensFunctions.jl:334-346 (1-based i, j, τ; σ ∈ {1, 2} picks the pair member)."""
        m = (j - 1) * self.L + tau1 - 1
        n = (i - 1) * self.L + tau2 - 1
        if sigma == 1:
            return self.x1[n] * self.r1[m]
        if sigma == 2:
            return self.x2[n] * self.r2[m]
        raise ValueError("sigma")
