"""elphdynamics_amd — MI355X (gfx950) fermion-force solver for ElPhDynamics.

The product is libelphgpu.so (hand-written HIP, C ABI in include/elph_gpu.h); this package is its
host-side mirror of the reference's operator API (Models.jl / IterativeSolvers.jl /
KPMPreconditioners.jl / FourierAcceleration.jl) plus the integer set-up code (lattice.py).
There is no CPU compute path: importing works anywhere, but every operator needs the built
library and a gfx950 device.
"""
from . import lattice, synth  # noqa: F401
from .lattice import Lattice  # noqa: F401

__all__ = ["lattice", "synth", "Lattice", "models", "preconditioners", "configs", "hmc", "greens", "io"]


def __getattr__(name):
    # models / preconditioners / configs pull in the ctypes binding lazily
    if name in ("models", "preconditioners", "configs", "hmc", "langevin", "greens", "io", "sharded", "dist"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
