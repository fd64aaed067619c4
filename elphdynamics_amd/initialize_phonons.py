"""Host-side mirror of InitializePhonons.jl: the half-filling start configuration of the phonon field.

    init_phonons_half_filled_(model, rng)     InitializePhonons.jl:71-101 (Holstein), :11-69 (SSH)
    sample_qho(omega, beta, rng)              :106-114

`rng` is a numpy Generator drawn from in the reference's order (one integer in {-1, 0, 1} and one normal per Holstein site; one
normal per SSH bond phonon); Julia's Xoshiro stream itself cannot be reproduced here.
"""
import math

import numpy as np

from . import models


def sample_qho(omega, beta, rng):
    """Position of a quantum harmonic oscillator of frequency omega at inverse temperature beta (sigma = 1 for omega <= 0)."""
    sigma = 1.0 / math.sqrt(2.0 * omega * math.tanh(beta * omega / 2.0)) if omega > 0 else 1.0
    return sigma * rng.standard_normal()


def init_phonons_half_filled_(model, rng):
    L = model.Ltau
    if model.kind == models.HOLSTEIN:
        for site in range(model.Nsites):
            om, lam = model.omega[site], model.lam[site]
            x0 = lam / om ** 2 * int(rng.integers(-1, 2))                   # density 0, 1 or 2 on the site (:92)
            model.x[site * L:(site + 1) * L] = x0 + sample_qho(om, model.beta, rng)
    else:
        names = model.phonon_names
        per_type = model.Nph // max(model.nph, 1)
        for ph in range(model.Nph):
            name = names[ph // per_type]                                     # the phonon's type = its bond definition (:36-38)
            x0 = sample_qho(model.omega[ph], model.beta, rng)
            if names.count(name) == 1:                                       # :47-50: offset only for phonon types of their own
                x0 -= 2.0 * model.alpha[ph] / model.omega[ph] ** 2
            model.x[ph * L:(ph + 1) * L] = x0
        model.x[:] = model.x[model.primary_field]                            # :63
    models.update_model_(model)
