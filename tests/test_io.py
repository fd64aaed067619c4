"""Phonon-configuration / matrix-dump file formats of the reference: byte-level format checks on
the CPU (stub model, no device), and round trips through the device model on the GPU."""
import types

import numpy as np
import pytest

from conftest import golden
from elphdynamics_amd import io as eio
from elphdynamics_amd import lattice as lat
from elphdynamics_amd import models


def _stub_holstein(norb, L1, L2, L3, Ltau):
    la = lat.Lattice(norb, L1, L2, L3)
    return types.SimpleNamespace(kind=models.HOLSTEIN, lattice=la, Ltau=Ltau, x=np.zeros(la.nsites * Ltau))


def _stub_ssh(nph_types, N, Ltau):
    defs = [dict(has_phonon=True)] * nph_types + [dict(has_phonon=False)]
    return types.SimpleNamespace(kind=models.SSH, Ltau=Ltau, Nph=nph_types * N, bond_definitions=defs,
                                 x=np.zeros(nph_types * N * Ltau))


@pytest.fixture
def no_device(monkeypatch):
    calls = []
    monkeypatch.setattr(models, "update_model_", lambda m: calls.append(m))
    return calls


def test_holstein_phonon_file_format(tmp_path, no_device):
    """Line order (l3, l2, l1, orbit, tau — HolsteinModels.jl:779-802), header, %.6f, 1-based orbit/tau, 0-based cells."""
    m = _stub_holstein(2, 3, 2, 1, 4)
    m.x = np.arange(m.x.size) * 0.12345678 - 1.5
    f = tmp_path / "ph.out"
    eio.write_phonons_(m, str(f))
    lines = f.read_text().split("\n")
    assert lines[0] == "L3 L2 L1 orbit tau x" and lines[-1] == "" and len(lines) == m.x.size + 2
    assert lines[1] == "0 0 0 1 1 -1.500000"
    assert lines[2] == "0 0 0 1 2 -1.376543"
    assert lines[5] == "0 0 0 2 1 %.6f" % m.x[4]                       # site 2 = orbit 2 of cell 0
    assert lines[9] == "0 0 1 1 1 %.6f" % m.x[8]                       # l1 runs before l2
    assert lines[1 + 3 * 2 * 4] == "0 1 0 1 1 %.6f" % m.x[24]
    # read back into a fresh model: values rounded to 6 decimals, update_model! called once (:849)
    m2 = _stub_holstein(2, 3, 2, 1, 4)
    eio.read_phonons_(m2, str(f))
    assert np.array_equal(m2.x, np.array([float("%.6f" % v) for v in m.x]))
    assert no_device == [m2]


def test_holstein_phonon_file_partial_and_shuffled(tmp_path, no_device):
    """read_phonons! assigns only the entries the file names, in any order (:824-846)."""
    m = _stub_holstein(1, 2, 2, 1, 3)
    m.x[:] = 7.0
    f = tmp_path / "p.out"
    f.write_text("L3 L2 L1 orbit tau x\n0 1 1 1 3 0.250000\n0 0 1 1 1 -2.000000\n")
    eio.read_phonons_(m, str(f))
    exp = np.full(12, 7.0)
    exp[(4 - 1) * 3 + 2] = 0.25          # site 4 = cell (1,1)
    exp[(2 - 1) * 3 + 0] = -2.0
    assert np.array_equal(m.x, exp)
    f.write_text("L3 L2 L1 orbit tau x\n0 0 0 1 4 1.0\n")
    with pytest.raises(IndexError):
        eio.read_phonons_(m, str(f))


def test_ssh_phonon_file_format(tmp_path, no_device):
    """'type loc tau x' with x reshaped (L, N, nph) — SSHModels.jl:838-913."""
    m = _stub_ssh(2, 3, 4)
    m.x = np.linspace(-1, 1, m.x.size)
    f = tmp_path / "s.out"
    eio.write_phonons_(m, str(f))
    lines = f.read_text().split("\n")
    assert lines[0] == "type loc tau x" and len(lines) == m.x.size + 2
    assert lines[1] == "1 1 1 -1.000000"
    assert lines[1 + 4] == "1 2 1 %.6f" % m.x[4]
    assert lines[1 + 12] == "2 1 1 %.6f" % m.x[12]
    m2 = _stub_ssh(2, 3, 4)
    eio.read_phonons_(m2, str(f))
    assert np.allclose(m2.x, m.x, atol=5.1e-7, rtol=0)
    # no phonons: nothing is written (:840)
    m0 = _stub_ssh(0, 0, 4)
    f0 = tmp_path / "none.out"
    eio.write_phonons_(m0, str(f0))
    assert not f0.exists()


def test_M_matrix_file_reader(tmp_path):
    f = tmp_path / "M.out"
    f.write_text("col row real imag\n1 1 1.0000000000 0.0000000000\n1 2 -0.5000000000 0.0000000000\n")
    r, c, v = eio.read_M_matrix(str(f))
    assert r.tolist() == [1, 2] and c.tolist() == [1, 1] and v.tolist() == [1.0, -0.5]


# ---------------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_phonon_round_trip_through_device_model(tmp_path):
    """write → read on a second model → the two fermion matrices agree to the 6 printed decimals."""
    from elphdynamics_amd import configs
    for tag in ("d", "e"):
        m, m2 = configs.make_model(tag), configs.make_model(tag, seed=99)
        f = tmp_path / ("ph_%s.out" % tag)
        eio.write_phonons_(m, str(f))
        eio.read_phonons_(m2, str(f))
        assert np.allclose(m2.x, m.x, atol=5.1e-7, rtol=0) and not np.array_equal(m2.x, m.x)
        v = np.cos(np.arange(m.Ndim) * 0.37)
        y, y2 = np.zeros(m.Ndim), np.zeros(m.Ndim)
        models.mulM_(y, m, v)
        models.mulM_(y2, m2, v)
        assert np.abs(y - y2).max() < 1e-5 * np.abs(y).max()
        # replay: rewriting what was read reproduces the file byte for byte
        f2 = tmp_path / ("ph2_%s.out" % tag)
        eio.write_phonons_(m2, str(f2))
        assert f.read_bytes() == f2.read_bytes()
        m.close(); m2.close()


@pytest.mark.gpu
def test_construct_M_matches_golden_dense(tmp_path):
    """construct_M / write_M_matrix! (Models.jl:300-367) on the golden 4x4, Ltau=8 Holstein case: the triplets
    rebuild the dense M whose products are in the fixture."""
    g = golden("holstein_sq4_L8.npz")
    la = lat.Lattice(1, 4, 4, 1)
    m = models.HolsteinModel(la, 0.8, float(g["dtau"]), tol=1e-10)
    m.neighbor_table, m.t = np.array(g["raw"]), np.array(g["t_raw"])
    m.initialize_model_()
    assert np.array_equal(m.neighbor_table, g["table"])
    m.lam[:], m.lam2[:], m.mu[:] = g["lam"], g["lam2"], g["mu"]
    m.x[:] = g["x"]
    models.update_model_(m)
    rows, cols, vals = eio.construct_M(m)
    n = m.Ndim
    M = np.zeros((n, n))
    M[rows - 1, cols - 1] = vals
    assert np.allclose(M @ g["v"], g["Mv"], rtol=1e-12, atol=1e-13)
    assert np.allclose(M.T @ g["v"], g["MTv"], rtol=1e-12, atol=1e-13)
    sign, logdet = np.linalg.slogdet(M)
    assert sign > 0 and abs(logdet - float(g["logdetM"])) < 1e-9
    f = tmp_path / "M.out"
    eio.write_M_matrix_(m, str(f))
    r2, c2, v2 = eio.read_M_matrix(str(f))
    keep = np.abs(vals) > 1e-10
    assert np.array_equal(r2, rows[keep]) and np.array_equal(c2, cols[keep])
    assert np.allclose(v2, vals[keep], atol=5.1e-11, rtol=0)
    m.close()
