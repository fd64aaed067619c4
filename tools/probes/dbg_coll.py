"""(round 6 probe) the slab operator of the collective transport against the oracle on the same open-slab bond table"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from elphdynamics_amd import sharded, sharded_rccl, synth, lattice as lat
from oracle.oracle import Oracle
orc = Oracle()
for (norb, Ls, bonds, Ltau, P) in ((1, 8, lat.SQUARE_BONDS, 8, 2), (2, 4, lat.HONEYCOMB_BONDS, 6, 2), (1, 16, lat.SQUARE_BONDS, 160, 2)):
    dtau = 0.1
    la = lat.Lattice(norb, Ls, Ls, 1)
    raw = np.concatenate([la.calc_neighbor_table(o1, o2, d) for (o1, o2, d) in bonds], axis=0)
    tv = np.ones(raw.shape[0]) if Ls == 16 else 1.0 + 0.1 * synth.randn(5, raw.shape[0])
    cb = lat.initialize_checkerboard(raw, tv, dtau)
    N = la.nsites
    S = sharded.SpatialSlabs(norb, Ls, Ls, cb["table"], P)
    for q in range(P):
        sl = S.slabs[q]
        Nloc = sl["rows"].size * S.row
        ltab = np.ascontiguousarray(S.local_table(q, cb["table"]), dtype=np.int64)
        c, s_ = np.asarray(cb["cosht"])[sl["bonds"]], np.asarray(cb["sinht"])[sl["bonds"]]
        loc = sharded_rccl.LibraryLocal(torch, Nloc, Ltau, ltab, c, s_, 0)
        E = np.exp(-dtau * synth.phonon_field(Nloc, Ltau, Ltau * dtau, dtau, seed=5 + q))
        loc.set_expV(E)
        om = orc.make_model(0, Nloc, Ltau, ltab, np.ascontiguousarray(c), np.ascontiguousarray(s_), np.ascontiguousarray(E))
        v = synth.randn(11 + q, Nloc * Ltau)
        p = torch.from_numpy(v.copy()).cuda(); z = torch.empty_like(p)
        loc.mtm(z, p); torch.cuda.synchronize()
        ref = orc.mulMTM(om, v)
        d = (z.cpu().numpy() - ref).reshape(Nloc, Ltau)
        own = slice(sl["lo"] * S.row, (sl["lo"] + sl["R"]) * S.row)
        print(f"L={Ls} norb={norb} rank {q}: Nloc {Nloc} bonds {ltab.shape[0]}: mtm err all {np.linalg.norm(d)/np.linalg.norm(ref):.2e} own rows {np.linalg.norm(d[own])/np.linalg.norm(ref):.2e}", flush=True)
        loc.close()
