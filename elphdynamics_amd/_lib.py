"""ctypes binding of libelphgpu.so — the C ABI declared in include/elph_gpu.h.

The library is hand-written HIP for gfx950; there is no CPU path.  Loading fails loudly if the
shared object has not been built (`python -c "import __graft_entry__ as g; g.build()"`), and every
compute entry point returns ELPH_E_NOGPU / ELPH_E_HIP without a usable MI355X.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

c_i64, c_dbl, c_int = C.c_int64, C.c_double, C.c_int
P_i64, P_dbl, P_int = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)
Handle = C.c_void_p

ELPH_OK = 0
ABI_VERSION = 2          # include/elph_gpu.h: ELPH_ABI_VERSION this binding was written against (checked exactly at load)
ELPH_E_ARG, ELPH_E_HIP, ELPH_E_STATE, ELPH_E_NOGPU, ELPH_E_UNSUPPORTED = -1, -2, -3, -4, -5
ERRORS = {-1: "ELPH_E_ARG", -2: "ELPH_E_HIP", -3: "ELPH_E_STATE", -4: "ELPH_E_NOGPU", -5: "ELPH_E_UNSUPPORTED"}

# every symbol include/elph_gpu.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "elph_last_error": (C.c_char_p, []),
    "elph_abi_version": (c_int, []),
    "elph_build_info": (C.c_char_p, []),
    "elph_device_count": (c_int, []),
    "elph_create": (c_int, [C.POINTER(Handle), c_int, c_i64, c_i64, c_i64, P_i64, P_dbl, P_dbl, c_int]),
    "elph_destroy": (c_int, [Handle]),
    "elph_set_stream": (c_int, [Handle, C.c_void_p]),
    "elph_synchronize": (c_int, [Handle]),
    "elph_update_model_holstein": (c_int, [Handle, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl]),
    "elph_update_model_holstein_chains": (c_int, [Handle, c_int, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl]),
    "elph_set_expV": (c_int, [Handle, P_dbl]),
    "elph_update_model_ssh": (c_int, [Handle, P_dbl, P_dbl, P_dbl]),
    "elph_update_model_ssh_fields": (c_int, [Handle, P_dbl, c_i64, P_i64, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl]),
    "elph_update_model_ssh_fields_chains": (c_int, [Handle, c_int, P_dbl, c_i64, P_i64, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl]),
    "elph_get_cosh_sinh": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_mulM": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_mulMT": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_mulMTM": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_mulMMT": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_mulM_dev": (c_int, [Handle, C.c_void_p, C.c_void_p]),
    "elph_mulMT_dev": (c_int, [Handle, C.c_void_p, C.c_void_p]),
    "elph_mulMTM_dev": (c_int, [Handle, C.c_void_p, C.c_void_p]),
    "elph_solver_set": (c_int, [Handle, c_dbl, c_i64, c_dbl]),
    "elph_cg_solve": (c_int, [Handle, P_dbl, P_dbl, c_dbl, c_i64, c_dbl, c_int, P_i64, P_dbl]),
    "elph_ldiv": (c_int, [Handle, P_dbl, P_dbl, c_int, c_i64, P_i64, P_dbl, P_int]),
    "elph_ldiv_batched": (c_int, [Handle, c_int, P_dbl, P_dbl, c_int, c_i64, P_i64, P_dbl, P_int]),
    "elph_ldiv_dev": (c_int, [Handle, C.c_void_p, C.c_void_p, c_int, c_i64, P_i64, P_dbl, P_int]),
    "elph_ldiv_batched_dev": (c_int, [Handle, c_int, C.c_void_p, C.c_void_p, c_int, c_i64, P_i64, P_dbl, P_int]),
    "elph_fermion_force_holstein": (c_int, [Handle, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl, P_dbl, c_int, c_dbl, P_dbl, P_dbl, P_dbl,
                                            P_i64, P_int]),
    "elph_fermion_force_ssh": (c_int, [Handle, P_dbl, P_dbl, c_int, c_dbl, P_dbl, P_dbl, P_dbl, P_i64, P_int]),
    "elph_fermion_force_ssh_fields": (c_int, [Handle, P_dbl, P_dbl, c_int, c_dbl, P_dbl, P_dbl, P_dbl, P_i64, P_int]),
    "elph_muldMdx_holstein": (c_int, [Handle, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl]),
    "elph_muldMdx_holstein_dev": (c_int, [Handle, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P_dbl, P_dbl, c_dbl]),
    "elph_muldMdx_ssh": (c_int, [Handle, P_dbl, P_dbl, P_dbl]),
    "elph_muldMdx_ssh_fields": (c_int, [Handle, P_dbl, P_dbl, P_dbl]),
    "elph_muldMdx_ssh_fields_dev": (c_int, [Handle, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elph_hmc_create": (c_int, [Handle, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_hmc_set_state": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_hmc_get_state": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_hmc_update": (c_int, [Handle, c_dbl, c_i64, c_int, c_dbl, c_int, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_int, P_dbl, P_dbl,
                                P_int]),
    "elph_hmc_set_mu": (c_int, [Handle, P_dbl]),
    "elph_hmc_set_mu_chains": (c_int, [Handle, P_dbl]),
    "elph_hmc_set_shared_fields": (c_int, [Handle, P_i64]),
    "elph_hmc_set_rng": (c_int, [Handle, C.c_uint64]),
    "elph_hmc_rng_batches": (c_int, [Handle, C.POINTER(C.c_uint64)]),
    "elph_hmc_special_move_chains": (c_int, [Handle, c_int, P_i64, P_i64, P_dbl, P_dbl, c_int, P_dbl, P_dbl, P_int, P_dbl, P_dbl, P_i64, P_int]),
    "elph_hmc_special_move": (c_int, [Handle, c_int, c_i64, c_i64, P_dbl, P_dbl, c_int, P_dbl, c_dbl, P_int, P_dbl, P_dbl, P_i64, P_int]),
    "elph_langevin_create": (c_int, [Handle, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_langevin_create_ssh": (c_int, [Handle, c_i64, P_dbl, P_dbl, P_i64, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_langevin_create_chains": (c_int, [Handle, c_int, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_langevin_evolve": (c_int, [Handle, c_int, c_dbl, c_int, P_dbl, P_dbl, P_dbl, P_dbl, P_i64, P_int]),
    "elph_greens_create": (c_int, [Handle, c_int, c_int, c_int, c_int, c_int]),
    "elph_greens_nv": (c_int, [Handle, P_int]),
    "elph_greens_update": (c_int, [Handle, P_dbl, c_int, P_i64, P_dbl, P_int]),
    "elph_greens_set_vectors": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_greens_get_vectors": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_greens_setup": (c_int, [Handle, c_int, c_int, P_dbl, P_dbl, P_dbl, P_dbl]),
    "elph_greens_dev_arrays": (c_int, [Handle, C.POINTER(C.c_void_p), P_i64]),
    "elph_hmc_create_ssh": (c_int, [Handle, c_i64, P_dbl, P_dbl, P_i64, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_hmc_create_ssh_chains": (c_int, [Handle, c_int, c_i64, P_dbl, P_dbl, P_i64, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_hmc_create_chains": (c_int, [Handle, c_int, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl]),
    "elph_hmc_update_chains": (c_int, [Handle, c_dbl, c_i64, c_int, c_dbl, c_int, P_dbl, P_dbl, P_dbl, P_dbl, P_dbl, P_int, P_dbl, P_dbl,
                                       P_int]),
    "elph_kpm_create": (c_int, [Handle, c_int, c_dbl, c_dbl, c_dbl]),
    "elph_kpm_setup": (c_int, [Handle, P_dbl, P_dbl, c_dbl, c_dbl, P_int, P_dbl, P_dbl]),
    "elph_kpm_setup_chains": (c_int, [Handle, P_dbl, P_dbl, P_dbl, P_dbl, P_int, P_dbl, P_dbl]),
    "elph_kpm_orders": (c_int, [Handle, P_i64, P_i64]),
    "elph_kpm_apply": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_kpm_apply_dev": (c_int, [Handle, C.c_void_p, C.c_void_p]),
    "elph_fourier_accelerate": (c_int, [Handle, P_dbl, P_dbl, P_dbl, c_dbl, c_i64]),
    "elph_tau_to_omega": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_omega_to_tau": (c_int, [Handle, P_dbl, P_dbl]),
    "elph_wg_status": (c_int, [Handle, P_int, P_i64]),
    "elph_shard_shape": (c_int, [c_i64, c_int, P_int, P_int, P_int, P_int]),
    "elph_shard_create": (c_int, [Handle, c_int, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, P_i64, C.c_void_p]),
    "elph_shard_connect": (c_int, [Handle, C.c_void_p]),
    "elph_shard_prepare": (c_int, [Handle]),
    "elph_shard_selftest": (c_int, [Handle, c_int, P_dbl, P_dbl]),
    "elph_peer_access": (c_int, [c_int, c_int, P_int]),
    "elph_shard_solve": (c_int, [Handle, P_dbl, P_dbl, c_dbl, c_i64, c_dbl, P_i64, P_int, P_dbl]),
    "elph_shard_solve_kpm": (c_int, [Handle, Handle, P_dbl, P_dbl, c_dbl, c_i64, c_dbl, P_i64, P_int, P_dbl]),
    "elph_shard_destroy": (c_int, [Handle]),
    "elph_shard_set_collectives": (c_int, [Handle, C.c_void_p, C.c_void_p, C.c_void_p]),
    "elph_shard_hmc_set_columns": (c_int, [Handle, P_i64, c_i64, P_dbl]),
    "elph_shard_set_full_lattice": (c_int, [Handle, Handle]),
    "elph_shard_set_bonds": (c_int, [Handle, P_i64, c_i64, P_dbl]),
    "elph_shard_ghost_stats": (c_int, [Handle, P_i64, P_i64]),
    "elph_shard_ldiv": (c_int, [Handle, Handle, P_dbl, P_dbl, c_int, c_i64, P_i64, P_dbl, P_int]),
    "elph_shard_fermion_force_holstein": (c_int, [Handle, Handle, P_dbl, P_dbl, P_dbl, P_dbl, c_dbl, P_dbl, P_dbl, c_int, c_dbl, P_dbl, P_dbl,
                                                  P_dbl, P_i64, P_int]),
    "elph_shard_fermion_force_ssh": (c_int, [Handle, Handle, P_dbl, P_dbl, c_int, c_dbl, P_dbl, P_dbl, P_dbl, P_i64, P_int]),
}

# private measurement hooks (csrc/elph_bench.h: exported by the library, not part of the drop-in ABI; bench.py and tools/ use them)
BENCH_SIGNATURES = {
    "elph_shard_iterate": (c_int, [Handle, P_dbl, c_i64, P_dbl]),
    "elph_bench_prepare": (c_int, [Handle, c_int, c_int, P_dbl]),
    "elph_bench_run": (c_int, [Handle, c_int, c_int, c_int, c_int, P_dbl]),
    "elph_bench_info": (c_int, [Handle, c_int, P_int]),
    "elph_bench_wg_info": (c_int, [Handle, c_int, P_int, P_int, P_int, P_int]),
    "elph_bench_px_info": (c_int, [Handle, P_int]),
    "elph_bench_pg_info": (c_int, [Handle, P_int, P_int, P_int, P_int, P_int]),
    "elph_bench_slabs_info": (c_int, [Handle, c_int, P_int, P_int, P_int, P_int]),
}


class ElphError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{ERRORS.get(code, code)}: {text}")
        self.code = code


_lib = None


def library_path():
    return _build.LIB


def load():
    """Load libelphgpu.so (once). Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with hipcc first (python -c 'import __graft_entry__ as g; g.build()'); "
                          "elphdynamics_amd has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in list(SIGNATURES.items()) + list(BENCH_SIGNATURES.items()):
        fn = getattr(lib, name)   # AttributeError here == ABI mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.elph_abi_version() != ABI_VERSION:
        raise ImportError(f"{path} speaks ABI {lib.elph_abi_version()}, this binding was written for {ABI_VERSION}: rebuild the library")
    _lib = lib
    return lib


def check(code):
    if code != ELPH_OK:
        raise ElphError(code, load().elph_last_error().decode())


def dptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "float64 C-contiguous array required"
    return a.ctypes.data_as(P_dbl)


def iptr(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"], "int64 C-contiguous array required"
    return a.ctypes.data_as(P_i64)
