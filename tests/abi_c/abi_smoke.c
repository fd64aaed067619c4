/* abi_smoke.c — the C ABI of libelphgpu.so called from plain C, exactly as the `ccall`s of julia/ElPhGPU.jl would call it: no Python, no
 * ctypes, no torch in the process.  Built with gcc against include/elph_gpu.h and the in-tree library; run by
 * tests/test_gpu_parity.py::test_c_abi_smoke_program on the GPU box.
 *
 * The sequence mirrors the reference's use of a model (file:line under the reference's src/):
 *   elph_create                    HolsteinModel(...) + initialize_model!          HolsteinModels.jl:192-314,484-517
 *   elph_update_model_holstein     update_model!(model)                            HolsteinModels.jl:526-549
 *   elph_mulM / _mulMT / _mulMTM   mulM!, mulMᵀ!, mulMᵀM!                          HolsteinModels.jl:569-684, Models.jl:215-224
 *   elph_solver_set + elph_ldiv    ldiv!(x, model, b) -> (iters, err, flag)        Models.jl:139-186
 *   elph_kpm_create / _setup       SymmetricKPMPreconditioner, setup!(P)           KPMPreconditioners.jl:219-235,259-321
 *   elph_kpm_orders / _apply       P.order, ldiv!(z, P, r)                         KPMPreconditioners.jl:296-308,426-481
 *   elph_ldiv(use_precond = 1)     ldiv!(x, model, b, P)                           Models.jl:74-137
 *   elph_muldMdx_holstein          muldMdx!(dMdx, u, model, v)                     HolsteinModels.jl:691-755
 *   elph_destroy                   finalizer
 * against the golden vectors of tests/golden/ (independent dense numpy/scipy restatement), exported to a flat binary by
 * tests/abi_c/export_fixture.py.  Exit code 0 and a last line "ABI SMOKE OK" on success; any mismatch prints what and exits 1.
 *
 *   gcc -O1 -Wall -I include tests/abi_c/abi_smoke.c -o abi_smoke -L elphdynamics_amd -lelphgpu -Wl,-rpath,$PWD/elphdynamics_amd -lm
 *   ./abi_smoke tests/abi_c/holstein_sq4_L8.bin
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "elph_gpu.h"

static int failures = 0;

#define CHECK_RC(call)                                                                                   \
    do {                                                                                                 \
        int rc_ = (call);                                                                                \
        if (rc_ != ELPH_OK) { fprintf(stderr, "FAIL %s -> %d: %s\n", #call, rc_, elph_last_error()); return 1; } \
    } while (0)

static double rel_err(const double *a, const double *ref, int64_t n) {
    double num = 0.0, den = 0.0;
    for (int64_t i = 0; i < n; ++i) { num += (a[i] - ref[i]) * (a[i] - ref[i]); den += ref[i] * ref[i]; }
    return sqrt(num / den);
}

static void expect(const char *what, double err, double tol) {
    printf("  %-34s rel err %.3e (tol %.0e) %s\n", what, err, tol, err < tol ? "ok" : "MISMATCH");
    if (!(err < tol)) ++failures;
}

static double *rd(FILE *f, int64_t n) {
    double *p = (double *)malloc(sizeof(double) * (size_t)n);
    if (!p || fread(p, sizeof(double), (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s fixture.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    char magic[8];
    int64_t hdr[3];
    double par[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "ELPHFIX1", 8) != 0 || fread(hdr, sizeof(int64_t), 3, f) != 3 ||
        fread(par, sizeof(double), 8, f) != 8) { fprintf(stderr, "bad fixture header\n"); return 2; }
    const int64_t N = hdr[0], L = hdr[1], nb = hdr[2], ndim = N * L, Lo2 = (L + 1) / 2;
    const double dtau = par[0], kbuf = par[1], kc1 = par[2], kc2 = par[3], e_min = par[4], e_max = par[5], lam_lo = par[6], lam_hi = par[7];
    int64_t *table = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * nb)), *orders_ref = (int64_t *)malloc(sizeof(int64_t) * (size_t)Lo2);
    if (fread(table, sizeof(int64_t), (size_t)(2 * nb), f) != (size_t)(2 * nb) || fread(orders_ref, sizeof(int64_t), (size_t)Lo2, f) != (size_t)Lo2) return 2;
    double *cosht = rd(f, nb), *sinht = rd(f, nb), *lam = rd(f, N), *lam2 = rd(f, N), *mu = rd(f, N);
    double *x = rd(f, ndim), *E = rd(f, ndim), *v = rd(f, ndim), *Mv = rd(f, ndim), *MTv = rd(f, ndim), *MTMv = rd(f, ndim);
    double *b = rd(f, ndim), *xsol = rd(f, ndim), *kin = rd(f, ndim), *kout = rd(f, ndim);
    fclose(f);
    (void)E;

    printf("abi version %d; devices %d\n%s\n", elph_abi_version(), elph_device_count(), elph_build_info());
    if (elph_abi_version() != ELPH_ABI_VERSION) { fprintf(stderr, "unexpected ABI version\n"); return 1; }
    if (elph_device_count() < 1) { fprintf(stderr, "no HIP device: the library has no CPU path\n"); return 3; }

    elph_handle h = NULL;
    CHECK_RC(elph_create(&h, ELPH_MODEL_HOLSTEIN, N, L, nb, table, cosht, sinht, 0));
    CHECK_RC(elph_update_model_holstein(h, x, lam, lam2, mu, dtau));

    double *y = (double *)calloc((size_t)ndim, sizeof(double));
    CHECK_RC(elph_mulM(h, y, v));    expect("mulM!   vs dense M v", rel_err(y, Mv, ndim), 1e-13);
    CHECK_RC(elph_mulMT(h, y, v));   expect("mulMT!  vs dense M^T v", rel_err(y, MTv, ndim), 1e-13);
    CHECK_RC(elph_mulMTM(h, y, v));  expect("mulMTM! vs dense M^T M v", rel_err(y, MTMv, ndim), 1e-13);

    /* ldiv!(x, model, b): (iters, residual_error, flag) — Models.jl:139-186 */
    int64_t iters = 0;
    double resid = 0.0;
    int flag = -1;
    CHECK_RC(elph_solver_set(h, 1e-13, 5000, 1e12));
    memset(y, 0, sizeof(double) * (size_t)ndim);
    CHECK_RC(elph_ldiv(h, y, b, 0, 0, &iters, &resid, &flag));
    printf("  ldiv!: %lld iterations, residual %.2e, flag %d\n", (long long)iters, resid, flag);
    if (flag != 0) { printf("  ldiv! flag %d MISMATCH\n", flag); ++failures; }
    expect("ldiv!   vs dense solve", rel_err(y, xsol, ndim), 1e-10);
    const int64_t iters_plain = iters;
    /* flag logic: a solve cut short is reported and zero-filled, never thrown — Models.jl:157-180 */
    CHECK_RC(elph_solver_set(h, 1e-14, 3, 1e12));
    memset(y, 0, sizeof(double) * (size_t)ndim);           /* ldiv! starts from the caller's x (callers pass zeros: HMC.jl:854) */
    CHECK_RC(elph_ldiv(h, y, b, 0, 0, &iters, &resid, &flag));
    {
        double nz = 0.0;
        for (int64_t i = 0; i < ndim; ++i) nz += fabs(y[i]);
        const int ok = (iters == 3 && flag == 1 && nz == 0.0);
        printf("  %-34s iters %lld flag %d |x|_1 %.1e %s\n", "maxiter hit: flag 1, x zeroed", (long long)iters, flag, nz, ok ? "ok" : "MISMATCH");
        if (!ok) ++failures;
    }

    /* SymmetricKPMPreconditioner + setup! with injected eigenvalue bounds (the reference draws Arnoldi start vectors from model.rng) */
    int active = 0;
    double lo = 0.0, hi = 0.0;
    CHECK_RC(elph_kpm_create(h, 20, kbuf, kc1, kc2));
    CHECK_RC(elph_kpm_setup(h, NULL, NULL, e_min, e_max, &active, &lo, &hi));
    {
        const int ok = active == 1 && lo == lam_lo && hi == lam_hi;
        printf("  %-34s active %d lam_lo %.17g lam_hi %.17g %s\n", "setup!(P): bounds bit-equal", active, lo, hi, ok ? "ok" : "MISMATCH");
        if (!ok) ++failures;
    }
    int64_t *orders = (int64_t *)calloc((size_t)Lo2, sizeof(int64_t)), total = 0;
    CHECK_RC(elph_kpm_orders(h, orders, &total));
    {
        int ok = 1;
        for (int64_t w = 0; w < Lo2; ++w) ok = ok && orders[w] == orders_ref[w];
        printf("  %-34s total %lld %s\n", "Chebyshev orders per frequency", (long long)total, ok ? "ok" : "MISMATCH");
        if (!ok) ++failures;
    }
    CHECK_RC(elph_kpm_apply(h, y, kin));
    expect("ldiv!(z,P,r) vs dense Chebyshev", rel_err(y, kout, ndim), 1e-12);
    CHECK_RC(elph_solver_set(h, 1e-13, 5000, 1e12));
    memset(y, 0, sizeof(double) * (size_t)ndim);
    CHECK_RC(elph_ldiv(h, y, b, 1, 0, &iters, &resid, &flag));
    printf("  ldiv!(…, P): %lld iterations (plain: %lld), residual %.2e, flag %d\n", (long long)iters, (long long)iters_plain, resid, flag);
    if (flag != 0) ++failures;
    expect("ldiv!(…, P) vs dense solve", rel_err(y, xsol, ndim), 1e-10);

    /* muldMdx!(dMdx, u, model, v) — HolsteinModels.jl:691-755, the operator calc_dSfdx! applies (HMC.jl:799,804) — against its own
     * definition: dMdx[f] = d/dx_f (u . M(x) v), central difference through update_model! + mulM! of this same ABI (u = b, v = v) */
    {
        double *d = (double *)calloc((size_t)ndim, sizeof(double)), *xp = (double *)malloc(sizeof(double) * (size_t)ndim);
        CHECK_RC(elph_update_model_holstein(h, x, lam, lam2, mu, dtau));
        CHECK_RC(elph_muldMdx_holstein(h, d, b, v, x, lam, lam2, dtau));
        const int64_t probe[4] = {0, L - 1, ndim / 2 + 1, ndim - 1};
        const double eps = 1e-6;
        double worst = 0.0;
        for (int k = 0; k < 4; ++k) {
            double s[2];
            for (int sg = 0; sg < 2; ++sg) {
                memcpy(xp, x, sizeof(double) * (size_t)ndim);
                xp[probe[k]] += sg ? -eps : eps;
                CHECK_RC(elph_update_model_holstein(h, xp, lam, lam2, mu, dtau));
                CHECK_RC(elph_mulM(h, y, v));
                s[sg] = 0.0;
                for (int64_t i = 0; i < ndim; ++i) s[sg] += b[i] * y[i];
            }
            const double fd = (s[0] - s[1]) / (2 * eps), err = fabs(fd - d[probe[k]]) / fmax(1.0, fabs(fd));
            if (err > worst) worst = err;
        }
        expect("muldMdx! vs d(u.Mv)/dx (4 fields)", worst, 1e-7);
        CHECK_RC(elph_update_model_holstein(h, x, lam, lam2, mu, dtau));
        free(d); free(xp);
    }

    /* usage errors come back as codes with a message, never as a crash */
    {
        const int rc = elph_mulM(h, NULL, v);
        const int ok = rc == ELPH_E_ARG && strlen(elph_last_error()) > 0;
        printf("  %-34s rc %d \"%s\" %s\n", "NULL argument -> ELPH_E_ARG", rc, elph_last_error(), ok ? "ok" : "MISMATCH");
        if (!ok) ++failures;
    }
    CHECK_RC(elph_destroy(h));
    if (failures) { printf("ABI SMOKE FAILED: %d mismatch(es)\n", failures); return 1; }
    printf("ABI SMOKE OK\n");
    return 0;
}
