/*
 * elph_oracle.c — CPU restatement (plain C) of the ElPhDynamics hot path.
 *
 * TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline); see elph_oracle.h.
 * PARITY UNPINNED BY THE REFERENCE (no reference tests/golden vectors, no Julia here).
 *
 * Loop structure, pass structure and operation order follow the reference
 * file:line cited at each function; nothing here is fused or reordered, so that
 * the timed build (-O3 -march=native -ffast-math, the analogue of the reference's
 * @fastmath @inbounds @simd) is a fair single-thread stand-in for the Julia code.
 */
#include "elph_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ====================================================================== */
/* geometry                                                               */
/* ====================================================================== */

static int64_t jl_mod(int64_t a, int64_t m) { /* Julia mod(): result has sign of m */
    int64_t r = a % m;
    return (r < 0) ? r + m : r;
}

/* Lattices.jl:149-168, 384-391 */
int64_t elpho_loc_to_site(int64_t norbits, int64_t L1, int64_t L2, int64_t L3, int64_t orbit,
                          int64_t l1, int64_t l2, int64_t l3) {
    int64_t l1p = jl_mod(l1, L1), l2p = jl_mod(l2, L2), l3p = jl_mod(l3, L3);
    int64_t cell = l1p + l2p * L1 + l3p * L1 * L2 + 1; /* loc_to_cell, 1-based */
    return norbits * (cell - 1) + orbit;
}

/* Lattices.jl:176-191 with site_to_cell / cell_loc as built at Lattices.jl:86-104 */
int64_t elpho_site_to_site(int64_t norbits, int64_t L1, int64_t L2, int64_t L3, int64_t isite,
                           const int64_t d[3], int64_t orbit) {
    int64_t cell = (isite - 1) / norbits; /* 0-based cell */
    int64_t l1 = cell % L1;
    int64_t l2 = (cell / L1) % L2;
    int64_t l3 = cell / (L1 * L2);
    return elpho_loc_to_site(norbits, L1, L2, L3, orbit, l1 + d[0], l2 + d[1], l3 + d[2]);
}

/* Lattices.jl:265-316 */
int64_t elpho_calc_neighbor_table(int64_t norbits, int64_t L1, int64_t L2, int64_t L3, int64_t o1,
                                  int64_t o2, const int64_t d[3], int remove_duplicates,
                                  int64_t *table) {
    int64_t nsites = norbits * L1 * L2 * L3;
    int64_t N = nsites / norbits;
    int64_t cnt = 0;
    for (int64_t isite = o1; isite <= nsites; isite += norbits) {
        int64_t fsite = elpho_site_to_site(norbits, L1, L2, L3, isite, d, o2);
        table[2 * cnt + 0] = isite;
        table[2 * cnt + 1] = fsite;
        cnt++;
    }
    if (!remove_duplicates) return N;
    char *keep = (char *)malloc((size_t)N);
    memset(keep, 1, (size_t)N);
    for (int64_t i = 0; i < N - 1; i++) {
        if (!keep[i]) continue;
        int64_t a = table[2 * i], b = table[2 * i + 1];
        for (int64_t j = i + 1; j < N; j++) {
            int64_t a2 = table[2 * j], b2 = table[2 * j + 1];
            if ((a == a2 && b == b2) || (a == b2 && b == a2)) keep[j] = 0;
        }
    }
    int64_t out = 0;
    for (int64_t i = 0; i < N; i++) {
        if (keep[i]) {
            table[2 * out] = table[2 * i];
            table[2 * out + 1] = table[2 * i + 1];
            out++;
        }
    }
    free(keep);
    return out;
}

/* stable merge sort of indices by key (Julia sortperm is stable) */
static void msort_idx(const int64_t *keys, int64_t *idx, int64_t *tmp, int64_t lo, int64_t hi) {
    if (hi - lo < 2) return;
    int64_t mid = lo + (hi - lo) / 2;
    msort_idx(keys, idx, tmp, lo, mid);
    msort_idx(keys, idx, tmp, mid, hi);
    int64_t a = lo, b = mid, k = lo;
    while (a < mid && b < hi) {
        if (keys[idx[b]] < keys[idx[a]]) tmp[k++] = idx[b++];
        else tmp[k++] = idx[a++];
    }
    while (a < mid) tmp[k++] = idx[a++];
    while (b < hi) tmp[k++] = idx[b++];
    for (int64_t i = lo; i < hi; i++) idx[i] = tmp[i];
}

void elpho_sortperm(const int64_t *keys, int64_t n, int64_t *perm) {
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    msort_idx(keys, perm, tmp, 0, n);
    for (int64_t i = 0; i < n; i++) perm[i] += 1; /* 1-based like Julia */
    free(tmp);
}

/* Lattices.jl:323-340 */
void elpho_sorted_neighbor_table_perm(int64_t *table, int64_t nb, int64_t *perm) {
    int64_t mx = 0;
    for (int64_t i = 0; i < nb; i++) {
        int64_t c1 = table[2 * i], c2 = table[2 * i + 1];
        if (c1 > c2) {
            table[2 * i] = c2;
            table[2 * i + 1] = c1;
        }
        if (table[2 * i] > mx) mx = table[2 * i];
        if (table[2 * i + 1] > mx) mx = table[2 * i + 1];
    }
    int64_t *vals = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nb > 0 ? nb : 1));
    for (int64_t i = 0; i < nb; i++) vals[i] = mx * table[2 * i] + table[2 * i + 1];
    elpho_sortperm(vals, nb, perm);
    free(vals);
}

/* Checkerboard.jl:471-515 */
int64_t elpho_checkerboard_groups(const int64_t *table, int64_t nb, int64_t *groups) {
    for (int64_t i = 0; i < nb; i++) groups[i] = 0;
    int64_t group = 0, nassigned = 0;
    while (nassigned < nb) {
        group += 1;
        for (int64_t n = 0; n < nb; n++) {
            if (groups[n] != 0) continue;
            groups[n] = group;
            nassigned += 1;
            for (int64_t p = 0; p < n; p++) {
                if (groups[p] != group) continue;
                if (table[2 * n] == table[2 * p] || table[2 * n + 1] == table[2 * p + 1] ||
                    table[2 * n] == table[2 * p + 1] || table[2 * n + 1] == table[2 * p]) {
                    groups[n] = 0;
                    nassigned -= 1;
                    break;
                }
            }
        }
    }
    return group;
}

static void permute_table(int64_t *table, int64_t nb, const int64_t *perm1) {
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)(nb > 0 ? nb : 1));
    memcpy(tmp, table, sizeof(int64_t) * 2 * (size_t)nb);
    for (int64_t i = 0; i < nb; i++) {
        table[2 * i] = tmp[2 * (perm1[i] - 1)];
        table[2 * i + 1] = tmp[2 * (perm1[i] - 1) + 1];
    }
    free(tmp);
}
static void permute_d(double *v, int64_t nb, const int64_t *perm1) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(nb > 0 ? nb : 1));
    memcpy(tmp, v, sizeof(double) * (size_t)nb);
    for (int64_t i = 0; i < nb; i++) v[i] = tmp[perm1[i] - 1];
    free(tmp);
}

/* HolsteinModels.jl:484-517 */
int64_t elpho_holstein_initialize_model(int64_t *table, int64_t nb, const double *t, double dtau,
                                        double *cosht, double *sinht, int64_t *cb_perm,
                                        int64_t *groups_sorted) {
    if (nb <= 0) return 0;
    for (int64_t i = 0; i < nb; i++) {
        cosht[i] = cosh(dtau * t[i]);
        sinht[i] = sinh(dtau * t[i]);
    }
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *new_perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *groups = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *comp = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    elpho_sorted_neighbor_table_perm(table, nb, perm);
    permute_table(table, nb, perm);
    permute_d(cosht, nb, perm);
    permute_d(sinht, nb, perm);
    int64_t ng = elpho_checkerboard_groups(table, nb, groups);
    elpho_sortperm(groups, nb, new_perm); /* Checkerboard.jl:442-446 */
    permute_table(table, nb, new_perm);
    permute_d(cosht, nb, new_perm);
    permute_d(sinht, nb, new_perm);
    for (int64_t i = 0; i < nb; i++) comp[i] = perm[new_perm[i] - 1];
    elpho_sortperm(comp, nb, cb_perm); /* checkerboard_perm = sortperm(perm[new_perm]) */
    if (groups_sorted)
        for (int64_t i = 0; i < nb; i++) groups_sorted[i] = groups[new_perm[i] - 1];
    free(perm);
    free(new_perm);
    free(groups);
    free(comp);
    return ng;
}

/* SSHModels.jl:435-448 */
int64_t elpho_ssh_initialize_table(int64_t *table, int64_t nb, int64_t *cb_perm,
                                   int64_t *inv_cb_perm, int64_t *groups_sorted) {
    if (nb <= 0) return 0;
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *new_perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    int64_t *groups = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
    elpho_sorted_neighbor_table_perm(table, nb, perm);
    permute_table(table, nb, perm);
    int64_t ng = elpho_checkerboard_groups(table, nb, groups);
    elpho_sortperm(groups, nb, new_perm);
    permute_table(table, nb, new_perm);
    for (int64_t i = 0; i < nb; i++) inv_cb_perm[i] = perm[new_perm[i] - 1];
    elpho_sortperm(inv_cb_perm, nb, cb_perm);
    if (groups_sorted)
        for (int64_t i = 0; i < nb; i++) groups_sorted[i] = groups[new_perm[i] - 1];
    free(perm);
    free(new_perm);
    free(groups);
    return ng;
}

/* HolsteinModels.jl:205 — Julia round(Int, x) rounds ties to even */
int64_t elpho_ltau(double beta, double dtau) { return (int64_t)rint(beta / dtau); }

/* ====================================================================== */
/* model update                                                            */
/* ====================================================================== */

/* HolsteinModels.jl:526-549 */
void elpho_update_model_holstein(int64_t N, int64_t L, double dtau, const double *x,
                                 const double *lambda, const double *lambda2, const double *mu,
                                 double *expV) {
    for (int64_t i = 0; i < N; i++) {
        for (int64_t tau = 0; tau < L; tau++) {
            int64_t idx = i * L + tau;
            expV[idx] = exp(-dtau * (lambda[i] * x[idx] + lambda2[i] * (x[idx] * x[idx]) + -mu[i]));
        }
    }
}

static double jl_sign(double x) { return (x > 0) ? 1.0 : ((x < 0) ? -1.0 : x); }

/* SSHModels.jl:510-535 */
void elpho_update_model_ssh(int64_t N, int64_t L, int64_t nb, int64_t Nph, double dtau,
                            const double *x, const double *t, const double *alpha,
                            const double *alpha2, const double *mu,
                            const int64_t *phonon_to_bond, const int64_t *cb_perm, double *cosht,
                            double *sinht, double *expDtauMu) {
    (void)nb;
    for (int64_t i = 0; i < N; i++) expDtauMu[i] = exp(dtau * mu[i]);
    for (int64_t field = 0; field < Nph * L; field++) {
        int64_t phonon = field / L; /* field_to_phonon = repeat(1:Nph, inner=L) */
        int64_t tau = field % L;    /* field_to_tau    = repeat(1:L, outer=Nph) */
        int64_t bond = phonon_to_bond[phonon] - 1;
        int64_t index = cb_perm[bond] - 1;
        double xt = x[field];
        double v = alpha[phonon] * xt + jl_sign(xt) * alpha2[phonon] * (xt * xt);
        double tp = t[bond] - v;
        cosht[tau + L * index] = cosh(dtau * tp);
        sinht[tau + L * index] = sinh(dtau * tp);
    }
}

/* ====================================================================== */
/* checkerboard products                                                   */
/* ====================================================================== */

/* Checkerboard.jl:57-83 */
void elpho_checkerboard_mul(double *y, const int64_t *table, const double *c, const double *s,
                            int64_t nb, int64_t L) {
    for (int64_t n = 0; n < nb; n++) {
        double cn = c[n], sn = s[n];
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double *yi = y + i * L, *yj = y + j * L;
        for (int64_t tau = 0; tau < L; tau++) {
            double t1 = yi[tau], t2 = yj[tau];
            yi[tau] = cn * t1 + sn * t2;
            yj[tau] = cn * t2 + sn * t1;
        }
    }
}

/* Checkerboard.jl:149-175 */
void elpho_checkerboard_transpose_mul(double *y, const int64_t *table, const double *c,
                                      const double *s, int64_t nb, int64_t L) {
    for (int64_t n = nb - 1; n >= 0; n--) {
        double cn = c[n], sn = s[n];
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double *yi = y + i * L, *yj = y + j * L;
        for (int64_t tau = 0; tau < L; tau++) {
            double t1 = yi[tau], t2 = yj[tau];
            yi[tau] = cn * t1 + sn * t2;
            yj[tau] = cn * t2 + sn * t1;
        }
    }
}

/* Checkerboard.jl:238-264 */
void elpho_checkerboard_inverse_mul(double *y, const int64_t *table, const double *c,
                                    const double *s, int64_t nb, int64_t L) {
    for (int64_t n = nb - 1; n >= 0; n--) {
        double cn = c[n], sn = s[n];
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double *yi = y + i * L, *yj = y + j * L;
        for (int64_t tau = 0; tau < L; tau++) {
            double t1 = yi[tau], t2 = yj[tau];
            yi[tau] = cn * t1 - sn * t2;
            yj[tau] = cn * t2 - sn * t1;
        }
    }
}

/* Checkerboard.jl:323-349 */
void elpho_checkerboard_inverse_transpose_mul(double *y, const int64_t *table, const double *c,
                                              const double *s, int64_t nb, int64_t L) {
    for (int64_t n = 0; n < nb; n++) {
        double cn = c[n], sn = s[n];
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double *yi = y + i * L, *yj = y + j * L;
        for (int64_t tau = 0; tau < L; tau++) {
            double t1 = yi[tau], t2 = yj[tau];
            yi[tau] = cn * t1 - sn * t2;
            yj[tau] = cn * t2 - sn * t1;
        }
    }
}

/* Checkerboard.jl:86-121 */
void elpho_checkerboard_mul_mat(double *y, const int64_t *table, const double *c, const double *s,
                                int64_t nb, int64_t L) {
    for (int64_t n = 0; n < nb; n++) {
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double *yi = y + i * L, *yj = y + j * L;
        const double *cn = c + n * L, *sn = s + n * L;
        for (int64_t tau = 0; tau < L; tau++) {
            double t1 = yi[tau], t2 = yj[tau];
            yi[tau] = cn[tau] * t1 + sn[tau] * t2;
            yj[tau] = cn[tau] * t2 + sn[tau] * t1;
        }
    }
}

/* Checkerboard.jl:177-210 */
void elpho_checkerboard_transpose_mul_mat(double *y, const int64_t *table, const double *c,
                                          const double *s, int64_t nb, int64_t L) {
    for (int64_t n = nb - 1; n >= 0; n--) {
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double *yi = y + i * L, *yj = y + j * L;
        const double *cn = c + n * L, *sn = s + n * L;
        for (int64_t tau = 0; tau < L; tau++) {
            double t1 = yi[tau], t2 = yj[tau];
            yi[tau] = cn[tau] * t1 + sn[tau] * t2;
            yj[tau] = cn[tau] * t2 + sn[tau] * t1;
        }
    }
}

/* Checkerboard.jl:123-141, complex y */
void elpho_checkerboard_mul_nvec_z(double *y, const int64_t *table, const double *c,
                                   const double *s, int64_t nb) {
    for (int64_t n = 0; n < nb; n++) {
        double cn = c[n], sn = s[n];
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double t1r = y[2 * i], t1i = y[2 * i + 1], t2r = y[2 * j], t2i = y[2 * j + 1];
        y[2 * i] = cn * t1r + sn * t2r;
        y[2 * i + 1] = cn * t1i + sn * t2i;
        y[2 * j] = cn * t2r + sn * t1r;
        y[2 * j + 1] = cn * t2i + sn * t1i;
    }
}

/* Checkerboard.jl:212-230, complex y */
void elpho_checkerboard_transpose_mul_nvec_z(double *y, const int64_t *table, const double *c,
                                             const double *s, int64_t nb) {
    for (int64_t n = nb - 1; n >= 0; n--) {
        double cn = c[n], sn = s[n];
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double t1r = y[2 * i], t1i = y[2 * i + 1], t2r = y[2 * j], t2i = y[2 * j + 1];
        y[2 * i] = cn * t1r + sn * t2r;
        y[2 * i + 1] = cn * t1i + sn * t2i;
        y[2 * j] = cn * t2r + sn * t1r;
        y[2 * j + 1] = cn * t2i + sn * t1i;
    }
}

/* Checkerboard.jl:123-141, real y */
void elpho_checkerboard_mul_nvec(double *y, const int64_t *table, const double *c, const double *s,
                                 int64_t nb) {
    for (int64_t n = 0; n < nb; n++) {
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double t1 = y[i], t2 = y[j];
        y[i] = c[n] * t1 + s[n] * t2;
        y[j] = c[n] * t2 + s[n] * t1;
    }
}

/* Checkerboard.jl:298-316, real y */
void elpho_checkerboard_inverse_mul_nvec(double *y, const int64_t *table, const double *c,
                                         const double *s, int64_t nb) {
    for (int64_t n = nb - 1; n >= 0; n--) {
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        double t1 = y[i], t2 = y[j];
        y[i] = c[n] * t1 - s[n] * t2;
        y[j] = c[n] * t2 - s[n] * t1;
    }
}

/* ====================================================================== */
/* M, M^T, M^T M                                                           */
/* ====================================================================== */

/* HolsteinModels.jl:569-626 ; SSHModels.jl:581-640 */
void elpho_mulM(double *y, const elpho_model *m, const double *v) {
    const int64_t N = m->N, L = m->L;
    /* pass 1: y(tau) = E .* v(tau-1)   (mod1 wrap) */
    for (int64_t i = 0; i < N; i++) {
        for (int64_t tau = 0; tau < L; tau++) {
            int64_t taum1 = (tau == 0) ? L - 1 : tau - 1;
            double e = (m->kind == 0) ? m->E[i * L + tau] : m->E[i];
            y[i * L + tau] = e * v[i * L + taum1];
        }
    }
    /* pass 2: checkerboard */
    if (m->nb > 0) {
        if (m->kind == 0) elpho_checkerboard_mul(y, m->table, m->c, m->s, m->nb, L);
        else elpho_checkerboard_mul_mat(y, m->table, m->c, m->s, m->nb, L);
    }
    /* pass 3: combine */
    for (int64_t i = 0; i < N; i++) {
        y[i * L] = v[i * L] + y[i * L];
        for (int64_t tau = 1; tau < L; tau++) y[i * L + tau] = v[i * L + tau] - y[i * L + tau];
    }
}

/* HolsteinModels.jl:631-684 ; SSHModels.jl:646-701 */
void elpho_mulMT(double *y, const elpho_model *m, const double *v) {
    const int64_t N = m->N, L = m->L;
    memcpy(y, v, sizeof(double) * (size_t)(N * L));
    if (m->nb > 0) {
        if (m->kind == 0) elpho_checkerboard_transpose_mul(y, m->table, m->c, m->s, m->nb, L);
        else elpho_checkerboard_transpose_mul_mat(y, m->table, m->c, m->s, m->nb, L);
    }
    for (int64_t i = 0; i < N; i++) {
        const double *Ei = (m->kind == 0) ? (m->E + i * L) : NULL;
        double ei = (m->kind == 0) ? 0.0 : m->E[i];
        double *yi = y + i * L;
        const double *vi = v + i * L;
        double y_iL = vi[L - 1] + ((m->kind == 0) ? Ei[0] : ei) * yi[0];
        for (int64_t tau = 0; tau < L - 1; tau++)
            yi[tau] = vi[tau] - ((m->kind == 0) ? Ei[tau + 1] : ei) * yi[tau + 1];
        yi[L - 1] = y_iL;
    }
}

/* Models.jl:215-224 */
void elpho_mulMTM(double *y, const elpho_model *m, const double *v) {
    elpho_mulM(m->vp, m, v);
    elpho_mulMT(y, m, m->vp);
}

/* ====================================================================== */
/* DFT along tau                                                           */
/* ====================================================================== */

/* FFTW.jl conventions (un-vendored dep FFTW.jl 1.3.2 / FFTW_jll 3.3.9):
 * fft: X[k] = sum_t x[t] exp(-2 pi i k t / L); ifft: same with +, scaled 1/L.
 * Restated as a direct O(L^2) sum with an exactly index-reduced twiddle table. */
void elpho_dft_tau(double *out_z, const double *in_z, int64_t N, int64_t L, int sign) {
    double *wr = (double *)malloc(sizeof(double) * (size_t)L);
    double *wi = (double *)malloc(sizeof(double) * (size_t)L);
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)L);
    for (int64_t m = 0; m < L; m++) {
        double a = 2.0 * M_PI * (double)m / (double)L;
        wr[m] = cos(a);
        wi[m] = (sign < 0) ? -sin(a) : sin(a);
    }
    double scale = (sign < 0) ? 1.0 : 1.0 / (double)L;
    for (int64_t i = 0; i < N; i++) {
        const double *x = in_z + 2 * i * L;
        for (int64_t k = 0; k < L; k++) {
            double sr = 0.0, si = 0.0;
            int64_t m = 0;
            for (int64_t t = 0; t < L; t++) {
                double xr = x[2 * t], xi = x[2 * t + 1];
                sr += xr * wr[m] - xi * wi[m];
                si += xr * wi[m] + xi * wr[m];
                m += k;
                if (m >= L) m -= L;
            }
            tmp[2 * k] = sr * scale;
            tmp[2 * k + 1] = si * scale;
        }
        memcpy(out_z + 2 * i * L, tmp, sizeof(double) * 2 * (size_t)L);
    }
    free(wr);
    free(wi);
    free(tmp);
}

/* TimeFreqFFTs.jl:37 (Theta), :55-73 */
void elpho_tau_to_omega(double *out_z, const double *in, int64_t N, int64_t L) {
    double *vtemp = (double *)malloc(sizeof(double) * 2 * (size_t)(N * L));
    for (int64_t i = 0; i < N; i++) {
        for (int64_t t = 0; t < L; t++) {
            double a = -M_PI * (double)t / (double)L; /* Theta[t] = exp(-i pi t/L), t 0-based */
            vtemp[2 * (i * L + t)] = cos(a) * in[i * L + t];
            vtemp[2 * (i * L + t) + 1] = sin(a) * in[i * L + t];
        }
    }
    elpho_dft_tau(out_z, vtemp, N, L, -1);
    free(vtemp);
}

/* TimeFreqFFTs.jl:112-130 */
void elpho_omega_to_tau(double *out, const double *in_z, int64_t N, int64_t L) {
    double *vtemp = (double *)malloc(sizeof(double) * 2 * (size_t)(N * L));
    elpho_dft_tau(vtemp, in_z, N, L, +1);
    for (int64_t i = 0; i < N; i++) {
        for (int64_t t = 0; t < L; t++) {
            double a = -M_PI * (double)t / (double)L;
            double cr = cos(a), ci = -sin(a); /* conj(Theta) */
            double vr = vtemp[2 * (i * L + t)], vi = vtemp[2 * (i * L + t) + 1];
            out[i * L + t] = cr * vr - ci * vi;
        }
    }
    free(vtemp);
}

/* ====================================================================== */
/* Fourier acceleration                                                    */
/* ====================================================================== */

/* FourierAcceleration.jl:260-266 */
double elpho_element_Mi(int64_t k, double omega, double dtau, double m0, double c, int64_t L) {
    int64_t kp = (k < L - k) ? k : L - k;
    double q = c * (double)kp / (double)L;
    double m = m0 * exp(-(q * q));
    return dtau * (m * m + omega * omega + (2.0 - 2.0 * cos(2.0 * M_PI * (double)kp / (double)L)) / (dtau * dtau)) /
           (m * m + omega * omega);
}

/* FourierAcceleration.jl:213-217 */
double elpho_element_Qi(int64_t k, double omega, double dtau, double m, int64_t L) {
    return (m * m + dtau * omega * omega + 4.0 / dtau) /
           (m * m + dtau * omega * omega + (2.0 - 2.0 * cos(2.0 * M_PI * (double)k / (double)L)) / dtau);
}

/* FourierAcceleration.jl:222-240,245-255 */
void elpho_update_M(double *Mdiag, int64_t Nph, int64_t L, double dtau, const double *omega,
                    double wmin, double wmax, double m0, double c) {
    for (int64_t ph = 0; ph < Nph; ph++)
        if (wmin < omega[ph] && omega[ph] < wmax)
            for (int64_t k = 0; k < L; k++) Mdiag[ph * L + k] = elpho_element_Mi(k, omega[ph], dtau, m0, c, L);
}

/* FourierAcceleration.jl:176-208 */
void elpho_update_Q(double *Qdiag, int64_t Nph, int64_t L, double dtau, const double *omega,
                    double wmin, double wmax, double m) {
    for (int64_t ph = 0; ph < Nph; ph++)
        if (wmin < omega[ph] && omega[ph] < wmax)
            for (int64_t k = 0; k < L; k++) Qdiag[ph * L + k] = elpho_element_Qi(k, omega[ph], dtau, m, L);
}

/* FourierAcceleration.jl:91-114,137-143 */
void elpho_fourier_accelerate(double *out, const double *in, const double *diag, double power,
                              int64_t N, int64_t L) {
    size_t n = (size_t)(N * L);
    double *vin = (double *)malloc(sizeof(double) * 2 * n);
    double *u = (double *)malloc(sizeof(double) * 2 * n);
    for (size_t i = 0; i < n; i++) {
        vin[2 * i] = in[i];
        vin[2 * i + 1] = 0.0;
    }
    elpho_dft_tau(u, vin, N, L, -1);
    for (size_t i = 0; i < n; i++) {
        double f = pow(diag[i], power);
        u[2 * i] *= f;
        u[2 * i + 1] *= f;
    }
    elpho_dft_tau(vin, u, N, L, +1);
    for (size_t i = 0; i < n; i++) out[i] = vin[2 * i];
    free(vin);
    free(u);
}

/* ====================================================================== */
/* small dense eigenvalues (stand-in for LAPACK eigvals!, KPMPreconditioners.jl:891,935) */
/* Hessenberg reduction + shifted QR (EISPACK elmhes/hqr algorithm).        */
/* ====================================================================== */

#define A_(i, j) a[(i) + (j) * n]

static void elmhes(double *a, int64_t n) {
    for (int64_t m = 1; m < n - 1; m++) {
        double x = 0.0;
        int64_t i = m;
        for (int64_t j = m; j < n; j++) {
            if (fabs(A_(j, m - 1)) > fabs(x)) {
                x = A_(j, m - 1);
                i = j;
            }
        }
        if (i != m) {
            for (int64_t j = m - 1; j < n; j++) {
                double t = A_(i, j);
                A_(i, j) = A_(m, j);
                A_(m, j) = t;
            }
            for (int64_t j = 0; j < n; j++) {
                double t = A_(j, i);
                A_(j, i) = A_(j, m);
                A_(j, m) = t;
            }
        }
        if (x != 0.0) {
            for (i = m + 1; i < n; i++) {
                double y = A_(i, m - 1);
                if (y != 0.0) {
                    y /= x;
                    A_(i, m - 1) = y;
                    for (int64_t j = m; j < n; j++) A_(i, j) -= y * A_(m, j);
                    for (int64_t j = 0; j < n; j++) A_(j, m) += y * A_(j, i);
                }
            }
        }
    }
    for (int64_t j = 0; j < n; j++)
        for (int64_t i = j + 2; i < n; i++) A_(i, j) = 0.0;
}

static double sign_of(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }

static int hqr(double *a, int64_t n, double *wr, double *wi) {
    int64_t nn, m, l, k, j, its, i, mmin;
    double z, y, x, w, v, u, t, s, r = 0, q = 0, p = 0, anorm = 0.0;
    for (i = 0; i < n; i++)
        for (j = (i > 0 ? i - 1 : 0); j < n; j++) anorm += fabs(A_(i, j));
    nn = n - 1;
    t = 0.0;
    while (nn >= 0) {
        its = 0;
        do {
            for (l = nn; l >= 1; l--) {
                s = fabs(A_(l - 1, l - 1)) + fabs(A_(l, l));
                if (s == 0.0) s = anorm;
                if (fabs(A_(l, l - 1)) + s == s) {
                    A_(l, l - 1) = 0.0;
                    break;
                }
            }
            x = A_(nn, nn);
            if (l == nn) {
                wr[nn] = x + t;
                wi[nn--] = 0.0;
            } else {
                y = A_(nn - 1, nn - 1);
                w = A_(nn, nn - 1) * A_(nn - 1, nn);
                if (l == nn - 1) {
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = sqrt(fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + sign_of(z, p);
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn - 1] = -(wi[nn] = z);
                    }
                    nn -= 2;
                } else {
                    if (its == 60) return -1;
                    if (its == 10 || its == 20) {
                        t += x;
                        for (i = 0; i <= nn; i++) A_(i, i) -= x;
                        s = fabs(A_(nn, nn - 1)) + fabs(A_(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    for (m = nn - 2; m >= l; m--) {
                        z = A_(m, m);
                        r = x - z;
                        s = y - z;
                        p = (r * s - w) / A_(m + 1, m) + A_(m, m + 1);
                        q = A_(m + 1, m + 1) - z - r - s;
                        r = A_(m + 2, m + 1);
                        s = fabs(p) + fabs(q) + fabs(r);
                        p /= s;
                        q /= s;
                        r /= s;
                        if (m == l) break;
                        u = fabs(A_(m, m - 1)) * (fabs(q) + fabs(r));
                        v = fabs(p) * (fabs(A_(m - 1, m - 1)) + fabs(z) + fabs(A_(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (i = m + 2; i <= nn; i++) {
                        A_(i, i - 2) = 0.0;
                        if (i != m + 2) A_(i, i - 3) = 0.0;
                    }
                    for (k = m; k <= nn - 1; k++) {
                        if (k != m) {
                            p = A_(k, k - 1);
                            q = A_(k + 1, k - 1);
                            r = 0.0;
                            if (k != nn - 1) r = A_(k + 2, k - 1);
                            if ((x = fabs(p) + fabs(q) + fabs(r)) != 0.0) {
                                p /= x;
                                q /= x;
                                r /= x;
                            }
                        }
                        if ((s = sign_of(sqrt(p * p + q * q + r * r), p)) != 0.0) {
                            if (k == m) {
                                if (l != m) A_(k, k - 1) = -A_(k, k - 1);
                            } else {
                                A_(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s;
                            y = q / s;
                            z = r / s;
                            q /= p;
                            r /= p;
                            for (j = k; j <= nn; j++) {
                                p = A_(k, j) + q * A_(k + 1, j);
                                if (k != nn - 1) {
                                    p += r * A_(k + 2, j);
                                    A_(k + 2, j) -= p * z;
                                }
                                A_(k + 1, j) -= p * y;
                                A_(k, j) -= p * x;
                            }
                            mmin = nn < k + 3 ? nn : k + 3;
                            for (i = l; i <= mmin; i++) {
                                p = x * A_(i, k) + y * A_(i, k + 1);
                                if (k != nn - 1) {
                                    p += z * A_(i, k + 2);
                                    A_(i, k + 2) -= p * r;
                                }
                                A_(i, k + 1) -= p * q;
                                A_(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return 0;
}
#undef A_

int elpho_eigvals(double *a, int64_t n, double *wr, double *wi) {
    if (n <= 0) return 0;
    if (n == 1) {
        wr[0] = a[0];
        wi[0] = 0.0;
        return 0;
    }
    elmhes(a, n);
    return hqr(a, n, wr, wi);
}

/* ====================================================================== */
/* KPM preconditioner                                                      */
/* ====================================================================== */

/* KPMPreconditioners.jl:332-349 (Holstein), 355-381 (SSH) */
void elpho_kpm_update_A(elpho_kpm *P, const elpho_model *m) {
    int64_t N = m->N, L = m->L;
    if (m->kind == 0) {
        for (int64_t i = 0; i < N; i++) {
            P->Ebar[i] = 0.0;
            for (int64_t tau = 0; tau < L; tau++) P->Ebar[i] += m->E[i * L + tau];
            P->Ebar[i] /= (double)L;
        }
        /* ctor, KPMPreconditioners.jl:124-126: cbar/sbar = model.cosht/sinht */
        for (int64_t n = 0; n < m->nb; n++) {
            P->cbar[n] = m->c[n];
            P->sbar[n] = m->s[n];
        }
    } else {
        for (int64_t n = 0; n < m->nb; n++) {
            P->cbar[n] = 0.0;
            P->sbar[n] = 0.0;
            for (int64_t tau = 0; tau < L; tau++) {
                P->cbar[n] += m->c[tau + L * n];
                P->sbar[n] += m->s[tau + L * n];
            }
            P->cbar[n] /= (double)L;
            P->sbar[n] /= (double)L;
        }
        for (int64_t i = 0; i < N; i++) P->Ebar[i] = m->E[i];
    }
}

/* KPMPreconditioners.jl:789-839 with scalar_invM :948-951.
 * The reference evaluates the sums with FFTW.dct! (unitary DCT-II) and undoes the
 * normalisation; restated here as the direct DCT-II sum it equals:
 *   c_m = (2 - delta_m0)/N_M * sum_{n<N_M} f(x_n) cos(pi m (n+1/2)/N_M),  N_M = 2*order. */
void elpho_kpm_coefficients(double *c_z, int64_t order, double lam_lo, double lam_hi, double phi) {
    int64_t M = order, NM = 2 * M;
    double lam_avg = (lam_hi + lam_lo) / 2, lam_mag = (lam_hi - lam_lo) / 2;
    double *fr = (double *)malloc(sizeof(double) * (size_t)NM);
    double *fi = (double *)malloc(sizeof(double) * (size_t)NM);
    double er = cos(phi), ei = -sin(phi); /* exp(-i phi) */
    for (int64_t n = 0; n < NM; n++) {
        double x = lam_mag * cos(M_PI * ((double)n + 0.5) / (double)NM) + lam_avg;
        double dr = 1.0 - er * x, di = -ei * x; /* 1 - exp(-i phi) x */
        double den = dr * dr + di * di;
        fr[n] = dr / den;
        fi[n] = -di / den;
    }
    for (int64_t m = 0; m < M; m++) {
        double sr = 0.0, si = 0.0;
        for (int64_t n = 0; n < NM; n++) {
            double cs = cos(M_PI * (double)m * ((double)n + 0.5) / (double)NM);
            sr += fr[n] * cs;
            si += fi[n] * cs;
        }
        double f = ((m == 0) ? 1.0 : 2.0) / (double)NM;
        c_z[2 * m] = f * sr;
        c_z[2 * m + 1] = f * si;
    }
    free(fr);
    free(fi);
}

/* A v = CBbar (Ebar .* v) — KPMPreconditioners.jl:387-401 (real arithmetic) */
static void kpm_A_real(double *vp, const elpho_kpm *P, const double *v) {
    for (int64_t i = 0; i < P->N; i++) vp[i] = P->Ebar[i] * v[i];
    elpho_checkerboard_mul_nvec(vp, P->table, P->cbar, P->sbar, P->nb);
}
/* A^-1 v — KPMPreconditioners.jl:406-420 */
static void kpm_Ainv_real(double *vp, const elpho_kpm *P, const double *v) {
    for (int64_t i = 0; i < P->N; i++) vp[i] = v[i];
    elpho_checkerboard_inverse_mul_nvec(vp, P->table, P->cbar, P->sbar, P->nb);
    for (int64_t i = 0; i < P->N; i++) vp[i] /= P->Ebar[i];
}

static double arnoldi_max_ritz(const elpho_kpm *P, int64_t n, const double *b0, int inverse) {
    int64_t m = P->N;
    double *Q = (double *)calloc((size_t)(m * (n + 1)), sizeof(double));
    double *h = (double *)calloc((size_t)((n + 1) * n), sizeof(double)); /* (n+1) x n col-major */
    double *b = (double *)malloc(sizeof(double) * (size_t)m);
    double *v = (double *)malloc(sizeof(double) * (size_t)m);
    double nrm = 0.0;
    for (int64_t i = 0; i < m; i++) nrm += b0[i] * b0[i];
    nrm = sqrt(nrm);
    for (int64_t i = 0; i < m; i++) {
        b[i] = b0[i] / nrm;
        Q[i] = b[i];
    }
    int64_t l = n;
    for (int64_t k = 0; k < n; k++) {
        if (inverse) kpm_Ainv_real(v, P, b);
        else kpm_A_real(v, P, b);
        for (int64_t j = 0; j <= k; j++) {
            const double *Qj = Q + j * m;
            double d = 0.0;
            for (int64_t i = 0; i < m; i++) d += Qj[i] * v[i];
            h[j + (n + 1) * k] = d;
            for (int64_t i = 0; i < m; i++) v[i] -= d * Qj[i];
        }
        double nv = 0.0;
        for (int64_t i = 0; i < m; i++) nv += v[i] * v[i];
        nv = sqrt(nv);
        h[(k + 1) + (n + 1) * k] = nv;
        if (nv > 1e-12) {
            for (int64_t i = 0; i < m; i++) {
                b[i] = v[i] / nv;
                Q[(k + 1) * m + i] = b[i];
            }
        } else {
            l = k + 1;
            break;
        }
    }
    double *hp = (double *)malloc(sizeof(double) * (size_t)(l * l));
    int finite = 1;
    for (int64_t j = 0; j < l; j++)
        for (int64_t i = 0; i < l; i++) {
            hp[i + l * j] = h[i + (n + 1) * j];
            if (!isfinite(hp[i + l * j])) finite = 0;
        }
    double res = INFINITY;
    if (finite) {
        double *wr = (double *)malloc(sizeof(double) * (size_t)l);
        double *wi = (double *)malloc(sizeof(double) * (size_t)l);
        if (elpho_eigvals(hp, l, wr, wi) == 0) {
            res = wr[0];
            for (int64_t i = 1; i < l; i++)
                if (wr[i] > res) res = wr[i];
        }
        free(wr);
        free(wi);
    }
    free(hp);
    free(Q);
    free(h);
    free(b);
    free(v);
    return res;
}

/* KPMPreconditioners.jl:845-942 (random start vectors injected) */
void elpho_kpm_arnoldi_bounds(const elpho_kpm *P, int64_t n, const double *b_max,
                              const double *b_min, double *e_min, double *e_max) {
    if (n > P->N) n = P->N; /* KPMPreconditioners.jl:136 */
    double emax = arnoldi_max_ritz(P, n, b_max, 0);
    double r = arnoldi_max_ritz(P, n, b_min, 1);
    *e_max = emax;
    *e_min = isfinite(r) ? 1.0 / r : -INFINITY;
}

static int jl_isapprox(double x, double y, double rtol) {
    double ax = fabs(x), ay = fabs(y);
    return x == y || (isfinite(x) && isfinite(y) && fabs(x - y) <= rtol * (ax > ay ? ax : ay));
}

/* KPMPreconditioners.jl:269-321 */
void elpho_kpm_setup_from_bounds(elpho_kpm *P, double e_min, double e_max) {
    if ((0.0 < e_min && e_min < 1.0) && (1.0 < e_max) && (e_max - e_min) < 2.0) {
        double lo = (1 - 2 * P->buf) * e_min;
        if (lo < 0.0) lo = 0.0;
        double hi = (1 + 2 * P->buf) * e_max;
        if (!jl_isapprox(lo, P->lam_lo, P->buf) || !jl_isapprox(hi, P->lam_hi, P->buf)) {
            P->lam_lo = lo;
            P->lam_hi = hi;
            P->lam_avg = (hi + lo) / 2;
            P->lam_mag = (hi - lo) / 2;
            int64_t off = 0;
            for (int64_t w = 0; w < P->Lo2; w++) {
                double phi = 2.0 * M_PI / (double)P->L * ((double)w + 0.5);
                int64_t order = (int64_t)floor((hi - lo) * (P->c1 / phi + P->c2));
                if (order < 1) order = 1;
                P->order[w] = order;
                P->coff[w] = off;
                if (off + order <= P->coeff_cap)
                    elpho_kpm_coefficients(P->coeff + 2 * off, order, lo, hi, phi);
                off += order;
            }
            P->coff[P->Lo2] = off;
        }
        P->active = 1;
    } else {
        P->active = 0;
    }
}

/* KPMPreconditioners.jl:758-778 (SymmetricKPMPreconditioner), complex vectors length N */
static void kpm_mulA_z(double *vp, elpho_kpm *P, const double *v, int transposed) {
    int64_t N = P->N;
    if (transposed) {
        memcpy(vp, v, sizeof(double) * 2 * (size_t)N);
        elpho_checkerboard_transpose_mul_nvec_z(vp, P->table, P->cbar, P->sbar, P->nb);
        for (int64_t i = 0; i < N; i++) {
            vp[2 * i] *= P->Ebar[i];
            vp[2 * i + 1] *= P->Ebar[i];
        }
    } else {
        for (int64_t i = 0; i < N; i++) {
            vp[2 * i] = P->Ebar[i] * v[2 * i];
            vp[2 * i + 1] = P->Ebar[i] * v[2 * i + 1];
        }
        elpho_checkerboard_mul_nvec_z(vp, P->table, P->cbar, P->sbar, P->nb);
    }
    P->checkerboard_count += 1;
}

/* KPMPreconditioners.jl:685-693 */
static void kpm_mulAprime_z(double *vp, elpho_kpm *P, const double *v, int transposed) {
    kpm_mulA_z(vp, P, v, transposed);
    double a = 1.0 / P->lam_mag, b = P->lam_avg / P->lam_mag;
    for (int64_t i = 0; i < 2 * P->N; i++) vp[i] = a * vp[i] - b * v[i];
}

/* one Chebyshev series  vp = sum_m cc_m T_m(A') u1, with cc = conj(c) if conjc.
 * KPMPreconditioners.jl:623-648 (first half) and :653-677 (second half) share this shape. */
static void kpm_series(double *vp, elpho_kpm *P, const double *c, int64_t order, int conjc,
                       int transposed, int from_vp, const double *v) {
    int64_t N = P->N;
    double *um1 = P->v3, *un = P->v4, *up1 = P->v5;
    double c0r = c[0], c0i = conjc ? -c[1] : c[1];
    if (from_vp) {
        /* second half: u1 = v'; v' = c1 * v' */
        if (order > 1) memcpy(un, vp, sizeof(double) * 2 * (size_t)N);
        for (int64_t i = 0; i < N; i++) {
            double xr = vp[2 * i], xi = vp[2 * i + 1];
            vp[2 * i] = c0r * xr - c0i * xi;
            vp[2 * i + 1] = c0r * xi + c0i * xr;
        }
    } else {
        for (int64_t i = 0; i < N; i++) {
            double xr = v[2 * i], xi = v[2 * i + 1];
            vp[2 * i] = c0r * xr - c0i * xi;
            vp[2 * i + 1] = c0r * xi + c0i * xr;
        }
        if (order > 1) memcpy(un, v, sizeof(double) * 2 * (size_t)N);
    }
    if (order > 1) {
        int64_t n = 1;
        kpm_mulAprime_z(up1, P, un, transposed);
        for (;;) {
            n += 1;
            double *tmp = um1;
            um1 = un;
            un = up1;
            up1 = tmp;
            double cr = c[2 * (n - 1)], ci = conjc ? -c[2 * (n - 1) + 1] : c[2 * (n - 1) + 1];
            for (int64_t i = 0; i < N; i++) {
                double xr = un[2 * i], xi = un[2 * i + 1];
                vp[2 * i] += cr * xr - ci * xi;
                vp[2 * i + 1] += cr * xi + ci * xr;
            }
            if (n == order) break;
            kpm_mulAprime_z(up1, P, un, transposed);
            for (int64_t i = 0; i < 2 * N; i++) up1[i] = 2 * up1[i] - um1[i];
        }
    }
}

/* KPMPreconditioners.jl:606-679 */
static void kpm_sym_mul(double *vp, elpho_kpm *P, const double *v, int64_t w) {
    int64_t order = P->order[w];
    const double *c = P->coeff + 2 * P->coff[w];
    kpm_series(vp, P, c, order, 1, 1, 0, v);    /* M^-T[w,w], conj coefficients */
    kpm_series(vp, P, c, order, 0, 0, 1, NULL); /* M^-1[w,w] */
}

/* KPMPreconditioners.jl:426-481 */
void elpho_kpm_apply(double *out, elpho_kpm *P, const double *in) {
    int64_t N = P->N, L = P->L;
    P->checkerboard_count = 0;
    if (!P->active) {
        memcpy(out, in, sizeof(double) * (size_t)(N * L));
        return;
    }
    double *v1 = P->v1, *v2 = P->v2;
    elpho_tau_to_omega(v2, in, N, L);
    /* transpose!(a1T, a2): a1T[i,w] = a2[w,i] */
    for (int64_t i = 0; i < N; i++)
        for (int64_t w = 0; w < L; w++) {
            v1[2 * (i + N * w)] = v2[2 * (w + L * i)];
            v1[2 * (i + N * w) + 1] = v2[2 * (w + L * i) + 1];
        }
    for (int64_t w = 0; w < P->Lo2; w++) {
        const double *u1 = v1 + 2 * N * w;
        double *u2 = v2 + 2 * N * w;
        kpm_sym_mul(u2, P, u1, w);
        double *u2c = v2 + 2 * N * (L - 1 - w);
        for (int64_t i = 0; i < N; i++) {
            double re = u2[2 * i], im = u2[2 * i + 1];
            u2c[2 * i] = re;
            u2c[2 * i + 1] = -im;
        }
    }
    /* transpose!(a1, a2T): a1[w,i] = a2T[i,w] */
    for (int64_t w = 0; w < L; w++)
        for (int64_t i = 0; i < N; i++) {
            v1[2 * (w + L * i)] = v2[2 * (i + N * w)];
            v1[2 * (w + L * i) + 1] = v2[2 * (i + N * w) + 1];
        }
    elpho_omega_to_tau(out, v1, N, L);
}

/* ====================================================================== */
/* conjugate gradient                                                      */
/* ====================================================================== */

static double dotp(const double *a, const double *b, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* IterativeSolvers.jl:239-314 (no preconditioner) and :153-234 (preconditioner) */
int64_t elpho_cg_solve(const elpho_model *m, double *x, const double *b, double tol,
                       int64_t maxiter, double kmax, elpho_kpm *P, double *r, double *p, double *z,
                       double *hist) {
    const int64_t n = m->N * m->L;
    double normb = sqrt(dotp(b, b, n));
    /* r0 = b - A x0 */
    elpho_mulMTM(r, m, x);
    for (int64_t i = 0; i < n; i++) r[i] = 1.0 * b[i] + -1.0 * r[i]; /* axpby!(1,b,-1,r) */
    double rdotz;
    if (P) {
        elpho_kpm_apply(z, P, r);
        memcpy(p, z, sizeof(double) * (size_t)n);
        rdotz = dotp(r, z, n);
    } else {
        memcpy(p, r, sizeof(double) * (size_t)n);
        rdotz = dotp(r, r, n);
    }
    double eps0 = sqrt(dotp(r, r, n)) / normb;
    double eps = eps0;
    double kmin = 0.0;
    if (hist) hist[0] = eps0;
    for (int64_t j = 1; j <= maxiter; j++) {
        elpho_mulMTM(z, m, p);
        double alpha = rdotz / dotp(p, z, n);
        for (int64_t i = 0; i < n; i++) x[i] += alpha * p[i];
        for (int64_t i = 0; i < n; i++) r[i] += -alpha * z[i];
        eps = sqrt(dotp(r, r, n)) / normb;
        if (hist) hist[j] = eps;
        double lg = log(2 * eps0 / eps);
        double q = (2 * (double)j / lg);
        double val = q * q;
        kmin = (val > kmin) ? val : kmin; /* @fastmath max: NaN never wins */
        if (eps < tol || kmin > kmax) return j;
        double new_rdotz;
        if (P) {
            elpho_kpm_apply(z, P, r);
            new_rdotz = dotp(r, z, n);
        } else {
            new_rdotz = dotp(r, r, n);
        }
        double beta = new_rdotz / rdotz;
        rdotz = new_rdotz;
        const double *zz = P ? z : r;
        for (int64_t i = 0; i < n; i++) p[i] = 1.0 * zz[i] + beta * p[i]; /* axpby!(1,z,beta,p) */
    }
    return maxiter;
}

static void ldiv_noP(const elpho_model *m, double *x, const double *b, int64_t maxiter,
                     double solver_tol, int64_t solver_maxiter, double kmax, double *r, double *p,
                     double *z, int64_t *iters, double *resid, int64_t *flag) {
    const int64_t n = m->N * m->L;
    if (maxiter == 0) maxiter = solver_maxiter;
    *iters = elpho_cg_solve(m, x, b, solver_tol, maxiter, kmax, NULL, r, p, z, NULL);
    double *v = m->vppp;
    elpho_mulMTM(v, m, x);
    for (int64_t i = 0; i < n; i++) v[i] = v[i] - b[i];
    *resid = sqrt(dotp(v, v, n)) / sqrt(dotp(b, b, n));
    if (*resid > sqrt(solver_tol)) {
        *flag = (*iters == solver_maxiter) ? 1 : 2; /* Models.jl:160 compares solver.maxiter */
        memset(x, 0, sizeof(double) * (size_t)n);
    } else {
        *flag = 0;
    }
}

/* Models.jl:74-137 (P != NULL), :139-186 (P == NULL) */
void elpho_ldiv(const elpho_model *m, double *x, const double *b, elpho_kpm *P, int64_t maxiter,
                double solver_tol, int64_t solver_maxiter, double kmax, double *r, double *p,
                double *z, int64_t *iters, double *resid, int64_t *flag) {
    const int64_t n = m->N * m->L;
    if (maxiter == 0) maxiter = solver_maxiter;
    if (!P) {
        ldiv_noP(m, x, b, maxiter, solver_tol, solver_maxiter, kmax, r, p, z, iters, resid, flag);
        return;
    }
    *iters = elpho_cg_solve(m, x, b, solver_tol, maxiter, kmax, P, r, p, z, NULL);
    double *v = m->vppp;
    elpho_mulMTM(v, m, x);
    for (int64_t i = 0; i < n; i++) v[i] = v[i] - b[i];
    *resid = sqrt(dotp(v, v, n)) / sqrt(dotp(b, b, n));
    if (*resid > sqrt(solver_tol)) {
        *flag = (*iters == maxiter) ? 1 : 2;
        memset(x, 0, sizeof(double) * (size_t)n);
    } else {
        *flag = 0;
    }
    if (*flag > 0)
        ldiv_noP(m, x, b, 10 * maxiter, solver_tol, solver_maxiter, kmax, r, p, z, iters, resid, flag);
}

/* ====================================================================== */
/* callers' helpers                                                        */
/* ====================================================================== */

/* HMC.jl:921-941 */
void elpho_update_Lambda(double *Lam, int64_t N, int64_t L, double dtau, const double *x,
                         const double *lambda, const double *lambda2) {
    for (int64_t i = 0; i < N; i++)
        for (int64_t tau = 0; tau < L; tau++) {
            double xt = x[i * L + tau];
            Lam[i * L + tau] = exp(-dtau * (lambda[i] * xt + lambda2[i] * (xt * xt)) / 2);
        }
}

/* HMC.jl:951-968 */
void elpho_mulLambda(double *out, const double *in, const double *Lam, int64_t N, int64_t L) {
    for (int64_t i = 0; i < N; i++) {
        double u1 = in[i * L];
        for (int64_t tau = 0; tau < L - 1; tau++)
            out[i * L + tau] = -Lam[i * L + tau + 1] * in[i * L + tau + 1];
        out[i * L + L - 1] = Lam[i * L] * u1;
    }
}

/* HMC.jl:978-995 */
void elpho_mulLambdaInv(double *out, const double *in, const double *Lam, int64_t N, int64_t L) {
    for (int64_t i = 0; i < N; i++) {
        double uL = in[i * L + L - 1];
        for (int64_t tau = L - 1; tau >= 1; tau--)
            out[i * L + tau] = -(1.0 / Lam[i * L + tau]) * in[i * L + tau - 1];
        out[i * L] = (1.0 / Lam[i * L]) * uL;
    }
}

/* HolsteinModels.jl:691-755 */
void elpho_muldMdx_holstein(double *dMdx, const double *u, const elpho_model *m, const double *v,
                            double dtau, const double *lambda, const double *lambda2,
                            const double *x) {
    const int64_t N = m->N, L = m->L;
    double *y = m->vp;
    for (int64_t i = 0; i < N; i++) {
        int64_t i1 = i * L, iL = i * L + L - 1;
        dMdx[i1] = -dtau * (lambda[i] + 2 * lambda2[i] * x[i1]) * m->E[i1] * v[iL];
        for (int64_t tau = 1; tau < L; tau++) {
            int64_t it = i * L + tau;
            dMdx[it] = dtau * (lambda[i] + 2 * lambda2[i] * x[it]) * m->E[it] * v[it - 1];
        }
    }
    memcpy(y, u, sizeof(double) * (size_t)(N * L));
    if (m->nb > 0) elpho_checkerboard_transpose_mul(y, m->table, m->c, m->s, m->nb, L);
    for (int64_t i = 0; i < N * L; i++) dMdx[i] = y[i] * dMdx[i];
}

/* HMC.jl:1005-1025: dLdx[n] += vl[n] * (sg dtau (lambda/2 + lambda2 x[n])) Lam[n] * vr[n'], n' = previous tau (wrapped),
 * sg = -1 for tau = 1 (0-based 0), +1 otherwise */
void elpho_muldLambdadx_holstein(double *dLdx, const double *vl, const double *vr, const double *Lam, int64_t N,
                                 int64_t L, double dtau, const double *lambda, const double *lambda2, const double *x) {
    for (int64_t i = 0; i < N; i++) {
        int64_t n = i * L, np = i * L + L - 1;
        dLdx[n] += vl[n] * (-dtau * (lambda[i] / 2 + lambda2[i] * x[n])) * Lam[n] * vr[np];
        for (int64_t tau = 1; tau < L; tau++) {
            n = i * L + tau;
            np = n - 1;
            dLdx[n] += vl[n] * (dtau * (lambda[i] / 2 + lambda2[i] * x[n])) * Lam[n] * vr[np];
        }
    }
}

/* HMC.jl:790-814: dSfdx += -dMdx(M X+, X+) - dMdx(M X-, X-) + dLdx(phi+, X+) + dLdx(phi-, X-).
 * scratch: u[N*L], d[N*L] */
void elpho_calc_dSfdx_holstein(double *dSfdx, const elpho_model *m, const double *Xp, const double *Xm,
                               const double *phip, const double *phim, const double *Lam, double dtau,
                               const double *lambda, const double *lambda2, const double *x, double *u, double *d) {
    const int64_t n = m->N * m->L;
    const double *X[2] = {Xp, Xm};
    const double *phi[2] = {phip, phim};
    for (int k = 0; k < 2; k++) {
        elpho_mulM(u, m, X[k]);
        elpho_muldMdx_holstein(d, u, m, X[k], dtau, lambda, lambda2, x);
        for (int64_t i = 0; i < n; i++) dSfdx[i] += -d[i];
    }
    for (int k = 0; k < 2; k++)
        elpho_muldLambdadx_holstein(dSfdx, phi[k], X[k], Lam, m->N, m->L, dtau, lambda, lambda2, x);
}

/* ====================================================================== */
/* HMC trajectory                                           */
/* ====================================================================== */

/* PhononAction.jl:11-66 (Holstein, shifted = false, no dispersive modes: the reference's loop over them reads an
 * undefined variable `L`, :49, so a deck with dispersion cannot run through calc_Sb there) */
double elpho_calc_Sb_holstein(int64_t N, int64_t L, double dtau, const double *x, const double *omega,
                              const double *omega4) {
    double Sb = 0.0;
    for (int64_t i = 0; i < N; i++)
        for (int64_t tau = 0; tau < L; tau++) {
            int64_t tm1 = (tau + L - 1) % L;
            double xt = x[i * L + tau], xm = x[i * L + tm1];
            Sb += omega[i] * omega[i] * (xt * xt) / 2 + omega4[i] * (xt * xt * xt * xt);
            Sb += (xt - xm) * (xt - xm) / (dtau * dtau) / 2;
        }
    return dtau * Sb;
}

/* PhononAction.jl:114-187 (accumulates) */
void elpho_calc_dSbdx_holstein(double *dSbdx, int64_t N, int64_t L, double dtau, const double *x, const double *omega,
                               const double *omega4) {
    for (int64_t i = 0; i < N; i++) {
        double a = dtau * omega[i] * omega[i], b = dtau * 4 * omega4[i];
        for (int64_t tau = 0; tau < L; tau++) {
            int64_t tp1 = (tau + 1) % L, tm1 = (tau + L - 1) % L, n = i * L + tau;
            double xt = x[n];
            dSbdx[n] += a * xt;
            dSbdx[n] += b * xt * xt * xt;
            dSbdx[n] -= (x[i * L + tp1] + x[i * L + tm1] - 2.0 * xt) / dtau;
        }
    }
}

typedef struct {
    const elpho_hmc_params *hp;
    const elpho_hmc_ssh *ssh;       /* NULL: Holstein */
    elpho_model *m;
    elpho_kpm *P;
    double *x, *v;
    double *phi[2], *Lphi[2], *X[2], *Lam, *u, *y, *dSdx, *r, *p, *z;
    const double *kpm_randn;
    int64_t kpm_calls;
    int64_t nf;                     /* phonon columns: Nsites (Holstein) or Nph (SSH) */
    double solver_iters;
} hmc_ws;

static void hmc_update_model(hmc_ws *w) {
    const elpho_hmc_params *hp = w->hp;
    if (w->ssh) {                                                 /* SSHModels.jl:510-562 */
        const elpho_hmc_ssh *q = w->ssh;
        elpho_update_model_ssh(hp->N, hp->L, w->m->nb, q->Nph, hp->dtau, w->x, q->t, q->alpha, q->alpha2, hp->mu,
                               q->phonon_to_bond, q->cb_perm, (double *)w->m->c, (double *)w->m->s, (double *)w->m->E);
        return;
    }
    elpho_update_model_holstein(hp->N, hp->L, hp->dtau, w->x, hp->lambda, hp->lambda2, hp->mu, (double *)w->m->E);
}

/* HMC.jl:820-915 (CG branch): returns iters, sets *flag */
static int64_t hmc_calc_OinvLphi(hmc_ws *w, double power, int64_t *flag) {
    const elpho_hmc_params *hp = w->hp;
    const int64_t n = hp->N * hp->L;
    const double tol = pow(hp->solver_tol, power);
    if (w->P) {                                                   /* setup!(P), KPMPreconditioners.jl:259-321 */
        const double *bmax = w->kpm_randn + (2 * w->kpm_calls) * hp->N, *bmin = bmax + hp->N;
        double emin, emax;
        w->kpm_calls++;
        elpho_kpm_update_A(w->P, w->m);
        elpho_kpm_arnoldi_bounds(w->P, hp->kpm_n, bmax, bmin, &emin, &emax);
        elpho_kpm_setup_from_bounds(w->P, emin, emax);
    }
    if (w->ssh) {                                                 /* Λ ≡ 1: update_Λ!/mulΛ! are no-ops (HMC.jl:943-946,970-973) */
        memcpy(w->Lphi[0], w->phi[0], sizeof(double) * (size_t)n);
        memcpy(w->Lphi[1], w->phi[1], sizeof(double) * (size_t)n);
    } else {
        elpho_update_Lambda(w->Lam, hp->N, hp->L, hp->dtau, w->x, hp->lambda, hp->lambda2);
        elpho_mulLambda(w->Lphi[0], w->phi[0], w->Lam, hp->N, hp->L);
        elpho_mulLambda(w->Lphi[1], w->phi[1], w->Lam, hp->N, hp->L);
    }
    int64_t iters = 0, it, fl = 0;
    double res;
    for (int k = 0; k < 2 && fl == 0; k++) {
        memset(w->X[k], 0, sizeof(double) * (size_t)n);
        elpho_ldiv(w->m, w->X[k], w->Lphi[k], w->P, 0, tol, hp->solver_maxiter, hp->kmax,
                   w->r, w->p, w->z, &it, &res, &fl);
        iters += it;
    }
    if (fl == 0) iters = (iters + 1) / 2;                          /* cld(iters, 2) */
    *flag = fl;
    return iters;
}

static double hmc_calc_Sf(hmc_ws *w) {                             /* HMC.jl:768-784 */
    const int64_t n = w->hp->N * w->hp->L;
    double Sf = dotp(w->Lphi[0], w->X[0], n) / 2;
    Sf += dotp(w->Lphi[1], w->X[1], n) / 2;
    return Sf;
}

/* dMdx[primary_field[f]] += dMdx[f] for the secondary fields, then dMdx = dMdx[primary_field]  (muldMdx!, SSHModels.jl:820-826) */
static void ssh_share_force(double *d, const int64_t *pf, int64_t nfl) {
    if (!pf) return;
    for (int64_t f = 0; f < nfl; f++)
        if (pf[f] != f) d[pf[f]] += d[f];
    for (int64_t f = 0; f < nfl; f++) d[f] = d[pf[f]];
}

static void hmc_calc_H(hmc_ws *w, double *H, double *S, double *K) {   /* HMC.jl:697-721,745-756 */
    const elpho_hmc_params *hp = w->hp;
    const int64_t nfl = w->nf * hp->L, L = hp->L;
    const int64_t *pf = w->ssh ? w->ssh->primary_field : NULL;
    *S = hmc_calc_Sf(w);
    elpho_fourier_accelerate(w->y, w->v, hp->fa_M, 1.0, w->nf, hp->L);
    if (pf) {      /* SSH with shared fields: calc_Sb (PhononAction.jl:68-97) and calc_K (HMC.jl:720-738) over primary fields */
        double k = 0.0;
        for (int64_t i = 0; i < w->nf; i++)
            if (pf[i * L] == i * L) *S += elpho_calc_Sb_holstein(1, L, hp->dtau, w->x + i * L, hp->omega + i, hp->omega4 + i);
        for (int64_t f = 0; f < nfl; f++)
            if (pf[f] == f) k += w->v[f] * w->y[f] / 2;
        *K = k;
    } else {
        /* calc_Sb: the SSH version with every field its own primary field is the Holstein sum over Nph columns */
        *S += elpho_calc_Sb_holstein(w->nf, hp->L, hp->dtau, w->x, hp->omega, hp->omega4);
        *K = dotp(w->v, w->y, nfl) / 2;
    }
    *H = *S + *K;
}

static void hmc_calc_dSfdx(hmc_ws *w) {                            /* HMC.jl:790-814 */
    const elpho_hmc_params *hp = w->hp;
    if (w->ssh) {                                                  /* dSf/dx += -dMdx(M X±, X±); muldΛdx! is a no-op (:1027-1030) */
        const elpho_hmc_ssh *q = w->ssh;
        const int64_t nfl = w->nf * hp->L;
        for (int k = 0; k < 2; k++) {
            elpho_mulM(w->u, w->m, w->X[k]);
            elpho_muldMdx_ssh(w->y, w->u, w->m, w->X[k], hp->dtau, q->bond_to_phonon_cb, q->alpha, q->alpha2, w->x, q->Nph);
            ssh_share_force(w->y, q->primary_field, nfl);
            for (int64_t i = 0; i < nfl; i++) w->dSdx[i] += -w->y[i];
        }
        return;
    }
    elpho_calc_dSfdx_holstein(w->dSdx, w->m, w->X[0], w->X[1], w->phi[0], w->phi[1], w->Lam, hp->dtau, hp->lambda,
                              hp->lambda2, w->x, w->u, w->y);
}

/* HMC.jl:343-463 (standard_update!, nb == 1) and :469-638 (multitimestep_update!, nb > 1); update! :313-337.
 * The random numbers the reference draws from model.rng are inputs: R[Ndof] (refresh_v!, :648-659),
 * Rp, Rm [Ndim] (refresh_ϕ!, :665-692), kpm_randn[(nt+2)*2*N] (one pair of Arnoldi start vectors per setup!,
 * consumed in call order; may be NULL when P is NULL) and the uniform u of the accept/reject step (:441,:617).
 * out[0..7] = H0, H1, S (last calc_H), K (last calc_H), returned iters = cld(iters, nt+2), flag, P_accept, solver calls.
 * ssh == NULL: Holstein (fields on sites, Λ from x); ssh != NULL: SSH (Nph bond-phonon columns, Λ ≡ 1, hp->omega,
 * omega4, fa_M per phonon; hp->lambda, lambda2 unused). */
static int64_t hmc_update_generic(const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P, double *x,
                                  double *v, const double *R, const double *Rp, const double *Rm, const double *kpm_randn,
                                  double u, double *out) {
    const int64_t N = hp->N, L = hp->L, n = N * L, nt = hp->nt, nb = hp->nb;
    const int64_t nf = ssh ? ssh->Nph : N, nfl = nf * L, nmax = (nfl > n) ? nfl : n;
    const double dt = hp->dt, dtp = hp->dt / (double)hp->nb;
    double *buf = (double *)calloc((size_t)(16 * nmax), sizeof(double));
    hmc_ws w;
    memset(&w, 0, sizeof w);
    w.hp = hp; w.ssh = ssh; w.m = m; w.P = P; w.x = x; w.v = v; w.kpm_randn = kpm_randn; w.nf = nf;
    double *q = buf;
    w.phi[0] = q; q += nmax; w.phi[1] = q; q += nmax; w.Lphi[0] = q; q += nmax; w.Lphi[1] = q; q += nmax;
    w.X[0] = q; q += nmax; w.X[1] = q; q += nmax; w.Lam = q; q += nmax; w.u = q; q += nmax; w.y = q; q += nmax;
    w.dSdx = q; q += nmax; w.r = q; q += nmax; w.p = q; q += nmax; w.z = q; q += nmax;
    double *x0 = q; q += nmax;
    double *v0 = q; q += nmax;
    double *Q = w.dSdx;                                            /* QdSdx aliases dSdx (:349) */
    int64_t flag = 0, iters = 0, itrs;
    double H0 = 0, H1 = 0, S = 0, K = 0;

    hmc_update_model(&w);
    /* refresh_v! (:648-659) */
    elpho_fourier_accelerate(w.y, R, hp->fa_M, -0.5, nf, L);
    for (int64_t i = 0; i < nfl; i++) v[i] = hp->alpha * v[i] + sqrt(1.0 - hp->alpha * hp->alpha) * w.y[i];
    memcpy(x0, x, sizeof(double) * (size_t)nfl);
    memcpy(v0, v, sizeof(double) * (size_t)nfl);
    /* refresh_ϕ! (:665-692): ϕ± = Λ⁻¹ Mᵀ R± */
    if (ssh) {
        elpho_mulMT(w.phi[0], m, Rp);
        elpho_mulMT(w.phi[1], m, Rm);
    } else {
        elpho_update_Lambda(w.Lam, N, L, hp->dtau, x, hp->lambda, hp->lambda2);
        elpho_mulMT(w.Lphi[0], m, Rp);
        elpho_mulLambdaInv(w.phi[0], w.Lphi[0], w.Lam, N, L);
        elpho_mulMT(w.Lphi[1], m, Rm);
        elpho_mulLambdaInv(w.phi[1], w.Lphi[1], w.Lam, N, L);
    }

    itrs = hmc_calc_OinvLphi(&w, 2.0, &flag);
    if (nb == 1) iters = itrs;                                     /* :373;  the multi-timestep variant has "iters += iters" (:507) */
    if (flag == 0) {
        hmc_calc_H(&w, &H0, &S, &K);
        memset(w.dSdx, 0, sizeof(double) * (size_t)nfl);
        hmc_calc_dSfdx(&w);
        if (nb == 1) elpho_calc_dSbdx_holstein(w.dSdx, nf, L, hp->dtau, x, hp->omega, hp->omega4);
        elpho_fourier_accelerate(Q, w.dSdx, hp->fa_M, -1.0, nf, L);
        for (int64_t t = 1; t <= nt; t++) {
            for (int64_t i = 0; i < nfl; i++) v[i] = v[i] - dt / 2 * Q[i];
            if (nb == 1) {
                for (int64_t i = 0; i < nfl; i++) x[i] = x[i] + dt * v[i];
            } else {
                memset(w.dSdx, 0, sizeof(double) * (size_t)nfl);
                elpho_calc_dSbdx_holstein(w.dSdx, nf, L, hp->dtau, x, hp->omega, hp->omega4);
                elpho_fourier_accelerate(Q, w.dSdx, hp->fa_M, -1.0, nf, L);
                for (int64_t tp = 1; tp <= nb; tp++) {
                    for (int64_t i = 0; i < nfl; i++) v[i] = v[i] - dtp / 2 * Q[i];
                    for (int64_t i = 0; i < nfl; i++) x[i] = x[i] + dtp * v[i];
                    memset(w.dSdx, 0, sizeof(double) * (size_t)nfl);
                    elpho_calc_dSbdx_holstein(w.dSdx, nf, L, hp->dtau, x, hp->omega, hp->omega4);
                    elpho_fourier_accelerate(Q, w.dSdx, hp->fa_M, -1.0, nf, L);
                    for (int64_t i = 0; i < nfl; i++) v[i] = v[i] - dtp / 2 * Q[i];
                }
            }
            hmc_update_model(&w);
            itrs = hmc_calc_OinvLphi(&w, 1.0, &flag);
            iters += itrs;
            if (flag > 0) break;
            memset(w.dSdx, 0, sizeof(double) * (size_t)nfl);
            hmc_calc_dSfdx(&w);
            if (nb == 1) elpho_calc_dSbdx_holstein(w.dSdx, nf, L, hp->dtau, x, hp->omega, hp->omega4);
            elpho_fourier_accelerate(Q, w.dSdx, hp->fa_M, -1.0, nf, L);
            for (int64_t i = 0; i < nfl; i++) v[i] = v[i] - dt / 2 * Q[i];
        }
    }
    double Pacc = 0.0;
    if (flag == 0) {
        itrs = hmc_calc_OinvLphi(&w, 2.0, &flag);
        iters += itrs;
        if (flag == 0) {
            hmc_calc_H(&w, &H1, &S, &K);
            double dH = H1 - H0, e = exp(-dH);
            Pacc = (1.0 < e) ? 1.0 : e;                            /* min(1, exp(-ΔH)) */
        }
    }
    int64_t accepted = (u < Pacc && flag == 0) ? 1 : 0;
    if (!accepted) {
        memcpy(x, x0, sizeof(double) * (size_t)nfl);
        for (int64_t i = 0; i < nfl; i++) v[i] = -v0[i];
        hmc_update_model(&w);
    }
    out[0] = H0; out[1] = H1; out[2] = S; out[3] = K;
    out[4] = (double)((iters + (nt + 2) - 1) / (nt + 2));          /* cld(iters, Nt+2) */
    out[5] = (double)flag; out[6] = Pacc; out[7] = (double)w.kpm_calls;
    free(buf);
    return accepted;
}

int64_t elpho_hmc_update_holstein(const elpho_hmc_params *hp, elpho_model *m, elpho_kpm *P, double *x, double *v,
                                  const double *R, const double *Rp, const double *Rm, const double *kpm_randn, double u,
                                  double *out) {
    return hmc_update_generic(hp, NULL, m, P, x, v, R, Rp, Rm, kpm_randn, u, out);
}

/* The same update for the SSH model (bond phonons): x, v, R have Nph*L entries, Rp, Rm N*L; m->c, m->s, m->E writable. */
int64_t elpho_hmc_update_ssh(const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P, double *x,
                             double *v, const double *R, const double *Rp, const double *Rm, const double *kpm_randn, double u,
                             double *out) {
    return hmc_update_generic(hp, ssh, m, P, x, v, R, Rp, Rm, kpm_randn, u, out);
}

/* ====================================================================== */
/* Special updates (SpecialUpdates.jl): one proposed move                   */
/* ====================================================================== */

/* The body of the loops of special_update! — ReflectionUpdate (SpecialUpdates.jl:103-136: x_i(τ) → −x_i(τ) on one site) and
 * SwapUpdate (:205-236 Holstein, :241-275 SSH: two phonon columns exchange their world lines):
 *   S₀ = refresh_ϕ!(hmc, model, sample_R = true)      fresh pseudofermions for the CURRENT field, S₀ = (R₊² + R₋²)/2 + S_b
 *   apply the move, update_model!, calc_O⁻¹Λϕ!(…, 2.0), S₁ = calc_S, accept iff u < min(1, e^{−(S₁−S₀)}) and flag == 0,
 *   otherwise undo the move and update_model! again.
 * kind 0: reflect column ci; kind 1: swap columns ci, cj (0-based phonon columns).  ssh == NULL: Holstein.
 * out[0..4] = S₀, S₁, iters, flag, acceptance probability.  Returns accepted. */
int64_t elpho_special_move(const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P, double *x, int kind,
                           int64_t ci, int64_t cj, const double *Rp, const double *Rm, const double *kpm_randn, double u,
                           double *out) {
    const int64_t N = hp->N, L = hp->L, n = N * L;
    const int64_t nf = ssh ? ssh->Nph : N, nfl = nf * L, nmax = (nfl > n) ? nfl : n;
    double *buf = (double *)calloc((size_t)(13 * nmax), sizeof(double));
    hmc_ws w;
    memset(&w, 0, sizeof w);
    w.hp = hp; w.ssh = ssh; w.m = m; w.P = P; w.x = x; w.kpm_randn = kpm_randn; w.nf = nf;
    double *q = buf;
    w.phi[0] = q; q += nmax; w.phi[1] = q; q += nmax; w.Lphi[0] = q; q += nmax; w.Lphi[1] = q; q += nmax;
    w.X[0] = q; q += nmax; w.X[1] = q; q += nmax; w.Lam = q; q += nmax; w.u = q; q += nmax; w.y = q; q += nmax;
    w.dSdx = q; q += nmax; w.r = q; q += nmax; w.p = q; q += nmax; w.z = q; q += nmax;
    hmc_update_model(&w);
    /* refresh_ϕ!(…, sample_R = true), HMC.jl:665-692 */
    if (ssh) {
        elpho_mulMT(w.phi[0], m, Rp);
        elpho_mulMT(w.phi[1], m, Rm);
    } else {
        elpho_update_Lambda(w.Lam, N, L, hp->dtau, x, hp->lambda, hp->lambda2);
        elpho_mulMT(w.Lphi[0], m, Rp);
        elpho_mulLambdaInv(w.phi[0], w.Lphi[0], w.Lam, N, L);
        elpho_mulMT(w.Lphi[1], m, Rm);
        elpho_mulLambdaInv(w.phi[1], w.Lphi[1], w.Lam, N, L);
    }
    const double S0 = dotp(Rp, Rp, n) / 2 + dotp(Rm, Rm, n) / 2 + elpho_calc_Sb_holstein(nf, L, hp->dtau, x, hp->omega, hp->omega4);
    for (int rep = 0; rep < 2; rep++) {                            /* rep 0: the move; rep 1: its undo when rejected */
        for (int64_t t = 0; t < L; t++) {
            if (kind == 0) x[ci * L + t] = -x[ci * L + t];
            else { const double a = x[ci * L + t]; x[ci * L + t] = x[cj * L + t]; x[cj * L + t] = a; }
        }
        hmc_update_model(&w);
        if (rep == 1) break;
        int64_t flag = 0;
        const int64_t iters = hmc_calc_OinvLphi(&w, 2.0, &flag);
        const double S1 = hmc_calc_Sf(&w) + elpho_calc_Sb_holstein(nf, L, hp->dtau, x, hp->omega, hp->omega4);
        const double e = exp(-(S1 - S0)), Pf = (1.0 < e) ? 1.0 : e;
        out[0] = S0; out[1] = S1; out[2] = (double)iters; out[3] = (double)flag; out[4] = Pf;
        if (u < Pf && flag == 0) { free(buf); return 1; }
    }
    free(buf);
    return 0;
}

/* ====================================================================== */
/* Langevin dynamics (LangevinDynamics.jl) — Holstein                      */
/* ====================================================================== */

/* PhononAction.jl:114-187 with shifted = true: the -Δτ λ term of the particle-hole shifted action (accumulates) */
static void calc_dSbdx_holstein_shifted(double *dSbdx, int64_t N, int64_t L, double dtau, const double *x, const double *omega,
                                        const double *omega4, const double *lambda) {
    elpho_calc_dSbdx_holstein(dSbdx, N, L, dtau, x, omega, omega4);
    for (int64_t i = 0; i < N; i++)
        for (int64_t tau = 0; tau < L; tau++) dSbdx[i * L + tau] -= dtau * lambda[i];
}

/* calc_dSdx! = calc_dSfdx! + calc_dSbdx!(…, true)  (LangevinDynamics.jl:334-384): g is the noise vector the reference draws
 * with randn!(model.rng, g); solve MᵀM x = Mᵀg; dSf/dx = -2 gᵀ (∂M/∂x) M⁻¹g (muldMdx! of the model).  The flag of ldiv! is
 * ignored there (a failed solve leaves M⁻¹g = 0).  ssh == NULL: Holstein; otherwise bond phonons (dSdx has Nph*L entries, the
 * shifted flag changes nothing, PhononAction.jl:189-234).  Returns the iteration count.  work: 5 max(N, Nph) L doubles. */
static int64_t langevin_dSdx(double *dSdx, const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P,
                             const double *x, const double *g, const double *b_max, const double *b_min, double *Minv_g, double *work) {
    const int64_t N = hp->N, L = hp->L, n = N * L;
    const int64_t nf = ssh ? ssh->Nph : N, nfl = nf * L, nmax = (nfl > n) ? nfl : n;
    double *b = work, *r = work + nmax, *p = work + 2 * nmax, *z = work + 3 * nmax, *d = work + 4 * nmax;
    int64_t it = 0, fl = 0;
    double res;
    if (P) {                                                       /* setup!(preconditioner), :366 */
        double emin, emax;
        elpho_kpm_update_A(P, m);
        elpho_kpm_arnoldi_bounds(P, hp->kpm_n, b_max, b_min, &emin, &emax);
        elpho_kpm_setup_from_bounds(P, emin, emax);
    }
    memset(Minv_g, 0, sizeof(double) * (size_t)n);
    elpho_mulMT(b, m, g);
    elpho_ldiv(m, Minv_g, b, P, 0, hp->solver_tol, hp->solver_maxiter, hp->kmax, r, p, z, &it, &res, &fl);
    if (ssh) {
        elpho_muldMdx_ssh(d, g, m, Minv_g, hp->dtau, ssh->bond_to_phonon_cb, ssh->alpha, ssh->alpha2, x, ssh->Nph);
        ssh_share_force(d, ssh->primary_field, nfl);
        for (int64_t i = 0; i < nfl; i++) dSdx[i] = -2.0 * d[i];
        elpho_calc_dSbdx_holstein(dSdx, nf, L, hp->dtau, x, hp->omega, hp->omega4);
    } else {
        elpho_muldMdx_holstein(d, g, m, Minv_g, hp->dtau, hp->lambda, hp->lambda2, x);
        for (int64_t i = 0; i < n; i++) dSdx[i] = -2.0 * d[i];
        calc_dSbdx_holstein_shifted(dSdx, N, L, hp->dtau, x, hp->omega, hp->omega4, hp->lambda);
    }
    return it;
}

int64_t elpho_langevin_dSdx(double *dSdx, const elpho_hmc_params *hp, elpho_model *m, elpho_kpm *P, const double *x, const double *g,
                            const double *b_max, const double *b_min, double *Minv_g, double *work) {
    return langevin_dSdx(dSdx, hp, NULL, m, P, x, g, b_max, b_min, Minv_g, work);
}

static void langevin_update_model(const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, const double *x) {
    if (ssh)
        elpho_update_model_ssh(hp->N, hp->L, m->nb, ssh->Nph, hp->dtau, x, ssh->t, ssh->alpha, ssh->alpha2, hp->mu, ssh->phonon_to_bond,
                               ssh->cb_perm, (double *)m->c, (double *)m->s, (double *)m->E);
    else
        elpho_update_model_holstein(hp->N, hp->L, hp->dtau, x, hp->lambda, hp->lambda2, hp->mu, (double *)m->E);
}

/* evolve!(model, dyn, fa, P) — scheme 0: EulerDynamics (:81-130), 1: RungeKuttaDynamics (:162-232), 2: HeunsDynamics
 * (:272-328).  fa_Q = FourierAccelerator.Q (fourier_accelerate! without use_mass).  eta [Ndof], g1, g2 [N L] (the second
 * noise vector of the two-stage schemes), kpm_randn: b_max, b_min of the first and of the second set-up (4 N doubles).
 * x is advanced in place; the model tables follow (update_model!).  Returns the reference's iteration count. */
static int64_t langevin_evolve(int scheme, const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P, double *x,
                               const double *fa_Q, double dt, const double *eta, const double *g1, const double *g2,
                               const double *kpm_randn) {
    const int64_t N = hp->N, L = hp->L;
    const int64_t nf = ssh ? ssh->Nph : N, n = nf * L, nmax = (n > N * L) ? n : N * L;
    double *buf = (double *)calloc((size_t)(12 * nmax), sizeof(double));
    double *F1 = buf, *F2 = buf + nmax, *xi = buf + 2 * nmax, *dx = buf + 3 * nmax, *Mg = buf + 4 * nmax, *Q = buf + 5 * nmax,
           *work = buf + 6 * nmax;
    const double *bm1 = kpm_randn, *bn1 = kpm_randn ? kpm_randn + N : NULL;
    const double *bm2 = kpm_randn ? kpm_randn + 2 * N : NULL, *bn2 = kpm_randn ? kpm_randn + 3 * N : NULL;
    int64_t iters = 0;
    langevin_update_model(hp, ssh, m, x);
    if (scheme == 0) {
        iters = langevin_dSdx(F1, hp, ssh, m, P, x, g1, bm1, bn1, Mg, work);
        elpho_fourier_accelerate(Q, F1, fa_Q, 1.0, nf, L);
        elpho_fourier_accelerate(xi, eta, fa_Q, 0.5, nf, L);
        for (int64_t i = 0; i < n; i++) x[i] += sqrt(2.0 * dt) * xi[i] - dt * Q[i];
    } else if (scheme == 1) {
        (void)langevin_dSdx(F1, hp, ssh, m, P, x, g1, bm1, bn1, Mg, work);
        for (int64_t i = 0; i < n; i++) { dx[i] = sqrt(2 * dt) * eta[i] - dt * F1[i]; x[i] = x[i] + dx[i]; }
        langevin_update_model(hp, ssh, m, x);
        iters = langevin_dSdx(F2, hp, ssh, m, P, x, g2, bm2, bn2, Mg, work);
        for (int64_t i = 0; i < n; i++) { x[i] = x[i] - dx[i]; F1[i] = (F2[i] + F1[i]) / 2.0; }
        elpho_fourier_accelerate(Q, F1, fa_Q, 1.0, nf, L);
        elpho_fourier_accelerate(xi, eta, fa_Q, 0.5, nf, L);
        for (int64_t i = 0; i < n; i++) x[i] = x[i] + (sqrt(2.0 * dt) * xi[i] - dt * Q[i]);
    } else {
        elpho_fourier_accelerate(xi, eta, fa_Q, 0.5, nf, L);
        const int64_t it1 = langevin_dSdx(F1, hp, ssh, m, P, x, g1, bm1, bn1, Mg, work);
        elpho_fourier_accelerate(F1, F1, fa_Q, 1.0, nf, L);          /* dΓdx aliases dSdx */
        for (int64_t i = 0; i < n; i++) { dx[i] = sqrt(2 * dt) * xi[i] - dt * F1[i]; x[i] = x[i] + dx[i]; }
        langevin_update_model(hp, ssh, m, x);
        const int64_t it2 = langevin_dSdx(F2, hp, ssh, m, P, x, g2, bm2, bn2, Mg, work);
        elpho_fourier_accelerate(F2, F2, fa_Q, 1.0, nf, L);
        for (int64_t i = 0; i < n; i++) {
            x[i] = x[i] - dx[i];
            x[i] = x[i] + sqrt(2 * dt) * xi[i] - dt * (F1[i] + F2[i]) / 2;
        }
        iters = (it1 + it2) / 2;
    }
    langevin_update_model(hp, ssh, m, x);
    free(buf);
    return iters;
}

int64_t elpho_langevin_evolve(int scheme, const elpho_hmc_params *hp, elpho_model *m, elpho_kpm *P, double *x, const double *fa_Q,
                              double dt, const double *eta, const double *g1, const double *g2, const double *kpm_randn) {
    return langevin_evolve(scheme, hp, NULL, m, P, x, fa_Q, dt, eta, g1, g2, kpm_randn);
}

int64_t elpho_langevin_evolve_ssh(int scheme, const elpho_hmc_params *hp, const elpho_hmc_ssh *ssh, elpho_model *m, elpho_kpm *P,
                                  double *x, const double *fa_Q, double dt, const double *eta, const double *g1, const double *g2,
                                  const double *kpm_randn) {
    return langevin_evolve(scheme, hp, ssh, m, P, x, fa_Q, dt, eta, g1, g2, kpm_randn);
}

/* SSHModels.jl:707-829 without the equivalent-field bookkeeping (primary_field == identity):
 * dMdx[field(phonon,tau)] = sg(tau) * <c_n(tau)| dtau dK_n/dx |b_n(tau)>, accumulated bond by bond in checkerboard order.
 * bond_to_phonon_cb[n]: 1-based phonon living on checkerboard bond n (0 = none); x, alpha, alpha2 per phonon.
 * Uses m->vp (b) and m->vppp (c) as scratch, like the reference uses v' and v''. */
void elpho_muldMdx_ssh(double *dMdx, const double *u, const elpho_model *m, const double *v, double dtau,
                       const int64_t *bond_to_phonon_cb, const double *alpha, const double *alpha2, const double *x,
                       int64_t Nph) {
    const int64_t N = m->N, L = m->L;
    double *b = m->vp, *c = m->vppp;
    for (int64_t i = 0; i < N; i++)
        for (int64_t tau = 0; tau < L; tau++) b[i * L + tau] = m->E[i] * v[i * L + (tau == 0 ? L - 1 : tau - 1)];
    memcpy(c, u, sizeof(double) * (size_t)(N * L));
    elpho_checkerboard_transpose_mul_mat(c, m->table, m->c, m->s, m->nb, L);
    for (int64_t k = 0; k < Nph * L; k++) dMdx[k] = 0.0;
    for (int64_t n = 0; n < m->nb; n++) {
        const int64_t ph = bond_to_phonon_cb[n];
        const int64_t i = m->table[2 * n] - 1, j = m->table[2 * n + 1] - 1;
        for (int64_t tau = 0; tau < L; tau++) {
            const double ct = m->c[tau + L * n], st = m->s[tau + L * n];
            const int64_t it = i * L + tau, jt = j * L + tau;
            const double bi = b[it], bj = b[jt];
            b[it] = ct * bi + st * bj;
            b[jt] = ct * bj + st * bi;
            const double ci = c[it], cj = c[jt];
            c[it] = ct * ci - st * cj;
            c[jt] = ct * cj - st * ci;
            if (ph != 0) {
                const int64_t field = (ph - 1) * L + tau;
                const double dKdx = alpha[ph - 1] + 2 * alpha2[ph - 1] * x[field];
                double dmdx = c[jt] * dtau * dKdx * b[it] + c[it] * dtau * dKdx * b[jt];
                if (tau == 0) dmdx = -dmdx;
                dMdx[field] += dmdx;
            }
        }
    }
}

/* ====================================================================== */
/* OpenMP variant of the un-preconditioned CG iteration (CPU baseline only) */
/* ====================================================================== */
/* NOT the reference's configuration: the reference pins BLAS and FFTW to one thread and has no threading
 * (ElPhDynamics.jl:74-75).  This is the "what if the CPU path used all host cores" number BASELINE.md asks for:
 * same passes as above, with  (i) the site loops of mulM!/mulMᵀ! split over threads, (ii) the bonds of one
 * checkerboard colour (site-disjoint by construction) split over threads, (iii) BLAS-1 as parallel reductions.
 * Built only into libelph_oracle_omp.so (-fopenmp). */
#ifdef _OPENMP
#include <omp.h>

static int64_t omp_colours(const int64_t *table, int64_t nb, int64_t N, int64_t *off /* nb+1 */) {
    char *used = (char *)calloc((size_t)N, 1);
    int64_t nc = 0;
    off[0] = 0;
    for (int64_t n = 0; n < nb; n++) {
        int64_t i = table[2 * n] - 1, j = table[2 * n + 1] - 1;
        if (used[i] || used[j]) {
            off[++nc] = n;
            memset(used, 0, (size_t)N);
        }
        used[i] = used[j] = 1;
    }
    if (nb > 0) off[++nc] = nb;
    free(used);
    return nc;
}

static void omp_cb(double *y, const elpho_model *m, const int64_t *off, int64_t nc, int reverse) {
    const int64_t L = m->L;
    for (int64_t cc = 0; cc < nc; cc++) {
        const int64_t col = reverse ? nc - 1 - cc : cc;
#pragma omp for schedule(static)
        for (int64_t n = off[col]; n < off[col + 1]; n++) {
            const double cn = m->c[n], sn = m->s[n];
            double *yi = y + (m->table[2 * n] - 1) * L, *yj = y + (m->table[2 * n + 1] - 1) * L;
            for (int64_t tau = 0; tau < L; tau++) {
                const double t1 = yi[tau], t2 = yj[tau];
                yi[tau] = cn * t1 + sn * t2;
                yj[tau] = cn * t2 + sn * t1;
            }
        }
    }
}

/* Runs `niter` CG iterations (no stop test) on a Holstein model with `nthreads` threads; returns seconds. */
double elpho_cg_iterations_omp(const elpho_model *m, double *x, const double *b, int64_t niter, int nthreads) {
    const int64_t N = m->N, L = m->L, n = N * L;
    if (m->kind != 0) return -1.0;
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m->nb + 2));
    const int64_t nc = omp_colours(m->table, m->nb, N, off);
    double *r = (double *)malloc(sizeof(double) * (size_t)n), *p = (double *)malloc(sizeof(double) * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n), *w = (double *)malloc(sizeof(double) * (size_t)n);
    memset(x, 0, sizeof(double) * (size_t)n);
    memcpy(r, b, sizeof(double) * (size_t)n);
    memcpy(p, b, sizeof(double) * (size_t)n);
    double rho = 0.0, pap = 0.0, rr = 0.0;
    for (int64_t i = 0; i < n; i++) rho += r[i] * r[i];
    omp_set_num_threads(nthreads);
    const double t0 = omp_get_wtime();
#pragma omp parallel
    {
        for (int64_t it = 0; it < niter; it++) {
            /* w = M p */
#pragma omp for schedule(static)
            for (int64_t i = 0; i < N; i++)
                for (int64_t tau = 0; tau < L; tau++) w[i * L + tau] = m->E[i * L + tau] * p[i * L + (tau == 0 ? L - 1 : tau - 1)];
            omp_cb(w, m, off, nc, 0);
#pragma omp for schedule(static)
            for (int64_t i = 0; i < N; i++) {
                w[i * L] = p[i * L] + w[i * L];
                for (int64_t tau = 1; tau < L; tau++) w[i * L + tau] = p[i * L + tau] - w[i * L + tau];
            }
            /* z = M^T w */
#pragma omp for schedule(static)
            for (int64_t i = 0; i < n; i++) z[i] = w[i];
            omp_cb(z, m, off, nc, 1);
#pragma omp single
            pap = 0.0;
#pragma omp for schedule(static) reduction(+ : pap)
            for (int64_t i = 0; i < N; i++) {
                double *zi = z + i * L;
                const double *wi = w + i * L, *Ei = m->E + i * L;
                const double zL = wi[L - 1] + Ei[0] * zi[0];
                for (int64_t tau = 0; tau < L - 1; tau++) zi[tau] = wi[tau] - Ei[tau + 1] * zi[tau + 1];
                zi[L - 1] = zL;
                for (int64_t tau = 0; tau < L; tau++) pap += p[i * L + tau] * zi[tau];
            }
#pragma omp single
            rr = 0.0;
            const double alpha = rho / pap;
#pragma omp for schedule(static) reduction(+ : rr)
            for (int64_t i = 0; i < n; i++) {
                x[i] += alpha * p[i];
                r[i] -= alpha * z[i];
                rr += r[i] * r[i];
            }
            const double beta = rr / rho;
#pragma omp for schedule(static)
            for (int64_t i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
#pragma omp single
            rho = rr;
        }
    }
    const double t1 = omp_get_wtime();
    free(off); free(r); free(p); free(z); free(w);
    return t1 - t0;
}
#endif
