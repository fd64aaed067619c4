// kpm_host.cpp — host-side set-up of the KPM preconditioner (runs once per setup!, not per CG
// iteration): Arnoldi eigenvalue bounds of A = CBbar diag(Ebar) on N-vectors, the small dense
// eigen-solve the reference delegates to LAPACK, and the Chebyshev coefficients.
//
// Reference: KPMPreconditioners.jl:259-321 (setup!), :387-420 (A, A^-1), :789-839 (coefficients),
// :845-942 (Arnoldi).  N <= 512 and n <= 20, so this is microseconds of scalar work; the per-iteration
// apply (ldiv!, :426-481) is on the GPU (kernels.hip).

#include <cmath>
#include <cstring>

#include "elph_internal.h"

namespace {

// bonds are stored in checkerboard order; the sequential product over n = 0..nb-1 is the
// reference's checkerboard_mul! (Checkerboard.jl:123-141), reversed with -s its inverse (:298-316).
void cb_mul(std::vector<double> &y, const elph_handle_s *h, const double *cbar, const double *sbar) {
    for (int64_t n = 0; n < h->nb; ++n) {
        const int i = h->h_bi[n], j = h->h_bj[n];
        const double c = cbar[n], s = sbar[n];
        const double t1 = y[i], t2 = y[j];
        y[i] = c * t1 + s * t2;
        y[j] = c * t2 + s * t1;
    }
}

void cb_inv_mul(std::vector<double> &y, const elph_handle_s *h, const double *cbar, const double *sbar) {
    for (int64_t n = h->nb - 1; n >= 0; --n) {
        const int i = h->h_bi[n], j = h->h_bj[n];
        const double c = cbar[n], s = sbar[n];
        const double t1 = y[i], t2 = y[j];
        y[i] = c * t1 - s * t2;
        y[j] = c * t2 - s * t1;
    }
}

void apply_A(std::vector<double> &out, const std::vector<double> &in, const elph_handle_s *h, const double *Ebar, bool inverse,
             const double *cbar, const double *sbar) {
    const int64_t N = h->N;
    if (!inverse) {  // A v = CBbar (Ebar .* v), :387-401
        for (int64_t i = 0; i < N; ++i) out[i] = Ebar[i] * in[i];
        cb_mul(out, h, cbar, sbar);
    } else {         // A^-1 v = (CBbar^-1 v) ./ Ebar, :406-420
        out = in;
        cb_inv_mul(out, h, cbar, sbar);
        for (int64_t i = 0; i < N; ++i) out[i] /= Ebar[i];
    }
}

double max_ritz(const elph_handle_s *h, const double *Ebar, int n, const double *b0, bool inverse, const double *cbar,
                const double *sbar) {
    const int64_t m = h->N;
    std::vector<double> Q((size_t)m * (n + 1), 0.0), H((size_t)(n + 1) * n, 0.0), b(m), v(m);
    double nrm = 0.0;
    for (int64_t i = 0; i < m; ++i) nrm += b0[i] * b0[i];
    nrm = std::sqrt(nrm);
    for (int64_t i = 0; i < m; ++i) { b[i] = b0[i] / nrm; Q[i] = b[i]; }
    int l = n;
    for (int k = 0; k < n; ++k) {
        apply_A(v, b, h, Ebar, inverse, cbar, sbar);
        for (int j = 0; j <= k; ++j) {
            const double *Qj = &Q[(size_t)j * m];
            double d = 0.0;
            for (int64_t i = 0; i < m; ++i) d += Qj[i] * v[i];
            H[j + (size_t)(n + 1) * k] = d;
            for (int64_t i = 0; i < m; ++i) v[i] -= d * Qj[i];
        }
        double nv = 0.0;
        for (int64_t i = 0; i < m; ++i) nv += v[i] * v[i];
        nv = std::sqrt(nv);
        H[(k + 1) + (size_t)(n + 1) * k] = nv;
        if (nv > 1e-12) {
            for (int64_t i = 0; i < m; ++i) { b[i] = v[i] / nv; Q[(size_t)(k + 1) * m + i] = b[i]; }
        } else {
            l = k + 1;
            break;
        }
    }
    std::vector<double> hp((size_t)l * l), wr(l), wi(l);
    for (int j = 0; j < l; ++j)
        for (int i = 0; i < l; ++i) {
            hp[i + (size_t)l * j] = H[i + (size_t)(n + 1) * j];
            if (!std::isfinite(hp[i + (size_t)l * j])) return INFINITY;
        }
    if (elph_hess_eigvals(hp, l, wr, wi) != 0) return INFINITY;
    double best = wr[0];
    for (int i = 1; i < l; ++i) best = std::max(best, wr[i]);
    return best;
}

}  // namespace

// Eigenvalues of a real upper-Hessenberg matrix (column-major n x n) by the implicit double-shift
// QR iteration (the algorithm behind LAPACK dhseqr / EISPACK hqr) — stands in for eigvals!
// at KPMPreconditioners.jl:891,935.  Only the eigenvalues are needed.
int elph_hess_eigvals(std::vector<double> &a, int n, std::vector<double> &wr, std::vector<double> &wi) {
    auto A = [&](int i, int j) -> double & { return a[(size_t)i + (size_t)j * n]; };
    auto sgn = [](double x, double y) { return (y >= 0.0) ? std::fabs(x) : -std::fabs(x); };
    if (n == 1) { wr[0] = a[0]; wi[0] = 0.0; return 0; }
    double anorm = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = std::max(i - 1, 0); j < n; ++j) anorm += std::fabs(A(i, j));
    int nn = n - 1;
    double t = 0.0, p = 0, q = 0, r = 0, s, x, y, z, w, u, v;
    while (nn >= 0) {
        int its = 0, l;
        do {
            for (l = nn; l >= 1; --l) {
                s = std::fabs(A(l - 1, l - 1)) + std::fabs(A(l, l));
                if (s == 0.0) s = anorm;
                if (std::fabs(A(l, l - 1)) + s == s) { A(l, l - 1) = 0.0; break; }
            }
            x = A(nn, nn);
            if (l == nn) {                       // one root found
                wr[nn] = x + t; wi[nn] = 0.0; --nn;
            } else {
                y = A(nn - 1, nn - 1);
                w = A(nn, nn - 1) * A(nn - 1, nn);
                if (l == nn - 1) {               // two roots found
                    p = 0.5 * (y - x);
                    q = p * p + w;
                    z = std::sqrt(std::fabs(q));
                    x += t;
                    if (q >= 0.0) {
                        z = p + sgn(z, p);
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != 0.0) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = 0.0;
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn] = z; wi[nn - 1] = -z;
                    }
                    nn -= 2;
                } else {                         // no roots yet: QR step
                    if (its == 60) return -1;
                    if (its == 10 || its == 20) {   // exceptional shift
                        t += x;
                        for (int i = 0; i <= nn; ++i) A(i, i) -= x;
                        s = std::fabs(A(nn, nn - 1)) + std::fabs(A(nn - 1, nn - 2));
                        y = x = 0.75 * s;
                        w = -0.4375 * s * s;
                    }
                    ++its;
                    int m;
                    for (m = nn - 2; m >= l; --m) {
                        z = A(m, m);
                        r = x - z; s = y - z;
                        p = (r * s - w) / A(m + 1, m) + A(m, m + 1);
                        q = A(m + 1, m + 1) - z - r - s;
                        r = A(m + 2, m + 1);
                        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        u = std::fabs(A(m, m - 1)) * (std::fabs(q) + std::fabs(r));
                        v = std::fabs(p) * (std::fabs(A(m - 1, m - 1)) + std::fabs(z) + std::fabs(A(m + 1, m + 1)));
                        if (u + v == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        A(i, i - 2) = 0.0;
                        if (i != m + 2) A(i, i - 3) = 0.0;
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = A(k, k - 1); q = A(k + 1, k - 1); r = 0.0;
                            if (k != nn - 1) r = A(k + 2, k - 1);
                            if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
                        }
                        if ((s = sgn(std::sqrt(p * p + q * q + r * r), p)) != 0.0) {
                            if (k == m) {
                                if (l != m) A(k, k - 1) = -A(k, k - 1);
                            } else {
                                A(k, k - 1) = -s * x;
                            }
                            p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
                            for (int j = k; j <= nn; ++j) {
                                p = A(k, j) + q * A(k + 1, j);
                                if (k != nn - 1) { p += r * A(k + 2, j); A(k + 2, j) -= p * z; }
                                A(k + 1, j) -= p * y;
                                A(k, j) -= p * x;
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * A(i, k) + y * A(i, k + 1);
                                if (k != nn - 1) { p += z * A(i, k + 2); A(i, k + 2) -= p * r; }
                                A(i, k + 1) -= p * q;
                                A(i, k) -= p;
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
    return 0;
}

// KPMPreconditioners.jl:845-942 with the random start vectors supplied by the caller.
// `chain` selects the configuration's Ē in h->h_Ebar; re-entrant (chains are set up on parallel host threads).
int elph_kpm_arnoldi(const elph_handle_s *h, int chain, const double *b_max, const double *b_min, double *e_min, double *e_max) {
    const double *Ebar = h->h_Ebar.data() + (size_t)chain * (size_t)h->N;
    int n = h->kpm_n;
    if (n > h->N) n = (int)h->N;   // :136
    if (n < 1) n = 1;
    // averaged hopping of this chain: one table per chain for SSH chains, one shared table otherwise
    const size_t hop = (h->h_cbar.size() >= (size_t)(chain + 1) * (size_t)h->nb && h->kpm_hop_per_chain) ? (size_t)chain * (size_t)h->nb : 0;
    const double *cbar = h->h_cbar.data() + hop, *sbar = h->h_sbar.data() + hop;
    const double emax = max_ritz(h, Ebar, n, b_max, false, cbar, sbar);
    const double r = max_ritz(h, Ebar, n, b_min, true, cbar, sbar);
    *e_max = emax;
    *e_min = std::isfinite(r) ? 1.0 / r : -INFINITY;
    return 0;
}

// KPMPreconditioners.jl:789-839 (+ scalar_invM :948-951).  The reference takes a unitary DCT-II of
// f(x_n) and rescales; that equals  c_m = (2 - [m==0])/N_M * sum_n f(x_n) cos(pi m (n+1/2)/N_M),
// N_M = 2*order, x_n = lam_mag cos(pi (n+1/2)/N_M) + lam_avg, f(x) = 1/(1 - exp(-i phi) x).
void elph_kpm_coefficients(double *c_z, int order, double lam_lo, double lam_hi, double phi) {
    const int M = order, NM = 2 * order;
    const double avg = 0.5 * (lam_hi + lam_lo), mag = 0.5 * (lam_hi - lam_lo);
    std::vector<double> fr(NM), fi(NM);
    const double er = std::cos(phi), ei = -std::sin(phi);
    for (int n = 0; n < NM; ++n) {
        const double x = mag * std::cos(M_PI * (n + 0.5) / NM) + avg;
        const double dr = 1.0 - er * x, di = -ei * x;
        const double den = dr * dr + di * di;
        fr[n] = dr / den;
        fi[n] = -di / den;
    }
    for (int m = 0; m < M; ++m) {
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < NM; ++n) {
            const double cs = std::cos(M_PI * m * (n + 0.5) / NM);
            sr += fr[n] * cs;
            si += fi[n] * cs;
        }
        const double f = ((m == 0) ? 1.0 : 2.0) / NM;
        c_z[2 * m] = f * sr;
        c_z[2 * m + 1] = f * si;
    }
}
