// Probe (round 6): does the tau-transform's access pattern — a wave touches FOUR 128-byte pieces of four different time slices (2 KB apart) per
// load instruction, the 16 column tiles of a slice being read by 16 different waves at different times — cost HBM efficiency against a wave
// that reads whole 2 KB slices?  Same bytes either way: read r and z ([nrhs][L][N] doubles), write r' = r - a z.
//   pattern 0 ("tile"):  the forward transform's: block (ct4, rhs), wave w -> column tile 4 ct4 + w, lane (jj, col): rows 2 (4 tt + jj), +1
//   pattern 1 ("slice"): wave -> (rhs, chunk of T slices), lane -> 4 consecutive sites (two 16-byte accesses per slice), cg_sq16.hip's
//   pattern 2 ("tile64"): as 0 but a wave covers 64 consecutive sites of ONE row per instruction (512 B contiguous), 4 rows in 4 instructions
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/access_pattern_probe.cpp -o /tmp/access_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int N = 256, L = 160;

__global__ void __launch_bounds__(256) k_tile(double *__restrict__ r, const double *__restrict__ z, double a) {
    const int rhs = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 15, jj = lane >> 4;
    const int s = (blockIdx.x * 4 + wv) * 16 + col;
    double *rw = r + (size_t)rhs * N * L;
    const double *zz = z + (size_t)rhs * N * L;
    constexpr int PF = 8, NT = L / 8;
    double q[PF][4];
#pragma unroll
    for (int p = 0; p < PF; ++p) { const size_t i0 = (size_t)(2 * (4 * p + jj)) * N + s; q[p][0] = rw[i0]; q[p][1] = rw[i0 + N]; q[p][2] = zz[i0]; q[p][3] = zz[i0 + N]; }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const size_t i0 = (size_t)(2 * (4 * tt + jj)) * N + s;
        const double n0 = q[tt % PF][0] - a * q[tt % PF][2], n1 = q[tt % PF][1] - a * q[tt % PF][3];
        rw[i0] = n0; rw[i0 + N] = n1;
        if (tt + PF < NT) { const size_t i1 = (size_t)(2 * (4 * (tt + PF) + jj)) * N + s; q[tt % PF][0] = rw[i1]; q[tt % PF][1] = rw[i1 + N]; q[tt % PF][2] = zz[i1]; q[tt % PF][3] = zz[i1 + N]; }
    }
}

__global__ void __launch_bounds__(256) k_tile64(double *__restrict__ r, const double *__restrict__ z, double a) {
    // a wave covers 64 consecutive sites of a row per instruction; block (quarter of the sites, rhs): wave w takes rows w, w + 4, ...
    const int rhs = blockIdx.z, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    double *rw = r + (size_t)rhs * N * L;
    const double *zz = z + (size_t)rhs * N * L;
    constexpr int PF = 8, NT = L / 4;      // rows of this wave
    double q[PF][2];
#pragma unroll
    for (int p = 0; p < PF; ++p) { const size_t i0 = (size_t)(4 * p + wv) * N + s; q[p][0] = rw[i0]; q[p][1] = zz[i0]; }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const size_t i0 = (size_t)(4 * tt + wv) * N + s;
        rw[i0] = q[tt % PF][0] - a * q[tt % PF][1];
        if (tt + PF < NT) { const size_t i1 = (size_t)(4 * (tt + PF) + wv) * N + s; q[tt % PF][0] = rw[i1]; q[tt % PF][1] = zz[i1]; }
    }
}

template <int T>
__global__ void __launch_bounds__(64) k_slice(double *__restrict__ r, const double *__restrict__ z, double a) {
    const int nch = L / T, rhs = blockIdx.x / nch, ch = blockIdx.x - rhs * nch, lane = threadIdx.x;
    double2 *rw = reinterpret_cast<double2 *>(r + (size_t)rhs * N * L + (size_t)ch * T * N);
    const double2 *zz = reinterpret_cast<const double2 *>(z + (size_t)rhs * N * L + (size_t)ch * T * N);
    constexpr int PF = 4;
    double2 qr[PF][2], qz[PF][2];
#pragma unroll
    for (int p = 0; p < PF; ++p) { qr[p][0] = rw[p * 128 + lane]; qr[p][1] = rw[p * 128 + 64 + lane]; qz[p][0] = zz[p * 128 + lane]; qz[p][1] = zz[p * 128 + 64 + lane]; }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        double2 o0, o1;
        o0.x = qr[t % PF][0].x - a * qz[t % PF][0].x; o0.y = qr[t % PF][0].y - a * qz[t % PF][0].y;
        o1.x = qr[t % PF][1].x - a * qz[t % PF][1].x; o1.y = qr[t % PF][1].y - a * qz[t % PF][1].y;
        rw[t * 128 + lane] = o0; rw[t * 128 + 64 + lane] = o1;
        if (t + PF < T) { const int u = t + PF; qr[t % PF][0] = rw[u * 128 + lane]; qr[t % PF][1] = rw[u * 128 + 64 + lane]; qz[t % PF][0] = zz[u * 128 + lane]; qz[t % PF][1] = zz[u * 128 + 64 + lane]; }
    }
}

int main(int argc, char **argv) {
    const int nrhs = argc > 1 ? atoi(argv[1]) : 288, nbuf = 6;      // several buffer pairs in rotation: more than the 256 MB Infinity Cache holds
    const size_t n = (size_t)nrhs * N * L;
    double *r[nbuf], *z[nbuf];
    for (int b = 0; b < nbuf; ++b) { hipMalloc(&r[b], n * 8); hipMalloc(&z[b], n * 8); hipMemset(r[b], 0, n * 8); hipMemset(z[b], 0, n * 8); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = 3.0 * n * 8;
    for (int pat = 0; pat < 4; ++pat) {
        for (int rot = 0; rot < 2; ++rot) {      // rot 0: the same pair every time (Infinity-Cache warm), 1: rotating pairs
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                for (int it = 0; it < 24; ++it) {
                    const int b = rot ? it % nbuf : 0;
                    if (pat == 0) hipLaunchKernelGGL(k_tile, dim3(4, 1, nrhs), dim3(256), 0, 0, r[b], z[b], 1e-3);
                    else if (pat == 1) hipLaunchKernelGGL(k_slice<16>, dim3(nrhs * (L / 16)), dim3(64), 0, 0, r[b], z[b], 1e-3);
                    else if (pat == 2) hipLaunchKernelGGL(k_tile64, dim3(4, 1, nrhs), dim3(256), 0, 0, r[b], z[b], 1e-3);
                    else hipLaunchKernelGGL(k_slice<8>, dim3(nrhs * (L / 8)), dim3(64), 0, 0, r[b], z[b], 1e-3);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const char *names[] = {"tile (4 x 128 B of 4 slices per instruction)", "slice (2 KB contiguous per wave, T = 16)", "tile64 (512 B of one slice per instruction)", "slice (T = 8)"};
            printf("nrhs %d  %-48s %s: %7.2f us per launch, %6.2f TB/s\n", nrhs, names[pat], rot ? "rotating buffers" : "same buffers    ", 1e3 * best / 24, bytes / (best / 24 * 1e-3) / 1e12);
        }
    }
    return 0;
}
