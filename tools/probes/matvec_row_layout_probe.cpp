// Probe (round 5): the fused M^T M apply of config C (16 x 16 sites, uniform hopping) in a layout where a 16-lane DPP ROW holds one time
// slice — lane (X, Y) = (l & 3, (l >> 2) & 3) a 4 x 4 patch of sites (16 registers per vector), the four rows of a wave four consecutive
// slices — against the instruction budget of the product's 2 x 2 patch layout (k_cg_wg, 4 slices per wave: 9 slab sweeps per mat-vec pair,
// 12 DPP moves of an f64 + 4 ds_bpermute per slab sweep).  Here: 2 sweeps of 64 fma + 16 DPP moves of an f64, every crossing inside the
// row; the tau shift goes through LDS (16 ds_write_b64 + 16 ds_read_b64 per shift and lane, one barrier each).
// What it measures: time per mat-vec pair of a workgroup of 8 waves (32 slices of one right-hand side) with every CU busy.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/matvec_row_layout_probe.cpp -o /tmp/matvec_row_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// one colour on the 4 x 4 patch v[cx + 4 cy]: v <- (I + th P_colour) v.  COL 0: x-even, 1: x-odd, 2: y-even, 3: y-odd
template <int COL>
__device__ __forceinline__ void colour(double (&v)[16], double th) {
    if (COL == 0) {
#pragma unroll
        for (int cy = 0; cy < 4; ++cy)
#pragma unroll
            for (int cx = 0; cx < 4; cx += 2) { const int i = cx + 4 * cy, j = i + 1; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
    } else if (COL == 2) {
#pragma unroll
        for (int cx = 0; cx < 4; ++cx)
#pragma unroll
            for (int cy = 0; cy < 4; cy += 2) { const int i = cx + 4 * cy, j = i + 4; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
    } else if (COL == 1) {
        double fu[4], fd[4];
#pragma unroll
        for (int cy = 0; cy < 4; ++cy) { fu[cy] = dpp<0x39>(v[0 + 4 * cy]); fd[cy] = dpp<0x93>(v[3 + 4 * cy]); }      // quad_perm [1,2,3,0]: from lane X + 1; [3,0,1,2]: from X - 1
#pragma unroll
        for (int cy = 0; cy < 4; ++cy) { const int i = 1 + 4 * cy, j = i + 1; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
#pragma unroll
        for (int cy = 0; cy < 4; ++cy) { v[3 + 4 * cy] += th * fu[cy]; v[0 + 4 * cy] += th * fd[cy]; }
    } else {
        double fu[4], fd[4];
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) { fu[cx] = dpp<0x12C>(v[cx + 0]); fd[cx] = dpp<0x124>(v[cx + 12]); }            // row_ror:12: from lane + 4 (Y + 1); row_ror:4: from lane - 4
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) { const int i = cx + 4, j = i + 4; const double a = v[i] + th * v[j], b = v[j] + th * v[i]; v[i] = a; v[j] = b; }
#pragma unroll
        for (int cx = 0; cx < 4; ++cx) { v[cx + 12] += th * fu[cx]; v[cx + 0] += th * fd[cx]; }
    }
}

__global__ void __launch_bounds__(512) k(double *out, const double *in, int iters, double th) {
    extern __shared__ double lds[];      // [34 slices][256 + 16 pad]
    constexpr int SL = 256 + 16;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, row = lane >> 4, l16 = lane & 15;
    const int slice = 1 + 4 * wv + row;                      // 1 .. 32; 0 and 33: the neighbouring workgroups' boundary slices
    double p[16], e[16], z[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { p[q] = in[(blockIdx.x * 512 + threadIdx.x) * 16 + q]; e[q] = 0.9 + 0.001 * q; }
    if (threadIdx.x < 256) { lds[0 * SL + threadIdx.x] = 0.5; lds[33 * SL + threadIdx.x] = 0.25; }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        // tau shift of p through LDS
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[slice * SL + q * 16 + l16] = p[q];
        __syncthreads();
        double u[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) u[q] = e[q] * lds[(slice - 1) * SL + q * 16 + l16];
        colour<0>(u, th); colour<1>(u, th); colour<2>(u, th); colour<3>(u, th);
        double m[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { m[q] = p[q] - u[q]; u[q] = m[q]; }
        colour<3>(u, th); colour<2>(u, th); colour<1>(u, th); colour<0>(u, th);
        __syncthreads();                                     // (everybody has read p of the slice below)
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[slice * SL + q * 16 + l16] = e[q] * u[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) z[q] = m[q] - lds[(slice + 1) * SL + q * 16 + l16];
        // four inner products' worth of fmas and the vector updates (x, r, p), as the iteration has them
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { s0 += p[q] * z[q]; s1 += m[q] * z[q]; s2 += z[q] * z[q]; s3 += m[q] * m[q]; }
        const double al = 1e-3 / (1.0 + s0 * s0 + s1 + s2 + s3);
#pragma unroll
        for (int q = 0; q < 16; ++q) p[q] = p[q] * 0.5 + al * z[q];
        __syncthreads();
    }
    double sum = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) sum += p[q];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

int main() {
    const int nblk = 256 * 6, iters = 2000;      // six rounds of the chip, as 288 right-hand sides x 5 workgroups make
    double *out, *in;
    hipMalloc(&out, (size_t)nblk * 512 * 8);
    hipMalloc(&in, (size_t)nblk * 512 * 16 * 8);
    hipMemset(in, 0, (size_t)nblk * 512 * 16 * 8);
    const size_t shm = 34 * (256 + 16) * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, nblk}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), shm, 0, out, in, iters, 0.1);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%4d workgroups of 8 waves (32 slices of a right-hand side each): %.3f us per mat-vec pair + sums + updates of one round (%d iterations, %.2f ms)\n", blocks,
               1e3 * ms / iters / (blocks / 256), iters, ms);
    }
    printf("(the product, 4 slices per wave in 2 x 2 patches: 5.0-5.2 us per iteration of a round of 48 right-hand sides, of which ~1.2 us are the meeting)\n");
    return 0;
}
