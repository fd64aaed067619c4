// Probe: issue rate of v_fma_f64 / v_mul_f64 / v_mov_dpp / ds_bpermute on one wave and on two waves of a SIMD (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/f64_rate_probe.cpp -o /tmp/f64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(512) k(double *out, long long *cyc, long long *wall, int iters) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
    const double c = 1.0000001, s = 1e-9;
    const int lane = threadIdx.x & 63;
    __syncthreads();
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = a[i] * c + s;                                   // v_fma_f64, 8 independent chains
                else if (KIND == 1) a[i] = a[i] * c;                                  // v_mul_f64
                else if (KIND == 2) { int lo = __double2loint(a[i]); lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
                else if (KIND == 3) { int lo = __double2loint(a[i]); lo = __builtin_amdgcn_ds_bpermute(((lane ^ 1) << 2), lo); a[i] = __hiloint2double(__double2hiint(a[i]), lo); }
                else { float f = (float)a[i]; f = f * 1.0000001f + 1e-9f; a[i] = (double)f; }
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double sum = 0; for (int i = 0; i < 8; ++i) sum += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; wall[0] = w1 - w0; }
}

int main() {
    double *out; long long *cyc, *wall;
    hipMalloc(&out, 512 * 8 * 1024); hipMalloc(&cyc, 8); hipMalloc(&wall, 8);
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_mov_b32_dpp", "ds_bpermute_b32"};
    const int iters = 2000;
    for (int kind = 0; kind < 4; ++kind)
        for (int threads : {64, 256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, wall, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, wall, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, cyc, wall, iters);
                if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, out, cyc, wall, iters);
                hipDeviceSynchronize();
            }
            long long c, w; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
            const double n = (double)iters * 64;
            printf("%-16s %3d threads (%d wave(s)/SIMD): %.2f clock64 ticks / instr, %.2f ns / instr (wall 100 MHz) -> %.2f GHz if clock64 counts shader cycles\n",
                   names[kind], threads, threads <= 256 ? 1 : 2, c / n, w * 10.0 / n, (c / n) / (w * 10.0 / n));
        }
    return 0;
}
