// cg_fast_common.h — helpers shared by the two compilations of cg_fast_impl.inc (cg_fast.hip: 4-colour lane programs,
// cg_fast6.hip: 6-colour ones; separate translation units so that they build in parallel).
#pragma once
#include <cstdlib>

#include "elph_internal.h"

#define WAVE ELPH_WAVE

// Ordering of LDS traffic inside ONE wavefront.  Every slab in this file is private to a wave, and the LDS
// pipeline executes a wave's DS instructions in issue order, so a ds_read issued after a ds_write of the same
// wave observes it without any s_waitcnt/s_barrier in between.  All that is needed is that the COMPILER keeps
// the program order of possibly-aliasing LDS accesses: a pure compiler barrier, no instruction.
// (Using __syncthreads() here costs an s_waitcnt lgkmcnt(0) per colour: one extra LDS round trip per stage.)
#ifdef ELPH_LDS_SYNC
#define WAVE_LDS_ORDER() __syncthreads()
#else
#define WAVE_LDS_ORDER() asm volatile("" ::: "memory")
#endif

// device-coherent scalar traffic (experiment): agent-scope relaxed atomics => sc1 loads/stores that bypass L1/K$
__device__ __forceinline__ double ld_coh(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ CgState ld_state(const CgState *p) {
    CgState s;
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    unsigned long long w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_memcpy(&s, w, sizeof(CgState));
    return s;
}

__device__ __forceinline__ double wave_sum2(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__device__ __forceinline__ double reduce_partials2(const double *p, int n) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += WAVE) a += ld_coh(p + i);
    return wave_sum2(a);
}

// XCD-aware mapping of a 1-D grid of 8*C*nrhs workgroups onto (tau, rhs); C = ceil(L/8)
__device__ __forceinline__ bool xcd_map(int L, int &t, int &rhs) {
    const int b = blockIdx.x;
    const int C = (L + 7) >> 3;
    const int xcd = b & 7, k = b >> 3;
    rhs = k / C;
    t = xcd * C + (k - rhs * C);
    return t < L;
}
