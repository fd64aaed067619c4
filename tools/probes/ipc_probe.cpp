// ipc_probe.cpp — can two PROCESSES on one GPU hand data to each other from inside running kernels through an IPC-mapped
// buffer?  (the transport of the sharded solve, csrc/shard.hip: device-initiated stores into the peer's mailbox)
//   ./ipc_probe <rank 0|1> <dir> [alloc: 0 hipMalloc, 1 fine-grained, 2 uncached]
// Both ranks run a ping-pong kernel: rank 0 stores tag k into the peer's word, rank 1 answers with k; bounded spins.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); exit(2); } } while (0)

__global__ void pingpong(unsigned long long *mine, unsigned long long *peer, int rank, int rounds, unsigned long long *out) {
    if (threadIdx.x != 0) return;
    long long t0 = wall_clock64();
    unsigned long long fails = 0;
    for (int k = 1; k <= rounds; ++k) {
        if (rank == 0) __hip_atomic_store(peer, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        long long spins = 0;
        while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)k) {
            if (++spins > (1LL << 24)) { fails++; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (fails) break;
        if (rank == 1) __hip_atomic_store(peer, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    out[0] = (unsigned long long)(wall_clock64() - t0);
    out[1] = fails;
}

static bool read_file(const std::string &p, void *buf, size_t n) {
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    size_t r = fread(buf, 1, n, f);
    fclose(f);
    return r == n;
}

int main(int argc, char **argv) {
    const int rank = atoi(argv[1]);
    const std::string dir = argv[2];
    const int alloc = argc > 3 ? atoi(argv[3]) : 1;
    CK(hipSetDevice(0));
    unsigned long long *mine = nullptr;
    const size_t bytes = 1 << 20;
    if (alloc == 0) CK(hipMalloc((void **)&mine, bytes));
    else CK(hipExtMallocWithFlags((void **)&mine, bytes, alloc == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    CK(hipMemset(mine, 0, bytes));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hm, hp;
    CK(hipIpcGetMemHandle(&hm, mine));
    {
        const std::string tmp = dir + "/h" + std::to_string(rank) + ".tmp", fin = dir + "/h" + std::to_string(rank) + ".bin";
        FILE *f = fopen(tmp.c_str(), "wb"); fwrite(&hm, 1, sizeof(hm), f); fclose(f);
        rename(tmp.c_str(), fin.c_str());
    }
    const std::string pf = dir + "/h" + std::to_string(1 - rank) + ".bin";
    for (int i = 0; i < 6000 && !read_file(pf, &hp, sizeof(hp)); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    unsigned long long *peer = nullptr;
    hipError_t e = hipIpcOpenMemHandle((void **)&peer, hp, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) { printf("rank %d alloc %d: hipIpcOpenMemHandle failed: %s\n", rank, alloc, hipGetErrorString(e)); return 3; }
    unsigned long long *out;
    CK(hipMalloc((void **)&out, 64));
    const int rounds = 2000;
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, mine, peer, rank, rounds, out);
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("rank %d alloc %d: %d round trips, %.3f us each (wall_clock64 @100MHz), fails=%llu\n", rank, alloc, rounds,
           (double)h[0] / 100.0 / rounds, h[1]);
    CK(hipIpcCloseMemHandle(peer));
    return h[1] ? 4 : 0;
}
