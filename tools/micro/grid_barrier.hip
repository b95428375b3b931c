// What does a barrier across the workgroups of ONE launch cost on an MI355X (8 XCDs, one L2 each)?  Decides whether a
// persistent multi-phase kernel (csrc/pk_grid.h) can beat one launch per phase (back-to-back dependent launches: 5 - 6 us).
// G workgroups x 512 threads, cooperative launch, ITER barriers; every workgroup writes 64 KB before each barrier and reads
// its neighbour's 64 KB after it (so the fences have real work).  Modes:
//   0  arrive + spin only (relaxed agent-scope atomics, no fences): the floor -- NOT a correct barrier for cross-XCD data
//   1  every thread: agent-scope release fence before, acquire fence after (pk_grid.h as first written)
//   2  __syncthreads, then ONE thread: release fence, arrive, spin, acquire fence; __syncthreads (pk_grid.h now)
//   3  as 2 with the data written by sc1 stores... (not built: needs inline asm)
// Prints us per barrier and whether the neighbour's data arrived intact.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 200, WORDS = 16384;   // 64 KB per workgroup

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* count, unsigned* data, unsigned* bad) {
    const unsigned g = gridDim.x, me = blockIdx.x, nb = (me + g / 2 + 1) % g;   // a neighbour on another XCD (b % 8)
    unsigned errs = 0;
    for (int it = 0; it < ITER; ++it) {
        for (int i = threadIdx.x; i < WORDS; i += 512) data[(size_t)me * WORDS + i] = (unsigned)(it * 7919 + me * 31 + i);
        const unsigned target = (unsigned)(it + 1) * g;
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int i = threadIdx.x; i < WORDS; i += 512)
            errs += data[(size_t)nb * WORDS + i] != (unsigned)(it * 7919 + nb * 31 + i);
        // (a second barrier so that nobody overwrites data a slower neighbour still reads; same kind, counted below)
        const unsigned target2 = target + 0x40000000u;
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(count + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(it + 1) * g) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        (void)target2;
    }
    if (errs) atomicAdd(bad, errs);
}

template <int MODE>
int run(int G) {
    unsigned *count, *data, *bad;
    CK(hipMalloc(&count, 8)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&data, (size_t)G * WORDS * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; unsigned hbad = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(count, 0, 8)); CK(hipMemset(bad, 0, 4));
        void* args[3] = {&count, &data, &bad};
        CK(hipEventRecord(e0));
        CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k<MODE>), dim3(G), dim3(512), args, 0, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    }
    printf("mode %d  G %3d: %7.2f us per iteration (write 64 KB + barrier + read 64 KB + plain barrier), stale words seen: %u\n", MODE, G, best * 1e3f / ITER, hbad);
    CK(hipFree(count)); CK(hipFree(bad)); CK(hipFree(data));
    return 0;
}
int main() {
    for (int G : {32, 128, 236, 256}) { if (run<0>(G)) return 1; if (run<1>(G)) return 1; if (run<2>(G)) return 1; }
    return 0;
}
