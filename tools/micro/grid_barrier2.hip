// Second look at barriers across the workgroups of one launch on an MI355X (8 XCDs, one L2 each); follows grid_barrier.hip,
// whose single-counter barrier with whole-L2 fences cost 27 us at 256 workgroups.  Questions here:
//   * what does a barrier cost without read-modify-write atomics on ONE address (a flag per workgroup, everybody polls all flags)?
//   * can the data that crosses the barrier travel with agent-scope (sc1) stores and loads instead of whole-L2 write-back /
//     invalidate, and what does that cost per byte?
//   * what does the command processor charge for the same thing (one launch per phase)?
// Per iteration: every workgroup writes W words, barrier, reads the W words of a workgroup on another XCD, second barrier
// (so that nobody overwrites what a slower neighbour still reads).  Both barriers are of the mode's kind.
//   mode 0  one counter, relaxed agent-scope atomics, no fences               (floor of grid_barrier.hip; stale data expected)
//   mode 2  one counter, one thread per workgroup fences (release before, acquire after)      (pk_grid.h as shipped)
//   mode 3  flag array, plain stores / loads for the data                       (floor of the flag barrier; stale data expected)
//   mode 4  flag array, data written with sc1 stores and read with sc1 loads, no cache maintenance
//   mode 5  flag array, one thread per workgroup: buffer_wbl2 sc1 before its flag, buffer_inv sc1 after the wait
//   mode 6  per-XCD counter + flag array over the 8 XCD leaders: the last arriver of an XCD writes back the L2 and raises the
//           XCD's flag; after the wait it invalidates the L2 and releases the XCD's workgroups, which invalidate their L1 only
//   mode 7  as 6, the released workgroups invalidate at agent scope (buffer_inv sc1) like the leader
//   mode 9  no barrier: ITER dependent launches of the same write / read phase (what the command processor charges)
// Every spin is bounded (SPIN_MAX polls): a lost workgroup ends the launch with the `lost` flag, never a hang.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 200;
constexpr unsigned SPIN_MAX = 1u << 18;   // polls; after the first loss every spin gives up within 256 polls

struct Bar {
    unsigned* count;      // [2] single counters (modes 0, 2)
    unsigned* flags;      // [2][256] one flag per workgroup and barrier slot (modes 3 - 5); [2][8] XCD flags at +512 (mode 6)
    unsigned* xcount;     // [2][8 * 32] per-XCD arrival counters, one 128-byte line each (mode 6)
    unsigned* xrelease;   // [2][8 * 32] per-XCD release words (mode 6)
    unsigned* xmembers;   // [8] workgroups per XCD (mode 6; filled by phase 0 of the launch)
    unsigned* lost;
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {   // past the L1, from this XCD's L2
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// one counter (grid_barrier.hip)
template <bool FENCE>
__device__ __forceinline__ void bar_counter(unsigned* count, unsigned target, unsigned* lost) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n = 0;
        while (ld_sc1(count) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++n > SPIN_MAX || ((n & 255u) == 0 && ld_sc1(lost))) { *lost = 1; break; }
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// flag array: workgroup `me` raises flags[me] to `epoch`; wave 0 polls all G flags (lane l holds flags 4 l .. 4 l + 3)
template <int MAINT>   // 0: nothing, 1: wbl2 before / inv after by one thread
__device__ __forceinline__ void bar_flags(unsigned* flags, unsigned G, unsigned epoch, unsigned* lost) {
    __syncthreads();   // (every wave has waited for its own stores: see the callers)
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) {
            if (MAINT == 1) asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(flags + blockIdx.x, epoch);
        }
        unsigned n = 0;
        for (;;) {
            bool ok = true;
            for (int j = 0; j < 4; ++j) {
                const unsigned i = threadIdx.x * 4 + j;
                if (i < G) ok = ok && (int)(ld_sc1(flags + i) - epoch) >= 0;
            }
            if (__all(ok)) break;
            if (++n > SPIN_MAX || ((n & 255u) == 0 && ld_sc1(lost))) { if (threadIdx.x == 0) *lost = 1; break; }
        }
        if (MAINT == 1 && threadIdx.x == 0) asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// two levels: per-XCD counter (the last arriver is the XCD's leader for this barrier), flags over the XCDs
template <bool L1_ONLY>
__device__ __forceinline__ void bar_xcd(const Bar& b, int slot, unsigned xcc, unsigned members, unsigned nx_mask, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* xc = b.xcount + slot * 256 + xcc * 32;
        unsigned* xr = b.xrelease + slot * 256 + xcc * 32;
        unsigned* xf = b.flags + 512 + slot * 8;
        const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned n = 0;
        if (old + 1 == epoch * members) {   // the XCD's last arriver: every store of this XCD's workgroups is in its L2
            asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(xf + xcc, epoch);
            for (;;) {
                bool ok = true;
                for (unsigned x = 0; x < 8; ++x)
                    if (nx_mask >> x & 1) ok = ok && (int)(ld_sc1(xf + x) - epoch) >= 0;
                if (ok) break;
                if (++n > SPIN_MAX || ((n & 255u) == 0 && ld_sc1(b.lost))) { *b.lost = 1; break; }
            }
            asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(xr, epoch);
        } else {
            while ((int)(ld_sc1(xr) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++n > SPIN_MAX || ((n & 255u) == 0 && ld_sc1(b.lost))) { *b.lost = 1; break; }
            }
            if (L1_ONLY) asm volatile("buffer_inv sc0\n s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
}

template <int MODE, int WORDS>
__global__ __launch_bounds__(512) void k(Bar b, unsigned* data, unsigned* bad) {
    const unsigned g = gridDim.x, me = blockIdx.x, nb = (me + g / 2 + 1) % g;   // a neighbour on another XCD (b % 8)
    unsigned errs = 0;
    unsigned xcc = 0, members = 0, nx_mask = 0;
    if (MODE == 6 || MODE == 7) {   // who shares my L2?  (once per launch, behind a plain single-counter barrier)
        xcc = xcc_id() & 7u;
        if (threadIdx.x == 0) atomicAdd(b.xmembers + xcc, 1u);
        bar_counter<true>(b.count + 2, g, b.lost);
        members = ld_sc1(b.xmembers + xcc);
        for (unsigned x = 0; x < 8; ++x) nx_mask |= (ld_sc1(b.xmembers + x) != 0) << x;
    }
    for (int it = 0; it < ITER; ++it) {
        unsigned* mine = data + (size_t)me * WORDS;
        const unsigned* theirs = data + (size_t)nb * WORDS;
        for (int i = threadIdx.x; i < WORDS; i += 512) {
            const unsigned v = (unsigned)(it * 7919 + me * 31 + i);
            if (MODE == 4) st_sc1(mine + i, v);
            else mine[i] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned e1 = 2 * it + 1, e2 = 2 * it + 2;
        if (MODE == 0) bar_counter<false>(b.count, e1 * g, b.lost);
        if (MODE == 2) bar_counter<true>(b.count, e1 * g, b.lost);
        if (MODE == 3 || MODE == 4) bar_flags<0>(b.flags, g, e1, b.lost);
        if (MODE == 5) bar_flags<1>(b.flags, g, e1, b.lost);
        if (MODE == 6) bar_xcd<true>(b, 0, xcc, members, nx_mask, e1);
        if (MODE == 7) bar_xcd<false>(b, 0, xcc, members, nx_mask, e1);
        for (int i = threadIdx.x; i < WORDS; i += 512) {
            const unsigned v = MODE == 4 ? ld_sc1(theirs + i) : theirs[i];
            errs += v != (unsigned)(it * 7919 + nb * 31 + i);
        }
        if (MODE == 0) bar_counter<false>(b.count, e2 * g, b.lost);
        if (MODE == 2) bar_counter<false>(b.count, e2 * g, b.lost);
        if (MODE == 3 || MODE == 4 || MODE == 5) bar_flags<0>(b.flags, g, e2, b.lost);
        if (MODE == 6) bar_xcd<true>(b, 0, xcc, members, nx_mask, e2);
        if (MODE == 7) bar_xcd<false>(b, 0, xcc, members, nx_mask, e2);
    }
    if (errs) atomicAdd(bad, errs);
}

// one launch per phase
template <int WORDS>
__global__ __launch_bounds__(512) void k_phase(unsigned* data, unsigned* bad, int it) {
    const unsigned g = gridDim.x, me = blockIdx.x, nb = (me + g / 2 + 1) % g;
    // read what the neighbour wrote in the previous launch (buffer it & 1 ^ 1), write mine (buffer it & 1)
    const unsigned* theirs = data + (size_t)((it & 1) ^ 1) * g * WORDS + (size_t)nb * WORDS;
    unsigned* mine = data + (size_t)(it & 1) * g * WORDS + (size_t)me * WORDS;
    unsigned errs = 0;
    if (it > 0)
        for (int i = threadIdx.x; i < WORDS; i += 512) errs += theirs[i] != (unsigned)((it - 1) * 7919 + nb * 31 + i);
    for (int i = threadIdx.x; i < WORDS; i += 512) mine[i] = (unsigned)(it * 7919 + me * 31 + i);
    if (errs) atomicAdd(bad, errs);
}

template <int MODE, int WORDS>
int run(int G) {
    Bar b;
    unsigned *data, *bad;
    CK(hipMalloc(&b.count, 16)); CK(hipMalloc(&b.flags, 4 * 1024)); CK(hipMalloc(&b.xcount, 4 * 512)); CK(hipMalloc(&b.xrelease, 4 * 512));
    CK(hipMalloc(&b.xmembers, 32)); CK(hipMalloc(&b.lost, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&data, (size_t)2 * G * WORDS * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; unsigned hbad = 0, hlost = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(b.count, 0, 16)); CK(hipMemset(b.flags, 0, 4 * 1024)); CK(hipMemset(b.xcount, 0, 4 * 512)); CK(hipMemset(b.xrelease, 0, 4 * 512));
        CK(hipMemset(b.xmembers, 0, 32)); CK(hipMemset(b.lost, 0, 4)); CK(hipMemset(bad, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (MODE == 9) {
            for (int it = 0; it < ITER; ++it) hipLaunchKernelGGL(k_phase<WORDS>, dim3(G), dim3(512), 0, 0, data, bad, it);
        } else {
            void* args[3] = {&b, &data, &bad};
            CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k<MODE, WORDS>), dim3(G), dim3(512), args, 0, 0));
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        unsigned x; CK(hipMemcpy(&x, bad, 4, hipMemcpyDeviceToHost)); hbad += x;
        CK(hipMemcpy(&x, b.lost, 4, hipMemcpyDeviceToHost)); hlost += x;
    }
    printf("mode %d  G %3d  %5d B per workgroup: %7.2f us per iteration, stale words %u, lost %u\n", MODE, G, WORDS * 4, best * 1e3f / ITER, hbad, hlost);
    fflush(stdout);
    CK(hipFree(b.count)); CK(hipFree(b.flags)); CK(hipFree(b.xcount)); CK(hipFree(b.xrelease)); CK(hipFree(b.xmembers)); CK(hipFree(b.lost));
    CK(hipFree(bad)); CK(hipFree(data));
    return 0;
}
template <int WORDS>
int sweep() {
    for (int G : {32, 64, 128, 256}) {
        if (run<9, WORDS>(G)) return 1;
        if (run<0, WORDS>(G)) return 1;
        if (run<2, WORDS>(G)) return 1;
        if (run<3, WORDS>(G)) return 1;
        if (run<4, WORDS>(G)) return 1;
        if (run<5, WORDS>(G)) return 1;
        if (run<6, WORDS>(G)) return 1;
        if (run<7, WORDS>(G)) return 1;
    }
    return 0;
}
int main() {
    if (sweep<1024>()) return 1;
    if (sweep<16384>()) return 1;
    return 0;
}
