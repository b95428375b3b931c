// pk_grid.h -- a barrier across ALL workgroups of one launch, for persistent kernels that run several dependent phases in
// one launch (wf_layer.hip: the eight residual layers of a WaveFlow row; a phase reads what other workgroups wrote in the
// previous one).  The workgroups must be co-resident: such kernels are launched with hipLaunchCooperativeKernel (PK_LAUNCH_COOP),
// which fails instead of deadlocking when the grid does not fit the device.
//
// Protocol: `count` is zeroed by the host before the launch (stream-ordered memset); barrier k (k = 1, 2, ...) is passed when
// count >= k * gridDim.x.  Memory: the workgroups of a launch sit on 8 XCDs with one L2 each, and hipMalloc memory is not
// kept coherent between them inside a kernel -- one thread per workgroup therefore performs an agent-scope RELEASE fence before
// arriving (the XCD's L2 is written back; the other waves' stores have reached it: __syncthreads) and an agent-scope ACQUIRE
// fence after leaving (the CU's L1 and the L2's non-coherent lines are invalidated), the instructions the LLVM AMDGPU memory
// model prescribes for gfx942 / gfx950 agent-scope synchronisation.  A workgroup that waits longer than PK_GRID_TIMEOUT cycles of s_memtime (a lost workgroup: never seen, but a
// hang would cost the GPU) sets *err and goes on: the launch then ends with wrong data and a flag instead of never ending.
#pragma once
#include <hip/hip_runtime.h>

#include "pk_common.h"

// The host emulation of tools/hipemu runs the workgroups of a launch one after the other: a kernel that waits for another
// workgroup cannot run there.  Launchers ask pk_grid_available() and fall back to one launch per phase (same kernel, one phase).
#ifdef PK_HIPEMU
constexpr bool PK_GRID_AVAILABLE = false;
#else
constexpr bool PK_GRID_AVAILABLE = true;
#endif
static inline bool pk_grid_available() { return PK_GRID_AVAILABLE; }

constexpr unsigned long long PK_GRID_TIMEOUT = 1ull << 28;   // s_memtime ticks: 0.1 - 3 s depending on the counter's clock (a barrier wait is < 100 us)

__device__ __forceinline__ void pk_grid_barrier(unsigned* count, unsigned target, int* err) {
#ifndef PK_HIPEMU
    // __syncthreads(): every wave has waited for its own stores (they are performed at this XCD's L2) before ONE thread does the
    // agent-scope part -- write back the L2, arrive, wait, invalidate this CU's L1 and the L2's non-coherent lines.  (First
    // version: every thread fenced at agent scope -- 12 waves x 236 workgroups each writing back and invalidating a whole L2:
    // 90 us per barrier in the WaveFlow row kernel, tools/micro/grid_barrier.hip mode 1.)
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memtime() - t0 > PK_GRID_TIMEOUT) {
                if (err) *err = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
#else
    (void)count; (void)target;
    if (err) *err = 2;   // never reached: launchers split the phases over launches under the emulation
#endif
}

// Cooperative launch with the engine's profiler bracket (PK_LAUNCH of pk_common.h).  `args_struct` is the kernel's single
// by-value argument.
#define PK_LAUNCH_COOP(ctx, name, kernel, grid, block, args_struct)                                               \
    do {                                                                                                          \
        int _rec = (ctx)->prof_on ? (ctx)->prof_begin(name) : -1;                                                 \
        void* _kargs[1] = {const_cast<void*>(static_cast<const void*>(&(args_struct)))};                          \
        hipError_t _ce = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kernel), grid, block, _kargs, 0, \
                                                    (ctx)->stream);                                               \
        if (_rec >= 0) (ctx)->prof_end(_rec);                                                                     \
        if (_ce != hipSuccess) {                                                                                  \
            pk_set_error("cooperative launch of %s failed: %s", name, hipGetErrorString(_ce));                    \
            return PK_EHIP;                                                                                       \
        }                                                                                                         \
    } while (0)
