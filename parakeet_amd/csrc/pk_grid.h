// pk_grid.h -- a barrier across ALL workgroups of one launch, for persistent kernels that run several dependent phases in
// one launch (wf_layer.hip: the eight residual layers of a WaveFlow row; a phase reads what other workgroups wrote in the
// previous one).  The workgroups must be co-resident: such kernels are launched with hipLaunchCooperativeKernel (PK_LAUNCH_COOP),
// which fails instead of deadlocking when the grid does not fit the device.
//
// Protocol: `count` is zeroed by the host before the launch (stream-ordered memset); barrier k (k = 1, 2, ...) is passed when
// count >= k * gridDim.x.  Memory: the workgroups of a launch sit on 8 XCDs with one L2 each, and hipMalloc memory is not
// kept coherent between them inside a kernel -- every thread therefore performs an agent-scope RELEASE fence before arriving
// (its stores are written back from this XCD's L2) and an agent-scope ACQUIRE fence after leaving (this XCD's non-coherent
// lines are invalidated), the construction the LLVM AMDGPU memory model prescribes for gfx942 / gfx950 agent-scope
// synchronisation.  A workgroup that waits longer than PK_GRID_TIMEOUT cycles of s_memtime (a lost workgroup: never seen, but a
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

constexpr unsigned long long PK_GRID_TIMEOUT = 1ull << 31;   // s_memtime ticks (100 MHz constant clock on gfx9: ~20 s; at core clock ~1 s)

__device__ __forceinline__ void pk_grid_barrier(unsigned* count, unsigned target, int* err) {
#ifndef PK_HIPEMU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (__builtin_amdgcn_s_memtime() - t0 > PK_GRID_TIMEOUT) {
                if (err) *err = 1;
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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
