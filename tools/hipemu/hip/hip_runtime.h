// hipemu: a host stand-in for <hip/hip_runtime.h>, for DEVELOPMENT ONLY.
//
// tools/hipemu/build.py compiles the engine's .hip translation units as plain C++ against this header into
// tools/hipemu/_build/libpk_synth_emu.so.  Every kernel launch then runs on the host: one fiber per work-item,
// workgroups one after the other, __syncthreads / wave shuffles / DPP / MFMA as rendezvous between the fibers of a
// workgroup / wave.  It exists to debug kernel LOGIC (indexing, barriers, hand-offs) without a GPU box.  It is not
// a fallback: parakeet_amd never loads this library (only tools/hipemu/harness.py does, for tests/test_emu_*.py),
// the hardware's approximate instructions (v_exp_f32, v_rcp_f32 ...) become libm calls, and nothing measured or
// claimed anywhere in this repository comes from it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <type_traits>
#include <utility>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

// ---------------------------------------------------------------------------------------------- types
struct uint3 {
    unsigned x, y, z;
};
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 {
    float x, y;
};
struct alignas(16) float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorLaunchFailure = 719 };
typedef struct hipemu_stream* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

// ---------------------------------------------------------------------------------------------- runtime API
extern "C" {
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                            hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);
hipError_t hipMemset(void* dst, int value, size_t bytes);
typedef void* hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t dst, int value, size_t count, hipStream_t) {
    int* p = static_cast<int*>(dst);
    for (size_t i = 0; i < count; ++i) p[i] = value;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipSetDevice(int dev);
hipError_t hipGetDevice(int* dev);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // one host thread: issue order
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { return hipStreamCreate(s); }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { return hipStreamCreate(s); }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e);
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // everything issued has already run
enum { hipHostMallocDefault = 0 };
static inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
}
template <class T>
static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
    size_t sharedMemPerBlock;
    char gcnArchName[256];
};
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int dev);
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// ---------------------------------------------------------------------------------------------- execution model
namespace hipemu {
constexpr int WAVE = 64;

struct Rendezvous {        // all live fibers of a group arrive, then the generation advances
    int arrived = 0;
    int live = 0;
    unsigned gen = 0;
};
struct WaveState {
    Rendezvous rv;
    alignas(16) unsigned char slot[2][WAVE][64];   // deposit area of the collectives, double-buffered on rv.gen & 1
};
struct Lane {
    uint3 tid;
    int linear;       // work-item index in the workgroup
    int lane;         // linear % 64
    WaveState* wave;
    // fiber
    void* sp;
    bool done;
    Rendezvous* waiting;
    unsigned wait_gen;
};

extern Lane* cur;
extern uint3 g_block_idx;
extern dim3 g_block_dim, g_grid_dim;
extern Rendezvous g_wg;

void yield_to_scheduler();
void arrive(Rendezvous* rv);             // blocks (yields) until every live fiber of the group has arrived
void* dynamic_lds();
void fail(const char* fmt, ...);         // records a launch failure; the current kernel is abandoned

typedef void (*Thunk)(void*);
void run_grid(dim3 grid, dim3 block, size_t lds_bytes, Thunk fn, void* closure);

template <class Tuple, class F, size_t... I>
void apply_impl(F f, Tuple& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }

template <class... P, class... A>
void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds_bytes, hipStream_t, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
    struct Closure {
        void (*k)(P...);
        std::tuple<std::decay_t<P>...> args;
    } c{kernel, std::tuple<std::decay_t<P>...>(static_cast<std::decay_t<P>>(a)...)};
    run_grid(grid, block, lds_bytes,
             [](void* p) {
                 Closure* cl = static_cast<Closure*>(p);
                 apply_impl(cl->k, cl->args, std::index_sequence_for<P...>{});
             },
             &c);
}

// ---- wave collectives.  Every live lane of the wave must execute the same sequence of collectives.
template <class T>
inline void deposit(const T& v) {
    static_assert(sizeof(T) <= 64, "collective payload");
    std::memcpy(cur->wave->slot[cur->wave->rv.gen & 1][cur->lane], &v, sizeof(T));
}
template <class T>
inline T peek(int p, int lane) {
    T r;
    std::memcpy(&r, cur->wave->slot[p][lane & (WAVE - 1)], sizeof(T));
    return r;
}
// deposit v, wait for the wave, return the parity of the buffer that now holds everybody's deposit
template <class T>
inline int exchange(const T& v) {
    const int p = (int)(cur->wave->rv.gen & 1);
    deposit(v);
    arrive(&cur->wave->rv);
    return p;
}

template <class T>
inline T shfl(T v, int src, int width = WAVE) {
    const int p = exchange(v);
    const int base = cur->lane & ~(width - 1);
    return peek<T>(p, base + (src & (width - 1)));
}
template <class T>
inline T shfl_xor(T v, int mask, int width = WAVE) {
    const int p = exchange(v);
    const int base = cur->lane & ~(width - 1);
    return peek<T>(p, base + ((cur->lane ^ mask) & (width - 1)));
}
template <class T>
inline T shfl_down(T v, int d, int width = WAVE) {
    const int p = exchange(v);
    const int l = cur->lane & (width - 1);
    return l + d < width ? peek<T>(p, cur->lane + d) : v;
}
template <class T>
inline T shfl_up(T v, int d, int width = WAVE) {
    const int p = exchange(v);
    const int l = cur->lane & (width - 1);
    return l - d >= 0 ? peek<T>(p, cur->lane - d) : v;
}
extern bool g_poison;
// build.py appends this to every __shared__ array declaration (HIPEMU_POISON=1: NaN bytes at the start of each workgroup)
inline void poison_lds(void* p, size_t bytes) {
    if (g_poison && cur->linear == 0) std::memset(p, 0xFF, bytes);
}
// rendezvous of the live lanes of the wave (build.py puts it where a "// [wave-lds-exchange]" marker stands)
inline void wave_sync() { arrive(&cur->wave->rv); }
inline int readlane(int v, int lane) {
    const int p = exchange(v);
    return peek<int>(p, lane);
}
int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl);

// ---- MFMA (gfx950 register layouts; lane l of the wave, 0 <= l < 64)
//   32x32 accumulators: 16 floats per lane, element r <-> row 8 (r / 4) + 4 (l / 32) + r % 4, column l % 32
//   32x32x2 f32:   A row l % 32, k = l / 32;                       B column l % 32, k = l / 32
//   32x32x16 f16:  A row l % 32, k = 8 (l / 32) + j, j = 0 .. 7;   B column l % 32, same k
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
v4f mfma_16x16x4_f32(float a, float b, v4f c);
v16f mfma_32x32x2_f32(float a, float b, v16f c);
v16f mfma_32x32x16_f16(v8h a, v8h b, v16f c);
v16f mfma_32x32x16_bf16(v8bf a, v8bf b, v16f c);

typedef __fp16 v2fp16 __attribute__((ext_vector_type(2)));
v2fp16 cvt_pkrtz(float a, float b);
// v_fma_mix_f32 d, h[sel], -1.0, x: x - float(h[sel]) in one rounding (exact whenever h is the rtz fp16 of x)
static inline float fma_mix_sub(unsigned packed_halves, int sel, float x) {
    const unsigned short bits = (unsigned short)(sel ? packed_halves >> 16 : packed_halves & 0xffffu);
    _Float16 h;
    std::memcpy(&h, &bits, 2);
    return std::fmaf((float)h, -1.0f, x);
}
}   // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::g_block_idx)
#define blockDim (hipemu::g_block_dim)
#define gridDim (hipemu::g_grid_dim)
#define warpSize 64

// The emulation runs the workgroups of a launch one after the other: kernels that synchronise across workgroups (pk_grid.h)
// cannot run here.  PK_HIPEMU tells the launchers, which then issue one launch per phase; a cooperative launch is an error.
#define PK_HIPEMU 1
static inline hipError_t hipLaunchCooperativeKernel(const void*, dim3, dim3, void**, unsigned, hipStream_t) { return hipErrorLaunchFailure; }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
    hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(lds), (hipStream_t)(stream), __VA_ARGS__)

static inline void __syncthreads() { hipemu::arrive(&hipemu::g_wg); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T> static inline T __shfl(T v, int src, int width = 64) { return hipemu::shfl(v, src, width); }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { return hipemu::shfl_xor(v, mask, width); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) { return hipemu::shfl_down(v, d, width); }
template <class T> static inline T __shfl_up(T v, int d, int width = 64) { return hipemu::shfl_up(v, d, width); }

// ---------------------------------------------------------------------------------------------- scalar intrinsics
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }
#define __expf(x) std::exp((float)(x))     // glibc declares __expf / __logf itself
#define __logf(x) std::log((float)(x))
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
// sincosf is provided by glibc
using std::max;
using std::min;

// single host thread: plain read-modify-write is atomic with respect to the other fibers
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }

#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_s_sleep(x) hipemu::yield_to_scheduler()
#define __builtin_amdgcn_exp2f(x) std::exp2((float)(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (float)(x))
#define __builtin_amdgcn_rsqf(x) (1.0f / std::sqrt((float)(x)))
static inline float hipemu_fmed3f(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) hipemu_fmed3f(a, b, c)
// used in this code base only to pin wave-uniform values into scalar registers
// v_mbcnt_lo / _hi with an all-ones mask (the only use): base + the lane's index among the lower / upper 32 lanes below it
#define __builtin_amdgcn_mbcnt_lo(m, b) ((unsigned)(b) + ((threadIdx.x & 63u) < 32u ? (threadIdx.x & 63u) : 32u))
#define __builtin_amdgcn_mbcnt_hi(m, b) ((unsigned)(b) + ((threadIdx.x & 63u) < 32u ? 0u : (threadIdx.x & 63u) - 32u))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_readlane(v, lane) hipemu::readlane(v, lane)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu::update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_cvt_pkrtz(a, b) hipemu::cvt_pkrtz(a, b)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_16x16x4_f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu::mfma_32x32x16_f16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_32x32x16_bf16(a, b, c)
