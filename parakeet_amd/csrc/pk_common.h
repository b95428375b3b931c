// pk_common.h -- internals shared by the translation units of libpk_synth.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pk_synth.h"

void pk_set_error(const char* fmt, ...);

// Measurement / ablation switches (environment variables read by the launchers: tile overrides, timing ablations that give
// WRONG results, traces) exist only in the profile build -- parakeet_amd/build.py build(profile=True) compiles
// libpk_synth_prof.so with -DPK_PROFILE_BUILD=1, the tools/ scripts load that one.  In the product library this function is
// a constant nullptr: every `if (pk_prof_env("PK_..."))` folds away at compile time, the names do not reach the binary and
// no environment variable changes what the library computes.  Behaviour a caller may legitimately choose goes through the
// documented pk_*_set_math / pk_*_set_option calls of pk_synth.h.
#ifndef PK_PROFILE_BUILD
#define PK_PROFILE_BUILD 0
#endif
static inline const char* pk_prof_env(const char* name) {
    if constexpr (PK_PROFILE_BUILD != 0) return std::getenv(name);
    else return nullptr;
}

#define PK_FAIL(code, ...)            \
    do {                              \
        pk_set_error(__VA_ARGS__);    \
        return (code);                \
    } while (0)

#define PK_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            pk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                         __FILE__, __LINE__);                                     \
            return PK_EHIP;                                                       \
        }                                                                         \
    } while (0)

#define PK_TRY(expr)                  \
    do {                              \
        int _s = (expr);              \
        if (_s != PK_OK) return _s;   \
    } while (0)

// Engine calls run on the context's device and leave the caller's current HIP device as they found it
// (torch tracks its own current device; a stray hipSetDevice inside a ctypes call would silently redirect
// the caller's next allocations in single-process multi-GPU use).
struct pk_device_guard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit pk_device_guard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
        else prev = -1;   // nothing to restore
    }
    ~pk_device_guard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    pk_device_guard(const pk_device_guard&) = delete;
    pk_device_guard& operator=(const pk_device_guard&) = delete;
};
#define PK_DEVICE(dev)                                                                          \
    pk_device_guard _dg(dev);                                                                   \
    if (_dg.err != hipSuccess) PK_FAIL(PK_EHIP, "hipSetDevice(%d) failed: %s", (int)(dev), hipGetErrorString(_dg.err))

struct pk_prof_rec {
    int name_id;
    hipEvent_t start, stop;
};

struct pk_ctx_scratch;
struct pk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 256;
    // profiler
    bool prof_on = false;
    std::vector<std::string> prof_names;
    std::map<std::string, int> prof_ids;
    std::vector<pk_prof_rec> prof_recs;
    std::vector<hipEvent_t> event_pool;
    // small grow-only device buffers shared by the launchers.  Stream-ordered, so one set PER STREAM the context has been
    // bound to: two engine handles issued concurrently on two streams of one context (Synthesizer.issue_acoustic) never
    // share a scratch buffer.
    std::map<hipStream_t, pk_ctx_scratch*> scratch;

    int prof_begin(const char* name);   // returns record index or -1
    void prof_end(int rec);
};

// Bracket a kernel launch with profiler events when enabled.
#define PK_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                       \
    do {                                                                            \
        int _rec = (ctx)->prof_on ? (ctx)->prof_begin(name) : -1;                   \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__); \
        if (_rec >= 0) (ctx)->prof_end(_rec);                                       \
        PK_HIP(hipGetLastError());                                                  \
    } while (0)

// Grow-only device buffer.
struct pk_dbuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return PK_OK;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) PK_FAIL(PK_EHIP, "hipFree failed: %s", hipGetErrorString(e));
        }
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            PK_FAIL(PK_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return PK_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// per-(context, stream) scratch (allocated on first use, freed by pk_ctx_destroy)
struct pk_ctx_scratch {
    pk_dbuf row_amax;    // pk_gemm_launch: max|A[r, :]| per row when the caller does not supply it
    pk_dbuf row_amax2;
    pk_dbuf attn_amax;   // run_attention: max|q|, |k|, |v| per (utterance, head)
};
pk_ctx_scratch* pk_ctx_get_scratch(pk_ctx* ctx);

struct pk_param {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

typedef std::map<std::string, pk_param> pk_param_map;

int pk_store_param(pk_param_map& m, const char* name, const float* data, const int64_t* shape,
                   int32_t ndim);
// Returns the plain weight for `base` ("conv_layers.0.conv"): either
// base.weight, or base.weight_g * base.weight_v / ||v|| (weight-norm fold,
// nn.utils.remove_weight_norm; norm over all axes but 0).
int pk_get_weight(const pk_param_map& m, const std::string& base, const std::vector<int64_t>& shape,
                  std::vector<float>& out);
int pk_get_vector(const pk_param_map& m, const std::string& name, int64_t n, std::vector<float>& out);

int pk_upload(pk_ctx* ctx, pk_dbuf& buf, const void* host, size_t bytes);

// ops.hip: standard-normal stream (Philox4x32-10 + Box-Muller) into device memory, on ctx->stream
int pk_randn_device(pk_ctx* ctx, float* d_out, long n, unsigned long long seed, unsigned long long offset);

static inline int pk_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
