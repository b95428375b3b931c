// Can the VALU work of one wave hide under the MFMAs of another wave of the same SIMD (and of the same wave)?
// 256 workgroups x 8 waves (2 per SIMD).  Modes: 0 = all waves MFMA only, 1 = all waves VALU only,
// 2 = waves 0-3 MFMA / waves 4-7 VALU (one of each per SIMD), 3 = every wave alternates 1 MFMA : 8 VALU in
// one instruction stream, 4 = every wave does the MFMA block then the VALU block (phases, like a real kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2000;

__global__ __launch_bounds__(512, 2) void k(float* out, int mode) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const bool do_m = mode == 0 || (mode == 2 && wave < 4);
    const bool do_v = mode == 1 || (mode == 2 && wave >= 4);
    if (do_m) {
        for (int i = 0; i < ITER; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
    }
    if (do_v) {
        for (int i = 0; i < ITER; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        }
    }
    if (mode == 3) {
        for (int i = 0; i < ITER; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
            v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
            v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (mode == 4) {
        for (int i = 0; i < ITER / 50; ++i) {
            for (int m = 0; m < 100; ++m) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int m = 0; m < 200; ++m) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"all waves: MFMA only (4 x ITER per wave)", "all waves: VALU only (32 x ITER FMAs per wave)",
                           "per SIMD: one MFMA wave + one VALU wave", "every wave: 1 MFMA : 8 VALU interleaved (2 x ITER MFMA, 16 x ITER VALU)",
                           "every wave: phases of 200 MFMA then 1600 VALU (4 x ITER MFMA, 32 x ITER VALU)"};
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode %d  %-90s %.3f ms\n", mode, names[mode], ms / 5);
    }
    return 0;
}
