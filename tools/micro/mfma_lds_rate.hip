// Does the LDS operand traffic of the PWG layer kernel (1.26 ds_read_b128 per MFMA, A fragments re-read for
// every 32-sample tile) cost matrix throughput?  256 workgroups x 8 waves; per iteration 12 MFMAs
// (32x32x16 f16) with R ds_read_b128 of fresh A fragments: R = 0 (registers only), 4, 8 (layer kernel: 2 per
// 3 MFMAs), 16, 24.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 1000;

template <int R>
__global__ __launch_bounds__(512, 2) void k(float* out) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[8192];   // 128 KB
    for (int i = threadIdx.x; i < 8192; i += 512) {
        f16x8 v;
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)((i + e) * 1e-4f);
        lds[i] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f16x8 b;
    for (int e = 0; e < 8; ++e) b[e] = (_Float16)(e * 0.25f);
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f16x8 a[12];
    for (int m = 0; m < 12; ++m) a[m] = lds[m * 64 + lane];
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (m * R / 12 != (m + 1) * R / 12 || (R >= 12)) {
#pragma unroll
                for (int q = 0; q < (R >= 12 ? R / 12 : 1); ++q)
                    a[(m + q) % 12] = lds[(((it * 12 + m) * 2 + q) * 64 + lane) & 8191];
            }
            if ((m & 3) == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b, c0, 0, 0, 0);
            if ((m & 3) == 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b, c1, 0, 0, 0);
            if ((m & 3) == 2) c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b, c2, 0, 0, 0);
            if ((m & 3) == 3) c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        const double mfma_per_simd = 2.0 * ITER * 12;
        printf("%-44s %.3f ms  -> %.1f ns per MFMA per SIMD (32 cycles at %.2f GHz)\n", name, ms, ms * 1e6 / mfma_per_simd,
               32.0 / (ms * 1e6 / mfma_per_simd));
        return 0;
    };
    run("0 ds_read_b128 per 12 MFMAs", k<0>);
    run("4 ds_read_b128 per 12 MFMAs", k<4>);
    run("8 ds_read_b128 per 12 MFMAs (layer kernel)", k<8>);
    run("12 ds_read_b128 per 12 MFMAs", k<12>);
    run("24 ds_read_b128 per 12 MFMAs", k<24>);
    return 0;
}
