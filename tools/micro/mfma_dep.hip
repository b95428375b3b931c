// How much does a chain of MFMAs on ONE accumulator cost against the same MFMAs spread over independent accumulators?
// The split-fp16 kernels issue a_hi*b_hi, a_lo*b_hi, a_hi*b_lo back to back into the same accumulator tile, for four tiles
// in turn ("triples").  256 workgroups x W waves (W = 4: one wave per SIMD, W = 8: two).  Modes:
//   0  one accumulator, every MFMA depends on the previous one
//   1  triples: c0 c0 c0 c1 c1 c1 c2 c2 c2 c3 c3 c3            (the kernels' order)
//   2  term-major: c0 c1 c2 c3 c0 c1 c2 c3 c0 c1 c2 c3          (distance 4)
//   3  pairs: c0 c1 c0 c1 c0 c1 c2 c3 c2 c3 c2 c3               (distance 2)
//   4  triples with 4 independent VALU instructions after every MFMA (does VALU hide in the dependency gap?)
//   5  term-major with the same VALU instructions
// Prints ns per MFMA per SIMD-wave and the implied cycles at the measured clock (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 1000;
#define M(c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define V4 v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f); __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
        if (MODE == 0) { M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); }
        if (MODE == 1) { M(c0); M(c0); M(c0); M(c1); M(c1); M(c1); M(c2); M(c2); M(c2); M(c3); M(c3); M(c3); }
        if (MODE == 2) { M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); }
        if (MODE == 3) { M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); M(c2); M(c3); M(c2); M(c3); M(c2); M(c3); }
        if (MODE == 4) { M(c0); V4; M(c0); V4; M(c0); V4; M(c1); V4; M(c1); V4; M(c1); V4; M(c2); V4; M(c2); V4; M(c2); V4; M(c3); V4; M(c3); V4; M(c3); V4; }
        if (MODE == 5) { M(c0); V4; M(c1); V4; M(c2); V4; M(c3); V4; M(c0); V4; M(c1); V4; M(c2); V4; M(c3); V4; M(c0); V4; M(c1); V4; M(c2); V4; M(c3); V4; }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = t1 - t0;
}

template <int MODE>
int run(float* out, unsigned long long* clk, int waves) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, out, clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double per_wave = ms * 1e6 / (ITER * 12.0);                 // ns per MFMA of one wave
    const double per_simd = per_wave / (waves / 4.0);                  // ns per MFMA per SIMD
    printf("mode %d waves/SIMD %d: %.3f ms, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz), s_memtime ticks per MFMA of wave 0: %.1f\n",
           MODE, waves / 4, ms, per_simd, per_simd * 2.4, (double)c / (ITER * 12.0));
    return 0;
}

int main() {
    float* out; unsigned long long* clk;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&clk, 8));
    for (int waves : {4, 8}) {
        if (run<0>(out, clk, waves) || run<1>(out, clk, waves) || run<2>(out, clk, waves) || run<3>(out, clk, waves) ||
            run<4>(out, clk, waves) || run<5>(out, clk, waves)) return 1;
    }
    return 0;
}
