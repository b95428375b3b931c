// Memory-system ceiling for the PWG residual-block access pattern (no matrix work):
// per 32-sample wave-tile read 3 taps x 64 channels of x (blocked [t/32][ch][32] layout), read 64 skip
// channels, write 64 x channels and 64 skip channels.  Build: hipcc --offload-arch=gfx950 -O3 stream_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int XB = 2048;
__device__ __forceinline__ long xoff(long t) { return (t >> 5) * XB + (t & 31); }
__device__ __forceinline__ int mrow(int r) { return (r & 3) + 8 * (r >> 2); }

template <int TAPS, bool SKIP>
__global__ __launch_bounds__(512, 2) void k_stream(const float* __restrict__ xin, float* __restrict__ xout,
                                                   float* __restrict__ skip, long t_first, int n_wtiles, int d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
    const int per_xcd = gridDim.x >> 3;
    const int wg_slot = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int nw = blockDim.x >> 6;
    for (int wt = wg_slot * nw + wave; wt < n_wtiles; wt += gridDim.x * nw) {
        const long t0 = t_first + (long)wt * 32 + j;
        float acc[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) acc[e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const long tt = t0 + (long)(tap - TAPS / 2) * d;
            const float* p = xin + xoff(tt) + 4 * hi * 32;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[16 * q + r] += p[(long)(32 * q + mrow(r)) * 32];
        }
        const long o = xoff(t0) + 4 * hi * 32;
        float sk[32];
        if (SKIP) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) sk[16 * q + r] = skip[o + (long)(32 * q + mrow(r)) * 32];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xout[o + (long)(32 * q + mrow(r)) * 32] = acc[16 * q + r];
                if (SKIP) skip[o + (long)(32 * q + mrow(r)) * 32] = sk[16 * q + r] + acc[16 * q + r];
            }
    }
}

int main() {
    const long S = 32L * 640 * 256, GAP = 1024, T = S + 2 * GAP;
    float *x0, *x1, *sk;
    CK(hipMalloc(&x0, T * 64 * 4 + (64 << 20))); CK(hipMalloc(&x1, T * 64 * 4 + (64 << 20))); CK(hipMalloc(&sk, T * 64 * 4 + (64 << 20)));
    CK(hipMemset(x0, 0, T * 64 * 4)); CK(hipMemset(x1, 0, T * 64 * 4)); CK(hipMemset(sk, 0, T * 64 * 4));
    const int n_wt = (int)(S / 32);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern, int grid, int block, int d, double bytes_per_sample) -> int {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, x0, x1, sk, GAP, n_wt, d);
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, x0, x1, sk, GAP, n_wt, d);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-28s grid %4d x %3d  d=%3d : %.3f ms  -> %.2f TB/s of minimal traffic (%.0f B/sample)\n", name, grid, block, d, ms,
               bytes_per_sample * S / ms / 1e9, bytes_per_sample);
        return 0;
    };
    for (long shift : {0L, 8192L + 512, 65536L + 4096 + 256, (1L << 20) + 8192 + 1024}) {
        printf("-- buffer shifts: x1 += %ld floats, skip += %ld floats\n", shift, 2 * shift);
        float* x1s = x1 + shift; float* sks = sk + 2 * shift;
        auto run2 = [&](const char* name, auto kern, int grid, int block, int d) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, x0, x1s, sks, GAP, n_wt, d);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, x0, x1s, sks, GAP, n_wt, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%-28s d=%3d : %.3f ms\n", name, d, ms);
        };
        run2("1 tap + skip rmw", k_stream<1, true>, 256, 512, 1);
        run2("3 taps + skip rmw", k_stream<3, true>, 256, 512, 4);
        run2("3 taps + skip rmw", k_stream<3, true>, 256, 512, 128);
    }
    for (int grid : {256}) for (int block : {512}) {
        run("x copy (1 tap, no skip)", k_stream<1, false>, grid, block, 1, 512);
        run("1 tap + skip rmw", k_stream<1, true>, grid, block, 1, 1024);
        for (int d : {1, 32, 512}) run("3 taps + skip rmw", k_stream<3, true>, grid, block, d, 1024);
    }
    run("3 taps + skip rmw", k_stream<3, true>, 1024, 256, 512, 1024);
    run("3 taps + skip rmw", k_stream<3, true>, 2048, 256, 512, 1024);
    return 0;
}
