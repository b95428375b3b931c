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
    // working-set sweep: does a chunk that fits the 256 MB Infinity Cache stream faster than the full batch?
    const long GAP = 1024;
    const long Smax = 32L * 640 * 256, Tmax = Smax + 2 * GAP;
    float *x0, *x1, *sk;
    CK(hipMalloc(&x0, Tmax * 64 * 4)); CK(hipMalloc(&x1, Tmax * 64 * 4)); CK(hipMalloc(&sk, Tmax * 64 * 4));
    CK(hipMemset(x0, 0, Tmax * 64 * 4)); CK(hipMemset(x1, 0, Tmax * 64 * 4)); CK(hipMemset(sk, 0, Tmax * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int utts : {1, 2, 3, 4, 8, 32}) {
        const long S = (long)utts * 640 * 256;
        const int n_wt = (int)(S / 32);
        for (int d : {1, 128}) {
            // ping-pong like the layer stack: x0 -> x1, x1 -> x0, ... (30 passes)
            for (int i = 0; i < 4; ++i)
                hipLaunchKernelGGL((k_stream<3, true>), dim3(256), dim3(512), 0, 0, (i & 1) ? x1 : x0, (i & 1) ? x0 : x1, sk, GAP, n_wt, d);
            CK(hipEventRecord(e0));
            const int reps = 30;
            for (int i = 0; i < reps; ++i)
                hipLaunchKernelGGL((k_stream<3, true>), dim3(256), dim3(512), 0, 0, (i & 1) ? x1 : x0, (i & 1) ? x0 : x1, sk, GAP, n_wt, d);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("utterances %2d (working set %6.1f MB) d=%3d : %.4f ms per pass -> %.2f TB/s of minimal traffic, %.3f ms per 32-utt equivalent\n",
                   utts, 3.0 * S * 256 / 1e6, d, ms, 1024.0 * S / ms / 1e9, ms * 32 / utts);
        }
    }
    return 0;
}
