// Sustained dense bf16 MFMA rate of the card with NO memory traffic at all: every wave issues TILES independent
// v_mfma_f32_16x16x32_bf16 per loop trip (TILES = 36 is the accumulator shape of conv3x3_kernel / conv3x3_wgrad_kernel).
// Also reports the in-kernel shader clock (s_memtime ticks per s_memrealtime tick of 10 ns).  Measurement tool, not product.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TILES, int WPS>
__global__ void __launch_bounds__(256, WPS) mfma_loop(float *out, unsigned long long *clk, int trips, float seed)
{
    f32x4 acc[TILES];
    for (int i = 0; i < TILES; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (int i = 0; i < 8; ++i) {                       // pseudo-random operands in [-1, 1): DVFS depends on the data
        h = h * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(h >> 8) & 0xFFFF) / 32768.0f - 1.0f);
        h = h * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(h >> 8) & 0xFFFF) / 32768.0f - 1.0f);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < TILES; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < TILES; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x < 1024) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

extern "C" int mfma_peak_launch(float *out, unsigned long long *clk, int blocks, int trips, int tiles, int wps, void *stream)
{
#define GO(T, W) if (tiles == T && wps == W) { hipLaunchKernelGGL((mfma_loop<T, W>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, clk, trips, seed); return (int)hipGetLastError(); }
    const float seed = 1.0f;
    GO(36, 1) GO(36, 2) GO(36, 3) GO(24, 2) GO(24, 4) GO(16, 1) GO(16, 2) GO(16, 4) GO(16, 8) GO(8, 4) GO(8, 8)
    return -1;
}
