// Sustained dense bf16 MFMA rate of the card with NO memory traffic at all: every wave issues 36 independent
// v_mfma_f32_16x16x32_bf16 per loop trip (the accumulator shape of conv3x3_kernel).  Measurement tool, not product.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TILES>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int trips, float seed)
{
    f32x4 acc[TILES];
    for (int i = 0; i < TILES; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 1e-3f); b[i] = (__bf16)(seed * 0.5f + i); }
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < TILES; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < TILES; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int mfma_peak_launch(float *out, int blocks, int trips, int tiles, void *stream)
{
    if (tiles == 36) hipLaunchKernelGGL(mfma_loop<36>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, trips, 1.0f);
    else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, trips, 1.0f);
    return (int)hipGetLastError();
}
