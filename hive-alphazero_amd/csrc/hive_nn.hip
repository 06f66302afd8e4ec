// hive_nn.hip -- 3x3 convolution of the 12x12x256 residual tower as an implicit GEMM on the CDNA4
// matrix cores (v_mfma_f32_16x16x32_bf16), fused with bias, skip connection and ReLU.
//
// One workgroup (4 waves) = one board.  GEMM view per board: D[k][pixel] = sum_{tap,c} W[tap][k][c] *
// X[pixel + tap][c]  (M = 256 output channels, N = 144 pixels, K = 9 * C_in).
//   * The board (144 pixels x C_in bf16, 74 KB) is staged ONCE in LDS; the nine taps are nine
//     shifted views of it, out-of-board pixels read as zero (no halo in LDS, so two workgroups fit
//     a CU and one stages while the other computes).  Pixel stride = 2*C_in + 32 bytes: the
//     ds_read_b128 of a B fragment (16 pixels x 4 k-groups) is bank-conflict free.
//   * Each wave owns 64 output channels (4 M tiles) x all 9 pixel tiles = 36 accumulator tiles
//     (144 VGPRs); per 32-deep k-step it reads 9 B fragments from LDS (shared by the 4 waves) and
//     4 A fragments (weights) straight from L2 into registers, double buffered one step ahead.
//   * D's layout puts 4 consecutive output channels of one pixel in each lane: the epilogue adds
//     bias (+ skip), applies ReLU, rounds once to bf16 and stores 8 bytes per lane, channels-last.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hive_abi.h"
#include "../../include/hive_nn.h"

#ifndef HIVE_CONV_PAD
#define HIVE_CONV_PAD 48       // bytes added to the LDS pixel stride (bank spreading of the B-fragment reads)
#endif
#ifndef HIVE_CONV_BDEPTH
#define HIVE_CONV_BDEPTH 3     // LDS B-fragment reads kept in flight
#endif
#ifndef HIVE_CONV_WAVES
#define HIVE_CONV_WAVES 4      // waves per board workgroup (4: 4 M tiles per wave, 2 workgroups per CU; measured faster than 8)
#endif

namespace hive {
int set_error(int code, const std::string &msg);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CINP, bool RES, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64, NWAVE / 2)
conv3x3_kernel(const __bf16 *__restrict__ X, int cin, const __bf16 *__restrict__ W, const float *__restrict__ bias,
               const __bf16 *__restrict__ R, __bf16 *__restrict__ Y, int relu)
{
    constexpr int PS = CINP * 2 + HIVE_CONV_PAD;   // pixel stride in LDS, bytes
    constexpr int KC = CINP / 32;              // 32-deep k-steps per tap
    constexpr int MT = 16 / NWAVE;             // 16-channel M tiles per wave
    constexpr int NT = NWAVE * 64;             // threads
    constexpr unsigned ZOFF = 144 * PS;        // a zeroed pixel: what every off-board tap reads
    __shared__ __attribute__((aligned(16))) unsigned char lds[145 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long long b = blockIdx.x;

    // ---- stage the board: 16-byte chunks, zero the channel padding
    {
        const int cpp = cin * 2 / 16;                       // chunks per pixel in global memory
        const uint4 *src = reinterpret_cast<const uint4 *>(X + b * 144 * cin);
#ifdef HIVE_CONV_ABL_STAGE
        for (int i = tid; i < 1 * cpp; i += NT) {
#else
        for (int i = tid; i < 144 * cpp; i += NT) {
#endif
            int pix = i / cpp, c = i - pix * cpp;
            *reinterpret_cast<uint4 *>(lds + pix * PS + c * 16) = src[i];
        }
        for (int i = tid; i < PS / 16; i += NT)
            *reinterpret_cast<uint4 *>(lds + ZOFF + i * 16) = make_uint4(0u, 0u, 0u, 0u);
        const int padc = CINP * 2 / 16 - cpp;               // 0 or 1 chunk of zero channels
        for (int i = tid; i < 144 * padc; i += NT) {
            int pix = i / padc, c = cpp + (i - pix * padc);
            *reinterpret_cast<uint4 *>(lds + pix * PS + c * 16) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    f32x4 acc[MT][9];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weights are stored fragment-major: W[tap][kc][m-tile][lane][8], so one A fragment of a wave is
    // one contiguous 1 KiB block (full 128-byte lines from L2, no 16-row gather)
    const __bf16 *wbase = W + ((size_t)(wave * MT) * 64 + lane) * 8;
    bf16x8 A[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) A[0][mt] = *reinterpret_cast<const bf16x8 *>(wbase + (size_t)mt * 512);
    __syncthreads();

    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        // LDS byte offset of the shifted pixel; off-board taps read the zero pixel (no exec-masked loads,
        // so the ds_reads can be issued ahead of the MFMAs that consume them)
        unsigned boff[9];
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int pixel = nt * 16 + lr, y0 = pixel / 12;      // recomputed per tap: cheaper than 9 live VGPRs
            int sy = y0 + dy, sx = pixel - 12 * y0 + dx;
            bool inb = (unsigned)sy < 12u && (unsigned)sx < 12u;
            boff[nt] = (inb ? (unsigned)((sy * 12 + sx) * PS) : ZOFF) + (unsigned)(lg * 16);
        }
#pragma unroll 1
        for (int kc2 = 0; kc2 < KC; kc2 += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kc = kc2 + half;
                // prefetch the next step's weight fragments into the other buffer
                {
                    int nkc = kc + 1, ntap = tap;
                    if (nkc == KC) { nkc = 0; ntap = tap + 1; }
#ifdef HIVE_CONV_ABL_A
                    if (false) {
#else
                    if (ntap < 9) {
#endif
                        const __bf16 *wp = wbase + (size_t)(ntap * KC + nkc) * (16 * 512);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            A[half ^ 1][mt] = *reinterpret_cast<const bf16x8 *>(wp + (size_t)mt * 512);
                    }
                }
                // B fragments: keep HIVE_CONV_BDEPTH reads in flight ahead of the MFMAs that consume them
                bf16x8 Bf[9];
#pragma unroll
                for (int nt = 0; nt < HIVE_CONV_BDEPTH; ++nt)
                    Bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + boff[nt] + kc * 64);
#pragma unroll
                for (int nt = 0; nt < 9; ++nt) {
                    if (nt + HIVE_CONV_BDEPTH < 9)
                        Bf[nt + HIVE_CONV_BDEPTH] =
                            *reinterpret_cast<const bf16x8 *>(lds + boff[nt + HIVE_CONV_BDEPTH] + kc * 64);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[half][mt], Bf[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds D[ch0 .. ch0+3][pixel], ch0 = (wave*MT + mt)*16 + lg*4, pixel = nt*16 + lr
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ch0 = (wave * MT + mt) * 16 + lg * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(bias + ch0);
#ifdef HIVE_CONV_ABL_EPI
        for (int nt = 0; nt < 9; ++nt) {
            if (acc[mt][nt][0] != 12345.678f) continue;
#else
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
#endif
            const int pixel = nt * 16 + lr;
            const size_t o = ((size_t)b * 144 + pixel) * 256 + ch0;
            float v0 = acc[mt][nt][0] + bv.x, v1 = acc[mt][nt][1] + bv.y, v2 = acc[mt][nt][2] + bv.z,
                  v3 = acc[mt][nt][3] + bv.w;
            if (RES) {
                bf16x4 r = *reinterpret_cast<const bf16x4 *>(R + o);
                v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
            }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            bf16x4 out = {(__bf16)v0, (__bf16)v1, (__bf16)v2, (__bf16)v3};
            *reinterpret_cast<bf16x4 *>(Y + o) = out;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// One residual block in one launch:  y = relu(conv2(relu(conv1(x) + b1)) + b2 + x)   (alpha_net.py:36-54)
// The intermediate activation never leaves the CU: after conv1 its bf16 result overwrites the
// board in LDS (all waves have finished reading x by then), conv2 reads it from there, and the
// skip connection re-reads x from global memory (L2 / Infinity-Cache resident: it was fetched by
// this very workgroup microseconds earlier).  Per block this removes one 75 MB write and one
// 75 MB read of the intermediate tensor and one staging + one store phase.
template <int MT, int KC, int PS>
__device__ __forceinline__ void conv_taps(const unsigned char *lds, unsigned zoff, const __bf16 *wbase, int lr, int lg,
                                          f32x4 (&acc)[MT][9], bf16x8 (&A)[2][MT])
{
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        unsigned boff[9];
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int pixel = nt * 16 + lr, y0 = pixel / 12;
            int sy = y0 + dy, sx = pixel - 12 * y0 + dx;
            bool inb = (unsigned)sy < 12u && (unsigned)sx < 12u;
            boff[nt] = (inb ? (unsigned)((sy * 12 + sx) * PS) : zoff) + (unsigned)(lg * 16);
        }
#pragma unroll 1
        for (int kc2 = 0; kc2 < KC; kc2 += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kc = kc2 + half;
                {
                    int nkc = kc + 1, ntap = tap;
                    if (nkc == KC) { nkc = 0; ntap = tap + 1; }
                    if (ntap < 9) {
                        const __bf16 *wp = wbase + (size_t)(ntap * KC + nkc) * (16 * 512);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            A[half ^ 1][mt] = *reinterpret_cast<const bf16x8 *>(wp + (size_t)mt * 512);
                    }
                }
                bf16x8 Bf[9];
#pragma unroll
                for (int nt = 0; nt < HIVE_CONV_BDEPTH; ++nt)
                    Bf[nt] = *reinterpret_cast<const bf16x8 *>(lds + boff[nt] + kc * 64);
#pragma unroll
                for (int nt = 0; nt < 9; ++nt) {
                    if (nt + HIVE_CONV_BDEPTH < 9)
                        Bf[nt + HIVE_CONV_BDEPTH] =
                            *reinterpret_cast<const bf16x8 *>(lds + boff[nt + HIVE_CONV_BDEPTH] + kc * 64);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[half][mt], Bf[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2)
resblock_kernel(const __bf16 *__restrict__ X, const __bf16 *__restrict__ W1, const float *__restrict__ b1,
                const __bf16 *__restrict__ W2, const float *__restrict__ b2, __bf16 *__restrict__ Y)
{
    constexpr int CINP = 256, KC = 8, MT = 4, NT = 256;
    constexpr int PS = CINP * 2 + HIVE_CONV_PAD;
    constexpr unsigned ZOFF = 144 * PS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[145 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long long b = blockIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(X + b * 144 * 256);
        for (int i = tid; i < 144 * 32; i += NT) *reinterpret_cast<uint4 *>(lds + (i >> 5) * PS + (i & 31) * 16) = src[i];
        for (int i = tid; i < PS / 16; i += NT) *reinterpret_cast<uint4 *>(lds + ZOFF + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    f32x4 acc[MT][9];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t woff = ((size_t)(wave * MT) * 64 + lane) * 8;
    bf16x8 A[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) A[0][mt] = *reinterpret_cast<const bf16x8 *>(W1 + woff + (size_t)mt * 512);
    __syncthreads();
    conv_taps<MT, KC, PS>(lds, ZOFF, W1 + woff, lr, lg, acc, A);

    // conv1 epilogue: relu(acc + b1) -> bf16, written over the board in LDS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) A[0][mt] = *reinterpret_cast<const bf16x8 *>(W2 + woff + (size_t)mt * 512);
    __syncthreads();                      // every wave has finished reading x from LDS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ch0 = (wave * MT + mt) * 16 + lg * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(b1 + ch0);
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int pixel = nt * 16 + lr;
            bf16x4 out = {(__bf16)fmaxf(acc[mt][nt][0] + bv.x, 0.f), (__bf16)fmaxf(acc[mt][nt][1] + bv.y, 0.f),
                          (__bf16)fmaxf(acc[mt][nt][2] + bv.z, 0.f), (__bf16)fmaxf(acc[mt][nt][3] + bv.w, 0.f)};
            *reinterpret_cast<bf16x4 *>(lds + pixel * PS + ch0 * 2) = out;
            acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();                      // the intermediate board is complete
    conv_taps<MT, KC, PS>(lds, ZOFF, W2 + woff, lr, lg, acc, A);

    // conv2 epilogue: + b2 + x (skip), relu, one rounding, channels-last store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ch0 = (wave * MT + mt) * 16 + lg * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(b2 + ch0);
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int pixel = nt * 16 + lr;
            const size_t o = ((size_t)b * 144 + pixel) * 256 + ch0;
            bf16x4 r = *reinterpret_cast<const bf16x4 *>(X + o);
            bf16x4 out = {(__bf16)fmaxf(acc[mt][nt][0] + bv.x + (float)r[0], 0.f),
                          (__bf16)fmaxf(acc[mt][nt][1] + bv.y + (float)r[1], 0.f),
                          (__bf16)fmaxf(acc[mt][nt][2] + bv.z + (float)r[2], 0.f),
                          (__bf16)fmaxf(acc[mt][nt][3] + bv.w + (float)r[3], 0.f)};
            *reinterpret_cast<bf16x4 *>(Y + o) = out;
        }
    }
}

}  // namespace hive

using namespace hive;

extern "C" int hive_nn_conv3x3(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                               int batch, int relu, void *stream)
{
    if (!x || !w || !bias || !y || batch <= 0) return set_error(HIVE_E_ARG, "hive_nn_conv3x3: bad argument");
    const __bf16 *X = (const __bf16 *)x, *Wt = (const __bf16 *)w, *R = (const __bf16 *)residual;
    __bf16 *Y = (__bf16 *)y;
    hipStream_t s = (hipStream_t)stream;
    constexpr int NWV = HIVE_CONV_WAVES;
    dim3 grid((unsigned)batch), block(NWV * 64);
    if (cin == 256) {
        if (R) hipLaunchKernelGGL((conv3x3_kernel<256, true, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu);
        else hipLaunchKernelGGL((conv3x3_kernel<256, false, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu);
    } else if (cin == 56) {
        if (R) hipLaunchKernelGGL((conv3x3_kernel<64, true, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu);
        else hipLaunchKernelGGL((conv3x3_kernel<64, false, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu);
    } else {
        return set_error(HIVE_E_ARG, "hive_nn_conv3x3: cin must be 56 or 256");
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_conv3x3: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" int hive_nn_resblock(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                                int batch, void *stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || batch <= 0 || x == y)
        return set_error(HIVE_E_ARG, "hive_nn_resblock: bad argument (y must not alias x)");
    hipLaunchKernelGGL(resblock_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                       (const __bf16 *)w1, b1, (const __bf16 *)w2, b2, (__bf16 *)y);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_resblock: ") + hipGetErrorString(e));
    return HIVE_OK;
}
