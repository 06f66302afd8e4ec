// hive_nn.hip -- the 12x12x256 residual tower of the leaf evaluator as implicit GEMMs on the CDNA4 matrix cores
// (v_mfma_f32_16x16x32_bf16 / _f16), fused with bias, skip connection and ReLU.
//
// GEMM view per board: D[k][pixel] = sum_{tap,c} W[tap][k][c] * X[pixel + tap][c]   (M = 256 output channels,
// N = 144 pixels, K = 9 * C_in).  Common to every kernel here:
//   * The board (144 pixels x C_in, 74 KB) is staged ONCE in LDS; the nine taps are nine shifted views of it,
//     out-of-board pixels read a zeroed pixel (no halo, no exec-masked reads).  Pixel stride = 2*C_in + pad bytes so
//     that the ds_read_b128 of a B fragment (16 pixels x 4 k-groups) spreads over the banks.
//   * Weights are stored fragment-major (one A fragment = one contiguous 1 KiB block) and go straight from L2 into
//     registers, double buffered one 32-deep k-step ahead; a wave owns 64 output channels (4 M tiles).
//   * D's layout puts 4 consecutive output channels of one pixel in each lane: the epilogue adds bias (+ skip),
//     applies ReLU, rounds once and stores 8 bytes per lane, channels-last.
//
// conv3x3_kernel   one convolution, one board per 4-wave workgroup, two workgroups per CU (stem, training step).
// resblock_kernel  both convolutions of one residual block, the intermediate activation stays in LDS.
// tower_kernel     the WHOLE tower (any number of residual blocks) in one launch: the board never leaves LDS between
//                  blocks; only the skip operand is re-read from (and every block's output written to) global memory.
//                  NB = 2 puts TWO boards in one workgroup's LDS (157 KB) and gives every wave both boards' pixel
//                  tiles (4 M tiles x 18 N tiles = 72 accumulator tiles, one wave per SIMD): every weight fragment
//                  fetched from L2 feeds two boards -- half the weight stream of the one-board form at the same LDS
//                  fragment traffic per MFMA.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <string>

#include "../../include/hive_abi.h"
#include "../../include/hive_nn.h"

#ifndef HIVE_CONV_PAD
#define HIVE_CONV_PAD 48       // bytes added to the LDS pixel stride (bank spreading of the B-fragment reads)
#endif
#ifndef HIVE_TOWER_PAD
#define HIVE_TOWER_PAD 32      // tower_kernel: pixel stride = 34 sixteen-byte slots -> the 16 lanes of a ds_read_b128 phase
#endif                         // (8 pixels at k-group g, the other 8 at g+1) fall on 16 different slots
#ifndef HIVE_CONV_BDEPTH
#define HIVE_CONV_BDEPTH 3     // LDS B-fragment reads kept in flight
#endif
#ifndef HIVE_TOWER_BDEPTH
#define HIVE_TOWER_BDEPTH 8    // tower_kernel at one wave per SIMD (512 registers): B-fragment reads in flight ...
#endif
#ifndef HIVE_TOWER_ADIST
#define HIVE_TOWER_ADIST 2     // ... and k-steps the weight fragments are fetched ahead
#endif
#ifndef HIVE_CONV_WAVES
#define HIVE_CONV_WAVES 4      // waves per board workgroup (4: 4 M tiles per wave, 2 workgroups per CU; measured faster than 8)
#endif

namespace hive {
int set_error(int code, const std::string &msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// element traits: the two 16-bit formats the matrix cores take at the same rate
struct Bf16 {
    typedef __bf16 T;
    typedef __bf16 v8 __attribute__((ext_vector_type(8)));
    typedef __bf16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
struct F16 {
    typedef _Float16 T;
    typedef _Float16 v8 __attribute__((ext_vector_type(8)));
    typedef _Float16 v4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// ---------------------------------------------------------------------------------------------
// The nine taps of one 256 -> 256 convolution over NTL pixel tiles of the LDS image (tile 0 = the first tile of board
// slot0; 9 tiles per board): acc += W * shifted views.  72 k-steps (tap, 32 input channels) of 4 x NTL MFMAs, software
// pipelined across k-steps:
//   * weights: on entry A[0] holds the fragments of k-step 0; every k-step opens by issuing the NEXT k-step's four
//     fragment loads into the other buffer -- pinned there with a scheduling barrier: left alone, the compiler sinks
//     those loads to the end of the k-step to save registers and every wave then waits out an L2 round trip per k-step.
//     While the last k-step runs, the fragments of `wnext`'s k-step 0 (the next convolution of the tower, or nullptr)
//     are fetched, so a chain of convolutions never restarts its weight pipeline.
//   * B fragments: HIVE_CONV_BDEPTH LDS reads stay in flight ahead of the MFMAs that consume them, ACROSS the k-step
//     (and tap) boundary: the first reads of the next k-step are issued during the last MFMAs of this one.
template <typename E, int NTL, int PS, int D, int AD, bool UNROLL>
__device__ __forceinline__ void conv_taps(const unsigned char *lds, unsigned zoff, int slot0, const typename E::T *wbase,
                                          const typename E::T *wnext, unsigned wlane, int lr, int lg, f32x4 (&acc)[4][NTL],
                                          typename E::v8 (&A)[2 * AD][4])
{
    // wbase / wnext are the convolutions' packed weights (wave-uniform pointers: the loads use the scalar-base +
    // 32-bit lane offset form, no 64-bit vector address arithmetic); wlane = this lane's byte offset inside a k-step
    // D = B-fragment reads in flight; AD = how many k-steps ahead the weight fragments are fetched (ring of 2 AD buffers;
    // on entry A[0 .. AD-1] hold k-steps 0 .. AD-1, on exit they hold those of `wnext` if it is not null)
    typedef typename E::T T;
    typedef typename E::v8 v8;
    constexpr int MT = 4, KC = 8, RING = 2 * AD, KCU = UNROLL ? KC / RING : 1;
    constexpr size_t KSTEP = 16 * 512;                              // elements of one k-step's fragments (16 KiB)
    static_assert(D >= 1 && D < NTL, "B prefetch depth");
    static_assert(KC % RING == 0, "the k-steps of a tap must be a multiple of the weight ring");
    // LDS byte offset of this lane's fragment row of pixel tile nt under tap `tap`; off-board pixels read the zero pixel
    // (no exec-masked reads, so the ds_reads can be issued ahead of the MFMAs that consume them)
    auto tap_off = [&](int tap, int nt) -> unsigned {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int slot = slot0 + nt / 9;                           // board of this pixel tile (16 | 144: tiles never straddle)
        const int pixel = (nt % 9) * 16 + lr, y0 = pixel / 12;
        const int sy = y0 + dy, sx = pixel - 12 * y0 + dx;
        const bool inb = (unsigned)sy < 12u && (unsigned)sx < 12u;
        return (inb ? (unsigned)((slot * 144 + sy * 12 + sx) * PS) : zoff) + (unsigned)(lg * 16);
    };
    unsigned boff[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) boff[nt] = tap_off(0, nt);
    v8 Bn[D];                                                       // the first D fragments of the coming k-step
#pragma unroll
    for (int d = 0; d < D; ++d) Bn[d] = *reinterpret_cast<const v8 *>(lds + boff[d]);
    const char *wp = reinterpret_cast<const char *>(wbase + (size_t)(AD - 1) * KSTEP);   // newest k-step already requested
    const char *wtail = reinterpret_cast<const char *>(wnext ? wnext : wbase);   // fetched past the last k-step (unused if null)
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        unsigned bnext[D];                                          // where the next tap's first D fragments are
#pragma unroll
        for (int d = 0; d < D; ++d) bnext[d] = tap_off(tap < 8 ? tap + 1 : 8, d);
        // UNROLL: all 8 k-steps of a tap unrolled, so that the LDS offsets of the fragment reads become instruction
        // immediates (12 -> 4 address VALU instructions per k-step; the loop is short of issue slots, not of MFMA time)
#pragma unroll KCU
        for (int kcr = 0; kcr < KC; kcr += RING) {
#pragma unroll
            for (int q = 0; q < RING; ++q) {
                const int kc = kcr + q;
                const bool tap_ends = q == RING - 1 && kcr == KC - RING;
                // request the weight fragments of the k-step AD ahead (k-steps are 16 KiB apart; past the last one: the
                // next convolution's first ones)
                wp += KSTEP * sizeof(T);
                const int ahead = tap * KC + kc + AD - 9 * KC;      // >= 0: that k-step belongs to the next convolution
                const char *wl = ahead >= 0 ? wtail + (size_t)ahead * KSTEP * sizeof(T) : wp;
#if defined(HIVE_ABL_NOA)               /* timing-only ablation (profiles/r03_net_tower.md): no weight stream */
                if (wl == nullptr) A[(q + AD) % RING][0] = *reinterpret_cast<const v8 *>(wl + wlane);
#else
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    A[(q + AD) % RING][mt] = *reinterpret_cast<const v8 *>(wl + wlane + mt * 1024);
#endif
                __builtin_amdgcn_sched_barrier(0);
                v8 Bf[NTL];
#pragma unroll
                for (int d = 0; d < D; ++d) Bf[d] = Bn[d];
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
#ifdef HIVE_ABL_NOB                     /* timing-only ablation (profiles/r03_net_tower.md): no LDS fragment reads */
                    if (nt + D < NTL) {
                        Bf[nt + D] = Bf[nt];
                    } else if (boff[0] == 0xffffffffu) {
#else
                    if (nt + D < NTL) {
                        Bf[nt + D] = *reinterpret_cast<const v8 *>(lds + boff[nt + D] + kc * 64);
                    } else {
#endif
                        const int t = nt + D - NTL;                 // belongs to the next k-step
                        const unsigned o = tap_ends ? bnext[t] : boff[t] + (kc + 1) * 64;
                        Bn[t] = *reinterpret_cast<const v8 *>(lds + o);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][nt] = E::mfma(A[q][mt], Bf[nt], acc[mt][nt]);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) boff[nt] = tap_off(tap < 8 ? tap + 1 : 8, nt);
    }
}

template <typename E, int CINP, bool RES, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64, NWAVE / 2)
conv3x3_kernel(const typename E::T *__restrict__ X, int cin, const typename E::T *__restrict__ W, const float *__restrict__ bias,
               const typename E::T *__restrict__ R, typename E::T *__restrict__ Y, int relu, const int8_t *__restrict__ need)
{
    if (need && !need[blockIdx.x]) return;      // a row nobody will read (hive_search_leaf_need): its output keeps what it held
    typedef typename E::T T;
    typedef typename E::v8 v8;
    typedef typename E::v4 v4;
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

    if constexpr (CINP == 256) {
        // the tower's convolution: the software-pipelined tap loop shared with resblock_kernel / tower_kernel
        static_assert(NWAVE == 4, "conv_taps gives every wave 4 M tiles");
        const unsigned wlane = (unsigned)(((wave * MT) * 64 + lane) * 16);
        v8 A[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            A[0][mt] = *reinterpret_cast<const v8 *>(reinterpret_cast<const char *>(W) + wlane + mt * 1024);
        __syncthreads();
        conv_taps<E, 9, PS, HIVE_CONV_BDEPTH, 1, true>(lds, ZOFF, 0, W, nullptr, wlane, lr, lg, acc, A);
    } else {
        // weights are stored fragment-major: W[tap][kc][m-tile][lane][8], so one A fragment of a wave is
        // one contiguous 1 KiB block (full 128-byte lines from L2, no 16-row gather)
        const T *wbase = W + ((size_t)(wave * MT) * 64 + lane) * 8;
        v8 A[2][MT];
    #pragma unroll
        for (int mt = 0; mt < MT; ++mt) A[0][mt] = *reinterpret_cast<const v8 *>(wbase + (size_t)mt * 512);
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
                            const T *wp = wbase + (size_t)(ntap * KC + nkc) * (16 * 512);
    #pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                A[half ^ 1][mt] = *reinterpret_cast<const v8 *>(wp + (size_t)mt * 512);
                        }
                    }
                    // B fragments: keep HIVE_CONV_BDEPTH reads in flight ahead of the MFMAs that consume them
                    v8 Bf[9];
    #pragma unroll
                    for (int nt = 0; nt < HIVE_CONV_BDEPTH; ++nt)
                        Bf[nt] = *reinterpret_cast<const v8 *>(lds + boff[nt] + kc * 64);
    #pragma unroll
                    for (int nt = 0; nt < 9; ++nt) {
                        if (nt + HIVE_CONV_BDEPTH < 9)
                            Bf[nt + HIVE_CONV_BDEPTH] =
                                *reinterpret_cast<const v8 *>(lds + boff[nt + HIVE_CONV_BDEPTH] + kc * 64);
    #pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][nt] = E::mfma(A[half][mt], Bf[nt], acc[mt][nt]);
                    }
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
                v4 r = *reinterpret_cast<const v4 *>(R + o);
                v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
            }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            v4 out = {(T)v0, (T)v1, (T)v2, (T)v3};
            *reinterpret_cast<v4 *>(Y + o) = out;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// One residual block in one launch:  y = relu(conv2(relu(conv1(x) + b1)) + b2 + x)   (alpha_net.py:36-54)
// The intermediate activation never leaves the CU: after conv1 its 16-bit result overwrites the board in LDS (all
// waves have finished reading x by then), conv2 reads it from there, and the skip connection re-reads x from global
// memory (L2 / Infinity-Cache resident: it was fetched by this very workgroup microseconds earlier).
template <typename E>
__global__ void __launch_bounds__(256, 2)
resblock_kernel(const typename E::T *__restrict__ X, const typename E::T *__restrict__ W1, const float *__restrict__ b1,
                const typename E::T *__restrict__ W2, const float *__restrict__ b2, typename E::T *__restrict__ Y,
                const int8_t *__restrict__ need)
{
    if (need && !need[blockIdx.x]) return;      // workgroup-uniform, before any barrier
    typedef typename E::T T;
    typedef typename E::v8 v8;
    typedef typename E::v4 v4;
    constexpr int MT = 4, NT = 256;
    constexpr int PS = 512 + HIVE_CONV_PAD;
    constexpr unsigned ZOFF = 144 * PS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[145 * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const long long b = blockIdx.x;
    {
        // all 18 sixteen-byte loads of a thread are in flight together (as a loop the compiler waited for each one before
        // issuing the next: 18 dependent L2 / HBM round trips per workgroup before its first MFMA)
        const uint4 *src = reinterpret_cast<const uint4 *>(X + b * 144 * 256);
        uint4 stage[144 * 32 / NT];
#pragma unroll
        for (int j = 0; j < 144 * 32 / NT; ++j) stage[j] = src[tid + j * NT];
#pragma unroll
        for (int j = 0; j < 144 * 32 / NT; ++j) {
            const int i = tid + j * NT;
            *reinterpret_cast<uint4 *>(lds + (i >> 5) * PS + (i & 31) * 16) = stage[j];
        }
        for (int i = tid; i < PS / 16; i += NT) *reinterpret_cast<uint4 *>(lds + ZOFF + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    f32x4 acc[MT][9];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned wlane = (unsigned)(((wave * MT) * 64 + lane) * 16);      // this lane's byte offset inside a k-step's fragments
    v8 A[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        A[0][mt] = *reinterpret_cast<const v8 *>(reinterpret_cast<const char *>(W1) + wlane + mt * 1024);
    __syncthreads();
    conv_taps<E, 9, PS, HIVE_CONV_BDEPTH, 1, true>(lds, ZOFF, 0, W1, W2, wlane, lr, lg, acc, A);
    __syncthreads();                      // every wave has finished reading x from LDS

    // conv1 epilogue: relu(acc + b1) -> 16 bits, written over the board in LDS
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ch0 = (wave * MT + mt) * 16 + lg * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(b1 + ch0);
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int pixel = nt * 16 + lr;
            v4 out = {(T)fmaxf(acc[mt][nt][0] + bv.x, 0.f), (T)fmaxf(acc[mt][nt][1] + bv.y, 0.f),
                      (T)fmaxf(acc[mt][nt][2] + bv.z, 0.f), (T)fmaxf(acc[mt][nt][3] + bv.w, 0.f)};
            *reinterpret_cast<v4 *>(lds + pixel * PS + ch0 * 2) = out;
            acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();                      // the intermediate board is complete
    conv_taps<E, 9, PS, HIVE_CONV_BDEPTH, 1, true>(lds, ZOFF, 0, W2, nullptr, wlane, lr, lg, acc, A);

    // conv2 epilogue: + b2 + x (skip), relu, one rounding, channels-last store
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ch0 = (wave * MT + mt) * 16 + lg * 4;
        const float4 bv = *reinterpret_cast<const float4 *>(b2 + ch0);
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) {
            const int pixel = nt * 16 + lr;
            const size_t o = ((size_t)b * 144 + pixel) * 256 + ch0;
            v4 r = *reinterpret_cast<const v4 *>(X + o);
            v4 out = {(T)fmaxf(acc[mt][nt][0] + bv.x + (float)r[0], 0.f),
                      (T)fmaxf(acc[mt][nt][1] + bv.y + (float)r[1], 0.f),
                      (T)fmaxf(acc[mt][nt][2] + bv.z + (float)r[2], 0.f),
                      (T)fmaxf(acc[mt][nt][3] + bv.w + (float)r[3], 0.f)};
            *reinterpret_cast<v4 *>(Y + o) = out;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The whole residual tower in one launch (alpha_net.py:87-99, the loop over res_0 .. res_18).
//   X, Y   [batch][144][256] channels-last;  W [2*nblocks] fragment-major convolutions back to back;  bias [2*nblocks][256]
// A workgroup owns NB consecutive boards for the whole tower.  Per block: conv1 reads the boards from LDS, its result
// (relu(. + b1), rounded once) overwrites them in place; conv2 reads that, and its epilogue adds the skip operand --
// the block's input, which this very lane stored to Y (or, for the first block, which sits in X) -- applies ReLU,
// rounds once, stores the block's output to Y (the next block's skip operand, and the tower's result) and writes it
// into the LDS image for the next block.  Arithmetic and rounding points are those of resblock_kernel / conv3x3_kernel,
// so the tower is bit-identical to the launch-per-block form.
template <typename E, int NB, int NG, int WPS, int D, int AD>
__global__ void __launch_bounds__(256 * NG, WPS)
tower_kernel(const typename E::T *__restrict__ X, const typename E::T *__restrict__ W, const float *__restrict__ bias,
             typename E::T *__restrict__ Y, int batch, int nblocks)
{
    // NG wave groups of 4 waves each split the NB boards' pixel tiles between them (NG = 2: 8 waves, wave w and w + 4
    // own the same 64 output channels for board 0 / board 1 and fetch the same weight fragments at about the same time)
    typedef typename E::T T;
    typedef typename E::v8 v8;
    typedef typename E::v4 v4;
    constexpr int MT = 4, NT = 256 * NG, NTL = 9 * NB / NG;
    constexpr int PS = 512 + (NB == 1 ? HIVE_CONV_PAD : HIVE_TOWER_PAD);
    constexpr unsigned ZOFF = 144 * NB * PS;
    constexpr size_t CONVSZ = (size_t)9 * 8 * 16 * 512;          // elements of one packed 256 -> 256 convolution
    __shared__ __attribute__((aligned(16))) unsigned char lds[(144 * NB + 1) * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;
    const int lr = lane & 15, lg = lane >> 4;
    const int slot0 = grp * (NB / NG);                                  // first board of this wave's pixel tiles
    const long long b0 = (long long)blockIdx.x * NB;
    const int nvalid = (batch - b0) < NB ? (int)(batch - b0) : NB;      // boards of this workgroup that exist
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(X + b0 * 144 * 256);
        for (int i = tid; i < NB * 144 * 32; i += NT) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (i < nvalid * 144 * 32) v = src[i];
            *reinterpret_cast<uint4 *>(lds + (i >> 5) * PS + (i & 31) * 16) = v;
        }
        for (int i = tid; i < PS / 16; i += NT) *reinterpret_cast<uint4 *>(lds + ZOFF + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    f32x4 acc[MT][NTL];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned wlane = (unsigned)(((wave * MT) * 64 + lane) * 16);      // this lane's byte offset inside a k-step's fragments
    v8 A[2 * AD][MT];
#pragma unroll
    for (int j = 0; j < AD; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            A[j][mt] = *reinterpret_cast<const v8 *>(reinterpret_cast<const char *>(W + (size_t)j * (16 * 512)) + wlane + mt * 1024);
    __syncthreads();

#pragma unroll 1
    for (int blk = 0; blk < nblocks; ++blk) {
        const T *w1 = W + (size_t)(2 * blk) * CONVSZ, *w2 = w1 + CONVSZ;
        const float *b1 = bias + (size_t)(2 * blk) * 256, *b2 = b1 + 256;
        const bool last = blk + 1 == nblocks;
        conv_taps<E, NTL, PS, D, AD, WPS == 1>(lds, ZOFF, slot0, w1, w2, wlane, lr, lg, acc, A);
        __syncthreads();                      // every wave has finished reading the block's input from LDS
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch0 = (wave * MT + mt) * 16 + lg * 4;
            const float4 bv = *reinterpret_cast<const float4 *>(b1 + ch0);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int g = (slot0 * 9 + nt) * 16 + lr;       // pixel index over the NB boards (board-major, as the LDS image)
                v4 out = {(T)fmaxf(acc[mt][nt][0] + bv.x, 0.f), (T)fmaxf(acc[mt][nt][1] + bv.y, 0.f),
                          (T)fmaxf(acc[mt][nt][2] + bv.z, 0.f), (T)fmaxf(acc[mt][nt][3] + bv.w, 0.f)};
                *reinterpret_cast<v4 *>(lds + g * PS + ch0 * 2) = out;
                acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();                      // the intermediate boards are complete
        conv_taps<E, NTL, PS, D, AD, WPS == 1>(lds, ZOFF, slot0, w2, last ? (const T *)nullptr : w2 + CONVSZ, wlane, lr, lg, acc, A);
        if (!last) __syncthreads();           // every wave has finished reading the intermediate boards
        const T *S = blk == 0 ? X : Y;        // the block's input: what this lane stored one block ago
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch0 = (wave * MT + mt) * 16 + lg * 4;
            const float4 bv = *reinterpret_cast<const float4 *>(b2 + ch0);
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int g = (slot0 * 9 + nt) * 16 + lr;
                if (slot0 + nt / 9 < nvalid) {    // (wave-uniform: a pixel tile never straddles two boards)
                    const size_t o = ((size_t)b0 * 144 + g) * 256 + ch0;
                    v4 r = *reinterpret_cast<const v4 *>(S + o);
                    v4 out = {(T)fmaxf(acc[mt][nt][0] + bv.x + (float)r[0], 0.f),
                              (T)fmaxf(acc[mt][nt][1] + bv.y + (float)r[1], 0.f),
                              (T)fmaxf(acc[mt][nt][2] + bv.z + (float)r[2], 0.f),
                              (T)fmaxf(acc[mt][nt][3] + bv.w + (float)r[3], 0.f)};
                    *reinterpret_cast<v4 *>(Y + o) = out;
                    if (!last) *reinterpret_cast<v4 *>(lds + g * PS + ch0 * 2) = out;
                }
                acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (!last) __syncthreads();           // the next block's input is complete
    }
}


// rows of equal leaves (hive_leaf_dedup_launch): row i takes the bytes of row rep[i] (a representative, rep[r] == r, is
// never written, so the copy is safe in place)
__global__ void __launch_bounds__(256)
copy_rows_kernel(uint4 *__restrict__ y, const int32_t *__restrict__ rep, int row16)
{
    const int i = blockIdx.x, r = rep[i];
    if (r == i) return;
    const uint4 *src = y + (size_t)r * row16;
    uint4 *dst = y + (size_t)i * row16;
    for (int k = threadIdx.x; k < row16; k += 256) dst[k] = src[k];
}

template <typename E>
static int launch_conv(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y, int batch,
                       int relu, const int8_t *need, hipStream_t s)
{
    typedef typename E::T T;
    const T *X = (const T *)x, *Wt = (const T *)w, *R = (const T *)residual;
    T *Y = (T *)y;
    constexpr int NWV = HIVE_CONV_WAVES;
    dim3 grid((unsigned)batch), block(NWV * 64);
    if (cin == 256) {
        if (R) hipLaunchKernelGGL((conv3x3_kernel<E, 256, true, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu, need);
        else hipLaunchKernelGGL((conv3x3_kernel<E, 256, false, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu, need);
    } else {
        if (R) hipLaunchKernelGGL((conv3x3_kernel<E, 64, true, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu, need);
        else hipLaunchKernelGGL((conv3x3_kernel<E, 64, false, NWV>), grid, block, 0, s, X, cin, Wt, bias, R, Y, relu, need);
    }
    return 0;
}

}  // namespace hive

using namespace hive;

static bool dtype_ok(int dtype) { return dtype == HIVE_BF16 || dtype == HIVE_F16; }

extern "C" int hive_nn_conv3x3_sel(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                                   int batch, int relu, int dtype, const int8_t *need, void *stream)
{
    if (!x || !w || !bias || !y || batch <= 0) return set_error(HIVE_E_ARG, "hive_nn_conv3x3: bad argument");
    if (cin != 256 && cin != 56) return set_error(HIVE_E_ARG, "hive_nn_conv3x3: cin must be 56 or 256");
    if (!dtype_ok(dtype)) return set_error(HIVE_E_ARG, "hive_nn_conv3x3: dtype must be HIVE_BF16 or HIVE_F16");
    if (dtype == HIVE_BF16) launch_conv<Bf16>(x, cin, w, bias, residual, y, batch, relu, need, (hipStream_t)stream);
    else launch_conv<F16>(x, cin, w, bias, residual, y, batch, relu, need, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_conv3x3: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" int hive_nn_conv3x3_dt(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                                  int batch, int relu, int dtype, void *stream)
{
    return hive_nn_conv3x3_sel(x, cin, w, bias, residual, y, batch, relu, dtype, nullptr, stream);
}

extern "C" int hive_nn_conv3x3(const void *x, int cin, const void *w, const float *bias, const void *residual, void *y,
                               int batch, int relu, void *stream)
{
    return hive_nn_conv3x3_dt(x, cin, w, bias, residual, y, batch, relu, HIVE_BF16, stream);
}

extern "C" int hive_nn_resblock_sel(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                                    int batch, int dtype, const int8_t *need, void *stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || batch <= 0 || x == y)
        return set_error(HIVE_E_ARG, "hive_nn_resblock: bad argument (y must not alias x)");
    if (!dtype_ok(dtype)) return set_error(HIVE_E_ARG, "hive_nn_resblock: dtype must be HIVE_BF16 or HIVE_F16");
    if (dtype == HIVE_BF16)
        hipLaunchKernelGGL(resblock_kernel<Bf16>, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                           (const __bf16 *)w1, b1, (const __bf16 *)w2, b2, (__bf16 *)y, need);
    else
        hipLaunchKernelGGL(resblock_kernel<F16>, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x,
                           (const _Float16 *)w1, b1, (const _Float16 *)w2, b2, (_Float16 *)y, need);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_resblock: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" int hive_nn_resblock_dt(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                                   int batch, int dtype, void *stream)
{
    return hive_nn_resblock_sel(x, w1, b1, w2, b2, y, batch, dtype, nullptr, stream);
}

extern "C" int hive_nn_copy_rows(void *y, const int32_t *rep, int batch, long long row_bytes, void *stream)
{
    if (!y || !rep || batch <= 0 || row_bytes <= 0 || (row_bytes & 15) || row_bytes > (1ll << 30) || ((uintptr_t)y & 15))
        return set_error(HIVE_E_ARG, "hive_nn_copy_rows: bad argument (rows of a multiple of 16 bytes, 16-byte aligned)");
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (uint4 *)y, rep,
                       (int)(row_bytes / 16));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_copy_rows: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" int hive_nn_resblock(const void *x, const void *w1, const float *b1, const void *w2, const float *b2, void *y,
                                int batch, void *stream)
{
    return hive_nn_resblock_dt(x, w1, b1, w2, b2, y, batch, HIVE_BF16, stream);
}

template <typename E>
static void launch_tower(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int mode, hipStream_t s)
{
    typedef typename E::T T;
    const T *X = (const T *)x, *Wt = (const T *)w;
    T *Y = (T *)y;
    if (mode == 1) hipLaunchKernelGGL((tower_kernel<E, 1, 1, 2, HIVE_CONV_BDEPTH, 1>), dim3((unsigned)batch), dim3(256), 0, s, X, Wt, bias, Y, batch, nblocks);
    else if (mode == 2) hipLaunchKernelGGL((tower_kernel<E, 2, 2, 2, HIVE_CONV_BDEPTH, 1>), dim3((unsigned)((batch + 1) / 2)), dim3(512), 0, s, X, Wt, bias, Y, batch, nblocks);
    else hipLaunchKernelGGL((tower_kernel<E, 1, 1, 1, HIVE_TOWER_BDEPTH, HIVE_TOWER_ADIST>), dim3((unsigned)batch), dim3(256), 0, s, X, Wt, bias, Y, batch, nblocks);
}

// ---------------------------------------------------------------------------------------------
// hive_tower72_{bf16,f16}: the tower with a 72-tile wave, written in assembly (gen_tower_asm.py -> tower72_gfx950.hsaco,
// embedded here): two boards per workgroup, one wave per SIMD, 288 accumulator registers per lane.
#if !defined(__HIP_DEVICE_COMPILE__)
asm(".section .rodata\n"
    ".p2align 12\n"
    ".global hive_tower72_hsaco\n"
    "hive_tower72_hsaco:\n"
    ".incbin \"tower72_gfx950.hsaco\"\n"
    ".global hive_tower72_hsaco_end\n"
    "hive_tower72_hsaco_end:\n"
    ".byte 0\n"
    ".text\n");
#endif
extern "C" const unsigned char hive_tower72_hsaco[];

namespace {
struct Tower72Args {
    const void *x, *w;
    const float *bias;
    void *y;
    const int32_t *rows, *nrows;
    int32_t batch, nblocks;
    const int32_t *plan;
    int32_t flags, pad;
    const void *residual;
};
static_assert(sizeof(Tower72Args) == 80, "kernarg layout of hive_tower72_* (gen_tower_asm.py)");
constexpr int kPlanMaxWg = 256, kPlanStride = 8;      // MAX_WG, PLAN_STRIDE / 4 of gen_tower_asm.py

// The launch plan of a balanced tower: `grid` workgroups (one per CU, all resident) share pairs * nblocks block-steps evenly.
// The pairs are laid end to end on a line of block-steps, workgroup c owns [c T, (c + 1) T) with T = ceil(steps / grid): a pair
// cut by a boundary has its FIRST blocks (the later workgroup's share ... of the line, but run first thing there: "head") and
// its last blocks (the earlier workgroup's share, run last there: "tail") on two workgroups; T >= nblocks guarantees the
// head has finished long before the tail starts.  Entry = {head pair, head end block, first full pair, end of the full
// pairs, tail pair, tail first block, 0, 0}; the hand-over flags (one int per pair) follow the kPlanMaxWg entries.
__global__ void __launch_bounds__(256)
tower72_plan_kernel(const int32_t *__restrict__ nrows, int batch, int nblocks, int grid, int32_t *__restrict__ plan, int maxpairs)
{
    const int c = threadIdx.x;
    const int n = nrows ? *nrows : batch, pairs = (n + 1) / 2;
    int32_t *flags = plan + kPlanMaxWg * kPlanStride;
    for (int i = c; i < maxpairs; i += 256) flags[i] = 0;
    int e[8] = {-1, 0, 0, 0, -1, 0, 0, 0};
    if (c < grid) {
        if (pairs <= grid) {
            if (c < pairs) { e[2] = c; e[3] = c + 1; }
        } else {
            const long long steps = (long long)pairs * nblocks, T = (steps + grid - 1) / grid;
            const long long v0 = c * T, v1 = v0 + T < steps ? v0 + T : steps;
            if (v0 < v1) {
                const int p0 = (int)(v0 / nblocks), o0 = (int)(v0 % nblocks), p1 = (int)(v1 / nblocks), o1 = (int)(v1 % nblocks);
                e[2] = o0 ? p0 + 1 : p0;
                e[3] = p1;
                if (o0) { e[4] = p0; e[5] = o0; }
                if (o1) { e[0] = p1; e[1] = o1; }
            }
        }
    }
    for (int k = 0; k < 8; ++k) plan[c * kPlanStride + k] = e[k];
}

struct Tower72Module {
    hipModule_t mod = nullptr;
    hipFunction_t fn[2] = {nullptr, nullptr};
};

// one module per device (a code object is loaded into the current device's context)
int tower72_function(int dtype, hipFunction_t *out)
{
    static std::mutex mu;
    static Tower72Module mods[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hive::set_error(HIVE_E_DEVICE, "hive_nn_tower72: no current device");
    std::lock_guard<std::mutex> lock(mu);
    Tower72Module &m = mods[dev];
    if (!m.mod) {
        // (HIVE_TOWER72_HSACO: a development switch -- load another build of the generated kernel, e.g. a debug or A/B variant)
        const char *alt = getenv("HIVE_TOWER72_HSACO");
        hipError_t e = alt && *alt ? hipModuleLoad(&m.mod, alt) : hipModuleLoadData(&m.mod, hive_tower72_hsaco);
        if (e == hipSuccess) e = hipModuleGetFunction(&m.fn[0], m.mod, "hive_tower72_bf16");
        if (e == hipSuccess) e = hipModuleGetFunction(&m.fn[1], m.mod, "hive_tower72_f16");
        if (e != hipSuccess) {
            m.mod = nullptr;
            return hive::set_error(HIVE_E_DEVICE, std::string("hive_nn_tower72: loading the code object: ") + hipGetErrorString(e));
        }
    }
    *out = m.fn[dtype == HIVE_BF16 ? 0 : 1];
    return HIVE_OK;
}

// rows[k] = index of the k-th board flagged in `need` (ascending), *nrows = how many: what hive_tower72_* iterates over
__global__ void __launch_bounds__(1024)
compact_rows_kernel(const int8_t *__restrict__ need, int batch, int32_t *__restrict__ rows, int32_t *__restrict__ nrows)
{
    __shared__ int wave_count[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int lo = 0; lo < batch; lo += 1024) {
        const int i = lo + tid;
        const bool on = i < batch && need[i] != 0;
        const unsigned long long bal = __ballot(on);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_count[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_count[w];
        if (on) rows[off + before] = i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wave_count[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *nrows = base;
}
}  // namespace

extern "C" int hive_nn_compact_rows(const int8_t *need, int batch, int32_t *rows, int32_t *nrows, void *stream)
{
    if (!need || !rows || !nrows || batch <= 0) return set_error(HIVE_E_ARG, "hive_nn_compact_rows: bad argument");
    hipLaunchKernelGGL(compact_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, need, batch, rows, nrows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_compact_rows: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" long long hive_nn_tower72_plan_bytes(int batch)
{
    return (long long)(kPlanMaxWg * kPlanStride + (batch + 1) / 2 + 64) * 4;
}

static int tower72_launch(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                          const int32_t *rows, const int32_t *nrows, int32_t *plan, void *stream, int flags = 0,
                          const void *residual = nullptr);

extern "C" int hive_nn_tower72(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                               const int32_t *rows, const int32_t *nrows, void *stream)
{
    return tower72_launch(x, w, bias, y, batch, nblocks, dtype, rows, nrows, nullptr, stream);
}

extern "C" int hive_nn_tower72_balanced(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                                        const int32_t *rows, const int32_t *nrows, void *plan_workspace, void *stream)
{
    if (!plan_workspace || ((uintptr_t)plan_workspace & 31))
        return set_error(HIVE_E_ARG, "hive_nn_tower72_balanced: plan_workspace (hive_nn_tower72_plan_bytes, 32-byte aligned) is missing");
    return tower72_launch(x, w, bias, y, batch, nblocks, dtype, rows, nrows, (int32_t *)plan_workspace, stream);
}

extern "C" int hive_nn_conv72(const void *x, const void *w, const float *bias, void *y, int batch, int relu, int dtype, void *stream)
{
    // one 256 -> 256 convolution (+ bias, optional ReLU) on the 72-tile kernel: flags bit 0 = one convolution per block,
    // bit 1 = no ReLU
    return tower72_launch(x, w, bias, y, batch, 1, dtype, nullptr, nullptr, nullptr, stream, 1 | (relu ? 0 : 2));
}

extern "C" int hive_nn_conv72_add(const void *x, const void *w, const float *bias, const void *residual, void *y, int batch, int relu,
                                  int dtype, void *stream)
{
    // ... + residual (flags bit 2): the convolution runs as the kernel's SECOND convolution of a block, whose epilogue adds a skip
    // operand in fp32 before the one rounding -- here the rows of `residual` (it may be y itself: a workgroup has read its two
    // rows of the residual before it stores them)
    if (!residual) return set_error(HIVE_E_ARG, "hive_nn_conv72_add: residual == NULL");
    return tower72_launch(x, w, bias, y, batch, 1, dtype, nullptr, nullptr, nullptr, stream, 1 | (relu ? 0 : 2) | 4, residual);
}

extern "C" int hive_nn_conv72_stats(const void *x, const void *w, const float *bias, void *y, int batch, int relu, int dtype,
                                    float *partial, void *stream)
{
    // ... and (flags bit 3) every workgroup's per-channel sum / sum of squares of what it stored: partial[batch / 2][2][256]
    if (!partial || batch % 2 != 0)
        return set_error(HIVE_E_ARG, "hive_nn_conv72_stats: partial == NULL or an odd batch (a repeated tail board would be counted twice)");
    return tower72_launch(x, w, bias, y, batch, 1, dtype, nullptr, nullptr, nullptr, stream, 1 | (relu ? 0 : 2) | 8, partial);
}

static int tower72_launch(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                          const int32_t *rows, const int32_t *nrows, int32_t *plan, void *stream, int flags, const void *residual)
{
    if (!x || !w || !bias || !y || batch <= 0 || nblocks <= 0 || x == y)
        return set_error(HIVE_E_ARG, "hive_nn_tower72: bad argument (y must not alias x)");
    if (!dtype_ok(dtype)) return set_error(HIVE_E_ARG, "hive_nn_tower72: dtype must be HIVE_BF16 or HIVE_F16");
    if ((rows == nullptr) != (nrows == nullptr))
        return set_error(HIVE_E_ARG, "hive_nn_tower72: rows and nrows come together (hive_nn_compact_rows) or not at all");
    hipFunction_t fn = nullptr;
    int rc = tower72_function(dtype, &fn);
    if (rc != HIVE_OK) return rc;
    unsigned grid = (unsigned)((batch + 1) / 2);
    if (plan) {
        // one workgroup per CU, all resident at once (157 KB of LDS each): the grid is the chip, the plan deals the work
        static int cus[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && cus[dev] == 0) {
            hipDeviceProp_t prop;
            cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : kPlanMaxWg;
        }
        const int chip = dev >= 0 && dev < 64 && cus[dev] > 0 ? (cus[dev] < kPlanMaxWg ? cus[dev] : kPlanMaxWg) : kPlanMaxWg;
        if ((int)grid > chip) grid = (unsigned)chip;
        hipLaunchKernelGGL(tower72_plan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nrows, batch, nblocks, (int)grid, plan,
                           (batch + 1) / 2);
    }
    Tower72Args args{x, w, bias, y, rows, nrows, batch, nblocks, plan, flags, 0, residual};
    size_t size = sizeof(args);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    hipError_t e = hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, nullptr, config);
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_tower72: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" int hive_nn_tower(const void *x, const void *w, const float *bias, void *y, int batch, int nblocks, int dtype,
                             int boards_per_group, void *stream)
{
    if (!x || !w || !bias || !y || batch <= 0 || nblocks <= 0 || x == y)
        return set_error(HIVE_E_ARG, "hive_nn_tower: bad argument (y must not alias x)");
    if (!dtype_ok(dtype)) return set_error(HIVE_E_ARG, "hive_nn_tower: dtype must be HIVE_BF16 or HIVE_F16");
    if (boards_per_group < 0 || boards_per_group > 3)
        return set_error(HIVE_E_ARG, "hive_nn_tower: boards_per_group must be 0 (choose), 1, 2 or 3 (one wave per SIMD)");
    // two boards per workgroup halve the weight stream but leave one workgroup per CU: worth it once the launch has more
    // boards than the chip has workgroup slots of the one-board form (2 x 256)
    const int mode = boards_per_group ? boards_per_group : (batch > 512 ? 2 : 1);
    if (dtype == HIVE_BF16) launch_tower<Bf16>(x, w, bias, y, batch, nblocks, mode, (hipStream_t)stream);
    else launch_tower<F16>(x, w, bias, y, batch, nblocks, mode, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_tower: ") + hipGetErrorString(e));
    return HIVE_OK;
}
