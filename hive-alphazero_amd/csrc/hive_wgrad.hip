// hive_wgrad.hip -- weight gradient of the 3x3 convolutions of the 256-channel tower on the CDNA4 matrix cores
// (training step of alpha_zero/alpha_net.py:117-162; what loss.backward() computes for ResBlock.conv1/conv2.weight).
//
//   dW[tap][k][c] = sum over boards b and pixels p of  dY[b][p][k] * X[b][p + tap][c]        (zero outside the board)
//
// GEMM view: M = k (256 output channels), N = c (256 input channels) for each of the 9 taps, contraction over the
// batch * 144 pixels.  Both operands are stored channels-last, i.e. with the CONTRACTION index as the slow one, so both
// MFMA operands are read from LDS with gfx950's transposing read ds_read_b64_tr_b16 (4 pixel rows x 16 channels per
// 16-lane group, delivered channel-major) -- no transposed copy of the activations is ever made.
//
// Decomposition: a workgroup (8 waves) owns the output slice [9 taps][128 k][64 c] and a share of the boards
// (split-K); a wave owns [9 taps][64 k][16 c] = 36 accumulator tiles (144 VGPRs).  Per board the slice's dY
// (144 x 128) and X (144 x 64, inside a zero halo so that a tap is a constant address offset) are staged in LDS, double
// buffered: the next board's LDS-DMA loads (global_load_lds_dwordx4) are issued before the MFMA loop into the other buffer.
// Per 32-pixel k-step a wave issues 8 + 18 transposed reads for 36 MFMAs (4 x 1 tiles per tap; 4 + 36 in the 2 x 2 form).
//   dY image: rows of 256 B, the 32-byte channel block XOR-swizzled with (row & 7); X image: 14 x 20 halo grid, rows of
//   160 B (128 + 32 pad): any 8 rows that are distinct mod 8 -- which 8 consecutive pixels, shifted by any tap, are on
//   a 20-wide grid -- hit all 64 banks once: every transposed read is conflict free.
// Split-K reduction: the 2048 waves hold 75 MB of fp32 partial sums (the output itself is 2.4 MB).  fp32 atomics on
// dW cost 52 of 140 us (18.9 M atomic dwords; per-XCD private buffers did not help: the atomic units, not cross-XCD
// traffic, are the limit), so every wave writes its accumulators as they sit in its registers -- one float4 per lane,
// 1 KiB per store instruction -- and a second kernel sums the board ranges and scatters the 2.4 MB into dW's layout.
// The XCD-aware workgroup order keeps the eight slices of one board range on one XCD so that the staged activations are
// shared through its L2.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hive_abi.h"
#include "../../include/hive_nn.h"

namespace hive {
int set_error(int code, const std::string &msg);

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgKS = 128, kWgCS = 64;            // output slice of a workgroup
constexpr int kWgDyRow = 256;                     // bytes per dY image row (128 channels)
constexpr int kWgDyRows = 145;                    // 144 pixels + one zero row (pixels 144..159 of the last k-step)
constexpr int kWgXRow = 160;                      // bytes per X image row (64 channels + 32 pad)
constexpr int kWgXW = 20;                         // halo grid width (12 + 8: row-to-row jump of 8 keeps rows distinct mod 8)
constexpr int kWgXRows = 13 * kWgXW + 14;         // last row any tap can touch: (13, 13)
constexpr int kWgDyBytes = kWgDyRows * kWgDyRow;  // 37,120
constexpr int kWgXBytes = kWgXRows * kWgXRow;     // 43,840
constexpr int kWgBuf = kWgDyBytes + kWgXBytes;    // 80,960 per buffer, 161,920 for both (160 KiB = 163,840)
constexpr int kWgThreads = 512;
#ifndef HIVE_WG_MT
#define HIVE_WG_MT 4
#endif
// A wave's share of the slice: kWM x kWN accumulator tiles per tap (k x c).  4 x 1: per 32-pixel k-step 8 transposing reads of
// dY fragments (each feeds 9 MFMAs) and 18 of X fragments (each feeds 4) = 26 reads for 36 MFMAs; the 2 x 2 form of rounds 2-3
// read 4 + 36 = 40 (an X fragment fed only 2 MFMAs) and was bound by those reads (LDS at its rate before the matrix pipes were).
constexpr int kWM = HIVE_WG_MT, kWN = 4 / kWM;
constexpr int kWavesC = kWgCS / (16 * kWN);       // waves side by side along c (4 for 4 x 1, 2 for 2 x 2); the rest along k
static_assert(kWM * kWN == 4 && (kWgKS / (16 * kWM)) * kWavesC == 8, "eight waves tile the [128 k][64 c] slice");
constexpr int kWgOut = 9 * 256 * 256;             // entries of dW

__device__ __forceinline__ ws16x4 lds_tr(const unsigned char *lds, unsigned off)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) ws16x4 *)(__attribute__((address_space(3))) void *)(lds + off));
}
__device__ __forceinline__ wbf16x8 frag(ws16x4 lo, ws16x4 hi)
{
    ws16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(wbf16x8, v);
}
__device__ __forceinline__ unsigned x_row_of_pixel(unsigned p) { return (p / 12u + 1u) * kWgXW + (p % 12u) + 1u; }

__global__ void __launch_bounds__(kWgThreads, 2)
conv3x3_wgrad_kernel(const __bf16 *__restrict__ X, const __bf16 *__restrict__ DY, float *__restrict__ WS, int batch,
                     int splits, int xcd_order)
{
    // two DISTINCT LDS objects: the compiler then knows that the LDS-DMA into one never aliases the transposed reads of
    // the other and does not park a vmcnt(0) in front of every read
    __shared__ __attribute__((aligned(16))) unsigned char lds0[kWgBuf];
    __shared__ __attribute__((aligned(16))) unsigned char lds1[kWgBuf];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / kWavesC, wc = wave % kWavesC;
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, p4 = li & 3;

    // workgroup -> (split, slice): with xcd_order the eight slices of a split are consecutive on ONE XCD
    // (workgroups are dealt to the 8 XCDs round robin), so they share the staged activations through its L2
    int split, slice;
    {
        const int id = blockIdx.x;
        if (xcd_order) { const int xcd = id & 7, t = id >> 3; slice = t & 7; split = (t >> 3) * 8 + xcd; }
        else { split = id >> 3; slice = id & 7; }
    }
    const int ks = slice >> 2, cs = slice & 3;                // 2 k-slices x 4 c-slices
    const int per = (batch + splits - 1) / splits;
    const int b0 = split * per, b1 = (b0 + per < batch) ? b0 + per : batch;

    // ---- zero both buffers once: halo, pad and the zero row stay zero, staging only writes the interior
    for (int i = tid; i < kWgBuf / 16; i += kWgThreads) {
        reinterpret_cast<uint4 *>(lds0)[i] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4 *>(lds1)[i] = make_uint4(0u, 0u, 0u, 0u);
    }

    // Staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land in 1 KiB of consecutive LDS, no VGPRs, fully
    // asynchronous).  The images are cut into 1-KiB spans: 36 of dY (four 256-byte rows each; the XOR swizzle is applied
    // on the SOURCE side -- lane l fetches the channel chunk that belongs in its slot) and 37 over the interior of the X
    // halo grid (lanes that fall on a halo cell or on the 32 pad bytes of a row are masked off, so those stay zero).
    // Wave w issues spans w, w + 8, ...
    constexpr int kDySpans = 36, kSpans = 73, kXFirst = (kWgXW + 1) * kWgXRow;     // byte offset of pixel 0's row in the X image
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // element offset inside one board of the 16 bytes this lane moves in span sp; bit 31: from X; 0xFFFFFFFF: nothing
    auto span_src = [&](int sp) -> unsigned {
        unsigned v = 0xFFFFFFFFu;
        if (sp < kDySpans) {
            const unsigned r = 4u * sp + ((unsigned)lane >> 4), pos = (unsigned)lane & 15u;
            const unsigned ch = (((pos >> 1) ^ (r & 7u)) << 1) | (pos & 1u);
            v = r * 256u + (unsigned)(ks * kWgKS) + ch * 8u;
        } else if (sp < kSpans) {
            const unsigned o = (unsigned)kXFirst + 1024u * (unsigned)(sp - kDySpans) + 16u * (unsigned)lane;
            const unsigned rho = o / kWgXRow, u = (o % kWgXRow) >> 4;
            const unsigned row = rho / kWgXW, col = rho % kWgXW;
            if (u < 8u && row >= 1u && row <= 12u && col >= 1u && col <= 12u)
                v = 0x80000000u | (((row - 1u) * 12u + (col - 1u)) * 256u + (unsigned)(cs * kWgCS) + u * 8u);
        }
        return v;
    };
    // span i (0..9) of this wave for board b into `image`
    auto stage_span = [&](int b, unsigned char *image, int i) {
        const int sp = wave_u + 8 * i;
        if (sp < kSpans) {
            const unsigned off = span_src(sp);
            unsigned char *dst = image + (sp < kDySpans ? sp * 1024 : kWgDyBytes + kXFirst + (sp - kDySpans) * 1024);
            if (off != 0xFFFFFFFFu) {
                const __bf16 *g = ((off >> 31) ? X : DY) + (long long)b * (144 * 256) + (off & 0x7FFFFFFFu);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
            }
        }
    };
    auto stage_board = [&](int b, unsigned char *image) {
#pragma unroll
        for (int i = 0; i < 10; ++i) stage_span(b, image, i);
    };

    wf32x4 acc[9][kWM][kWN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < kWM; ++a)
#pragma unroll
            for (int c = 0; c < kWN; ++c) acc[t][a][c] = wf32x4{0.f, 0.f, 0.f, 0.f};

    // the MFMA work on one staged board
    // ... during which the spans of the NEXT board (nb >= 0) are sent off a few per k-step: ten LDS-DMA instructions issued
    // back to back at the board boundary would stall both waves of a SIMD at the same moment
    auto board_mfma = [&](const unsigned char *dyi, int nb, unsigned char *nimage) {
        const unsigned char *xi = dyi + kWgDyBytes;
#pragma unroll 1
        for (int s = 0; s < 5; ++s) {
#ifndef HIVE_WG_ABL_NOSTAGE
            if (nb >= 0) {
#ifdef HIVE_WG_EARLY_DMA
                // (experiment: three spans per k-step in the first three k-steps, none in the last -- measured 3 % slower)
                if (s < 4) {
                    stage_span(nb, nimage, 3 * s);
                    if (s < 3) {
                        stage_span(nb, nimage, 3 * s + 1);
                        stage_span(nb, nimage, 3 * s + 2);
                    }
                }
#else
                stage_span(nb, nimage, 2 * s);
                stage_span(nb, nimage, 2 * s + 1);
#endif
            }
#endif
            // this lane's pixel rows: k-group grp holds pixels 32 s + 4 grp + {0..3} (first read) and + 16 (second read);
            // lane 4 q + p4 of a 16-lane group addresses row q, channels 4 p4 .. 4 p4 + 3 of the block
            unsigned pa = 32u * s + 4u * grp + q, pb = pa + 16u;
            const unsigned ra = pa < 144u ? pa : 144u, rb = pb < 144u ? pb : 144u;        // the zero row beyond the board
            wbf16x8 A[kWM];
#pragma unroll
            for (int mt = 0; mt < kWM; ++mt) {
                const unsigned mb = (unsigned)(kWM * wk + mt);
                A[mt] = frag(lds_tr(dyi, ra * kWgDyRow + ((mb ^ (ra & 7u)) << 5) + (p4 << 3)),
                             lds_tr(dyi, rb * kWgDyRow + ((mb ^ (rb & 7u)) << 5) + (p4 << 3)));
            }
            // dY is zero beyond pixel 143, so X only has to be finite there: read pixel 0's neighbourhood
            if (pa >= 144u) pa = 0u;
            if (pb >= 144u) pb = 0u;
            const unsigned xa = (x_row_of_pixel(pa) - (kWgXW + 1)) * kWgXRow + (unsigned)(kWN * wc) * 32u + (p4 << 3);
            const unsigned xb = (x_row_of_pixel(pb) - (kWgXW + 1)) * kWgXRow + (unsigned)(kWN * wc) * 32u + (p4 << 3);
            // B fragments two taps ahead of the MFMAs that consume them (a transposing LDS read takes longer than the two
            // MFMAs the compiler's own schedule left between issue and use); sched_barrier pins the order
            wbf16x8 Bf[3][kWN];
#define WG_LOAD_B(tap_)                                                                                              \
    {                                                                                                               \
        const unsigned toff = (unsigned)((((tap_) / 3) * kWgXW + ((tap_) % 3)) * kWgXRow); /* (dy + 1, dx + 1) rows */ \
        _Pragma("unroll") for (int nt_ = 0; nt_ < kWN; ++nt_)                                                        \
            Bf[(tap_) % 3][nt_] = frag(lds_tr(xi, xa + toff + 32u * nt_), lds_tr(xi, xb + toff + 32u * nt_));         \
    }
            WG_LOAD_B(0)
            WG_LOAD_B(1)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 2 < 9) WG_LOAD_B(tap + 2)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < kWN; ++nt)
#pragma unroll
                    for (int mt = 0; mt < kWM; ++mt)
                        acc[tap][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt], Bf[tap % 3][nt], acc[tap][mt][nt], 0, 0, 0);
            }
#undef WG_LOAD_B
        }
    };

    __syncthreads();                                          // buffers are zero
    if (b0 < b1) stage_board(b0, lds0);
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): this wave's DMA has landed
    __syncthreads();

    for (int b = b0; b < b1; b += 2) {
        board_mfma(lds0, b + 1 < b1 ? b + 1 : -1, lds1);      // nobody reads lds1 during this board
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // the next board has landed (this wave's share)
        __syncthreads();
        if (b + 1 < b1) {
            board_mfma(lds1, b + 2 < b1 ? b + 2 : -1, lds0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
    }

    // ---- partial sums out, register layout: WS[split][slice][wave][tile = (tap, mt, nt)][lane] = float4 (rows 4 grp + j)
    {
        float4 *out = reinterpret_cast<float4 *>(WS) + ((((long long)split * 8 + slice) * 8 + wave) * 36) * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mt = 0; mt < kWM; ++mt)
#pragma unroll
                for (int nt = 0; nt < kWN; ++nt) {
                    const wf32x4 v = acc[tap][mt][nt];
#ifdef HIVE_WG_ABL_NOATOMIC
                    if (v[0] == 12345.678f)
#endif
                    out[((tap * kWM + mt) * kWN + nt) * 64] = make_float4(v[0], v[1], v[2], v[3]);
                }
    }
}

// dW[tap][k][c] = sum over the board ranges of the partial tiles; one thread per (slice, wave, tile, lane) float4
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float *__restrict__ WS, float *__restrict__ DW, int splits, int k_major)
{
    const int i = blockIdx.x * 256 + threadIdx.x;              // = ((slice * 8 + wave) * 36 + tile) * 64 + lane
    if (i >= 8 * 8 * 36 * 64) return;
    const float4 *p = reinterpret_cast<const float4 *>(WS) + i;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < splits; ++sp) {
        const float4 v = p[(long long)sp * (8 * 8 * 36 * 64)];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const int lane = i & 63, tile = (i >> 6) % 36, wave = ((i >> 6) / 36) & 7, slice = (i >> 6) / (36 * 8);
    const int tap = tile >> 2, mt = (tile & 3) / kWN, nt = (tile & 3) % kWN;
    const int ks = slice >> 2, cs = slice & 3, wk = wave / kWavesC, wc = wave % kWavesC, grp = lane >> 4, li = lane & 15;
    const int k0 = ks * kWgKS + 16 * kWM * wk + 16 * mt + 4 * grp, c = cs * kWgCS + 16 * kWN * wc + 16 * nt + li;
    if (k_major) {                     // [k][tap][c]: the memory of a torch.channels_last nn.Conv2d weight
        float *d = DW + ((long long)k0 * 9 + tap) * 256 + c;
        d[0] = a.x; d[9 * 256] = a.y; d[18 * 256] = a.z; d[27 * 256] = a.w;
        return;
    }
    float *d = DW + ((long long)tap * 256 + k0) * 256 + c;
    d[0] = a.x; d[256] = a.y; d[512] = a.z; d[768] = a.w;
}

}  // namespace hive

using namespace hive;

constexpr int kWgMaxSplits = 32;
extern "C" int hive_nn_wgrad_workspace_floats(void) { return kWgMaxSplits * kWgOut; }


#ifdef HIVE_WG_DEBUG
static int g_wg_debug_order = -1;
extern "C" void hive_nn_wgrad_debug_order(int v) { g_wg_debug_order = v; }
#endif

static int wgrad_launch(const void *x, const void *dy, float *dw, int batch, float *workspace, int k_major, void *stream);

extern "C" int hive_nn_conv3x3_wgrad(const void *x, const void *dy, float *dw, int batch, float *workspace, void *stream)
{
    return wgrad_launch(x, dy, dw, batch, workspace, 0, stream);
}

extern "C" int hive_nn_conv3x3_wgrad_layout(const void *x, const void *dy, float *dw, int batch, float *workspace, int layout,
                                            void *stream)
{
    if (layout != 0 && layout != 1) return set_error(HIVE_E_ARG, "hive_nn_conv3x3_wgrad_layout: layout must be 0 or 1");
    return wgrad_launch(x, dy, dw, batch, workspace, layout, stream);
}

static int wgrad_launch(const void *x, const void *dy, float *dw, int batch, float *workspace, int k_major, void *stream)
{
    if (!x || !dy || !dw || !workspace || batch <= 0) return set_error(HIVE_E_ARG, "hive_nn_conv3x3_wgrad: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    // one workgroup per CU and per (slice, board range): 8 slices x `splits` ranges
    int splits = batch < kWgMaxSplits ? batch : kWgMaxSplits;
    int xcd_order = (splits % 8 == 0) ? 1 : 0;
#ifdef HIVE_WG_DEBUG
    if (g_wg_debug_order >= 0) xcd_order = g_wg_debug_order && (splits % 8 == 0);
#endif
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((unsigned)(8 * splits)), dim3(kWgThreads), 0, s, (const __bf16 *)x,
                       (const __bf16 *)dy, workspace, batch, splits, xcd_order);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((kWgOut / 4 + 255) / 256)), dim3(256), 0, s, workspace, dw, splits, k_major);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_conv3x3_wgrad: ") + hipGetErrorString(e));
    return HIVE_OK;
}
