// hive_heads.hip -- the two heads of the leaf evaluator (alpha_net.py:56-80 of the reference: OutBlock) as hand-written
// kernels, so that a forward contains no library GEMM (no per-process tuning, nothing that spins on another workgroup):
//
//   head_conv_kernel   both 1x1 convolutions over the tower's NHWC output (256 -> 128 policy channels + 1 value channel,
//                      BatchNorm folded, ReLU) as one MFMA GEMM per pixel with the 16 BOARDS of a workgroup as the N
//                      dimension: the policy activations leave directly as the A fragments of the policy FC.
//   policy_fc_kernel   logits = p1 [B x 18432] . Wfc^T [18432 x 1584]: 256 boards x 176 actions per workgroup, the action
//                      fragments shared by the four waves through LDS, the board fragments straight from L2, K split over
//                      workgroups into fp32 partial sums (summed in a fixed order: deterministic).
//   head_finish_kernel per board: partial sums + bias -> softmax (fp32), and the value MLP 144 -> 64 -> 1 -> tanh.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hive_abi.h"
#include "../../include/hive_nn.h"

namespace hive {
int set_error(int code, const std::string &msg);

namespace heads {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

constexpr int kPolicyCh = 128, kHeadTiles = 9, kPixels = 144, kFcK = kPixels * kPolicyCh, kFcKsteps = kFcK / 32, kActions = 1584;
constexpr int kFcNT = kActions / 16;               // 99 action tiles
constexpr int kFcNB = 11;                          // action tiles per workgroup (9 workgroups cover the 99)
constexpr int kFcMB = 16;                          // board tiles per workgroup (256 boards), 4 per wave

// ---------------------------------------------------------------------------------------------
// D_p[out channel][board] = sum_c W[out channel][c] * x[board][p][c] for every pixel p of 16 boards.
//   x     [B][144][256]                                  tower output, channels-last
//   w     [9][8][64][8]   A fragments (M tile of 16 output channels, k-step of 32 input channels): rows 0..127 the policy
//                         convolution, row 128 the value convolution, rows 129..143 zero
//   bias  f32 [144]       (BatchNorm folded; entries beyond 128 unused)
//   p1    [ceil(B/16)][576][64][8]   relu(policy conv) as the FC's A fragments: board tile, k-step = pixel * 4 + channel / 32,
//                                   lane = (channel % 32) / 8 * 16 + board % 16, 8 consecutive channels
//   v1    f32 [B][144]    relu(value conv), rounded to the 16-bit type first (as the library path did)
// grid (ceil(B/16), 4): a workgroup = 16 boards x 36 pixels; a wave = 9 of those pixels, three at a time, all 9 M tiles;
// the 72 KiB of weight fragments sit in LDS.
template <typename E>
__global__ void __launch_bounds__(256, 1)
head_conv_kernel(const typename E::T *__restrict__ x, const typename E::T *__restrict__ w, const float *__restrict__ bias,
                 typename E::T *__restrict__ p1, float *__restrict__ v1, int batch)
{
    typedef typename E::T T;
    typedef typename E::v8 v8;
    typedef typename E::v4 v4;
    __shared__ __attribute__((aligned(16))) unsigned char wl[kHeadTiles * 8 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(w);
        uint4 tmp[kHeadTiles * 8 * 64 / 256];
#pragma unroll
        for (int j = 0; j < kHeadTiles * 8 * 64 / 256; ++j) tmp[j] = src[tid + j * 256];
#pragma unroll
        for (int j = 0; j < kHeadTiles * 8 * 64 / 256; ++j) *reinterpret_cast<uint4 *>(wl + (size_t)(tid + j * 256) * 16) = tmp[j];
    }
    const int mtile = blockIdx.x;
    const int board = mtile * 16 + lr;
    const bool valid = board < batch;
    const size_t brow = (size_t)(valid ? board : batch - 1) * kPixels * 256;      // (tail lanes read a real board, store nothing)
    __syncthreads();
    for (int grp = 0; grp < 3; ++grp) {
        const int p0 = blockIdx.y * 36 + wave * 9 + grp * 3;
        const T *xp = x + brow + (size_t)p0 * 256 + lg * 8;
        f32x4 acc[kHeadTiles][3];
#pragma unroll
        for (int mt = 0; mt < kHeadTiles; ++mt)
#pragma unroll
            for (int px = 0; px < 3; ++px) acc[mt][px] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k-step outer, M tile inner: three board fragments live per k-step (the next k-step's are in flight), every weight
        // fragment read from LDS feeds three MFMAs
        v8 X[2][3];
#pragma unroll
        for (int px = 0; px < 3; ++px) X[0][px] = *reinterpret_cast<const v8 *>(xp + px * 256);
        // (the k-step pairs are a real loop: fully unrolled, the compiler hoists all 72 weight-fragment reads and spills)
#pragma unroll 1
        for (int ks2 = 0; ks2 < 8; ks2 += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ks = ks2 + half;
                if (ks + 1 < 8) {
#pragma unroll
                    for (int px = 0; px < 3; ++px)
                        X[half ^ 1][px] = *reinterpret_cast<const v8 *>(xp + px * 256 + (ks + 1) * 32);
                }
#pragma unroll
                for (int mt = 0; mt < kHeadTiles; ++mt) {
                    const v8 a = *reinterpret_cast<const v8 *>(wl + ((size_t)(mt * 8 + ks) * 64 + lane) * 16);
#pragma unroll
                    for (int px = 0; px < 3; ++px) acc[mt][px] = E::mfma(a, X[half][px], acc[mt][px]);
                }
            }
        }
        // lane holds D[4 channels mt*16 + lg*4 ..][board lr] of three pixels
#pragma unroll
        for (int px = 0; px < 3; ++px) {
            const int p = p0 + px;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + mt * 16 + lg * 4);
                v4 out = {(T)fmaxf(acc[mt][px][0] + bv.x, 0.f), (T)fmaxf(acc[mt][px][1] + bv.y, 0.f),
                          (T)fmaxf(acc[mt][px][2] + bv.z, 0.f), (T)fmaxf(acc[mt][px][3] + bv.w, 0.f)};
                if (!valid) out = v4{(T)0.f, (T)0.f, (T)0.f, (T)0.f};
                // channel c = mt*16 + lg*4 + r -> k-step p*4 + mt/2, k-group (mt&1)*2 + lg/2, element (lg&1)*4 + r
                const size_t frag = ((size_t)mtile * kFcKsteps + p * 4 + (mt >> 1)) * 512;          // elements
                const int within = (((mt & 1) * 2 + (lg >> 1)) * 16 + lr) * 8 + (lg & 1) * 4;
                *reinterpret_cast<v4 *>(p1 + frag + within) = out;
            }
            if (lg == 0 && valid) {
                const float v = fmaxf(acc[8][px][0] + bias[128], 0.f);
                v1[(size_t)board * kPixels + p] = (float)(T)v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// part[split][board][action] = sum over the split's k-steps of p1 . Wfc^T
//   a     [mtiles][576][64][8]  p1 as A fragments (head_conv_kernel)
//   w     [99][576][64][8]      FC weights as B fragments: action tile, k-step, lane = (k % 32) / 8 * 16 + action % 16
//   part  f32 [splits][mtiles][99][64][4]  fragment-major: lane = (board % 16) / 4 * 16 + action % 16, 4 consecutive boards
// grid ceil(mtiles / 16) * 9 * splits, 512 threads: wave w owns board tiles 2w, 2w+1 of the workgroup's 16 and all 11
// action tiles.  Stages of four k-steps (one stage of compute = 2.8 k cycles per SIMD covers the L2 / HBM latency of the next
// stage's loads): the 44 action fragments of a stage go through LDS (double buffered, one barrier per stage), the wave's 8
// board fragments straight into registers one stage ahead.
template <typename E>
__global__ void __launch_bounds__(512, 2)
policy_fc_kernel(const typename E::T *__restrict__ a, const typename E::T *__restrict__ w, float *__restrict__ part, int mtiles,
                 int splits)
{
    typedef typename E::v8 v8;
    typedef typename E::T T;
    constexpr int KS = 4;                                           // k-steps per stage
    constexpr int STAGE_FRAGS = KS * kFcNB;                         // 44 fragments of 1 KiB per stage
    constexpr int LOADS = (STAGE_FRAGS * 64 + 511) / 512;           // 16-byte pieces per thread and stage (6, the last partial)
    constexpr int WM = kFcMB / 8;                                   // board tiles per wave (2): 88 accumulator registers, so
                                                                    // that everything fits the 256 registers two waves per
                                                                    // SIMD leave each (176 accumulators made hipcc shuttle
                                                                    // them through the AGPR half every stage: 212 us)
    __shared__ __attribute__((aligned(16))) unsigned char bl[2][STAGE_FRAGS * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    // Workgroup -> (board block, action block, K range).  Workgroups are dealt round-robin over the 8 XCDs (observed, not
    // promised: only speed could depend on it), and each XCD has its own L2: the logical index runs XCD-major, K range
    // slowest, so that the ~32 workgroups of one XCD share one K range (its slices of p1 and of the FC weights).  Measured:
    // no difference to the plain order (95 -> 101 us, box-to-box noise): the kernel streams 572 MB of operands for 96 MB of
    // unique bytes at ~6 TB/s, the rate of the level behind the L2s (profiles/r04_heads.md); kept because it is harmless.
    const int mblocks = (mtiles + kFcMB - 1) / kFcMB, total = mblocks * 9 * splits;
    const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
    const int logical = xcd * (total >> 3) + (xcd < (total & 7) ? xcd : (total & 7)) + slot;
    const int zz = logical / (mblocks * 9), rem = logical - zz * (mblocks * 9);
    const int nblk = rem / mblocks, mblk = rem - nblk * mblocks;
    const int mt0 = mblk * kFcMB + wave * WM;
    const int nt0 = nblk * kFcNB;
    const int nstages = kFcKsteps / KS;
    const int s0 = (int)((long long)nstages * zz / splits), s1 = (int)((long long)nstages * (zz + 1) / splits);
    const int m0 = mt0 < mtiles ? mt0 : mtiles - 1, m1 = mt0 + 1 < mtiles ? mt0 + 1 : mtiles - 1;   // tail tiles repeat the last one, never stored

    // piece q of a stage: fragment q / 64 (= kk * 11 + n), lane q % 64; thread t moves pieces t, t + 512, ..., t + 2560 (the
    // last one only for t < 256).  Explicit scalars, unconditional loads (the odd piece reads a clamped address): as an
    // array filled under a condition hipcc parked these registers in LDS (48 KiB of "promoted alloca", every piece written
    // and read back once more, and the wait for the loads moved in front of the stage's MFMAs).
    static_assert(LOADS == 6 && STAGE_FRAGS * 64 == 5 * 512 + 256, "piece schedule of a stage");
    const bool last_piece = tid < 256;
    const T *bsrc[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int q = j < 5 ? tid + j * 512 : (last_piece ? tid + 2560 : tid + 2048);
        const int f = q >> 6, kk = f / kFcNB, n = f - kk * kFcNB;
        bsrc[j] = w + (((size_t)(nt0 + n) * kFcKsteps + kk) * 64 + (q & 63)) * 8;       // + stage * KS * 512 elements
    }
    uint4 b0, b1, b2, b3, b4, b5;
    auto load_b = [&](int st) {
        const size_t o = (size_t)st * KS * 512;
        b0 = *reinterpret_cast<const uint4 *>(bsrc[0] + o);
        b1 = *reinterpret_cast<const uint4 *>(bsrc[1] + o);
        b2 = *reinterpret_cast<const uint4 *>(bsrc[2] + o);
        b3 = *reinterpret_cast<const uint4 *>(bsrc[3] + o);
        b4 = *reinterpret_cast<const uint4 *>(bsrc[4] + o);
        b5 = *reinterpret_cast<const uint4 *>(bsrc[5] + o);
    };
    auto store_b = [&](int buf) {
        unsigned char *d = bl[buf] + (size_t)tid * 16;
        *reinterpret_cast<uint4 *>(d) = b0;
        *reinterpret_cast<uint4 *>(d + 512 * 16) = b1;
        *reinterpret_cast<uint4 *>(d + 2 * 512 * 16) = b2;
        *reinterpret_cast<uint4 *>(d + 3 * 512 * 16) = b3;
        *reinterpret_cast<uint4 *>(d + 4 * 512 * 16) = b4;
        if (last_piece) *reinterpret_cast<uint4 *>(d + 5 * 512 * 16) = b5;
    };
    v8 A[2][KS][WM];
    auto load_a = [&](int st, int buf) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            A[buf][kk][0] = *reinterpret_cast<const v8 *>(a + (((size_t)m0 * kFcKsteps + st * KS + kk) * 64 + lane) * 8);
            A[buf][kk][1] = *reinterpret_cast<const v8 *>(a + (((size_t)m1 * kFcKsteps + st * KS + kk) * 64 + lane) * 8);
        }
    };
    f32x4 acc[WM][kFcNB];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < kFcNB; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (the two stage buffers alternate at compile time -- `half` -- so that A[][] and the LDS buffer are never indexed by
    // a run-time value: a dynamically indexed register array lives in scratch memory)
    auto compute = [&](int cur) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int n = 0; n < kFcNB; ++n) {
                const v8 b = *reinterpret_cast<const v8 *>(bl[cur] + ((size_t)(kk * kFcNB + n) * 64 + lane) * 16);
#pragma unroll
                for (int i = 0; i < WM; ++i) acc[i][n] = E::mfma(A[cur][kk][i], b, acc[i][n]);
            }
    };
    load_b(s0);
    load_a(s0, 0);
    store_b(0);
    __syncthreads();
    for (int st = s0; st < s1; st += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cs = st + half;
            if (cs < s1) {
                const bool more = cs + 1 < s1;
                if (more) {
                    load_b(cs + 1);
                    load_a(cs + 1, half ^ 1);
                }
                // (pinned, for the optimizer's memory motion and for the scheduler: left alone, the compiler moves the LDS stores of the next stage -- which depend on nothing the
                // MFMAs produce -- right behind their loads, in FRONT of this stage's 88 MFMAs, and every stage then waits
                // out a full L2 round trip: 60 % of the wave cycles were s_waitcnt, profiles/r04_heads.md)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                compute(half);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::: "memory");
                if (more) store_b(half ^ 1);
            }
            __syncthreads();
        }
    }
    // lane holds D[boards (mt0+i)*16 + lg*4 + r][action (nt0+n)*16 + lr], r = 0..3: the partial sums leave fragment-major --
    // part[K range][board tile][action tile][lane][4] -- one 16-byte store per lane, 1 KiB per wave-instruction (as a
    // [board][action] matrix every store instruction wrote 64-byte pieces of four rows: 45 MB that way cost more than the GEMM)
    float4 *dst = reinterpret_cast<float4 *>(part) + (size_t)zz * mtiles * kFcNT * 64;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        if (mt0 + i >= mtiles) continue;
#pragma unroll
        for (int n = 0; n < kFcNB; ++n)
            dst[((size_t)(mt0 + i) * kFcNT + nt0 + n) * 64 + lane] = float4{acc[i][n][0], acc[i][n][1], acc[i][n][2], acc[i][n][3]};
    }
    (void)lr;
    (void)lg;
}

// ---------------------------------------------------------------------------------------------
// One workgroup per FOUR consecutive boards (the four rows one lane of the FC's accumulator layout holds: every read of the
// partial sums is a whole 16-byte piece): logits = bias + the partial sums in ascending K-range order, p = softmax(logits);
// value = tanh(w2 . relu(W1 v1 + b1) + b2) (alpha_net.py:66-69), wave w of the workgroup taking board w.
__global__ void __launch_bounds__(256)
head_finish_kernel(const float *__restrict__ part, int splits, int mtiles, const float *__restrict__ fcb, const float *__restrict__ v1,
                   const float *__restrict__ w1t, const float *__restrict__ b1, const float *__restrict__ w2,
                   const float *__restrict__ b2, float *__restrict__ p, float *__restrict__ v, int batch)
{
    __shared__ float red[2][4][4];
    __shared__ float vin[4][kPixels];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = g * 4, mtile = b0 >> 4, lgq = (b0 & 15) >> 2;
    constexpr int PER = (kActions + 255) / 256;                      // 7
    const float4 *part4 = reinterpret_cast<const float4 *>(part);
    float z[PER][4];
    float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int aidx = tid + j * 256;
        float s4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        if (aidx < kActions) {
            float4 ps[8];                            // (all loads in flight, then the sums in ascending K-range order)
            const size_t at = ((size_t)mtile * kFcNT + (aidx >> 4)) * 64 + lgq * 16 + (aidx & 15);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                ps[k] = k < splits ? part4[(size_t)k * mtiles * kFcNT * 64 + at] : float4{0.f, 0.f, 0.f, 0.f};
            const float bias = fcb[aidx];
            s4[0] = s4[1] = s4[2] = s4[3] = bias;
#pragma unroll
            for (int k = 0; k < 8; ++k) { s4[0] += ps[k].x; s4[1] += ps[k].y; s4[2] += ps[k].z; s4[3] += ps[k].w; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { z[j][r] = s4[r]; mx[r] = fmaxf(mx[r], s4[r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        for (int d = 32; d >= 1; d >>= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], d, 64));
        if (lane == 0) red[0][wave][r] = mx[r];
    }
    if (tid < 4 * 36) {                              // the four boards' value-convolution rows -> LDS (144 floats each)
        const int r = tid / 36, i4 = tid - r * 36;
        if (b0 + r < batch)
            reinterpret_cast<float4 *>(vin[r])[i4] = reinterpret_cast<const float4 *>(v1 + (size_t)(b0 + r) * kPixels)[i4];
    }
    __syncthreads();
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(fmaxf(red[0][0][r], red[0][1][r]), fmaxf(red[0][2][r], red[0][3][r]));
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            z[j][r] = tid + j * 256 < kActions ? expf(z[j][r] - mx[r]) : 0.f;
            sum[r] += z[j][r];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        for (int d = 32; d >= 1; d >>= 1) sum[r] += __shfl_xor(sum[r], d, 64);
        if (lane == 0) red[1][wave][r] = sum[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (b0 + r >= batch) continue;
        const float inv = 1.0f / ((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r]));
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (tid + j * 256 < kActions) p[(size_t)(b0 + r) * kActions + tid + j * 256] = z[j][r] * inv;
    }
    // value head: wave w = board b0 + w, lane = hidden unit; w1t is fc1's weight TRANSPOSED ([144][64]: consecutive lanes
    // read consecutive floats)
    if (b0 + wave < batch) {
        float h = b1[lane];
#pragma unroll 8
        for (int i = 0; i < kPixels; ++i) h += w1t[i * 64 + lane] * vin[wave][i];
        h = fmaxf(h, 0.f) * w2[lane];
        for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d, 64);
        if (lane == 0) v[b0 + wave] = tanhf(h + b2[0]);
    }
}

}  // namespace heads
}  // namespace hive

using namespace hive;
using namespace hive::heads;

extern "C" int hive_nn_heads_splits(int batch)
{
    // A FIXED number of K ranges: a board's logits are then the same bits in a batch of 32 and in a batch of 4096 (the
    // engine's reproducibility across batch sizes and GPU counts rests on that).  7 ranges x 9 action blocks x 4 board
    // blocks = 252 workgroups at 1024 boards, one round of the chip.
    (void)batch;
    return 7;                                      // (<= 8: head_finish_kernel keeps one register per range)
}

extern "C" long long hive_nn_heads_workspace_bytes(int batch)
{
    // p1 fragments (16-bit) + v1 (f32) + the FC's partial sums (f32)
    const long long mtiles = (batch + 15) / 16;
    const long long p1 = mtiles * kFcKsteps * 1024, v1 = mtiles * 16 * kPixels * 4;
    const long long part = (long long)hive_nn_heads_splits(batch) * mtiles * 16 * kActions * 4;
    return p1 + v1 + part;
}

extern "C" int hive_nn_heads(const void *x, int batch, int dtype, const void *wconv, const float *bconv, const void *wfc,
                             const float *bfc, const float *w1, const float *b1, const float *w2, const float *b2, void *workspace,
                             float *p, float *v, void *stream)
{
    if (!x || !wconv || !bconv || !wfc || !bfc || !w1 || !b1 || !w2 || !b2 || !workspace || !p || !v || batch <= 0)
        return set_error(HIVE_E_ARG, "hive_nn_heads: bad argument");
    if (dtype != HIVE_BF16 && dtype != HIVE_F16) return set_error(HIVE_E_ARG, "hive_nn_heads: dtype must be HIVE_BF16 or HIVE_F16");
    if ((uintptr_t)workspace & 15) return set_error(HIVE_E_ARG, "hive_nn_heads: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int mtiles = (batch + 15) / 16, splits = hive_nn_heads_splits(batch);
    unsigned char *ws = (unsigned char *)workspace;
    void *p1 = ws;
    float *v1 = (float *)(ws + (size_t)mtiles * kFcKsteps * 1024);
    float *part = v1 + (size_t)mtiles * 16 * kPixels;
    const dim3 g1((unsigned)mtiles, 4), g2((unsigned)(((mtiles + kFcMB - 1) / kFcMB) * 9 * splits));
    if (dtype == HIVE_BF16) {
        hipLaunchKernelGGL(head_conv_kernel<Bf16>, g1, dim3(256), 0, s, (const __bf16 *)x, (const __bf16 *)wconv, bconv, (__bf16 *)p1, v1, batch);
        hipLaunchKernelGGL(policy_fc_kernel<Bf16>, g2, dim3(512), 0, s, (const __bf16 *)p1, (const __bf16 *)wfc, part, mtiles, splits);
    } else {
        hipLaunchKernelGGL(head_conv_kernel<F16>, g1, dim3(256), 0, s, (const _Float16 *)x, (const _Float16 *)wconv, bconv, (_Float16 *)p1, v1, batch);
        hipLaunchKernelGGL(policy_fc_kernel<F16>, g2, dim3(512), 0, s, (const _Float16 *)p1, (const _Float16 *)wfc, part, mtiles, splits);
    }
    hipLaunchKernelGGL(head_finish_kernel, dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, s, part, splits, mtiles, bfc, v1, w1, b1, w2, b2, p,
                       v, batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_heads: ") + hipGetErrorString(e));
    return HIVE_OK;
}
