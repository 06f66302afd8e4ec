// hive_train.hip -- training-mode BatchNorm (+ skip connection + ReLU) of the 256-channel residual tower,
// forward and backward, for the training step of alpha_zero/alpha_net.py:117-162.
//
// Activations are channels-last bf16, i.e. a [rows][256] matrix with rows = batch * 144; everything here is
// HBM-bound streaming: a thread owns 8 consecutive channels (one 16-byte load per row) and walks rows with a grid
// stride, so a wavefront reads two full 512-byte rows per instruction and the per-channel factors live in registers.
//
//   forward : pass 1  per-channel sum / sum of squares   (x read once)            -> partial[grid][512]
//             finish  mean, 1/std, scale = gamma/std, shift = beta - mean*scale, running statistics
//             pass 2  y = relu(x*scale + shift (+ residual))                      (x (+res) read, y written)
//   backward: pass 1  g = dy * (y > 0);  dbeta = sum g,  dgamma = sum g * xhat    (dy, y, x read)
//             finish  dgamma, dbeta and the three per-channel coefficients of dx
//             pass 2  dx = gamma/std * (g - dbeta/N - xhat * dgamma/N), dres = g  (dy, y, x read; dx (, dres) written)
//
// ~0.5 GB of traffic per layer at batch 512 for forward + backward together, against ~1.4 GB-equivalent time in the
// library kernels it replaces (tools/bn_bench.py).
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/hive_abi.h"
#include "../../include/hive_nn.h"

namespace hive {
int set_error(int code, const std::string &msg);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kBnC = 256;            // channels of the tower
constexpr int kBnGrid = 512;         // workgroups of the streaming passes (2 per CU)
constexpr int kBnThreads = 256;      // 8 row lanes x 32 channel groups of 8

__device__ __forceinline__ void load8(const __bf16 *p, float (&v)[8])
{
    const bf16x8 r = *reinterpret_cast<const bf16x8 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
}
__device__ __forceinline__ void store8(__bf16 *p, const float (&v)[8])
{
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
    *reinterpret_cast<bf16x8 *>(p) = r;
}

// Reduce the 8 row lanes of a workgroup and write its partial sums: partial[wg][0..255] = a, [256..511] = b.
// C < 256 channels (the heads' BatchNorm2d(128) / (1): a [positions][C] matrix read as [positions * C / 256][256], so
// columns j, j + C, j + 2C, ... are the same channel): the columns of a channel are added up here and only entries
// 0 .. C-1 of each half are written.
__device__ __forceinline__ void reduce_rows(float (&a)[8], float (&b)[8], float *partial_wg, int C)
{
    __shared__ float red[8][2 * kBnC];
    __shared__ float col[2 * kBnC];
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[rl][cg * 8 + i] = a[i];
        red[rl][kBnC + cg * 8 + i] = b[i];
    }
    __syncthreads();
    if (C == kBnC) {
        for (int j = threadIdx.x; j < 2 * kBnC; j += kBnThreads) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r][j];
            partial_wg[j] = s;
        }
        return;
    }
    for (int j = threadIdx.x; j < 2 * kBnC; j += kBnThreads) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][j];
        col[j] = s;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += kBnThreads) {
        const int half = j / C, c = j - half * C;
        float s = 0.f;
        for (int m = c; m < kBnC; m += C) s += col[half * kBnC + m];
        partial_wg[half * kBnC + c] = s;
    }
}

// Finish kernels: 32 workgroups x 256 threads, workgroup w owns channels 8 w .. 8 w + 7; thread (k, j) adds the
// partials of streaming workgroups k, k+32, ... for channel c = 8 w + j (16 independent loads per array); the threads
// with k == 0 return true with the two totals of their channel.
__device__ __forceinline__ bool sum_partials(const float *partial, int grid, int &c, float &a, float &b)
{
    __shared__ float tot[32][16];
    const int j = threadIdx.x & 7, k = threadIdx.x >> 3;
    c = blockIdx.x * 8 + j;
    float s = 0.f, q = 0.f;
#pragma unroll 4
    for (int g = k; g < grid; g += 32) {
        s += partial[(long long)g * 2 * kBnC + c];
        q += partial[(long long)g * 2 * kBnC + kBnC + c];
    }
    tot[k][j] = s;
    tot[k][8 + j] = q;
    __syncthreads();
    if (k != 0) return false;
    a = b = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) { a += tot[r][j]; b += tot[r][8 + j]; }
    return true;
}

// ---------------------------------------------------------------- forward
__global__ void __launch_bounds__(kBnThreads)
bn_fwd_stats_kernel(const __bf16 *__restrict__ x, long long rows, float *__restrict__ partial, int C)
{
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (long long r = (long long)blockIdx.x * 8 + rl; r < rows; r += (long long)gridDim.x * 8) {
        float v[8];
        load8(x + r * kBnC + cg * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s[i] += v[i]; q[i] += v[i] * v[i]; }
    }
    reduce_rows(s, q, partial + (long long)blockIdx.x * 2 * kBnC, C);
}

// coef[0..255] = scale, [256..511] = shift
__global__ void __launch_bounds__(kBnThreads)
bn_fwd_finish_kernel(const float *__restrict__ partial, int grid, long long rows, const float *__restrict__ gamma,
                     const float *__restrict__ beta, float *__restrict__ running_mean, float *__restrict__ running_var,
                     float momentum, float eps, float *__restrict__ save_mean, float *__restrict__ save_invstd,
                     float *__restrict__ coef, int C)
{
    float s, q;
    int c;
    if (!sum_partials(partial, grid, c, s, q) || c >= C) return;
    const float n = (float)rows * (float)(kBnC / C);            // values per channel
    const float mean = s / n;
    const float var = fmaxf(q / n - mean * mean, 0.f);          // biased, as BatchNorm normalises with it
    const float invstd = rsqrtf(var + eps);
    const float scale = gamma[c] * invstd;
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    coef[c] = scale;
    coef[kBnC + c] = beta[c] - mean * scale;
    if (running_mean != nullptr) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    }
}

template <bool RES>
__global__ void __launch_bounds__(kBnThreads)
bn_fwd_apply_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ res, const float *__restrict__ coef,
                    long long rows, int relu, __bf16 *__restrict__ y, int cmask)
{
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float sc[8], sh[8];
    const int c0 = (cg * 8) & cmask;            // 8 consecutive channels (two 16-byte loads per array) unless C < 8
    if (cmask >= 7) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = coef[c0 + i]; sh[i] = coef[kBnC + c0 + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = coef[i & cmask]; sh[i] = coef[kBnC + (i & cmask)]; }
    }
#pragma unroll 4
    for (long long r = (long long)blockIdx.x * 8 + rl; r < rows; r += (long long)gridDim.x * 8) {
        const long long o = r * kBnC + cg * 8;
        float v[8];
        load8(x + o, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * sc[i] + sh[i];
        if (RES) {
            float w[8];
            load8(res + o, w);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += w[i];
        }
        if (relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        store8(y + o, v);
    }
}

// ---------------------------------------------------------------- backward
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_stats_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ y, const __bf16 *__restrict__ x,
                    const float *__restrict__ save_mean, const float *__restrict__ save_invstd, long long rows, int relu,
                    float *__restrict__ partial, int C)
{
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int cmask = C - 1;
    float mu[8], is[8];
    const int c0 = (cg * 8) & cmask;
    if (cmask >= 7) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { mu[i] = save_mean[c0 + i]; is[i] = save_invstd[c0 + i]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { mu[i] = save_mean[i & cmask]; is[i] = save_invstd[i & cmask]; }
    }
    float db[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (long long r = (long long)blockIdx.x * 8 + rl; r < rows; r += (long long)gridDim.x * 8) {
        const long long o = r * kBnC + cg * 8;
        float g[8], xv[8];
        load8(dy + o, g);
        load8(x + o, xv);
        if (relu) {
            float yv[8];
            load8(y + o, yv);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { db[i] += g[i]; dg[i] += g[i] * (xv[i] - mu[i]) * is[i]; }
    }
    reduce_rows(db, dg, partial + (long long)blockIdx.x * 2 * kBnC, C);
}

// coef[0..255] = gamma/std, [256..511] = dbeta/N, [512..767] = dgamma/N
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_finish_kernel(const float *__restrict__ partial, int grid, long long rows, const float *__restrict__ gamma,
                     const float *__restrict__ save_invstd, float *__restrict__ dgamma, float *__restrict__ dbeta,
                     float *__restrict__ coef, int C)
{
    float b, g;
    int c;
    if (!sum_partials(partial, grid, c, b, g) || c >= C) return;
    dbeta[c] = b;
    dgamma[c] = g;
    const float n = (float)rows * (float)(kBnC / C);
    coef[c] = gamma[c] * save_invstd[c];
    coef[kBnC + c] = b / n;
    coef[2 * kBnC + c] = g / n;
}

template <bool RES>
__global__ void __launch_bounds__(kBnThreads)
bn_bwd_apply_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ y, const __bf16 *__restrict__ x,
                    const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                    const float *__restrict__ coef, long long rows, int relu, __bf16 *__restrict__ dx,
                    __bf16 *__restrict__ dres, int cmask)
{
    const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float mu[8], is[8], a[8], cb[8], cgm[8];
    const int c0 = (cg * 8) & cmask;
    if (cmask >= 7) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mu[i] = save_mean[c0 + i];
            is[i] = save_invstd[c0 + i];
            a[i] = coef[c0 + i];
            cb[i] = coef[kBnC + c0 + i];
            cgm[i] = coef[2 * kBnC + c0 + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = i & cmask;
            mu[i] = save_mean[c];
            is[i] = save_invstd[c];
            a[i] = coef[c];
            cb[i] = coef[kBnC + c];
            cgm[i] = coef[2 * kBnC + c];
        }
    }
#pragma unroll 4
    for (long long r = (long long)blockIdx.x * 8 + rl; r < rows; r += (long long)gridDim.x * 8) {
        const long long o = r * kBnC + cg * 8;
        float g[8], xv[8];
        load8(dy + o, g);
        load8(x + o, xv);
        if (relu) {
            float yv[8];
            load8(y + o, yv);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
        }
        if (RES) store8(dres + o, g);
        float d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = a[i] * (g[i] - cb[i] - (xv[i] - mu[i]) * is[i] * cgm[i]);
        store8(dx + o, d);
    }
}

// ---------------------------------------------------------------- weights of the training-step convolutions
// fp32 [256][cin][3][3] (the nn.Conv2d parameter) -> bf16 fragment-major [9][cinp/32][16][4][16][8] of
// hive_nn_conv3x3 (include/hive_nn.h), once per step and layer.  transpose != 0 builds the weights of the
// data-gradient convolution instead: dx[q][ci] = sum_{tap,co} dy[q + tap][co] * w[co][ci][-tap], i.e. the same
// kernel run on dy with input/output channels swapped and the taps rotated by 180 degrees.
__global__ void __launch_bounds__(256)
pack_weights_kernel(const float *__restrict__ w, int cin, int cinp, int transpose, int channels_last,
                    __bf16 *__restrict__ out)
{
    const int ks_n = cinp / 32;
    const int idx = blockIdx.x * 256 + threadIdx.x;              // one 8-channel vector of the output
    if (idx >= 9 * ks_n * 16 * 4 * 16) return;
    const int row = idx & 15, kg = (idx >> 4) & 3, mt = (idx >> 6) & 15, rest = idx >> 10;
    const int ks = rest % ks_n, tap = rest / ks_n;
    const int m = mt * 16 + row, c0 = ks * 32 + kg * 8;         // output channel of the GEMM, first of 8 reduction channels
    const int wt = transpose ? 8 - tap : tap;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        float f = 0.f;
        if (c < cin) {
            const int ko = transpose ? c : m, ki = transpose ? m : c;      // indices into w[out channel][in channel][tap]
            f = channels_last ? w[((long long)ko * 9 + wt) * cin + ki] : w[((long long)ko * cin + ki) * 9 + wt];
        }
        v[e] = (__bf16)f;
    }
    *reinterpret_cast<bf16x8 *>(out + (long long)idx * 8) = v;
}

// All 256 -> 256 convolutions of a training step at once: blockIdx.y = 2 * layer + form (0 forward, 1 data gradient); one launch
// instead of one per convolution and direction (77 launches x 5.6 us of a 15 ms step).
__global__ void __launch_bounds__(256)
pack_weights_multi_kernel(const float *const *__restrict__ ws, int channels_last, __bf16 *__restrict__ out_fwd,
                          __bf16 *__restrict__ out_t)
{
    constexpr int cin = 256, ks_n = 8;
    const int layer = blockIdx.y >> 1, transpose = blockIdx.y & 1;
    const float *__restrict__ w = ws[layer];
    __bf16 *out = (transpose ? out_t : out_fwd) + (long long)layer * 9 * 256 * 256;
    const int idx = blockIdx.x * 256 + threadIdx.x;              // one 8-channel vector of the output
    if (idx >= 9 * ks_n * 16 * 4 * 16) return;
    const int row = idx & 15, kg = (idx >> 4) & 3, mt = (idx >> 6) & 15, rest = idx >> 10;
    const int ks = rest % ks_n, tap = rest / ks_n;
    const int m = mt * 16 + row, c0 = ks * 32 + kg * 8;
    const int wt = transpose ? 8 - tap : tap;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const int ko = transpose ? c : m, ki = transpose ? m : c;
        v[e] = (__bf16)(channels_last ? w[((long long)ko * 9 + wt) * cin + ki] : w[((long long)ko * cin + ki) * 9 + wt]);
    }
    *reinterpret_cast<bf16x8 *>(out + (long long)idx * 8) = v;
}

}  // namespace hive

using namespace hive;

extern "C" int hive_nn_pack_conv3x3_weights_multi(const float *const *weights, int n, int channels_last, void *out_fwd,
                                                  void *out_t, void *stream)
{
    if (!weights || !out_fwd || !out_t || n <= 0 || n > 32767)
        return set_error(HIVE_E_ARG, "hive_nn_pack_conv3x3_weights_multi: bad argument");
    const int vecs = 9 * 8 * 16 * 4 * 16;
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)(vecs / 256), (unsigned)(2 * n)), dim3(256), 0, (hipStream_t)stream,
                       weights, channels_last, (__bf16 *)out_fwd, (__bf16 *)out_t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_pack_conv3x3_weights_multi: ") + hipGetErrorString(e));
    return HIVE_OK;
}

extern "C" int hive_nn_pack_conv3x3_weights(const float *w, int cin, int transpose, int channels_last, void *out,
                                            void *stream)
{
    if (!w || !out || (cin != 56 && cin != 256) || (transpose && cin != 256))
        return set_error(HIVE_E_ARG, "hive_nn_pack_conv3x3_weights: cin must be 56 or 256 (256 for the transposed form)");
    const int cinp = (cin + 63) / 64 * 64;
    const int vecs = 9 * (cinp / 32) * 16 * 4 * 16;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cin,
                       cinp, transpose, channels_last, (__bf16 *)out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(HIVE_E_DEVICE, std::string("hive_nn_pack_conv3x3_weights: ") + hipGetErrorString(e));
    return HIVE_OK;
}

#define BN_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return set_error(HIVE_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// channels: a power of two <= 256 whose [positions][channels] matrix is a whole number of 256-wide rows
static bool bn_shape_ok(long long rows, int channels)
{
    return channels >= 1 && channels <= kBnC && (channels & (channels - 1)) == 0 && (rows * channels) % kBnC == 0;
}
static const char *kBnShapeMsg = "hive_nn_bn_act: channels must be a power of two <= 256 and rows * channels a multiple of 256";

extern "C" int hive_nn_bn_workspace_floats(void) { return kBnGrid * 2 * kBnC + 3 * kBnC; }

extern "C" int hive_nn_bn_act_fwd(const void *x, const void *residual, const float *gamma, const float *beta,
                                  float *running_mean, float *running_var, float momentum, float eps, void *y,
                                  float *save_mean, float *save_invstd, float *workspace, long long rows, int channels,
                                  int relu, void *stream)
{
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || rows <= 0)
        return set_error(HIVE_E_ARG, "hive_nn_bn_act_fwd: bad argument");
    if (!bn_shape_ok(rows, channels)) return set_error(HIVE_E_ARG, kBnShapeMsg);
    if ((running_mean == nullptr) != (running_var == nullptr))
        return set_error(HIVE_E_ARG, "hive_nn_bn_act_fwd: running_mean and running_var go together");
    hipStream_t s = (hipStream_t)stream;
    const int C = channels;
    rows = rows * C / kBnC;                                     // the matrix as [rows][256]
    const int grid = (int)((rows + 7) / 8 < kBnGrid ? (rows + 7) / 8 : kBnGrid);
    const int fgrid = C >= 8 ? C / 8 : 1;
    float *partial = workspace, *coef = workspace + (long long)kBnGrid * 2 * kBnC;
    const __bf16 *X = (const __bf16 *)x, *R = (const __bf16 *)residual;
    hipLaunchKernelGGL(bn_fwd_stats_kernel, dim3(grid), dim3(kBnThreads), 0, s, X, rows, partial, C);
    hipLaunchKernelGGL(bn_fwd_finish_kernel, dim3(fgrid), dim3(kBnThreads), 0, s, partial, grid, rows, gamma, beta, running_mean,
                       running_var, momentum, eps, save_mean, save_invstd, coef, C);
    if (R) hipLaunchKernelGGL((bn_fwd_apply_kernel<true>), dim3(grid), dim3(kBnThreads), 0, s, X, R, coef, rows, relu, (__bf16 *)y, C - 1);
    else hipLaunchKernelGGL((bn_fwd_apply_kernel<false>), dim3(grid), dim3(kBnThreads), 0, s, X, R, coef, rows, relu, (__bf16 *)y, C - 1);
    BN_TRY(hipGetLastError());
    return HIVE_OK;
}

extern "C" int hive_nn_bn_act_fwd_partial(const void *x, const void *residual, const float *gamma, const float *beta,
                                          float *running_mean, float *running_var, float momentum, float eps, void *y,
                                          float *save_mean, float *save_invstd, float *workspace, const float *partial,
                                          int parts, long long rows, int relu, void *stream)
{
    // hive_nn_bn_act_fwd for 256 channels without its statistics pass: `partial` = parts x [2][256] per-channel sums / sums of
    // squares that together cover all `rows` positions (written by hive_nn_conv72_stats, the convolution in front)
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || !partial || parts <= 0 || rows <= 0)
        return set_error(HIVE_E_ARG, "hive_nn_bn_act_fwd_partial: bad argument");
    if ((running_mean == nullptr) != (running_var == nullptr))
        return set_error(HIVE_E_ARG, "hive_nn_bn_act_fwd_partial: running_mean and running_var go together");
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((rows + 7) / 8 < kBnGrid ? (rows + 7) / 8 : kBnGrid);
    float *coef = workspace + (long long)kBnGrid * 2 * kBnC;
    const __bf16 *X = (const __bf16 *)x, *R = (const __bf16 *)residual;
    hipLaunchKernelGGL(bn_fwd_finish_kernel, dim3(kBnC / 8), dim3(kBnThreads), 0, s, partial, parts, rows, gamma, beta, running_mean,
                       running_var, momentum, eps, save_mean, save_invstd, coef, kBnC);
    if (R) hipLaunchKernelGGL((bn_fwd_apply_kernel<true>), dim3(grid), dim3(kBnThreads), 0, s, X, R, coef, rows, relu, (__bf16 *)y, kBnC - 1);
    else hipLaunchKernelGGL((bn_fwd_apply_kernel<false>), dim3(grid), dim3(kBnThreads), 0, s, X, R, coef, rows, relu, (__bf16 *)y, kBnC - 1);
    BN_TRY(hipGetLastError());
    return HIVE_OK;
}

extern "C" int hive_nn_bn_act_bwd(const void *dy, const void *x, const void *y, const float *gamma, const float *save_mean,
                                  const float *save_invstd, void *dx, void *dresidual, float *dgamma, float *dbeta,
                                  float *workspace, long long rows, int channels, int relu, void *stream)
{
    if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace || rows <= 0 ||
        (relu && !y))
        return set_error(HIVE_E_ARG, "hive_nn_bn_act_bwd: bad argument");
    if (!bn_shape_ok(rows, channels)) return set_error(HIVE_E_ARG, kBnShapeMsg);
    hipStream_t s = (hipStream_t)stream;
    const int C = channels;
    rows = rows * C / kBnC;
    const int grid = (int)((rows + 7) / 8 < kBnGrid ? (rows + 7) / 8 : kBnGrid);
    const int fgrid = C >= 8 ? C / 8 : 1;
    float *partial = workspace, *coef = workspace + (long long)kBnGrid * 2 * kBnC;
    const __bf16 *DY = (const __bf16 *)dy, *X = (const __bf16 *)x, *Y = (const __bf16 *)y;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(grid), dim3(kBnThreads), 0, s, DY, Y, X, save_mean, save_invstd, rows, relu,
                       partial, C);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(fgrid), dim3(kBnThreads), 0, s, partial, grid, rows, gamma, save_invstd, dgamma,
                       dbeta, coef, C);
    if (dresidual)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true>), dim3(grid), dim3(kBnThreads), 0, s, DY, Y, X, save_mean, save_invstd,
                           coef, rows, relu, (__bf16 *)dx, (__bf16 *)dresidual, C - 1);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false>), dim3(grid), dim3(kBnThreads), 0, s, DY, Y, X, save_mean, save_invstd,
                           coef, rows, relu, (__bf16 *)dx, (__bf16 *)dresidual, C - 1);
    BN_TRY(hipGetLastError());
    return HIVE_OK;
}
