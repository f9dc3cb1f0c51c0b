// The classification loss of the criterion on the dense class map, for gfx950 (MI355X): what the reference's
// `cls_criterion` computes from `pred_logits` and the [B, Q] class map the matching indices were scattered into
// (/root/reference/models/detr/loss.py:127) -- `FocalLoss` (:198-213, `sigmoid_focal_loss` :217-242) or
// `CrossEntropy` (:185-195, nn.CrossEntropyLoss with the no-object class weight of :60-62).  The reference
// expands the class map into a one-hot tensor and runs about a dozen elementwise torch kernels over [B, Q, C]
// forward and as many backward; here one pass reads the logits once, looks the class of a row up in the map,
// and writes the sum and (when asked) the gradient of every logit.  HBM-bound streaming kernels: no tiling, no
// MFMA; the reductions have a fixed order, so results are reproducible run to run.
#include <algorithm>

#include "dm_common.h"

namespace dm {
int check_hip(hipError_t e, const char *what);
}

namespace {

constexpr int CLS_THREADS = 256;
constexpr int CLS_MAX_BLOCKS = DM_CLS_LOSS_PARTIALS / 2;     // two partial sums per workgroup

// Sum over the workgroup, valid in thread 0: butterflies inside the wave, then the waves in index order.
__device__ __forceinline__ double block_sum(double v, double *lds)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                 // (the previous use of lds has been read)
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < CLS_THREADS / 64; ++w) total += lds[w];
    return total;
}

struct FocalParams {
    float alpha, gamma, scale;
    int64_t ignore_index;
    int use_ignore;
};

// One logit x of a row whose class is `positive` (t = 1) or not (t = 0): sigmoid_focal_loss (loss.py:233-240) in the
// reference's own operation order, and d loss / d x as torch autograd derives it from those operations.
template <bool GAMMA2>
__device__ __forceinline__ float focal_term(float x, bool positive, const FocalParams &fp, float *grad)
{
    const float p = 1.0f / (1.0f + expf(-x));                                // prob = inputs.sigmoid()
    // binary_cross_entropy_with_logits: softplus(-x) for t = 1, softplus(x) for t = 0
    const float z = positive ? -x : x;
    const float ce = fmaxf(z, 0.0f) + log1pf(expf(-fabsf(x)));
    const float q = 1.0f - p;
    const float base = positive ? q : 1.0f - q;                               // 1 - p_t, p_t = p t + (1 - p)(1 - t)
    float m, dm;                                                              // (1 - p_t) ** gamma and its derivative in base
    if (GAMMA2) {
        m = base * base;
        dm = 2.0f * base;
    } else {
        m = powf(base, fp.gamma);
        dm = fp.gamma == 0.0f ? 0.0f : fp.gamma * powf(base, fp.gamma - 1.0f);      // (torch: no gradient through x ** 0)
    }
    const float a_t = fp.alpha >= 0.0f ? (positive ? fp.alpha : 1.0f - fp.alpha) : 1.0f;
    // d ce / d x = p - t;  d base / d x = -(2 t - 1) p (1 - p)
    const float dbase = positive ? -(p * q) : p * q;
    *grad = a_t * ((p - (positive ? 1.0f : 0.0f)) * m + ce * (dm * dbase)) * fp.scale;
    return a_t * (ce * m);
}

// Elementwise over the [rows, C] logits, four consecutive elements per thread and trip (one 16-byte load, one
// 16-byte store); the row of an element is looked up in the class map.
template <bool GAMMA2>
__global__ __launch_bounds__(CLS_THREADS) void focal_loss_kernel(const float *__restrict__ x, const int64_t *__restrict__ cls,
                                                                 const unsigned n, const unsigned C, const FocalParams fp,
                                                                 float *__restrict__ grad, double *__restrict__ partials,
                                                                 int32_t *__restrict__ status)
{
    __shared__ double lds[CLS_THREADS / 64];
    float acc = 0.0f;
    bool bad = false;
    const unsigned chunks = (n + 3) >> 2;
    for (unsigned i = blockIdx.x * CLS_THREADS + threadIdx.x; i < chunks; i += gridDim.x * CLS_THREADS) {
        const unsigned e0 = i << 2;
        const bool whole = e0 + 4 <= n;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (whole) {
            const float4 t = reinterpret_cast<const float4 *>(x)[i];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            for (unsigned k = 0; e0 + k < n; ++k) v[k] = x[e0 + k];
        }
        unsigned row = e0 / C, c = e0 - row * C;
        int64_t label = cls[row];
        float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float part = 0.0f;
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            if (e0 + k < n) {
                const bool ignored = fp.use_ignore && label == fp.ignore_index;
                if (!ignored) {
                    if (label < 0 || label > (int64_t)C) bad = true;       // one_hot(targets, n_cls + 1) refuses these
                    part += focal_term<GAMMA2>(v[k], label == (int64_t)c, fp, &g[k]);
                }
                if (++c == C && e0 + k + 1 < n) {
                    c = 0;
                    label = cls[++row];
                }
            }
        }
        acc += part;
        if (grad) {
            if (whole) {
                reinterpret_cast<float4 *>(grad)[i] = make_float4(g[0], g[1], g[2], g[3]);
            } else {
                for (unsigned k = 0; e0 + k < n; ++k) grad[e0 + k] = g[k];
            }
        }
    }
    if (bad) dm::raise_status(status, DM_ST_LABEL_RANGE);
    const double total = block_sum((double)acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// Weighted cross entropy (nn.CrossEntropyLoss(weight), reduction 'mean'): L lanes share a row (L a power of two up to
// 64, consecutive lanes read consecutive logits).  Three passes over the row -- maximum, sum of exponentials,
// gradient -- of which the second and third find the row in the cache.
__global__ __launch_bounds__(CLS_THREADS) void cross_entropy_kernel(const float *__restrict__ x, const int64_t *__restrict__ cls,
                                                                    const float *__restrict__ weight, const unsigned rows,
                                                                    const unsigned C, const unsigned L,
                                                                    const int64_t ignore_index, float *__restrict__ grad,
                                                                    double *__restrict__ partials, int32_t *__restrict__ status)
{
    __shared__ double lds[CLS_THREADS / 64];
    const unsigned sub = threadIdx.x & (L - 1);
    const unsigned rows_per_block = CLS_THREADS / L;
    float num = 0.0f, den = 0.0f;
    bool bad = false;
    // every lane of a row group runs the same trips (the shuffles below need all of its lanes)
    for (unsigned r0 = blockIdx.x * rows_per_block; r0 < rows; r0 += gridDim.x * rows_per_block) {
        const unsigned row = r0 + threadIdx.x / L;
        const bool live = row < rows;
        const float *lg = x + (size_t)(live ? row : 0) * C;
        float mx = -__builtin_huge_valf();
        for (unsigned c = sub; c < C; c += L) mx = fmaxf(mx, lg[c]);
        for (unsigned off = L >> 1; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.0f;
        for (unsigned c = sub; c < C; c += L) sum += expf(lg[c] - mx);
        for (unsigned off = L >> 1; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        const float lse = mx + logf(sum);
        if (!live) continue;
        const int64_t y = cls[row];
        float w = 0.0f;
        if (y != ignore_index) {
            if (y < 0 || y >= (int64_t)C) bad = true;
            else w = weight ? weight[y] : 1.0f;
        }
        if (sub == 0 && w != 0.0f) {
            num += w * (lse - lg[y]);
            den += w;
        }
        if (grad) {
            float *gr = grad + (size_t)row * C;
            for (unsigned c = sub; c < C; c += L)
                gr[c] = w == 0.0f ? 0.0f : w * (expf(lg[c] - lse) - ((int64_t)c == y ? 1.0f : 0.0f));
        }
    }
    if (bad) dm::raise_status(status, DM_ST_LABEL_RANGE);
    const double a = block_sum((double)num, lds);
    const double b = block_sum((double)den, lds);
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
    }
}

// The workgroups' partial sums in index order.  FOCAL: loss = scale * sum.  Otherwise loss = numerator / denominator,
// loss_out[1] = 1 / denominator (what the backward pass scales the stored gradient by).
template <bool FOCAL>
__global__ __launch_bounds__(CLS_THREADS) void cls_loss_finish_kernel(const double *__restrict__ partials, const int blocks,
                                                                      const float scale, float *__restrict__ loss_out)
{
    __shared__ double lds[CLS_THREADS / 64];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < blocks; i += CLS_THREADS) {
        if (FOCAL) a += partials[i];
        else { a += partials[2 * i]; b += partials[2 * i + 1]; }
    }
    a = block_sum(a, lds);
    if (!FOCAL) b = block_sum(b, lds);
    if (threadIdx.x == 0) {
        if (FOCAL) {
            loss_out[0] = (float)(a * (double)scale);
        } else {
            loss_out[0] = (float)(a / b);
            loss_out[1] = (float)(1.0 / b);
        }
    }
}

int check_common(const float *logits, const int64_t *target_cls, int64_t rows, int C, float *loss_out, double *partials,
                 int32_t *status)
{
    if (rows < 0 || C < 1) return DM_E_INVALID_ARG;
    if (!loss_out || !partials || !status) return DM_E_INVALID_ARG;
    if (rows > 0 && (!logits || !target_cls)) return DM_E_INVALID_ARG;
    if ((unsigned long long)rows * (unsigned long long)C >= (1ull << 31)) return DM_E_UNSUPPORTED;
    return DM_OK;
}

}  // namespace

extern "C" int dm_focal_loss_f32(const float *logits, const int64_t *target_cls, int64_t rows, int C, float alpha,
                                 float gamma, float scale, int64_t ignore_index, int use_ignore, float *loss_out,
                                 float *grad_out, double *partials, int32_t *status, void *stream)
{
    int rc = check_common(logits, target_cls, rows, C, loss_out, partials, status);
    if (rc != DM_OK) return rc;
    if (((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(grad_out)) & 15) != 0) return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned n = (unsigned)(rows * C);
    const unsigned chunks = (n + 3) >> 2;
    const int blocks = (int)std::max(1u, std::min((unsigned)CLS_MAX_BLOCKS, (chunks + CLS_THREADS - 1) / CLS_THREADS));
    const FocalParams fp{alpha, gamma, scale, ignore_index, use_ignore};
    if (gamma == 2.0f)
        hipLaunchKernelGGL(focal_loss_kernel<true>, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, n, (unsigned)C,
                           fp, grad_out, partials, status);
    else
        hipLaunchKernelGGL(focal_loss_kernel<false>, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, n, (unsigned)C,
                           fp, grad_out, partials, status);
    rc = dm::check_hip(hipGetLastError(), "focal_loss_kernel launch");
    if (rc != DM_OK) return rc;
    hipLaunchKernelGGL(cls_loss_finish_kernel<true>, dim3(1), dim3(CLS_THREADS), 0, st, partials, blocks, scale, loss_out);
    return dm::check_hip(hipGetLastError(), "cls_loss_finish_kernel launch");
}

extern "C" int dm_cross_entropy_f32(const float *logits, const int64_t *target_cls, const float *weight, int64_t rows,
                                    int C, int64_t ignore_index, float *loss_out, float *grad_out, double *partials,
                                    int32_t *status, void *stream)
{
    int rc = check_common(logits, target_cls, rows, C, loss_out, partials, status);
    if (rc != DM_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned L = 1;                                   // lanes per row: about four logits per lane, at most a wavefront
    while (L < 64 && L * 4 < (unsigned)C) L <<= 1;
    const unsigned rows_per_block = CLS_THREADS / L;
    const unsigned want = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
    const int blocks = (int)std::max(1u, std::min((unsigned)CLS_MAX_BLOCKS, want));
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, weight, (unsigned)rows,
                       (unsigned)C, L, ignore_index, grad_out, partials, status);
    rc = dm::check_hip(hipGetLastError(), "cross_entropy_kernel launch");
    if (rc != DM_OK) return rc;
    hipLaunchKernelGGL(cls_loss_finish_kernel<false>, dim3(1), dim3(CLS_THREADS), 0, st, partials, blocks, 1.0f, loss_out);
    return dm::check_hip(hipGetLastError(), "cls_loss_finish_kernel launch");
}
