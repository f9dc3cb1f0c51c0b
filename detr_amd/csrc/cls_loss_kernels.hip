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

// What the class map says about a row, in 32 bits: the class (0 .. C, C = no object), or one of two codes.
constexpr int ROW_IGNORED = -1;     // contributes nothing
constexpr int ROW_BAD = -2;         // a class the reference's one_hot refuses: contributes nothing, and is reported

__device__ __forceinline__ int row_code(int64_t label, unsigned C, const FocalParams &fp)
{
    if (fp.use_ignore && label == fp.ignore_index) return ROW_IGNORED;
    if (label < 0 || label > (int64_t)C) return ROW_BAD;
    return (int)label;
}

// One logit x of a row whose class is `positive` (t = 1) or not (t = 0): sigmoid_focal_loss (loss.py:233-240) and
// d loss / d x as torch autograd derives it from those operations.  With z = -x for the positive logit and z = x for
// the others, the reference's terms are functions of z alone: 1 - p_t = sigmoid(z) and
// binary_cross_entropy_with_logits = softplus(z), so that
//     loss = alpha_t * softplus(z) * sigmoid(z)^gamma,
//     d loss / d z = alpha_t * sigmoid(z)^(gamma - 1) * sigmoid(z) * (sigmoid(z) + gamma * softplus(z) * (1 - sigmoid(z)))
// and d loss / d x is that with the sign of dz/dx.  One hardware exponential serves both: with e = exp(-|z|),
// sigmoid(z) = 1 / (1 + e) or e / (1 + e) (no cancellation in 1 - sigmoid) and softplus(z) = max(z, 0) + log1p(e).
// (v_exp_f32 / v_log_f32 / v_rcp_f32 are good to about 1e-6 relative, an order of magnitude inside the tolerance the
// loss is held to; with the library's expf / log1pf / division the kernel is bound by their instruction count, not by
// memory.)  No branches: the terms of a quad run interleaved.
template <bool GAMMA2>
__device__ __forceinline__ float focal_term(float x, bool positive, const FocalParams &fp, float *grad)
{
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const unsigned flip = positive ? 0x80000000u : 0u;
    const float z = __uint_as_float(__float_as_uint(x) ^ flip);
    const float e = __builtin_amdgcn_exp2f(-fabsf(z) * LOG2E);
    const float one_e = 1.0f + e;
    const float r = __builtin_amdgcn_rcpf(one_e);
    const float er = e * r;
    const bool up = z >= 0.0f;
    const float sg = up ? r : er;                                             // sigmoid(z) = 1 - p_t
    const float co = up ? er : r;                                             // 1 - sigmoid(z)
    // log1p(e) = log(1 + e) + the part of e that the rounding of 1 + e dropped, over 1 + e: exact to first order for
    // every e (a tiny e gives log(1) + e), without a branch
    const float l1p = __builtin_amdgcn_logf(one_e) * LN2 + (e - (one_e - 1.0f)) * r;
    const float sp = __builtin_fmaxf(z, 0.0f) + l1p;                          // softplus(z)
    const float a_t = fp.alpha >= 0.0f ? (positive ? fp.alpha : 1.0f - fp.alpha) : 1.0f;
    float m, slope;                                                           // sigmoid^gamma, d (softplus * sigmoid^gamma) / dz
    if (GAMMA2) {
        m = sg * sg;
        const float u = sp * co;
        slope = m * (sg + (u + u));
    } else {
        m = powf(sg, fp.gamma);
        // (torch: no gradient through x ** 0)
        const float dm = fp.gamma == 0.0f ? 0.0f : fp.gamma * powf(sg, fp.gamma - 1.0f);
        slope = sg * m + sp * (dm * (sg * co));
    }
    *grad = __uint_as_float(__float_as_uint((a_t * fp.scale) * slope) ^ flip);
    return a_t * (sp * m);
}

// Elementwise over the [rows, C] logits in quads of four consecutive elements (one 16-byte load, one 16-byte store).
// With two or more classes a quad lies in at most two rows, whose classes are requested together with the logits;
// ONE_CLASS: every element is a row of its own.  A thread requests the logits and classes of FOCAL_QUADS quads before
// it computes on any of them (at 8 TB/s and ~2 us of memory latency some 16 MB have to be in flight, twice what one
// 16-byte load per lane of a full chip holds), and walks from quad to quad by additions: the kernel is close to being
// bound by its instruction count (about 50 per logit), a division per quad would show.  The up to three logits behind
// the last whole quad are handled by the first threads of the first workgroup.
constexpr int FOCAL_QUADS = 2;

template <bool GAMMA2, bool ONE_CLASS>
__global__ __launch_bounds__(CLS_THREADS) void focal_loss_kernel(const float *__restrict__ x, const int64_t *__restrict__ cls,
                                                                 const unsigned n, const unsigned C, const unsigned rows,
                                                                 const FocalParams fp, float *__restrict__ grad,
                                                                 double *__restrict__ partials, int32_t *__restrict__ status)
{
    __shared__ double lds[CLS_THREADS / 64];
    float acc = 0.0f;
    bool bad = false;
    const unsigned quads = n >> 2;                                   // whole quads
    const unsigned stride = gridDim.x * CLS_THREADS;
    // (row, column) of a quad's first element, advanced by `stride` quads at a time
    const unsigned step_row = (stride << 2) / C, step_col = (stride << 2) - step_row * C;      // wave-uniform
    unsigned i0 = blockIdx.x * CLS_THREADS + threadIdx.x;
    unsigned row = (i0 << 2) / C, col = (i0 << 2) - row * C;
    auto advance = [&](unsigned &r, unsigned &c) {
        c += step_col;
        const bool over = c >= C;
        c = over ? c - C : c;
        r += step_row + (over ? 1u : 0u);
    };
    for (; i0 < quads; i0 += FOCAL_QUADS * stride) {
        float v[FOCAL_QUADS][4];
        int64_t label[FOCAL_QUADS][ONE_CLASS ? 4 : 2];
        unsigned c0[FOCAL_QUADS];
        // requests
#pragma unroll
        for (int u = 0; u < FOCAL_QUADS; ++u) {
            const unsigned i = i0 + u * stride;
            c0[u] = col;
            if (i < quads) {
                const float4 t = reinterpret_cast<const float4 *>(x)[i];
                v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
                if (ONE_CLASS) {
#pragma unroll
                    for (unsigned k = 0; k < 4; ++k) label[u][k] = cls[(i << 2) + k];
                } else {
                    label[u][0] = cls[row];
                    label[u][1] = cls[min(row + 1, rows - 1)];
                }
            }
            if (!ONE_CLASS) advance(row, col);
        }
        // arithmetic
#pragma unroll
        for (int u = 0; u < FOCAL_QUADS; ++u) {
            const unsigned i = i0 + u * stride;
            if (i < quads) {
                int code[ONE_CLASS ? 4 : 2];
#pragma unroll
                for (int j = 0; j < (ONE_CLASS ? 4 : 2); ++j) code[j] = row_code(label[u][j], C, fp);
                float g[4];
#pragma unroll
                for (unsigned k = 0; k < 4; ++k) {
                    const bool wrapped = !ONE_CLASS && c0[u] + k >= C;              // (c0 + 3 < 2 C for C >= 2)
                    const int row_class = ONE_CLASS ? code[k] : (wrapped ? code[1] : code[0]);
                    const unsigned column = ONE_CLASS ? 0u : (wrapped ? c0[u] + k - C : c0[u] + k);
                    const float term = focal_term<GAMMA2>(v[u][k], row_class == (int)column, fp, &g[k]);
                    const bool counts = row_class >= 0;
                    acc += counts ? term : 0.0f;
                    g[k] = counts ? g[k] : 0.0f;
                    bad |= row_class == ROW_BAD;
                }
                if (grad) reinterpret_cast<float4 *>(grad)[i] = make_float4(g[0], g[1], g[2], g[3]);
            }
        }
    }
    // the logits behind the last whole quad
    const unsigned e = (quads << 2) + threadIdx.x;
    if (blockIdx.x == 0 && e < n) {
        const unsigned r = e / C;
        const int row_class = row_code(cls[r], C, fp);
        float g;
        const float term = focal_term<GAMMA2>(x[e], row_class == (int)(e - r * C), fp, &g);
        if (row_class >= 0) acc += term;
        if (grad) grad[e] = row_class >= 0 ? g : 0.0f;
        bad |= row_class == ROW_BAD;
    }
    if (bad) dm::raise_status(status, DM_ST_LABEL_RANGE);
    const double total = block_sum((double)acc, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// Weighted cross entropy (nn.CrossEntropyLoss(weight), reduction 'mean'): L lanes share a row (L a power of two up to
// 64, consecutive lanes read consecutive logits).  KEEP: a lane's share of the row is at most four logits and stays in
// registers -- read once, exponentials taken once (softmax = exp(x - max) / sum serves loss and gradient); otherwise
// (more than 256 classes) the row is walked three times, the second and third time out of the cache.
template <bool KEEP>
__global__ __launch_bounds__(CLS_THREADS) void cross_entropy_kernel(const float *__restrict__ x, const int64_t *__restrict__ cls,
                                                                    const float *__restrict__ weight, const unsigned rows,
                                                                    const unsigned C, const unsigned L,
                                                                    const int64_t ignore_index, float *__restrict__ grad,
                                                                    double *__restrict__ partials, int32_t *__restrict__ status)
{
    __shared__ double lds[CLS_THREADS / 64];
    const unsigned sub = threadIdx.x & (L - 1);
    const unsigned rows_per_block = CLS_THREADS / L;
    constexpr float LOG2E = 1.4426950408889634f;
    float num = 0.0f, den = 0.0f;
    bool bad = false;
    // every lane of a row group runs the same trips (the shuffles below need all of its lanes)
    for (unsigned r0 = blockIdx.x * rows_per_block; r0 < rows; r0 += gridDim.x * rows_per_block) {
        const unsigned row = r0 + threadIdx.x / L;
        const bool live = row < rows;
        const float *lg = x + (size_t)(live ? row : 0) * C;
        float v[4];
        float mx = -__builtin_huge_valf();
        if (KEEP) {
#pragma unroll
            for (unsigned k = 0; k < 4; ++k) {
                const unsigned c = sub + k * L;
                v[k] = c < C ? lg[c] : -__builtin_huge_valf();
                mx = fmaxf(mx, v[k]);
            }
        } else {
            for (unsigned c = sub; c < C; c += L) mx = fmaxf(mx, lg[c]);
        }
        for (unsigned off = L >> 1; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.0f;
        if (KEEP) {
#pragma unroll
            for (unsigned k = 0; k < 4; ++k) {
                v[k] = __builtin_amdgcn_exp2f((v[k] - mx) * LOG2E);        // (a slot past the row: exp2(-inf) = 0)
                sum += v[k];
            }
        } else {
            for (unsigned c = sub; c < C; c += L) sum += __builtin_amdgcn_exp2f((lg[c] - mx) * LOG2E);
        }
        for (unsigned off = L >> 1; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        if (!live) continue;
        const int64_t y = cls[row];
        float w = 0.0f;
        if (y != ignore_index) {
            if (y < 0 || y >= (int64_t)C) bad = true;
            else w = weight ? weight[y] : 1.0f;
        }
        if (sub == 0 && w != 0.0f) {
            num += w * ((mx + __logf(sum)) - lg[y]);
            den += w;
        }
        if (grad) {
            float *gr = grad + (size_t)row * C;
            const float scale = w == 0.0f ? 0.0f : w * __frcp_rn(sum);
            if (KEEP) {
#pragma unroll
                for (unsigned k = 0; k < 4; ++k) {
                    const unsigned c = sub + k * L;
                    if (c < C) gr[c] = w == 0.0f ? 0.0f : v[k] * scale - ((int64_t)c == y ? w : 0.0f);
                }
            } else {
                for (unsigned c = sub; c < C; c += L)
                    gr[c] = w == 0.0f ? 0.0f
                                      : __builtin_amdgcn_exp2f((lg[c] - mx) * LOG2E) * scale - ((int64_t)c == y ? w : 0.0f);
            }
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

// The same for a class count that is a multiple of four (rows then start on 16-byte boundaries when the array does):
// a lane's share of the row is up to four float4, so that every memory instruction of a wavefront moves 1 KB
// instead of 256 bytes.  L lanes per row with 16 L >= C.
__global__ __launch_bounds__(CLS_THREADS) void cross_entropy_vec4_kernel(const float *__restrict__ x, const int64_t *__restrict__ cls,
                                                                         const float *__restrict__ weight, const unsigned rows,
                                                                         const unsigned C, const unsigned L,
                                                                         const int64_t ignore_index, float *__restrict__ grad,
                                                                         double *__restrict__ partials, int32_t *__restrict__ status)
{
    __shared__ double lds[CLS_THREADS / 64];
    const unsigned sub = threadIdx.x & (L - 1);
    const unsigned rows_per_block = CLS_THREADS / L;
    const unsigned C4 = C >> 2;
    constexpr float LOG2E = 1.4426950408889634f;
    const float ninf = -__builtin_huge_valf();
    float num = 0.0f, den = 0.0f;
    bool bad = false;
    for (unsigned r0 = blockIdx.x * rows_per_block; r0 < rows; r0 += gridDim.x * rows_per_block) {
        const unsigned row = r0 + threadIdx.x / L;
        const bool live = row < rows;
        const float4 *lg4 = reinterpret_cast<const float4 *>(x) + (size_t)(live ? row : 0) * C4;
        float4 v[4];
        float mx = ninf;
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            const unsigned c4 = sub + k * L;
            v[k] = c4 < C4 ? lg4[c4] : make_float4(ninf, ninf, ninf, ninf);
            mx = fmaxf(fmaxf(mx, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
        }
        for (unsigned off = L >> 1; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sum = 0.0f;
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            v[k].x = __builtin_amdgcn_exp2f((v[k].x - mx) * LOG2E);            // (a slot past the row: exp2(-inf) = 0)
            v[k].y = __builtin_amdgcn_exp2f((v[k].y - mx) * LOG2E);
            v[k].z = __builtin_amdgcn_exp2f((v[k].z - mx) * LOG2E);
            v[k].w = __builtin_amdgcn_exp2f((v[k].w - mx) * LOG2E);
            sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        for (unsigned off = L >> 1; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
        if (!live) continue;
        const int64_t y = cls[row];
        float w = 0.0f;
        if (y != ignore_index) {
            if (y < 0 || y >= (int64_t)C) bad = true;
            else w = weight ? weight[y] : 1.0f;
        }
        if (sub == 0 && w != 0.0f) {
            num += w * ((mx + __logf(sum)) - x[(size_t)row * C + y]);
            den += w;
        }
        if (grad) {
            float4 *gr = reinterpret_cast<float4 *>(grad) + (size_t)row * C4;
            const float scale = w == 0.0f ? 0.0f : w * __frcp_rn(sum);
            const unsigned hot = w == 0.0f ? 0xFFFFFFFFu : (unsigned)y;        // the logit that also gets -w
#pragma unroll
            for (unsigned k = 0; k < 4; ++k) {
                const unsigned c4 = sub + k * L;
                if (c4 < C4) {
                    float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (w != 0.0f) {
                        const unsigned c = c4 << 2;
                        g.x = v[k].x * scale - (c == hot ? w : 0.0f);
                        g.y = v[k].y * scale - (c + 1 == hot ? w : 0.0f);
                        g.z = v[k].z * scale - (c + 2 == hot ? w : 0.0f);
                        g.w = v[k].w * scale - (c + 3 == hot ? w : 0.0f);
                    }
                    gr[c4] = g;
                }
            }
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
    const unsigned quads = n >> 2;
    const unsigned per_block = CLS_THREADS * FOCAL_QUADS;
    const int blocks = (int)std::max(1u, std::min((unsigned)CLS_MAX_BLOCKS, (quads + per_block - 1) / per_block));
    const FocalParams fp{alpha, gamma, scale, ignore_index, use_ignore};
    auto kernel = gamma == 2.0f ? (C == 1 ? focal_loss_kernel<true, true> : focal_loss_kernel<true, false>)
                                : (C == 1 ? focal_loss_kernel<false, true> : focal_loss_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, n, (unsigned)C, (unsigned)rows, fp,
                       grad_out, partials, status);
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
    const bool vec4 = (C & 3) == 0 && C <= 1024 &&
                      ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(grad_out)) & 15) == 0;
    unsigned L = 1;                                   // lanes per row: about four logits (float4) per lane, at most a wavefront
    while (L < 64 && L * (vec4 ? 16 : 4) < (unsigned)C) L <<= 1;
    const unsigned rows_per_block = CLS_THREADS / L;
    const unsigned want = (unsigned)((rows + rows_per_block - 1) / rows_per_block);
    const int blocks = (int)std::max(1u, std::min((unsigned)CLS_MAX_BLOCKS, want));
    if (vec4)
        hipLaunchKernelGGL(cross_entropy_vec4_kernel, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, weight,
                           (unsigned)rows, (unsigned)C, L, ignore_index, grad_out, partials, status);
    else if ((unsigned)C <= 4 * L)
        hipLaunchKernelGGL(cross_entropy_kernel<true>, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, weight,
                           (unsigned)rows, (unsigned)C, L, ignore_index, grad_out, partials, status);
    else
        hipLaunchKernelGGL(cross_entropy_kernel<false>, dim3(blocks), dim3(CLS_THREADS), 0, st, logits, target_cls, weight,
                           (unsigned)rows, (unsigned)C, L, ignore_index, grad_out, partials, status);
    rc = dm::check_hip(hipGetLastError(), "cross_entropy_kernel launch");
    if (rc != DM_OK) return rc;
    hipLaunchKernelGGL(cls_loss_finish_kernel<false>, dim3(1), dim3(CLS_THREADS), 0, st, partials, blocks, 1.0f, loss_out);
    return dm::check_hip(hipGetLastError(), "cls_loss_finish_kernel launch");
}
