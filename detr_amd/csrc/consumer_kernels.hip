// Consumers of the matching indices, for gfx950 (MI355X): what the reference's criterion does
// with the (src, tgt) tuples right after the matcher returns them
// (/root/reference/models/detr/loss.py:71-75 gather indices, :97-114 L1 + GIoU loss of the
// matched pairs, :121-125 dense class map).  The reference re-uploads the CPU indices and
// launches about fifteen small torch kernels per output set -- and builds an [N, N] GIoU matrix to
// read its diagonal; here the indices never leave the device and every output set of the step
// is handled by one launch per operation, O(number of pairs).
//
// These are tiny, latency-bound launches (a few thousand pairs): no tiling, no MFMA; the
// reductions have a fixed order, so results are reproducible run to run.
#include <algorithm>

#include "dm_common.h"

namespace {

struct DmIndexSetArray {
    DmIndexSet s[DM_MAX_SETS_PER_LAUNCH];
};

// image of packed slot k: the b with tgt_off[b] <= k < tgt_off[b + 1]
__device__ __forceinline__ int image_of_slot(const int32_t *__restrict__ tgt_off, int B, int k)
{
    int lo = 0, hi = B;   // invariant: tgt_off[lo] <= k < tgt_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tgt_off[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}

struct Corners {
    float x0, y0, x1, y1;
};

__device__ __forceinline__ Corners corners(const float4 b)   // box_ops.py:17-21
{
    return {b.x - 0.5f * b.z, b.y - 0.5f * b.w, b.x + 0.5f * b.z, b.y + 0.5f * b.w};
}

// share of the gradient of max(a, b) / min(a, b) that reaches a (torch splits it evenly at ties)
__device__ __forceinline__ float hi_share(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); }
__device__ __forceinline__ float lo_share(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); }

// GIoU of one (prediction, target) pair -- the diagonal entry of box_ops.py:48-69 -- and the
// gradient of (1 - giou) w.r.t. the cxcywh prediction.
__device__ __forceinline__ float pair_giou_loss(const float4 pb, const float4 gb, float grad[4])
{
    const Corners a = corners(pb), g = corners(gb);
    const float aw = a.x1 - a.x0, ah = a.y1 - a.y0;
    const float area_a = aw * ah;
    const float area_g = (g.x1 - g.x0) * (g.y1 - g.y0);
    const float dx = fminf(a.x1, g.x1) - fmaxf(a.x0, g.x0);
    const float dy = fminf(a.y1, g.y1) - fmaxf(a.y0, g.y0);
    const float wx = fmaxf(dx, 0.0f), wy = fmaxf(dy, 0.0f);
    const float inter = wx * wy;
    const float uni = (area_a + area_g) - inter;
    const float iou = inter / uni;
    const float edx = fmaxf(a.x1, g.x1) - fminf(a.x0, g.x0);
    const float edy = fmaxf(a.y1, g.y1) - fminf(a.y0, g.y0);
    const float ewx = fmaxf(edx, 0.0f), ewy = fmaxf(edy, 0.0f);
    const float earea = ewx * ewy;
    const float giou = iou - (earea - uni) / earea;

    // d giou / d(inter, union, enclosing area)
    const float g_union = -inter / (uni * uni) + 1.0f / earea;
    const float g_inter = 1.0f / uni - g_union;
    const float g_earea = -uni / (earea * earea);
    const float g_dx = dx >= 0.0f ? g_inter * wy : 0.0f;
    const float g_dy = dy >= 0.0f ? g_inter * wx : 0.0f;
    const float g_edx = edx >= 0.0f ? g_earea * ewy : 0.0f;
    const float g_edy = edy >= 0.0f ? g_earea * ewx : 0.0f;
    const float g_x1 = g_dx * lo_share(a.x1, g.x1) + g_edx * hi_share(a.x1, g.x1) + g_union * ah;
    const float g_x0 = -g_dx * hi_share(a.x0, g.x0) - g_edx * lo_share(a.x0, g.x0) - g_union * ah;
    const float g_y1 = g_dy * lo_share(a.y1, g.y1) + g_edy * hi_share(a.y1, g.y1) + g_union * aw;
    const float g_y0 = -g_dy * hi_share(a.y0, g.y0) - g_edy * lo_share(a.y0, g.y0) - g_union * aw;
    // loss = 1 - giou
    grad[0] = -(g_x0 + g_x1);
    grad[1] = -(g_y0 + g_y1);
    grad[2] = -0.5f * (g_x1 - g_x0);
    grad[3] = -0.5f * (g_y1 - g_y0);
    return 1.0f - giou;
}

__device__ __forceinline__ float sign_of(float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); }

constexpr int LOSS_THREADS = 1024;

// One workgroup per output set: sums of |pred - target| and of (1 - giou) over the set's matched
// pairs (loss.py:105-112 before the division by num_boxes), and the per-pair gradients.
__global__ __launch_bounds__(LOSS_THREADS) void pair_loss_kernel(const DmIndexSetArray sets, const int B, const int M,
                                                                 const float *__restrict__ tgt_boxes,
                                                                 const int32_t *__restrict__ tgt_off,
                                                                 const int64_t *__restrict__ src,
                                                                 const int64_t *__restrict__ tgt,
                                                                 float *__restrict__ sums, float *__restrict__ pair_grad)
{
    __shared__ float red[2][LOSS_THREADS / 64];
    const DmIndexSet &set = sets.s[blockIdx.x];
    const int Q = set.Q;
    const float4 *pred = reinterpret_cast<const float4 *>(set.pred_boxes);
    const size_t image_rows = set.boxes_stride ? (size_t)(set.boxes_stride >> 2) : (size_t)Q;      // float4 rows from image to image
    const float4 *gts = reinterpret_cast<const float4 *>(tgt_boxes);
    float l1 = 0.0f, gl = 0.0f;
    for (int k = threadIdx.x; k < M; k += LOSS_THREADS) {
        const int b = image_of_slot(tgt_off, B, k);
        const int first = tgt_off[b];
        const int n = min(Q, tgt_off[b + 1] - first);
        float out[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (k - first < n) {
            const int64_t q = src[set.out_off + k];
            const int64_t t = first + tgt[set.out_off + k];
            const float4 pb = pred[(size_t)b * image_rows + q];
            const float4 gb = gts[t];
            const float d0 = pb.x - gb.x, d1 = pb.y - gb.y, d2 = pb.z - gb.z, d3 = pb.w - gb.w;
            l1 += ((fabsf(d0) + fabsf(d1)) + fabsf(d2)) + fabsf(d3);
            out[0] = sign_of(d0); out[1] = sign_of(d1); out[2] = sign_of(d2); out[3] = sign_of(d3);
            gl += pair_giou_loss(pb, gb, out + 4);
        }
        if (pair_grad) {
            float4 *dst = reinterpret_cast<float4 *>(pair_grad + ((size_t)blockIdx.x * M + k) * 8);
            dst[0] = make_float4(out[0], out[1], out[2], out[3]);
            dst[1] = make_float4(out[4], out[5], out[6], out[7]);
        }
    }
    // fixed-order reduction: butterflies inside the wave, then the waves in index order
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        l1 += __shfl_xor(l1, off);
        gl += __shfl_xor(gl, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = l1; red[1][wave] = gl; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, c = 0.0f;
        for (int w = 0; w < LOSS_THREADS / 64; ++w) { a += red[0][w]; c += red[1][w]; }
        sums[2 * blockIdx.x] = a;
        sums[2 * blockIdx.x + 1] = c;
    }
}

// One workgroup per (image, output set): the slot of every matched query is scattered into an
// LDS row, then the whole [Q] row of the output is written once, coalesced.
template <bool CLASSES>
__global__ __launch_bounds__(256) void scatter_rows_kernel(const DmIndexSetArray sets, const int M,
                                                           const int32_t *__restrict__ tgt_off,
                                                           const int64_t *__restrict__ src,
                                                           const int64_t *__restrict__ tgt,
                                                           const float *__restrict__ pair_grad,
                                                           const float *__restrict__ grad_sums,
                                                           const int64_t *__restrict__ tgt_labels,
                                                           const int64_t no_object)
{
    extern __shared__ int slot[];   // [Q]: packed slot of the pair that uses query q, or -1
    const DmIndexSet &set = sets.s[blockIdx.y];
    const int b = blockIdx.x, Q = set.Q;
    const int first = tgt_off[b];
    const int n = min(Q, tgt_off[b + 1] - first);
    for (int q = threadIdx.x; q < Q; q += blockDim.x) slot[q] = -1;
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += blockDim.x) slot[src[set.out_off + first + k]] = first + k;
    __syncthreads();
    if (CLASSES) {
        int64_t *row = set.target_cls + (size_t)b * Q;
        for (int q = threadIdx.x; q < Q; q += blockDim.x) {
            const int k = slot[q];
            row[q] = k < 0 ? no_object : tgt_labels[first + tgt[set.out_off + k]];
        }
    } else {
        float4 *row = reinterpret_cast<float4 *>(set.grad_boxes) + (size_t)b * Q;
        const float w_l1 = grad_sums[2 * blockIdx.y], w_giou = grad_sums[2 * blockIdx.y + 1];
        for (int q = threadIdx.x; q < Q; q += blockDim.x) {
            const int k = slot[q];
            float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k >= 0) {
                const float4 *pg = reinterpret_cast<const float4 *>(pair_grad + ((size_t)blockIdx.y * M + k) * 8);
                const float4 a = pg[0], c = pg[1];
                g = make_float4(w_l1 * a.x + w_giou * c.x, w_l1 * a.y + w_giou * c.y, w_l1 * a.z + w_giou * c.z,
                                w_l1 * a.w + w_giou * c.w);
            }
            row[q] = g;
        }
    }
}

// The same two sums for an explicit list of (image, query, packed target) pairs -- indices that did not come from the
// matcher: the denoising groups of DN-DETR / DINO / Co-DETR bring theirs along (several groups of queries per target),
// an auxiliary assigner may pair one prediction with several targets.  Any number of workgroups; each leaves its two
// partial sums, which pair_list_finish_kernel adds in index order.
constexpr int PAIR_LIST_THREADS = 256;
constexpr int PAIR_LIST_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(PAIR_LIST_THREADS) void pair_list_loss_kernel(const float *__restrict__ pred_boxes, const size_t image_rows,
                                                                          const int B, const int Q, const int M,
                                                                          const float *__restrict__ tgt_boxes,
                                                                          const int64_t *__restrict__ pair_b,
                                                                          const int64_t *__restrict__ pair_q,
                                                                          const int64_t *__restrict__ pair_t, const int64_t N,
                                                                          float *__restrict__ pair_grad,
                                                                          double *__restrict__ partials, int32_t *__restrict__ status)
{
    __shared__ double red[2][PAIR_LIST_THREADS / 64];
    const float4 *pred = reinterpret_cast<const float4 *>(pred_boxes);
    const float4 *gts = reinterpret_cast<const float4 *>(tgt_boxes);
    float l1 = 0.0f, gl = 0.0f;
    bool bad = false;
    for (int64_t k = (int64_t)blockIdx.x * PAIR_LIST_THREADS + threadIdx.x; k < N; k += (int64_t)gridDim.x * PAIR_LIST_THREADS) {
        const int64_t b = pair_b[k], q = pair_q[k], t = pair_t[k];
        float out[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (b < 0 || b >= B || q < 0 || q >= Q || t < 0 || t >= M) {
            bad = true;                     // (the reference's gathers raise IndexError; the pair contributes nothing)
        } else {
            const float4 pb = pred[(size_t)b * image_rows + (size_t)q];
            const float4 gb = gts[t];
            const float d0 = pb.x - gb.x, d1 = pb.y - gb.y, d2 = pb.z - gb.z, d3 = pb.w - gb.w;
            l1 += ((fabsf(d0) + fabsf(d1)) + fabsf(d2)) + fabsf(d3);
            out[0] = sign_of(d0); out[1] = sign_of(d1); out[2] = sign_of(d2); out[3] = sign_of(d3);
            gl += pair_giou_loss(pb, gb, out + 4);
        }
        if (pair_grad) {
            float4 *dst = reinterpret_cast<float4 *>(pair_grad + (size_t)k * 8);
            dst[0] = make_float4(out[0], out[1], out[2], out[3]);
            dst[1] = make_float4(out[4], out[5], out[6], out[7]);
        }
    }
    if (bad) dm::raise_status(status, DM_ST_LABEL_RANGE);
    double a = (double)l1, c = (double)gl;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off);
        c += __shfl_xor(c, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = a; red[1][wave] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sc = 0.0;
        for (int w = 0; w < PAIR_LIST_THREADS / 64; ++w) { sa += red[0][w]; sc += red[1][w]; }
        partials[2 * blockIdx.x] = sa;
        partials[2 * blockIdx.x + 1] = sc;
    }
}

__global__ __launch_bounds__(64) void pair_list_finish_kernel(const double *__restrict__ partials, const int blocks,
                                                               float *__restrict__ sums)
{
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 64) { a += partials[2 * i]; c += partials[2 * i + 1]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        a += __shfl_xor(a, off);
        c += __shfl_xor(c, off);
    }
    if (threadIdx.x == 0) { sums[0] = (float)a; sums[1] = (float)c; }
}

// Backward of the pair list: grad_boxes[b][q] += w_l1 * d l1 + w_giou * d (1 - giou) for every pair.  A query may occur in
// several pairs, so the four components are added with hardware float atomics (the order of the additions, and with
// it the last bit of such a query's gradient, is then not fixed; a query in one pair gets exactly its value).
__global__ __launch_bounds__(PAIR_LIST_THREADS) void pair_list_backward_kernel(const float *__restrict__ pair_grad,
                                                                              const float *__restrict__ grad_sums,
                                                                              const int64_t *__restrict__ pair_b,
                                                                              const int64_t *__restrict__ pair_q, const int64_t N,
                                                                              const int B, const int Q, float *__restrict__ grad_boxes)
{
    const float w_l1 = grad_sums[0], w_giou = grad_sums[1];
    for (int64_t k = (int64_t)blockIdx.x * PAIR_LIST_THREADS + threadIdx.x; k < N; k += (int64_t)gridDim.x * PAIR_LIST_THREADS) {
        const int64_t b = pair_b[k], q = pair_q[k];
        if (b < 0 || b >= B || q < 0 || q >= Q) continue;
        const float4 *pg = reinterpret_cast<const float4 *>(pair_grad + (size_t)k * 8);
        const float4 a = pg[0], c = pg[1];
        float *dst = grad_boxes + ((size_t)b * Q + (size_t)q) * 4;
        unsafeAtomicAdd(dst + 0, w_l1 * a.x + w_giou * c.x);
        unsafeAtomicAdd(dst + 1, w_l1 * a.y + w_giou * c.y);
        unsafeAtomicAdd(dst + 2, w_l1 * a.z + w_giou * c.z);
        unsafeAtomicAdd(dst + 3, w_l1 * a.w + w_giou * c.w);
    }
}

int check_sets(const DmIndexSet *sets, int S, int B, int M)
{
    if (S < 0 || B < 0 || M < 0) return DM_E_INVALID_ARG;
    if (S > 0 && !sets) return DM_E_INVALID_ARG;
    if (B > 65535) return DM_E_UNSUPPORTED;
    for (int s = 0; s < S; ++s)
        if (sets[s].Q < 1) return DM_E_INVALID_ARG;
    return DM_OK;
}

}  // namespace

namespace dm {
int check_hip(hipError_t e, const char *what);
}

extern "C" int dm_pair_loss_f32(const DmIndexSet *sets, int S, int B, int M, const float *tgt_boxes,
                                const int32_t *tgt_off, const int64_t *src, const int64_t *tgt, float *sums,
                                float *pair_grad, void *stream)
{
    int rc = check_sets(sets, S, B, M);
    if (rc != DM_OK) return rc;
    if (S == 0) return DM_OK;
    if (!sums || !tgt_off || (M > 0 && (!tgt_boxes || !src || !tgt))) return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s0 = 0; s0 < S; s0 += DM_MAX_SETS_PER_LAUNCH) {
        const int n = std::min(DM_MAX_SETS_PER_LAUNCH, S - s0);
        DmIndexSetArray arr;
        for (int i = 0; i < n; ++i) {
            if (!sets[s0 + i].pred_boxes || !dm::index_set_boxes_ok(sets[s0 + i])) return DM_E_INVALID_ARG;
            arr.s[i] = sets[s0 + i];
        }
        hipLaunchKernelGGL(pair_loss_kernel, dim3(n), dim3(LOSS_THREADS), 0, st, arr, B, M, tgt_boxes, tgt_off, src, tgt,
                           sums + 2 * s0, pair_grad ? pair_grad + (size_t)s0 * M * 8 : nullptr);
        rc = dm::check_hip(hipGetLastError(), "pair_loss_kernel launch");
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}

namespace {
template <bool CLASSES>
int launch_scatter(const DmIndexSet *sets, int S, int B, int M, const int32_t *tgt_off, const int64_t *src,
                   const int64_t *tgt, const float *pair_grad, const float *grad_sums, const int64_t *tgt_labels,
                   int64_t no_object, hipStream_t st)
{
    int rc = check_sets(sets, S, B, M);
    if (rc != DM_OK) return rc;
    if (S == 0 || B == 0) return DM_OK;
    if (!tgt_off || (M > 0 && (!src || !tgt))) return DM_E_INVALID_ARG;
    auto kernel = scatter_rows_kernel<CLASSES>;
    for (int s0 = 0; s0 < S; s0 += DM_MAX_SETS_PER_LAUNCH) {
        const int n = std::min(DM_MAX_SETS_PER_LAUNCH, S - s0);
        DmIndexSetArray arr;
        int max_q = 0;
        for (int i = 0; i < n; ++i) {
            arr.s[i] = sets[s0 + i];
            if (CLASSES ? !arr.s[i].target_cls : !arr.s[i].grad_boxes) return DM_E_INVALID_ARG;
            max_q = std::max(max_q, arr.s[i].Q);
        }
        const size_t lds = (size_t)max_q * sizeof(int);
        if (lds > 160 * 1024) return DM_E_UNSUPPORTED;
        if (lds > 64 * 1024) {
            rc = dm::allow_lds(reinterpret_cast<const void *>(kernel), lds, "hipFuncSetAttribute(scatter_rows_kernel)");
            if (rc != DM_OK) return rc;
        }
        hipLaunchKernelGGL(kernel, dim3(B, n), dim3(256), lds, st, arr, M, tgt_off, src, tgt,
                           pair_grad ? pair_grad + (size_t)s0 * M * 8 : nullptr,
                           grad_sums ? grad_sums + 2 * s0 : nullptr, tgt_labels, no_object);
        rc = dm::check_hip(hipGetLastError(), "scatter_rows_kernel launch");
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}
}  // namespace

extern "C" int dm_pair_loss_backward_f32(const DmIndexSet *sets, int S, int B, int M, const int32_t *tgt_off,
                                         const int64_t *src, const int64_t *tgt, const float *pair_grad,
                                         const float *grad_sums, void *stream)
{
    if (!pair_grad || !grad_sums) return DM_E_INVALID_ARG;
    return launch_scatter<false>(sets, S, B, M, tgt_off, src, tgt, pair_grad, grad_sums, nullptr, 0,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int dm_target_classes_i64(const DmIndexSet *sets, int S, int B, int M, const int64_t *tgt_labels,
                                     const int32_t *tgt_off, const int64_t *src, const int64_t *tgt,
                                     int64_t no_object, void *stream)
{
    if (M > 0 && !tgt_labels) return DM_E_INVALID_ARG;
    return launch_scatter<true>(sets, S, B, M, tgt_off, src, tgt, nullptr, nullptr, tgt_labels, no_object,
                                static_cast<hipStream_t>(stream));
}

extern "C" int dm_pair_list_loss_f32(const float *pred_boxes, int boxes_stride, int B, int Q, const float *tgt_boxes, int M,
                                     const int64_t *pair_b, const int64_t *pair_q, const int64_t *pair_t, int64_t N,
                                     float *sums, float *pair_grad, double *partials, int32_t *status, void *stream)
{
    if (B < 0 || Q < 1 || M < 0 || N < 0 || !sums || !partials || !status) return DM_E_INVALID_ARG;
    if (N > 0 && (!pred_boxes || !tgt_boxes || !pair_b || !pair_q || !pair_t)) return DM_E_INVALID_ARG;
    if (boxes_stride != 0 && (boxes_stride < Q * 4 || (boxes_stride & 3) != 0)) return DM_E_INVALID_ARG;
    if (((reinterpret_cast<uintptr_t>(pred_boxes) | reinterpret_cast<uintptr_t>(tgt_boxes) |
          reinterpret_cast<uintptr_t>(pair_grad)) & 15) != 0)
        return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t image_rows = boxes_stride ? (size_t)(boxes_stride >> 2) : (size_t)Q;
    const int64_t want = (N + PAIR_LIST_THREADS - 1) / PAIR_LIST_THREADS;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(PAIR_LIST_MAX_BLOCKS, want));
    static_assert(2 * PAIR_LIST_MAX_BLOCKS <= DM_CLS_LOSS_PARTIALS, "the partial sums share the classification loss's workspace size");
    hipLaunchKernelGGL(pair_list_loss_kernel, dim3(blocks), dim3(PAIR_LIST_THREADS), 0, st, pred_boxes, image_rows, B, Q, M,
                       tgt_boxes, pair_b, pair_q, pair_t, N, pair_grad, partials, status);
    int rc = dm::check_hip(hipGetLastError(), "pair_list_loss_kernel launch");
    if (rc != DM_OK) return rc;
    hipLaunchKernelGGL(pair_list_finish_kernel, dim3(1), dim3(64), 0, st, partials, blocks, sums);
    return dm::check_hip(hipGetLastError(), "pair_list_finish_kernel launch");
}

extern "C" int dm_pair_list_backward_f32(const float *pair_grad, const float *grad_sums, const int64_t *pair_b,
                                         const int64_t *pair_q, int64_t N, int B, int Q, float *grad_boxes, void *stream)
{
    if (B < 0 || Q < 1 || N < 0) return DM_E_INVALID_ARG;
    if (N == 0) return DM_OK;
    if (!pair_grad || !grad_sums || !pair_b || !pair_q || !grad_boxes) return DM_E_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(pair_grad) & 15) != 0) return DM_E_INVALID_ARG;
    const int64_t want = (N + PAIR_LIST_THREADS - 1) / PAIR_LIST_THREADS;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(PAIR_LIST_MAX_BLOCKS, want));
    hipLaunchKernelGGL(pair_list_backward_kernel, dim3(blocks), dim3(PAIR_LIST_THREADS), 0, static_cast<hipStream_t>(stream),
                       pair_grad, grad_sums, pair_b, pair_q, N, B, Q, grad_boxes);
    return dm::check_hip(hipGetLastError(), "pair_list_backward_kernel launch");
}
