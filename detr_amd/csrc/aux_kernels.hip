// Co-DETR auxiliary label assigners for gfx950 (MI355X): ATSS and Faster-RCNN style assignment
// of predictions to targets (/root/reference/models/co_detr/auxiliary_head.py:22-60, 69-96).
// The reference walks images, targets and candidates in Python with an `.item()` per candidate;
// here one wavefront takes one (output set, target) -- or one (set, image, 64 predictions) for
// the negatives -- and all output sets of a step go out in one launch per assigner.
// Results are fixed-size tables (candidate lists + bit masks); turning them into the
// variable-length index lists of the reference's return value is left to the caller.
//
// Latency-bound integer/compare work on a few thousand boxes: no LDS tiling, no MFMA.
#include <algorithm>

#include "dm_common.h"

namespace {

struct DmIndexSetArray {
    DmIndexSet s[DM_MAX_SETS_PER_LAUNCH];
};

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

// torchvision.ops.box_iou for one pair: inter / (area_a + area_b - inter)
__device__ __forceinline__ float iou_of(const Corners a, const Corners b)
{
    const float area_a = (a.x1 - a.x0) * (a.y1 - a.y0);
    const float area_b = (b.x1 - b.x0) * (b.y1 - b.y0);
    const float w = fmaxf(fminf(a.x1, b.x1) - fmaxf(a.x0, b.x0), 0.0f);
    const float h = fmaxf(fminf(a.y1, b.y1) - fmaxf(a.y0, b.y0), 0.0f);
    const float inter = w * h;
    return inter / ((area_a + area_b) - inter);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

// ATSS, one wavefront per (set, target): auxiliary_head.py:35-57.
//   candidates: the k predictions of the image closest to the target's centre (Euclidean distance
//   of the centres, ascending; equal distances by prediction index);
//   positives: candidates with IoU >= mean + std (unbiased) of the k IoUs whose centre lies
//   inside the target box.
__global__ __launch_bounds__(64) void atss_kernel(const DmIndexSetArray sets, const int B, const int M, const int k,
                                                  const float *__restrict__ tgt_boxes,
                                                  const int32_t *__restrict__ tgt_off, int32_t *__restrict__ cand,
                                                  unsigned long long *__restrict__ pos_mask)
{
    const DmIndexSet &set = sets.s[blockIdx.y];
    const int t = blockIdx.x, lane = threadIdx.x, Q = set.Q;
    const int b = image_of_slot(tgt_off, B, t);
    const float4 gb = reinterpret_cast<const float4 *>(tgt_boxes)[t];
    const float4 *pred = dm::image_pred_boxes(set, b);

    // k rounds of "smallest key above the last one", key = (distance bits, prediction index):
    // distances are >= 0, so their bit patterns order like the values
    unsigned long long last = 0;
    int mine = -1;                      // lane j keeps candidate j
    for (int j = 0; j < k; ++j) {
        unsigned long long best = ~0ull;
        for (int q = lane; q < Q; q += 64) {
            const float4 pb = pred[q];
            const float dx = gb.x - pb.x, dy = gb.y - pb.y;
            const float d = sqrtf(dx * dx + dy * dy);
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)q;
            if ((j == 0 || key > last) && key < best) best = key;
        }
        best = wave_min_u64(best);
        last = best;
        if (lane == j) mine = (int)(unsigned)best;
    }

    const Corners g = corners(gb);
    float iou = 0.0f;
    float4 pb = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (lane < k) {
        pb = pred[mine];
        iou = iou_of(g, corners(pb));
    }
    // mean and unbiased standard deviation of the k IoUs, summed in candidate order
    float sum = 0.0f;
    for (int j = 0; j < k; ++j) sum += __shfl(iou, j);
    const float mean = sum / (float)k;
    const float dev = iou - mean;
    float ss = 0.0f;
    for (int j = 0; j < k; ++j) ss += __shfl(dev * dev, j);
    const float thr = mean + (k > 1 ? sqrtf(ss / (float)(k - 1)) : __builtin_nanf(""));
    const bool positive = lane < k && iou >= thr && g.x0 <= pb.x && pb.x <= g.x1 && g.y0 <= pb.y && pb.y <= g.y1;
    const unsigned long long mask = __ballot(positive);
    const size_t slot = (size_t)blockIdx.y * M + t;
    if (lane < k) cand[slot * k + lane] = mine;
    if (lane == 0) pos_mask[slot] = mask;
}

// Faster-RCNN positives, one wavefront per (set, target): auxiliary_head.py:87-93.
//   pos_bits[slot][w] bit i: IoU(target, prediction 64 w + i) > pos_threshold
//   best[slot]: the prediction with the largest IoU (the first one among equals)
__global__ __launch_bounds__(64) void rcnn_pos_kernel(const DmIndexSetArray sets, const int B, const int M, const int W,
                                                      const float pos_threshold, const float *__restrict__ tgt_boxes,
                                                      const int32_t *__restrict__ tgt_off,
                                                      unsigned long long *__restrict__ pos_bits,
                                                      int32_t *__restrict__ best)
{
    const DmIndexSet &set = sets.s[blockIdx.y];
    const int t = blockIdx.x, lane = threadIdx.x, Q = set.Q;
    const int b = image_of_slot(tgt_off, B, t);
    const Corners g = corners(reinterpret_cast<const float4 *>(tgt_boxes)[t]);
    const float4 *pred = dm::image_pred_boxes(set, b);
    const size_t slot = (size_t)blockIdx.y * M + t;
    float top = -1.0f;
    int top_q = 0x7fffffff;
    for (int w = 0; w < W; ++w) {
        const int q = w * 64 + lane;
        float iou = -1.0f;
        if (q < Q) iou = iou_of(g, corners(pred[q]));
        const unsigned long long bits = __ballot(q < Q && iou > pos_threshold);
        if (lane == 0) pos_bits[slot * W + w] = bits;
        if (q < Q && (iou > top || top_q == 0x7fffffff)) { top = iou; top_q = q; }
    }
    // largest IoU, smallest index among equals: order-preserving key (IoU >= 0 or the -1 filler)
    const unsigned hi = ~(__float_as_uint(fmaxf(top, 0.0f)));
    unsigned long long key = ((unsigned long long)hi << 32) | (unsigned)top_q;
    key = wave_min_u64(key);
    if (lane == 0) best[slot] = (int)(unsigned)key;
}

// Faster-RCNN negatives, one wavefront per (set, image, 64 predictions): auxiliary_head.py:79, 84.
//   neg_bits bit i: max over the image's targets of IoU(target, prediction) < neg_threshold;
//   every prediction of an image without targets is negative.
__global__ __launch_bounds__(64) void rcnn_neg_kernel(const DmIndexSetArray sets, const int B, const int W,
                                                      const float neg_threshold, const float *__restrict__ tgt_boxes,
                                                      const int32_t *__restrict__ tgt_off,
                                                      unsigned long long *__restrict__ neg_bits)
{
    const DmIndexSet &set = sets.s[blockIdx.z];
    const int w = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, Q = set.Q;
    const int q = w * 64 + lane;
    const int first = tgt_off[b], T = tgt_off[b + 1] - first;
    bool negative = q < Q;
    if (q < Q && T > 0) {
        const Corners p = corners(dm::image_pred_boxes(set, b)[q]);
        float top = -__builtin_huge_valf();
        for (int t = 0; t < T; ++t)
            top = fmaxf(top, iou_of(corners(reinterpret_cast<const float4 *>(tgt_boxes)[first + t]), p));
        negative = top < neg_threshold;
    }
    const unsigned long long bits = __ballot(negative);
    if (lane == 0) neg_bits[((size_t)blockIdx.z * B + b) * W + w] = bits;
}

// Faster-RCNN dense class targets, one wavefront per (set, image, 64 predictions): the [B, Q] map
// that the reference's calc_cls_loss_with_ignore builds from the positive pairs and the negative
// list (/root/reference/models/co_detr/loss.py:153-159), without materialising either list:
//   negative (best IoU < neg_threshold, or an image without targets)  -> n_cls
//   else positive for at least one target (IoU > pos_threshold, or the target's best prediction)
//        -> that target's label; the LAST such target wins, as the reference's CPU scatter does
//   else ignored                                                        -> -1
// `best` comes from rcnn_pos_kernel.  num_pos[set] (optional, zeroed by the caller) receives the number
// of positive (prediction, target) pairs, the reference's normaliser (loss.py:151).
__global__ __launch_bounds__(64) void rcnn_dense_kernel(const DmIndexSetArray sets, const int B, const int M,
                                                        const float pos_threshold, const float neg_threshold,
                                                        const float *__restrict__ tgt_boxes,
                                                        const int64_t *__restrict__ tgt_labels,
                                                        const int32_t *__restrict__ tgt_off,
                                                        const int32_t *__restrict__ best, const int64_t n_cls,
                                                        int32_t *__restrict__ num_pos)
{
    const DmIndexSet &set = sets.s[blockIdx.z];
    const int b = blockIdx.y, lane = threadIdx.x, Q = set.Q;
    const int q = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= Q) return;                     // whole wavefront past the end of a shorter set
    const int first = tgt_off[b], T = tgt_off[b + 1] - first;
    int64_t value = n_cls;
    int pairs = 0;
    if (q < Q && T > 0) {
        const Corners p = corners(dm::image_pred_boxes(set, b)[q]);
        const int32_t *bst = best + (size_t)blockIdx.z * M + first;
        float top = -__builtin_huge_valf();
        int winner = -1;
        for (int t = 0; t < T; ++t) {
            const float iou = iou_of(corners(reinterpret_cast<const float4 *>(tgt_boxes)[first + t]), p);
            top = fmaxf(top, iou);
            if (iou > pos_threshold || bst[t] == q) { winner = t; ++pairs; }
        }
        if (!(top < neg_threshold)) value = winner >= 0 ? tgt_labels[first + winner] : (int64_t)-1;
    }
    if (q < Q) set.target_cls[(size_t)b * Q + q] = value;
    if (num_pos) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) pairs += __shfl_xor(pairs, off);
        if (lane == 0 && pairs) atomicAdd(num_pos + blockIdx.z, pairs);
    }
}

int check_sets(const DmIndexSet *sets, int S, int B, int M)
{
    if (S < 0 || B < 0 || M < 0 || (S > 0 && !sets)) return DM_E_INVALID_ARG;
    if (B > 65535) return DM_E_UNSUPPORTED;
    for (int s = 0; s < S; ++s)
        if (sets[s].Q < 1 || !sets[s].pred_boxes || !dm::index_set_boxes_ok(sets[s])) return DM_E_INVALID_ARG;
    return DM_OK;
}

}  // namespace

namespace dm {
int check_hip(hipError_t e, const char *what);
}

extern "C" int dm_atss_f32(const DmIndexSet *sets, int S, int B, int M, int k, const float *tgt_boxes,
                           const int32_t *tgt_off, int32_t *cand, uint64_t *pos_mask, void *stream)
{
    int rc = check_sets(sets, S, B, M);
    if (rc != DM_OK) return rc;
    if (k < 1 || k > 64) return DM_E_UNSUPPORTED;
    if (S == 0 || M == 0) return DM_OK;
    if (!tgt_boxes || !tgt_off || !cand || !pos_mask) return DM_E_INVALID_ARG;
    for (int s = 0; s < S; ++s)
        if (sets[s].Q < k) return DM_E_INVALID_ARG;      // the reference's topk raises
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s0 = 0; s0 < S; s0 += DM_MAX_SETS_PER_LAUNCH) {
        const int n = std::min(DM_MAX_SETS_PER_LAUNCH, S - s0);
        DmIndexSetArray arr;
        for (int i = 0; i < n; ++i) arr.s[i] = sets[s0 + i];
        hipLaunchKernelGGL(atss_kernel, dim3(M, n), dim3(64), 0, st, arr, B, M, k, tgt_boxes, tgt_off,
                           cand + (size_t)s0 * M * k, reinterpret_cast<unsigned long long *>(pos_mask) + (size_t)s0 * M);
        rc = dm::check_hip(hipGetLastError(), "atss_kernel launch");
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}

extern "C" int dm_frcnn_f32(const DmIndexSet *sets, int S, int B, int M, int W, float pos_threshold,
                            float neg_threshold, const float *tgt_boxes, const int32_t *tgt_off,
                            uint64_t *pos_bits, int32_t *best, uint64_t *neg_bits, void *stream)
{
    int rc = check_sets(sets, S, B, M);
    if (rc != DM_OK) return rc;
    if (S == 0 || B == 0) return DM_OK;
    if (!tgt_off || !neg_bits || W < 1 || W > 65535) return DM_E_INVALID_ARG;
    if (M > 0 && (!tgt_boxes || !pos_bits || !best)) return DM_E_INVALID_ARG;
    for (int s = 0; s < S; ++s)
        if ((sets[s].Q + 63) / 64 > W) return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s0 = 0; s0 < S; s0 += DM_MAX_SETS_PER_LAUNCH) {
        const int n = std::min(DM_MAX_SETS_PER_LAUNCH, S - s0);
        DmIndexSetArray arr;
        for (int i = 0; i < n; ++i) arr.s[i] = sets[s0 + i];
        if (M > 0) {
            hipLaunchKernelGGL(rcnn_pos_kernel, dim3(M, n), dim3(64), 0, st, arr, B, M, W, pos_threshold, tgt_boxes,
                               tgt_off, reinterpret_cast<unsigned long long *>(pos_bits) + (size_t)s0 * M * W,
                               best + (size_t)s0 * M);
            rc = dm::check_hip(hipGetLastError(), "rcnn_pos_kernel launch");
            if (rc != DM_OK) return rc;
        }
        hipLaunchKernelGGL(rcnn_neg_kernel, dim3(W, B, n), dim3(64), 0, st, arr, B, W, neg_threshold, tgt_boxes, tgt_off,
                           reinterpret_cast<unsigned long long *>(neg_bits) + (size_t)s0 * B * W);
        rc = dm::check_hip(hipGetLastError(), "rcnn_neg_kernel launch");
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}

extern "C" int dm_frcnn_dense_i64(const DmIndexSet *sets, int S, int B, int M, float pos_threshold,
                                  float neg_threshold, const float *tgt_boxes, const int64_t *tgt_labels,
                                  const int32_t *tgt_off, const int32_t *best, int64_t n_cls, int32_t *num_pos,
                                  void *stream)
{
    int rc = check_sets(sets, S, B, M);
    if (rc != DM_OK) return rc;
    if (S == 0) return DM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (num_pos) {
        rc = dm::check_hip(hipMemsetAsync(num_pos, 0, sizeof(int32_t) * (size_t)S, st), "hipMemsetAsync(num_pos)");
        if (rc != DM_OK) return rc;
    }
    if (B == 0) return DM_OK;
    if (!tgt_off || (M > 0 && (!tgt_boxes || !tgt_labels || !best))) return DM_E_INVALID_ARG;
    for (int s0 = 0; s0 < S; s0 += DM_MAX_SETS_PER_LAUNCH) {
        const int n = std::min(DM_MAX_SETS_PER_LAUNCH, S - s0);
        DmIndexSetArray arr;
        int max_q = 0;
        for (int i = 0; i < n; ++i) {
            arr.s[i] = sets[s0 + i];
            if (!arr.s[i].target_cls) return DM_E_INVALID_ARG;
            max_q = std::max(max_q, arr.s[i].Q);
        }
        hipLaunchKernelGGL(rcnn_dense_kernel, dim3((max_q + 63) / 64, B, n), dim3(64), 0, st, arr, B, M, pos_threshold,
                           neg_threshold, tgt_boxes, tgt_labels, tgt_off, best ? best + (size_t)s0 * M : nullptr, n_cls,
                           num_pos ? num_pos + s0 : nullptr);
        rc = dm::check_hip(hipGetLastError(), "rcnn_dense_kernel launch");
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}
