/*
 * detr_match.h -- C ABI of libdetr_match.so, the MI355X (gfx950) implementation of the
 * DETR-family bipartite-matching path.
 *
 * The path it replaces is HungarianMatcher.forward of the reference
 * (/root/reference/models/matcher.py:38-95) with its box helpers
 * (/root/reference/utils/box_ops.py:17-21, 32-45, 48-69).  The reference has no FFI of
 * its own (it is pure Python calling torch and scipy); these entry points are what a
 * ctypes binding inside that matcher would call.
 *
 * Conventions
 *   - Every pointer marked "device" is GPU memory of the current HIP device; "host"
 *     pointers are ordinary process memory and are only read during the call.
 *   - All launches go to `stream` (a hipStream_t passed as void*; NULL = the null stream)
 *     and the calls return without synchronising.
 *   - Return value: DM_OK or a negative DM_E_* code for argument / launch errors.
 *     Data-dependent errors (the exceptions the reference raises) are reported
 *     asynchronously through the device `status` word as an OR of DM_ST_* bits;
 *     the caller zeroes it before the call and reads it back after synchronising.
 *
 * Data layout
 *   One call handles S "output sets" (decoder layers, two-stage encoder head ...) over the
 *   same B images and the same targets.  Problem (s, b) matches the Q_s queries of set s
 *   in image b to the T_b targets of image b.
 *   - targets are packed over the batch: tgt_labels[M], tgt_boxes[M][4] (cxcywh),
 *     tgt_off[B+1] exclusive prefix sums of T_b, M = tgt_off[B].
 *   - cost of set s: rows are TARGETS: cost[set.cost_off + (tgt_off[b] + t) * set.q_stride + q]
 *     is the matching cost of query q and target t of image b, i.e. the transpose of the
 *     reference's per-image block C[b][:, off_b:off_b+T_b] (matcher.py:91-94).
 *     A set's region is M * q_stride floats.
 *   - indices of set s: image b's min(Q_s, T_b) pairs are written to
 *     src_out/tgt_out[set.out_off + tgt_off[b] + k], src ascending -- the order
 *     scipy.optimize.linear_sum_assignment returns (matcher.py:94-95).  A set's region is
 *     M int64 entries; entries beyond min(Q_s, T_b) of an image are left untouched.
 */
#ifndef DETR_MATCH_H
#define DETR_MATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_ABI_VERSION 2
#define DM_MAX_SETS_PER_LAUNCH 16

/* return codes */
#define DM_OK 0
#define DM_E_INVALID_ARG (-1)
#define DM_E_LAUNCH (-2)       /* a HIP call failed; dm_last_hip_error() has the text */
#define DM_E_UNSUPPORTED (-3)  /* problem exceeds a documented limit */
#define DM_NEED_BUFFERS 1      /* dm_match_targets_f32 only: nothing was launched, call again with the buffers */

/* status-word bits (device int32), and the exception the reference raises for each */
#define DM_ST_LABEL_RANGE 1     /* IndexError: label outside [-C, C)   (matcher.py:73,81) */
#define DM_ST_DEGENERATE_BOX 2  /* AssertionError: x1<x0, y1<y0 or NaN (box_ops.py:59-60) */
#define DM_ST_INVALID_COST 4    /* ValueError: NaN or -inf cost entry  (scipy, matcher.py:94) */
#define DM_ST_INFEASIBLE 8      /* ValueError: cost matrix is infeasible (scipy) */
#define DM_ST_BAD_SIZES 16      /* no counterpart: an image has more targets (tgt_off) than the call's max_T says;
                                   that problem was skipped (the kernels size their LDS arrays by max_T) */

/* One output set: the `output` dict of one matcher call (matcher.py:43-45). */
typedef struct DmSet {
    const float *logits;  /* device, [B, Q, C]: image b's rows start at logits + b * logits_stride, dense from there
                             (outputs["pred_logits"]) */
    const float *boxes;   /* device, [B, Q, 4] cxcywh, 16-byte aligned: image b's rows start at boxes + b * boxes_stride
                             (outputs["pred_boxes"]) */
    int64_t cost_off;     /* element offset of this set's region in the cost buffer */
    int64_t out_off;      /* element offset of this set's region in src_out / tgt_out */
    int32_t Q;            /* queries per image */
    int32_t q_stride;     /* row stride of the cost region in floats, >= Q */
    int32_t logits_stride; /* floats from one image's logits to the next; 0 = contiguous (Q * C).  The denoising models hand
                              the matcher the tail of every image's queries -- pred_logits.split(sizes, dim=1)[1]
                              (/root/reference/models/dino_detr/dino_func.py:164-182, dn_detr/dn_func.py:167-188) -- which is
                              such a view: Q rows out of Q + Q_dn per image */
    int32_t boxes_stride;  /* floats from one image's boxes to the next, a multiple of 4; 0 = contiguous (Q * 4) */
} DmSet;

/* Matcher hyper-parameters: HungarianMatcher.__init__ (matcher.py:21-36). */
typedef struct DmParams {
    int32_t cls_mode;     /* 0: 'ce' softmax class cost (matcher.py:68-73); 1: focal (74-81) */
    float w_class;        /* cost_class */
    float w_bbox;         /* cost_bbox  */
    float w_giou;         /* cost_giou  */
    float focal_alpha;
    float focal_gamma;
} DmParams;

int dm_abi_version(void);
const char *dm_strerror(int code);
const char *dm_last_hip_error(void);

/* Number of GPUs HIP sees; <= 0 when there is none (the library still loads). */
int dm_device_count(void);

/*
 * Packs the per-image targets of a batch: replaces the two torch.cat calls of matcher.py:65-66 and
 * produces tgt_off / order as well, in one launch: up to 96 images without any upload (pointers, offsets and
 * order travel as kernel arguments), larger batches through a table in pinned host memory that the library keeps
 * per host thread (the kernel reads it across the bus).
 *   pointers  host uint64[2 * B]: per image the device address of its int64 labels [T_b] and of its float32
 *             cxcywh boxes [T_b][4] (16-byte aligned), both contiguous
 *   meta      host int32[2 * B + 1]: tgt_off (B + 1 exclusive prefix sums of T_b), then the processing order (B)
 *   labels_out device int64[M], boxes_out device float[M][4] (16-byte aligned), meta_out device int32[2 * B + 1]
 *             (tgt_off = meta_out, order = meta_out + B + 1)
 */
int dm_pack_targets(const uint64_t *pointers, const int32_t *meta, int B, int64_t *labels_out,
                    float *boxes_out, int32_t *meta_out, void *stream);

/*
 * Fused matching-cost kernel: replaces matcher.py:59-90 and box_ops.py:17-69 for the
 * B diagonal blocks the reference consumes (class cost + L1 + GIoU, weighted sum).
 *   sets     host, S entries            tgt_labels device int64[M]
 *   tgt_boxes device float[M][4]        tgt_off    device int32[B+1]
 *   max_T    host-known max_b T_b       cost       device, see layout above
 *   status   device int32[1], OR-ed with DM_ST_LABEL_RANGE / DM_ST_DEGENERATE_BOX
 * Prediction boxes of every image are validated, also of images without targets,
 * as the reference's batch-wide assert does.
 * Limits (DM_E_UNSUPPORTED beyond): B <= 65535 images, S <= 4096 sets.  Any C: while a tile of 64 logit rows
 * fits in LDS (64 * C * 4 bytes + 3.5 KB <= 160 KB, C <= 626) the logits are streamed through LDS once; beyond
 * that a slower kernel reads them from memory (softmax statistics in 64-byte pieces, one gathered logit per pair).
 */
int dm_cost_f32(const DmSet *sets, int S, int B, int C,
                const int64_t *tgt_labels, const float *tgt_boxes, const int32_t *tgt_off,
                int max_T, const DmParams *params, float *cost, int32_t *status, void *stream);

/*
 * Batched rectangular linear-sum-assignment: replaces the per-image
 * scipy.optimize.linear_sum_assignment loop (matcher.py:93-95) for all S*B problems in
 * one pass, reproducing scipy's result including its choice among tied optima.
 *   cost     device, layout above (produced by dm_cost_f32 or by the caller)
 *   order    device int32[B] or NULL: image processing order (largest T first helps)
 *   src_out, tgt_out  device int64, layout above
 *   status   OR-ed with DM_ST_INVALID_COST / DM_ST_INFEASIBLE
 * Limits: min(Q, T) <= 4096 and max(Q, T) <= 4096 per problem (DM_E_UNSUPPORTED beyond).
 */
int dm_lap_f32(const DmSet *sets, int S, int B, const float *cost,
               const int32_t *tgt_off, const int32_t *order, int max_T,
               int64_t *src_out, int64_t *tgt_out, int32_t *status, void *stream);

/* dm_cost_f32 followed by dm_lap_f32 on the same stream: one whole matcher step. */
int dm_match_f32(const DmSet *sets, int S, int B, int C,
                 const int64_t *tgt_labels, const float *tgt_boxes, const int32_t *tgt_off,
                 const int32_t *order, int max_T, const DmParams *params,
                 float *cost, int64_t *src_out, int64_t *tgt_out, int32_t *status, void *stream);

/*
 * dm_lap_f32 / dm_match_f32 with one more output for callers that keep the indices on the device:
 * flat_out[k] = tgt_out[k] + tgt_off[image of k], the matched target's position among the packed
 * targets of the batch -- what `torch.cat([t['boxes'][i] for t, (_, i) in zip(targets, indices)])`
 * (detr/loss.py:103) gathers, as one index into the packed boxes.  Same layout as tgt_out; may be
 * NULL (then identical to the calls above).
 */
int dm_lap_flat_f32(const DmSet *sets, int S, int B, const float *cost,
                    const int32_t *tgt_off, const int32_t *order, int max_T,
                    int64_t *src_out, int64_t *tgt_out, int64_t *flat_out, int32_t *status, void *stream);
int dm_match_flat_f32(const DmSet *sets, int S, int B, int C,
                      const int64_t *tgt_labels, const float *tgt_boxes, const int32_t *tgt_off,
                      const int32_t *order, int max_T, const DmParams *params,
                      float *cost, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out,
                      int32_t *status, void *stream);

/*
 * dm_match_flat_f32 as ONE kernel per group of output sets where the shapes allow it: replaces
 * matcher.py:59-95 as a unit.  With C <= 4 logits per query and max_T <= 256, every assignment workgroup
 * computes the cost block of its own problem (same arithmetic, same bits as dm_cost_f32) straight into
 * LDS and solves it there; the cost matrix never travels through memory.  Sets that can meet an image
 * with T >= Q, and calls outside those limits, take the two-kernel path of dm_match_flat_f32.  The
 * resulting indices are identical to dm_match_flat_f32's in every case.
 *   workspace  device, sized like `cost` of dm_match_f32; its contents after the call are unspecified
 *              (rows of a cost block that do not fit in LDS are kept there)
 */
int dm_match_fused_f32(const DmSet *sets, int S, int B, int C,
                       const int64_t *tgt_labels, const float *tgt_boxes, const int32_t *tgt_off,
                       const int32_t *order, int max_T, const DmParams *params,
                       float *workspace, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out,
                       int32_t *status, void *stream);

/*
 * A whole matcher call from the reference's own inputs: dm_pack_targets followed by dm_match_fused_f32 -- or,
 * when the step runs as one fused kernel and B <= 128, that kernel alone, reading every image's labels and boxes
 * through the pointers (they travel, with offsets and order, as kernel arguments): no packed copy, no second launch.
 *   pointers, meta              as for dm_pack_targets
 *   labels_buf, boxes_buf, meta_buf   device room for the packed targets (M int64, M x 4 float 16-byte aligned,
 *                               2 * B + 1 int32); filled only when the call had to pack: *packed says so.
 *                               All three may be NULL: a call that would have to pack then launches nothing
 *                               and returns DM_NEED_BUFFERS
 *   the rest                    as for dm_match_fused_f32
 * The per-image tensors must stay alive and unchanged until the call's work on `stream` has completed.
 */
int dm_match_targets_f32(const DmSet *sets, int S, int B, int C, const uint64_t *pointers, const int32_t *meta,
                         int max_T, const DmParams *params, int64_t *labels_buf, float *boxes_buf, int32_t *meta_buf,
                         float *workspace, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out, int32_t *status,
                         void *stream, int *packed);

/*
 * dm_match_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  What dm_match_f32
 * keeps in its `cost` workspace -- the cost blocks on their way from the cost kernel to the assignment kernel
 * (C > 4, sets that can meet an image with T >= Q) and, where the assignment workgroups compute their own
 * blocks, the rows that do not fit in LDS -- lives in a fixed array of DM_NOALLOC_MAX_COST_FLOATS floats (32 MiB)
 * that the library holds on every GPU it runs on until the process ends, so there is no `cost` argument; the
 * call itself calls no hipMalloc / hipFree or other allocating routine.
 *   M   host-known total number of targets, tgt_off[B]
 * sets[s].cost_off / q_stride place set s inside that array as they do inside `cost` above: every set's
 * region [cost_off, cost_off + M * q_stride) must end at or before DM_NOALLOC_MAX_COST_FLOATS, else
 * DM_E_UNSUPPORTED (nothing is launched).  All calls on one GPU share the array: calls that overlap in time
 * on the same GPU (two streams, two threads) race; serialise them.  Nothing in it needs clearing between
 * calls: a kernel reads only entries that a kernel (or the same thread) of the same call has written before,
 * so what an earlier call left behind is never seen.
 */
#define DM_NOALLOC_MAX_COST_FLOATS (8u << 20)
int dm_match_noalloc_f32(const DmSet *sets, int S, int B, int C,
                         const int64_t *tgt_labels, const float *tgt_boxes, const int32_t *tgt_off,
                         const int32_t *order, int max_T, int M, const DmParams *params,
                         int64_t *src_out, int64_t *tgt_out, int32_t *status, void *stream);
/*
 * Testing aid (no counterpart in the reference): the address of that array on the current GPU and its length
 * in floats, so that a test can fill it with a marker before a call and read back what the call left in it.
 */
int dm_noalloc_scratch(float **ptr, uint64_t *floats);

/*
 * ---- Consumers of the indices (the criterion's first steps after the matcher) ----------------
 * One output set as these entry points see it.  src / tgt are the buffers dm_lap_f32 wrote
 * (same layout, same out_off); everything stays on the device.
 */
typedef struct DmIndexSet {
    const float *pred_boxes; /* device [B, Q, 4] cxcywh, 16-byte aligned, image b's rows at pred_boxes + b * boxes_stride
                                (dm_pair_loss_f32, the auxiliary assigners) or NULL */
    float *grad_boxes;       /* device [B, Q, 4], every element written (dm_pair_loss_backward_f32) or NULL */
    int64_t *target_cls;     /* device [B, Q], every element written (dm_target_classes_i64) or NULL */
    int64_t out_off;         /* element offset of the set's region in src / tgt */
    int32_t Q;               /* queries per image */
    int32_t boxes_stride;    /* floats from one image's prediction boxes to the next, a multiple of 4; 0 = contiguous (Q * 4) */
} DmIndexSet;

/*
 * L1 and GIoU loss sums of the matched pairs of every set: replaces the gathers, the [N, N]
 * generalized_box_iou + diag and the reductions of calc_box_loss
 * (/root/reference/models/detr/loss.py:97-112; the caller divides by num_boxes, :106,112).
 *   sums       device float[S][2]: sum |pred - target| and sum (1 - giou) of set s
 *   pair_grad  device float[S][M][8] or NULL: per packed slot, d l1 / d pred (4) and
 *              d (1 - giou) / d pred (4) in cxcywh; zero for slots without a pair
 * M = tgt_off[B] (host-known).  Fixed-order reductions: bitwise reproducible.
 */
int dm_pair_loss_f32(const DmIndexSet *sets, int S, int B, int M, const float *tgt_boxes,
                     const int32_t *tgt_off, const int64_t *src, const int64_t *tgt,
                     float *sums, float *pair_grad, void *stream);

/*
 * Backward of dm_pair_loss_f32: grad_boxes[b][q] = grad_sums[s][0] * d l1 + grad_sums[s][1] *
 * d (1 - giou) for matched queries, 0 elsewhere (what autograd produces for loss.py:102-112).
 *   grad_sums  device float[S][2]
 * Limit: Q <= 40960 (DM_E_UNSUPPORTED beyond).
 */
int dm_pair_loss_backward_f32(const DmIndexSet *sets, int S, int B, int M, const int32_t *tgt_off,
                              const int64_t *src, const int64_t *tgt, const float *pair_grad,
                              const float *grad_sums, void *stream);

/*
 * The same two sums for an explicit list of pairs -- indices that did not come from the matcher: the denoising groups
 * of DN-DETR / DINO / Co-DETR bring their own (/root/reference/models/dino_detr/loss.py:170-172, built at
 * dino_func.py:117-124: several groups of queries per target), an auxiliary assigner may pair one prediction with
 * several targets.  The reference's calc_box_loss builds an [N, N] GIoU matrix for them too (loss.py:108-111), N being
 * groups x targets of the whole batch.
 *   pred_boxes  device [B, Q, 4], 16-byte aligned, image b at pred_boxes + b * boxes_stride floats (0 = Q * 4)
 *   tgt_boxes   device [M, 4]: the packed targets
 *   pair_b / pair_q / pair_t   device int64[N]: image, query and packed target of every pair
 *   sums        device float[2]: sum |pred - target| and sum (1 - giou) (the caller divides by num_boxes)
 *   pair_grad   device float[N][8] or NULL: d l1 / d pred (4) and d (1 - giou) / d pred (4) of every pair; the
 *               caller accumulates them into [B, Q, 4] (a query may occur in several pairs)
 *   partials    device double[DM_CLS_LOSS_PARTIALS] workspace;  status: DM_ST_LABEL_RANGE for an index out of range
 *               (the reference's gathers raise IndexError; the pair contributes nothing)
 */
int dm_pair_list_loss_f32(const float *pred_boxes, int boxes_stride, int B, int Q, const float *tgt_boxes, int M,
                          const int64_t *pair_b, const int64_t *pair_q, const int64_t *pair_t, int64_t N,
                          float *sums, float *pair_grad, double *partials, int32_t *status, void *stream);

/*
 * Backward of dm_pair_list_loss_f32 (what autograd produces for loss.py:102-112 on such indices):
 * grad_boxes[b][q] += grad_sums[0] * d l1 + grad_sums[1] * d (1 - giou) over the pairs of (b, q).
 *   grad_sums   device float[2];  grad_boxes  device [B, Q, 4] contiguous, zeroed by the caller
 * The additions are float atomics: a query that occurs in several pairs receives them in no fixed order.
 */
int dm_pair_list_backward_f32(const float *pair_grad, const float *grad_sums, const int64_t *pair_b,
                              const int64_t *pair_q, int64_t N, int B, int Q, float *grad_boxes, void *stream);

/*
 * Dense class map of every set: target_cls[b][q] = label of the target matched to query q, else
 * no_object (/root/reference/models/detr/loss.py:121-125).  Limit: Q <= 40960.
 */
int dm_target_classes_i64(const DmIndexSet *sets, int S, int B, int M, const int64_t *tgt_labels,
                          const int32_t *tgt_off, const int64_t *src, const int64_t *tgt,
                          int64_t no_object, void *stream);

/*
 * ---- The classification loss on the dense class map ---------------------------------------------
 * What the criterion's `cls_criterion` computes from pred_logits and the [B, Q] class map
 * (/root/reference/models/detr/loss.py:127), in one pass over the logits: the loss and, when grad_out is given,
 * d loss / d logit for every logit (what autograd hands back for a unit upstream gradient).
 *   logits      device float[rows][C], contiguous (16-byte aligned for dm_focal_loss_f32), rows = B * Q
 *   target_cls  device int64[rows]
 *   partials    device double[DM_CLS_LOSS_PARTIALS]: workspace, overwritten (calls that share it must be ordered)
 *   status      device int32, bits are OR-ed in: DM_ST_LABEL_RANGE for a class outside the accepted range (the
 *               reference's one_hot / cross_entropy raise for it); such a row counts as ignored
 * rows * C < 2^31 (DM_E_UNSUPPORTED beyond).  Fixed-order reductions: bitwise reproducible.
 *
 * dm_focal_loss_f32: FocalLoss.forward + sigmoid_focal_loss (loss.py:198-213, 217-242) on
 * one_hot(target_cls, C + 1)[..., :-1]: class C is "no object" (no positive logit in the row).
 *   loss_out[0] = scale * sum over all logits; the reference's `loss.mean(1).sum() / num_boxes * Q` is
 *   scale = 1 / num_boxes.  alpha < 0: no alpha weighting (:238).  With use_ignore, rows whose class is
 *   ignore_index contribute nothing (the foreground selection of calc_cls_loss_with_ignore,
 *   /root/reference/models/co_detr/loss.py:161-166).
 */
#define DM_CLS_LOSS_PARTIALS 4096
int dm_focal_loss_f32(const float *logits, const int64_t *target_cls, int64_t rows, int C, float alpha,
                      float gamma, float scale, int64_t ignore_index, int use_ignore, float *loss_out,
                      float *grad_out, double *partials, int32_t *status, void *stream);

/*
 * dm_cross_entropy_f32: CrossEntropy.forward (loss.py:185-195) = nn.CrossEntropyLoss(weight), reduction 'mean':
 * sum_i w[y_i] * (logsumexp(x_i) - x_i[y_i]) / sum_i w[y_i] over the rows whose class is not ignore_index.
 *   weight      device float[C] or NULL (all ones)
 *   loss_out    device float[2]: the loss, and 1 / sum_i w[y_i]
 *   grad_out    w[y_i] * (softmax(x_i) - onehot(y_i)) -- NOT yet divided by sum_i w[y_i]: the caller scales by
 *               loss_out[1] (a device value, so that nothing synchronises)
 */
int dm_cross_entropy_f32(const float *logits, const int64_t *target_cls, const float *weight, int64_t rows,
                         int C, int64_t ignore_index, float *loss_out, float *grad_out, double *partials,
                         int32_t *status, void *stream);

/*
 * ---- Co-DETR auxiliary assigners ---------------------------------------------------------------
 * Replace ATSS.forward and FasterRCNN.forward of
 * /root/reference/models/co_detr/auxiliary_head.py:22-60 and 69-96 for all output sets of a step
 * (sets[s].pred_boxes and Q are used).  They fill fixed-size tables; the caller turns them into
 * the reference's variable-length index lists.
 *
 * dm_atss_f32: for target slot t of set s
 *   cand[(s*M + t)*k + j]   the j-th closest prediction of the target's image by Euclidean distance
 *                           of the box centres (equal distances: lower index first)
 *   pos_mask[s*M + t]       bit j set: candidate j is a positive (IoU >= mean + unbiased std of the
 *                           k IoUs, and its centre lies inside the target box)
 * 1 <= k <= 64 (DM_E_UNSUPPORTED beyond); k > Q is DM_E_INVALID_ARG (the reference's topk raises).
 */
int dm_atss_f32(const DmIndexSet *sets, int S, int B, int M, int k, const float *tgt_boxes,
                const int32_t *tgt_off, int32_t *cand, uint64_t *pos_mask, void *stream);

/*
 * dm_frcnn_f32: W = words per row of the bit tables, >= ceil(Q/64) of every set.
 *   pos_bits[(s*M + t)*W + w]  bit i: IoU(target t, prediction 64w+i) > pos_threshold
 *   best[s*M + t]              the prediction with the largest IoU for target t (first among equals);
 *                              the reference adds it to the target's positives
 *   neg_bits[(s*B + b)*W + w]  bit i: prediction 64w+i of image b has IoU < neg_threshold with every
 *                              target of the image (all predictions when the image has no target)
 */
int dm_frcnn_f32(const DmIndexSet *sets, int S, int B, int M, int W, float pos_threshold,
                 float neg_threshold, const float *tgt_boxes, const int32_t *tgt_off,
                 uint64_t *pos_bits, int32_t *best, uint64_t *neg_bits, void *stream);

/*
 * dm_frcnn_dense_i64: the dense [B, Q] class targets that calc_cls_loss_with_ignore
 * (/root/reference/models/co_detr/loss.py:143-161) scatters from the Faster-RCNN positives and negatives,
 * written straight to sets[s].target_cls: n_cls for negatives, the label of the last positive target
 * otherwise, -1 for predictions that are neither (ignored).  `best` is dm_frcnn_f32's output.
 * num_pos (int32[S], may be NULL) receives the number of positive pairs of each set, the
 * normaliser of loss.py:151 before its max(., 1).
 */
int dm_frcnn_dense_i64(const DmIndexSet *sets, int S, int B, int M, float pos_threshold,
                       float neg_threshold, const float *tgt_boxes, const int64_t *tgt_labels,
                       const int32_t *tgt_off, const int32_t *best, int64_t n_cls, int32_t *num_pos,
                       void *stream);

/*
 * Measurement aids (no counterpart in the reference).  With profiling enabled,
 * dm_match_f32 records HIP events on its stream before the cost kernels, between the cost
 * and the assignment kernels, and after the assignment kernels; dm_profile_read returns the
 * two elapsed times of the most recent dm_match_f32 call in milliseconds and must be called
 * after that call's stream has been synchronised.  Not thread-safe.
 */
int dm_profile_enable(int on);
int dm_profile_read(float *cost_ms, float *lap_ms);

/*
 * Measurement aid (no counterpart in the reference): how a dm_match_fused_f32 / dm_match_targets_f32 call of this
 * shape would run.  *cost_sets = the number of the S output sets whose cost blocks go through the cost kernel and
 * memory first (0: every assignment workgroup computes its own, the call is assignment launches only).  Launches
 * nothing.  Only Q and q_stride of the sets are read.
 */
int dm_match_plan(const DmSet *sets, int S, int B, int C, int max_T, int *cost_sets);

/*
 * Testing / tuning aid: launch-shape knobs.  None of them changes a result (the tests check that);
 * value 0 restores the automatic choice.  Returns DM_OK or DM_E_INVALID_ARG.  Not thread-safe.
 */
#define DM_KNOB_LAP_KERNEL 0     /* 1: list-order kernel only, 2: workgroup-per-problem, 3: wavefront-per-problem */
#define DM_KNOB_LAP_MAX_WAVES 1  /* 1..16: wavefronts per problem in the workgroup-per-problem kernel */
#define DM_KNOB_COST_QPT 2       /* 1 / 2: queries per thread in the C <= 4 cost kernel */
#define DM_KNOB_COST_TSPLIT 3    /* 1 / 2 / 4: workgroups sharing the targets of an image in the C <= 4 cost kernel */
#define DM_KNOB_TILE_WAVES 4     /* 4 / 8: wavefronts per workgroup in the C > 4 cost kernel */
#define DM_KNOB_LAP_LDS_KB 5     /* 1..80: LDS budget per problem of the register assignment kernels, in KB */
#define DM_KNOB_LAP_WAVE_BUDGET 6  /* wavefronts one launch of the workgroup-per-problem kernel may ask for */
#define DM_KNOB_DIRECT_TARGETS 8  /* 1: dm_match_targets_f32 packs the targets whatever the shapes */
#define DM_KNOB_LAP_PLAIN_SHAPE 9  /* workgroup-per-problem kernel with few problems in flight.  1: one wavefront per 64-column
                                      slice (up to 16); 2..16: exactly that many wavefronts (automatic: at most twelve, the first
                                      ones take the slices left over as second columns) */
#define DM_KNOB_FUSE 7           /* 1: dm_match_fused_f32 takes the two-kernel path whatever the shapes */
int dm_debug_set_knob(int knob, int value);

#ifdef __cplusplus
}
#endif
#endif /* DETR_MATCH_H */
