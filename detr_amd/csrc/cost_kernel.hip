// Fused matching-cost kernel for gfx950 (MI355X).
//
// Replaces, for the B per-image blocks the matcher consumes, the chain of whole-batch
// torch ops at /root/reference/models/matcher.py:59-90 and
// /root/reference/utils/box_ops.py:17-21, 32-45, 48-69: class cost (softmax gather or
// focal), L1 (torch.cdist p=1), GIoU, weighted sum -- one pass, one write per cost entry.
//
// Work split: one lane per query; the image's targets are staged once per workgroup in LDS
// as pre-converted records (xyxy, area, cxcywh, label) that every lane reads as broadcasts;
// the cost row of a target is written with consecutive lanes on consecutive queries
// (coalesced 256-byte stores).
//   cost_small_kernel  C <= 4: the logits go straight to registers and the class cost of
//                      every class is computed once per query.
//   cost_tile_kernel   C > 4: a workgroup of four wavefronts shares one tile of 64 logit
//                      rows, streamed into LDS as a linear image (LDS-DMA); the wavefronts
//                      split the targets four ways.
//
// HBM-bound streaming work: no contraction, so no MFMA.  Compiled with -ffp-contract=off;
// the two divisions of the GIoU are evaluated as reciprocal * numerator (v_rcp_f32, 1 ulp)
// and the final sums use explicit fma -- within 1e-7 relative of the reference's
// separately rounded operations, far inside the 1e-5 parity bar.
#include <algorithm>
#include <mutex>
#include <vector>

#include "cost_math.h"

namespace {

using namespace dm;

constexpr int TCHUNK = 64;   // targets staged per pass
#ifndef DM_TILE_ABL
#define DM_TILE_ABL 0
#endif
// cache-policy bits of the tile's LDS-DMA loads: 2 = nt (the logits are read once; keeping them
// out of the way of the cost rows being assembled in L2 measured 6 % faster)
#ifndef DM_TILE_LDAUX
#define DM_TILE_LDAUX 2
#endif
// nontemporal stores of the small-C kernel's cost rows (measured 8 % faster on the largest batches)
#ifndef DM_SMALL_NTST
#define DM_SMALL_NTST 1
#endif
#ifndef DM_SMALL_ABL
#define DM_SMALL_ABL 0
#endif
#ifndef DM_TILE_NTST
#define DM_TILE_NTST 0
#endif

// Stage targets [tc, tc + nt) of the image in LDS (validated, pre-converted).
__device__ __forceinline__ void stage_targets(TgtRec *tg, const int64_t *__restrict__ labels,
                                              const float *__restrict__ tboxes, int first, int nt, int C,
                                              int32_t *status)
{
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
        const int t = first + i;
        store_target(tg + i, reinterpret_cast<const float4 *>(tboxes)[t], labels[t], C, status);
    }
}

// C <= 4.  Every thread owns QPT consecutive queries of one image and loops over the image's
// targets; the QPT costs of a target go out as one QPT*4-byte store.
template <int C, int QPT>
__global__ __launch_bounds__(512, 2) void cost_small_kernel(const DmSetArray sets,
                                                            const int64_t *__restrict__ labels,
                                                            const float *__restrict__ tboxes,
                                                            const int32_t *__restrict__ tgt_off, const DmParams prm,
                                                            float *__restrict__ cost, int32_t *status, const int tsplit)
{
    __shared__ TgtRec tg[TCHUNK + 1];        // (+1: the record read ahead of the last one)
    typedef float vecq __attribute__((ext_vector_type(QPT)));

    const DmSet &set = sets.s[blockIdx.z];
    const int b = blockIdx.y;
    const int Q = set.Q;
    // blockIdx.x = query tile * tsplit + target slice: with few problems in flight the targets of
    // an image are split over `tsplit` workgroups to shorten every thread's serial loop
    const int tile = blockIdx.x / tsplit, slice = blockIdx.x - tile * tsplit;
    const int q0 = (tile * blockDim.x + threadIdx.x) * QPT;     // first query of this thread
    if ((int)(tile * blockDim.x) * QPT >= Q) return;
    const int t_all0 = tgt_off[b];
    const int T_all = tgt_off[b + 1] - t_all0;
    const int per = (T_all + tsplit - 1) / tsplit;
    const int t0 = t_all0 + min(slice * per, T_all);
    const int T = min(per, t_all0 + T_all - t0);

    // own prediction boxes and logits (clamped to the last query when past the end); the
    // reference validates every prediction of the batch
    float4 raw[QPT];
    float x[QPT][4];
    const float4 *img_boxes = image_boxes(set, b);
    const float *img_logits = image_logits(set, b, C);
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int row = min(q0 + i, Q - 1);
        raw[i] = img_boxes[row];
        const float *lg = img_logits + (size_t)row * C;
#pragma unroll
        for (int c = 0; c < 4; ++c) x[i][c] = c < C ? lg[c] : 0.0f;
    }
    if (T > 0) stage_targets(tg, labels, tboxes, t0, min(TCHUNK, T), C, status);

    PredBox p[QPT];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        p[i] = make_pred(raw[i]);
        ok = ok && (pred_valid(p[i]) || q0 + i >= Q);
    }
    if (!ok && slice == 0) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    if (T <= 0) return;

    // cost_class * class cost of every class, once per query
    float wc[QPT][4];
#pragma unroll
    for (int i = 0; i < QPT; ++i) small_class_costs(x[i], C, prm, wc[i]);

    // (opaque from here on: keeps the selection among the finished class costs from being rewritten
    // as cost_class * (selection among the class costs), one more multiplication per pair.  Held as one
    // vector value per query, not as an array: the compiler turns a selection among three or four array
    // elements into an indexed read of a per-thread table in LDS, whose addressing reads the workgroup size
    // from the dispatch packet in host memory -- 14 us per launch.)
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 wv[QPT];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        wv[i] = f4{wc[i][0], wc[i][1], wc[i][2], wc[i][3]};
        asm("" : "+v"(wv[i]));
    }

    const int qs = set.q_stride;
    // wave-uniform row pointer (scalar registers) + the thread's query as a 32-bit offset
    float *row_out = cost + set.cost_off + (size_t)t0 * qs;
    const unsigned qu = (unsigned)q0;
    // a QPT-wide store needs all its queries in range and natural alignment of the row start
    const bool wide = (q0 + QPT <= Q) && (qs % QPT == 0) && ((reinterpret_cast<uintptr_t>(row_out + q0) & (4 * QPT - 1)) == 0);

    for (int tc = 0; tc < T; tc += TCHUNK) {
        const int nt = min(TCHUNK, T - tc);
        if (tc > 0) {
            __syncthreads();
            stage_targets(tg, labels, tboxes, t0 + tc, nt, C, status);
        }
        __syncthreads();
        if (q0 < Q) {
            TgtRec g = tg[0];
            float *dst = row_out + (size_t)tc * qs;
            auto entry = [&](const int i) {
                const TgtRec gn = tg[i + 1];     // next record in flight while this one is used (the one past the end is never used)
                // select by data, not by branch (the label is wave-uniform and the compiler would
                // otherwise turn the selects into a scalar branch tree)
                int l = g.label;
                asm("" : "+v"(l));
                vecq r;
#pragma unroll
                for (int k = 0; k < QPT; ++k) {
                    float wcls = wv[k].x;
                    if (C > 1) wcls = (l == 1) ? wv[k].y : wcls;
                    if (C > 2) wcls = (l == 2) ? wv[k].z : wcls;
                    if (C > 3) wcls = (l == 3) ? wv[k].w : wcls;
#if DM_SMALL_ABL != 2
                    r[k] = pair_cost(p[k], g, wcls, prm.w_bbox, prm.w_giou);
#endif
                }
#if DM_SMALL_ABL == 2
                r[0] = g.area;
#endif
                if (QPT == 1) {
#if DM_SMALL_ABL == 1
                    if (r[0] == 123.456f) dst[qu] = r[0];
#elif DM_SMALL_NTST
                    // scalar row pointer + 32-bit lane offset, spelled out: the compiler would carry a
                    // 64-bit address per lane through the loop instead
                    asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(qu * 4u), "v"(r[0]), "s"(dst) : "memory");
#else
                    dst[qu] = r[0];
#endif
                } else if (wide) {
#if DM_SMALL_NTST
                    __builtin_nontemporal_store(r, reinterpret_cast<vecq *>(dst + qu));
#else
                    *reinterpret_cast<vecq *>(dst + qu) = r;
#endif
                } else {
#pragma unroll
                    for (int k = 0; k < QPT; ++k) if (q0 + k < Q) dst[qu + k] = r[k];
                }
                g = gn;
                dst += qs;
            };
            // (unrolled by hand: inline assembly counts as a convergent operation, which bars the
            // compiler from unrolling a loop whose trip count it does not know)
            int i = 0;
            for (; i + 4 <= nt; i += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) entry(i + u);
            }
            for (; i < nt; ++i) entry(i);
        }
    }
}

// C > 4.  blockDim = 64 * W: wavefront w of the workgroup handles targets t = w (mod W) of one
// tile of 64 queries of one image, two targets per trip (two independent dependency chains per
// lane); all W wavefronts stream the 64 x C logit tile into LDS together as a linear image of the
// rows (LDS-DMA when the tile starts on a 16-byte boundary: no registers, every piece in flight at
// once).  Lane l reads row l: with an odd C the row stride is odd and the lanes hit distinct banks;
// with an even C lane l walks its row rotated by l columns, which has the same effect.
// Grid (tile, image, set).  (Giving all tiles of an
// image to workgroups of one XCD, so that the 256-byte pieces of a cost row meet in one L2, measured
// 5 % slower: it concentrates the heavy images on single XCDs.)
// max(a, b) as the single instruction it is (fmaxf adds a canonicalising v_max per operand: the tile kernel issues
// one of these per logit).  NaN operands: the instruction returns the other operand, like fmaxf.
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Sum (or maximum) over each group of 16 consecutive lanes, returned in all 16: pairs, quads, halves, row --
// one DPP-operand instruction per step (the s_nop covers the VALU-write -> DPP-read hazard, which the
// compiler does not track through inline assembly).
template <bool SUM>
__device__ __forceinline__ float row16_all(float x)
{
#define DM_ROW16_STEP(CTRL)                                                                                     \
    if (SUM) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(x)); \
    else asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(x));
    DM_ROW16_STEP("quad_perm:[1,0,3,2]")
    DM_ROW16_STEP("quad_perm:[2,3,0,1]")
    DM_ROW16_STEP("row_half_mirror")
    DM_ROW16_STEP("row_mirror")
#undef DM_ROW16_STEP
    return x;
}

// MODE: 0 softmax class cost, 1 focal with gamma == 2, 2 focal with any gamma.
template <int W, int MODE>
__global__ __launch_bounds__(64 * W) void cost_tile_kernel(const DmSetArray sets, const int nsets, const int B,
                                                           const int C, const int64_t *__restrict__ labels,
                                                           const float *__restrict__ tboxes,
                                                           const int32_t *__restrict__ tgt_off, const DmParams prm,
                                                           float *__restrict__ cost, int32_t *status, const int ntiles)
{
    constexpr int NT = 64 * W;
    extern __shared__ __align__(16) unsigned char smem[];
    TgtRec *tg = reinterpret_cast<TgtRec *>(smem);                                   // [TCHUNK]
    float *part = reinterpret_cast<float *>(smem + TCHUNK * sizeof(TgtRec));         // [2][64] softmax statistics
    float *tile = part + (MODE == 0 ? 64 * 2 : 0);                                    // [64][C]

    // grid (tiles, images, sets): no integer division in the prologue (each costs ~25 instructions, and
    // this kernel is bound by the instructions it issues); x runs fastest, so the tiles of an image
    // are still dispatched together
    const int tile_id = blockIdx.x, b = blockIdx.y, si = blockIdx.z;
    const DmSet &set = sets.s[si];
    const int Q = set.Q;
    const int q0 = tile_id * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: loop counters stay scalar
    const int q = q0 + lane;
    const bool active = q < Q;
    const int nq = min(64, Q - q0);
    const int t0 = tgt_off[b];
    const int T = tgt_off[b + 1] - t0;
    const float4 raw = image_boxes(set, b)[active ? q : Q - 1];
    // the first chunk of targets (one per thread) is requested before the tile, so that its latency
    // runs under the tile's: loads complete in order, and the records are built while the tile streams
    const bool stager = tid < min(TCHUNK, T);
    float4 tgt_box = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    long long tgt_label = 0;
    if (stager) {
        tgt_box = reinterpret_cast<const float4 *>(tboxes)[t0 + tid];
        tgt_label = labels[t0 + tid];
    }
    if (T > 0) {
        // stream the tile: rows q0 .. q0+nq-1 are contiguous in memory and stay so in LDS
        const float *src = image_logits(set, b, C) + (size_t)q0 * (size_t)C;
        const int n = nq * C;
#if DM_TILE_ABL == 2
        if (false) {
#else
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
#endif
            // LDS-DMA, 1 KB per wave instruction
            for (int base = wave * 256; base < n; base += W * 256) {
                const int e = base + lane * 4;
                if (e + 3 < n)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e),
                                                     (__attribute__((address_space(3))) void *)(tile + base), 16, 0, DM_TILE_LDAUX);
            }
            const int n4 = n & ~3;
            if (tid < n - n4) tile[n4 + tid] = src[n4 + tid];
            if (stager) store_target(tg + tid, tgt_box, tgt_label, C, status);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if DM_TILE_ABL == 2
        } else if (false) {
#else
        } else {
#endif
            constexpr int UL = 8;
            for (int base = 0; base < n; base += NT * UL) {
                float v[UL];
#pragma unroll
                for (int k = 0; k < UL; ++k) v[k] = src[min(base + k * NT + tid, n - 1)];
#pragma unroll
                for (int k = 0; k < UL; ++k) {
                    const int e = base + k * NT + tid;
                    if (e < n) tile[e] = v[k];
                }
            }
            if (stager) store_target(tg + tid, tgt_box, tgt_label, C, status);
        }
    }
    const PredBox p = make_pred(raw);
    if (active && wave == 0 && !pred_valid(p)) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    if (T == 0) return;
    __syncthreads();

    const float *xr = tile + lane * C;
    constexpr float LOG2E = 1.44269504088896f;
    float ce_max = 0.0f, ce_scale = 1.0f;      // class cost -softmax = exp2((x - ce_max) * LOG2E) * ce_scale
    if (MODE == 0) {
        // Softmax statistics (row maximum, -1 / sum of exp(x - max): the reference's own formulation) of the 64 rows.
        // Sixteen lanes share a row and walk its columns side by side (consecutive words: no index arithmetic per
        // element), four rows per wavefront instruction, the row groups of a wavefront interleaved; the sixteen partial
        // results meet through four DPP steps in a fixed order.
        // Which rows meet in a wavefront instruction: the two rows of a 32-lane half (the unit in which LDS banks
        // conflict) sit k rows apart with k * C = 16 (mod 32), so that their 16-word windows fall on disjoint banks
        // (k = 16 >> number of trailing zero bits of C; none exists when C is a multiple of 32).
        constexpr int NG = 16 / W;                  // row groups per wavefront
        constexpr int GBITS = NG == 4 ? 2 : 1;
        typedef float f2 __attribute__((ext_vector_type(2)));
        const int sub = lane & 15;
        const int kb = 4 - min(__builtin_ctz(C), 4);
        const int slot0 = (lane >> 4) & 1, slot1 = lane >> 5;
        int rg[NG];
        const float *pg[NG];
        float mx[NG];
        f2 sum[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int n = slot1 | (g << 1) | (wave << (1 + GBITS));        // the other five bits of the row number
            rg[g] = ((n >> kb) << (kb + 1)) | (slot0 << kb) | (n & ((1 << kb) - 1));
            pg[g] = tile + rg[g] * C + sub;      // rows >= nq read tile space nobody wrote: results unused
            mx[g] = -INFINITY;
            sum[g] = f2{0.0f, 0.0f};
        }
        const int nfull = C >> 4;                   // chunks of 16 columns that every lane of the group has
        const bool tail = sub < (C & 15);
        {
            int k = nfull, o = 0;
            for (; k >= 4; k -= 4, o += 64) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < NG; ++g) mx[g] = vmax(mx[g], pg[g][o + 16 * j]);
            }
            for (; k > 0; --k, o += 16) {
#pragma unroll
                for (int g = 0; g < NG; ++g) mx[g] = vmax(mx[g], pg[g][o]);
            }
            if (tail) {
#pragma unroll
                for (int g = 0; g < NG; ++g) mx[g] = vmax(mx[g], pg[g][o]);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) mx[g] = row16_all<false>(mx[g]);
        {
            // two logits per packed subtraction / multiplication / addition; the exponentials stay single
            auto term2 = [&](int g, float x0, float x1) {
                const f2 t = (f2{x0, x1} - f2{mx[g], mx[g]}) * f2{LOG2E, LOG2E};
                sum[g] += f2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            };
            auto term = [&](int g, float x) { sum[g].x += __builtin_amdgcn_exp2f((x - mx[g]) * LOG2E); };
            int k = nfull, o = 0;
            for (; k >= 4; k -= 4, o += 64) {
#pragma unroll
                for (int j = 0; j < 4; j += 2)
#pragma unroll
                    for (int g = 0; g < NG; ++g) term2(g, pg[g][o + 16 * j], pg[g][o + 16 * j + 16]);
            }
            for (; k > 0; --k, o += 16) {
#pragma unroll
                for (int g = 0; g < NG; ++g) term(g, pg[g][o]);
            }
            if (tail) {
#pragma unroll
                for (int g = 0; g < NG; ++g) term(g, pg[g][o]);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float total = row16_all<true>(sum[g].x + sum[g].y);
            if (sub == 0) {
                part[rg[g]] = mx[g];
                part[64 + rg[g]] = -1.0f / total;       // class cost = -softmax
            }
        }
        __syncthreads();
        ce_max = part[lane];
        ce_scale = part[64 + lane];
    }

    const int qs = set.q_stride;
    // wave-uniform row pointers (scalar registers); the lane's query is the 32-bit offset of the store
    float *out = cost + set.cost_off + (size_t)t0 * qs;

    for (int tc = 0; tc < T; tc += TCHUNK) {
        const int nt = min(TCHUNK, T - tc);
        if (tc > 0) {
            __syncthreads();
            stage_targets(tg, labels, tboxes, t0 + tc, nt, C, status);
            __syncthreads();
        }
#if DM_TILE_ABL == 1
        if (active && prm.w_bbox == 123.0f) {
#else
        if (active) {
#endif
            auto element = [&](int i) {
                const TgtRec g = tg[i];
                const float xl = xr[g.label];
                float cls;
                if (MODE == 0) cls = __builtin_amdgcn_exp2f((xl - ce_max) * LOG2E) * ce_scale;
                else cls = focal_cost<MODE == 1>(xl, prm.focal_alpha, prm.focal_gamma);
                return pair_cost(p, g, prm.w_class * cls, prm.w_bbox, prm.w_giou);
            };
            float *dst = out + (size_t)(tc + wave) * qs;
            const size_t hop = (size_t)W * qs;
            const unsigned qu = (unsigned)q;      // unsigned 32-bit lane offset: scalar base + vector offset stores
            int i = wave;
            for (; i + W < nt; i += 2 * W, dst += 2 * hop) {
                const float r0 = element(i), r1 = element(i + W);
#if DM_TILE_ABL == 3
                if (r0 == 123.456f) { dst[qu] = r0; dst[hop + qu] = r1; }
#elif DM_TILE_NTST
                __builtin_nontemporal_store(r0, dst + qu);
                __builtin_nontemporal_store(r1, dst + hop + qu);
#else
                dst[qu] = r0;
                dst[hop + qu] = r1;
#endif
            }
            if (i < nt) {
                const float r0 = element(i);
#if DM_TILE_ABL == 3
                if (r0 == 123.456f) dst[qu] = r0;
#elif DM_TILE_NTST
                __builtin_nontemporal_store(r0, dst + qu);
#else
                dst[qu] = r0;
#endif
            }
        }
    }
}

// C beyond what a 64-row logit tile in LDS can hold (about 600 classes): the same decomposition as the tile
// kernel -- four wavefronts per tile of 64 queries, targets split four ways -- with the logits read from memory
// where the tile kernel reads its LDS tile: sixteen lanes walk a row's consecutive words for the softmax
// statistics (64-byte pieces), and the one logit a (query, target) pair needs is gathered by the query's lane.
// MODE as for the tile kernel.
template <int MODE>
__global__ __launch_bounds__(256) void cost_wide_kernel(const DmSetArray sets, const int nsets, const int B, const int C,
                                                        const int64_t *__restrict__ labels,
                                                        const float *__restrict__ tboxes,
                                                        const int32_t *__restrict__ tgt_off, const DmParams prm,
                                                        float *__restrict__ cost, int32_t *status)
{
    constexpr int W = 4;
    __shared__ TgtRec tg[TCHUNK];
    __shared__ float part[128];           // softmax statistics of the tile's rows: maximum | -1 / sum

    const int b = blockIdx.y;
    const DmSet &set = sets.s[blockIdx.z];
    const int Q = set.Q;
    const int q0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = q0 + lane;
    const bool active = q < Q;
    const int t0 = tgt_off[b];
    const int T = tgt_off[b + 1] - t0;
    const int row = active ? q : Q - 1;
    const PredBox p = make_pred(image_boxes(set, b)[row]);
    if (active && wave == 0 && !pred_valid(p)) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    if (T == 0) return;

    const float *img_logits = image_logits(set, b, C);
    const float *xr = img_logits + (size_t)row * (size_t)C;
    constexpr float LOG2E = 1.44269504088896f;
    float ce_max = 0.0f, ce_scale = 1.0f;
    if (MODE == 0) {
        const int sub = lane & 15;
        for (int g = 0; g < 16 / W; ++g) {
            const int r = 4 * wave + 4 * W * g + (lane >> 4);
            const float *xrow = img_logits + (size_t)min(q0 + r, Q - 1) * (size_t)C;
            float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
            int c = sub;
            for (; c + 48 < C; c += 64) {
                m0 = vmax(m0, xrow[c]); m1 = vmax(m1, xrow[c + 16]); m2 = vmax(m2, xrow[c + 32]); m3 = vmax(m3, xrow[c + 48]);
            }
            for (; c < C; c += 16) m0 = vmax(m0, xrow[c]);
            const float mx = row16_all<false>(vmax(vmax(m0, m1), vmax(m2, m3)));
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            c = sub;
            for (; c + 48 < C; c += 64) {
                s0 += __builtin_amdgcn_exp2f((xrow[c] - mx) * LOG2E);
                s1 += __builtin_amdgcn_exp2f((xrow[c + 16] - mx) * LOG2E);
                s2 += __builtin_amdgcn_exp2f((xrow[c + 32] - mx) * LOG2E);
                s3 += __builtin_amdgcn_exp2f((xrow[c + 48] - mx) * LOG2E);
            }
            for (; c < C; c += 16) s0 += __builtin_amdgcn_exp2f((xrow[c] - mx) * LOG2E);
            const float sum = row16_all<true>((s0 + s1) + (s2 + s3));
            if (sub == 0) {
                part[r] = mx;
                part[64 + r] = -1.0f / sum;       // class cost = -softmax
            }
        }
        __syncthreads();
        ce_max = part[lane];
        ce_scale = part[64 + lane];
    }

    const int qs = set.q_stride;
    float *out = cost + set.cost_off + (size_t)t0 * qs;
    for (int tc = 0; tc < T; tc += TCHUNK) {
        const int nt = min(TCHUNK, T - tc);
        if (tc > 0) __syncthreads();
        stage_targets(tg, labels, tboxes, t0 + tc, nt, C, status);
        __syncthreads();
        if (active) {
            for (int i = wave; i < nt; i += W) {
                const TgtRec g = tg[i];
                const float xl = xr[g.label];
                float cls;
                if (MODE == 0) cls = __builtin_amdgcn_exp2f((xl - ce_max) * LOG2E) * ce_scale;
                else cls = focal_cost<MODE == 1>(xl, prm.focal_alpha, prm.focal_gamma);
                out[(size_t)(tc + i) * qs + q] = pair_cost(p, g, prm.w_class * cls, prm.w_bbox, prm.w_giou);
            }
        }
    }
}

// waves per workgroup: fewest padded lanes over the tiles of one image, larger blocks on ties
int pick_waves(int Q, int max_waves)
{
    const int need = (Q + 63) / 64;
    int best_k = 1, best_total = 1 << 30;
    for (int k = max_waves; k >= 1; --k) {
        const int total = ((need + k - 1) / k) * k;
        if (total < best_total) { best_total = total; best_k = k; }
    }
    return best_k;
}

thread_local char g_hip_err[256] = "";
int g_cost_tsplit = 0;   // 0: automatic; otherwise forced target split (testing aid)
int g_tile_waves = 0;     // 0: automatic; 4 / 8: wavefronts per workgroup of the tile kernel (testing aid)
int g_cost_qpt = 0;   // 0: automatic; 1 / 2: force the queries-per-thread variant (testing aid)

}  // namespace

extern "C" const char *dm_last_hip_error(void) { return g_hip_err; }

namespace dm {
// launch-shape knobs of the cost kernels (dm_debug_set_knob); 0 = automatic
int set_cost_knob(int knob, int value)
{
    switch (knob) {
    case DM_KNOB_COST_QPT:
        if (value < 0 || value > 2) return DM_E_INVALID_ARG;
        g_cost_qpt = value;
        return DM_OK;
    case DM_KNOB_COST_TSPLIT:
        if (value != 0 && value != 1 && value != 2 && value != 4) return DM_E_INVALID_ARG;
        g_cost_tsplit = value;
        return DM_OK;
    case DM_KNOB_TILE_WAVES:
        if (value != 0 && value != 4 && value != 8) return DM_E_INVALID_ARG;
        g_tile_waves = value;
        return DM_OK;
    }
    return DM_E_INVALID_ARG;
}
}  // namespace dm

namespace dm {
int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return DM_OK;
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return DM_E_LAUNCH;
}

// Kernels that may need more than 64 KB of dynamic LDS: the attribute is raised once per (function, device)
// to the largest size asked for so far; the record of what has been set is shared by all host threads.
int allow_lds(const void *fn, size_t bytes, const char *what)
{
    struct Entry { const void *fn; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Entry> set_on;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    std::lock_guard<std::mutex> lock(mu);
    Entry *mine = nullptr;
    for (auto &e : set_on)
        if (e.fn == fn && e.dev == dev) { mine = &e; break; }
    if (mine && mine->bytes >= bytes) return DM_OK;
    int rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), what);
    if (rc == DM_OK && dev >= 0) {
        if (mine) mine->bytes = bytes;
        else set_on.push_back(Entry{fn, dev, bytes});
    }
    return rc;
}
}  // namespace dm

extern "C" int dm_cost_f32(const DmSet *sets, int S, int B, int C, const int64_t *tgt_labels,
                           const float *tgt_boxes, const int32_t *tgt_off, int max_T,
                           const DmParams *params, float *cost, int32_t *status, void *stream)
{
    if (S < 0 || B < 0 || C < 1 || max_T < 0 || !params || !status) return DM_E_INVALID_ARG;
    if (S == 0 || B == 0) return DM_OK;
    if (!sets || !tgt_off || (max_T > 0 && (!tgt_labels || !tgt_boxes || !cost))) return DM_E_INVALID_ARG;
    if (B > 65535 || S > 4096) return DM_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tile = C > 4;
    // tile kernel: LDS = staged targets + softmax statistics (softmax variant) + the logit tile
    const int mode = params->cls_mode == 0 ? 0 : (params->focal_gamma == 2.0f ? 1 : 2);
    const int tile_waves = g_tile_waves == 8 ? 8 : 4;
    const size_t tile_lds = TCHUNK * sizeof(TgtRec) + (mode == 0 ? 64 * 2 * sizeof(float) : 0) +
                            (size_t)64 * C * sizeof(float);
    // (160 KB is the CU's LDS; the tile kernel is used while a workgroup's share stays within it)
    const bool wide = tile && tile_lds > 160 * 1024;
    typedef void (*TileFn)(const DmSetArray, const int, const int, const int, const int64_t *, const float *,
                           const int32_t *, const DmParams, float *, int32_t *, const int);
#define DM_TILE_FN(WW) (mode == 0 ? cost_tile_kernel<WW, 0> : mode == 1 ? cost_tile_kernel<WW, 1> : cost_tile_kernel<WW, 2>)
    const TileFn tile_fn = tile_waves == 4 ? DM_TILE_FN(4) : DM_TILE_FN(8);
#undef DM_TILE_FN
    if (tile && !wide && tile_lds > 64 * 1024) {
        int rc = dm::allow_lds(reinterpret_cast<const void *>(tile_fn), tile_lds, "hipFuncSetAttribute(cost_tile_kernel)");
        if (rc != DM_OK) return rc;
    }

    // one launch per group of sets with equal Q (the tile decomposition depends on Q)
    bool done[4096] = {false};
    for (int s0 = 0; s0 < S; ++s0) {
        if (done[s0]) continue;
        const int Q = sets[s0].Q;
        DmSetArray arr;
        int n = 0;
        for (int s = s0; s <= S; ++s) {
            if (s < S && !done[s] && sets[s].Q == Q) {
                if (Q < 1 || sets[s].q_stride < Q || !sets[s].logits || !sets[s].boxes || !dm::strides_ok(sets[s], C))
                    return DM_E_INVALID_ARG;
                arr.s[n++] = sets[s];
                done[s] = true;
            }
            if (n == DM_MAX_SETS_PER_LAUNCH || (s == S && n > 0)) {
                if (wide) {
                    const dim3 grid((unsigned)((Q + 63) / 64), (unsigned)B, (unsigned)n);
                    if (mode == 0)
                        hipLaunchKernelGGL(cost_wide_kernel<0>, grid, dim3(256), 0, st, arr, n, B, C, tgt_labels, tgt_boxes,
                                           tgt_off, *params, cost, status);
                    else if (mode == 1)
                        hipLaunchKernelGGL(cost_wide_kernel<1>, grid, dim3(256), 0, st, arr, n, B, C, tgt_labels, tgt_boxes,
                                           tgt_off, *params, cost, status);
                    else
                        hipLaunchKernelGGL(cost_wide_kernel<2>, grid, dim3(256), 0, st, arr, n, B, C, tgt_labels, tgt_boxes,
                                           tgt_off, *params, cost, status);
                } else if (tile) {
                    const int ntiles = (Q + 63) / 64;
                    hipLaunchKernelGGL(tile_fn, dim3((unsigned)ntiles, (unsigned)B, (unsigned)n), dim3(64 * tile_waves),
                                       tile_lds, st, arr, n, B, C, tgt_labels, tgt_boxes, tgt_off, *params, cost, status,
                                       ntiles);
                } else {
                    // one query per thread keeps the most wavefronts in flight; the two-query
                    // variant (8-byte stores) measured no faster at any size and stays a testing aid
                    const int qpt = g_cost_qpt ? g_cost_qpt : 1;
                    const int lanes = (Q + qpt - 1) / qpt;
                    const int threads = 64 * pick_waves(lanes, 8);
                    // few wavefronts in flight: split every image's targets over 2 or 4 workgroups
                    const long long waves_total = (long long)((lanes + 63) / 64) * B * n;
                    int tsplit = 1;      // measured on MI355X: 4-way below ~1k wavefronts, 2-way below ~6k
                    if (waves_total <= 1024 && max_T >= 8) tsplit = 4;
                    else if (waves_total <= 6144 && max_T >= 4) tsplit = 2;
                    if (g_cost_tsplit) tsplit = g_cost_tsplit;
                    dim3 grid(((lanes + threads - 1) / threads) * tsplit, B, n);
#define DM_LAUNCH_SMALL(CC, QQ)                                                                                  \
    hipLaunchKernelGGL((cost_small_kernel<CC, QQ>), grid, dim3(threads), 0, st, arr, tgt_labels, tgt_boxes, tgt_off, \
                       *params, cost, status, tsplit)
                    if (qpt == 1) {
                        switch (C) {
                        case 1: DM_LAUNCH_SMALL(1, 1); break;
                        case 2: DM_LAUNCH_SMALL(2, 1); break;
                        case 3: DM_LAUNCH_SMALL(3, 1); break;
                        default: DM_LAUNCH_SMALL(4, 1); break;
                        }
                    } else {
                        switch (C) {
                        case 1: DM_LAUNCH_SMALL(1, 2); break;
                        case 2: DM_LAUNCH_SMALL(2, 2); break;
                        case 3: DM_LAUNCH_SMALL(3, 2); break;
                        default: DM_LAUNCH_SMALL(4, 2); break;
                        }
                    }
#undef DM_LAUNCH_SMALL
                }
                int rc = dm::check_hip(hipGetLastError(), "cost kernel launch");
                if (rc != DM_OK) return rc;
                n = 0;
            }
        }
    }
    return DM_OK;
}
