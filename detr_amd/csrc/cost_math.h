// Per-pair arithmetic of the matching cost, shared by the cost kernels (cost_kernel.hip) and by the
// assignment kernels that compute their own cost blocks (lap_kernel.hip): both produce the same bits.
//
// Follows /root/reference/models/matcher.py:68-90 and /root/reference/utils/box_ops.py:17-21, 32-45,
// 48-69.  Compiled with -ffp-contract=off; the two divisions of the GIoU are evaluated as
// reciprocal * numerator (v_rcp_f32, 1 ulp) and the final sums use explicit fma.
#pragma once

#include "dm_common.h"

namespace dm {

struct __align__(16) TgtRec {
    float x0, y0, x1, y1;    // box_ops.py:17-21
    float cx, cy, w, h;
    float area;              // (x1 - x0) * (y1 - y0), torchvision box_area
    int label;               // wrapped into [0, C)
    int pad0, pad1;
};
static_assert(sizeof(TgtRec) == 48, "TgtRec is read as three 16-byte LDS loads");

// v_min_f32 / v_max_f32 without the canonicalising copies the compiler adds for possibly
// signalling NaN inputs: boxes with NaN coordinates are rejected through the status word.
__device__ __forceinline__ float fast_min(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fast_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float pow_gamma(float x, float gamma)
{
    // torch computes pow(x, 2.0) as x * x
    if (gamma == 2.0f) return x * x;
    if (gamma == 1.0f) return x;
    if (gamma == 0.0f) return 1.0f;
    return powf(x, gamma);
}

// matcher.py:76-81 for one logit, in the reference's own form: p = 1 / (1 + exp(-x)) rounded to fp32,
// then 1 - p from that rounded p (NOT the exactly complementary sigmoid(-x): for x > 9 the reference's
// log(1 - p + 1e-8) is a function of the last bits of p, up to p == 1 for x > 16.6, and the assignment
// has to see the same cost).  exp, reciprocal and log are the hardware v_exp_f32 / v_rcp_f32 /
// v_log_f32 (1 ulp each): p lands within a few ulp of torch's, like torch's own CPU and GPU kernels
// differ from one another.
// G2: gamma == 2 is known at compile time (keeps the general powf out of the instruction stream).
template <bool G2 = false>
__device__ __forceinline__ float focal_cost(float x, float alpha, float gamma)
{
    const float a = __builtin_amdgcn_exp2f(x * -1.44269504088896f);     // exp(-x); +inf for x < -88.7: p = 0
    const float p = __builtin_amdgcn_rcpf(1.0f + a);
    const float q = 1.0f - p;
    const float nlog_q = -(__builtin_amdgcn_logf(q + 1e-8f) * 0.693147180559945f);
    const float nlog_p = -(__builtin_amdgcn_logf(p + 1e-8f) * 0.693147180559945f);
    const float neg = ((1.0f - alpha) * (G2 ? p * p : pow_gamma(p, gamma))) * nlog_q;
    const float pos = (alpha * (G2 ? q * q : pow_gamma(q, gamma))) * nlog_p;
    return pos - neg;
}

struct PredBox {
    float cx, cy, w, h, x0, y0, x1, y1, area;
};

__device__ __forceinline__ PredBox make_pred(const float4 raw)
{
    PredBox p;
    p.cx = raw.x; p.cy = raw.y; p.w = raw.z; p.h = raw.w;
    p.x0 = p.cx - 0.5f * p.w; p.y0 = p.cy - 0.5f * p.h;
    p.x1 = p.cx + 0.5f * p.w; p.y1 = p.cy + 0.5f * p.h;
    p.area = (p.x1 - p.x0) * (p.y1 - p.y0);
    return p;
}

__device__ __forceinline__ bool pred_valid(const PredBox &p)
{
    return (p.x1 >= p.x0) && (p.y1 >= p.y0);   // false for NaN, like box_ops.py:59
}

// matcher.py:84-90 and box_ops.py:32-45, 61-69 for one (query, target) pair.
// `wcls` is cost_class * class cost.
__device__ __forceinline__ float pair_cost(const PredBox &p, const TgtRec &g, float wcls, float w_bbox, float w_giou)
{
    // (two-element vectors: the differences of an axis pair are one packed instruction each)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 dc = f2{p.cx, p.cy} - f2{g.cx, g.cy};
    const f2 ds = f2{p.w, p.h} - f2{g.w, g.h};
    const float l1 = ((fabsf(dc.x) + fabsf(dc.y)) + fabsf(ds.x)) + fabsf(ds.y);
    // per axis: (overlap, enclosing extent) = (min of the upper edges, max of them) - (max of the lower edges, min)
    f2 ex = f2{fast_min(p.x1, g.x1), fast_max(p.x1, g.x1)} - f2{fast_max(p.x0, g.x0), fast_min(p.x0, g.x0)};
    f2 ey = f2{fast_min(p.y1, g.y1), fast_max(p.y1, g.y1)} - f2{fast_max(p.y0, g.y0), fast_min(p.y0, g.y0)};
    // the overlap is clamped at zero; the enclosing box is never negative for valid boxes, so the
    // reference's clamp is a no-op there
    ex.x = fast_max(ex.x, 0.0f);
    ey.x = fast_max(ey.x, 0.0f);
    const f2 areas = ex * ey;
    const float inter = areas.x, earea = areas.y;
    const float uni = (p.area + g.area) - inter;
    // -giou = -(inter/uni - (earea - uni)/earea) = (1 - inter/uni) - uni/earea
    const float neg_giou = __builtin_fmaf(-uni, __builtin_amdgcn_rcpf(earea),
                                          __builtin_fmaf(-inter, __builtin_amdgcn_rcpf(uni), 1.0f));
    return __builtin_fmaf(w_bbox, l1, __builtin_fmaf(w_giou, neg_giou, wcls));
}

// One target, validated and pre-converted, into its LDS record.
__device__ __forceinline__ void store_target(TgtRec *slot, const float4 gb, long long l, int C, int32_t *status)
{
    TgtRec g;
    g.cx = gb.x; g.cy = gb.y; g.w = gb.z; g.h = gb.w;
    g.x0 = g.cx - 0.5f * g.w; g.y0 = g.cy - 0.5f * g.h;
    g.x1 = g.cx + 0.5f * g.w; g.y1 = g.cy + 0.5f * g.h;
    g.area = (g.x1 - g.x0) * (g.y1 - g.y0);
    if (!((g.x1 >= g.x0) && (g.y1 >= g.y0))) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    if (l < -(long long)C || l >= (long long)C) { dm::raise_status(status, DM_ST_LABEL_RANGE); l = 0; }
    if (l < 0) l += C;   // python-style negative index
    g.label = (int)l; g.pad0 = 0; g.pad1 = 0;
    *slot = g;
}


// cost_class * class cost of each of the C <= 4 classes of one query (matcher.py:68-81).
// x: the query's logits, entries beyond C are ignored.  The same operations in the same order whether
// C is a template constant of the caller or a run-time value.
__device__ __forceinline__ void small_class_costs(const float (&x)[4], const int C, const DmParams &prm, float (&wc)[4])
{
    if (prm.cls_mode == 0) {
        float mx = x[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) if (c < C) mx = fmaxf(mx, x[c]);
        float e[4], den = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) { e[c] = expf(x[c] - mx); den += e[c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) wc[c] = (c < C) ? prm.w_class * (-(e[c] / den)) : 0.0f;
    } else if (prm.focal_gamma == 2.0f) {
        // (a branch of its own: the general power function stays out of the usual path)
#pragma unroll
        for (int c = 0; c < 4; ++c) wc[c] = (c < C) ? prm.w_class * focal_cost<true>(x[c], prm.focal_alpha, 2.0f) : 0.0f;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) wc[c] = (c < C) ? prm.w_class * focal_cost<false>(x[c], prm.focal_alpha, prm.focal_gamma) : 0.0f;
    }
}

// What an assignment kernel that computes its own cost blocks needs besides the output sets.
// The targets of up to DIRECT_MAX_IMAGES images as the caller holds them -- one labels and one boxes tensor per
// image -- with the offsets and the processing order, all travelling in the kernel-argument segment: a
// kernel that takes them needs no packed copy of the targets and no upload.
constexpr int DIRECT_MAX_IMAGES = 128;      // (3076 bytes of the 4 KB a kernel may take as arguments)
struct DirectTargets {
    const int64_t *labels[DIRECT_MAX_IMAGES];
    const float *boxes[DIRECT_MAX_IMAGES];
    int32_t offsets[DIRECT_MAX_IMAGES + 1];
    int32_t order[DIRECT_MAX_IMAGES];
};

struct FusedIn {
    const int64_t *labels;   // device, packed target labels
    const float *tboxes;     // device, packed target boxes [M][4] cxcywh
    DmParams prm;
    int C;                   // logits per query, 1..4
    int direct;              // != 0: the targets are read through `table` instead of the packed arrays
    DirectTargets table;
};

// The cost block of problem (set, image b) computed by the workgroup that is going to solve it
// (`nthreads` threads, this one is `tid`; thread `tid` owns the columns (queries) tid + nthreads * k,
// k < NCOL).  Row i (target t0 + i) goes to lds_rows + i * ls for i < R and to spill + i * qs beyond;
// every entry is written by the thread that owns its column -- the only thread that reads it later.
// tl, tb: the image's nr labels and boxes.  tg: LDS room for the image's nr target records.  Returns whether this thread met an entry scipy
// rejects (NaN or -inf); raises the status bits of the cost kernel (label range, degenerate boxes).
// Ends with a barrier.
template <int NCOL>
__device__ __forceinline__ bool compute_cost_block(const DmSet &set, const FusedIn &fin, const int b, const int nc,
                                                   const int nr, const int64_t *tl, const float4 *tb, const int tid,
                                                   const int nthreads, TgtRec *tg, float *lds_rows, const int ls,
                                                   const int R, float *spill, const int qs, int32_t *status)
{
    const int C = fin.C;
    for (int i = tid; i < nr; i += nthreads) store_target(tg + i, tb[i], tl[i], C, status);
    __syncthreads();
    bool bad = false;
    const float4 *img_boxes = image_boxes(set, b);
    const float *img_logits = image_logits(set, b, C);
    // (Requesting the thread's first prediction ahead of the targets, to overlap the two memory latencies,
    // keeps the compiler from unrolling the pair loop below and measured 19 us slower at 384 problems.)
#pragma unroll 1
    for (int k = 0; k < NCOL; ++k) {
        const int j = tid + nthreads * k;
        if (j >= nc) break;
        const float4 raw = img_boxes[j];
        const float *lg = img_logits + (size_t)j * (size_t)C;
        float x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = c < C ? lg[c] : 0.0f;
        const PredBox p = make_pred(raw);
        if (!pred_valid(p)) raise_status(status, DM_ST_DEGENERATE_BOX);
        float wc[4];
        small_class_costs(x, C, fin.prm, wc);
        TgtRec g = tg[0];
        // four independent dependency chains per trip: a lone wavefront issues a dependent instruction
        // only every few cycles
#pragma unroll 4
        for (int i = 0; i < nr; ++i) {
            const TgtRec gn = tg[min(i + 1, nr - 1)];     // next record in flight while this one is used
            int l = g.label;
            asm("" : "+v"(l));                            // select by data, not by a scalar branch tree
            float wcls = wc[0];
#pragma unroll
            for (int c = 1; c < 4; ++c) wcls = (l == c) ? wc[c] : wcls;
            const float cst = pair_cost(p, g, wcls, fin.prm.w_bbox, fin.prm.w_giou);
            bad |= (cst != cst) || (cst == -__builtin_huge_valf());
            if (i < R) lds_rows[i * ls + j] = cst;
            else spill[(size_t)i * qs + j] = cst;
            g = gn;
        }
    }
    __threadfence_block();
    __syncthreads();
    return bad;
}

}  // namespace dm
