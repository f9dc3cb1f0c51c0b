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
//                      rows, streamed with coalesced loads into LDS (odd row stride, so each
//                      lane reads its own row conflict-free); the wavefronts split the
//                      targets four ways.
//
// HBM-bound streaming work: no contraction, so no MFMA.  Compiled with -ffp-contract=off;
// the two divisions of the GIoU are evaluated as reciprocal * numerator (v_rcp_f32, 1 ulp)
// and the final sums use explicit fma -- within 1e-7 relative of the reference's
// separately rounded operations, far inside the 1e-5 parity bar.
#include "dm_common.h"

namespace {

constexpr int TCHUNK = 64;   // targets staged per pass

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

// matcher.py:76-81 for one logit.
// sigmoid and its complement are formed from a = exp(-|x|) as 1/(1+a) and a/(1+a): the
// complement keeps full relative accuracy where the reference's fp32 `1 - p` has lost it
// (it stays within one rounding of p of the reference's value); exp and log are the
// hardware v_exp_f32 / v_log_f32 (1 ulp).
__device__ __forceinline__ float focal_cost(float x, float alpha, float gamma)
{
    const float a = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896f);
    const float big = __builtin_amdgcn_rcpf(1.0f + a);     // sigmoid(|x|)
    const float small = a * big;                            // sigmoid(-|x|)
    const float p = x >= 0.0f ? big : small;
    const float q = x >= 0.0f ? small : big;                // 1 - p
    const float nlog_q = -(__builtin_amdgcn_logf(q + 1e-8f) * 0.693147180559945f);
    const float nlog_p = -(__builtin_amdgcn_logf(p + 1e-8f) * 0.693147180559945f);
    const float neg = ((1.0f - alpha) * pow_gamma(p, gamma)) * nlog_q;
    const float pos = (alpha * pow_gamma(q, gamma)) * nlog_p;
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
    const float l1 = ((fabsf(p.cx - g.cx) + fabsf(p.cy - g.cy)) + fabsf(p.w - g.w)) + fabsf(p.h - g.h);
    const float iw = fast_max(fast_min(p.x1, g.x1) - fast_max(p.x0, g.x0), 0.0f);
    const float ih = fast_max(fast_min(p.y1, g.y1) - fast_max(p.y0, g.y0), 0.0f);
    const float inter = iw * ih;
    const float uni = (p.area + g.area) - inter;
    // enclosing box: never negative for valid boxes, so the reference's clamp is a no-op
    const float ew = fast_max(p.x1, g.x1) - fast_min(p.x0, g.x0);
    const float eh = fast_max(p.y1, g.y1) - fast_min(p.y0, g.y0);
    const float earea = ew * eh;
    // -giou = -(inter/uni - (earea - uni)/earea) = (1 - inter/uni) - uni/earea
    const float neg_giou = __builtin_fmaf(-uni, __builtin_amdgcn_rcpf(earea),
                                          __builtin_fmaf(-inter, __builtin_amdgcn_rcpf(uni), 1.0f));
    return __builtin_fmaf(w_bbox, l1, __builtin_fmaf(w_giou, neg_giou, wcls));
}

// Stage targets [tc, tc + nt) of the image in LDS (validated, pre-converted).
__device__ __forceinline__ void stage_targets(TgtRec *tg, const int64_t *__restrict__ labels,
                                              const float *__restrict__ tboxes, int first, int nt, int C,
                                              int32_t *status)
{
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
        const int t = first + i;
        const float4 gb = reinterpret_cast<const float4 *>(tboxes)[t];
        long long l = labels[t];
        TgtRec g;
        g.cx = gb.x; g.cy = gb.y; g.w = gb.z; g.h = gb.w;
        g.x0 = g.cx - 0.5f * g.w; g.y0 = g.cy - 0.5f * g.h;
        g.x1 = g.cx + 0.5f * g.w; g.y1 = g.cy + 0.5f * g.h;
        g.area = (g.x1 - g.x0) * (g.y1 - g.y0);
        if (!((g.x1 >= g.x0) && (g.y1 >= g.y0))) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
        if (l < -(long long)C || l >= (long long)C) { dm::raise_status(status, DM_ST_LABEL_RANGE); l = 0; }
        if (l < 0) l += C;   // python-style negative index
        g.label = (int)l; g.pad0 = 0; g.pad1 = 0;
        tg[i] = g;
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
    __shared__ TgtRec tg[TCHUNK];
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
    float x[QPT][C];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const size_t row = (size_t)b * Q + min(q0 + i, Q - 1);
        raw[i] = reinterpret_cast<const float4 *>(set.boxes)[row];
        const float *lg = set.logits + row * (size_t)C;
#pragma unroll
        for (int c = 0; c < C; ++c) x[i][c] = lg[c];
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
    float wc[QPT][C];
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        if (prm.cls_mode == 0) {
            float mx = x[i][0];
#pragma unroll
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[i][c]);
            float e[C], den = 0.0f;
#pragma unroll
            for (int c = 0; c < C; ++c) { e[c] = expf(x[i][c] - mx); den += e[c]; }
#pragma unroll
            for (int c = 0; c < C; ++c) wc[i][c] = prm.w_class * (-(e[c] / den));
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) wc[i][c] = prm.w_class * focal_cost(x[i][c], prm.focal_alpha, prm.focal_gamma);
        }
    }

    const int qs = set.q_stride;
    float *out = cost + set.cost_off + (size_t)t0 * qs + q0;
    // a QPT-wide store needs all its queries in range and natural alignment of the row start
    const bool wide = (q0 + QPT <= Q) && (qs % QPT == 0) && ((reinterpret_cast<uintptr_t>(out) & (4 * QPT - 1)) == 0);

    for (int tc = 0; tc < T; tc += TCHUNK) {
        const int nt = min(TCHUNK, T - tc);
        if (tc > 0) {
            __syncthreads();
            stage_targets(tg, labels, tboxes, t0 + tc, nt, C, status);
        }
        __syncthreads();
        if (q0 < Q) {
            TgtRec g = tg[0];
            for (int i = 0; i < nt; ++i) {
                const TgtRec gn = tg[min(i + 1, nt - 1)];     // next record in flight while this one is used
                // select by data, not by branch (the label is wave-uniform and the compiler would
                // otherwise turn the selects into a scalar branch tree)
                int l = g.label;
                asm("" : "+v"(l));
                vecq r;
#pragma unroll
                for (int k = 0; k < QPT; ++k) {
                    float wcls = wc[k][0];
#pragma unroll
                    for (int c = 1; c < C; ++c) wcls = (l == c) ? wc[k][c] : wcls;
                    r[k] = pair_cost(p[k], g, wcls, prm.w_bbox, prm.w_giou);
                }
                float *dst = out + (size_t)(tc + i) * qs;
                if (QPT == 1 || wide) {
                    *reinterpret_cast<vecq *>(dst) = r;
                } else {
#pragma unroll
                    for (int k = 0; k < QPT; ++k) if (q0 + k < Q) dst[k] = r[k];
                }
                g = gn;
            }
        }
    }
}

// C > 4.  blockDim = 256: wavefront w of the workgroup handles targets t = w (mod 4) of the
// 64 queries [64 * blockIdx.x, +64); all four stream the 64 x C logit tile into LDS together.
__global__ __launch_bounds__(256) void cost_tile_kernel(const DmSetArray sets, const int C,
                                                        const int64_t *__restrict__ labels,
                                                        const float *__restrict__ tboxes,
                                                        const int32_t *__restrict__ tgt_off, const DmParams prm,
                                                        float *__restrict__ cost, int32_t *status)
{
    extern __shared__ __align__(16) unsigned char smem[];
    TgtRec *tg = reinterpret_cast<TgtRec *>(smem);                                   // [TCHUNK]
    float *part = reinterpret_cast<float *>(smem + TCHUNK * sizeof(TgtRec));         // [4][64] softmax partials
    float *tile = part + 256;                                                        // [64][Cs]

    const DmSet &set = sets.s[blockIdx.z];
    const int b = blockIdx.y;
    const int Q = set.Q;
    const int q0 = blockIdx.x * 64;
    if (q0 >= Q) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = q0 + lane;
    const bool active = q < Q;
    const int nq = min(64, Q - q0);
    const int t0 = tgt_off[b];
    const int T = tgt_off[b + 1] - t0;
    const size_t row = (size_t)b * Q + (active ? q : Q - 1);
    const int Cs = C | 1;                      // odd LDS row stride

    const float4 raw = reinterpret_cast<const float4 *>(set.boxes)[row];
    if (T > 0) {
        // stream the tile: rows q0 .. q0+nq-1 are contiguous in memory; lane-consecutive dwords
        // land on consecutive LDS banks
        const float *src = set.logits + ((size_t)b * Q + q0) * (size_t)C;
        const int n = nq * C;
        if ((C & 1) && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            // odd C: the LDS image (row stride C) is a linear copy of the rows -> LDS-DMA, 1 KB per
            // wave instruction, no registers, every piece in flight at once
            for (int base = wave * 256; base < n; base += 4 * 256) {
                const int e = base + lane * 4;
                if (e + 3 < n)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + e),
                                                     (__attribute__((address_space(3))) void *)(tile + base), 16, 0, 0);
            }
            const int n4 = n & ~3;
            if (tid < n - n4) tile[n4 + tid] = src[n4 + tid];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int step_r = 256 / C, step_c = 256 % C;
            int r = tid / C, c = tid - r * C;
            constexpr int U = 8;
            for (int base = 0; base < n; base += 256 * U) {
                float v[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int e = base + k * 256 + tid;
                    v[k] = src[min(e, n - 1)];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int e = base + k * 256 + tid;
                    if (e < n) tile[r * Cs + c] = v[k];
                    c += step_c; r += step_r;
                    if (c >= C) { c -= C; ++r; }
                }
            }
        }
        stage_targets(tg, labels, tboxes, t0, min(TCHUNK, T), C, status);
    }
    const PredBox p = make_pred(raw);
    if (active && wave == 0 && !pred_valid(p)) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    if (T == 0) return;
    __syncthreads();

    const float *xr = tile + lane * Cs;
    float ce_max = 0.0f, ce_den = 1.0f;
    if (prm.cls_mode == 0) {
        // softmax statistics of the lane's row, the class range split over the four wavefronts
        float mx = -__builtin_huge_valf();
        for (int c = wave; c < C; c += 4) mx = fmaxf(mx, xr[c]);
        part[wave * 64 + lane] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(part[lane], part[64 + lane]), fmaxf(part[128 + lane], part[192 + lane]));
        __syncthreads();
        float den = 0.0f;
        for (int c = wave; c < C; c += 4) den += expf(xr[c] - mx);
        part[wave * 64 + lane] = den;
        __syncthreads();
        ce_max = mx;
        ce_den = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
    }

    const int qs = set.q_stride;
    float *out = cost + set.cost_off + (size_t)t0 * qs + q;

    for (int tc = 0; tc < T; tc += TCHUNK) {
        const int nt = min(TCHUNK, T - tc);
        if (tc > 0) {
            __syncthreads();
            stage_targets(tg, labels, tboxes, t0 + tc, nt, C, status);
            __syncthreads();
        }
        if (active) {
            for (int i = wave; i < nt; i += 4) {
                const TgtRec g = tg[i];
                const float xl = xr[g.label];
                const float cls = prm.cls_mode == 0 ? -(expf(xl - ce_max) / ce_den)
                                                    : focal_cost(xl, prm.focal_alpha, prm.focal_gamma);
                out[(size_t)(tc + i) * qs] = pair_cost(p, g, prm.w_class * cls, prm.w_bbox, prm.w_giou);
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
int g_cost_qpt = 0;   // 0: automatic; 1 / 2: force the queries-per-thread variant (testing aid)

}  // namespace

extern "C" const char *dm_last_hip_error(void) { return g_hip_err; }

// Testing aid: 0 restores the automatic choice, 1 / 2 force one / two queries per thread in
// the small-C cost kernel.
extern "C" void dm_debug_cost_queries_per_thread(int qpt)
{
    if (qpt >= 200) { g_cost_tsplit = qpt - 200; return; }   // 201, 202, 204: forced target split
    g_cost_qpt = (qpt == 1 || qpt == 2) ? qpt : 0;
}

namespace dm {
int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return DM_OK;
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return DM_E_LAUNCH;
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
    const size_t tile_lds = TCHUNK * sizeof(TgtRec) + 256 * sizeof(float) + (size_t)64 * (C | 1) * sizeof(float);
    if (tile && tile_lds > 160 * 1024) return DM_E_UNSUPPORTED;
    if (tile && tile_lds > 64 * 1024) {
        int rc = dm::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&cost_tile_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds),
                               "hipFuncSetAttribute(cost_tile_kernel)");
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
                if (Q < 1 || sets[s].q_stride < Q || !sets[s].logits || !sets[s].boxes) return DM_E_INVALID_ARG;
                arr.s[n++] = sets[s];
                done[s] = true;
            }
            if (n == DM_MAX_SETS_PER_LAUNCH || (s == S && n > 0)) {
                if (tile) {
                    dim3 grid((Q + 63) / 64, B, n);
                    hipLaunchKernelGGL(cost_tile_kernel, grid, dim3(256), tile_lds, st, arr, C, tgt_labels, tgt_boxes,
                                       tgt_off, *params, cost, status);
                } else {
                    // one query per thread keeps the most wavefronts in flight; the two-query
                    // variant (8-byte stores) measured no faster at any size and stays a testing aid
                    const int qpt = g_cost_qpt ? g_cost_qpt : 1;
                    const int lanes = (Q + qpt - 1) / qpt;
                    const int threads = 64 * pick_waves(lanes, 8);
                    // few wavefronts in flight: split every image's targets over 2 or 4 workgroups
                    const long long waves_total = (long long)((lanes + 63) / 64) * B * n;
                    int tsplit = 1;      // measured on MI355X: 4-way below ~1k wavefronts, 2-way below ~8k
                    if (waves_total <= 1024 && max_T >= 8) tsplit = 4;
                    else if (waves_total <= 8192 && max_T >= 4) tsplit = 2;
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
