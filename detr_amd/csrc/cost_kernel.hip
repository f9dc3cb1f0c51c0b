// Fused matching-cost kernel for gfx950 (MI355X).
//
// Replaces, for the B per-image blocks the matcher consumes, the chain of whole-batch
// torch ops at /root/reference/models/matcher.py:59-90 and
// /root/reference/utils/box_ops.py:17-21, 32-45, 48-69: class cost (softmax gather or
// focal), L1 (torch.cdist p=1), GIoU, weighted sum -- one pass, one write per cost entry.
//
// Mapping: one thread per query, one workgroup per (query tile, image, set).  The
// image's targets are staged once per workgroup in LDS as pre-converted records
// (xyxy, area, cxcywh, label) and broadcast-read by every lane; the cost row of a
// target is written with consecutive lanes on consecutive queries (coalesced).
// Logits with C <= 4 go straight to registers; wider logit rows are streamed with
// 16-byte loads into a wave-private LDS tile with an odd row stride so that each
// lane can read its own row bank-conflict-free.
//
// The kernel is HBM-bound streaming work (no contraction, so no MFMA).  It is compiled
// with -ffp-contract=off and evaluates every expression in the reference's order.
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

__device__ __forceinline__ float pow_gamma(float x, float gamma)
{
    // torch computes pow(x, 2.0) as x * x
    if (gamma == 2.0f) return x * x;
    if (gamma == 1.0f) return x;
    if (gamma == 0.0f) return 1.0f;
    return powf(x, gamma);
}

// matcher.py:76-81 for one logit
__device__ __forceinline__ float focal_cost(float x, float alpha, float gamma)
{
    const float p = 1.0f / (1.0f + expf(-x));
    const float neg = ((1.0f - alpha) * pow_gamma(p, gamma)) * (-logf((1.0f - p) + 1e-8f));
    const float pos = (alpha * pow_gamma(1.0f - p, gamma)) * (-logf(p + 1e-8f));
    return pos - neg;
}

struct PredBox {
    float cx, cy, w, h, x0, y0, x1, y1, area;
};

// matcher.py:84-90 and box_ops.py:32-45, 61-69 for one (query, target) pair
__device__ __forceinline__ float pair_cost(const PredBox &p, const TgtRec &g, float cls, const DmParams &prm)
{
    const float l1 = ((fabsf(p.cx - g.cx) + fabsf(p.cy - g.cy)) + fabsf(p.w - g.w)) + fabsf(p.h - g.h);
    const float iw = fmaxf(fminf(p.x1, g.x1) - fmaxf(p.x0, g.x0), 0.0f);
    const float ih = fmaxf(fminf(p.y1, g.y1) - fmaxf(p.y0, g.y0), 0.0f);
    const float inter = iw * ih;
    const float uni = (p.area + g.area) - inter;
    const float iou = inter / uni;
    const float ew = fmaxf(fmaxf(p.x1, g.x1) - fminf(p.x0, g.x0), 0.0f);
    const float eh = fmaxf(fmaxf(p.y1, g.y1) - fminf(p.y0, g.y0), 0.0f);
    const float earea = ew * eh;
    const float giou = iou - (earea - uni) / earea;
    return ((prm.w_bbox * l1) + (prm.w_class * cls)) + (prm.w_giou * (-giou));
}

template <bool TILE>
__global__ void cost_kernel(const DmSetArray sets, const int C, const int64_t *__restrict__ labels,
                            const float *__restrict__ tboxes, const int32_t *__restrict__ tgt_off,
                            const DmParams prm, float *__restrict__ cost, int32_t *status)
{
    extern __shared__ __align__(16) unsigned char smem[];
    TgtRec *tg = reinterpret_cast<TgtRec *>(smem);

    const DmSet &set = sets.s[blockIdx.z];
    const int b = blockIdx.y;
    const int Q = set.Q;
    const int q0 = blockIdx.x * blockDim.x;
    if (q0 >= Q) return;
    const int q = q0 + threadIdx.x;
    const bool active = q < Q;
    const int t0 = tgt_off[b];
    const int T = tgt_off[b + 1] - t0;
    const size_t row = (size_t)b * Q + (active ? q : Q - 1);

    // own prediction box; the reference validates every prediction of the batch
    const float4 raw = reinterpret_cast<const float4 *>(set.boxes)[row];
    PredBox p;
    p.cx = raw.x; p.cy = raw.y; p.w = raw.z; p.h = raw.w;
    p.x0 = p.cx - 0.5f * p.w; p.y0 = p.cy - 0.5f * p.h;
    p.x1 = p.cx + 0.5f * p.w; p.y1 = p.cy + 0.5f * p.h;
    p.area = (p.x1 - p.x0) * (p.y1 - p.y0);
    if (active && !((p.x1 >= p.x0) && (p.y1 >= p.y0))) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    if (T == 0) return;

    const int Cs = C | 1;                      // odd LDS row stride
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *wt = reinterpret_cast<float *>(smem + TCHUNK * sizeof(TgtRec)) + (size_t)wave * 64 * Cs;

    float cv0 = 0.f, cv1 = 0.f, cv2 = 0.f, cv3 = 0.f;  // per-class cost, C <= 4
    float ce_max = 0.f, ce_den = 1.f;                   // softmax pieces, C > 4

    if (!TILE) {
        const float *lg = set.logits + row * (size_t)C;
        float x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) x[c] = (c < C) ? lg[c] : 0.0f;
        if (prm.cls_mode == 0) {
            float mx = x[0];
#pragma unroll
            for (int c = 1; c < 4; ++c) if (c < C) mx = fmaxf(mx, x[c]);
            float e[4], den = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) { e[c] = (c < C) ? expf(x[c] - mx) : 0.0f; if (c < C) den += e[c]; }
            cv0 = -(e[0] / den); cv1 = -(e[1] / den); cv2 = -(e[2] / den); cv3 = -(e[3] / den);
        } else {
            cv0 = focal_cost(x[0], prm.focal_alpha, prm.focal_gamma);
            if (C > 1) cv1 = focal_cost(x[1], prm.focal_alpha, prm.focal_gamma);
            if (C > 2) cv2 = focal_cost(x[2], prm.focal_alpha, prm.focal_gamma);
            if (C > 3) cv3 = focal_cost(x[3], prm.focal_alpha, prm.focal_gamma);
        }
    } else {
        // stream this wave's 64 logit rows (contiguous in memory) into its LDS tile
        const int qw0 = q0 + wave * 64;
        const int nq = min(64, Q - qw0);
        if (nq > 0) {
            const float *src = set.logits + ((size_t)b * Q + qw0) * (size_t)C;
            const int n = nq * C;
            const int step_r = 256 / C, step_c = 256 % C;
            int e0 = lane * 4;
            int r = e0 / C, c = e0 - r * C;
            const bool vec_ok = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
            constexpr int U = 8;   // 16-byte loads kept in flight per lane
            for (int base = 0; base < n; base += 256 * U) {
                float4 v[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int e = base + k * 256 + lane * 4;
                    if (vec_ok && e + 3 < n) {
                        v[k] = *reinterpret_cast<const float4 *>(src + e);
                    } else {
                        v[k].x = (e < n) ? src[e] : 0.f;
                        v[k].y = (e + 1 < n) ? src[e + 1] : 0.f;
                        v[k].z = (e + 2 < n) ? src[e + 2] : 0.f;
                        v[k].w = (e + 3 < n) ? src[e + 3] : 0.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int e = base + k * 256 + lane * 4;
                    const float vv[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                    int rr = r, cc = c;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (e + i < n) wt[rr * Cs + cc] = vv[i];
                        if (++cc == C) { cc = 0; ++rr; }
                    }
                    c += step_c; r += step_r;
                    if (c >= C) { c -= C; ++r; }
                }
            }
        }
        __syncthreads();
        if (prm.cls_mode == 0) {
            const float *xr = wt + lane * Cs;
            float mx = xr[0];
            for (int c = 1; c < C; ++c) mx = fmaxf(mx, xr[c]);
            float den = 0.0f;
            for (int c = 0; c < C; ++c) den += expf(xr[c] - mx);
            ce_max = mx; ce_den = den;
        }
    }

    float *out = cost + set.cost_off + (size_t)t0 * set.q_stride + q;

    for (int tc = 0; tc < T; tc += TCHUNK) {
        const int nt = min(TCHUNK, T - tc);
        if (tc > 0) __syncthreads();
        for (int i = threadIdx.x; i < nt; i += blockDim.x) {
            const int t = t0 + tc + i;
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
        __syncthreads();
        if (active) {
#pragma unroll 4
            for (int i = 0; i < nt; ++i) {
                const TgtRec g = tg[i];
                float cls;
                if (!TILE) {
                    cls = g.label == 0 ? cv0 : (g.label == 1 ? cv1 : (g.label == 2 ? cv2 : cv3));
                } else {
                    const float x = wt[lane * Cs + g.label];
                    cls = prm.cls_mode == 0 ? -(expf(x - ce_max) / ce_den)
                                            : focal_cost(x, prm.focal_alpha, prm.focal_gamma);
                }
                out[(size_t)(tc + i) * set.q_stride] = pair_cost(p, g, cls, prm);
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

}  // namespace

extern "C" const char *dm_last_hip_error(void) { return g_hip_err; }

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
    if (B > 65535) return DM_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tile = C > 4;
    const int Cs = C | 1;
    const size_t tile_wave_bytes = tile ? (size_t)64 * Cs * sizeof(float) : 0;
    if (tile_wave_bytes + TCHUNK * sizeof(TgtRec) > 160 * 1024) return DM_E_UNSUPPORTED;
    int max_waves = 8;
    if (tile) {
        max_waves = (int)((48 * 1024) / tile_wave_bytes);
        if (max_waves < 1) max_waves = 1;
        if (max_waves > 4) max_waves = 4;
    }

    // one launch per group of sets with equal Q (the tile decomposition depends on Q)
    bool done[4096] = {false};
    if (S > 4096) return DM_E_UNSUPPORTED;
    for (int s0 = 0; s0 < S; ++s0) {
        if (done[s0]) continue;
        const int Q = sets[s0].Q;
        if (Q < 1 || sets[s0].q_stride < Q || !sets[s0].logits || !sets[s0].boxes) return DM_E_INVALID_ARG;
        DmSetArray arr;
        int n = 0;
        for (int s = s0; s <= S; ++s) {
            const bool take = s < S && !done[s] && sets[s].Q == Q;
            if (take) {
                if (sets[s].q_stride < Q || !sets[s].logits || !sets[s].boxes) return DM_E_INVALID_ARG;
                arr.s[n++] = sets[s];
                done[s] = true;
            }
            if (n == DM_MAX_SETS_PER_LAUNCH || (s == S && n > 0)) {
                const int waves = pick_waves(Q, max_waves);
                const int threads = waves * 64;
                const size_t lds = TCHUNK * sizeof(TgtRec) + tile_wave_bytes * waves;
                dim3 grid((Q + threads - 1) / threads, B, n);
                if (tile) {
                    if (lds > 64 * 1024) {
                        int rc = dm::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&cost_kernel<true>),
                                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                                               "hipFuncSetAttribute(cost_kernel)");
                        if (rc != DM_OK) return rc;
                    }
                    hipLaunchKernelGGL(cost_kernel<true>, grid, dim3(threads), lds, st, arr, C, tgt_labels,
                                       tgt_boxes, tgt_off, *params, cost, status);
                } else {
                    hipLaunchKernelGGL(cost_kernel<false>, grid, dim3(threads), lds, st, arr, C, tgt_labels,
                                       tgt_boxes, tgt_off, *params, cost, status);
                }
                int rc = dm::check_hip(hipGetLastError(), "cost_kernel launch");
                if (rc != DM_OK) return rc;
                n = 0;
            }
        }
    }
    return DM_OK;
}
