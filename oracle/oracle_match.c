/*
 * CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the DETR bipartite-matching path, used as the
 * checker for the HIP kernels.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this file's shared object; the
 * product package (detr_amd/) never does.
 *
 * What is restated, and from where:
 *   - per-image matching cost block (class cost + L1 + GIoU, weighted sum):
 *       /root/reference/models/matcher.py:59-90
 *       /root/reference/utils/box_ops.py:17-21 (cxcywh -> xyxy)
 *       /root/reference/utils/box_ops.py:32-45 (IoU, union)
 *       /root/reference/utils/box_ops.py:48-69 (GIoU, validity asserts)
 *   - per-image linear sum assignment (matcher.py:93-95 calls
 *     scipy.optimize.linear_sum_assignment, a third-party compiled module whose
 *     source is not in the reference tree; scipy is not version-pinned by the
 *     reference, the build container holds scipy 1.15.3).  The solver below
 *     restates SciPy's published algorithm: D. F. Crouse, "On implementing 2D
 *     rectangular assignment algorithms", IEEE TAES 52(4), 2016 -- a
 *     shortest-augmenting-path (Jonker-Volgenant style) solver without an
 *     initialisation phase, run on the transposed matrix when it has more
 *     rows than columns.
 *
 * Parity pin: tests/test_oracle.py checks this file against the golden
 * vectors under tests/golden/ (produced by running the reference's own
 * matcher.py + box_ops.py in the build container, see
 * tests/golden/gen_golden.py) and against scipy itself.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off keeps every fp32 operation separately rounded, as the
 * reference's elementwise torch ops are.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_LABEL_RANGE 1   /* IndexError in the reference (matcher.py:73,81) */
#define ORC_ERR_DEGENERATE_BOX 2 /* AssertionError (box_ops.py:59-60) */
#define ORC_ERR_INVALID_COST 3  /* scipy ValueError: NaN or -inf entry */
#define ORC_ERR_INFEASIBLE 4    /* scipy ValueError: infeasible */
#define ORC_ERR_NOMEM 5

/* ------------------------------------------------------------------ */
/* box helpers: utils/box_ops.py                                        */
/* ------------------------------------------------------------------ */

/* box_ops.py:17-21 */
static void cxcywh_to_xyxy(const float *b, float *o)
{
    o[0] = b[0] - 0.5f * b[2];
    o[1] = b[1] - 0.5f * b[3];
    o[2] = b[0] + 0.5f * b[2];
    o[3] = b[1] + 0.5f * b[3];
}

/* box_ops.py:59-60 -- `(b[2:] >= b[:2]).all()`; a NaN compares false. */
static int box_is_valid(const float *x)
{
    return (x[2] >= x[0]) && (x[3] >= x[1]);
}

/* box_ops.py:32-45 and 61-69 for one (prediction, target) pair of xyxy boxes */
static float giou_pair(const float *a, const float *g)
{
    float area1 = (a[2] - a[0]) * (a[3] - a[1]);
    float area2 = (g[2] - g[0]) * (g[3] - g[1]);
    float ltx = fmaxf(a[0], g[0]), lty = fmaxf(a[1], g[1]);
    float rbx = fminf(a[2], g[2]), rby = fminf(a[3], g[3]);
    float w = rbx - ltx, h = rby - lty;
    if (w < 0.0f) w = 0.0f;          /* clamp(min=0); NaN stays NaN */
    if (h < 0.0f) h = 0.0f;
    float inter = w * h;
    float uni = (area1 + area2) - inter;
    float iou = inter / uni;
    float ex0 = fminf(a[0], g[0]), ey0 = fminf(a[1], g[1]);
    float ex1 = fmaxf(a[2], g[2]), ey1 = fmaxf(a[3], g[3]);
    float ew = ex1 - ex0, eh = ey1 - ey0;
    if (ew < 0.0f) ew = 0.0f;
    if (eh < 0.0f) eh = 0.0f;
    float earea = ew * eh;
    return iou - (earea - uni) / earea;
}

/* ------------------------------------------------------------------ */
/* class cost: matcher.py:68-81                                         */
/* ------------------------------------------------------------------ */

static float pow_gamma(float x, float gamma)
{
    /* torch's pow(x, 2.0) is computed as x*x */
    if (gamma == 2.0f) return x * x;
    if (gamma == 1.0f) return x;
    if (gamma == 0.0f) return 1.0f;
    return powf(x, gamma);
}

/* focal branch, matcher.py:76-81, for one logit */
static float focal_cost(float x, float alpha, float gamma)
{
    float p = 1.0f / (1.0f + expf(-x));
    float neg = ((1.0f - alpha) * pow_gamma(p, gamma)) * (-logf((1.0f - p) + 1e-8f));
    float pos = (alpha * pow_gamma(1.0f - p, gamma)) * (-logf(p + 1e-8f));
    return pos - neg;
}

/*
 * Cost block of ONE image: out[q*T + t], q-major like the reference's
 * C[i] slice of shape [Q, T_b] (matcher.py:91-94).
 *   logits [Q, C], boxes [Q, 4] cxcywh, labels [T] int64, tboxes [T, 4] cxcywh.
 *   cls_mode 0 = 'ce' (matcher.py:68-73), otherwise focal (matcher.py:74-81).
 * Returns ORC_OK or the first error the reference would raise.
 */
int orc_cost_block_f32(const float *logits, const float *boxes,
                       const int64_t *labels, const float *tboxes,
                       int Q, int T, int C, int cls_mode,
                       float w_class, float w_bbox, float w_giou,
                       float alpha, float gamma, float *out)
{
    int q, t, c;
    /* label range first: the class-cost gather precedes the box asserts */
    for (t = 0; t < T; t++)
        if (labels[t] < -(int64_t)C || labels[t] >= (int64_t)C) return ORC_ERR_LABEL_RANGE;
    for (q = 0; q < Q; q++) {
        float a[4];
        cxcywh_to_xyxy(boxes + 4 * q, a);
        if (!box_is_valid(a)) return ORC_ERR_DEGENERATE_BOX;
    }
    for (t = 0; t < T; t++) {
        float g[4];
        cxcywh_to_xyxy(tboxes + 4 * t, g);
        if (!box_is_valid(g)) return ORC_ERR_DEGENERATE_BOX;
    }
    for (q = 0; q < Q; q++) {
        const float *x = logits + (size_t)q * C;
        const float *b = boxes + 4 * q;
        float a[4];
        float mx = 0.0f, den = 0.0f;
        cxcywh_to_xyxy(b, a);
        if (cls_mode == 0) {
            mx = x[0];
            for (c = 1; c < C; c++) if (x[c] > mx) mx = x[c];
            for (c = 0; c < C; c++) den += expf(x[c] - mx);
        }
        for (t = 0; t < T; t++) {
            const float *gb = tboxes + 4 * t;
            float g[4];
            int64_t l = labels[t];
            float cc, l1, giou;
            if (l < 0) l += C;           /* python-style negative index */
            cxcywh_to_xyxy(gb, g);
            if (cls_mode == 0)
                cc = -(expf(x[l] - mx) / den);
            else
                cc = focal_cost(x[l], alpha, gamma);
            /* torch.cdist(p=1): sequential sum of |d| over the 4 coords */
            l1 = ((fabsf(b[0] - gb[0]) + fabsf(b[1] - gb[1])) + fabsf(b[2] - gb[2])) + fabsf(b[3] - gb[3]);
            giou = giou_pair(a, g);
            /* matcher.py:90: w_bbox*L1 + w_class*cls + w_giou*(-giou) */
            out[(size_t)q * T + t] = ((w_bbox * l1) + (w_class * cc)) + (w_giou * (-giou));
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------ */
/* rectangular linear sum assignment (what matcher.py:94 calls)         */
/* ------------------------------------------------------------------ */

typedef struct {
    int nr, nc;
    const double *cost;   /* [nr][nc], nr <= nc */
    double *u, *v, *spc;
    int *path, *col4row, *row4col, *remaining;
    unsigned char *SR, *SC;
} lsap_ws;

/* One shortest augmenting path from row `cur`; returns the sink column or -1. */
static int augmenting_path(lsap_ws *w, int cur, double *p_min)
{
    const int nc = w->nc;
    double min_val = 0.0;
    int num_remaining = nc, it, sink = -1, i = cur;

    for (it = 0; it < nc; it++) w->remaining[it] = nc - it - 1;
    memset(w->SR, 0, (size_t)w->nr);
    memset(w->SC, 0, (size_t)nc);
    for (it = 0; it < nc; it++) w->spc[it] = INFINITY;

    while (sink == -1) {
        int index = -1, j;
        double lowest = INFINITY;
        w->SR[i] = 1;
        for (it = 0; it < num_remaining; it++) {
            double r;
            j = w->remaining[it];
            r = min_val + w->cost[(size_t)i * nc + j] - w->u[i] - w->v[j];
            if (r < w->spc[j]) {
                w->path[j] = i;
                w->spc[j] = r;
            }
            /* among equal minima prefer a column that is still unassigned */
            if (w->spc[j] < lowest || (w->spc[j] == lowest && w->row4col[j] == -1)) {
                lowest = w->spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        j = w->remaining[index];
        if (w->row4col[j] == -1) sink = j;
        else i = w->row4col[j];
        w->SC[j] = 1;
        w->remaining[index] = w->remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

static int cmp_pair(const void *a, const void *b)
{
    const int *x = (const int *)a, *y = (const int *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return (x[1] > y[1]) - (x[1] < y[1]);
}

/*
 * Solve min sum cost[a[k]][b[k]] over a [nr][nc] row-major fp32 matrix
 * (widened to fp64 exactly, as scipy does).  Writes min(nr, nc) pairs;
 * `a` is ascending.  Returns ORC_OK / ORC_ERR_*.
 */
int orc_lsap_f32(const float *cost32, int nr, int nc, int64_t *a, int64_t *b)
{
    int transpose = nc < nr;
    int R = transpose ? nc : nr, Cn = transpose ? nr : nc;
    int i, j, cur, rc = ORC_OK;
    size_t n = (size_t)nr * (size_t)nc;
    lsap_ws w;
    double *cost;

    if (nr == 0 || nc == 0) return ORC_OK;

    cost = (double *)malloc(n * sizeof(double));
    if (!cost) return ORC_ERR_NOMEM;
    if (transpose) {
        for (i = 0; i < nr; i++)
            for (j = 0; j < nc; j++) cost[(size_t)j * nr + i] = (double)cost32[(size_t)i * nc + j];
    } else {
        for (i = 0; (size_t)i < n; i++) cost[i] = (double)cost32[i];
    }
    for (i = 0; (size_t)i < n; i++) {
        if (cost[i] != cost[i] || cost[i] == -INFINITY) { free(cost); return ORC_ERR_INVALID_COST; }
    }

    memset(&w, 0, sizeof(w));
    w.nr = R; w.nc = Cn; w.cost = cost;
    w.u = (double *)calloc((size_t)R, sizeof(double));
    w.v = (double *)calloc((size_t)Cn, sizeof(double));
    w.spc = (double *)malloc((size_t)Cn * sizeof(double));
    w.path = (int *)malloc((size_t)Cn * sizeof(int));
    w.col4row = (int *)malloc((size_t)R * sizeof(int));
    w.row4col = (int *)malloc((size_t)Cn * sizeof(int));
    w.remaining = (int *)malloc((size_t)Cn * sizeof(int));
    w.SR = (unsigned char *)malloc((size_t)R);
    w.SC = (unsigned char *)malloc((size_t)Cn);
    if (!w.u || !w.v || !w.spc || !w.path || !w.col4row || !w.row4col || !w.remaining || !w.SR || !w.SC) {
        rc = ORC_ERR_NOMEM;
        goto done;
    }
    for (i = 0; i < R; i++) w.col4row[i] = -1;
    for (j = 0; j < Cn; j++) { w.row4col[j] = -1; w.path[j] = -1; }

    for (cur = 0; cur < R; cur++) {
        double min_val = 0.0;
        int sink = augmenting_path(&w, cur, &min_val);
        if (sink < 0) { rc = ORC_ERR_INFEASIBLE; goto done; }

        /* dual update */
        w.u[cur] += min_val;
        for (i = 0; i < R; i++)
            if (w.SR[i] && i != cur) w.u[i] += min_val - w.spc[w.col4row[i]];
        for (j = 0; j < Cn; j++)
            if (w.SC[j]) w.v[j] -= min_val - w.spc[j];

        /* augment along the alternating path back to `cur` */
        j = sink;
        for (;;) {
            int tmp;
            i = w.path[j];
            w.row4col[j] = i;
            tmp = w.col4row[i]; w.col4row[i] = j; j = tmp;
            if (i == cur) break;
        }
    }

    if (transpose) {
        /* rows of the transposed problem are original columns: report the
         * pairs ordered by original row index (scipy: argsort of col4row) */
        int *pairs = (int *)malloc((size_t)R * 2 * sizeof(int));
        if (!pairs) { rc = ORC_ERR_NOMEM; goto done; }
        for (i = 0; i < R; i++) { pairs[2 * i] = w.col4row[i]; pairs[2 * i + 1] = i; }
        qsort(pairs, (size_t)R, 2 * sizeof(int), cmp_pair);
        for (i = 0; i < R; i++) { a[i] = pairs[2 * i]; b[i] = pairs[2 * i + 1]; }
        free(pairs);
    } else {
        for (i = 0; i < R; i++) { a[i] = i; b[i] = w.col4row[i]; }
    }

done:
    free(cost); free(w.u); free(w.v); free(w.spc); free(w.path); free(w.col4row);
    free(w.row4col); free(w.remaining); free(w.SR); free(w.SC);
    return rc;
}

/*
 * Whole matcher call for one output set, per-image evaluation
 * (matcher.py:59-95 restricted to the diagonal blocks it consumes).
 *   logits [B, Q, C], boxes [B, Q, 4]; targets packed: labels [M], tboxes [M, 4],
 *   tgt_off [B+1] prefix sums of T_b.
 *   cost_out (optional, may be NULL): concatenation of the [Q, T_b] blocks.
 *   src_out/tgt_out: concatenation of min(Q, T_b) pairs per image.
 * Error precedence follows the reference: label range, then box validity over
 * the WHOLE batch (the asserts see the flattened batch), then per-image LAP errors.
 */
int orc_match_f32(const float *logits, const float *boxes,
                  const int64_t *labels, const float *tboxes, const int64_t *tgt_off,
                  int B, int Q, int C, int cls_mode,
                  float w_class, float w_bbox, float w_giou, float alpha, float gamma,
                  float *cost_out, int64_t *src_out, int64_t *tgt_out)
{
    int b, q, rc;
    int64_t M = tgt_off[B], t;
    size_t cost_pos = 0, out_pos = 0;

    for (t = 0; t < M; t++)
        if (labels[t] < -(int64_t)C || labels[t] >= (int64_t)C) return ORC_ERR_LABEL_RANGE;
    for (q = 0; q < B * Q; q++) {
        float a[4];
        cxcywh_to_xyxy(boxes + 4 * (size_t)q, a);
        if (!box_is_valid(a)) return ORC_ERR_DEGENERATE_BOX;
    }
    for (t = 0; t < M; t++) {
        float g[4];
        cxcywh_to_xyxy(tboxes + 4 * (size_t)t, g);
        if (!box_is_valid(g)) return ORC_ERR_DEGENERATE_BOX;
    }
    for (b = 0; b < B; b++) {
        int T = (int)(tgt_off[b + 1] - tgt_off[b]);
        int npair = T < Q ? T : Q;
        float *blk;
        if (T == 0 || Q == 0) continue;
        blk = cost_out ? cost_out + cost_pos : (float *)malloc((size_t)Q * T * sizeof(float));
        if (!blk) return ORC_ERR_NOMEM;
        rc = orc_cost_block_f32(logits + (size_t)b * Q * C, boxes + (size_t)b * Q * 4,
                                labels + tgt_off[b], tboxes + 4 * tgt_off[b],
                                Q, T, C, cls_mode, w_class, w_bbox, w_giou, alpha, gamma, blk);
        if (rc == ORC_OK) rc = orc_lsap_f32(blk, Q, T, src_out + out_pos, tgt_out + out_pos);
        if (!cost_out) free(blk);
        if (rc != ORC_OK) return rc;
        cost_pos += (size_t)Q * T;
        out_pos += (size_t)npair;
    }
    return ORC_OK;
}
