// Batched rectangular linear-sum-assignment for gfx950 (MI355X).
//
// Replaces the serial per-image scipy.optimize.linear_sum_assignment loop at
// /root/reference/models/matcher.py:93-95.  scipy's solver is the shortest-augmenting-
// path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment
// algorithms", IEEE TAES 2016 (no initialisation phase, fp64 duals, the matrix is
// transposed when it has more rows than columns).  Both kernels below run that
// algorithm with the same fp64 operation order and reproduce scipy's choice among
// tied minima, which depends on the position of a column in scipy's swap-with-last
// "remaining" list:
//     among the remaining columns of minimal reduced cost, take the unassigned one at
//     the LARGEST list position if there is one, else the one at the SMALLEST position.
//
// One wavefront solves one (set, image) problem; there is no inter-workgroup traffic.
//   lap_wave_kernel<NCOL>  T < Q <= 64*NCOL: every lane owns NCOL columns (queries) and
//                          keeps their reduced cost, dual, assignment and predecessor in
//                          registers; the column minimum is a DPP wave reduction; the
//                          first rows of the cost block are staged in LDS.
//   lap_generic_kernel     any shape up to 4096 x 4096 (also T >= Q, where scipy does
//                          not transpose): literal list-order scan with all state in LDS.
#include "dm_common.h"

namespace dm {
int check_hip(hipError_t e, const char *what);
}

namespace {

constexpr double INF = __builtin_huge_val();
constexpr int FAST_MAX_NCOL = 17;
constexpr int FAST_MAX_Q = 64 * FAST_MAX_NCOL;
constexpr int GENERIC_MAX = 4096;

int g_force_generic = 0;

__device__ __forceinline__ bool cost_is_invalid(float c)
{
    // scipy rejects NaN and -inf entries
    return (c != c) || (c == -__builtin_huge_valf());
}

// Writes the matched (query, target) pairs in ascending query order: lanes hold, for
// query j = lane + 64*k, the matched target (or -1) in r4c.
__device__ __forceinline__ void emit_pair(bool has, int query, int target, int &base, int lane,
                                          int64_t *src, int64_t *tgt)
{
    const unsigned long long bal = __ballot(has);
    if (has) {
        const int o = base + __popcll(bal & ((1ull << lane) - 1ull));
        src[o] = query;
        tgt[o] = target;
    }
    base += __popcll(bal);
}

template <int NCOL>
__global__ __launch_bounds__(64) void lap_wave_kernel(const DmSetArray sets, const int S,
                                                       const float *__restrict__ cost,
                                                       const int32_t *__restrict__ tgt_off,
                                                       const int32_t *__restrict__ order,
                                                       int64_t *__restrict__ src_out, int64_t *__restrict__ tgt_out,
                                                       int32_t *status, const int t_cap, const int lds_cost_floats)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x % S;
    const int bi = blockIdx.x / S;
    const int b = order ? order[bi] : bi;
    const DmSet &set = sets.s[s];
    const int nc = set.Q;                 // columns = queries
    const int t0 = tgt_off[b];
    const int nr = tgt_off[b + 1] - t0;   // rows = targets
    if (nr == 0 || nr >= nc || nc > 64 * NCOL) return;   // not this kernel's problem

    constexpr int QCAP = 64 * NCOL;
    double *u = reinterpret_cast<double *>(smem);          // [t_cap] row duals
    int2 *undo = reinterpret_cast<int2 *>(u + t_cap);      // [t_cap] (position, column) moved per removal
    int *col4row = reinterpret_cast<int *>(undo + t_cap);  // [t_cap]
    int *row4col = col4row + t_cap;                        // [QCAP]
    int *pathL = row4col + QCAP;                           // [QCAP]
    int *remaining = pathL + QCAP;                         // [QCAP] list position -> column
    int *pos = remaining + QCAP;                           // [QCAP] column -> list position
    float *crow = reinterpret_cast<float *>(pos + QCAP);   // [lds_cost_floats] first rows of the block

    const int qs = set.q_stride;
    const float *cg = cost + set.cost_off + (size_t)t0 * qs;
    const int R = min(nr, lds_cost_floats / nc);           // rows resident in LDS

    double v[NCOL], spc[NCOL];
    int r4c[NCOL], pth[NCOL];
    unsigned valid = 0;
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
        const int j = lane + 64 * k;
        v[k] = 0.0; spc[k] = INF; r4c[k] = -1; pth[k] = -1;
        if (j < nc) {
            valid |= 1u << k;
            row4col[j] = -1;
            remaining[j] = nc - 1 - j;
            pos[j] = nc - 1 - j;
        }
    }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }

    // stage the first R rows; every entry read is validated (the others are on first touch)
    bool bad = false;
    for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
            const int j = lane + 64 * k;
            if ((valid >> k) & 1u) {
                const float c = cg[(size_t)i * qs + j];
                bad |= cost_is_invalid(c);
                crow[i * nc + j] = c;
            }
        }
    }
    if (__any(bad)) {
        if (lane == 0) dm::raise_status(status, DM_ST_INVALID_COST);
        return;
    }
    __syncthreads();

    int nundo = 0;
    for (int cur = 0; cur < nr; ++cur) {
        // bring the remaining list back to its initial (reversed) order
        for (int e = lane; e < nundo; e += 64) {
            const int2 w = undo[e];
            remaining[w.x] = nc - 1 - w.x;
            pos[w.y] = nc - 1 - w.y;
        }
        nundo = 0;
#pragma unroll
        for (int k = 0; k < NCOL; ++k) spc[k] = INF;
        unsigned sc = 0;            // own columns already scanned (in the tree)
        double min_val = 0.0;
        int i = cur, n = nc, sink = -1;
        __syncthreads();

        while (true) {
            const double ui = u[i];
            const int jl = __builtin_amdgcn_readfirstlane(remaining[n > 0 ? n - 1 : 0]);  // column at the last position
            const bool in_lds = i < R;
            const bool first_touch = (i == cur) && !in_lds;   // row `cur` is read in full exactly here
            const unsigned live = valid & ~sc;
            double lmin = INF;
            bool badrow = false;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const int j = lane + 64 * k;
                float c = 0.0f;
                if ((valid >> k) & 1u) c = in_lds ? crow[i * nc + j] : cg[(size_t)i * qs + j];
                if (first_touch) badrow |= cost_is_invalid(c);
                const double r = ((min_val + (double)c) - ui) - v[k];
                const bool lv = (live >> k) & 1u;
                const bool better = lv && (r < spc[k]);
                spc[k] = better ? r : spc[k];
                pth[k] = better ? i : pth[k];
                lmin = lv ? fmin(lmin, spc[k]) : lmin;
            }
            if (first_touch && __any(badrow)) {
                if (lane == 0) dm::raise_status(status, DM_ST_INVALID_COST);
                return;
            }
            const double m = dm::wave_min_f64(lmin);
            if (!(m < INF)) {
                if (lane == 0) dm::raise_status(status, DM_ST_INFEASIBLE);
                return;
            }

            // which column attains the minimum
            int cnt = 0, cj = -1, cr = -1;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const bool hit = ((live >> k) & 1u) && (spc[k] == m);
                cnt += hit ? 1 : 0;
                cj = hit ? lane + 64 * k : cj;
                cr = hit ? r4c[k] : cr;
            }
            const unsigned long long has = __ballot(cnt > 0);
            const unsigned long long multi = __ballot(cnt > 1);
            int jw, rw, pw;
            if (__popcll(has) == 1 && multi == 0) {
                const int wl = __ffsll((long long)has) - 1;
                jw = dm::readlane_i32(cj, wl);
                rw = dm::readlane_i32(cr, wl);
                pw = __builtin_amdgcn_readfirstlane(pos[jw]);
            } else {
                // several columns tie: apply the list-position rule
                unsigned key = 0;
                cj = -1; cr = -1;
#pragma unroll
                for (int k = 0; k < NCOL; ++k) {
                    const bool hit = ((live >> k) & 1u) && (spc[k] == m);
                    if (hit) {
                        const int j = lane + 64 * k;
                        const int pp = pos[j];
                        const unsigned kk = (r4c[k] == -1) ? (unsigned)(nc + 1 + pp) : (unsigned)(nc - pp);
                        if (kk > key) { key = kk; cj = j; cr = r4c[k]; }
                    }
                }
                const unsigned K = dm::wave_max_u32(key);
                const unsigned long long win = __ballot(key == K);
                const int wl = __ffsll((long long)win) - 1;
                jw = dm::readlane_i32(cj, wl);
                rw = dm::readlane_i32(cr, wl);
                pw = (K > (unsigned)nc) ? (int)(K - (unsigned)nc - 1u) : (int)((unsigned)nc - K);
            }
            min_val = m;
            // swap-with-last removal of column jw from the remaining list
            --n;
            if (lane == 0) {
                remaining[pw] = jl;
                pos[jl] = pw;
                undo[nundo] = make_int2(pw, jl);
            }
            ++nundo;
            if ((jw & 63) == lane) sc |= 1u << (jw >> 6);
            if (rw == -1) { sink = jw; break; }
            i = rw;
        }

        // dual update, then publish the predecessors of the scanned columns
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
            if ((sc >> k) & 1u) {
                const int j = lane + 64 * k;
                const double d = min_val - spc[k];
                if (j != sink) u[r4c[k]] += d;
                v[k] -= d;
                pathL[j] = pth[k];
            }
        }
        if (lane == 0) u[cur] += min_val;
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            for (int guard = 0; guard <= nr; ++guard) {
                const int ii = pathL[j];
                row4col[j] = ii;
                const int jn = col4row[ii];
                col4row[ii] = j;
                j = jn;
                if (ii == cur) break;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
            if ((sc >> k) & 1u) r4c[k] = row4col[lane + 64 * k];
    }

    int64_t *so = src_out + set.out_off + t0;
    int64_t *to = tgt_out + set.out_off + t0;
    int base = 0;
#pragma unroll
    for (int k = 0; k < NCOL; ++k)
        emit_pair(((valid >> k) & 1u) && r4c[k] >= 0, lane + 64 * k, r4c[k], base, lane, so, to);
}

__global__ __launch_bounds__(64) void lap_generic_kernel(const DmSetArray sets, const int S,
                                                          const float *__restrict__ cost,
                                                          const int32_t *__restrict__ tgt_off,
                                                          const int32_t *__restrict__ order,
                                                          int64_t *__restrict__ src_out, int64_t *__restrict__ tgt_out,
                                                          int32_t *status, const int c_cap, const int r_cap,
                                                          const int all_problems)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int s = blockIdx.x % S;
    const int bi = blockIdx.x / S;
    const int b = order ? order[bi] : bi;
    const DmSet &set = sets.s[s];
    const int Q = set.Q;
    const int t0 = tgt_off[b];
    const int T = tgt_off[b + 1] - t0;
    if (T == 0) return;
    const bool transposed = T < Q;        // scipy transposes when the matrix has more rows (queries) than columns
    if (!all_problems && transposed && Q <= FAST_MAX_Q) return;   // lap_wave_kernel's problem
    const int nr = transposed ? T : Q;
    const int nc = transposed ? Q : T;

    double *spc = reinterpret_cast<double *>(smem);      // [c_cap]
    double *v = spc + c_cap;                              // [c_cap]
    double *u = v + c_cap;                                // [r_cap]
    int *row4col = reinterpret_cast<int *>(u + r_cap);    // [c_cap]
    int *path = row4col + c_cap;                          // [c_cap]
    int *remaining = path + c_cap;                        // [c_cap]
    int *col4row = remaining + c_cap;                     // [r_cap]

    const int qs = set.q_stride;
    const float *cg = cost + set.cost_off + (size_t)t0 * qs;
    const size_t rs = transposed ? (size_t)qs : 1, cs = transposed ? 1 : (size_t)qs;

    bool bad = false;
    for (int t = 0; t < T; ++t)
        for (int q = lane; q < Q; q += 64) bad |= cost_is_invalid(cg[(size_t)t * qs + q]);
    if (__any(bad)) {
        if (lane == 0) dm::raise_status(status, DM_ST_INVALID_COST);
        return;
    }
    for (int j = lane; j < nc; j += 64) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }

    for (int cur = 0; cur < nr; ++cur) {
        for (int j = lane; j < nc; j += 64) { spc[j] = INF; remaining[j] = nc - 1 - j; }
        double min_val = 0.0;
        int i = cur, n = nc, sink = -1;
        __syncthreads();
        while (true) {
            const double ui = u[i];
            double best = INF;
            int bit = -1;
            bool bun = false;
            for (int it = lane; it < n; it += 64) {
                const int j = remaining[it];
                const double c = (double)cg[(size_t)i * rs + (size_t)j * cs];
                const double r = ((min_val + c) - ui) - v[j];
                double sp = spc[j];
                if (r < sp) { spc[j] = r; path[j] = i; sp = r; }
                const bool un = row4col[j] == -1;
                if (sp < best || (sp == best && un)) { best = sp; bit = it; bun = un; }
            }
            const double m = dm::wave_min_f64(best);
            if (!(m < INF)) {
                if (lane == 0) dm::raise_status(status, DM_ST_INFEASIBLE);
                return;
            }
            unsigned key = 0;
            if (bit >= 0 && best == m) key = bun ? (unsigned)(nc + 1 + bit) : (unsigned)(nc - bit);
            const unsigned K = dm::wave_max_u32(key);
            const int pw = (K > (unsigned)nc) ? (int)(K - (unsigned)nc - 1u) : (int)((unsigned)nc - K);
            __syncthreads();
            const int jw = __builtin_amdgcn_readfirstlane(remaining[pw]);
            const int rw = __builtin_amdgcn_readfirstlane(row4col[jw]);
            --n;
            const int jl = __builtin_amdgcn_readfirstlane(remaining[n]);
            __syncthreads();
            if (lane == 0) { remaining[pw] = jl; remaining[n] = jw; }   // scanned columns collect behind position n
            __syncthreads();
            min_val = m;
            if (rw == -1) { sink = jw; break; }
            i = rw;
        }
        for (int it = n + lane; it < nc; it += 64) {
            const int j = remaining[it];
            const double d = min_val - spc[j];
            if (j != sink) u[row4col[j]] += d;
            v[j] -= d;
        }
        if (lane == 0) u[cur] += min_val;
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            for (int guard = 0; guard <= nr; ++guard) {
                const int ii = path[j];
                row4col[j] = ii;
                const int jn = col4row[ii];
                col4row[ii] = j;
                j = jn;
                if (ii == cur) break;
            }
        }
        __syncthreads();
    }

    int64_t *so = src_out + set.out_off + t0;
    int64_t *to = tgt_out + set.out_off + t0;
    if (transposed) {
        int base = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
            const int j = j0 + lane;
            const int r = j < nc ? row4col[j] : -1;
            emit_pair(r >= 0, j, r, base, lane, so, to);
        }
    } else {
        for (int i = lane; i < nr; i += 64) { so[i] = i; to[i] = col4row[i]; }
    }
}

int ncol_class(int Q)
{
    static const int classes[] = {2, 5, 10, 15, FAST_MAX_NCOL};
    for (int c : classes) if (Q <= 64 * c) return c;
    return 0;
}

template <int NCOL>
int launch_wave(const DmSetArray &arr, int n, int B, const float *cost, const int32_t *tgt_off,
                const int32_t *order, int max_T, int max_Q, int64_t *src_out, int64_t *tgt_out,
                int32_t *status, hipStream_t st)
{
    const int fast_T = max_T < max_Q ? max_T : max_Q - 1;     // largest row count this kernel can meet
    const int t_cap = (fast_T + 3) & ~3;
    const size_t fixed = (size_t)t_cap * (8 + 8 + 4) + (size_t)4 * 64 * NCOL * 4;
    size_t want = (size_t)fast_T * max_Q * sizeof(float);
    const size_t budget = 64 * 1024;
    if (fixed >= budget) return DM_E_UNSUPPORTED;
    if (fixed + want > budget) want = budget - fixed;
    want &= ~(size_t)15;
    hipLaunchKernelGGL(lap_wave_kernel<NCOL>, dim3((unsigned)n * B), dim3(64), fixed + want, st, arr, n, cost,
                       tgt_off, order, src_out, tgt_out, status, t_cap, (int)(want / sizeof(float)));
    return dm::check_hip(hipGetLastError(), "lap_wave_kernel launch");
}

}  // namespace

// Testing aid: route every problem through the list-order kernel (non-zero) or restore
// the default split between the two kernels (zero).
extern "C" void dm_debug_force_generic_lap(int on) { g_force_generic = on; }

extern "C" int dm_lap_f32(const DmSet *sets, int S, int B, const float *cost, const int32_t *tgt_off,
                          const int32_t *order, int max_T, int64_t *src_out, int64_t *tgt_out,
                          int32_t *status, void *stream)
{
    if (S < 0 || B < 0 || max_T < 0 || !status) return DM_E_INVALID_ARG;
    if (S == 0 || B == 0 || max_T == 0) return DM_OK;
    if (!sets || !cost || !tgt_off || !src_out || !tgt_out) return DM_E_INVALID_ARG;
    if (S > 4096) return DM_E_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s = 0; s < S; ++s) {
        const int Q = sets[s].Q;
        if (Q < 1 || sets[s].q_stride < Q) return DM_E_INVALID_ARG;
        const bool needs_generic = g_force_generic || Q > FAST_MAX_Q || max_T >= Q;
        if (needs_generic && (Q > GENERIC_MAX || max_T > GENERIC_MAX)) return DM_E_UNSUPPORTED;
    }

    static const int classes[] = {2, 5, 10, 15, FAST_MAX_NCOL};
    if (!g_force_generic) {
        for (int cls : classes) {
            DmSetArray arr;
            int n = 0, max_Q = 0;
            for (int s = 0; s <= S; ++s) {
                if (s < S && ncol_class(sets[s].Q) == cls) {
                    arr.s[n++] = sets[s];
                    if (sets[s].Q > max_Q) max_Q = sets[s].Q;
                }
                if (n == DM_MAX_SETS_PER_LAUNCH || (s == S && n > 0)) {
                    int rc;
                    switch (cls) {
                    case 2: rc = launch_wave<2>(arr, n, B, cost, tgt_off, order, max_T, max_Q, src_out, tgt_out, status, st); break;
                    case 5: rc = launch_wave<5>(arr, n, B, cost, tgt_off, order, max_T, max_Q, src_out, tgt_out, status, st); break;
                    case 10: rc = launch_wave<10>(arr, n, B, cost, tgt_off, order, max_T, max_Q, src_out, tgt_out, status, st); break;
                    case 15: rc = launch_wave<15>(arr, n, B, cost, tgt_off, order, max_T, max_Q, src_out, tgt_out, status, st); break;
                    default: rc = launch_wave<FAST_MAX_NCOL>(arr, n, B, cost, tgt_off, order, max_T, max_Q, src_out, tgt_out, status, st); break;
                    }
                    if (rc != DM_OK) return rc;
                    n = 0; max_Q = 0;
                }
            }
        }
    }

    // problems the register kernel does not take: T >= Q, or Q beyond its column capacity
    DmSetArray arr;
    int n = 0, max_Q = 0;
    for (int s = 0; s <= S; ++s) {
        if (s < S && (g_force_generic || sets[s].Q > FAST_MAX_Q || max_T >= sets[s].Q)) {
            arr.s[n++] = sets[s];
            if (sets[s].Q > max_Q) max_Q = sets[s].Q;
        }
        if (n == DM_MAX_SETS_PER_LAUNCH || (s == S && n > 0)) {
            int c_cap = max_Q > max_T ? max_Q : max_T;
            int r_cap = max_Q < max_T ? max_Q : max_T;
            c_cap = (c_cap + 3) & ~3;
            r_cap = (r_cap + 3) & ~3;
            const size_t lds = (size_t)c_cap * 28 + (size_t)r_cap * 12;
            if (lds > 160 * 1024) return DM_E_UNSUPPORTED;
            if (lds > 64 * 1024) {
                int rc = dm::check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(&lap_generic_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                                       "hipFuncSetAttribute(lap_generic_kernel)");
                if (rc != DM_OK) return rc;
            }
            hipLaunchKernelGGL(lap_generic_kernel, dim3((unsigned)n * B), dim3(64), lds, st, arr, n, cost, tgt_off,
                               order, src_out, tgt_out, status, c_cap, r_cap, g_force_generic ? 1 : 0);
            int rc = dm::check_hip(hipGetLastError(), "lap_generic_kernel launch");
            if (rc != DM_OK) return rc;
            n = 0; max_Q = 0;
        }
    }
    return DM_OK;
}
