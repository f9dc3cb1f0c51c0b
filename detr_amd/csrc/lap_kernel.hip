// Batched rectangular linear-sum-assignment for gfx950 (MI355X).
//
// Replaces the serial per-image scipy.optimize.linear_sum_assignment loop at
// /root/reference/models/matcher.py:93-95.  scipy's solver is the shortest-augmenting-
// path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment
// algorithms", IEEE TAES 2016 (no initialisation phase, fp64 duals, the matrix is
// transposed when it has more rows than columns).  The kernels below run that
// algorithm with the same fp64 operation order and reproduce scipy's choice among
// tied minima, which depends on the position of a column in scipy's swap-with-last
// "remaining" list:
//     among the remaining columns of minimal reduced cost, take the unassigned one at
//     the LARGEST list position if there is one, else the one at the SMALLEST position.
//
// Every (set, image) problem is solved by one workgroup; there is no inter-workgroup traffic.
//   lap_block_kernel<NCOL, WIDE>  T < Q <= 4096: a workgroup of 1..16 wavefronts per problem, a
//                          thread owns NCOL columns (queries) in registers; per Dijkstra step a
//                          DPP reduction per wavefront, one 128-bit record per wavefront in LDS,
//                          one barrier, a second reduction.  Few problems in flight: latency.
//   lap_wave_kernel<NCOL>  T < Q <= 64*NCOL <= 1088: one wavefront per problem, no barrier; every
//                          lane owns NCOL columns.  Many problems in flight: throughput.
//   lap_generic_kernel     any shape up to 4096 x 4096 (also T >= Q, where scipy does
//                          not transpose): literal list-order scan with all state in LDS.
// The first rows of a cost block are staged in LDS by LDS-DMA.
// FUSED variants of the two register kernels do not read a cost block at all: every thread computes
// the entries of its own columns (queries) from the predictions and the image's targets -- the
// arithmetic of the cost kernel (cost_math.h), bit for bit -- straight into LDS; rows that do not fit
// there go to the problem's region of the cost workspace and are read back by the thread that wrote them.
#include <algorithm>
#include <initializer_list>
#include <mutex>
#include <vector>
#include <type_traits>
#include <utility>

#include "cost_math.h"

namespace dm {
int check_hip(hipError_t e, const char *what);
}

namespace {

constexpr double INF = __builtin_huge_val();
constexpr int FAST_MAX_NCOL = 17;
constexpr int FAST_MAX_Q = 64 * FAST_MAX_NCOL;
constexpr int BLOCK_MAX_NCOL_WIDE = 4;                 // lap_block_kernel<NCOL <= 4> may use 1024 threads ...
constexpr int BLOCK_MAX_Q = 1024 * BLOCK_MAX_NCOL_WIDE;   // ... and then takes up to 4096 columns
constexpr int GENERIC_MAX = 4096;

int g_lap_wave_budget = 0;   // 0: automatic (12288); otherwise wavefronts a launch of lap_block_kernel may ask for
int g_lap_lds_kb = 0;        // 0: automatic; otherwise LDS budget per problem in KB (tuning knob)
int g_lap_max_waves = 16;   // wavefronts per problem in lap_block_kernel (tuning knob)
int g_lap_plain_shape = 0;   // testing aid.  1: one wavefront per 64-column slice up to 16; 2..16: exactly that many wavefronts
int g_lap_mode = 0;   // 0: automatic; 1: list-order kernel only; 2: workgroup kernel; 3: wavefront kernel

// Diagnostic build only (-DDM_LAP_STAMPS): cycle stamps around the segments of a step.
#ifdef DM_LAP_STAMPS
#define DM_STAMP(idx)                                                                              \
    {                                                                                              \
        unsigned long long t_;                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        st_seg[idx] += t_ - st_t;                                                                  \
        st_t = t_;                                                                                 \
    }
#else
#define DM_STAMP(idx)
#endif

__device__ __forceinline__ bool cost_is_invalid(float c)
{
    // scipy rejects NaN and -inf entries
    return (c != c) || (c == -__builtin_huge_valf());
}

// Writes the matched (query, target) pairs in ascending query order: lanes hold, for
// query j = lane + 64*k, the matched target (or -1) in r4c.
// flat (optional): the target's position among the packed targets of the batch, target + flat_base.
__device__ __forceinline__ void emit_pair(bool has, int query, int target, int &base, int lane,
                                          int64_t *src, int64_t *tgt, int64_t *flat = nullptr, int flat_base = 0)
{
    const unsigned long long bal = __ballot(has);
    if (has) {
        const int o = base + __popcll(bal & ((1ull << lane) - 1ull));
        src[o] = query;
        tgt[o] = target;
        if (flat) flat[o] = target + flat_base;
    }
    base += __popcll(bal);
}

// Calls f(std::integral_constant<int, K>) for the K in [0, N) equal to the wave-uniform k:
// a register array can then be indexed by a run-time (but uniform) k without leaving registers.
template <typename F, int... Is>
__device__ __forceinline__ void with_static_index(int k, F &&f, std::integer_sequence<int, Is...>)
{
    (void)std::initializer_list<int>{((k == Is) ? (f(std::integral_constant<int, Is>{}), 0) : 0)...};
}

// Order-preserving map of a double's high/low words to unsigned integers.
__device__ __forceinline__ void sortable_words(double x, unsigned &hi, unsigned &lo)
{
    const int h = __double2hiint(x);
    const unsigned m = (unsigned)(h >> 31);          // all ones for negative values
    hi = (unsigned)h ^ (m | 0x80000000u);
    lo = (unsigned)__double2loint(x) ^ m;
}

// More than one bit set in a wave-uniform lane mask (s_bcnt1 + compare; the generic popcount idiom
// compiles to a four-instruction x & (x - 1) test).
__device__ __forceinline__ bool several_lanes(unsigned long long mask)
{
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(mask) : "scc");
    return n > 1;
}

// One wavefront per problem with T < Q <= 64*NCOL.  Lane l owns the columns (queries)
// j = l + 64*k, k < NCOL, and keeps their reduced cost, dual, assignment and predecessor in
// registers.  A column that has been scanned (or lies past Q) is parked with reduced cost +inf
// and dual -inf, which makes it inert in the scan without a mask test; its real values wait in
// LDS, at the column's place in the row's scan order, until the row's dual update (in registers they
// cost 4 VGPRs per column: 208 instead of 152 for ten columns, two wavefronts per SIMD instead of three).
// Per Dijkstra step the minimum is located
// with 32-bit DPP reductions on the order-preserving high word (low word and list-position tie
// key only when needed).  No barrier: a single wave executes its LDS operations in order.
template <int NCOL>
__global__ __launch_bounds__(64) void lap_wave_kernel(const DmSetArray sets, const int S,
                                                       const float *__restrict__ cost,
                                                       const int32_t *__restrict__ tgt_off,
                                                       const int32_t *__restrict__ order,
                                                       int64_t *__restrict__ src_out, int64_t *__restrict__ tgt_out,
                                                         int64_t *__restrict__ flat_out,
                                                       int32_t *status, const int t_cap, const int lds_cost_floats)
{
    extern __shared__ __align__(16) unsigned char smem[];
    using KSeq = std::make_integer_sequence<int, NCOL>;
    constexpr int QCAP = 64 * NCOL;
    constexpr double NEG_INF = -INF;
    const int lane = threadIdx.x;
    const int s = blockIdx.x % S;
    const int bi = blockIdx.x / S;
    const int b = order ? order[bi] : bi;
    const DmSet &set = sets.s[s];
    const int nc = set.Q;                 // columns = queries
    const int t0 = tgt_off[b];
    const int nr = tgt_off[b + 1] - t0;   // rows = targets
    if (nr == 0 || nr >= nc || nc > QCAP) return;   // not this kernel's problem
    if (nr > t_cap || nr < 0) {      // more targets than the caller's max_T said: the LDS arrays are sized by it
        if (lane == 0) dm::raise_status(status, DM_ST_BAD_SIZES);
        return;
    }
#ifdef DM_LAP_STAMPS
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_c1 = 0;
    int st_iters = 0;
    unsigned long long st_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
#endif

    double *u = reinterpret_cast<double *>(smem);           // [t_cap] row duals
    double *shadow = u + t_cap;                             // [t_cap][2] reduced cost and dual of the columns parked in the
                                                            // current row, in scan order (registers hold +inf / -inf meanwhile)
    int *col4row = reinterpret_cast<int *>(shadow + 2 * t_cap);      // [t_cap]
    int *sclist = col4row + t_cap;                          // [t_cap] columns scanned in the current row, in order
    int *row4col = sclist + t_cap;                          // [QCAP]
    int *pathL = row4col + QCAP;                            // [QCAP]
    int *remaining = pathL + QCAP;                          // [QCAP] list position -> column
    int *pos = remaining + QCAP;                            // [QCAP] column -> list position
    float *crow = reinterpret_cast<float *>(pos + QCAP);    // [lds_cost_floats] first rows of the block

    const int qs = set.q_stride;
    const float *cg = cost + set.cost_off + (size_t)t0 * qs;
    // rows resident in LDS keep the global row stride when they can be copied linearly
    const bool linear = (qs & 3) == 0 && (reinterpret_cast<uintptr_t>(cg) & 15) == 0;
    const int ls = linear ? qs : nc;                        // LDS row stride
    const int R = min(nr, lds_cost_floats / ls);            // rows resident in LDS

    double v[NCOL], spc[NCOL];
    int r4c[NCOL], pth[NCOL], jc[NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
        const int j = lane + 64 * k;
        const bool ok = j < nc;
        v[k] = ok ? 0.0 : NEG_INF;
        spc[k] = INF; r4c[k] = -1; pth[k] = -1;
        jc[k] = min(j, nc - 1);      // columns past the end alias the last one and stay parked
        if (ok) {
            row4col[j] = -1;
            remaining[nc - 1 - j] = j;
            pos[j] = nc - 1 - j;
        }
    }
    for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }

    // Stage the first R rows in LDS.  A dense, 16-byte aligned block goes through LDS-DMA
    // (no registers, every piece in flight at once); anything else through registers.
    const int nfl = R * ls;
    bool bad = false;
    if (linear) {
        for (int base = 0; base < nfl; base += 256) {
            const int e = base + lane * 4;
            if (e < nfl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(cg + e),
                                                 (__attribute__((address_space(3))) void *)(crow + base), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = 0; i < R; ++i)      // (the padding behind column nc is never written: skip it)
            for (int e = lane * 4; e < nc; e += 256) {
                const float4 x = *reinterpret_cast<const float4 *>(crow + i * ls + e);
                bad |= cost_is_invalid(x.x) | (e + 1 < nc && cost_is_invalid(x.y)) |
                       (e + 2 < nc && cost_is_invalid(x.z)) | (e + 3 < nc && cost_is_invalid(x.w));
            }
    } else {
        for (int i = 0; i < R; ++i) {
            float c[NCOL];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) c[k] = cg[(size_t)i * qs + jc[k]];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                bad |= cost_is_invalid(c[k]);
                crow[i * ls + jc[k]] = c[k];
            }
        }
    }
    if (__any(bad)) {
        if (lane == 0) dm::raise_status(status, DM_ST_INVALID_COST);
        return;
    }
    __syncthreads();
#ifdef DM_LAP_STAMPS
    st_c1 = __builtin_amdgcn_s_memtime();
#endif

    // This lane's entries of row i.  `fresh`: the first read of a row that is not resident in LDS (every row is
    // read in full exactly once that way, as the row a search starts from): validated here.
    auto read_row = [&](int i, float (&c)[NCOL], bool fresh) __attribute__((always_inline)) {
        if (i < R) {
            const float *row = crow + i * ls;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) c[k] = row[jc[k]];
        } else {
            const float *row = cg + (size_t)i * qs;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) c[k] = row[jc[k]];
            if (fresh) {
                bool badrow = false;
#pragma unroll
                for (int k = 0; k < NCOL; ++k) badrow |= cost_is_invalid(c[k]);
                if (badrow) dm::raise_status(status, DM_ST_INVALID_COST);
            }
        }
    };
    // The first step of a row's search waits for no read: its row was requested while the row before was being
    // searched, its dual is still 0 and the last list position holds column 0 at every row start.
    float cnext[NCOL];
    read_row(0, cnext, true);
    for (int cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int i = cur, n = nc, sink = -1;
        double ui = 0.0;
        int jl = 0;
        float c[NCOL];
#pragma unroll
        for (int k = 0; k < NCOL; ++k) c[k] = cnext[k];
        if (cur + 1 < nr) read_row(cur + 1, cnext, true);
        bool first_step = true;

        // every step parks one column or ends the row; with no column left every candidate is +inf,
        // which ends the loop as infeasible (sink stays -1): two exits only
        for (;;) {
#ifdef DM_LAP_STAMPS
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t)::"memory");
#endif
            if (!first_step) {      // a row reached through an assigned column
                ui = u[i];
                jl = __builtin_amdgcn_readfirstlane(remaining[n - 1]);   // column at the last list position
                read_row(i, c, false);
            }
            first_step = false;
            DM_STAMP(0)
            double lmin = INF;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const double r = ((min_val + (double)c[k]) - ui) - v[k];   // +inf for parked columns
                const bool better = r < spc[k];
                spc[k] = better ? r : spc[k];
                pth[k] = better ? i : pth[k];
                lmin = (NCOL == 1) ? spc[k] : fmin(lmin, spc[k]);
            }
            DM_STAMP(1)
            // locate the minimum: high word, then (if several lanes share it) low word
            unsigned hi, lo;
            sortable_words(lmin, hi, lo);
            const unsigned mhi = dm::wave_min_u32(hi);
            unsigned long long at = __builtin_amdgcn_ballot_w64(hi == mhi);
            if (several_lanes(at)) {
                const unsigned mlo = dm::wave_min_u32(__builtin_amdgcn_inverse_ballot_w64(at) ? lo : 0xFFFFFFFFu);
                at &= __builtin_amdgcn_ballot_w64(lo == mlo);
            }
            const int wl0 = __ffsll((long long)at) - 1;
            const double m = __hiloint2double(dm::readlane_i32(__double2hiint(lmin), wl0),
                                              dm::readlane_i32(__double2loint(lmin), wl0));
            if (!(m < INF)) break;
            DM_STAMP(2)
            // columns attaining it
            int hits = 0, cj = -1, cr = -1;
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                const bool hit = spc[k] == m;
                hits += __popcll(__ballot(hit));
                cj = hit ? lane + 64 * k : cj;
                cr = hit ? r4c[k] : cr;
            }
            int jw, rw, pw = -1;
            if (hits == 1) {
                jw = dm::readlane_i32(cj, wl0);
                rw = dm::readlane_i32(cr, wl0);
            } else {
                // several columns tie: the list-position rule decides
                unsigned key = 0;
                cj = -1; cr = -1;
#pragma unroll
                for (int k = 0; k < NCOL; ++k) {
                    if (spc[k] == m) {
                        const int j = lane + 64 * k;
                        const int pp = pos[j];
                        const unsigned kk = (r4c[k] == -1) ? (unsigned)(nc + 1 + pp) : (unsigned)(nc - pp);
                        if (kk > key) { key = kk; cj = j; cr = r4c[k]; }
                    }
                }
                const unsigned K = dm::wave_max_u32(key);
                const int wl = __ffsll((long long)__ballot(key == K)) - 1;
                jw = dm::readlane_i32(cj, wl);
                rw = dm::readlane_i32(cr, wl);
                pw = (K > (unsigned)nc) ? (int)(K - (unsigned)nc - 1u) : (int)((unsigned)nc - K);
            }
            DM_STAMP(3)
            min_val = m;
#ifdef DM_LAP_STAMPS
            ++st_iters;
#endif
            // an unassigned column ends the row: it is neither removed from the list nor parked
            if (rw == -1) { sink = jw; DM_STAMP(4) break; }
            if (pw < 0) pw = __builtin_amdgcn_readfirstlane(pos[jw]);
            // swap-with-last removal of column jw from the list
            const int scan_at = nc - n;      // how many columns this row has scanned before this one
            if (lane == 0) {
                remaining[pw] = jl;
                pos[jl] = pw;
                sclist[scan_at] = jw;
            }
            --n;
            // park the column: its real values wait in LDS, at its place in the scan order
            with_static_index(jw >> 6, [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (lane == (jw & 63)) {
                    shadow[2 * scan_at] = spc[K]; shadow[2 * scan_at + 1] = v[K];
                    spc[K] = INF; v[K] = NEG_INF;
                }
            }, KSeq{});
            DM_STAMP(4)
            i = rw;
        }

        if (sink < 0) {      // ran out of columns without reaching an unassigned one
            if (lane == 0) dm::raise_status(status, DM_ST_INFEASIBLE);
            return;
        }
        // Row end: dual update of the scanned columns, augmentation along the predecessor
        // chain, un-parking, and the list back in its initial (reversed) order.
        const int nsc = nc - n;      // scanned (parked) columns: all of them assigned
        if (nsc == 0) {
            // the first step already found an unassigned column: nothing changes but u[cur] and that
            // column's assignment; the list is untouched
            with_static_index(sink >> 6, [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (lane == (sink & 63)) r4c[K] = cur;
            }, KSeq{});
            if (lane == 0) {
                u[cur] += min_val;
                col4row[cur] = sink;
                row4col[sink] = cur;
            }
        } else {
            for (int e = 0; e < nsc; ++e) {
                const int j = __builtin_amdgcn_readfirstlane(sclist[e]);
                with_static_index(j >> 6, [&](auto kc) {
                    constexpr int K = decltype(kc)::value;
                    if (lane == (j & 63)) {
                        const double d = min_val - shadow[2 * e];
                        u[r4c[K]] += d;
                        v[K] = shadow[2 * e + 1] - d;
                        pathL[j] = pth[K];
                    }
                }, KSeq{});
            }
            with_static_index(sink >> 6, [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (lane == (sink & 63)) pathL[sink] = pth[K];
            }, KSeq{});
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
            for (int e = 0; e < nsc; ++e) {
                const int j = __builtin_amdgcn_readfirstlane(sclist[e]);
                with_static_index(j >> 6, [&](auto kc) {
                    constexpr int K = decltype(kc)::value;
                    if (lane == (j & 63)) r4c[K] = row4col[j];
                }, KSeq{});
            }
            with_static_index(sink >> 6, [&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (lane == (sink & 63)) r4c[K] = row4col[sink];
            }, KSeq{});
            // Every list slot that changed is either the initial slot of a scanned column or one
            // of the last nsc slots; both loops write initial values, overlaps agree.
            for (int e = lane; e < nsc; e += 64) {
                const int j = sclist[e];
                remaining[nc - 1 - j] = j;
                pos[j] = nc - 1 - j;
            }
            for (int e = n + lane; e < nc; e += 64) {
                remaining[e] = nc - 1 - e;
                pos[nc - 1 - e] = e;
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < NCOL; ++k) spc[k] = INF;
    }

#ifdef DM_LAP_STAMPS
    if (lane == 0) {
        const unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *rec = reinterpret_cast<unsigned long long *>(status + 16) + (size_t)blockIdx.x * 12;
        rec[0] = st_c1 - st_c0; rec[1] = c2 - st_c1; rec[2] = r2 - st_r0; rec[3] = st_iters; rec[4] = nr; rec[5] = c2 - st_c0;
        for (int q = 0; q < 5; ++q) rec[6 + q] = st_seg[q];
    }
#endif
    int64_t *so = src_out + set.out_off + t0;
    int64_t *to = tgt_out + set.out_off + t0;
    int64_t *fo = flat_out ? flat_out + set.out_off + t0 : nullptr;
    int base = 0;
#pragma unroll
    for (int k = 0; k < NCOL; ++k) emit_pair(r4c[k] >= 0, lane + 64 * k, r4c[k], base, lane, so, to, fo, t0);
}

// Inverse of sortable_words.
__device__ __forceinline__ double from_sortable_words(unsigned hi, unsigned lo)
{
    const unsigned m = (hi & 0x80000000u) ? 0u : 0xFFFFFFFFu;    // top bit set <=> value was non-negative
    return __hiloint2double((int)(hi ^ (m | 0x80000000u)), (int)(lo ^ m));
}

// Lanes of the wave-uniform mask `cand` whose (hi, lo[, key]) is the lexicographic minimum (key:
// maximum); returns the lane mask after as few 32-bit reductions as the data needs.  All masks stay
// in scalar registers.  ROW0: the candidates sit in lanes 0..15 (one DPP row: four reduction steps).
template <bool ROW0>
__device__ __forceinline__ unsigned long long narrow_to_minimum(unsigned long long cand, unsigned hi, unsigned lo,
                                                                unsigned key)
{
    auto reduce_min = [](unsigned x) { return ROW0 ? dm::row0_min_u32(x) : dm::wave_min_u32(x); };
    // ROW0 == false: every lane is a candidate
    const unsigned mhi = reduce_min(ROW0 ? (__builtin_amdgcn_inverse_ballot_w64(cand) ? hi : 0xFFFFFFFFu) : hi);
    unsigned long long at = ROW0 ? __builtin_amdgcn_ballot_w64(hi == mhi) & cand : __builtin_amdgcn_ballot_w64(hi == mhi);
    if (several_lanes(at)) {
        const unsigned mlo = reduce_min(__builtin_amdgcn_inverse_ballot_w64(at) ? lo : 0xFFFFFFFFu);
        at &= __builtin_amdgcn_ballot_w64(lo == mlo);
        if (several_lanes(at)) {
            const unsigned K = dm::wave_max_u32(__builtin_amdgcn_inverse_ballot_w64(at) ? key : 0u);
            at &= __builtin_amdgcn_ballot_w64(key == K);
        }
    }
    return at;
}

// One workgroup of W wavefronts (W = blockDim.x / 64, 1..16) per problem with T < Q <= 64*W*NCOL
// (up to 4096 columns with 16 wavefronts and 4 columns per thread).
// Thread `tid` owns the columns (queries) j = tid + 64*W*k, k < NCOL, and keeps their reduced
// cost, dual, assignment, predecessor and list position in registers; scanned columns are
// parked (reduced cost +inf, dual -inf) as in lap_wave_kernel.  Per Dijkstra step every wave
// narrows its columns to one candidate with 32-bit DPP reductions on order-preserving words,
// the candidate's lane publishes one record in LDS, ONE barrier, and every wave narrows the W
// records the same way -- all waves agree on the step's column without a second barrier (the
// records are double-buffered by step parity).  State that every wave needs (list, row
// assignment of a one-step row) is written redundantly by each wave, so a wave only ever
// depends on its own LDS writes between barriers.  The step loop has two exits only (row end,
// nothing finite left) and its lane masks live in scalar registers: half of a step's instructions
// used to be scalar control flow.  flat_out (optional): target index + tgt_off[image].
// WIDE: compiled for up to 1024 threads whatever NCOL (the register budget of 16 wavefronts).
template <int NCOL, bool WIDE, bool FUSED>
__global__ __launch_bounds__((WIDE || NCOL == 1) ? 1024 : 64 * ((FAST_MAX_NCOL + NCOL - 1) / NCOL)) void lap_block_kernel(const DmSetArray sets, const int S,
                                                         const dm::FusedIn fin, float *cost_w,
                                                         const float *__restrict__ cost,
                                                         const int32_t *__restrict__ tgt_off,
                                                         const int32_t *__restrict__ order,
                                                         int64_t *__restrict__ src_out, int64_t *__restrict__ tgt_out,
                                                         int64_t *__restrict__ flat_out,
                                                         int32_t *status, const int t_cap, const int lds_cost_floats)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr double NEG_INF = -INF;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int BS = blockDim.x, W = BS >> 6;
    const int QCAP = BS * NCOL;
    const int s = blockIdx.x % S;
    const int bi = blockIdx.x / S;
    const DmSet &set = sets.s[s];
    const int nc = set.Q;                 // columns = queries
    int b, t0, nr;                        // image, its first packed target, rows = targets
    const int64_t *tl = nullptr;          // FUSED: the image's labels and boxes
    const float4 *tb = nullptr;
    if (FUSED && fin.direct) {
        b = fin.table.order[bi];
        t0 = fin.table.offsets[b];
        nr = fin.table.offsets[b + 1] - t0;
        tl = fin.table.labels[b];
        tb = reinterpret_cast<const float4 *>(fin.table.boxes[b]);
    } else {
        b = order ? order[bi] : bi;
        t0 = tgt_off[b];
        nr = tgt_off[b + 1] - t0;
        if (FUSED) {
            tl = fin.labels + t0;
            tb = reinterpret_cast<const float4 *>(fin.tboxes) + t0;
        }
    }
    if (FUSED && nr == 0) {
        // an image without targets: its predictions are still validated (the reference asserts on the whole batch)
        bool okb = true;
        for (int j = tid; j < nc; j += BS)
            okb = okb && dm::pred_valid(dm::make_pred(dm::image_boxes(set, b)[j]));
        if (!okb) dm::raise_status(status, DM_ST_DEGENERATE_BOX);
    }
    if (nr == 0 || nr >= nc || nc > QCAP) return;   // not this kernel's problem
    if (nr > t_cap || nr < 0) {      // more targets than the caller's max_T said: the LDS arrays are sized by it
        if (tid == 0) dm::raise_status(status, DM_ST_BAD_SIZES);
        return;
    }
#ifdef DM_LAP_STAMPS
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_c1 = 0;
    int st_iters = 0;
    __shared__ unsigned long long st_arr[16];      // arrival time of every wavefront at the step's barrier
    unsigned long long st_spread = 0, st_release = 0, st_late[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0;
#endif

    double *u = reinterpret_cast<double *>(smem);           // [t_cap] row duals
    // [2][16] published candidates: hi, lo, key, column | (row assigned to that column + 1) << 16
    uint4 *rec = reinterpret_cast<uint4 *>(u + t_cap);
    int *col4row = reinterpret_cast<int *>(rec + 32);       // [t_cap]
    int *row4col = col4row + t_cap;                         // [QCAP]
    int *pathL = row4col + QCAP;                            // [QCAP]
    int *remaining = pathL + QCAP;                          // [QCAP] list position -> column
    dm::TgtRec *tg = reinterpret_cast<dm::TgtRec *>(remaining + QCAP);     // FUSED: [t_cap] the image's targets
    float *crow = FUSED ? reinterpret_cast<float *>(tg + t_cap)
                        : reinterpret_cast<float *>(remaining + QCAP);   // [lds_cost_floats] first rows of the block

    const int qs = set.q_stride;
    // FUSED: the rows beyond LDS live in the workspace, written and read by the same thread
    const float *cg = (FUSED ? cost_w : cost) + set.cost_off + (size_t)t0 * qs;
    // rows resident in LDS keep the global row stride when they can be copied linearly
    const bool linear = FUSED || ((qs & 3) == 0 && (reinterpret_cast<uintptr_t>(cg) & 15) == 0);
    const int ls = linear ? qs : nc;                        // LDS row stride
    const int R = min(nr, lds_cost_floats / ls);            // rows resident in LDS

    double v[NCOL], spc[NCOL], v_real[NCOL], spc_real[NCOL];
    int r4c[NCOL], pth[NCOL], jc[NCOL], posr[NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
        const int j = tid + BS * k;
        const bool ok = j < nc;
        v[k] = ok ? 0.0 : NEG_INF;
        spc[k] = INF; v_real[k] = 0.0; spc_real[k] = INF; r4c[k] = -1; pth[k] = -1;
        jc[k] = min(j, nc - 1);      // columns past the end alias the last one and stay parked
        posr[k] = nc - 1 - j;
        if (ok) {
            row4col[j] = -1;
            remaining[nc - 1 - j] = j;
        }
    }
    for (int i = tid; i < nr; i += BS) { u[i] = 0.0; col4row[i] = -1; }

    // Stage the first R rows in LDS.  A 16-byte aligned block goes through LDS-DMA (no
    // registers, every piece in flight at once); anything else through registers.
    const int nfl = R * ls;
    bool bad = false;
    if (FUSED) {
        bad = dm::compute_cost_block<NCOL>(set, fin, b, nc, nr, tl, tb, tid, BS, tg, crow, ls, R,
                                           cost_w + set.cost_off + (size_t)t0 * qs, qs, status);
    } else if (linear) {
        for (int base = wave * 256; base < nfl; base += W * 256) {
            const int e = base + lane * 4;
            if (e < nfl)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(cg + e),
                                                 (__attribute__((address_space(3))) void *)(crow + base), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = wave; i < R; i += W)      // one row per wave and trip (the padding behind column nc is
            for (int e = lane * 4; e < nc; e += 256) {      // never written: skip it)
                const float4 x = *reinterpret_cast<const float4 *>(crow + i * ls + e);
                bad |= cost_is_invalid(x.x) | (e + 1 < nc && cost_is_invalid(x.y)) |
                       (e + 2 < nc && cost_is_invalid(x.z)) | (e + 3 < nc && cost_is_invalid(x.w));
            }
    } else {
        for (int i = 0; i < R; ++i) {
            float c[NCOL];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) c[k] = cg[(size_t)i * qs + jc[k]];
#pragma unroll
            for (int k = 0; k < NCOL; ++k) {
                bad |= cost_is_invalid(c[k]);
                crow[i * ls + jc[k]] = c[k];
            }
        }
    }
    if (__syncthreads_or(bad)) {
        if (tid == 0) dm::raise_status(status, DM_ST_INVALID_COST);
        return;
    }

#ifdef DM_LAP_STAMPS
    st_c1 = __builtin_amdgcn_s_memtime();
#endif
    // The solver proper, compiled twice: with every row of the block resident in LDS (the usual case)
    // the per-step choice of where a row lives disappears from the step.
    // NC: the columns per thread that this WAVEFRONT really has (the rest of its NCOL lie past the last column: with
    // a thread count that is not a multiple of what the columns need, only the first wavefronts own NCOL columns).
    auto solve = [&](auto all_resident, auto columns_here) __attribute__((always_inline)) -> bool {
        constexpr bool ALLRES = decltype(all_resident)::value;
        constexpr int NC = decltype(columns_here)::value;
        int par = 0;
        unsigned scanned = 0;      // bit k: this thread's column k has been scanned (parked) in the current row
        // This thread's entries of row i.  `fresh`: the first read of a row that is not resident in LDS (every row is
        // read in full exactly once that way, as the row a search starts from): validated here.
        auto read_row = [&](int i, float (&c)[NC], bool fresh) __attribute__((always_inline)) {
            if (ALLRES || i < R) {
                const float *row = crow + i * ls;
#pragma unroll
                for (int k = 0; k < NC; ++k) c[k] = row[jc[k]];
            } else {
                const float *row = cg + (size_t)i * qs;
#pragma unroll
                for (int k = 0; k < NC; ++k) c[k] = row[jc[k]];
                if (!FUSED && fresh) {
                    bool badrow = false;
#pragma unroll
                    for (int k = 0; k < NC; ++k) badrow |= cost_is_invalid(c[k]);
                    if (badrow) dm::raise_status(status, DM_ST_INVALID_COST);
                }
            }
        };
        // The first step of a row's search waits for no LDS read: the entries of the row it starts from were requested
        // while the row before was being searched, the row's dual is still 0 and the last list position holds column 0
        // at every row start.  (Requesting the reads of the later steps ahead as well -- right after the step before
        // has picked its column, in front of its bookkeeping -- measured slower.)
        float cnext[NC];
        read_row(0, cnext, true);
        for (int cur = 0; cur < nr; ++cur) {
            double min_val = 0.0;
            int i = cur, n = nc, sink = -1;
            double ui = 0.0;                              // u[cur]: untouched before its own row
            int jl_read = 0;                              // column at the last list position: 0 at a row start
            float c[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) c[k] = cnext[k];
            if (cur + 1 < nr) read_row(cur + 1, cnext, true);

            // Every step parks one column or ends the row; when no column is left every candidate is +inf,
            // which ends the loop as infeasible: two exits only.
            bool first_step = true;
            for (;;) {
#ifdef DM_LAP_STAMPS
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t)::"memory");
#endif
                if (!first_step) {      // a row reached through an assigned column
                    ui = u[i];
                    jl_read = remaining[n - 1];
                    read_row(i, c, false);
                }
                first_step = false;
                DM_STAMP(0)
                double lmin = INF;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const double r = ((min_val + (double)c[k]) - ui) - v[k];   // +inf for parked columns
                    const bool better = r < spc[k];
                    spc[k] = better ? r : spc[k];
                    pth[k] = better ? i : pth[k];
                    lmin = (NC == 1) ? spc[k] : fmin(lmin, spc[k]);
                }
                // the lane's candidate among its own columns (list-position rule on ties)
                unsigned key = 0;
                int cj = -1, cr = -1;
                if (NC == 1) {
                    key = (r4c[0] == -1) ? (unsigned)(nc + 1 + posr[0]) : (unsigned)(nc - posr[0]);
                    cj = tid;
                    cr = r4c[0];
                } else {
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const bool hit = spc[k] == lmin;
                        const unsigned kk = (r4c[k] == -1) ? (unsigned)(nc + 1 + posr[k]) : (unsigned)(nc - posr[k]);
                        if (hit && kk > key) { key = kk; cj = tid + BS * k; cr = r4c[k]; }
                    }
                }
                unsigned hi, lo;
                sortable_words(lmin, hi, lo);
                DM_STAMP(1)
                // the wave's candidate, published by its lane
                const unsigned long long at = narrow_to_minimum<false>(~0ull, hi, lo, key);
                if (__builtin_amdgcn_inverse_ballot_w64(at & (0ull - at)))      // its lowest lane
                    rec[par * 16 + wave] = make_uint4(hi, lo, key, (unsigned)cj | ((unsigned)(cr + 1) << 16));
                DM_STAMP(2)
#ifdef DM_LAP_STAMPS
                if (lane == 0) st_arr[wave] = st_t;
#endif
                __syncthreads();
                DM_STAMP(3)
#ifdef DM_LAP_STAMPS
                if (tid == 0) {
                    unsigned long long lo_ = ~0ull, hi_ = 0;
                    for (int w_ = 0; w_ < W; ++w_) { lo_ = st_arr[w_] < lo_ ? st_arr[w_] : lo_; hi_ = st_arr[w_] > hi_ ? st_arr[w_] : hi_; }
                    for (int w_ = 0; w_ < W; ++w_) st_late[w_] += st_arr[w_] - lo_;
                    st_spread += hi_ - lo_;
                    st_release += st_t - hi_;
                }
                __syncthreads();
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t)::"memory");
#endif
                // every wave picks the workgroup's candidate from the W records
                const uint4 x = rec[par * 16 + (lane & 15)];
                const unsigned long long win = narrow_to_minimum<true>((1ull << W) - 1ull, x.x, x.y, x.z);
                const int wl = __builtin_ctzll(win);      // never empty: some lane holds the minimum
                const unsigned whi = (unsigned)dm::readlane_i32((int)x.x, wl);
                const unsigned wlo = (unsigned)dm::readlane_i32((int)x.y, wl);
                const unsigned K = (unsigned)dm::readlane_i32((int)x.z, wl);
                const unsigned jr = (unsigned)dm::readlane_i32((int)x.w, wl);
                const int jw = (int)(jr & 0xFFFFu);
                const int rw = (int)(jr >> 16) - 1;
                // +inf maps to (0xFFF00000, 0), NaN above it: no finite candidate left
                if (whi >= 0xFFF00000u) break;      // sink stays -1
                const double m = from_sortable_words(whi, wlo);
                min_val = m;
                par ^= 1;
#ifdef DM_LAP_STAMPS
                ++st_iters;
#endif
                // an unassigned column ends the row: it is neither removed from the list nor parked
                if (rw == -1) { sink = jw; DM_STAMP(4) break; }
                const int pw = (K > (unsigned)nc) ? (int)(K - (unsigned)nc - 1u) : (int)((unsigned)nc - K);
                // swap-with-last removal of column jw (every wave writes the same values)
                const int jl = __builtin_amdgcn_readfirstlane(jl_read);
                if (lane == 0) remaining[pw] = jl;
                --n;
                i = rw;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int j = tid + BS * k;
                    posr[k] = (j == jl) ? pw : posr[k];
                    if (j == jw) {       // park the column in its owner's registers
                        spc_real[k] = spc[k]; v_real[k] = v[k];
                        spc[k] = INF; v[k] = NEG_INF;
                        scanned |= 1u << k;
                    }
                }
                DM_STAMP(4)
            }
            if (sink < 0) {      // ran out of columns without reaching an unassigned one
                if (tid == 0) dm::raise_status(status, DM_ST_INFEASIBLE);
                return false;
            }

            // Row end: dual update of the scanned columns, augmentation along the predecessor
            // chain, un-parking, and the list back in its initial (reversed) order.
            if (n == nc) {
                // the first step already found an unassigned column: nothing changes but u[cur] and the
                // assignment of that column; list and positions are untouched.
                // No barrier: every wave writes the shared words itself.
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    if (tid + BS * k == sink) r4c[k] = cur;
                    spc[k] = INF;
                }
                if (lane == 0) {
                    u[cur] = min_val;                     // it was 0
                    col4row[cur] = sink;
                    row4col[sink] = cur;
                }
            } else {
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int j = tid + BS * k;
                    if ((scanned >> k) & 1u) {      // a column scanned in this row (all of them are assigned)
                        const double d = min_val - spc_real[k];
                        u[r4c[k]] += d;
                        v[k] = v_real[k] - d;
                        pathL[j] = pth[k];
                    }
                    if (j == sink) pathL[j] = pth[k];
                }
                if (tid == 0) u[cur] += min_val;
                __syncthreads();
                if (tid == 0) {
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
                // refresh the assignments of the scanned columns and of the sink (every owner for its own
                // columns, in parallel); list and positions back to initial
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int j = tid + BS * k;
                    if ((scanned >> k) & 1u) {
                        r4c[k] = row4col[j];
                        remaining[nc - 1 - j] = j;
                    }
                    if (j == sink) r4c[k] = row4col[j];
                }
                scanned = 0;
                for (int e = n + tid; e < nc; e += BS) remaining[e] = nc - 1 - e;
#pragma unroll
                for (int k = 0; k < NC; ++k) { posr[k] = nc - 1 - (tid + BS * k); spc[k] = INF; }
                // No barrier here: before the barrier of the next row's first step a wave reads only u[cur + 1]
                // (untouched so far), cost rows and the last list slot, which it restores itself.
                if (lane == 0) remaining[nc - 1] = 0;
            }
        }

        return true;
    };
    // Wavefronts whose last column slice lies entirely past the end run the solver for one column per thread less
    // (same barriers; fewer instructions and a shorter dependency chain per step: measured 71 -> 56 us for six sets of
    // 300 queries sharing a launch shaped for a seventh of 1050).
    constexpr bool CAN_TRIM = NCOL >= 2 && NCOL <= 4;
    constexpr int NTRIM = CAN_TRIM ? NCOL - 1 : NCOL;
    const bool trimmed = CAN_TRIM && __builtin_amdgcn_readfirstlane(wave) * 64 + BS * (NCOL - 1) >= nc;
    bool solved;
    if (R >= nr)
        solved = trimmed ? solve(std::true_type{}, std::integral_constant<int, NTRIM>{})
                         : solve(std::true_type{}, std::integral_constant<int, NCOL>{});
    else
        solved = trimmed ? solve(std::false_type{}, std::integral_constant<int, NTRIM>{})
                         : solve(std::false_type{}, std::integral_constant<int, NCOL>{});
    if (!solved) return;

    __syncthreads();
#ifdef DM_LAP_STAMPS
    if (tid == 0) {
        const unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *rec2 = reinterpret_cast<unsigned long long *>(status + 16) + (size_t)blockIdx.x * 12;
        rec2[0] = st_c1 - st_c0; rec2[1] = c2 - st_c1; rec2[2] = r2 - st_r0; rec2[3] = st_iters; rec2[4] = nr; rec2[5] = c2 - st_c0;
        for (int q = 0; q < 5; ++q) rec2[6 + q] = st_seg[q];
        rec2[11] = st_spread;
        unsigned long long *ext = reinterpret_cast<unsigned long long *>(status + 16) + (size_t)gridDim.x * 12 + (size_t)blockIdx.x * 18;
        ext[0] = st_release; ext[1] = W;
        for (int w_ = 0; w_ < 16; ++w_) ext[2 + w_] = st_late[w_];
    }
#endif
    if (wave == 0) {
        int64_t *so = src_out + set.out_off + t0;
        int64_t *to = tgt_out + set.out_off + t0;
        int64_t *fo = flat_out ? flat_out + set.out_off + t0 : nullptr;
        int base = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
            const int j = j0 + lane;
            const int r = j < nc ? row4col[j] : -1;
            emit_pair(r >= 0, j, r, base, lane, so, to, fo, t0);
        }
    }
}

__global__ __launch_bounds__(64) void lap_generic_kernel(const DmSetArray sets, const int S,
                                                          const float *__restrict__ cost,
                                                          const int32_t *__restrict__ tgt_off,
                                                          const int32_t *__restrict__ order,
                                                          int64_t *__restrict__ src_out, int64_t *__restrict__ tgt_out,
                                                         int64_t *__restrict__ flat_out,
                                                          int32_t *status, const int c_cap, const int r_cap,
                                                          const int all_problems, const int fast_max_q)
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
    if (!all_problems && transposed && Q <= fast_max_q) return;   // a register kernel's problem
    const int nr = transposed ? T : Q;
    const int nc = transposed ? Q : T;
    if (T < 0 || nr > r_cap || nc > c_cap) {      // more targets than the caller's max_T said: the LDS arrays are sized by it
        if (lane == 0) dm::raise_status(status, DM_ST_BAD_SIZES);
        return;
    }

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
    int64_t *fo = flat_out ? flat_out + set.out_off + t0 : nullptr;
    if (transposed) {
        int base = 0;
        for (int j0 = 0; j0 < nc; j0 += 64) {
            const int j = j0 + lane;
            const int r = j < nc ? row4col[j] : -1;
            emit_pair(r >= 0, j, r, base, lane, so, to, fo, t0);
        }
    } else {
        for (int i = lane; i < nr; i += 64) {
            so[i] = i;
            to[i] = col4row[i];
            if (fo) fo[i] = col4row[i] + t0;
        }
    }
}

constexpr int NCOL_CLASSES[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 15, FAST_MAX_NCOL};
// a kernel may take 4 KB of arguments: the output sets, the in-place target table and the 12 scalars / pointers
// of lap_block_kernel have to fit
static_assert(sizeof(DmSetArray) + sizeof(dm::FusedIn) + 12 * 8 <= 4096, "kernel-argument segment of lap_block_kernel");
constexpr int FUSED_MAX_T = 256;      // the image's target records (48 bytes each) share LDS with the solver

int ncol_class(int Q)
{
    for (int c : NCOL_CLASSES) if (Q <= 64 * c) return c;
    return 0;
}

// Everything a launch of a register kernel needs besides its shape.
struct LapArgs {
    int B;
    const float *cost;
    const int32_t *tgt_off, *order;
    int max_T;
    int64_t *src_out, *tgt_out, *flat_out;
    int32_t *status;
    hipStream_t st;
    const dm::FusedIn *fin;     // non-null: the kernels compute their own cost blocks ...
    float *cost_w;              // ... and keep the rows that do not fit in LDS here
};

template <int NCOL>
int launch_wave(const DmSetArray &arr, int n, const LapArgs &a, int max_Q, int max_stride)
{
    const int fast_T = a.max_T < max_Q ? a.max_T : max_Q - 1;     // largest row count this kernel can meet
    const int t_cap = (fast_T + 3) & ~3;
    const size_t fixed = (size_t)t_cap * (8 + 16 + 4 + 4) + (size_t)4 * 64 * NCOL * 4;
    size_t want = (size_t)fast_T * max_stride * sizeof(float);
    // few problems: two workgroups per CU can share the 160 KB of LDS and keep their cost rows in
    // it; many: resident wavefronts per SIMD matter more than resident rows (measured 1.8x at 12 288
    // problems), so only the state stays in LDS and the rows are read from L2 / Infinity Cache
    size_t budget = ((size_t)n * a.B <= 512) ? 80 * 1024 : fixed + 2048;
    if (g_lap_lds_kb) budget = std::max((size_t)g_lap_lds_kb * 1024, fixed + 1040);
    if (fixed + 1024 >= budget) return DM_E_UNSUPPORTED;
    if (fixed + want > budget) want = budget - fixed;
    want &= ~(size_t)15;
    if (fixed + want > 64 * 1024) {
        int rc = dm::allow_lds(reinterpret_cast<const void *>(&lap_wave_kernel<NCOL>), 80 * 1024, "hipFuncSetAttribute(lap_wave_kernel)");
        if (rc != DM_OK) return rc;
    }
    hipLaunchKernelGGL(lap_wave_kernel<NCOL>, dim3((unsigned)n * a.B), dim3(64), fixed + want, a.st, arr, n, a.cost,
                       a.tgt_off, a.order, a.src_out, a.tgt_out, a.flat_out, a.status, t_cap, (int)(want / sizeof(float)));
    return dm::check_hip(hipGetLastError(), "lap_wave_kernel launch");
}

int launch_wave_dyn(int ncol, const DmSetArray &arr, int n, const LapArgs &a, int max_Q, int max_stride)
{
#define DM_CASE(N) case N: return launch_wave<N>(arr, n, a, max_Q, max_stride)
    switch (ncol) {
        DM_CASE(1); DM_CASE(2); DM_CASE(3); DM_CASE(4); DM_CASE(5); DM_CASE(6); DM_CASE(8); DM_CASE(10);
        DM_CASE(12); DM_CASE(15); DM_CASE(FAST_MAX_NCOL);
    }
#undef DM_CASE
    return DM_E_INVALID_ARG;
}

// (NCOL, W) of the workgroup that solves problems with Q queries when P problems are in flight.
// Few problems: as many wavefronts per problem as there are 64-column slices (latency);
// many problems: fewer, fatter wavefronts so that the grid fits the chip in few rounds.
// heavy: problems with many rows (max T > 32), whose launch takes as long as the chain of its heaviest problem.
void pick_shape(int Q, long long P, bool heavy, int *ncol, int *waves)
{
    const int need = (Q + 63) / 64;                    // 64-column slices
    long long w = need;
    // 1.5 x the chip's wavefront slots: the heaviest problems come first and the tail back-fills (measured)
    const long long wave_budget = g_lap_wave_budget ? g_lap_wave_budget : 12288;
    if (P * need > wave_budget) w = wave_budget / P;
    if (w > g_lap_max_waves) w = g_lap_max_waves;
    if (w < 1) w = 1;
    // 13 to 16 column slices (Q = 769 .. 1024): more than twelve wavefronts per problem -- three per SIMD -- cost more
    // than they bring, because a step takes as long as the wavefront that shares its SIMD with the most others (the
    // SIMD serves the oldest first; measured: wavefronts 0-3 reach the step's barrier together, 4-7 some 220 cycles
    // later, 8-11 some 470).  The problem gets twelve wavefronts and the slices left over are second columns of the
    // first ones (Q = 900: 114 us for a problem with 100 targets against 127 us with 15 wavefronts and 123 us with
    // 8 x 2 columns; Q = 1024: 70 / 76 / 73) -- as long as the launch's wavefronts are resident together (twelve
    // per problem up to 512 problems: 384 problems of Q = 900, T ~ 16: 54 us against 57 with eight and 60 with
    // fifteen; 768 problems: 69 / 63 / 109) or the launch is as long as its heaviest problem anyway (T up to 100:
    // 960 problems 124 against 130 us with eight, 1536 problems 135 / 140); eight wavefronts x 2 columns beyond that.
    // The wavefronts without a second column run the solver compiled for one (lap_block_kernel, `trimmed`).
    if (g_lap_plain_shape != 1 && g_lap_max_waves == 16) {
        int exact = 0;
        if (g_lap_plain_shape > 1) {      // testing aid
            if (P * need <= wave_budget) exact = std::min(g_lap_plain_shape, need);
        } else if (need > 12 && need <= 16 && !g_lap_wave_budget) {
            if (P * 12 <= (heavy ? 18432 : 6144)) exact = 12;
            else if (P * 8 <= 12288) exact = 8;
        } else if (need >= 5 && need <= 12 && need != 8 && w == need && P * need > 9216 && !heavy && !g_lap_wave_budget) {
            // One wavefront per slice only while (nearly) all of them are resident (8192 with one column per thread):
            // beyond that two columns per thread in about six wavefronts (Q = 600, T ~ 16: 960 problems 60 us against
            // 78 us with ten wavefronts, 1152 problems 74 / 88; Q = 768: 960 problems 67 / 75; up to 864 problems one
            // wavefront per slice is as fast or faster: 79 / 82; Q = 300, 2240 problems: 66 against 84 us; Q = 384, 1800
            // problems: 79 / 83).  Eight slices -- two wavefronts on every SIMD -- are best left alone (Q = 512: 1152
            // problems 63 / 68, 1398 problems 68 / 72).
            exact = (int)std::max<long long>((need + 1) / 2, std::min<long long>(need, 6144 / P));
        }
        if (exact > 0 && (need + exact - 1) / exact <= BLOCK_MAX_NCOL_WIDE) {
            *ncol = (need + exact - 1) / exact;
            *waves = exact;
            return;
        }
    }
    int n = (need + (int)w - 1) / (int)w;
    for (int c : NCOL_CLASSES) if (c >= n) { n = c; break; }
    *ncol = n;
    *waves = (need + n - 1) / n;
}

template <int NCOL>
int launch_block(const DmSetArray &arr, int n, const LapArgs &a, int waves, int max_Q, int max_stride)
{
    const bool fused = a.fin != nullptr;
    const int fast_T = a.max_T < max_Q ? a.max_T : max_Q - 1;     // largest row count this kernel can meet
    const int t_cap = (fast_T + 3) & ~3;
    const int qcap = 64 * waves * NCOL;
    const size_t fixed = (size_t)t_cap * (8 + 4) + 32 * 16 + (size_t)3 * qcap * 4 + (fused ? (size_t)t_cap * sizeof(dm::TgtRec) : 0);
    size_t want = (size_t)fast_T * max_stride * sizeof(float);
    // few problems: two workgroups per CU can share the 160 KB of LDS and keep their cost rows in
    // it; many: more resident workgroups beat resident rows
    size_t budget = ((size_t)n * a.B <= 512) ? 80 * 1024 : std::max<size_t>(16 * 1024, fixed + 2048);
    if (g_lap_lds_kb) budget = std::max((size_t)g_lap_lds_kb * 1024, fixed + 1040);
    if (fixed + 1024 >= budget) return DM_E_UNSUPPORTED;
    if (fixed + want > budget) want = budget - fixed;
    want &= ~(size_t)15;
    // shapes beyond the narrow variant's thread limit use the variant compiled for 1024 threads
    constexpr int narrow_threads = NCOL == 1 ? 1024 : 64 * ((FAST_MAX_NCOL + NCOL - 1) / NCOL);
    const bool wide = 64 * waves > narrow_threads;
    if (wide && NCOL > BLOCK_MAX_NCOL_WIDE) return DM_E_UNSUPPORTED;
    constexpr bool CAN_WIDE = NCOL <= BLOCK_MAX_NCOL_WIDE;
    auto kernel = fused ? (wide ? lap_block_kernel<NCOL, CAN_WIDE, true> : lap_block_kernel<NCOL, false, true>)
                        : (wide ? lap_block_kernel<NCOL, CAN_WIDE, false> : lap_block_kernel<NCOL, false, false>);
    if (fixed + want > 64 * 1024) {
        int rc = dm::allow_lds(reinterpret_cast<const void *>(kernel), 80 * 1024, "hipFuncSetAttribute(lap_block_kernel)");
        if (rc != DM_OK) return rc;
    }
    const dm::FusedIn fin = fused ? *a.fin : dm::FusedIn{};
    hipLaunchKernelGGL(kernel, dim3((unsigned)n * a.B), dim3(64 * waves), fixed + want, a.st, arr, n, fin, a.cost_w,
                       a.cost, a.tgt_off, a.order, a.src_out, a.tgt_out, a.flat_out, a.status, t_cap,
                       (int)(want / sizeof(float)));
    return dm::check_hip(hipGetLastError(), "lap_block_kernel launch");
}

int launch_block_dyn(int ncol, const DmSetArray &arr, int n, const LapArgs &a, int waves, int max_Q, int max_stride)
{
#define DM_CASE(N) case N: return launch_block<N>(arr, n, a, waves, max_Q, max_stride)
    switch (ncol) {
        DM_CASE(1); DM_CASE(2); DM_CASE(3); DM_CASE(4); DM_CASE(5); DM_CASE(6); DM_CASE(8); DM_CASE(10);
        DM_CASE(12); DM_CASE(15); DM_CASE(FAST_MAX_NCOL);
    }
#undef DM_CASE
    return DM_E_INVALID_ARG;
}

}  // namespace

namespace dm {
// launch-shape knobs of the assignment kernels (dm_debug_set_knob); 0 = automatic
int set_lap_knob(int knob, int value)
{
    switch (knob) {
    case DM_KNOB_LAP_KERNEL:
        if (value < 0 || value > 3) return DM_E_INVALID_ARG;
        g_lap_mode = value;
        return DM_OK;
    case DM_KNOB_LAP_WAVE_BUDGET:
        if (value < 0 || value > (1 << 20)) return DM_E_INVALID_ARG;
        g_lap_wave_budget = value;
        return DM_OK;
    case DM_KNOB_LAP_LDS_KB:
        if (value < 0 || value > 80) return DM_E_INVALID_ARG;
        g_lap_lds_kb = value;
        return DM_OK;
    case DM_KNOB_LAP_PLAIN_SHAPE:
        if (value < 0 || value > 16) return DM_E_INVALID_ARG;
        g_lap_plain_shape = value;
        return DM_OK;
    case DM_KNOB_LAP_MAX_WAVES:
        if (value < 0 || value > 16) return DM_E_INVALID_ARG;
        g_lap_max_waves = value ? value : 16;
        return DM_OK;
    }
    return DM_E_INVALID_ARG;
}

// The launches of one call.  Every set belongs to at most one register-kernel launch (problems with
// T < Q) and, when one of its images can have T >= Q or it has more columns than the register kernels
// hold, to the list-order kernel as well.
struct Launch {
    bool block;              // workgroup-per-problem kernel (else wavefront-per-problem)
    int ncol, waves, max_Q, max_stride;
    std::vector<int> members;
};
struct Plan {
    int fast_max_q = FAST_MAX_Q;
    std::vector<Launch> launches;
    std::vector<unsigned char> generic;      // [S] the list-order kernel visits this set
    std::vector<unsigned char> fusable;      // [S] its register launch may compute the cost blocks itself
};

static int make_plan(const DmSet *sets, int S, int B, int max_T, Plan &plan)
{
    if (S > 4096) return DM_E_UNSUPPORTED;
    // Sets wider than the wavefront kernel's 1088 columns (up to 4096) go to the workgroup kernel when
    // it may use its widest shapes for them (16 wavefronts x at most 4 columns per thread, state
    // within the LDS budget); otherwise the list-order kernel takes them.
    int fast_max_q = FAST_MAX_Q;
    if ((g_lap_mode == 0 || g_lap_mode == 2) && max_T <= 1024) {
        int wide = 0, widest = 0;
        for (int s = 0; s < S; ++s)
            if (sets[s].Q > FAST_MAX_Q && sets[s].Q <= BLOCK_MAX_Q && max_T < sets[s].Q) {
                ++wide;
                widest = std::max(widest, sets[s].Q);
            }
        if (wide > 0) {
            int ncol = 0, waves = 0;
            pick_shape(widest, (long long)wide * B, max_T > 32, &ncol, &waves);
            if (ncol <= BLOCK_MAX_NCOL_WIDE) fast_max_q = BLOCK_MAX_Q;
        }
    }
    plan.fast_max_q = fast_max_q;
    plan.generic.assign((size_t)S, 0);
    plan.fusable.assign((size_t)S, 0);
    for (int s = 0; s < S; ++s) {
        const int Q = sets[s].Q;
        if (Q < 1 || sets[s].q_stride < Q) return DM_E_INVALID_ARG;
        const bool needs_generic = (g_lap_mode == 1) || Q > fast_max_q || max_T >= Q;
        if (needs_generic && (Q > GENERIC_MAX || max_T > GENERIC_MAX)) return DM_E_UNSUPPORTED;
        plan.generic[(size_t)s] = needs_generic ? 1 : 0;
    }
    if (g_lap_mode == 1) return DM_OK;

    std::vector<unsigned char> done((size_t)S, 0);
    if (g_lap_mode == 0 || g_lap_mode == 2) {
        // Sets of different Q (decoder layers + a first-stage set over all encoder tokens) in ONE
        // launch shaped for the largest Q, when that does not squeeze the wavefronts per problem:
        // the kernel is latency-bound, idle lanes cost nothing, and the launches no longer queue
        // behind each other.
        int max_Q = 0, count = 0, distinct = 0;
        for (int s = 0; s < S; ++s) {
            if (plan.generic[(size_t)s]) continue;
            bool seen = false;
            for (int r = 0; r < s; ++r) seen |= sets[r].Q == sets[s].Q && !plan.generic[(size_t)r];
            distinct += seen ? 0 : 1;
            max_Q = std::max(max_Q, sets[s].Q);
            ++count;
        }
        const long long P = (long long)count * B;
        int ncol = 1, waves = 1;
        if (distinct > 1) pick_shape(max_Q, P, max_T > 32, &ncol, &waves);
        // ... as long as all workgroups are resident at once (8192 wavefront slots on the chip)
        if (distinct > 1 && count <= DM_MAX_SETS_PER_LAUNCH && P * waves <= 8192 &&
            (max_Q <= FAST_MAX_Q || ncol <= BLOCK_MAX_NCOL_WIDE)) {
            Launch l{true, ncol, waves, max_Q, max_Q, {}};
            for (int s = 0; s < S; ++s) {
                if (plan.generic[(size_t)s]) continue;
                l.members.push_back(s);
                done[(size_t)s] = 1;
                plan.fusable[(size_t)s] = 1;
                l.max_stride = std::max(l.max_stride, sets[s].q_stride);
            }
            plan.launches.push_back(std::move(l));
        }
    }
    // one launch per distinct Q among the (remaining) sets the register kernels can take; a set whose
    // images may have T >= Q is visited by both kinds of kernel, each taking its own problems
    for (int s0 = 0; s0 < S; ++s0) {
        const int Q = sets[s0].Q;
        if (done[(size_t)s0] || Q > fast_max_q) continue;
        int total = 0;
        for (int s = s0; s < S; ++s) total += (!done[(size_t)s] && sets[s].Q == Q) ? 1 : 0;
        // few problems: a workgroup of several wavefronts per problem (latency); many: one
        // wavefront per problem (throughput)
        int ncol = ncol_class(Q), waves = 1;
        const long long P = (long long)total * B;
        // crossover measured at Q=600, T~16: about 2800 problems (2688: 104 vs 109 us, 3072: 116 vs 101 us), i.e.
        // 28672 column slices; problems with many rows (T up to 100 at Q=900) stay faster in the workgroup kernel
        // well beyond that
        const long long slices = P * ((Q + 63) / 64);
        const bool use_block = g_lap_mode == 2 || Q > FAST_MAX_Q ||
                               (g_lap_mode == 0 && slices <= (max_T > 32 ? 98304 : 28672));
        if (use_block) pick_shape(Q, P, max_T > 32, &ncol, &waves);
        Launch l{use_block, ncol, waves, Q, Q, {}};
        for (int s = s0; s <= S; ++s) {
            if (s < S && !done[(size_t)s] && sets[s].Q == Q) {
                l.members.push_back(s);
                done[(size_t)s] = 1;
                // the workgroup kernel can compute the set's cost blocks itself -- unless the list-order
                // kernel needs them in memory anyway
                plan.fusable[(size_t)s] = (use_block && !plan.generic[(size_t)s]) ? 1 : 0;
                l.max_stride = std::max(l.max_stride, sets[s].q_stride);
            }
            if ((int)l.members.size() == DM_MAX_SETS_PER_LAUNCH || (s == S && !l.members.empty())) {
                plan.launches.push_back(l);
                l.members.clear();
                l.max_stride = Q;
            }
        }
    }
    return DM_OK;
}

// For every set: does some kernel of the call read its cost blocks from memory (so that the cost
// kernel has to run for it first)?  Only meaningful when the register kernels are allowed to compute
// cost blocks themselves; otherwise every set needs them.
int lap_plan(const DmSet *sets, int S, int B, int max_T, unsigned char *needs_cost)
{
    Plan plan;
    int rc = make_plan(sets, S, B, max_T, plan);
    if (rc != DM_OK) return rc;
    for (int s = 0; s < S; ++s) needs_cost[s] = plan.fusable[(size_t)s] ? 0 : 1;
    return DM_OK;
}

// Whether the register kernels may compute their own cost blocks for a call of this shape.
bool lap_can_fuse(int C, int max_T) { return C >= 1 && C <= 4 && max_T <= FUSED_MAX_T; }

// All assignment launches of one call.  fin != nullptr: the workgroup kernel computes the cost blocks
// of the sets lap_plan reports as not needing them in memory (cost_w: workspace for rows beyond LDS);
// every other kernel reads `cost`.
int lap_run(const DmSet *sets, int S, int B, const float *cost, const int32_t *tgt_off, const int32_t *order,
            int max_T, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out, int32_t *status, hipStream_t st,
            const FusedIn *fin, float *cost_w)
{
    Plan plan;
    int rc = make_plan(sets, S, B, max_T, plan);
    if (rc != DM_OK) return rc;
    LapArgs a{B, cost, tgt_off, order, max_T, src_out, tgt_out, flat_out, status, st, nullptr, cost_w};
    for (const Launch &l : plan.launches) {
        DmSetArray arr;
        int n = 0;
        bool fuse = fin != nullptr && l.block;
        for (int s : l.members) {
            arr.s[n++] = sets[s];
            fuse = fuse && plan.fusable[(size_t)s];
        }
        a.fin = fuse ? fin : nullptr;
        rc = l.block ? launch_block_dyn(l.ncol, arr, n, a, l.waves, l.max_Q, l.max_stride)
                     : launch_wave_dyn(l.ncol, arr, n, a, l.max_Q, l.max_stride);
        if (rc != DM_OK) return rc;
    }

    // problems the register kernels do not take: T >= Q, or Q beyond their column capacity
    DmSetArray arr;
    int n = 0, max_Q = 0;
    for (int s = 0; s <= S; ++s) {
        if (s < S && plan.generic[(size_t)s]) {
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
                rc = dm::allow_lds(reinterpret_cast<const void *>(&lap_generic_kernel), lds,
                                   "hipFuncSetAttribute(lap_generic_kernel)");
                if (rc != DM_OK) return rc;
            }
            hipLaunchKernelGGL(lap_generic_kernel, dim3((unsigned)n * B), dim3(64), lds, st, arr, n, cost, tgt_off,
                               order, src_out, tgt_out, flat_out, status, c_cap, r_cap, (g_lap_mode == 1) ? 1 : 0,
                               plan.fast_max_q);
            rc = dm::check_hip(hipGetLastError(), "lap_generic_kernel launch");
            if (rc != DM_OK) return rc;
            n = 0; max_Q = 0;
        }
    }
    return DM_OK;
}
}  // namespace dm

extern "C" int dm_lap_flat_f32(const DmSet *sets, int S, int B, const float *cost, const int32_t *tgt_off,
                               const int32_t *order, int max_T, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out,
                               int32_t *status, void *stream)
{
    if (S < 0 || B < 0 || max_T < 0 || !status) return DM_E_INVALID_ARG;
    if (S == 0 || B == 0 || max_T == 0) return DM_OK;
    if (!sets || !cost || !tgt_off || !src_out || !tgt_out) return DM_E_INVALID_ARG;
    return dm::lap_run(sets, S, B, cost, tgt_off, order, max_T, src_out, tgt_out, flat_out, status,
                       static_cast<hipStream_t>(stream), nullptr, nullptr);
}

extern "C" int dm_lap_f32(const DmSet *sets, int S, int B, const float *cost, const int32_t *tgt_off,
                          const int32_t *order, int max_T, int64_t *src_out, int64_t *tgt_out,
                          int32_t *status, void *stream)
{
    return dm_lap_flat_f32(sets, S, B, cost, tgt_off, order, max_T, src_out, tgt_out, nullptr, status, stream);
}
