// Entry points of libdetr_match.so that are not a kernel launch of their own.
#include <atomic>

#include "cost_math.h"

extern "C" int dm_abi_version(void) { return DM_ABI_VERSION; }

extern "C" const char *dm_strerror(int code)
{
    switch (code) {
    case DM_OK: return "ok";
    case DM_E_INVALID_ARG: return "invalid argument";
    case DM_E_LAUNCH: return "HIP call failed (see dm_last_hip_error)";
    case DM_E_UNSUPPORTED: return "problem size exceeds a documented limit";
    default: return "unknown error code";
    }
}

namespace dm {
int set_cost_knob(int knob, int value);
int set_lap_knob(int knob, int value);
}  // namespace dm

namespace {
int g_no_direct = 0;     // DM_KNOB_DIRECT_TARGETS: 1 = dm_match_targets_f32 always packs the targets first (testing aid)
int g_no_fuse = 0;      // DM_KNOB_FUSE: 1 = dm_match_fused_f32 always takes the two-kernel path (testing aid)
}

extern "C" int dm_debug_set_knob(int knob, int value)
{
    switch (knob) {
    case DM_KNOB_LAP_KERNEL:
    case DM_KNOB_LAP_MAX_WAVES:
    case DM_KNOB_LAP_WAVE_BUDGET:
    case DM_KNOB_LAP_PLAIN_SHAPE:
    case DM_KNOB_LAP_LDS_KB: return dm::set_lap_knob(knob, value);
    case DM_KNOB_COST_QPT:
    case DM_KNOB_COST_TSPLIT:
    case DM_KNOB_TILE_WAVES: return dm::set_cost_knob(knob, value);
    case DM_KNOB_FUSE:
        if (value < 0 || value > 1) return DM_E_INVALID_ARG;
        g_no_fuse = value;
        return DM_OK;
    case DM_KNOB_DIRECT_TARGETS:
        if (value < 0 || value > 1) return DM_E_INVALID_ARG;
        g_no_direct = value;
        return DM_OK;
    }
    return DM_E_INVALID_ARG;
}

extern "C" int dm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {
// optional timing of the two kernel groups of dm_match_f32 with HIP events on the call's stream
bool g_profile = false;
hipEvent_t g_ev[3] = {nullptr, nullptr, nullptr};
bool g_ev_valid = false;
}  // namespace

extern "C" int dm_profile_enable(int on)
{
    g_profile = on != 0;
    g_ev_valid = false;
    if (g_profile && !g_ev[0]) {
        for (auto &e : g_ev)
            if (hipEventCreate(&e) != hipSuccess) { g_profile = false; return DM_E_LAUNCH; }
    }
    return DM_OK;
}

extern "C" int dm_profile_read(float *cost_ms, float *lap_ms)
{
    if (!g_profile || !g_ev_valid || !cost_ms || !lap_ms) return DM_E_INVALID_ARG;
    if (hipEventElapsedTime(cost_ms, g_ev[0], g_ev[1]) != hipSuccess) return DM_E_LAUNCH;
    if (hipEventElapsedTime(lap_ms, g_ev[1], g_ev[2]) != hipSuccess) return DM_E_LAUNCH;
    return DM_OK;
}

extern "C" int dm_match_flat_f32(const DmSet *sets, int S, int B, int C, const int64_t *tgt_labels,
                                 const float *tgt_boxes, const int32_t *tgt_off, const int32_t *order, int max_T,
                                 const DmParams *params, float *cost, int64_t *src_out, int64_t *tgt_out,
                                 int64_t *flat_out, int32_t *status, void *stream)
{
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_profile) (void)hipEventRecord(g_ev[0], st);
    int rc = dm_cost_f32(sets, S, B, C, tgt_labels, tgt_boxes, tgt_off, max_T, params, cost, status, stream);
    if (rc != DM_OK) return rc;
    if (g_profile) (void)hipEventRecord(g_ev[1], st);
    rc = dm_lap_flat_f32(sets, S, B, cost, tgt_off, order, max_T, src_out, tgt_out, flat_out, status, stream);
    if (g_profile) { (void)hipEventRecord(g_ev[2], st); g_ev_valid = rc == DM_OK; }
    return rc;
}

// One step with the cost blocks computed inside the assignment kernels wherever that is possible
// (C <= 4 and the set's problems all have T < Q); the other sets go through the cost kernel first.
extern "C" int dm_match_fused_f32(const DmSet *sets, int S, int B, int C, const int64_t *tgt_labels,
                                  const float *tgt_boxes, const int32_t *tgt_off, const int32_t *order, int max_T,
                                  const DmParams *params, float *workspace, int64_t *src_out, int64_t *tgt_out,
                                  int64_t *flat_out, int32_t *status, void *stream)
{
    if (S < 0 || B < 0 || C < 1 || max_T < 0 || !params || !status) return DM_E_INVALID_ARG;
    if (g_no_fuse || !dm::lap_can_fuse(C, max_T) || S > 4096)
        return dm_match_flat_f32(sets, S, B, C, tgt_labels, tgt_boxes, tgt_off, order, max_T, params, workspace,
                                 src_out, tgt_out, flat_out, status, stream);
    if (S == 0 || B == 0) return DM_OK;
    if (!sets || !tgt_off || !src_out || !tgt_out || (max_T > 0 && (!tgt_labels || !tgt_boxes || !workspace)))
        return DM_E_INVALID_ARG;
    for (int s = 0; s < S; ++s)
        if (!sets[s].logits || !sets[s].boxes || !dm::strides_ok(sets[s], C)) return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char generic[4096];      // sets whose cost blocks some kernel reads from memory
    int rc = dm::lap_plan(sets, S, B, max_T, generic);
    if (rc != DM_OK) return rc;
    if (g_profile) (void)hipEventRecord(g_ev[0], st);
    // the sets the list-order kernel will touch need their cost blocks in memory
    DmSet subset[DM_MAX_SETS_PER_LAUNCH];
    int n = 0;
    for (int s = 0; s <= S; ++s) {
        if (s < S && generic[s]) subset[n++] = sets[s];
        if (n == DM_MAX_SETS_PER_LAUNCH || (s == S && n > 0)) {
            rc = dm_cost_f32(subset, n, B, C, tgt_labels, tgt_boxes, tgt_off, max_T, params, workspace, status, stream);
            if (rc != DM_OK) return rc;
            n = 0;
        }
    }
    if (g_profile) (void)hipEventRecord(g_ev[1], st);
    if (max_T == 0) {
        // no targets anywhere: nothing to assign, but the reference still validates every prediction box
        rc = dm_cost_f32(sets, S, B, C, tgt_labels, tgt_boxes, tgt_off, max_T, params, workspace, status, stream);
    } else {
        const dm::FusedIn fin{tgt_labels, tgt_boxes, *params, C};
        rc = dm::lap_run(sets, S, B, workspace, tgt_off, order, max_T, src_out, tgt_out, flat_out, status, st, &fin,
                         workspace);
    }
    if (g_profile) { (void)hipEventRecord(g_ev[2], st); g_ev_valid = rc == DM_OK; }
    return rc;
}

extern "C" int dm_match_plan(const DmSet *sets, int S, int B, int C, int max_T, int *cost_sets)
{
    if (S < 0 || B < 0 || C < 1 || max_T < 0 || !cost_sets || (S > 0 && !sets)) return DM_E_INVALID_ARG;
    *cost_sets = S;
    if (S == 0 || g_no_fuse || !dm::lap_can_fuse(C, max_T) || S > 4096) return DM_OK;
    unsigned char needs_cost[4096];
    int rc = dm::lap_plan(sets, S, B, max_T, needs_cost);
    if (rc != DM_OK) return rc;
    int n = 0;
    for (int s = 0; s < S; ++s) n += needs_cost[s] ? 1 : 0;
    *cost_sets = n;
    return DM_OK;
}

// dm_pack_targets + dm_match_fused_f32 in one call -- and, when the step runs as one fused kernel and the batch
// fits the kernel-argument table, without the packing: the kernel reads every image's targets through the
// pointers it is handed.
extern "C" int dm_pack_targets(const uint64_t *pointers, const int32_t *meta, int B, int64_t *labels_out,
                               float *boxes_out, int32_t *meta_out, void *stream);

extern "C" int dm_match_targets_f32(const DmSet *sets, int S, int B, int C, const uint64_t *pointers,
                                    const int32_t *meta, int max_T, const DmParams *params, int64_t *labels_buf,
                                    float *boxes_buf, int32_t *meta_buf, float *workspace, int64_t *src_out,
                                    int64_t *tgt_out, int64_t *flat_out, int32_t *status, void *stream, int *packed)
{
    if (S < 0 || B < 0 || C < 1 || max_T < 0 || !params || !status || !packed) return DM_E_INVALID_ARG;
    *packed = 0;
    if (S == 0 || B == 0) return DM_OK;
    if (!sets || !pointers || !meta || !workspace || !src_out || !tgt_out) return DM_E_INVALID_ARG;
    const bool have_buffers = labels_buf && boxes_buf && meta_buf;
    if (!have_buffers && (labels_buf || boxes_buf || meta_buf)) return DM_E_INVALID_ARG;
    bool direct = !g_no_fuse && !g_no_direct && B <= dm::DIRECT_MAX_IMAGES && max_T > 0 && dm::lap_can_fuse(C, max_T) && S <= 4096;
    if (direct) {
        for (int s = 0; s < S; ++s)
            if (!sets[s].logits || !sets[s].boxes || !dm::strides_ok(sets[s], C)) return DM_E_INVALID_ARG;
        unsigned char needs_cost[4096];
        int rc = dm::lap_plan(sets, S, B, max_T, needs_cost);
        if (rc != DM_OK) return rc;
        for (int s = 0; s < S; ++s) direct = direct && !needs_cost[s];      // every set in a fused launch
    }
    if (!direct) {
        if (!have_buffers) return DM_NEED_BUFFERS;       // nothing launched
        int rc = dm_pack_targets(pointers, meta, B, labels_buf, boxes_buf, meta_buf, stream);
        if (rc != DM_OK) return rc;
        *packed = 1;
        return dm_match_fused_f32(sets, S, B, C, labels_buf, boxes_buf, meta_buf, meta_buf + B + 1, max_T, params,
                                  workspace, src_out, tgt_out, flat_out, status, stream);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    dm::FusedIn fin;
    fin.labels = nullptr;
    fin.tboxes = nullptr;
    fin.prm = *params;
    fin.C = C;
    fin.direct = 1;
    for (int b = 0; b < B; ++b) {
        fin.table.labels[b] = reinterpret_cast<const int64_t *>(pointers[2 * (size_t)b]);
        fin.table.boxes[b] = reinterpret_cast<const float *>(pointers[2 * (size_t)b + 1]);
        fin.table.offsets[b] = meta[b];
        fin.table.order[b] = meta[B + 1 + b];
    }
    fin.table.offsets[B] = meta[B];
    if (g_profile) { (void)hipEventRecord(g_ev[0], st); (void)hipEventRecord(g_ev[1], st); }
    // (tgt_off / order are not read by a launch that has the table: every set of this call is in one)
    int rc = dm::lap_run(sets, S, B, workspace, nullptr, nullptr, max_T, src_out, tgt_out, flat_out, status, st,
                         &fin, workspace);
    if (g_profile) { (void)hipEventRecord(g_ev[2], st); g_ev_valid = rc == DM_OK; }
    return rc;
}

extern "C" int dm_match_f32(const DmSet *sets, int S, int B, int C, const int64_t *tgt_labels,
                            const float *tgt_boxes, const int32_t *tgt_off, const int32_t *order, int max_T,
                            const DmParams *params, float *cost, int64_t *src_out, int64_t *tgt_out,
                            int32_t *status, void *stream)
{
    return dm_match_fused_f32(sets, S, B, C, tgt_labels, tgt_boxes, tgt_off, order, max_T, params, cost, src_out,
                              tgt_out, nullptr, status, stream);
}

// ---- the entry point that allocates nothing ----------------------------------------------------
// What dm_match_fused_f32 keeps in its caller's workspace -- the cost blocks that travel from the cost kernel
// to the assignment kernel (C > 4, sets with T >= Q) and the rows of a block that a fused assignment workgroup
// cannot hold in LDS -- lives in a fixed array of device memory that belongs to the library (one copy per GPU,
// held until the process ends).
namespace {
__device__ __align__(256) float scratch[DM_NOALLOC_MAX_COST_FLOATS];

// In host code `scratch` is only a handle for hipGetSymbolAddress; the array's address differs from GPU to
// GPU and is looked up once per device, with that device current.
constexpr int MAX_DEVICES = 64;
std::atomic<float *> g_scratch_ptr[MAX_DEVICES];

int scratch_address(float **out)
{
    int dev = 0;
    int rc = dm::check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc != DM_OK) return rc;
    if (dev < 0 || dev >= MAX_DEVICES) return DM_E_UNSUPPORTED;
    float *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        void *addr = nullptr;
        rc = dm::check_hip(hipGetSymbolAddress(&addr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
        if (rc != DM_OK) return rc;
        if (!addr) return DM_E_LAUNCH;
        p = static_cast<float *>(addr);
        g_scratch_ptr[dev].store(p, std::memory_order_release);     // racing threads store the same address
    }
    *out = p;
    return DM_OK;
}
}  // namespace

extern "C" int dm_noalloc_scratch(float **ptr, uint64_t *floats)
{
    if (!ptr || !floats) return DM_E_INVALID_ARG;
    *floats = DM_NOALLOC_MAX_COST_FLOATS;
    return scratch_address(ptr);
}

extern "C" int dm_match_noalloc_f32(const DmSet *sets, int S, int B, int C, const int64_t *tgt_labels,
                                    const float *tgt_boxes, const int32_t *tgt_off, const int32_t *order, int max_T,
                                    int M, const DmParams *params, int64_t *src_out, int64_t *tgt_out,
                                    int32_t *status, void *stream)
{
    if (S < 0 || B < 0 || M < 0 || max_T < 0 || max_T > M) return DM_E_INVALID_ARG;
    if (S > 0 && !sets) return DM_E_INVALID_ARG;
    // every set's cost region [cost_off, cost_off + M * q_stride) must lie inside the fixed array
    for (int s = 0; s < S; ++s) {
        if (sets[s].cost_off < 0 || sets[s].q_stride < 1) return DM_E_INVALID_ARG;
        const unsigned long long end = (unsigned long long)sets[s].cost_off +
                                       (unsigned long long)M * (unsigned long long)sets[s].q_stride;
        if (end > (unsigned long long)DM_NOALLOC_MAX_COST_FLOATS) return DM_E_UNSUPPORTED;
    }
    float *cost = nullptr;
    const int rc = scratch_address(&cost);
    if (rc != DM_OK) return rc;
    return dm_match_fused_f32(sets, S, B, C, tgt_labels, tgt_boxes, tgt_off, order, max_T, params, cost, src_out,
                              tgt_out, nullptr, status, stream);
}
