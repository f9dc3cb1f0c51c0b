// Entry points of libdetr_match.so that are not a kernel launch of their own.
#include "dm_common.h"

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

extern "C" int dm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int dm_match_f32(const DmSet *sets, int S, int B, int C, const int64_t *tgt_labels,
                            const float *tgt_boxes, const int32_t *tgt_off, const int32_t *order, int max_T,
                            const DmParams *params, float *cost, int64_t *src_out, int64_t *tgt_out,
                            int32_t *status, void *stream)
{
    int rc = dm_cost_f32(sets, S, B, C, tgt_labels, tgt_boxes, tgt_off, max_T, params, cost, status, stream);
    if (rc != DM_OK) return rc;
    return dm_lap_f32(sets, S, B, cost, tgt_off, order, max_T, src_out, tgt_out, status, stream);
}
