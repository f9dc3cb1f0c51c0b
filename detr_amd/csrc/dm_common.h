// Shared pieces of the gfx950 matching kernels: kernel-argument structs and wave64
// cross-lane reductions built on DPP.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detr_match.h"

// The sets of one launch travel by value in the kernel-argument segment: no descriptor
// table in device memory, nothing to upload per call.
struct DmSetArray {
    DmSet s[DM_MAX_SETS_PER_LAUNCH];
};

namespace dm {

constexpr int WAVE = 64;

// First logit / first box of image b of a set (an image's rows are dense; images may lie further apart than
// Q rows: stride 0 means they do not).
__device__ __forceinline__ const float *image_logits(const DmSet &set, int b, int C)
{
    return set.logits + (size_t)b * (set.logits_stride ? (size_t)set.logits_stride : (size_t)set.Q * (size_t)C);
}
__device__ __forceinline__ const float4 *image_boxes(const DmSet &set, int b)
{
    return reinterpret_cast<const float4 *>(set.boxes + (size_t)b * (set.boxes_stride ? (size_t)set.boxes_stride : (size_t)set.Q * 4));
}

// DPP controls (GCN3/GFX9 encoding, all valid on gfx950)
constexpr int DPP_QUAD_SWAP1 = 0xB1;   // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;   // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_SHR4 = 0x114;
constexpr int DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142;
constexpr int DPP_ROW_BCAST31 = 0x143;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move(int v)
{
    // lanes without a source (or masked rows) keep their own value: harmless for min/max
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v)
{
    int lo = dpp_move<CTRL, ROW_MASK>(__double2loint(v));
    int hi = dpp_move<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Minimum of a double over the 64 lanes of the wave, returned in every lane (wave-uniform).
// Values must not be NaN.
__device__ __forceinline__ double wave_min_f64(double x)
{
    x = fmin(x, dpp_move<DPP_QUAD_SWAP1, 0xF>(x));
    x = fmin(x, dpp_move<DPP_QUAD_SWAP2, 0xF>(x));
    x = fmin(x, dpp_move<DPP_ROW_SHR4, 0xF>(x));
    x = fmin(x, dpp_move<DPP_ROW_SHR8, 0xF>(x));      // lanes 12..15 of each row: row minimum
    x = fmin(x, dpp_move<DPP_ROW_BCAST15, 0xA>(x));   // lane 31: rows 0-1, lane 63: rows 2-3
    x = fmin(x, dpp_move<DPP_ROW_BCAST31, 0xC>(x));   // lane 63: all four rows
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

// Maximum of an unsigned over the wave, wave-uniform result.
__device__ __forceinline__ unsigned wave_max_u32(unsigned x)
{
    int v = (int)x;
    auto mx = [](int a, int b) { return (int)max((unsigned)a, (unsigned)b); };
    v = mx(v, dpp_move<DPP_QUAD_SWAP1, 0xF>(v));
    v = mx(v, dpp_move<DPP_QUAD_SWAP2, 0xF>(v));
    v = mx(v, dpp_move<DPP_ROW_SHR4, 0xF>(v));
    v = mx(v, dpp_move<DPP_ROW_SHR8, 0xF>(v));
    v = mx(v, dpp_move<DPP_ROW_BCAST15, 0xA>(v));
    v = mx(v, dpp_move<DPP_ROW_BCAST31, 0xC>(v));
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}

// Minimum of an unsigned over the wave, wave-uniform result.  Lanes without a DPP source take
// the identity, which lets the compiler fold each step into one v_min_u32 with a DPP operand.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_step(unsigned x)
{
    const unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, CTRL, ROW_MASK, 0xF, false);
    return min(x, y);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned x)
{
    x = dpp_min_step<DPP_QUAD_SWAP1, 0xF>(x);
    x = dpp_min_step<DPP_QUAD_SWAP2, 0xF>(x);
    x = dpp_min_step<DPP_ROW_SHR4, 0xF>(x);
    x = dpp_min_step<DPP_ROW_SHR8, 0xF>(x);
    x = dpp_min_step<DPP_ROW_BCAST15, 0xA>(x);
    x = dpp_min_step<DPP_ROW_BCAST31, 0xC>(x);
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// Minimum over lanes 0..15 (the first DPP row), wave-uniform result.
__device__ __forceinline__ unsigned row0_min_u32(unsigned x)
{
    x = dpp_min_step<DPP_QUAD_SWAP1, 0xF>(x);
    x = dpp_min_step<DPP_QUAD_SWAP2, 0xF>(x);
    x = dpp_min_step<DPP_ROW_SHR4, 0xF>(x);
    x = dpp_min_step<DPP_ROW_SHR8, 0xF>(x);      // lanes 12..15 of each row: row minimum
    return (unsigned)__builtin_amdgcn_readlane((int)x, 15);
}

__device__ __forceinline__ int readlane_i32(int v, int lane)
{
    return __builtin_amdgcn_readlane(v, lane);
}

__device__ __forceinline__ void raise_status(int32_t *status, int bits)
{
    atomicOr(status, bits);
}

// Prediction boxes of image b for the consumers of the indices and the auxiliary assigners (DmIndexSet).
__device__ __forceinline__ const float4 *image_pred_boxes(const DmIndexSet &set, int b)
{
    return reinterpret_cast<const float4 *>(set.pred_boxes) +
           (size_t)b * (set.boxes_stride ? (size_t)(set.boxes_stride >> 2) : (size_t)set.Q);
}
inline bool index_set_boxes_ok(const DmIndexSet &set)
{
    if (set.boxes_stride != 0 && (set.boxes_stride < set.Q * 4 || (set.boxes_stride & 3) != 0)) return false;
    return (reinterpret_cast<uintptr_t>(set.pred_boxes) & 15) == 0;
}

// Image strides of a set: 0 (contiguous) or at least one image's rows; boxes stay 16-byte aligned.
inline bool strides_ok(const DmSet &set, int C)
{
    if (set.logits_stride != 0 && (long long)set.logits_stride < (long long)set.Q * C) return false;
    if (set.boxes_stride != 0 && (set.boxes_stride < set.Q * 4 || (set.boxes_stride & 3) != 0)) return false;
    return (reinterpret_cast<uintptr_t>(set.boxes) & 15) == 0;
}

struct FusedIn;
int check_hip(hipError_t e, const char *what);
int allow_lds(const void *fn, size_t bytes, const char *what);
int lap_plan(const DmSet *sets, int S, int B, int max_T, unsigned char *needs_cost);
bool lap_can_fuse(int C, int max_T);
int lap_run(const DmSet *sets, int S, int B, const float *cost, const int32_t *tgt_off, const int32_t *order,
            int max_T, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out, int32_t *status, hipStream_t st,
            const FusedIn *fin, float *cost_w);

}  // namespace dm
