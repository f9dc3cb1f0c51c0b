// Packs the targets of a batch: the per-image label and box tensors of the reference's `targets` list
// (/root/reference/models/matcher.py:47-50) into the batch-wide arrays the kernels read -- what the
// reference does with two torch.cat calls per matcher call (matcher.py:65-66) -- plus the per-image
// offsets and the processing order, in ONE launch: the per-image pointers, offsets and order travel
// in the kernel-argument segment, so nothing is uploaded.
#include "dm_common.h"

namespace {

constexpr int PACK_CHUNK = 96;      // images per launch: 96 * (8 + 8 + 4 + 4) bytes of kernel arguments

struct PackArgs {
    const int64_t *labels[PACK_CHUNK];
    const float *boxes[PACK_CHUNK];
    int32_t offsets[PACK_CHUNK + 1];
    int32_t order[PACK_CHUNK];
    int32_t first, count, batch;
};

__global__ __launch_bounds__(64) void pack_targets_kernel(const PackArgs a, int64_t *__restrict__ labels_out,
                                                           float *__restrict__ boxes_out, int32_t *__restrict__ meta_out)
{
    const int b = blockIdx.x;
    const int t0 = a.offsets[b], n = a.offsets[b + 1] - t0;
    const int64_t *lab = a.labels[b];
    const float4 *box = reinterpret_cast<const float4 *>(a.boxes[b]);
    for (int i = threadIdx.x; i < n; i += 64) {
        labels_out[t0 + i] = lab[i];
        reinterpret_cast<float4 *>(boxes_out)[t0 + i] = box[i];
    }
    if (threadIdx.x == 0) {
        meta_out[a.first + b] = t0;
        meta_out[a.batch + 1 + a.first + b] = a.order[b];
        if (a.first + b == a.batch - 1) meta_out[a.batch] = t0 + n;
    }
}

}  // namespace

extern "C" int dm_pack_targets(const uint64_t *pointers, const int32_t *meta, int B, int64_t *labels_out,
                               float *boxes_out, int32_t *meta_out, void *stream)
{
    if (B < 0 || (B > 0 && (!pointers || !meta || !labels_out || !boxes_out || !meta_out))) return DM_E_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(boxes_out) & 15) != 0) return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int first = 0; first < B; first += PACK_CHUNK) {
        PackArgs a;
        a.first = first;
        a.count = B - first < PACK_CHUNK ? B - first : PACK_CHUNK;
        a.batch = B;
        for (int i = 0; i < a.count; ++i) {
            a.labels[i] = reinterpret_cast<const int64_t *>(pointers[2 * (size_t)(first + i)]);
            a.boxes[i] = reinterpret_cast<const float *>(pointers[2 * (size_t)(first + i) + 1]);
            a.offsets[i] = meta[first + i];
            a.order[i] = meta[B + 1 + first + i];
        }
        a.offsets[a.count] = meta[first + a.count];
        hipLaunchKernelGGL(pack_targets_kernel, dim3((unsigned)a.count), dim3(64), 0, st, a, labels_out, boxes_out, meta_out);
        int rc = dm::check_hip(hipGetLastError(), "pack_targets_kernel launch");
        if (rc != DM_OK) return rc;
    }
    return DM_OK;
}
