// Packs the targets of a batch: the per-image label and box tensors of the reference's `targets` list
// (/root/reference/models/matcher.py:47-50) into the batch-wide arrays the kernels read -- what the
// reference does with two torch.cat calls per matcher call (matcher.py:65-66) -- plus the per-image
// offsets and the processing order, in ONE launch.  Up to 96 images the per-image pointers, offsets and
// order travel in the kernel-argument segment, so nothing is uploaded; larger batches hand the same table
// over in pinned host memory that the kernel reads across the bus (16 + 8 bytes per image): one launch
// whatever the batch size, where a launch per 96 images left the GPU waiting for the host between launches
// (three launches ~10 us apart at 256 images, eleven at 1024).
#include <cstring>
#include <vector>

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

// The same with the table in memory: pointers[2 * b], pointers[2 * b + 1] = labels and boxes of image b,
// meta = offsets (B + 1) | order (B).
__global__ __launch_bounds__(64) void pack_targets_table_kernel(const uint64_t *__restrict__ pointers,
                                                                 const int32_t *__restrict__ meta, const int B,
                                                                 int64_t *__restrict__ labels_out,
                                                                 float *__restrict__ boxes_out, int32_t *__restrict__ meta_out)
{
    const int b = blockIdx.x;
    const int t0 = meta[b], n = meta[b + 1] - t0;
    const int64_t *lab = reinterpret_cast<const int64_t *>(pointers[2 * (size_t)b]);
    const float4 *box = reinterpret_cast<const float4 *>(pointers[2 * (size_t)b + 1]);
    for (int i = threadIdx.x; i < n; i += 64) {
        labels_out[t0 + i] = lab[i];
        reinterpret_cast<float4 *>(boxes_out)[t0 + i] = box[i];
    }
    if (threadIdx.x == 0) {
        meta_out[b] = t0;
        meta_out[B + 1 + b] = meta[B + 1 + b];
        if (b == B - 1) meta_out[B] = t0 + n;
    }
}

// Pinned host memory for the tables of the launches in flight: per host thread and device a ring of slots, each
// with the event recorded behind the launch that reads it; a slot is written again only after that event has
// completed (normally long ago: the ring has eight slots).
struct TableRing {
    static constexpr int SLOTS = 8;
    int dev = -1;
    unsigned char *base = nullptr;
    size_t slot_bytes = 0;
    hipEvent_t done[SLOTS] = {};
    bool used[SLOTS] = {};
    int next = 0;
};
thread_local std::vector<TableRing> t_rings;

int table_slot(size_t bytes, unsigned char **slot, hipEvent_t *done)
{
    int dev = 0;
    int rc = dm::check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc != DM_OK) return rc;
    TableRing *ring = nullptr;
    for (TableRing &r : t_rings)
        if (r.dev == dev) ring = &r;
    if (!ring) {
        t_rings.emplace_back();
        ring = &t_rings.back();
        ring->dev = dev;
        for (hipEvent_t &e : ring->done) {
            rc = dm::check_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
            if (rc != DM_OK) { t_rings.pop_back(); return rc; }
        }
    }
    if (bytes > ring->slot_bytes) {
        // a larger batch than any before: every launch that reads the old ring has to finish first
        for (int i = 0; i < TableRing::SLOTS; ++i)
            if (ring->used[i]) { (void)hipEventSynchronize(ring->done[i]); ring->used[i] = false; }
        if (ring->base) (void)hipHostFree(ring->base);
        ring->base = nullptr;
        ring->slot_bytes = 0;
        size_t want = 64 * 1024;
        while (want < bytes) want *= 2;
        void *mem = nullptr;
        rc = dm::check_hip(hipHostMalloc(&mem, want * TableRing::SLOTS, hipHostMallocDefault), "hipHostMalloc");
        if (rc != DM_OK) return rc;
        ring->base = static_cast<unsigned char *>(mem);
        ring->slot_bytes = want;
    }
    const int i = ring->next;
    ring->next = (i + 1) % TableRing::SLOTS;
    if (ring->used[i]) {
        rc = dm::check_hip(hipEventSynchronize(ring->done[i]), "hipEventSynchronize");
        if (rc != DM_OK) return rc;
    }
    ring->used[i] = true;
    *slot = ring->base + (size_t)i * ring->slot_bytes;
    *done = ring->done[i];
    return DM_OK;
}

}  // namespace

extern "C" int dm_pack_targets(const uint64_t *pointers, const int32_t *meta, int B, int64_t *labels_out,
                               float *boxes_out, int32_t *meta_out, void *stream)
{
    if (B < 0 || (B > 0 && (!pointers || !meta || !labels_out || !boxes_out || !meta_out))) return DM_E_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(boxes_out) & 15) != 0) return DM_E_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (B > PACK_CHUNK) {
        const size_t ptr_bytes = (size_t)B * 16, meta_bytes = (size_t)(2 * B + 1) * 4;
        unsigned char *slot = nullptr;
        hipEvent_t done = nullptr;
        int rc = table_slot(ptr_bytes + meta_bytes, &slot, &done);
        if (rc != DM_OK) return rc;
        std::memcpy(slot, pointers, ptr_bytes);
        std::memcpy(slot + ptr_bytes, meta, meta_bytes);
        hipLaunchKernelGGL(pack_targets_table_kernel, dim3((unsigned)B), dim3(64), 0, st,
                           reinterpret_cast<const uint64_t *>(slot), reinterpret_cast<const int32_t *>(slot + ptr_bytes), B,
                           labels_out, boxes_out, meta_out);
        rc = dm::check_hip(hipGetLastError(), "pack_targets_table_kernel launch");
        if (rc != DM_OK) return rc;
        return dm::check_hip(hipEventRecord(done, st), "hipEventRecord");
    }
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
