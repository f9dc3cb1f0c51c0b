/*
 * The matcher step from plain C: nothing but include/detr_match.h, the HIP runtime's C API and malloc -- no torch,
 * no Python.  What a binding in any host language does (the call sequence of
 * /root/reference/models/matcher.py:59-95 through the C ABI):
 *
 *     abi_demo <fixture>
 *
 * fixture (little endian, written by tests/test_gpu_c_abi.py):
 *     int32  S, B, Q, C, cls_mode, M            sets, images, queries, logits per query, 0 ce / 1 focal, total targets
 *     float  w_class, w_bbox, w_giou, alpha, gamma
 *     int32  sizes[B]
 *     float  logits[S][B][Q][C], float boxes[S][B][Q][4]
 *     int64  labels[M], float tboxes[M][4]
 *     int64  expect_src[S][M], expect_tgt[S][M]      (-7 where no pair is written)
 * Runs the step twice -- dm_match_targets_f32 on the per-image target arrays where they are (with room to pack,
 * should the shapes ask for it), then dm_pack_targets + dm_match_f32 -- and compares both results with the
 * expected indices.  Prints "ok" and exits 0 when everything agrees.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "detr_match.h"

#define HIP(call)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                      \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)

static int read_all(FILE *f, void *dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: abi_demo <fixture>\n"); return 2; }
    if (dm_abi_version() != DM_ABI_VERSION) { fprintf(stderr, "library / header ABI mismatch\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int32_t head[6];
    float prm_f[5];
    if (read_all(f, head, sizeof head) || read_all(f, prm_f, sizeof prm_f)) return 2;
    const int S = head[0], B = head[1], Q = head[2], C = head[3], M = head[5];
    DmParams prm = {head[4], prm_f[0], prm_f[1], prm_f[2], prm_f[3], prm_f[4]};
    int32_t *sizes = malloc(sizeof(int32_t) * (size_t)B);
    const size_t n_logits = (size_t)S * B * Q * C, n_boxes = (size_t)S * B * Q * 4, n_idx = (size_t)S * (M > 0 ? M : 1);
    float *logits = malloc(4 * n_logits), *boxes = malloc(4 * n_boxes), *tboxes = malloc(16 * (size_t)(M > 0 ? M : 1));
    int64_t *labels = malloc(8 * (size_t)(M > 0 ? M : 1)), *want = malloc(16 * n_idx), *got = malloc(16 * n_idx);
    if (read_all(f, sizes, 4 * (size_t)B) || read_all(f, logits, 4 * n_logits) || read_all(f, boxes, 4 * n_boxes) ||
        read_all(f, labels, 8 * (size_t)M) || read_all(f, tboxes, 16 * (size_t)M) || read_all(f, want, 16 * (size_t)S * M)) {
        fprintf(stderr, "short fixture\n");
        return 2;
    }
    fclose(f);

    /* device copies: predictions per set; targets per image, each image an allocation of its own (as the reference's
       list of target dicts holds them) */
    float *d_logits, *d_boxes, *d_cost, *d_pack_boxes;
    int64_t *d_idx, *d_pack_labels;
    int32_t *d_status, *d_meta;
    HIP(hipMalloc((void **)&d_logits, 4 * n_logits));
    HIP(hipMalloc((void **)&d_boxes, 4 * n_boxes));
    HIP(hipMemcpy(d_logits, logits, 4 * n_logits, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_boxes, boxes, 4 * n_boxes, hipMemcpyHostToDevice));
    uint64_t *pointers = malloc(16 * (size_t)B);
    int32_t *meta = malloc(4 * (size_t)(2 * B + 1));
    int max_T = 0, off = 0;
    for (int b = 0; b < B; ++b) {
        void *dl, *db;
        const size_t t = (size_t)(sizes[b] > 0 ? sizes[b] : 1);
        HIP(hipMalloc(&dl, 8 * t));
        HIP(hipMalloc(&db, 16 * t));
        HIP(hipMemcpy(dl, labels + off, 8 * (size_t)sizes[b], hipMemcpyHostToDevice));
        HIP(hipMemcpy(db, tboxes + 4 * (size_t)off, 16 * (size_t)sizes[b], hipMemcpyHostToDevice));
        pointers[2 * b] = (uint64_t)(uintptr_t)dl;
        pointers[2 * b + 1] = (uint64_t)(uintptr_t)db;
        meta[b] = off;
        off += sizes[b];
        if (sizes[b] > max_T) max_T = sizes[b];
    }
    meta[B] = off;
    if (off != M) { fprintf(stderr, "sizes do not add up to M\n"); return 2; }
    /* processing order: images by falling number of targets (stable) */
    for (int b = 0; b < B; ++b) meta[B + 1 + b] = b;
    for (int i = 1; i < B; ++i) {
        const int32_t v = meta[B + 1 + i];
        int j = i;
        while (j > 0 && sizes[meta[B + j]] < sizes[v]) { meta[B + 1 + j] = meta[B + j]; --j; }
        meta[B + 1 + j] = v;
    }

    DmSet *sets = calloc((size_t)S, sizeof(DmSet));
    const int q_stride = (Q + 31) / 32 * 32;
    int64_t cost_floats = 0;
    for (int s = 0; s < S; ++s) {
        sets[s].logits = d_logits + (size_t)s * B * Q * C;
        sets[s].boxes = d_boxes + (size_t)s * B * Q * 4;
        sets[s].cost_off = cost_floats;
        sets[s].out_off = (int64_t)s * M;
        sets[s].Q = Q;
        sets[s].q_stride = q_stride;      /* (logits_stride / boxes_stride 0: contiguous) */
        cost_floats += (int64_t)M * q_stride;
    }
    HIP(hipMalloc((void **)&d_cost, 4 * (size_t)(cost_floats > 0 ? cost_floats : 1)));
    HIP(hipMalloc((void **)&d_idx, 16 * n_idx));
    HIP(hipMalloc((void **)&d_status, 4));
    HIP(hipMalloc((void **)&d_pack_labels, 8 * (size_t)(M > 0 ? M : 1)));
    HIP(hipMalloc((void **)&d_pack_boxes, 16 * (size_t)(M > 0 ? M : 1)));
    HIP(hipMalloc((void **)&d_meta, 4 * (size_t)(2 * B + 1)));
    hipStream_t stream;
    HIP(hipStreamCreate(&stream));

    for (int pass = 0; pass < 2; ++pass) {
        HIP(hipMemsetAsync(d_status, 0, 4, stream));
        HIP(hipMemsetAsync(d_idx, 0xF9, 16 * n_idx, stream));      /* 0xF9F9...F9 marks untouched slots */
        int rc, packed = -1;
        if (pass == 0) {
            rc = dm_match_targets_f32(sets, S, B, C, pointers, meta, max_T, &prm, d_pack_labels, d_pack_boxes, d_meta, d_cost,
                                      d_idx, d_idx + (size_t)S * M, NULL, d_status, stream, &packed);
        } else {
            rc = dm_pack_targets(pointers, meta, B, d_pack_labels, d_pack_boxes, d_meta, stream);
            if (rc == DM_OK)
                rc = dm_match_f32(sets, S, B, C, d_pack_labels, d_pack_boxes, d_meta, d_meta + B + 1, max_T, &prm, d_cost,
                                  d_idx, d_idx + (size_t)S * M, d_status, stream);
        }
        if (rc != DM_OK) { fprintf(stderr, "pass %d: %s (%s)\n", pass, dm_strerror(rc), dm_last_hip_error()); return 1; }
        HIP(hipStreamSynchronize(stream));
        int32_t status = -1;
        HIP(hipMemcpy(&status, d_status, 4, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(got, d_idx, 16 * (size_t)S * M, hipMemcpyDeviceToHost));
        if (status != 0) { fprintf(stderr, "pass %d: status word %d\n", pass, status); return 1; }
        long bad = 0;
        for (size_t k = 0; k < 2 * (size_t)S * M; ++k) {
            const int64_t w = want[k], g = got[k] == (int64_t)0xF9F9F9F9F9F9F9F9LL ? -7 : got[k];
            bad += w != g;
        }
        if (bad) { fprintf(stderr, "pass %d: %ld of %zu index slots differ\n", pass, bad, 2 * (size_t)S * M); return 1; }
        printf("pass %d (%s): %d sets x %d images, %d targets: indices as expected\n", pass,
               pass == 0 ? (packed ? "dm_match_targets_f32, packed" : "dm_match_targets_f32, targets read in place")
                         : "dm_pack_targets + dm_match_f32", S, B, M);
    }
    printf("ok\n");
    return 0;
}
