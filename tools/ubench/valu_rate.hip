// Microbenchmark: cycles per wave64 VALU instruction per SIMD on gfx950, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {          // 8 independent fma
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (OP == 1) {   // dependent chain
            x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
            x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b); x0 = __builtin_fmaf(x0, a, b);
        } else if (OP == 2) {   // 8 independent rcp
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else if (OP == 3) {   // 8 independent min (asm, no canonicalize)
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(x0) : "v"(a)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x1) : "v"(a));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(x2) : "v"(a)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x3) : "v"(a));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(x4) : "v"(a)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x5) : "v"(a));
            asm volatile("v_min_f32 %0, %0, %1" : "+v"(x6) : "v"(a)); asm volatile("v_min_f32 %0, %0, %1" : "+v"(x7) : "v"(a));
        } else if (OP == 4) {   // 4 independent pk_fma (8 lanes-ops)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7}, aa = {a, a}, bb = {b, b};
            y0 = __builtin_elementwise_fma(y0, aa, bb); y1 = __builtin_elementwise_fma(y1, aa, bb);
            y2 = __builtin_elementwise_fma(y2, aa, bb); y3 = __builtin_elementwise_fma(y3, aa, bb);
            x0 = y0.x; x1 = y0.y; x2 = y1.x; x3 = y1.y; x4 = y2.x; x5 = y2.y; x6 = y3.x; x7 = y3.y;
        } else if (OP == 5) {   // 8 independent f64 add
            double d0 = x0, d1 = x1; (void)d1;
            asm volatile("v_add_f64 %0, %0, %0" : "+v"(d0)); x0 = (float)d0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int OP>
void run(const char *name, int insts_per_iter)
{
    float *out;
    hipMalloc(&out, 256 * 32 * 64 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpe : {1, 2, 4, 8}) {        // waves per SIMD: block = 256 threads (one wave per SIMD), wpe blocks per CU
        dim3 grid(256 * wpe), block(256);
        k<OP><<<grid, block>>>(out, 100, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<OP><<<grid, block>>>(out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_insts_per_simd = (double)iters * insts_per_iter * wpe;
        printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, wpe, ms,
               ms * 1e6 / wave_insts_per_simd, ms * 1e6 / wave_insts_per_simd * 2.4);
    }
    hipFree(out);
}

int main()
{
    run<0>("fma x8 independent", 8);
    run<1>("fma x8 dependent", 8);
    run<2>("rcp x8 independent", 8);
    run<3>("min x8 independent", 8);
    run<4>("pk_fma x4 independent", 4);
    return 0;
}
