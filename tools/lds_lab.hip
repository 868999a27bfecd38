// lds_lab.hip -- DIAGNOSTIC microbenchmarks: LDS float atomics vs plain LDS writes, v_exp_f32 rate, fma issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ void __launch_bounds__(512) k_lds(float* out, int iters) {
    __shared__ float t[8 * 64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 64 * 8; i += 512) t[i] = 0.f;
    __syncthreads();
    float v = lane * 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) atomicAdd(&t[k * 64 + lane], v);                 // all waves, same addresses
            if (MODE == 1) atomicAdd(&t[(wave * 8 + k) * 64 + lane], v);    // wave-private addresses
            if (MODE == 2) t[(wave * 8 + k) * 64 + lane] = v;               // plain write
            if (MODE == 3) v += t[(wave * 8 + k) * 64 + lane];              // plain read
        }
        v += 1e-6f;
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = t[threadIdx.x] + v;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_valu(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
                         a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7); }
        if (MODE == 1) { a0 = fmaf(a0, 0.999f, 0.1f); a1 = fmaf(a1, 0.999f, 0.1f); a2 = fmaf(a2, 0.999f, 0.1f); a3 = fmaf(a3, 0.999f, 0.1f);
                         a4 = fmaf(a4, 0.999f, 0.1f); a5 = fmaf(a5, 0.999f, 0.1f); a6 = fmaf(a6, 0.999f, 0.1f); a7 = fmaf(a7, 0.999f, 0.1f); }
        if (MODE == 2) { a0 = fmaf(a0, 0.999f, 0.1f); a0 = fmaf(a0, 0.999f, 0.1f); a0 = fmaf(a0, 0.999f, 0.1f); a0 = fmaf(a0, 0.999f, 0.1f);
                         a0 = fmaf(a0, 0.999f, 0.1f); a0 = fmaf(a0, 0.999f, 0.1f); a0 = fmaf(a0, 0.999f, 0.1f); a0 = fmaf(a0, 0.999f, 0.1f); }   // dependent chain
    }
    if (MODE == 3) {      // the same eight chains as four packed pairs: one v_pk_fma_f32 does the work of two v_fma_f32
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
        const f2 m = {0.999f, 0.999f}, c = {0.1f, 0.1f};
        for (int it = 0; it < iters; ++it) {
            p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
            p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));      // keep them packed and in the loop
        }
        a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; CK(hipMalloc(&out, 1 << 26));
    const int iters = 2000; const double clk = 2.4e9;
    const char* names[] = {"ds_add_f32 shared addr", "ds_add_f32 private addr", "ds_write_b32", "ds_read_b32"};
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_lds<0>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("%-26s 1 WG/CU x 8 waves: %.1f cyc per wave-instr per CU\n", names[0], ms * 1e-3 * clk / (iters * 8.0 * 8));
    ms = timeit([&] { hipLaunchKernelGGL(k_lds<1>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("%-26s 1 WG/CU x 8 waves: %.1f cyc per wave-instr per CU\n", names[1], ms * 1e-3 * clk / (iters * 8.0 * 8));
    ms = timeit([&] { hipLaunchKernelGGL(k_lds<2>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("%-26s 1 WG/CU x 8 waves: %.1f cyc per wave-instr per CU\n", names[2], ms * 1e-3 * clk / (iters * 8.0 * 8));
    ms = timeit([&] { hipLaunchKernelGGL(k_lds<3>, dim3(256), dim3(512), 0, 0, out, iters); }); printf("%-26s 1 WG/CU x 8 waves: %.1f cyc per wave-instr per CU\n", names[3], ms * 1e-3 * clk / (iters * 8.0 * 8));
    for (int wpb = 1; wpb <= 8; wpb *= 2) {   // waves per SIMD = blocks per CU (256-thread blocks = 1 wave per SIMD each)
        const int it2 = 20000;
        float m0 = timeit([&] { hipLaunchKernelGGL(k_valu<0>, dim3(256 * wpb), dim3(256), 0, 0, out, it2); });
        float m1 = timeit([&] { hipLaunchKernelGGL(k_valu<1>, dim3(256 * wpb), dim3(256), 0, 0, out, it2); });
        float m2 = timeit([&] { hipLaunchKernelGGL(k_valu<2>, dim3(256 * wpb), dim3(256), 0, 0, out, it2); });
        float m3 = timeit([&] { hipLaunchKernelGGL(k_valu<3>, dim3(256 * wpb), dim3(256), 0, 0, out, it2); });
        printf("%d wave(s)/SIMD: v_exp_f32 %.2f cyc/instr/SIMD, independent fma %.2f, dependent fma chain %.2f, v_pk_fma_f32 %.2f "
               "(= %.2f per fma it replaces)\n", wpb,
               m0 * 1e-3 * clk / (it2 * 8.0 * wpb), m1 * 1e-3 * clk / (it2 * 8.0 * wpb), m2 * 1e-3 * clk / (it2 * 8.0 * wpb),
               m3 * 1e-3 * clk / (it2 * 4.0 * wpb), m3 * 1e-3 * clk / (it2 * 8.0 * wpb));
    }
    return 0;
}
