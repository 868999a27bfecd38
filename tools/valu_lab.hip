// valu_lab.hip -- DIAGNOSTIC microbenchmarks for the "lanes = states" scan design (round 2):
//   * do v_exp_f32 and v_fma_f32 from different waves of one SIMD overlap, or do their issue costs add?
//   * cost of DPP operands (row_newbcast) on v_mul / v_fmac, of v_permlane32_swap / v_permlane16_swap,
//     of v_cndmask + DPP adds (the transposed 16-lane reduction), of ds_read_b128 beside VALU work.
// Every figure is SIMD cycles per loop body, from s_memtime inside the kernel (clock-independent), median over waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REGS16 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
               "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])

// MODE 0: 16 fma                      MODE 1: 4 exp                      MODE 2: 16 fma + 4 exp (interleaved 4:1)
// MODE 3: 16 v_mul_f32_dpp row_newbcast   MODE 4: 16 v_fmac_f32_dpp row_newbcast
// MODE 5: 8 x (v_permlane32_swap)     MODE 6: 8 x v_permlane16_swap      MODE 7: 16 v_add_f32_dpp row_ror:8 bank_mask
// MODE 8: 16 fma + 2 ds_read_b128     MODE 9: 8 x (cndmask, cndmask, add_dpp quad_perm)   MODE 10: 16 v_mul with SGPR operand
// MODE 11: 16 fma + 8 exp             MODE 12: 8 exp
template <int MODE>
__global__ void __launch_bounds__(256) k_lab(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    float r[16], e[8];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) e[i] = -threadIdx.x * 1e-3f - i;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    const float m = 0.999f, c = 0.1f;
    const unsigned lp = (unsigned)(size_t)(lds + (threadIdx.x & 63) * 4);   // low 32 bits of a flat LDS address = LDS offset
    float sgp = __builtin_amdgcn_readfirstlane(__float_as_int(m)) ? m : c;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 8 || MODE == 11) {
#define F4(a, b, c2, d) "v_fma_f32 %" #a ", %" #a ", %[m], %[c]\n\tv_fma_f32 %" #b ", %" #b ", %[m], %[c]\n\t" \
                        "v_fma_f32 %" #c2 ", %" #c2 ", %[m], %[c]\n\tv_fma_f32 %" #d ", %" #d ", %[m], %[c]\n\t"
            if (MODE == 0)
                asm volatile(F4(0, 1, 2, 3) F4(4, 5, 6, 7) F4(8, 9, 10, 11) F4(12, 13, 14, 15) : REGS16 : [m] "v"(m), [c] "v"(c));
            if (MODE == 2)
                asm volatile(F4(0, 1, 2, 3) "v_exp_f32 %16, %16\n\t" F4(4, 5, 6, 7) "v_exp_f32 %17, %17\n\t"
                             F4(8, 9, 10, 11) "v_exp_f32 %18, %18\n\t" F4(12, 13, 14, 15) "v_exp_f32 %19, %19\n\t"
                             : REGS16, "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]) : [m] "v"(m), [c] "v"(c));
            if (MODE == 11)
                asm volatile(F4(0, 1, 2, 3) "v_exp_f32 %16, %16\n\tv_exp_f32 %20, %20\n\t" F4(4, 5, 6, 7) "v_exp_f32 %17, %17\n\tv_exp_f32 %21, %21\n\t"
                             F4(8, 9, 10, 11) "v_exp_f32 %18, %18\n\tv_exp_f32 %22, %22\n\t" F4(12, 13, 14, 15) "v_exp_f32 %19, %19\n\tv_exp_f32 %23, %23\n\t"
                             : REGS16, "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]) : [m] "v"(m), [c] "v"(c));
            if (MODE == 8) {
                float4 a, b;
                asm volatile("ds_read_b128 %16, %18\n\tds_read_b128 %17, %18 offset:1024\n\t"
                             F4(0, 1, 2, 3) F4(4, 5, 6, 7) F4(8, 9, 10, 11) F4(12, 13, 14, 15) "s_waitcnt lgkmcnt(0)\n\t"
                             : REGS16, "=&v"(a), "=&v"(b) : "v"(lp), [m] "v"(m), [c] "v"(c) : "memory");
                e[0] += a.x + b.y;
            }
        }
        if (MODE == 1)
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                         : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
        if (MODE == 12)
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                         "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
                         : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]));
        if (MODE == 3) {
#define D4(op, a, b, c2, d, k) op " %" #a ", %[s], %" #a " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t" \
                               op " %" #b ", %[s], %" #b " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t" \
                               op " %" #c2 ", %[s], %" #c2 " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t" \
                               op " %" #d ", %[s], %" #d " row_newbcast:" #k " row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(D4("v_mul_f32_dpp", 0, 1, 2, 3, 1) D4("v_mul_f32_dpp", 4, 5, 6, 7, 5) D4("v_mul_f32_dpp", 8, 9, 10, 11, 9)
                         D4("v_mul_f32_dpp", 12, 13, 14, 15, 13) : REGS16 : [s] "v"(m));
        }
        if (MODE == 4)
            asm volatile(D4("v_fmac_f32_dpp", 0, 1, 2, 3, 1) D4("v_fmac_f32_dpp", 4, 5, 6, 7, 5) D4("v_fmac_f32_dpp", 8, 9, 10, 11, 9)
                         D4("v_fmac_f32_dpp", 12, 13, 14, 15, 13) : REGS16 : [s] "v"(c));
        if (MODE == 5)
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\t"
                         "v_permlane32_swap_b32 %6, %7\n\tv_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\t"
                         "v_permlane32_swap_b32 %12, %13\n\tv_permlane32_swap_b32 %14, %15\n\t" : REGS16);
        if (MODE == 6)
            asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\t"
                         "v_permlane16_swap_b32 %6, %7\n\tv_permlane16_swap_b32 %8, %9\n\tv_permlane16_swap_b32 %10, %11\n\t"
                         "v_permlane16_swap_b32 %12, %13\n\tv_permlane16_swap_b32 %14, %15\n\t" : REGS16);
        if (MODE == 7) {
#define A4(a, b, c2, d) "v_add_f32_dpp %" #a ", %" #b ", %" #b " row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
                        "v_add_f32_dpp %" #a ", %" #c2 ", %" #c2 " row_ror:8 row_mask:0xf bank_mask:0xc\n\t" \
                        "v_add_f32_dpp %" #d ", %" #b ", %" #b " row_ror:4 row_mask:0xf bank_mask:0x5\n\t" \
                        "v_add_f32_dpp %" #d ", %" #c2 ", %" #c2 " row_ror:12 row_mask:0xf bank_mask:0xa\n\t"
            asm volatile(A4(0, 1, 2, 3) A4(4, 5, 6, 7) A4(8, 9, 10, 11) A4(12, 13, 14, 15) : REGS16);
        }
        if (MODE == 9) {
            const unsigned long long msk = 0xccccccccccccccccull;
#define C3(a, b, t) "v_cndmask_b32 %" #t ", %" #a ", %" #b ", %[k]\n\tv_cndmask_b32 %" #a ", %" #b ", %" #a ", %[k]\n\t" \
                    "s_nop 0\n\tv_add_f32_dpp %" #a ", %" #t ", %" #a " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(C3(0, 1, 16) C3(2, 3, 17) C3(4, 5, 18) C3(6, 7, 19) C3(8, 9, 16) C3(10, 11, 17) C3(12, 13, 18) C3(14, 15, 19)
                         : REGS16, "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]) : [k] "s"(msk));
        }
        if (MODE == 10) {
#define M4(a, b, c2, d) "v_mul_f32 %" #a ", %[s], %" #a "\n\tv_mul_f32 %" #b ", %[s], %" #b "\n\tv_mul_f32 %" #c2 ", %[s], %" #c2 "\n\tv_mul_f32 %" #d ", %[s], %" #d "\n\t"
            asm volatile(M4(0, 1, 2, 3) M4(4, 5, 6, 7) M4(8, 9, 10, 11) M4(12, 13, 14, 15) : REGS16 : [s] "s"(sgp));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int i = 0; i < 8; ++i) s += e[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static double run(int wps, int iters, float* out, unsigned long long* cyc, double* wall_us) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k_lab<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_lab<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    *wall_us = ms * 1e3;
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / iters;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 4 * 8));
    const int iters = 4000;
    const char* names[] = {"16 fma", "4 exp", "16 fma + 4 exp", "16 mul_dpp newbcast", "16 fmac_dpp newbcast", "8 permlane32_swap",
                           "8 permlane16_swap", "16 add_dpp bank_mask", "16 fma + 2 ds_read_b128", "8 x (2 cndmask + add_dpp)",
                           "16 mul sgpr", "16 fma + 8 exp", "8 exp"};
    const int wpss[] = {1, 2, 3, 4, 6, 8};
    printf("cycles of one wave per loop body (s_memtime, median over waves); the SIMD hosts `w` such waves, so SIMD cycles per body = cyc / w\n");
    printf("%-28s", "body \\ waves per SIMD");
    for (int w : wpss) printf("  w=%d cyc (per-SIMD) wall", w);
    printf("\n");
#define ROW(M) { printf("%-28s", names[M]); for (int w : wpss) { double us; double c = run<M>(w, iters, out, cyc, &us); \
                 printf("  %8.1f (%6.1f) %6.0fus", c, c / w, us); } printf("\n"); }
    ROW(0) ROW(1) ROW(12) ROW(2) ROW(11) ROW(3) ROW(4) ROW(10) ROW(5) ROW(6) ROW(7) ROW(9) ROW(8)
    return 0;
}
