// ls_lab.hip -- DIAGNOSTIC: the reverse sweep of the lanes = states backward (scan_ls.hip) as a register-only loop, to find
// what its VALU instructions cost at 3 waves per SIMD and which part of the sequence is responsible.
//   MODE 0: the sweep as in the kernel (5 DPP fmacs + 4 muls per token, 2 transposed reductions per 16 tokens)
//   MODE 1: without the reductions        MODE 2: reductions only
//   MODE 3: the g chain only (fmac_dpp, mul)      MODE 4: everything but with plain (non-DPP) fmas instead of tok_fma
//   MODE 5: MODE 0 with the merges' s_nop removed (timing only: the hazard is then unprotected)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
__host__ __device__ constexpr int br4(int k) { return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3); }
template <int K, bool DPP> __device__ __forceinline__ float tok_fma(float acc, float src, float mul) {
    if (DPP) asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(br4(K)));
    else acc = fmaf(src, mul, acc);
    return acc;
}
template <int I, typename F> __device__ __forceinline__ void sfor_down(F&& f) {
    if constexpr (I > 0) { f(std::integral_constant<int, I - 1>{}); sfor_down<I - 1>(f); }
}
template <bool NOP> __device__ __forceinline__ float merge8(float X, float Y) {
    float Z;
    if (NOP) asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(Z) : "v"(X), "v"(Y));
    else asm("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(Z) : "v"(X), "v"(Y));
    return Z;
}
template <bool NOP> __device__ __forceinline__ float merge4(float X, float Y) {
    float Z;
    if (NOP) asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(Z) : "v"(X), "v"(Y));
    else asm("v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(Z) : "v"(X), "v"(Y));
    return Z;
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float merge2(float X, float Y, bool hi) { const float k = hi ? Y : X, s = hi ? X : Y; return k + dpp_mov<0x4e>(s); }
__device__ __forceinline__ float merge1(float X, float Y, bool hi) { const float k = hi ? Y : X, s = hi ? X : Y; return k + dpp_mov<0xb1>(s); }
template <int K, bool NOP> __device__ __forceinline__ void reduce_down(float (&s)[16], float (&z)[8], float (&w)[4], float (&v)[2], float& out, int li) {
    if constexpr ((K & 1) == 0) z[K / 2] = merge8<NOP>(s[K], s[K + 1]);
    if constexpr ((K & 3) == 0) w[K / 4] = merge4<NOP>(z[K / 2], z[K / 2 + 1]);
    if constexpr ((K & 7) == 0) v[K / 8] = merge2(w[K / 4], w[K / 4 + 1], (li & 2) != 0);
    if constexpr (K == 0) out = merge1(v[0], v[1], (li & 1) != 0);
}

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* in, int iters) {
    extern __shared__ float pad[];
    const int li = threadIdx.x & 15;
    float Bv[16], Cv[16], a[16], h[16], dBv[16], dCv[16];
    for (int i = 0; i < 16; ++i) { Bv[i] = in[i]; Cv[i] = in[16 + i]; a[i] = 0.9f + 1e-3f * in[32 + i]; h[i] = in[48 + i]; dBv[i] = 0; dCv[i] = 0; }
    float dl = in[64 + threadIdx.x], w = in[128 + threadIdx.x], dy = in[192 + threadIdx.x], A2 = -0.3f, ag = 0.1f, dA0 = 0, dA1 = 0, acc = 0;
    asm volatile("s_nop 1" : "+v"(dl), "+v"(w), "+v"(dy));
    constexpr bool DPP = MODE != 4, NOP = MODE != 5;
    for (int it = 0; it < iters; ++it) {
        float s1[16], s2[16], z1[8], z2[8], w1[4], w2[4], v1[2], v2[2], S1 = 0, S2 = 0;
        sfor_down<16>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (MODE == 2) { s1[k] = Bv[k] * ag; s2[k] = Cv[k] * ag; }
            else {
                const float gk = tok_fma<k, DPP>(ag, dy, Cv[k]);
                ag = gk * a[k];
                if (MODE != 3) {
                    const float x = ag * h[k > 0 ? k - 1 : 0];
                    s1[k] = gk * Bv[k];
                    s2[k] = A2 * x;
                    if constexpr (k & 1) dA1 = tok_fma<k, DPP>(dA1, dl, x); else dA0 = tok_fma<k, DPP>(dA0, dl, x);
                    dBv[k] = tok_fma<k, DPP>(dBv[k], w, gk);
                    dCv[k] = tok_fma<k, DPP>(dCv[k], dy, h[k]);
                }
            }
            if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5) {
                reduce_down<k, NOP>(s1, z1, w1, v1, S1, li);
                reduce_down<k, NOP>(s2, z2, w2, v2, S2, li);
            } else if (MODE == 1) { acc += s1[k] + s2[k]; }
        });
        acc += S1 + S2;
        ag = ag * 0.5f + 0.01f;
        a[it & 15] += 1e-6f;                 // keep the loop body from being hoisted
    }
    float r = acc + ag + dA0 + dA1;
    for (int i = 0; i < 16; ++i) r += dBv[i] + dCv[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> static void run(const char* name, int valu_per_iter, float* out, float* in) {
    const int iters = 2000, blocks = 256 * 3;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 53000, 0, out, in, iters); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 53000, 0, out, in, iters); (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 3 blocks x 4 waves per CU = 3 waves per SIMD; per SIMD: 3 * iters bodies
    const double ns_body = ms * 1e6 / (3.0 * iters);
    printf("%-44s %8.1f ns per 16-token sweep per SIMD  (%5.2f ns per token; ~%d VALU -> %.2f ns each)\n", name, ns_body, ns_body / 16, valu_per_iter, ns_body / valu_per_iter);
}
int main() {
    float *out, *in; (void)hipMalloc(&out, 256 * 3 * 256 * 4); (void)hipMalloc(&in, 4096 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 97) + 0.1f; (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 53000);
    run<0>("0 sweep as in the kernel", 16 * 9 + 2 * 39, out, in);
    run<1>("1 without the two reductions", 16 * 11, out, in);
    run<2>("2 the two reductions only", 32 + 2 * 39, out, in);
    run<3>("3 g chain only (fmac_dpp + mul)", 32, out, in);
    run<4>("4 plain fma instead of DPP operands", 16 * 9 + 2 * 39, out, in);
    run<5>("5 as 0, merges without s_nop", 16 * 9 + 2 * 39, out, in);
    return 0;
}
