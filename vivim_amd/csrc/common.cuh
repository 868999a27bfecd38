// common.cuh -- device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vivim_hip.h"

namespace vivim {

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;

using f16_t = _Float16;
using bf16_t = __bf16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---- fast transcendental forms (one hardware op each, as the reference's --use_fast_math build) ----
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_fast(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// softplus with the reference's threshold (fwd_kernel.cuh:155: x <= 20 ? log1p(exp(x)) : x).
// log1p(e) via Kahan's correction so that small e keeps full relative precision.
__device__ __forceinline__ float softplus_ref(float x) {
    if (x > 20.0f) return x;
    const float e = fast_exp(x);
    const float w = 1.0f + e;
    const float d = w - 1.0f;
    return d == 0.0f ? e : fast_log(w) * (e * fast_rcp(d));
}

// ---- element conversion ----
template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// ---- K consecutive elements per lane: 16-byte vector path when in range and aligned ----
template <typename T, int BYTES> struct Pack;
template <typename T> struct Pack<T, 16> { using type = u32x4; };
template <typename T> struct Pack<T, 8> { using type = u32x2; };
template <typename T> struct Pack<T, 4> { using type = uint32_t; };

// nv = number of valid elements at p[0..K) (<=0: none). Invalid slots read as 0.
template <typename T, int K>
__device__ __forceinline__ void load_k(const T* __restrict__ p, int nv, float (&v)[K]) {
    constexpr int BYTES = K * (int)sizeof(T);
    constexpr int VB = BYTES >= 16 ? 16 : BYTES;        // bytes per vector access
    constexpr int NV = BYTES / VB;                      // vector accesses
    constexpr int EPV = VB / (int)sizeof(T);            // elements per vector
    if (nv >= K && (reinterpret_cast<uintptr_t>(p) & (VB - 1)) == 0) {
        using V = typename Pack<T, VB>::type;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            union { V raw; T e[EPV]; } u;
            u.raw = reinterpret_cast<const V*>(p)[i];
#pragma unroll
            for (int j = 0; j < EPV; ++j) v[i * EPV + j] = to_f32<T>(u.e[j]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = k < nv ? to_f32<T>(p[k]) : 0.0f;
    }
}

template <typename T, int K>
__device__ __forceinline__ void store_k(T* __restrict__ p, int nv, const float (&v)[K]) {
    constexpr int BYTES = K * (int)sizeof(T);
    constexpr int VB = BYTES >= 16 ? 16 : BYTES;
    constexpr int NV = BYTES / VB;
    constexpr int EPV = VB / (int)sizeof(T);
    if (nv >= K && (reinterpret_cast<uintptr_t>(p) & (VB - 1)) == 0) {
        using V = typename Pack<T, VB>::type;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            union { V raw; T e[EPV]; } u;
#pragma unroll
            for (int j = 0; j < EPV; ++j) u.e[j] = from_f32<T>(v[i * EPV + j]);
            reinterpret_cast<V*>(p)[i] = u.raw;
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (k < nv) p[k] = from_f32<T>(v[k]);
    }
}

// ---- wave64 scans of affine maps  x -> P*x + H  ----
// Forward: lane l ends up with the composition of lanes 0..l (lane 0 applied first).
__device__ __forceinline__ void wave_scan_affine_fwd(float& P, float& H, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float Pp = __shfl_up(P, off, kWave);
        const float Hp = __shfl_up(H, off, kWave);
        if (lane >= off) {
            H = fmaf(P, Hp, H);   // apply the earlier segment first, then ours
            P = P * Pp;
        }
    }
}
// Reverse: lane l ends up with the composition of lanes l..63 (lane 63 applied first).
__device__ __forceinline__ void wave_scan_affine_rev(float& P, float& H, int lane) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const float Pn = __shfl_down(P, off, kWave);
        const float Hn = __shfl_down(H, off, kWave);
        if (lane + off < kWave) {
            H = fmaf(P, Hn, H);
            P = P * Pn;
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// compiler-level ordering of this wave's LDS traffic (hardware keeps a wave's DS ops in order)
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

}  // namespace vivim
