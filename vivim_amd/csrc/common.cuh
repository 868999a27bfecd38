// common.cuh -- device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vivim_hip.h"

namespace vivim {

// Kernel-selection overrides (capi.hip): 0 = automatic.  Initialised from VIVIM_FWD_VARIANT / VIVIM_BWD_VARIANT,
// changed at run time through vivim_set_tuning() (tests and tools sweep the variants in one process).
//   forward : 1 n-split K=8, 2 n-split K=4, 3 generic, 5 lanes=channels (needs the forward workspace)
//   backward: 3 generic
int tuning_fwd_variant();
int tuning_bwd_variant();

constexpr int kWave = 64;
constexpr int kChunk = 256;   // tokens per row of the checkpoint tensor x (contract between fwd and bwd kernels)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

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
    // Branch-free (selects only): divergent branches in the scan inner loops split the scheduling regions and
    // force s_waitcnt drains.  min() keeps exp finite on the x > 20 side whose result is discarded.
    const float e = fast_exp(fminf(x, 20.0f));
    const float w = 1.0f + e;
    const float d = w - 1.0f;
    const float r = fast_log(w) * (e * fast_rcp(d == 0.0f ? 1.0f : d));
    return x > 20.0f ? x : (d == 0.0f ? e : r);
}

// ---- element conversion ----
template <typename T> __device__ __forceinline__ float to_f32(T v) { return static_cast<float>(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return static_cast<T>(v); }

// ---- K consecutive elements per lane: 16-byte vector path when in range and aligned ----
template <typename T, int BYTES> struct Pack;
template <typename T> struct Pack<T, 16> { using type = u32x4; };
template <typename T> struct Pack<T, 8> { using type = u32x2; };
template <typename T> struct Pack<T, 4> { using type = uint32_t; };
template <typename T> struct Pack<T, 2> { using type = uint16_t; };

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

// Raw (unconverted) K elements kept in registers, so a load can be issued long before its use.
template <typename T, int K> struct RawK { T e[K]; };

template <typename T, int K>
__device__ __forceinline__ RawK<T, K> load_raw(const T* __restrict__ p, int nv) {
    constexpr int BYTES = K * (int)sizeof(T);
    constexpr int VB = BYTES >= 16 ? 16 : BYTES;
    constexpr int NV = BYTES / VB;
    constexpr int EPV = VB / (int)sizeof(T);
    RawK<T, K> r;
    if (nv >= K && (reinterpret_cast<uintptr_t>(p) & (VB - 1)) == 0) {
        using V = typename Pack<T, VB>::type;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            union { V raw; T e[EPV]; } u;
            u.raw = reinterpret_cast<const V*>(p)[i];
#pragma unroll
            for (int j = 0; j < EPV; ++j) r.e[i * EPV + j] = u.e[j];
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) r.e[k] = k < nv ? p[k] : from_f32<T>(0.0f);
    }
    return r;
}
template <typename T, int K>
__device__ __forceinline__ void unpack(const RawK<T, K>& r, float (&v)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = to_f32<T>(r.e[k]);
}

// Branch-free forms for kernels whose host dispatch guarantees 16-byte-aligned rows and a sequence
// length that is a multiple of K: a lane is either fully inside the row (pred) or fully outside.
template <typename T, int K>
__device__ __forceinline__ RawK<T, K> load_vec(const T* __restrict__ p, bool pred) {
    constexpr int BYTES = K * (int)sizeof(T);
    constexpr int VB = BYTES >= 16 ? 16 : BYTES;
    constexpr int NV = BYTES / VB;
    using V = typename Pack<T, VB>::type;
    union { V raw[NV]; RawK<T, K> r; } u;
#pragma unroll
    for (int i = 0; i < NV; ++i) u.raw[i] = V(0);
    if (pred) {
#pragma unroll
        for (int i = 0; i < NV; ++i) u.raw[i] = reinterpret_cast<const V*>(p)[i];
    }
    return u.r;
}
// Same, but the load is ALWAYS issued (from `safe` when !pred) so that the compiler can count it in
// s_waitcnt vmcnt(N): a load under a divergent branch makes every later wait a full vmcnt(0) drain.
template <typename T, int K>
__device__ __forceinline__ RawK<T, K> load_vec_always(const T* __restrict__ p, bool pred, const T* __restrict__ safe) {
    constexpr int BYTES = K * (int)sizeof(T);
    constexpr int VB = BYTES >= 16 ? 16 : BYTES;
    constexpr int NV = BYTES / VB;
    using V = typename Pack<T, VB>::type;
    union { V raw[NV]; RawK<T, K> r; } u;
    const V* q = reinterpret_cast<const V*>(pred ? p : safe);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const V v = q[i];
        u.raw[i] = pred ? v : V(0);
    }
    return u.r;
}

template <typename T, int K>
__device__ __forceinline__ void store_vec(T* __restrict__ p, bool pred, const float (&v)[K]) {
    constexpr int BYTES = K * (int)sizeof(T);
    constexpr int VB = BYTES >= 16 ? 16 : BYTES;
    constexpr int NV = BYTES / VB;
    using V = typename Pack<T, VB>::type;
    union { V raw[NV]; T e[K]; } u;
#pragma unroll
    for (int k = 0; k < K; ++k) u.e[k] = from_f32<T>(v[k]);
    if (pred) {
#pragma unroll
        for (int i = 0; i < NV; ++i) reinterpret_cast<V*>(p)[i] = u.raw[i];
    }
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the
// wave's outstanding global loads/stores (hipcc emits s_waitcnt vmcnt(0) for those).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- in-kernel stamps: DIAGNOSTIC builds only (tools/scan_lab.hip defines VIVIM_STAMPS) ----
#ifdef VIVIM_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;   // [block][wave][step][slot]
constexpr int kStampSlots = 16, kStampSteps = 8, kStampWaves = 8;
__device__ __forceinline__ void stamp(int step, int slot, int wave, int lane) {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (g_stamp_buf && lane == 0 && step < kStampSteps && blockIdx.x < 2 && blockIdx.y == 0 && blockIdx.z == 0)
        g_stamp_buf[((blockIdx.x * kStampWaves + wave) * kStampSteps + step) * kStampSlots + slot] = t;
}
#define VIVIM_STAMP(step, slot, wave, lane) stamp(step, slot, wave, lane)
#else
#define VIVIM_STAMP(step, slot, wave, lane) ((void)0)
#endif

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

// ---- DPP forms (no LDS crossbar): row_shr 1/2/4/8 inside 16-lane rows, then row_bcast 15 / 31 ----
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float read_lane(float v, int lane) {   // wave-uniform result (SGPR)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
constexpr int kDppWaveShr1 = 0x138;   // lane l <- lane l-1 (lane 0 keeps `old`)
constexpr int kDppWaveShl1 = 0x130;   // lane l <- lane l+1 (lane 63 keeps `old`)

// Two independent forward scans of affine maps, interleaved so that every DPP read of a VGPR is at
// least two instructions behind its last VALU write (the gfx9 DPP hazard the compiler cannot see
// inside asm); the leading s_nop covers the compiler-generated producer.  Lanes without a source
// (row start, masked rows) are not written, i.e. they compose with the identity.  Needs EXEC = all ones.
#define VIVIM_SCAN2_STEP(ctrl)                                   \
    "v_fmac_f32_dpp %0, %0, %2 " ctrl " bank_mask:0xf\n\t"      \
    "v_fmac_f32_dpp %1, %1, %3 " ctrl " bank_mask:0xf\n\t"      \
    "v_mul_f32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n\t"       \
    "v_mul_f32_dpp %3, %3, %3 " ctrl " bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_scan2_affine_fwd(float& P0, float& H0, float& P1, float& H1) {
    asm volatile("s_nop 1\n\t"
                 VIVIM_SCAN2_STEP("row_shr:1 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_shr:2 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_shr:4 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_shr:8 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_bcast:15 row_mask:0xa")
                 VIVIM_SCAN2_STEP("row_bcast:31 row_mask:0xc")
                 "s_nop 1"
                 : "+v"(H0), "+v"(H1), "+v"(P0), "+v"(P1));
}

// Reverse direction: in-row part with row_shl (lane l combines with lanes l+1.. of its 16-lane row);
// rows are joined afterwards by wave_scan2_rev_join (there is no reverse row_bcast).
#define VIVIM_SCAN2_STEP_L(ctrl) VIVIM_SCAN2_STEP(ctrl)
__device__ __forceinline__ void wave_scan2_affine_rev_rows(float& P0, float& H0, float& P1, float& H1) {
    asm volatile("s_nop 1\n\t"
                 VIVIM_SCAN2_STEP("row_shl:1 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_shl:2 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_shl:4 row_mask:0xf")
                 VIVIM_SCAN2_STEP("row_shl:8 row_mask:0xf")
                 "s_nop 1"
                 : "+v"(H0), "+v"(H1), "+v"(P0), "+v"(P1));
}
// After the in-row scan lane 16r holds row r's total map.  Compose, for every lane, the maps of the rows
// to its right (rows r+1..3) and apply its own after them: lane l ends with the composition of lanes l..63.
__device__ __forceinline__ void wave_scan_rev_join(float& P, float& H, int lane) {
    const float p1 = read_lane(P, 16), h1 = read_lane(H, 16);
    const float p2 = read_lane(P, 32), h2 = read_lane(H, 32);
    const float p3 = read_lane(P, 48), h3 = read_lane(H, 48);
    const float q2p = p2 * p3, q2h = fmaf(p2, h3, h2);          // rows 2..3
    const float q1p = p1 * q2p, q1h = fmaf(p1, q2h, h1);        // rows 1..3
    const int row = lane >> 4;
    const float sp = row == 0 ? q1p : row == 1 ? q2p : row == 2 ? p3 : 1.0f;
    const float sh = row == 0 ? q1h : row == 1 ? q2h : row == 2 ? h3 : 0.0f;
    H = fmaf(P, sh, H);
    P = P * sp;
}

// wave64 sum with DPP adds; the total lands in lane 63 (read it with read_lane(v, 63)).
__device__ __forceinline__ float wave_sum_dpp_to63(float v) {
    v += dpp_mov<0x111>(0.0f, v);
    v += dpp_mov<0x112>(0.0f, v);
    v += dpp_mov<0x114>(0.0f, v);
    v += dpp_mov<0x118>(0.0f, v);
    v += dpp_mov<0x142, 0xa>(0.0f, v);
    v += dpp_mov<0x143, 0xc>(0.0f, v);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// compiler-level ordering of this wave's LDS traffic (hardware keeps a wave's DS ops in order)
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

}  // namespace vivim
