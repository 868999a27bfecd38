// ls_common.cuh -- helpers shared by the lanes = states kernels (scan_ls.hip, scan_ls2.hip): the 16-lane row
// primitives (DPP broadcasts, transposed reductions), raw buffer accessors, B / C row loads and the token-axis
// segment descriptor.  Moved verbatim out of scan_ls.hip.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.cuh"

namespace vivim {

constexpr int kLsT = 16;           // tokens per tile
constexpr int kLsCPR = 4;          // channels a row walks per tile

__host__ __device__ constexpr int br4(int k) { return ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3); }

// value held by lane K of this lane's 16-lane row
template <int K> __device__ __forceinline__ float row_bc(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, true));
}
// value of token K of the tile (token K lives in lane bitrev4(K) of every row)
template <int K> __device__ __forceinline__ float tok(float v) { return row_bc<br4(K)>(v); }

// acc + token_K(src) * mul as ONE instruction (v_fmac_f32 with a DPP row broadcast on src).  hipcc folds the broadcast into
// v_mul_f32 by itself but not into v_fmac_f32 (it emits v_mov_b32_dpp + v_fmac_f32).  Written as a dependent update of
// `acc` on purpose: a product that does not depend on the running value gets hoisted out of the 16-token recurrence by
// the scheduler, 16 temporaries at a time.
template <int K> __device__ __forceinline__ float tok_fma(float acc, float src, float mul) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(br4(K)));
    return acc;
}

// The per-token values (delta, delta * u, dy) are read through DPP by inline asm (tok_fma), where hipcc does not see the
// "VALU write -> DPP read needs two wait states" hazard: pass them through this once, right after they are computed --
// from here on they are only read.  (Without it the broadcasts of the LAST channel of a tile returned stale registers.)
__device__ __forceinline__ void ls_settle(float& a, float& b, float& c) { asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void ls_settle(float& a, float& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
// "These sixteen registers are used HERE": placed right after the loads of a tile's B / C rows, it makes hipcc wait for them
// before the channel loop instead of at their first use inside it (where it cannot count what else is in flight).
__device__ __forceinline__ void ls_arrive(float (&v)[16]) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                      "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int I, typename F> __device__ __forceinline__ void sfor_down(F&& f) {      // I-1, I-2, ..., 0
    if constexpr (I > 0) { f(std::integral_constant<int, I - 1>{}); sfor_down<I - 1>(f); }
}

// ---- transposed 16-lane reduction: merge two vectors, each lane keeps the pair sum of ONE of them --------------------
// level 8: lanes 0-7 of a row get X[r] + X[r+8], lanes 8-15 get Y[r-8] + Y[r].  The leading s_nop covers the DPP read
// hazard against the (compiler-scheduled) VALU producers of X / Y.  EXEC must be all ones (it is: uniform code).
__device__ __forceinline__ float ls_merge8(float X, float Y) {
    float Z;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(Z) : "v"(X), "v"(Y));
    return Z;
}
// level 4: lanes with bit 2 clear get X[r] + X[r+4], lanes with bit 2 set get Y[r-4] + Y[r]
__device__ __forceinline__ float ls_merge4(float X, float Y) {
    float Z;
    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(Z) : "v"(X), "v"(Y));
    return Z;
}
// levels 2 and 1: no DPP write mask is finer than a bank of four lanes, and an EXEC mask would also invalidate the SOURCE
// lanes of the DPP read, so each lane first selects what it keeps and what it sends (v_cndmask issues beside the
// plain VALU work, valu_lab3: pattern CP), then one DPP add exchanges inside the quad.
// level 2: lanes 0,1 of a quad get X[r] + X[r+2], lanes 2,3 get Y[r-2] + Y[r]
__device__ __forceinline__ float ls_merge2(float X, float Y, bool hi) {
    const float keep = hi ? Y : X, send = hi ? X : Y;
    return keep + dpp_mov<0x4e>(0.0f, send);            // quad_perm:[2,3,0,1]
}
// level 1: even lanes get X[r] + X[r+1], odd lanes get Y[r-1] + Y[r]
__device__ __forceinline__ float ls_merge1(float X, float Y, bool hi) {
    const float keep = hi ? Y : X, send = hi ? X : Y;
    return keep + dpp_mov<0xb1>(0.0f, send);            // quad_perm:[1,0,3,2]
}
// Incremental use: feed the 16 per-token vectors in an order that completes pairs (2j, 2j+1) -- an ascending or a
// descending loop -- so that at most five partial vectors are live.  Result: lane r holds the total of token
// bitrev4(r), i.e. of the token that lane owns.
// merges for a DESCENDING token loop: call after token K has been produced
template <int K> __device__ __forceinline__ void ls_reduce_down(float (&s)[16], float (&z)[8], float (&w)[4], float (&v)[2], float& out, int li) {
    if constexpr ((K & 1) == 0) z[K / 2] = ls_merge8(s[K], s[K + 1]);
    if constexpr ((K & 3) == 0) w[K / 4] = ls_merge4(z[K / 2], z[K / 2 + 1]);
    if constexpr ((K & 7) == 0) v[K / 8] = ls_merge2(w[K / 4], w[K / 4 + 1], (li & 2) != 0);
    if constexpr (K == 0) out = ls_merge1(v[0], v[1], (li & 1) != 0);
}
// merges for an ASCENDING token loop
template <int K> __device__ __forceinline__ void ls_reduce_up(float (&s)[16], float (&z)[8], float (&w)[4], float (&v)[2], float& out, int li) {
    if constexpr ((K & 1) == 1) z[K / 2] = ls_merge8(s[K - 1], s[K]);
    if constexpr ((K & 3) == 3) w[K / 4] = ls_merge4(z[K / 2 - 1], z[K / 2]);
    if constexpr ((K & 7) == 7) v[K / 8] = ls_merge2(w[K / 4 - 1], w[K / 4], (li & 2) != 0);
    if constexpr (K == 15) out = ls_merge1(v[0], v[1], (li & 1) != 0);
}

// sum over the RPS rows of a stream (dstate 32 / 64): every row ends with the total
template <int RPS> __device__ __forceinline__ float ls_rows_sum(float x) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    if constexpr (RPS >= 2) {
        const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    if constexpr (RPS >= 4) {
        const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        x = __uint_as_float(r.x) + __uint_as_float(r.y);
    }
    return x;
}
// sum over the 16 lanes of a row, total in every lane
__device__ __forceinline__ float ls_row_total(float v) {
    v += dpp_mov<0x128>(0.0f, v);       // row_ror:8
    v += dpp_mov<0x124>(0.0f, v);       // row_ror:4
    v += dpp_mov<0x122>(0.0f, v);       // row_ror:2
    v += dpp_mov<0x121>(0.0f, v);       // row_ror:1
    return v;
}

// ---- activation I/O: raw buffer accesses ------------------------------------------------------------------------------------
// Every lane touches ONE element per tensor and channel; with plain pointers that is a 64-bit address per (tensor,
// channel) and lane, and hipcc keeps all of them live across the tile loop (the first build of the backward: 300 VGPRs
// of spills).  A buffer access splits the address into the resource's base (SGPRs: tensor + batch offset), a scalar
// offset (SGPR: the wave's channel) and ONE 32-bit lane offset per tensor (the row's channel offset + the token).
// The host checks that a (channel, token) offset inside one batch element fits 32 bits (ls_offsets_ok).
// Kernel arguments arrive as s_load_dwordx16 tuples; when SGPRs run short hipcc spills whole tuples to VGPR lanes and
// reloads all sixteen (v_readlane) for every field it touches inside the loops.  Values that are used in the loops are
// therefore copied out once, through a VGPR and v_readfirstlane, into scalars of their own.
__device__ __forceinline__ unsigned ls_own(unsigned v) {
    asm volatile("" : "+v"(v));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int ls_own(int v) { return (int)ls_own((unsigned)v); }
template <typename P> __device__ __forceinline__ P* ls_own(P* q) {
    const uint64_t a = reinterpret_cast<uint64_t>(q);
    return reinterpret_cast<P*>(((uint64_t)ls_own((unsigned)(a >> 32)) << 32) | ls_own((unsigned)a));
}
typedef __amdgpu_buffer_rsrc_t ls_rsrc;
__device__ __forceinline__ ls_rsrc ls_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, -1, 0x00020000);
}
template <typename T> struct LsElem;
template <> struct LsElem<float> {
    static __device__ __forceinline__ float ld(ls_rsrc r, unsigned voff, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    }
    static __device__ __forceinline__ void st(ls_rsrc r, unsigned voff, unsigned soff, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
    }
};
template <> struct LsElem<bf16_t> {
    static __device__ __forceinline__ float ld(ls_rsrc r, unsigned voff, unsigned soff) {
        return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0) << 16);
    }
    static __device__ __forceinline__ void st(ls_rsrc r, unsigned voff, unsigned soff, float v) {
        union { bf16_t h; unsigned short u; } c;
        c.h = from_f32<bf16_t>(v);
        __builtin_amdgcn_raw_buffer_store_b16(c.u, r, voff, soff, 0);
    }
};
template <> struct LsElem<f16_t> {
    static __device__ __forceinline__ float ld(ls_rsrc r, unsigned voff, unsigned soff) {
        union { unsigned short u; f16_t h; } c;
        c.u = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
        return static_cast<float>(c.h);
    }
    static __device__ __forceinline__ void st(ls_rsrc r, unsigned voff, unsigned soff, float v) {
        union { f16_t h; unsigned short u; } c;
        c.h = from_f32<f16_t>(v);
        __builtin_amdgcn_raw_buffer_store_b16(c.u, r, voff, soff, 0);
    }
};
// One tensor of shape (batch, dim, seqlen): resource of this batch element, channel stride in bytes, and the lane's own
// offset (its row's first channel + its token of the current tile).  Loads are unconditional (a load under a divergent
// branch turns every later wait into a full drain): a lane that is off reads the element at offset 0 and drops it.
// The kernel arguments, re-read through a pointer the compiler cannot see through (scalar loads out of the constant
// cache, merged per step): a tensor's base / strides are then live only around its access instead of sitting in
// SGPRs for the whole kernel -- twelve tensors do not fit, and hipcc spilled them to VGPR lanes (v_readlane for every
// access: 525 per tile in the first build).
typedef const __attribute__((address_space(4))) char* ls_kargs;
__device__ __forceinline__ ls_kargs ls_fresh_kargs() {
    ls_kargs q = (ls_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}
template <typename V> __device__ __forceinline__ V ls_karg(ls_kargs q, int off) {
    return *reinterpret_cast<const __attribute__((address_space(4))) V*>(q + off);
}
#define LS_OFF(S, fld) ((int)__builtin_offsetof(S, fld))

template <typename T> struct LsTensor {
    int off_ptr, off_bs;           // byte offsets of the pointer and of {batch stride, channel stride} in the kernel arguments
    int b;                         // batch element
    unsigned lane_off;             // bytes: (this row's first channel - the wave's first channel) * channel stride
    __device__ __forceinline__ void init(int off_ptr_, int off_bs_, int b_, int64_t d_stride, int row_ch) {
        off_ptr = off_ptr_; off_bs = off_bs_; b = b_;
        lane_off = (unsigned)row_ch * (unsigned)d_stride * (unsigned)sizeof(T);
    }
    struct Acc { ls_rsrc r; unsigned soff; };
    // chu: wave-uniform channel (the wave's first channel + c, clamped to a valid one).  The resource's size field carries
    // chu so that it is rebuilt (scalar moves) per access instead of being kept.
    __device__ __forceinline__ Acc acc(ls_kargs q, int chu) const {
        const T* ptr = ls_karg<const T*>(q, off_ptr);
        const int64_t bs = ls_karg<int64_t>(q, off_bs), ds = ls_karg<int64_t>(q, off_bs + 8);
        Acc a;
        a.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ptr + (int64_t)b * bs), (short)0, (int)(0xffff0000u + (unsigned)chu), 0x00020000);
        a.soff = (unsigned)chu * ((unsigned)ds * (unsigned)sizeof(T));
        return a;
    }
    // t: this lane's token
    __device__ __forceinline__ float ld(ls_kargs q, int chu, int t, bool ok) const { const float v = ld_raw(q, chu, t, ok); return ok ? v : 0.0f; }
    __device__ __forceinline__ float ld_raw(ls_kargs q, int chu, int t, bool ok) const {
        const Acc a = acc(q, chu);
        return LsElem<T>::ld(a.r, ok ? lane_off + (unsigned)t * (unsigned)sizeof(T) : 0u, a.soff);
    }
    // Stores are issued by every lane: a store under a divergent branch makes hipcc's s_waitcnt counting give up (every
    // later wait becomes vmcnt(0), which also waits for the NEXT step's prefetched loads).  A lane that is off gets an
    // offset beyond the resource's size: the buffer range check (voffset >= num_records) drops its store.
    __device__ __forceinline__ void st(ls_kargs q, int chu, int t, bool ok, float v) const {
        const Acc a = acc(q, chu);
        LsElem<T>::st(a.r, ok ? lane_off + (unsigned)t * (unsigned)sizeof(T) : 0xfffffff0u, a.soff, v);
    }
};
// The same with base and stride held in SGPRs of their own for the whole kernel: for the few tensors that are LOADED at
// the top of every step, where the latency of the scalar loads above would be exposed.
template <typename T> struct LsTensorR {
    const T* base;                 // this batch element's (channel 0, token 0), wave-uniform
    unsigned dstride;              // bytes between channels
    unsigned lane_off;
    __device__ __forceinline__ void init(const void* p, int64_t batch_off, int64_t d_stride, int row_ch) {
        base = ls_own(static_cast<const T*>(p) + batch_off);
        dstride = ls_own((unsigned)d_stride * (unsigned)sizeof(T));
        lane_off = (unsigned)row_ch * dstride;
    }
    __device__ __forceinline__ float ld(int chu, int t, bool ok) const { const float v = ld_raw(chu, t, ok); return ok ? v : 0.0f; }
    // without the final select: for values that are requested a step ahead -- the select would be scheduled right behind
    // the load and wait for it; the consumer applies `ok` when it uses the value
    __device__ __forceinline__ float ld_raw(int chu, int t, bool ok) const {
        const ls_rsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), (short)0, (int)(0xffff0000u + (unsigned)chu), 0x00020000);
        return LsElem<T>::ld(r, ok ? lane_off + (unsigned)t * (unsigned)sizeof(T) : 0u, (unsigned)chu * dstride);
    }
};
// the checkpoint tensor x (batch, dim, nck, dstate) f32: "token" = (row, state), channel stride = nck * dstate floats
struct LsCkpt {
    int off_ptr;
    unsigned bstride_ch;           // channels per batch element (dim)
    unsigned dstride;              // bytes between channels
    int b;
    unsigned lane_off;
    __device__ __forceinline__ void init(int off_ptr_, int b_, int dim, int nck, int NS, int row_ch) {
        off_ptr = off_ptr_; b = b_; bstride_ch = (unsigned)dim;
        dstride = (unsigned)nck * (unsigned)NS * 4u;
        lane_off = (unsigned)row_ch * dstride;
    }
    __device__ __forceinline__ LsTensor<float>::Acc acc(ls_kargs q, int chu) const {
        const float* ptr = ls_karg<const float*>(q, off_ptr);
        LsTensor<float>::Acc a;
        a.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr) + (int64_t)b * bstride_ch * (dstride / 4u), (short)0,
                                                (int)(0xffff0000u + (unsigned)chu), 0x00020000);
        a.soff = (unsigned)chu * dstride;
        return a;
    }
    __device__ __forceinline__ float ld(ls_kargs q, int chu, int idx, bool ok) const { const float v = ld_raw(q, chu, idx, ok); return ok ? v : 0.0f; }
    __device__ __forceinline__ float ld_raw(ls_kargs q, int chu, int idx, bool ok) const {
        const auto a = acc(q, chu);
        return LsElem<float>::ld(a.r, ok ? lane_off + (unsigned)idx * 4u : 0u, a.soff);
    }
    __device__ __forceinline__ void st(ls_kargs q, int chu, int idx, bool ok, float v) const {
        const auto a = acc(q, chu);
        LsElem<float>::st(a.r, ok ? lane_off + (unsigned)idx * 4u : 0xfffffff0u, a.soff, v);
    }
};

// 16 consecutive tokens of one row of B or C (this lane's state): resource of the (batch, group) block, lane offset =
// the state's row.  vec: rows and t0 are 16-byte aligned and the tile is whole; otherwise element loads with a tail predicate.
// sixteen elements out of 16-byte vectors, without a union (hipcc left the union of some instantiations on the stack)
template <typename T> struct LsUnpack;
template <> struct LsUnpack<float> {
    static __device__ __forceinline__ void run(const u32x4 (&raw)[4], float (&v)[16]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i * 4 + 0] = __uint_as_float(raw[i].x); v[i * 4 + 1] = __uint_as_float(raw[i].y);
            v[i * 4 + 2] = __uint_as_float(raw[i].z); v[i * 4 + 3] = __uint_as_float(raw[i].w);
        }
    }
};
template <> struct LsUnpack<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4 (&raw)[2], float (&v)[16]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[i * 8 + q * 2 + 0] = __uint_as_float(w[q] << 16);
                v[i * 8 + q * 2 + 1] = __uint_as_float(w[q] & 0xffff0000u);
            }
        }
    }
};
template <> struct LsUnpack<f16_t> {
    static __device__ __forceinline__ void run(const u32x4 (&raw)[2], float (&v)[16]) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned w[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const h2 hh = __builtin_bit_cast(h2, w[q]);
                v[i * 8 + q * 2 + 0] = static_cast<float>(hh.x);
                v[i * 8 + q * 2 + 1] = static_cast<float>(hh.y);
            }
        }
    }
};

template <typename T> struct LsRow {
    int off_ptr, off_bs;           // kernel-argument offsets of the pointer and of {batch stride, group stride, dstate stride}
    int b, g;
    unsigned lane_off;             // bytes: state * dstate stride
    __device__ __forceinline__ void init(int off_ptr_, int off_bs_, int b_, int g_, int64_t n_stride, int n) {
        off_ptr = off_ptr_; off_bs = off_bs_; b = b_; g = g_;
        lane_off = (unsigned)n * (unsigned)n_stride * (unsigned)sizeof(T);
    }
    // one 16-byte piece of a whole, aligned tile: row `pn`, piece `pp` of the row's sizeof(T) pieces
    __device__ __forceinline__ u32x4 piece(ls_kargs q, int t0, unsigned pn, unsigned pp) const {
        const T* ptr = ls_karg<const T*>(q, off_ptr);
        const int64_t bs = ls_karg<int64_t>(q, off_bs), gs = ls_karg<int64_t>(q, off_bs + 8), ns = ls_karg<int64_t>(q, off_bs + 16);
        const ls_rsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ptr + (int64_t)b * bs + (int64_t)g * gs), (short)0,
                                                            (int)(0xffff0000u + (unsigned)(t0 & 0xfff0)), 0x00020000);
        return __builtin_amdgcn_raw_buffer_load_b128(r, pn * ((unsigned)ns * (unsigned)sizeof(T)) + pp * 16u, (unsigned)t0 * (unsigned)sizeof(T), 0);
    }
    __device__ __forceinline__ void load16(ls_kargs q, int t0, int L, bool vec, float (&v)[16]) const {
        const T* ptr = ls_karg<const T*>(q, off_ptr);
        const int64_t bs = ls_karg<int64_t>(q, off_bs), gs = ls_karg<int64_t>(q, off_bs + 8);
        const ls_rsrc r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ptr + (int64_t)b * bs + (int64_t)g * gs), (short)0,
                                                            (int)(0xffff0000u + (unsigned)(t0 & 0xfff0)), 0x00020000);
        if (vec) {
            constexpr int NV = (int)sizeof(T);          // 16-byte vectors per 16 elements
            u32x4 raw[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i)
                raw[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off + (unsigned)i * 16u, (unsigned)t0 * (unsigned)sizeof(T), 0);
            LsUnpack<T>::run(raw, v);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool ok = t0 + k < L;
                const float x = LsElem<T>::ld(r, lane_off + (ok ? (unsigned)k * (unsigned)sizeof(T) : 0u), (ok ? (unsigned)t0 : 0u) * (unsigned)sizeof(T));
                v[k] = ok ? x : 0.0f;
            }
        }
    }
};

// ---- geometry shared by the kernels and the host -----------------------------------------------------------------------
struct LsSeg {
    int S, seg_blocks;             // segments, checkpoint blocks (16 * RPS tokens) per segment
    float* agg;                    // [batch][dim][S][dstate]   pre-pass: the segment's aggregate for zero inflow
    float* dsum;                   // [batch][dim][S]           pre-pass: the segment's sum of delta (see the kernels)
    float* gin;                    // [batch][dim][S][dstate]   carry kernel: inflow of segment s
    int bc_vec;                    // B / C rows may be read with 16-byte vectors
    int dbg;                       // DIAGNOSTIC (VIVIM_LS_DBG): 1 skips the backward's tile epilogue, 2 its B / C loads -- wrong results
};

template <int NS> struct LsGeom {
    static constexpr int RPS = NS / 16;                 // rows per stream
    static constexpr int SPW = 4 / RPS;                 // streams per wave
    static constexpr int CPW = SPW * kLsCPR;            // channels per wave
    static constexpr int CK = 16 * RPS;                 // tokens per checkpoint row of x
};

// The wave's first channel, re-read through a value the compiler cannot see through: the scalar offsets of the four
// channels of a tile (8 tensors x 4 channels) would otherwise all be hoisted out of the tile loop and spill.
__device__ __forceinline__ int ls_fresh_uniform(int& v) {
    asm volatile("" : "+v"(v));
    return __builtin_amdgcn_readfirstlane(v);
}
// ... and the same value made to depend on `result`: the next channel's loads (whose scalar offsets come from it) cannot
// be issued before `result` exists, so hipcc cannot sink the recurrences of all four channels of a tile below the four
// prologues and run them interleaved (it did: 4 x the live registers, 800 bytes of spills per lane).
__device__ __forceinline__ void ls_tie(int& v, float result) { asm volatile("" : "+v"(v) : "v"(result)); }

}  // namespace vivim
