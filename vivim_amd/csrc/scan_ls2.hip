// scan_ls2.hip -- selective SSM scan backward, "lanes = states" main kernel, second generation (gfx950, wave64).
//
// Same math and same lane mapping as ssm_ls_bwd_kernel (scan_ls.hip; reference: selective_scan_bwd_kernel.cuh:146-489):
// a 16-lane row is one channel, the lane is the state, a tile is 16 tokens, a row walks kLsCPR channels per tile and keeps
// their dB / dC sum in registers; token-axis segments get their inflow from a pre-pass + carry.  What changed is everything
// AROUND the two sweeps, following what round 2 measured and what this round's stamps said:
//   * activations move in SPANS of one 128-byte line per channel row (64 16-bit / 32 fp32 tokens) and 16 channels per wave:
//     u, delta, dout, z, out are read with 16-byte vectors, whole lines, once per span; du / ddelta leave the same way.
//     The first-generation kernel asked for ONE element per lane and tensor in every (tile, channel) step: 32-byte row
//     pieces whose lines were evicted between visits (3.8 x the algorithmic traffic; here 1.7 x), ten vector-memory
//     instructions and ~150 scalar instructions of descriptor arithmetic per step.
//   * everything a token needs that does not depend on the scan runs at SPAN level, where a lane holds 8 (4) consecutive
//     tokens of one channel and their dependency chains interleave: softplus, the z gate, dz (stored right away), dD, and at
//     the span's end the sigmoid factor of ddelta and dbias.  A (tile, channel) step is then the forward sweep, the reverse
//     sweep with its two transposed reductions, and four instructions of output arithmetic.
//   * the per-token scalars (delta, delta * u, dy; f32) live in LDS in token order and reach the 16 lanes of a row as
//     broadcast ds_read_b128 of four tokens -- no DPP operands in the sweeps (a VALU instruction with a DPP operand issues at
//     half rate and is not hidden by its neighbours, profiles/r02_valu_lab3.log; the first generation carried six per update).
//   * the transposed reductions are written with builtins (v_cndmask pair + one DPP add per merge; level 4 pairs lane l with
//     l ^ 7 = row_half_mirror so that every level is an involution): hipcc sees the DPP hazards itself -- no s_nop.
//   * what a channel carries from tile to tile (reverse carry, dA, A log2e, the next checkpoint) sits in registers that
//     rotate through slot 0; the checkpoints of a tile are requested a tile ahead and taken out of the memory queue before
//     that tile's dB / dC atomics go in: the step loop contains no vector-memory wait.
//   * dB / dC: the four rows of a wave are summed in registers (v_permlane16/32_swap merges) before they go to LDS.
// Per wave 18 KB (16-bit) / 12 KB (fp32) of LDS and 232-251 VGPRs: two 4-wave workgroups per CU, two waves per SIMD.
//
// Measured (MI355X, tools/kbench.py grouped shapes; profiles/r03_*): the same time as the first generation on 16-bit rows
// (cfg 2 stages 1-3: 274 / 164 / 89 us against 274 / 174 / 84) and 16 % less on long fp32 rows; with the lanes = tokens
// closed-form pre-pass in front (scan_ls.hip: segments cut at multiples of 256 tokens) cfg 3 grouped stage 0 takes 5480 us
// against 5760 for the lanes = tokens family -- whose forward writes 16 x fewer checkpoints, which is why the automatic
// dispatch keeps it for rows longer than 8192 tokens.  SQ counters (profiles/r03_pmc_sq_ls2_*): 26 VALU instructions per
// state update (12 the recurrences, 5.6 the two reductions, the rest span level, dB / dC, bookkeeping), the vector ALU busy
// ~80 % of the time at two waves per SIMD.  What did NOT help, each built, checked and timed (profiles/r03_ls2_experiments.log):
// two channels of a row side by side in one step for instruction-level parallelism (+5 %), B / C rows read from the
// staged LDS copy instead of 32 registers (+6 %), the reductions as one block behind the sweep (+1 %), 8-wave workgroups
// with plain dB / dC stores (+5 ... +18 %), three waves per SIMD (32-token spans, B / C from LDS, 168 VGPRs with spills outside the
// step loop; tools/experiments/ls2_w3.patch: +8 ... +20 % on 16-bit rows, +1 ... +2 % on fp32).  Timing
// ablations (LS2_ABL): no LDS reads in the sweeps -11 %, no reductions -15 %, no forward rebuild -14 %, no epilogue -3 %:
// the cost is spread over the instruction stream, which is what "VALU-bound" looks like from outside.
#include "ls_common.cuh"

namespace vivim {

template <typename T> struct Ls2Geom {
    static constexpr int EPV = 16 / (int)sizeof(T);      // elements per 16-byte piece
    static constexpr int TS = 128 / (int)sizeof(T);      // tokens per span (one line per channel row)
    static constexpr int TPS = TS / kLsT;                // tiles per span
    // per-wave LDS block, bytes
    static constexpr int RAWU = 0, RAWD = 2048, DL = 4096;
    static constexpr int WU = DL + TS * 16 * 4, DY = WU + TS * 16 * 4;
    static constexpr int SLOT = DY + TS * 16 * 4;       // [2 planes][lane][4] f32: the wave's dB / dC sums of a tile
    static constexpr int WB = SLOT + 2 * kWave * 16;
};

// ---- 16-byte pieces <-> floats ---------------------------------------------------------------------------------------
template <typename T> struct Ls2Piece;
template <> struct Ls2Piece<float> {
    static __device__ __forceinline__ void unpack(const u32x4 r, float (&v)[4]) {
        v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[4]) {
        return u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    }
};
template <> struct Ls2Piece<bf16_t> {
    static __device__ __forceinline__ void unpack(const u32x4 r, float (&v)[8]) {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            union { bf16_t h[2]; unsigned u; } c;
            c.h[0] = from_f32<bf16_t>(v[2 * q]); c.h[1] = from_f32<bf16_t>(v[2 * q + 1]);
            w[q] = c.u;
        }
        return u32x4{w[0], w[1], w[2], w[3]};
    }
};
template <> struct Ls2Piece<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void unpack(const u32x4 r, float (&v)[8]) {
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const h2 hh = __builtin_bit_cast(h2, w[q]);
            v[2 * q] = static_cast<float>(hh.x); v[2 * q + 1] = static_cast<float>(hh.y);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h2 hh; hh.x = static_cast<_Float16>(v[2 * q]); hh.y = static_cast<_Float16>(v[2 * q + 1]);
            w[q] = __builtin_bit_cast(unsigned, hh);
        }
        return u32x4{w[0], w[1], w[2], w[3]};
    }
};

// ---- transposed 16-lane reduction, compiler-visible form: each lane keeps one of the two vectors and receives the
// other one's partner lane through ONE DPP add (the selects pair with plain VALU work, valu_lab3 pattern CP) -------------
template <int CTRL> __device__ __forceinline__ float ls2_xchg_add(float keep, float send) {
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float ls2_merge8(float X, float Y, bool hi) { return ls2_xchg_add<0x128>(hi ? Y : X, hi ? X : Y); }   // row_ror:8
// level 4 pairs lane l with l ^ 7 (row_half_mirror, an involution like the others; a rotation by 4 is not): the partners differ
// in bit 2, which is all the merge needs, and the four levels {8, 7, 2, 1} still generate every lane of the row
__device__ __forceinline__ float ls2_merge4(float X, float Y, bool hi) { return ls2_xchg_add<0x141>(hi ? Y : X, hi ? X : Y); }
__device__ __forceinline__ float ls2_merge2(float X, float Y, bool hi) { return ls2_xchg_add<0x4e>(hi ? Y : X, hi ? X : Y); }    // quad_perm:[2,3,0,1]
__device__ __forceinline__ float ls2_merge1(float X, float Y, bool hi) { return ls2_xchg_add<0xb1>(hi ? Y : X, hi ? X : Y); }    // quad_perm:[1,0,3,2]
// merges for a DESCENDING token loop: call after token K has been produced.  Lane r ends with the total of token bitrev4(r).
template <int K> __device__ __forceinline__ void ls2_reduce_down(float (&s)[16], float (&z)[8], float (&w)[4], float (&v)[2], float& out, int li) {
    if constexpr ((K & 1) == 0) z[K / 2] = ls2_merge8(s[K], s[K + 1], (li & 8) != 0);
    if constexpr ((K & 3) == 0) w[K / 4] = ls2_merge4(z[K / 2], z[K / 2 + 1], (li & 4) != 0);
    if constexpr ((K & 7) == 0) v[K / 8] = ls2_merge2(w[K / 4], w[K / 4 + 1], (li & 2) != 0);
    if constexpr (K == 0) out = ls2_merge1(v[0], v[1], (li & 1) != 0);
}

// sum of two vectors over row pairs: permlane16_swap exchanges the odd rows of P with the even rows of Q, so P' + Q' holds
// rows {P.r0 + P.r1, Q.r0 + Q.r1, P.r2 + P.r3, Q.r2 + Q.r3}
__device__ __forceinline__ float ls2_rows16(float P, float Q) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(P), __float_as_uint(Q), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// permlane32_swap exchanges the upper half of P with the lower half of Q: P' + Q' = {P.lo + P.hi, Q.lo + Q.hi}
__device__ __forceinline__ float ls2_rows32(float P, float Q) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(P), __float_as_uint(Q), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// keeps hipcc from hoisting the LDS reads of later token groups above this point (it clusters all twelve of a sweep
// otherwise: 48 more live registers, spilled)
__device__ __forceinline__ void ls2_sched_fence() { asm volatile("" ::: "memory"); }

#ifndef LS2_ABL
#define LS2_ABL 0
#endif
constexpr int kAbl2 = LS2_ABL;       // timing experiments only (tools/abl.sh ls2build): results are WRONG for any value but 0

// One (tile, channel) step is the forward sweep, the reverse sweep and four instructions of output arithmetic: everything a
// token needs that does not depend on the scan (softplus, the z gate, dz, the sigmoid factor of ddelta) is done at SPAN
// level, where a lane holds EPV consecutive tokens of one channel and their dependency chains interleave -- in-kernel
// stamps of the first build had a step spend 1000 + 600 cycles of pure latency in these chains against 2700 in the sweeps.
template <typename T, bool HAS_Z>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) ssm_ls2_bwd_kernel(const vivim_ssm_bwd_params p, const LsSeg sg) {
    typedef Ls2Geom<T> G2;
    typedef Ls2Piece<T> PK;
    constexpr int EPV = G2::EPV, TS = G2::TS, TPS = G2::TPS, NS = 16, CPR = kLsCPR;
    const vivim_ssm_fwd_params& f = p.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = blockDim.x >> 6;
    const int row = lane >> 4, li = lane & 15;
    const int n = li;                                         // this lane's state
    const int tk = br4(li);                                   // this lane's token of a tile
    const int b = blockIdx.y, seg = blockIdx.z;
    const int L = ls_own(f.seqlen), cpg = f.dim / f.n_groups;
    const int cpb = W * 4 * CPR;                              // channels per workgroup
    const int bpg = (cpg + cpb - 1) / cpb;                    // workgroups per B/C group
    const int g = blockIdx.x / bpg;
    const int d_end = ls_own((g + 1) * cpg);
    const int dwave = ls_own(g * cpg + (blockIdx.x - g * bpg) * cpb + wave * 4 * CPR);   // first channel of this wave (uniform)
    const int rowch = row * CPR;                              // this row's first channel, relative to the wave's
    const int ntiles = (L + kLsT - 1) / kLsT;
    const int nck = ntiles;                                   // checkpoint rows of x (16 tokens each)
    const int tile_lo = seg * sg.seg_blocks, tile_hi = min(ntiles, tile_lo + sg.seg_blocks);
    const int t_next = tile_hi * kLsT;                        // first token right of the segment
    const bool single = ls_own((int)(bpg == 1)) != 0;         // this workgroup is the only contributor to its dB / dC rows

    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    unsigned char* wb = smem2 + wave * G2::WB;
    unsigned char* raw_u = wb + G2::RAWU;                     // [channel of the wave][128 bytes]: u as loaded
    unsigned char* raw_d = wb + G2::RAWD;                     // delta as loaded
    float* dlb = reinterpret_cast<float*>(wb + G2::DL);       // [channel][TS] f32: delta after softplus
    float* wub = reinterpret_cast<float*>(wb + G2::WU);       // delta * u; after a token's step: its du
    float* dyb = reinterpret_cast<float*>(wb + G2::DY);       // dy (gated dout); after a token's step: its ddelta before the sigmoid factor
    float* slot = reinterpret_cast<float*>(wb + G2::SLOT);
    float* ctab = reinterpret_cast<float*>(smem2 + W * G2::WB) + (wave * CPR * 4 + row) * 2;   // [wave][channel][row][D | -]
    unsigned char* stage = smem2 + W * G2::WB + W * CPR * 8 * 4;               // [B | C][state][16 tokens] raw, next tile
    constexpr int PT = NS * (int)sizeof(T);                   // 16-byte pieces per tensor and tile

    typedef vivim_ssm_bwd_params BP;
    LsRow<T> rB, rC;
    rB.init(LS_OFF(BP, f.B), LS_OFF(BP, f.B_batch_stride), b, g, f.B_dstate_stride, n);
    rC.init(LS_OFF(BP, f.C), LS_OFF(BP, f.C_batch_stride), b, g, f.C_dstate_stride, n);
    const bool softplus = ls_own((int)f.delta_softplus) != 0;
    const bool bc_vec = ls_own(sg.bc_vec) != 0;
    // checkpoints: this lane's state of this row's first channel, checkpoint row 0 (clamped to a valid channel)
    const float* xrow = static_cast<const float*>(f.x) + (((int64_t)b * f.dim + min(dwave + rowch, d_end - 1)) * nck) * NS + n;
    const int xcs = nck * NS;                                 // floats between the checkpoint rows of neighbouring channels

    // ---- per-channel state of the row's channels: A * log2e, the reverse carry a_{t+1} g_{t+1}, the running dA.  The four
    // channels take turns in slot 0 (a tile is exactly four steps, so the order is restored at every tile edge) ----
    float A2r[CPR], gcar[CPR], dacc[CPR];
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
        const int d = dwave + rowch + c;
        const bool cv = d < d_end;
        const int dc = cv ? d : d_end - 1;
        const float A2 = static_cast<const float*>(f.A)[dc * f.A_d_stride + n * f.A_dstate_stride] * kLog2e;
        const float bias = f.delta_bias ? static_cast<const float*>(f.delta_bias)[dc] : 0.0f;
        const float gin = (sg.S > 1 && cv) ? sg.gin[(((int64_t)b * f.dim + dc) * sg.S + seg) * NS + n] : 0.0f;
        float dl_nx = 0.0f;                                   // delta of the first token right of the segment
        if (t_next < L) {
            const float raw = to_f32<T>(static_cast<const T*>(f.delta)[b * f.delta_batch_stride + dc * f.delta_d_stride + t_next]) + bias;
            dl_nx = softplus ? softplus_ref(raw) : raw;
        }
        gcar[c] = gin * fast_exp2(dl_nx * A2);
        dacc[c] = 0.0f;
        A2r[c] = A2;
        ctab[c * 8 + 0] = f.D ? static_cast<const float*>(f.D)[dc] : 0.0f;      // every lane of the row writes the same value
    }
    // ---- span level: lane = (channel (lane >> 3) + 8 i, 16-byte piece lane & 7) ----
    float bias2[2], accD2[2] = {0.0f, 0.0f}, accB2[2] = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = min(dwave + (lane >> 3) + 8 * i, d_end - 1);
        bias2[i] = f.delta_bias ? static_cast<const float*>(f.delta_bias)[d] : 0.0f;
    }
    // (the lane's coordinates are re-derived from an opaque copy in every call: hipcc otherwise hoists the sixteen 64-bit
    // addresses of a span out of the span loop and keeps them live across the sweeps)
    auto load_span = [&](int sp) __attribute__((always_inline)) {
        const ls_kargs q = ls_fresh_kargs();
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int pc = ln >> 3, pp = ln & 7;
        const bool want_oz = HAS_Z && ls_karg<const void*>(q, LS_OFF(BP, f.out_z)) != nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cwi = pc + 8 * i;
            const int d = dwave + cwi;
            const int t = sp * TS + pp * EPV;
            const bool pv = d < d_end && t < L;               // seqlen % EPV == 0 (host check): a piece is whole or absent
            const bool mine = pv && (t >> 4) >= tile_lo && (t >> 4) < tile_hi;
            const int64_t dc = min(d, d_end - 1), tc = pv ? t : 0;
            auto ld = [&](int off_ptr, int off_bs) __attribute__((always_inline)) -> u32x4 {
                const T* ptr = ls_karg<const T*>(q, off_ptr);
                const int64_t bs = ls_karg<int64_t>(q, off_bs), ds = ls_karg<int64_t>(q, off_bs + 8);
                return *reinterpret_cast<const u32x4*>(ptr + b * bs + dc * ds + tc);
            };
            auto st = [&](int off_ptr, int off_bs, const u32x4 v) __attribute__((always_inline)) {
                T* ptr = ls_karg<T*>(q, off_ptr);
                const int64_t bs = ls_karg<int64_t>(q, off_bs), ds = ls_karg<int64_t>(q, off_bs + 8);
                if (mine) *reinterpret_cast<u32x4*>(ptr + b * bs + dc * ds + tc) = v;
            };
            const u32x4 ru = ld(LS_OFF(BP, f.u), LS_OFF(BP, f.u_batch_stride));
            const u32x4 rd = ld(LS_OFF(BP, f.delta), LS_OFF(BP, f.delta_batch_stride));
            const u32x4 rdo = ld(LS_OFF(BP, dout), LS_OFF(BP, dout_batch_stride));
            float dy[EPV], uu[EPV], dl[EPV], wu[EPV];
            PK::unpack(rdo, dy);
            PK::unpack(ru, uu);
            PK::unpack(rd, dl);
            if (HAS_Z) {
                const u32x4 rz = ld(LS_OFF(BP, f.z), LS_OFF(BP, f.z_batch_stride));
                const u32x4 ro = ld(LS_OFF(BP, f.out), LS_OFF(BP, f.out_batch_stride));
                float zf[EPV], of[EPV], dzv[EPV], ozv[EPV];
                PK::unpack(rz, zf);
                PK::unpack(ro, of);
#pragma unroll
                for (int e = 0; e < EPV; ++e) {
                    const float sgm = sigmoidf_fast(zf[e]);
                    dzv[e] = dy[e] * of[e] * sgm * (1.0f + zf[e] * (1.0f - sgm));        // bwd_kernel.cuh:186-191
                    ozv[e] = of[e] * zf[e] * sgm;                                         // bwd_kernel.cuh:193-204
                    dy[e] *= zf[e] * sgm;
                }
                st(LS_OFF(BP, dz), LS_OFF(BP, dz_batch_stride), PK::pack(dzv));
                if (want_oz) st(LS_OFF(BP, f.out_z), LS_OFF(BP, f.out_z_batch_stride), PK::pack(ozv));
            }
            float sD = 0.0f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float raw = dl[e] + bias2[i];
                dl[e] = pv ? (softplus ? softplus_ref(raw) : raw) : 0.0f;                 // absent tokens: identity steps
                dy[e] = pv ? dy[e] : 0.0f;
                wu[e] = dl[e] * uu[e];
                sD = fmaf(dy[e], uu[e], sD);
            }
            accD2[i] += mine ? sD : 0.0f;
            *reinterpret_cast<u32x4*>(raw_u + cwi * 128 + pp * 16) = ru;
            *reinterpret_cast<u32x4*>(raw_d + cwi * 128 + pp * 16) = rd;
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                *reinterpret_cast<float4*>(dlb + cwi * TS + pp * EPV + e) = float4{dl[e], dl[e + 1], dl[e + 2], dl[e + 3]};
                *reinterpret_cast<float4*>(wub + cwi * TS + pp * EPV + e) = float4{wu[e], wu[e + 1], wu[e + 2], wu[e + 3]};
                *reinterpret_cast<float4*>(dyb + cwi * TS + pp * EPV + e) = float4{dy[e], dy[e + 1], dy[e + 2], dy[e + 3]};
            }
        }
        wave_lds_fence();
    };
    auto store_span = [&](int sp) __attribute__((always_inline)) {
        if constexpr (kAbl2 == 3) return;
        wave_lds_fence();
        const ls_kargs q = ls_fresh_kargs();
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int pc = ln >> 3, pp = ln & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cwi = pc + 8 * i;
            const int d = dwave + cwi;
            const int t = sp * TS + pp * EPV;
            const bool mine = d < d_end && t < L && (t >> 4) >= tile_lo && (t >> 4) < tile_hi;
            float du[EPV], dd[EPV], raw[EPV];
            PK::unpack(*reinterpret_cast<const u32x4*>(raw_d + cwi * 128 + pp * 16), raw);
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(wub + cwi * TS + pp * EPV + e);
                const float4 b4 = *reinterpret_cast<const float4*>(dyb + cwi * TS + pp * EPV + e);
                du[e] = a4.x; du[e + 1] = a4.y; du[e + 2] = a4.z; du[e + 3] = a4.w;
                dd[e] = b4.x; dd[e + 1] = b4.y; dd[e + 2] = b4.z; dd[e + 3] = b4.w;
            }
            float sB = 0.0f;
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                const float r = raw[e] + bias2[i];
                if (softplus) dd[e] *= r <= 20.0f ? sigmoidf_fast(r) : 1.0f;              // bwd_kernel.cuh:439-452
                sB += dd[e];
            }
            accB2[i] += mine ? sB : 0.0f;
            T* pu = ls_karg<T*>(q, LS_OFF(BP, du));
            const int64_t ubs = ls_karg<int64_t>(q, LS_OFF(BP, du_batch_stride)), uds = ls_karg<int64_t>(q, LS_OFF(BP, du_d_stride));
            T* pd = ls_karg<T*>(q, LS_OFF(BP, ddelta));
            const int64_t dbs = ls_karg<int64_t>(q, LS_OFF(BP, ddelta_batch_stride)), dds = ls_karg<int64_t>(q, LS_OFF(BP, ddelta_d_stride));
            if (mine) {
                *reinterpret_cast<u32x4*>(pu + b * ubs + (int64_t)d * uds + t) = PK::pack(du);
                *reinterpret_cast<u32x4*>(pd + b * dbs + (int64_t)d * dds + t) = PK::pack(dd);
            }
        }
        wave_lds_fence();
    };

    // The checkpoints (state at the left edge of a tile) of the row's four channels are requested a whole tile ahead and taken
    // out of the memory queue BEFORE that tile's dB / dC atomics are issued: the queue completes in order, so a load that is
    // waited for behind an atomic waits for the atomic (~3000 cycles with every CU adding).  The step loop itself then
    // contains no vector-memory wait at all.
    auto fetch_h = [&](int tile, float (&hn)[CPR]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < CPR; ++c) {
            const bool v = tile > 0 && tile >= tile_lo && dwave + rowch + c < d_end;
            hn[c] = xrow[v ? c * xcs + (tile - 1) * NS : 0];
        }
    };

    float Bv[16], Cv[16], dBv[16], dCv[16];
    bool staged = false;
    float hcur[CPR], hnxt[CPR];
    fetch_h(tile_hi - 1, hcur);
#ifdef VIVIM_STAMPS
    int stamp_step = 0;                                       // DIAGNOSTIC builds only (tools/ls2_lab.hip)
#define LS2_STAMP(slot) VIVIM_STAMP(stamp_step, slot, wave, lane)
#else
#define LS2_STAMP(slot) ((void)0)
#endif

    const int sp_hi = (tile_hi - 1) / TPS, sp_lo = tile_lo / TPS;
#pragma unroll 1
    for (int sp = sp_hi; sp >= sp_lo; --sp) {
        load_span(sp);
        const int tl_hi = min(tile_hi, (sp + 1) * TPS) - 1, tl_lo = max(tile_lo, sp * TPS);
#pragma unroll 1
        for (int tile = tl_hi; tile >= tl_lo; --tile) {
            const int t0 = tile * kLsT;
            const int tokb = (tile - sp * TPS) * kLsT;        // the tile's first token inside the span
            if (staged) {                                     // this tile's rows were staged during the previous one
                constexpr int NV = (int)sizeof(T);
                u32x4 rb[NV], rc[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    rb[i] = *reinterpret_cast<const u32x4*>(stage + (n * NV + i) * 16);
                    rc[i] = *reinterpret_cast<const u32x4*>(stage + (PT + n * NV + i) * 16);
                }
                LsUnpack<T>::run(rb, Bv);
                LsUnpack<T>::run(rc, Cv);
            } else {
                const bool vec = bc_vec && t0 + kLsT <= L;
                const ls_kargs q = ls_fresh_kargs();
                rB.load16(q, t0, L, vec, Bv);
                rC.load16(q, t0, L, vec, Cv);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { dBv[k] = 0.0f; dCv[k] = 0.0f; }
            const bool stage_next = W == 4 && bc_vec && tile - 1 >= tile_lo;   // (a tile left of another one is whole)
            u32x4 sb = {0u, 0u, 0u, 0u}, sc = {0u, 0u, 0u, 0u};
            fetch_h(tile - 1, hnxt);
#pragma unroll 1
            for (int c = 0; c < CPR; ++c) {
                const int d = dwave + rowch + c;
                const bool cv = d < d_end;                    // uniform per row
                const int cw = rowch + c;
                const float h_in = (cv && tile > 0) ? hcur[0] : 0.0f;
                hcur[0] = hcur[1]; hcur[1] = hcur[2]; hcur[2] = hcur[3];          // the next channel moves into slot 0
                if (c == CPR - 1 && stage_next) {             // the next tile's B / C rows: one 16-byte piece of each per thread
                    const ls_kargs q = ls_fresh_kargs();
                    const bool mine = tid < PT;
                    const unsigned pn = (unsigned)tid / (unsigned)sizeof(T), pq = (unsigned)tid % (unsigned)sizeof(T);
                    sb = rB.piece(q, (tile - 1) * kLsT, mine ? pn : 0u, mine ? pq : 0u);
                    sc = rC.piece(q, (tile - 1) * kLsT, mine ? pn : 0u, mine ? pq : 0u);
                }
                LS2_STAMP(0);
                const float* dlr = dlb + cw * TS + tokb;      // the 16 tokens of this step: the same address in all lanes of a row
                float* wur = wub + cw * TS + tokb;
                float* dyr = dyb + cw * TS + tokb;
                // this lane's own token (for the outputs): requested now, used after the sweeps
                const float dl_t = dlr[tk], dy_t = dyr[tk];
                const float uu = to_f32<T>(*reinterpret_cast<const T*>(raw_u + cw * 128 + (tokb + tk) * (int)sizeof(T)));
                const float A2 = A2r[0];
                LS2_STAMP(1);
                // ---- forward states of the tile, from the checkpoint ----
                float a[16], h[16];
                {
                    float hp = h_in;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        ls2_sched_fence();
                        float4 d4, w4;
                        if constexpr (kAbl2 == 9) { d4 = float4{dl_t, dl_t, dl_t, dl_t}; w4 = float4{dy_t, dy_t, uu, uu}; }   // timing only: no LDS reads in the sweeps
                        else { d4 = *reinterpret_cast<const float4*>(dlr + 4 * m); w4 = *reinterpret_cast<const float4*>(wur + 4 * m); }
                        const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k = 4 * m + j;
                            if constexpr (kAbl2 == 6) { a[k] = A2; h[k] = hp + Bv[k]; continue; }
                            a[k] = fast_exp2(dd[j] * A2);
                            hp = fmaf(a[k], hp, ww[j] * Bv[k]);                          // h_t = a_t h_{t-1} + d_t u_t B_t
                            h[k] = hp;
                        }
                    }
                }
                LS2_STAMP(2);
                // ---- reverse sweep: g_t = a_{t+1} g_{t+1} + C_t dy_t; ag = a_t g_t is the carry to the left ----
                float s1[16], s2[16], z1[8], z2[8], w1[4], w2[4], v1[2], v2[2], S1 = 0.0f, S2 = 0.0f;
                {
                    float ag = gcar[0], dA0 = dacc[0], dA1 = 0.0f;
                    sfor_down<4>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        ls2_sched_fence();
                        float4 d4, w4, y4;
                        if constexpr (kAbl2 == 9) { d4 = float4{dl_t, dl_t, dl_t, dl_t}; w4 = float4{dy_t, dy_t, uu, uu}; y4 = float4{uu, dy_t, dl_t, uu}; }
                        else { d4 = *reinterpret_cast<const float4*>(dlr + 4 * m); w4 = *reinterpret_cast<const float4*>(wur + 4 * m);
                               y4 = *reinterpret_cast<const float4*>(dyr + 4 * m); }
                        const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
                        sfor_down<4>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            constexpr int k = 4 * m + j;
                            const float gk = fmaf(yy[j], Cv[k], ag);                        // g_t
                            ag = gk * a[k];
                            const float x = ag * (k > 0 ? h[k > 0 ? k - 1 : 0] : h_in);    // g_t a_t h_{t-1}
                            s1[k] = gk * Bv[k];
                            s2[k] = A2 * x;
                            if constexpr (k & 1) dA1 = fmaf(dd[j], x, dA1); else dA0 = fmaf(dd[j], x, dA0);   // two chains
                            dBv[k] = fmaf(ww[j], gk, dBv[k]);
                            dCv[k] = fmaf(yy[j], h[k], dCv[k]);
                            if constexpr (kAbl2 == 7) { if (k == 0) { S1 = s1[0] + s1[5] + s1[15]; S2 = s2[0] + s2[7] + s2[15]; } return; }
                            ls2_reduce_down<k>(s1, z1, w1, v1, S1, li);
                            ls2_reduce_down<k>(s2, z2, w2, v2, S2, li);
                        });
                    });
                    // the next channel moves into slot 0
                    gcar[0] = gcar[1]; gcar[1] = gcar[2]; gcar[2] = gcar[3]; gcar[3] = ag;
                    dacc[0] = dacc[1]; dacc[1] = dacc[2]; dacc[2] = dacc[3]; dacc[3] = dA0 + dA1;
                    A2r[0] = A2r[1]; A2r[1] = A2r[2]; A2r[2] = A2r[3]; A2r[3] = A2;
                }
                LS2_STAMP(3);
                // ---- this lane's token: du, and ddelta up to the sigmoid factor (applied at span level), into the delta u / dy
                // slots of the token (all sixteen were consumed by the sweeps above) ----
                wave_lds_fence();
                wur[tk] = fmaf(dl_t, S1, ctab[c * 8 + 0] * dy_t);
                dyr[tk] = fmaf(uu, S1, S2 * kLn2);                                        // S2 carries A * log2e
                LS2_STAMP(4);
#ifdef VIVIM_STAMPS
                if (c < CPR - 1) ++stamp_step;
#endif
            }
            // ---- dB / dC of the tile: the row's channels are summed in dBv / dCv; add the four rows in registers, then the
            // waves of the workgroup through LDS.  R[k], row r = the wave's total of vector 4 k + r (0-15 dB, 16-31 dC by token).
            if constexpr (kAbl2 == 1) continue;
            LS2_STAMP(5);
            // (the dB / dC bases: scalar loads from the argument block, requested here so that they are in when the barriers are)
            const ls_kargs qe = ls_fresh_kargs();
            float* __restrict__ dBg = ls_karg<float*>(qe, LS_OFF(BP, dB)) + b * ls_karg<int64_t>(qe, LS_OFF(BP, dB_batch_stride)) + g * ls_karg<int64_t>(qe, LS_OFF(BP, dB_group_stride));
            float* __restrict__ dCg = ls_karg<float*>(qe, LS_OFF(BP, dC)) + b * ls_karg<int64_t>(qe, LS_OFF(BP, dC_batch_stride)) + g * ls_karg<int64_t>(qe, LS_OFF(BP, dC_group_stride));
            const int dBns = (int)ls_karg<int64_t>(qe, LS_OFF(BP, dB_dstate_stride)), dCns = (int)ls_karg<int64_t>(qe, LS_OFF(BP, dC_dstate_stride));
            float R[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                R[k] = ls2_rows32(ls2_rows16(dBv[4 * k], dBv[4 * k + 1]), ls2_rows16(dBv[4 * k + 2], dBv[4 * k + 3]));
                R[4 + k] = ls2_rows32(ls2_rows16(dCv[4 * k], dCv[4 * k + 1]), ls2_rows16(dCv[4 * k + 2], dCv[4 * k + 3]));
            }
            // Both barriers sit around the slot writes: "everybody is done reading the previous tile's slots and this tile's
            // staged rows", then "slots and the next tile's rows are written".
            LS2_STAMP(6);
            lds_barrier();
            LS2_STAMP(7);
            if (stage_next && tid < PT) {
                *reinterpret_cast<u32x4*>(stage + tid * 16) = sb;
                *reinterpret_cast<u32x4*>(stage + (PT + tid) * 16) = sc;
            }
            staged = stage_next;
            *reinterpret_cast<float4*>(slot + lane * 4) = float4{R[0], R[1], R[2], R[3]};
            *reinterpret_cast<float4*>(slot + (kWave + lane) * 4) = float4{R[4], R[5], R[6], R[7]};
            lds_barrier();
            LS2_STAMP(8);
            // the next tile's checkpoints: out of the memory queue before this tile's atomics go in
            asm volatile("" : "+v"(hnxt[0]), "+v"(hnxt[1]), "+v"(hnxt[2]), "+v"(hnxt[3]));
#pragma unroll
            for (int c = 0; c < CPR; ++c) hcur[c] = hnxt[c];
            {
                auto slot_of = [&](int e) __attribute__((always_inline)) -> const float* {
                    const int isC = e >> 8, en = (e >> 4) & 15, ek = e & 15;
                    const int i = isC * 16 + ek, k = i >> 2, r = i & 3;
                    return reinterpret_cast<const float*>(smem2 + G2::SLOT) + ((k >> 2) * kWave + r * 16 + en) * 4 + (k & 3);
                };
                auto emit = [&](int e, float acc) __attribute__((always_inline)) {
                    const int isC = e >> 8, en = (e >> 4) & 15, ek = e & 15;
                    if (t0 + ek < L) {
                        float* dst = isC ? dCg + en * dCns : dBg + en * dBns;
                        if (single) dst[t0 + ek] = acc;                   // the only contributor: plain store, deterministic
                        else atomicAdd(dst + t0 + ek, acc);
                    }
                };
                if (W == 4) {                                 // two outputs per thread, their eight slot reads in flight together
                    const float* p0 = slot_of(tid);
                    const float* p1 = slot_of(tid + 256);
                    constexpr int WS = G2::WB / 4;
                    const float a0 = p0[0], a1 = p0[WS], a2 = p0[2 * WS], a3 = p0[3 * WS];
                    const float b0 = p1[0], b1 = p1[WS], b2 = p1[2 * WS], b3 = p1[3 * WS];
                    emit(tid, (a0 + a1) + (a2 + a3));
                    emit(tid + 256, (b0 + b1) + (b2 + b3));
                } else {
                    for (int e = tid; e < 2 * NS * 16; e += blockDim.x) {
                        const float* sp0 = slot_of(e);
                        float acc = sp0[0];
                        for (int s = 1; s < W; ++s) acc += sp0[s * (G2::WB / 4)];
                        emit(e, acc);
                    }
                }
            }
            LS2_STAMP(9);
#ifdef VIVIM_STAMPS
            ++stamp_step;
#endif
        }
        store_span(sp);
    }
    // ---- per-channel sums ----
#pragma unroll
    for (int c = 0; c < CPR; ++c) {
        const int d = dwave + rowch + c;
        if (d >= d_end) continue;                             // uniform per row
        atomicAdd(static_cast<float*>(p.dA) + d * p.dA_d_stride + n * p.dA_dstate_stride, dacc[c]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                             // dD, dbias: the eight lanes that share a channel at span level
        float sD = accD2[i], sBs = accB2[i];
        sD += dpp_mov<0xb1>(0.0f, sD);   sBs += dpp_mov<0xb1>(0.0f, sBs);      // lane ^ 1
        sD += dpp_mov<0x4e>(0.0f, sD);   sBs += dpp_mov<0x4e>(0.0f, sBs);      // lane ^ 2
        sD += dpp_mov<0x141>(0.0f, sD);  sBs += dpp_mov<0x141>(0.0f, sBs);     // lane ^ 7 (row_half_mirror): the other quad
        const int d = dwave + (lane >> 3) + 8 * i;
        if ((lane & 7) == 0 && d < d_end) {
            if (p.dD) atomicAdd(static_cast<float*>(p.dD) + d, sD);
            if (p.ddelta_bias) atomicAdd(static_cast<float*>(p.ddelta_bias) + d, sBs);
        }
    }
}

// =========================================================================================================================
// Host side
// =========================================================================================================================
size_t ls2_bwd_smem(int W, int itype) {
    const int wb = itype == VIVIM_F32 ? Ls2Geom<float>::WB : Ls2Geom<bf16_t>::WB;
    const int es = itype == VIVIM_F32 ? 4 : 2;
    static const size_t pad = getenv("VIVIM_LS2_SMEM_PAD") ? (size_t)atoi(getenv("VIVIM_LS2_SMEM_PAD")) : 0;   // occupancy experiments
    return (size_t)W * wb + (size_t)W * kLsCPR * 8 * 4 + (size_t)2 * 16 * 16 * es + pad;
}

// Vector path: every activation row 16-byte aligned and a whole number of 16-byte pieces long.
bool ls2_bwd_ok(const vivim_ssm_bwd_params& p) {
    const vivim_ssm_fwd_params& f = p.f;
    if (f.dstate != 16 || !f.is_variable_B || !f.is_variable_C) return false;
    const int64_t epv = f.itype == VIVIM_F32 ? 4 : 8;
    auto al = [&](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto st = [&](int64_t e) { return e % epv == 0; };
    if (f.seqlen % epv != 0 || !al(f.u) || !al(f.delta) || !al(p.dout) || !al(p.du) || !al(p.ddelta) ||
        !st(f.u_batch_stride) || !st(f.u_d_stride) || !st(f.delta_batch_stride) || !st(f.delta_d_stride) ||
        !st(p.dout_batch_stride) || !st(p.dout_d_stride) || !st(p.du_batch_stride) || !st(p.du_d_stride) ||
        !st(p.ddelta_batch_stride) || !st(p.ddelta_d_stride))
        return false;
    if (f.z && (!al(f.z) || !al(f.out) || !al(p.dz) || !st(f.z_batch_stride) || !st(f.z_d_stride) ||
                !st(f.out_batch_stride) || !st(f.out_d_stride) || !st(p.dz_batch_stride) || !st(p.dz_d_stride) ||
                (f.out_z && (!al(f.out_z) || !st(f.out_z_batch_stride) || !st(f.out_z_d_stride)))))
        return false;
    return true;
}

template <typename T> static void ls2_launch_t(const vivim_ssm_bwd_params& p, const LsSeg& sg, int W, hipStream_t stream) {
    const vivim_ssm_fwd_params& f = p.f;
    const int cpg = f.dim / f.n_groups;
    const int cpb = W * 4 * kLsCPR;
    const int bpg = (cpg + cpb - 1) / cpb;
    const dim3 grid(bpg * f.n_groups, f.batch, sg.S);
    const size_t smem = ls2_bwd_smem(W, f.itype);
    auto launch = [&](auto kernel) {
        if (smem > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kernel, grid, dim3(W * kWave), smem, stream, p, sg);
    };
    if (f.z) launch(ssm_ls2_bwd_kernel<T, true>); else launch(ssm_ls2_bwd_kernel<T, false>);
}

void ls2_bwd_launch(const vivim_ssm_bwd_params& p, const LsSeg& sg, int W, hipStream_t stream) {
    switch (p.f.itype) {
        case VIVIM_F32: ls2_launch_t<float>(p, sg, W, stream); break;
        case VIVIM_F16: ls2_launch_t<f16_t>(p, sg, W, stream); break;
        case VIVIM_BF16: ls2_launch_t<bf16_t>(p, sg, W, stream); break;
    }
}

int ls2_bwd_blocks_per_cu(int itype, bool has_z, int W) {
    int nb = 0;
    const size_t smem = ls2_bwd_smem(W, itype);
    hipError_t e = hipSuccess;
    auto q = [&](auto kernel) {
        if (smem > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, W * kWave, smem);
    };
    auto by_t = [&](auto tag) {
        typedef decltype(tag) T;
        if (has_z) q(ssm_ls2_bwd_kernel<T, true>); else q(ssm_ls2_bwd_kernel<T, false>);
    };
    if (itype == VIVIM_F32) by_t(float{}); else if (itype == VIVIM_F16) by_t(f16_t{}); else by_t(bf16_t{});
    if (e != hipSuccess || nb <= 0) { (void)hipGetLastError(); nb = 2; }
    return nb;
}

}  // namespace vivim
