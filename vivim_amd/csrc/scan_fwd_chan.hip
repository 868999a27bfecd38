// scan_fwd_chan.hip -- selective SSM scan forward, "lanes = channels" kernels (gfx950, wave64).
//
// Same math as scan_fwd.hip (selective_scan_fwd_kernel.cuh:67-303); a different mapping, built because the
// n-split kernel turned out instruction- and phase-bound:
//   * a WAVE owns 64 channels of one (batch, B/C group) and a SEGMENT of the token axis; lane = channel.  Each
//     lane walks its tokens serially with all N=16 states h[n] in registers: the recurrence needs no cross-lane
//     operation, no barrier and no second "apply" sweep -- 4 VALU ops + 1 exp per state update, 16 independent
//     dependency chains per lane;
//   * B_n[t] and C_n[t] are the same for the 64 channels of the wave, so they are read with SCALAR loads (the
//     pointers are cast to the constant address space: the rows are never written by this kernel) and enter
//     the fma as SGPR operands -- no LDS traffic, no VGPRs;
//   * u / delta / z / out / out_z rows are token-contiguous while lanes are channels: tiles of 64 channels x TT
//     tokens go through wave-private LDS (coalesced 16-byte global accesses on one side, conflict-free
//     ds_read_b128 / ds_write_b128 of a lane's own row on the other; rows padded by 16 bytes);
//   * the token axis is split so that >= ~2048 waves exist.  PASS 1 gives every segment's end state for zero
//     inflow plus sum(delta); a carry kernel chains them (h_in[s+1] = exp2(A*log2e*sum_s) * h_in[s] + H_s); PASS 2
//     recomputes the recurrence from the true inflow and produces out / out_z and the checkpoints x.  The state
//     is re-derived instead of stored: 2 exp per state update instead of a 16-float-per-token round trip.
#include <stdlib.h>
#include "common.cuh"

namespace vivim {

constexpr int kChN = 16;           // states (compile time: they live in registers)
constexpr int kChWaves = 2;        // independent waves per workgroup (15 KB of LDS each)

struct FwdSeg {
    int S, seg_tiles;              // segments, TT-token tiles per segment
    float* H;                      // [batch][dim][S][N]  PASS 1: end state for zero inflow; after the carry kernel: inflow
    float* dsum;                   // [batch][dim][S]     sum of softplus(delta + bias) over the segment
};

typedef const __attribute__((address_space(4))) uint32_t* cptr32;

// B/C scalars of GT consecutive tokens of one state row: 8 bytes through the scalar cache.
template <typename T> struct ScalarRow;
template <> struct ScalarRow<float> {
    static constexpr int GT = 2;
    uint32_t w0, w1;
    __device__ __forceinline__ void load(const float* row, int t) {
        cptr32 q = (cptr32)(uintptr_t)(row + t);
        w0 = q[0]; w1 = q[1];
    }
    __device__ __forceinline__ float get(int j) const { return __uint_as_float(j == 0 ? w0 : w1); }
};
template <> struct ScalarRow<bf16_t> {
    static constexpr int GT = 4;
    uint32_t w0, w1;
    __device__ __forceinline__ void load(const bf16_t* row, int t) {
        cptr32 q = (cptr32)(uintptr_t)(row + t);
        w0 = q[0]; w1 = q[1];
    }
    __device__ __forceinline__ float get(int j) const {
        const uint32_t w = j < 2 ? w0 : w1;
        return __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
    }
};
template <> struct ScalarRow<f16_t> {
    static constexpr int GT = 4;
    uint32_t w0, w1;
    __device__ __forceinline__ void load(const f16_t* row, int t) {
        cptr32 q = (cptr32)(uintptr_t)(row + t);
        w0 = q[0]; w1 = q[1];
    }
    __device__ __forceinline__ float get(int j) const {
        const uint32_t w = j < 2 ? w0 : w1;
        const uint16_t hbits = (uint16_t)((j & 1) ? (w >> 16) : (w & 0xffffu));
        return (float)__builtin_bit_cast(f16_t, hbits);
    }
};

template <typename T, int PASS, bool HAS_Z>
__global__ void __launch_bounds__(kChWaves * kWave) ssm_fwd_chan_kernel(const vivim_ssm_fwd_params p, const FwdSeg sg) {
    constexpr int N = kChN;
    constexpr int EPV = 16 / (int)sizeof(T);          // elements per 16-byte vector
    constexpr int TT = 4 * EPV;                       // tokens per tile: 64 bytes of a row (32 for 16-bit, 16 for fp32)
    constexpr int ROWB = 64 + 16;                     // padded LDS row, bytes
    constexpr int GT = ScalarRow<T>::GT;
    constexpr int NARR = PASS == 2 && HAS_Z ? 3 : 2;  // resident tiles: u, delta (, z)
    __shared__ __attribute__((aligned(16))) unsigned char lds[kChWaves * NARR * kWave * ROWB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, seg = blockIdx.z;
    const int L = p.seqlen;
    const int cpg = p.dim / p.n_groups;
    const int bpg = (cpg + kWave - 1) / kWave;        // 64-channel blocks per B/C group
    const int cb = blockIdx.x * kChWaves + wave;
    if (cb >= bpg * p.n_groups) return;               // waves are independent: no barrier below
    const int g = cb / bpg;
    const int c0 = g * cpg + (cb - g * bpg) * kWave;
    const int cend = (g + 1) * cpg;
    const int d = min(c0 + lane, cend - 1);           // surplus lanes shadow the last channel, never stored
    const bool dvalid = c0 + lane < cend;

    unsigned char* tile_u = lds + (wave * NARR + 0) * kWave * ROWB;
    unsigned char* tile_d = lds + (wave * NARR + 1) * kWave * ROWB;
    unsigned char* tile_z = lds + (wave * NARR + (NARR - 1)) * kWave * ROWB;   // only used when NARR == 3

    const float* __restrict__ A = static_cast<const float*>(p.A);
    float A2[N], h[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        A2[n] = A[d * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;     // fwd_kernel.cuh:168-175
        h[n] = (PASS == 2 && seg > 0) ? sg.H[(((int64_t)b * p.dim + d) * sg.S + seg) * N + n] : 0.0f;
    }
    const float Dv = p.D ? static_cast<const float*>(p.D)[d] : 0.0f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.0f;
    const T* __restrict__ Brow = static_cast<const T*>(p.B) + b * p.B_batch_stride + g * p.B_group_stride;
    const T* __restrict__ Crow = static_cast<const T*>(p.C) + b * p.C_batch_stride + g * p.C_group_stride;
    const int64_t sB = __builtin_amdgcn_readfirstlane((int)p.B_dstate_stride);
    const int64_t sC = __builtin_amdgcn_readfirstlane((int)p.C_dstate_stride);

    // cooperative tile I/O: instruction i moves rows i*16 + lane/4, 16-byte column lane%4
    const int io_col = lane & 3;
    const int io_row0 = lane >> 2;
    const T* __restrict__ gu = static_cast<const T*>(p.u) + b * p.u_batch_stride;
    const T* __restrict__ gd = static_cast<const T*>(p.delta) + b * p.delta_batch_stride;
    const T* __restrict__ gz = HAS_Z ? static_cast<const T*>(p.z) + b * p.z_batch_stride : nullptr;
    T* __restrict__ go = static_cast<T*>(p.out) + b * p.out_batch_stride;
    T* __restrict__ goz = HAS_Z ? static_cast<T*>(p.out_z) + b * p.out_z_batch_stride : nullptr;
    float* __restrict__ xck = static_cast<float*>(p.x);
    const int nck = (L + kChunk - 1) / kChunk;

    const int ntiles = (L + TT - 1) / TT;
    const int tile_lo = seg * sg.seg_tiles, tile_hi = min(ntiles, tile_lo + sg.seg_tiles);
    float dsum = 0.0f;

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int t0 = tile * TT;
        // ---- global -> LDS (coalesced 64-byte row segments) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 16 + io_row0;
            const int ch = min(c0 + row, cend - 1);
            const int t = t0 + io_col * EPV;
            const bool ok = t < L;                    // L % EPV == 0 (host): a 16-byte column is all-in or all-out
            typedef uint32_t __attribute__((ext_vector_type(4))) v4;
            union { RawK<T, EPV> r; v4 v; } cu, cd, cz;
            cu.r = load_vec<T, EPV>(gu + ch * p.u_d_stride + t, ok);
            cd.r = load_vec<T, EPV>(gd + ch * p.delta_d_stride + t, ok);
            *reinterpret_cast<v4*>(tile_u + row * ROWB + io_col * 16) = cu.v;
            *reinterpret_cast<v4*>(tile_d + row * ROWB + io_col * 16) = cd.v;
            if (NARR == 3) {
                cz.r = load_vec<T, EPV>(gz + ch * p.z_d_stride + t, ok);
                *reinterpret_cast<v4*>(tile_z + row * ROWB + io_col * 16) = cz.v;
            }
        }
        wave_lds_fence();
        // ---- the lane's own row: TT tokens in blocks of EPV ----
#pragma unroll 1
        for (int blk = 0; blk < 4; ++blk) {
            const int tb = t0 + blk * EPV;
            float uf[EPV], df[EPV], zf[EPV], yo[EPV], yz[EPV];
            {
                typedef uint32_t __attribute__((ext_vector_type(4))) v4;
                union { RawK<T, EPV> r; v4 v; } cu, cd, cz;
                cu.v = *reinterpret_cast<const v4*>(tile_u + lane * ROWB + blk * 16);
                cd.v = *reinterpret_cast<const v4*>(tile_d + lane * ROWB + blk * 16);
                unpack(cu.r, uf);
                unpack(cd.r, df);
                if (NARR == 3) {
                    cz.v = *reinterpret_cast<const v4*>(tile_z + lane * ROWB + blk * 16);
                    unpack(cz.r, zf);
                }
            }
#pragma unroll
            for (int q = 0; q < EPV / GT; ++q) {       // groups of GT tokens share one set of scalar B/C loads
                const int tq = __builtin_amdgcn_readfirstlane(tb + q * GT);
                const int ts = min(tq, L - GT);        // past the end: any valid address; those tokens are identity maps
                ScalarRow<T> Bs[N], Cs[N];
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    Bs[n].load(Brow + n * sB, ts);
                    if (PASS == 2) Cs[n].load(Crow + n * sC, ts);
                }
#pragma unroll
                for (int j = 0; j < GT; ++j) {
                    const int k = q * GT + j;
                    const bool in = tq + j < L;
                    const float raw = df[k] + bias;
                    const float sp = p.delta_softplus ? softplus_ref(raw) : raw;
                    const float dl = in ? sp : 0.0f;   // padded token: exp2(0) = 1, drive 0
                    const float w = dl * uf[k];
                    dsum += dl;
                    float y = Dv * uf[k];
#pragma unroll
                    for (int n = 0; n < N; ++n) {
                        const float a = fast_exp2(dl * A2[n]);
                        h[n] = fmaf(a, h[n], w * Bs[n].get(j));
                        if (PASS == 2) y = fmaf(h[n], Cs[n].get(j), y);
                    }
                    if (PASS == 2) {
                        yo[k] = y;
                        if (HAS_Z) yz[k] = y * zf[k] * sigmoidf_fast(zf[k]);        // fwd_kernel.cuh:290
                        const int tk = tq + j;
                        if (((tk + 1) & (kChunk - 1)) == 0 || tk == L - 1) {        // state after every kChunk tokens
                            const int row = tk / kChunk;
                            if (dvalid && row < nck) {
                                float* xr = xck + (((int64_t)b * p.dim + d) * nck + row) * N;
#pragma unroll
                                for (int n = 0; n < N; ++n) xr[n] = h[n];
                            }
                        }
                    }
                }
            }
            if (PASS == 2) {                           // results overwrite the lane's own consumed input columns
                typedef uint32_t __attribute__((ext_vector_type(4))) v4;
                union { T e[EPV]; v4 v; } co, cz;
#pragma unroll
                for (int k = 0; k < EPV; ++k) co.e[k] = from_f32<T>(yo[k]);
                *reinterpret_cast<v4*>(tile_u + lane * ROWB + blk * 16) = co.v;
                if (HAS_Z) {
#pragma unroll
                    for (int k = 0; k < EPV; ++k) cz.e[k] = from_f32<T>(yz[k]);
                    *reinterpret_cast<v4*>(tile_z + lane * ROWB + blk * 16) = cz.v;
                }
            }
        }
        wave_lds_fence();
        if (PASS == 2) {                               // LDS -> global, coalesced
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + io_row0;
                const int ch = c0 + row;
                const int t = t0 + io_col * EPV;
                if (ch < cend && t < L) {
                    typedef uint32_t __attribute__((ext_vector_type(4))) v4;
                    *reinterpret_cast<v4*>(go + ch * p.out_d_stride + t) =
                        *reinterpret_cast<const v4*>(tile_u + row * ROWB + io_col * 16);
                    if (HAS_Z)
                        *reinterpret_cast<v4*>(goz + ch * p.out_z_d_stride + t) =
                            *reinterpret_cast<const v4*>(tile_z + row * ROWB + io_col * 16);
                }
            }
            wave_lds_fence();
        }
    }
    if (PASS == 1 && dvalid) {
        float* Hs = sg.H + (((int64_t)b * p.dim + d) * sg.S + seg) * N;
#pragma unroll
        for (int n = 0; n < N; ++n) Hs[n] = h[n];
        sg.dsum[((int64_t)b * p.dim + d) * sg.S + seg] = dsum;
    }
}

// In place: H[s] (end state of segment s for zero inflow) becomes the state flowing INTO segment s.
__global__ void ssm_fwd_carry_kernel(const vivim_ssm_fwd_params p, const FwdSeg sg) {
    constexpr int N = kChN;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.batch * p.dim * N) return;
    const int n = (int)(i % N);
    const int64_t bd = i / N;
    const int dch = (int)(bd % p.dim);
    const float A2 = static_cast<const float*>(p.A)[dch * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;
    float hin = 0.0f;
    for (int s = 0; s < sg.S; ++s) {
        const int64_t k = (bd * sg.S + s) * N + n;
        const float Hs = sg.H[k];
        sg.H[k] = hin;
        hin = fmaf(fast_exp2(A2 * sg.dsum[bd * sg.S + s]), hin, Hs);
    }
}

static void fwd_chan_segmentation(const vivim_ssm_fwd_params& f, int tt, int& S, int& seg_tiles) {
    const int ntiles = (f.seqlen + tt - 1) / tt;
    const int cpg = f.dim / f.n_groups;
    const int64_t waves = (int64_t)((cpg + kWave - 1) / kWave) * f.n_groups * f.batch;
    int64_t want = (2048 + waves - 1) / waves;
    if (want > ntiles) want = ntiles;
    if (want > 1024) want = 1024;
    if (want < 1) want = 1;
    seg_tiles = (int)((ntiles + want - 1) / want);
    S = (ntiles + seg_tiles - 1) / seg_tiles;
}

// shape_only: pointers are not inspected (the workspace query may come before they are final)
static bool fwd_chan_eligible(const vivim_ssm_fwd_params& p, bool shape_only = false) {
    if (!p.is_variable_B || !p.is_variable_C || p.dstate != kChN || p.seqlen % 8 != 0) return false;
    if (p.B_dstate_stride > 0x7fffffff || p.C_dstate_stride > 0x7fffffff) return false;
    static const int forced = [] { const char* e = getenv("VIVIM_FWD_VARIANT"); return e ? atoi(e) : 0; }();
    if (forced != 0 && forced != 5) return false;      // tuning: another forward kernel was requested
    const int64_t epv = p.itype == VIVIM_F32 ? 4 : 8;
    auto al = [&](const void* q) { return shape_only || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto st = [&](int64_t e) { return e % epv == 0; };
    if (!al(p.u) || !al(p.delta) || !al(p.B) || !al(p.C) || !al(p.out) ||
        !st(p.u_batch_stride) || !st(p.u_d_stride) || !st(p.delta_batch_stride) || !st(p.delta_d_stride) ||
        !st(p.out_batch_stride) || !st(p.out_d_stride) || !st(p.B_batch_stride) || !st(p.B_group_stride) ||
        !st(p.B_dstate_stride) || !st(p.C_batch_stride) || !st(p.C_group_stride) || !st(p.C_dstate_stride))
        return false;
    if (p.z && (!al(p.z) || !al(p.out_z) || !st(p.z_batch_stride) || !st(p.z_d_stride) ||
                !st(p.out_z_batch_stride) || !st(p.out_z_d_stride)))
        return false;
    return true;
}

size_t scan_fwd_workspace_bytes(const vivim_ssm_fwd_params& f) {
    if (!fwd_chan_eligible(f, true)) return 0;
    int S, seg_tiles;
    fwd_chan_segmentation(f, f.itype == VIVIM_F32 ? 16 : 32, S, seg_tiles);
    if (S <= 1) return 16;   // still selects the channel kernel (non-zero), nothing is stored
    return (size_t)f.batch * f.dim * S * (kChN + 1) * sizeof(float);
}

template <typename T>
static bool launch_fwd_chan(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    constexpr int TT = 4 * (16 / (int)sizeof(T));
    FwdSeg sg = {1, (p.seqlen + TT - 1) / TT, nullptr, nullptr};
    int S, seg_tiles;
    fwd_chan_segmentation(p, TT, S, seg_tiles);
    const size_t need = (size_t)p.batch * p.dim * S * (kChN + 1) * sizeof(float);
    if (S > 1) {
        if (!p.workspace || (size_t)p.workspace_bytes < need) return false;
        sg.S = S;
        sg.seg_tiles = seg_tiles;
        sg.H = static_cast<float*>(p.workspace);
        sg.dsum = sg.H + (size_t)p.batch * p.dim * S * kChN;
    }
    const int cpg = p.dim / p.n_groups;
    const int blocks = (((cpg + kWave - 1) / kWave) * p.n_groups + kChWaves - 1) / kChWaves;
    const dim3 block(kChWaves * kWave);
    if (sg.S > 1) {
        hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 1, false>), dim3(blocks, p.batch, sg.S), block, 0, stream, p, sg);
        const int64_t nthr = (int64_t)p.batch * p.dim * kChN;
        hipLaunchKernelGGL(ssm_fwd_carry_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, p, sg);
    }
    if (p.z) hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 2, true>), dim3(blocks, p.batch, sg.S), block, 0, stream, p, sg);
    else     hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 2, false>), dim3(blocks, p.batch, sg.S), block, 0, stream, p, sg);
    return true;
}

bool try_fwd_chan(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    if (!fwd_chan_eligible(p)) return false;
    switch (p.itype) {
        case VIVIM_F32: return launch_fwd_chan<float>(p, stream);
        case VIVIM_F16: return launch_fwd_chan<f16_t>(p, stream);
        case VIVIM_BF16: return launch_fwd_chan<bf16_t>(p, stream);
    }
    return false;
}

}  // namespace vivim
