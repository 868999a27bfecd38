// layernorm.hip -- LayerNorm over the channel axis of a CHANNEL-major token tensor, forward and backward (SURVEY.md 8f row 4:
// the two nn.LayerNorms around the Mamba call, modeling/vivim.py:155-156).
//
// MambaLayer holds its activations as (B, C, nf*H*W) and hands the norm the transposed VIEW (B, L, C) with strides
// (C*L, 1, L) (modeling/vivim.py:151-155).  The ATen path first makes that view contiguous (one full copy kernel) and then
// runs its row kernels; here the transpose IS the kernel.
//
// One WAVE owns a tile of TT tokens x all C channels of one batch element and nothing is shared between waves: no workgroup
// barrier anywhere (the first version of this file had six in the forward and eight in the backward and spent its time in
// them: 15-21 us for the 4-10 MB of stages 2 / 3).  The wave reads every channel's TT-token piece with 16-byte vectors
// (coalesced along the tokens), keeps the tile in its LDS as f32 [channel][TT + 1], takes the per-token sums with lanes =
// (token, channel part) -- one sweep, shifted by the token's first channel so that sum and sum of squares do not cancel --
// and writes the normalised rows token-major (coalesced along the channels) in the dtype autocast would give them.  The
// backward does the same in the other direction: dy arrives token-major, dx leaves channel-major (the layout of the
// residual stream it is added to).  dweight / dbias: the wave leaves its tile's partial sums as one row of a workspace and a
// second small kernel adds the rows up (two launches, no same-address atomic traffic from thousands of waves; the only
// atomics are one per row group of the reduce kernel, 128 groups at most).
// TT is 32, 16 or 8: the largest that still gives a couple of thousand waves and fits the LDS.  Waves are numbered so that the
// tiles an XCD works on are neighbours in memory (pieces shorter than a 128-byte line meet in one L2).
// HBM-bound: forward reads x once and writes y once; backward reads dy and x once and writes dx once.
#include "common.cuh"

namespace vivim {

constexpr int kLnMaxC = 512;         // backward: 2 tiles x 512 channels x 9 floats = 37 KB of LDS per wave at TT = 8

template <typename T> struct LnVec {
    static constexpr int E = 16 / (int)sizeof(T);
    typedef typename Pack<T, 16>::type vec;
    union U { vec v; T e[16 / sizeof(T)]; };
};

// Which tile this wave (= workgroup) works on.  Workgroups are dealt round-robin over the 8 XCDs: XCD k takes the k-th eighth
// of the tiles, in order.
__device__ __forceinline__ bool ln_tile(int ntiles, int tpb, int TT, int& b, int& t0) {
    const int per = (ntiles + 7) / 8;
    const int tile = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if ((int)(blockIdx.x / 8) >= per || tile >= ntiles) return false;
    b = tile / tpb;
    t0 = (tile - b * tpb) * TT;
    return true;
}

// tile[c][t] <- x[b][c][t0 + t] as f32 (zero beyond the row's end); seqlen % E == 0 (host check).  Eight vectors per lane are
// in flight before the first LDS write.
template <typename T, int TT>
__device__ __forceinline__ void ln_load_cm(float* tile, const T* __restrict__ xb, int64_t c_stride, int C, int t0, int L, int lane) {
    constexpr int E = LnVec<T>::E;
    constexpr int VPR = TT / E;                        // 16-byte vectors per channel piece
    constexpr int PAD = TT + 1;
    const int nvec = C * VPR;
    for (int base = 0; base < nvec; base += 64 * 8) {
        typename LnVec<T>::U u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = base + i * 64 + lane;
            const int c = idx / VPR, v = idx - c * VPR;
            const int t = t0 + v * E;
            if (idx < nvec && t < L) u[i].v = *reinterpret_cast<const typename LnVec<T>::vec*>(xb + (int64_t)c * c_stride + t);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = base + i * 64 + lane;
            const int c = idx / VPR, v = idx - c * VPR;
            const bool ok = t0 + v * E < L;
            if (idx < nvec) {
#pragma unroll
                for (int e = 0; e < E; ++e) tile[c * PAD + v * E + e] = ok ? to_f32<T>(u[i].e[e]) : 0.0f;
            }
        }
    }
}

// sum over the 64 / TT channel parts of a token (lanes t, t + TT, t + 2 TT, ...)
template <int TT>
__device__ __forceinline__ float ln_parts_sum(float v) {
#pragma unroll
    for (int off = TT; off < kWave; off <<= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

template <typename TI, typename TO, int TT>
__global__ void __launch_bounds__(kWave) ln_cm_fwd_kernel(const vivim_layernorm_params p, const int ntiles, const int tpb) {
    extern __shared__ __attribute__((aligned(16))) float ln_smem[];
    constexpr int PAD = TT + 1, P = kWave / TT;
    const int C = p.channels, L = p.seqlen, lane = threadIdx.x;
    int b, t0;
    if (!ln_tile(ntiles, tpb, TT, b, t0)) return;
    float* tile = ln_smem;                             // [C][TT + 1]
    float* stat = tile + C * PAD;                      // [2][TT]: mean, rstd
    float* gb = stat + 2 * TT;                         // [2][C]: weight, bias (a global load per output element would serialise the store loop)
    for (int c = lane; c < C; c += kWave) {
        gb[c] = p.weight ? static_cast<const float*>(p.weight)[c] : 1.0f;
        gb[C + c] = p.bias ? static_cast<const float*>(p.bias)[c] : 0.0f;
    }
    ln_load_cm<TI, TT>(tile, static_cast<const TI*>(p.x) + (int64_t)b * p.x_batch_stride, p.x_c_stride, C, t0, L, lane);
    wave_lds_fence();
    const int t = lane % TT, part = lane / TT;
    const float pivot = tile[t];                       // channel 0 of the token
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 8
    for (int c = part; c < C; c += P) {
        const float d = tile[c * PAD + t] - pivot;
        s1 += d;
        s2 = fmaf(d, d, s2);
    }
    s1 = ln_parts_sum<TT>(s1);
    s2 = ln_parts_sum<TT>(s2);
    const float inv_c = 1.0f / (float)C;
    const float m = s1 * inv_c;
    const float mean = pivot + m;
    const float rstd = rsqrtf(fmaxf(s2 * inv_c - m * m, 0.0f) + p.eps);
    if (part == 0) {
        stat[t] = mean;
        stat[TT + t] = rstd;
        if (t0 + t < L) {
            static_cast<float*>(p.mean)[(int64_t)b * L + t0 + t] = mean;
            static_cast<float*>(p.rstd)[(int64_t)b * L + t0 + t] = rstd;
        }
    }
    wave_lds_fence();
    TO* __restrict__ yb = static_cast<TO*>(p.y) + (int64_t)b * p.y_batch_stride + (int64_t)t0 * p.y_token_stride;
    const int nt = min(TT, L - t0);
    int tt = 0, c = lane;                              // token-major: the wave writes 64 consecutive channels of one token
    while (c >= C) { c -= C; ++tt; }
    const int iters = (nt * C + kWave - 1) / kWave;
#pragma unroll 4
    for (int k = 0; k < iters; ++k) {
        if (tt < nt) {
            const float v = (tile[c * PAD + tt] - stat[tt]) * stat[TT + tt] * gb[c] + gb[C + c];
            yb[(int64_t)tt * p.y_token_stride + c] = from_f32<TO>(v);
        }
        c += kWave;
        while (c >= C) { c -= C; ++tt; }
    }
}

template <typename TI, typename TO, int TT>
__global__ void __launch_bounds__(kWave) ln_cm_bwd_kernel(const vivim_layernorm_params p, const int ntiles, const int tpb) {
    extern __shared__ __attribute__((aligned(16))) float ln_smem[];
    constexpr int PAD = TT + 1, P = kWave / TT;
    const int C = p.channels, L = p.seqlen, lane = threadIdx.x;
    int b, t0;
    if (!ln_tile(ntiles, tpb, TT, b, t0)) return;
    const int tile_id = b * tpb + t0 / TT;
    float* xt = ln_smem;                               // [C][TT + 1]: x
    float* gt = xt + C * PAD;                          // [C][TT + 1]: dy
    float* stat = gt + C * PAD;                        // [4][TT]: mean, rstd, S1 / C, S2 / C
    float* gam = stat + 4 * TT;                        // [C]: weight
    const int nt = min(TT, L - t0);
    for (int c = lane; c < C; c += kWave) gam[c] = p.weight ? static_cast<const float*>(p.weight)[c] : 1.0f;
    // dy: token-major reads (64 consecutive channels of a token per instruction), transposed into the tile; eight in flight
    {
        const TO* __restrict__ dyb = static_cast<const TO*>(p.dy) + (int64_t)b * p.y_batch_stride + (int64_t)t0 * p.y_token_stride;
        int tt = 0, c = lane;
        while (c >= C) { c -= C; ++tt; }
        while (tt < TT) {
            float v[8];
            int ct[8], tk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ct[i] = c; tk[i] = tt;
                v[i] = tt < nt ? to_f32<TO>(dyb[(int64_t)tt * p.y_token_stride + c]) : 0.0f;
                c += kWave;
                while (c >= C) { c -= C; ++tt; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (tk[i] < TT) gt[ct[i] * PAD + tk[i]] = v[i];
        }
    }
    ln_load_cm<TI, TT>(xt, static_cast<const TI*>(p.x) + (int64_t)b * p.x_batch_stride, p.x_c_stride, C, t0, L, lane);
    const int t = lane % TT, part = lane / TT;
    const bool tok = t0 + t < L;
    const float mean = tok ? static_cast<const float*>(p.mean)[(int64_t)b * L + t0 + t] : 0.0f;
    const float rstd = tok ? static_cast<const float*>(p.rstd)[(int64_t)b * L + t0 + t] : 0.0f;
    wave_lds_fence();
    // per-token S1 = sum_c dy gamma, S2 = sum_c dy gamma xhat
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 8
    for (int c = part; c < C; c += P) {
        const float g = gt[c * PAD + t] * gam[c];
        s1 += g;
        s2 = fmaf(g, (xt[c * PAD + t] - mean) * rstd, s2);
    }
    s1 = ln_parts_sum<TT>(s1);
    s2 = ln_parts_sum<TT>(s2);
    const float inv_c = 1.0f / (float)C;
    if (part == 0) {
        stat[t] = mean;
        stat[TT + t] = rstd;                           // 0 for a token beyond the row: its xhat and dx vanish
        stat[2 * TT + t] = s1 * inv_c;
        stat[3 * TT + t] = s2 * inv_c;
    }
    wave_lds_fence();
    // dweight, dbias partials of this tile: lane = channel, sum over the tile's tokens; row tile_id of the workspace
    if (p.workspace) {
        float* __restrict__ row = static_cast<float*>(p.workspace) + (int64_t)tile_id * 2 * C;
        for (int c = lane; c < C; c += kWave) {
            float dw = 0.0f, db = 0.0f;
#pragma unroll
            for (int k = 0; k < TT; ++k) {
                const float g = gt[c * PAD + k];
                dw = fmaf(g, (xt[c * PAD + k] - stat[k]) * stat[TT + k], dw);
                db += g;
            }
            row[c] = dw;
            row[C + c] = db;
        }
    }
    // dx = rstd * (dy gamma - S1 / C - xhat S2 / C), channel-major with 16-byte vectors along the tokens
    constexpr int E = LnVec<TI>::E;
    constexpr int VPR = TT / E;
    TI* __restrict__ dxb = static_cast<TI*>(p.dx) + (int64_t)b * p.dx_batch_stride;
    const int nvec = C * VPR;
    for (int idx = lane; idx < nvec; idx += kWave) {
        const int c = idx / VPR, v = idx - c * VPR;
        const int tg = t0 + v * E;
        if (tg >= L) continue;
        const float g = gam[c];
        typename LnVec<TI>::U u;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = v * E + e;
            const float r = stat[TT + k];
            const float xh = (xt[c * PAD + k] - stat[k]) * r;
            u.e[e] = from_f32<TI>(r * (gt[c * PAD + k] * g - stat[2 * TT + k] - xh * stat[3 * TT + k]));
        }
        *reinterpret_cast<typename LnVec<TI>::vec*>(dxb + (int64_t)c * p.dx_c_stride + tg) = u.v;
    }
}

// dweight[c] += sum over the workspace rows of column c, dbias[c] += ... of column C + c.  grid (ceil(2C / 64), row groups),
// 256 threads: lane = column, the four waves of a workgroup and the row groups interleave the rows (about eight rows per wave).
__global__ void __launch_bounds__(256) ln_reduce_kernel(const float* __restrict__ ws, int ntiles, int C, float* dweight, float* dbias) {
    __shared__ float part[4][kWave];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * kWave + lane;
    float s = 0.0f;
    if (col < 2 * C) {
        const int step = 4 * (int)gridDim.y;
        for (int r = blockIdx.y * 4 + wave; r < ntiles; r += step) s += ws[(int64_t)r * 2 * C + col];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < 2 * C) {
        s = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        if (col < C) { if (dweight) atomicAdd(dweight + col, s); }
        else if (dbias) atomicAdd(dbias + col - C, s);
    }
}

static size_t ln_fwd_smem(int C, int TT) { return ((size_t)C * (TT + 1) + 2 * TT + 2 * C) * sizeof(float); }
static size_t ln_bwd_smem(int C, int TT) { return ((size_t)2 * C * (TT + 1) + 4 * TT + C) * sizeof(float); }

// tokens per wave: the largest of 32 / 16 / 8 that leaves a couple of thousand waves and at most 64 KB of LDS per wave
// (VIVIM_LN_TT overrides, for the tests and for tuning)
int layernorm_tile_tokens(const vivim_layernorm_params& p) {
    const char* e = getenv("VIVIM_LN_TT");
    const int forced = e ? atoi(e) : 0;
    const int vec = p.itype == VIVIM_F32 ? 4 : 8;
    for (int TT : {32, 16, 8}) {
        if (TT % vec != 0 || ln_bwd_smem(p.channels, TT) > 65536) continue;
        if (forced == TT) return TT;
        if (!forced && (int64_t)p.batch * ((p.seqlen + TT - 1) / TT) >= (TT == 32 ? 4096 : 2048)) return TT;
    }
    return 8;
}
size_t layernorm_bwd_workspace_bytes(const vivim_layernorm_params& p) {
    const int TT = layernorm_tile_tokens(p);
    return (size_t)p.batch * ((p.seqlen + TT - 1) / TT) * 2 * p.channels * sizeof(float);
}

template <typename TI, typename TO, int TT>
static void ln_launch_tt(const vivim_layernorm_params& p, bool bwd, hipStream_t stream) {
    const int tpb = (p.seqlen + TT - 1) / TT, ntiles = p.batch * tpb;
    const dim3 grid((unsigned)((ntiles + 7) / 8 * 8)), block(kWave);
    if (!bwd) {
        hipLaunchKernelGGL((ln_cm_fwd_kernel<TI, TO, TT>), grid, block, ln_fwd_smem(p.channels, TT), stream, p, ntiles, tpb);
        return;
    }
    hipLaunchKernelGGL((ln_cm_bwd_kernel<TI, TO, TT>), grid, block, ln_bwd_smem(p.channels, TT), stream, p, ntiles, tpb);
    if (p.workspace && (p.dweight || p.dbias))
        hipLaunchKernelGGL(ln_reduce_kernel, dim3((2 * p.channels + kWave - 1) / kWave, std::min(128, std::max(1, ntiles / 32))), dim3(256), 0, stream,
                           static_cast<const float*>(p.workspace), ntiles, p.channels, static_cast<float*>(p.dweight),
                           static_cast<float*>(p.dbias));
}
template <typename TI, typename TO>
static void ln_launch(const vivim_layernorm_params& p, bool bwd, hipStream_t stream) {
    switch (layernorm_tile_tokens(p)) {
        case 32: ln_launch_tt<TI, TO, 32>(p, bwd, stream); break;
        case 16: ln_launch_tt<TI, TO, 16>(p, bwd, stream); break;
        default: ln_launch_tt<TI, TO, 8>(p, bwd, stream); break;
    }
}

bool layernorm_dispatch(const vivim_layernorm_params& p, bool bwd, hipStream_t stream) {
    if (p.channels > kLnMaxC) return false;
    // the output / incoming-gradient side is f32 (what autocast makes of layer_norm) or the input's own type
    if (p.otype == VIVIM_F32) {
        switch (p.itype) {
            case VIVIM_F32: ln_launch<float, float>(p, bwd, stream); return true;
            case VIVIM_F16: ln_launch<f16_t, float>(p, bwd, stream); return true;
            case VIVIM_BF16: ln_launch<bf16_t, float>(p, bwd, stream); return true;
        }
    } else if (p.otype == p.itype) {
        switch (p.itype) {
            case VIVIM_F16: ln_launch<f16_t, f16_t>(p, bwd, stream); return true;
            case VIVIM_BF16: ln_launch<bf16_t, bf16_t>(p, bwd, stream); return true;
        }
    }
    return false;
}

}  // namespace vivim
