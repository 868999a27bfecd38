// layernorm.hip -- LayerNorm over the channel axis of a CHANNEL-major token tensor, forward and backward (SURVEY.md 8f row 4:
// the two nn.LayerNorms around the Mamba call, modeling/vivim.py:155-156).
//
// MambaLayer holds its activations as (B, C, nf*H*W) and hands the norm the transposed VIEW (B, L, C) with strides
// (C*L, 1, L) (modeling/vivim.py:151-155).  The ATen path first makes that view contiguous (one full copy kernel) and then
// runs its row kernel; here the transpose IS the kernel: a workgroup owns 32 tokens of one batch element, reads every
// channel's 32-token piece with 16-byte vectors (coalesced along the tokens), keeps the tile in LDS as f32 [channel][33],
// takes mean and variance per token from LDS (two sweeps: sum, then squared deviations -- no E[x^2] - mean^2 cancellation),
// and writes the normalised rows token-major (coalesced along the channels) in the dtype autocast would give them.
// The backward does the same in the other direction: dy arrives token-major, dx leaves channel-major (the layout of the
// residual stream it is added to), dweight / dbias leave as one fp32 atomic per (workgroup, channel).
// HBM-bound: forward reads x once and writes y once; backward reads dy and x once and writes dx once.
#include "common.cuh"

namespace vivim {

constexpr int kLnTT = 32;            // tokens per workgroup
constexpr int kLnPad = kLnTT + 1;    // LDS row stride in floats: lane = channel reads hit 32 different banks
constexpr int kLnMaxC = 512;         // 2 tiles x 512 channels x 33 floats = 135 KB of LDS in the backward

template <typename T> struct LnVec {
    static constexpr int E = 16 / (int)sizeof(T);
    typedef typename Pack<T, 16>::type vec;
    union U { vec v; T e[16 / sizeof(T)]; };
};

// tile[c][t] <- x[b][c][t0 + t] as f32 (zero beyond the row's end); seqlen % E == 0 (host check)
template <typename T>
__device__ __forceinline__ void ln_load_cm(float* tile, const T* __restrict__ xb, int64_t c_stride, int C, int t0, int L) {
    constexpr int E = LnVec<T>::E;
    constexpr int VPR = kLnTT / E;                     // 16-byte vectors per channel piece
    for (int idx = threadIdx.x; idx < C * VPR; idx += blockDim.x) {
        const int c = idx / VPR, v = idx - c * VPR;
        const int t = t0 + v * E;
        typename LnVec<T>::U u;
        if (t < L) u.v = *reinterpret_cast<const typename LnVec<T>::vec*>(xb + (int64_t)c * c_stride + t);
#pragma unroll
        for (int e = 0; e < E; ++e) tile[c * kLnPad + v * E + e] = t < L ? to_f32<T>(u.e[e]) : 0.0f;
    }
}

// per-token sum over the channels of f(c, t): thread (t = tid & 31, part = tid >> 5) sums its channels, the 8 parts meet in LDS
template <typename F>
__device__ __forceinline__ float ln_token_sum(float* red, int C, F&& f) {
    const int t = threadIdx.x & 31, part = threadIdx.x >> 5, nparts = blockDim.x >> 5;
    float s = 0.0f;
    for (int c = part; c < C; c += nparts) s += f(c, t);
    red[part * kLnTT + t] = s;
    __syncthreads();
    float tot = 0.0f;
    for (int q = 0; q < nparts; ++q) tot += red[q * kLnTT + t];
    __syncthreads();
    return tot;                                        // every thread with the same t holds the token's total
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) ln_cm_fwd_kernel(const vivim_layernorm_params p) {
    extern __shared__ __attribute__((aligned(16))) float ln_smem[];
    const int C = p.channels, L = p.seqlen;
    float* tile = ln_smem;                             // [C][33]
    float* red = tile + C * kLnPad;                    // [8][32]
    float* gb = red + 8 * kLnTT;                       // [2][C]: weight, bias
    const int b = blockIdx.y, t0 = blockIdx.x * kLnTT;
    const TI* __restrict__ xb = static_cast<const TI*>(p.x) + (int64_t)b * p.x_batch_stride;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        gb[c] = p.weight ? static_cast<const float*>(p.weight)[c] : 1.0f;
        gb[C + c] = p.bias ? static_cast<const float*>(p.bias)[c] : 0.0f;
    }
    ln_load_cm<TI>(tile, xb, p.x_c_stride, C, t0, L);
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    const float mean = ln_token_sum(red, C, [&](int c, int t) { return tile[c * kLnPad + t]; }) * inv_c;
    const float var = ln_token_sum(red, C, [&](int c, int t) { const float d = tile[c * kLnPad + t] - mean; return d * d; }) * inv_c;
    const float rstd = rsqrtf(var + p.eps);
    const int tt = threadIdx.x & 31;
    if (threadIdx.x < kLnTT && t0 + tt < L) {
        static_cast<float*>(p.mean)[(int64_t)b * L + t0 + tt] = mean;
        static_cast<float*>(p.rstd)[(int64_t)b * L + t0 + tt] = rstd;
    }
    red[tt] = mean; red[kLnTT + tt] = rstd;            // (all threads of a token write the same values)
    __syncthreads();
    TO* __restrict__ yb = static_cast<TO*>(p.y) + (int64_t)b * p.y_batch_stride;
    const int nt = min(kLnTT, L - t0);
    for (int idx = threadIdx.x; idx < nt * C; idx += blockDim.x) {     // token-major: a wave writes consecutive channels of one token
        const int t = idx / C, c = idx - t * C;
        const float v = (tile[c * kLnPad + t] - red[t]) * red[kLnTT + t] * gb[c] + gb[C + c];
        yb[(int64_t)(t0 + t) * p.y_token_stride + c] = from_f32<TO>(v);
    }
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) ln_cm_bwd_kernel(const vivim_layernorm_params p) {
    extern __shared__ __attribute__((aligned(16))) float ln_smem[];
    const int C = p.channels, L = p.seqlen;
    float* xh = ln_smem;                               // [C][33]: x normalised
    float* gy = xh + C * kLnPad;                       // [C][33]: dy
    float* red = gy + C * kLnPad;                      // [8][32]
    float* gam = red + 8 * kLnTT;                      // [C]
    float* stat = gam + C;                             // [2][32]: mean, rstd; later S1, S2
    const int b = blockIdx.y, t0 = blockIdx.x * kLnTT;
    const int nt = min(kLnTT, L - t0);
    const TI* __restrict__ xb = static_cast<const TI*>(p.x) + (int64_t)b * p.x_batch_stride;
    for (int c = threadIdx.x; c < C; c += blockDim.x) gam[c] = p.weight ? static_cast<const float*>(p.weight)[c] : 1.0f;
    if (threadIdx.x < kLnTT) {
        const bool ok = threadIdx.x < nt;
        stat[threadIdx.x] = ok ? static_cast<const float*>(p.mean)[(int64_t)b * L + t0 + threadIdx.x] : 0.0f;
        stat[kLnTT + threadIdx.x] = ok ? static_cast<const float*>(p.rstd)[(int64_t)b * L + t0 + threadIdx.x] : 0.0f;
    }
    ln_load_cm<TI>(xh, xb, p.x_c_stride, C, t0, L);
    const TO* __restrict__ dyb = static_cast<const TO*>(p.dy) + (int64_t)b * p.y_batch_stride;
    for (int idx = threadIdx.x; idx < kLnTT * C; idx += blockDim.x) {   // token-major reads, transposed into the tile
        const int t = idx / C, c = idx - t * C;
        gy[c * kLnPad + t] = t < nt ? to_f32<TO>(dyb[(int64_t)(t0 + t) * p.y_token_stride + c]) : 0.0f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < kLnTT * C; idx += blockDim.x) {   // x -> (x - mean) * rstd, in place (lane = token: conflict-free)
        const int c = idx / kLnTT, t = idx - c * kLnTT;
        xh[c * kLnPad + t] = t < nt ? (xh[c * kLnPad + t] - stat[t]) * stat[kLnTT + t] : 0.0f;
    }
    __syncthreads();
    // dweight, dbias: lane = channel, sum over the tile's tokens; one atomic per (workgroup, channel)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float dw = 0.0f, db = 0.0f;
#pragma unroll 8
        for (int t = 0; t < kLnTT; ++t) { const float g = gy[c * kLnPad + t]; dw = fmaf(g, xh[c * kLnPad + t], dw); db += g; }
        if (p.dweight) atomicAdd(static_cast<float*>(p.dweight) + c, dw);
        if (p.dbias) atomicAdd(static_cast<float*>(p.dbias) + c, db);
    }
    // per-token S1 = sum_c dy gamma, S2 = sum_c dy gamma xhat
    const float s1 = ln_token_sum(red, C, [&](int c, int t) { return gy[c * kLnPad + t] * gam[c]; });
    const float s2 = ln_token_sum(red, C, [&](int c, int t) { return gy[c * kLnPad + t] * gam[c] * xh[c * kLnPad + t]; });
    const int tt = threadIdx.x & 31;
    const float rstd_t = stat[kLnTT + tt];
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    stat[tt] = s1 * inv_c; stat[kLnTT + tt] = s2 * inv_c; red[tt] = rstd_t;
    __syncthreads();
    // dx = rstd * (dy gamma - S1 / C - xhat S2 / C), channel-major with 16-byte vectors along the tokens
    constexpr int E = LnVec<TI>::E;
    constexpr int VPR = kLnTT / E;
    TI* __restrict__ dxb = static_cast<TI*>(p.dx) + (int64_t)b * p.dx_batch_stride;
    for (int idx = threadIdx.x; idx < C * VPR; idx += blockDim.x) {
        const int c = idx / VPR, v = idx - c * VPR;
        const int t = t0 + v * E;
        if (t >= L) continue;
        typename LnVec<TI>::U u;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int tl = v * E + e;
            const float d = red[tl] * (gy[c * kLnPad + tl] * gam[c] - stat[tl] - xh[c * kLnPad + tl] * stat[kLnTT + tl]);
            u.e[e] = from_f32<TI>(d);
        }
        *reinterpret_cast<typename LnVec<TI>::vec*>(dxb + (int64_t)c * p.dx_c_stride + t) = u.v;
    }
}

static size_t ln_fwd_smem(int C) { return ((size_t)C * kLnPad + 8 * kLnTT + 2 * C) * sizeof(float); }
static size_t ln_bwd_smem(int C) { return ((size_t)2 * C * kLnPad + 8 * kLnTT + C + 2 * kLnTT) * sizeof(float); }

template <typename TI, typename TO>
static void ln_launch(const vivim_layernorm_params& p, bool bwd, hipStream_t stream) {
    const dim3 grid((p.seqlen + kLnTT - 1) / kLnTT, p.batch), block(256);
    const size_t smem = bwd ? ln_bwd_smem(p.channels) : ln_fwd_smem(p.channels);
    auto kernel = bwd ? ln_cm_bwd_kernel<TI, TO> : ln_cm_fwd_kernel<TI, TO>;
    if (smem > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, p);
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
