// conv1d.hip -- depthwise causal conv1d (+ fused SiLU), channel-first, forward and backward, gfx950.
//
// What it computes is what the reference kernels compute (causal-conv1d/csrc/causal_conv1d_fwd.cu:39-130,
// causal_conv1d_bwd.cu:46-240); how is different: the reference walks one (batch, channel) row per
// 128-thread block, chunk after chunk, passing halos through shared memory with three barriers per
// chunk.  Here every wave owns an independent 64*E-token tile of one row (grid = tiles x dim x batch, so
// a 20480-token row is ten 256-thread blocks in flight instead of one), halos move between neighbouring
// lanes with wave shuffles, the two lanes at the wave edges fetch theirs with guarded scalar loads, and
// there is no barrier in the forward at all.  Pure stream: 2*s bytes/token forward, 3*s backward.
#include "common.cuh"

namespace vivim {

constexpr int kConvThreads = 256;

// Which (channel, tile) a wave works on.  Long rows: the workgroup's four waves take four consecutive 64*E-token tiles of
// the row blockIdx.y.  PACK (rows shorter than four tiles -- Vivim's stages 2 and 3: 1280 and 320 tokens): the waves of
// the launch are numbered through (channel, tile) pairs, so a 320-token row is one wave instead of a 256-thread
// workgroup with 216 idle lanes; a wave past the last channel has nothing to do.
template <bool PACK, int E>
__device__ __forceinline__ bool conv_tile(int L, int dim, int& c, int& t0) {
    const int lane = threadIdx.x & 63;
    if (!PACK) {
        c = blockIdx.y;
        t0 = (blockIdx.x * kConvThreads + threadIdx.x) * E;
        return true;
    }
    const int tpr = (L + kWave * E - 1) / (kWave * E);
    const int wid = blockIdx.x * (kConvThreads / kWave) + (threadIdx.x >> 6);
    c = __builtin_amdgcn_readfirstlane(wid / tpr);
    t0 = ((wid - c * tpr) * kWave + lane) * E;
    return c < dim;
}

// taps are right-aligned into 4 slots so one code path serves width 2..4:
//   out[t] = bias + sum_{j<4} w4[j] * x[t - 3 + j],   w4[j] = weight[j - (4 - W)] (0 for j < 4 - W)
template <typename WT>
__device__ __forceinline__ void load_taps(const WT* w, int64_t wstride, int width, float (&w4)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int src = j - (4 - width);
        w4[j] = src >= 0 ? to_f32<WT>(w[src * wstride]) : 0.0f;
    }
}

template <typename T, typename WT, int E, bool PACK>
__global__ void __launch_bounds__(kConvThreads) conv1d_fwd_kernel(const vivim_conv_fwd_params p) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.z;
    const int L = p.seqlen;
    int c, t0;
    if (!conv_tile<PACK, E>(L, p.dim, c, t0)) return;           // whole waves only; no barrier in this kernel
    const T* __restrict__ x = static_cast<const T*>(p.x) + b * p.x_batch_stride + c * p.x_c_stride;
    T* __restrict__ out = static_cast<T*>(p.out) + b * p.out_batch_stride + c * p.out_c_stride;

    float w4[4];
    load_taps<WT>(static_cast<const WT*>(p.weight) + c * p.weight_c_stride, p.weight_width_stride, p.width, w4);
    const float bias = p.bias ? to_f32<WT>(static_cast<const WT*>(p.bias)[c]) : 0.0f;

    float xx[E + 3];   // x[t0-3 .. t0+E)
    {
        float xv[E];
        load_k<T, E>(x + t0, L - t0, xv);
#pragma unroll
        for (int k = 0; k < E; ++k) xx[3 + k] = xv[k];
    }
    // left halo: previous lane's last three tokens; lane 0 of each wave reads them itself
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float h = __shfl_up(xx[E + j], 1, kWave);
        if (lane == 0) {
            const int t = t0 - 3 + j;
            h = (t >= 0 && t < L) ? to_f32<T>(x[t]) : 0.0f;
        }
        xx[j] = h;
    }
    float o[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        float acc = bias;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(w4[j], xx[k + j], acc);
        o[k] = p.silu_activation ? acc * sigmoidf_fast(acc) : acc;
    }
    store_k<T, E>(out + t0, L - t0, o);
}

// Backward.  Per lane: tokens [t0, t0+E).  It needs x on [t0-3, t0+E+3) and dout on [t0, t0+E+3):
//   g[t]   = dout[t] * silu'(pre[t])            (pre recomputed from x, bwd.cu:163-175)
//   dx[s]  = sum_j w4[j] * g[s + 3 - j]
//   dw4[j] = sum_t g[t] * x[t - 3 + j],  dbias = sum_t g[t]     (own tokens only, then reduced)
template <typename T, typename WT, int E, bool PACK>
__global__ void __launch_bounds__(kConvThreads) conv1d_bwd_kernel(const vivim_conv_bwd_params p) {
    const vivim_conv_fwd_params& f = p.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int L = f.seqlen;
    int c, t0;
    if (!conv_tile<PACK, E>(L, f.dim, c, t0)) return;           // PACK only (whole waves; the PACK path has no barrier)
    const T* __restrict__ x = static_cast<const T*>(f.x) + b * f.x_batch_stride + c * f.x_c_stride;
    const T* __restrict__ dout = static_cast<const T*>(p.dout) + b * p.dout_batch_stride + c * p.dout_c_stride;
    T* __restrict__ dx = static_cast<T*>(p.dx) + b * p.dx_batch_stride + c * p.dx_c_stride;

    float w4[4];
    load_taps<WT>(static_cast<const WT*>(f.weight) + c * f.weight_c_stride, f.weight_width_stride, f.width, w4);
    const float bias = f.bias ? to_f32<WT>(static_cast<const WT*>(f.bias)[c]) : 0.0f;

    float X[E + 6];    // x[t0-3 .. t0+E+3)
    float dO[E + 3];   // dout[t0 .. t0+E+3)
    {
        float xv[E], dv[E];
        load_k<T, E>(x + t0, L - t0, xv);
        load_k<T, E>(dout + t0, L - t0, dv);
#pragma unroll
        for (int k = 0; k < E; ++k) { X[3 + k] = xv[k]; dO[k] = dv[k]; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float hl = __shfl_up(X[E + j], 1, kWave);          // previous lane's tokens E-3..E-1
        float hr = __shfl_down(X[3 + j], 1, kWave);        // next lane's tokens 0..2
        float dr = __shfl_down(dO[j], 1, kWave);
        if (lane == 0) {
            const int t = t0 - 3 + j;
            hl = (t >= 0 && t < L) ? to_f32<T>(x[t]) : 0.0f;
        }
        if (lane == kWave - 1) {
            const int t = t0 + E + j;
            hr = t < L ? to_f32<T>(x[t]) : 0.0f;
            dr = t < L ? to_f32<T>(dout[t]) : 0.0f;
        }
        X[j] = hl;
        X[E + 3 + j] = hr;
        dO[E + j] = dr;
    }
    float g[E + 3];
#pragma unroll
    for (int j = 0; j < E + 3; ++j) {
        float gj = dO[j];
        if (f.silu_activation) {
            float pre = bias;
#pragma unroll
            for (int i = 0; i < 4; ++i) pre = fmaf(w4[i], X[j + i], pre);
            const float sg = sigmoidf_fast(pre);
            gj *= sg * (1.0f + pre * (1.0f - sg));
        }
        g[j] = gj;      // tokens >= L carry dout == 0, hence g == 0
    }
    float o[E];
#pragma unroll
    for (int k = 0; k < E; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf(w4[i], g[k + 3 - i], acc);
        o[k] = acc;
    }
    store_k<T, E>(dx + t0, L - t0, o);

    float red[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // dw4[0..3], dbias
#pragma unroll
    for (int k = 0; k < E; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[i] = fmaf(g[k], X[k + i], red[i]);
        red[4] += g[k];
    }
    if (PACK) {                                   // the waves of a workgroup hold different channels: one wave, five atomics
        float mine = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float s = wave_sum(red[i]);
            if (lane == i) mine = s;
        }
        if (lane < 4) {
            const int src = lane - (4 - f.width);
            if (src >= 0)
                atomicAdd(static_cast<float*>(p.dweight) + c * p.dweight_c_stride + src * p.dweight_width_stride, mine);
        } else if (lane == 4 && p.dbias) {
            atomicAdd(static_cast<float*>(p.dbias) + c, mine);
        }
        return;
    }
    __shared__ float part[kConvThreads / kWave][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float s = wave_sum(red[i]);
        if (lane == 0) part[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kConvThreads / kWave; ++w) s += part[w][threadIdx.x];
        if (threadIdx.x < 4) {
            const int src = (int)threadIdx.x - (4 - f.width);
            if (src >= 0)
                atomicAdd(static_cast<float*>(p.dweight) + c * p.dweight_c_stride + src * p.dweight_width_stride, s);
        } else if (p.dbias) {
            atomicAdd(static_cast<float*>(p.dbias) + c, s);
        }
    }
}

template <typename T, typename WT>
static void launch_conv_fwd(const vivim_conv_fwd_params& p, hipStream_t stream) {
    constexpr int E = 16 / sizeof(T);   // one 16-byte access per lane
    const int tpr = (p.seqlen + kWave * E - 1) / (kWave * E);          // 64*E-token tiles per row
    if (tpr < kConvThreads / kWave) {                                   // short rows: waves numbered through (channel, tile)
        const int waves = p.dim * tpr, wpb = kConvThreads / kWave;
        hipLaunchKernelGGL((conv1d_fwd_kernel<T, WT, E, true>), dim3((waves + wpb - 1) / wpb, 1, p.batch), dim3(kConvThreads), 0, stream, p);
        return;
    }
    dim3 grid((p.seqlen + kConvThreads * E - 1) / (kConvThreads * E), p.dim, p.batch);
    hipLaunchKernelGGL((conv1d_fwd_kernel<T, WT, E, false>), grid, dim3(kConvThreads), 0, stream, p);
}
template <typename T, typename WT>
static void launch_conv_bwd(const vivim_conv_bwd_params& p, hipStream_t stream) {
    constexpr int E = 16 / sizeof(T);
    const int tpr = (p.f.seqlen + kWave * E - 1) / (kWave * E);
    if (tpr < kConvThreads / kWave) {
        const int waves = p.f.dim * tpr, wpb = kConvThreads / kWave;
        hipLaunchKernelGGL((conv1d_bwd_kernel<T, WT, E, true>), dim3((waves + wpb - 1) / wpb, 1, p.f.batch), dim3(kConvThreads), 0, stream, p);
        return;
    }
    dim3 grid((p.f.seqlen + kConvThreads * E - 1) / (kConvThreads * E), p.f.dim, p.f.batch);
    hipLaunchKernelGGL((conv1d_bwd_kernel<T, WT, E, false>), grid, dim3(kConvThreads), 0, stream, p);
}

template <typename T>
static bool dispatch_w_fwd(const vivim_conv_fwd_params& p, hipStream_t s) {
    switch (p.wtype) {
        case VIVIM_F32: launch_conv_fwd<T, float>(p, s); return true;
        case VIVIM_F16: launch_conv_fwd<T, f16_t>(p, s); return true;
        case VIVIM_BF16: launch_conv_fwd<T, bf16_t>(p, s); return true;
    }
    return false;
}
template <typename T>
static bool dispatch_w_bwd(const vivim_conv_bwd_params& p, hipStream_t s) {
    switch (p.f.wtype) {
        case VIVIM_F32: launch_conv_bwd<T, float>(p, s); return true;
        case VIVIM_F16: launch_conv_bwd<T, f16_t>(p, s); return true;
        case VIVIM_BF16: launch_conv_bwd<T, bf16_t>(p, s); return true;
    }
    return false;
}

bool conv_fwd_dispatch(const vivim_conv_fwd_params& p, hipStream_t s) {
    switch (p.itype) {
        case VIVIM_F32: return dispatch_w_fwd<float>(p, s);
        case VIVIM_F16: return dispatch_w_fwd<f16_t>(p, s);
        case VIVIM_BF16: return dispatch_w_fwd<bf16_t>(p, s);
    }
    return false;
}
bool conv_bwd_dispatch(const vivim_conv_bwd_params& p, hipStream_t s) {
    switch (p.f.itype) {
        case VIVIM_F32: return dispatch_w_bwd<float>(p, s);
        case VIVIM_F16: return dispatch_w_bwd<f16_t>(p, s);
        case VIVIM_BF16: return dispatch_w_bwd<bf16_t>(p, s);
    }
    return false;
}

}  // namespace vivim
