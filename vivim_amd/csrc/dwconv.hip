// dwconv.hip -- depthwise 3x3 (2-D) / 3x3x3 (3-D) convolution on TOKEN-MAJOR tensors, stride 1, zero "same"
// padding, forward + input gradient (same kernel, flipped taps) + weight/bias gradient.  gfx950.
//
// Where it sits: the depthwise Conv3d of MambaLayer's Mlp, modeling/vivim.py:57-68,
// which the reference runs through ATen (x.transpose(1, 2).view(B, C, nf, H, W) -> nn.Conv3d(groups=C) ->
// flatten(2).transpose(1, 2)).  On MI355X that call lands on MIOpen's `naive_conv_*` kernels (0.8 ms each)
// and a 1.5-4 ms CK weight-gradient kernel: together 60 % of the Vivim train step (profiles/r01_v1_*).
//
// Layout: x, y are (batch, tokens = D*H*W, channels) with channels contiguous -- exactly what the Mlp already
// holds, so no transpose pass exists.  A thread owns 16 bytes of consecutive channels (8 bf16 / 4 fp32) at
// TW = 4 consecutive w positions: every load is a 16-byte vector, 8 neighbouring lanes cover 128 contiguous
// bytes of one token, each loaded input vector feeds up to 3 of the 4 outputs, taps live in LDS as fp32.
// Memory-bound: 2*s bytes per element forward (the 27x neighbour re-reads are L1/L2 hits).
#include "common.cuh"

namespace vivim {

constexpr int kDwTW = 4;           // w positions per thread
constexpr int kDwThreads = 256;

// erf-form GELU (torch.nn.GELU(), approximate = 'none') and its derivative, on the fp32 accumulator
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float v) {
    return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * 0.39894228040143268f * __expf(-0.5f * v * v);
}

// ACT: 0 plain, 1 y = gelu(conv), 2 y = aux * gelu'(conv) (include/vivim_hip.h: vivim_dwconv_params.act)
template <typename T, int KD, bool FLIP, int ACT = 0>
__global__ void __launch_bounds__(kDwThreads) dwconv_fwd_kernel(const vivim_dwconv_params p) {
    constexpr int CV = 16 / (int)sizeof(T);           // channels per thread
    constexpr int CB = 8 * CV;                        // channels per block (8 lanes x CV = 128 B of one token)
    constexpr int TAPS = KD * 9;
    constexpr int TW = kDwTW;
    __shared__ __attribute__((aligned(16))) float sw[TAPS * CB];
    const int tid = threadIdx.x;
    const int C = p.channels, D = p.depth, H = p.height, W = p.width;
    const int c0 = blockIdx.y * CB, b = blockIdx.z;
    const float* __restrict__ wt = static_cast<const float*>(p.wt);
    for (int i = tid; i < TAPS * CB; i += kDwThreads) {
        const int tap = i / CB, cc = i - tap * CB;
        sw[i] = (c0 + cc < C) ? wt[(int64_t)(FLIP ? TAPS - 1 - tap : tap) * C + c0 + cc] : 0.0f;
    }
    __syncthreads();
    const int cv = tid & 7, sp = tid >> 3;
    const int wtiles = (W + TW - 1) / TW;
    const int tile = blockIdx.x * (kDwThreads / 8) + sp;
    const int c = c0 + cv * CV;
    if (tile >= D * H * wtiles || c >= C) return;
    const int w0 = (tile % wtiles) * TW;
    const int h = (tile / wtiles) % H;
    const int d = tile / (wtiles * H);
    const T* __restrict__ x = static_cast<const T*>(p.x) + b * p.x_batch_stride + c;
    T* __restrict__ y = static_cast<T*>(p.y) + b * p.y_batch_stride + c;

    float acc[TW][CV];
#pragma unroll
    for (int j = 0; j < TW; ++j)
#pragma unroll
        for (int v = 0; v < CV; ++v) acc[j][v] = p.bias ? static_cast<const float*>(p.bias)[c + v] : 0.0f;
#pragma unroll
    for (int kd = 0; kd < KD; ++kd) {
        const int dd = d + kd - KD / 2;
        if (dd < 0 || dd >= D) continue;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
            if (hh < 0 || hh >= H) continue;
            const int64_t row = ((int64_t)dd * H + hh) * W;
#pragma unroll
            for (int iw = -1; iw <= TW; ++iw) {
                const int ww = w0 + iw;
                float xv[CV];
                unpack(load_vec<T, CV>(x + (row + ww) * p.x_token_stride, ww >= 0 && ww < W), xv);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int kw = iw - j + 1;
                    if (kw < 0 || kw > 2) continue;
                    const float* wp = sw + ((kd * 3 + kh) * 3 + kw) * CB + cv * CV;
#pragma unroll
                    for (int v = 0; v < CV; ++v) acc[j][v] = fmaf(wp[v], xv[v], acc[j][v]);
                }
            }
        }
    }
    const int64_t tok = ((int64_t)d * H + h) * W + w0;
    if (ACT == 1) {
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int v = 0; v < CV; ++v) acc[j][v] = gelu_erf(acc[j][v]);
    } else if (ACT == 2) {
        const T* __restrict__ aux = static_cast<const T*>(p.aux) + b * p.aux_batch_stride + c;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            float gv[CV];
            unpack(load_vec<T, CV>(aux + (tok + j) * p.aux_token_stride, w0 + j < W), gv);
#pragma unroll
            for (int v = 0; v < CV; ++v) acc[j][v] = gv[v] * gelu_erf_grad(acc[j][v]);
        }
    }
#pragma unroll
    for (int j = 0; j < TW; ++j) store_vec<T, CV>(y + (tok + j) * p.y_token_stride, w0 + j < W, acc[j]);
}

// dwt[tap][c] += sum_{b, token} x[token + off(tap)][c] * dy[token][c];  dbias[c] += sum dy[token][c].
// A lane owns 2 consecutive channels (a wave: 128 contiguous channels of a token) and keeps all TAPS x 2
// partial sums in registers while it walks its share of the (d, h, w-tile) tiles; the 4 waves of a block are
// summed through LDS, then one fp32 atomic per (tap, channel) and block.
template <typename T, int KD>
__global__ void __launch_bounds__(kDwThreads) dwconv_wgrad_kernel(const vivim_dwconv_wgrad_params p, const int tiles_per_wave) {
    constexpr int CP = 2;
    constexpr int TAPS = KD * 9;
    constexpr int TW = kDwTW;
    __shared__ float red[(kDwThreads / kWave) * (TAPS + 1) * CP * kWave];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.channels, D = p.depth, H = p.height, W = p.width;
    const int c = blockIdx.y * (kWave * CP) + lane * CP, b = blockIdx.z;
    const bool cok = c < C;                           // C % 2 == 0 guaranteed by the host
    const T* __restrict__ x = static_cast<const T*>(p.x) + b * p.x_batch_stride + c;
    const T* __restrict__ dy = static_cast<const T*>(p.dy) + b * p.dy_batch_stride + c;
    const int wtiles = (W + TW - 1) / TW;
    const int ntiles = D * H * wtiles;
    const int t_lo = (blockIdx.x * (kDwThreads / kWave) + wave) * tiles_per_wave;
    const int t_hi = min(ntiles, t_lo + tiles_per_wave);

    float acc[TAPS][CP], db[CP] = {0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < TAPS; ++t) { acc[t][0] = 0.0f; acc[t][1] = 0.0f; }
    // Every load of a tile is issued before its first use, and unconditionally (from `safe` when its token lies outside the
    // volume or the lane's channels outside the tensor, the value then replaced by zero): a load under a branch makes hipcc
    // wait for ALL outstanding loads at every later use, which turned the nine (kd, kh) rows of a tile into nine memory round
    // trips in a row (round 3: 103 -> 87 us at stage 0 of configs[1], 41 -> 30 us at stage 3).
    const T* __restrict__ safe_x = static_cast<const T*>(p.x) + b * p.x_batch_stride;       // channel 0 of token 0: always there
    const T* __restrict__ safe_dy = static_cast<const T*>(p.dy) + b * p.dy_batch_stride;
    for (int tile = t_lo; tile < t_hi; ++tile) {
        const int w0 = (tile % wtiles) * TW;
        const int h = (tile / wtiles) % H;
        const int d = tile / (wtiles * H);
        const int64_t tok = ((int64_t)d * H + h) * W + w0;
        RawK<T, CP> graw[TW], xraw[KD * 3][TW + 2];
#pragma unroll
        for (int j = 0; j < TW; ++j)
            graw[j] = load_vec_always<T, CP>(dy + (tok + j) * p.dy_token_stride, cok && w0 + j < W, safe_dy);
#pragma unroll
        for (int kd = 0; kd < KD; ++kd) {
            const int dd = d + kd - KD / 2;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int hh = h + kh - 1;
                const bool rowok = cok && dd >= 0 && dd < D && hh >= 0 && hh < H;
                const int64_t row = ((int64_t)dd * H + hh) * W;
#pragma unroll
                for (int iw = -1; iw <= TW; ++iw) {
                    const int ww = w0 + iw;
                    xraw[kd * 3 + kh][iw + 1] = load_vec_always<T, CP>(x + (row + ww) * p.x_token_stride, rowok && ww >= 0 && ww < W, safe_x);
                }
            }
        }
        float g[TW][CP];
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            unpack(graw[j], g[j]);
            db[0] += g[j][0];
            db[1] += g[j][1];
        }
#pragma unroll
        for (int r = 0; r < KD * 3; ++r) {
#pragma unroll
            for (int iw = -1; iw <= TW; ++iw) {
                float xv[CP];
                unpack(xraw[r][iw + 1], xv);
#pragma unroll
                for (int j = 0; j < TW; ++j) {
                    const int kw = iw - j + 1;
                    if (kw < 0 || kw > 2) continue;
                    const int tap = r * 3 + kw;
                    acc[tap][0] = fmaf(xv[0], g[j][0], acc[tap][0]);
                    acc[tap][1] = fmaf(xv[1], g[j][1], acc[tap][1]);
                }
            }
        }
    }
    // block reduction: [wave][tap or bias][cp][lane]
    float* mine = red + wave * (TAPS + 1) * CP * kWave;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        mine[(t * CP + 0) * kWave + lane] = acc[t][0];
        mine[(t * CP + 1) * kWave + lane] = acc[t][1];
    }
    mine[(TAPS * CP + 0) * kWave + lane] = db[0];
    mine[(TAPS * CP + 1) * kWave + lane] = db[1];
    __syncthreads();
    float* dwt = static_cast<float*>(p.dwt);
    for (int i = tid; i < (TAPS + 1) * CP * kWave; i += kDwThreads) {
        float s = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kDwThreads / kWave; ++wv) s += red[wv * (TAPS + 1) * CP * kWave + i];
        const int ln = i & 63, cp = (i >> 6) & 1, t = i >> 7;
        const int cc = blockIdx.y * (kWave * CP) + ln * CP + cp;
        if (cc < C) {
            if (t < TAPS) atomicAdd(dwt + (int64_t)t * C + cc, s);
            else if (p.dbias) atomicAdd(static_cast<float*>(p.dbias) + cc, s);
        }
    }
}

template <typename T>
static bool dw_fwd(const vivim_dwconv_params& p, hipStream_t s) {
    constexpr int CV = 16 / (int)sizeof(T);
    const int wtiles = (p.width + kDwTW - 1) / kDwTW;
    const int ntiles = p.depth * p.height * wtiles;
    dim3 grid((ntiles + kDwThreads / 8 - 1) / (kDwThreads / 8), (p.channels + 8 * CV - 1) / (8 * CV), p.batch);
    if (p.act == 1 || p.act == 2) {
        if (p.kd == 3) {
            if (p.act == 1) hipLaunchKernelGGL((dwconv_fwd_kernel<T, 3, false, 1>), grid, dim3(kDwThreads), 0, s, p);
            else            hipLaunchKernelGGL((dwconv_fwd_kernel<T, 3, false, 2>), grid, dim3(kDwThreads), 0, s, p);
        } else {
            if (p.act == 1) hipLaunchKernelGGL((dwconv_fwd_kernel<T, 1, false, 1>), grid, dim3(kDwThreads), 0, s, p);
            else            hipLaunchKernelGGL((dwconv_fwd_kernel<T, 1, false, 2>), grid, dim3(kDwThreads), 0, s, p);
        }
        return true;
    }
    if (p.kd == 3) {
        if (p.flip) hipLaunchKernelGGL((dwconv_fwd_kernel<T, 3, true>), grid, dim3(kDwThreads), 0, s, p);
        else        hipLaunchKernelGGL((dwconv_fwd_kernel<T, 3, false>), grid, dim3(kDwThreads), 0, s, p);
    } else {
        if (p.flip) hipLaunchKernelGGL((dwconv_fwd_kernel<T, 1, true>), grid, dim3(kDwThreads), 0, s, p);
        else        hipLaunchKernelGGL((dwconv_fwd_kernel<T, 1, false>), grid, dim3(kDwThreads), 0, s, p);
    }
    return true;
}
template <typename T>
static bool dw_wgrad(const vivim_dwconv_wgrad_params& p, hipStream_t s) {
    const int wtiles = (p.width + kDwTW - 1) / kDwTW;
    const int ntiles = p.depth * p.height * wtiles;
    const int cgroups = (p.channels + 127) / 128;
    // ~2048 blocks in total, at least 8 tiles per wave
    int blocks_x = (2048 + cgroups * p.batch - 1) / (cgroups * p.batch);
    int tpw = (ntiles + blocks_x * 4 - 1) / (blocks_x * 4);
    if (tpw < 8) tpw = 8;
    blocks_x = (ntiles + tpw * 4 - 1) / (tpw * 4);
    dim3 grid(blocks_x, cgroups, p.batch);
    if (p.kd == 3) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 3>), grid, dim3(kDwThreads), 0, s, p, tpw);
    else           hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 1>), grid, dim3(kDwThreads), 0, s, p, tpw);
    return true;
}

bool dwconv_fwd_dispatch(const vivim_dwconv_params& p, hipStream_t s) {
    switch (p.itype) {
        case VIVIM_F32: return dw_fwd<float>(p, s);
        case VIVIM_F16: return dw_fwd<f16_t>(p, s);
        case VIVIM_BF16: return dw_fwd<bf16_t>(p, s);
    }
    return false;
}
bool dwconv_wgrad_dispatch(const vivim_dwconv_wgrad_params& p, hipStream_t s) {
    switch (p.itype) {
        case VIVIM_F32: return dw_wgrad<float>(p, s);
        case VIVIM_F16: return dw_wgrad<f16_t>(p, s);
        case VIVIM_BF16: return dw_wgrad<bf16_t>(p, s);
    }
    return false;
}

}  // namespace vivim
