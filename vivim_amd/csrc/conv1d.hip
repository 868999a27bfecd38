// conv1d.hip -- depthwise causal conv1d (+ fused SiLU), channel-first, forward and backward, gfx950.
//
// What it computes is what the reference kernels compute (causal-conv1d/csrc/causal_conv1d_fwd.cu:39-130,
// causal_conv1d_bwd.cu:46-240); how is different: the reference walks one (batch, channel) row per
// 128-thread block, chunk after chunk, passing halos through shared memory with three barriers per
// chunk.  Here a row is cut into independent 64*E-token tiles (E = 16 bytes of elements per lane) and there is no
// barrier anywhere.  Pure stream: 2*s bytes/token forward, 3*s backward.
//   * conv1d_{fwd,bwd}_pipe_kernel (second half of the file): 16-byte aligned rows of a multiple of E tokens and at
//     least two tiles -- every launch of Vivim's stages 0-2.  A wave walks up to eight consecutive tiles with the next
//     tile's loads in flight under the current tile's arithmetic; halos are loaded, not shuffled.
//   * conv1d_{fwd,bwd}_kernel: everything else (ragged or unaligned rows, rows of one tile: stage 3's 320 tokens).  One
//     tile per wave; halos move between neighbouring lanes with wave shuffles, the two lanes at the wave edges fetch
//     theirs with guarded scalar loads.
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

// ---- long aligned rows: several tiles per wave, the next tile's loads in flight under the current tile's arithmetic -----
// The one-tile-per-wave kernels above finish a tile as load -> arithmetic -> store with nothing of their own to overlap;
// measured on cfg 2's stage 0 the backward took the SUM of its memory time (26 us at the copy rate) and its instruction
// time (21 us): waves that start together stay in phase.  Here a wave walks `nt` consecutive tiles of its row and issues
// tile k+1's loads before it touches tile k.  Halos come from memory as well (4-element vectors either side of the lane's
// own 16 bytes, L1 hits: the neighbouring lane loads the same line) instead of from lane shuffles, so a lane needs nothing
// from its neighbours and the edges need no branch: the row is a raw buffer resource of exactly L elements, a vector that
// starts before the row (wrapped offset) or behind it reads as zeros -- the causal left padding and the tile tail -- and a
// store behind the row is dropped.  Host-side conditions: 16-byte aligned rows, L a multiple of E.
using conv_rsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned kConvNowhere = 0x80000000u;      // an offset behind every row (rows are < 2 GiB): loads return zeros without traffic
__device__ __forceinline__ conv_rsrc conv_row(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
template <typename T> struct ConvHalo;                      // 4 elements
template <> struct ConvHalo<float> {
    using V = u32x4;
    static __device__ __forceinline__ V load(conv_rsrc r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
};
template <typename T> struct ConvHalo {
    using V = u32x2;
    static __device__ __forceinline__ V load(conv_rsrc r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
};
template <typename T, typename V, int N>
__device__ __forceinline__ void conv_unpack(const V& raw, float* v) {
    union { V raw; T e[N]; } u;
    u.raw = raw;
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = to_f32<T>(u.e[j]);
}
template <typename T, int E>
__device__ __forceinline__ void conv_store(conv_rsrc r, unsigned off, const float (&o)[E]) {
    union { u32x4 raw; T e[E]; } u;
#pragma unroll
    for (int k = 0; k < E; ++k) u.e[k] = from_f32<T>(o[k]);
    __builtin_amdgcn_raw_buffer_store_b128(u.raw, r, off, 0, 0);
}

template <typename T, bool BWD> struct ConvTile {
    typename ConvHalo<T>::V xl, xr, dr;      // x[t0-4, t0), x[t0+E, t0+E+4), dout[t0+E, t0+E+4)   (xr, dr: backward only)
    u32x4 xm, dm;                            // x[t0, t0+E), dout[t0, t0+E)
};
template <typename T, int E, bool BWD>
__device__ __forceinline__ ConvTile<T, BWD> conv_tile_load(conv_rsrc rx, conv_rsrc rd, unsigned off) {
    ConvTile<T, BWD> t;
    t.xl = ConvHalo<T>::load(rx, off - 4u * (unsigned)sizeof(T));
    t.xm = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    if (BWD) {
        t.xr = ConvHalo<T>::load(rx, off + 16u);
        t.dm = __builtin_amdgcn_raw_buffer_load_b128(rd, off, 0, 0);
        t.dr = ConvHalo<T>::load(rd, off + 16u);
    }
    return t;
}

// The waves of the launch are numbered through (channel, chunk) pairs, a chunk being nt consecutive 64*E-token tiles of the
// row (blockIdx.z, channel); grid (ceil(dim * cpr / 4), 1, batch).  A wave past the last channel has nothing to do (no barrier).
__device__ __forceinline__ bool conv_chunk(int dim, int cpr, int& c, int& chunk) {
    const int wid = blockIdx.x * (kConvThreads / kWave) + (threadIdx.x >> 6);
    c = __builtin_amdgcn_readfirstlane(wid / cpr);
    chunk = __builtin_amdgcn_readfirstlane(wid - c * cpr);
    return c < dim;
}

template <typename T, typename WT, int E, bool SILU>
__global__ void __launch_bounds__(kConvThreads) conv1d_fwd_pipe_kernel(const vivim_conv_fwd_params p, const int nt, const int cpr) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.z;
    int c, chunk;
    if (!conv_chunk(p.dim, cpr, c, chunk)) return;
    const unsigned row_bytes = (unsigned)p.seqlen * (unsigned)sizeof(T);
    const conv_rsrc rx = conv_row(static_cast<const T*>(p.x) + b * p.x_batch_stride + c * p.x_c_stride, row_bytes);
    const conv_rsrc ro = conv_row(static_cast<T*>(p.out) + b * p.out_batch_stride + c * p.out_c_stride, row_bytes);
    float w4[4];
    load_taps<WT>(static_cast<const WT*>(p.weight) + c * p.weight_c_stride, p.weight_width_stride, p.width, w4);
    const float bias = p.bias ? to_f32<WT>(static_cast<const WT*>(p.bias)[c]) : 0.0f;

    constexpr unsigned kTileBytes = kWave * 16u;
    unsigned off = (unsigned)(chunk * nt) * kTileBytes + (unsigned)lane * 16u;
    ConvTile<T, false> cur = conv_tile_load<T, E, false>(rx, rx, off);
    for (int k = 0; k < nt; ++k, off += kTileBytes) {
        const ConvTile<T, false> nxt = conv_tile_load<T, E, false>(rx, rx, k + 1 < nt ? off + kTileBytes : kConvNowhere);
        float X[E + 4];                          // X[i] = x[t0 - 4 + i]
        conv_unpack<T, typename ConvHalo<T>::V, 4>(cur.xl, X);
        conv_unpack<T, u32x4, E>(cur.xm, X + 4);
        float o[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float acc = bias;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = fmaf(w4[i], X[j + 1 + i], acc);
            o[j] = SILU ? acc * sigmoidf_fast(acc) : acc;
        }
        conv_store<T, E>(ro, off, o);
        cur = nxt;
    }
}

template <typename T, typename WT, int E, bool SILU>
__global__ void __launch_bounds__(kConvThreads) conv1d_bwd_pipe_kernel(const vivim_conv_bwd_params p, const int nt, const int cpr, const int share) {
    const vivim_conv_fwd_params& f = p.f;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.z;
    int c, chunk;
    if (!conv_chunk(f.dim, cpr, c, chunk)) return;
    const unsigned row_bytes = (unsigned)f.seqlen * (unsigned)sizeof(T);
    const conv_rsrc rx = conv_row(static_cast<const T*>(f.x) + b * f.x_batch_stride + c * f.x_c_stride, row_bytes);
    const conv_rsrc rd = conv_row(static_cast<const T*>(p.dout) + b * p.dout_batch_stride + c * p.dout_c_stride, row_bytes);
    const conv_rsrc ro = conv_row(static_cast<T*>(p.dx) + b * p.dx_batch_stride + c * p.dx_c_stride, row_bytes);
    float w4[4];
    load_taps<WT>(static_cast<const WT*>(f.weight) + c * f.weight_c_stride, f.weight_width_stride, f.width, w4);
    const float bias = f.bias ? to_f32<WT>(static_cast<const WT*>(f.bias)[c]) : 0.0f;

    constexpr unsigned kTileBytes = kWave * 16u;
    unsigned off = (unsigned)(chunk * nt) * kTileBytes + (unsigned)lane * 16u;
    float red[5] = {0.f, 0.f, 0.f, 0.f, 0.f};   // dw4[0..3], dbias over the wave's tiles
    ConvTile<T, true> cur = conv_tile_load<T, E, true>(rx, rd, off);
    for (int k = 0; k < nt; ++k, off += kTileBytes) {
        const ConvTile<T, true> nxt = conv_tile_load<T, E, true>(rx, rd, k + 1 < nt ? off + kTileBytes : kConvNowhere);
        float X[E + 8];                          // X[i] = x[t0 - 4 + i]
        float dO[E + 4];                         // dout[t0 + j]
        conv_unpack<T, typename ConvHalo<T>::V, 4>(cur.xl, X);
        conv_unpack<T, u32x4, E>(cur.xm, X + 4);
        conv_unpack<T, typename ConvHalo<T>::V, 4>(cur.xr, X + 4 + E);
        conv_unpack<T, u32x4, E>(cur.dm, dO);
        conv_unpack<T, typename ConvHalo<T>::V, 4>(cur.dr, dO + E);
        float g[E + 3];
#pragma unroll
        for (int j = 0; j < E + 3; ++j) {
            float gj = dO[j];
            if (SILU) {
                float pre = bias;
#pragma unroll
                for (int i = 0; i < 4; ++i) pre = fmaf(w4[i], X[j + 1 + i], pre);
                const float sg = sigmoidf_fast(pre);
                gj *= sg * (1.0f + pre * (1.0f - sg));
            }
            g[j] = gj;      // tokens >= L carry dout == 0, hence g == 0
        }
        float o[E];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = fmaf(w4[i], g[j + 3 - i], acc);
            o[j] = acc;
        }
        conv_store<T, E>(ro, off, o);
#pragma unroll
        for (int j = 0; j < E; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[i] = fmaf(g[j], X[j + 1 + i], red[i]);
            red[4] += g[j];
        }
        cur = nxt;
    }
    // dweight / dbias.  Device-scope float atomics are resolved behind the XCDs' L2s, one cache line at a time: they cost about
    // 70 ns of kernel time per thousand (one tile per wave, 691 k of them at cfg 2's stage 0: 74 us against 28), so as few as
    // possible: when the host made the chunks per row a multiple of four (share), the workgroup's four waves hold the same
    // channel and add up in LDS first.
    __shared__ float part[kConvThreads / kWave][5];
    float mine = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float s = wave_sum(red[i]);
        if (lane == i) mine = s;
    }
    if (share) {                                 // uniform over the workgroup: all four waves are here (same channel, c < dim)
        if (lane < 5) part[threadIdx.x >> 6][lane] = mine;
        __syncthreads();
        if (threadIdx.x >= 5) return;
        mine = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    }
    if (lane < 4) {
        const int src = lane - (4 - f.width);
        if (src >= 0)
            atomicAdd(static_cast<float*>(p.dweight) + c * p.dweight_c_stride + src * p.dweight_width_stride, mine);
    } else if (lane == 4 && p.dbias) {
        atomicAdd(static_cast<float*>(p.dbias) + c, mine);
    }
}

// Tiles per wave (nt) and chunks per row (cpr): chains as long as a wave target allows (VIVIM_CONV_WAVES overrides it for
// experiments; at most 16 tiles), and the chunks per row rounded up to a multiple of four where a row has that many tiles, so
// that a workgroup's four waves share their channel (the backward's atomics, see the kernel).
static void conv_pipe_plan(int64_t rows, int tpr, int& nt, int& cpr) {
    const char* e = getenv("VIVIM_CONV_WAVES");                  // read per call: tests sweep it
    const int target = e && atoi(e) > 0 ? atoi(e) : 4096;
    const int nt0 = (int)std::min<int64_t>(16, std::max<int64_t>(1, rows * tpr / target));
    cpr = (tpr + nt0 - 1) / nt0;
    if (cpr >= 2 && (cpr + 3) / 4 * 4 <= tpr) cpr = (cpr + 3) / 4 * 4;
    nt = (tpr + cpr - 1) / cpr;
}
static bool conv_aligned16(const void* p, int64_t sb, int64_t sc, size_t es) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (sb * (int64_t)es) % 16 == 0 && (sc * (int64_t)es) % 16 == 0;
}
template <typename T>
static bool conv_pipe_ok(const vivim_conv_fwd_params& f, int E) {
    const bool off = getenv("VIVIM_CONV_NO_PIPE") != nullptr;
    const int tpr = (f.seqlen + kWave * E - 1) / (kWave * E);
    return !off && f.seqlen % E == 0 && tpr >= 2 && (int64_t)f.seqlen * (int64_t)sizeof(T) < (1ll << 31) &&
           f.batch <= 65535 && conv_aligned16(f.x, f.x_batch_stride, f.x_c_stride, sizeof(T));
}

template <typename T, typename WT>
static void launch_conv_fwd(const vivim_conv_fwd_params& p, hipStream_t stream) {
    constexpr int E = 16 / sizeof(T);   // one 16-byte access per lane
    const int tpr = (p.seqlen + kWave * E - 1) / (kWave * E);          // 64*E-token tiles per row
    if (conv_pipe_ok<T>(p, E) && conv_aligned16(p.out, p.out_batch_stride, p.out_c_stride, sizeof(T))) {
        int nt, cpr;
        conv_pipe_plan((int64_t)p.batch * p.dim, tpr, nt, cpr);
        const int wpb = kConvThreads / kWave;
        const dim3 grid((unsigned)(((int64_t)p.dim * cpr + wpb - 1) / wpb), 1, p.batch);
        if (p.silu_activation) hipLaunchKernelGGL((conv1d_fwd_pipe_kernel<T, WT, E, true>), grid, dim3(kConvThreads), 0, stream, p, nt, cpr);
        else hipLaunchKernelGGL((conv1d_fwd_pipe_kernel<T, WT, E, false>), grid, dim3(kConvThreads), 0, stream, p, nt, cpr);
        return;
    }
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
    if (conv_pipe_ok<T>(p.f, E) && conv_aligned16(p.dout, p.dout_batch_stride, p.dout_c_stride, sizeof(T)) &&
        conv_aligned16(p.dx, p.dx_batch_stride, p.dx_c_stride, sizeof(T))) {
        int nt, cpr;
        conv_pipe_plan((int64_t)p.f.batch * p.f.dim, tpr, nt, cpr);
        const int wpb = kConvThreads / kWave, share = cpr % wpb == 0;
        const dim3 grid((unsigned)(((int64_t)p.f.dim * cpr + wpb - 1) / wpb), 1, p.f.batch);
        if (p.f.silu_activation) hipLaunchKernelGGL((conv1d_bwd_pipe_kernel<T, WT, E, true>), grid, dim3(kConvThreads), 0, stream, p, nt, cpr, share);
        else hipLaunchKernelGGL((conv1d_bwd_pipe_kernel<T, WT, E, false>), grid, dim3(kConvThreads), 0, stream, p, nt, cpr, share);
        return;
    }
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
