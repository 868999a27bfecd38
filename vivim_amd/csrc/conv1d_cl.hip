// conv1d_cl.hip -- depthwise causal conv1d (+ fused SiLU) on CHANNEL-LAST tensors, forward and backward, gfx950.
//
// Layout: x is (batch, dim, seqlen) with unit stride along dim and a free token stride (causal_conv1d.cpp:151-152,
// is_channel_last).  The reference has separate kernels for it (causal_conv1d_fwd.cu:193-298, causal_conv1d_bwd.cu:306-472)
// that stage (tokens x 64 channels) tiles through shared memory and transpose them so that threads run along tokens.
// Here lanes run along CHANNELS, which is the contiguous axis already: a lane owns E adjacent channels and one chunk of
// tokens, keeps the last three rows of its channels in registers as the sliding window and walks its chunk row by row.
// Every row access of a wave is one contiguous run of 64 * E elements; there is no shared memory and no barrier.  The halo
// costs three extra rows per chunk (forward) or six (backward).  Not on Vivim's path (its x has unit seqlen stride).
#include "common.cuh"

namespace vivim {

constexpr int kClThreads = 256;
constexpr int kClE = 4;              // channels per lane

__device__ __forceinline__ float cl_weight(const void* w, int wtype, int64_t i) {
    switch (wtype) {
        case VIVIM_F32: return static_cast<const float*>(w)[i];
        case VIVIM_F16: return to_f32<f16_t>(static_cast<const f16_t*>(w)[i]);
        default:        return to_f32<bf16_t>(static_cast<const bf16_t*>(w)[i]);
    }
}

// taps right-aligned into 4 slots (as conv1d.hip): out[t] = bias + sum_j w4[j] * x[t - 3 + j]
template <int E>
__device__ __forceinline__ void cl_load_taps(const vivim_conv_fwd_params& p, int c0, int nv, float (&w4)[4][E], float (&bias)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool in = e < nv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int src = j - (4 - p.width);
            w4[j][e] = (in && src >= 0) ? cl_weight(p.weight, p.wtype, (int64_t)(c0 + e) * p.weight_c_stride + src * p.weight_width_stride) : 0.0f;
        }
        bias[e] = (in && p.bias) ? cl_weight(p.bias, p.wtype, c0 + e) : 0.0f;
    }
}

struct ClGeom { int ncv, nchunk, chunk_len; };

template <typename T, int E>
__global__ void __launch_bounds__(kClThreads) conv1d_cl_fwd_kernel(const vivim_conv_fwd_params p, const ClGeom g) {
    const int64_t gid = (int64_t)blockIdx.x * kClThreads + threadIdx.x;
    const int cv = (int)(gid % g.ncv);
    const int64_t rest = gid / g.ncv;
    const int chunk = (int)(rest % g.nchunk), b = (int)(rest / g.nchunk);
    if (b >= p.batch) return;
    const int c0 = cv * E, nv = p.dim - c0, L = p.seqlen;
    const T* __restrict__ x = static_cast<const T*>(p.x) + b * p.x_batch_stride + c0;
    T* __restrict__ out = static_cast<T*>(p.out) + b * p.out_batch_stride + c0;
    float w4[4][E], bias[E];
    cl_load_taps<E>(p, c0, nv, w4, bias);

    const int t0 = chunk * g.chunk_len, t1 = min(t0 + g.chunk_len, L);
    float xw[3][E];                                   // rows t-3, t-2, t-1
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int t = t0 - 3 + j;
        if (t >= 0) load_k<T, E>(x + (int64_t)t * p.x_l_stride, nv, xw[j]);
        else {
#pragma unroll
            for (int e = 0; e < E; ++e) xw[j][e] = 0.0f;
        }
    }
    for (int t = t0; t < t1; ++t) {
        float xv[E], o[E];
        load_k<T, E>(x + (int64_t)t * p.x_l_stride, nv, xv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float acc = bias[e];
            acc = fmaf(w4[0][e], xw[0][e], acc);
            acc = fmaf(w4[1][e], xw[1][e], acc);
            acc = fmaf(w4[2][e], xw[2][e], acc);
            acc = fmaf(w4[3][e], xv[e], acc);
            o[e] = p.silu_activation ? acc * sigmoidf_fast(acc) : acc;
            xw[0][e] = xw[1][e]; xw[1][e] = xw[2][e]; xw[2][e] = xv[e];
        }
        store_k<T, E>(out + (int64_t)t * p.out_l_stride, nv, o);
    }
}

// Backward, one forward walk over [t0, t1 + 3): at row t the lane forms g[t] = dout[t] * silu'(pre[t]) from its x window
// and emits dx[t-3] = w4[3] g[t-3] + w4[2] g[t-2] + w4[1] g[t-1] + w4[0] g[t] from its g window.  dweight / dbias sums run
// over the lane's own rows [t0, t1) and leave through one atomic per (channel, tap) per chunk.
template <typename T, int E>
__global__ void __launch_bounds__(kClThreads) conv1d_cl_bwd_kernel(const vivim_conv_bwd_params p, const ClGeom g) {
    const vivim_conv_fwd_params& f = p.f;
    const int64_t gid = (int64_t)blockIdx.x * kClThreads + threadIdx.x;
    const int cv = (int)(gid % g.ncv);
    const int64_t rest = gid / g.ncv;
    const int chunk = (int)(rest % g.nchunk), b = (int)(rest / g.nchunk);
    if (b >= f.batch) return;
    const int c0 = cv * E, nv = f.dim - c0, L = f.seqlen;
    const T* __restrict__ x = static_cast<const T*>(f.x) + b * f.x_batch_stride + c0;
    const T* __restrict__ dout = static_cast<const T*>(p.dout) + b * p.dout_batch_stride + c0;
    T* __restrict__ dx = static_cast<T*>(p.dx) + b * p.dx_batch_stride + c0;
    float w4[4][E], bias[E];
    cl_load_taps<E>(f, c0, nv, w4, bias);

    const int t0 = chunk * g.chunk_len, t1 = min(t0 + g.chunk_len, L);
    float xw[3][E], gw[3][E], dw[4][E], db[E];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int t = t0 - 3 + j;
        if (t >= 0) load_k<T, E>(x + (int64_t)t * f.x_l_stride, nv, xw[j]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (t < 0) xw[j][e] = 0.0f;
            gw[j][e] = 0.0f;
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { dw[0][e] = dw[1][e] = dw[2][e] = dw[3][e] = 0.0f; db[e] = 0.0f; }

    for (int t = t0; t < t1 + 3; ++t) {
        float xv[E], gv[E];
        const bool live = t < L;                      // rows past the end carry no gradient
        if (live) {
            load_k<T, E>(x + (int64_t)t * f.x_l_stride, nv, xv);
            load_k<T, E>(dout + (int64_t)t * p.dout_l_stride, nv, gv);
        }
        const bool own = t < t1;
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (!live) { xv[e] = 0.0f; gv[e] = 0.0f; }
            if (f.silu_activation) {
                float pre = bias[e];
                pre = fmaf(w4[0][e], xw[0][e], pre);
                pre = fmaf(w4[1][e], xw[1][e], pre);
                pre = fmaf(w4[2][e], xw[2][e], pre);
                pre = fmaf(w4[3][e], xv[e], pre);
                const float sg = sigmoidf_fast(pre);
                gv[e] *= sg * (1.0f + pre * (1.0f - sg));
            }
            if (own) {
                dw[0][e] = fmaf(gv[e], xw[0][e], dw[0][e]);
                dw[1][e] = fmaf(gv[e], xw[1][e], dw[1][e]);
                dw[2][e] = fmaf(gv[e], xw[2][e], dw[2][e]);
                dw[3][e] = fmaf(gv[e], xv[e], dw[3][e]);
                db[e] += gv[e];
            }
            float acc = w4[0][e] * gv[e];             // same summation order as the channel-first kernel (conv1d.hip)
            acc = fmaf(w4[1][e], gw[2][e], acc);
            acc = fmaf(w4[2][e], gw[1][e], acc);
            o[e] = fmaf(w4[3][e], gw[0][e], acc);
            xw[0][e] = xw[1][e]; xw[1][e] = xw[2][e]; xw[2][e] = xv[e];
            gw[0][e] = gw[1][e]; gw[1][e] = gw[2][e]; gw[2][e] = gv[e];
        }
        const int s = t - 3;
        if (s >= t0 && s < t1) store_k<T, E>(dx + (int64_t)s * p.dx_l_stride, nv, o);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (e >= nv) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int src = j - (4 - f.width);
            if (src >= 0)
                atomicAdd(static_cast<float*>(p.dweight) + (int64_t)(c0 + e) * p.dweight_c_stride + src * p.dweight_width_stride, dw[j][e]);
        }
        if (p.dbias) atomicAdd(static_cast<float*>(p.dbias) + c0 + e, db[e]);
    }
}

// chunks: enough lanes to fill the chip (>= 64 K when the tensor allows), rows per chunk >= 16 so that the halo stays small
static ClGeom cl_geometry(const vivim_conv_fwd_params& p) {
    ClGeom g;
    g.ncv = (p.dim + kClE - 1) / kClE;
    const int64_t lanes_per_chunk = (int64_t)p.batch * g.ncv;
    int64_t want = (65536 + lanes_per_chunk - 1) / lanes_per_chunk;
    const int64_t most = (p.seqlen + 15) / 16;
    if (want > most) want = most;
    if (want < 1) want = 1;
    g.chunk_len = (int)((p.seqlen + want - 1) / want);
    g.nchunk = (p.seqlen + g.chunk_len - 1) / g.chunk_len;
    return g;
}

static unsigned cl_blocks(const vivim_conv_fwd_params& p, const ClGeom& g) {
    const int64_t threads = (int64_t)p.batch * g.nchunk * g.ncv;
    return (unsigned)((threads + kClThreads - 1) / kClThreads);
}

bool conv_cl_fwd_dispatch(const vivim_conv_fwd_params& p, hipStream_t s) {
    const ClGeom g = cl_geometry(p);
    const dim3 grid(cl_blocks(p, g)), block(kClThreads);
    switch (p.itype) {
        case VIVIM_F32:  hipLaunchKernelGGL((conv1d_cl_fwd_kernel<float, kClE>), grid, block, 0, s, p, g); return true;
        case VIVIM_F16:  hipLaunchKernelGGL((conv1d_cl_fwd_kernel<f16_t, kClE>), grid, block, 0, s, p, g); return true;
        case VIVIM_BF16: hipLaunchKernelGGL((conv1d_cl_fwd_kernel<bf16_t, kClE>), grid, block, 0, s, p, g); return true;
    }
    return false;
}

bool conv_cl_bwd_dispatch(const vivim_conv_bwd_params& p, hipStream_t s) {
    const ClGeom g = cl_geometry(p.f);
    const dim3 grid(cl_blocks(p.f, g)), block(kClThreads);
    switch (p.f.itype) {
        case VIVIM_F32:  hipLaunchKernelGGL((conv1d_cl_bwd_kernel<float, kClE>), grid, block, 0, s, p, g); return true;
        case VIVIM_F16:  hipLaunchKernelGGL((conv1d_cl_bwd_kernel<f16_t, kClE>), grid, block, 0, s, p, g); return true;
        case VIVIM_BF16: hipLaunchKernelGGL((conv1d_cl_bwd_kernel<bf16_t, kClE>), grid, block, 0, s, p, g); return true;
    }
    return false;
}

}  // namespace vivim
