// dirmap.hip -- the v3 block's three scan directions as index maps (include/vivim_hip.h: vivim_dir_params).
//
// scatter: one read of the (batch, channels, seqlen) tensor, three writes into the stacked
//          (batch, halves, 3, csplit, seqlen) tensor the grouped fused op consumes;
// gather : three reads of the stacked tensor, one write -- the (out + out_b.flip + out_s^-1) / 3 combination of
//          mamba_simple.py:261-264 and, with scale 1, the gradient of scatter.
// A thread owns one 16-byte vector of consecutive tokens on the FLAT side (coalesced there); on the stacked side
// direction 0 is the same vector, direction 1 the mirrored vector with its elements reversed (still one 16-byte
// access), direction 2 (frame interleave, token t*hw+p <-> p*nf+t) is element-wise: consecutive flat tokens are
// nf elements apart there, the lines are shared between neighbouring threads and served by L2.
// Pure data movement: HBM-bound, algorithmic bytes 4 * batch * channels * seqlen * sizeof(T) either way.
#include "common.cuh"

namespace vivim {

template <typename T, bool GATHER>
__global__ void __launch_bounds__(256) dir_kernel(const vivim_dir_params p) {
    constexpr int E = 16 / (int)sizeof(T);
    typedef typename Pack<T, 16>::type vec;
    const int L = p.seqlen, nf = p.nframes, hw = L / nf;
    const int l0 = (blockIdx.x * 256 + threadIdx.x) * E;
    if (l0 >= L) return;                               // L % E == 0 (host)
    const int c = blockIdx.y, b = blockIdx.z;
    const int64_t stk = (int64_t)b * p.stk_batch_stride + (int64_t)(c / p.csplit) * p.stk_half_stride +
                        (int64_t)(c % p.csplit) * p.stk_c_stride;
    const int64_t flat = (int64_t)b * p.flat_batch_stride + (int64_t)c * p.flat_c_stride + l0;
    const float scale = p.scale;
    union U { vec v; T e[E]; };
    if (GATHER) {
        const T* __restrict__ s = static_cast<const T*>(p.src) + stk;
        U a0, a1, r;
        a0.v = *reinterpret_cast<const vec*>(s + l0);
        a1.v = *reinterpret_cast<const vec*>(s + p.stk_dir_stride + (L - E - l0));
        const T* __restrict__ s2 = s + 2 * p.stk_dir_stride;
        int t = l0 / hw, q = l0 - t * hw;              // flat token l0 + j = t*hw + q
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const float v2 = to_f32<T>(s2[(int64_t)q * nf + t]);
            r.e[j] = from_f32<T>((to_f32<T>(a0.e[j]) + to_f32<T>(a1.e[E - 1 - j]) + v2) * scale);
            if (++q == hw) { q = 0; ++t; }
        }
        *reinterpret_cast<vec*>(static_cast<T*>(p.dst) + flat) = r.v;
    } else {
        U a, o0, o1;
        a.v = *reinterpret_cast<const vec*>(static_cast<const T*>(p.src) + flat);
        T* __restrict__ d = static_cast<T*>(p.dst) + stk;
        T* __restrict__ d2 = d + 2 * p.stk_dir_stride;
        int t = l0 / hw, q = l0 - t * hw;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const T v = from_f32<T>(to_f32<T>(a.e[j]) * scale);
            o0.e[j] = v;
            o1.e[E - 1 - j] = v;
            d2[(int64_t)q * nf + t] = v;
            if (++q == hw) { q = 0; ++t; }
        }
        *reinterpret_cast<vec*>(d + l0) = o0.v;
        *reinterpret_cast<vec*>(d + p.stk_dir_stride + (L - E - l0)) = o1.v;
    }
}

// The same maps with the frame interleave transposed through LDS: a workgroup owns Q pixel positions of ALL frames of one
// (batch, channel) -- nf contiguous Q-token pieces on the flat side, ONE contiguous nf*Q-token piece on the interleaved
// side -- so direction 2 is written (read) with whole 16-byte vectors too.  The element-wise kernel above issued E two-byte
// stores per thread into lines that four other workgroups complete (30.8 us per launch in the bench, 16 launches per
// step: half the time of all forward scans).  Needs hw % E == 0 (vectors of a frame's piece stay aligned) and nf <= 16.
// DIR_ABL (tools/abl.sh dirbuild; timing only, results WRONG): 1 = directions 0 and 1 are neither written (scatter) nor read
// (gather) -- what the two maps would cost if the conv / scan kernels read xz themselves, forward and reversed (SURVEY.md
// 8f row 1), and only the frame interleave stayed a copy.
#ifndef DIR_ABL
#define DIR_ABL 0
#endif
constexpr int kDirMaxFrames = 16;
template <typename T, bool GATHER>
__global__ void __launch_bounds__(256) dir_tile_kernel(const vivim_dir_params p) {
    constexpr int E = 16 / (int)sizeof(T);
    constexpr int Q = 64 * E;                          // pixel positions per workgroup: 64 vectors per frame
    typedef typename Pack<T, 16>::type vec;
    __shared__ __attribute__((aligned(16))) T lds[kDirMaxFrames * Q];
    const int L = p.seqlen, nf = p.nframes, hw = L / nf;
    const int q0 = blockIdx.x * Q, c = blockIdx.y, b = blockIdx.z;
    const int qn = min(Q, hw - q0);                    // % E == 0
    const int64_t stk = (int64_t)b * p.stk_batch_stride + (int64_t)(c / p.csplit) * p.stk_half_stride +
                        (int64_t)(c % p.csplit) * p.stk_c_stride;
    const int64_t flat = (int64_t)b * p.flat_batch_stride + (int64_t)c * p.flat_c_stride;
    const float scale = p.scale;
    const int vpf = qn / E;                            // vectors per frame piece
    union U { vec v; T e[E]; };
    if (GATHER) {
        const T* __restrict__ s = static_cast<const T*>(p.src) + stk;
        const T* __restrict__ s2 = s + 2 * p.stk_dir_stride + (int64_t)q0 * nf;
        for (int j = threadIdx.x; j < vpf * nf; j += 256) {               // the interleaved piece, vector by vector
            U a;
            a.v = *reinterpret_cast<const vec*>(s2 + j * E);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = j * E + e, q = idx / nf, t = idx - q * nf;
                lds[t * Q + q] = a.e[e];
            }
        }
        __syncthreads();
        T* __restrict__ d = static_cast<T*>(p.dst) + flat;
        for (int i = threadIdx.x; i < vpf * nf; i += 256) {
            const int t = i / vpf, v = i - t * vpf;
            const int l0 = t * hw + q0 + v * E;
            U a0, a1, a2, r;
            a2.v = *reinterpret_cast<const vec*>(lds + t * Q + v * E);
            if (DIR_ABL == 1) { a0.v = a2.v; a1.v = a2.v; }
            else {
                a0.v = *reinterpret_cast<const vec*>(s + l0);
                a1.v = *reinterpret_cast<const vec*>(s + p.stk_dir_stride + (L - E - l0));
            }
#pragma unroll
            for (int e = 0; e < E; ++e)
                r.e[e] = from_f32<T>((to_f32<T>(a0.e[e]) + to_f32<T>(a1.e[E - 1 - e]) + to_f32<T>(a2.e[e])) * scale);
            *reinterpret_cast<vec*>(d + l0) = r.v;
        }
    } else {
        const T* __restrict__ sflat = static_cast<const T*>(p.src) + flat;
        T* __restrict__ d = static_cast<T*>(p.dst) + stk;
        for (int i = threadIdx.x; i < vpf * nf; i += 256) {
            const int t = i / vpf, v = i - t * vpf;
            const int l0 = t * hw + q0 + v * E;
            U a, o0, o1;
            a.v = *reinterpret_cast<const vec*>(sflat + l0);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const T x = from_f32<T>(to_f32<T>(a.e[e]) * scale);
                o0.e[e] = x;
                o1.e[E - 1 - e] = x;
            }
            if (DIR_ABL != 1) {
                *reinterpret_cast<vec*>(d + l0) = o0.v;
                *reinterpret_cast<vec*>(d + p.stk_dir_stride + (L - E - l0)) = o1.v;
            }
            *reinterpret_cast<vec*>(lds + t * Q + v * E) = o0.v;
        }
        __syncthreads();
        T* __restrict__ d2 = d + 2 * p.stk_dir_stride + (int64_t)q0 * nf;
        for (int j = threadIdx.x; j < vpf * nf; j += 256) {
            U o;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int idx = j * E + e, q = idx / nf, t = idx - q * nf;
                o.e[e] = lds[t * Q + q];
            }
            *reinterpret_cast<vec*>(d2 + j * E) = o.v;
        }
    }
}

template <bool GATHER>
bool dir_dispatch(const vivim_dir_params& p, hipStream_t stream) {
    const int E = p.itype == VIVIM_F32 ? 4 : 8;
    const int hw = p.seqlen / p.nframes;
    if (hw % E == 0 && p.nframes <= kDirMaxFrames && p.nframes > 1) {
        const dim3 grid((hw + 64 * E - 1) / (64 * E), p.channels, p.batch), block(256);
        switch (p.itype) {
            case VIVIM_F32: hipLaunchKernelGGL((dir_tile_kernel<float, GATHER>), grid, block, 0, stream, p); return true;
            case VIVIM_F16: hipLaunchKernelGGL((dir_tile_kernel<f16_t, GATHER>), grid, block, 0, stream, p); return true;
            case VIVIM_BF16: hipLaunchKernelGGL((dir_tile_kernel<bf16_t, GATHER>), grid, block, 0, stream, p); return true;
        }
        return false;
    }
    const dim3 grid((p.seqlen / E + 255) / 256, p.channels, p.batch), block(256);
    switch (p.itype) {
        case VIVIM_F32: hipLaunchKernelGGL((dir_kernel<float, GATHER>), grid, block, 0, stream, p); return true;
        case VIVIM_F16: hipLaunchKernelGGL((dir_kernel<f16_t, GATHER>), grid, block, 0, stream, p); return true;
        case VIVIM_BF16: hipLaunchKernelGGL((dir_kernel<bf16_t, GATHER>), grid, block, 0, stream, p); return true;
    }
    return false;
}
template bool dir_dispatch<false>(const vivim_dir_params&, hipStream_t);
template bool dir_dispatch<true>(const vivim_dir_params&, hipStream_t);

}  // namespace vivim
