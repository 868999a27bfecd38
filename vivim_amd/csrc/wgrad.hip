// wgrad.hip -- the two weight-gradient products of the fused inner op's backward (selective_scan_interface.py:273, 276:
//   ddelta_proj_weight = einsum("dB,Br->dr", ddelta, x_dbl[:, :R]),   dx_proj_weight = einsum("Br,Bd->rd", dx_dbl, conv1d_out))
// as a split-k MFMA kernel.
//
//   out[g][i][j] = sum_t a[g][i][t] * b[g][j][t]        a: (G, M, K), b: (G, N, K), both with unit stride along t
//
// K is every token of every clip (61 440 at stage 0 of the 256 x 256 configs) while M x N is 128 x 4 or 36 x 128: two or
// four output tiles per direction.  The library GEMM behind torch.bmm runs such a product on as many workgroups as it has
// output tiles -- 77-100 us each at stage 0 for 16-60 MB of operands (tools/layer_prof.sh) -- so here the token axis is split
// over a couple of thousand waves instead.  In the grouped op both operands are channel-major (t contiguous), which is exactly
// the operand order of v_mfma_f32_16x16x32: lane l of a fragment holds row l & 15, k = 8 (l >> 4) .. + 7, eight consecutive
// tokens = one 16-byte load, so the fragments come straight from memory -- no LDS, no transpose.  A wave owns a 64 x (16 | 64)
// output tile and a token range, keeps two or three 32-token steps of fragments in flight beyond the one it multiplies (a step is
// otherwise one memory round trip, about a microsecond), and adds its tile to the f32 output with atomics (the caller zeroes
// `out`).  Device-scope float atomics are the expensive part -- about 1.3 TB/s of added bytes chip-wide (MI355X_MICROARCH.md,
// atomics) -- so the number of splits is bounded by a minimum of sixteen steps per wave: with two steps per wave (and one step in flight) the
// stage-0 products took 166 / 103 us, with sixteen 24 / 22 (profiles/r03_wgrad_sweep.txt; torch.bmm: 74 / 63).  (Adding the four waves of a workgroup up in LDS first
// was tried: 64 ds_add_f32 per lane cost 20-38 us, more than the global atomics they saved.)  Rows beyond M / N are read from
// the last valid row and dropped.  HBM-bound by design: both operands once (a again per column tile of b).
#include "common.cuh"

namespace vivim {

typedef __attribute__((ext_vector_type(8))) __bf16 wg_bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 wg_h8;
typedef __attribute__((ext_vector_type(4))) float wg_f4;

template <typename T> struct WgFrag;
template <> struct WgFrag<bf16_t> {
    typedef wg_bf8 type;
    static __device__ __forceinline__ wg_f4 mma(type a, type b, wg_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WgFrag<f16_t> {
    typedef wg_h8 type;
    static __device__ __forceinline__ wg_f4 mma(type a, type b, wg_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int kWgMF = 4;             // 16-row fragments of a per wave tile: 64 rows of the output


// grid (k splits, row tiles x column tiles, groups), one wave per workgroup.  NF = 16-column fragments per wave tile (1 or 4).
// DEPTH = step buffers of the register ring (DEPTH - 1 steps in flight beyond the current one).
template <typename T, int NF, int DEPTH>
__global__ void __launch_bounds__(kWave) wgrad_nt_kernel(const vivim_wgrad_nt_params p, const int steps_per_split, const int col_tiles) {
    typedef typename WgFrag<T>::type frag;
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const int g = blockIdx.z;
    const int rt = blockIdx.y / col_tiles, ct = blockIdx.y - rt * col_tiles;
    const int m0 = rt * 16 * kWgMF, n0 = ct * 16 * NF;
    const T* __restrict__ a = static_cast<const T*>(p.a) + (int64_t)g * p.a_group_stride;
    const T* __restrict__ b = static_cast<const T*>(p.b) + (int64_t)g * p.b_group_stride;
    const T* arow[kWgMF];
    const T* brow[NF];
#pragma unroll
    for (int i = 0; i < kWgMF; ++i) arow[i] = a + (int64_t)min(m0 + 16 * i + r, p.m - 1) * p.a_row_stride + 8 * kq;
#pragma unroll
    for (int j = 0; j < NF; ++j) brow[j] = b + (int64_t)min(n0 + 16 * j + r, p.n - 1) * p.b_row_stride + 8 * kq;

    wg_f4 acc[kWgMF][NF];
#pragma unroll
    for (int i = 0; i < kWgMF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = wg_f4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (p.k + 31) / 32;                               // 32 tokens per MFMA step
    const int s0 = blockIdx.x * steps_per_split, s1 = min(nsteps, s0 + steps_per_split);
    const frag zero = {};
    // tokens [32 s + 8 kq, + 8) of a row: all inside or all outside (k % 8 == 0); a step behind the wave's range loads nothing
    auto load = [&](const T* row, int s) -> frag {
        return (s < s1 && 32 * s + 8 * kq < p.k) ? *reinterpret_cast<const frag*>(row + 32 * s) : zero;
    };
    frag fa[DEPTH][kWgMF], fb[DEPTH][NF];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
#pragma unroll
        for (int i = 0; i < kWgMF; ++i) fa[d][i] = load(arow[i], s0 + d);
#pragma unroll
        for (int j = 0; j < NF; ++j) fb[d][j] = load(brow[j], s0 + d);
    }
    for (int s = s0; s < s1; s += DEPTH) {                         // DEPTH steps per trip: the register ring needs constant indices
#pragma unroll
        for (int h = 0; h < DEPTH; ++h) {
            constexpr int D1 = DEPTH - 1;
            const int nxt = (h + D1) % DEPTH;
#pragma unroll
            for (int i = 0; i < kWgMF; ++i) fa[nxt][i] = load(arow[i], s + h + D1);
#pragma unroll
            for (int j = 0; j < NF; ++j) fb[nxt][j] = load(brow[j], s + h + D1);
#pragma unroll
            for (int i = 0; i < kWgMF; ++i)                           // (a step behind the range multiplies zeros)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = WgFrag<T>::mma(fa[h][i], fb[h][j], acc[i][j]);
        }
    }
    // C / D map of the 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + register
    float* __restrict__ out = static_cast<float*>(p.out) + (int64_t)g * p.out_group_stride;
#pragma unroll
    for (int i = 0; i < kWgMF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int col = n0 + 16 * j + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + 16 * i + 4 * kq + e;
                if (row < p.m && col < p.n) atomicAdd(out + (int64_t)row * p.out_row_stride + col, acc[i][j][e]);
            }
        }
}

template <typename T>
static void wgrad_launch(const vivim_wgrad_nt_params& p, hipStream_t stream) {
    const int nf = p.n <= 16 ? 1 : 4;
    const int row_tiles = (p.m + 16 * kWgMF - 1) / (16 * kWgMF), col_tiles = (p.n + 16 * nf - 1) / (16 * nf);
    const int nsteps = (p.k + 31) / 32;
    // a couple of thousand waves, at least sixteen 32-token steps each (VIVIM_WGRAD_WAVES / VIVIM_WGRAD_MINSTEPS: experiments)
    const int64_t tiles = (int64_t)row_tiles * col_tiles * p.groups;
    const char* ew = getenv("VIVIM_WGRAD_WAVES");
    const char* es = getenv("VIVIM_WGRAD_MINSTEPS");
    const int waves = ew && atoi(ew) > 0 ? atoi(ew) : 2048, minsteps = es && atoi(es) > 0 ? atoi(es) : 16;
    int splits = (int)std::min<int64_t>(std::max<int64_t>(1, waves / tiles), std::max(1, nsteps / minsteps));
    const int sps = (nsteps + splits - 1) / splits;
    splits = (nsteps + sps - 1) / sps;
    const dim3 grid(splits, row_tiles * col_tiles, p.groups), block(kWave);
    if (nf == 1) hipLaunchKernelGGL((wgrad_nt_kernel<T, 1, 4>), grid, block, 0, stream, p, sps, col_tiles);
    else hipLaunchKernelGGL((wgrad_nt_kernel<T, 4, 3>), grid, block, 0, stream, p, sps, col_tiles);
}

bool wgrad_nt_dispatch(const vivim_wgrad_nt_params& p, hipStream_t stream) {
    switch (p.itype) {
        case VIVIM_BF16: wgrad_launch<bf16_t>(p, stream); return true;
        case VIVIM_F16: wgrad_launch<f16_t>(p, stream); return true;
    }
    return false;                                                     // f32 operands: the caller keeps its library GEMM
}

}  // namespace vivim
