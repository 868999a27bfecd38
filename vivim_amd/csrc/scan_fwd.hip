// scan_fwd.hip -- selective SSM scan, forward, gfx950 (wave64).
//
// Math (selective_scan_fwd_kernel.cuh:67-303; oracle: selective_scan_interface.py:86-152):
//   d_t = softplus(delta_t + bias);  h_{n,t} = exp(d_t A_n) h_{n,t-1} + d_t u_t B_{n,t}
//   out_t = sum_n C_{n,t} h_{n,t} + D u_t;   out_z_t = out_t * silu(z_t)
//
// Mapping (NOT the reference's one-block-per-(batch,channel) BlockScan):
//   * a WAVE owns R channels of one batch element and walks the token axis in steps of 64*K tokens;
//     lane l holds tokens [l*K, l*K+K) of the step in registers, so u / delta / z / out / out_z move as
//     one 16-byte access per lane and B_n / C_n rows are loaded once and reused by the R channels;
//   * per (channel, n): a K-long in-register scan, one wave64 scan of the lane aggregates (affine maps
//     h -> P h + H), and the running state between steps lives in a per-wave LDS row -- waves never
//     talk to each other, so the kernel has no barrier;
//   * the state after every step is written to `x` (batch, dim, n_chunks, dstate): the backward's
//     forward re-scan restarts from it (role of the reference's x, selective_scan.cpp:307-313).
#include "common.cuh"

namespace vivim {

constexpr int kScanWaves = 4;   // waves per workgroup (independent of each other)

template <typename T, int K, int R, bool HAS_Z, bool VAR_BC>
__global__ void __launch_bounds__(kScanWaves * kWave) ssm_fwd_kernel(const vivim_ssm_fwd_params p) {
    constexpr int TILE = kWave * K;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int N = p.dstate, L = p.seqlen;
    const int cpg = p.dim / p.n_groups;               // channels per B/C group
    const int wpg = (cpg + R - 1) / R;                // channel-sets (waves) per group
    const int ws = blockIdx.x * kScanWaves + wave;    // this wave's channel-set
    if (ws >= wpg * p.n_groups) return;               // no barriers below: early exit is safe
    const int g = ws / wpg;
    const int d0 = g * cpg + (ws - g * wpg) * R;
    const int nvalid = min(R, (g + 1) * cpg - d0);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* carry = smem + wave * R * N;               // h at the end of the previous step, [r][n]
    for (int i = lane; i < R * N; i += kWave) carry[i] = 0.0f;
    wave_lds_fence();

    int d[R];
    float Dv[R], bias[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        d[r] = d0 + min(r, nvalid - 1);               // clamp: surplus slots shadow the last channel, never stored
        Dv[r] = p.D ? static_cast<const float*>(p.D)[d[r]] : 0.0f;
        bias[r] = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d[r]] : 0.0f;
    }
    const T* __restrict__ uB = static_cast<const T*>(p.u) + b * p.u_batch_stride;
    const T* __restrict__ dB_ = static_cast<const T*>(p.delta) + b * p.delta_batch_stride;
    const float* __restrict__ A = static_cast<const float*>(p.A);
    const T* __restrict__ Bv = static_cast<const T*>(p.B) + b * p.B_batch_stride + g * p.B_group_stride;
    const T* __restrict__ Cv = static_cast<const T*>(p.C) + b * p.C_batch_stride + g * p.C_group_stride;
    const float* __restrict__ Bc = static_cast<const float*>(p.B);   // constant (dim, dstate) forms
    const float* __restrict__ Cc = static_cast<const float*>(p.C);
    float* __restrict__ xck = static_cast<float*>(p.x);
    const int nsteps = (L + TILE - 1) / TILE;

    for (int step = 0; step < nsteps; ++step) {
        const int t0 = step * TILE + lane * K;
        const int nv = L - t0;                        // valid tokens of this lane (may be <= 0 or > K)
        float dl[R][K], w[R][K], y[R][K];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float uf[K], df[K];
            load_k<T, K>(uB + d[r] * p.u_d_stride + t0, nv, uf);
            load_k<T, K>(dB_ + d[r] * p.delta_d_stride + t0, nv, df);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float raw = df[k] + bias[r];
                const float sp = p.delta_softplus ? softplus_ref(raw) : raw;
                dl[r][k] = k < nv ? sp : 0.0f;        // padded tokens: exp2(0)=1, drive 0 -> identity map
                w[r][k] = dl[r][k] * uf[k];
                y[r][k] = Dv[r] * uf[k];
            }
        }
        for (int n = 0; n < N; ++n) {
            float Bn[K], Cn[K];
            if (VAR_BC) {
                load_k<T, K>(Bv + n * p.B_dstate_stride + t0, nv, Bn);
                load_k<T, K>(Cv + n * p.C_dstate_stride + t0, nv, Cn);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!VAR_BC) {
                    const float bc = Bc[d[r] * p.B_group_stride + n * p.B_dstate_stride];
                    const float cc = Cc[d[r] * p.C_group_stride + n * p.C_dstate_stride];
#pragma unroll
                    for (int k = 0; k < K; ++k) { Bn[k] = bc; Cn[k] = cc; }
                }
                const float A2 = A[d[r] * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;   // fwd_kernel.cuh:168-175
                float pl[K], hl[K];
                float P = 1.0f, H = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float a = fast_exp2(dl[r][k] * A2);
                    H = fmaf(a, H, w[r][k] * Bn[k]);
                    P *= a;
                    pl[k] = P;
                    hl[k] = H;
                }
                float Pi = P, Hi = H;
                wave_scan_affine_fwd(Pi, Hi, lane);
                float Pe = __shfl_up(Pi, 1, kWave), He = __shfl_up(Hi, 1, kWave);
                if (lane == 0) { Pe = 1.0f; He = 0.0f; }
                const float cin = carry[r * N + n];
                const float hin = fmaf(Pe, cin, He);          // state entering this lane's first token
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float h = fmaf(pl[k], hin, hl[k]);
                    y[r][k] = fmaf(h, Cn[k], y[r][k]);
                }
                wave_lds_fence();
                if (lane == kWave - 1) {
                    const float cnew = fmaf(Pi, cin, Hi);     // state after the step's last token
                    carry[r * N + n] = cnew;
                    if (r < nvalid)
                        xck[(((int64_t)b * p.dim + d[r]) * nsteps + step) * N + n] = cnew;
                }
                wave_lds_fence();
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r >= nvalid) break;
            T* out = static_cast<T*>(p.out) + b * p.out_batch_stride + d[r] * p.out_d_stride + t0;
            store_k<T, K>(out, nv, y[r]);
            if (HAS_Z) {
                float zf[K], oz[K];
                load_k<T, K>(static_cast<const T*>(p.z) + b * p.z_batch_stride + d[r] * p.z_d_stride + t0, nv, zf);
#pragma unroll
                for (int k = 0; k < K; ++k) oz[k] = y[r][k] * zf[k] * sigmoidf_fast(zf[k]);   // fwd_kernel.cuh:290
                T* outz = static_cast<T*>(p.out_z) + b * p.out_z_batch_stride + d[r] * p.out_z_d_stride + t0;
                store_k<T, K>(outz, nv, oz);
            }
        }
    }
}

template <typename T, int K, int R>
static void launch_fwd(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    const int cpg = p.dim / p.n_groups;
    const int sets = ((cpg + R - 1) / R) * p.n_groups;
    dim3 grid((sets + kScanWaves - 1) / kScanWaves, p.batch);
    const size_t smem = (size_t)kScanWaves * R * p.dstate * sizeof(float);
    const bool var = p.is_variable_B;   // capi enforces is_variable_B == is_variable_C
    if (p.z) {
        if (var) hipLaunchKernelGGL((ssm_fwd_kernel<T, K, R, true, true>), grid, dim3(kScanWaves * kWave), smem, stream, p);
        else     hipLaunchKernelGGL((ssm_fwd_kernel<T, K, R, true, false>), grid, dim3(kScanWaves * kWave), smem, stream, p);
    } else {
        if (var) hipLaunchKernelGGL((ssm_fwd_kernel<T, K, R, false, true>), grid, dim3(kScanWaves * kWave), smem, stream, p);
        else     hipLaunchKernelGGL((ssm_fwd_kernel<T, K, R, false, false>), grid, dim3(kScanWaves * kWave), smem, stream, p);
    }
}

// tokens per checkpoint row of x: 64 lanes * K tokens.  K = 4 for every dtype in this build so that the
// forward's and the backward's steps coincide.
int scan_chunk_len(int) { return kWave * 4; }

bool ssm_fwd_dispatch(const vivim_ssm_fwd_params& p, hipStream_t s) {
    switch (p.itype) {
        case VIVIM_F32: launch_fwd<float, 4, 2>(p, s); return true;
        case VIVIM_F16: launch_fwd<f16_t, 4, 2>(p, s); return true;
        case VIVIM_BF16: launch_fwd<bf16_t, 4, 2>(p, s); return true;
    }
    return false;
}

}  // namespace vivim
