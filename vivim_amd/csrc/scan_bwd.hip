// scan_bwd.hip -- selective SSM scan, backward, gfx950 (wave64).
//
// Math (selective_scan_bwd_kernel.cuh:146-489, real weights), with a_t = exp(d_t A_n), b_t = d_t u_t B_{n,t},
// h_t = a_t h_{t-1} + b_t, dy_t = dout_t * silu(z_t):
//   g_t   = C_{n,t} dy_t + a_{t+1} g_{t+1}                         (reverse scan)
//   du_t  = D dy_t + d_t sum_n g_t B_{n,t}
//   dd_t  = sum_n g_t (B_{n,t} u_t + A_n (h_t - b_t));  ddelta_t = dd_t * sigmoid(delta_t + bias) (<= 20)
//   dA_n  = sum_{b,t} g_t d_t (h_t - b_t);  dB_{n,t} = sum_d g_t d_t u_t;  dC_{n,t} = sum_d dy_t h_t
//   dD    = sum dy_t u_t;  dbias = sum ddelta_t;  dz = dout * out * sig(z) (1 + z (1 - sig(z)))  (saved, rounded out)
//
// Mapping: same as the forward (a wave = R channels x 64*K tokens per step, lanes hold K consecutive
// tokens) but the steps are walked last -> first.  Per (channel, n) and step: the forward states are
// rebuilt from the forward kernel's checkpoint x[step-1] (one wave scan), the reverse recurrence is a
// second wave scan seeded by the carry of the step to the right (kept in per-wave LDS together with that
// step's first decay factor), dB/dC are summed over the wave's R channels in registers before they leave
// the wave.  No inter-wave communication, no barrier.
#include "common.cuh"

namespace vivim {

constexpr int kBwdWaves = 4;

template <typename T, int K, int R, bool HAS_Z, bool VAR_BC>
__global__ void __launch_bounds__(kBwdWaves * kWave) ssm_bwd_kernel(const vivim_ssm_bwd_params p) {
    constexpr int TILE = kWave * K;
    const vivim_ssm_fwd_params& f = p.f;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int N = f.dstate, L = f.seqlen;
    const int cpg = f.dim / f.n_groups;
    const int wpg = (cpg + R - 1) / R;
    const int ws = blockIdx.x * kBwdWaves + wave;
    if (ws >= wpg * f.n_groups) return;
    const int g = ws / wpg;
    const int d0 = g * cpg + (ws - g * wpg) * R;
    const int nvalid = min(R, (g + 1) * cpg - d0);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* gcarry = smem + wave * 3 * R * N;          // g at the first token of the step to the right
    float* afirst = gcarry + R * N;                   // a at the first token of the step to the right
    float* dAacc = afirst + R * N;                    // running dA[r][n] of this wave
    for (int i = lane; i < R * N; i += kWave) { gcarry[i] = 0.0f; afirst[i] = 1.0f; dAacc[i] = 0.0f; }
    wave_lds_fence();

    int d[R];
    float Dv[R], bias[R], dD_acc[R], dbias_acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        d[r] = d0 + min(r, nvalid - 1);
        Dv[r] = f.D ? static_cast<const float*>(f.D)[d[r]] : 0.0f;
        bias[r] = f.delta_bias ? static_cast<const float*>(f.delta_bias)[d[r]] : 0.0f;
        dD_acc[r] = 0.0f;
        dbias_acc[r] = 0.0f;
    }
    const T* __restrict__ uB = static_cast<const T*>(f.u) + b * f.u_batch_stride;
    const T* __restrict__ dlB = static_cast<const T*>(f.delta) + b * f.delta_batch_stride;
    const T* __restrict__ doB = static_cast<const T*>(p.dout) + b * p.dout_batch_stride;
    const float* __restrict__ A = static_cast<const float*>(f.A);
    const T* __restrict__ Bv = static_cast<const T*>(f.B) + b * f.B_batch_stride + g * f.B_group_stride;
    const T* __restrict__ Cv = static_cast<const T*>(f.C) + b * f.C_batch_stride + g * f.C_group_stride;
    const float* __restrict__ Bc = static_cast<const float*>(f.B);
    const float* __restrict__ Cc = static_cast<const float*>(f.C);
    float* __restrict__ dBg = static_cast<float*>(p.dB);
    float* __restrict__ dCg = static_cast<float*>(p.dC);
    const float* __restrict__ xck = static_cast<const float*>(f.x);
    const int nsteps = (L + TILE - 1) / TILE;

    for (int step = nsteps - 1; step >= 0; --step) {
        const int t0 = step * TILE + lane * K;
        const int nv = L - t0;
        float dl[R][K], uu[R][K], dy[R][K], du[R][K], dd[R][K];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float df[K], dof[K];
            load_k<T, K>(uB + d[r] * f.u_d_stride + t0, nv, uu[r]);
            load_k<T, K>(dlB + d[r] * f.delta_d_stride + t0, nv, df);
            load_k<T, K>(doB + d[r] * p.dout_d_stride + t0, nv, dof);
            if (HAS_Z) {
                float zf[K], of[K], dzv[K];
                load_k<T, K>(static_cast<const T*>(f.z) + b * f.z_batch_stride + d[r] * f.z_d_stride + t0, nv, zf);
                load_k<T, K>(static_cast<const T*>(f.out) + b * f.out_batch_stride + d[r] * f.out_d_stride + t0, nv, of);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float sg = sigmoidf_fast(zf[k]);
                    dzv[k] = dof[k] * of[k] * sg * (1.0f + zf[k] * (1.0f - sg));     // bwd_kernel.cuh:186-191
                    dof[k] *= zf[k] * sg;
                }
                if (r < nvalid) {
                    store_k<T, K>(static_cast<T*>(p.dz) + b * p.dz_batch_stride + d[r] * p.dz_d_stride + t0, nv, dzv);
                    if (f.out_z) {                                                    // bwd_kernel.cuh:193-204
                        float oz[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) oz[k] = of[k] * zf[k] * sigmoidf_fast(zf[k]);
                        store_k<T, K>(static_cast<T*>(f.out_z) + b * f.out_z_batch_stride + d[r] * f.out_z_d_stride + t0, nv, oz);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float raw = df[k] + bias[r];
                const float sp = f.delta_softplus ? softplus_ref(raw) : raw;
                dl[r][k] = k < nv ? sp : 0.0f;
                dy[r][k] = dof[k];                     // 0 on padded tokens (dout loads as 0)
                du[r][k] = Dv[r] * dof[k];
                dd[r][k] = 0.0f;
                dD_acc[r] = fmaf(dof[k], uu[r][k], dD_acc[r]);
            }
        }
        for (int n = 0; n < N; ++n) {
            float Bn[K], Cn[K], dBv[K], dCv[K];
            if (VAR_BC) {
                load_k<T, K>(Bv + n * f.B_dstate_stride + t0, nv, Bn);
                load_k<T, K>(Cv + n * f.C_dstate_stride + t0, nv, Cn);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) { dBv[k] = 0.0f; dCv[k] = 0.0f; }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!VAR_BC) {
                    const float bc = Bc[d[r] * f.B_group_stride + n * f.B_dstate_stride];
                    const float cc = Cc[d[r] * f.C_group_stride + n * f.C_dstate_stride];
#pragma unroll
                    for (int k = 0; k < K; ++k) { Bn[k] = bc; Cn[k] = cc; dBv[k] = 0.0f; dCv[k] = 0.0f; }
                }
                const float An = A[d[r] * f.A_d_stride + n * f.A_dstate_stride];
                const float A2 = An * kLog2e;
                // ---- forward re-scan: lane aggregate, wave scan, then the K states ----
                float a[K], h[K];
                float P = 1.0f, H = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    a[k] = fast_exp2(dl[r][k] * A2);
                    H = fmaf(a[k], H, dl[r][k] * uu[r][k] * Bn[k]);
                    P *= a[k];
                }
                wave_scan_affine_fwd(P, H, lane);
                float Pe = __shfl_up(P, 1, kWave), He = __shfl_up(H, 1, kWave);
                if (lane == 0) { Pe = 1.0f; He = 0.0f; }
                const float hstep = step > 0
                    ? xck[(((int64_t)b * f.dim + d[r]) * nsteps + (step - 1)) * N + n] : 0.0f;
                float hc = fmaf(Pe, hstep, He);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    hc = fmaf(a[k], hc, dl[r][k] * uu[r][k] * Bn[k]);
                    h[k] = hc;
                }
                // ---- reverse scan of g_t = a_{t+1} g_{t+1} + C_t dy_t ----
                float an = __shfl_down(a[0], 1, kWave);          // decay of the token right of this lane's last
                if (lane == kWave - 1) an = afirst[r * N + n];
                float Pr = 1.0f, G = 0.0f;
#pragma unroll
                for (int k = K - 1; k >= 0; --k) {
                    const float al = k == K - 1 ? an : a[k + 1];
                    G = fmaf(al, G, Cn[k] * dy[r][k]);
                    Pr *= al;
                }
                wave_scan_affine_rev(Pr, G, lane);
                float Pre = __shfl_down(Pr, 1, kWave), Gre = __shfl_down(G, 1, kWave);
                if (lane == kWave - 1) { Pre = 1.0f; Gre = 0.0f; }
                const float gstep = gcarry[r * N + n];
                float gc = fmaf(Pre, gstep, Gre);                // g at the token right of this lane's last
                float dA_part = 0.0f;
#pragma unroll
                for (int k = K - 1; k >= 0; --k) {
                    const float al = k == K - 1 ? an : a[k + 1];
                    gc = fmaf(al, gc, Cn[k] * dy[r][k]);         // g_t
                    const float bt = dl[r][k] * uu[r][k] * Bn[k];
                    const float ahp = h[k] - bt;                 // a_t h_{t-1}
                    const float gB = gc * Bn[k];
                    du[r][k] = fmaf(gB, dl[r][k], du[r][k]);
                    dd[r][k] += fmaf(gB, uu[r][k], gc * An * ahp);
                    dA_part = fmaf(gc * dl[r][k], ahp, dA_part);
                    dBv[k] = fmaf(gc * dl[r][k], uu[r][k], dBv[k]);
                    dCv[k] = fmaf(dy[r][k], h[k], dCv[k]);
                }
                dA_part = wave_sum(dA_part);
                wave_lds_fence();
                if (lane == 0) {
                    gcarry[r * N + n] = fmaf(Pr, gstep, G);      // g at this step's first token
                    afirst[r * N + n] = a[0];
                    dAacc[r * N + n] += r < nvalid ? dA_part : 0.0f;
                }
                wave_lds_fence();
                if (!VAR_BC && r < nvalid) {                      // constant B/C: (dim, dstate) gradients
                    float sB = 0.f, sC = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) { sB += dBv[k]; sC += dCv[k]; }
                    sB = wave_sum(sB);
                    sC = wave_sum(sC);
                    if (lane == 0) {
                        atomicAdd(dBg + d[r] * p.dB_group_stride + n * p.dB_dstate_stride, sB);
                        atomicAdd(dCg + d[r] * p.dC_group_stride + n * p.dC_dstate_stride, sC);
                    }
                }
                if (VAR_BC && r >= nvalid - 1) break;             // shadow slots must not count twice
            }
            if (VAR_BC) {
                float* dBp = dBg + b * p.dB_batch_stride + g * p.dB_group_stride + n * p.dB_dstate_stride + t0;
                float* dCp = dCg + b * p.dC_batch_stride + g * p.dC_group_stride + n * p.dC_dstate_stride + t0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k < nv) {
                        atomicAdd(dBp + k, dBv[k]);               // fp32 sum over channel sets (bwd_kernel.cuh:312-313)
                        atomicAdd(dCp + k, dCv[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r >= nvalid) break;
            if (f.delta_softplus) {                               // bwd_kernel.cuh:439-452 (delta re-read)
                float df[K];
                load_k<T, K>(dlB + d[r] * f.delta_d_stride + t0, nv, df);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float raw = df[k] + bias[r];
                    if (raw <= 20.0f) dd[r][k] *= sigmoidf_fast(raw);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) dbias_acc[r] += k < nv ? dd[r][k] : 0.0f;
            store_k<T, K>(static_cast<T*>(p.du) + b * p.du_batch_stride + d[r] * p.du_d_stride + t0, nv, du[r]);
            store_k<T, K>(static_cast<T*>(p.ddelta) + b * p.ddelta_batch_stride + d[r] * p.ddelta_d_stride + t0, nv, dd[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r >= nvalid) break;
        const float sD = wave_sum(dD_acc[r]);
        const float sb = wave_sum(dbias_acc[r]);
        if (lane == 0) {
            if (p.dD) atomicAdd(static_cast<float*>(p.dD) + d[r], sD);
            if (p.ddelta_bias) atomicAdd(static_cast<float*>(p.ddelta_bias) + d[r], sb);
        }
    }
    wave_lds_fence();
    for (int i = lane; i < nvalid * N; i += kWave) {
        const int r = i / N, n = i - r * N;
        atomicAdd(static_cast<float*>(p.dA) + (d0 + r) * p.dA_d_stride + n * p.dA_dstate_stride, dAacc[i]);
    }
}

template <typename T, int K, int R>
static void launch_bwd(const vivim_ssm_bwd_params& p, hipStream_t stream) {
    const vivim_ssm_fwd_params& f = p.f;
    const int cpg = f.dim / f.n_groups;
    const int sets = ((cpg + R - 1) / R) * f.n_groups;
    dim3 grid((sets + kBwdWaves - 1) / kBwdWaves, f.batch);
    const size_t smem = (size_t)kBwdWaves * 3 * R * f.dstate * sizeof(float);
    const bool var = f.is_variable_B;
    if (f.z) {
        if (var) hipLaunchKernelGGL((ssm_bwd_kernel<T, K, R, true, true>), grid, dim3(kBwdWaves * kWave), smem, stream, p);
        else     hipLaunchKernelGGL((ssm_bwd_kernel<T, K, R, true, false>), grid, dim3(kBwdWaves * kWave), smem, stream, p);
    } else {
        if (var) hipLaunchKernelGGL((ssm_bwd_kernel<T, K, R, false, true>), grid, dim3(kBwdWaves * kWave), smem, stream, p);
        else     hipLaunchKernelGGL((ssm_bwd_kernel<T, K, R, false, false>), grid, dim3(kBwdWaves * kWave), smem, stream, p);
    }
}

bool ssm_bwd_dispatch(const vivim_ssm_bwd_params& p, hipStream_t s) {
    switch (p.f.itype) {
        case VIVIM_F32: launch_bwd<float, 4, 2>(p, s); return true;
        case VIVIM_F16: launch_bwd<f16_t, 4, 2>(p, s); return true;
        case VIVIM_BF16: launch_bwd<bf16_t, 4, 2>(p, s); return true;
    }
    return false;
}

}  // namespace vivim
