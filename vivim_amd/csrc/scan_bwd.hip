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
#include <stdlib.h>
#include "common.cuh"

namespace vivim {

int scan_ckpt_len(const vivim_ssm_fwd_params&);                          // scan_fwd.hip
bool ls_shape_ok(const vivim_ssm_fwd_params&);                           // scan_ls.hip
int ls_ckpt_len(const vivim_ssm_fwd_params&);
bool try_ls_bwd(const vivim_ssm_bwd_params&, hipStream_t);
size_t ls_bwd_workspace_bytes(const vivim_ssm_fwd_params&);

constexpr int kBwdWaves = 4;

template <typename T, int K, int R, bool HAS_Z, bool VAR_BC>
__global__ void __launch_bounds__(kBwdWaves * kWave) ssm_bwd_generic_kernel(const vivim_ssm_bwd_params p, const int ck) {
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
    const int nck = (L + ck - 1) / ck;                 // checkpoint rows of x: one per ck tokens (ck divides TILE)

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
                    ? xck[(((int64_t)b * f.dim + d[r]) * nck + (step * (TILE / ck) - 1)) * N + n] : 0.0f;
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


// ------------------------------------------------------------------------------------------------
// Fast path (variable B/C, dstate % 8 == 0 and <= 64, aligned rows, seqlen % K == 0).
//   * workgroup = 8 waves = 16 channels of one batch element and one B/C group; a wave owns a channel
//     PAIR and all N states; lane l holds K consecutive tokens of the 64*K-token step; steps are walked
//     last -> first;
//   * per state: forward re-scan of both channels (DPP scan, seeded by the forward kernel's checkpoint)
//     and reverse scan of g (DPP in-row + read_lane row join), outputs accumulated in registers;
//   * dB / dC: the pair's sum stays in registers; per state every wave drops its 2*K values per lane into
//     its own LDS slot (plain stores -- ds_add_f32 measures 193 cycles per wave-instruction on gfx950,
//     50x a ds_write), ONE barrier, then each thread sums the 8 slots of one (token, dB|dC) element in
//     fixed order and issues one fp32 atomic to HBM: 1/16 of the per-channel atomics of the reference
//     (bwd_kernel.cuh:312-313), which on MI355X would cap the kernel at the chip's ~1.3 TB/s atomic
//     rate.  Slots are double-buffered on the state parity, so one barrier per state suffices;
//   * every per-(channel, state) scalar the step needs (checkpoint, g carry, first decay of the step to the
//     right, A, A*log2e, running dA) sits in a per-wave LDS record read with one ds_read_b128 pair.
constexpr int kBwWmax = 8;         // waves per workgroup: 8 or 4 (template parameter W of the fast kernel)
constexpr int kBwR = 2;            // channels per wave
constexpr int kRec = 8;            // floats per (state, channel) record

// Segment scratch (only when the token axis is split over S > 1 workgroups): the reverse recurrence needs,
// at the right edge of every segment, g of the first token of the next one.
struct BwdSeg {
    int S, seg_steps;          // segments, steps per segment
    float* agg;                // [batch][dim][S][dstate]  g at the segment's first token for zero inflow
    float* dsum;               // [batch][dim][S]          sum over the segment of delta_{t+1}
    float* gin;                // [batch][dim][S][dstate]  g flowing into the segment from the right
};

// Packed fp32 pairs: the two channels a wave owns run the same instruction stream, so every element-wise step of
// the per-state loop is written on float2 = (channel 0, channel 1) and compiles to v_pk_mul/fma/add_f32 -- two
// channels per VALU issue (the unpacked build issued 105 scalar mul/fma per state and packed 18).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 exp2_2(f2 v) { return f2{fast_exp2(v.x), fast_exp2(v.y)}; }
template <int CTRL> __device__ __forceinline__ f2 dpp_mov2(f2 old, f2 src) {
    return f2{dpp_mov<CTRL>(old.x, src.x), dpp_mov<CTRL>(old.y, src.y)};
}

// The backward of one segment of the token axis (the whole sequence when S == 1).
// DA_LDS (N <= 16): dA partial sums stay per lane in LDS and are reduced over the lanes once per segment, instead of
// one wave reduction per (state, channel) and step.
template <typename T, int K, bool HAS_Z, int W, bool DA_LDS>
__global__ void __launch_bounds__(W * kWave, (K == 4 && W == 4) ? 3 : 2) ssm_bwd_fast_kernel(const vivim_ssm_bwd_params p, const BwdSeg sg) {
    constexpr int R = kBwR;
    static_assert(R == 2, "the state loop is written on channel pairs");
    constexpr int TILE = kWave * K;
    static_assert(TILE % kChunk == 0, "a step starts on a checkpoint row");
    constexpr int EPT = 2 * TILE / (W * kWave);     // (token, dB|dC) elements of a step each thread reduces
    static_assert(EPT * W * kWave == 2 * TILE, "whole elements per thread");
    const vivim_ssm_fwd_params& f = p.f;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int N = f.dstate, L = f.seqlen;
    const int cpg = f.dim / f.n_groups;
    const int ppg = (cpg + R - 1) / R;                 // channel pairs per B/C group
    const int bpg = (ppg + W - 1) / W;           // workgroups per group
    const int g = blockIdx.x / bpg;
    const int pair = (blockIdx.x - g * bpg) * W + wave;
    const bool active = pair < ppg;                    // surplus waves only help with barriers and the flush
    const int d0 = g * cpg + min(pair, ppg - 1) * R;
    const int nvalid = min(R, (g + 1) * cpg - d0);
    const int nsteps = (L + TILE - 1) / TILE;
    const int seg = blockIdx.z;
    const int s_lo = seg * sg.seg_steps, s_hi = min(nsteps, s_lo + sg.seg_steps);
    const int t_next = s_hi * TILE;                   // first token right of the segment

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SLOT = 2 * K * kWave;                // floats per wave slot: [dB k0..k3 | dC k0..k3][lane]
    // slot buffers: double-buffered on the state parity, one workgroup barrier per state (four buffers and a barrier
    // every second state measured 4-10 % slower: the skew between waves grows with the distance between barriers)
    constexpr int NBUF = 2;
    float* slots = smem;                               // [n % NBUF][wave][2K][64]
    // wave-private records, [state][field][channel]: the pair of a field is one 8-byte read
    float* rec = smem + NBUF * W * SLOT + wave * (N * R * kRec);
    f2* dAl = reinterpret_cast<f2*>(smem + NBUF * W * SLOT + W * (N * R * kRec)) + wave * (N * kWave);   // [state][lane]
    if (DA_LDS)
        for (int i = lane; i < N * kWave; i += kWave) dAl[i] = f2{0.0f, 0.0f};
    enum { HCK = 0, GCAR = 1, AFIRST = 2, A2VAL = 3, DAACC = 4 };
#define VIVIM_REC(n, F, r) rec[((n) * kRec + (F)) * R + (r)]

    int d[R];
    float Dv[R], bias[R], msk[R], dD_acc[R], dbias_acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        d[r] = d0 + min(r, nvalid - 1);
        Dv[r] = f.D ? static_cast<const float*>(f.D)[d[r]] : 0.0f;
        bias[r] = f.delta_bias ? static_cast<const float*>(f.delta_bias)[d[r]] : 0.0f;
        msk[r] = (active && r < nvalid) ? 1.0f : 0.0f;  // shadow slots / surplus waves contribute nothing
        dD_acc[r] = 0.0f;
        dbias_acc[r] = 0.0f;
    }
    const float* __restrict__ A = static_cast<const float*>(f.A);
    float dl_nx[R];                                    // softplus(delta + bias) at the first token of the next segment
#pragma unroll
    for (int r = 0; r < R; ++r) {
        dl_nx[r] = 0.0f;
        if (t_next < L) {
            const float raw = to_f32<T>((static_cast<const T*>(f.delta) + b * f.delta_batch_stride + d[r] * f.delta_d_stride)[t_next]) + bias[r];
            dl_nx[r] = f.delta_softplus ? softplus_ref(raw) : raw;
        }
    }
    for (int i = lane; i < N * R; i += kWave) {
        const int n = i / R, r = i - n * R;
        const float a = A[d[r] * f.A_d_stride + n * f.A_dstate_stride];
        // shadow slots (odd channel count, surplus waves) get no inflow either: with dy == 0 their g stays 0
        VIVIM_REC(n, GCAR, r) = (sg.S > 1 && active && r < nvalid)
                                    ? sg.gin[(((int64_t)b * f.dim + d[r]) * sg.S + seg) * N + n] : 0.0f;
        VIVIM_REC(n, AFIRST, r) = fast_exp2((r == 0 ? dl_nx[0] : dl_nx[1]) * a * kLog2e);   // exp2(0) = 1 past the end
        VIVIM_REC(n, A2VAL, r) = a * kLog2e;
        VIVIM_REC(n, DAACC, r) = 0.0f;
    }
    wave_lds_fence();
    // the (token, dB|dC) element this thread reduces after every state
    const int nck = (L + kChunk - 1) / kChunk;                     // checkpoint rows of x (forward kernel's contract)

    const T* __restrict__ uB = static_cast<const T*>(f.u) + b * f.u_batch_stride;
    const T* __restrict__ dlB = static_cast<const T*>(f.delta) + b * f.delta_batch_stride;
    const T* __restrict__ doB = static_cast<const T*>(p.dout) + b * p.dout_batch_stride;
    const T* __restrict__ zB = HAS_Z ? static_cast<const T*>(f.z) + b * f.z_batch_stride : nullptr;
    const T* __restrict__ oB = HAS_Z ? static_cast<const T*>(f.out) + b * f.out_batch_stride : nullptr;
    const T* __restrict__ Bv = static_cast<const T*>(f.B) + b * f.B_batch_stride + g * f.B_group_stride;
    const T* __restrict__ Cv = static_cast<const T*>(f.C) + b * f.C_batch_stride + g * f.C_group_stride;
    float* __restrict__ dBg = static_cast<float*>(p.dB) + b * p.dB_batch_stride + g * p.dB_group_stride;
    float* __restrict__ dCg = static_cast<float*>(p.dC) + b * p.dC_batch_stride + g * p.dC_group_stride;
    const float* __restrict__ xck = static_cast<const float*>(f.x);

    for (int step = s_hi - 1; step >= s_lo; --step) {
        const int t0 = step * TILE + lane * K;
        const bool in = t0 < L;                         // host guarantees L % K == 0: all-in or all-out
        VIVIM_STAMP(nsteps - 1 - step, 0, wave, lane);
        // forward checkpoints entering this step, one float per (state, channel)
        for (int i = lane; i < N * R; i += kWave) {
            const int n = i / R, r = i - n * R;
            VIVIM_REC(n, HCK, r) = step > 0 ? xck[(((int64_t)b * f.dim + d[r]) * nck + (step * (TILE / kChunk) - 1)) * N + n] : 0.0f;
        }
        f2 dl[K], w[K], dy[K], S1[K], S2[K], dsum = {0.0f, 0.0f};
        // (loads are issued unconditionally, from the row's first vector for a lane beyond the row, and zeroed by a select: under
        // `if (in)` hipcc waits for every outstanding load at each use -- cfg 3: -7 % for the backward, round 3)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float df[K], dof[K], uu[K];                 // u is re-read for ddelta at the end of the step: 8 VGPRs less
            unpack(load_vec_always<T, K>(uB + d[r] * f.u_d_stride + t0, in, uB + d[r] * f.u_d_stride), uu);
            unpack(load_vec_always<T, K>(dlB + d[r] * f.delta_d_stride + t0, in, dlB + d[r] * f.delta_d_stride), df);
            unpack(load_vec_always<T, K>(doB + d[r] * p.dout_d_stride + t0, in, doB + d[r] * p.dout_d_stride), dof);
            if (HAS_Z) {
                float zf[K], of[K], dzv[K];
                unpack(load_vec_always<T, K>(zB + d[r] * f.z_d_stride + t0, in, zB + d[r] * f.z_d_stride), zf);
                unpack(load_vec_always<T, K>(oB + d[r] * f.out_d_stride + t0, in, oB + d[r] * f.out_d_stride), of);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float sg = sigmoidf_fast(zf[k]);
                    dzv[k] = dof[k] * of[k] * sg * (1.0f + zf[k] * (1.0f - sg));     // bwd_kernel.cuh:186-191
                    dof[k] *= zf[k] * sg;
                }
                // shadow slots (r >= nvalid: odd channel count) must not store: their dy is zeroed, so their du / ddelta are 0
                store_vec<T, K>(static_cast<T*>(p.dz) + b * p.dz_batch_stride + d[r] * p.dz_d_stride + t0, in && active && r < nvalid, dzv);
                if (f.out_z) {                                                        // bwd_kernel.cuh:193-204
                    float oz[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) oz[k] = of[k] * zf[k] * sigmoidf_fast(zf[k]);
                    store_vec<T, K>(static_cast<T*>(f.out_z) + b * f.out_z_batch_stride + d[r] * f.out_z_d_stride + t0, in && active && r < nvalid, oz);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float raw = df[k] + bias[r];
                const float sp = f.delta_softplus ? softplus_ref(raw) : raw;
                const float dlv = in ? sp : 0.0f;       // padded tokens: identity maps both ways
                dl[k][r] = dlv;
                dsum[r] += dlv;
                w[k][r] = dlv * uu[k];
                dy[k][r] = dof[k] * msk[r];            // shadow slots / surplus waves: g == 0, dB = dC = dA = 0
                dD_acc[r] = fmaf(dof[k], uu[k], dD_acc[r]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { S1[k] = f2{0.0f, 0.0f}; S2[k] = f2{0.0f, 0.0f}; }
        wave_lds_fence();
        VIVIM_STAMP(nsteps - 1 - step, 1, wave, lane);
        {
            RawK<T, K> Braw = load_vec_always<T, K>(Bv + t0, in, Bv);
            RawK<T, K> Craw = load_vec_always<T, K>(Cv + t0, in, Cv);
#pragma unroll 1
            for (int n = 0; n < N; ++n) {
                float Bn[K], Cn[K];
                unpack(Braw, Bn);
                unpack(Craw, Cn);
                {   // next state's rows fly during this state (always issued, so vmcnt stays countable); two states
                    // ahead measured the same
                    const bool nx = in && (n + 1 < N);
                    Braw = load_vec_always<T, K>(Bv + (n + 1) * f.B_dstate_stride + t0, nx, Bv);
                    Craw = load_vec_always<T, K>(Cv + (n + 1) * f.C_dstate_stride + t0, nx, Cv);
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 2, wave, lane);
                // the state's record: checkpoint, g carry, first decay of the step to the right, A * log2e (pairs)
                const f2* q2 = reinterpret_cast<const f2*>(rec + n * kRec * R);
                const f2 hck = q2[HCK], gcar = q2[GCAR], afirst = q2[AFIRST], a2v = q2[A2VAL];
                // ---- forward re-scan ----
                f2 a[K], hs[K], wB[K];
                f2 P = exp2_2(dsum * a2v), H = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    a[k] = exp2_2(dl[k] * a2v);
                    wB[k] = w[k] * Bn[k];
                    H = fma2(a[k], H, wB[k]);
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 3, wave, lane);
                {
                    float P0 = P.x, H0 = H.x, P1 = P.y, H1 = H.y;
                    wave_scan2_affine_fwd(P0, H0, P1, H1);
                    P = f2{P0, P1};
                    H = f2{H0, H1};
                }
                {
                    const f2 hend = fma2(P, hck, H);
                    f2 h = dpp_mov2<kDppWaveShr1>(hck, hend);                // state entering this lane
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        h = fma2(a[k], h, wB[k]);
                        hs[k] = h;
                    }
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 4, wave, lane);
                // ---- reverse scan of g_t = a_{t+1} g_{t+1} + C_t dy_t ----
                const f2 an = dpp_mov2<kDppWaveShl1>(afirst, a[0]);          // decay of the token right of this lane
                f2 cdy[K], Pr = an, G = {0.0f, 0.0f};
#pragma unroll
                for (int k = K - 1; k >= 0; --k) {
                    const f2 al = k == K - 1 ? an : a[k + 1];
                    cdy[k] = dy[k] * Cn[k];
                    G = fma2(al, G, cdy[k]);
                    if (k < K - 1) Pr *= al;
                }
                f2 gfirst;
                {
                    float P0 = Pr.x, G0 = G.x, P1 = Pr.y, G1 = G.y;
                    wave_scan2_affine_rev_rows(P0, G0, P1, G1);
                    wave_scan_rev_join(P0, G0, lane);
                    wave_scan_rev_join(P1, G1, lane);
                    gfirst = fma2(f2{P0, P1}, gcar, f2{G0, G1});             // g at this lane's first token
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 5, wave, lane);
                f2 gc = dpp_mov2<kDppWaveShl1>(gcar, gfirst);                // g at the token right of this lane
                f2 dA_part = {0.0f, 0.0f};
                float dBv[K], dCv[K];
#pragma unroll
                for (int k = K - 1; k >= 0; --k) {
                    const f2 al = k == K - 1 ? an : a[k + 1];
                    gc = fma2(al, gc, cdy[k]);                               // g_t
                    const f2 t = gc * (hs[k] - wB[k]);                       // g_t * a_t h_{t-1}
                    S1[k] = fma2(gc, f2{Bn[k], Bn[k]}, S1[k]);               // du = D dy + d * S1
                    S2[k] = fma2(t, a2v, S2[k]);                             // dd = u * S1 + ln2 * S2
                    dA_part = fma2(t, dl[k], dA_part);
                    const f2 db = gc * w[k], dc = dy[k] * hs[k];             // shadow slots: dy == 0, so gc == 0
                    dBv[k] = db.x + db.y;                                    // the pair's sum
                    dCv[k] = dc.x + dc.y;
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 6, wave, lane);
                if (DA_LDS) {
                    dAl[n * kWave + lane] += dA_part;                        // wave-private: no barrier
                    if (lane == 0) {                                         // lane 0 already holds both values
                        f2* qq = reinterpret_cast<f2*>(rec + n * kRec * R);
                        qq[GCAR] = gfirst;                                   // g at this step's first token
                        qq[AFIRST] = a[0];
                    }
                } else {
                    const float dA0 = read_lane(wave_sum_dpp_to63(dA_part.x), 63);
                    const float dA1 = read_lane(wave_sum_dpp_to63(dA_part.y), 63);
                    if (lane == 0) {
                        f2* qq = reinterpret_cast<f2*>(rec + n * kRec * R);
                        qq[GCAR] = gfirst;
                        qq[AFIRST] = a[0];
                        qq[DAACC] += f2{dA0, dA1};
                    }
                }
                {
                    float* sl = slots + ((n & (NBUF - 1)) * W + wave) * SLOT + lane;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        sl[k * kWave] = dBv[k];
                        sl[(K + k) * kWave] = dCv[k];
                    }
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 7, wave, lane);
                lds_barrier();
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 8, wave, lane);
#pragma unroll
                for (int j = 0; j < EPT; ++j) {
                    // fixed-order sum over the workgroup's 8 channel pairs, then one fp32 atomic per element
                    const int e = tid + j * (W * kWave);
                    const int e_tok = e & (TILE - 1), e_isC = e / TILE;
                    const float* sp = slots + (n & (NBUF - 1)) * W * SLOT +
                                      (e_isC * K + (e_tok & (K - 1))) * kWave + (e_tok / K);
                    float acc = sp[0];
#pragma unroll
                    for (int wv = 1; wv < W; ++wv) acc += sp[wv * SLOT];
                    // fp32 sum over workgroups (bwd_kernel.cuh:312-313).  Unconditional: elements past the end
                    // carry acc == 0 (their dy and delta*u are 0) and are wrapped onto distinct valid tokens.
                    const int tq = step * TILE + e_tok;
                    const int t = tq < L ? tq : tq % L;
#if defined(BW_ABL) && BW_ABL == 1               // timing experiment (tools/abl.sh bwdbuild): no dB / dC atomics -- wrong results
                    asm volatile("" : : "v"(acc), "v"(t));
#else
                    atomicAdd((e_isC ? dCg + n * p.dC_dstate_stride : dBg + n * p.dB_dstate_stride) + t, acc);
#endif
                }
                if (n == 1) VIVIM_STAMP(nsteps - 1 - step, 9, wave, lane);
            }
        }
        VIVIM_STAMP(nsteps - 1 - step, 11, wave, lane);
        // ---- per-channel outputs of the step ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float duv[K], ddv[K], uu[K];
            unpack(load_vec_always<T, K>(uB + d[r] * f.u_d_stride + t0, in, uB + d[r] * f.u_d_stride), uu);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                duv[k] = fmaf(dl[k][r], S1[k][r], Dv[r] * dy[k][r]);
                ddv[k] = fmaf(uu[k], S1[k][r], S2[k][r] * kLn2);         // S2 was accumulated with A * log2e
            }
            if (f.delta_softplus) {                                           // bwd_kernel.cuh:439-452
                // sigmoid(raw) is applied for raw <= 20 only (the reference leaves ddelta unscaled above)
                float df[K];
                unpack(load_vec_always<T, K>(dlB + d[r] * f.delta_d_stride + t0, in, dlB + d[r] * f.delta_d_stride), df);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float raw = df[k] + bias[r];
                    if (raw <= 20.0f) ddv[k] *= sigmoidf_fast(raw);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) dbias_acc[r] += in ? ddv[k] : 0.0f;
            store_vec<T, K>(static_cast<T*>(p.du) + b * p.du_batch_stride + d[r] * p.du_d_stride + t0, in && active && r < nvalid, duv);
            store_vec<T, K>(static_cast<T*>(p.ddelta) + b * p.ddelta_batch_stride + d[r] * p.ddelta_d_stride + t0, in && active && r < nvalid, ddv);
        }
        VIVIM_STAMP(nsteps - 1 - step, 10, wave, lane);
    }
    if (active) {
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
        for (int i = lane; i < N * R; i += kWave) {
            const int n = i / R, r = i - n * R;
            float tot = VIVIM_REC(n, DAACC, r);
            if (DA_LDS) {                                                     // sum the 64 per-lane partials of (n, r)
                const float* q = reinterpret_cast<const float*>(dAl + n * kWave) + r;
                tot = 0.0f;
                for (int j = 0; j < kWave; ++j) tot += q[j * R];
            }
            if (r < nvalid)
                atomicAdd(static_cast<float*>(p.dA) + (d0 + r) * p.dA_d_stride + n * p.dA_dstate_stride, tot);
        }
    }
#undef VIVIM_REC
}

// ---- pre-pass of the token-axis split: per (batch, channel, segment >= 1, state) the reverse aggregate ----------
//   agg  = g at the segment's first token t0 for zero inflow from the right
//        = sum_{t in seg} (prod_{tau = t0+1 .. t} a_tau) * C_{n,t} * dy_t,   a_tau = exp(delta_tau * A_n)
//        = sum_t exp2(A_n * log2e * (cdelta_t - delta_{t0})) * C_{n,t} * dy_t      (cdelta = inclusive prefix of delta)
//   dsum = sum_{t in seg} delta_{t+1}
// One prefix sum of delta per channel serves all N states, so a state update costs mul + exp + fma here instead of
// the two scans of the main kernel (SQ counters: 18 VALU lane-instructions per update when the pre-pass was the
// main kernel with its outputs switched off).  exp2 of a large negative argument underflows to 0
// exactly where the true weight is negligible.  A wave owns a channel pair and one segment; lane = K consecutive
// tokens of a 256-token step; the running sums live in wave-private LDS [state][channel][lane] because the state
// loop cannot be unrolled for a run-time N; they are reduced over the lanes once per segment.
constexpr int kPreW = 4;           // independent waves per workgroup (fewer when N * 512 bytes per wave would pass 64 KB)
// NS = 16: the state count is known at compile time (Vivim), the 32 running sums of a lane stay in registers and the
// state loop is unrolled on packed channel pairs; NS = 0: any N, running sums in LDS.
template <typename T, int K, bool HAS_Z, int NS>
__global__ void __launch_bounds__(kPreW * kWave) ssm_bwd_prepass_kernel(const vivim_ssm_bwd_params p, const BwdSeg sg) {
    constexpr int R = kBwR;
    constexpr int TILE = kWave * K;
    const vivim_ssm_fwd_params& f = p.f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int N = f.dstate, L = f.seqlen;
    const int cpg = f.dim / f.n_groups;
    const int ppg = (cpg + R - 1) / R;                 // channel pairs per B/C group
    const int nw = blockDim.x >> 6;                    // waves in this workgroup
    const int bpg = (ppg + nw - 1) / nw;
    const int g = blockIdx.x / bpg;
    const int pair = (blockIdx.x - g * bpg) * nw + wave;
    if (pair >= ppg) return;                           // waves are independent: no workgroup barrier below
    const int d0 = g * cpg + pair * R;
    const int nvalid = min(R, (g + 1) * cpg - d0);
    const int nsteps = (L + TILE - 1) / TILE;
    const int seg = blockIdx.z + 1;                    // segment 0 has no left neighbour to feed
    const int s_lo = seg * sg.seg_steps, s_hi = min(nsteps, s_lo + sg.seg_steps);
    const int t_next = s_hi * TILE;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* acc = smem + wave * (N * R * kWave);        // [n][r][lane]
    if (NS == 0)
        for (int i = lane; i < N * R * kWave; i += kWave) acc[i] = 0.0f;
    f2 racc[NS > 0 ? NS : 1];
#pragma unroll
    for (int n = 0; n < (NS > 0 ? NS : 1); ++n) racc[n] = f2{0.0f, 0.0f};

    int d[R];
    float bias[R], a2v[R];                             // a2v: lane n holds A[d][n] * log2e
#pragma unroll
    for (int r = 0; r < R; ++r) {
        d[r] = d0 + min(r, nvalid - 1);
        bias[r] = f.delta_bias ? static_cast<const float*>(f.delta_bias)[d[r]] : 0.0f;
        a2v[r] = lane < N ? static_cast<const float*>(f.A)[d[r] * f.A_d_stride + lane * f.A_dstate_stride] * kLog2e : 0.0f;
    }
    const T* __restrict__ dlB = static_cast<const T*>(f.delta) + b * f.delta_batch_stride;
    const T* __restrict__ doB = static_cast<const T*>(p.dout) + b * p.dout_batch_stride;
    const T* __restrict__ zB = HAS_Z ? static_cast<const T*>(f.z) + b * f.z_batch_stride : nullptr;
    const T* __restrict__ Cv = static_cast<const T*>(f.C) + b * f.C_batch_stride + g * f.C_group_stride;
    wave_lds_fence();

    float base[R] = {0.0f, 0.0f}, dfirst[R] = {0.0f, 0.0f};
    // The next step's delta / dout / z vectors are requested before this step's arithmetic, unconditionally (a lane beyond the
    // row or a step beyond the segment reads its row's first vector and gets zeros): loads under `if (in)` made every later
    // wait a full drain and left each step one exposed memory round trip (round 3).
    RawK<T, K> ndl[R], ndo[R], nz[R];
    auto request = [&](int step) __attribute__((always_inline)) {
        const int t0 = step * TILE + lane * K;
        const bool in = step < s_hi && t0 < L;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ndl[r] = load_vec_always<T, K>(dlB + d[r] * f.delta_d_stride + t0, in, dlB + d[r] * f.delta_d_stride);
            ndo[r] = load_vec_always<T, K>(doB + d[r] * p.dout_d_stride + t0, in, doB + d[r] * p.dout_d_stride);
            if (HAS_Z) nz[r] = load_vec_always<T, K>(zB + d[r] * f.z_d_stride + t0, in, zB + d[r] * f.z_d_stride);
        }
    };
    // (f32 vectors of 8 tokens: the second set of registers costs a wave per SIMD, cfg 3 was 2 % slower with it: request and use)
    constexpr bool kAhead = sizeof(T) * K <= 16;
    if (kAhead) request(s_lo);
    for (int step = s_lo; step < s_hi; ++step) {
        const int t0 = step * TILE + lane * K;
        const bool in = t0 < L;                         // L % K == 0 (host): a lane is all-in or all-out
        float c[R][K], dy[R][K];
        RawK<T, K> cdl[R], cdo[R], cz[R];
        if (!kAhead) request(step);
#pragma unroll
        for (int r = 0; r < R; ++r) { cdl[r] = ndl[r]; cdo[r] = ndo[r]; if (HAS_Z) cz[r] = nz[r]; }
        if (kAhead) request(step + 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float df[K];
            unpack(cdl[r], df);
            unpack(cdo[r], dy[r]);
            if (HAS_Z) {
                float zf[K];
                unpack(cz[r], zf);
#pragma unroll
                for (int k = 0; k < K; ++k) dy[r][k] *= zf[k] * sigmoidf_fast(zf[k]);
            }
            float run = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float raw = df[k] + bias[r];
                const float sp = f.delta_softplus ? softplus_ref(raw) : raw;
                run += in ? sp : 0.0f;                  // padded tokens: delta 0 (and dy 0: they add nothing)
                c[r][k] = run;
            }
            if (step == s_lo) {
                dfirst[r] = read_lane(c[r][0], 0);      // delta of the segment's first token
                base[r] = -dfirst[r];
            }
            const float incl = wave_sum_dpp_to63(run);  // inclusive prefix of the lane totals
            const float off = incl - run + base[r];
#pragma unroll
            for (int k = 0; k < K; ++k) c[r][k] += off;
            base[r] += read_lane(incl, 63);
        }
        RawK<T, K> Craw = load_vec_always<T, K>(Cv + t0, in, Cv);
        if (NS > 0) {
            f2 cp[K], dyp[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { cp[k] = f2{c[0][k], c[1][k]}; dyp[k] = f2{dy[0][k], dy[1][k]}; }
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                float Cn[K];
                unpack(Craw, Cn);
                Craw = load_vec_always<T, K>(Cv + (n + 1) * f.C_dstate_stride + t0, in && (n + 1 < NS), Cv);
                const f2 A2 = {read_lane(a2v[0], n), read_lane(a2v[1], n)};
#pragma unroll
                for (int k = 0; k < K; ++k) racc[n] = fma2(exp2_2(A2 * cp[k]), dyp[k] * Cn[k], racc[n]);
            }
        } else {
#pragma unroll 1
            for (int n = 0; n < N; ++n) {
                float Cn[K];
                unpack(Craw, Cn);
                Craw = load_vec_always<T, K>(Cv + (n + 1) * f.C_dstate_stride + t0, in && (n + 1 < N), Cv);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float A2 = read_lane(a2v[r], n);
                    float* q = acc + (n * R + r) * kWave + lane;
                    float v = *q;
#pragma unroll
                    for (int k = 0; k < K; ++k) v = fmaf(fast_exp2(A2 * c[r][k]), Cn[k] * dy[r][k], v);
                    *q = v;
                }
            }
        }
    }
    if (NS > 0) {                                       // hand the register sums to the common reduction below
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            acc[(n * R + 0) * kWave + lane] = racc[n].x;
            acc[(n * R + 1) * kWave + lane] = racc[n].y;
        }
    }
    wave_lds_fence();
    // lane pair (2i, 2i+1) sums the 64 partials of row i = (n, r): 32 rows per sweep
    for (int row0 = 0; row0 < N * R; row0 += kWave / 2) {
        const int row = row0 + (lane >> 1);
        float v = 0.0f;
        if (row < N * R) {
            const float* q = acc + row * kWave + (lane & 1) * (kWave / 2);
#pragma unroll 8
            for (int j = 0; j < kWave / 2; ++j) v += q[j];
        }
        v += __shfl_xor(v, 1, kWave);
        const int n = row / R, r = row - n * R;
        if (row < N * R && (lane & 1) == 0 && r < nvalid)
            sg.agg[(((int64_t)b * f.dim + d0 + r) * sg.S + seg) * N + n] = v;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float dl_nx = 0.0f;                             // softplus(delta + bias) at the first token of the next segment
        if (t_next < L) {
            const float raw = to_f32<T>((dlB + d[r] * f.delta_d_stride)[t_next]) + bias[r];
            dl_nx = f.delta_softplus ? softplus_ref(raw) : raw;
        }
        // base = (sum of delta over the segment) - delta_first  ->  sum_{t in seg} delta_{t+1}
        if (lane == 0 && r < nvalid) sg.dsum[((int64_t)b * f.dim + d[r]) * sg.S + seg] = base[r] + dl_nx;
    }
}

// gin[seg-1] = exp2(A2 * dsum[seg]) * gin[seg] + agg[seg], right to left; one thread per (batch, channel, state)
__global__ void ssm_bwd_carry_kernel(const vivim_ssm_bwd_params p, const BwdSeg sg) {
    const vivim_ssm_fwd_params& f = p.f;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = f.dstate;
    if (i >= (int64_t)f.batch * f.dim * N) return;
    const int n = (int)(i % N);
    const int64_t bd = i / N;
    const int dch = (int)(bd % f.dim);
    const float A2 = static_cast<const float*>(f.A)[dch * f.A_d_stride + n * f.A_dstate_stride] * kLog2e;
    float g = 0.0f;
    const int S = sg.S;
    sg.gin[(bd * S + S - 1) * N + n] = 0.0f;
    // The chain's operands do not depend on the chain: eight segments' worth are fetched together (a dependent walk over global
    // memory costs one memory latency per segment: 11 us for 8 segments at cfg 2's stage 0, the launch floor is 5).
    for (int i0 = 0; i0 < S - 1; i0 += 8) {
        float ag[8], ds[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int s = S - 1 - (i0 + q);
            const bool ok = s >= 1;
            ag[q] = ok ? sg.agg[(bd * S + s) * N + n] : 0.0f;
            ds[q] = ok ? sg.dsum[bd * S + s] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int s = S - 1 - (i0 + q);
            if (s >= 1) {
                g = fmaf(fast_exp2(A2 * ds[q]), g, ag[q]);
                sg.gin[(bd * S + s - 1) * N + n] = g;
            }
        }
    }
}

// Tokens per lane of the fast backward.  Most of a state iteration is scan machinery whose cost does not depend on K
// (DESIGN.md 4.3), so 8 tokens per lane (512-token steps, 229-243 VGPRs of the 256 available at two waves per SIMD, no
// scratch) nearly halve the instructions per state update: measured -4 ... -25 % on Vivim's bf16 shapes, -4 ... -9 % on
// the fp32 ones -- except where the last 512-token step would be mostly empty (L = 1280: three steps, 20 % of the
// slots idle, +6 % against five full 256-token steps).  A pure function of the shape: the workspace query and the
// launch must agree.
static int bwd_tokens_per_lane(int /*itype*/, int seqlen) {
    const int64_t slots8 = (int64_t)((seqlen + 511) / 512) * 512, slots4 = (int64_t)((seqlen + 255) / 256) * 256;
    return slots8 * 100 > slots4 * 115 ? 4 : 8;
}

// How the token axis is cut.  One 8-wave workgroup is resident per CU and all workgroups of a launch take the same
// time, so the launch runs in rounds of `ncu` workgroups: cost(S) ~ ceil(base * S / ncu) * (ceil(nsteps / S) + fixed), with
// base = workgroups before the split and `fixed` ~ 0.3 step for a workgroup's prologue / final reductions.  Measured
// against the former fixed target of 1024 workgroups: per-direction stage 0 287 -> 243 us (S 40 -> 10), grouped
// stage 2 255 -> 227 us (S 3 -> 2); grouped stage 0 unchanged (S 14 -> 7..10).  Ties go to the smaller S (less pre-pass).
static int device_cu_count() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;                                    // MI355X; also what a GPU-less build host reports
        return v;
    }();
    return n;
}

// Token-axis cut for workgroups of W waves: minimise rounds-of-the-chip x (steps per segment + fixed cost).
static void bwd_segmentation(const vivim_ssm_fwd_params& f, int W, int& S, int& seg_steps, int64_t& workgroups) {
    const int tile = kWave * bwd_tokens_per_lane(f.itype, f.seqlen);
    const int nsteps = (f.seqlen + tile - 1) / tile;
    const int cpg = f.dim / f.n_groups;
    const int ppg = (cpg + kBwR - 1) / kBwR;
    const int64_t base = (int64_t)((ppg + W - 1) / W) * f.n_groups * f.batch;
    // workgroups in flight per CU: two waves per SIMD at K = 8 (240-248 VGPRs), three at K = 4 with 4-wave workgroups (145)
    const int slots = device_cu_count() * (tile == 4 * kWave && W == 4 ? 3 : kBwWmax / W);
    int best = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= nsteps && s <= 64; ++s) {
        const int steps = (nsteps + s - 1) / s;
        if ((nsteps + steps - 1) / steps != s) continue;            // not a distinct cut
        const double rounds = (double)((base * s + slots - 1) / slots);
        // a cut adds the pre-pass over all segments but the first (~0.4 of a main-pass step per step) and the carry kernel
        const double cost = rounds * (steps * (1.0 + 0.4 * (s - 1) / s) + 0.3);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    seg_steps = (nsteps + best - 1) / best;
    S = (nsteps + seg_steps - 1) / seg_steps;
    workgroups = base * S;
}

// Waves per workgroup.  Eight waves share one B/C tile and one set of dB/dC atomics (half as many atomics per address as
// two 4-wave workgroups) and win on short rows that fit the chip in one round (D 1024, L 320: 55 us with 8 waves, 73 us
// with 4; D 640, L 1280: 96 vs 102 us).  Two independent 4-wave workgroups per CU win when a workgroup walks several
// steps -- one runs while the other waits at its per-state barrier -- and past one round, where the last round is cut
// finer (MI355X, cfg 2 grouped stages 0-3: 597/314/225/147 -> 583/293/198/145 us; cfg 3 stages 0-2:
// 2241/1103/779 -> 2151/981/625 us).  vivim_set_tuning(1, 1 / 2) pins 8 / 4.
struct BwdPlan { int W, S, seg_steps; };
static BwdPlan bwd_plan(const vivim_ssm_fwd_params& f) {
    BwdPlan q;
    int64_t wgs = 0;
    q.W = kBwWmax;
    bwd_segmentation(f, q.W, q.S, q.seg_steps, wgs);
    const int tv = tuning_bwd_variant();
    if (tv == 2 || (tv != 1 && (wgs > device_cu_count() || q.seg_steps >= 4))) {
        q.W = 4;
        bwd_segmentation(f, q.W, q.S, q.seg_steps, wgs);
    }
    return q;
}

// The lanes = states family (scan_ls.hip) takes every shape whose checkpoints were written for it (scan_ckpt_len), unless
// the tuning selector pins one of the kernels of this file (1 / 2: fast kernel with 8 / 4 waves, 3: generic; 4 pins its first-generation main kernel, 5 the second-generation one of scan_ls2.hip).
static bool bwd_takes_ls(const vivim_ssm_fwd_params& f) {
    const int tv = tuning_bwd_variant();
    return ls_shape_ok(f) && scan_ckpt_len(f) == ls_ckpt_len(f) && (tv == 0 || tv == 4 || tv == 5);
}

static size_t fast_bwd_workspace_bytes(const vivim_ssm_fwd_params& f);
size_t scan_bwd_workspace_bytes(const vivim_ssm_fwd_params& f) {
    if (bwd_takes_ls(f)) return ls_bwd_workspace_bytes(f);
    return fast_bwd_workspace_bytes(f);
}
static size_t fast_bwd_workspace_bytes(const vivim_ssm_fwd_params& f) {
    if (!f.is_variable_B || !f.is_variable_C || f.dstate > 64) return 0;
    const BwdPlan q = bwd_plan(f);
    if (q.S <= 1) return 0;
    return ((size_t)f.batch * f.dim * q.S * (2 * f.dstate + 1)) * sizeof(float);
}

template <typename T, int K, int W>
static void launch_bwd_fast(const vivim_ssm_bwd_params& p, const BwdPlan& plan, hipStream_t stream) {
    const vivim_ssm_fwd_params& f = p.f;
    const int cpg = f.dim / f.n_groups;
    const int ppg = (cpg + kBwR - 1) / kBwR;
    const int bpg = (ppg + W - 1) / W;
    BwdSeg sg = {1, (f.seqlen + kWave * K - 1) / (kWave * K), nullptr, nullptr, nullptr};
    const size_t need = fast_bwd_workspace_bytes(f);
    if (need && p.workspace && (size_t)p.workspace_bytes >= need) {
        sg.S = plan.S;
        sg.seg_steps = plan.seg_steps;
        const size_t nbd = (size_t)f.batch * f.dim * sg.S;
        sg.agg = static_cast<float*>(p.workspace);
        sg.gin = sg.agg + nbd * f.dstate;
        sg.dsum = sg.gin + nbd * f.dstate;
    }
    const dim3 block(W * kWave);
    // LDS at W = 8: 2 slot buffers (32 KB) + records (N * 512 B) + per-lane dA partials when used (N * 4 KB): 104 KB at
    // N = 16, half of that at W = 4 -- the registers (two waves per SIMD) bound the residency either way
    // per-lane dA partials in LDS pay off once a workgroup walks several steps (grouped stage 0, 6 steps: 724 -> 698 us);
    // for one or two steps their zero-fill and final reduction cost more than the per-state wave reductions they
    // replace (stage 3: 74 -> 80 us)
    const bool da_lds = f.dstate <= 16 && sg.seg_steps >= 2;
    const size_t smem = ((size_t)2 * W * 2 * K * kWave + (size_t)W * f.dstate * kBwR * kRec +
                         (da_lds ? (size_t)W * f.dstate * kWave * 2 : 0)) * sizeof(float);
    if (sg.S > 1) {
        const size_t per_wave = (size_t)f.dstate * kBwR * kWave * sizeof(float);            // 8 KB at N = 16, 32 KB at N = 64
        int nw = (int)((size_t)65536 / per_wave);
        nw = nw > kPreW ? kPreW : (nw < 1 ? 1 : nw);
        dim3 gpre(((ppg + nw - 1) / nw) * f.n_groups, f.batch, sg.S - 1);
        if (f.dstate == 16) {
            if (f.z) hipLaunchKernelGGL((ssm_bwd_prepass_kernel<T, K, true, 16>), gpre, dim3(nw * kWave), nw * per_wave, stream, p, sg);
            else     hipLaunchKernelGGL((ssm_bwd_prepass_kernel<T, K, false, 16>), gpre, dim3(nw * kWave), nw * per_wave, stream, p, sg);
        } else {
            if (f.z) hipLaunchKernelGGL((ssm_bwd_prepass_kernel<T, K, true, 0>), gpre, dim3(nw * kWave), nw * per_wave, stream, p, sg);
            else     hipLaunchKernelGGL((ssm_bwd_prepass_kernel<T, K, false, 0>), gpre, dim3(nw * kWave), nw * per_wave, stream, p, sg);
        }
        const int64_t nthr = (int64_t)f.batch * f.dim * f.dstate;
        hipLaunchKernelGGL(ssm_bwd_carry_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, p, sg);
    }
    dim3 grid(bpg * f.n_groups, f.batch, sg.S);
    auto launch = [&](auto kernel) {
        // The four <HAS_Z, DA_LDS> instantiations share one function-pointer TYPE, so a cache inside this generic lambda would
        // be shared between them (and between devices): the attribute is simply set on every launch that needs it -- a
        // host-side table write, no synchronisation.
        if (smem > 65536)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kernel, grid, block, smem, stream, p, sg);
    };
    if (f.z) { if (da_lds) launch(ssm_bwd_fast_kernel<T, K, true, W, true>); else launch(ssm_bwd_fast_kernel<T, K, true, W, false>); }
    else     { if (da_lds) launch(ssm_bwd_fast_kernel<T, K, false, W, true>); else launch(ssm_bwd_fast_kernel<T, K, false, W, false>); }
}

// The lanes = tokens pre-pass + carry on somebody else's cut of the token axis: S segments of seg_tokens tokens each, a
// multiple of 256 (scan_ls.hip uses it for the lanes = states main kernels on long rows: the closed-form pre-pass of this
// file runs at half the time of the recurrence form there -- cfg 3 grouped stage 0: 989 against 1931 us).  Same workspace
// layout and the same meaning of agg / dsum / gin.  The caller has checked 16-byte aligned rows of delta, dout, z and C.
template <typename T, int K>
static void launch_prepass_only(const vivim_ssm_bwd_params& p, const BwdSeg& sg, hipStream_t stream) {
    const vivim_ssm_fwd_params& f = p.f;
    const int cpg = f.dim / f.n_groups;
    const int ppg = (cpg + kBwR - 1) / kBwR;
    const size_t per_wave = (size_t)f.dstate * kBwR * kWave * sizeof(float);
    int nw = (int)((size_t)65536 / per_wave);
    nw = nw > kPreW ? kPreW : (nw < 1 ? 1 : nw);
    dim3 gpre(((ppg + nw - 1) / nw) * f.n_groups, f.batch, sg.S - 1);
    if (f.z) hipLaunchKernelGGL((ssm_bwd_prepass_kernel<T, K, true, 16>), gpre, dim3(nw * kWave), nw * per_wave, stream, p, sg);
    else     hipLaunchKernelGGL((ssm_bwd_prepass_kernel<T, K, false, 16>), gpre, dim3(nw * kWave), nw * per_wave, stream, p, sg);
    const int64_t nthr = (int64_t)f.batch * f.dim * f.dstate;
    hipLaunchKernelGGL(ssm_bwd_carry_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, p, sg);
}
bool fast_bwd_prepass(const vivim_ssm_bwd_params& p, int S, int seg_tokens, float* agg, float* gin, float* dsum, hipStream_t stream) {
    const vivim_ssm_fwd_params& f = p.f;
    if (S <= 1 || f.dstate != 16 || seg_tokens % 256 != 0) return false;
    const bool k8 = seg_tokens % 512 == 0 && f.seqlen % 8 == 0;
    if (!k8 && f.seqlen % 4 != 0) return false;
    const BwdSeg sg = {S, seg_tokens / (k8 ? 512 : 256), agg, dsum, gin};
    switch (f.itype) {
        case VIVIM_F32: if (k8) launch_prepass_only<float, 8>(p, sg, stream); else launch_prepass_only<float, 4>(p, sg, stream); break;
        case VIVIM_F16: if (k8) launch_prepass_only<f16_t, 8>(p, sg, stream); else launch_prepass_only<f16_t, 4>(p, sg, stream); break;
        case VIVIM_BF16: if (k8) launch_prepass_only<bf16_t, 8>(p, sg, stream); else launch_prepass_only<bf16_t, 4>(p, sg, stream); break;
        default: return false;
    }
    return true;
}

template <typename T, int K>
static bool try_bwd_fast_k(const vivim_ssm_bwd_params& p, hipStream_t stream) {
    const vivim_ssm_fwd_params& f = p.f;
    if (!f.is_variable_B || !f.is_variable_C || f.dstate > 64 || f.x == nullptr) return false;
    if (tuning_bwd_variant() == 3) return false;
    if (scan_ckpt_len(f) != kChunk) return false;          // this family reads one checkpoint row per kChunk tokens
    // unconditional K-element vectors: rows aligned to the vector size, seqlen a whole number of lanes
    const int64_t vb = K * (int64_t)sizeof(T) >= 16 ? 16 : K * (int64_t)sizeof(T);
    const int64_t epv = vb / (int64_t)sizeof(T);
    auto al = [&](const void* q) { return (reinterpret_cast<uintptr_t>(q) & (vb - 1)) == 0; };
    auto st = [&](int64_t e) { return e % epv == 0; };
    if (f.seqlen % K != 0 || !al(f.u) || !al(f.delta) || !al(f.B) || !al(f.C) || !al(p.dout) || !al(p.du) || !al(p.ddelta) ||
        !st(f.u_batch_stride) || !st(f.u_d_stride) || !st(f.delta_batch_stride) || !st(f.delta_d_stride) ||
        !st(p.dout_batch_stride) || !st(p.dout_d_stride) || !st(p.du_batch_stride) || !st(p.du_d_stride) ||
        !st(p.ddelta_batch_stride) || !st(p.ddelta_d_stride) || !st(f.B_batch_stride) || !st(f.B_group_stride) ||
        !st(f.B_dstate_stride) || !st(f.C_batch_stride) || !st(f.C_group_stride) || !st(f.C_dstate_stride))
        return false;
    if (f.z && (!al(f.z) || !al(f.out) || !al(p.dz) || !st(f.z_batch_stride) || !st(f.z_d_stride) ||
                !st(f.out_batch_stride) || !st(f.out_d_stride) || !st(p.dz_batch_stride) || !st(p.dz_d_stride) ||
                (f.out_z && (!al(f.out_z) || !st(f.out_z_batch_stride) || !st(f.out_z_d_stride)))))
        return false;
    // Two waves per SIMD (__launch_bounds__(W * 64, 2)): registers uncapped (144-152 VGPRs at K = 4, 229-248 at K = 8; one
    // 8-wave or two 4-wave workgroups per CU, 256 VGPRs available).  A 128-VGPR build of the K = 4 kernel (four waves per SIMD, two
    // workgroups per CU) measured ~12% faster but needs 28-48 bytes of scratch per lane, and hipcc (ROCm 7.2) may
    // place such a VGPR spill at the top of the join block of a divergent loop, BEFORE the s_or_b64 that
    // restores EXEC: the store then runs with EXEC = 0, nothing is saved, and the reload returns garbage (seen as
    // wrong gradients and a GPU memory fault in fp32 once an unrelated edit changed the allocation).  No kernel
    // of this library may use scratch: `make check-scratch` (part of the default build) enforces it.
    const BwdPlan plan = bwd_plan(f);
    if (plan.W == 4) launch_bwd_fast<T, K, 4>(p, plan, stream);
    else             launch_bwd_fast<T, K, kBwWmax>(p, plan, stream);
    return true;
}

template <typename T>
static bool try_bwd_fast(const vivim_ssm_bwd_params& p, hipStream_t stream) {
    if (bwd_tokens_per_lane(p.f.itype, p.f.seqlen) == 8) return try_bwd_fast_k<T, 8>(p, stream);
    return try_bwd_fast_k<T, 4>(p, stream);
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
        if (var) hipLaunchKernelGGL((ssm_bwd_generic_kernel<T, K, R, true, true>), grid, dim3(kBwdWaves * kWave), smem, stream, p, scan_ckpt_len(f));
        else     hipLaunchKernelGGL((ssm_bwd_generic_kernel<T, K, R, true, false>), grid, dim3(kBwdWaves * kWave), smem, stream, p, scan_ckpt_len(f));
    } else {
        if (var) hipLaunchKernelGGL((ssm_bwd_generic_kernel<T, K, R, false, true>), grid, dim3(kBwdWaves * kWave), smem, stream, p, scan_ckpt_len(f));
        else     hipLaunchKernelGGL((ssm_bwd_generic_kernel<T, K, R, false, false>), grid, dim3(kBwdWaves * kWave), smem, stream, p, scan_ckpt_len(f));
    }
}

bool ssm_bwd_dispatch(const vivim_ssm_bwd_params& p, hipStream_t s) {
    if (bwd_takes_ls(p.f) && try_ls_bwd(p, s)) return true;
    switch (p.f.itype) {
        case VIVIM_F32: if (!try_bwd_fast<float>(p, s)) launch_bwd<float, 4, 2>(p, s); return true;
        case VIVIM_F16: if (!try_bwd_fast<f16_t>(p, s)) launch_bwd<f16_t, 4, 2>(p, s); return true;
        case VIVIM_BF16: if (!try_bwd_fast<bf16_t>(p, s)) launch_bwd<bf16_t, 4, 2>(p, s); return true;
    }
    return false;
}

}  // namespace vivim
