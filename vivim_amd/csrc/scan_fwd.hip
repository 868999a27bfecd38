// scan_fwd.hip -- selective SSM scan, forward, gfx950 (wave64).
//
// Math (selective_scan_fwd_kernel.cuh:67-303; oracle: selective_scan_interface.py:86-152):
//   d_t = softplus(delta_t + bias);  h_{n,t} = exp(d_t A_n) h_{n,t-1} + d_t u_t B_{n,t}
//   out_t = sum_n C_{n,t} h_{n,t} + D u_t;   out_z_t = out_t * silu(z_t)
//
// Two kernels (NOT the reference's one-block-per-(batch,channel) BlockScan):
//
// ssm_fwd_nsplit_kernel -- the fast path (variable B/C, dstate a multiple of 8).
//   * a WORKGROUP owns R=2 channels of one batch element and walks the token axis in steps of 64*K
//     tokens; its NS waves all see the same tokens but each owns SPW = N/NS states, so the serial
//     recurrence along L costs nothing in parallelism: B*(D/2)*NS waves are in flight (the reference:
//     B*D blocks, serial in L) and no cross-workgroup carry exists;
//   * lane l holds tokens [l*K, l*K+K) of the step.  softplus(delta+bias), delta*u and the D*u term are
//     computed ONCE per token by the whole workgroup (R*K/NS tokens per thread) and shared through LDS;
//     each wave accumulates sum_n C h over its own states in registers and adds it into the y tile in
//     LDS (ds_add_f32); the epilogue (z gate, out/out_z stores) is again spread over all threads;
//   * per (state, channel pair): K-long in-register recurrence giving the lane's affine map h -> P h + H,
//     one wave64 scan of the 64 maps done with DPP row operations (two channels interleaved, no LDS
//     crossbar, no nops), carry between steps in registers;
//   * the state after every step goes to `x` (batch, dim, n_chunks, dstate): the backward's forward
//     re-scan restarts from it (role of the reference's x, selective_scan.cpp:307-313).
//
// ssm_fwd_generic_kernel -- any dstate, constant B/C: a wave owns R channels and ALL states (carry in a
//   per-wave LDS row, shuffle-based scan); slower, used only off Vivim's path.
#include <stdlib.h>
#include "common.cuh"

namespace vivim {

constexpr int kScanWaves = 4;   // waves per workgroup (independent of each other)

template <typename T, int K, int R, bool HAS_Z, bool VAR_BC>
__global__ void __launch_bounds__(kScanWaves * kWave) ssm_fwd_generic_kernel(const vivim_ssm_fwd_params p) {
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


// ------------------------------------------------------------------------------------------------
constexpr int kNsR = 2;            // channels per workgroup
constexpr int kLdsRow = 65;        // [slot][k][lane] tiles, rows padded to 65 floats: conflict-free both ways

template <typename T, int K, int NS, bool HAS_Z, int MINW>
__global__ void __launch_bounds__(NS * kWave, MINW) ssm_fwd_nsplit_kernel(const vivim_ssm_fwd_params p) {
    constexpr int R = kNsR;
    constexpr int TILE = kWave * K;
    constexpr int NT = NS * kWave;
    constexpr int IPT = R * TILE / NT;                 // tokens per thread in the shared prologue / epilogue
    static_assert(R * TILE % NT == 0 && IPT >= 1 && K % IPT == 0, "bad tiling");
    constexpr int TSZ = R * K * kLdsRow;               // floats per [r][k][lane] tile
    __shared__ float s_dl[TSZ];                        // softplus(delta + bias)   (0 on padded tokens)
    __shared__ float s_w[TSZ];                         // delta * u
    __shared__ float s_du[TSZ];                        // D * u
    __shared__ float s_part[NS * TSZ];                 // per-wave partial sum_n C h (summed in fixed order)
    __shared__ float s_carry[256 * R];                 // running state h[n][r] between steps (wave-private rows)
    __shared__ float s_A2[256 * R];                    // A[d][n] * log2(e)                    (wave-private rows)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: A / carry go scalar
    const int b = blockIdx.y;
    const int N = p.dstate, L = p.seqlen;
    const int cpg = p.dim / p.n_groups;
    const int wpg = (cpg + R - 1) / R;
    const int g = blockIdx.x / wpg;
    const int d0 = g * cpg + (blockIdx.x - g * wpg) * R;
    const int nvalid = min(R, (g + 1) * cpg - d0);
    const int SPW = N / NS;                            // states owned by this wave: [n0, n0 + SPW)
    const int n0 = wave * SPW;
    const int nsteps = (L + TILE - 1) / TILE;
    constexpr int CPS = TILE / kChunk;                 // checkpoint rows per step
    constexpr int LPC = kChunk / K;                    // lanes per checkpoint chunk
    const int nck = (L + kChunk - 1) / kChunk;

    // ---- this thread's slice of the shared per-token work ----
    const int i0 = tid * IPT;
    const int pr = i0 / TILE;                          // channel slot
    const int ptok = i0 - pr * TILE;                   // first token inside the tile
    const int plane = ptok / K, pj = ptok - plane * K; // where those tokens live in the [k][lane] tiles
    const int pd = d0 + min(pr, nvalid - 1);
    const float pD = p.D ? static_cast<const float*>(p.D)[pd] : 0.0f;
    const float pbias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[pd] : 0.0f;
    const T* __restrict__ pu = static_cast<const T*>(p.u) + b * p.u_batch_stride + pd * p.u_d_stride;
    const T* __restrict__ pdl = static_cast<const T*>(p.delta) + b * p.delta_batch_stride + pd * p.delta_d_stride;
    const T* __restrict__ pz = HAS_Z ? static_cast<const T*>(p.z) + b * p.z_batch_stride + pd * p.z_d_stride : nullptr;
    T* __restrict__ pout = static_cast<T*>(p.out) + b * p.out_batch_stride + pd * p.out_d_stride;
    T* __restrict__ poutz = HAS_Z ? static_cast<T*>(p.out_z) + b * p.out_z_batch_stride + pd * p.out_z_d_stride : nullptr;
    const int pbase = (pr * K + pj) * kLdsRow + plane;

    // per-token work of one step from raw u / delta: writes the three shared tiles
    auto prologue = [&](int step, const RawK<T, IPT>& ur, const RawK<T, IPT>& dr) {
        const bool pin = step * TILE + ptok < L;
        float uf[IPT], df[IPT];
        unpack(ur, uf);
        unpack(dr, df);
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const float raw = df[i] + pbias;
            const float sp = p.delta_softplus ? softplus_ref(raw) : raw;
            const float dlv = pin ? sp : 0.0f;          // padded tokens: exp2(0)=1 and drive 0 -> identity map
            s_dl[pbase + i * kLdsRow] = dlv;
            s_w[pbase + i * kLdsRow] = dlv * uf[i];
            s_du[pbase + i * kLdsRow] = pD * uf[i];
        }
    };

    const float* __restrict__ A = static_cast<const float*>(p.A);
    const T* __restrict__ Bv = static_cast<const T*>(p.B) + b * p.B_batch_stride + g * p.B_group_stride;
    const T* __restrict__ Cv = static_cast<const T*>(p.C) + b * p.C_batch_stride + g * p.C_group_stride;
    float* __restrict__ xck = static_cast<float*>(p.x);
    int d[R];
#pragma unroll
    for (int r = 0; r < R; ++r) d[r] = d0 + min(r, nvalid - 1);
    for (int i = lane; i < SPW * R; i += kWave) {
        const int n = n0 + i / R, r = i - (i / R) * R;
        s_carry[n0 * R + i] = 0.0f;
        s_A2[n0 * R + i] = A[d[r] * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;      // fwd_kernel.cuh:168-175
    }
    // B/C rows of the first two states of the NEXT step are fetched one step ahead (Bq/Cq)
    RawK<T, K> Bq[2], Cq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool ok = lane * K < L && i < SPW;
        Bq[i] = load_vec<T, K>(Bv + (n0 + i) * p.B_dstate_stride + lane * K, ok);
        Cq[i] = load_vec<T, K>(Cv + (n0 + i) * p.C_dstate_stride + lane * K, ok);
    }

    // host dispatch guarantees aligned rows and L % K == 0: lanes / threads are all-in or all-out
    prologue(0, load_vec<T, IPT>(pu + ptok, ptok < L), load_vec<T, IPT>(pdl + ptok, ptok < L));
    lds_barrier();
    for (int step = 0; step < nsteps; ++step) {
        const int t0 = step * TILE + lane * K;
        VIVIM_STAMP(step, 0, wave, lane);
        // ---- issue every global load of this step up front; they are consumed after LDS/VALU work ----
        const int tp = step * TILE + ptok;             // this thread's epilogue tokens
        const int tn = tp + TILE;                      // ... and next step's prologue tokens
        RawK<T, IPT> zr, unx, dnx;
        if (HAS_Z) zr = load_vec<T, IPT>(pz + tp, tp < L);
        const bool more = step + 1 < nsteps;
        unx = load_vec<T, IPT>(pu + tn, tn < L);
        dnx = load_vec<T, IPT>(pdl + tn, tn < L);
        const bool in = t0 < L;

        float dl[R][K], w[R][K], yp[R][K], dsum[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            dsum[r] = 0.0f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                dl[r][k] = s_dl[(r * K + k) * kLdsRow + lane];
                w[r][k] = s_w[(r * K + k) * kLdsRow + lane];
                yp[r][k] = 0.0f;
                dsum[r] += dl[r][k];
            }
        }
        // the rows prefetched one step ago become current; only now do we depend on vmcnt (the LDS reads
        // above were issued first), then the next step's first two states are requested
        RawK<T, K> Bc[2] = {Bq[0], Bq[1]}, Cc[2] = {Cq[0], Cq[1]};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = t0 + TILE < L && i < SPW;
            Bq[i] = load_vec<T, K>(Bv + (n0 + i) * p.B_dstate_stride + t0 + TILE, ok);
            Cq[i] = load_vec<T, K>(Cv + (n0 + i) * p.C_dstate_stride + t0 + TILE, ok);
        }
        VIVIM_STAMP(step, 1, wave, lane);
        // one state for the wave's R channels: lane maps, DPP scan, apply, checkpoint
        auto do_state = [&](int n, const RawK<T, K>& Braw, const RawK<T, K>& Craw) {
            float A2[R], cin[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                A2[r] = s_A2[n * R + r];
                cin[r] = s_carry[n * R + r];
            }
            float Bn[K], Cn[K];
            unpack(Braw, Bn);
            unpack(Craw, Cn);
            // lane map h -> P h + H over its K tokens: P = exp2(A2 * sum_k delta_k), H by the recurrence
            float a[R][K], P[R], H[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                P[r] = fast_exp2(dsum[r] * A2[r]);
                H[r] = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    a[r][k] = fast_exp2(dl[r][k] * A2[r]);
                    H[r] = fmaf(a[r][k], H[r], w[r][k] * Bn[k]);
                }
            }
            wave_scan2_affine_fwd(P[0], H[0], P[1], H[1]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float hend = fmaf(P[r], cin[r], H[r]);                 // state after this lane's last token
                float h = dpp_mov<kDppWaveShr1>(cin[r], hend);               // state entering this lane
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    h = fmaf(a[r][k], h, w[r][k] * Bn[k]);
                    yp[r][k] = fmaf(h, Cn[k], yp[r][k]);
                }
#pragma unroll
                for (int q = 0; q < CPS; ++q) {                              // state after every kChunk tokens
                    const float v = read_lane(hend, (q + 1) * LPC - 1);
                    const int row = step * CPS + q;
                    if (lane == 0 && r < nvalid && row < nck)
                        xck[(((int64_t)b * p.dim + d[r]) * nck + row) * N + n] = v;
                    if (q == CPS - 1) {                                      // state after the step
                        wave_lds_fence();
                        if (lane == 0) s_carry[n * R + r] = v;
                        wave_lds_fence();
                    }
                }
            }
        };
        do_state(n0, Bc[0], Cc[0]);
        VIVIM_STAMP(step, 2, wave, lane);
        if (SPW > 1) {
            RawK<T, K> Bx = load_vec<T, K>(Bv + (n0 + 2) * p.B_dstate_stride + t0, in && SPW > 2);
            RawK<T, K> Cx = load_vec<T, K>(Cv + (n0 + 2) * p.C_dstate_stride + t0, in && SPW > 2);
            do_state(n0 + 1, Bc[1], Cc[1]);
#pragma unroll 1
            for (int n = n0 + 2; n < n0 + SPW; ++n) {
                const RawK<T, K> Bn_ = Bx, Cn_ = Cx;
                const bool nx = in && (n + 1 < n0 + SPW);
                Bx = load_vec<T, K>(Bv + (n + 1) * p.B_dstate_stride + t0, nx);   // flies during this state
                Cx = load_vec<T, K>(Cv + (n + 1) * p.C_dstate_stride + t0, nx);
                do_state(n, Bn_, Cn_);
            }
        }
        VIVIM_STAMP(step, 3, wave, lane);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) s_part[wave * TSZ + (r * K + k) * kLdsRow + lane] = yp[r][k];
        VIVIM_STAMP(step, 4, wave, lane);
        lds_barrier();
        VIVIM_STAMP(step, 5, wave, lane);
        // ---- epilogue of this step + prologue of the next, same token slice per thread ----
        {
            float y[IPT];
#pragma unroll
            for (int i = 0; i < IPT; ++i) {
                float acc = s_du[pbase + i * kLdsRow];
#pragma unroll
                for (int wv = 0; wv < NS; ++wv) acc += s_part[wv * TSZ + pbase + i * kLdsRow];   // fixed order
                y[i] = acc;
            }
            if (pr < nvalid) {
                store_vec<T, IPT>(pout + tp, tp < L, y);
                if (HAS_Z) {
                    float zf[IPT];
                    unpack(zr, zf);
#pragma unroll
                    for (int i = 0; i < IPT; ++i) y[i] *= zf[i] * sigmoidf_fast(zf[i]);       // fwd_kernel.cuh:290
                    store_vec<T, IPT>(poutz + tp, tp < L, y);
                }
            }
        }
        VIVIM_STAMP(step, 6, wave, lane);
        if (more) prologue(step + 1, unx, dnx);
        VIVIM_STAMP(step, 7, wave, lane);
        lds_barrier();
        VIVIM_STAMP(step, 8, wave, lane);
    }
}

template <typename T, int K, int NS, int MINW>
static void launch_fwd_nsplit(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    const int cpg = p.dim / p.n_groups;
    dim3 grid(((cpg + kNsR - 1) / kNsR) * p.n_groups, p.batch);
    if (p.z) hipLaunchKernelGGL((ssm_fwd_nsplit_kernel<T, K, NS, true, MINW>), grid, dim3(NS * kWave), 0, stream, p);
    else     hipLaunchKernelGGL((ssm_fwd_nsplit_kernel<T, K, NS, false, MINW>), grid, dim3(NS * kWave), 0, stream, p);
}

// 8 waves per workgroup, dstate / 8 states per wave.  VIVIM_FWD_VARIANT (tuning only) picks the tiling.
template <typename T>
static bool try_fwd_nsplit(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    if (!p.is_variable_B || !p.is_variable_C || p.dstate % 8 != 0) return false;
    // the fast kernel uses unconditional 16-byte vectors: every row must be 16-byte aligned and the
    // sequence a whole number of 8-token lanes; anything else takes the generic kernel.
    const int64_t epv = 16 / (int64_t)sizeof(T);
    auto al = [&](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto st = [&](int64_t e) { return e % epv == 0; };
    if (p.seqlen % 8 != 0 || !al(p.u) || !al(p.delta) || !al(p.B) || !al(p.C) || !al(p.out) ||
        !st(p.u_batch_stride) || !st(p.u_d_stride) || !st(p.delta_batch_stride) || !st(p.delta_d_stride) ||
        !st(p.out_batch_stride) || !st(p.out_d_stride) || !st(p.B_batch_stride) || !st(p.B_group_stride) ||
        !st(p.B_dstate_stride) || !st(p.C_batch_stride) || !st(p.C_group_stride) || !st(p.C_dstate_stride))
        return false;
    if (p.z && (!al(p.z) || !al(p.out_z) || !st(p.z_batch_stride) || !st(p.z_d_stride) ||
                !st(p.out_z_batch_stride) || !st(p.out_z_d_stride)))
        return false;
    const int forced = tuning_fwd_variant();
    // 512-token steps (K=8) halve the per-step fixed cost; 256-token steps (K=4) need 100 instead of 160 VGPRs,
    // so several workgroups share a CU -- better once there are enough workgroups to fill the chip twice AND the
    // rows are short (few steps per row: finer steps waste less of the last one).
    const int64_t nwg = (int64_t)((p.dim / p.n_groups + kNsR - 1) / kNsR) * p.n_groups * p.batch;
    // Measured at the grouped v3 shapes (dim = 3 * d_inner, tools/kbench.py --groups 3): K=8 wins for long rows (L 20480:
    // 365 vs 503 us, L 5120: 160 vs 188 us), K=4 for short ones (L 1280: 112 vs 130 us, L 320: 68 vs 87 us).
    int variant = (forced && forced < 5) ? forced : ((nwg >= 512 && p.seqlen < 4096) ? 2 : 1);
    switch (variant) {
        case 2:  launch_fwd_nsplit<T, 4, 8, 2>(p, stream); break;     // K=4
        case 3:  return false;                                         // generic kernel (tuning only)
        default: launch_fwd_nsplit<T, 8, 8, 2>(p, stream); break;     // K=8
    }
    return true;
}

template <typename T, int K, int R>
static void launch_fwd(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    const int cpg = p.dim / p.n_groups;
    const int sets = ((cpg + R - 1) / R) * p.n_groups;
    dim3 grid((sets + kScanWaves - 1) / kScanWaves, p.batch);
    const size_t smem = (size_t)kScanWaves * R * p.dstate * sizeof(float);
    const bool var = p.is_variable_B;   // capi enforces is_variable_B == is_variable_C
    if (p.z) {
        if (var) hipLaunchKernelGGL((ssm_fwd_generic_kernel<T, K, R, true, true>), grid, dim3(kScanWaves * kWave), smem, stream, p);
        else     hipLaunchKernelGGL((ssm_fwd_generic_kernel<T, K, R, true, false>), grid, dim3(kScanWaves * kWave), smem, stream, p);
    } else {
        if (var) hipLaunchKernelGGL((ssm_fwd_generic_kernel<T, K, R, false, true>), grid, dim3(kScanWaves * kWave), smem, stream, p);
        else     hipLaunchKernelGGL((ssm_fwd_generic_kernel<T, K, R, false, false>), grid, dim3(kScanWaves * kWave), smem, stream, p);
    }
}

// Tokens per checkpoint row of x.  The lanes = states backward (scan_ls.hip) rebuilds the forward states of a 16-token tile
// from a checkpoint, so every shape it takes gets one row per 16 * (dstate / 16) tokens, written by the lanes = channels
// and lanes = states forward kernels; the n-split / generic kernels write one row per kChunk tokens (two per 512-token
// n-split step) and are only reached for other shapes or when the tuning selector pins them.  A pure function of the
// shape and of the forward tuning value: forward and backward calls must see the same one.
bool ls_shape_ok(const vivim_ssm_fwd_params&);                           // scan_ls.hip
int ls_ckpt_len(const vivim_ssm_fwd_params&);
bool try_ls_fwd(const vivim_ssm_fwd_params&, hipStream_t);
size_t ls_fwd_workspace_bytes(const vivim_ssm_fwd_params&);
bool try_fwd_chan(const vivim_ssm_fwd_params& p, hipStream_t stream);   // scan_fwd_chan.hip
size_t fwd_chan_workspace_bytes(const vivim_ssm_fwd_params&);

// Which checkpoint rows a shape gets -- and with them which backward family (the forward families that can write them
// follow).  Measured on MI355X (tools/kb_round2.sh, profiles/r02_kbench_families.log), lanes = states against the round-1
// families at dstate 16: the backward wins on short rows (cfg 2 grouped stages 1-3: 273 / 172 / 84 us against 288 / 185 /
// 149) and loses a few per cent on long ones (L 20480: 592 against 564 us; L 81920: 2214 against 2111), where the
// lanes = tokens kernel amortises its scans over 512-token steps; at dstate 32 / 64 its extra forward sweep per checkpoint
// block costs more than it gains (cfg 5: 1018 against 820 us).  Forward tuning 5 / 6 pin the short rows for any such shape.
static bool ls_plan(const vivim_ssm_fwd_params& f) {
    if (!ls_shape_ok(f)) return false;
    const int t = tuning_fwd_variant();
    if (t == 5 || t == 6) return true;
    if (t != 0) return false;
    // Round 3: with the second-generation lanes = states backward (scan_ls2.hip) and the closed-form pre-pass the two backward
    // families take the same time on long 16-bit rows (cfg 2 grouped stage 0: 547-561 against 555-588 us) while the lanes =
    // states one moves half the bytes (0.58 against 1.08 GB per launch); the forward pays 20 us there for the denser
    // checkpoints (262 against 242).  fp32 rows keep the old limit: the denser checkpoints cost the forward 15 % (cfg 3
    // grouped stage 0: 2974 against 2586 us) for 4 % of the backward.
    return f.dstate == 16 && f.seqlen <= (f.itype == VIVIM_F32 ? 8192 : 32768);
}
int scan_ckpt_len(const vivim_ssm_fwd_params& f) { return ls_plan(f) ? ls_ckpt_len(f) : kChunk; }
int scan_chunk_len(int) { return kChunk; }
static_assert(kWave * 4 == kChunk, "generic kernel step must equal the checkpoint chunk");

size_t scan_fwd_workspace_bytes(const vivim_ssm_fwd_params& f) {
    // either of two families may run (the lanes = channels one also wants aligned rows, known only at launch): the larger
    const size_t a = fwd_chan_workspace_bytes(f), b = ls_plan(f) ? ls_fwd_workspace_bytes(f) : 0;
    return a > b ? a : b;
}

bool ssm_fwd_dispatch(const vivim_ssm_fwd_params& p, hipStream_t s) {
    const int tune = tuning_fwd_variant();
    if (tune != 6 && try_fwd_chan(p, s)) return true;      // lanes = channels: long, wide problems (or tuning 5); either row length
    if (ls_plan(p)) return try_ls_fwd(p, s);               // lanes = states (short checkpoint rows)
    switch (p.itype) {
        case VIVIM_F32: if (!try_fwd_nsplit<float>(p, s)) launch_fwd<float, 4, 2>(p, s); return true;
        case VIVIM_F16: if (!try_fwd_nsplit<f16_t>(p, s)) launch_fwd<f16_t, 4, 2>(p, s); return true;
        case VIVIM_BF16: if (!try_fwd_nsplit<bf16_t>(p, s)) launch_fwd<bf16_t, 4, 2>(p, s); return true;
    }
    return false;
}

}  // namespace vivim
