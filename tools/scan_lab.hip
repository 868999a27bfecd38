// scan_lab.hip -- DIAGNOSTIC harness (not part of the product): runs the forward scan kernel of
// vivim_amd/csrc/scan_fwd.hip standalone with in-kernel s_memtime stamps and prints where a step
// spends its cycles.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -DVIVIM_STAMPS
//                          tools/scan_lab.hip -o tools/scan_lab        Run on the GPU box: tools/scan_lab [B D L N variant]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../vivim_amd/csrc/scan_fwd_chan.hip"
#include "../vivim_amd/csrc/scan_fwd.hip"
#include "../vivim_amd/csrc/scan_bwd.hip"

namespace vivim { int tuning_fwd_variant() { const char* e = getenv("VIVIM_FWD_VARIANT"); return e ? atoi(e) : 0; }
                  int tuning_bwd_variant() { return 0; } }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 3, D = argc > 2 ? atoi(argv[2]) : 128, L = argc > 3 ? atoi(argv[3]) : 20480,
              N = argc > 4 ? atoi(argv[4]) : 16;
    using T = vivim::bf16_t;
    const size_t nact = (size_t)B * D * L, nbc = (size_t)B * N * L;
    std::vector<unsigned short> h(nact);
    for (size_t i = 0; i < nact; ++i) h[i] = 0x3c00 + (rand() & 0xff);   // bf16 values in [0.0078, 0.0156): small positive
    unsigned short *u, *dl, *z, *out, *outz, *Bm, *Cm;
    float *A, *Dv, *bias, *x;
    CK(hipMalloc(&u, nact * 2)); CK(hipMalloc(&dl, nact * 2)); CK(hipMalloc(&z, nact * 2));
    CK(hipMalloc(&out, nact * 2)); CK(hipMalloc(&outz, nact * 2)); CK(hipMalloc(&Bm, nbc * 2)); CK(hipMalloc(&Cm, nbc * 2));
    CK(hipMalloc(&A, D * N * 4)); CK(hipMalloc(&Dv, D * 4)); CK(hipMalloc(&bias, D * 4));
    const int nck = (L + 255) / 256;
    CK(hipMalloc(&x, (size_t)B * D * nck * N * 4));
    CK(hipMemcpy(u, h.data(), nact * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, h.data(), nact * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(z, h.data(), nact * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Bm, h.data(), nbc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Cm, h.data(), nbc * 2, hipMemcpyHostToDevice));
    std::vector<float> hA(D * N), hD(D, 1.f), hb(D, -4.f);
    for (int d = 0; d < D; ++d) for (int n = 0; n < N; ++n) hA[d * N + n] = -(float)(n + 1);
    CK(hipMemcpy(A, hA.data(), D * N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Dv, hD.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), D * 4, hipMemcpyHostToDevice));

    vivim_ssm_fwd_params p = {};
    p.batch = B; p.dim = D; p.seqlen = L; p.dstate = N; p.n_groups = 1; p.itype = VIVIM_BF16;
    p.is_variable_B = p.is_variable_C = 1; p.delta_softplus = 1;
    p.u_batch_stride = p.delta_batch_stride = p.z_batch_stride = p.out_batch_stride = p.out_z_batch_stride = (int64_t)D * L;
    p.u_d_stride = p.delta_d_stride = p.z_d_stride = p.out_d_stride = p.out_z_d_stride = L;
    p.A_d_stride = N; p.A_dstate_stride = 1;
    p.B_batch_stride = p.C_batch_stride = (int64_t)N * L; p.B_group_stride = p.C_group_stride = (int64_t)N * L;
    p.B_dstate_stride = p.C_dstate_stride = L;
    p.workspace_bytes = (int64_t)vivim::scan_fwd_workspace_bytes(p); if (p.workspace_bytes) CK(hipMalloc(&p.workspace, p.workspace_bytes));
    p.u = u; p.delta = dl; p.A = A; p.B = Bm; p.C = Cm; p.D = Dv; p.delta_bias = bias; p.z = z; p.out = out; p.out_z = outz; p.x = x;

    const int nstamp = 2 * vivim::kStampWaves * vivim::kStampSteps * vivim::kStampSlots;
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, nstamp * 8)); CK(hipMemset(dbg, 0, nstamp * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(vivim::g_stamp_buf), &dbg, sizeof(dbg)));

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) vivim::ssm_fwd_dispatch(p, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int iters = 10;
    for (int i = 0; i < iters; ++i) vivim::ssm_fwd_dispatch(p, 0);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d D=%d L=%d N=%d: %.1f us per launch (stamped build; read the SHARES, not the length)\n", B, D, L, N, ms * 1e3 / iters);
    std::vector<unsigned long long> hs(nstamp);
    CK(hipMemcpy(hs.data(), dbg, nstamp * 8, hipMemcpyDeviceToHost));
    const char* names[] = {"top->lds-read", "state0", "state1+", "(gap)", "part-write", "barrierB", "epilogue", "prologue", "barrierA"};
    for (int blk = 0; blk < 2; ++blk)
        for (int w = 0; w < vivim::kStampWaves; w += 7) {
            printf("block %d wave %d (cycles per segment; steps 2..7):\n", blk, w);
            for (int st = 2; st < vivim::kStampSteps; ++st) {
                const unsigned long long* s = &hs[((blk * vivim::kStampWaves + w) * vivim::kStampSteps + st) * vivim::kStampSlots];
                printf("  step %d:", st);
                for (int k = 0; k < 8; ++k) printf(" %s=%lld", names[k], (long long)(s[k + 1] - s[k]));
                printf("  total=%lld\n", (long long)(s[8] - s[0]));
            }
        }
    if (argc > 5 && atoi(argv[5]) == 1) {   // ---- backward ----
        unsigned short *dout, *du, *ddl, *dz;
        float *dA, *dB, *dC, *dD, *dbias;
        CK(hipMalloc(&dout, nact * 2)); CK(hipMalloc(&du, nact * 2)); CK(hipMalloc(&ddl, nact * 2)); CK(hipMalloc(&dz, nact * 2));
        CK(hipMemcpy(dout, h.data(), nact * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dA, D * N * 4)); CK(hipMalloc(&dB, nbc * 4)); CK(hipMalloc(&dC, nbc * 4)); CK(hipMalloc(&dD, D * 4)); CK(hipMalloc(&dbias, D * 4));
        CK(hipMemset(dA, 0, D * N * 4)); CK(hipMemset(dB, 0, nbc * 4)); CK(hipMemset(dC, 0, nbc * 4)); CK(hipMemset(dD, 0, D * 4)); CK(hipMemset(dbias, 0, D * 4));
        vivim_ssm_bwd_params q = {};
        q.f = p; q.f.out_z = nullptr;
        q.dout_batch_stride = q.du_batch_stride = q.ddelta_batch_stride = q.dz_batch_stride = (int64_t)D * L;
        q.dout_d_stride = q.du_d_stride = q.ddelta_d_stride = q.dz_d_stride = L;
        q.dA_d_stride = N; q.dA_dstate_stride = 1;
        q.dB_batch_stride = q.dC_batch_stride = (int64_t)N * L; q.dB_group_stride = q.dC_group_stride = (int64_t)N * L;
        q.dB_dstate_stride = q.dC_dstate_stride = L;
        q.dout = dout; q.du = du; q.ddelta = ddl; q.dz = dz; q.dA = dA; q.dB = dB; q.dC = dC; q.dD = dD; q.ddelta_bias = dbias;
        if (argc > 6 && atoi(argv[6]) == 1) {
            q.workspace_bytes = (int64_t)vivim::scan_bwd_workspace_bytes(q.f);
            if (q.workspace_bytes) CK(hipMalloc(&q.workspace, q.workspace_bytes));
            printf("workspace %lld bytes\n", (long long)q.workspace_bytes);
        }
        CK(hipMemset(dbg, 0, nstamp * 8));
        vivim::ssm_bwd_dispatch(q, 0);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) vivim::ssm_bwd_dispatch(q, 0);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("BWD B=%d D=%d L=%d N=%d: %.1f us per launch (stamped)\n", B, D, L, N, ms * 1e3 / 3);
        CK(hipMemcpy(hs.data(), dbg, nstamp * 8, hipMemcpyDeviceToHost));
        // stamps of ssm_bwd_fast_kernel: 0 step start, 1 loads+softplus done; inside state n = 1: 2 unpack/prefetch,
        // 3 record + local forward, 4 forward scan + h, 5 reverse local + scan + join, 6 accumulation loop,
        // 7 dA reduce + record update + slot stores, 8 barrier, 9 slot sum + atomic; 11 all states done, 10 outputs stored
        for (int w = 0; w < vivim::kStampWaves; w += 7) {
            printf("bwd block 0 wave %d:\n", w);
            for (int st = 1; st < 4; ++st) {
                const unsigned long long* s = &hs[((0 * vivim::kStampWaves + w) * vivim::kStampSteps + st) * vivim::kStampSlots];
                printf("  step %d: loads+prep=%lld  all-states=%lld  outputs=%lld  total=%lld\n", st, (long long)(s[1] - s[0]),
                       (long long)(s[11] - s[1]), (long long)(s[10] - s[11]), (long long)(s[10] - s[0]));
                printf("      state 1: rec+local-fwd=%lld fwd-scan+h=%lld rev-local+scan+join=%lld accumulate=%lld "
                       "dA+rec+slots=%lld barrier=%lld slot-sum+atomic=%lld  (sum %lld)\n",
                       (long long)(s[3] - s[2]), (long long)(s[4] - s[3]), (long long)(s[5] - s[4]), (long long)(s[6] - s[5]),
                       (long long)(s[7] - s[6]), (long long)(s[8] - s[7]), (long long)(s[9] - s[8]), (long long)(s[9] - s[2]));
            }
        }
    }
    return 0;
}
