// ls2_lab.hip -- DIAGNOSTIC harness (not part of the product): runs the lanes=states backward (scan_ls.hip pre-pass + carry,
// scan_ls2.hip main kernel) standalone with in-kernel s_memtime stamps and prints where a (tile, channel) step spends its cycles.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -DVIVIM_STAMPS tools/ls2_lab.hip -o tools/ls2_lab
// Run on the GPU box: tools/ls2_lab [B D L G]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../vivim_amd/csrc/scan_ls.hip"
#include "../vivim_amd/csrc/scan_ls2.hip"

namespace vivim { bool fast_bwd_prepass(const vivim_ssm_bwd_params&, int, int, float*, float*, float*, hipStream_t) { return false; }   // (scan_bwd.hip is not part of the lab: the recurrence pre-pass of scan_ls.hip runs)
                  int tuning_fwd_variant() { return 6; }
                  int tuning_bwd_variant() { const char* e = getenv("VIVIM_BWD_VARIANT"); return e ? atoi(e) : 5; } }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 3, D = argc > 2 ? atoi(argv[2]) : 768, L = argc > 3 ? atoi(argv[3]) : 5120,
              G = argc > 4 ? atoi(argv[4]) : 3, N = 16;
    const size_t nact = (size_t)B * D * L, nbc = (size_t)B * G * N * L;
    std::vector<unsigned short> h(nact > nbc ? nact : nbc);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0xff);   // bf16 values in [0.0078, 0.0156)
    unsigned short *u, *dl, *z, *out, *outz, *Bm, *Cm, *dout, *du, *ddl, *dz;
    float *A, *Dv, *bias, *x, *dA, *dB, *dC, *dD, *dbias;
    for (unsigned short** q : {&u, &dl, &z, &out, &outz, &dout, &du, &ddl, &dz}) CK(hipMalloc(q, nact * 2));
    CK(hipMalloc(&Bm, nbc * 2)); CK(hipMalloc(&Cm, nbc * 2));
    CK(hipMalloc(&A, D * N * 4)); CK(hipMalloc(&Dv, D * 4)); CK(hipMalloc(&bias, D * 4));
    const int nck = (L + 15) / 16;
    CK(hipMalloc(&x, (size_t)B * D * nck * N * 4));
    for (unsigned short* q : {u, dl, z, dout}) CK(hipMemcpy(q, h.data(), nact * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(Bm, h.data(), nbc * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Cm, h.data(), nbc * 2, hipMemcpyHostToDevice));
    std::vector<float> hA(D * N), hD(D, 1.f), hb(D, -4.f);
    for (int d = 0; d < D; ++d) for (int n = 0; n < N; ++n) hA[d * N + n] = -(float)(n + 1);
    CK(hipMemcpy(A, hA.data(), D * N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Dv, hD.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), D * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dA, D * N * 4)); CK(hipMalloc(&dB, nbc * 4)); CK(hipMalloc(&dC, nbc * 4)); CK(hipMalloc(&dD, D * 4)); CK(hipMalloc(&dbias, D * 4));

    vivim_ssm_fwd_params p = {};
    p.batch = B; p.dim = D; p.seqlen = L; p.dstate = N; p.n_groups = G; p.itype = VIVIM_BF16;
    p.is_variable_B = p.is_variable_C = 1; p.delta_softplus = 1;
    p.u_batch_stride = p.delta_batch_stride = p.z_batch_stride = p.out_batch_stride = p.out_z_batch_stride = (int64_t)D * L;
    p.u_d_stride = p.delta_d_stride = p.z_d_stride = p.out_d_stride = p.out_z_d_stride = L;
    p.A_d_stride = N; p.A_dstate_stride = 1;
    p.B_batch_stride = p.C_batch_stride = (int64_t)G * N * L; p.B_group_stride = p.C_group_stride = (int64_t)N * L;
    p.B_dstate_stride = p.C_dstate_stride = L;
    p.u = u; p.delta = dl; p.A = A; p.B = Bm; p.C = Cm; p.D = Dv; p.delta_bias = bias; p.z = z; p.out = out; p.out_z = outz; p.x = x;
    p.workspace_bytes = (int64_t)vivim::ls_fwd_workspace_bytes(p); if (p.workspace_bytes) CK(hipMalloc(&p.workspace, p.workspace_bytes));
    if (!vivim::try_ls_fwd(p, 0)) { printf("forward refused\n"); return 1; }
    CK(hipDeviceSynchronize());

    vivim_ssm_bwd_params q = {};
    q.f = p; q.f.out_z = nullptr;
    q.dout_batch_stride = q.du_batch_stride = q.ddelta_batch_stride = q.dz_batch_stride = (int64_t)D * L;
    q.dout_d_stride = q.du_d_stride = q.ddelta_d_stride = q.dz_d_stride = L;
    q.dA_d_stride = N; q.dA_dstate_stride = 1;
    q.dB_batch_stride = q.dC_batch_stride = (int64_t)G * N * L; q.dB_group_stride = q.dC_group_stride = (int64_t)N * L;
    q.dB_dstate_stride = q.dC_dstate_stride = L;
    q.dout = dout; q.du = du; q.ddelta = ddl; q.dz = dz; q.dA = dA; q.dB = dB; q.dC = dC; q.dD = dD; q.ddelta_bias = dbias;
    q.workspace_bytes = (int64_t)vivim::ls_bwd_workspace_bytes(q.f);
    if (q.workspace_bytes) CK(hipMalloc(&q.workspace, q.workspace_bytes));

    const int nstamp = 2 * vivim::kStampWaves * vivim::kStampSteps * vivim::kStampSlots;
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, nstamp * 8)); CK(hipMemset(dbg, 0, nstamp * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(vivim::g_stamp_buf), &dbg, sizeof(dbg)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) if (!vivim::try_ls_bwd(q, 0)) { printf("backward refused\n"); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemset(dbg, 0, nstamp * 8));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) vivim::try_ls_bwd(q, 0);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("BWD B=%d D=%d L=%d G=%d: %.1f us per launch (stamped build; read the SHARES)\n", B, D, L, G, ms * 1e3 / 3);
    std::vector<unsigned long long> hs(nstamp);
    CK(hipMemcpy(hs.data(), dbg, nstamp * 8, hipMemcpyDeviceToHost));
    // stamps: 0 step top (checkpoint taken, next one requested), 1 token scalars in LDS, 2 forward sweep done, 3 reverse sweep done,
    // 4 outputs written (steps of channels 0-2 end here); last channel of a tile: 5 tile sums start, 6 rows summed, 7 past barrier 1,
    // 8 past barrier 2, 9 sums stored
    for (int blk = 0; blk < 2; ++blk)
        for (int w = 0; w < 4; w += 3) {
            printf("block %d wave %d:\n", blk, w);
            for (int st = 0; st < vivim::kStampSteps; ++st) {
                const unsigned long long* s = &hs[((blk * vivim::kStampWaves + w) * vivim::kStampSteps + st) * vivim::kStampSlots];
                if (!s[0]) continue;
                printf("  step %d: scalars=%lld fwd=%lld rev=%lld post=%lld", st, (long long)(s[1] - s[0]), (long long)(s[2] - s[1]),
                       (long long)(s[3] - s[2]), (long long)(s[4] - s[3]));
                if (s[5]) printf("  | tile: gap=%lld rowsum=%lld barrier1=%lld write+barrier2=%lld reduce=%lld", (long long)(s[5] - s[4]),
                                 (long long)(s[6] - s[5]), (long long)(s[7] - s[6]), (long long)(s[8] - s[7]), (long long)(s[9] - s[8]));
                const unsigned long long* nx = s + vivim::kStampSlots;
                if (st + 1 < vivim::kStampSteps && nx[0]) printf("  -> next step top after %lld", (long long)(nx[0] - (s[5] ? s[9] : s[4])));
                printf("\n");
            }
        }
    return 0;
}
