// scan_fwd_chan.hip -- selective SSM scan forward, "lanes = channels" kernels (gfx950, wave64).
//
// Same math as scan_fwd.hip (selective_scan_fwd_kernel.cuh:67-303); a different mapping, built because the
// n-split kernel turned out instruction- and phase-bound:
//   * a WAVE owns 64 channels of one (batch, B/C group) and a SEGMENT of the token axis; lane = channel.  Each
//     lane walks its tokens serially with all N=16 states h[n] in registers: the recurrence needs no cross-lane
//     operation, no barrier and no second "apply" sweep -- 4 VALU ops + 1 exp per state update, 16 independent
//     dependency chains per lane;
//   * B_n[t] and C_n[t] are the same for the 64 channels of the wave, so they are read with SCALAR loads and
//     enter the fma as SGPR operands -- no LDS traffic, no VGPRs.  A small pre-kernel repacks B and C into fp32
//     token-major rows BC[t] = {B_0..B_15, C_0..C_15} (128 bytes per token, in the workspace): one
//     s_load_dwordx8 per operand per half token, no per-element unpacking on the scalar ALU.  Two scalar sets
//     (states 0-7, states 8-15; 32 SGPRs in all) ping-pong half a token ahead.  Scalar loads return out of order,
//     so every wait on them is lgkmcnt(0): each set is re-loaded right AFTER the explicit wait that precedes the
//     other set's use, never right before one (that would expose the full scalar-load latency);
//   * u / delta / z / out / out_z rows are token-contiguous while lanes are channels: tiles of 64 channels x TT
//     tokens go through wave-private LDS (coalesced 16-byte global accesses on one side, conflict-free
//     ds_read_b128 / ds_write_b128 of a lane's own row on the other; rows padded by 16 bytes);
//   * the token axis is split so that >= ~2048 waves exist.  PASS 1 gives every segment's end state for zero
//     inflow plus sum(delta); a carry kernel chains them (h_in[s+1] = exp2(A*log2e*sum_s) * h_in[s] + H_s); PASS 2
//     recomputes the recurrence from the true inflow and produces out / out_z and the checkpoints x.  The state
//     is re-derived instead of stored: 2 exp per state update instead of a 16-float-per-token round trip.
#include <stdlib.h>
#include "common.cuh"

namespace vivim {

// CH_ABL: timing experiments (tools/abl.sh chan): 1 no scalar B / C loads, 2 no tile loads, 3 no tile stores, 4 no exp -- results
// are wrong for any value but 0; never set in the product build.
#ifndef CH_ABL
#define CH_ABL 0
#endif
constexpr int kChAbl = CH_ABL;
constexpr int kChN = 16;           // states (compile time: they live in registers)
constexpr int kChWaves = 2;        // independent waves per workgroup
// Tokens per tile: a tile row is ONE 128-byte line for every I/O type (32 fp32 / 64 16-bit tokens).  With 32-byte pieces (16
// tokens of bf16, the first version) a line of a row was fetched for four separate tiles, microseconds apart, with 2048
// waves x 64 rows x 3 streams of such lines in flight -- far more than the L2s hold: rocprofv3 counted 1.17 GB of HBM
// traffic per launch against 0.25 GB algorithmic at cfg 2's grouped stage 0 (4.1 TB/s in 283 us) and 4.6 GB against 1.76 GB
// at cfg 3's stage 0 (4.8 TB/s in 962 us): the launches were bound by their own over-fetch
// (profiles/r02_hbm_counters_per_kernel.txt).  Whole lines need 18 KB of LDS per wave for TWO resident streams, which is
// what 8 waves per CU can have: z no longer goes through LDS -- the gate out * silu(z) is applied in the store phase, where
// the lanes lie along tokens again and z is read with the same coalesced vectors the outputs are written with; y waits for it
// in LDS as fp32, in the bytes of the u / delta tokens it was computed from.
template <typename T> struct ChTile { static constexpr int TT = 128 / (int)sizeof(T); };

int scan_ckpt_len(const vivim_ssm_fwd_params&);                          // scan_fwd.hip

struct FwdSeg {
    int S, seg_tiles;              // segments, TT-token tiles per segment
    float* H;                      // [batch][dim][S][N]  PASS 1: end state for zero inflow; after the carry kernel: inflow
    float* dsum;                   // [batch][dim][S]     sum of softplus(delta + bias) over the segment
    const float* BC;               // [batch][groups][Lpad + 1][32]  fp32 token-major B / C (ssm_fwd_bc_kernel)
    int Lpad;
    int ck;                        // tokens per checkpoint row of x (scan_ckpt_len: 16 for the lanes = states backward, else kChunk)
    int xcd;                       // re-number the workgroups so that those sharing B / C rows share an XCD (see the kernel)
};


// BC[b][g][t][chunk] = [B_0..7 | C_0..7 | B_8..15 | C_8..15] of the chunk's 16 states (chunk = state / 16; one chunk at
// dstate 16, four at dstate 64) as fp32; zero rows for L <= t <= Lpad (Lpad + 1 rows).  grid.y = groups * chunks.
template <typename T>
__global__ void __launch_bounds__(256) ssm_fwd_bc_kernel(const vivim_ssm_fwd_params p, float* __restrict__ BC, int Lpad) {
    __shared__ float tile[64][33];
    const int tid = threadIdx.x;
    const int nch = p.dstate / 16;
    const int t0 = blockIdx.x * 64, g = blockIdx.y / nch, kch = blockIdx.y - g * nch, b = blockIdx.z;
    const T* __restrict__ Bp = static_cast<const T*>(p.B) + b * p.B_batch_stride + g * p.B_group_stride + kch * 16 * p.B_dstate_stride;
    const T* __restrict__ Cp = static_cast<const T*>(p.C) + b * p.C_batch_stride + g * p.C_group_stride + kch * 16 * p.C_dstate_stride;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = i * 4 + (tid >> 6), t = t0 + (tid & 63);
        float v = 0.0f;
        if (t < p.seqlen)
            v = to_f32(row < 16 ? Bp[row * p.B_dstate_stride + t] : Cp[(row - 16) * p.C_dstate_stride + t]);
        tile[tid & 63][((row & 8) << 1) | ((row >> 4) << 3) | (row & 7)] = v;   // [B0-7 | C0-7 | B8-15 | C8-15]
    }
    __syncthreads();
    float* __restrict__ dst = BC + (((int64_t)(b * p.n_groups + g) * (Lpad + 1) + t0) * nch + kch) * 32;   // + 1: prefetch overrun row
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + tid;
        if (t0 + (idx >> 5) <= Lpad) dst[(int64_t)(idx >> 5) * 32 * nch + (idx & 31)] = tile[idx >> 5][idx & 31];
    }
}

// ---- the per-token state update --------------------------------------------------------------------------------
// 16 states of one token: a = exp2(dl*A2); h = a*h + (w*B); y += h*C  with B, C as SGPR operands.
// Two FIXED scalar sets ping-pong, each one whole BC row [B0-7 | C0-7 | B8-15 | C8-15]: X = s[68:99], Y = s[36:67].
// A token first issues BOTH loads of the row the NEXT token consumes (into the other set), computes its 16 states from
// its own set and ends with s_waitcnt lgkmcnt(0).  Scalar loads return out of order, so every wait on them is
// lgkmcnt(0): one wait per token, a whole token of work between issue and wait.  (The first version waited per HALF
// token: the SQ counters showed the waves parked 42 % of the time.)  The kernel is compiled with
// amdgpu_num_sgpr(kChSgprLimit): the compiler never allocates s[36:99] itself, so the sets survive the
// compiler-generated code around the asm statements (register allocation kept spilling the scalar sets to VGPR lanes
// inside the hot loop when B and C were plain C++ values).
#define CH_CLOB                                                                                                  \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",  \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",  \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",  \
    "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

constexpr int kChSgprLimit = 40;   // advisory: the compiler peaks at s39 (tile store phase) whatever is asked below ~48

// X = s[68:99]: B0-7 s68-75 | C0-7 s76-83 | B8-15 s84-91 | C8-15 s92-99;  Y = s[36:67] likewise.
// X is the set that is live at block and tile boundaries (the row of the next token to process), where the compiler's
// own code peaks at s39: it sits high.  Y is only live between the two tokens of a pair inside a block, where the
// compiler emits no scalar code above s35 -- `make check-chan-sgpr` verifies both statements on the generated ISA.
// PASS 1 needs B only: two 8-dword loads into the B positions of the set.
#define CH_LOAD_Y2 "s_load_dwordx16 s[36:51], %[ptr], 0x0\n\ts_load_dwordx16 s[52:67], %[ptr], 0x40\n\t"
#define CH_LOAD_X2 "s_load_dwordx16 s[68:83], %[ptr], 0x0\n\ts_load_dwordx16 s[84:99], %[ptr], 0x40\n\t"
#define CH_LOAD_Y1 "s_load_dwordx8 s[36:43], %[ptr], 0x0\n\ts_load_dwordx8 s[52:59], %[ptr], 0x40\n\t"
#define CH_LOAD_X1 "s_load_dwordx8 s[68:75], %[ptr], 0x0\n\ts_load_dwordx8 s[84:91], %[ptr], 0x40\n\t"

// ---- state pairs on v_pk_*_f32 ----------------------------------------------------------------------------------
// The arithmetic of a state pair (2n, 2n+1) is  t = dl * A2 (pk_mul), two v_exp_f32, u = B * w (pk_mul, B an SGPR pair),
// h = t * h + u (pk_fma), y += h * C (pk_fma, C an SGPR pair): 6 instructions for two state updates (the first version
// spent 10 scalar ones inside two 40-instruction asm blocks per token; same roundings in the same order -- y.x collects
// the even states, y.y the odd ones -- so the results did not change by a bit).  Only the two instructions that name the
// fixed scalar registers are inline asm (an asm operand cannot name one half of a 64-bit register pair, which the
// v_exp_f32 pair would need); the rest is C++ on 2-vectors, which hipcc maps to v_pk_mul/fma_f32 with op_sel broadcasts
// and interleaves across the eight pairs.  All asm statements are volatile, so they keep their order among themselves:
// a set is never read before the wait that completes it nor after the load that overwrites it.
// Measured: 40 % fewer VALU instructions per token bought only 2-5 % (cfg 3 stage 0: 1005 -> 954 us, grouped 2745 ->
// 2610 us): on this chip a v_pk_fma_f32 occupies the issue port as long as the two v_fma_f32 it replaces.
typedef float cf2 __attribute__((ext_vector_type(2)));
#define CHP_PAIR(j, BP, CP)                                                                        \
    {                                                                                              \
        cf2 t = dlp * ap[j];                                                                       \
        if (kChAbl != 4) { t.x = fast_exp2(t.x); t.y = fast_exp2(t.y); }                           \
        cf2 u;                                                                                     \
        asm volatile("v_pk_mul_f32 %0, " BP ", %1" : "=v"(u) : "v"(wp));                           \
        hp[j] = __builtin_elementwise_fma(t, hp[j], u);                                            \
        if (PASS == 2) asm volatile("v_pk_fma_f32 %0, %1, " CP ", %0" : "+v"(y2) : "v"(hp[j]));    \
    }
#define CH_CLOB_Y                                                                                                \
    "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",  \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67"
#define CH_CLOB_X                                                                                                \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",  \
    "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99"

// one token whose row is in X (s[68:99]); issues the loads of `next` (the following token's row) into Y first and waits
// for them after the token's last instruction.  hp / ap: the 8 state pairs and their A * log2e.
template <int PASS>
__device__ __forceinline__ void chan_token_x(cf2* hp, const cf2* ap, float dl, float w, cf2& y2, const float* next) {
    const cf2 dlp = {dl, dl}, wp = {w, w};
    if (kChAbl == 1) asm volatile("; CHAN lo_x\n\t" : : [ptr] "s"(next) : CH_CLOB_Y);
    else if (PASS == 2) asm volatile("; CHAN lo_x\n\t" CH_LOAD_Y2 : : [ptr] "s"(next) : CH_CLOB_Y);
    else           asm volatile("; CHAN lo_x\n\t" CH_LOAD_Y1 : : [ptr] "s"(next) : CH_CLOB_Y);
    CHP_PAIR(0, "s[68:69]", "s[76:77]") CHP_PAIR(1, "s[70:71]", "s[78:79]")
    CHP_PAIR(2, "s[72:73]", "s[80:81]") CHP_PAIR(3, "s[74:75]", "s[82:83]")
    CHP_PAIR(4, "s[84:85]", "s[92:93]") CHP_PAIR(5, "s[86:87]", "s[94:95]")
    CHP_PAIR(6, "s[88:89]", "s[96:97]") CHP_PAIR(7, "s[90:91]", "s[98:99]")
    asm volatile("; CHAN hi_x\n\ts_waitcnt lgkmcnt(0)");
}
// the same for a token whose row is in Y (s[36:67]); loads the following row into X
template <int PASS>
__device__ __forceinline__ void chan_token_y(cf2* hp, const cf2* ap, float dl, float w, cf2& y2, const float* next) {
    const cf2 dlp = {dl, dl}, wp = {w, w};
    if (kChAbl == 1) asm volatile("; CHAN lo_y\n\t" : : [ptr] "s"(next) : CH_CLOB_X);
    else if (PASS == 2) asm volatile("; CHAN lo_y\n\t" CH_LOAD_X2 : : [ptr] "s"(next) : CH_CLOB_X);
    else           asm volatile("; CHAN lo_y\n\t" CH_LOAD_X1 : : [ptr] "s"(next) : CH_CLOB_X);
    CHP_PAIR(0, "s[36:37]", "s[44:45]") CHP_PAIR(1, "s[38:39]", "s[46:47]")
    CHP_PAIR(2, "s[40:41]", "s[48:49]") CHP_PAIR(3, "s[42:43]", "s[50:51]")
    CHP_PAIR(4, "s[52:53]", "s[60:61]") CHP_PAIR(5, "s[54:55]", "s[62:63]")
    CHP_PAIR(6, "s[56:57]", "s[64:65]") CHP_PAIR(7, "s[58:59]", "s[66:67]")
    asm volatile("; CHAN hi_y\n\ts_waitcnt lgkmcnt(0)");
}

// the first row of a segment into X
template <int PASS>
__device__ __forceinline__ void chan_prime_x(const float* next) {
    if (PASS == 2) asm volatile(CH_LOAD_X2 "s_waitcnt lgkmcnt(0)" : : [ptr] "s"(next) : CH_CLOB);
    else           asm volatile(CH_LOAD_X1 "s_waitcnt lgkmcnt(0)" : : [ptr] "s"(next) : CH_CLOB);
}

typedef const __attribute__((address_space(4))) vivim_ssm_fwd_params* kparams_t;

// The kernel arguments, re-read through a pointer the compiler cannot see through.  The tile I/O phases use this
// instead of `p`: their ~40 SGPRs of pointers and strides would otherwise stay live across the compute loop and
// push the B/C scalars out of the SGPR file (spills to VGPR lanes inside the hot loop).  `p` is argument 0.
__device__ __forceinline__ kparams_t fresh_params() {
    kparams_t q = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q));
    return q;
}

// (amdgpu_waves_per_eu(2): the f32 / z / 64-state PASS 2 otherwise takes 255 VGPRs + 4 AGPRs = one wave per SIMD; with the
// bound it fits 253 and the other 22 instantiations compile as before, the 16-state ones register for register)
template <typename T, int PASS, bool HAS_Z, int NST = kChN>
__global__ void __launch_bounds__(kChWaves * kWave) __attribute__((amdgpu_num_sgpr(kChSgprLimit), amdgpu_waves_per_eu(2)))
ssm_fwd_chan_kernel(const vivim_ssm_fwd_params p, const FwdSeg sg) {
    constexpr int N = NST;                            // 16, or 64 as four chunks of 16 per token (the scalar sets hold one chunk)
    constexpr int NCH = N / 16;
    constexpr int EPV = 16 / (int)sizeof(T);          // elements per 16-byte vector
    constexpr int TT = ChTile<T>::TT;                 // tokens per tile
    constexpr int RB = TT * (int)sizeof(T);           // bytes of a row inside a tile (32 for 16-bit, 64 for fp32)
    constexpr int LPR = RB / 16;                      // lanes (16-byte columns) per row
    constexpr int RPI = kWave / LPR;                  // rows per cooperative load/store instruction
    constexpr int NIO = kWave / RPI;                  // instructions per tile and stream
    constexpr int ROWB = RB + 16;                     // padded LDS row, bytes (conflict-free 8/16-byte own-row access)
    constexpr int NARR = 2;                           // resident tiles: u, delta
    constexpr int TB = 4;                             // tokens per compute block (L % TB == 0, sg.ck % TB == 0)
    typedef uint32_t __attribute__((ext_vector_type(4))) v4;
    typedef typename Pack<T, TB * (int)sizeof(T)>::type vblk;
    __shared__ __attribute__((aligned(16))) unsigned char lds[kChWaves * NARR * kWave * ROWB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (blocks i and i + 8 share one; MI355X_MICROARCH.md, workgroup
    // dispatch).  The workgroups of one (batch, group, segment) read the same B / C rows: numbered consecutively they
    // would pull those rows into three or more L2s (a third of this kernel's reads at cfg 2's stage 0); re-numbered so
    // that consecutive work items are 8 blocks apart they meet in one.  Speed only: nothing depends on the placement.
    int bx = blockIdx.x, b = blockIdx.y, seg = blockIdx.z;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
        if (sg.xcd && total % 8 == 0) {
            const unsigned flat = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
            const unsigned w = (flat % 8) * (total / 8) + flat / 8;
            bx = (int)(w % gx); b = (int)((w / gx) % gy); seg = (int)(w / (gx * gy));
        }
    }
    const int L = p.seqlen;
    const int cpg = p.dim / p.n_groups;               // % 64 == 0 (host)
    const int bpg = cpg / kWave;                      // 64-channel blocks per B/C group
    const int cb = bx * kChWaves + wave;
    if (cb >= bpg * p.n_groups) return;               // waves are independent: no barrier below
    const int g = cb / bpg;
    const int c0 = cb * kWave;
    const int d = c0 + lane;

    unsigned char* tile_u = lds + (wave * NARR + 0) * kWave * ROWB;
    unsigned char* tile_d = lds + (wave * NARR + 1) * kWave * ROWB;

    cf2 A2p[N / 2], hp[N / 2];                        // state pairs (2j, 2j + 1)
    {
        const float* __restrict__ A = static_cast<const float*>(p.A);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const float a2 = A[d * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;     // fwd_kernel.cuh:168-175
            const float h0 = (PASS == 2 && seg > 0) ? sg.H[(((int64_t)b * p.dim + d) * sg.S + seg) * N + n] : 0.0f;
            if (n & 1) { A2p[n / 2].y = a2; hp[n / 2].y = h0; } else { A2p[n / 2].x = a2; hp[n / 2].x = h0; }
        }
    }
    const float Dv = p.D ? static_cast<const float*>(p.D)[d] : 0.0f;
    const float bias = p.delta_bias ? static_cast<const float*>(p.delta_bias)[d] : 0.0f;
    const bool sp_on = p.delta_softplus;
    const float* __restrict__ bc = sg.BC + (int64_t)(b * p.n_groups + g) * (sg.Lpad + 1) * 32 * NCH;
    const int ck = sg.ck;
    const int nck = (L + ck - 1) / ck;
    float* __restrict__ xlane = static_cast<float*>(p.x) + ((int64_t)b * p.dim + d) * nck * N;   // per-lane (VGPRs)

    // cooperative tile I/O: instruction i moves rows i*16 + lane/4, 16-byte column lane%4
    const int io_col = lane % LPR;
    const int io_row0 = lane / LPR;
    const int ntiles = (L + TT - 1) / TT;
    const int tile_lo = seg * sg.seg_tiles, tile_hi = min(ntiles, tile_lo + sg.seg_tiles);
    float dsum = 0.0f;

    float l2_touch = 0.0f;
    // one BC row is [B0-7 | C0-7 | B8-15 | C8-15]: a half token = 16 consecutive floats (t <= Lpad: Lpad + 1 rows)
    // the segment's first tile is touched here (vector load, L2 allocate) so that its scalar loads do not go to HBM
    if (lane < TT) l2_touch = bc[(int64_t)min(tile_lo * TT + lane, sg.Lpad) * 32 * NCH];   // one lane per BC row (its first line)
    asm volatile("s_waitcnt vmcnt(0)" : : "v"(l2_touch));
    chan_prime_x<PASS>(bc + (int64_t)tile_lo * TT * 32 * NCH);

    // Tile I/O is double-buffered through registers: while tile i is processed out of LDS, the global loads of tile
    // i + 1 are in flight into `nu / nd / nz`; they are written to LDS after tile i's results have left it.
    union tile_regs { RawK<T, EPV> r; v4 v; };
    tile_regs nu[NIO], nd[NIO];
    auto issue_tile_loads = [&](int tile) {
        if (kChAbl == 2 && tile > tile_lo) return;
        kparams_t q = fresh_params();
        // Unconditional loads (countable vmcnt: a load under a branch would make the next wait a full drain); columns past
        // the end are clamped onto the row's last vector -- their tokens are never processed (blocks stop at L, the
        // tile after the segment's last is never read)
        // (the request after the segment's last tile re-reads that tile -- an L2 hit -- instead of fetching a tile of the
        // next segment that nobody uses: with three or four tiles per segment that was a quarter of the launch's reads)
        const int t = min(__builtin_amdgcn_readfirstlane(min(tile, tile_hi - 1) * TT) + io_col * EPV, L - EPV);
        const int64_t su = q->u_d_stride, sd = q->delta_d_stride;
        const T* gu = static_cast<const T*>(q->u) + b * q->u_batch_stride + (c0 + io_row0) * su + t;
        const T* gd = static_cast<const T*>(q->delta) + b * q->delta_batch_stride + (c0 + io_row0) * sd + t;
#pragma unroll
        for (int i = 0; i < NIO; ++i) {
            nu[i].v = *reinterpret_cast<const v4*>(gu + i * RPI * su);
            nd[i].v = *reinterpret_cast<const v4*>(gd + i * RPI * sd);
        }
    };
    auto tile_regs_to_lds = [&] {
#pragma unroll
        for (int i = 0; i < NIO; ++i) {
            const int off = (i * RPI + io_row0) * ROWB + io_col * 16;
            *reinterpret_cast<v4*>(tile_u + off) = nu[i].v;
            *reinterpret_cast<v4*>(tile_d + off) = nd[i].v;
        }
    };
    // z of the tile being computed: requested at the tile's start with the vectors of the store phase, consumed there.
    // (Registers are free for it: two resident tile streams already cap the kernel at two waves per SIMD.)
    tile_regs gz[NIO];
    auto issue_z_loads = [&](int tile) {
        if (!(PASS == 2 && HAS_Z) || kChAbl == 2) return;
        kparams_t q = fresh_params();
        const int t = min(__builtin_amdgcn_readfirstlane(tile * TT) + io_col * EPV, L - EPV);
        const int64_t sz = q->z_d_stride;
        const T* gzp = static_cast<const T*>(q->z) + b * q->z_batch_stride + (c0 + io_row0) * sz + t;
#pragma unroll
        for (int i = 0; i < NIO; ++i) gz[i].v = *reinterpret_cast<const v4*>(gzp + i * RPI * sz);
    };
    issue_tile_loads(tile_lo);
    tile_regs_to_lds();
    wave_lds_fence();

#pragma unroll 1
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int t0 = __builtin_amdgcn_readfirstlane(tile * TT);
        // one vector load touches the 64-byte lines of the NEXT tile's BC rows: they are in this XCD's L2 by the time
        // the scalar loads want them (first touch would otherwise come from beyond the L2)
        if (lane < TT) l2_touch = bc[(int64_t)min(t0 + TT + lane, sg.Lpad) * 32 * NCH];
        issue_tile_loads(tile + 1);                       // (past the segment's last tile: that tile again)
        if (NCH == 1) issue_z_loads(tile);                // (64 states: no registers left -- requested in the store phase)
        // ---- the lane's own row: TT tokens in blocks of TB (one 8- or 16-byte LDS access per stream) ----
#pragma unroll 1
        for (int blk = 0; blk < TT / TB; ++blk) {
            const int tb = __builtin_amdgcn_readfirstlane(t0 + blk * TB);
            if (tb >= L) break;                          // blocks never straddle L: nothing to mask below
            float uf[TB], df[TB], yo[TB], dl[TB], w[TB];
            {
                union { RawK<T, TB> r; vblk v; } cu, cd;
                cu.v = *reinterpret_cast<const vblk*>(tile_u + lane * ROWB + blk * TB * (int)sizeof(T));
                cd.v = *reinterpret_cast<const vblk*>(tile_d + lane * ROWB + blk * TB * (int)sizeof(T));
                unpack(cu.r, uf);
                unpack(cd.r, df);
            }
#pragma unroll
            for (int k = 0; k < TB; ++k) {
                const float raw = df[k] + bias;
                dl[k] = sp_on ? softplus_ref(raw) : raw;
                w[k] = dl[k] * uf[k];
                dsum += dl[k];
                yo[k] = Dv * uf[k];
            }
            const float* bct = bc + (int64_t)tb * 32 * NCH;   // uniform: this block's first BC row
            if constexpr (NCH == 1) {
#pragma unroll
                for (int k = 0; k < TB; k += 2) {         // tokens alternate between the two scalar sets
                    cf2 y2 = {yo[k], 0.0f};
                    chan_token_x<PASS>(hp, A2p, dl[k], w[k], y2, bct + (k + 1) * 32);             // prefetch: token k + 1 -> Y
                    if (PASS == 2) yo[k] = y2.x + y2.y;
                    y2 = cf2{yo[k + 1], 0.0f};
                    chan_token_y<PASS>(hp, A2p, dl[k + 1], w[k + 1], y2, bct + (k + 2) * 32);     // token k + 2 -> X
                    if (PASS == 2) yo[k + 1] = y2.x + y2.y;
                }
            } else {
                static_assert(NCH == 1 || NCH == 4, "the chunk sequence X, Y, X, Y is written for four chunks");
#pragma unroll
                for (int k = 0; k < TB; ++k) {            // a token's four 16-state chunks alternate between the sets
                    const float* row = bct + k * 32 * NCH;
                    cf2 y2 = {yo[k], 0.0f};
                    chan_token_x<PASS>(hp, A2p, dl[k], w[k], y2, row + 32);                  // chunk 1 -> Y
                    chan_token_y<PASS>(hp + 8, A2p + 8, dl[k], w[k], y2, row + 64);          // chunk 2 -> X
                    chan_token_x<PASS>(hp + 16, A2p + 16, dl[k], w[k], y2, row + 96);        // chunk 3 -> Y
                    chan_token_y<PASS>(hp + 24, A2p + 24, dl[k], w[k], y2, row + 32 * NCH);  // the next token's chunk 0 -> X
                    if (PASS == 2) yo[k] = y2.x + y2.y;
                }
            }
            if (PASS == 2) {
                // state after every ck tokens and after the last one: always the last token of a block
                const int tl = tb + TB - 1;
                if (((tl + 1) & (ck - 1)) == 0 || tl == L - 1) {
                    float* xr = xlane + (tl / ck) * N;
#pragma unroll
                    for (int n = 0; n < N; ++n) xr[n] = (n & 1) ? hp[n / 2].y : hp[n / 2].x;
                }
                // y (fp32) replaces the inputs it was computed from: exactly as many bytes.  fp32 I/O: the block's four u
                // values.  16-bit I/O: tokens 0, 1 of the block over its four u values, tokens 2, 3 over its four delta values.
                static_assert(TB == 4, "the in-place y layout is written for blocks of four tokens");
                if (sizeof(T) == 4) {
                    *reinterpret_cast<float4*>(tile_u + lane * ROWB + blk * 16) = float4{yo[0], yo[1], yo[2], yo[3]};
                } else {
                    *reinterpret_cast<float2*>(tile_u + lane * ROWB + blk * 8) = float2{yo[0], yo[1]};
                    *reinterpret_cast<float2*>(tile_d + lane * ROWB + blk * 8) = float2{yo[2], yo[3]};
                }
            }
        }
        wave_lds_fence();
        if (PASS == 2 && kChAbl != 3) {                // LDS -> global, coalesced; the gate is applied here
            kparams_t q = fresh_params();
            const int t = t0 + io_col * EPV;
            if (t < L) {
                if (NCH != 1) issue_z_loads(tile);
                const int64_t so = q->out_d_stride;
                T* go = static_cast<T*>(q->out) + b * q->out_batch_stride + (c0 + io_row0) * so + t;
                const int64_t soz = HAS_Z ? q->out_z_d_stride : 0;
                T* goz = HAS_Z ? static_cast<T*>(q->out_z) + b * q->out_z_batch_stride + (c0 + io_row0) * soz + t : nullptr;
#pragma unroll
                for (int i = 0; i < NIO; ++i) {
                    const int off = (i * RPI + io_row0) * ROWB + io_col * 16;
                    float y[EPV];
                    if (sizeof(T) == 4) {
                        const float4 a = *reinterpret_cast<const float4*>(tile_u + off);
                        y[0] = a.x; y[1] = a.y; y[2] = a.z; y[3] = a.w;
                    } else {                           // tokens 8c .. 8c+7: (0, 1, 4, 5) from the u row, (2, 3, 6, 7) from the delta row
                        const float4 a = *reinterpret_cast<const float4*>(tile_u + off);
                        const float4 d4 = *reinterpret_cast<const float4*>(tile_d + off);
                        y[0] = a.x; y[1] = a.y; y[2] = d4.x; y[3] = d4.y;
                        y[4 % EPV] = a.z; y[5 % EPV] = a.w; y[6 % EPV] = d4.z; y[7 % EPV] = d4.w;
                    }
                    union { T e[EPV]; v4 v; } co;
#pragma unroll
                    for (int k = 0; k < EPV; ++k) co.e[k] = from_f32<T>(y[k]);
                    *reinterpret_cast<v4*>(go + i * RPI * so) = co.v;
                    if (HAS_Z) {
                        float zf[EPV];
                        unpack(gz[i].r, zf);
#pragma unroll
                        for (int k = 0; k < EPV; ++k) co.e[k] = from_f32<T>(y[k] * zf[k] * sigmoidf_fast(zf[k]));   // fwd_kernel.cuh:290
                        *reinterpret_cast<v4*>(goz + i * RPI * soz) = co.v;
                    }
                }
            }
            wave_lds_fence();
        }
        asm volatile("" : : "v"(l2_touch));              // the touch load must not be optimised away (consumed only now)
        tile_regs_to_lds();                              // tile + 1: its loads had a whole tile to land
        wave_lds_fence();
    }
    if (PASS == 1) {
        float* Hs = sg.H + (((int64_t)b * p.dim + d) * sg.S + seg) * N;
#pragma unroll
        for (int n = 0; n < N; ++n) Hs[n] = (n & 1) ? hp[n / 2].y : hp[n / 2].x;
        sg.dsum[((int64_t)b * p.dim + d) * sg.S + seg] = dsum;
    }
}

// In place: H[s] (end state of segment s for zero inflow) becomes the state flowing INTO segment s.
// One wave (= one workgroup) per (batch, channel).  The whole chain (S x 16 states + S delta sums) is first pulled
// into LDS with coalesced loads that are all in flight together; the serial part then runs out of LDS:
// lane = (segment j of a group of 4, state n), the 4 affine maps x -> P x + H of a group are composed with two
// shuffle steps, the carry of the previous group comes from lanes 48..63, the next group's LDS reads are issued
// before the current group is processed.  (The first version walked global memory directly: one memory latency per
// segment, 90 us for S = 342 -- more than the two scan passes together.)
__global__ void __launch_bounds__(kWave) ssm_fwd_carry_kernel(const vivim_ssm_fwd_params p, const FwdSeg sg) {
    constexpr int N = kChN;
    extern __shared__ __attribute__((aligned(16))) float cs[];
    const int lane = threadIdx.x;
    const int64_t bd = blockIdx.x;
    const int S = sg.S;
    float* Hs = cs;                                     // [S][N]
    float* ds = cs + S * N;                             // [S]
    float* __restrict__ Hrow = sg.H + bd * S * N;
    const float* __restrict__ drow = sg.dsum + bd * S;
    for (int i = lane * 4; i < S * N; i += kWave * 4)   // S * N % 4 == 0, rows are 64-byte aligned
        *reinterpret_cast<float4*>(Hs + i) = *reinterpret_cast<const float4*>(Hrow + i);
    for (int i = lane; i < S; i += kWave) ds[i] = drow[i];
    __syncthreads();
    const int n = lane & 15, j = lane >> 4;
    const int dch = (int)(bd % p.dim);
    const float A2 = static_cast<const float*>(p.A)[dch * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;
    float carry = 0.0f;                                 // state flowing into the current group (per n, replicated over j)
    float Hn = j < S ? Hs[j * N + n] : 0.0f;
    float dn = j < S ? ds[j] : 0.0f;
    for (int s0 = 0; s0 < S; s0 += 4) {
        const float Hc = Hn, dc = dn;
        const int sn = s0 + 4 + j;
        Hn = sn < S ? Hs[sn * N + n] : 0.0f;            // next group
        dn = sn < S ? ds[sn] : 0.0f;
        // inclusive scan over j of the maps (P, H); segments past S are identity maps (P = 1, H = 0)
        float P = s0 + j < S ? fast_exp2(A2 * dc) : 1.0f, H = Hc;
        float Pp = __shfl_up(P, 16, kWave), Hp = __shfl_up(H, 16, kWave);
        if (j >= 1) { H = fmaf(P, Hp, H); P *= Pp; }
        Pp = __shfl_up(P, 32, kWave); Hp = __shfl_up(H, 32, kWave);
        if (j >= 2) { H = fmaf(P, Hp, H); P *= Pp; }
        // exclusive form: the inflow of segment s0 + j is the inclusive result of j - 1 applied to the carry
        float Pe = __shfl_up(P, 16, kWave), He = __shfl_up(H, 16, kWave);
        if (j == 0) { Pe = 1.0f; He = 0.0f; }
        const float hin = fmaf(Pe, carry, He);
        if (s0 + j < S) Hrow[(s0 + j) * N + n] = hin;
        const float cnext = fmaf(P, carry, H);          // valid in j == 3: state after the whole group
        carry = __shfl(cnext, 48 + n, kWave);
    }
}

// The same chain for 64 states: lane = state, segments in order, eight segments' operands requested together.
__global__ void __launch_bounds__(kWave) ssm_fwd_carry64_kernel(const vivim_ssm_fwd_params p, const FwdSeg sg) {
    constexpr int N = 64;
    const int n = threadIdx.x;
    const int64_t bd = blockIdx.x;
    const int S = sg.S;
    float* __restrict__ Hrow = sg.H + bd * S * N;
    const float* __restrict__ drow = sg.dsum + bd * S;
    const int dch = (int)(bd % p.dim);
    const float A2 = static_cast<const float*>(p.A)[dch * p.A_d_stride + n * p.A_dstate_stride] * kLog2e;
    float carry = 0.0f;
    for (int s0 = 0; s0 < S; s0 += 8) {
        float H[8], ds[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool ok = s0 + q < S;
            H[q] = ok ? Hrow[(s0 + q) * N + n] : 0.0f;
            ds[q] = ok ? drow[s0 + q] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (s0 + q < S) {
                Hrow[(s0 + q) * N + n] = carry;                       // the state flowing INTO segment s0 + q
                carry = fmaf(fast_exp2(A2 * ds[q]), carry, H[q]);
            }
        }
    }
}

static void fwd_chan_segmentation(const vivim_ssm_fwd_params& f, int tt, int& S, int& seg_tiles) {
    const int ntiles = (f.seqlen + tt - 1) / tt;
    const int cpg = f.dim / f.n_groups;
    const int64_t waves = (int64_t)((cpg + kWave - 1) / kWave) * f.n_groups * f.batch;
    // 2048 waves: swept 512 ... 8192 on the grouped cfg-2 shapes (309/144/106/64 us at 2048; 377/198/130/68 at 1024;
    // 320/167/125/70 at 4096); again with the packed token update: 293/146/101/69 at 1536, 283/141/104/70 at 2048,
    // 300/164/115/70 at 3072, 304/165/124/70 at 4096
    static const int target = getenv("VIVIM_CHAN_WAVES") ? atoi(getenv("VIVIM_CHAN_WAVES")) : 2048;   // (sweeps)
    int64_t want = (target + waves - 1) / waves;
    if (want > ntiles) want = ntiles;
    if (want > 512) want = 512;       // the carry kernel keeps a whole chain in LDS: 512 * 17 * 4 = 34 KB
    if (want < 1) want = 1;
    seg_tiles = (int)((ntiles + want - 1) / want);
    S = (ntiles + seg_tiles - 1) / seg_tiles;
}

// shape_only: pointers are not inspected (the workspace query may come before they are final)
static bool fwd_chan_eligible(const vivim_ssm_fwd_params& p, bool shape_only = false) {
    if (!p.is_variable_B || !p.is_variable_C || (p.dstate != 16 && p.dstate != 64) || p.seqlen % 8 != 0) return false;
    if (p.dim % p.n_groups != 0 || (p.dim / p.n_groups) % kWave != 0) return false;   // whole 64-channel blocks per group
    // Automatic choice, from tools/kbench.py on MI355X (us, this family vs n-split; cols = batch * dim / 64 waves' worth of
    // channels, work = cols * seqlen wave-tokens):
    //   grouped v3 stages 0-3 (cols 18/36/90/144, work 368k/184k/115k/46k): 309/140/104/64 vs 369/159/112/68
    //   per-direction stages 0-3 (cols 6/12/30/48, work 123k/61k/38k/15k):  135/84/60/37  vs 131/65/39/29
    //   cfg 3 stage 0 fp32 (cols 16, work 1.3M): 997 vs 1189;  grouped (cols 48, 3.9M): 2883 vs 3453
    // -> enough total work AND enough independent channel blocks; otherwise the two passes + carry are latency-bound
    // and n-split wins.  Tuning 5 forces this family, any other non-zero value excludes it.
    const int tune = tuning_fwd_variant();
    if (tune != 5) {
        if (tune != 0) return false;
        if (p.dstate == 64 && p.itype == VIVIM_F32) return false;   // 255 + 4 registers: one wave per SIMD (n-split is faster)
        const int64_t cols = (int64_t)p.batch * (p.dim / kWave);
        // short checkpoint rows (scan_ckpt_len): the alternative is the lanes = states forward, which wins below ~150 k
        // wave-tokens (cfg 2 grouped stages 1-3, 184 k / 115 k / 46 k: 150 / 98 / 51 us against 137 / 100 / 59 with 64-byte
        // tile rows; cfg 3 stage 2, 410 k: 361 against 302)
        const int64_t least = scan_ckpt_len(p) < kChunk ? 150000 : 110000;
        if (cols < 8 || cols * p.seqlen < least) return false;
    }
    const int64_t epv = p.itype == VIVIM_F32 ? 4 : 8;
    auto al = [&](const void* q) { return shape_only || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    auto st = [&](int64_t e) { return e % epv == 0; };
    if (!al(p.u) || !al(p.delta) || !al(p.out) ||
        !st(p.u_batch_stride) || !st(p.u_d_stride) || !st(p.delta_batch_stride) || !st(p.delta_d_stride) ||
        !st(p.out_batch_stride) || !st(p.out_d_stride))
        return false;
    if (p.z && (!al(p.z) || !al(p.out_z) || !st(p.z_batch_stride) || !st(p.z_d_stride) ||
                !st(p.out_z_batch_stride) || !st(p.out_z_d_stride)))
        return false;
    return true;
}

static size_t fwd_chan_layout(const vivim_ssm_fwd_params& f, int tt, int& S, int& seg_tiles, int& Lpad,
                              size_t& bc_floats) {
    fwd_chan_segmentation(f, tt, S, seg_tiles);
    Lpad = (f.seqlen + tt - 1) / tt * tt;
    bc_floats = (size_t)f.batch * f.n_groups * (Lpad + 1) * 32 * (f.dstate / 16);
    const size_t h_floats = S > 1 ? (size_t)f.batch * f.dim * S * (f.dstate + 1) : 0;
    return (bc_floats + h_floats) * sizeof(float);
}

size_t fwd_chan_workspace_bytes(const vivim_ssm_fwd_params& f) {
    if (!fwd_chan_eligible(f, true)) return 0;
    int S, seg_tiles, Lpad;
    size_t bc;
    return fwd_chan_layout(f, f.itype == VIVIM_F32 ? ChTile<float>::TT : ChTile<bf16_t>::TT, S, seg_tiles, Lpad, bc);
}

template <typename T>
static bool launch_fwd_chan(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    constexpr int TT = ChTile<T>::TT;
    int S, seg_tiles, Lpad;
    size_t bc_floats;
    const size_t need = fwd_chan_layout(p, TT, S, seg_tiles, Lpad, bc_floats);
    if (!p.workspace || (size_t)p.workspace_bytes < need || (reinterpret_cast<uintptr_t>(p.workspace) & 63)) return false;
    // Measured, VIVIM_CHAN_XCD=0 / 1 (profiles/r02_chan_xcd_ab.log): the re-numbering gains 2 - 5 % on the bf16 grouped shapes
    // and where two workgroups share a group's B / C rows (cfg 3 stage 1), and loses 2 - 10 % on fp32 problems with one
    // workgroup per group (cfg 3 stage 0: 831 -> 845 us; grouped 2563 -> 2807 us).
    static const int xcd_env = getenv("VIVIM_CHAN_XCD") ? atoi(getenv("VIVIM_CHAN_XCD")) : -1;
    const int wg_per_group = ((p.dim / p.n_groups) / kWave + kChWaves - 1) / kChWaves;
    const int xcd = xcd_env >= 0 ? xcd_env : ((wg_per_group >= 2 || p.itype != VIVIM_F32) ? 1 : 0);
    FwdSeg sg = {S, seg_tiles, nullptr, nullptr, static_cast<const float*>(p.workspace), Lpad, scan_ckpt_len(p), xcd};
    if (S > 1) {
        sg.H = static_cast<float*>(p.workspace) + bc_floats;
        sg.dsum = sg.H + (size_t)p.batch * p.dim * S * p.dstate;
    }
    hipLaunchKernelGGL((ssm_fwd_bc_kernel<T>), dim3((Lpad + 1 + 63) / 64, p.n_groups * (p.dstate / 16), p.batch), dim3(256), 0, stream, p,
                       static_cast<float*>(p.workspace), Lpad);
    const int cpg = p.dim / p.n_groups;
    const int blocks = ((cpg / kWave) * p.n_groups + kChWaves - 1) / kChWaves;
    const dim3 block(kChWaves * kWave), grid(blocks, p.batch, sg.S);
    if (p.dstate == 64) {
        if (sg.S > 1) {
            hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 1, false, 64>), grid, block, 0, stream, p, sg);
            hipLaunchKernelGGL(ssm_fwd_carry64_kernel, dim3((unsigned)(p.batch * p.dim)), dim3(kWave), 0, stream, p, sg);
        }
        if (p.z) hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 2, true, 64>), grid, block, 0, stream, p, sg);
        else     hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 2, false, 64>), grid, block, 0, stream, p, sg);
        return true;
    }
    if (sg.S > 1) {
        hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 1, false>), grid, block, 0, stream, p, sg);
        const size_t carry_lds = (size_t)sg.S * (kChN + 1) * sizeof(float);      // <= 34 KB (S <= 512)
        hipLaunchKernelGGL(ssm_fwd_carry_kernel, dim3((unsigned)(p.batch * p.dim)), dim3(kWave), carry_lds, stream, p, sg);
    }
    if (p.z) hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 2, true>), grid, block, 0, stream, p, sg);
    else     hipLaunchKernelGGL((ssm_fwd_chan_kernel<T, 2, false>), grid, block, 0, stream, p, sg);
    return true;
}

bool try_fwd_chan(const vivim_ssm_fwd_params& p, hipStream_t stream) {
    if (!fwd_chan_eligible(p)) return false;
    switch (p.itype) {
        case VIVIM_F32: return launch_fwd_chan<float>(p, stream);
        case VIVIM_F16: return launch_fwd_chan<f16_t>(p, stream);
        case VIVIM_BF16: return launch_fwd_chan<bf16_t>(p, stream);
    }
    return false;
}

}  // namespace vivim
