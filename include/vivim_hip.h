/*
 * vivim_hip.h -- C ABI of libvivim_hip.so: the MI355X (gfx950) drop-in for the two CUDA extension
 * modules on Vivim's Temporal-Mamba hot path.
 *
 * Each entry point replaces one function the reference binds with pybind11 (all citations are
 * file:line under /root/reference):
 *
 *   vivim_selective_scan_fwd  <- selective_scan_cuda.fwd   mamba/csrc/selective_scan/selective_scan.cpp:226-336
 *                                params struct SSMParamsBase mamba/csrc/selective_scan/selective_scan.h:26-69
 *   vivim_selective_scan_bwd  <- selective_scan_cuda.bwd   mamba/csrc/selective_scan/selective_scan.cpp:338-492
 *                                params struct SSMParamsBwd  mamba/csrc/selective_scan/selective_scan.h:71-101
 *   vivim_causal_conv1d_fwd   <- causal_conv1d_cuda.causal_conv1d_fwd  causal-conv1d/csrc/causal_conv1d.cpp:130-189
 *                                params struct ConvParamsBase causal-conv1d/csrc/causal_conv1d.h:9-35
 *   vivim_causal_conv1d_bwd   <- causal_conv1d_cuda.causal_conv1d_bwd  causal-conv1d/csrc/causal_conv1d.cpp:191-268
 *                                params struct ConvParamsBwd  causal-conv1d/csrc/causal_conv1d.h:37-52
 *
 * Contract (same as the reference bindings after their ATen part):
 *   - every pointer is a DEVICE pointer on the current device; strides are in ELEMENTS; the token
 *     (seqlen) axis of every activation tensor has unit stride; batch / channel strides are free
 *     (Vivim passes halves of `xz`, strides (L, B*L, 1) -- mamba_simple.py:204-208).
 *   - the call only enqueues kernels on `stream` (a hipStream_t, NULL = default stream): it never
 *     synchronises, allocates or frees, and keeps no per-call state (re-entrant).  The one piece of process-global
 *     mutable state is the kernel-selection override of vivim_set_tuning() (tests / tuning only, default 0 = automatic):
 *     it also changes vivim_scan_ckpt_len(), so it must not be changed between a forward call and its backward call or
 *     while another thread is inside the library.
 *   - outputs are caller-allocated.  Accumulated outputs (dA, dB, dC, dD, ddelta_bias, dweight,
 *     dbias) are float32 and MUST be zero-filled by the caller before the call, exactly as the
 *     reference binding does (selective_scan.cpp:458-466, causal_conv1d.cpp:247-249).
 *   - returns 0 on success; otherwise a VIVIM_ERR_* code, and vivim_last_error() returns a
 *     thread-local message naming the failed check (the reference raises RuntimeError there).
 *
 * Differences from the reference structs, on purpose: strides are int64 (a 288 GB HBM3E part holds
 * tensors past 2^32 elements; the reference uses uint32, selective_scan.h:27); the input dtype is an
 * explicit enum instead of a C++ template dispatch; `x` (scan checkpoints) has OUR chunk length,
 * vivim_scan_ckpt_len(), instead of the reference's fixed 2048 (selective_scan.cpp:307).
 */
#ifndef VIVIM_HIP_H
#define VIVIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIVIM_ABI_VERSION 8

typedef enum { VIVIM_F32 = 0, VIVIM_F16 = 1, VIVIM_BF16 = 2 } vivim_dtype_t;

enum {
    VIVIM_OK = 0,
    VIVIM_ERR_INVALID = 1,      /* failed shape / stride / pointer check */
    VIVIM_ERR_UNSUPPORTED = 2,  /* valid in the reference but not built here (see message) */
    VIVIM_ERR_LAUNCH = 3        /* hipGetLastError() after the launch */
};

/* ---- selective scan, forward (selective_scan.h:26-69) --------------------------------------- */
typedef struct {
    int32_t batch, dim, seqlen, dstate, n_groups;
    int32_t itype;              /* vivim_dtype_t of u, delta, z, out, out_z and of variable B / C */
    int32_t is_variable_B;      /* B is (batch, n_groups, dstate, seqlen) itype; else (dim, dstate) f32 */
    int32_t is_variable_C;
    int32_t delta_softplus;
    int32_t _pad0;
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t z_batch_stride, z_d_stride;
    int64_t out_batch_stride, out_d_stride;
    int64_t out_z_batch_stride, out_z_d_stride;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;   /* constant B: batch/group stride unused, */
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;   /* B_d_stride == B_group_stride slot     */
    const void *u, *delta;      /* (batch, dim, seqlen) itype */
    const void *A;              /* (dim, dstate) f32 */
    const void *B, *C;
    const void *D;              /* (dim) f32 or NULL */
    const void *delta_bias;     /* (dim) f32 or NULL */
    const void *z;              /* (batch, dim, seqlen) itype or NULL */
    void *out;                  /* (batch, dim, seqlen) itype: y + D*u, before gating.  The reference allocates out / out_z with
                                   delta's / z's strides (selective_scan.cpp:311-313); a shape that gets the short checkpoint
                                   rows (vivim_scan_ckpt_len() == 16 * (dstate / 16)) needs every (channel, token) byte offset
                                   of out / out_z inside one batch element below 2^32 - 2^16, as it already holds for delta / z
                                   there: other strides return VIVIM_ERR_UNSUPPORTED */
    void *out_z;                /* out * silu(z); required iff z != NULL */
    void *x;                    /* (batch, dim, n_chunks, dstate) f32 contiguous: state after each chunk of
                                   vivim_scan_ckpt_len() tokens; x[:, :, -1, :] is the final state
                                   (the reference's x[:, :, -1, 1::2], selective_scan_interface.py:40) */
    void *workspace;            /* forward only: device scratch of >= vivim_scan_fwd_workspace_bytes() bytes or NULL
                                   (NULL selects a kernel that needs none); ignored inside vivim_ssm_bwd_params.f */
    int64_t workspace_bytes;
} vivim_ssm_fwd_params;

/* ---- selective scan, backward (selective_scan.h:71-101) ------------------------------------- */
typedef struct {
    vivim_ssm_fwd_params f;     /* forward tensors; f.out = SAVED forward `out` (needed iff z != NULL);
                                   f.out_z = optional recomputed out_z destination (may be NULL);
                                   f.x = forward checkpoints (required when seqlen > chunk_len) */
    int64_t dout_batch_stride, dout_d_stride;
    int64_t du_batch_stride, du_d_stride;
    int64_t ddelta_batch_stride, ddelta_d_stride;
    int64_t dz_batch_stride, dz_d_stride;
    int64_t dA_d_stride, dA_dstate_stride;
    int64_t dB_batch_stride, dB_group_stride, dB_dstate_stride;
    int64_t dC_batch_stride, dC_group_stride, dC_dstate_stride;
    const void *dout;           /* (batch, dim, seqlen) itype */
    void *du, *ddelta;          /* itype */
    void *dz;                   /* itype; required iff z != NULL (may alias a caller view of dxz) */
    void *dA;                   /* (dim, dstate) f32, pre-zeroed */
    void *dB, *dC;              /* f32, pre-zeroed: (batch, n_groups, dstate, seqlen) if variable else (dim, dstate) */
    void *dD;                   /* (dim) f32 pre-zeroed, or NULL iff D == NULL */
    void *ddelta_bias;          /* (dim) f32 pre-zeroed, or NULL iff delta_bias == NULL */
    void *workspace;            /* device scratch of >= vivim_scan_bwd_workspace_bytes() bytes, or NULL: the
                                   kernel then walks the whole sequence inside one workgroup per 16 channels
                                   (correct, but far fewer workgroups in flight).  Contents are don't-care. */
    int64_t workspace_bytes;
} vivim_ssm_bwd_params;

/* ---- causal depthwise conv1d (causal_conv1d.h:9-52) ------------------------------------------ */
typedef struct {
    int32_t batch, dim, seqlen, width;   /* width in [2, 4] (causal_conv1d.cpp:157) */
    int32_t itype;                       /* dtype of x, out, dout, dx */
    int32_t wtype;                       /* dtype of weight and bias */
    int32_t silu_activation;
    int32_t _pad0;
    int64_t x_batch_stride, x_c_stride, x_l_stride;         /* x_l_stride == 1 (channel-first), or x_c_stride == 1 and
                                                               x_l_stride > 1 (channel-last, causal_conv1d.cpp:151):
                                                               out / dout / dx must then have unit channel stride too */
    int64_t out_batch_stride, out_c_stride, out_l_stride;
    int64_t weight_c_stride, weight_width_stride;
    const void *x;              /* (batch, dim, seqlen) */
    const void *weight;         /* (dim, width) */
    const void *bias;           /* (dim) or NULL */
    void *out;                  /* forward output; unused by the backward */
} vivim_conv_fwd_params;

typedef struct {
    vivim_conv_fwd_params f;
    int64_t dout_batch_stride, dout_c_stride, dout_l_stride;
    int64_t dx_batch_stride, dx_c_stride, dx_l_stride;
    int64_t dweight_c_stride, dweight_width_stride;
    const void *dout;
    void *dx;                   /* itype, may be a strided caller view (causal_conv1d.cpp:232-238) */
    void *dweight;              /* (dim, width) f32 pre-zeroed */
    void *dbias;                /* (dim) f32 pre-zeroed, or NULL iff bias == NULL */
} vivim_conv_bwd_params;

/* ---- depthwise 3x3 / 3x3x3 convolution on token-major tensors ---------------
 * Replaces the ATen call behind modeling/vivim.py:57-68 (DWConv: nn.Conv3d(dim, dim, 3, 1, 1, groups=dim) applied
 * to x.transpose(1, 2).view(B, C, nf, H, W)); the reference has no kernel of its own there.
 * x, y: (batch, depth*height*width, channels), channels contiguous, token stride and batch stride free.
 * wt: (kd*9, channels) f32, TAP-major: wt[(kd*3 + kh)*3 + kw][c] = conv.weight[c][0][kd][kh][kw].
 * flip = 1 correlates with the reversed tap order: the input gradient of the same convolution.
 * act (ABI v7) fuses the Mlp's activation (modeling/vivim.py:99-106: act(dwconv(fc1(x))), act = nn.GELU, erf form) into
 * the convolution's epilogue: 0 y = conv; 1 y = gelu(conv); 2 y = aux * gelu'(conv) -- the gradient with respect to the
 * pre-activation, the convolution recomputed instead of stored (aux = the gradient of the activation's output, laid out
 * like y).  act != 0 requires flip == 0. */
typedef struct {
    int32_t batch, depth, height, width, channels;
    int32_t kd;                 /* 1 (2-D, 3x3) or 3 (3-D, 3x3x3) */
    int32_t itype;              /* dtype of x and y; channels % (16 / sizeof) == 0 and 16-byte aligned rows */
    int32_t flip;
    int64_t x_batch_stride, x_token_stride;
    int64_t y_batch_stride, y_token_stride;
    const void *x, *wt, *bias;  /* bias (channels) f32 or NULL */
    void *y;
    int32_t act, _pad1;
    const void *aux;            /* act == 2: (batch, tokens, channels) itype, 16-byte aligned rows; else ignored */
    int64_t aux_batch_stride, aux_token_stride;
} vivim_dwconv_params;

typedef struct {
    int32_t batch, depth, height, width, channels;
    int32_t kd;
    int32_t itype;              /* dtype of x and dy; channels % 2 == 0 */
    int32_t _pad0;
    int64_t x_batch_stride, x_token_stride;
    int64_t dy_batch_stride, dy_token_stride;
    const void *x, *dy;
    void *dwt;                  /* (kd*9, channels) f32 tap-major, pre-zeroed */
    void *dbias;                /* (channels) f32 pre-zeroed, or NULL */
} vivim_dwconv_wgrad_params;

/* The three scan directions of the v3 block (mamba_simple.py:220-262) as index maps over the token axis of a
 * frame-major clip, l = t*hw + p with t < nframes, p < hw = seqlen / nframes:
 *   direction 0: l            (forward in time)
 *   direction 1: seqlen-1-l   (xz.flip(-1), :231 / out_b.flip(-1), :264)
 *   direction 2: p*nframes+t  (chunk(nframes) + stack(-1) + flatten, :245-247; inverse at :261)
 * scatter: dst[b][c / csplit][g][c % csplit][m_g(l)] = scale * src[b][c][l]   for g = 0, 1, 2    (one read, three writes)
 * gather : dst[b][c][l] = scale * sum_g src[b][c / csplit][g][c % csplit][m_g(l)]               (three reads, one write)
 * They replace flip + stack/permute copies and the out + out_b + out_s sum, and are each other's gradient. */
typedef struct {
    int32_t batch, channels, seqlen, nframes;
    int32_t csplit;              /* channels per half: the stacked tensor is (batch, channels/csplit, 3, csplit, seqlen) */
    int32_t itype;               /* VIVIM_F32 / F16 / BF16 */
    float scale;
    int32_t _pad0;
    int64_t flat_batch_stride, flat_c_stride;                  /* the (batch, channels, seqlen) side; unit seqlen stride */
    int64_t stk_batch_stride, stk_half_stride, stk_dir_stride, stk_c_stride;   /* the stacked side; unit seqlen stride */
    const void *src;
    void *dst;
} vivim_dir_params;

/* ---- single-token steps for streaming inference ------------------------------------------
 * causal_conv1d_update (causal_conv1d.cpp:270-327, causal_conv1d_update.cu:26-66): conv_state is shifted left by one
 * along width, x is appended, out = sum_w conv_state[w] * weight[w] (+ bias) (silu). */
typedef struct {
    int32_t batch, dim, width;           /* width in [2, 4] */
    int32_t itype;                       /* x, conv_state, out */
    int32_t wtype;                       /* weight, bias */
    int32_t silu_activation;
    int64_t x_batch_stride, x_c_stride;
    int64_t state_batch_stride, state_c_stride, state_w_stride;
    int64_t weight_c_stride, weight_width_stride;
    int64_t out_batch_stride, out_c_stride;
    const void *x;              /* (batch, dim) */
    void *conv_state;           /* (batch, dim, width), updated in place */
    const void *weight;         /* (dim, width) */
    const void *bias;           /* (dim) or NULL */
    void *out;                  /* (batch, dim) */
} vivim_conv_update_params;

/* selective_state_update (mamba_ssm/ops/triton/selective_state_update.py:21-155): dt' = (softplus)(dt + dt_bias);
 * state = state * exp(dt' * A) + dt' * B * x (in place); out = sum_n state * C + D * x, gated by silu(z). */
typedef struct {
    int32_t batch, dim, dstate;
    int32_t itype;                       /* x, dt, B, C, z, out */
    int32_t stype;                       /* state: VIVIM_F32 or the same as itype */
    int32_t dt_softplus;
    int64_t state_batch_stride, state_d_stride, state_n_stride;
    int64_t x_batch_stride, x_d_stride, dt_batch_stride, dt_d_stride;
    int64_t A_d_stride, A_n_stride;
    int64_t B_batch_stride, B_n_stride, C_batch_stride, C_n_stride;
    int64_t z_batch_stride, z_d_stride, out_batch_stride, out_d_stride;
    void *state;                /* (batch, dim, dstate), updated in place */
    const void *x, *dt;         /* (batch, dim) */
    const void *A;              /* (dim, dstate) f32 */
    const void *B, *C;          /* (batch, dstate) */
    const void *D;              /* (dim) f32 or NULL */
    const void *z;              /* (batch, dim) or NULL */
    const void *dt_bias;        /* (dim) f32 or NULL */
    void *out;                  /* (batch, dim) */
} vivim_state_update_params;

/* ---- LayerNorm over the channels of a CHANNEL-major token tensor ---------------------------------------
 * Replaces the ATen calls behind modeling/vivim.py:155-156 (self.norm(x_flat) with x_flat = x.reshape(B, C, L).transpose(-1, -2):
 * a (B, L, C) view with strides (C*L, 1, L)); the reference has no kernel of its own there.
 *   forward : y[b][t][c] = (x[b][c][t] - mean[b][t]) * rstd[b][t] * weight[c] + bias[c]      mean / rstd over c, biased variance
 *   backward: dx[b][c][t] (channel-major like x), dweight[c] += sum dy * xhat, dbias[c] += sum dy   (f32, pre-zeroed by the caller;
 *             the per-tile partial sums go through `workspace`, vivim_layernorm_bwd_workspace_bytes() of it, added up by a
 *             second small kernel on the same stream: required whenever dweight or dbias is given)
 * x, dx: itype, unit token stride, 16-byte aligned rows, seqlen a whole number of 16-byte pieces; y, dy: otype = VIVIM_F32 (what
 * autocast makes of layer_norm) or itype, unit channel stride; channels <= 512. */
typedef struct {
    int32_t batch, seqlen, channels;
    int32_t itype, otype;
    float eps;
    int64_t x_batch_stride, x_c_stride;      /* x: (batch, channels, seqlen) memory, token stride 1 */
    int64_t y_batch_stride, y_token_stride;  /* y and dy: (batch, seqlen, channels), channel stride 1 */
    int64_t dx_batch_stride, dx_c_stride;    /* dx: laid out like x */
    const void *x;
    const void *weight, *bias;               /* (channels) f32, or NULL (1 / 0) */
    void *y;                                 /* forward output */
    void *mean, *rstd;                       /* (batch, seqlen) f32: written by the forward, read by the backward */
    const void *dy;                          /* backward input */
    void *dx;                                /* backward outputs */
    void *dweight, *dbias;                   /* (channels) f32 pre-zeroed, or NULL */
    void *workspace;                         /* backward scratch (see above), 16-byte aligned; contents undefined afterwards */
} vivim_layernorm_params;

/* ---- weight-gradient products of the fused inner op's backward (mamba_ssm/ops/selective_scan_interface.py:273, 276) ------------
 * out[g][i][j] += sum_t a[g][i][t] * b[g][j][t]: both operands with unit stride along t (the grouped op keeps everything
 * channel-major), t = every token of every clip.  Replaces the two einsum calls there (ddelta_proj_weight: a = ddelta (d, B*l),
 * b = x_dbl[:, :R]^T; dx_proj_weight: a = dx_dbl^T, b = conv1d_out (d, B*l)) with a split-k MFMA kernel; f16 / bf16 operands,
 * f32 accumulation and output (pre-zeroed by the caller: the splits add atomically).  k % 8 == 0, 16-byte aligned rows. */
typedef struct {
    int32_t groups, m, n, k;
    int32_t itype;                              /* VIVIM_F16 or VIVIM_BF16, both operands */
    int32_t _pad0;
    int64_t a_group_stride, a_row_stride;       /* elements */
    int64_t b_group_stride, b_row_stride;
    int64_t out_group_stride, out_row_stride;   /* out: (groups, m, n) f32, unit stride along n */
    const void *a, *b;
    void *out;
} vivim_wgrad_nt_params;

int vivim_abi_version(void);
const char *vivim_last_error(void);

/* sizeof() of a params struct as this library was compiled, so a foreign-language binding can assert
 * its own layout: which = 0 ssm_fwd, 1 ssm_bwd, 2 conv_fwd, 3 conv_bwd, 4 dwconv, 5 dwconv_wgrad, 6 dir, 7 conv_update,
 * 8 state_update, 9 layernorm, 10 wgrad_nt; 0 for anything else. */
size_t vivim_sizeof(int which);

/* Tokens per checkpoint row of `x`: n_chunks = ceil(seqlen / vivim_scan_ckpt_len(f)).  Depends on the sizes and flags in
 * `f` (not on its pointers) and on the forward tuning value: 16 * (dstate / 16) for the shapes the lanes = states
 * backward takes (variable B / C, dstate 16 / 32 / 64), vivim_scan_chunk_len() otherwise.  The forward call, the
 * backward call and the allocation of `x` must see the same value. */
int vivim_scan_ckpt_len(const vivim_ssm_fwd_params *f);
/* The checkpoint length of the shapes the call above does not special-case. */
int vivim_scan_chunk_len(int itype);

/* Scratch the backward wants for splitting the token axis over workgroups (carries of the reverse
 * recurrence per (batch, channel, segment, state)); depends only on the sizes in `f`. */
size_t vivim_scan_bwd_workspace_bytes(const vivim_ssm_fwd_params *f);
/* Scratch the forward wants for its token-axis split (per-segment end states); 0 when the shape takes a kernel
 * that needs none. */
size_t vivim_scan_fwd_workspace_bytes(const vivim_ssm_fwd_params *f);

/* Kernel-selection override for tuning and tests: which = 0 forward scan (0 automatic, 1 n-split K=8, 2 n-split K=4,
 * 3 generic, 5 lanes=channels, 6 lanes=states), which = 1 backward scan (0 automatic, 1 / 2 lanes=tokens kernel with
 * 8 / 4 waves per workgroup, 3 generic, 4 lanes=states first generation, 5 lanes=states second generation = what
 * automatic takes for dstate 16 on 16-byte aligned rows).  Forward values 1-3 also select the long checkpoint rows
 * (vivim_scan_ckpt_len), which the lanes=states backward cannot use.  Returns the previous value, -1 on a bad argument.  Initial values come from VIVIM_FWD_VARIANT / VIVIM_BWD_VARIANT.  The forward workspace size depends on
 * the forward setting: query it after changing it. */
int vivim_set_tuning(int which, int value);

int vivim_selective_scan_fwd(const vivim_ssm_fwd_params *p, void *stream);
int vivim_selective_scan_bwd(const vivim_ssm_bwd_params *p, void *stream);
int vivim_causal_conv1d_fwd(const vivim_conv_fwd_params *p, void *stream);
int vivim_causal_conv1d_bwd(const vivim_conv_bwd_params *p, void *stream);
int vivim_dwconv_fwd(const vivim_dwconv_params *p, void *stream);            /* also the input gradient (flip = 1) */
int vivim_dwconv_wgrad(const vivim_dwconv_wgrad_params *p, void *stream);
int vivim_dir_scatter(const vivim_dir_params *p, void *stream);
int vivim_dir_gather(const vivim_dir_params *p, void *stream);
int vivim_causal_conv1d_update(const vivim_conv_update_params *p, void *stream);
int vivim_selective_state_update(const vivim_state_update_params *p, void *stream);
int vivim_layernorm_cm_fwd(const vivim_layernorm_params *p, void *stream);
int vivim_layernorm_cm_bwd(const vivim_layernorm_params *p, void *stream);
size_t vivim_layernorm_bwd_workspace_bytes(const vivim_layernorm_params *p);   /* from batch, seqlen, channels, itype */
int vivim_wgrad_nt(const vivim_wgrad_nt_params *p, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VIVIM_HIP_H */
