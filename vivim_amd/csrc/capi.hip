// capi.hip -- the extern "C" boundary of libvivim_hip.so (see include/vivim_hip.h).
// Host-side checks mirror the TORCH_CHECKs of the reference bindings that still make sense below the
// tensor layer (selective_scan.cpp:233-304, 352-438; causal_conv1d.cpp:135-163, 198-238).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vivim_hip.h"

namespace vivim {
bool conv_fwd_dispatch(const vivim_conv_fwd_params&, hipStream_t);
bool conv_bwd_dispatch(const vivim_conv_bwd_params&, hipStream_t);
bool conv_cl_fwd_dispatch(const vivim_conv_fwd_params&, hipStream_t);                // conv1d_cl.hip (channel-last)
bool conv_cl_bwd_dispatch(const vivim_conv_bwd_params&, hipStream_t);
bool ssm_fwd_dispatch(const vivim_ssm_fwd_params&, hipStream_t);
bool dwconv_fwd_dispatch(const vivim_dwconv_params&, hipStream_t);
bool dwconv_wgrad_dispatch(const vivim_dwconv_wgrad_params&, hipStream_t);
template <bool GATHER> bool dir_dispatch(const vivim_dir_params&, hipStream_t);     // dirmap.hip
void conv_update_launch(const vivim_conv_update_params&, hipStream_t);               // update.hip
void state_update_launch(const vivim_state_update_params&, hipStream_t);
bool ssm_bwd_dispatch(const vivim_ssm_bwd_params&, hipStream_t);
int scan_chunk_len(int itype);
int scan_ckpt_len(const vivim_ssm_fwd_params&);
size_t scan_bwd_workspace_bytes(const vivim_ssm_fwd_params&);
bool layernorm_dispatch(const vivim_layernorm_params&, bool bwd, hipStream_t);   // layernorm.hip
size_t layernorm_bwd_workspace_bytes(const vivim_layernorm_params&);
bool wgrad_nt_dispatch(const vivim_wgrad_nt_params&, hipStream_t);                 // wgrad.hip
size_t scan_fwd_workspace_bytes(const vivim_ssm_fwd_params&);
}  // namespace vivim

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define VCHECK(cond)                                                                   \
    do {                                                                               \
        if (!(cond)) return fail(VIVIM_ERR_INVALID, "%s: check failed: %s", __func__, #cond); \
    } while (0)

static int after_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VIVIM_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return VIVIM_OK;
}

static bool dtype_ok(int t) { return t == VIVIM_F32 || t == VIVIM_F16 || t == VIVIM_BF16; }

static int check_ssm_fwd(const vivim_ssm_fwd_params* p, bool is_bwd) {
    VCHECK(p != nullptr);
    VCHECK(dtype_ok(p->itype));
    VCHECK(p->batch > 0 && p->dim > 0 && p->seqlen > 0 && p->dstate > 0 && p->n_groups > 0);
    VCHECK(p->dstate <= 256);                       // selective_scan.cpp:262
    VCHECK(p->dim % p->n_groups == 0);
    VCHECK(p->u && p->delta && p->A && p->B && p->C);
    VCHECK(p->x != nullptr || (is_bwd && p->seqlen <= vivim::scan_ckpt_len(*p)));
    if (!is_bwd) VCHECK(p->out != nullptr);
    if (p->z) {
        if (!is_bwd) VCHECK(p->out_z != nullptr);
        else VCHECK(p->out != nullptr);             // selective_scan.cpp:423 (saved out needed for dz)
    }
    if (p->is_variable_B != p->is_variable_C)
        return fail(VIVIM_ERR_UNSUPPORTED,
                    "selective_scan: mixed constant/variable B and C is not built (Vivim uses variable B and C)");
    if (!p->is_variable_B) VCHECK(p->n_groups == 1);
    return VIVIM_OK;
}

namespace vivim {
static int g_tune[2] = {-1, -1};     // host-side; accessed with relaxed atomics
static int tuning_get(int which, const char* env) {
    int v = __atomic_load_n(&g_tune[which], __ATOMIC_RELAXED);
    if (v < 0) {
        const char* e = getenv(env);
        v = e ? atoi(e) : 0;
        if (v < 0) v = 0;
        __atomic_store_n(&g_tune[which], v, __ATOMIC_RELAXED);
    }
    return v;
}
int tuning_fwd_variant() { return tuning_get(0, "VIVIM_FWD_VARIANT"); }
int tuning_bwd_variant() { return tuning_get(1, "VIVIM_BWD_VARIANT"); }
}  // namespace vivim

extern "C" {

int vivim_set_tuning(int which, int value) {
    if (which < 0 || which > 1 || value < 0) return -1;
    const int prev = which == 0 ? vivim::tuning_fwd_variant() : vivim::tuning_bwd_variant();
    __atomic_store_n(&vivim::g_tune[which], value, __ATOMIC_RELAXED);
    return prev;
}
int vivim_abi_version(void) { return VIVIM_ABI_VERSION; }
const char* vivim_last_error(void) { return g_err; }
int vivim_scan_chunk_len(int itype) { return vivim::scan_chunk_len(itype); }
int vivim_scan_ckpt_len(const vivim_ssm_fwd_params* f) { return f ? vivim::scan_ckpt_len(*f) : 0; }
size_t vivim_scan_bwd_workspace_bytes(const vivim_ssm_fwd_params* f) {
    return f ? vivim::scan_bwd_workspace_bytes(*f) : 0;
}
size_t vivim_scan_fwd_workspace_bytes(const vivim_ssm_fwd_params* f) {
    return f ? vivim::scan_fwd_workspace_bytes(*f) : 0;
}
size_t vivim_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(vivim_ssm_fwd_params);
        case 1: return sizeof(vivim_ssm_bwd_params);
        case 2: return sizeof(vivim_conv_fwd_params);
        case 3: return sizeof(vivim_conv_bwd_params);
        case 4: return sizeof(vivim_dwconv_params);
        case 5: return sizeof(vivim_dwconv_wgrad_params);
        case 6: return sizeof(vivim_dir_params);
        case 7: return sizeof(vivim_conv_update_params);
        case 8: return sizeof(vivim_state_update_params);
        case 9: return sizeof(vivim_layernorm_params);
        case 10: return sizeof(vivim_wgrad_nt_params);
    }
    return 0;
}

int vivim_selective_scan_fwd(const vivim_ssm_fwd_params* p, void* stream) {
    if (int rc = check_ssm_fwd(p, false)) return rc;
    if (!vivim::ssm_fwd_dispatch(*p, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "selective_scan_fwd not implemented for input type %d", p->itype);
    return after_launch("selective_scan_fwd");
}

int vivim_selective_scan_bwd(const vivim_ssm_bwd_params* p, void* stream) {
    VCHECK(p != nullptr);
    if (int rc = check_ssm_fwd(&p->f, true)) return rc;
    VCHECK(p->dout && p->du && p->ddelta && p->dA && p->dB && p->dC);
    VCHECK((p->f.D == nullptr) == (p->dD == nullptr));
    VCHECK((p->f.delta_bias == nullptr) == (p->ddelta_bias == nullptr));
    VCHECK((p->f.z == nullptr) == (p->dz == nullptr));
    if (!vivim::ssm_bwd_dispatch(*p, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "selective_scan_bwd not implemented for input type %d", p->f.itype);
    return after_launch("selective_scan_bwd");
}

static int check_conv(const vivim_conv_fwd_params* p) {
    VCHECK(p != nullptr);
    VCHECK(dtype_ok(p->itype) && dtype_ok(p->wtype));
    VCHECK(p->batch > 0 && p->dim > 0 && p->seqlen > 0);
    VCHECK(p->batch <= 65535 && p->dim <= 65535);
    if (!(p->width >= 2 && p->width <= 4))          // causal_conv1d.cpp:157
        return fail(VIVIM_ERR_INVALID, "causal_conv1d only supports width between 2 and 4");
    VCHECK(p->x && p->weight);
    if (p->x_l_stride != 1 && p->x_c_stride != 1)   // causal_conv1d.cpp:151-152
        return fail(VIVIM_ERR_INVALID, "causal_conv1d: x must have unit stride along seqlen or along channels");
    return VIVIM_OK;
}

// causal_conv1d.cpp:151: is_channel_last = x.stride(1) == 1 && x.stride(2) > 1
static bool conv_channel_last(const vivim_conv_fwd_params* p) { return p->x_c_stride == 1 && p->x_l_stride > 1; }

int vivim_causal_conv1d_fwd(const vivim_conv_fwd_params* p, void* stream) {
    if (int rc = check_conv(p)) return rc;
    VCHECK(p->out != nullptr);
    const bool cl = conv_channel_last(p);
    if (cl) { VCHECK(p->out_c_stride == 1); } else { VCHECK(p->out_l_stride == 1); }
    if (!(cl ? vivim::conv_cl_fwd_dispatch(*p, static_cast<hipStream_t>(stream))
             : vivim::conv_fwd_dispatch(*p, static_cast<hipStream_t>(stream))))
        return fail(VIVIM_ERR_UNSUPPORTED, "causal_conv1d_fwd not implemented for input type %d / weight type %d",
                    p->itype, p->wtype);
    return after_launch("causal_conv1d_fwd");
}

int vivim_causal_conv1d_bwd(const vivim_conv_bwd_params* p, void* stream) {
    VCHECK(p != nullptr);
    if (int rc = check_conv(&p->f)) return rc;
    VCHECK(p->dout && p->dx && p->dweight);
    const bool cl = conv_channel_last(&p->f);
    if (cl) { VCHECK(p->dout_c_stride == 1 && p->dx_c_stride == 1); }      // causal_conv1d.cpp:221, 237
    else    { VCHECK(p->dout_l_stride == 1 && p->dx_l_stride == 1); }      // causal_conv1d.cpp:220, 236
    VCHECK((p->f.bias == nullptr) == (p->dbias == nullptr));
    if (!(cl ? vivim::conv_cl_bwd_dispatch(*p, static_cast<hipStream_t>(stream))
             : vivim::conv_bwd_dispatch(*p, static_cast<hipStream_t>(stream))))
        return fail(VIVIM_ERR_UNSUPPORTED, "causal_conv1d_bwd not implemented for input type %d / weight type %d",
                    p->f.itype, p->f.wtype);
    return after_launch("causal_conv1d_bwd");
}

static int check_dw_dims(int batch, int depth, int height, int width, int channels, int kd, int itype) {
    VCHECK(dtype_ok(itype));
    VCHECK(batch > 0 && depth > 0 && height > 0 && width > 0 && channels > 0);
    VCHECK(kd == 1 || kd == 3);
    VCHECK(batch <= 65535);
    return VIVIM_OK;
}

int vivim_dwconv_fwd(const vivim_dwconv_params* p, void* stream) {
    VCHECK(p != nullptr);
    if (int rc = check_dw_dims(p->batch, p->depth, p->height, p->width, p->channels, p->kd, p->itype)) return rc;
    VCHECK(p->x && p->wt && p->y);
    const int64_t cv = p->itype == VIVIM_F32 ? 4 : 8;
    VCHECK(p->channels % cv == 0 && p->x_token_stride % cv == 0 && p->x_batch_stride % cv == 0 &&
           p->y_token_stride % cv == 0 && p->y_batch_stride % cv == 0);
    VCHECK((reinterpret_cast<uintptr_t>(p->x) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->y) & 15) == 0);
    VCHECK(p->act >= 0 && p->act <= 2 && (p->act == 0 || p->flip == 0));
    if (p->act == 2)
        VCHECK(p->aux && (reinterpret_cast<uintptr_t>(p->aux) & 15) == 0 && p->aux_token_stride % cv == 0 &&
               p->aux_batch_stride % cv == 0);
    if (!vivim::dwconv_fwd_dispatch(*p, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "dwconv_fwd not implemented for input type %d", p->itype);
    return after_launch("dwconv_fwd");
}

int vivim_dwconv_wgrad(const vivim_dwconv_wgrad_params* p, void* stream) {
    VCHECK(p != nullptr);
    if (int rc = check_dw_dims(p->batch, p->depth, p->height, p->width, p->channels, p->kd, p->itype)) return rc;
    VCHECK(p->x && p->dy && p->dwt);
    VCHECK(p->channels % 2 == 0 && p->x_token_stride % 2 == 0 && p->x_batch_stride % 2 == 0 &&
           p->dy_token_stride % 2 == 0 && p->dy_batch_stride % 2 == 0);
    VCHECK((reinterpret_cast<uintptr_t>(p->x) & 7) == 0 && (reinterpret_cast<uintptr_t>(p->dy) & 7) == 0);
    if (!vivim::dwconv_wgrad_dispatch(*p, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "dwconv_wgrad not implemented for input type %d", p->itype);
    return after_launch("dwconv_wgrad");
}

static int check_dir(const vivim_dir_params* p) {
    VCHECK(p != nullptr);
    VCHECK(p->itype == VIVIM_F32 || p->itype == VIVIM_F16 || p->itype == VIVIM_BF16);
    VCHECK(p->batch > 0 && p->channels > 0 && p->seqlen > 0 && p->nframes > 0 && p->csplit > 0);
    VCHECK(p->batch <= 65535 && p->channels <= 65535);
    VCHECK(p->seqlen % p->nframes == 0 && p->channels % p->csplit == 0);
    VCHECK(p->src && p->dst);
    const int64_t e = p->itype == VIVIM_F32 ? 4 : 8;            // 16-byte vectors on both sides
    VCHECK(p->seqlen % e == 0);
    VCHECK((reinterpret_cast<uintptr_t>(p->src) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->dst) & 15) == 0);
    VCHECK(p->flat_batch_stride % e == 0 && p->flat_c_stride % e == 0 && p->stk_batch_stride % e == 0 &&
           p->stk_half_stride % e == 0 && p->stk_dir_stride % e == 0 && p->stk_c_stride % e == 0);
    return VIVIM_OK;
}

int vivim_dir_scatter(const vivim_dir_params* p, void* stream) {
    if (int rc = check_dir(p)) return rc;
    if (!vivim::dir_dispatch<false>(*p, static_cast<hipStream_t>(stream))) return fail(VIVIM_ERR_UNSUPPORTED, "dir_scatter: bad itype");
    return after_launch("dir_scatter");
}

int vivim_dir_gather(const vivim_dir_params* p, void* stream) {
    if (int rc = check_dir(p)) return rc;
    if (!vivim::dir_dispatch<true>(*p, static_cast<hipStream_t>(stream))) return fail(VIVIM_ERR_UNSUPPORTED, "dir_gather: bad itype");
    return after_launch("dir_gather");
}

int vivim_causal_conv1d_update(const vivim_conv_update_params* p, void* stream) {
    VCHECK(p != nullptr);
    VCHECK(dtype_ok(p->itype) && dtype_ok(p->wtype));
    VCHECK(p->batch > 0 && p->dim > 0 && p->batch <= 65535);
    if (p->width < 2 || p->width > 4)
        return fail(VIVIM_ERR_UNSUPPORTED, "causal_conv1d only supports width between 2 and 4");   // causal_conv1d.cpp:295
    VCHECK(p->x && p->conv_state && p->weight && p->out);
    vivim::conv_update_launch(*p, static_cast<hipStream_t>(stream));
    return after_launch("causal_conv1d_update");
}

int vivim_selective_state_update(const vivim_state_update_params* p, void* stream) {
    VCHECK(p != nullptr);
    VCHECK(dtype_ok(p->itype) && (p->stype == VIVIM_F32 || p->stype == p->itype));
    VCHECK(p->batch > 0 && p->dim > 0 && p->dstate > 0 && p->batch <= 65535);
    VCHECK(p->state && p->x && p->dt && p->A && p->B && p->C && p->out);
    vivim::state_update_launch(*p, static_cast<hipStream_t>(stream));
    return after_launch("selective_state_update");
}

static int check_layernorm(const vivim_layernorm_params* p) {
    VCHECK(p != nullptr);
    VCHECK(dtype_ok(p->itype) && (p->otype == VIVIM_F32 || p->otype == p->itype));
    VCHECK(p->batch > 0 && p->seqlen > 0 && p->channels > 0 && p->batch <= 65535);
    if (p->channels > 512)
        return fail(VIVIM_ERR_UNSUPPORTED, "layernorm_cm: more than 512 channels do not fit the backward's two LDS tiles");
    const int64_t e = p->itype == VIVIM_F32 ? 4 : 8;             // 16-byte vectors along the tokens of x / dx
    VCHECK(p->seqlen % e == 0 && p->x_batch_stride % e == 0 && p->x_c_stride % e == 0);
    VCHECK(p->x && (reinterpret_cast<uintptr_t>(p->x) & 15) == 0 && p->mean && p->rstd);
    return VIVIM_OK;
}

int vivim_layernorm_cm_fwd(const vivim_layernorm_params* p, void* stream) {
    if (int rc = check_layernorm(p)) return rc;
    VCHECK(p->y != nullptr);
    if (!vivim::layernorm_dispatch(*p, false, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "layernorm_cm_fwd not implemented for input type %d / output type %d", p->itype, p->otype);
    return after_launch("layernorm_cm_fwd");
}

size_t vivim_layernorm_bwd_workspace_bytes(const vivim_layernorm_params* p) {
    return p && p->batch > 0 && p->seqlen > 0 && p->channels > 0 ? vivim::layernorm_bwd_workspace_bytes(*p) : 0;
}

int vivim_layernorm_cm_bwd(const vivim_layernorm_params* p, void* stream) {
    if (int rc = check_layernorm(p)) return rc;
    VCHECK(p->dy && p->dx && (reinterpret_cast<uintptr_t>(p->dx) & 15) == 0);
    const int64_t e = p->itype == VIVIM_F32 ? 4 : 8;
    VCHECK(p->dx_batch_stride % e == 0 && p->dx_c_stride % e == 0);
    if ((p->dweight || p->dbias) && !p->workspace)
        return fail(VIVIM_ERR_INVALID, "layernorm_cm_bwd: dweight / dbias need the workspace (vivim_layernorm_bwd_workspace_bytes)");
    if (!vivim::layernorm_dispatch(*p, true, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "layernorm_cm_bwd not implemented for input type %d / output type %d", p->itype, p->otype);
    return after_launch("layernorm_cm_bwd");
}

int vivim_wgrad_nt(const vivim_wgrad_nt_params* p, void* stream) {
    VCHECK(p != nullptr);
    VCHECK(p->groups > 0 && p->groups <= 65535 && p->m > 0 && p->n > 0 && p->k > 0);
    VCHECK(p->a && p->b && p->out);
    if (p->itype != VIVIM_F16 && p->itype != VIVIM_BF16)
        return fail(VIVIM_ERR_UNSUPPORTED, "wgrad_nt takes f16 or bf16 operands (type %d given): f32 products stay on the library GEMM", p->itype);
    VCHECK(p->k % 8 == 0 && p->a_row_stride % 8 == 0 && p->b_row_stride % 8 == 0 && p->a_group_stride % 8 == 0 &&
           p->b_group_stride % 8 == 0);
    VCHECK((reinterpret_cast<uintptr_t>(p->a) & 15) == 0 && (reinterpret_cast<uintptr_t>(p->b) & 15) == 0);
    VCHECK((int64_t)((p->m + 63) / 64) * ((p->n + 15) / 16) <= 65535);
    if (!vivim::wgrad_nt_dispatch(*p, static_cast<hipStream_t>(stream)))
        return fail(VIVIM_ERR_UNSUPPORTED, "wgrad_nt not implemented for type %d", p->itype);
    return after_launch("wgrad_nt");
}

}  // extern "C"
