// update.hip -- single-token steps for streaming inference (include/vivim_hip.h: vivim_conv_update_params,
// vivim_state_update_params).  One thread per (batch, channel): the work is a few tens of bytes per thread and the
// launches are latency-bound (batch * dim threads), so the only design rule is coalescing: consecutive threads are
// consecutive channels, a thread's state row (width or dstate elements) is contiguous in the usual layouts, and a
// wave therefore touches one contiguous block.
#include "common.cuh"

namespace vivim {

__device__ __forceinline__ float ld_any(const void* p, int64_t i, int dt) {
    switch (dt) {
        case VIVIM_F32: return static_cast<const float*>(p)[i];
        case VIVIM_F16: return to_f32(static_cast<const f16_t*>(p)[i]);
        default:        return to_f32(static_cast<const bf16_t*>(p)[i]);
    }
}
__device__ __forceinline__ void st_any(void* p, int64_t i, int dt, float v) {
    switch (dt) {
        case VIVIM_F32: static_cast<float*>(p)[i] = v; break;
        case VIVIM_F16: static_cast<f16_t*>(p)[i] = from_f32<f16_t>(v); break;
        default:        static_cast<bf16_t*>(p)[i] = from_f32<bf16_t>(v); break;
    }
}

// causal_conv1d_update.cu:26-66
__global__ void __launch_bounds__(256) conv_update_kernel(const vivim_conv_update_params p) {
    const int d = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (d >= p.dim) return;
    const int W = p.width;
    const int64_t so = (int64_t)b * p.state_batch_stride + (int64_t)d * p.state_c_stride;
    float acc = p.bias ? ld_any(p.bias, d, p.wtype) : 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w >= W) break;
        // the new window: old window shifted left by one, the incoming sample last (values already in the I/O dtype)
        const float v = w + 1 < W ? ld_any(p.conv_state, so + (w + 1) * p.state_w_stride, p.itype)
                                  : ld_any(p.x, (int64_t)b * p.x_batch_stride + (int64_t)d * p.x_c_stride, p.itype);
        st_any(p.conv_state, so + w * p.state_w_stride, p.itype, v);   // thread-private row: ascending w is safe
        acc = fmaf(v, ld_any(p.weight, (int64_t)d * p.weight_c_stride + w * p.weight_width_stride, p.wtype), acc);
    }
    if (p.silu_activation) acc = acc * sigmoidf_fast(acc);
    st_any(p.out, (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_c_stride, p.itype, acc);
}

// selective_state_update.py:21-96
__global__ void __launch_bounds__(256) state_update_kernel(const vivim_state_update_params p) {
    const int d = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (d >= p.dim) return;
    const float x = ld_any(p.x, (int64_t)b * p.x_batch_stride + (int64_t)d * p.x_d_stride, p.itype);
    float dt = ld_any(p.dt, (int64_t)b * p.dt_batch_stride + (int64_t)d * p.dt_d_stride, p.itype);
    if (p.dt_bias) dt += static_cast<const float*>(p.dt_bias)[d];
    if (p.dt_softplus) dt = softplus_ref(dt);
    const float* __restrict__ A = static_cast<const float*>(p.A) + (int64_t)d * p.A_d_stride;
    const int64_t so = (int64_t)b * p.state_batch_stride + (int64_t)d * p.state_d_stride;
    const float dtx = dt * x, dt2 = dt * kLog2e;
    float acc = 0.0f;
    for (int n = 0; n < p.dstate; ++n) {
        const float Bn = ld_any(p.B, (int64_t)b * p.B_batch_stride + n * p.B_n_stride, p.itype);
        const float Cn = ld_any(p.C, (int64_t)b * p.C_batch_stride + n * p.C_n_stride, p.itype);
        const float s = fmaf(ld_any(p.state, so + n * p.state_n_stride, p.stype), fast_exp2(dt2 * A[n * p.A_n_stride]), dtx * Bn);
        st_any(p.state, so + n * p.state_n_stride, p.stype, s);
        acc = fmaf(s, Cn, acc);
    }
    if (p.D) acc = fmaf(static_cast<const float*>(p.D)[d], x, acc);
    if (p.z) {
        const float z = ld_any(p.z, (int64_t)b * p.z_batch_stride + (int64_t)d * p.z_d_stride, p.itype);
        acc *= z * sigmoidf_fast(z);
    }
    st_any(p.out, (int64_t)b * p.out_batch_stride + (int64_t)d * p.out_d_stride, p.itype, acc);
}

void conv_update_launch(const vivim_conv_update_params& p, hipStream_t s) {
    hipLaunchKernelGGL(conv_update_kernel, dim3((p.dim + 255) / 256, p.batch), dim3(256), 0, s, p);
}
void state_update_launch(const vivim_state_update_params& p, hipStream_t s) {
    hipLaunchKernelGGL(state_update_kernel, dim3((p.dim + 255) / 256, p.batch), dim3(256), 0, s, p);
}

}  // namespace vivim
