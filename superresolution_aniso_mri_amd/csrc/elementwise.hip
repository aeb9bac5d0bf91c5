// HBM-bound elementwise / reduction pieces of the ae_combined step:
//   lerp     z_mix[b] = a_from[b]*z[b] + a_to[b]*z[B+b]  (kwatsch/cardiac/trainer_ae.py:173,
//            kwatsch/brain/trainer_ae.py:264-266, generate_hr_volumes.py:88) + its gradient
//   mse      F.mse_loss(a, b) mean (kwatsch/base_trainer.py:177) + gradient, upstream scalar read on device
//   act_bwd  dpre = dout * act'(y) from the saved activation output (sigmoid of networks/acai_vanilla.py:98)
//   adam     torch.optim.Adam single-tensor semantics on one flat fp32 buffer (kwatsch/trainer_ae.py:29-30)
#include "aesr_kernels.h"

// ---- lerp ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lerp_fwd_kernel(const float* __restrict__ z, const float* __restrict__ af,
                                                       const float* __restrict__ at, float* __restrict__ zmix, int B,
                                                       size_t per4) {
    const size_t total = (size_t)B * per4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = idx / per4;
        const f32x4 v0 = ((const f32x4*)z)[idx], v1 = ((const f32x4*)z)[idx + total];
        ((f32x4*)zmix)[idx] = v0 * af[b] + v1 * at[b];
    }
}

__global__ __launch_bounds__(256) void lerp_bwd_kernel(const float* __restrict__ dmix, const float* __restrict__ af,
                                                       const float* __restrict__ at, float* __restrict__ dz, int B,
                                                       size_t per4) {
    const size_t total = (size_t)B * per4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = idx / per4;
        const f32x4 d = ((const f32x4*)dmix)[idx];
        ((f32x4*)dz)[idx] = d * af[b];
        ((f32x4*)dz)[idx + total] = d * at[b];
    }
}

// ---- mse -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             double* __restrict__ partial, size_t n) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void mean_finalize_kernel(const double* __restrict__ partial, int np, double inv_n,
                                                          float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 64) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) *out = (float)(s * inv_n);
}

// da = 2*(a-b)*g/n  (g: device scalar, upstream gradient of the loss)
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ g, float* __restrict__ da, size_t n,
                                                      float two_over_n) {
    const float k = two_over_n * g[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) da[i] = (a[i] - b[i]) * k;
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                      float* __restrict__ dpre, size_t n, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dpre[i] = dout[i] * act_grad_from_output(y[i], act, slope);
}

// ---- adam ----------------------------------------------------------------------------------------------------
// state[0] = step (float, exact below 2^24), state[1] = 1-beta1^t, state[2] = sqrt(1-beta2^t)
__global__ void adam_prep_kernel(float* __restrict__ state, float beta1, float beta2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float t = state[0] + 1.f;
        state[0] = t;
        state[1] = (float)(1.0 - pow((double)beta1, (double)t));
        state[2] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    }
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ state, size_t n, float lr, float beta1,
                                                        float beta2, float eps, float wd) {
    const float bc1 = state[1], bc2s = state[2];
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);          // torch: exp_avg.lerp_(grad, 1-beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;      // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

static inline int grid_for(size_t n, int cap) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}

int aesr_launch_lerp_fwd(const float* z, const float* af, const float* at, float* zmix, int B, size_t per, hipStream_t st) {
    hipLaunchKernelGGL(lerp_fwd_kernel, dim3(grid_for((size_t)B * per / 4, 4096)), dim3(256), 0, st, z, af, at, zmix, B, per / 4);
    AESR_LAUNCH_CHECK("lerp_fwd");
    return AESR_OK;
}

int aesr_launch_lerp_bwd(const float* dmix, const float* af, const float* at, float* dz, int B, size_t per, hipStream_t st) {
    hipLaunchKernelGGL(lerp_bwd_kernel, dim3(grid_for((size_t)B * per / 4, 4096)), dim3(256), 0, st, dmix, af, at, dz, B, per / 4);
    AESR_LAUNCH_CHECK("lerp_bwd");
    return AESR_OK;
}

int aesr_launch_mse_fwd(const float* a, const float* b, double* partial, int np, float* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(np), dim3(256), 0, st, a, b, partial, n);
    AESR_LAUNCH_CHECK("sqdiff_partial");
    hipLaunchKernelGGL(mean_finalize_kernel, dim3(1), dim3(64), 0, st, partial, np, 1.0 / (double)n, out);
    AESR_LAUNCH_CHECK("mean_finalize");
    return AESR_OK;
}

int aesr_launch_mse_bwd(const float* a, const float* b, const float* g, float* da, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, st, a, b, g, da, n, (float)(2.0 / (double)n));
    AESR_LAUNCH_CHECK("mse_bwd");
    return AESR_OK;
}

int aesr_launch_act_bwd(const float* dout, const float* y, float* dpre, size_t n, int act, float slope, hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, st, dout, y, dpre, n, act, slope);
    AESR_LAUNCH_CHECK("act_bwd");
    return AESR_OK;
}

int aesr_launch_adam(float* p, const float* g, float* m, float* v, float* state, size_t n, float lr, float beta1,
                     float beta2, float eps, float wd, hipStream_t st) {
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, st, state, beta1, beta2);
    AESR_LAUNCH_CHECK("adam_prep");
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, st, p, g, m, v, state, n, lr, beta1, beta2, eps, wd);
    AESR_LAUNCH_CHECK("adam_step");
    return AESR_OK;
}
