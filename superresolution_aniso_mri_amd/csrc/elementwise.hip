// HBM-bound elementwise / reduction pieces of the ae_combined step:
//   lerp     z_mix[b] = a_from[b]*z[b] + a_to[b]*z[B+b]  (kwatsch/cardiac/trainer_ae.py:173,
//            kwatsch/brain/trainer_ae.py:264-266, generate_hr_volumes.py:88) + its gradient
//   mse      F.mse_loss(a, b) mean (kwatsch/base_trainer.py:177) + gradient, upstream scalar read on device
//   act_bwd  dpre = dout * act'(y) from the saved activation output (sigmoid of networks/acai_vanilla.py:98)
//   adam     torch.optim.Adam single-tensor semantics on one flat fp32 buffer (kwatsch/trainer_ae.py:29-30)
#include "../../include/aesr_hip.h"
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

// the decoder's input of the ae_combined step in one pass: zcat = [z (2B rows) | z_mix (B rows)] -- the lerp and the
// concatenation that feeds dec([z | z_mix]) (kwatsch/cardiac/trainer_ae.py:20-30) -- and its gradient
// dz[b] = g[b] + a_from[b]*g[2B+b], dz[B+b] = g[B+b] + a_to[b]*g[2B+b]
__global__ __launch_bounds__(256) void lerp_cat_fwd_kernel(const float* __restrict__ z, const float* __restrict__ af,
                                                           const float* __restrict__ at, float* __restrict__ zcat, int B, size_t per4) {
    const size_t total = (size_t)B * per4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = idx / per4;
        const f32x4 v0 = ((const f32x4*)z)[idx], v1 = ((const f32x4*)z)[idx + total];
        ((f32x4*)zcat)[idx] = v0;
        ((f32x4*)zcat)[idx + total] = v1;
        ((f32x4*)zcat)[idx + 2 * total] = v0 * af[b] + v1 * at[b];
    }
}

// slice synthesis (generate_hr_volumes.py:46-53,88): for every alpha a_k and every pair of neighbouring slices (i, i + 1)
//   out[k][i] = act(a_k * z[i + 1] + (1 - a_k) * z[i])
// in ONE launch; z = latents (act none), or the PRE-ACTIVATIONS of the decoder's first convolution: that layer is linear, so its
// output for a latent mix is the same mix of its outputs for the two slices (bias included: the weights add up to 1) -- the layer
// runs once per slice instead of once per synthesised slice, and its LeakyReLU is applied here.
struct LerpMultiArgs { float a[16]; int n; };
__global__ __launch_bounds__(256) void lerp_multi_kernel(const float* __restrict__ z, float* __restrict__ out, int Zm1, size_t per4,
                                                         LerpMultiArgs al, float nslope) {
    const size_t total = (size_t)Zm1 * per4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const f32x4 lo = ((const f32x4*)z)[idx], hi = ((const f32x4*)z)[idx + per4];
        for (int k = 0; k < al.n; ++k) {
            f32x4 v = hi * al.a[k] + lo * (1.f - al.a[k]);
            const f32x4 vs = v * nslope;                      // none / LeakyReLU / ReLU as max(x, slope * x), 0 <= slope <= 1
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], vs[e]);
            ((f32x4*)out)[(size_t)k * total + idx] = v;
        }
    }
}

// the super-resolved volume (generate_hr_volumes.py:57-67): slot i * (n + 1) = slice i of `orig`, slot i * (n + 1) + k + 1 = synthesised slice
// synth[k][i], everything clamped to [lo, hi] -- one pass instead of n + 1 strided copies and a clamp over the result
__global__ __launch_bounds__(256) void interleave_clamp_kernel(const float* __restrict__ orig, const float* __restrict__ synth, float* __restrict__ out,
                                                               int Z, int n, size_t per4, float lo, float hi) {
    const size_t total = ((size_t)(Z - 1) * (n + 1) + 1) * per4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t slot = idx / per4, e = idx - slot * per4;
        const size_t i = slot / (n + 1);
        const int k = (int)(slot - i * (n + 1));
        f32x4 v = k == 0 ? ((const f32x4*)orig)[i * per4 + e] : ((const f32x4*)synth)[((size_t)(k - 1) * (Z - 1) + i) * per4 + e];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fminf(fmaxf(v[c], lo), hi);
        ((f32x4*)out)[idx] = v;
    }
}

__global__ __launch_bounds__(256) void lerp_cat_bwd_kernel(const float* __restrict__ g, const float* __restrict__ af,
                                                           const float* __restrict__ at, float* __restrict__ dz, int B, size_t per4) {
    const size_t total = (size_t)B * per4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = idx / per4;
        const f32x4 gm = ((const f32x4*)g)[idx + 2 * total];
        ((f32x4*)dz)[idx] = ((const f32x4*)g)[idx] + gm * af[b];
        ((f32x4*)dz)[idx + total] = ((const f32x4*)g)[idx + total] + gm * at[b];
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

// ---- the three mean-squared errors of the ae_combined step in ONE launch -------------------------------------------------
// out[0] = m1 + lam * m2, out[1] = m1, out[2] = lam * m2, out[3] = m3 (m_k = mean((a_k - b_k)^2), fp64 sums).  Every block leaves
// its three partial sums in `ws`, takes a ticket, and the block that draws the last one adds the partials up in block order:
// bitwise reproducible (no floating-point atomics), one graph node instead of the six of three aesr_mse_fwd calls + the torch
// glue that combined them.  ws[3 * gridDim.x] is the ticket counter: zero before the first launch, left at zero.
struct Mse3Args {
    const float* a[3]; const float* b[3]; size_t n[3];
    const float* lam; double* ws; float* out;
};
__global__ __launch_bounds__(256) void mse3_fwd_kernel(Mse3Args p) {
    __shared__ double red[3][4];
    __shared__ unsigned ticket;
    double s[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* a = p.a[k];
        const float* b = p.b[k];
        if (!a) continue;
        const size_t n4 = p.n[k] >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const f32x4 d = *(const f32x4*)(a + 4 * i) - *(const f32x4*)(b + 4 * i);
            s[k] += ((double)(d[0] * d[0]) + (double)(d[1] * d[1])) + ((double)(d[2] * d[2]) + (double)(d[3] * d[3]));
        }
        if (blockIdx.x == 0 && threadIdx.x < (p.n[k] & 3)) {
            const float d = a[4 * n4 + threadIdx.x] - b[4 * n4 + threadIdx.x];
            s[k] += (double)(d * d);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_down(s[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    unsigned* counter = (unsigned*)(p.ws + 3 * gridDim.x);
    if (threadIdx.x == 0) {
        // write-through stores, drained, then the ticket: no agent release (an L2 write-back) per workgroup; only the last one acquires
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __hip_atomic_store(p.ws + 3 * blockIdx.x + k, (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double t[3] = {0.0, 0.0, 0.0};
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] += __hip_atomic_load(p.ws + 3 * i + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t[k] += __shfl_down(t[k], o, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = t[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double m[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) m[k] = p.a[k] ? ((red[k][0] + red[k][1]) + (red[k][2] + red[k][3])) / (double)p.n[k] : 0.0;
        const float m1 = (float)m[0], m2w = p.lam[0] * (float)m[1];
        p.out[0] = m1 + m2w;
        p.out[1] = m1;
        p.out[2] = m2w;
        p.out[3] = (float)m[2];
        *counter = 0u;
    }
}

// gradient of out[0] of mse3_fwd with respect to a1 and a2: d1 = g * 2 (a1 - b1) / n1, d2 = g * lam * 2 (a2 - b2) / n2
__global__ __launch_bounds__(256) void mse3_bwd_kernel(const float* __restrict__ a1, const float* __restrict__ b1, size_t n1,
                                                       const float* __restrict__ a2, const float* __restrict__ b2, size_t n2,
                                                       const float* __restrict__ lam, const float* __restrict__ g,
                                                       float* __restrict__ d1, float* __restrict__ d2, float k1, float k2) {
    const float g1 = k1 * g[0], g2 = k2 * g[0] * lam[0];
    const size_t q1 = n1 >> 2, q2 = n2 >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < q1 + q2; i += (size_t)gridDim.x * 256) {
        if (i < q1) *(f32x4*)(d1 + 4 * i) = (*(const f32x4*)(a1 + 4 * i) - *(const f32x4*)(b1 + 4 * i)) * g1;
        else *(f32x4*)(d2 + 4 * (i - q1)) = (*(const f32x4*)(a2 + 4 * (i - q1)) - *(const f32x4*)(b2 + 4 * (i - q1))) * g2;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < (n1 & 3)) d1[4 * q1 + threadIdx.x] = (a1[4 * q1 + threadIdx.x] - b1[4 * q1 + threadIdx.x]) * g1;
        if (threadIdx.x < (n2 & 3)) d2[4 * q2 + threadIdx.x] = (a2[4 * q2 + threadIdx.x] - b2[4 * q2 + threadIdx.x]) * g2;
    }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                      float* __restrict__ dpre, size_t n, int act, float slope) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dpre[i] = dout[i] * act_grad_from_output(y[i], act, slope);
}

// ---- adam ----------------------------------------------------------------------------------------------------
// state: 8 floats = {step t done so far, 1-beta1^(t+1), sqrt(1-beta2^(t+1)), ticket counter (bits), beta1^(t+1) and beta2^(t+1) as two
// doubles}: the bias corrections of the step ABOUT to run are in the buffer (aesr_adam_state_init for a given t), every workgroup just
// reads them, and the workgroup that finishes last advances the step and the running powers (doubles: 1e-11 after 10^5 steps) for the
// next launch -- one launch, no pow() on any workgroup's critical path.  zero_g: the gradient buffer is left at zero.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ state, size_t n, float lr,
                                                        double beta1d, double beta2d, float eps, float wd, int zero_g) {
    const float beta1 = (float)beta1d, beta2 = (float)beta2d;        // element-wise arithmetic in fp32 as torch's; the powers in double
    const float bc1 = state[1], bc2s = state[2];
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i];
        const float pi = p[i];
        if (zero_g) g[i] = 0.f;
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);          // torch: exp_avg.lerp_(grad, 1-beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;      // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] = pi - step_size * (mi / denom);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // no data is handed over (the last workgroup only WRITES the state), so no fence: each workgroup read the state before its ticket
        unsigned* counter = (unsigned*)(state + 3);
        if (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            double* pw = (double*)(state + 4);
            const double b1p = pw[0] * beta1d, b2p = pw[1] * beta2d;
            pw[0] = b1p;
            pw[1] = b2p;
            state[0] += 1.f;
            state[1] = (float)(1.0 - b1p);
            state[2] = (float)sqrt(1.0 - b2p);
            *counter = 0u;
        }
    }
}

// ---- space-to-depth (2x2): the stride-2 2x2 convolution of networks/acai_vanilla_strided.py:19 as a 1x1 conv -------------
// out[n,yo,xo,(ky*2+kx)*C + c] = x[n,2yo+ky,2xo+kx,c]   (Ho = H/2, Wo = W/2; an odd last row / column is dropped)
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C,
                                                  int inverse) {
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    if (!inverse) {
        const int total = N * Ho * Wo * 4 * C4;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
            const int c4 = idx % C4;
            int r = idx / C4;
            const int q = r & 3;
            r >>= 2;
            const int xo = r % Wo;
            r /= Wo;
            const int yo = r % Ho;
            const int n = r / Ho;
            *(f32x4*)(out + (size_t)idx * 4) =
                *(const f32x4*)(x + ((size_t)(n * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1)) * C + c4 * 4);
        }
    } else {
        // x is the [N,Ho,Wo,4C] gradient, out the [N,H,W,C] gradient (zero in a dropped odd row / column)
        const int total = N * H * W * C4;
        for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
            const int c4 = idx % C4;
            int r = idx / C4;
            const int xx = r % W;
            r /= W;
            const int yy = r % H;
            const int n = r / H;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int yo = yy >> 1, xo = xx >> 1;
            if (yo < Ho && xo < Wo)
                v = *(const f32x4*)(x + (((size_t)(n * Ho + yo) * Wo + xo) * 4 + (yy & 1) * 2 + (xx & 1)) * C + c4 * 4);
            *(f32x4*)(out + (size_t)idx * 4) = v;
        }
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

int aesr_launch_lerp_cat_fwd(const float* z, const float* af, const float* at, float* zcat, int B, size_t per, hipStream_t st) {
    hipLaunchKernelGGL(lerp_cat_fwd_kernel, dim3(grid_for((size_t)B * per / 4, 4096)), dim3(256), 0, st, z, af, at, zcat, B, per / 4);
    AESR_LAUNCH_CHECK("lerp_cat_fwd");
    return AESR_OK;
}

int aesr_launch_lerp_multi(const float* z, float* out, int Z, size_t per, const float* alphas, int n, float nslope, hipStream_t st) {
    LerpMultiArgs al;
    al.n = n;
    for (int k = 0; k < 16; ++k) al.a[k] = k < n ? alphas[k] : 0.f;
    hipLaunchKernelGGL(lerp_multi_kernel, dim3(grid_for((size_t)(Z - 1) * per / 4, 4096)), dim3(256), 0, st, z, out, Z - 1, per / 4, al, nslope);
    AESR_LAUNCH_CHECK("lerp_multi");
    return AESR_OK;
}

int aesr_launch_interleave_clamp(const float* orig, const float* synth, float* out, int Z, int n, size_t per, float lo, float hi, hipStream_t st) {
    const size_t total4 = ((size_t)(Z - 1) * (n + 1) + 1) * (per / 4);
    hipLaunchKernelGGL(interleave_clamp_kernel, dim3(grid_for(total4, 4096)), dim3(256), 0, st, orig, synth, out, Z, n, per / 4, lo, hi);
    AESR_LAUNCH_CHECK("interleave_clamp");
    return AESR_OK;
}

int aesr_launch_lerp_cat_bwd(const float* g, const float* af, const float* at, float* dz, int B, size_t per, hipStream_t st) {
    hipLaunchKernelGGL(lerp_cat_bwd_kernel, dim3(grid_for((size_t)B * per / 4, 4096)), dim3(256), 0, st, g, af, at, dz, B, per / 4);
    AESR_LAUNCH_CHECK("lerp_cat_bwd");
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

int aesr_launch_mse3_fwd(const float* const* a, const float* const* b, const size_t* n, const float* lam, double* ws, float* out,
                         hipStream_t st) {
    Mse3Args p;
    for (int k = 0; k < 3; ++k) { p.a[k] = a[k]; p.b[k] = b[k]; p.n[k] = n[k]; }
    p.lam = lam; p.ws = ws; p.out = out;
    hipLaunchKernelGGL(mse3_fwd_kernel, dim3(AESR_MSE3_NPART), dim3(256), 0, st, p);
    AESR_LAUNCH_CHECK("mse3_fwd");
    return AESR_OK;
}

int aesr_launch_mse3_bwd(const float* a1, const float* b1, size_t n1, const float* a2, const float* b2, size_t n2, const float* lam,
                         const float* g, float* d1, float* d2, hipStream_t st) {
    hipLaunchKernelGGL(mse3_bwd_kernel, dim3(grid_for((n1 + n2) / 4 + 1, 2048)), dim3(256), 0, st, a1, b1, n1, a2, b2, n2, lam, g, d1, d2,
                       (float)(2.0 / (double)n1), (float)(2.0 / (double)n2));
    AESR_LAUNCH_CHECK("mse3_bwd");
    return AESR_OK;
}

int aesr_launch_act_bwd(const float* dout, const float* y, float* dpre, size_t n, int act, float slope, hipStream_t st) {
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, st, dout, y, dpre, n, act, slope);
    AESR_LAUNCH_CHECK("act_bwd");
    return AESR_OK;
}

int aesr_launch_adam(float* p, float* g, float* m, float* v, float* state, size_t n, float lr, double beta1,
                     double beta2, float eps, float wd, int zero_g, hipStream_t st) {
    // <= 128 workgroups: the tickets of the step counter are atomics on ONE word (~12 ns each, serialized): 1 700 of them took 20 us
    hipLaunchKernelGGL(adam_step_kernel, dim3(grid_for(n, 128)), dim3(256), 0, st, p, g, m, v, state, n, lr, beta1, beta2, eps, wd, zero_g);
    AESR_LAUNCH_CHECK("adam_step");
    return AESR_OK;
}

int aesr_launch_s2d(const float* x, float* out, int N, int H, int W, int C, int inverse, hipStream_t st) {
    const size_t total = inverse ? (size_t)N * H * W * (C / 4) : (size_t)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, st, x, out, N, H, W, C, inverse);
    AESR_LAUNCH_CHECK("s2d");
    return AESR_OK;
}
