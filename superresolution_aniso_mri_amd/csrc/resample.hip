// Stand-alone x2 resampling steps, NHWC fp32 (the BatchNorm-fused AvgPool / nearest forms live in bn.hip):
//   mode 1  AvgPool2d(2)                                   networks/acai_vanilla.py:59 without BatchNorm, networks/ae_standard.py:41
//   mode 2  Upsample(scale_factor=2, mode='nearest')       networks/acai_vanilla.py:92 without BatchNorm
//   mode 3  Upsample(scale_factor=2, mode='bilinear', align_corners=False)      networks/ae_standard.py:68
// forward: out[N,Ho,Wo,C] from x[N,H,W,C]; backward: dx[N,H,W,C] from gout[N,Ho,Wo,C] in gather form (no atomics, fixed
// order), optionally multiplied by the derivative of the activation that produced x (taken from x itself).
#include "aesr_kernels.h"

enum { RS_POOL = 1, RS_NEAREST = 2, RS_BILINEAR = 3 };

// PyTorch's area_pixel_compute_source_index for scale 2, align_corners=False: src = max(0, (o + 0.5) / 2 - 0.5)
__device__ __forceinline__ void bilinear_taps(int o, int n_in, int* i0, int* i1, float* lam) {
    float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    if (src < 0.f) src = 0.f;
    const int f = (int)src;
    *i0 = f;
    *i1 = f + 1 < n_in ? f + 1 : n_in - 1;
    *lam = src - (float)f;
}

// weight with which output index o reads input index i
__device__ __forceinline__ float bilinear_weight(int o, int i, int n_in) {
    int i0, i1;
    float lam;
    bilinear_taps(o, n_in, &i0, &i1, &lam);
    return (i0 == i ? 1.f - lam : 0.f) + (i1 == i ? lam : 0.f);
}

__global__ __launch_bounds__(256) void resample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W,
                                                            int C, int Ho, int Wo, int mode) {
    const int C4 = C >> 2;
    const size_t total = (size_t)N * Ho * Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = idx % C4;
        size_t pix = idx / C4;
        const int ox = pix % Wo;
        pix /= Wo;
        const int oy = pix % Ho;
        const int n = pix / Ho;
        const float* base = x + (size_t)n * H * W * C + c4 * 4;
        f32x4 v;
        if (mode == RS_POOL) {
            const float* b = base + ((size_t)(2 * oy) * W + 2 * ox) * C;
            v = ((*(const f32x4*)b + *(const f32x4*)(b + C)) + (*(const f32x4*)(b + (size_t)W * C) + *(const f32x4*)(b + (size_t)W * C + C))) * 0.25f;
        } else if (mode == RS_NEAREST) {
            v = *(const f32x4*)(base + ((size_t)(oy >> 1) * W + (ox >> 1)) * C);
        } else {
            int y0, y1, x0, x1;
            float ly, lx;
            bilinear_taps(oy, H, &y0, &y1, &ly);
            bilinear_taps(ox, W, &x0, &x1, &lx);
            const f32x4 v00 = *(const f32x4*)(base + ((size_t)y0 * W + x0) * C), v01 = *(const f32x4*)(base + ((size_t)y0 * W + x1) * C);
            const f32x4 v10 = *(const f32x4*)(base + ((size_t)y1 * W + x0) * C), v11 = *(const f32x4*)(base + ((size_t)y1 * W + x1) * C);
            // same association as ATen's upsample_bilinear2d: h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)
            v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        }
        *(f32x4*)(out + idx * 4) = v;
    }
}

__global__ __launch_bounds__(256) void resample2_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ xsave,
                                                            float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo,
                                                            int mode, int mask_act, float slope) {
    const int C4 = C >> 2;
    const size_t total = (size_t)N * H * W * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = idx % C4;
        size_t pix = idx / C4;
        const int ix = pix % W;
        pix /= W;
        const int iy = pix % H;
        const int n = pix / H;
        const float* gb = gout + (size_t)n * Ho * Wo * C + c4 * 4;
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (mode == RS_POOL) {
            const int py = iy >> 1, px = ix >> 1;
            if (py < Ho && px < Wo) g = *(const f32x4*)(gb + ((size_t)py * Wo + px) * C) * 0.25f;
        } else if (mode == RS_NEAREST) {
            const float* b = gb + ((size_t)(2 * iy) * Wo + 2 * ix) * C;
            g = (*(const f32x4*)b + *(const f32x4*)(b + C)) + (*(const f32x4*)(b + (size_t)Wo * C) + *(const f32x4*)(b + (size_t)Wo * C + C));
        } else {
            for (int oy = 2 * iy - 1; oy <= 2 * iy + 2; ++oy) {
                if (oy < 0 || oy >= Ho) continue;
                const float wy = bilinear_weight(oy, iy, H);
                if (wy == 0.f) continue;
                f32x4 row = {0.f, 0.f, 0.f, 0.f};
                for (int ox = 2 * ix - 1; ox <= 2 * ix + 2; ++ox) {
                    if (ox < 0 || ox >= Wo) continue;
                    const float wx = bilinear_weight(ox, ix, W);
                    if (wx != 0.f) row += wx * *(const f32x4*)(gb + ((size_t)oy * Wo + ox) * C);
                }
                g += wy * row;
            }
        }
        if (xsave && mask_act != ACT_NONE) {
            const f32x4 xs = *(const f32x4*)(xsave + idx * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_output(xs[e], mask_act, slope);
        }
        *(f32x4*)(dx + idx * 4) = g;
    }
}

int aesr_launch_resample2(const float* x, const float* gout, const float* xsave, float* dst, int N, int H, int W, int C, int mode,
                          int backward, int mask_act, float slope, hipStream_t st) {
    const int Ho = mode == RS_POOL ? H / 2 : 2 * H, Wo = mode == RS_POOL ? W / 2 : 2 * W;
    const size_t total = (size_t)N * (backward ? H * W : Ho * Wo) * (C / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (!backward)
        hipLaunchKernelGGL(resample2_fwd_kernel, dim3(grid), dim3(256), 0, st, x, dst, N, H, W, C, Ho, Wo, mode);
    else
        hipLaunchKernelGGL(resample2_bwd_kernel, dim3(grid), dim3(256), 0, st, gout, xsave, dst, N, H, W, C, Ho, Wo, mode, mask_act, slope);
    AESR_LAUNCH_CHECK("resample2");
    return AESR_OK;
}
