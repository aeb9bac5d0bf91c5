// LPIPS-VGG head and the VGG max-pool, NHWC fp32.
//
//   maxpool2 fwd / bwd   nn.MaxPool2d(2) of lpips/pretrained_networks.py:107-116 (torchvision vgg16.features);
//                        backward scatters to the FIRST maximum of each window (ATen tie rule), adds the tap
//                        gradient of the same layer and applies the ReLU mask of the producing conv in one pass
//   lpips_tap fwd / bwd  lpips/networks_basic.py:69-86 for one tap: unit-normalise both branches over channels
//                        (lpips/common.py:12-14, eps OUTSIDE the sqrt), squared difference, 1x1 "lin" weights
//                        (no bias, dropout inactive), spatial mean -- in ONE pass over both feature maps; one
//                        wave per pixel, channels spread over the 64 lanes.  Only branch 0 (the synthesised slice)
//                        gets a gradient; branch 1 (the reference slice) is constant.
//   lpips_finalize       d[n] = sum_k mean_hw(...)  (L_weights == 1, lpips/networks_basic.py:36,84-86)
#include "aesr_kernels.h"

constexpr int LP_NCH = 64;   // pixel chunks per image of the tap partial sums

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H,
                                                           int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const int total = N * Ho * Wo * C4;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int c4 = idx % C4;
        int pix = idx / C4;
        const int xo = pix % Wo;
        pix /= Wo;
        const int yo = pix % Ho;
        const int n = pix / Ho;
        const float* b = x + ((size_t)(n * H + 2 * yo) * W + 2 * xo) * C + c4 * 4;
        const f32x4 v00 = *(const f32x4*)b, v01 = *(const f32x4*)(b + C);
        const f32x4 v10 = *(const f32x4*)(b + (size_t)W * C), v11 = *(const f32x4*)(b + (size_t)W * C + C);
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
        *(f32x4*)(out + (size_t)idx * 4) = m;
    }
}

// dx[n,y,x,c] = ( [this is the first max of its window] * gout[n,y/2,x/2,c] + gadd[n,y,x,c] ) * (relu ? x>0 : 1)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                           const float* __restrict__ gadd, float* __restrict__ dx, int N, int H,
                                                           int W, int C, int relu) {
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const int total = N * H * W * C4;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int c4 = idx % C4;
        int pix = idx / C4;
        const int xx = pix % W;
        pix /= W;
        const int yy = pix % H;
        const int n = pix / H;
        const f32x4 xv = *(const f32x4*)(x + (size_t)idx * 4);
        f32x4 g = gadd ? *(const f32x4*)(gadd + (size_t)idx * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        // branch-free: the window of an unpooled last row / column (odd H or W) is clamped onto a valid one and its contribution
        // zeroed, so the five loads are issued together (a bounds branch in front of them serialises the loads)
        const bool pooled = (yy >> 1) < Ho && (xx >> 1) < Wo;
        const int yo = min(yy >> 1, Ho - 1), xo = min(xx >> 1, Wo - 1);
        {
            const float* b = x + ((size_t)(n * H + 2 * yo) * W + 2 * xo) * C + c4 * 4;
            const f32x4 v00 = *(const f32x4*)b, v01 = *(const f32x4*)(b + C);
            const f32x4 v10 = *(const f32x4*)(b + (size_t)W * C), v11 = *(const f32x4*)(b + (size_t)W * C + C);
            const f32x4 go = *(const f32x4*)(gout + ((size_t)(n * Ho + yo) * Wo + xo) * C + c4 * 4);
            const int me = pooled ? (yy & 1) * 2 + (xx & 1) : -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // first maximum in scan order (00, 01, 10, 11)
                int am = 0;
                float mv = v00[e];
                if (v01[e] > mv) { mv = v01[e]; am = 1; }
                if (v10[e] > mv) { mv = v10[e]; am = 2; }
                if (v11[e] > mv) { mv = v11[e]; am = 3; }
                if (am == me) g[e] += go[e];
            }
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = xv[e] > 0.f ? g[e] : 0.f;
        }
        *(f32x4*)(dx + (size_t)idx * 4) = g;
    }
}

// one wave per pixel; lane l owns channels l*CPL .. l*CPL+CPL-1
template <int CPL>
__device__ __forceinline__ void load_cpl(const float* p, int lane, float (&v)[CPL]) {
    if constexpr (CPL == 1) {
        v[0] = p[lane];
    } else if constexpr (CPL == 2) {
        const float2 t = *(const float2*)(p + lane * 2);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int k = 0; k < CPL / 4; ++k) {
            const f32x4 t = *(const f32x4*)(p + lane * CPL + k * 4);
            v[k * 4 + 0] = t[0]; v[k * 4 + 1] = t[1]; v[k * 4 + 2] = t[2]; v[k * 4 + 3] = t[3];
        }
    }
}

template <int CPL>
__device__ __forceinline__ void store_cpl(float* p, int lane, const float (&v)[CPL]) {
    if constexpr (CPL == 1) {
        p[lane] = v[0];
    } else if constexpr (CPL == 2) {
        *(float2*)(p + lane * 2) = make_float2(v[0], v[1]);
    } else {
#pragma unroll
        for (int k = 0; k < CPL / 4; ++k) *(f32x4*)(p + lane * CPL + k * 4) = (f32x4){v[k * 4], v[k * 4 + 1], v[k * 4 + 2], v[k * 4 + 3]};
    }
}

__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// partial[n][chunk] = sum over the chunk's pixels of sum_c w_c (f0hat_c - f1hat_c)^2 ; grid = (LP_NCH, B)
template <int CPL>
__global__ __launch_bounds__(256) void lpips_tap_fwd_kernel(const float* __restrict__ f, const float* __restrict__ lin,
                                                            float* __restrict__ partial, int B, int HW) {
    constexpr int C = CPL * 64;
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y;
    float w[CPL];
    load_cpl<CPL>(lin, lane, w);
    float acc = 0.f;
    for (int p = blockIdx.x * 4 + wave; p < HW; p += LP_NCH * 4) {
        float a[CPL], b[CPL];
        load_cpl<CPL>(f + ((size_t)n * HW + p) * C, lane, a);
        load_cpl<CPL>(f + ((size_t)(n + B) * HW + p) * C, lane, b);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) { sa = fmaf(a[k], a[k], sa); sb = fmaf(b[k], b[k], sb); }
        sa = wave_allsum(sa);
        sb = wave_allsum(sb);
        const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const float d = a[k] * ia - b[k] * ib;
            r = fmaf(w[k] * d, d, r);
        }
        acc += wave_allsum(r);
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[n * LP_NCH + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Narrow taps (C = 64, 128): LPP = C/4 lanes x float4 own one pixel, a wave works on 64/LPP pixels at once -- the wave-wide
// form above spends most of its time in 6-step butterflies for a 64-channel pixel.
template <int LPP>
__device__ __forceinline__ float group_allsum(float v) {
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LPP>
__global__ __launch_bounds__(256) void lpips_tap_fwd_sub_kernel(const float* __restrict__ f, const float* __restrict__ lin,
                                                                float* __restrict__ partial, int B, int HW) {
    constexpr int C = LPP * 4, PPW = 64 / LPP;
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, pg = lane / LPP;
    const int n = blockIdx.y;
    const f32x4 w = *(const f32x4*)(lin + sub * 4);
    float acc = 0.f;
    for (int p0 = (blockIdx.x * 4 + wave) * PPW; p0 < HW; p0 += LP_NCH * 4 * PPW) {
        const int p = p0 + pg;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (p < HW) {
            a = *(const f32x4*)(f + ((size_t)n * HW + p) * C + sub * 4);
            b = *(const f32x4*)(f + ((size_t)(n + B) * HW + p) * C + sub * 4);
        }
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { sa = fmaf(a[k], a[k], sa); sb = fmaf(b[k], b[k], sb); }
        sa = group_allsum<LPP>(sa);
        sb = group_allsum<LPP>(sb);
        const float ia = 1.f / (sqrtf(sa) + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = a[k] * ia - b[k] * ib;
            acc = fmaf(w[k] * d, d, acc);
        }
    }
    acc = wave_allsum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[n * LP_NCH + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int LPP>
__global__ __launch_bounds__(256) void lpips_tap_bwd_sub_kernel(const float* __restrict__ f, const float* __restrict__ lin,
                                                                const float* __restrict__ gd, float* __restrict__ gf0, int B,
                                                                int HW, float inv_hw) {
    constexpr int C = LPP * 4, PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, pg = lane / LPP;
    const int n = blockIdx.y;
    const f32x4 w = *(const f32x4*)(lin + sub * 4);
    const float up = gd[n] * inv_hw;
    for (int p0 = (blockIdx.x * 4 + wave) * PPW; p0 < HW; p0 += gridDim.x * 4 * PPW) {
        const int p = p0 + pg;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (p < HW) {
            a = *(const f32x4*)(f + ((size_t)n * HW + p) * C + sub * 4);
            b = *(const f32x4*)(f + ((size_t)(n + B) * HW + p) * C + sub * 4);
        }
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { sa = fmaf(a[k], a[k], sa); sb = fmaf(b[k], b[k], sb); }
        sa = group_allsum<LPP>(sa);
        sb = group_allsum<LPP>(sb);
        const float n0 = sqrtf(sa);
        const float ia = 1.f / (n0 + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
        f32x4 o;
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = a[k] * ia - b[k] * ib;
            o[k] = 2.f * w[k] * d;
            S = fmaf(o[k], a[k], S);
        }
        S = group_allsum<LPP>(S);
        const float kn = n0 > 0.f ? S * ia * ia / n0 : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = up * (o[k] * ia - a[k] * kn);
        if (p < HW) *(f32x4*)(gf0 + ((size_t)n * HW + p) * C + sub * 4) = o;
    }
}

// gf0[n,p,c] = gd[n]*inv_hw * ( 2 w_c D_c / (n0+eps) - f0_c * S / (n0 (n0+eps)^2) ),  S = sum_c 2 w_c D_c f0_c
// (an all-zero feature vector gets the finite limit 2 w_c D_c/(eps) -> its norm term is dropped, where autograd of the
//  reference would produce NaN)
template <int CPL>
__global__ __launch_bounds__(256) void lpips_tap_bwd_kernel(const float* __restrict__ f, const float* __restrict__ lin,
                                                            const float* __restrict__ gd, float* __restrict__ gf0, int B, int HW,
                                                            float inv_hw) {
    constexpr int C = CPL * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y;
    float w[CPL];
    load_cpl<CPL>(lin, lane, w);
    const float up = gd[n] * inv_hw;
    for (int p = blockIdx.x * 4 + wave; p < HW; p += gridDim.x * 4) {
        float a[CPL], b[CPL], o[CPL];
        load_cpl<CPL>(f + ((size_t)n * HW + p) * C, lane, a);
        load_cpl<CPL>(f + ((size_t)(n + B) * HW + p) * C, lane, b);
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) { sa = fmaf(a[k], a[k], sa); sb = fmaf(b[k], b[k], sb); }
        sa = wave_allsum(sa);
        sb = wave_allsum(sb);
        const float n0 = sqrtf(sa);
        const float ia = 1.f / (n0 + 1e-10f), ib = 1.f / (sqrtf(sb) + 1e-10f);
        float S = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            const float d = a[k] * ia - b[k] * ib;
            o[k] = 2.f * w[k] * d;
            S = fmaf(o[k], a[k], S);
        }
        S = wave_allsum(S);
        const float kn = n0 > 0.f ? S * ia * ia / n0 : 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = up * (o[k] * ia - a[k] * kn);
        store_cpl<CPL>(gf0 + ((size_t)n * HW + p) * C, lane, o);
    }
}

// ScalingLayer (+ optional 2x-1) materialised as a 4-channel image so that VGG conv1_1 runs on the MFMA kernel:
// out[p] = (ca0*x+cb0, ca1*x+cb1, ca2*x+cb2, 0);  backward: dx[p] = sum_c ca_c * d4[p][c]
struct ScaleArgs { float ca[3], cb[3]; };

__global__ __launch_bounds__(256) void scale_expand_fwd_kernel(const float* __restrict__ x, float* __restrict__ out4, int n, ScaleArgs a) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = x[i];
        *(f32x4*)(out4 + (size_t)i * 4) = (f32x4){a.ca[0] * v + a.cb[0], a.ca[1] * v + a.cb[1], a.ca[2] * v + a.cb[2], 0.f};
    }
}

__global__ __launch_bounds__(256) void scale_expand_bwd_kernel(const float* __restrict__ d4, float* __restrict__ dx, int n, ScaleArgs a) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const f32x4 d = *(const f32x4*)(d4 + (size_t)i * 4);
        dx[i] = a.ca[0] * d[0] + a.ca[1] * d[1] + a.ca[2] * d[2];
    }
}

int aesr_launch_scale_expand(const float* x, float* out4, int n, const float* ca, const float* cb, int backward, hipStream_t st) {
    ScaleArgs a;
    for (int c = 0; c < 3; ++c) { a.ca[c] = ca[c]; a.cb[c] = cb ? cb[c] : 0.f; }
    int grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (backward) hipLaunchKernelGGL(scale_expand_bwd_kernel, dim3(grid), dim3(256), 0, st, x, out4, n, a);
    else hipLaunchKernelGGL(scale_expand_fwd_kernel, dim3(grid), dim3(256), 0, st, x, out4, n, a);
    AESR_LAUNCH_CHECK("scale_expand");
    return AESR_OK;
}

struct LpFinalArgs {
    const float* partial[8];
    float scale[8];
    int ntaps;
};

__global__ void lpips_finalize_kernel(LpFinalArgs a, float* __restrict__ d, int B) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B) return;
    float tot = 0.f;
    for (int k = 0; k < a.ntaps; ++k) {
        float s = 0.f;
        for (int c = 0; c < LP_NCH; ++c) s += a.partial[k][n * LP_NCH + c];
        tot += s * a.scale[k];
    }
    d[n] = tot;
}

static inline int grid_cap(size_t n, int cap) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}

int aesr_launch_maxpool2_fwd(const float* x, float* out, int N, int H, int W, int C, hipStream_t st) {
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_cap((size_t)N * (H / 2) * (W / 2) * (C / 4), 8192)), dim3(256), 0, st, x, out, N, H, W, C);
    AESR_LAUNCH_CHECK("maxpool2_fwd");
    return AESR_OK;
}

int aesr_launch_maxpool2_bwd(const float* gout, const float* x, const float* gadd, float* dx, int N, int H, int W, int C, int relu,
                             hipStream_t st) {
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_cap((size_t)N * H * W * (C / 4), 8192)), dim3(256), 0, st, gout, x, gadd, dx, N, H, W, C, relu);
    AESR_LAUNCH_CHECK("maxpool2_bwd");
    return AESR_OK;
}

int aesr_launch_lpips_tap_fwd(const float* f, const float* lin, float* partial, int B, int HW, int C, hipStream_t st) {
    dim3 grid(LP_NCH, B);
    switch (C) {
        case 64: hipLaunchKernelGGL(lpips_tap_fwd_sub_kernel<16>, grid, dim3(256), 0, st, f, lin, partial, B, HW); break;
        case 128: hipLaunchKernelGGL(lpips_tap_fwd_sub_kernel<32>, grid, dim3(256), 0, st, f, lin, partial, B, HW); break;
        case 256: hipLaunchKernelGGL(lpips_tap_fwd_kernel<4>, grid, dim3(256), 0, st, f, lin, partial, B, HW); break;
        case 512: hipLaunchKernelGGL(lpips_tap_fwd_kernel<8>, grid, dim3(256), 0, st, f, lin, partial, B, HW); break;
        default: aesr_set_error("lpips_tap_fwd: unsupported channel count %d (64/128/256/512)", C); return AESR_ERR_UNSUPPORTED;
    }
    AESR_LAUNCH_CHECK("lpips_tap_fwd");
    return AESR_OK;
}

int aesr_launch_lpips_tap_bwd(const float* f, const float* lin, const float* gd, float* gf0, int B, int HW, int C, hipStream_t st) {
    int gx = (HW + 3) / 4;
    if (gx > 256) gx = 256;
    dim3 grid(gx, B);
    const float inv = 1.f / (float)HW;
    int gs = (HW + 15) / 16;                 // narrow taps: 4 (C=64) / 2 (C=128) pixels per wave
    if (gs > 256) gs = 256;
    switch (C) {
        case 64: hipLaunchKernelGGL(lpips_tap_bwd_sub_kernel<16>, dim3(gs, B), dim3(256), 0, st, f, lin, gd, gf0, B, HW, inv); break;
        case 128: hipLaunchKernelGGL(lpips_tap_bwd_sub_kernel<32>, dim3(gs, B), dim3(256), 0, st, f, lin, gd, gf0, B, HW, inv); break;
        case 256: hipLaunchKernelGGL(lpips_tap_bwd_kernel<4>, grid, dim3(256), 0, st, f, lin, gd, gf0, B, HW, inv); break;
        case 512: hipLaunchKernelGGL(lpips_tap_bwd_kernel<8>, grid, dim3(256), 0, st, f, lin, gd, gf0, B, HW, inv); break;
        default: aesr_set_error("lpips_tap_bwd: unsupported channel count %d (64/128/256/512)", C); return AESR_ERR_UNSUPPORTED;
    }
    AESR_LAUNCH_CHECK("lpips_tap_bwd");
    return AESR_OK;
}

int aesr_launch_lpips_finalize(const float* const* partials, const int* hw, int ntaps, float* d, int B, hipStream_t st) {
    LpFinalArgs a;
    a.ntaps = ntaps;
    for (int k = 0; k < ntaps; ++k) { a.partial[k] = partials[k]; a.scale[k] = 1.f / (float)hw[k]; }
    hipLaunchKernelGGL(lpips_finalize_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, a, d, B);
    AESR_LAUNCH_CHECK("lpips_finalize");
    return AESR_OK;
}
