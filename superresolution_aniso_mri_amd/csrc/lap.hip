// Laplacian-pyramid loss pieces (reference kwatsch/lap_pyramid_loss.py) on single-channel planes [P][H][W], P = N*C:
//   blur5      out = add + gain * G(in): 5x5 binomial filter (1 4 6 4 1)^T(1 4 6 4 1)/256 with REFLECT padding (:37-40), or the
//              transposed operator (adjoint = 1), which is what the gradient of G needs (reflect padding makes G non-symmetric
//              in the two border rows/columns)
//   down2      out[y][x] = in[2y][2x]                                (:23-24)
//   zero_ins2  out = 0 except out[2y][2x] = in[y][x]                 (:27-34, the tensor fed to 4*G in upsample())
//   l1         mean |a - b| (F.l1_loss) and its gradient sign(a-b) * g / n
// All HBM-bound, one thread per output element; used by superresolution_aniso_mri_amd/kwatsch/lap_pyramid_loss.py.
#include "aesr_kernels.h"

__device__ __forceinline__ float lap_k(int d) {      // (1 4 6 4 1) / 16
    return d == 2 ? 0.375f : ((d == 1 || d == 3) ? 0.25f : 0.0625f);
}

__device__ __forceinline__ int lap_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// coefficient of in[p] in (G in)[y] along one axis of length n:  sum of k[d] over taps d with reflect(y + d - 2) == p
__device__ __forceinline__ float lap_w(int p, int y, int n) {
    float w = 0.f;
    const int d0 = p - y + 2;
    if (d0 >= 0 && d0 <= 4) w += lap_k(d0);
    const int d1 = -p - y + 2;                       // tap landed at index -p < 0
    if (p > 0 && d1 >= 0 && d1 <= 4) w += lap_k(d1);
    const int d2 = 2 * (n - 1) - p - y + 2;          // tap landed at index 2(n-1)-p >= n
    if (p <= n - 2 && d2 >= 0 && d2 <= 4) w += lap_k(d2);
    return w;
}

__global__ __launch_bounds__(256) void lap_blur5_kernel(const float* __restrict__ in, const float* __restrict__ add,
                                                        float* __restrict__ out, int P, int H, int W, float gain, int adjoint) {
    const size_t total = (size_t)P * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = idx % W;
        const size_t r = idx / W;
        const int y = r % H;
        const float* plane = in + (r / H) * (size_t)H * W;
        float s = 0.f;
        if (!adjoint) {
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                const float* row = plane + (size_t)lap_reflect(y + dy - 2, H) * W;
                float t = 0.f;
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) t += lap_k(dx) * row[lap_reflect(x + dx - 2, W)];
                s += lap_k(dy) * t;
            }
        } else {
            for (int yy = max(0, y - 2); yy <= min(H - 1, y + 2); ++yy) {
                const float wy = lap_w(y, yy, H);
                const float* row = plane + (size_t)yy * W;
                float t = 0.f;
                for (int xx = max(0, x - 2); xx <= min(W - 1, x + 2); ++xx) t += lap_w(x, xx, W) * row[xx];
                s += wy * t;
            }
        }
        out[idx] = (add ? add[idx] : 0.f) + gain * s;
    }
}

__global__ __launch_bounds__(256) void lap_down2_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int H, int W,
                                                        int h, int w) {
    const size_t total = (size_t)P * h * w;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = idx % w;
        const size_t r = idx / w;
        const int y = r % h;
        out[idx] = in[((r / h) * H + 2 * y) * (size_t)W + 2 * x];
    }
}

__global__ __launch_bounds__(256) void lap_zero_insert2_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int h,
                                                               int w, int H, int W) {
    const size_t total = (size_t)P * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = idx % W;
        const size_t r = idx / W;
        const int y = r % H;
        const bool src = !(x & 1) && !(y & 1) && (y >> 1) < h && (x >> 1) < w;
        out[idx] = src ? in[((r / H) * h + (y >> 1)) * (size_t)w + (x >> 1)] : 0.f;
    }
}

__global__ __launch_bounds__(256) void absdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              double* __restrict__ partial, size_t n) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += (double)fabsf(a[i] - b[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void l1_finalize_kernel(const double* __restrict__ partial, int np, double inv_n, float* __restrict__ out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 64) s += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (threadIdx.x == 0) *out = (float)(s * inv_n);
}

// da = sign(a - b) * g / n   (torch: the subgradient at 0 is 0)
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                                     float* __restrict__ da, size_t n, float inv_n) {
    const float s = g[0] * inv_n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        da[i] = d > 0.f ? s : (d < 0.f ? -s : 0.f);
    }
}

// ---- per-row mean (the Discriminator head of networks/acai_vanilla.py:146-150: x.reshape(N,-1).mean(-1)) ----
__global__ __launch_bounds__(1024) void row_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, size_t M) {
    __shared__ double red[16];
    const float* row = x + (size_t)blockIdx.x * M;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < M; i += 1024) s += (double)row[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k];
        out[blockIdx.x] = (float)(t / (double)M);
    }
}

__global__ __launch_bounds__(256) void row_mean_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, size_t M, size_t total,
                                                           float inv_m) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) dx[i] = g[i / M] * inv_m;
}

static inline int lap_grid(size_t n) {
    const size_t g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g ? g : 1));
}

int aesr_launch_lap_blur5(const float* in, const float* add, float* out, int P, int H, int W, float gain, int adjoint, hipStream_t st) {
    hipLaunchKernelGGL(lap_blur5_kernel, dim3(lap_grid((size_t)P * H * W)), dim3(256), 0, st, in, add, out, P, H, W, gain, adjoint);
    AESR_LAUNCH_CHECK("lap_blur5");
    return AESR_OK;
}

int aesr_launch_lap_down2(const float* in, float* out, int P, int H, int W, hipStream_t st) {
    const int h = (H + 1) / 2, w = (W + 1) / 2;
    hipLaunchKernelGGL(lap_down2_kernel, dim3(lap_grid((size_t)P * h * w)), dim3(256), 0, st, in, out, P, H, W, h, w);
    AESR_LAUNCH_CHECK("lap_down2");
    return AESR_OK;
}

int aesr_launch_lap_zero_insert2(const float* in, float* out, int P, int h, int w, int H, int W, hipStream_t st) {
    hipLaunchKernelGGL(lap_zero_insert2_kernel, dim3(lap_grid((size_t)P * H * W)), dim3(256), 0, st, in, out, P, h, w, H, W);
    AESR_LAUNCH_CHECK("lap_zero_insert2");
    return AESR_OK;
}

int aesr_launch_l1_fwd(const float* a, const float* b, double* partial, int np, float* out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(absdiff_partial_kernel, dim3(np), dim3(256), 0, st, a, b, partial, n);
    AESR_LAUNCH_CHECK("absdiff_partial");
    hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(64), 0, st, partial, np, 1.0 / (double)n, out);
    AESR_LAUNCH_CHECK("l1_finalize");
    return AESR_OK;
}

int aesr_launch_l1_bwd(const float* a, const float* b, const float* g, float* da, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(lap_grid(n)), dim3(256), 0, st, a, b, g, da, n, (float)(1.0 / (double)n));
    AESR_LAUNCH_CHECK("l1_bwd");
    return AESR_OK;
}

int aesr_launch_row_mean_fwd(const float* x, float* out, int N, size_t M, hipStream_t st) {
    hipLaunchKernelGGL(row_mean_fwd_kernel, dim3(N), dim3(1024), 0, st, x, out, M);
    AESR_LAUNCH_CHECK("row_mean_fwd");
    return AESR_OK;
}

int aesr_launch_row_mean_bwd(const float* g, float* dx, int N, size_t M, hipStream_t st) {
    hipLaunchKernelGGL(row_mean_bwd_kernel, dim3(lap_grid((size_t)N * M)), dim3(256), 0, st, g, dx, M, (size_t)N * M, (float)(1.0 / (double)M));
    AESR_LAUNCH_CHECK("row_mean_bwd");
    return AESR_OK;
}
