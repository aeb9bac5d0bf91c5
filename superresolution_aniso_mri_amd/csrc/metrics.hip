// Image-quality metrics of the validation / model-selection loop on the device (evaluate/metrics.py:111-194 of the
// reference calls skimage's structural_similarity and peak_signal_noise_ratio slice by slice on the host):
//   per slice z of two volumes a, b [Z][H][W]:
//     ssim[z] = mean over the (H-win+1) x (W-win+1) windows of
//               ((2 ux uy + C1)(2 vxy + C2)) / ((ux^2 + uy^2 + C1)(vx + vy + C2)),   uniform win x win window,
//               sample covariance (n/(n-1)), C1 = (k1 R)^2, C2 = (k2 R)^2            [skimage defaults, R = data_range]
//     mse[z]  = mean (a - b)^2                                                        [psnr = 10 log10(R^2 / mse) on the host]
// All arithmetic in fp64 (the variance is a difference of nearly equal sums).  One 16x16 tile of window origins per
// workgroup, the (16+win-1)^2 patch of both images in LDS; partial sums per tile, then a fixed-order reduction per slice.
#include "aesr_kernels.h"

#define MT 16
#define MWIN_MAX 11

__global__ __launch_bounds__(256) void ssim_mse_tile_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            double* __restrict__ partial, int H, int W, int win, double c1,
                                                            double c2, int tiles_x, int tpi) {
    __shared__ float pa[MT + MWIN_MAX - 1][MT + MWIN_MAX], pb[MT + MWIN_MAX - 1][MT + MWIN_MAX];
    __shared__ double red[2][256];
    const int z = blockIdx.y, tile = blockIdx.x;
    const int ty0 = (tile / tiles_x) * MT, tx0 = (tile % tiles_x) * MT;
    const int P = MT + win - 1;
    const float* az = a + (size_t)z * H * W;
    const float* bz = b + (size_t)z * H * W;
    for (int q = threadIdx.x; q < P * P; q += 256) {
        const int r = q / P, c = q - r * P;
        const int y = ty0 + r, x = tx0 + c;
        const bool in = y < H && x < W;
        pa[r][c] = in ? az[(size_t)y * W + x] : 0.f;
        pb[r][c] = in ? bz[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / MT, lx = threadIdx.x % MT;
    const int y = ty0 + ly, x = tx0 + lx;
    double s_ssim = 0.0, s_sq = 0.0;
    if (y < H && x < W) {
        const double d = (double)pa[ly][lx] - (double)pb[ly][lx];
        s_sq = d * d;
    }
    if (y + win <= H && x + win <= W) {
        double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
        for (int r = 0; r < win; ++r)
            for (int c = 0; c < win; ++c) {
                const double u = (double)pa[ly + r][lx + c], v = (double)pb[ly + r][lx + c];
                sx += u;
                sy += v;
                sxx += u * u;
                syy += v * v;
                sxy += u * v;
            }
        const double n = (double)(win * win), cov = n / (n - 1.0);
        const double ux = sx / n, uy = sy / n;
        const double vx = cov * (sxx / n - ux * ux), vy = cov * (syy / n - uy * uy), vxy = cov * (sxy / n - ux * uy);
        s_ssim = ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
    }
    red[0][threadIdx.x] = s_ssim;
    red[1][threadIdx.x] = s_sq;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            red[0][threadIdx.x] += red[0][threadIdx.x + h];
            red[1][threadIdx.x] += red[1][threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[((size_t)z * tpi + tile) * 2 + 0] = red[0][0];
        partial[((size_t)z * tpi + tile) * 2 + 1] = red[1][0];
    }
}

__global__ __launch_bounds__(256) void ssim_mse_finish_kernel(const double* __restrict__ partial, double* __restrict__ ssim,
                                                              double* __restrict__ mse, int tpi, double nwin, double npix) {
    __shared__ double red[2][256];
    const int z = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int t = threadIdx.x; t < tpi; t += 256) {
        s0 += partial[((size_t)z * tpi + t) * 2 + 0];
        s1 += partial[((size_t)z * tpi + t) * 2 + 1];
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            red[0][threadIdx.x] += red[0][threadIdx.x + h];
            red[1][threadIdx.x] += red[1][threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ssim[z] = red[0][0] / nwin;
        mse[z] = red[1][0] / npix;
    }
}

int aesr_launch_ssim_mse(const float* a, const float* b, double* partial, double* ssim, double* mse, int Z, int H, int W, int win,
                         double data_range, double k1, double k2, hipStream_t st) {
    const int tiles_y = ceil_div(H, MT), tiles_x = ceil_div(W, MT), tpi = tiles_y * tiles_x;
    const double c1 = (k1 * data_range) * (k1 * data_range), c2 = (k2 * data_range) * (k2 * data_range);
    hipLaunchKernelGGL(ssim_mse_tile_kernel, dim3(tpi, Z), dim3(256), 0, st, a, b, partial, H, W, win, c1, c2, tiles_x, tpi);
    AESR_LAUNCH_CHECK("ssim_mse_tile");
    hipLaunchKernelGGL(ssim_mse_finish_kernel, dim3(Z), dim3(256), 0, st, partial, ssim, mse, tpi,
                       (double)(H - win + 1) * (double)(W - win + 1), (double)H * (double)W);
    AESR_LAUNCH_CHECK("ssim_mse_finish");
    return AESR_OK;
}
