// Bandwidth-bound convolutions that do not belong on the matrix cores (K or N of the GEMM <= 4), NHWC:
//   * conv_smallcin_fwd    Cin <= 4 (stem 1x1 pad 1 of networks/acai_vanilla.py:51, VGG conv1_1 of
//                          lpips/pretrained_networks.py:107 with the ScalingLayer of
//                          lpips/networks_basic.py:93-100 folded into the loader, and the data-gradient of
//                          the 32->1 output conv networks/acai_vanilla.py:98 via the flip/transpose flag)
//   * conv_smallcin_dgrad  data gradient towards <= 4 input channels (VGG conv1_1 -> image, stem -> image)
//   * conv_smallcin_wgrad  weight/bias gradient of a 1x1 small-Cin conv (the stem)
//   * conv_cout1_wgrad     weight/bias gradient of the Cout == 1 output conv
// plus the fixed-order partial-sum reducer they share.
#include "aesr_kernels.h"


__global__ __launch_bounds__(256) void conv_smallcin_fwd_kernel(SmallArgs a) {
    const size_t total = (size_t)a.N * a.Ho * a.Wo * a.Cout;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int co = idx % a.Cout;
        size_t pix = idx / a.Cout;
        const int x = pix % a.Wo;
        pix /= a.Wo;
        const int y = pix % a.Ho;
        const int n = pix / a.Ho;
        float s = a.bias ? a.bias[co] : 0.f;
        const int cmem = a.bcast ? 1 : a.Cin;
        for (int ky = 0; ky < a.KS; ++ky) {
            const int gy = y + ky - a.pad;
            if (gy < 0 || gy >= a.H) continue;
            for (int kx = 0; kx < a.KS; ++kx) {
                const int gx = x + kx - a.pad;
                if (gx < 0 || gx >= a.W) continue;
                const float* ip = a.in + (((size_t)n * a.H + gy) * a.W + gx) * cmem;
                for (int ci = 0; ci < a.Cin; ++ci) {
                    const float v = a.bcast ? a.ca[ci] * ip[0] + a.cb[ci] : ip[ci];
                    const float wv = a.transpose
                                         ? a.w[(((size_t)ci * a.Cout + co) * a.KS + (a.KS - 1 - ky)) * a.KS + (a.KS - 1 - kx)]
                                         : a.w[(((size_t)co * a.Cin + ci) * a.KS + ky) * a.KS + kx];
                    s = fmaf(v, wv, s);
                }
            }
        }
        s = act_apply(s, a.act, a.slope);
        if (a.ysave) s *= act_grad_from_output(a.ysave[idx], a.mask_act, a.slope);
        a.out[idx] = s;
    }
}

// Same op, one workgroup per output row (n, y); a thread owns one cout quad (256 % (Cout/4) == 0) and walks the row:
// no per-element integer division, 16-byte stores, the quad's filter taps cached in registers when Cin*KS*KS <= 9.
__global__ __launch_bounds__(256) void conv_smallcin_fwd4_kernel(SmallArgs a) {
    const int C4 = a.Cout >> 2;
    const int row = blockIdx.x;
    const int n = row / a.Ho, y = row - n * a.Ho;
    const int c4 = threadIdx.x % C4, xl = threadIdx.x / C4, XL = 256 / C4;
    const int cmem = a.bcast ? 1 : a.Cin;
    const int K = a.Cin * a.KS * a.KS;
    f32x4 wreg[9];
    const bool cached = K <= 9;
    if (cached) {
        for (int k = 0; k < 9; ++k) {
            wreg[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (k < K) {
                const int ci = k / (a.KS * a.KS), tap = k - ci * a.KS * a.KS;
                const int ky = tap / a.KS, kx = tap - ky * a.KS;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = c4 * 4 + e;
                    wreg[k][e] = a.transpose ? a.w[((ci * a.Cout + co) * a.KS + (a.KS - 1 - ky)) * a.KS + (a.KS - 1 - kx)]
                                             : a.w[((co * a.Cin + ci) * a.KS + ky) * a.KS + kx];
                }
            }
        }
    }
    const f32x4 bias = a.bias ? *(const f32x4*)(a.bias + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int x = xl; x < a.Wo; x += XL) {
        f32x4 s = bias;
        for (int ky = 0; ky < a.KS; ++ky) {
            const int gy = y + ky - a.pad;
            if (gy < 0 || gy >= a.H) continue;
            for (int kx = 0; kx < a.KS; ++kx) {
                const int gx = x + kx - a.pad;
                if (gx < 0 || gx >= a.W) continue;
                const float* ip = a.in + ((size_t)(n * a.H + gy) * a.W + gx) * cmem;
                for (int ci = 0; ci < a.Cin; ++ci) {
                    const float v = a.bcast ? a.ca[ci] * ip[0] + a.cb[ci] : ip[ci];
                    if (cached) {
                        // static register indexing: K <= 9 slots, select by comparison (wave-uniform)
                        const int k = ci * a.KS * a.KS + ky * a.KS + kx;
                        f32x4 wv = wreg[0];
#pragma unroll
                        for (int q = 1; q < 9; ++q)
                            if (k == q) wv = wreg[q];
                        s += wv * v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int co = c4 * 4 + e;
                            const float wv = a.transpose
                                                 ? a.w[((ci * a.Cout + co) * a.KS + (a.KS - 1 - ky)) * a.KS + (a.KS - 1 - kx)]
                                                 : a.w[((co * a.Cin + ci) * a.KS + ky) * a.KS + kx];
                            s[e] = fmaf(v, wv, s[e]);
                        }
                    }
                }
            }
        }
        const size_t o = ((size_t)(n * a.Ho + y) * a.Wo + x) * a.Cout + c4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = act_apply(s[e], a.act, a.slope);
        if (a.ysave) {
            const f32x4 ys = *(const f32x4*)(a.ysave + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] *= act_grad_from_output(ys[e], a.mask_act, a.slope);
        }
        *(f32x4*)(a.out + o) = s;
    }
}

// Cout == 1, 3x3 pad 1 output convolution (networks/acai_vanilla.py:98): bandwidth-bound; one thread per output pixel,
// the input patch (tile + halo, all Cin channels) staged once in LDS, weights broadcast from LDS.
__global__ __launch_bounds__(256) void conv_cout1_fwd_kernel(Cout1FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int PW = a.TW + 2, PH = a.TH + 2, PP = PH * PW, TP = a.TH * a.TW;
    const int CS = a.Cin + 4, C4 = a.Cin >> 2;
    float* ldsP = lds;               // [PP][CS]
    float* ldsW = lds + PP * CS;     // [9][Cin]
    int tile = blockIdx.x;
    const int tx = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    for (int q = threadIdx.x; q < PP * C4; q += 256) {
        const int p = q / C4, part = q - p * C4;
        const int gy = y0 + p / PW - 1, gx = x0 + p % PW - 1;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
            v = *(const f32x4*)(a.x + ((size_t)(n * a.H + gy) * a.W + gx) * a.Cin + part * 4);
        *(f32x4*)(ldsP + p * CS + part * 4) = v;
    }
    for (int q = threadIdx.x; q < 9 * a.Cin; q += 256) {
        const int tap = q / a.Cin, ci = q - tap * a.Cin;
        ldsW[q] = a.w[ci * 9 + tap];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < TP; p += 256) {
        const int r = p / a.TW, c = p - r * a.TW;
        const int y = y0 + r, x = x0 + c;
        if (y >= a.H || x >= a.W) continue;
        float s = a.bias ? a.bias[0] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* pp = ldsP + ((r + t / 3) * PW + c + t % 3) * CS;
            const float* ww = ldsW + t * a.Cin;
            for (int k = 0; k < C4; ++k) {
                const f32x4 v = *(const f32x4*)(pp + k * 4), wv = *(const f32x4*)(ww + k * 4);
                s = fmaf(v[0], wv[0], s);
                s = fmaf(v[1], wv[1], s);
                s = fmaf(v[2], wv[2], s);
                s = fmaf(v[3], wv[3], s);
            }
        }
        a.out[(size_t)(n * a.H + y) * a.W + x] = act_apply(s, a.act, a.slope);
    }
}

// dx[n,y,x,ci] = sum_{ky,kx,co} dy[n,y+pad-ky,x+pad-kx,co] * w[co,ci,ky,kx]      (forward conv stride 1)
// bcast: the forward input was 1 channel expanded by c -> ca[c]*x+cb[c]; returns dx1 = sum_c ca[c]*dx[c]

__global__ __launch_bounds__(256) void conv_smallcin_dgrad_kernel(SmallDgradArgs a) {
    // one wave-quarter (16 lanes) per output pixel: lanes split Cout, then shuffle-reduce
    const int sub = threadIdx.x & 15;
    const size_t npix = (size_t)a.N * a.H * a.W;
    for (size_t pix = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); pix < npix; pix += (size_t)gridDim.x * 16) {
        const int x = pix % a.W;
        const int y = (pix / a.W) % a.H;
        const int n = pix / ((size_t)a.W * a.H);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < a.KS; ++ky) {
            const int oy = y + a.pad - ky;
            if (oy < 0 || oy >= a.Ho) continue;
            for (int kx = 0; kx < a.KS; ++kx) {
                const int ox = x + a.pad - kx;
                if (ox < 0 || ox >= a.Wo) continue;
                const float* dp = a.dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout;
                for (int co = sub; co < a.Cout; co += 16) {
                    const float d = dp[co];
#pragma unroll
                    for (int ci = 0; ci < 4; ++ci)
                        if (ci < a.Cin) acc[ci] = fmaf(d, a.w[(((size_t)co * a.Cin + ci) * a.KS + ky) * a.KS + kx], acc[ci]);
                }
            }
        }
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            float v = acc[ci];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 1, 64);
            acc[ci] = v;
        }
        if (sub == 0) {
            if (a.bcast) {
                float s = 0.f;
                for (int ci = 0; ci < a.Cin; ++ci) s = fmaf(a.ca[ci], acc[ci], s);
                a.dx[pix] = s;
            } else {
                for (int ci = 0; ci < a.Cin; ++ci) a.dx[pix * a.Cin + ci] = acc[ci];
            }
        }
    }
}

// stem wgrad (KS == 1): partial[wg][co*Cin + k] = sum dout[.,co]*in[.,k];  partial[wg][Cout*Cin + co] = sum dout

__global__ __launch_bounds__(256) void conv_smallcin_wgrad_kernel(SmallWgradArgs a) {
    // thread = (cout quad, pixel lane); workgroups stride over output rows; no per-pixel integer division
    extern __shared__ __attribute__((aligned(16))) float red[];   // [XL][Cout*(Cin+1)]
    const int C4 = a.Cout >> 2, c4 = threadIdx.x % C4, xl = threadIdx.x / C4, XL = 256 / C4;
    f32x4 acc[4], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nrows = a.N * a.Ho;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / a.Ho, y = row - n * a.Ho;
        const int gy = y - a.pad;
        const bool yin = gy >= 0 && gy < a.H;
        const float* drow = a.dout + (size_t)row * a.Wo * a.Cout + c4 * 4;
        const float* irow = a.in + (size_t)(n * a.H + (yin ? gy : 0)) * a.W * a.Cin;
        for (int x = xl; x < a.Wo; x += XL) {
            const f32x4 d = *(const f32x4*)(drow + (size_t)x * a.Cout);
            accb += d;
            const int gx = x - a.pad;
            if (yin && gx >= 0 && gx < a.W) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < a.Cin) acc[k] += d * irow[gx * a.Cin + k];
            }
        }
    }
    const int nout = a.Cout * (a.Cin + 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int co = c4 * 4 + e;
        for (int k = 0; k < a.Cin; ++k) red[xl * nout + co * a.Cin + k] = acc[k][e];
        red[xl * nout + a.Cout * a.Cin + co] = accb[e];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nout; o += 256) {
        float t = 0.f;
        for (int p = 0; p < XL; ++p) t += red[p * nout + o];
        a.partial[(size_t)blockIdx.x * nout + o] = t;
    }
}

// Cout == 1 output conv: partial[wg][ci*KS*KS + tap] = sum_px dy[px] * x[px+tap][ci];  partial[wg][Cin*KS*KS] = sum dy

__global__ __launch_bounds__(256) void conv_cout1_wgrad_kernel(Cout1WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int PW = a.TW + 2, PH = a.TH + 2, PP = PH * PW, TP = a.TH * a.TW;
    const int CS = a.Cin + 1;            // odd-ish stride: consecutive ci of one pixel stay conflict-free
    float* ldsP = lds;                   // [PP][CS]
    float* ldsD = lds + PP * CS;         // [TP]
    const int ci = threadIdx.x % a.Cin, pl = threadIdx.x / a.Cin, PL = 256 / a.Cin;
    float acc[9], accb = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    const int tpi = a.tiles_y * a.tiles_x;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int y0 = (trem / a.tiles_x) * a.TH, x0 = (trem % a.tiles_x) * a.TW;
        const int C4 = a.Cin >> 2;
        for (int q = threadIdx.x; q < PP * C4; q += 256) {
            const int p = q / C4, part = q - p * C4;
            const int gy = y0 + p / PW - 1, gx = x0 + p % PW - 1;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *(const f32x4*)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + part * 4);
            float* d = ldsP + p * CS + part * 4;
            d[0] = v[0];
            d[1] = v[1];
            d[2] = v[2];
            d[3] = v[3];
        }
        for (int p = threadIdx.x; p < TP; p += 256) {
            const int gy = y0 + p / a.TW, gx = x0 + p % a.TW;
            ldsD[p] = (gy < a.H && gx < a.W) ? a.dy[((size_t)n * a.H + gy) * a.W + gx] : 0.f;
        }
        __syncthreads();
        for (int p = pl; p < TP; p += PL) {
            const float d = ldsD[p];
            const int r = p / a.TW, c = p - r * a.TW;
            if (ci == 0) accb += d;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = fmaf(d, ldsP[((r + t / 3) * PW + c + t % 3) * CS + ci], acc[t]);
        }
        __syncthreads();
    }
    // reduce over pl through LDS (reuse the patch area)
    const int nout = a.Cin * 9 + 1;
    float* red = lds;
#pragma unroll
    for (int t = 0; t < 9; ++t) red[pl * nout + ci * 9 + t] = acc[t];
    if (ci == 0) red[pl * nout + a.Cin * 9] = accb;
    __syncthreads();
    for (int o = threadIdx.x; o < nout; o += 256) {
        float s = 0.f;
        for (int p = 0; p < PL; ++p) s += red[p * nout + o];
        a.partial[(size_t)blockIdx.x * nout + o] = s;
    }
}

// out[o] = sum_p partial[p][o]  (fixed order; double accumulate), optional second output split at n0
// block = 64 columns x 16 row-lanes; every row-lane keeps 4 independent loads in flight
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ partial, int np, int n,
                                                            float* __restrict__ out0, int n0, float* __restrict__ out1) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + col;
    double s = 0.0;
    if (o < n) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int p = rl;
        for (; p + 48 < np; p += 64) {
            const float v0 = partial[(size_t)p * n + o], v1 = partial[(size_t)(p + 16) * n + o];
            const float v2 = partial[(size_t)(p + 32) * n + o], v3 = partial[(size_t)(p + 48) * n + o];
            s0 += (double)v0;
            s1 += (double)v1;
            s2 += (double)v2;
            s3 += (double)v3;
        }
        for (; p < np; p += 16) s0 += (double)partial[(size_t)p * n + o];
        s = (s0 + s1) + (s2 + s3);
    }
    red[rl][col] = s;
    __syncthreads();
    if (rl == 0 && o < n) {
        s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][col];
        if (o < n0) out0[o] = (float)s;
        else out1[o - n0] = (float)s;
    }
}

int aesr_launch_cout1_fwd(const Cout1FwdArgs& a, hipStream_t st) {
    const size_t shmem = ((size_t)(a.TH + 2) * (a.TW + 2) * (a.Cin + 4) + 9 * a.Cin) * sizeof(float);
    hipLaunchKernelGGL(conv_cout1_fwd_kernel, dim3(a.N * a.tiles_y * a.tiles_x), dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_cout1_fwd");
    return AESR_OK;
}

int aesr_launch_smallcin_fwd(const SmallArgs& a, hipStream_t st) {
    const size_t total = (size_t)a.N * a.Ho * a.Wo * a.Cout;
    if (a.Cout % 4 == 0 && 256 % (a.Cout / 4) == 0 && total < (size_t)1 << 31) {
        hipLaunchKernelGGL(conv_smallcin_fwd4_kernel, dim3(a.N * a.Ho), dim3(256), 0, st, a);
        AESR_LAUNCH_CHECK("conv_smallcin_fwd4");
        return AESR_OK;
    }
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(conv_smallcin_fwd_kernel, dim3(grid), dim3(256), 0, st, a);
    AESR_LAUNCH_CHECK("conv_smallcin_fwd");
    return AESR_OK;
}

int aesr_launch_smallcin_dgrad(const SmallDgradArgs& a, hipStream_t st) {
    const size_t npix = (size_t)a.N * a.H * a.W;
    int grid = (int)((npix + 15) / 16);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(conv_smallcin_dgrad_kernel, dim3(grid), dim3(256), 0, st, a);
    AESR_LAUNCH_CHECK("conv_smallcin_dgrad");
    return AESR_OK;
}

int aesr_launch_sum_partials(const float* partial, int np, int n, float* out0, int n0, float* out1, hipStream_t st) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(n, 64)), dim3(1024), 0, st, partial, np, n, out0, n0, out1);
    AESR_LAUNCH_CHECK("sum_partials");
    return AESR_OK;
}

int aesr_launch_smallcin_wgrad(const SmallWgradArgs& a, int nwg, hipStream_t st) {
    const int XL = 256 / (a.Cout / 4);
    const size_t shmem = (size_t)XL * a.Cout * (a.Cin + 1) * sizeof(float);
    hipLaunchKernelGGL(conv_smallcin_wgrad_kernel, dim3(nwg), dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_smallcin_wgrad");
    return AESR_OK;
}

int aesr_launch_cout1_wgrad(const Cout1WgradArgs& a, int nwg, hipStream_t st) {
    const int PP = (a.TH + 2) * (a.TW + 2), TP = a.TH * a.TW, PL = 256 / a.Cin;
    size_t fl = (size_t)PP * (a.Cin + 1) + TP;
    const size_t redfl = (size_t)PL * (a.Cin * 9 + 1);
    if (redfl > fl) fl = redfl;
    hipLaunchKernelGGL(conv_cout1_wgrad_kernel, dim3(nwg), dim3(256), fl * sizeof(float), st, a);
    AESR_LAUNCH_CHECK("conv_cout1_wgrad");
    return AESR_OK;
}
