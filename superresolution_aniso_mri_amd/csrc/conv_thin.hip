// "Thin" 3x3 convolutions: one side of the convolution has a single channel, so the op is bandwidth-bound on the
// multi-channel tensor and does not belong on the matrix cores.
//
//   expand   out[n,y,x,c] = act(b[c] + sum_t ( w[t][c] * s[n, y+dy_t-ps, x+dx_t-ps]  +  [ (y+dy_t, x+dx_t) in grid ] * be[t][c] )) * mask
//   reduce   r[t][c]  = sum_{n,y,x} T[n,y,x,c] * s[n, y+dy_t-ps, x+dx_t-ps]
//            rb[t][c] = sum_{n,y,x} [ (y+dy_t, x+dx_t) in grid ] * T[n,y,x,c]          ssum = sum s over the grid
//
// (t = 3*(dy+1)+(dx+1), dy,dx in -1..1; s is zero outside its [Hs,Ws] extent; "grid" is the [Ho,Wo] extent of the
// multi-channel tensor.)  Users:
//   * the encoder stem folded into the first 3x3 convolution (networks/acai_vanilla.py:51,55): a 1x1 pad-1 conv
//     colors=1 -> Cs followed, with no non-linearity in between, by a 3x3 pad-1 conv Cs -> C1 is exactly a 1 -> C1
//     3x3 conv with the folded filter  weff[t][co] = sum_c W1[co,c,t]*ws[c]  plus a per-tap bias
//     beff[t][co] = sum_c W1[co,c,t]*bs[c]  that only counts for taps that land inside the stem's (H+2)x(W+2) output
//     grid.  Forward = expand; backward = reduce (+ the chain rule back to W1, ws, bs in thin_stem_finish_kernel).
//     The Cs-channel stem tensor never exists.
//   * the Cout == 1 output convolution (networks/acai_vanilla.py:98): its data gradient is an expand of dy with the
//     flipped filter (and the derivative of the LeakyReLU in front of it as the mask), its weight gradient a reduce of
//     the saved input against dy.
//
// Tile = 8 rows x min(64, 256 / (C/4)) columns; a thread owns one channel quad of one column and walks the 8 rows; the
// single-channel patch (tile + halo) sits in LDS.  16-byte accesses on the multi-channel side, 1 KiB contiguous per
// wave.
#include "aesr_kernels.h"
#include "aesr_pack_dev.h"

#define THIN_TH 8
#ifndef THIN_ETH
#define THIN_ETH 16              // rows of an expand tile: the patch staging and the 18 filter / bias loads of a thread are per tile
#endif
#define THIN_XS 68

__host__ __device__ __forceinline__ int thin_tw(int C4) { return 256 / C4 < 64 ? 256 / C4 : 64; }

template <int TH = THIN_TH>
__device__ __forceinline__ void thin_stage(const ThinArgs& a, float (*xs)[THIN_XS], int n, int y0, int x0, int TW) {
    const int PW = TW + 2;
    for (int q = threadIdx.x; q < (TH + 2) * PW; q += 256) {
        const int r = q / PW, c = q - r * PW;
        const int sy = y0 + r - 1 - a.ps, sx = x0 + c - 1 - a.ps;
        float v = 0.f;
        if (sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws) v = a.s[((size_t)n * a.Hs + sy) * a.Ws + sx];
        xs[r][c] = v;
    }
}

__global__ __launch_bounds__(256) void thin_expand_kernel(ThinArgs a) {
    __shared__ float xs[THIN_ETH + 2][THIN_XS];
    const int C4 = a.C >> 2, TW = thin_tw(C4);
    const int tid = threadIdx.x, c4 = tid % C4, pl = tid / C4;
    int tile = blockIdx.x;
    const int tx = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int n = tile / a.tiles_y;
    const int y0 = ty * THIN_ETH, x0 = tx * TW;
    thin_stage<THIN_ETH>(a, xs, n, y0, x0, TW);
    f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *(const f32x4*)(a.w + t * a.C + c4 * 4);
    const f32x4 b0 = a.b ? *(const f32x4*)(a.b + c4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bfull = b0;
    if (a.be) {
#pragma unroll
        for (int t = 0; t < 9; ++t) bfull += *(const f32x4*)(a.be + t * a.C + c4 * 4);
    }
    __syncthreads();
    const int ox = x0 + pl;
    if (pl >= TW || ox >= a.Wo) return;
    const bool xedge = a.be && (ox == 0 || ox == a.Wo - 1);
    // none / ReLU / LeakyReLU as one branch-free form max(v, v * slope) (0 <= slope <= 1); the sigmoid keeps its own path
    const bool sigm = a.act == ACT_SIGMOID;
    const float nslope = a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU ? 0.f : 1.f);
    const bool fastact = !sigm && nslope >= 0.f && nslope <= 1.f;
#pragma unroll 4
    for (int r = 0; r < THIN_ETH; ++r) {
        const int oy = y0 + r;
        if (oy >= a.Ho) break;
        f32x4 acc = bfull;
        if (xedge || (a.be && (oy == 0 || oy == a.Ho - 1))) {       // border of the grid: only the taps that land inside count
            acc = b0;
            for (int t = 0; t < 9; ++t) {
                const int yy = oy + t / 3 - 1, xx = ox + t % 3 - 1;
                if (yy >= 0 && yy < a.Ho && xx >= 0 && xx < a.Wo) acc += *(const f32x4*)(a.be + t * a.C + c4 * 4);
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = xs[r + t / 3][pl + t % 3];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[t][e], v, acc[e]);
        }
        const size_t o = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.C + c4 * 4;
        if (fastact) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], acc[e] * nslope);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = act_apply(acc[e], a.act, a.slope);
        }
        if (a.ysave) {
            const f32x4 ys = *(const f32x4*)(a.ysave + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= act_grad_from_output(ys[e], a.mask_act, a.slope);
        }
        *(f32x4*)(a.out + o) = acc;
    }
}

// partial[wg][row][c]: rows 0..8 = r, (WITH_BE: rows 9..17 = rb), last row: column 0 = ssum, other columns 0
template <bool WITH_BE>
__global__ __launch_bounds__(256) void thin_reduce_kernel(ThinArgs a) {
    __shared__ float xs[THIN_TH + 2][THIN_XS];
    extern __shared__ __attribute__((aligned(16))) float red[];      // [4 waves][NROW][C]
    constexpr int NROW = WITH_BE ? 19 : 10;
    const int C4 = a.C >> 2, TW = thin_tw(C4);
    const int tid = threadIdx.x, c4 = tid % C4, pl = tid / C4;
    const int lane = tid & 63, wave = tid >> 6;
    f32x4 accW[9], accB[9];
    float accs = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        accW[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int tpi = a.tiles_y * a.tiles_x;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * THIN_TH, x0 = tx * TW;
        __syncthreads();
        thin_stage(a, xs, n, y0, x0, TW);
        const int ox = x0 + pl;
        const bool xin = pl < TW && ox < a.Wo;
        const int plr = pl < TW ? pl : 0;               // idle threads (C < 16) read a valid LDS column, their g is 0
        f32x4 g[THIN_TH];
#pragma unroll
        for (int r = 0; r < THIN_TH; ++r) {
            const int oy = y0 + r;
            g[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xin && oy < a.Ho) g[r] = *(const f32x4*)(a.t + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.C + c4 * 4);
        }
        float mxf[3];
        mxf[0] = (ox - 1 >= 0) ? 1.f : 0.f;
        mxf[1] = 1.f;
        mxf[2] = (ox + 1 < a.Wo) ? 1.f : 0.f;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < THIN_TH; ++r) {
            const int oy = y0 + r;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = xs[r + t / 3][plr + t % 3];
#pragma unroll
                for (int e = 0; e < 4; ++e) accW[t][e] = fmaf(g[r][e], v, accW[t][e]);
            }
            if (WITH_BE) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = oy + dy - 1;
                    const float myf = (yy >= 0 && yy < a.Ho) ? 1.f : 0.f;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float m = myf * mxf[dx];
#pragma unroll
                        for (int e = 0; e < 4; ++e) accB[dy * 3 + dx][e] = fmaf(g[r][e], m, accB[dy * 3 + dx][e]);
                    }
                }
            }
            if (c4 == 0 && xin && oy < a.Ho) accs += xs[r + 1][plr + 1];
        }
    }
    // lanes that share a channel quad (same lane % C4) -> butterfly; then the 4 waves through LDS, fixed order
    for (int m = C4; m < 64; m <<= 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                accW[t][e] += __shfl_xor(accW[t][e], m, 64);
                if (WITH_BE) accB[t][e] += __shfl_xor(accB[t][e], m, 64);
            }
    }
    for (int m = 1; m < 64; m <<= 1) accs += __shfl_xor(accs, m, 64);
    __syncthreads();
    if (lane < C4) {                              // C4 <= 64: lane == c4 here
        float* rw = red + (size_t)wave * NROW * a.C;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            *(f32x4*)(rw + t * a.C + c4 * 4) = accW[t];
            if (WITH_BE) *(f32x4*)(rw + (9 + t) * a.C + c4 * 4) = accB[t];
        }
        *(f32x4*)(rw + (NROW - 1) * a.C + c4 * 4) = (f32x4){c4 == 0 ? accs : 0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    const int nout = NROW * a.C;
    for (int o = tid; o < nout; o += 256)
        a.partial[(size_t)blockIdx.x * nout + o] =
            (red[o] + red[(size_t)nout + o]) + (red[2 * (size_t)nout + o] + red[3 * (size_t)nout + o]);
}

// collapse  out[n,y,x] = act(b + sum_{t,c} w[c][t] * T[n, y+dy_t, x+dx_t, c])      (C -> 1, 3x3 pad 1; w in PyTorch [1][C][3][3] order)
// The Cout == 1 output convolution forward (networks/acai_vanilla.py:98 + Sigmoid).  Thread = (channel quad, input column):
// it walks the rows of its column (every element of T is loaded exactly once, 16 B per lane, plus a one-column halo per
// tile) and produces, per output row, the three partial sums its column contributes to the outputs at x-1, x, x+1
// (12 fma each); the C/4 quad-lanes are summed with a butterfly and the three neighbours meet through a tiny LDS
// exchange.  Rows go in chunks of 4 so that 4 loads per lane are in flight.
__global__ __launch_bounds__(256) void thin_collapse_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out, int N, int H,
                                                            int W, int C, int TH, int tiles_y, int tiles_x, int act, float slope) {
    __shared__ float S[2][3][4][64];         // [chunk parity][kx][row of the chunk][pixel lane]
    const int C4 = C >> 2, TW = thin_tw(C4), TO = TW - 2;
    const int tid = threadIdx.x, c4 = tid % C4, pl = tid / C4;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int n = tile / tiles_y;
    const int y0 = ty * TH, x0 = tx * TO;
    const int xc = x0 - 1 + pl;                          // the input column this thread owns
    const bool live = pl < TW && xc >= 0 && xc < W;
    f32x4 wq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) wq[t][e] = w[(c4 * 4 + e) * 9 + t];
    const float b0 = bias ? bias[0] : 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const float* base = x + ((size_t)n * H * W + (live ? xc : 0)) * C + c4 * 4;
    // branch-free: every lane loads from a clamped (always valid) row and the value is zeroed afterwards, so the 4 loads of a
    // chunk are issued back to back (a bounds branch in front of each load serialised them: 3.2 TB/s)
    auto load1 = [&](int yy) -> f32x4 {
        const f32x4 v = *(const f32x4*)(base + (size_t)min(max(yy, 0), H - 1) * W * C);
        return (live && yy >= 0 && yy < H) ? v : zero;
    };
    f32x4 r[6];
    r[0] = load1(y0 - 1);
    r[1] = load1(y0);
    int par = 0;
    for (int i = 0; i < TH; i += 4, par ^= 1) {
        const int oy = y0 + i;
        if (oy >= H) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[2 + k] = load1(oy + 1 + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sk[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                f32x4 acc = zero;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(wq[ky * 3 + kx][e], r[k + ky][e], acc[e]);
                sk[kx] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
            }
            for (int m = 1; m < C4; m <<= 1) {
                sk[0] += __shfl_xor(sk[0], m, 64);
                sk[1] += __shfl_xor(sk[1], m, 64);
                sk[2] += __shfl_xor(sk[2], m, 64);
            }
            if (c4 == 0 && pl < TW) {
                S[par][0][k][pl] = sk[0];
                S[par][1][k][pl] = sk[1];
                S[par][2][k][pl] = sk[2];
            }
        }
        __syncthreads();
        for (int o = tid; o < 4 * TO; o += 256) {
            const int k = o / TO, j = o - k * TO;
            const int xo = x0 + j, yo = oy + k;
            if (xo < W && yo < H && i + k < TH) {
                // output column xo: kx = 0 comes from input column xo-1 (lane j), kx = 1 from xo (lane j+1), kx = 2 from xo+1
                const float v = (S[par][0][k][j] + S[par][1][k][j + 1]) + S[par][2][k][j + 2];
                out[((size_t)n * H + yo) * W + xo] = act_apply(v + b0, act, slope);
            }
        }
        r[0] = r[4];
        r[1] = r[5];
    }
}

// ---- tiny parameter-side kernels ----------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void thin_stem_fold_kernel(const float* __restrict__ ws, const float* __restrict__ bs,
                                                             const float* __restrict__ w1, float* __restrict__ folded, int Cs,
                                                             int C1) {
    stem_fold_elements(ws, bs, w1, folded, Cs, C1, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// R rows: [0..8] dweff[t][co], [9..17] dbeff[t][co].  Chain rule back to the four parameter tensors.
// blocks [0, Cs): dws[c], dbs[c] (one block per stem channel, fixed-order tree over the C1*9 terms);
// blocks [Cs, ...): dw1 / db1 element-wise.
__global__ __launch_bounds__(256) void thin_stem_finish_kernel(const float* __restrict__ R, const float* __restrict__ ws,
                                                               const float* __restrict__ bs, const float* __restrict__ w1,
                                                               float* __restrict__ dws, float* __restrict__ dbs,
                                                               float* __restrict__ dw1, float* __restrict__ db1, int Cs, int C1) {
    __shared__ double redw[256], redb[256];
    if ((int)blockIdx.x < Cs) {
        const int c = blockIdx.x;
        double sw = 0.0, sb = 0.0;
        for (int o = threadIdx.x; o < C1 * 9; o += 256) {
            const int co = o / 9, t = o - co * 9;
            const double wv = (double)w1[((size_t)co * Cs + c) * 9 + t];
            sw += wv * (double)R[t * C1 + co];
            sb += wv * (double)R[(9 + t) * C1 + co];
        }
        redw[threadIdx.x] = sw;
        redb[threadIdx.x] = sb;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) {
                redw[threadIdx.x] += redw[threadIdx.x + h];
                redb[threadIdx.x] += redb[threadIdx.x + h];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            dws[c] = (float)redw[0];
            if (dbs) dbs[c] = (float)redb[0];
        }
        return;
    }
    const int tid = (blockIdx.x - Cs) * 256 + threadIdx.x, nth = (gridDim.x - Cs) * 256;
    for (int o = tid; o < C1 * Cs * 9; o += nth) {
        const int t = o % 9, c = (o / 9) % Cs, co = o / (9 * Cs);
        float v = R[t * C1 + co] * ws[c];
        if (bs) v = fmaf(R[(9 + t) * C1 + co], bs[c], v);
        dw1[o] = v;
    }
    if (db1)
        for (int co = tid; co < C1; co += nth) db1[co] = R[(9 + 4) * C1 + co];     // centre tap is always inside the grid
}

__global__ __launch_bounds__(256) void thin_cout1_flip_kernel(const float* __restrict__ w, float* __restrict__ wexp, int Cin) {
    cout1_flip_elements(w, wexp, Cin, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// Cout == 1 conv: dw[ci*9+k] = R[8-k][ci], db = R[9][0]
__global__ __launch_bounds__(256) void thin_cout1_finish_kernel(const float* __restrict__ R, float* __restrict__ dw,
                                                                float* __restrict__ db, int Cin) {
    for (int o = blockIdx.x * 256 + threadIdx.x; o < 9 * Cin; o += gridDim.x * 256) {
        const int ci = o / 9, k = o - ci * 9;
        dw[o] = R[(8 - k) * Cin + ci];
    }
    if (db && blockIdx.x == 0 && threadIdx.x == 0) db[0] = R[9 * Cin];
}

// ---- launchers ------------------------------------------------------------------------------------------------------

static bool thin_shape_ok(const ThinArgs& a) {
    const int C4 = a.C / 4;
    return a.C >= 4 && a.C % 4 == 0 && a.C <= 256 && (C4 & (C4 - 1)) == 0;
}

int aesr_launch_thin_expand(ThinArgs a, hipStream_t st) {
    if (!thin_shape_ok(a)) {
        aesr_set_error("thin conv: C=%d must be 4 times a power of two (4..256)", a.C);
        return AESR_ERR_ARG;
    }
    const int TW = thin_tw(a.C / 4);
    a.tiles_y = ceil_div(a.Ho, THIN_ETH);
    a.tiles_x = ceil_div(a.Wo, TW);
    a.ntiles = a.N * a.tiles_y * a.tiles_x;
    hipLaunchKernelGGL(thin_expand_kernel, dim3(a.ntiles), dim3(256), 0, st, a);
    AESR_LAUNCH_CHECK("thin_expand");
    return AESR_OK;
}

int aesr_launch_thin_reduce(ThinArgs a, int nwg, hipStream_t st) {
    if (!thin_shape_ok(a)) {
        aesr_set_error("thin conv: C=%d must be 4 times a power of two (4..256)", a.C);
        return AESR_ERR_ARG;
    }
    const int TW = thin_tw(a.C / 4);
    a.tiles_y = ceil_div(a.Ho, THIN_TH);
    a.tiles_x = ceil_div(a.Wo, TW);
    a.ntiles = a.N * a.tiles_y * a.tiles_x;
    const int nrow = a.with_be ? 19 : 10;
    const size_t shmem = (size_t)4 * nrow * a.C * sizeof(float);
    if (a.with_be)
        hipLaunchKernelGGL(thin_reduce_kernel<true>, dim3(nwg), dim3(256), shmem, st, a);
    else
        hipLaunchKernelGGL(thin_reduce_kernel<false>, dim3(nwg), dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("thin_reduce");
    return AESR_OK;
}

int aesr_launch_thin_collapse(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int act,
                              float slope, hipStream_t st) {
    const int C4 = C / 4;
    if (C < 4 || C % 4 != 0 || C > 256 || (C4 & (C4 - 1)) != 0) {
        aesr_set_error("thin conv: C=%d must be 4 times a power of two (4..256)", C);
        return AESR_ERR_ARG;
    }
    const int TW = thin_tw(C4), TH = H < 20 ? H : 20;
    const int tiles_y = ceil_div(H, TH), tiles_x = ceil_div(W, TW - 2);
    hipLaunchKernelGGL(thin_collapse_kernel, dim3(N * tiles_y * tiles_x), dim3(256), 0, st, x, w, bias, out, N, H, W, C, TH, tiles_y,
                       tiles_x, act, slope);
    AESR_LAUNCH_CHECK("thin_collapse");
    return AESR_OK;
}

int aesr_launch_thin_stem_fold(const float* ws, const float* bs, const float* w1, float* folded, int Cs, int C1, hipStream_t st) {
    hipLaunchKernelGGL(thin_stem_fold_kernel, dim3(ceil_div(9 * C1, 256)), dim3(256), 0, st, ws, bs, w1, folded, Cs, C1);
    AESR_LAUNCH_CHECK("thin_stem_fold");
    return AESR_OK;
}

int aesr_launch_thin_stem_finish(const float* R, const float* ws, const float* bs, const float* w1, float* dws, float* dbs,
                                 float* dw1, float* db1, int Cs, int C1, hipStream_t st) {
    int grid = ceil_div(C1 * Cs * 9, 256);
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(thin_stem_finish_kernel, dim3(Cs + grid), dim3(256), 0, st, R, ws, bs, w1, dws, dbs, dw1, db1, Cs, C1);
    AESR_LAUNCH_CHECK("thin_stem_finish");
    return AESR_OK;
}

int aesr_launch_thin_cout1_flip(const float* w, float* wexp, int Cin, hipStream_t st) {
    hipLaunchKernelGGL(thin_cout1_flip_kernel, dim3(ceil_div(9 * Cin, 256)), dim3(256), 0, st, w, wexp, Cin);
    AESR_LAUNCH_CHECK("thin_cout1_flip");
    return AESR_OK;
}

int aesr_launch_thin_cout1_finish(const float* R, float* dw, float* db, int Cin, hipStream_t st) {
    hipLaunchKernelGGL(thin_cout1_finish_kernel, dim3(ceil_div(9 * Cin, 256)), dim3(256), 0, st, R, dw, db, Cin);
    AESR_LAUNCH_CHECK("thin_cout1_finish");
    return AESR_OK;
}
