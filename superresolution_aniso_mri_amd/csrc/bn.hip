// BatchNorm2d (train + eval) with the following AvgPool2d(2) / nearest Upsample(x2) fused, NHWC, with
// *statistic groups*: one launch normalises several independent sub-batches (e.g. enc(x[2B]) and
// enc(slice_between[B]) of kwatsch/cardiac/trainer_ae.py:18,180), each with its own batch statistics,
// exactly as if the reference had called the network once per sub-batch (running statistics are
// updated group after group, in order).
//
//   stats    : per-channel sum / sum of squares partials  -> bn_reduce -> double [G][2][C]
//              (under data parallel these sums are all-reduced across ranks = SyncBN)
//   finalize : mean, biased var -> invstd, scale = gamma*invstd, shift = beta - mean*scale; running stats
//              with momentum (unbiased var), num_batches_tracked (nn.BatchNorm2d semantics)
//   apply    : out = scale*pool_or_up(y) + shift          (affine commutes with the 2x2 mean)
//   backward : reduce sum(g), sum(g*xhat) -> apply dy = scale*(g - s1/M - xhat*s2/M) * act'(y)
//
// Replaces nn.BatchNorm2d + nn.AvgPool2d / nn.Upsample of networks/acai_vanilla.py:58-59,90-92.
#include "aesr_kernels.h"

enum { BN_MODE_NONE = 0, BN_MODE_POOL = 1, BN_MODE_UP = 2 };


__device__ __forceinline__ int group_of(const BnGroups& gr, int n) {
    int g = 0;
    for (int k = 1; k < gr.G; ++k)
        if (n >= gr.nstart[k]) g = k;
    return g;
}

// ---- forward statistics ---------------------------------------------------------------------------------
// partial[g][wg][2][C]; grid = (nwg, G)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ partial, int HW,
                                                       int C, BnGroups gr) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [PL][2][C]
    const int C4 = C >> 2, PL = 256 / C4;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int g = blockIdx.y;
    const size_t p0 = (size_t)gr.nstart[g] * HW, p1 = (size_t)gr.nstart[g + 1] * HW;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (pl < PL) {
        for (size_t p = p0 + (size_t)blockIdx.x * PL + pl; p < p1; p += (size_t)gridDim.x * PL) {
            const f32x4 v = *(const f32x4*)(y + p * C + c4 * 4);
            s += v;
            q += v * v;
        }
        *(f32x4*)(red + (pl * 2 + 0) * C + c4 * 4) = s;
        *(f32x4*)(red + (pl * 2 + 1) * C + c4 * 4) = q;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        float t = 0.f;
        for (int k = 0; k < PL; ++k) t += red[k * 2 * C + o];
        partial[((size_t)g * gridDim.x + blockIdx.x) * 2 * C + o] = t;
    }
}

// sums[g][2][C] (double) = sum_wg partial[g][wg][2][C]
// block = 64 columns x 4 row-lanes: coalesced across columns, fixed summation order (bitwise reproducible)
__global__ __launch_bounds__(1024) void bn_reduce_kernel(const float* __restrict__ partial, double* __restrict__ sums, int nwg,
                                                         int C, int G) {
    __shared__ double red[16][64];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + col;
    const int n = G * 2 * C;
    double s = 0.0;
    if (o < n) {
        const int g = o / (2 * C), r = o - g * 2 * C;
        const float* base = partial + (size_t)g * nwg * 2 * C + r;
        const size_t rs = (size_t)2 * C;
        // 8 independent loads in flight per thread (one load per trip = a round of memory latency per row: 10 us for 512 rows)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = rl;
        for (; k + 7 * 16 < nwg; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(k + u * 16) * rs];
            s0 += (double)v[0] + (double)v[4];
            s1 += (double)v[1] + (double)v[5];
            s2 += (double)v[2] + (double)v[6];
            s3 += (double)v[3] + (double)v[7];
        }
        for (; k < nwg; k += 16) s0 += (double)base[(size_t)k * rs];
        s = (s0 + s1) + (s2 + s3);
    }
    red[rl][col] = s;
    __syncthreads();
    if (rl == 0 && o < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][col];
        sums[o] = t;
    }
}

// train: stats from sums/counts (+ running update, group after group); eval: stats from the running buffers
struct BnCounts { double c[4]; };

struct BnFinArgs {
    const float* gamma; const float* beta; float* running_mean; float* running_var; long long* nbt;
    float* mean; float* invstd; float* scale; float* shift;
    int C, G;
    float momentum, eps;
    int train, update_running;
    BnCounts counts;
};

// one channel, one group: (sum, sum of squares) -> mean / invstd / scale / shift (+ running statistics)
__device__ __forceinline__ void bn_finalize_one(const BnFinArgs& f, int c, int g, double s0, double s1) {
    float m, iv;
    if (f.train) {
        const double M = f.counts.c[g];
        const double mu = s0 / M;
        double var = s1 / M - mu * mu;
        if (var < 0.0) var = 0.0;
        m = (float)mu;
        iv = (float)(1.0 / sqrt(var + (double)f.eps));
        if (f.update_running) {
            const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * m;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
        }
    } else {
        m = f.running_mean[c];
        iv = 1.f / sqrtf(f.running_var[c] + f.eps);
    }
    f.mean[g * f.C + c] = m;
    f.invstd[g * f.C + c] = iv;
    const float sc = f.gamma[c] * iv;
    f.scale[g * f.C + c] = sc;
    f.shift[g * f.C + c] = f.beta[c] - m * sc;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, BnFinArgs f) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && f.train && f.update_running && f.nbt) *f.nbt += f.G;
    if (c >= f.C) return;
    for (int g = 0; g < f.G; ++g)
        bn_finalize_one(f, c, g, f.train ? sums[(g * 2 + 0) * f.C + c] : 0.0, f.train ? sums[(g * 2 + 1) * f.C + c] : 0.0);
}

// block layout of the finalize kernels: 16 channels x 64 row-lanes
#define BNR_COLS 16
#define BNR_RL 64

// All 2*G column totals of a block's 16 channels with ONE wait on global memory: the 64 row-lanes of the block are split over the
// 2*G quantities (q = 2*g + {0: first sum, 1: second sum}), each thread strides its quantity's nwg partial rows, the partial
// sums meet in LDS and 2*G*16 threads add them up in a fixed order.  (One total at a time = four dependent rounds of global
// latency + 32 barriers: 10 us for a few KB.)  tot[q][col] is valid for every thread after the call.  G <= 4.
__device__ __forceinline__ void bn_all_totals(const float* __restrict__ partial, int nwg, int C, int G, int c, bool live,
                                              double (*red)[BNR_COLS], double (*tot)[BNR_COLS], int col, int slot) {
    const int NQ = 2 * G, RLQ = BNR_RL / NQ;
    const int q = slot % NQ, rl = slot / NQ;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (live && rl < RLQ) {
        const float* base = partial + ((size_t)(q >> 1) * nwg * 2 + (q & 1)) * C + c;      // partial[g][k][s][c], row stride 2*C
        const size_t rs = (size_t)2 * C;
        int k = rl;
        for (; k + 7 * RLQ < nwg; k += 8 * RLQ) {          // 8 independent loads in flight per thread
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(k + u * RLQ) * rs];
            s0 += (double)v[0] + (double)v[4];
            s1 += (double)v[1] + (double)v[5];
            s2 += (double)v[2] + (double)v[6];
            s3 += (double)v[3] + (double)v[7];
        }
        for (; k < nwg; k += RLQ) s0 += (double)base[k * rs];
    }
    red[slot][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slot < NQ) {
        double t = 0.0;
        for (int r = 0; r < RLQ; ++r) t += red[r * NQ + slot][col];
        tot[slot][col] = t;
    }
    __syncthreads();
}

// Statistics partials -> finalize in ONE launch (no SyncBN exchange in between).  block = 16 channels x 64 row-lanes.
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ partial, int nwg, BnFinArgs f) {
    __shared__ double red[BNR_RL][BNR_COLS];
    __shared__ double tot[8][BNR_COLS];
    const int col = threadIdx.x % BNR_COLS, rl = threadIdx.x / BNR_COLS;
    const int c = blockIdx.x * BNR_COLS + col;
    const bool live = c < f.C;
    if (c == 0 && rl == 0 && f.update_running && f.nbt) *f.nbt += f.G;
    bn_all_totals(partial, nwg, f.C, f.G, c, live, red, tot, col, rl);
    if (rl == 0 && live)
        for (int g = 0; g < f.G; ++g) bn_finalize_one(f, c, g, tot[2 * g][col], tot[2 * g + 1][col]);      // group order: running stats
}

// ---- forward apply (+pool / +upsample) ----------------------------------------------------------------

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a) {
    const int C4 = a.C >> 2;
    const size_t total = (size_t)a.N * a.Ho * a.Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = idx % C4;
        size_t pix = idx / C4;
        const int x = pix % a.Wo;
        pix /= a.Wo;
        const int yy = pix % a.Ho;
        const int n = pix / a.Ho;
        const int g = group_of(a.gr, n);
        const f32x4 sc = *(const f32x4*)(a.scale + g * a.C + c4 * 4);
        const f32x4 sh = *(const f32x4*)(a.shift + g * a.C + c4 * 4);
        f32x4 v;
        if (a.mode == BN_MODE_POOL) {
            const float* b = a.y + (((size_t)n * a.H + 2 * yy) * a.W + 2 * x) * a.C + c4 * 4;
            const f32x4 v00 = *(const f32x4*)b, v01 = *(const f32x4*)(b + a.C);
            const f32x4 v10 = *(const f32x4*)(b + (size_t)a.W * a.C), v11 = *(const f32x4*)(b + (size_t)a.W * a.C + a.C);
            v = ((v00 + v01) + (v10 + v11)) * 0.25f;
        } else if (a.mode == BN_MODE_UP) {
            v = *(const f32x4*)(a.y + (((size_t)n * a.H + (yy >> 1)) * a.W + (x >> 1)) * a.C + c4 * 4);
        } else {
            v = *(const f32x4*)(a.y + (((size_t)n * a.H + yy) * a.W + x) * a.C + c4 * 4);
        }
        *(f32x4*)(a.out + idx * 4) = v * sc + sh;
    }
}

// Data parallel (SyncBN): finalize + apply in ONE launch.  The all-reduced sums are [G][2][C] doubles -- a few hundred values -- so every
// block derives scale / shift for all groups and channels itself (into LDS, the arithmetic of bn_finalize_one) and block 0 alone
// writes mean / invstd / scale / shift for the backward pass and updates the running statistics, group after group.
constexpr int BN_FUSE_MAX = 1024;       // G * C values a block keeps in LDS

__device__ __forceinline__ void bn_finalize_vals(const BnFinArgs& f, int g, double s0, double s1, float* m, float* iv, double* unb) {
    const double M = f.counts.c[g];
    const double mu = s0 / M;
    double var = s1 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    *m = (float)mu;
    *iv = (float)(1.0 / sqrt(var + (double)f.eps));
    *unb = M > 1.0 ? var * M / (M - 1.0) : var;
}

__global__ __launch_bounds__(256) void bn_finalize_apply_kernel(const double* __restrict__ sums, BnFinArgs f, BnApplyArgs a) {
    __shared__ __attribute__((aligned(16))) float s_sc[BN_FUSE_MAX], s_sh[BN_FUSE_MAX];
    const int GC = f.G * f.C;
    for (int i = threadIdx.x; i < GC; i += 256) {
        const int g = i / f.C, c = i - g * f.C;
        float m, iv;
        double unb;
        bn_finalize_vals(f, g, sums[(g * 2 + 0) * f.C + c], sums[(g * 2 + 1) * f.C + c], &m, &iv, &unb);
        const float sc = f.gamma[c] * iv, sh = f.beta[c] - m * sc;
        s_sc[i] = sc;
        s_sh[i] = sh;
        if (blockIdx.x == 0) {
            f.mean[i] = m;
            f.invstd[i] = iv;
            f.scale[i] = sc;
            f.shift[i] = sh;
        }
    }
    if (blockIdx.x == 0 && f.update_running) {
        if (threadIdx.x == 0 && f.nbt) *f.nbt += f.G;
        for (int c = threadIdx.x; c < f.C; c += 256) {
            float rm = f.running_mean[c], rv = f.running_var[c];
            for (int g = 0; g < f.G; ++g) {             // group after group, as the reference's successive calls
                float m, iv;
                double unb;
                bn_finalize_vals(f, g, sums[(g * 2 + 0) * f.C + c], sums[(g * 2 + 1) * f.C + c], &m, &iv, &unb);
                rm = (1.f - f.momentum) * rm + f.momentum * m;
                rv = (1.f - f.momentum) * rv + f.momentum * (float)unb;
            }
            f.running_mean[c] = rm;
            f.running_var[c] = rv;
        }
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const size_t total = (size_t)a.N * a.Ho * a.Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = idx % C4;
        size_t pix = idx / C4;
        const int x = pix % a.Wo;
        pix /= a.Wo;
        const int yy = pix % a.Ho;
        const int n = pix / a.Ho;
        const int g = group_of(a.gr, n);
        const f32x4 sc = *(const f32x4*)(s_sc + g * a.C + c4 * 4);
        const f32x4 sh = *(const f32x4*)(s_sh + g * a.C + c4 * 4);
        f32x4 v;
        if (a.mode == BN_MODE_POOL) {
            const float* b = a.y + (((size_t)n * a.H + 2 * yy) * a.W + 2 * x) * a.C + c4 * 4;
            const f32x4 v00 = *(const f32x4*)b, v01 = *(const f32x4*)(b + a.C);
            const f32x4 v10 = *(const f32x4*)(b + (size_t)a.W * a.C), v11 = *(const f32x4*)(b + (size_t)a.W * a.C + a.C);
            v = ((v00 + v01) + (v10 + v11)) * 0.25f;
        } else if (a.mode == BN_MODE_UP) {
            v = *(const f32x4*)(a.y + (((size_t)n * a.H + (yy >> 1)) * a.W + (x >> 1)) * a.C + c4 * 4);
        } else {
            v = *(const f32x4*)(a.y + (((size_t)n * a.H + yy) * a.W + x) * a.C + c4 * 4);
        }
        *(f32x4*)(a.out + idx * 4) = v * sc + sh;
    }
}

// ---- backward --------------------------------------------------------------------------------------------

// MODE >= 0: compile-time mode with branch-free loads (a divergent bounds branch in front of the load kept the compiler from
// batching the loads of the unrolled pixels); MODE < 0 keeps the run-time form
template <int MODE>
__device__ __forceinline__ f32x4 bn_gather_g_t(const BnBwdArgs& a, int n, int yy, int x, int c4) {
    // gradient w.r.t. the BN output at full-resolution BN pixel (yy, x)
    const int mode = MODE < 0 ? a.mode : MODE;
    if (mode == BN_MODE_POOL) {
        const int py = yy >> 1, px = x >> 1;
        if (MODE < 0) {
            if (py >= a.Ho || px >= a.Wo) return (f32x4){0.f, 0.f, 0.f, 0.f};
            return *(const f32x4*)(a.gout + (((size_t)n * a.Ho + py) * a.Wo + px) * a.C + c4 * 4) * 0.25f;
        }
        const float w = (py < a.Ho && px < a.Wo) ? 0.25f : 0.f;          // odd H / W: the last row / column is not pooled
        return *(const f32x4*)(a.gout + (((size_t)n * a.Ho + min(py, a.Ho - 1)) * a.Wo + min(px, a.Wo - 1)) * a.C + c4 * 4) * w;
    } else if (mode == BN_MODE_UP) {
        const float* b = a.gout + (((size_t)n * a.Ho + 2 * yy) * a.Wo + 2 * x) * a.C + c4 * 4;
        return (*(const f32x4*)b + *(const f32x4*)(b + a.C)) +
               (*(const f32x4*)(b + (size_t)a.Wo * a.C) + *(const f32x4*)(b + (size_t)a.Wo * a.C + a.C));
    }
    return *(const f32x4*)(a.gout + (((size_t)n * a.H + yy) * a.W + x) * a.C + c4 * 4);
}

// exact floor(p / d) for p < 2^31 with the host's (m, k) = (ceil(2^k / d), 31 + ceil(log2 d)): one 64-bit multiply and a shift
__device__ __forceinline__ unsigned bn_fastdiv(unsigned p, unsigned long long m, int k) {
    return (unsigned)(((unsigned long long)p * m) >> k);
}

// grid = (nwg, G)
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int C4 = a.C >> 2, PL = 256 / C4;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int g = blockIdx.y;
    const unsigned HW = a.H * a.W;
    const unsigned p0 = a.gr.nstart[g] * HW, p1 = a.gr.nstart[g + 1] * HW;       // host checks N*H*W < 2^31
    const f32x4 mu = *(const f32x4*)(a.mean + g * a.C + c4 * 4);
    const f32x4 iv = *(const f32x4*)(a.invstd + g * a.C + c4 * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (pl < PL) {
        // BNR_U pixels per trip: their loads (y and the 1..4 gathered gradient pieces each) are independent, which multiplies the
        // bytes in flight per thread -- one pixel per trip left the kernel latency-bound (2.2-3.6 TB/s with 8 waves per CU)
        constexpr int U = 2;
        const unsigned S = gridDim.x * PL;
        unsigned p = p0 + blockIdx.x * PL + pl;
        f32x4 a1[U], a2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a1[u] = a2[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (; p + (U - 1) * S < p1; p += U * S) {
            f32x4 v[U], gq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned q = p + u * S;
                const unsigned n = bn_fastdiv(q, a.m_hw, a.k_hw);          // q / HW, q % HW, rem / W without ~40-instruction
                const unsigned rem = q - n * HW;                            // divisions (C/4 threads repeat them per pixel)
                const unsigned yy = bn_fastdiv(rem, a.m_w, a.k_w);
                v[u] = *(const f32x4*)(a.y + (size_t)q * a.C + c4 * 4);
                gq[u] = bn_gather_g_t<MODE>(a, n, yy, rem - yy * a.W, c4);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                a1[u] += gq[u];
                a2[u] += gq[u] * ((v[u] - mu) * iv);
            }
        }
        for (; p < p1; p += S) {
            const unsigned n = bn_fastdiv(p, a.m_hw, a.k_hw);
            const unsigned rem = p - n * HW;
            const unsigned yy = bn_fastdiv(rem, a.m_w, a.k_w), x = rem - yy * a.W;
            const f32x4 gg = bn_gather_g_t<MODE>(a, n, yy, x, c4);
            const f32x4 xh = (*(const f32x4*)(a.y + (size_t)p * a.C + c4 * 4) - mu) * iv;
            s1 += gg;
            s2 += gg * xh;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s1 += a1[u];
            s2 += a2[u];
        }
        *(f32x4*)(red + (pl * 2 + 0) * a.C + c4 * 4) = s1;
        *(f32x4*)(red + (pl * 2 + 1) * a.C + c4 * 4) = s2;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * a.C; o += 256) {
        float t = 0.f;
        for (int k = 0; k < PL; ++k) t += red[k * 2 * a.C + o];
        a.partial[((size_t)g * gridDim.x + blockIdx.x) * 2 * a.C + o] = t;
    }
}

// coef[g][2][C] = sums/M ; dgamma[c] = sum_g s2 ; dbeta[c] = sum_g s1
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, BnCounts counts,
                                       float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       int C, int G) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double dg = 0.0, db = 0.0;
    for (int g = 0; g < G; ++g) {
        const double s1 = sums[(g * 2 + 0) * C + c], s2 = sums[(g * 2 + 1) * C + c];
        coef[(g * 2 + 0) * C + c] = (float)(s1 / counts.c[g]);
        coef[(g * 2 + 1) * C + c] = (float)(s2 / counts.c[g]);
        db += s1;
        dg += s2;
    }
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
}

// backward-reduce partials -> coef / dgamma / dbeta in ONE launch (same math as bn_reduce + bn_bwd_finalize)
__global__ __launch_bounds__(1024) void bn_bwd_reduce_finalize_kernel(const float* __restrict__ partial, int nwg, BnCounts counts,
                                                                      float* __restrict__ coef, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, int C, int G) {
    __shared__ double red[BNR_RL][BNR_COLS];
    const int col = threadIdx.x % BNR_COLS, rl = threadIdx.x / BNR_COLS;
    const int c = blockIdx.x * BNR_COLS + col;
    const bool live = c < C;
    __shared__ double tot[8][BNR_COLS];
    bn_all_totals(partial, nwg, C, G, c, live, red, tot, col, rl);
    if (rl == 0 && live) {
        double dg = 0.0, db = 0.0;
        for (int g = 0; g < G; ++g) {
            const double s1 = tot[2 * g][col], s2 = tot[2 * g + 1][col];
            coef[(g * 2 + 0) * C + c] = (float)(s1 / counts.c[g]);
            coef[(g * 2 + 1) * C + c] = (float)(s2 / counts.c[g]);
            db += s1;
            dg += s2;
        }
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
    const int C4 = a.C >> 2;
    const size_t total = (size_t)a.N * a.H * a.W * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = idx % C4;
        size_t pix = idx / C4;
        const int x = pix % a.W;
        pix /= a.W;
        const int yy = pix % a.H;
        const int n = pix / a.H;
        const int g = group_of(a.gr, n);
        const f32x4 mu = *(const f32x4*)(a.mean + g * a.C + c4 * 4);
        const f32x4 iv = *(const f32x4*)(a.invstd + g * a.C + c4 * 4);
        const f32x4 sc = *(const f32x4*)(a.scale + g * a.C + c4 * 4);
        const f32x4 k1 = *(const f32x4*)(a.coef + (g * 2 + 0) * a.C + c4 * 4);
        const f32x4 k2 = *(const f32x4*)(a.coef + (g * 2 + 1) * a.C + c4 * 4);
        const f32x4 yv = *(const f32x4*)(a.y + idx * 4);
        const f32x4 gg = bn_gather_g_t<MODE>(a, n, yy, x, c4);
        const f32x4 xh = (yv - mu) * iv;
        f32x4 d = sc * (gg - k1 - xh * k2);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] *= act_grad_from_output(yv[e], a.act, a.slope);
        *(f32x4*)(a.dpre + idx * 4) = d;
    }
}

// the same for the backward pass: coef = sums / M per block in LDS, block 0 writes coef / dgamma / dbeta
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_finalize_apply_kernel(const double* __restrict__ sums, BnCounts counts, float* __restrict__ coef,
                                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int G, BnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float s_k[2 * BN_FUSE_MAX];       // [g][2][C]
    const int GC2 = G * 2 * a.C;
    for (int i = threadIdx.x; i < GC2; i += 256) {
        const int g = i / (2 * a.C);
        const float k = (float)(sums[i] / counts.c[g]);
        s_k[i] = k;
        if (blockIdx.x == 0) coef[i] = k;
    }
    if (blockIdx.x == 0)
        for (int c = threadIdx.x; c < a.C; c += 256) {
            double dg = 0.0, db = 0.0;
            for (int g = 0; g < G; ++g) {
                db += sums[(g * 2 + 0) * a.C + c];
                dg += sums[(g * 2 + 1) * a.C + c];
            }
            dgamma[c] = (float)dg;
            dbeta[c] = (float)db;
        }
    __syncthreads();
    const int C4 = a.C >> 2;
    const size_t total = (size_t)a.N * a.H * a.W * C4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int c4 = idx % C4;
        size_t pix = idx / C4;
        const int x = pix % a.W;
        pix /= a.W;
        const int yy = pix % a.H;
        const int n = pix / a.H;
        const int g = group_of(a.gr, n);
        const f32x4 mu = *(const f32x4*)(a.mean + g * a.C + c4 * 4);
        const f32x4 iv = *(const f32x4*)(a.invstd + g * a.C + c4 * 4);
        const f32x4 sc = *(const f32x4*)(a.scale + g * a.C + c4 * 4);
        const f32x4 k1 = *(const f32x4*)(s_k + (g * 2 + 0) * a.C + c4 * 4);
        const f32x4 k2 = *(const f32x4*)(s_k + (g * 2 + 1) * a.C + c4 * 4);
        const f32x4 yv = *(const f32x4*)(a.y + idx * 4);
        const f32x4 gg = bn_gather_g_t<MODE>(a, n, yy, x, c4);
        const f32x4 xh = (yv - mu) * iv;
        f32x4 d = sc * (gg - k1 - xh * k2);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] *= act_grad_from_output(yv[e], a.act, a.slope);
        *(f32x4*)(a.dpre + idx * 4) = d;
    }
}

// ---- launchers ----------------------------------------------------------------------------------------------
static int bn_check_c(int C, const char* who) {
    if (C % 4 != 0 || 256 % (C / 4) != 0 || C > 1024) {
        aesr_set_error("%s: unsupported channel count %d (need C/4 | 256)", who, C);
        return AESR_ERR_ARG;
    }
    return AESR_OK;
}

int aesr_launch_bn_stats(const float* y, float* partial, int HW, int C, const BnGroups& gr, int nwg, hipStream_t st) {
    if (int e = bn_check_c(C, "bn_stats")) return e;
    const int PL = 256 / (C / 4);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nwg, gr.G), dim3(256), (size_t)PL * 2 * C * sizeof(float), st, y, partial, HW, C, gr);
    AESR_LAUNCH_CHECK("bn_stats");
    return AESR_OK;
}

int aesr_launch_bn_reduce(const float* partial, double* sums, int nwg, int C, int G, hipStream_t st) {
    hipLaunchKernelGGL(bn_reduce_kernel, dim3(ceil_div(G * 2 * C, 64)), dim3(1024), 0, st, partial, sums, nwg, C, G);
    AESR_LAUNCH_CHECK("bn_reduce");
    return AESR_OK;
}

static BnFinArgs bn_fin_args(const double* counts, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             long long* nbt, float* mean, float* invstd, float* scale, float* shift, int C, int G, float momentum,
                             float eps, int train, int update_running) {
    BnFinArgs f;
    f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var; f.nbt = nbt;
    f.mean = mean; f.invstd = invstd; f.scale = scale; f.shift = shift; f.C = C; f.G = G; f.momentum = momentum; f.eps = eps;
    f.train = train; f.update_running = update_running;
    for (int g = 0; g < 4; ++g) f.counts.c[g] = (counts && g < G) ? counts[g] : 1.0;
    return f;
}

int aesr_launch_bn_finalize(const double* sums, const double* counts, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* nbt, float* mean, float* invstd,
                            float* scale, float* shift, int C, int G, float momentum, float eps, int train,
                            int update_running, hipStream_t st) {
    const BnFinArgs f = bn_fin_args(counts, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift, C, G, momentum,
                                    eps, train, update_running);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, sums, f);
    AESR_LAUNCH_CHECK("bn_finalize");
    return AESR_OK;
}

int aesr_launch_bn_reduce_finalize(const float* partial, int nwg, const double* counts, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, long long* nbt, float* mean, float* invstd,
                                   float* scale, float* shift, int C, int G, float momentum, float eps, int update_running,
                                   hipStream_t st) {
    const BnFinArgs f = bn_fin_args(counts, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift, C, G, momentum,
                                    eps, 1, update_running);
    hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(ceil_div(C, BNR_COLS)), dim3(BNR_COLS * BNR_RL), 0, st, partial, nwg, f);
    AESR_LAUNCH_CHECK("bn_reduce_finalize");
    return AESR_OK;
}

int aesr_launch_bn_bwd_reduce_finalize(const float* partial, int nwg, const double* counts, float* coef, float* dgamma,
                                       float* dbeta, int C, int G, hipStream_t st) {
    BnCounts cnt;
    for (int g = 0; g < 4; ++g) cnt.c[g] = (counts && g < G) ? counts[g] : 1.0;
    hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3(ceil_div(C, BNR_COLS)), dim3(BNR_COLS * BNR_RL), 0, st, partial, nwg, cnt, coef, dgamma,
                       dbeta, C, G);
    AESR_LAUNCH_CHECK("bn_bwd_reduce_finalize");
    return AESR_OK;
}

int aesr_launch_bn_apply(const BnApplyArgs& a, hipStream_t st) {
    if (int e = bn_check_c(a.C, "bn_apply")) return e;
    const size_t total = (size_t)a.N * a.Ho * a.Wo * (a.C / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, st, a);
    AESR_LAUNCH_CHECK("bn_apply");
    return AESR_OK;
}

static void bn_magic(unsigned d, unsigned long long* m, int* k) {
    int L = 0;
    while ((1u << L) < d) ++L;
    *k = 31 + L;
    const unsigned __int128 one = (unsigned __int128)1 << *k;
    *m = (unsigned long long)((one + d - 1) / d);
}

int aesr_launch_bn_bwd_reduce(const BnBwdArgs& a_in, int nwg, hipStream_t st) {
    BnBwdArgs a = a_in;
    bn_magic((unsigned)(a.H * a.W), &a.m_hw, &a.k_hw);
    bn_magic((unsigned)a.W, &a.m_w, &a.k_w);
    if (int e = bn_check_c(a.C, "bn_bwd_reduce")) return e;
    const int PL = 256 / (a.C / 4);
    const dim3 grid(nwg, a.gr.G);
    const size_t shm = (size_t)PL * 2 * a.C * sizeof(float);
    if (a.mode == BN_MODE_POOL) hipLaunchKernelGGL(bn_bwd_reduce_kernel<BN_MODE_POOL>, grid, dim3(256), shm, st, a);
    else if (a.mode == BN_MODE_UP) hipLaunchKernelGGL(bn_bwd_reduce_kernel<BN_MODE_UP>, grid, dim3(256), shm, st, a);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, grid, dim3(256), shm, st, a);
    AESR_LAUNCH_CHECK("bn_bwd_reduce");
    return AESR_OK;
}

int aesr_launch_bn_bwd_finalize(const double* sums, const double* counts, float* coef, float* dgamma, float* dbeta, int C,
                                int G, hipStream_t st) {
    BnCounts cnt;
    for (int g = 0; g < 4; ++g) cnt.c[g] = (counts && g < G) ? counts[g] : 1.0;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, sums, cnt, coef, dgamma, dbeta, C, G);
    AESR_LAUNCH_CHECK("bn_bwd_finalize");
    return AESR_OK;
}

int aesr_launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t st) {
    if (int e = bn_check_c(a.C, "bn_bwd_apply")) return e;
    const size_t total = (size_t)a.N * a.H * a.W * (a.C / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (a.mode == BN_MODE_POOL) hipLaunchKernelGGL(bn_bwd_apply_kernel<BN_MODE_POOL>, dim3(grid), dim3(256), 0, st, a);
    else if (a.mode == BN_MODE_UP) hipLaunchKernelGGL(bn_bwd_apply_kernel<BN_MODE_UP>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(grid), dim3(256), 0, st, a);
    AESR_LAUNCH_CHECK("bn_bwd_apply");
    return AESR_OK;
}

// finalize + apply / backward finalize + apply in one launch (data parallel, train mode); false: too many values for the LDS tables
bool aesr_bn_fused_ok(int C, int G) { return G * C <= BN_FUSE_MAX && G >= 1 && G <= 4; }

int aesr_launch_bn_finalize_apply(const double* sums, const double* counts, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, long long* nbt, float* mean, float* invstd, float* scale, float* shift, float momentum,
                                  float eps, int update_running, int G, const BnApplyArgs& a, hipStream_t st) {
    if (int e = bn_check_c(a.C, "bn_finalize_apply")) return e;
    const BnFinArgs f = bn_fin_args(counts, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift, a.C, G, momentum, eps, 1,
                                    update_running);
    const size_t total = (size_t)a.N * a.Ho * a.Wo * (a.C / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(grid), dim3(256), 0, st, sums, f, a);
    AESR_LAUNCH_CHECK("bn_finalize_apply");
    return AESR_OK;
}

int aesr_launch_bn_bwd_finalize_apply(const double* sums, const double* counts, float* coef, float* dgamma, float* dbeta, int G,
                                      const BnBwdArgs& a, hipStream_t st) {
    if (int e = bn_check_c(a.C, "bn_bwd_finalize_apply")) return e;
    BnCounts cnt;
    for (int g = 0; g < 4; ++g) cnt.c[g] = (counts && g < G) ? counts[g] : 1.0;
    const size_t total = (size_t)a.N * a.H * a.W * (a.C / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (a.mode == BN_MODE_POOL) hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<BN_MODE_POOL>, dim3(grid), dim3(256), 0, st, sums, cnt, coef, dgamma, dbeta, G, a);
    else if (a.mode == BN_MODE_UP) hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<BN_MODE_UP>, dim3(grid), dim3(256), 0, st, sums, cnt, coef, dgamma, dbeta, G, a);
    else hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<0>, dim3(grid), dim3(256), 0, st, sums, cnt, coef, dgamma, dbeta, G, a);
    AESR_LAUNCH_CHECK("bn_bwd_finalize_apply");
    return AESR_OK;
}
