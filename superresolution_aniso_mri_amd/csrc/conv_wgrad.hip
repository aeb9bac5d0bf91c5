// Weight gradient of a stride-1 convolution on the fp32 matrix cores, NHWC.
//
//   dW[co,ci,ky,kx] = sum_{n,y,x} dY[n,y,x,co] * X[n,y+ky-pad,x+kx-pad,ci]      db[co] = sum dY[n,y,x,co]
//
// GEMM view per tap: M = ci, N = co, K = pixels (hundreds of thousands) -> a split-K design:
//  * grid = (S splits, Cin chunks of 16*CIB, Cout chunks of 16*COBW*WCO); each workgroup walks the
//    pixel tiles s, s+S, s+2S ... and keeps its whole [KS*KS][CIB][COBW] accumulator set in registers
//    for the entire walk, so the only global writes are ONE partial slab per wave at the end;
//  * a tile's X patch (tile + halo) and dY tile are staged in LDS (channel halves XOR-swizzled by
//    column parity so the two pixels of a half-wave hit different banks);
//  * waves are arranged WCO (over co blocks) x WK = 4/WCO (over the tile's pixels);
//  * the bias gradient rides along as one extra MFMA per k-step with A = e_0 (row 0 of the extra
//    accumulator block = column sums of dY), computed only by the ci-chunk-0 workgroups;
//  * a second kernel sums the slabs in a fixed order (bitwise reproducible, no atomics) and writes
//    dW in the PyTorch [Cout][Cin][KS][KS] layout.
//
// Replaces autograd's conv2d weight-gradient for the layers of networks/acai_vanilla.py:49-102.
#include "aesr_kernels.h"


template <int KS, int CIB, int COBW, int WCO>
__global__ __launch_bounds__(256) void conv_wgrad_f32(WgradArgs a) {
    constexpr int WK = 4 / WCO;
    constexpr int ROWP = CIB * 16;           // floats per patch pixel in LDS
    constexpr int COT = COBW * WCO * 16;     // couts per workgroup
    constexpr int ROWD = COT;                // floats per dY pixel in LDS
    constexpr int SWZP = ROWP >= 32 ? 16 : 0;
    constexpr int SWZD = ROWD >= 32 ? 16 : 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int wco = wave % WCO, wk = wave / WCO;
    const int ci0 = blockIdx.y * ROWP, co0 = blockIdx.z * COT;
    const int PW = a.TW + KS - 1, PH = a.TH + KS - 1, PP = PH * PW, TP = a.TH * a.TW;
    float* ldsP = lds;
    float* ldsD = lds + PP * ROWP;
    const bool do_bias = (blockIdx.y == 0);

    f32x4 acc[KS * KS][CIB][COBW];
    f32x4 accb[COBW];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int i = 0; i < CIB; ++i)
#pragma unroll
            for (int j = 0; j < COBW; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < COBW; ++j) accb[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a_one = (l15 == 0) ? 1.f : 0.f;

    const int tpi = a.tiles_y * a.tiles_x;
    const int kq = a.TW >> 2;               // k-steps per tile row
    const int nks = TP >> 2;                // k-steps per tile

    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S) {
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        // ---- stage X patch [PP][ROWP] ----
        for (int q = tid; q < PP * (ROWP / 4); q += 256) {
            const int p = q / (ROWP / 4), part = q - p * (ROWP / 4);
            const int pr = p / PW, pc = p - pr * PW;
            const int gy = y0 + pr - a.pad, gx = x0 + pc - a.pad, ci = ci0 + part * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci < a.Cin)
                v = *(const f32x4*)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + ci);
            *(f32x4*)(ldsP + p * ROWP + ((part * 4) ^ ((pc & 1) ? SWZP : 0))) = v;
        }
        // ---- stage dY tile [TP][ROWD] ----
        for (int q = tid; q < TP * (ROWD / 4); q += 256) {
            const int p = q / (ROWD / 4), part = q - p * (ROWD / 4);
            const int r = p / a.TW, c = p - r * a.TW;
            const int gy = y0 + r, gx = x0 + c, co = co0 + part * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy < a.Ho && gx < a.Wo && co < a.Cout)
                v = *(const f32x4*)(a.dy + (((size_t)n * a.Ho + gy) * a.Wo + gx) * a.Cout + co);
            *(f32x4*)(ldsD + p * ROWD + ((part * 4) ^ ((c & 1) ? SWZD : 0))) = v;
        }
        __syncthreads();
        for (int ks = wk; ks < nks; ks += WK) {
            const int r = ks / kq, c = (ks - r * kq) * 4 + g;      // this lane's pixel of the k-step
            float bv[COBW];
#pragma unroll
            for (int j = 0; j < COBW; ++j)
                bv[j] = ldsD[(r * a.TW + c) * ROWD + (((wco * COBW + j) * 16 + l15) ^ ((c & 1) ? SWZD : 0))];
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                const int pp = (r + t / KS) * PW + c + t % KS;
                const int sw = ((c + t % KS) & 1) ? SWZP : 0;
#pragma unroll
                for (int i = 0; i < CIB; ++i) {
                    const float av = ldsP[pp * ROWP + ((i * 16 + l15) ^ sw)];
#pragma unroll
                    for (int j = 0; j < COBW; ++j)
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[t][i][j], 0, 0, 0);
                }
            }
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < COBW; ++j)
                    accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_one, bv[j], accb[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- write this wave's partial slab: D layout col (co) = lane&15, row (ci) = 4*(lane>>4)+j ----
    const size_t plane = (size_t)a.CinP * a.CoutP;
    float* sl = a.slab + (size_t)(blockIdx.x * WK + wk) * (KS * KS + 1) * plane;
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int i = 0; i < CIB; ++i)
#pragma unroll
            for (int j = 0; j < COBW; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = ci0 + i * 16 + g * 4 + e;
                    const int co = co0 + (wco * COBW + j) * 16 + l15;
                    sl[t * plane + (size_t)ci * a.CoutP + co] = acc[t][i][j][e];
                }
    if (do_bias && g == 0) {
#pragma unroll
        for (int j = 0; j < COBW; ++j) sl[(KS * KS) * plane + co0 + (wco * COBW + j) * 16 + l15] = accb[j][0];
    }
}

// dW[co][ci][ky][kx] = sum_s slab[s][tap][ci][co];  db[co] = sum_s slab[s][KS*KS][0][co]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           float* __restrict__ db, int nslab, int KS2, int Cin, int CinP,
                                                           int Cout, int CoutP) {
    __shared__ float red[4][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sl = threadIdx.x >> 6;
    const int nw = KS2 * Cin * Cout;
    const int total = nw + (db ? Cout : 0);
    float s = 0.f;
    size_t off = 0;
    bool live = o < total;
    if (live) {
        if (o < nw) {
            const int co = o % Cout;
            int rest = o / Cout;
            const int ci = rest % Cin;
            const int tap = rest / Cin;
            off = ((size_t)tap * CinP + ci) * CoutP + co;
        } else {
            off = (size_t)KS2 * CinP * CoutP + (o - nw);
        }
        const size_t stride = (size_t)(KS2 + 1) * CinP * CoutP;
        for (int k = sl; k < nslab; k += 4) s += slab[(size_t)k * stride + off];
    }
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && live) {
        s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (o < nw) {
            const int co = o % Cout;
            int rest = o / Cout;
            const int ci = rest % Cin;
            const int tap = rest / Cin;
            dw[((size_t)co * Cin + ci) * KS2 + tap] = s;
        } else {
            db[o - nw] = s;
        }
    }
}

template <int KS, int CIB, int COBW, int WCO>
static int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    constexpr int ROWP = CIB * 16, COT = COBW * WCO * 16;
    const int PP = (a.TH + KS - 1) * (a.TW + KS - 1), TP = a.TH * a.TW;
    const size_t shmem = ((size_t)PP * ROWP + (size_t)TP * COT) * sizeof(float);
    if (shmem > 160 * 1024) {
        aesr_set_error("conv_wgrad: tile needs %zu B of LDS", shmem);
        return AESR_ERR_ARG;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_f32<KS, CIB, COBW, WCO>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(a.S, a.CinP / ROWP, a.CoutP / COT);
    hipLaunchKernelGGL((conv_wgrad_f32<KS, CIB, COBW, WCO>), grid, dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wgrad_f32");
    return AESR_OK;
}

// variant: 0 -> 32 ci x 32 co per workgroup (WCO=2, WK=2); 1 -> 32 ci x 64 co (WCO=4, WK=1); 2 -> 16 ci x 16 co (WCO=1, WK=4)
int aesr_launch_conv_wgrad(const WgradArgs& a, int KS, int variant, hipStream_t st) {
    if (a.TW % 4 != 0) {
        aesr_set_error("conv_wgrad: TW=%d must be a multiple of 4", a.TW);
        return AESR_ERR_ARG;
    }
#define WG_CASE(ks, v, cib, cobw, wco) \
    if (KS == ks && variant == v) return launch_wgrad<ks, cib, cobw, wco>(a, st);
    WG_CASE(3, 0, 2, 1, 2) WG_CASE(3, 1, 2, 1, 4) WG_CASE(3, 2, 1, 1, 1)
    WG_CASE(1, 0, 2, 1, 2) WG_CASE(1, 1, 2, 1, 4) WG_CASE(1, 2, 1, 1, 1)
#undef WG_CASE
    aesr_set_error("conv_wgrad: no instantiation for KS=%d variant=%d", KS, variant);
    return AESR_ERR_UNSUPPORTED;
}

int aesr_launch_wgrad_reduce(const float* slab, float* dw, float* db, int nslab, int KS, int Cin, int CinP, int Cout,
                             int CoutP, hipStream_t st) {
    const int total = KS * KS * Cin * Cout + (db ? Cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div(total, 64)), dim3(256), 0, st, slab, dw, db, nslab, KS * KS, Cin,
                       CinP, Cout, CoutP);
    AESR_LAUNCH_CHECK("wgrad_reduce");
    return AESR_OK;
}
