// Weight gradient of a stride-1 convolution on the fp32 matrix cores, NHWC.
//
//   dW[co,ci,ky,kx] = sum_{n,y,x} dY[n,y,x,co] * X[n,y+ky-pad,x+kx-pad,ci]      db[co] = sum dY[n,y,x,co]
//
// GEMM view per tap: M = ci, N = co, K = pixels (hundreds of thousands) -> a split-K design:
//  * grid = (S splits, Cin chunks of 32, Cout chunks of 32 or 64); each workgroup walks the pixel tiles s, s+S, ... and
//    keeps its whole accumulator set (all KS*KS taps) in registers for the entire walk; the only global writes are ONE
//    partial slab per workgroup at the end (the 4 waves own disjoint (ci-block, co-block) sets: no cross-wave reduction);
//  * LDS holds the X patch (tile + halo) and the dY tile CHANNEL-MAJOR ([channel][row][col], plane stride = 4 mod 64
//    floats): the MFMA k index (lane>>4) walks pixels, and one lane reads 4 consecutive pixels of its channel with two
//    aligned ds_read_b64 -> 2 pixel sub-steps x 3 horizontal taps = 6 MFMAs per read pair, conflict-free, no per-tap
//    address arithmetic and no division inside the loop;
//  * the bias gradient rides along as one extra MFMA per sub-step with A = e_0 (row 0 of that block = column sums of dY),
//    computed by the ci-chunk-0 workgroups only;
//  * a second kernel sums the slabs in a fixed order (bitwise reproducible, no atomics) and writes dW in the PyTorch
//    [Cout][Cin][KS][KS] layout.
//
// Replaces autograd's conv2d weight-gradient for the layers of networks/acai_vanilla.py:49-102.
#include "aesr_kernels.h"

// WG covers 32 ci x (16*NWCO) co.  NWCO == 2: wave = (ci block, co block), 1 ci block per wave;
// NWCO == 4: wave = co block, 2 ci blocks per wave.
template <int KS, int NWCO>
__global__ __launch_bounds__(256) void conv_wgrad_f32(WgradArgs a) {
    constexpr int CIBW = (NWCO == 2) ? 1 : 2;        // ci blocks per wave
    constexpr int CIT = 32, COT = 16 * NWCO;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int wco = (NWCO == 2) ? (wave >> 1) : wave;
    const int wci = (NWCO == 2) ? (wave & 1) : 0;
    const int ci0 = blockIdx.y * CIT, co0 = blockIdx.z * COT;
    const int PH = a.TH + KS - 1;
    const int PWS = a.PWS, TWS = a.TWS, PSX = a.PSX, PSD = a.PSD;      // row / plane strides (floats), host-chosen
    float* ldsX = lds;                      // [CIT][PSX]
    float* ldsD = lds + CIT * PSX;          // [COT][PSD]
    const bool do_bias = (blockIdx.y == 0) && (wci == 0);

    f32x4 acc[KS * KS][CIBW];
    f32x4 accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int i = 0; i < CIBW; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a_one = (l15 == 0) ? 1.f : 0.f;

    const int tpi = a.tiles_y * a.tiles_x;
    const int PWp = a.TW + KS - 1;          // patch pixels per row actually staged
    const int PP = PH * PWp, TP = a.TH * a.TW;
    // this lane's fixed offsets: X plane of (ci block, l15) + 2g ; dY plane of (co block, l15) + 2g
    const int xbase0 = ((wci * CIBW) * 16 + l15) * PSX + 2 * g;
    const int dbase = (wco * 16 + l15) * PSD + 2 * g;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += a.S) {
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        // ---- stage X patch, transposing to channel-major ----
        for (int q = tid; q < PP * (CIT / 4); q += 256) {
            const int p = q >> 3, part = q & 7;                  // CIT/4 == 8 pieces per pixel
            const int pr = p / PWp, pc = p - pr * PWp;
            const int gy = y0 + pr - a.pad, gx = x0 + pc - a.pad, ci = ci0 + part * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci < a.Cin)
                v = *(const f32x4*)(a.x + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + ci);
            float* d = ldsX + (part * 4) * PSX + pr * PWS + pc;
            d[0] = v[0];
            d[PSX] = v[1];
            d[2 * PSX] = v[2];
            d[3 * PSX] = v[3];
        }
        // ---- stage dY tile, transposing to channel-major ----
        for (int q = tid; q < TP * (COT / 4); q += 256) {
            const int p = q / (COT / 4), part = q - p * (COT / 4);
            const int r = p / a.TW, c = p - r * a.TW;
            const int gy = y0 + r, gx = x0 + c, co = co0 + part * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (gy < a.Ho && gx < a.Wo && co < a.Cout)
                v = *(const f32x4*)(a.dy + (((size_t)n * a.Ho + gy) * a.Wo + gx) * a.Cout + co);
            float* d = ldsD + (part * 4) * PSD + r * TWS + c;
            d[0] = v[0];
            d[PSD] = v[1];
            d[2 * PSD] = v[2];
            d[3 * PSD] = v[3];
        }
        __syncthreads();
        // ---- k loop: groups of 8 consecutive pixels of one row; lane (.,g) owns pixels c0+2g, c0+2g+1 (+ halo) ----
        for (int r = 0; r < a.TH; ++r) {
            for (int c0 = 0; c0 < a.TW; c0 += 8) {
                const float2 dv = *(const float2*)(ldsD + dbase + r * TWS + c0);
#pragma unroll
                for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
                    for (int i = 0; i < CIBW; ++i) {
                        const float* xp = ldsX + xbase0 + i * 16 * PSX + (r + ky) * PWS + c0;
                        float e[4];
                        const float2 e01 = *(const float2*)xp;
                        e[0] = e01.x;
                        e[1] = e01.y;
                        if (KS > 1) {
                            const float2 e23 = *(const float2*)(xp + 2);
                            e[2] = e23.x;
                            e[3] = e23.y;
                        }
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx)      // pixel sub-step 0, then sub-step 1: same accumulator 3 MFMAs apart
                            acc[ky * KS + kx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[kx], dv.x, acc[ky * KS + kx][i], 0, 0, 0);
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx)
                            acc[ky * KS + kx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[kx + 1], dv.y, acc[ky * KS + kx][i], 0, 0, 0);
                    }
                    if (do_bias && ky == 0) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(a_one, dv.x, accb, 0, 0, 0);
                    if (do_bias && ky == KS - 1) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(a_one, dv.y, accb, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- write the partial slab: D layout col (co) = lane&15, row (ci) = 4*(lane>>4)+j ----
    const size_t plane = (size_t)a.CinP * a.CoutP;
    float* sl = a.slab + (size_t)blockIdx.x * (KS * KS + 1) * plane;
    const int co = co0 + wco * 16 + l15;
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int i = 0; i < CIBW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = ci0 + (wci * CIBW + i) * 16 + g * 4 + e;
                sl[t * plane + (size_t)ci * a.CoutP + co] = acc[t][i][e];
            }
    if (do_bias && g == 0) sl[(KS * KS) * plane + co] = accb[0];
}

// dW[co][ci][ky][kx] = sum_s slab[s][tap][ci][co];  db[co] = sum_s slab[s][KS*KS][0][co]
// block = 64 outputs x 4 slab-lanes, fixed summation order
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
    const bool live = o < total;
    int co = 0, ci = 0, tap = 0;
    if (live) {
        if (o < nw) {
            co = o % Cout;
            const int rest = o / Cout;
            ci = rest % Cin;
            tap = rest / Cin;
            off = ((size_t)tap * CinP + ci) * CoutP + co;
        } else {
            off = (size_t)KS2 * CinP * CoutP + (o - nw);
        }
        const size_t stride = (size_t)(KS2 + 1) * CinP * CoutP;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = sl;
        for (; k + 12 < nslab; k += 16) {
            s0 += slab[(size_t)k * stride + off];
            s1 += slab[(size_t)(k + 4) * stride + off];
            s2 += slab[(size_t)(k + 8) * stride + off];
            s3 += slab[(size_t)(k + 12) * stride + off];
        }
        for (; k < nslab; k += 4) s0 += slab[(size_t)k * stride + off];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && live) {
        s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (o < nw) dw[((size_t)co * Cin + ci) * KS2 + tap] = s;
        else db[o - nw] = s;
    }
}

template <int KS, int NWCO>
static int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    constexpr int COT = 16 * NWCO;
    const size_t shmem = ((size_t)32 * a.PSX + (size_t)COT * a.PSD) * sizeof(float);
    if (shmem > 160 * 1024) {
        aesr_set_error("conv_wgrad: tile needs %zu B of LDS", shmem);
        return AESR_ERR_ARG;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_f32<KS, NWCO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(a.S, a.CinP / 32, a.CoutP / COT);
    hipLaunchKernelGGL((conv_wgrad_f32<KS, NWCO>), grid, dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wgrad_f32");
    return AESR_OK;
}

// variant: 0 -> 32 ci x 32 co per workgroup; 1 -> 32 ci x 64 co
int aesr_launch_conv_wgrad(const WgradArgs& a, int KS, int variant, hipStream_t st) {
    if (a.TW % 8 != 0 || (a.PWS & 1) || (a.TWS & 1) || (a.PSX & 1) || (a.PSD & 1)) {
        aesr_set_error("conv_wgrad: TW=%d must be a multiple of 8 and strides even", a.TW);
        return AESR_ERR_ARG;
    }
#define WG_CASE(ks, v, nwco) \
    if (KS == ks && variant == v) return launch_wgrad<ks, nwco>(a, st);
    WG_CASE(3, 0, 2) WG_CASE(3, 1, 4) WG_CASE(1, 0, 2) WG_CASE(1, 1, 4)
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
