// Weight gradient of a 3x3 / stride-1 / padding-1 convolution in Winograd F(2x2,3x3) form on the fp32 matrix cores, NHWC.
//
// Forward:  Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 2x2 output tile (conv_wino.hip).  Differentiating through it:
//       dM = A dY A^T (4x4 from the tile's 2x2 output gradient),   dU_xi[ci][co] = sum_tiles V_xi[ci][tile] * dM_xi[tile][co],
//       dg = G^T dU G (3x3)                                          with V = B^T d B the forward's transformed input tile.
// The 16 positions xi are 16 GEMMs with M = ci, N = co, K = tiles: 16 MFMA flops-blocks per tile instead of the 36 of the
// direct form (conv_wgrad.hip) -- 2.25x fewer matrix-core flops.
//
// What bounds it (scripts/micro/mfma_gap.hip, profiles/r02_micro_mfma_gap.txt): on gfx950 v_mfma_f32_16x16x4_f32 and the vector
// ALU do NOT overlap -- one wave per SIMD pays 32 cycles per MFMA plus ~4.3 per other instruction, wherever it stands -- so the
// kernel is built to issue FEW instructions besides its MFMAs:
//  * one 4-wave workgroup per CU, ONE wave per SIMD with the whole 512-register budget: a wave keeps the accumulators of all 16
//    positions for a 32 ci x 32 co block (256 accumulation registers, pinned by asm constraints) and takes every 4th k-step (a
//    k-step = 4 horizontally adjacent tiles = the K of one MFMA): one input transform feeds 32 MFMAs, one dM transform 32;
//  * both transforms run on register PAIRS with v_pk_add_f32 and its source-select / negate modifiers (two adds per instruction,
//    no shuffles: the LDS reads already deliver vertical pixel pairs, ds_read2st64_b32): 48 instructions per 64 MFMAs.  dM is
//    built with the sign-free matrix A' = [[1,0],[1,1],[1,-1],[0,1]] (A's last row negated); the signs (-1)^(i==3) (-1)^(j==3)
//    are applied once, in the epilogue.  The bias gradient is one of the dM sums (dY00 + dY01 + dY10 + dY11);
//  * the X patch and the dY tile go global -> LDS by DMA (buffer_load ... lds), pixel-major as they lie in memory
//    ([row][pixel][32 channels]); one DMA instruction of a wave = 8 pixels of ONE row, and a wave always fetches the same
//    8-pixel segment: everything per lane (channel slot, column bound, the folded Upsample's source column) is computed once per
//    tile, a DMA costs 4 instructions.  Rows outside the image are answered by the range check of a per-image buffer resource.
//    The DMA source swaps the two 16-channel halves of every other pixel pair (slot = quad ^ 4*((col >> 1) & 1)) so the lanes
//    of a half-wave (channel = lane & 15, tile = lane >> 4) read conflict free;
//  * tile v lives in LDS buffer v & 1.  The barrier of a tile stands before its LAST k-step: by then every wave has the last
//    operands it needs from the buffer in registers and tile v + 1 has landed, so the last k-step already fetches the first
//    operands of tile v + 1 and issues the DMA of tile v + 2 into the buffer just freed -- a whole k-step (> 2 000 cycles) before
//    anything waits for it.  No prologue per tile, no fetch burst;
//  * epilogue: dg = G^T dU G per lane (16 -> 9 values), the four waves' partial sums are added through LDS in a fixed order,
//    ONE slab per workgroup in the layout of conv_wgrad.hip ([split][tap | bias][ci][co]); wgrad_reduce_kernel sums the slabs
//    (bitwise reproducible, no atomics).
//
// Replaces autograd's conv2d weight-gradient for the 3x3 layers of networks/acai_vanilla.py:49-102 with >= 32 channels.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "aesr_kernels.h"

constexpr int WW_OOB = 0x70000000;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef WW_ABL
#define WW_ABL 0                // timing-only ablations of the tile loop (scripts/wgrad_ablation.sh): 1 no DMA, 2 no transforms, 3 no LDS reads, 4 MFMAs only
#endif

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int... I, class F>
__device__ __forceinline__ void ww_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ww_rep(F&& f) { ww_for(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ void ww_dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}

// v_pk_add_f32 forms (checked on the device by scripts/micro/pk_mods.hip); a = (a.lo, a.hi), b likewise
// (the operands are bound to locals first: clang rejects enclosing-function variables as asm operands inside nested generic lambdas)
#define WW_PK(dst, a, b, mods)                                                          \
    do {                                                                                \
        f32x2& d_ = dst;                                                                \
        const f32x2 a_ = a, b_ = b;                                                     \
        asm volatile("v_pk_add_f32 %0, %1, %2 " mods : "=v"(d_) : "v"(a_), "v"(b_));    \
    } while (0)
#define WW_M_ADD ""                                                                 /* (a.lo + b.lo, a.hi + b.hi) */
#define WW_M_SUB "neg_lo:[0,1] neg_hi:[0,1]"                                        /* (a.lo - b.lo, a.hi - b.hi) */
#define WW_M_T01 "op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]"           /* (a.lo - b.lo, a.hi + b.lo) */
#define WW_M_T23 "op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]"           /* (b.lo - a.hi, a.hi - b.hi) */
#define WW_M_LL "op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]"            /* (a.lo + b.lo, a.lo - b.lo) */
#define WW_M_HH "op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,0] neg_hi:[0,1]"            /* (a.hi + b.hi, a.hi - b.hi) */
#define WW_M_SELF "op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]"          /* with b = a: (a.lo + a.hi, a.lo - a.hi) */

// A spatial tile this workgroup visits; all uniform (SGPRs)
struct WwTile {
    int n, ty, tx;              // image, tile row, tile column
};

template <int TH, int TW, bool STAMP>
__global__ __launch_bounds__(256, 1) void conv_wgrad_wino_f32(WgradArgs a) {
    constexpr int CIT = 32, COT = 32;
    static_assert((TH == 16 && TW == 8) || (TH == 8 && TW == 16), "tiles: 8 k-steps, rows of whole 8-pixel DMA segments");
    constexpr int PH = TH + 2, PWL = TW == 8 ? 16 : 32, TWL = TW;     // LDS row strides in pixels
    constexpr int XFL = PH * PWL * 32, DFL = TH * TWL * 32;            // floats of one X / dY buffer
    constexpr int KPR = TW >> 3;                                        // k-steps per tile row; 8 per tile, 2 per wave
    constexpr int SPR = PWL / 8, SPRD = TW / 8;                         // 8-pixel DMA segments per X / dY row
    constexpr int NXS = PH * SPR / 4, NDS = TH * SPRD / 4;              // segments a wave fetches per tile
    constexpr int RXS = 4 / SPR, RDS = 4 / SPRD;                        // ... every RXS-th / RDS-th row
    static_assert(2 * (XFL + DFL) * 4 <= 160 * 1024 && (PH * SPR) % 4 == 0 && (TH * SPRD) % 4 == 0, "LDS / segment split");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // workgroup -> (ci chunk, co chunk, split); the chunks that walk the same pixel tiles sit on one XCD back to back
    const int nci = a.CinP / CIT, nchunks = nci * (a.CoutP / COT);
    int chunk, split;
    if ((a.S & 7) == 0) {
        const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        chunk = local % nchunks;
        split = (local / nchunks) * 8 + xcd;
    } else {
        chunk = blockIdx.x % nchunks;
        split = blockIdx.x / nchunks;
    }
    const int ciy = chunk % nci, coz = chunk / nci;
    const int ci0 = ciy * CIT, co0 = coz * COT;
    float* const ldsX0 = lds;                        // [2][PH][PWL][32]
    float* const ldsD0 = lds + 2 * XFL;              // [2][TH][TWL][32]
    const bool do_bias = ciy == 0;

    // ---- DMA.  Wave w always fetches segment w % SPR of rows w / SPR, + RXS, ... of the X patch (segment w % SPRD of rows w / SPRD,
    // + RDS, ... of the dY tile): per lane only the row changes between its DMAs ----
    const int sh = a.x_up2 ? 1 : 0;                                     // nearest Upsample x2 folded in: x is stored at half size, pixel (y, x) <- (y/2, x/2)
    const int xH = a.H >> sh, xW = a.W >> sh;
    const int ximg = xH * xW * a.Cin * 4, dimg = a.H * a.W * a.Cout * 4; // bytes of one image
    const int xrow = xW * a.Cin * 4, drow = a.W * a.Cout * 4;            // ... of one stored row
    const int sgx = wave % SPR, rx0 = wave / SPR, sgd = wave % SPRD, rd0 = wave / SPRD;
    const int pxl = lane >> 3, pos = lane & 7;
    const int quad = pos ^ (((pxl >> 1) & 1) << 2);                     // channel quad ^ 4 * ((col >> 1) & 1): 8 sg does not touch bit 1 of the column
    const int pcx = 8 * sgx + pxl, pcd = 8 * sgd + pxl;                 // this lane's patch / tile column
    const int lcx = (((pcx - sh) >> sh) * a.Cin + ci0 + 4 * quad) * 4;  // byte offset of the lane's 16 B relative to (row start + tile column origin)
    const int lcd = (pcd * a.Cout + co0 + 4 * quad) * 4;
    const int tpi = a.tiles_y * a.tiles_x;
    const int dS_n = a.S / tpi, dS_r = a.S - dS_n * tpi, dS_ty = dS_r / a.tiles_x, dS_tx = dS_r - dS_ty * a.tiles_x;
    auto advance = [&](WwTile t) {
        t.tx += dS_tx;
        if (t.tx >= a.tiles_x) { t.tx -= a.tiles_x; ++t.ty; }
        t.ty += dS_ty;
        if (t.ty >= a.tiles_y) { t.ty -= a.tiles_y; ++t.n; }
        t.n += dS_n;
        return t;
    };
    // all DMAs of one tile: per-IMAGE buffer resources (rows above / below the image fall outside their range: zero, no fetch;
    // a tile past the last one gets an empty range: its buffer is zero-filled, so whatever is computed from it adds nothing)
    auto fetch = [&](const WwTile& t, int buf) {
        const bool valid = t.n < a.N;
        const int n = valid ? t.n : 0, y0 = t.ty * TH, x0 = t.tx * TW;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x + (size_t)n * ximg), 0, valid ? ximg : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.dy + (size_t)n * dimg), 0, valid ? dimg : 0, 0x00020000);
        // the lane's offset inside a row, or the out-of-range marker: patch column beyond the patch or outside the image
        const unsigned gxx = (unsigned)(x0 - 1 + pcx), gxd = (unsigned)(x0 + pcd);
        const int offx = (pcx < TW + 2 && gxx < (unsigned)a.W) ? lcx + ((x0 >> sh) - 1 + sh) * a.Cin * 4 : WW_OOB;
        const int offd = gxd < (unsigned)a.W ? lcd + x0 * a.Cout * 4 : WW_OOB;
        // X rows y0 - 1 + rx0 + RXS k (source row (.) >> 1 under the folded Upsample): a running offset, two alternating steps
        const int r0 = y0 - 1 + rx0;
        int ux = (r0 >> sh) * xrow;                  // arithmetic shift: row -1 stays -1
        const int step_odd = sh == 0 ? RXS * xrow : RXS == 2 ? xrow : (r0 & 1) * xrow;      // step into an odd k
        const int step_even = sh == 0 ? RXS * xrow : RXS == 2 ? xrow : xrow - step_odd;
        float* const xdst = ldsX0 + buf * XFL + (rx0 * PWL + 8 * sgx) * 32;
        int uxk[NXS];
        ww_rep<NXS>([&](auto Kc) {
            constexpr int k = decltype(Kc)::value;
            if constexpr (k > 0) ux += (k & 1) ? step_odd : step_even;
            uxk[k] = ux;
        });
        if (pcx < TW + 2) {         // lanes past the patch's last column take no part: a DMA of 16 lanes issues faster than one of 64
            ww_rep<NXS>([&](auto Kc) {
                constexpr int k = decltype(Kc)::value;
                ww_dma(rs_x, xdst + k * RXS * PWL * 32, offx + uxk[k]);
            });
        }
        int ud = (y0 + rd0) * drow;
        float* const ddst = ldsD0 + buf * DFL + (rd0 * TWL + 8 * sgd) * 32;
        ww_rep<NDS>([&](auto Kc) {
            constexpr int k = decltype(Kc)::value;
            if constexpr (k > 0) ud += RDS * drow;
            ww_dma(rs_d, ddst + k * RDS * TWL * 32, offd + ud);
        });
    };

    f32x4 acc[16][2][2];                    // [position][ci block][co block]: D rows = ci 4g..4g+3, column = co l15
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int o = 0; o < 2; ++o) acc[x][i][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accb[2] = {0.f, 0.f};

    // this lane's channel inside a pixel: quad l15 >> 2 (+ 4 for the second 16-channel block), element l15 & 3; the swizzle swaps
    // the blocks for odd tile pairs: position = quad ^ 4 * ((col >> 1) & 1), col = 2 * tile + j -> parity of (tile + (j >> 1))
    const int sw = (g & 1) << 2;                                   // the tile index of a k-step is 4 * kx + g: its parity is g's
    const int chA = (((l15 >> 2) ^ sw) << 2) + (l15 & 3);          // float offset of block 0's channel in pixels with (col >> 1) even
    const int chB = (((l15 >> 2) ^ sw ^ 4) << 2) + (l15 & 3);      // ... with (col >> 1) odd
    // LDS byte addresses (without the k-step's uniform part) of patch pixel (0, c) of the lane's tile (2 g pixels to the right),
    // block i, and of dY pixel (0, e): columns 0,1 share (col >> 1) parity, 2,3 flip it
    const unsigned lds0 = (unsigned)(unsigned long long)lds;       // the LDS offset is the low half of the flat address
    unsigned lx[2][4], ld[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) lx[i][c] = lds0 + (unsigned)(((2 * g + c) * 32 + ((c < 2 ? chA : chB) ^ (i ? 16 : 0))) * 4);
#pragma unroll
        for (int e = 0; e < 2; ++e) ld[i][e] = lds0 + (unsigned)((2 * XFL + (2 * g + e) * 32 + (chA ^ (i ? 16 : 0))) * 4);
    }

    // operands of a k-step, as register pairs.  V = B^T d B of block i: Vp[i][s][k] = (V[2s][k], V[2s+1][k]) -> positions
    // 8 s + k and 8 s + 4 + k.  M' = A' dY A'^T of block o: Mp[o][0..3] = Ya = (y00, y10), Yb = (y01, y11), Ra = (y00 + y10,
    // y00 - y10), Rb likewise; Mp[o][4 + r] = (row r: a + b, a - b) with rows (a, b) = (Ya.lo, Yb.lo), (Ra.lo, Rb.lo),
    // (Ra.hi, Rb.hi), (Ya.hi, Yb.hi).  M'[4 r + 0..3] = a, a + b, a - b, b.
    f32x2 Vp[2][2][2][4], Mp[2][2][8];      // [buffer][block]...
    f32x2 P[2][4][2];                       // raw patch pairs [block][column][(d0, d1) | (d2, d3)]
    // LDS reads of k-step ks = wave + 4 j of the tile in buffer `buf` (uniform float offsets xu / du): 20 ds_read2st64_b32
    auto op_read = [&](int xu, int du, int nb) {
        ww_rep<8>([&](auto Kc) {
            constexpr int i = decltype(Kc)::value >> 2, c = decltype(Kc)::value & 3;
            const unsigned ad = lx[i][c] + (unsigned)(xu * 4);
            f32x2 &p0_ = P[i][c][0], &p1_ = P[i][c][1];
            asm volatile("ds_read2st64_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read2st64_b32 %1, %2 offset0:%5 offset1:%6"
                         : "=&v"(p0_), "=&v"(p1_) : "v"(ad), "n"(0), "n"(PWL / 2), "n"(PWL), "n"(3 * PWL / 2) : "memory");
        });
        ww_rep<4>([&](auto Kc) {
            constexpr int o = decltype(Kc)::value >> 1, e = decltype(Kc)::value & 1;
            const unsigned ad = ld[o][e] + (unsigned)(du * 4);
            f32x2& y_ = Mp[nb][o][e];
            asm volatile("ds_read2st64_b32 %0, %1 offset0:0 offset1:%2" : "=&v"(y_) : "v"(ad), "n"(TWL / 2) : "memory");
        });
    };
    // both transforms of the operands read: 48 v_pk_add_f32 + 2 adds for the bias gradient
    auto op_transform = [&](int nb) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ww_rep<2>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            f32x2 T[2][4];                  // [rows 0,1 | rows 2,3][column]
            ww_rep<4>([&](auto Cc) {
                constexpr int c = decltype(Cc)::value;
                WW_PK(T[0][c], P[i][c][0], P[i][c][1], WW_M_T01);      // (d0 - d2, d1 + d2)
                WW_PK(T[1][c], P[i][c][0], P[i][c][1], WW_M_T23);      // (d2 - d1, d1 - d3)
            });
            ww_rep<2>([&](auto Sc) {
                constexpr int s_ = decltype(Sc)::value;
                WW_PK(Vp[nb][i][s_][0], T[s_][0], T[s_][2], WW_M_SUB);
                WW_PK(Vp[nb][i][s_][1], T[s_][1], T[s_][2], WW_M_ADD);
                WW_PK(Vp[nb][i][s_][2], T[s_][2], T[s_][1], WW_M_SUB);
                WW_PK(Vp[nb][i][s_][3], T[s_][1], T[s_][3], WW_M_SUB);
            });
        });
        ww_rep<2>([&](auto Oc) {
            constexpr int o = decltype(Oc)::value;
            WW_PK(Mp[nb][o][2], Mp[nb][o][0], Mp[nb][o][0], WW_M_SELF);
            WW_PK(Mp[nb][o][3], Mp[nb][o][1], Mp[nb][o][1], WW_M_SELF);
            WW_PK(Mp[nb][o][4], Mp[nb][o][0], Mp[nb][o][1], WW_M_LL);
            WW_PK(Mp[nb][o][7], Mp[nb][o][0], Mp[nb][o][1], WW_M_HH);
            WW_PK(Mp[nb][o][5], Mp[nb][o][2], Mp[nb][o][3], WW_M_LL);
            WW_PK(Mp[nb][o][6], Mp[nb][o][2], Mp[nb][o][3], WW_M_HH);
            accb[o] += Mp[nb][o][5][0];     // (y00 + y10) + (y01 + y11)
        });
    };
    // MFMAs [Q0, Q0 + N) of the 64 of a k-step: A = V (M = ci), B = M' (N = co), K = the 4 tiles of the k-step.  The accumulators
    // are pinned to the accumulation registers ("+a"): left to itself the allocator moves parts of them through the (full) vector
    // registers and scratch.  Operands are a k-step old and an accumulator is touched once per k-step: no MFMA hazard is near.
    auto mfmas = [&](auto Q0c, auto Nc, int cb) {
        ww_rep<decltype(Nc)::value>([&](auto Dc) {
            constexpr int q = decltype(Q0c)::value + decltype(Dc)::value, x = q >> 2, i = (q >> 1) & 1, o = q & 1;
            constexpr int r = x >> 2, k = x & 3;
            constexpr int mi = k == 0 ? (r == 0 || r == 3 ? 0 : 2) : k == 3 ? (r == 0 || r == 3 ? 1 : 3) : 4 + r;
            constexpr int mh = k == 0 || k == 3 ? (r >= 2 ? 1 : 0) : k - 1;
            f32x4& acc_ = acc[x][i][o];
            const float va_ = Vp[cb][i][r >> 1][k][r & 1], vb_ = Mp[cb][o][mi][mh];
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(va_), "v"(vb_));
        });
    };
    // uniform LDS float offsets of k-step ks = wave + 4 j: tile (tyl, 4 kx + g) has its top-left patch pixel at (2 tyl, 8 kx + 2 g)
    auto step_xu = [&](int j, int buf) { const int ks = wave + 4 * j, tyl = ks / KPR, kx = ks - tyl * KPR; return buf * XFL + ((2 * tyl) * PWL + 8 * kx) * 32; };
    auto step_du = [&](int j, int buf) { const int ks = wave + 4 * j, tyl = ks / KPR, kx = ks - tyl * KPR; return buf * DFL + ((2 * tyl) * TWL + 8 * kx) * 32; };
    using C0 = std::integral_constant<int, 0>;
    using C16 = std::integral_constant<int, 16>;
    using C32 = std::integral_constant<int, 32>;

    // debug instantiation of the same loop (STAMP, AESR_WGRAD_WINO_DBG=1): cycles per wave in the k-step loop and at the barrier
    long long tph[3] = {0, 0, 0}, tq = 0;
#define WW_STAMP(k)                                                   \
    if constexpr (STAMP) {                                            \
        const long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
        tph[k] += t_ - tq;                                            \
        tq = t_;                                                      \
    }
    if constexpr (STAMP) tq = (long long)__builtin_amdgcn_s_memtime();

    WwTile t0, t1, t2;
    {
        t0.n = split / tpi;
        const int rem = split - t0.n * tpi;
        t0.ty = rem / a.tiles_x;
        t0.tx = rem - t0.ty * a.tiles_x;
        if (split >= a.ntiles) t0.n = a.N;
    }
    t1 = advance(t0);
    t2 = advance(t1);
    fetch(t0, 0);
    fetch(t1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    op_read(step_xu(0, 0), step_du(0, 0), 0);
    op_transform(0);
    WW_STAMP(2)
    if (t0.n < a.N) do {
        // k-step 0 of the tile: MFMAs on operand set 0, the operands of k-step 1 into set 1
        __builtin_amdgcn_sched_barrier(0);
#if WW_ABL != 3 && WW_ABL != 4
        op_read(step_xu(1, buf), step_du(1, buf), 1);
#endif
        mfmas(C0{}, C16{}, 0);
#if WW_ABL != 2 && WW_ABL != 4
        op_transform(1);
#endif
        mfmas(C16{}, std::integral_constant<int, 48>{}, 0);
        __builtin_amdgcn_sched_barrier(0);
        WW_STAMP(0)
        // The wait is OURS to place: the LDS reads of this kernel are asm statements the compiler does not see, so its own
        // bookkeeping finds no reader of the DMA'd bytes and leaves __syncthreads() a bare s_barrier -- a wave then read tile
        // v + 1 before another wave's DMA had landed (rare wrong tiles at 36 x 162 x 162, never at the small test sizes).
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();            // tile v + 1 has landed; nobody reads buffer `buf` any more
        WW_STAMP(1)
        // k-step 1: MFMAs on set 1, the first operands of the NEXT tile into set 0, the DMA of the tile after it into `buf`
        __builtin_amdgcn_sched_barrier(0);
#if WW_ABL != 3 && WW_ABL != 4
        op_read(step_xu(0, buf ^ 1), step_du(0, buf ^ 1), 0);
#endif
        mfmas(C0{}, C16{}, 1);
        __builtin_amdgcn_sched_barrier(0);
#if WW_ABL != 1 && WW_ABL != 4
        fetch(t2, buf);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfmas(C16{}, C16{}, 1);
#if WW_ABL != 2 && WW_ABL != 4
        op_transform(0);
#endif
        mfmas(C32{}, C32{}, 1);
        __builtin_amdgcn_sched_barrier(0);
        t0 = t1;
        t1 = t2;
        t2 = advance(t2);
        buf ^= 1;
    } while (t0.n < a.N);
    WW_STAMP(0)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();            // DMAs still in flight (zero fills past the last tile) must land before the exchange reuses the LDS
    if (STAMP && lane == 0)
        for (int k = 0; k < 3; ++k) a.dbgbuf[(blockIdx.x * 4 + wave) * 3 + k] = (float)tph[k];

    // ---- epilogue: signs of A, dg = G^T dU G, sum of the four waves through LDS, ONE slab per workgroup ----
    // G^T = [[1, 1/2, 1/2, 0], [0, 1/2, -1/2, 0], [0, 1/2, 1/2, 1]]
    f32x4 dg[9][2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            f32x4 R[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 u0_ = acc[0 + j][i][o], u1 = acc[4 + j][i][o], u2 = acc[8 + j][i][o], u3 = -acc[12 + j][i][o];
                const f32x4 s = j == 3 ? (f32x4){-1.f, -1.f, -1.f, -1.f} : (f32x4){1.f, 1.f, 1.f, 1.f};
                R[0][j] = s * (u0_ + 0.5f * (u1 + u2));
                R[1][j] = s * (0.5f * (u1 - u2));
                R[2][j] = s * (0.5f * (u1 + u2) + u3);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                dg[r * 3 + 0][i][o] = R[r][0] + 0.5f * (R[r][1] + R[r][2]);
                dg[r * 3 + 1][i][o] = 0.5f * (R[r][1] - R[r][2]);
                dg[r * 3 + 2][i][o] = 0.5f * (R[r][1] + R[r][2]) + R[r][3];
            }
        }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        accb[o] += __shfl_xor(accb[o], 16, 64);
        accb[o] += __shfl_xor(accb[o], 32, 64);
    }
    // the planes are dead (barrier at the end of the last tile): every wave parks its partial sums in LDS and then adds up and stores a
    // QUARTER of the slab (9 of the 36 [tap][ci block][co block] pieces) -- waves 0..3 in the same fixed order as before (wave 0 alone
    // used to read 108 KB and issue all 144 stores while the other three waves idled)
    f32x4* ex = (f32x4*)lds;                               // [wave][36][64 lanes]
    float* exb = lds + 4 * 36 * 64 * 4;                    // [wave][2][16]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int o = 0; o < 2; ++o) ex[(wave * 36 + (t * 4 + i * 2 + o)) * 64 + lane] = dg[t][i][o];
    if (g == 0) {
        exb[wave * 32 + l15] = accb[0];
        exb[wave * 32 + 16 + l15] = accb[1];
    }
    __syncthreads();
    const size_t plane = (size_t)a.CinP * a.CoutP;
    float* sl = a.slab + (size_t)split * 10 * plane;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int blk = wave * 9 + q;                      // uniform: piece (t, i, o)
        const int t = blk >> 2, i = (blk >> 1) & 1, o = blk & 1;
        f32x4 v = ex[(0 * 36 + blk) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += ex[(w * 36 + blk) * 64 + lane];
        const int co = co0 + o * 16 + l15;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = ci0 + i * 16 + g * 4 + e;
            sl[t * plane + (size_t)ci * a.CoutP + co] = v[e];
        }
    }
    if (wave == 0 && do_bias && g == 0) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float bsum = exb[0 * 32 + o * 16 + l15];
#pragma unroll
            for (int w = 1; w < 4; ++w) bsum += exb[w * 32 + o * 16 + l15];
            sl[9 * plane + co0 + o * 16 + l15] = bsum;
        }
    }
}

size_t aesr_wgrad_wino_lds_bytes(int TH, int TW) {
    const size_t xfl = (size_t)(TH + 2) * (TW == 8 ? 16 : 32) * 32, dfl = (size_t)TH * TW * 32;
    const size_t bufs = 2 * (xfl + dfl) * sizeof(float);
    const size_t exch = ((size_t)4 * 36 * 64 * 4 + 4 * 32) * sizeof(float);
    return bufs > exch ? bufs : exch;
}

// the tiles the kernel is built for: 8 k-steps ((TH / 2) * (TW / 8)), rows of whole 8-pixel DMA segments split evenly over 4 waves
bool aesr_wgrad_wino_tile_ok(int TH, int TW) { return (TH == 16 && TW == 8) || (TH == 8 && TW == 16); }

template <int TH, int TW>
static int launch_wgrad_wino(const WgradArgs& a, hipStream_t st) {
    const size_t shmem = aesr_wgrad_wino_lds_bytes(TH, TW);
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wgrad_wino_f32<TH, TW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ == hipSuccess) (void)hipFuncSetAttribute((const void*)conv_wgrad_wino_f32<TH, TW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wgrad_wino_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    dim3 grid(a.S * (a.CinP / 32) * (a.CoutP / 32));
    if (getenv("AESR_WGRAD_WINO_DBG") && grid.x <= 4096) {     // debug: per-phase cycle stamps, printed after a host sync
        static float* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 4 * 3 * sizeof(float));
        WgradArgs b = a;
        b.dbgbuf = dbuf;
        hipLaunchKernelGGL((conv_wgrad_wino_f32<TH, TW, true>), grid, dim3(256), shmem, st, b);
        (void)hipStreamSynchronize(st);
        static float host[4096 * 4 * 3];
        (void)hipMemcpy(host, dbuf, (size_t)grid.x * 4 * 3 * sizeof(float), hipMemcpyDeviceToHost);
        double s3[3] = {0, 0, 0};
        for (unsigned i = 0; i < grid.x * 4; ++i) for (int k = 0; k < 3; ++k) s3[k] += host[i * 3 + k];
        const double visits = (double)((a.ntiles + a.S - 1) / a.S);
        const int nks = (TH >> 1) * (TW >> 3);
        fprintf(stderr, "[wgrad-wino stamps] grid=%u tile %dx%d S=%d (%.0f visits, %d k-steps = %d MFMA cycles per wave and visit) per visit, cycles: "
                "k-step loop %.0f | barrier %.0f || fill %.0f\n", grid.x, TH, TW, a.S, visits, nks, nks / 4 * 64 * 32,
                s3[0] / grid.x / 4 / visits, s3[1] / grid.x / 4 / visits, s3[2] / grid.x / 4);
        return AESR_OK;
    }
    hipLaunchKernelGGL((conv_wgrad_wino_f32<TH, TW, false>), grid, dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wgrad_wino_f32");
    return AESR_OK;
}

int aesr_launch_conv_wgrad_wino(const WgradArgs& a, hipStream_t st) {
    // a lane's byte offset inside ONE image is 32-bit and shares its range with the out-of-range marker
    if ((size_t)a.H * a.W * a.Cin * 4 >= (size_t)0x10000000 || (size_t)a.H * a.W * a.Cout * 4 >= (size_t)0x10000000) {
        aesr_set_error("conv_wgrad_wino: images of 256 MB or more need 64-bit lane offsets (not built)");
        return AESR_ERR_UNSUPPORTED;
    }
    if (a.x_up2 && ((a.H | a.W) & 1)) {
        aesr_set_error("conv_wgrad_wino: the folded Upsample(x2) needs even convolution sizes (got %dx%d)", a.H, a.W);
        return AESR_ERR_ARG;
    }
    if (!aesr_wgrad_wino_tile_ok(a.TH, a.TW) || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) {
        aesr_set_error("conv_wgrad_wino: tile %dx%d is not 16x8 or 8x16 (or not 3x3 / padding 1)", a.TH, a.TW);
        return AESR_ERR_ARG;
    }
    if (a.CinP % 32 != 0 || a.CoutP % 32 != 0 || a.Cin != a.CinP || a.Cout != a.CoutP || a.S < 1 || a.S > a.ntiles ||
        a.tiles_y != ceil_div(a.H, a.TH) || a.tiles_x != ceil_div(a.W, a.TW) || a.ntiles != a.N * a.tiles_y * a.tiles_x) {
        aesr_set_error("conv_wgrad_wino: channel counts must be multiples of 32 and the tile grid consistent (S=%d, %d tiles)", a.S, a.ntiles);
        return AESR_ERR_ARG;
    }
    if (a.TH == 16) return launch_wgrad_wino<16, 8>(a, st);
    return launch_wgrad_wino<8, 16>(a, st);
}
