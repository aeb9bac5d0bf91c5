// Weight gradient of a 3x3 / stride-1 / padding-1 convolution in Winograd F(2x2,3x3) form on the fp32 matrix cores, NHWC.
//
// Forward:  Y = A^T [ (G g G^T) .* (B^T d B) ] A   per 2x2 output tile (conv_wino.hip).  Differentiating through it:
//       dM = A dY A^T (4x4 from the tile's 2x2 output gradient),   dU_xi[ci][co] = sum_tiles V_xi[ci][tile] * dM_xi[tile][co],
//       dg = G^T dU G (3x3)                                          with V = B^T d B the forward's transformed input tile.
// The 16 positions xi are 16 GEMMs with M = ci, N = co, K = tiles: 16 MFMA flops-blocks per tile instead of the 36 of the
// direct form (conv_wgrad.hip) -- 2.25x fewer matrix-core flops.
//
//  * one 4-wave workgroup per CU, ONE wave per SIMD with the whole 512-register budget: a wave keeps the accumulators of all 16
//    positions for a 32 ci x 32 co block (256 accumulator registers) and takes every 4th k-step (a k-step = 4 horizontally
//    adjacent tiles = the K of one MFMA); with 2 ci x 2 co blocks per wave one input transform feeds 32 MFMAs and one dM
//    transform 32: ~2 other instructions per MFMA, produced one k-step ahead in the shadow of the previous step's 64 MFMAs;
//  * the X patch and the dY tile go global -> LDS by DMA (buffer_load ... lds), pixel-major as they lie in memory
//    ([pixel][32 channels]), double buffered: the next spatial tile lands while this one is consumed, ONE barrier per tile, no
//    staging registers and no transposing pass.  Lane (channel = lane & 15, tile = lane >> 4) reads single floats: two lanes of
//    a half-wave that differ in the tile would hit the same 16 banks, so the DMA source swaps the two 16-channel halves of every
//    other pixel pair (slot = quad ^ 4*((col >> 1) & 1)): tiles t and t + 1 then read from opposite bank halves -- conflict free;
//  * the transforms run in registers; dM is built with the sign-free matrix A' = [[1,0],[1,1],[1,-1],[0,1]] (A's last row
//    negated) and the signs (-1)^(i==3) (-1)^(j==3) are applied once, in the epilogue, to the accumulators;
//  * epilogue: dg = G^T dU G per lane (16 -> 9 values), the four waves' partial sums are added through LDS in a fixed order,
//    ONE slab per workgroup in the layout of conv_wgrad.hip ([split][tap | bias][ci][co]); wgrad_reduce_kernel sums the slabs
//    (bitwise reproducible, no atomics).  The bias gradient is the sum of the dY values the lanes read anyway.
//
// Replaces autograd's conv2d weight-gradient for the 3x3 layers of networks/acai_vanilla.py:49-102 with >= 32 channels.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "aesr_kernels.h"

constexpr int WW_OOB = 0x70000000;
#ifndef WW_G
#define WW_G 8                  // MFMAs issued back to back
#endif

__device__ __forceinline__ void ww_dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int... I, class F>
__device__ __forceinline__ void ww_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// A spatial tile this workgroup visits.  All uniform (SGPRs).  The two streams get per-IMAGE buffer resources: the hardware range
// check of the resource then answers "row outside the image" by itself (zero, no fetch).
struct WwTile {
    int n, ty, tx;              // image, tile row, tile column
    const char* xb;             // start of image n of x / of dy
    const char* db;
    int orgx, orgd;             // byte offset of the patch / tile origin inside the image, or the out-of-range marker past the last tile
    int x0;                     // first column of the tile
};

template <int TH, int TW, bool STAMP>
__global__ __launch_bounds__(256, 1) void conv_wgrad_wino_f32(WgradArgs a) {
    constexpr int CIT = 32, COT = 32;
    constexpr int PH = TH + 2, PWL = (TW + 2 + 3) & ~3, TWL = TW;     // LDS row strides in PIXELS (multiples of 4: the swizzle then depends on the column only)
    // floats of one X / dY buffer, in whole DMA rounds (256 lanes x 16 B): the last round of a buffer must not spill into its
    // neighbour, which is being read
    constexpr int XFL = (PH * PWL * 32 + 1023) & ~1023, DFL = (TH * TWL * 32 + 1023) & ~1023;
    constexpr int NPX = XFL >> 10, NPD = DFL >> 10, NPIECE = NPX + NPD;          // DMA rounds of a tile
    constexpr int KPR = TW >> 3, NKS = (TH >> 1) * KPR, NST = NKS >> 2;           // k-steps per tile row / per tile / per wave and tile
    constexpr int NSLOT = (NPIECE + NST - 1) / NST;                               // DMA rounds issued per k-step
    static_assert(NKS % 8 == 0, "an even number of k-steps per wave: the operand buffers keep fixed roles around the loop");
    static_assert(2 * (XFL + DFL) * 4 <= 160 * 1024, "LDS");
    constexpr unsigned MGX = (65536 + PWL - 1) / PWL, MGD = (65536 + TWL - 1) / TWL;      // pixel -> row by multiply-high (pixel < 1820)

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

    // ---- DMA: round m of a tile = 256 lanes x 16 B; lane -> (pixel (tid >> 3) + 32 m, position tid & 7) of the buffer.  Nothing per
    // lane is kept between rounds (a descriptor table would cost 40 registers the operand pipeline needs): ~15 instructions per
    // round rebuild it, dealt out between the MFMAs like everything else ----
    const int sh = a.x_up2 ? 1 : 0;                                     // nearest Upsample x2 folded in: x is stored at half size, pixel (y, x) <- (y/2, x/2)
    const int xH = a.H >> sh, xW = a.W >> sh;
    const int ximg = xH * xW * a.Cin * 4, dimg = a.H * a.W * a.Cout * 4; // bytes of one image
    const int xrow = xW * a.Cin * 4, drow = a.W * a.Cout * 4, xpx = a.Cin * 4, dpx = a.Cout * 4;
    int prow = tid >> 3;                  // made opaque inside the loop: otherwise the compiler hoists all 20 rounds' lane terms out of it
    const int pos16 = (tid & 7) << 4;
    const int tpi = a.tiles_y * a.tiles_x;
    const int dS_n = a.S / tpi, dS_r = a.S - dS_n * tpi, dS_ty = dS_r / a.tiles_x, dS_tx = dS_r - dS_ty * a.tiles_x;
    auto locate = [&](WwTile& t) {
        const bool valid = t.n < a.N;
        const int n = valid ? t.n : 0, y0 = t.ty * TH, x0 = t.tx * TW;
        t.xb = (const char*)a.x + (size_t)n * ximg;
        t.db = (const char*)a.dy + (size_t)n * dimg;
        const int ox = ((y0 >> sh) - 1 + sh) * xrow + ((x0 >> sh) - 1 + sh) * xpx + ci0 * 4, od = y0 * drow + x0 * dpx + co0 * 4;
        t.orgx = valid ? ox : WW_OOB;
        t.orgd = valid ? od : WW_OOB;
        t.x0 = x0;
    };
    auto advance = [&](WwTile t) {
        t.tx += dS_tx;
        if (t.tx >= a.tiles_x) { t.tx -= a.tiles_x; ++t.ty; }
        t.ty += dS_ty;
        if (t.ty >= a.tiles_y) { t.ty -= a.tiles_y; ++t.n; }
        t.n += dS_n;
        locate(t);
        return t;
    };
    // ---- everything besides the MFMAs is cut into ATOMS of one or two instructions and dealt out by hand, a few behind every MFMA,
    // scheduling fences in between.  ONE wave per SIMD issues in order: an instruction behind an MFMA waits until the matrix pipe
    // takes that MFMA (32 cycles after the previous one), so whatever follows a RUN of MFMAs is not hidden by it; ~28 cycles of
    // other work (7 instructions) behind each single MFMA are.  [measured: groups of 4 MFMAs + 10 others ran at 71 % of the pipe]

    // B atoms -- one DMA round (M compile-time: stream, LDS slot) in 12 atoms; the lane state of the round in flight:
    unsigned b_pix = 0, b_pr = 0, b_pc = 0, b_gx = 0;
    int b_w = 0, b_sy = 0, b_sx = 0, b_rel = 0;
    auto dma_atom = [&](auto Mc, auto Kc, const WwTile& t, int buf) {
        constexpr int M = decltype(Mc)::value, K = decltype(Kc)::value;
        static_assert(M < NPIECE, "round");
        constexpr bool IS_X = M < NPX;
        constexpr int KM = IS_X ? M : M - NPX;
        if constexpr (K == 0) b_pix = (unsigned)prow + 32u * KM;
        if constexpr (K == 1) b_pr = (b_pix * (IS_X ? MGX : MGD)) >> 16;
        if constexpr (K == 2) b_pc = b_pix - b_pr * (IS_X ? PWL : TWL);
        if constexpr (K == 3) b_w = (int)((b_pc & 2u) << 5);                             // channel quad ^ 4 * ((col >> 1) & 1), in bytes
        if constexpr (K == 4) b_w ^= pos16;
        if constexpr (IS_X) {
            if constexpr (K == 5) b_sy = ((int)b_pr - sh) >> sh;                         // relative source pixel (arithmetic shift: -1 stays -1)
            if constexpr (K == 6) b_sx = ((int)b_pc - sh) >> sh;
            if constexpr (K == 7) b_rel = __mul24(b_sx, xpx) + b_w;
            if constexpr (K == 8) b_rel = __mul24(b_sy, xrow) + b_rel;
            if constexpr (K == 9) b_gx = (unsigned)(t.x0 - 1) + b_pc;
            if constexpr (K == 10) b_rel = b_gx < (unsigned)a.W ? t.orgx + b_rel : WW_OOB;    // rows outside the image: the per-image resource says so
            if constexpr (K == 11) ww_dma(__builtin_amdgcn_make_buffer_rsrc((void*)t.xb, 0, ximg, 0x00020000), ldsX0 + buf * XFL + KM * 1024 + wave * 256, b_rel);
        } else {
            if constexpr (K == 7) b_rel = __mul24((int)b_pc, dpx) + b_w;
            if constexpr (K == 8) b_rel = __mul24((int)b_pr, drow) + b_rel;
            if constexpr (K == 9) b_gx = (unsigned)t.x0 + b_pc;
            if constexpr (K == 10) b_rel = b_gx < (unsigned)a.W ? t.orgd + b_rel : WW_OOB;
            if constexpr (K == 11) ww_dma(__builtin_amdgcn_make_buffer_rsrc((void*)t.db, 0, dimg, 0x00020000), ldsD0 + buf * DFL + KM * 1024 + wave * 256, b_rel);
        }
    };
    constexpr int NBA = 12;                     // B atoms per round

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
    const int lnA = 2 * g * 32 + chA, lnB = 2 * g * 32 + chB;       // + the lane's tile (2 g pixels to the right)

    // A atoms -- the operands of one k-step: V = B^T d B of this lane's (channel, tile) for both ci blocks (block 1 = the other half
    // of the pixel) and M' = A' dY A'^T for both co blocks; the bias gradient is the sum of the dY values read.  6 address atoms,
    // 40 LDS reads (immediate offsets off the six lane bases), 32 + 32 transform atoms for V, 32 for M'.
    float V[2][2][16], M[2][2][16];             // [buffer][block][position]
    const float* a_px[2][2] = {{lds, lds}, {lds, lds}};
    const float* a_pd[2] = {lds, lds};
    float a_d[2][4][4], a_y[2][4], a_t[2][4][4], a_s[2], a_r[2][2];
    constexpr int NAA = 6 + 40 + 32 + 32 + 32;
    auto op_atom = [&](auto Kc, int xu, int du, float bw, bool weigh, float (&Vn)[2][16], float (&Mn)[2][16]) {
        constexpr int K = decltype(Kc)::value;
        if constexpr (K < 6) {
            if constexpr (K == 0) a_px[0][0] = ldsX0 + xu + lnA;
            if constexpr (K == 1) a_px[0][1] = ldsX0 + xu + lnB;
            if constexpr (K == 2) a_px[1][0] = ldsX0 + xu + (lnA ^ 16);
            if constexpr (K == 3) a_px[1][1] = ldsX0 + xu + (lnB ^ 16);
            if constexpr (K == 4) a_pd[0] = ldsD0 + du + lnA;
            if constexpr (K == 5) a_pd[1] = ldsD0 + du + (lnA ^ 16);
        } else if constexpr (K < 38) {              // patch pixel (r, c) of the lane's tile; columns 0,1 share (col >> 1) parity, 2,3 flip it
            constexpr int k = K - 6, i = k >> 4, c = (k >> 2) & 3, r = k & 3;
            a_d[i][r][c] = a_px[i][c >> 1][(r * PWL + c) * 32];
        } else if constexpr (K < 46) {
            constexpr int k = K - 38, o = k >> 2, e = k & 3;
            a_y[o][e] = a_pd[o][(e >> 1) * TWL * 32 + (e & 1) * 32];
        } else if constexpr (K < 78) {              // columns: t = B^T d
            constexpr int k = K - 46, i = k >> 4, c = (k >> 2) & 3, r = k & 3;
            if constexpr (r == 0) a_t[i][0][c] = a_d[i][0][c] - a_d[i][2][c];
            if constexpr (r == 1) a_t[i][1][c] = a_d[i][1][c] + a_d[i][2][c];
            if constexpr (r == 2) a_t[i][2][c] = a_d[i][2][c] - a_d[i][1][c];
            if constexpr (r == 3) a_t[i][3][c] = a_d[i][1][c] - a_d[i][3][c];
        } else if constexpr (K < 110) {             // rows: V = t B
            constexpr int k = K - 78, i = k >> 4, r = (k >> 2) & 3, c = k & 3;
            if constexpr (c == 0) Vn[i][r * 4 + 0] = a_t[i][r][0] - a_t[i][r][2];
            if constexpr (c == 1) Vn[i][r * 4 + 1] = a_t[i][r][1] + a_t[i][r][2];
            if constexpr (c == 2) Vn[i][r * 4 + 2] = a_t[i][r][2] - a_t[i][r][1];
            if constexpr (c == 3) Vn[i][r * 4 + 3] = a_t[i][r][1] - a_t[i][r][3];
        } else {                                    // M' rows {y0}, {y0 + y1}, {y0 - y1}, {y1} (y0 = top pixel pair, y1 = bottom), then the same along columns
            constexpr int k = K - 110, o = k >> 4, e = k & 15;
            const float y00 = a_y[o][0], y01 = a_y[o][1], y10 = a_y[o][2], y11 = a_y[o][3];
            if constexpr (e == 0) a_s[0] = y00 + y01;
            if constexpr (e == 1) a_s[1] = y10 + y11;
            if constexpr (e == 2) a_s[0] = a_s[0] + a_s[1];
            if constexpr (e == 3) accb[o] += weigh ? bw * a_s[0] : a_s[0];
            if constexpr (e == 4) a_r[0][0] = y00 + y10;
            if constexpr (e == 5) a_r[0][1] = y01 + y11;
            if constexpr (e == 6) a_r[1][0] = y00 - y10;
            if constexpr (e == 7) a_r[1][1] = y01 - y11;
            if constexpr (e >= 8) {
                constexpr int i = (e - 8) >> 1;
                const float r0 = i == 0 ? y00 : i == 1 ? a_r[0][0] : i == 2 ? a_r[1][0] : y10;
                const float r1 = i == 0 ? y01 : i == 1 ? a_r[0][1] : i == 2 ? a_r[1][1] : y11;
                if constexpr ((e & 1) == 0) { Mn[o][i * 4 + 0] = r0; Mn[o][i * 4 + 1] = r0 + r1; }
                else { Mn[o][i * 4 + 3] = r1; Mn[o][i * 4 + 2] = r0 - r1; }
            }
        }
    };
    // LDS float offsets of k-step ks = wave + 4 j of the tile in buffer `buf`: tile (tyl, 4 kx + g) has its top-left patch pixel at
    // (2 tyl, 8 kx + 2 g)
    auto step_xu = [&](int j, int buf) { const int ks = wave + 4 * j, tyl = ks / KPR, kx = ks - tyl * KPR; return buf * XFL + ((2 * tyl) * PWL + 8 * kx) * 32; };
    auto step_du = [&](int j, int buf) { const int ks = wave + 4 * j, tyl = ks / KPR, kx = ks - tyl * KPR; return buf * DFL + ((2 * tyl) * TWL + 8 * kx) * 32; };

    // DMA rounds per k-step of the window a tile is fetched in (it opens behind the barrier that frees its buffer -- position 0 is
    // the LAST k-step of the tile before the previous one -- and closes at the next barrier): front-loaded, the rounds issued
    // last have a whole k-step (>= 2048 cycles) to land
    constexpr int NR0 = NST == 2 ? NPIECE : 8, NR1 = NST == 2 ? 0 : 7, NR2 = NST == 2 ? 0 : NPIECE - 15;
    static_assert(NST == 2 || NST == 4, "window");
    static_assert(NR2 >= 0 && NR0 * NBA <= 2 * 64 + 16, "rounds per k-step");

    // debug instantiation of the same loop (STAMP, AESR_WGRAD_WINO_DBG=1): cycles per wave in the k-step loop and at the barrier.
    // Compile-time: a branch here would split the loop body into blocks and the compiler sinks the transforms across them.
    long long tph[3] = {0, 0, 0}, tq = 0;
#define WW_STAMP(k)                                                   \
    if constexpr (STAMP) {                                            \
        const long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
        tph[k] += t_ - tq;                                            \
        tq = t_;                                                      \
    }
    if constexpr (STAMP) tq = (long long)__builtin_amdgcn_s_memtime();

    // ---- the pipeline.  Tile v lives in buffer v & 1.  The barrier of tile v stands before its LAST k-step: by then every wave has
    // read the last operands it needs from buffer v & 1 (they are in registers) and tile v + 1 has landed in the other buffer,
    // so the last k-step already fetches the first operands of tile v + 1, and the DMA rounds of tile v + 2 start into buffer
    // v & 1 right behind the barrier -- spread over k-steps, never as a burst. ----
    WwTile t0, t1, t2;
    {
        t0.n = split / tpi;
        const int rem = split - t0.n * tpi;
        t0.ty = rem / a.tiles_x;
        t0.tx = rem - t0.ty * a.tiles_x;
        if (split >= a.ntiles) t0.n = a.N;
        locate(t0);
    }
    t1 = advance(t0);
    t2 = advance(t1);
    ww_for(std::make_integer_sequence<int, NPIECE>{}, [&](auto Mc) {
        ww_for(std::make_integer_sequence<int, NBA>{}, [&](auto Kc) { dma_atom(Mc, Kc, t0, 0); });
    });
    ww_for(std::make_integer_sequence<int, NR0>{}, [&](auto Mc) {
        ww_for(std::make_integer_sequence<int, NBA>{}, [&](auto Kc) { dma_atom(Mc, Kc, t1, 1); });
    });
    __syncthreads();
    int buf = 0;
    {
        const int xu = step_xu(0, 0), du = step_du(0, 0);
        ww_for(std::make_integer_sequence<int, NAA>{}, [&](auto Kc) { op_atom(Kc, xu, du, 1.f, false, V[0], M[0]); });
    }
    WW_STAMP(2)
    if (t0.n < a.N) do {
        asm volatile("" : "+v"(prow));
        ww_for(std::make_integer_sequence<int, NST>{}, [&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            constexpr bool LAST = j == NST - 1;
            if constexpr (LAST) {
                WW_STAMP(0)
                __syncthreads();
                WW_STAMP(1)
            }
            // this k-step: MFMAs on V/M[j & 1]; the operands of the next one (the first of the next tile behind the barrier) into the
            // other pair; DMA rounds [R0, R0 + NR) of tile v + 1 (v + 2 behind the barrier)
            constexpr int NR = LAST ? NR0 : j == 0 ? NR1 : j == 1 ? NR2 : 0;
            constexpr int R0 = LAST ? 0 : j == 0 ? NR0 : NR0 + NR1;
            constexpr int NB = NR * NBA;
            const int xu = LAST ? step_xu(0, buf ^ 1) : step_xu(j + 1, buf), du = LAST ? step_du(0, buf ^ 1) : step_du(j + 1, buf);
            const float bw = t1.n < a.N ? 1.f : 0.f;
            const WwTile& tt = LAST ? t2 : t1;
            const int tb = LAST ? buf : buf ^ 1;
            __builtin_amdgcn_sched_barrier(0);
            // groups of WW_G MFMAs, then the atoms of WW_G slots: on this chip the f32 MFMA and the vector ALU do not overlap (a filler
            // costs its 4 issue cycles wherever it stands, scripts/micro/mfma_gap.hip) and every MFMA -> VALU switch costs ~4 more
            ww_for(std::make_integer_sequence<int, 64 / WW_G>{}, [&](auto Qc) {
                constexpr int q0 = decltype(Qc)::value * WW_G;
                ww_for(std::make_integer_sequence<int, WW_G>{}, [&](auto Dc) {
                    constexpr int q = q0 + decltype(Dc)::value, x = q >> 2, i = (q >> 1) & 1, o = q & 1;
                    // the accumulators are pinned to the accumulation registers ("+a"): left to itself the allocator moves parts of
                    // them through the (full) vector registers and scratch.  Operands are a k-step old and an accumulator is touched
                    // once per k-step, so no MFMA hazard is near an asm statement.
                    f32x4& acc_ = acc[x][i][o];
                    const float va_ = V[j & 1][i][x], vb_ = M[j & 1][o][x];
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(va_), "v"(vb_));
                });
                constexpr int A0 = q0 * NAA / 64, A1 = (q0 + WW_G) * NAA / 64;
                ww_for(std::make_integer_sequence<int, A1 - A0>{}, [&](auto Kc) {
                    op_atom(std::integral_constant<int, A0 + decltype(Kc)::value>{}, xu, du, bw, LAST, V[(j + 1) & 1], M[(j + 1) & 1]);
                });
                constexpr int B0 = q0 * NB / 64, B1 = (q0 + WW_G) * NB / 64;
                ww_for(std::make_integer_sequence<int, B1 - B0>{}, [&](auto Kc) {
                    constexpr int b = B0 + decltype(Kc)::value;
                    dma_atom(std::integral_constant<int, R0 + b / NBA>{}, std::integral_constant<int, b % NBA>{}, tt, tb);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        t0 = t1;
        t1 = t2;
        t2 = advance(t2);
        buf ^= 1;
    } while (t0.n < a.N);
    WW_STAMP(0)
    __syncthreads();            // DMA rounds still in flight (zero fills past the last tile) must land before the exchange reuses the LDS
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
    // the planes are dead (barrier at the end of the last tile): waves 1-3 park their partial sums in LDS, wave 0 adds them up
    f32x4* ex = (f32x4*)lds;                               // [wave - 1][36][64 lanes]
    float* exb = lds + 3 * 36 * 64 * 4;                    // [wave - 1][2][16]
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int o = 0; o < 2; ++o) ex[((wave - 1) * 36 + (t * 4 + i * 2 + o)) * 64 + lane] = dg[t][i][o];
        if (g == 0) {
            exb[(wave - 1) * 32 + l15] = accb[0];
            exb[(wave - 1) * 32 + 16 + l15] = accb[1];
        }
    }
    __syncthreads();
    if (wave == 0) {
        const size_t plane = (size_t)a.CinP * a.CoutP;
        float* sl = a.slab + (size_t)split * 10 * plane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    f32x4 v = dg[t][i][o];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += ex[(w * 36 + (t * 4 + i * 2 + o)) * 64 + lane];
                    const int co = co0 + o * 16 + l15;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ci = ci0 + i * 16 + g * 4 + e;
                        sl[t * plane + (size_t)ci * a.CoutP + co] = v[e];
                    }
                }
        if (do_bias && g == 0) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                float b = accb[o];
#pragma unroll
                for (int w = 0; w < 3; ++w) b += exb[w * 32 + o * 16 + l15];
                sl[9 * plane + co0 + o * 16 + l15] = b;
            }
        }
    }
}

size_t aesr_wgrad_wino_lds_bytes(int TH, int TW) {
    const size_t xfl = ((size_t)(TH + 2) * round_up(TW + 2, 4) * 32 + 1023) & ~(size_t)1023;
    const size_t dfl = ((size_t)TH * round_up(TW, 4) * 32 + 1023) & ~(size_t)1023;
    const size_t bufs = 2 * (xfl + dfl) * sizeof(float);
    const size_t exch = ((size_t)3 * 36 * 64 * 4 + 3 * 32) * sizeof(float);
    return bufs > exch ? bufs : exch;
}

// the tiles the kernel is built for: (TH / 2) * (TW / 8) k-steps a multiple of 8, both double buffers within 160 KB
bool aesr_wgrad_wino_tile_ok(int TH, int TW) {
    return (TH == 16 && TW == 8) || (TH == 8 && TW == 16) || (TH == 16 && TW == 16) || (TH == 4 && TW == 32) || (TH == 8 && TW == 32);
}

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
        aesr_set_error("conv_wgrad_wino: tile %dx%d is not one of 16x8, 8x16, 16x16, 4x32, 8x32 (or not 3x3 / padding 1)", a.TH, a.TW);
        return AESR_ERR_ARG;
    }
    if (a.CinP % 32 != 0 || a.CoutP % 32 != 0 || a.Cin != a.CinP || a.Cout != a.CoutP || a.S < 1 || a.S > a.ntiles ||
        a.tiles_y != ceil_div(a.H, a.TH) || a.tiles_x != ceil_div(a.W, a.TW) || a.ntiles != a.N * a.tiles_y * a.tiles_x) {
        aesr_set_error("conv_wgrad_wino: channel counts must be multiples of 32 and the tile grid consistent (S=%d, %d tiles)", a.S, a.ntiles);
        return AESR_ERR_ARG;
    }
    if (a.TH == 16 && a.TW == 8) return launch_wgrad_wino<16, 8>(a, st);
    if (a.TH == 8 && a.TW == 16) return launch_wgrad_wino<8, 16>(a, st);
    if (a.TH == 16 && a.TW == 16) return launch_wgrad_wino<16, 16>(a, st);
    if (a.TH == 4 && a.TW == 32) return launch_wgrad_wino<4, 32>(a, st);
    return launch_wgrad_wino<8, 32>(a, st);
}
