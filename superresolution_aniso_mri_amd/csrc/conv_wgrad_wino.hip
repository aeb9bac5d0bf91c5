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

#include "aesr_kernels.h"

constexpr int WW_NX = 12;     // DMA pieces (16 B) per thread for the X patch: (TH + 2) * round_up(TW + 2, 4) * 8 <= 256 * WW_NX
constexpr int WW_ND = 8;      // ... for the dY tile: TH * TW * 8 <= 256 * WW_ND
constexpr int WW_OOB = 0x70000000;

__device__ __forceinline__ void ww_dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}

__global__ __launch_bounds__(256, 1) void conv_wgrad_wino_f32(WgradArgs a) {
    constexpr int CIT = 32, COT = 32;
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
    const int PH = a.TH + 2, PWp = a.TW + 2;
    const int PWL = a.PWS, TWL = a.TWS;               // LDS row strides in PIXELS (multiples of 4: the swizzle then depends on the column only)
    // floats of one X / dY buffer, in whole DMA rounds (256 lanes x 16 B): the last round of a buffer must not spill into its
    // neighbour, which is being read
    const int XFL = (PH * PWL * 32 + 1023) & ~1023, DFL = (a.TH * TWL * 32 + 1023) & ~1023;
    float* const ldsX0 = lds;                        // [2][PH][PWL][32]
    float* const ldsD0 = lds + 2 * XFL;              // [2][TH][TWL][32]
    const bool do_bias = ciy == 0;
    const int tpi = a.tiles_y * a.tiles_x;

    // ---- DMA pieces: piece m = tid + 256 k fills LDS float4 slot m = (pixel m >> 3, position m & 7) of the buffer ----
    const int xH = a.x_up2 ? a.H >> 1 : a.H, xW = a.x_up2 ? a.W >> 1 : a.W;          // stored size of x (nearest Upsample x2 folded in: pixel (y, x) <- (y/2, x/2))
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * xH * xW * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);
    int xrc[WW_NX], drc[WW_ND], xrel[WW_NX], drel[WW_ND];      // (row << 16 | col) or -1; byte offset relative to the tile origin or OOB
    const int pos = tid & 7;
#pragma unroll
    for (int k = 0; k < WW_NX; ++k) {
        const int pix = (tid + 256 * k) >> 3;
        const int pr = pix / PWL, pc = pix - pr * PWL;
        const int q = pos ^ (((pc >> 1) & 1) << 2);
        const int ci = ci0 + q * 4;
        const bool ok = pr < PH && pc < PWp;
        xrc[k] = ok ? ((pr << 16) | pc) : -1;
        xrel[k] = (ok && ci < a.Cin) ? ((pr * a.W + pc) * a.Cin + ci) * 4 : WW_OOB;
    }
#pragma unroll
    for (int k = 0; k < WW_ND; ++k) {
        const int pix = (tid + 256 * k) >> 3;
        const int r = pix / TWL, c = pix - r * TWL;
        const int q = pos ^ (((c >> 1) & 1) << 2);
        const int co = co0 + q * 4;
        const bool ok = r < a.TH && c < a.TW;
        drc[k] = ok ? ((r << 16) | c) : -1;
        drel[k] = (ok && co < a.Cout) ? ((r * a.W + c) * a.Cout + co) * 4 : WW_OOB;
    }
    const int npx = XFL >> 10, npd = DFL >> 10;                        // DMA rounds of a buffer (1024 floats per round)
    auto stage = [&](int tile, int buf) {
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        const int xorg = ((n * a.H + y0 - 1) * a.W + (x0 - 1)) * a.Cin * 4;       // patch origin (may lie in the padding)
        const int dorg = ((n * a.H + y0) * a.W + x0) * a.Cout * 4;
        float* xdst = ldsX0 + buf * XFL + wave * 256;
        float* ddst = ldsD0 + buf * DFL + wave * 256;
#pragma unroll
        for (int k = 0; k < WW_NX; ++k) {
            if (k < npx) {
                const unsigned gy = (unsigned)(y0 + (xrc[k] >> 16) - 1), gx = (unsigned)(x0 + (xrc[k] & 0xffff) - 1);
                int off = (xrc[k] >= 0 && gy < (unsigned)a.H && gx < (unsigned)a.W) ? xorg + xrel[k] : WW_OOB;
                if (a.x_up2 && off != WW_OOB && xrel[k] != WW_OOB) {
                    const int q = pos ^ ((((xrc[k] & 0xffff) >> 1) & 1) << 2);
                    off = (((n * xH + (int)(gy >> 1)) * xW + (int)(gx >> 1)) * a.Cin + ci0 + q * 4) * 4;
                }
                ww_dma(rs_x, xdst + k * 1024, off);
            }
        }
#pragma unroll
        for (int k = 0; k < WW_ND; ++k) {
            if (k < npd) {
                const int gy = y0 + (drc[k] >> 16), gx = x0 + (drc[k] & 0xffff);
                const int off = (drc[k] >= 0 && gy < a.H && gx < a.W) ? dorg + drel[k] : WW_OOB;
                ww_dma(rs_d, ddst + k * 1024, off);
            }
        }
    };

    f32x4 acc[16][2][2];                    // [position][ci block][co block]: D rows = ci 4g..4g+3, column = co l15
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int o = 0; o < 2; ++o) acc[x][i][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float accb[2] = {0.f, 0.f};

    const int kpr = a.TW >> 3, nks = (a.TH >> 1) * kpr;          // k-steps per tile row / per spatial tile
    // this lane's channel inside a pixel: quad l15 >> 2 (+ 4 for the second 16-channel block), element l15 & 3; the swizzle swaps
    // the blocks for odd tile pairs: position = quad ^ 4 * ((col >> 1) & 1), col = 2 * tile + j -> parity of (tile + (j >> 1))
    const int sw = (g & 1) << 2;                                   // the tile index of a k-step is 4 * kx + g: its parity is g's
    const int chA = (((l15 >> 2) ^ sw) << 2) + (l15 & 3);          // float offset of block 0's channel in pixels with (col >> 1) even
    const int chB = (((l15 >> 2) ^ sw ^ 4) << 2) + (l15 & 3);      // ... with (col >> 1) odd

    // debug build of the same loop (a.dbgbuf != nullptr, AESR_WGRAD_WINO_DBG=1): cycles per wave spent staging, in the unhidden first
    // operands of a tile, in the pair loop and at the barrier
    const bool stamp = a.dbgbuf != nullptr;
    long long tph[5] = {0, 0, 0, 0, 0}, tq = 0;
#define WW_STAMP(k)                                                 \
    if (stamp) {                                                    \
        const long long t_ = (long long)__builtin_amdgcn_s_memtime(); \
        tph[k] += t_ - tq;                                          \
        tq = t_;                                                    \
    }
    if (stamp) tq = (long long)__builtin_amdgcn_s_memtime();
    int tile = split, buf = 0;
    if (tile < a.ntiles) stage(tile, 0);
    __syncthreads();
    WW_STAMP(4)
    while (tile < a.ntiles) {
        const int next = tile + a.S;
        if (next < a.ntiles) stage(next, buf ^ 1);                 // lands before the barrier at the end of this tile
        WW_STAMP(0)
        const float* xbuf = ldsX0 + buf * XFL;
        const float* dbuf = ldsD0 + buf * DFL;

        // software pipeline over this wave's k-steps: the operands of step s + 1 (LDS reads + both transforms, ~130 instructions)
        // are produced in the shadow of the 64 MFMAs of step s; sched_group_barrier deals them out.  With ONE wave per SIMD
        // nothing else hides them.
        float V[2][2][16], M[2][2][16];             // [buffer][block][position]
        auto operands = [&](int ks, float bw, float (&Vn)[2][16], float (&Mn)[2][16]) {
            const int tyl = ks / kpr, kx = ks - tyl * kpr;
            // tile (tyl, 4 kx + g): top-left patch pixel (2 tyl, 8 kx + 2 g); columns j = 0,1 share (col >> 1) parity, j = 2,3 flip it
            const float* xb = xbuf + ((2 * tyl) * PWL + 8 * kx + 2 * g) * 32;
            const float* db = dbuf + ((2 * tyl) * TWL + 8 * kx + 2 * g) * 32;
            // ---- V = B^T d B for this lane's (channel, tile), both ci blocks (block 1 = the other half of the pixel) ----
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c01 = i == 0 ? chA : (chA ^ 16), c23 = i == 0 ? chB : (chB ^ 16);
                float d[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    d[r][0] = xb[(r * PWL + 0) * 32 + c01];
                    d[r][1] = xb[(r * PWL + 1) * 32 + c01];
                    d[r][2] = xb[(r * PWL + 2) * 32 + c23];
                    d[r][3] = xb[(r * PWL + 3) * 32 + c23];
                }
                float t[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    t[0][c] = d[0][c] - d[2][c];
                    t[1][c] = d[1][c] + d[2][c];
                    t[2][c] = d[2][c] - d[1][c];
                    t[3][c] = d[1][c] - d[3][c];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Vn[i][r * 4 + 0] = t[r][0] - t[r][2];
                    Vn[i][r * 4 + 1] = t[r][1] + t[r][2];
                    Vn[i][r * 4 + 2] = t[r][2] - t[r][1];
                    Vn[i][r * 4 + 3] = t[r][1] - t[r][3];
                }
            }
            // ---- M' = A' dY A'^T for this lane's (channel, tile), both co blocks; bias gradient = sum of the dY read ----
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int c = o == 0 ? chA : (chA ^ 16);
                const float y00 = db[c], y01 = db[32 + c], y10 = db[TWL * 32 + c], y11 = db[TWL * 32 + 32 + c];
                accb[o] += bw * ((y00 + y01) + (y10 + y11));
                const float r[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {y10, y11}};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    Mn[o][i * 4 + 0] = r[i][0];
                    Mn[o][i * 4 + 1] = r[i][0] + r[i][1];
                    Mn[o][i * 4 + 2] = r[i][0] - r[i][1];
                    Mn[o][i * 4 + 3] = r[i][1];
                }
            }
        };
        auto mfmas = [&](const float (&Vc)[2][16], const float (&Mc)[2][16]) {
            // 64 MFMAs: A = V (M = ci), B = M' (N = co), K = the 4 tiles of the k-step
#pragma unroll
            for (int x = 0; x < 16; ++x)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        acc[x][i][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vc[i][x], Mc[o][x], acc[x][i][o], 0, 0, 0);
        };
#define WW_DEAL()                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) {                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);               \
        if (q_ < 10) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  \
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);               \
    }
        // Steps come in pairs (the planner picks tiles whose k-steps are a multiple of 8 = an even count per wave), so the two
        // operand buffers keep fixed roles around the back edge -- a loop that may leave between the halves makes the compiler
        // rotate the buffers with ~110 register moves per pair.  The operands of a step past the end are those of the last step
        // again, with weight 0 in the bias sum: no branch splits a scheduling region.
        const int nst = nks > wave ? (nks - wave + 3) >> 2 : 0;            // k-steps of this wave: wave, wave + 4, ...
        const int npair = nst >> 1;
        if (npair > 0) operands(wave, 1.f, V[0], M[0]);
        WW_STAMP(1)
        for (int s2 = 0; s2 < npair; ++s2) {
            const int ks = wave + 8 * s2;
            __builtin_amdgcn_sched_barrier(0);
            operands(ks + 4, 1.f, V[1], M[1]);
            mfmas(V[0], M[0]);
            WW_DEAL()
            __builtin_amdgcn_sched_barrier(0);
            const bool more = s2 + 1 < npair;
            operands(more ? ks + 8 : ks + 4, more ? 1.f : 0.f, V[0], M[0]);
            mfmas(V[1], M[1]);
            WW_DEAL()
            __builtin_amdgcn_sched_barrier(0);
        }
        if (nst & 1) {              // odd count (tiles the planner does not choose): the last step, unpipelined
            operands(wave + 4 * (nst - 1), 1.f, V[1], M[1]);
            mfmas(V[1], M[1]);
        }
#undef WW_DEAL
        WW_STAMP(2)
        __syncthreads();            // every wave is done with `buf`; the next tile is complete in `buf ^ 1`
        WW_STAMP(3)
        buf ^= 1;
        tile = next;
    }

    if (stamp && lane == 0)
        for (int k = 0; k < 5; ++k) a.dbgbuf[(blockIdx.x * 4 + wave) * 5 + k] = (float)tph[k];
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

int aesr_launch_conv_wgrad_wino(const WgradArgs& a, hipStream_t st) {
    if ((size_t)a.N * a.H * a.W * a.Cin >= (size_t)0x1C000000 || (size_t)a.N * a.H * a.W * a.Cout >= (size_t)0x1C000000) {
        aesr_set_error("conv_wgrad_wino: tensors of 469M elements (1.75 GB) or more need 64-bit indexing (not built)");
        return AESR_ERR_UNSUPPORTED;
    }
    if (a.x_up2 && ((a.H | a.W) & 1)) {
        aesr_set_error("conv_wgrad_wino: the folded Upsample(x2) needs even convolution sizes (got %dx%d)", a.H, a.W);
        return AESR_ERR_ARG;
    }
    if (a.TW % 8 != 0 || a.TH % 2 != 0 || a.PWS != round_up(a.TW + 2, 4) || a.TWS != round_up(a.TW, 4) || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) {
        aesr_set_error("conv_wgrad_wino: TW=%d must be a multiple of 8, TH=%d even, row strides multiples of 4, 3x3 padding 1", a.TW, a.TH);
        return AESR_ERR_ARG;
    }
    if ((a.TH + 2) * a.PWS * 8 > 256 * WW_NX || a.TH * a.TWS * 8 > 256 * WW_ND || a.CinP % 32 != 0 || a.CoutP % 32 != 0) {
        aesr_set_error("conv_wgrad_wino: tile %dx%d does not fit the DMA piece slots (or bad channel padding)", a.TH, a.TW);
        return AESR_ERR_ARG;
    }
    const size_t shmem = aesr_wgrad_wino_lds_bytes(a.TH, a.TW);
    if (shmem > (size_t)160 * 1024) {
        aesr_set_error("conv_wgrad_wino: tile needs %zu B of LDS", shmem);
        return AESR_ERR_ARG;
    }
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wgrad_wino_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wgrad_wino_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    dim3 grid(a.S * (a.CinP / 32) * (a.CoutP / 32));
    if (getenv("AESR_WGRAD_WINO_DBG") && grid.x <= 4096) {     // debug: per-phase cycle stamps, printed after a host sync
        static float* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 4 * 5 * sizeof(float));
        WgradArgs b = a;
        b.dbgbuf = dbuf;
        hipLaunchKernelGGL(conv_wgrad_wino_f32, grid, dim3(256), shmem, st, b);
        (void)hipStreamSynchronize(st);
        static float host[4096 * 4 * 5];
        (void)hipMemcpy(host, dbuf, (size_t)grid.x * 4 * 5 * sizeof(float), hipMemcpyDeviceToHost);
        double s5[5] = {0, 0, 0, 0, 0};
        for (unsigned i = 0; i < grid.x * 4; ++i) for (int k = 0; k < 5; ++k) s5[k] += host[i * 5 + k];
        const double visits = (double)((a.ntiles + a.S - 1) / a.S);
        const int nks = (a.TH >> 1) * (a.TW >> 3);
        fprintf(stderr, "[wgrad-wino stamps] grid=%u tile %dx%d S=%d (%.0f visits, %d k-steps = %d MFMA cycles per wave and visit) per visit, cycles: "
                "stage %.0f | first operands %.0f | pair loop %.0f | barrier %.0f || first fill %.0f\n", grid.x, a.TH, a.TW, a.S, visits, nks,
                nks / 4 * 64 * 32, s5[0] / grid.x / 4 / visits, s5[1] / grid.x / 4 / visits, s5[2] / grid.x / 4 / visits, s5[3] / grid.x / 4 / visits,
                s5[4] / grid.x / 4);
        return AESR_OK;
    }
    hipLaunchKernelGGL(conv_wgrad_wino_f32, grid, dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wgrad_wino_f32");
    return AESR_OK;
}
