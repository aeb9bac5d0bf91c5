// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC, stride 1, padding 1.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A            (Lavin & Gray 2016, minimal filtering F(2x2, 3x3))
//
// A 4x4 input tile d (stride 2) gives a 2x2 output tile with 16 multiplies per (ci, co) instead of 36: 2.25x fewer MFMA flops
// than the implicit GEMM of conv_igemm.hip.  The 16 transform positions xi are 16 independent GEMMs
//       M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile]
// that share nothing but the loop: one wave owns 16 Winograd tiles x 32 output channels and keeps the accumulators of ALL 16
// positions (128 VGPRs), so the output transform Y = A^T M A is a per-lane affair in the epilogue (the D layout of 16x16x4 puts
// one tile and four consecutive couts on a lane for every xi).
//
//  * U = G g G^T is computed once per optimizer step by the packing kernel (wino_pack_*), stored as the LDS image
//    [ci chunk(16)][cout tile(32)][xi(16)][ci/4][cout(32)][4]: the A operand of a position is one conflict-free ds_read_b128.
//  * V = B^T d B is computed on the fly IN REGISTERS: a lane reads the 4x4 input pixels of its tile for 4 input channels
//    (16 ds_read_b128 from the raw patch in LDS, laid out [pixel][16 ch]), applies the separable
//    transform (128 v_sub/v_add per 16-channel chunk = 1 per MFMA) and feeds the results straight into the MFMA B operand:
//    V never exists in memory.
//  * 512 threads (8 waves) = ONE workgroup per CU; LDS is double buffered (patch + U chunk, 2 x (<= 40 KB + 32 KB)); the next
//    chunk goes global -> LDS by DMA (buffer_load_dwordx4 ... lds: no staging registers -- the accumulators leave none -- and no
//    ds_write pass), issued right after the barrier and waited for (vmcnt(0)) only at the next barrier, a whole chunk of matrix
//    work later: memory latency under load (2-3 us) never reaches the MFMA pipe.  ONE barrier per chunk.  The DMA destination is
//    lane-linear, so the patch is unpadded (64 B per pixel; the 4x4-pixel reads of a tile row are 4-way bank conflicted, 256 of
//    ~8200 LDS cycles per chunk).  Out-of-range buffer offsets deliver zeros: halo, padding and tails need no masking.
//  * the bias rides in the accumulator of position (1,1), whose value reaches all four outputs of the tile with weight +1.
//  * the same kernel is the data gradient (run on dY with U of the flipped / transposed filter, derivative mask in the epilogue).
//  * nearest-neighbour Upsample(x2) in front of the convolution (networks/acai_vanilla.py:92) is folded in: with `in_up2` the DMA
//    source of patch pixel (y, x) is pixel (y/2, x/2) of the half-resolution tensor -- the upsampled tensor never exists; its
//    adjoint in the data gradient (`out_sum2`) is the sum of the lane's 2x2 output tile, stored as ONE half-resolution pixel.
//
// Accuracy: products and sums are fp32 (exact-fp32 MFMA); the transforms add a few roundings (G has entries 1/2: exact).
// Measured against fp64: 2-4e-7 relative (tests/test_gpu_kernels.py::test_conv_wino_*), inside the stated 1e-5 forward bound.
//
// Replaces the ATen/cuDNN conv2d calls behind networks/acai_vanilla.py:55-56,68,70,87-88,96 and
// lpips/pretrained_networks.py:107-116 for 3x3 / padding 1 layers with Cin % 16 == 0 and Cout % 32 == 0.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "aesr_kernels.h"
#include "aesr_pack_dev.h"

constexpr int WN_S = 16;            // LDS floats per patch pixel: 16 channels, no padding (the DMA destination is lane-linear)
constexpr int WN_NT = 512;          // threads per workgroup
constexpr int WN_TN = 32;           // output channels per work item (2 MFMA blocks)
constexpr int WN_NB = 2;
constexpr int WN_WFL = 16 * 4 * WN_TN * 4;      // floats of one U chunk in LDS (16 positions x 16 ci x 32 co) = 8192
constexpr int WN_WP = WN_WFL / 4 / WN_NT;       // 16-byte weight pieces per thread and chunk = 4
constexpr int WN_OOB = 0x70000000;              // byte offset that every buffer descriptor rejects (see conv_igemm.hip)
constexpr int WN_PIECE_FL = WN_NT * 4;          // floats one DMA round of the workgroup fills (512 lanes x 16 B = 8 KB)

// global -> LDS without registers: lane l of the wave writes 16 bytes at lds_wave_base + 16 l (lds_wave_base is wave-uniform: M0)
__device__ __forceinline__ void wn_dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}

__device__ __forceinline__ f32x4 wn_ld(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
__device__ __forceinline__ void wn_st(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

// NPP = patch staging pieces (16 B) per thread: patch pixels * 4 <= 512 * NPP
template <int NPP, bool MASK>
__global__ __launch_bounds__(WN_NT, 2) void conv_wino_f32(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int G = gridDim.x;
    const int nitems = a.nitems, ncot = a.CoutP / WN_TN, nchunks = a.CinP >> 4;

    const int PW = 2 * a.TWt + 2, PH = 2 * a.THt + 2;
    const int PPI = PH * PW, PP = a.TI * PPI;
    const int TPI = a.THt * a.TWt, TP = a.TI * TPI;
    constexpr int PPS = NPP * WN_PIECE_FL;               // floats of one patch buffer (whole DMA rounds: PP * 16 rounded up)
    float* const ldsP0 = lds;
    float* const ldsW0 = lds + 2 * PPS;
    float* const ldsBias = ldsW0 + 2 * WN_WFL;           // [CoutP] (zeros when there is no bias)
    for (int c = tid; c < a.CoutP; c += WN_NT) ldsBias[c] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;

    const int inH = a.in_up2 ? a.H >> 1 : a.H, inW = a.in_up2 ? a.W >> 1 : a.W;            // stored size of the input tensor
    const int outH = a.out_sum2 ? a.H >> 1 : a.H, outW = a.out_sum2 ? a.W >> 1 : a.W;      // stored size of the output tensor
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * inH * inW * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.upk, 0, (int)((size_t)16 * a.CinP * a.CoutP * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)((size_t)a.N * outH * outW * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ys = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK ? a.ysave : a.out), 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);

    // ---- position-independent maps (computed once) ------------------------------------------------------------
    // this lane's Winograd tile: flattened (image, tile row, tile col) index wave*16 + l15
    int a_off, tpix;                          // LDS float offset of the tile's top-left patch pixel (+ channel quad); img<<20 | tr<<10 | tc or -1
    {
        const int t = wave * 16 + l15;
        const bool valid = t < TP;
        const int tt = valid ? t : 0;
        const int img = tt / TPI;
        const int rem = tt - img * TPI;
        const int tr = rem / a.TWt;
        const int tc = rem - tr * a.TWt;
        a_off = (img * PPI + 2 * tr * PW + 2 * tc) * WN_S + 4 * g;
        tpix = valid ? ((img << 20) | (tr << 10) | tc) : -1;
    }
    int piece[NPP];                           // img<<20 | pr<<10 | pc of the patch pixel of staging piece j, or -1
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int q = tid + WN_NT * j;
        int v = -1;
        if (q < PP * 4) {
            const int p = q >> 2;
            const int img = p / PPI;
            const int rem = p - img * PPI;
            const int pr = rem / PW;
            v = (img << 20) | (pr << 10) | (rem - pr * PW);
        }
        piece[j] = v;
    }
    const int part4 = (tid & 3) * 4;          // channel offset of this thread's patch pieces inside a chunk

    int l_item = blockIdx.x, l_cc = 0;
    if (l_item >= nitems) return;
    int goff[NPP];

    // item -> (image group, region row, region column, cout tile): divisions by launch constants as multiply-high with the
    // host's magic numbers (x / d == mulhi(x, ceil(2^32 / d)) for x * d < 2^32): a runtime integer division costs ~40 instructions
    // and three of them sit on every item boundary, where no MFMA overlaps them
#define WN_DIV(x, m) ((m) ? (int)__umulhi((unsigned)(x), (m)) : (int)(x))          /* m == 0: divisor 1 */
#define WN_ITEM_ORIGIN(item, n0, ty0, tx0, co0)               \
    {                                                         \
        int reg_ = WN_DIV(item, a.m_ncot);                    \
        co0 = ((item) - reg_ * ncot) * WN_TN;                 \
        const int q1_ = WN_DIV(reg_, a.m_regs_x);             \
        const int rx_ = reg_ - q1_ * a.regs_x;                \
        const int q2_ = WN_DIV(q1_, a.m_regs_y);              \
        const int ry_ = q1_ - q2_ * a.regs_y;                 \
        n0 = q2_ * a.TI;                                      \
        ty0 = ry_ * a.THt;                                    \
        tx0 = rx_ * a.TWt;                                    \
    }
#define WN_COMPUTE_GOFF(item)                                                                                     \
    {                                                                                                             \
        int n0_, ty0_, tx0_, co0_;                                                                                \
        WN_ITEM_ORIGIN(item, n0_, ty0_, tx0_, co0_)                                                               \
        (void)co0_;                                                                                               \
        _Pragma("unroll") for (int j = 0; j < NPP; ++j) {                                                         \
            int go = WN_OOB;                                                                                      \
            if (piece[j] >= 0) {                                                                                  \
                const int n = n0_ + (piece[j] >> 20), gy = 2 * ty0_ + ((piece[j] >> 10) & 1023) - 1;              \
                const int gx = 2 * tx0_ + (piece[j] & 1023) - 1;                                                  \
                const int sy = a.in_up2 ? gy >> 1 : gy, sx = a.in_up2 ? gx >> 1 : gx;                             \
                go = (n < a.N && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (((n * inH + sy) * inW + sx) * a.Cin + part4) * 4 : WN_OOB; \
            }                                                                                                     \
            goff[j] = go;                                                                                         \
        }                                                                                                         \
    }
    // byte offsets of the load state's chunk: weights of (chunk, cout tile) are one contiguous 32 KB block
#define WN_CHUNK_OFFS(item, cc, wbase, coff)                                                             \
    const int wbase = (int)(((size_t)(cc) * ncot + ((item) % ncot)) * (WN_WFL * 4));                     \
    const int coff = ((cc) * 16 + part4 < a.Cin) ? (cc) * 64 : WN_OOB;

    // one chunk (16 input channels of the patch + the U block of the cout tile) global -> LDS buffer, NPP + 4 DMAs per thread
#define WN_STAGE(item, cc, pdst, wdst)                                                                         \
    {                                                                                                          \
        WN_CHUNK_OFFS(item, cc, wbase_, coff_)                                                                 \
        _Pragma("unroll") for (int j = 0; j < WN_WP; ++j)                                                      \
            wn_dma(rs_w, (wdst) + j * WN_PIECE_FL + wave * 256, wbase_ + (tid + WN_NT * j) * 16);              \
        _Pragma("unroll") for (int j = 0; j < NPP; ++j)                                                        \
            wn_dma(rs_in, (pdst) + j * WN_PIECE_FL + wave * 256, goff[j] + coff_);                             \
    }
    // ---- prologue: chunk 0 of the first item into buffer 0 -----------------------------------------------------
    WN_COMPUTE_GOFF(l_item)
    WN_STAGE(l_item, 0, ldsP0, ldsW0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (explicit, as at the chunk barrier below)
    __syncthreads();
    // advance the load state to the chunk that will be staged during the first compute phase
    if (1 < nchunks) {
        l_cc = 1;
    } else {
        l_item += G;
        l_cc = 0;
        if (l_item < nitems) WN_COMPUTE_GOFF(l_item)
    }

    // ---- compute state ----
    int c_item = blockIdx.x, cc = 0;
    int cn0, cty0, ctx0, co0;
    WN_ITEM_ORIGIN(c_item, cn0, cty0, ctx0, co0)
    f32x4 acc[16][WN_NB];
    const float mslope = a.mask_act == ACT_LRELU ? a.slope : (a.mask_act == ACT_RELU ? 0.f : 1.f);
    const float nslope = a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU ? 0.f : 1.f);      // ACT_NONE / ACT_SIGMOID: identity
    const bool sigm = a.act == ACT_SIGMOID;
    const bool wave_active = wave * 16 < TP;           // waves whose 16 tile slots are all past the item's tiles skip the arithmetic

    // debug build of the same loop (a.dbgbuf != nullptr): cycles per wave spent in [loop top .. last MFMA], at the barrier, in the
    // item epilogue; shares only -- the stamps themselves perturb the schedule
    const bool stamp = a.dbgbuf != nullptr;
    unsigned long long tph[5] = {0, 0, 0, 0, 0}, tlast = stamp ? __builtin_amdgcn_s_memtime() : 0;
#define WN_STAMP(k)                                                        \
    if (stamp) {                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
        tph[k] += t_ - tlast;                                              \
        tlast = t_;                                                        \
    }
    if ((a.flags & 2) && wave >= 4) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for the younger half

#define WN_INIT_ACC(eco0)                                                                              \
    {                                                                                                  \
        _Pragma("unroll") for (int x = 0; x < 16; ++x)                                                 \
            _Pragma("unroll") for (int nb = 0; nb < WN_NB; ++nb) acc[x][nb] = (f32x4){0.f, 0.f, 0.f, 0.f}; \
        _Pragma("unroll") for (int nb = 0; nb < WN_NB; ++nb)                                           \
            acc[5][nb] = *(const f32x4*)(ldsBias + (eco0) + nb * 16 + 4 * g);                          \
    }
    WN_INIT_ACC(co0)

    int buf = 0;
    while (true) {
        WN_STAMP(3)
        const float* pb = ldsP0 + buf * PPS;
        const float* wb = ldsW0 + buf * WN_WFL + (g * WN_TN + l15) * 4;        // + xi * (4*TN*4) + nb*64
        const bool do_load = l_item < nitems;
        // the staging DMAs of the next chunk (7-9 per wave, ~100 cycles of issue each) are NOT issued by all waves at once: the
        // waves that are served first (0-3) put theirs behind their first position row, so that a SIMD whose one wave is
        // stalled at the DMA queue has the other one in its MFMAs
        const bool stage_late = !(a.flags & 4) && wave < 4 && wave_active;        // AESR_WINO_FLAGS=4: everybody stages first (A/B)
        if (do_load && !stage_late) WN_STAGE(l_item, l_cc, ldsP0 + (buf ^ 1) * PPS, ldsW0 + (buf ^ 1) * WN_WFL)      // lands before the barrier below
        if (wave_active) {
            // ---- the 4x4 input pixels of this lane's tile, 4 channels each, and the row half of the transform (B^T d) ----
            f32x4 t[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[i][j] = *(const f32x4*)(pb + a_off + (i * PW + j) * WN_S);
            f32x4 wnx[WN_NB];
#pragma unroll
            for (int nb = 0; nb < WN_NB; ++nb) wnx[nb] = *(const f32x4*)(wb + nb * 64);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 d0 = t[0][j], d1 = t[1][j], d2 = t[2][j], d3 = t[3][j];
                t[0][j] = aesr_sub4(d0, d2);
                t[1][j] = d1 + d2;
                t[2][j] = aesr_sub4(d2, d1);
                t[3][j] = aesr_sub4(d1, d3);
            }
            // column half of the transform, one position ahead of the MFMAs that consume it: V[i][j] = (t[i] B)[j]
#define WN_V(i, j) ((j) == 0 ? aesr_sub4(t[i][0], t[i][2]) : (j) == 1 ? t[i][1] + t[i][2] : (j) == 2 ? aesr_sub4(t[i][2], t[i][1]) : aesr_sub4(t[i][1], t[i][3]))
            // (taking 0 / the bias as the C operand in the first chunk of an item instead of zeroing the accumulators -- what
            // conv_wino_res.hip does -- needs a second copy of this block: 4-17 registers spilled here, where the staging maps live)
            f32x4 vnx = WN_V(0, 0);
            auto positions = [&](auto I0c, auto I1c) {
#pragma unroll
                for (int i = decltype(I0c)::value; i < decltype(I1c)::value; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int xi = i * 4 + j;
                        // software pipeline, pinned with sched_barrier: the weight fragments and the transformed operand of position
                        // xi + 1 are produced BEFORE the 8 MFMAs of position xi (left alone, the scheduler sinks the LDS reads to
                        // just before their first use); inside the two groups the compiler's own order is kept (a fully
                        // hand-interleaved order pinned per MFMA and a sched_group_barrier deal-out both measured 0-3 % slower:
                        // profiles/r02_wino_experiments.txt)
                        f32x4 wc[WN_NB];
#pragma unroll
                        for (int nb = 0; nb < WN_NB; ++nb) wc[nb] = wnx[nb];
                        const f32x4 vc = vnx;
                        if (xi + 1 < 16) {
#pragma unroll
                            for (int nb = 0; nb < WN_NB; ++nb) wnx[nb] = *(const f32x4*)(wb + (xi + 1) * (4 * WN_TN * 4) + nb * 64);
                            vnx = WN_V((xi + 1) >> 2, (xi + 1) & 3);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int nb = 0; nb < WN_NB; ++nb) {
                                acc[xi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nb][r], vc[r], acc[xi][nb], 0, 0, 0);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            positions(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            if (do_load && stage_late) WN_STAGE(l_item, l_cc, ldsP0 + (buf ^ 1) * PPS, ldsW0 + (buf ^ 1) * WN_WFL)
            positions(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
#undef WN_V
        }
        WN_STAMP(0)
        // OUR wait, not the compiler's: the barrier below is what makes the other waves' DMAs (buffer_load ... lds) visible, which holds
        // only if every wave has waited for its own before arriving (the compiler emits this wait today; conv_wgrad_wino once raced
        // when it did not)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (stamp) { WN_STAMP(4) }
        __syncthreads();            // every wave is done with `buf`; the next chunk is complete in `buf ^ 1`
        WN_STAMP(1)
        buf ^= 1;
        // advance the load state
        if (do_load) {
            if (l_cc + 1 < nchunks) {
                ++l_cc;
            } else {
                l_item += G;
                l_cc = 0;
                if (l_item < nitems) WN_COMPUTE_GOFF(l_item)
            }
        }
        if (cc + 1 < nchunks) {
            ++cc;
            continue;
        }
        // ---- item finished: output transform Y = A^T M A, activation, (data gradient) derivative mask, store ----------
        if (wave_active) {
            const bool okt = tpix >= 0;
            const int n = cn0 + (tpix >> 20), y0 = 2 * (cty0 + ((tpix >> 10) & 1023)), x0 = 2 * (ctx0 + (tpix & 1023));
            const bool okn = okt && n < a.N;
            int ob[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    ob[p][q] = (okn && y0 + p < a.H && x0 + q < a.W) ? ((n * a.H + y0 + p) * a.W + x0 + q) * a.Cout * 4 : WN_OOB;
#pragma unroll
            for (int nb = 0; nb < WN_NB; ++nb) {
                const int co = co0 + nb * 16 + 4 * g;
                const int cob = co < a.Cout ? co * 4 : WN_OOB;
                f32x4 ys[2][2];
                if (MASK) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int q = 0; q < 2; ++q) ys[p][q] = wn_ld(rs_ys, ob[p][q] + cob);
                }
                f32x4 P[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    P[0][j] = acc[0 + j][nb] + acc[4 + j][nb] + acc[8 + j][nb];
                    P[1][j] = aesr_sub4(aesr_sub4(acc[4 + j][nb], acc[8 + j][nb]), acc[12 + j][nb]);
                }
                if (a.out_sum2) {
                    // adjoint of the nearest Upsample(x2) in front of this layer's forward: the 2x2 tile collapses to one pixel
                    // (the sum of A^T M A over its four entries = the corner combination below); no activation, no mask
                    const f32x4 s = aesr_sub4((P[0][0] + P[1][0]) + 2.f * (P[0][1] + P[1][1]), P[0][3] + P[1][3]);
                    const int obs = (okn && y0 < a.H && x0 < a.W) ? ((n * outH + (y0 >> 1)) * outW + (x0 >> 1)) * a.Cout * 4 : WN_OOB;
                    wn_st(rs_out, obs + cob, s);
                    continue;
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f32x4 Y[2];
                    Y[0] = P[p][0] + P[p][1] + P[p][2];
                    Y[1] = aesr_sub4(aesr_sub4(P[p][1], P[p][2]), P[p][3]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 o = Y[q];
                        // none / ReLU / LeakyReLU as ONE branch-free form, max(x, x * slope) for 0 <= slope <= 1; a per-element switch on
                        // the activation code costs a chain of uniform branches per element (~1000 instructions per item)
                        const f32x4 os = o * nslope;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], os[e]);
                        if (sigm) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] = 1.f / (1.f + expf(-o[e]));
                        }
                        if (MASK) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) o[e] *= (ys[p][q][e] > 0.f ? 1.f : mslope);
                        }
                        wn_st(rs_out, ob[p][q] + cob, o);
                    }
                }
            }
        }
        WN_STAMP(2)
        c_item += G;
        if (c_item >= nitems) break;
        cc = 0;
        WN_ITEM_ORIGIN(c_item, cn0, cty0, ctx0, co0)
        WN_INIT_ACC(co0)
        WN_STAMP(3)
    }
    if (stamp && lane == 0) {
        for (int k = 0; k < 5; ++k) a.dbgbuf[(blockIdx.x * 8 + wave) * 5 + k] = (float)tph[k];
    }
#undef WN_STAMP
#undef WN_INIT_ACC
#undef WN_STAGE
#undef WN_CHUNK_OFFS
#undef WN_COMPUTE_GOFF
#undef WN_ITEM_ORIGIN
#undef WN_DIV
}

// ---- weight transform + packing: wino_pack_elements lives in aesr_pack_dev.h (shared with prep.hip) ----

__global__ __launch_bounds__(256) void wino_pack_many_kernel(PackTable t) {
    int j = 0;
    for (int k = 1; k < t.njobs; ++k)
        if ((int)blockIdx.x >= t.job[k].block0) j = k;
    const PackJob& jb = t.job[j];
    const int b1 = (j + 1 < t.njobs) ? t.job[j + 1].block0 : t.nblocks;
    wino_pack_elements(jb.w, jb.p, jb.Cout, jb.Cin, jb.KinP, jb.NoutP, jb.transpose & 1,
                       (size_t)(blockIdx.x - jb.block0) * 256 + threadIdx.x, (size_t)(b1 - jb.block0) * 256);
}

int aesr_launch_wino_pack_many(const PackTable& t, hipStream_t st) {
    hipLaunchKernelGGL(wino_pack_many_kernel, dim3(t.nblocks), dim3(256), 0, st, t);
    AESR_LAUNCH_CHECK("wino_pack_many");
    return AESR_OK;
}

size_t aesr_wino_lds_bytes(int PP) { return ((size_t)2 * ceil_div(PP * 4, WN_NT) * WN_PIECE_FL + 2 * WN_WFL + 512) * sizeof(float); }

template <int NPP, bool MASK>
static int wino_launch_one(const WinoArgs& a, hipStream_t st) {
    const int PP = a.TI * (2 * a.THt + 2) * (2 * a.TWt + 2);
    const size_t shmem = aesr_wino_lds_bytes(PP);
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wino_f32<NPP, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wino_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    int grid = 256;                    // persistent: one 8-wave workgroup per CU
    if (const char* e = getenv("AESR_WINO_GRID")) grid = atoi(e);
    if (grid > a.nitems) grid = a.nitems;
    if (getenv("AESR_WINO_DBG")) {     // debug: per-phase cycle stamps, printed after a host sync
        static float* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 256 * 8 * 5 * sizeof(float));
        WinoArgs b = a;
        b.dbgbuf = dbuf;
        hipLaunchKernelGGL((conv_wino_f32<NPP, MASK>), dim3(grid), dim3(WN_NT), shmem, st, b);
        (void)hipStreamSynchronize(st);
        static float host[256 * 8 * 5];
        (void)hipMemcpy(host, dbuf, (size_t)grid * 8 * 5 * sizeof(float), hipMemcpyDeviceToHost);
        double s4[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < grid * 8; ++i) for (int k = 0; k < 5; ++k) s4[k] += host[i * 5 + k];
        const int nch = a.CinP / 16;
        const double items_per_wg = (double)a.nitems / grid;
        {
            double wv[8][2] = {};
            for (int i = 0; i < grid; ++i) for (int w = 0; w < 8; ++w) { wv[w][0] += host[(i * 8 + w) * 5 + 0]; wv[w][1] += host[(i * 8 + w) * 5 + 1]; }
            fprintf(stderr, "[wino stamps] per wave (compute/barrier kcycles per chunk):");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d %.1f/%.1f", w, wv[w][0] / grid / 1e3 / (items_per_wg * nch), wv[w][1] / grid / 1e3 / (items_per_wg * nch));
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[wino stamps] grid=%d items=%d (%.2f per WG, %d chunks each) per-wave kcycles: compute %.1f | barrier %.1f | epilogue %.1f | "
                "loop head %.1f || per chunk: compute %.2f dma-wait %.2f barrier %.2f, per item: epilogue %.2f\n", grid, a.nitems, items_per_wg, nch,
                s4[0] / grid / 8 / 1e3, s4[1] / grid / 8 / 1e3, s4[2] / grid / 8 / 1e3, s4[3] / grid / 8 / 1e3,
                s4[0] / grid / 8 / 1e3 / (items_per_wg * nch), s4[4] / grid / 8 / 1e3 / (items_per_wg * nch),
                s4[1] / grid / 8 / 1e3 / (items_per_wg * nch), s4[2] / grid / 8 / 1e3 / items_per_wg);
        return AESR_OK;
    }
    hipLaunchKernelGGL((conv_wino_f32<NPP, MASK>), dim3(grid), dim3(WN_NT), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wino_f32");
    return AESR_OK;
}

int aesr_launch_conv_wino(const WinoArgs& a_in, hipStream_t st) {
    WinoArgs a = a_in;
    if (const char* e = getenv("AESR_WINO_FLAGS")) a.flags = atoi(e);
    const int TP = a.TI * a.THt * a.TWt;
    const int PP = a.TI * (2 * a.THt + 2) * (2 * a.TWt + 2);
    if (TP < 1 || TP > 128) {
        aesr_set_error("conv_wino: %d Winograd tiles per work item (1..128)", TP);
        return AESR_ERR_ARG;
    }
    if (aesr_wino_lds_bytes(PP) > (size_t)160 * 1024 || PP * 4 > WN_NT * 5) {
        aesr_set_error("conv_wino: patch of %d pixels exceeds LDS / staging capacity", PP);
        return AESR_ERR_ARG;
    }
    if (a.CoutP % WN_TN != 0 || a.CinP % 16 != 0 || a.Cin % 4 != 0 || a.Cout % 4 != 0) {
        aesr_set_error("conv_wino: bad channel padding Cin=%d CinP=%d Cout=%d CoutP=%d", a.Cin, a.CinP, a.Cout, a.CoutP);
        return AESR_ERR_ARG;
    }
    if ((size_t)a.N * a.H * a.W * a.Cin >= (size_t)0x1C000000 || (size_t)a.N * a.H * a.W * a.Cout >= (size_t)0x1C000000) {
        aesr_set_error("conv_wino: tensors of 469M elements (1.75 GB) or more need 64-bit indexing (not built)");
        return AESR_ERR_UNSUPPORTED;
    }
    if (a.act == ACT_LRELU && !(a.slope >= 0.f && a.slope <= 1.f)) {
        aesr_set_error("conv_wino: the fused LeakyReLU is max(x, slope * x): slope %g is outside [0, 1] (use aesr_conv2d_fwd)", (double)a.slope);
        return AESR_ERR_UNSUPPORTED;
    }
    if (a.ysave && a.mask_act == ACT_SIGMOID) {
        aesr_set_error("conv_wino: a sigmoid derivative mask is not fused into the data gradient (use aesr_act_bwd)");
        return AESR_ERR_UNSUPPORTED;
    }
    if ((a.in_up2 || a.out_sum2) && ((a.H | a.W) & 1)) {
        aesr_set_error("conv_wino: the folded Upsample(x2) needs even convolution sizes (got %dx%d)", a.H, a.W);
        return AESR_ERR_ARG;
    }
    if (a.out_sum2 && (a.ysave || a.act != ACT_NONE || a.bias)) {
        aesr_set_error("conv_wino: the 2x2-summing epilogue takes no bias, activation or derivative mask");
        return AESR_ERR_ARG;
    }
    if (2 * a.THt + 2 > 1023 || 2 * a.TWt + 2 > 1023 || a.TI > 1023) {
        aesr_set_error("conv_wino: tile dimensions exceed the packed-coordinate range");
        return AESR_ERR_ARG;
    }
    if (aesr_wino_res_ok(a)) return aesr_launch_conv_wino_res(a, st);      // Cin <= 32: resident filter, independent waves (conv_wino_res.hip)
    if (aesr_wino_ring_takes(a)) return aesr_launch_conv_wino_ring(a, st);      // filter chunks through an LDS ring (conv_wino_ring.hip)
    if (a.post_scale) {
        aesr_set_error("conv_wino: the folded eval-mode BatchNorm epilogue exists in the resident-filter and ring kernels only (aesr_conv2d_wino_fwd_bn_supported)");
        return AESR_ERR_UNSUPPORTED;
    }
    a.regs_y = ceil_div(ceil_div(a.H, 2), a.THt);
    a.regs_x = ceil_div(ceil_div(a.W, 2), a.TWt);
    a.nitems = ceil_div(a.N, a.TI) * a.regs_y * a.regs_x * (a.CoutP / WN_TN);
    // magic numbers of the item decomposition: exact while item * divisor < 2^32
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + d - 1) / d); };
    a.m_ncot = magic(a.CoutP / WN_TN); a.m_regs_x = magic(a.regs_x); a.m_regs_y = magic(a.regs_y);
    if ((unsigned long long)a.nitems * (unsigned)(a.CoutP / WN_TN + a.regs_x + a.regs_y) >= ((unsigned long long)1 << 31)) {
        aesr_set_error("conv_wino: %d work items exceed the exact range of the item decomposition", a.nitems);
        return AESR_ERR_UNSUPPORTED;
    }
    const int NPP = ceil_div(PP * 4, WN_NT);
#define WN_CASE(npp)                                                                                     \
    if (NPP == npp) return a.ysave ? wino_launch_one<npp, true>(a, st) : wino_launch_one<npp, false>(a, st);
    WN_CASE(1) WN_CASE(2) WN_CASE(3) WN_CASE(4) WN_CASE(5)
#undef WN_CASE
    aesr_set_error("conv_wino: no instantiation for %d staging pieces", NPP);
    return AESR_ERR_UNSUPPORTED;
}
