// (SUPERSEDED by scripts/micro/wino_b3v2.hip, which is correct and 1.05 - 1.15 x: the spills and the errors below came from the "if (lane < 40)" around the
// patch DMA -- lane-dependent resources, waterfall loops -- and from register copies of hand-waited reads placed before their wait: profiles/r06_bf16x3_second_try.txt)
// ATTEMPT, NOT BUILT (round 6): the two-waves-per-SIMD bf16 x 3 form of conv_wino_res.hip INSIDE the library (dispatch from aesr_launch_conv_wino_res, the split
// filter image written by wino_pack_elements behind the f32 one, kernel id 4) -- tried and reverted the same day; kept for whoever picks it up.  Status:
//   * the plain-C++ form of this file (builtin MFMAs, C++ fragment loads) was CORRECT through the C ABI (tests/test_gpu_kernels.py: 301 passed) and SLOWER than
//     conv_wino_res_f32: 125 against 78 us on 32 -> 32 @ 160 x 160 x 36 -- the allocator spilled accumulator tuples around every MFMA group, and scratch fills
//     wait on vmcnt(0), i.e. on the patch DMA in flight;
//   * THIS form (accumulators pinned to the accumulation registers by "a" constraints, fragments by ds_read2st64_b64 asm, hand-counted lgkmcnt) removes those
//     spills but measured 139-159 us and is NOT correct (1.4e-3 .. 3e-3) -- do not build it as it is.  Most likely cause of both: ~50 registers still spill
//     (constants in the prologue, the epilogue), and scratch stores / fills are vector-memory operations -- they count in vmcnt, so the hand-counted
//     "s_waitcnt vmcnt(8)" at the top of an item (exactly 8 stores behind the patch DMAs) no longer guarantees that the patch has landed, while the
//     compiler's own vmcnt(0) in front of every fill waits for the patch DMA in flight.  A version of this kernel is only worth timing at ZERO spills.
// Why neither pays: the item loop came out at ~3 100 instructions per item (6 v_mov per position for the duplicated operand halves, 128 v_accvgpr_read, waits,
// address work) against ~1 850 in the timing skeleton (scripts/micro/bf16x3.hip, mode w32-pair: 1.34 x) and ~850 in conv_wino_res_f32, whose 256 f32 MFMAs
// fill 8 192 of its 13 800 cycles by themselves: on the bf16 pipe the kernel is bound by instruction ISSUE, and the compiler's schedule is 1.7 x the skeleton's
// count.  The form needs hand-written assembly to get under ~2 200 instructions per item (= 1.3 x).  profiles/EXPERIMENTS.md, round 6.
//
// Winograd F(2x2, 3x3) convolution, resident filter, K side <= 32 channels, 32 output channels per workgroup -- conv_wino_res.hip's kernel with the
// matrix work on the BF16 pipe: fp32 arithmetic by a THREE-TERM SPLIT of both operands (round 6; scripts/micro/bf16x3.hip, wino_b3_proto.hip).
//
//   x = hi + mid + lo, every term a bf16 (8 significant bits each, fp32's exponent range): the split of an fp32 number is EXACT.  A product needs the six
//   terms hh, hm, mh, hl, lh, mm (the dropped ml, lm, ll are 2^-27 of it and below), accumulated in fp32 by v_mfma_f32_16x16x32_bf16: measured rel-L2
//   2.5e-7 against fp64 on K = 288 dot products, 1.4e-7 on a whole convolution -- at least as close as v_mfma_f32_16x16x4_f32 (3.1e-7 / 2.0e-7).
//
// Why: the f32 MFMA runs at 1/16 of the bf16 rate AND blocks its SIMD's vector issue for all of its 32 cycles, so the Winograd transforms beside it bound
// the f32 kernels; a bf16 MFMA holds the issue for 8 of its 16 cycles.  What makes the form FIT (two waves per SIMD, 256 registers,
// 160 KB of LDS) is that one K = 32 instruction carries TWO of the six products of a 16-channel chunk: lane (l15, g) supplies k-slots 8 g .. 8 g + 7 =
// [4 channels of operand P | the same 4 channels of operand Q], so with A = [U_h | U_m], A' = [U_h | U_l] from LDS and B1 = [V_h | V_m], B2 = [V_m | V_h],
// B3 = [V_l | V_h] from the split
//        A' . B3 = hl + lh        A . B2 = hm + mh        A . B1 = hh + mm
// -- three MFMAs of 16 cycles per (chunk, position, 16 couts) instead of four f32 MFMAs of 32, a lane still holds 4 channels of its tile (64 patch
// registers, as in conv_wino_res.hip), and the chunk structure, the per-wave patch DMA, the transforms and the epilogue are that kernel's.
//   * filter: U = G g G^T split by the weight preparation (aesr_pack_dev.h: wino_pack_elements writes this image behind the f32 one) as
//     [chunk][cout tile][position 16][cout block 2][plane h, m, l][lane 64][4 bf16]; a workgroup keeps its cout tile's chunks in LDS: 48 KB per chunk;
//   * patches: 10 rows x 10 pixels x 16 channels per wave, rows pitched 656 B (= 41 x 16 B; with the channel-quad swap in pixels 4..7 of the DMA source the
//     16 lanes of a ds_read_b128 phase hit 16 different bank groups), DMA on 40 lanes: 8 x 6.4 KB + 96 KB of filter = 148 KB;
//   * per (chunk, position): V (4 channels of the lane's tile) -> and, sub, and, sub per value + v_perm_b32 packing -> B1, B2, B3.
// Timing skeleton 1.34 x conv_wino_res_f32's (8 250 against 11 050 cycles of SIMD time per item, profiles/r06_bf16x3_experiment.txt).
//
// Replaces the ATen/cuDNN conv2d calls behind networks/acai_vanilla.py:55-56,87-88,96 (the 32-channel layers) forward and as data gradient, where
// conv_wino_res_f32<32> ran them; AESR_WINO_B3=0 keeps that kernel.
#include <stdlib.h>

#include <type_traits>

#include "aesr_kernels.h"

constexpr int B3_NT = 512;                  // 8 independent waves, 2 per SIMD
constexpr int B3_RP = 164;                  // floats between patch rows: 10 pixels x 16 channels + 4
constexpr int B3_PFL = 10 * B3_RP;          // floats of a wave's patch buffer
constexpr int B3_CHUNK = 16 * 2 * 3 * 512;  // bytes of one 16-channel chunk of the split filter for 32 couts
constexpr int B3_OOB = 0x70000000;

typedef unsigned b3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void b3_dma(__amdgpu_buffer_rsrc_t rs, void* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ f32x4 b3_ld(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
__device__ __forceinline__ void b3_st(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

// the three B operands of a position: 4 transformed values of the lane's tile -> hi / mid / lo (truncation split: exact), packed in pairs
struct B3Ops { b3_u32x4 b1, b2, b3; };
__device__ __forceinline__ B3Ops b3_split(f32x4 v) {
    unsigned xr[4], rr[4], r2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xr[k] = __float_as_uint(v[k]);
        const float r = v[k] - __uint_as_float(xr[k] & 0xffff0000u);
        rr[k] = __float_as_uint(r);
        const float q = r - __uint_as_float(rr[k] & 0xffff0000u);
        r2[k] = __float_as_uint(q);
    }
    const unsigned h0 = __builtin_amdgcn_perm(xr[1], xr[0], 0x07060302u), h1 = __builtin_amdgcn_perm(xr[3], xr[2], 0x07060302u);
    const unsigned m0 = __builtin_amdgcn_perm(rr[1], rr[0], 0x07060302u), m1 = __builtin_amdgcn_perm(rr[3], rr[2], 0x07060302u);
    const unsigned l0 = __builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), l1 = __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u);
    B3Ops o;
    o.b1 = (b3_u32x4){h0, h1, m0, m1};
    o.b2 = (b3_u32x4){m0, m1, h0, h1};
    o.b3 = (b3_u32x4){l0, l1, h0, h1};
    return o;
}
// The 128 accumulators of a wave are pinned to the ACCUMULATION registers ("a" constraints, as conv_wgrad_wino.hip does): with two waves per SIMD a wave
// has 256 registers, and left to itself the allocator spilled accumulator tuples to scratch around every MFMA group -- whose fills wait on vmcnt(0),
// i.e. on the patch DMA in flight.  The vector side (128 registers) then holds the patch (64), the filter fragments (16), the split operands (12).
#define B3_MFMA(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))
#define B3_MFMA0(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc_) : "v"(a_), "v"(b_))
// two planes of a filter fragment (8 B per lane each, 512 B apart per plane) straight into one 4-register operand
#define B3_RD2(dst_, addr_, o0_, o1_) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(dst_) : "v"(addr_), "n"(o0_), "n"(o1_))
#define B3_LGKM(n_) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n_) : "memory")

template <int I, class F>
__device__ __forceinline__ void ww_unroll16_(F&& f) {
    if constexpr (I < 16) {
        f(std::integral_constant<int, I>{});
        ww_unroll16_<I + 1>(f);
    }
}
template <class F>
__device__ __forceinline__ void ww_unroll16(F&& f) { ww_unroll16_<0>(f); }

template <bool MASK>
__global__ __launch_bounds__(B3_NT, 2) void conv_wino_res_b3(WinoArgs a, const void* upk3) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int ncot = a.CoutP / 32, nchunks = a.CinP >> 4;
    // workgroup -> (cout tile, spatial worker), XCD-aware where the grid allows: conv_wino_res.hip
    const bool xmap = a.xcd_map != 0;
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int cot = xmap ? lw % ncot : blockIdx.x % ncot;
    const int wgc = xmap ? lw / ncot : blockIdx.x / ncot;
    const int nwgc = xmap ? (gridDim.x >> 3) / ncot : gridDim.x / ncot;
    const int co0 = cot * 32;
    char* const ldsW = (char*)lds;                                                  // [chunk][position][cout block][plane][lane][8 B]
    float* const ldsP = lds + nchunks * (B3_CHUNK / 4) + wave * B3_PFL;             // this wave's patch
    float* const ldsBias = lds + nchunks * (B3_CHUNK / 4) + 8 * B3_PFL;             // [32]

    const int sh = a.in_up2 ? 1 : 0;
    const int inH = a.H >> sh, inW = a.W >> sh;
    const bool halfout = a.out_sum2 != 0;
    const int outH = halfout ? a.H >> 1 : a.H, outW = halfout ? a.W >> 1 : a.W;
    const int inimg = inH * inW * a.Cin * 4, inrow = inW * a.Cin * 4;
    const int wbytes = __builtin_amdgcn_readfirstlane(nchunks * ncot * B3_CHUNK);
    const int obytes = __builtin_amdgcn_readfirstlane(a.N * outH * outW * a.Cout * 4), ybytes = __builtin_amdgcn_readfirstlane(a.N * a.H * a.W * a.Cout * 4);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)upk3, 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ys = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK ? a.ysave : a.out), 0, ybytes, 0x00020000);

    // ---- prologue: the workgroup's split filter (48 KB per chunk) and bias, once ----
    float bias_v = 0.f;
    if (tid < 32 && a.bias && co0 + tid < a.Cout) bias_v = a.bias[co0 + tid];
    for (int cc = 0; cc < nchunks; ++cc) {
        const int gbase = (cc * ncot + cot) * B3_CHUNK;
#pragma unroll
        for (int j = 0; j < B3_CHUNK / 1024 / 8; ++j) {
            const int kb = wave + 8 * j;
            b3_dma(rs_w, ldsW + cc * B3_CHUNK + kb * 1024, gbase + kb * 1024 + lane * 16);
        }
    }

    // ---- per-lane maps ----
    // DMA: lane -> pixel slot lane >> 2 of a patch row (lanes 0..39 = 10 pixels), channel quad (lane & 3) ^ ((slot >> 2) & 1)
    const int dpx = lane >> 2, dq = (lane & 3) ^ ((dpx >> 2) & 1);
    const int lcd = (((dpx - sh) >> sh) * a.Cin + 4 * dq) * 4;
    const int ty = l15 >> 2, tx = l15 & 3;
    const int offA = (2 * ty) * B3_RP + (2 * tx) * 16 + ((g ^ (tx >> 1)) << 2);                  // columns j = 0, 1
    const int offB = (2 * ty) * B3_RP + (2 * tx) * 16 + ((g ^ (((2 * tx + 2) >> 2) & 1)) << 2);  // columns j = 2, 3
    const char* const wbl = ldsW + lane * 8;                                // + chunk * B3_CHUNK + ((xi * 2 + nb) * 3 + plane) * 512

    const float mslope = a.mask_act == ACT_LRELU ? a.slope : (a.mask_act == ACT_RELU ? 0.f : 1.f);
    const float nslope = a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU ? 0.f : 1.f);
    const bool sigm = a.act == ACT_SIGMOID;

#define B3_DIV(x, m) ((m) ? (int)__umulhi((unsigned)(x), (m)) : (int)(x))
    auto item_of = [&](int j) { return xmap ? xcd * (8 * nwgc) + wgc + nwgc * (j & 7) + 64 * nwgc * (j >> 3) : wgc + nwgc * j; };
    int slot = wave;
    int item = item_of(slot);
    int in_n = 0, in_y0 = 0, in_x0 = 0;
    auto locate = [&](int it) {
        in_n = B3_DIV(it, a.m_bpi);
        const int rem = it - in_n * a.bpi;
        const int by = B3_DIV(rem, a.m_regs_x);
        in_y0 = by * 8;
        in_x0 = (rem - by * a.regs_x) * 8;
    };
    auto fetch = [&](int cc) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)in_n * inimg), 0, inimg, 0x00020000);
        const unsigned gx = (unsigned)(in_x0 - 1 + dpx);
        const int off = (gx < (unsigned)a.W && cc * 16 + 4 * dq < a.Cin) ? lcd + ((in_x0 >> sh) - 1 + sh) * a.Cin * 4 + cc * 64 : B3_OOB;
        if (lane < 40) {                                    // 10 pixels x 4 quads: the row is 656 B, lanes 40..63 would write into the next row / wave
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                const int urow = ((in_y0 - 1 + r) >> sh) * inrow;
                b3_dma(rs_in, ldsP + r * B3_RP, off + urow);
            }
        }
    };
    if (item < a.nblk) {
        locate(item);
        fetch(0);
    }
    if (tid < 32) ldsBias[tid] = bias_v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[16][2];
    int cc = 0;
    bool after_stores = false;
    while (item < a.nblk) {
        if (after_stores) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = *(const f32x4*)(ldsP + (j < 2 ? offA : offB) + i * B3_RP + j * 16);
        const char* wb = wbl + cc * B3_CHUNK;
        // filter fragments of a (position, cout block): A = [U_h | U_m] and A' = [U_h | U_l], one ds_read2st64_b64 each
        b3_u32x4 fa1[2], fa3[2];
        const unsigned waddr = (unsigned)(unsigned long long)(wbl + cc * B3_CHUNK);
#define B3_FRAGS(xi_, nb_)                                                                  \
    do {                                                                                   \
        B3_RD2(fa1[nb_], waddr, ((xi_) * 2 + (nb_)) * 3, ((xi_) * 2 + (nb_)) * 3 + 1);      \
        B3_RD2(fa3[nb_], waddr, ((xi_) * 2 + (nb_)) * 3, ((xi_) * 2 + (nb_)) * 3 + 2);      \
    } while (0)
        B3_FRAGS(0, 0);
        B3_FRAGS(0, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the patch is in registers: request the next one (next chunk, or chunk 0 of the next item) into the same buffer
        const int cur_n = in_n, cur_y0 = in_y0, cur_x0 = in_x0;
        const bool last = cc + 1 == nchunks;
        if (last) {
            slot += 8;
            item = item_of(slot);
            if (item < a.nblk) {
                locate(item);
                fetch(0);
            }
        } else {
            fetch(cc + 1);
        }
        // row half of the transform (B^T d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 d0 = t[0][j], d1 = t[1][j], d2 = t[2][j], d3 = t[3][j];
            t[0][j] = d0 - d2;
            t[1][j] = d1 + d2;
            t[2][j] = d2 - d1;
            t[3][j] = d1 - d3;
        }
#define B3_V(i, j) ((j) == 0 ? t[i][0] - t[i][2] : (j) == 1 ? t[i][1] + t[i][2] : (j) == 2 ? t[i][2] - t[i][1] : t[i][1] - t[i][3])
        auto positions = [&](auto firstc) {
            constexpr bool FIRST = decltype(firstc)::value;
            // No software pipelining inside the wave: with two waves per SIMD the partner's vector work fills the issue slots this wave's MFMAs leave
            // free (8 of 16 cycles each).  The fragments of (xi + 1, nb) are requested as soon as the MFMAs of (xi, nb) have been issued; at the top of
            // a position the four reads of (xi, 0) and (xi, 1) are in flight, oldest first.
            ww_unroll16([&](auto xic) {
                constexpr int xi = decltype(xic)::value;
                __builtin_amdgcn_sched_barrier(0);          // nothing of a later position is hoisted over this one (register pressure)
                const B3Ops v = b3_split(B3_V(xi >> 2, xi & 3));
                if (FIRST && xi == 5) {
                    acc[xi][0] = *(const f32x4*)(ldsBias + 4 * g);
                    acc[xi][1] = *(const f32x4*)(ldsBias + 16 + 4 * g);
                    B3_LGKM(0);
                } else {
                    B3_LGKM(2);                         // (xi, 0) has landed
                }
                if (FIRST && xi != 5) B3_MFMA0(acc[xi][0], fa3[0], v.b3);
                else B3_MFMA(acc[xi][0], fa3[0], v.b3);             // hl + lh
                B3_MFMA(acc[xi][0], fa1[0], v.b2);                  // hm + mh
                B3_MFMA(acc[xi][0], fa1[0], v.b1);                  // hh + mm
                if constexpr (xi + 1 < 16) {
                    B3_FRAGS(xi + 1, 0);
                    B3_LGKM(2);                         // (xi, 1) has landed
                } else {
                    B3_LGKM(0);
                }
                if (FIRST && xi != 5) B3_MFMA0(acc[xi][1], fa3[1], v.b3);
                else B3_MFMA(acc[xi][1], fa3[1], v.b3);
                B3_MFMA(acc[xi][1], fa1[1], v.b2);
                B3_MFMA(acc[xi][1], fa1[1], v.b1);
                if constexpr (xi + 1 < 16) B3_FRAGS(xi + 1, 1);
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if (cc == 0) positions(std::true_type{});
        else positions(std::false_type{});
#undef B3_V
        after_stores = false;
        if (!last) {
            ++cc;
            continue;
        }
        cc = 0;
        // ---- item finished: output transform Y = A^T M A, activation, (data gradient) derivative mask, store: conv_wino_res.hip's epilogue ----
        // (the MFMAs are inline asm: the compiler's hazard recognizer does not know that accumulators were just written by the matrix pipe)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        {
            const int y0 = cur_y0 + 2 * ty, x0 = cur_x0 + 2 * tx;
            int ob[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    ob[p][q] = (y0 + p < a.H && x0 + q < a.W) ? ((cur_n * a.H + y0 + p) * a.W + x0 + q) * a.Cout * 4 : B3_OOB;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = co0 + nb * 16 + 4 * g;
                const int cob = co < a.Cout ? co * 4 : B3_OOB;
                f32x4 ys[2][2];
                if (MASK) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int q = 0; q < 2; ++q) ys[p][q] = b3_ld(rs_ys, ob[p][q] + cob);
                }
                f32x4 P[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    P[0][j] = acc[0 + j][nb] + acc[4 + j][nb] + acc[8 + j][nb];
                    P[1][j] = acc[4 + j][nb] - acc[8 + j][nb] - acc[12 + j][nb];
                }
                if (a.out_sum2) {
                    const f32x4 s = (P[0][0] + P[1][0]) + 2.f * (P[0][1] + P[1][1]) - (P[0][3] + P[1][3]);
                    const int obs = (y0 < a.H && x0 < a.W) ? ((cur_n * outH + (y0 >> 1)) * outW + (x0 >> 1)) * a.Cout * 4 : B3_OOB;
                    b3_st(rs_out, obs + cob, s);
                    continue;
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f32x4 Y[2];
                    Y[0] = P[p][0] + P[p][1] + P[p][2];
                    Y[1] = P[p][1] - P[p][2] - P[p][3];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 o = Y[q];
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
                        b3_st(rs_out, ob[p][q] + cob, o);
                    }
                }
            }
        }
        after_stores = !MASK && !halfout;           // exactly 8 stores follow the next patch's DMAs
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no DMA may still be writing this workgroup's LDS when it is released
#undef B3_DIV
}

static size_t b3_lds_bytes(int CinP) { return ((size_t)(CinP >> 4) * (B3_CHUNK / 4) + 8 * B3_PFL + 32 + 4) * sizeof(float); }

bool aesr_wino_res_b3_enabled() {
    static const int on = getenv("AESR_WINO_B3") ? atoi(getenv("AESR_WINO_B3")) : 1;
    return on != 0;
}

template <bool MASK>
static int b3_launch_one(const WinoArgs& a, const void* upk3, hipStream_t st) {
    const size_t shmem = b3_lds_bytes(a.CinP);
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wino_res_b3<MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wino_res_b3: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    const int ncot = a.CoutP / 32;
    int grid = 256 / ncot * ncot;
    if (const char* e = getenv("AESR_WINO_GRID")) grid = atoi(e) / ncot * ncot;
    static const int wpw = getenv("AESR_WINO_RES_WPW") ? atoi(getenv("AESR_WINO_RES_WPW")) : 4;         // small layers: conv_wino_res.hip's rule
    const int per_cot = ceil_div(a.nblk, wpw >= 1 && wpw <= 8 ? wpw : 4);
    if (grid / ncot > per_cot) grid = per_cot * ncot;
    if (grid < ncot) grid = ncot;
    WinoArgs b = a;
    static const int xmap_on = getenv("AESR_WINO_XCD") ? atoi(getenv("AESR_WINO_XCD")) : 1;
    b.xcd_map = (xmap_on && grid % (8 * ncot) == 0 && a.nblk >= 8 * (grid / ncot)) ? 1 : 0;
    hipLaunchKernelGGL((conv_wino_res_b3<MASK>), dim3(grid), dim3(B3_NT), shmem, st, b, upk3);
    AESR_LAUNCH_CHECK("conv_wino_res_b3");
    return AESR_OK;
}

// called by aesr_launch_conv_wino_res (which has filled the block decomposition) for 32-cout workgroups with a K side of <= 32 channels
int aesr_launch_conv_wino_res_b3(const WinoArgs& a, hipStream_t st) {
    if (a.CinP > 32 || a.CoutP % 32 != 0 || a.post_scale) {
        aesr_set_error("conv_wino_res_b3: K side of %d channels / folded BatchNorm epilogue are not this kernel's", a.CinP);
        return AESR_ERR_ARG;
    }
    if (b3_lds_bytes(a.CinP) > (size_t)160 * 1024) return AESR_ERR_ARG;
    // the split filter image lies behind the f32 one in the packed buffer (aesr_conv2d_wino_packed_floats)
    const void* upk3 = (const void*)(a.upk + (size_t)16 * a.CinP * a.CoutP);
    return a.ysave ? b3_launch_one<true>(a, upk3, st) : b3_launch_one<false>(a, upk3, st);
}
