// EXPERIMENT (round 6), stand-alone, NOT part of the library: the two-waves-per-SIMD bf16 x 3 form of the resident-filter Winograd kernel
// (csrc/conv_wino_res.hip) -- the second try after scripts/micro/conv_wino_res_b3_attempt.hip.  This one is CORRECT on every epilogue form and ragged size
// (rel-L2 1.9e-7 against fp64, the f32-MFMA kernel's own 2.0e-7) and compiles without a spilled register; measured 74 us against conv_wino_res_f32's 81 us
// on 32 -> 32 @ 160 x 160 x 36 (rocprofv3; 1.09 x, 1.00 - 1.07 x on 4 and 12 images): profiles/r06_bf16x3_second_try.txt.  Not integrated: see the verdict there.
//
// What was wrong with the first attempt, and what this file does instead:
//   * the patch DMA's lane limit (40 of 64 lanes write a 656-byte row) is set in EXEC inside ONE asm block per fetch.  As a C++ branch ("if (lane < 40)") it made
//     the compiler treat the item's coordinates and the image's buffer resource as lane-dependent: vector registers for all of them and a waterfall loop around
//     each of the ten loads -- that, not the MFMA loop, was where the attempt's ~50 spilled registers and much of its time came from;
//   * hand-waited LDS reads must reach their MFMA WITHOUT a register copy in between: the compiler does not know that the asm's output is pending, and a copy
//     (a phi between blocks, a tuple rebuilt by v_mov) placed before the s_waitcnt reads data that is not there yet -- the likely cause of the attempt's 1e-3
//     errors.  Reads and their waits sit in one block here, the waits take the operands as in / out, and scripts/micro/check_pending_reads.py walks the ISA
//     listing and reports any instruction that touches a register between its read and the wait that completes it (0 here);
//   * filter fragments by two ds_read_b64 per operand (2 + 2 LDS-array cycles per KB; ds_read2st64_b64 costs 8) straight into the MFMA operands
//     [U_h | U_l], [U_m | U_h], [U_h | U_m]; B3_DEPTH register sets in rotation over the use order (cout blocks innermost, so neighbouring MFMAs never share
//     an accumulator); two B operands per position ([V_h | V_m] serves hh + mm and mh + hm);
//   * the split on PAIRS of values (and, and, v_pk_add_f32; twice; three v_perm_b32): 18 + 2 instructions per position instead of 24, packed transforms;
//   * builtin MFMAs (the compiler tracks accumulators and hazards); wave-uniform coordinates forced into scalar registers.
// Compile-time switches: B3_DEPTH (3), B3_LATE (refill the set consumed one use earlier), B3_PIPE (split of position xi + 1 written between the MFMAs of xi);
// timing only, WRONG results: B3_NODMA, B3_NOLOOP, B3_NOSPLIT, B3_NOREAD, B3_NOMFMA.  None of the schedule switches moves the time (74 - 78 us).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -Isuperresolution_aniso_mri_amd/csrc -shared -fPIC \
//         scripts/micro/wino_b3v2.hip -o scripts/micro/libwino_b3v2.so ;  python scripts/micro/wino_b3v2.py
//
// Filter image: [chunk][cout tile][position 16][cout block 2][plane h, m, l][lane 64][4 bf16] (lane (l15, g): cout 32 tile + 16 block + l15, channels
// 16 chunk + 4 g .. + 3), packed by the driver.  Everything else (patch layout, transforms, epilogue, item order) is conv_wino_res.hip's.
#include <stdlib.h>
#ifndef B3_DEPTH
#define B3_DEPTH 3     // operand register sets in rotation = uses (MFMAs) the fragment reads run ahead
#endif

#include <type_traits>

#include "aesr_kernels.h"

constexpr int B3_NT = 512;                  // 8 independent waves, 2 per SIMD
constexpr int B3_RP = 164;                  // floats between patch rows: 10 pixels x 16 channels + 4
constexpr int B3_PFL = 10 * B3_RP;          // floats of a wave's patch buffer
constexpr int B3_CHUNK = 16 * 2 * 3 * 512;  // bytes of one 16-channel chunk of the split filter for 32 couts
constexpr int B3_OOB = 0x70000000;

typedef unsigned b3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned b3_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned b3_u32x6 __attribute__((ext_vector_type(6)));
typedef float b3_f32x2 __attribute__((ext_vector_type(2)));
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

// f32x4 sums as two PACKED fp32 operations (v_pk_add_f32: two values per lane and instruction at the single rate)
__device__ __forceinline__ f32x4 b3_add(f32x4 a, f32x4 b) {
    const b3_f32x2 lo = __builtin_shufflevector(a, a, 0, 1) + __builtin_shufflevector(b, b, 0, 1), hi = __builtin_shufflevector(a, a, 2, 3) + __builtin_shufflevector(b, b, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 b3_sub(f32x4 a, f32x4 b) {
    const b3_f32x2 lo = __builtin_shufflevector(a, a, 0, 1) - __builtin_shufflevector(b, b, 0, 1), hi = __builtin_shufflevector(a, a, 2, 3) - __builtin_shufflevector(b, b, 2, 3);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// the three B operands of a position: 4 transformed values of the lane's tile -> hi / mid / lo (truncation split: exact), packed in pairs
struct B3Ops { b3_u32x4 b1, b3; };
__device__ __forceinline__ B3Ops b3_split(f32x4 v) {
#ifdef B3_NOSPLIT      // timing experiment: no split arithmetic
    {
        B3Ops o_;
        o_.b1 = __builtin_bit_cast(b3_u32x4, v);
        o_.b3 = __builtin_bit_cast(b3_u32x4, v + v);
        return o_;
    }
#endif
    // per PAIR of values: and, and, packed subtract (twice), then one v_perm_b32 per term packs the upper halves -- 9 instructions per pair
    b3_u32x6 o6;        // [V_l | V_h | V_m]: the operands are overlapping 4-register runs of it -- no duplicated halves
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const b3_f32x2 x = p == 0 ? __builtin_shufflevector(v, v, 0, 1) : __builtin_shufflevector(v, v, 2, 3);
        const unsigned xa = __float_as_uint(x[0]), xb = __float_as_uint(x[1]);
        const b3_f32x2 hi = {__uint_as_float(xa & 0xffff0000u), __uint_as_float(xb & 0xffff0000u)};
        const b3_f32x2 r = x - hi;
        const unsigned ra = __float_as_uint(r[0]), rb = __float_as_uint(r[1]);
        const b3_f32x2 mi = {__uint_as_float(ra & 0xffff0000u), __uint_as_float(rb & 0xffff0000u)};
        const b3_f32x2 q = r - mi;
        o6[0 + p] = __builtin_amdgcn_perm(__float_as_uint(q[1]), __float_as_uint(q[0]), 0x07060302u);
        o6[2 + p] = __builtin_amdgcn_perm(xb, xa, 0x07060302u);
        o6[4 + p] = __builtin_amdgcn_perm(rb, ra, 0x07060302u);
    }
    B3Ops o;
    o.b3 = __builtin_shufflevector(o6, o6, 0, 1, 2, 3);      // [V_l | V_h]
    o.b1 = __builtin_shufflevector(o6, o6, 2, 3, 4, 5);      // [V_h | V_m]
    return o;
}
#define B3_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b3_bf16x8, a_), __builtin_bit_cast(b3_bf16x8, b_), acc_, 0, 0, 0)
// Two planes of a filter fragment (8 B per lane each, 512 B apart per plane) straight into one 4-register MFMA operand.  As asm, so that the compiler
// neither merges the three reads of a (position, cout block) differently nor rebuilds the operands with v_mov (338 per chunk when left to itself); the
// wait that follows takes the operands as in / out so that no consumer can be scheduled above it.
// (ds_read2st64_b64 costs 8 LDS-array cycles per KB, two ds_read_b64 cost 2 + 2: MI355X_MICROARCH.md, LDS table)
#define B3_RD1(dst_, addr_, o_) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(o_))
#define B3_WAIT2(n_, a_, b_) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a_), "+v"(b_) : "n"(n_))

template <int I, class F>
__device__ __forceinline__ void ww_unroll16_(F&& f) {
    if constexpr (I < 16) {
        f(std::integral_constant<int, I>{});
        ww_unroll16_<I + 1>(f);
    }
}
template <class F>
__device__ __forceinline__ void ww_unroll16(F&& f) { ww_unroll16_<0>(f); }
template <int N, int I = 0, class F>
__device__ __forceinline__ void b3_unroll(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        b3_unroll<N, I + 1>(f);
    }
}

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
    const int cot = __builtin_amdgcn_readfirstlane(xmap ? lw % ncot : blockIdx.x % ncot);      // (integer division runs on the vector unit)
    const int wgc = __builtin_amdgcn_readfirstlane(xmap ? lw / ncot : blockIdx.x / ncot);
    const int nwgc = __builtin_amdgcn_readfirstlane(xmap ? (gridDim.x >> 3) / ncot : gridDim.x / ncot);
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
    auto item_of = [&](int j) { return __builtin_amdgcn_readfirstlane(xmap ? xcd * (8 * nwgc) + wgc + nwgc * (j & 7) + 64 * nwgc * (j >> 3) : wgc + nwgc * j); };
    int slot = wave;
    int item = item_of(slot);
    int in_n = 0, in_y0 = 0, in_x0 = 0;
    auto locate = [&](int it) {
        in_n = __builtin_amdgcn_readfirstlane(B3_DIV(it, a.m_bpi));          // wave-uniform: kept in scalar registers
        const int rem = it - in_n * a.bpi;
        const int by = __builtin_amdgcn_readfirstlane(B3_DIV(rem, a.m_regs_x));
        in_y0 = by * 8;
        in_x0 = __builtin_amdgcn_readfirstlane((rem - by * a.regs_x) * 8);
    };
    auto fetch = [&](int cc) {
#ifdef B3_NODMA       // timing experiment: no patch traffic (the loop computes on whatever the LDS holds)
        if (a.N > 0) return;
#endif
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)in_n * inimg), 0, inimg, 0x00020000);
        const unsigned gx = (unsigned)(in_x0 - 1 + dpx);
        const int off = (gx < (unsigned)a.W && cc * 16 + 4 * dq < a.Cin) ? lcd + ((in_x0 >> sh) - 1 + sh) * a.Cin * 4 + cc * 64 : B3_OOB;
        // 10 pixels x 4 quads per row: the row is 656 B, so only lanes 0..39 may write (40..63 would land in the next row / the next wave's patch).
        // The lane limit is set in EXEC by hand inside one asm block: as a C++ branch it makes the compiler treat the item's coordinates and
        // the image's resource as lane-dependent (vector registers, a waterfall loop around each of the ten loads).
        int u[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) u[r] = ((in_y0 - 1 + r) >> sh) * inrow;           // row -1 stays negative: outside the image's resource
        const unsigned l0 = (unsigned)(unsigned long long)ldsP;
        int tmp;
        asm volatile(
            "s_mov_b32 exec_hi, 0xff\n\t"
            "s_mov_b32 m0, %[l0]\n\t"
            "v_add_u32 %[t], %[off], %[u0]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u1]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u2]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u3]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u4]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u5]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u6]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u7]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u8]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_add_u32 m0, m0, %[rp]\n\t"
            "v_add_u32 %[t], %[off], %[u9]\n\t"
            "buffer_load_dwordx4 %[t], %[rs], 0 offen lds\n\t"
            "s_mov_b32 exec_hi, -1"
            : [t] "=&v"(tmp)
            : [off] "v"(off), [rs] "s"(rs_in), [l0] "s"(l0), [rp] "n"(B3_RP * 4), [u0] "s"(u[0]), [u1] "s"(u[1]), [u2] "s"(u[2]), [u3] "s"(u[3]), [u4] "s"(u[4]),
              [u5] "s"(u[5]), [u6] "s"(u[6]), [u7] "s"(u[7]), [u8] "s"(u[8]), [u9] "s"(u[9])
            : "memory", "scc");      // (M0 too: the compiler sets it before each of its own uses)
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
        // filter fragments of a (position, cout block): [U_h | U_l], [U_m | U_h], [U_h | U_m] -- two planes each (8 B per lane and plane, 512 B apart);
        // two register sets: the reads of step s + 1 are in flight while the MFMAs of step s run (a step = one (position, cout block))
        const unsigned waddr = (unsigned)(unsigned long long)(wbl + cc * B3_CHUNK);
        b3_u32x2 fl[B3_DEPTH], fh[B3_DEPTH];
        // use u = 6 position + 2 product pair + block: planes of step (2 position + block): pair 0 = [U_h | U_l], 1 = [U_m | U_h], 2 = [U_h | U_m]
#define B3_FRAG(lo_, hi_, u_)                                                                                          \
    do {                                                                                                              \
        constexpr int s_ = 2 * ((u_) / 6) + ((u_) % 6 & 1), pr_ = ((u_) % 6) >> 1;                                      \
        B3_RD1(lo_, waddr, (s_ * 3 + (pr_ == 1 ? 1 : 0)) * 512);                                                       \
        B3_RD1(hi_, waddr, (s_ * 3 + (pr_ == 0 ? 2 : pr_ == 1 ? 0 : 1)) * 512);                                        \
    } while (0)
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
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 d0 = t[0][j], d1 = t[1][j], d2 = t[2][j], d3 = t[3][j];
            t[0][j] = b3_sub(d0, d2);
            t[1][j] = b3_add(d1, d2);
            t[2][j] = b3_sub(d2, d1);
            t[3][j] = b3_sub(d1, d3);
            __builtin_amdgcn_sched_barrier(0);              // one column at a time: the transform's temporaries stay at one tuple
        }
#define B3_V(i, j) ((j) == 0 ? b3_sub(t[i][0], t[i][2]) : (j) == 1 ? b3_add(t[i][1], t[i][2]) : (j) == 2 ? b3_sub(t[i][2], t[i][1]) : b3_sub(t[i][1], t[i][3]))
        auto positions = [&](auto firstc) {
            constexpr bool FIRST = decltype(firstc)::value;
            // (inside the branch: a read in the block before it would reach its MFMA through a register copy -- made BEFORE the wait, of data not there yet)
#ifdef B3_LATE
            b3_unroll<B3_DEPTH - 1>([&](auto dc) { B3_FRAG(fl[decltype(dc)::value], fh[decltype(dc)::value], decltype(dc)::value); });
#else
            b3_unroll<B3_DEPTH>([&](auto dc) { B3_FRAG(fl[decltype(dc)::value], fh[decltype(dc)::value], decltype(dc)::value); });
#endif
#ifdef B3_PIPE
            B3Ops vcur;
#endif
            ww_unroll16([&](auto xic) {
                constexpr int xi = decltype(xic)::value;
                __builtin_amdgcn_sched_barrier(0);          // nothing of a later position is hoisted over this one (register pressure)
#ifdef B3_PIPE
                // software pipeline: the split of position xi + 1 is written BETWEEN the MFMAs of position xi (a wave issues in order: behind the six MFMAs
                // it would wait for the matrix pipe, in front of them the MFMAs would wait for it), in six stages of 3 - 4 instructions
                if constexpr (xi == 0) vcur = b3_split(B3_V(0, 0));
                const B3Ops v = vcur;
                b3_f32x2 px, pr, pq;
                b3_u32x6 n6;
                f32x4 nv;
                auto stage = [&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (xi + 1 < 16) {
                        constexpr int xn = xi + 1;
                        if constexpr (k == 0) nv = B3_V(xn >> 2, xn & 3);
                        if constexpr (k == 0 || k == 3) {
                            px = k == 0 ? __builtin_shufflevector(nv, nv, 0, 1) : __builtin_shufflevector(nv, nv, 2, 3);
                            const b3_f32x2 hi = {__uint_as_float(__float_as_uint(px[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(px[1]) & 0xffff0000u)};
                            pr = px - hi;
                        }
                        if constexpr (k == 1 || k == 4) {
                            const b3_f32x2 mi = {__uint_as_float(__float_as_uint(pr[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(pr[1]) & 0xffff0000u)};
                            pq = pr - mi;
                        }
                        if constexpr (k == 2 || k == 5) {
                            constexpr int p_ = k == 2 ? 0 : 1;
                            n6[0 + p_] = __builtin_amdgcn_perm(__float_as_uint(pq[1]), __float_as_uint(pq[0]), 0x07060302u);
                            n6[2 + p_] = __builtin_amdgcn_perm(__float_as_uint(px[1]), __float_as_uint(px[0]), 0x07060302u);
                            n6[4 + p_] = __builtin_amdgcn_perm(__float_as_uint(pr[1]), __float_as_uint(pr[0]), 0x07060302u);
                        }
                        if constexpr (k == 5) {
                            vcur.b3 = __builtin_shufflevector(n6, n6, 0, 1, 2, 3);
                            vcur.b1 = __builtin_shufflevector(n6, n6, 2, 3, 4, 5);
                        }
                    }
                };
#else
                const B3Ops v = b3_split(B3_V(xi >> 2, xi & 3));
#endif
                // three operand registers in rotation: as soon as an MFMA has consumed one, the read of the same operand of the next step
                // (position, cout block) is issued into it -- always two newer reads in flight behind the one waited for
                // Use order of a position: (block 0, 1) x (hl + lh, mh + hm, hh + mm) with the BLOCKS innermost -- neighbouring MFMAs never share an accumulator.
                // Three operand register sets in rotation over that order: as soon as an MFMA has consumed one, the read for the use three further on is
                // issued into it, so two reads (of two ds_read_b64 each) are always in flight behind the one waited for.
                f32x4 c0 = acc[xi][0], c1 = acc[xi][1];
                if (FIRST) {
                    c0 = xi == 5 ? *(const f32x4*)(ldsBias + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    c1 = xi == 5 ? *(const f32x4*)(ldsBias + 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                b3_unroll<6>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, u = 6 * xi + k;
                    constexpr int left = 95 - u;                     // reads issued after this use's
                    b3_u32x2& lo = fl[u % B3_DEPTH];
                    b3_u32x2& hi = fh[u % B3_DEPTH];
#ifndef B3_NOREAD
#ifdef B3_LATE
                    B3_WAIT2(left >= B3_DEPTH - 2 ? 2 * (B3_DEPTH - 2) : 2 * left, lo, hi);
#else
                    B3_WAIT2(left >= B3_DEPTH - 1 ? 2 * (B3_DEPTH - 1) : 2 * left, lo, hi);
#endif
#else
                    if constexpr (u == 0) B3_WAIT2(0, lo, hi);
#endif
                    const b3_u32x4 fa = __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
#ifndef B3_NOMFMA
                    if constexpr ((k & 1) == 0) B3_MFMA(c0, fa, (k >> 1) == 0 ? v.b3 : v.b1);
                    else B3_MFMA(c1, fa, (k >> 1) == 0 ? v.b3 : v.b1);
#else
                    if constexpr (k == 4) c0[0] += __uint_as_float(fa[0] ^ v.b1[0] ^ v.b3[1]);
                    if constexpr (k == 5) c1[1] += __uint_as_float(fa[1] ^ v.b1[2] ^ v.b3[3]);
#endif
#ifndef B3_NOREAD
#ifdef B3_LATE          // refill the register set consumed ONE use ago (that of use u - 1, or the spare one at u = 0), not the one this MFMA is still reading
                    if constexpr (u + B3_DEPTH - 1 < 96) B3_FRAG(fl[(u + B3_DEPTH - 1) % B3_DEPTH], fh[(u + B3_DEPTH - 1) % B3_DEPTH], u + B3_DEPTH - 1);
#else
                    if constexpr (u + B3_DEPTH < 96) B3_FRAG(lo, hi, u + B3_DEPTH);
#endif
#endif
#ifdef B3_PIPE
                    stage(kc);
                    __builtin_amdgcn_sched_barrier(0);
#endif
                });
                acc[xi][0] = c0;
                acc[xi][1] = c1;
                __builtin_amdgcn_sched_barrier(0);
            });
        };
#ifdef B3_NOLOOP      // timing experiment: everything but the position loop (patch traffic, transforms' loads, epilogue, stores)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi][0] = acc[xi][1] = t[xi >> 2][xi & 3];
        (void)positions;
#else
        if (cc == 0) positions(std::true_type{});
        else positions(std::false_type{});
#endif
#undef B3_V
        after_stores = false;
        if (!last) {
            ++cc;
            continue;
        }
        cc = 0;
        // ---- item finished: output transform Y = A^T M A, activation, (data gradient) derivative mask, store: conv_wino_res.hip's epilogue ----
        // (the MFMAs are inline asm: the compiler's hazard recognizer does not know that accumulators were just written by the matrix pipe)
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

template <bool MASK>
static int b3_launch_one(const WinoArgs& a, const void* upk3, int grid_req, hipStream_t st) {
    const size_t shmem = b3_lds_bytes(a.CinP);
    if (hipFuncSetAttribute((const void*)conv_wino_res_b3<MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    const int ncot = a.CoutP / 32;
    int grid = (grid_req > 0 ? grid_req : 256) / ncot * ncot;
    const int per_cot = (a.nblk + 3) / 4;
    if (grid / ncot > per_cot) grid = per_cot * ncot;
    if (grid < ncot) grid = ncot;
    WinoArgs b = a;
    b.xcd_map = (grid % (8 * ncot) == 0 && a.nblk >= 8 * (grid / ncot)) ? 1 : 0;
    hipLaunchKernelGGL((conv_wino_res_b3<MASK>), dim3(grid), dim3(B3_NT), shmem, st, b, upk3);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// out = act(conv3x3(in, w) + bias) [* act'(ysave) with mask_act: the data-gradient form]; NHWC fp32; upk3: the split filter image
extern "C" int b3v2_conv(const float* in, const void* upk3, const float* bias, const float* ysave, float* out, int N, int H, int W, int Cin, int Cout, int act,
                         float slope, int mask_act, int in_up2, int out_sum2, int grid, void* stream) {
    WinoArgs a = {};
    a.in = in; a.bias = bias; a.ysave = ysave; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.CinP = (Cin + 15) / 16 * 16; a.CoutP = (Cout + 31) / 32 * 32;
    a.act = act; a.mask_act = mask_act; a.slope = slope; a.in_up2 = in_up2; a.out_sum2 = out_sum2;
    if (a.CinP > 32 || b3_lds_bytes(a.CinP) > (size_t)160 * 1024) return -1;
    a.regs_y = (H + 7) / 8; a.regs_x = (W + 7) / 8;
    a.bpi = a.regs_y * a.regs_x; a.nblk = N * a.bpi;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + d - 1) / d); };
    a.m_bpi = magic(a.bpi); a.m_regs_x = magic(a.regs_x);
    return ysave ? b3_launch_one<true>(a, upk3, grid, (hipStream_t)stream) : b3_launch_one<false>(a, upk3, grid, (hipStream_t)stream);
}
