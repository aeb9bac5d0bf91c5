// Winograd F(2x2, 3x3) convolution for layers with FEW input channels (Cin <= 64): the transformed filter stays resident in LDS
// and every wave runs on its own -- no barrier after the prologue.
//
// The general kernel (conv_wino.hip) streams 16-channel chunks of patch + U through LDS for an 8-wave work item and meets at a
// barrier per chunk.  With Cin = 32 an item is only 2 chunks long, so the item epilogue (output transform, activation, stores,
// re-zeroing 128 accumulators, the next item's address work) -- during which all 8 waves leave the matrix pipe idle TOGETHER --
// was 30 % of the kernel (AESR_WINO_DBG stamps: 7.9 k of 27 k cycles per item), and on gfx950 the f32 MFMA shares its pipe with the
// vector ALU (scripts/micro/mfma_gap.hip), so nothing hides inside a wave either.  Here:
//  * all of U for the workgroup's output channels (64 KB: 32 couts x <= 32 input channels, or 16 couts x <= 64) is loaded ONCE per
//    workgroup; workgroup b serves the cout tile b % ncot for its whole life;
//  * a work item belongs to ONE wave: 4 x 4 Winograd tiles (8 x 8 outputs) x 32 output channels, the same 128 accumulators per
//    wave as in conv_wino.hip.  The wave fetches its own 10 x 10-pixel patch by DMA (buffer_load ... lds, one instruction per
//    patch row of 16 pixel slots, 4 instructions of address work each) into its own 10 KB of LDS and waits only on its own
//    vmcnt: the patch buffer is free again as soon as its 16 ds_read_b128 have returned (the raw tile lives in registers while
//    it is transformed), so the NEXT chunk's patch is requested before this chunk's 128 MFMAs start -- single buffered, a whole
//    chunk of matrix work ahead;
//  * the two waves of a SIMD drift apart (the older one is served first), so one's epilogue, stores and mask loads run beside
//    the other's MFMAs without any scheduling effort;
//  * patch rows are pitched 1040 B and the DMA source swaps channel quads q <-> q ^ 1 in pixels 4..7 and 12..15 of a row: the
//    16 lanes of a ds_read_b128 phase (16 tiles, one channel quad) then hit 16 different 16-byte bank groups -- conflict free
//    (conv_wino.hip pays 4-way conflicts for its register-free DMA).
// Everything else (U packing, the per-lane transforms, the bias in the accumulator of position (1,1), the derivative mask, the
// folded Upsample in both directions) is conv_wino.hip's; results are bit-identical in exact arithmetic and equal to rounding.
//
// Replaces the ATen/cuDNN conv2d calls behind networks/acai_vanilla.py:55-56,87-88,96 (the 32-channel layers) forward and as
// data gradient.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "aesr_kernels.h"

// experiment switches (scripts/variants.py builds variant libraries with -D...; the shipped build leaves them at their defaults)
#ifndef WR_EXP_PRIO
#define WR_EXP_PRIO 0       // s_setprio value OUTSIDE the MFMA runs (patch wait, LDS reads, DMA issue, row transform, epilogue); 0 = never touched
#endif
#ifndef WR_EXP_PRIO_MFMA
#define WR_EXP_PRIO_MFMA 0  // s_setprio value INSIDE the MFMA runs
#endif
#ifndef WR_EXP_STAGGER
#define WR_EXP_STAGGER 0    // wave w starts its items w * WR_EXP_STAGGER * 64 cycles behind the prologue barrier
#endif
#ifndef WR_EXP_PAIR
#define WR_EXP_PAIR 1       // 16-cout workgroups: MFMAs of two positions interleaved (a lone accumulator chain waits 40 cycles per MFMA, not 32)
#endif
#ifndef WR_DYN
#define WR_DYN 0            // 1: items of a workgroup dealt to its waves through an LDS counter (measured: +1 % on the biggest layer, -1 % on small ones, step flat)
#endif
#ifndef WR_EXP_HALF
#define WR_EXP_HALF 0       // 1: only waves 0..3 of a workgroup take items (one active wave per SIMD, the same code): what does the second wave buy?
#endif
#ifndef WR_EXP_SALU
#define WR_EXP_SALU 0       // n > 0: n extra scalar instructions in every patch fetch (two fetches per item): does a scalar instruction cost a wave time?
#endif
#ifndef WR_ABL
#define WR_ABL 0            // timing-only ablations (WRONG results): 1 epilogue without its stores, 2 epilogue stores raw accumulators (no transform / activation)
#endif
#ifndef WR_EXP_SPLITPRO
#define WR_EXP_SPLITPRO 0   // prologue: start on filter chunk 0 + first patch, meet again for the rest of the filter before chunk 1
#endif

constexpr int WR_NT = 512;          // threads per workgroup: 8 independent waves, 2 per SIMD
// TN = output channels of a workgroup: 32 (K side <= 32 channels: <= 64 KB of filter) or 16 (K side <= 64 channels: 64 KB)
constexpr int WR_RP = 260;          // floats between patch rows: 16 pixel slots x 16 channels + 4 (shifts a row by one 16-byte bank group)
constexpr int WR_PFL = 10 * WR_RP;  // floats of a wave's patch buffer (10 rows)
constexpr int WR_OOB = 0x70000000;

__device__ __forceinline__ void wr_dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ f32x4 wr_ld(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
__device__ __forceinline__ void wr_st(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

// POST: eval-mode BatchNorm (a per-channel affine behind the activation) and the AvgPool2d(2) that follows it, in the epilogue -- a lane
// holds one 2 x 2 output tile, which IS one pooling window; an instantiation of its own, so that the training kernels' code is untouched
// STAMP (debug, AESR_WINO_RES_DBG=1): wall-clock stamps (s_memrealtime, 10 ns) of the phases of every wave's FIRST item -> a.dbgbuf as
// [workgroup][wave][16] 64-bit ticks: 0 entry, 1 DMAs of the prologue issued, 2 behind the prologue barrier, 3 + 2 c patch of chunk c landed,
// 4 + 2 c MFMAs of chunk c issued (c < 4), 11 stores of the item issued, 12 exit, 13 items of the wave
template <int WR_TN, bool MASK, bool POST = false, bool STAMP = false>
__global__ __launch_bounds__(WR_NT, 2) void conv_wino_res_f32(WinoArgs a) {
    constexpr int WR_NB = WR_TN / 16;
    constexpr int WR_WFL = 16 * 4 * WR_TN * 4;      // floats of one U chunk (16 positions x 16 ci x TN co)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long stamp[14] = {};
    int stamped_items = 0;
#define WR_STAMP(k)                                                          \
    if constexpr (STAMP) {                                                   \
        if (stamped_items == 0) stamp[k] = __builtin_amdgcn_s_memrealtime(); \
    }
    WR_STAMP(0)
    const int l15 = lane & 15, g = lane >> 4;
    const int ncot = a.CoutP / WR_TN, nchunks = a.CinP >> 4;
    // Workgroup -> (cout tile, spatial worker).  Workgroups b and b + 8 share an XCD (round-robin placement: speed only, never
    // correctness): where the grid allows, XCD x takes a CONTIGUOUS run of blocks per round and all cout tiles of a block sit on it,
    // so the 10 x 10 patches of neighbouring blocks (1.56 x the image) and the re-reads by the other cout tiles hit that XCD's L2
    // instead of crossing the fabric (round 2 PMC: 2.4 x the algorithmic bytes read).  Otherwise: cout tile = b % ncot as before.
    const bool xmap = a.xcd_map != 0;
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3;
    const int cot = xmap ? lw % ncot : blockIdx.x % ncot;
    const int wgc = xmap ? lw / ncot : blockIdx.x / ncot;               // spatial worker (inside the XCD / of the grid)
    const int nwgc = xmap ? (gridDim.x >> 3) / ncot : gridDim.x / ncot;
    const int co0 = cot * WR_TN;
    float* const ldsW = lds;                                        // [nchunks][16 positions][4][32][4]
    float* const ldsP = lds + nchunks * WR_WFL + wave * WR_PFL;     // this wave's patch: [10 rows][16 pixel slots][16 ch] (+ pitch)
    float* const ldsBias = lds + nchunks * WR_WFL + 8 * WR_PFL;     // [32]
    int* const ldsNext = (int*)(ldsBias + 32);                      // the workgroup's next undealt item slot (WR_DYN)

    const int sh = a.in_up2 ? 1 : 0;
    const int inH = a.H >> sh, inW = a.W >> sh;                                              // stored size of the input tensor
    const bool halfout = a.out_sum2 || (POST && a.post_pool);
    const int outH = halfout ? a.H >> 1 : a.H, outW = halfout ? a.W >> 1 : a.W;              // stored size of the output tensor
    const int inimg = inH * inW * a.Cin * 4, inrow = inW * a.Cin * 4;                        // bytes of an input image / row
    // (sizes through readfirstlane: a 64-bit product lands in vector registers, and a resource there costs a waterfall loop per access)
    const int wbytes = __builtin_amdgcn_readfirstlane(16 * a.CinP * a.CoutP * 4);
    const int obytes = __builtin_amdgcn_readfirstlane(a.N * outH * outW * a.Cout * 4), ybytes = __builtin_amdgcn_readfirstlane(a.N * a.H * a.W * a.Cout * 4);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.upk, 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ys = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK ? a.ysave : a.out), 0, ybytes, 0x00020000);

    // ---- prologue: the workgroup's U block(s) and bias, once ----
    // (the bias is only REQUESTED here: parked in LDS behind the DMA issue below -- written first, its global round trip (~1 us on a cold
    // line) stood in front of every DMA of the launch: stamps of profiles/r04_small_shard_budget.txt, section 11)
    float bias_v = 0.f;
    if (tid < WR_TN && a.bias && co0 + tid < a.Cout) bias_v = a.bias[co0 + tid];
    // packed layout [chunk][32-cout tile][position][ci / 4][32 couts][4]: a 16-cout workgroup takes one half of every 32-cout row
    auto filter_chunk = [&](int cc) {
        const int wbase = (int)(((size_t)cc * (a.CoutP / 32) + (co0 >> 5)) * (8192 * 4)) + ((co0 >> 4) & 1) * (WR_TN == 16 ? 256 : 0);
#pragma unroll
        for (int j = 0; j < WR_WFL / 4 / WR_NT; ++j) {
            const int pc = tid + WR_NT * j;                        // 16-byte piece -> (row of TN couts x 4, piece in the row)
            wr_dma(rs_w, ldsW + cc * WR_WFL + j * (WR_NT * 4) + wave * 256, wbase + ((pc / WR_TN) * 128 + (pc % WR_TN) * 4) * 4);
        }
    };
    // (WR_EXP_SPLITPRO: only chunk 0 of the filter stands in front of the first patch; the rest is requested behind it and met at a second
    // barrier in front of the first chunk-1 MFMAs, so the first 128 MFMAs of every wave run while 32 KB of filter are still on their way)
    for (int cc = 0; cc < (WR_EXP_SPLITPRO ? 1 : nchunks); ++cc) filter_chunk(cc);

    // ---- per-lane maps ----
    // DMA: lane -> pixel slot lane >> 2 of a patch row, channel quad (lane & 3) ^ ((slot >> 2) & 1)
    const int dpx = lane >> 2, dq = (lane & 3) ^ ((dpx >> 2) & 1);
    const int lcd = (((dpx - sh) >> sh) * a.Cin + 4 * dq) * 4;          // bytes from (row start + item column origin); arithmetic shift: -1 stays -1
    // MFMA B operand: lane -> tile l15 = (ty, tx) of the 4 x 4 block, channel quad g; patch pixel (2 ty + i, 2 tx + j)
    const int ty = l15 >> 2, tx = l15 & 3;
    const int offA = (2 * ty) * WR_RP + (2 * tx) * 16 + ((g ^ (tx >> 1)) << 2);                  // columns j = 0, 1
    const int offB = (2 * ty) * WR_RP + (2 * tx) * 16 + ((g ^ (((2 * tx + 2) >> 2) & 1)) << 2);  // columns j = 2, 3
    const float* wbl = ldsW + (g * WR_TN + l15) * 4;                    // + chunk * WFL + xi * 512 + nb * 64

    const float mslope = a.mask_act == ACT_LRELU ? a.slope : (a.mask_act == ACT_RELU ? 0.f : 1.f);
    const float nslope = a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU ? 0.f : 1.f);      // ACT_NONE / ACT_SIGMOID: identity
    const bool sigm = a.act == ACT_SIGMOID;

    // item k of this wave: block wgc + nwgc * (wave + 8 k) of the N x BY x BX blocks of 8 x 8 outputs (a partial last round lands on
    // wave 0 of many workgroups, not on all waves of a few)
#define WR_DIV(x, m) ((m) ? (int)__umulhi((unsigned)(x), (m)) : (int)(x))          /* m == 0: divisor 1 */
    // XCD map: round k, XCD x = blocks [(8 k + x) * 8 nwgc, + 8 nwgc), wave w of worker s takes block w * nwgc + s of them
    // The workgroup's items in the order they are dealt: slot j -> block; wave w takes slots w, w + 8, ...  The two waves of a SIMD do NOT progress
    // alike -- the older one (waves 0..3) is served first and ran its equally long list in 74 us where waves 4..7 took 90 (stamps of 32 -> 32 @
    // 162 x 162 x 36, profiles/r05_res_wave_exit.txt) -- but dealing the slots through an LDS counter (WR_DYN=1: a wave takes the next undealt slot
    // when it finishes an item) bought only 1 % there and cost 1 % on small layers: the SIMD finishes items at one rate whichever wave owns them; the
    // younger wave simply runs faster once it is alone.  Slots are monotonic in the block index, so the first slot past the end ends a wave.
    auto item_of = [&](int j) { return xmap ? xcd * (8 * nwgc) + wgc + nwgc * (j & 7) + 64 * nwgc * (j >> 3) : wgc + nwgc * j; };
    int item = item_of(wave);
    [[maybe_unused]] int slot = wave;
    if (WR_EXP_HALF && wave >= 4) item = a.nblk;            // experiment: the younger wave of every SIMD stays idle
    if (WR_DYN && tid == 0) *ldsNext = 8;                   // visible behind the prologue barrier
    // the patch of (item, chunk) -> this wave's LDS buffer: 10 DMAs; out-of-range rows / columns deliver zeros
    int in_n = 0, in_y0 = 0, in_x0 = 0;
    auto locate = [&](int it) {
        in_n = WR_DIV(it, a.m_bpi);
        const int rem = it - in_n * a.bpi;
        const int by = WR_DIV(rem, a.m_regs_x);
        in_y0 = by * 8;
        in_x0 = (rem - by * a.regs_x) * 8;
    };
    auto fetch = [&](int, int cc) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)in_n * inimg), 0, inimg, 0x00020000);
        if constexpr (WR_EXP_SALU > 0) {
            int dummy_ = cc;
            asm volatile(".rept %1\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(dummy_) : "i"(WR_EXP_SALU));
            if (dummy_ == 0x7fffffff) return;
        }
        const unsigned gx = (unsigned)(in_x0 - 1 + dpx);
        const int off = (dpx < 10 && gx < (unsigned)a.W && cc * 16 + 4 * dq < a.Cin) ? lcd + ((in_x0 >> sh) - 1 + sh) * a.Cin * 4 + cc * 64 : WR_OOB;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const int urow = ((in_y0 - 1 + r) >> sh) * inrow;           // row -1 stays -1: outside the image's resource
            wr_dma(rs_in, ldsP + r * WR_RP, off + urow);
        }
    };
    if (item < a.nblk) {
        locate(item);
        fetch(item, 0);
    }
    WR_STAMP(1)
    bool second_barrier = false;            // uniform over the workgroup
    if constexpr (WR_EXP_SPLITPRO) {
        for (int cc = 1; cc < nchunks; ++cc) filter_chunk(cc);
        second_barrier = nchunks > 1;
    }
    if (tid < WR_TN) ldsBias[tid] = bias_v;
    if constexpr (WR_EXP_SPLITPRO) {
        // all but this wave's (nchunks - 1) * WR_WFL / 4 / WR_NT youngest DMAs: chunk 0 of the filter and the first patch
        constexpr int PER = WR_WFL / 4 / WR_NT;
        switch (nchunks) {
            case 1: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); break;
        }
        if (nchunks > 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // explicit: this wave's share of the filter DMAs has landed before it arrives
    }
    __syncthreads();            // U (chunk 0 at least) and bias are in LDS (every wave waited for its own part)
    WR_STAMP(2)
    if constexpr (WR_EXP_STAGGER > 0) {
        for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(WR_EXP_STAGGER);
    }

    f32x4 acc[16][WR_NB];
    int cc = 0;
    bool after_stores = false;
    while (item < a.nblk) {
        // ---- the 4x4 input pixels of this lane's tile, 4 channels each ----
        // the DMAs of this patch are the oldest outstanding memory operations; the previous item's stores may still be in flight
        if constexpr (WR_EXP_PRIO != WR_EXP_PRIO_MFMA) __builtin_amdgcn_s_setprio(WR_EXP_PRIO);
        int next_slot = 0;
        if constexpr (WR_DYN != 0) {
            // last chunk of the item: ask for the next slot now, one lane, so that the answer returns with the patch reads below
            if (cc + 1 == nchunks && lane == 0) next_slot = __hip_atomic_fetch_add(ldsNext, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (after_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * WR_NB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (WR_EXP_SPLITPRO) {
            if (second_barrier && cc == 1) {        // first item only; vmcnt(0) above: this wave's share of the later filter chunks has landed
                __syncthreads();
                second_barrier = false;
            }
        }
        if constexpr (STAMP) {
            if (stamped_items == 0 && cc < 4) stamp[3 + 2 * cc] = __builtin_amdgcn_s_memrealtime();
        }
        f32x4 t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) t[i][j] = *(const f32x4*)(ldsP + (j < 2 ? offA : offB) + i * WR_RP + j * 16);
        const float* wb = wbl + cc * WR_WFL;
        // PP positions per MFMA group: with ONE accumulator register set per position (16-cout workgroups) the four MFMAs of a position
        // form a dependent chain -- 40 cycles each instead of the 32 of independent ones -- so two positions are interleaved there
        constexpr int PP = (WR_NB == 1 && WR_EXP_PAIR) ? 2 : 1;
        f32x4 wnx[PP][WR_NB];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
#pragma unroll
            for (int nb = 0; nb < WR_NB; ++nb) wnx[pp][nb] = *(const f32x4*)(wb + pp * (4 * WR_TN * 4) + nb * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the patch is in registers: request the next one (next chunk, or chunk 0 of the next item) into the same buffer
        const int cur_item = item, cur_n = in_n, cur_y0 = in_y0, cur_x0 = in_x0;
        const bool last = cc + 1 == nchunks;
        if (last) {
            if constexpr (WR_DYN != 0) {
                slot = __builtin_amdgcn_readfirstlane(next_slot);
            } else {
                slot += WR_EXP_HALF ? 4 : 8;
            }
            item = item_of(slot);
            if (item < a.nblk) {
                locate(item);
                fetch(item, 0);
            }
        } else {
            fetch(item, cc + 1);
        }
        (void)cur_item;
        // row half of the transform (B^T d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 d0 = t[0][j], d1 = t[1][j], d2 = t[2][j], d3 = t[3][j];
            t[0][j] = aesr_sub4(d0, d2);
            t[1][j] = d1 + d2;
            t[2][j] = aesr_sub4(d2, d1);
            t[3][j] = aesr_sub4(d1, d3);
        }
        // column half, one position ahead of the MFMAs that consume it: V[i][j] = (t[i] B)[j].  In the FIRST chunk of an item the
        // MFMA of each accumulator's first use takes 0 (the bias in position (1,1)) as its C operand: nothing is zeroed between items
#define WR_V(i, j) ((j) == 0 ? aesr_sub4(t[i][0], t[i][2]) : (j) == 1 ? t[i][1] + t[i][2] : (j) == 2 ? aesr_sub4(t[i][2], t[i][1]) : aesr_sub4(t[i][1], t[i][3]))
        auto positions = [&](auto firstc) {
            constexpr bool FIRST = decltype(firstc)::value;
            f32x4 vnx[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) vnx[pp] = WR_V(pp >> 2, pp & 3);
#pragma unroll
            for (int grp = 0; grp < 16 / PP; ++grp) {
                f32x4 wc[PP][WR_NB], vc[PP];
#pragma unroll
                for (int pp = 0; pp < PP; ++pp) {
                    vc[pp] = vnx[pp];
#pragma unroll
                    for (int nb = 0; nb < WR_NB; ++nb) wc[pp][nb] = wnx[pp][nb];
                }
                if (grp + 1 < 16 / PP) {
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp) {
                        const int xn = (grp + 1) * PP + pp;
#pragma unroll
                        for (int nb = 0; nb < WR_NB; ++nb) wnx[pp][nb] = *(const f32x4*)(wb + xn * (4 * WR_TN * 4) + nb * 64);
                        vnx[pp] = WR_V(xn >> 2, xn & 3);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                        for (int nb = 0; nb < WR_NB; ++nb) {
                            const int xi = grp * PP + pp;
                            f32x4 c = acc[xi][nb];
                            if (FIRST && r == 0) c = xi == 5 ? *(const f32x4*)(ldsBias + nb * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                            acc[xi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[pp][nb][r], vc[pp][r], c, 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (WR_EXP_PRIO != WR_EXP_PRIO_MFMA) __builtin_amdgcn_s_setprio(WR_EXP_PRIO_MFMA);
        if (cc == 0) positions(std::true_type{});
        else positions(std::false_type{});
#undef WR_V
        if constexpr (STAMP) {
            if (stamped_items == 0 && cc < 4) stamp[4 + 2 * cc] = __builtin_amdgcn_s_memrealtime();
        }
        after_stores = false;
        if (!last) {
            ++cc;
            continue;
        }
        cc = 0;
        // ---- item finished: output transform Y = A^T M A, activation, (data gradient) derivative mask, store ----------
        {
            const int y0 = cur_y0 + 2 * ty, x0 = cur_x0 + 2 * tx;
            int ob[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    ob[p][q] = (y0 + p < a.H && x0 + q < a.W) ? ((cur_n * a.H + y0 + p) * a.W + x0 + q) * a.Cout * 4 : WR_OOB;
#pragma unroll
            for (int nb = 0; nb < WR_NB; ++nb) {
                const int co = co0 + nb * 16 + 4 * g;
                const int cob = co < a.Cout ? co * 4 : WR_OOB;
                f32x4 ys[2][2];
                if (MASK) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int q = 0; q < 2; ++q) ys[p][q] = wr_ld(rs_ys, ob[p][q] + cob);
                }
                f32x4 P[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    P[0][j] = acc[0 + j][nb] + acc[4 + j][nb] + acc[8 + j][nb];
                    P[1][j] = aesr_sub4(aesr_sub4(acc[4 + j][nb], acc[8 + j][nb]), acc[12 + j][nb]);
                }
                if (a.out_sum2) {
                    // adjoint of the nearest Upsample(x2) in front of this layer's forward: the 2x2 tile collapses to one pixel
                    const f32x4 s = aesr_sub4((P[0][0] + P[1][0]) + 2.f * (P[0][1] + P[1][1]), P[0][3] + P[1][3]);
                    const int obs = (y0 < a.H && x0 < a.W) ? ((cur_n * outH + (y0 >> 1)) * outW + (x0 >> 1)) * a.Cout * 4 : WR_OOB;
                    wr_st(rs_out, obs + cob, s);
                    continue;
                }
                f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f}, prow = psh, pm = psh;      // pooled: row sums as they come (8 registers, not 16)
                if (POST && co < a.Cout) {
                    psc = *(const f32x4*)(a.post_scale + co);
                    psh = *(const f32x4*)(a.post_shift + co);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    f32x4 Y[2];
                    Y[0] = P[p][0] + P[p][1] + P[p][2];
                    Y[1] = aesr_sub4(aesr_sub4(P[p][1], P[p][2]), P[p][3]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 o = Y[q];
                        // none / ReLU / LeakyReLU as ONE branch-free form: max(x, x * slope) for 0 <= slope <= 1
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
                        if (POST) {
                            if (a.post_pool) prow = q == 0 ? o : prow + o;
                            else wr_st(rs_out, ob[p][q] + cob, o * psc + psh);              // bn.hip bn_apply: v * scale + shift
                        } else {
#if WR_ABL == 1
                            if (a.slope == 12345.f) wr_st(rs_out, ob[p][q] + cob, o);          // never true: the arithmetic stays, the store does not issue
#elif WR_ABL == 2
                            wr_st(rs_out, ob[p][q] + cob, acc[p * 2 + q][nb]);
#else
                            wr_st(rs_out, ob[p][q] + cob, o);
#endif
                        }
                    }
                    if (POST && a.post_pool) pm = p == 0 ? prow : pm + prow;
                }
                if (POST && a.post_pool) {
                    // AvgPool2d(2) of the activated tile, then the affine: the arithmetic and order of bn.hip's bn_apply (pooling mode);
                    // a window that sticks out of an odd image has no output (floor)
                    const f32x4 m = pm * 0.25f;                     // ((o00 + o01) + (o10 + o11)) * 0.25
                    const int obs = (y0 + 1 < a.H && x0 + 1 < a.W) ? ((cur_n * outH + (y0 >> 1)) * outW + (x0 >> 1)) * a.Cout * 4 : WR_OOB;
                    wr_st(rs_out, obs + cob, m * psc + psh);
                }
            }
        }
        after_stores = !MASK && !halfout;           // exactly 4 NB stores follow the next patch's DMAs
        WR_STAMP(11)
        if constexpr (STAMP) ++stamped_items;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no DMA may still be writing this workgroup's LDS when it is released
    if constexpr (WR_EXP_SPLITPRO) {
        if (second_barrier) __syncthreads();        // a wave without items still owes the others its share of the filter
    }
    if constexpr (STAMP) {
        stamp[12] = __builtin_amdgcn_s_memrealtime();
        stamp[13] = (unsigned long long)stamped_items;
        if (lane == 0) {
            unsigned long long* dst = (unsigned long long*)a.dbgbuf + ((size_t)blockIdx.x * 8 + wave) * 16;
#pragma unroll
            for (int k = 0; k < 14; ++k) dst[k] = stamp[k];
        }
    }
#undef WR_STAMP
#undef WR_DIV
}

// Output channels per workgroup.  64 K-side channels only fit with 16; with <= 32 both do, and the choice is about balance: a
// 16-cout work item is 0.55 of a 32-cout one (half the MFMAs, the same transforms), so it wins when it saves rounds of
// items over the 2 048 waves -- at 6 images of 160 x 160: 2 400 items = 2 rounds, or 4 800 half items = 3 x 0.55 = 1.65
static int wino_res_tn(const WinoArgs& a) {
    if (a.CinP > 32) return 16;
    static const int forced = getenv("AESR_WINO_RES_TN") ? atoi(getenv("AESR_WINO_RES_TN")) : 0;
    if (forced == 16 || forced == 32) return forced;
    const long nblk = (long)a.N * ceil_div(a.H, 8) * ceil_div(a.W, 8);
    const double c32 = (double)((nblk * (a.CoutP / 32) + 2047) / 2048), c16 = 0.55 * (double)((nblk * (a.CoutP / 16) + 2047) / 2048);
    return c16 < c32 ? 16 : 32;
}
static size_t wino_res_lds_bytes(int CinP, int TN) { return ((size_t)(CinP >> 4) * (256 * TN) + 8 * WR_PFL + 32 + 4) * sizeof(float); }

bool aesr_wino_res_ok(const WinoArgs& a) {
    // AESR_WINO_RES: 0 = never, 1 = K side <= 32 channels only, unset / 2 = up to 64 channels (16-cout workgroups)
    static const int level = getenv("AESR_WINO_RES") ? atoi(getenv("AESR_WINO_RES")) : 2;
    if (level <= 0 || a.CinP % 16 != 0 || a.CoutP % 32 != 0 || a.CoutP / 16 > 256) return false;
    if (a.CinP <= 32) return true;
    // 16-cout workgroups pay twice the transform work per MFMA: worth it only where the 8 x 8-output blocks tile the image with
    // little waste (measured: 40 x 40 and 80 x 80 layers 5-13 % faster than the streamed kernel, 81 x 81 (+18 % padding) 5 % slower)
    const double waste = (double)(ceil_div(a.H, 8) * 8) * (ceil_div(a.W, 8) * 8) / ((double)a.H * a.W);
    return level >= 2 && a.CinP <= 64 && waste <= 1.10;
}

// debug launch (AESR_WINO_RES_DBG=1, forward without mask): the stamped instantiation, then a host sync and ONE line on stderr -- where the
// time of a launch goes for the waves that had work (profiles/r04_small_shard_budget.txt, section 11)
template <int WR_TN>
static int wino_res_stamped(const WinoArgs& b_in, int grid, size_t shmem, hipStream_t st) {
    static unsigned long long* dbuf = nullptr;
    static std::vector<unsigned long long> host;
    const size_t n = (size_t)grid * 8 * 16;
    if (grid > 1024) return AESR_ERR_ARG;
    if (!dbuf && hipMalloc(&dbuf, (size_t)1024 * 8 * 16 * sizeof(unsigned long long)) != hipSuccess) return AESR_ERR_HIP;
    (void)hipFuncSetAttribute((const void*)conv_wino_res_f32<WR_TN, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    WinoArgs b = b_in;
    b.dbgbuf = (float*)dbuf;
    (void)hipMemsetAsync(dbuf, 0, n * sizeof(unsigned long long), st);
    hipLaunchKernelGGL((conv_wino_res_f32<WR_TN, false, false, true>), dim3(grid), dim3(WR_NT), shmem, st, b);
    AESR_LAUNCH_CHECK("conv_wino_res_f32 (stamped)");
    (void)hipStreamSynchronize(st);
    host.resize(n);
    (void)hipMemcpy(host.data(), dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long t_first = ~0ull, t_last = 0;
    for (size_t w = 0; w < (size_t)grid * 8; ++w) {
        const unsigned long long* s_ = &host[w * 16];
        if (s_[0] && s_[0] < t_first) t_first = s_[0];
        if (s_[12] > t_last) t_last = s_[12];
    }
    const int nch = b.CinP / 16 < 4 ? b.CinP / 16 : 4;
    double sum[16] = {}, ramp_max = 0;
    int busy = 0, items = 0;
    for (size_t w = 0; w < (size_t)grid * 8; ++w) {
        const unsigned long long* s_ = &host[w * 16];
        if (!s_[13]) continue;
        ++busy;
        items += (int)s_[13];
        const double ramp = (double)(s_[0] - t_first);
        if (ramp > ramp_max) ramp_max = ramp;
        sum[0] += ramp;                                     // entry behind the first wave of the grid
        sum[1] += (double)(s_[1] - s_[0]);                  // address work + issue of the filter and first-patch DMAs
        sum[2] += (double)(s_[2] - s_[1]);                  // wait for them + the barrier
        unsigned long long prev = s_[2];
        for (int c = 0; c < nch; ++c) {
            sum[3 + 2 * c] += (double)(s_[3 + 2 * c] - prev);               // wait for the patch of chunk c
            sum[4 + 2 * c] += (double)(s_[4 + 2 * c] - s_[3 + 2 * c]);      // LDS reads, next DMAs, transform, MFMAs of chunk c
            prev = s_[4 + 2 * c];
        }
        sum[11] += (double)(s_[11] - prev);                 // output transform, activation, stores issued
        sum[12] += (double)(s_[12] - s_[11]);               // further items of the wave + drain of the stores
    }
    const double q = busy ? 0.01 / busy : 0.0;              // ticks of 10 ns -> us per busy wave
    fprintf(stderr, "[wino-res stamps] %d->%d N=%d %dx%d TN=%d grid=%d: %d of %d waves busy, %d items | span %.2f us | per busy wave (first item), us: "
            "entry behind first wave %.2f (max %.2f) | DMA issue %.2f | prologue wait+barrier %.2f |", b.Cin, b.Cout, b.N, b.H, b.W, WR_TN, grid, busy,
            grid * 8, items, (double)(t_last - t_first) * 0.01, sum[0] * q, ramp_max * 0.01, sum[1] * q, sum[2] * q);
    for (int c = 0; c < nch; ++c) fprintf(stderr, " chunk %d: wait %.2f work %.2f |", c, sum[3 + 2 * c] * q, sum[4 + 2 * c] * q);
    fprintf(stderr, " epilogue %.2f | rest of the wave %.2f\n", sum[11] * q, sum[12] * q);
    // do the eight waves of a workgroup (wave w and w + 4 share a SIMD) finish their -- equally long -- item lists at the same time?
    double ex[8] = {}, ex_min[8], ex_max[8] = {};
    int nw[8] = {};
    for (int w = 0; w < 8; ++w) ex_min[w] = 1e30;
    for (size_t w = 0; w < (size_t)grid * 8; ++w) {
        const unsigned long long* s_ = &host[w * 16];
        if (!s_[13]) continue;
        const double e = (double)(s_[12] - t_first) * 0.01;
        const int k = (int)(w & 7);
        ex[k] += e; ++nw[k];
        if (e < ex_min[k]) ex_min[k] = e;
        if (e > ex_max[k]) ex_max[k] = e;
    }
    fprintf(stderr, "[wino-res stamps] exit of wave w behind the grid's first entry, us (mean [min .. max] over the workgroups):");
    for (int w = 0; w < 8; ++w)
        if (nw[w]) fprintf(stderr, "  w%d %.1f [%.1f .. %.1f]", w, ex[w] / nw[w], ex_min[w], ex_max[w]);
    fprintf(stderr, "\n");
    return AESR_OK;
}

template <int WR_TN, bool MASK, bool POST = false>
static int wino_res_launch_one(const WinoArgs& a, hipStream_t st) {
    const size_t shmem = wino_res_lds_bytes(a.CinP, WR_TN);
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wino_res_f32<WR_TN, MASK, POST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wino_res_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    const int ncot = a.CoutP / WR_TN;
    int grid = 256 / ncot * ncot;                          // one workgroup per CU, a whole number of them per cout tile
    if (const char* e = getenv("AESR_WINO_GRID")) grid = atoi(e) / ncot * ncot;
    // Few blocks (a small data-parallel shard, the deep layers): at most FOUR per workgroup.  Waves 0..3 of a workgroup sit on the four
    // SIMDs of its CU (wave i -> SIMD i % 4: measured, scripts/res_stamps.py), so four one-item waves each have a matrix pipe to themselves;
    // the round-2 rule packed eight per workgroup and two one-item waves shared every SIMD (2.2 -> 1.5 us per 16-channel chunk at 6 images).
    static const int wpw = getenv("AESR_WINO_RES_WPW") ? atoi(getenv("AESR_WINO_RES_WPW")) : 4;
    const int per_cot = ceil_div(a.nblk, wpw >= 1 && wpw <= 8 ? wpw : 4);
    if (grid / ncot > per_cot) grid = per_cot * ncot;
    if (grid < ncot) grid = ncot;
    WinoArgs b = a;
    static const int xmap_on = getenv("AESR_WINO_XCD") ? atoi(getenv("AESR_WINO_XCD")) : 1;
    // (the XCD map hands every XCD a contiguous run of 8 x its workers blocks per round: with fewer blocks than waves it would fill the first
    // XCDs' workgroups with eight blocks each and leave the others idle)
    b.xcd_map = (xmap_on && grid % (8 * ncot) == 0 && a.nblk >= 8 * (grid / ncot)) ? 1 : 0;
    if constexpr (!MASK && !POST) {
        if (getenv("AESR_WINO_RES_DBG")) return wino_res_stamped<WR_TN>(b, grid, shmem, st);
    }
    hipLaunchKernelGGL((conv_wino_res_f32<WR_TN, MASK, POST>), dim3(grid), dim3(WR_NT), shmem, st, b);
    AESR_LAUNCH_CHECK("conv_wino_res_f32");
    return AESR_OK;
}

// called by aesr_launch_conv_wino (which has validated the arguments) when aesr_wino_res_ok(a)
int aesr_launch_conv_wino_res(const WinoArgs& a_in, hipStream_t st) {
    WinoArgs a = a_in;
    a.TI = 1; a.THt = 4; a.TWt = 4;
    a.regs_y = ceil_div(a.H, 8);
    a.regs_x = ceil_div(a.W, 8);
    a.bpi = a.regs_y * a.regs_x;
    a.nblk = a.N * a.bpi;
    const int TN = wino_res_tn(a);
    a.nitems = a.nblk * (a.CoutP / TN);
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + d - 1) / d); };
    a.m_bpi = magic(a.bpi); a.m_regs_x = magic(a.regs_x);
    // the wave's item index runs up to nblk + 8 * 256 past the end before it is compared: exactness of the multiply-high division
    if (((unsigned long long)a.nblk + 4096) * (unsigned)(a.bpi + a.regs_x) >= ((unsigned long long)1 << 31) || (size_t)a.H * a.W * a.Cin * 4 >= (size_t)0x10000000) {
        aesr_set_error("conv_wino_res: %d blocks exceed the exact range of the item decomposition (or images of 256 MB and more)", a.nblk);
        return AESR_ERR_UNSUPPORTED;
    }
    if (wino_res_lds_bytes(a.CinP, TN) > (size_t)160 * 1024) {
        aesr_set_error("conv_wino_res: %d input channels do not fit the resident filter", a.CinP);
        return AESR_ERR_ARG;
    }
    if (a.post_scale) {
        if (a.ysave || a.out_sum2 || !a.post_shift) {
            aesr_set_error("conv_wino_res: the folded eval-mode BatchNorm is a forward epilogue (no derivative mask, no 2x2-summing output)");
            return AESR_ERR_ARG;
        }
        return TN == 32 ? wino_res_launch_one<32, false, true>(a, st) : wino_res_launch_one<16, false, true>(a, st);
    }
    if (TN == 32) return a.ysave ? wino_res_launch_one<32, true>(a, st) : wino_res_launch_one<32, false>(a, st);
    return a.ysave ? wino_res_launch_one<16, true>(a, st) : wino_res_launch_one<16, false>(a, st);
}
