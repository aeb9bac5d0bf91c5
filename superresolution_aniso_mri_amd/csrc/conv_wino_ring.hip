// Winograd F(2x2, 3x3) convolution for layers with MANY K-side channels (the filter does not fit in LDS): the transformed filter
// streams through a three-slot LDS ring shared by the workgroup, every wave owns its tiles and its input patch, and the only
// coupling between the waves is a per-slot arrival counter in LDS -- no barrier after the prologue.
//
// The first streamed kernel (conv_wino.hip) met at one __syncthreads() per 16-channel chunk: its 8 waves entered the staging
// DMAs, the LDS read burst and the item epilogue TOGETHER, and on gfx950 nothing hides inside a wave (the f32 MFMA shares its
// pipe with the vector ALU, scripts/micro/mfma_gap.hip): SQ_VALU_MFMA_BUSY 47 %, 30 % of the wave time parked at the barrier,
// 4-way bank conflicts on the shared patch (profiles/r02_wino_experiments.txt).  conv_wino_res.hip showed what independent waves
// buy where the filter is resident (71 % busy).  Here, for any channel count:
//  * a wave's work item is a block of <= 16 Winograd tiles (TI images x THt x TWt tiles, 4 x 4 tiles = 8 x 8 outputs where the
//    layer allows) x 32 output channels: the same 128 accumulators per wave and per-lane output transform as the other kernels.
//    The wave fetches its OWN patch by DMA (one buffer_load ... lds per patch row) into its own LDS and waits on its own vmcnt;
//    rows are pitched so that the 16 ds_read_b128 of the input transform are conflict free (the launcher searches pitch, image
//    stride and channel-quad swizzle per block shape; 4 x 4 blocks: 656 B rows, quads swapped in pixels 4..7);
//  * the 8 waves of a workgroup share the cout tile and walk the chunks in the same order; each fetches 1/8 of every 32 KB filter
//    chunk.  Chunk number q lives in ring slot q % 3.  A wave issues its part of chunk q + 1 (and its patch of q + 1) at the top of
//    chunk q, and a quarter of a chunk later -- when its own vmcnt says they have landed -- adds 1 to the slot's arrival counter.
//    Before the first filter read of a chunk a wave spins (normally zero times) until the counter shows all 8 arrivals.  Why three
//    slots suffice: when the arrivals of chunk q are complete, every wave has passed the signal point of chunk q - 1, so nobody
//    reads chunk q - 2 any more, which is the slot chunk q + 1 is written to.  Waves may therefore drift apart by up to three
//    quarters of a chunk: one wave's epilogue, stores and DMA issue run beside its SIMD partner's MFMAs;
//  * work items are (group of 8 blocks, cout tile) for as many WHOLE rounds over the 256 CUs as the layer has; the blocks of the last,
//    partial round are dealt out <= 4 per group over all CUs (one wave per SIMD everywhere instead of two on a few CUs: half the
//    time of a full round), the other waves of such a group only stream their share of the filter.  Small layers (VGG conv5: 48
//    blocks x 16 cout tiles) are all "tail" and still occupy every CU;
//  * workgroups are renumbered so that the items of one spatial block group (all cout tiles) and its neighbours run on ONE XCD
//    at about the same time: the input patches are fetched into that XCD's L2 once;
//  * channel split: a layer with fewer items than the chip has SIMDs would stream its K side serially per item (512 channels = 32
//    chunks of ~3 us: VGG conv4 / conv5, every deep layer of a small data-parallel shard).  With a caller workspace the launcher may
//    make S <= 16 items per (block group, cout tile), each streaming 1/S of the chunks and storing raw partial sums into its own
//    output-shaped slab; wino_split_reduce_kernel adds the slabs in fixed order and applies bias, activation and derivative mask.
//    S comes from the same cost model (+ the reduce launch and its (S + 1) output-sized streams).
// U packing, the in-register input transform, bias in the accumulator of position (1,1), the derivative mask and the folded
// Upsample(x2) in both directions are those of conv_wino.hip; results equal to rounding (the summation order over channels is
// the same).
//
// Replaces the ATen/cuDNN conv2d calls behind networks/acai_vanilla.py:68,70,87 (layers with 64 and more K-side channels) and
// lpips/pretrained_networks.py:107-116 (every VGG layer but conv1_1), forward and as data gradient.
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

#include "aesr_kernels.h"

constexpr int RG_WFL = 16 * 4 * 32 * 4;     // floats of one U chunk (16 positions x 16 ci x 32 co) = 8192
constexpr int RG_SLOTS = 3;
constexpr int RG_LDS_MAX = 160 * 1024 - 256;      // dynamic LDS a launch may ask for: the arrival counters are static LDS of the kernel
constexpr int RG_OOB = 0x70000000;
#ifndef RG_SIG_POS
#define RG_SIG_POS 4
#endif
constexpr int RG_SIG = RG_SIG_POS;          // the arrival of the next chunk is signalled after this many positions (of 16)
constexpr int RG_SPIN_LIMIT = 1 << 22;      // a spin that long means a protocol bug: give up (flagged, results are garbage) instead of hanging the GPU

__device__ unsigned int g_ring_timeouts = 0;

__device__ __forceinline__ void rg_dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off, int uniform_off = 0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, uniform_off, 0, 0);
}
// lane id without a live register: the per-lane parts of the DMA addresses are rebuilt where they are used (the accumulators and
// the raw tile leave ~30 registers for everything else)
__device__ __forceinline__ int rg_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ f32x4 rg_ld(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
// The slab offset of a channel split is added to the per-lane offset, NOT passed as the instruction's scalar offset.  With an SGPR
// soffset the compiler (ROCm 7.2 clang) leaves out the wait state between "buffer_store_dwordx4 v[128:131], v, s[..], sN offen" and
// the next VALU instruction that overwrites v[128:131] -- its hazard recognizer holds that this store-data hazard "only exists if the
// instruction is not using a register in the soffset field" -- and on gfx950 the store then picks up the NEW contents in part of its
// lanes: odd output channels of tiles 12..15 came out wrong (scripts/diag_ring_tail.py; the ISA of the two forms differs by exactly
// that s_nop).  With soffset = 0 the compiler inserts the wait state.
__device__ __forceinline__ void rg_st(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

// PWT: pixel slots of a patch row (>= 2 TWt + 2) at compile time; rows are pitched PWT * 16 + 4 floats, so the 16 patch reads of a
// chunk share 4 address registers (a run-time pitch cost 11 spilled registers, reloaded behind s_waitcnt vmcnt(0))
// POST: the folded eval-mode BatchNorm (+ AvgPool2d(2)) epilogue of conv_wino_res.hip, for the forward of slice synthesis
template <int PWT, bool MASK, bool POST = false>
__global__ __launch_bounds__(512, 2) void conv_wino_ring_f32(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = 8;
    constexpr int rowP = PWT * 16 + 4;
    const int ncot = a.CoutP >> 5, nchunks = a.CinP >> 4;
    float* const ldsU = lds;                                            // [3][16 positions][4][32][4]
    float* const ldsP = lds + RG_SLOTS * RG_WFL + wave * a.patch_fl;    // this wave's patch: [TI][PH rows][PW pixel slots][16 ch] (+ pitches)
    float* const ldsBias = lds + RG_SLOTS * RG_WFL + NW * a.patch_fl;   // [CoutP]
    __shared__ unsigned cnt[4];                                         // [3] arrival counters (static: the compiler must see LDS, not flat, accesses)

    const int sh = a.in_up2 ? 1 : 0;
    const int inH = a.H >> sh, inW = a.W >> sh;                                              // stored size of the input tensor
    const bool halfout = a.out_sum2 || (POST && a.post_pool);
    const int outH = halfout ? a.H >> 1 : a.H, outW = halfout ? a.W >> 1 : a.W;              // stored size of the output tensor
    const int inimg = inH * inW * a.Cin * 4, inrow = inW * a.Cin * 4;                        // bytes of an input image / row
    const int ibytes = __builtin_amdgcn_readfirstlane(a.N * inimg);
    const int wbytes = __builtin_amdgcn_readfirstlane(16 * a.CinP * a.CoutP * 4);
    // with a channel split (a.ksplit > 1) a.out is the workspace of partial outputs: one output-shaped slab per split
    const int obytes = __builtin_amdgcn_readfirstlane(a.ksplit * a.split_bytes), ybytes = __builtin_amdgcn_readfirstlane(a.N * a.H * a.W * a.Cout * 4);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, ibytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.upk, 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, obytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ys = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK ? a.ysave : a.out), 0, ybytes, 0x00020000);

    // ---- prologue: bias, counters ----
    for (int c = tid; c < a.CoutP; c += NW * 64) ldsBias[c] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
    if (tid < RG_SLOTS) cnt[tid] = 0u;
    __syncthreads();            // the only barrier of the kernel

    // ---- per-lane maps (launch constants) ----
    const int PH = 2 * a.THt + 2, TPI = a.THt * a.TWt, TP = a.TI * TPI;
    constexpr int PW = PWT;
    // DMA of a patch row: lane -> pixel slot lane >> 2, channel quad (lane & 3) ^ f(slot), f = the launcher's bank swizzle
    // (lanes past the patch's last column sit out: their LDS slots belong to the next row)
    // MFMA B operand: lane -> tile l15 of the block (image ti, tile row tr, tile column tc), channel quad g
    int tmap;                                                           // ti << 16 | tr << 8 | tc, or -1 for a lane without a tile
    int offj[4];                                                        // float offset of patch pixel (2 tr, 2 tc + j), this lane's quad
    {
        const int l15 = lane & 15, g = lane >> 4;
        const int t = l15 < TP ? l15 : 0;
        const int ti_l = t / TPI;
        const int rem = t - ti_l * TPI;
        const int tr_l = rem / a.TWt;
        const int tc_l = rem - tr_l * a.TWt;
        tmap = l15 < TP ? (ti_l << 16) | (tr_l << 8) | tc_l : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int px = 2 * tc_l + j;
            offj[j] = ti_l * a.imgP + 2 * tr_l * rowP + px * 16 + ((g ^ ((((px >> a.sw_a) & a.sw_m) << a.sw_b) & 3)) << 2);
        }
    }
    constexpr int upw = 32 / NW;                                        // 1 KB pieces of a U chunk this wave fetches

    const float mslope = a.mask_act == ACT_LRELU ? a.slope : (a.mask_act == ACT_RELU ? 0.f : 1.f);
    const float nslope = a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU ? 0.f : 1.f);      // ACT_NONE / ACT_SIGMOID: identity
    const bool sigm = a.act == ACT_SIGMOID;

    // workgroup order: the workgroups of one XCD (blockIdx % 8 under round-robin placement: speed only) take consecutive items
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7;
    const int wg = xcd * (G >> 3) + min(xcd, G & 7) + (int)(blockIdx.x >> 3);          // XCD x owns (G - x + 7) / 8 consecutive items per round

#define RG_DIV(x, m) ((m) ? (int)__umulhi((unsigned)(x), (m)) : (int)(x))          /* m == 0: divisor 1 */
    // item -> (block group, cout tile); this wave's block of the group -> (image group, block row, block column)
    // channel split: item = ((block group, split), cout tile); split s streams chunks [s kchunks, (s + 1) kchunks) of the K side into its
    // own output-shaped slab (no bias, no activation: the split-reduce kernel finishes the layer)
    int in_n0 = 0, in_ty0 = 0, in_tx0 = 0, in_co0 = 0, in_c0 = 0, in_c1 = nchunks, in_so = 0;
    bool in_active = false;
    auto locate = [&](int it) {
        const int bgs = RG_DIV(it, a.m_ncot);
        in_co0 = (it - bgs * ncot) * 32;
        const int bg = RG_DIV(bgs, a.m_ksplit);
        const int sp = bgs - bg * a.ksplit;
        in_c0 = sp * a.kchunks;
        in_c1 = min(in_c0 + a.kchunks, nchunks);
        in_so = sp * a.split_bytes;
        // full groups of 8 blocks first (whole rounds over the grid); the blocks of the last, partial round are dealt a.tail_k <= 4 per
        // group -- one wave per SIMD on EVERY CU instead of eight waves on a few -- and waves tail_k.. only stream the filter
        const int tj = bg - a.nfull;
        const int blk = tj < 0 ? bg * NW + wave : a.nfull * NW + tj * a.tail_k + wave;
        in_active = blk < a.nblk && (tj < 0 || wave < a.tail_k);
        const int b = in_active ? blk : 0;
        const int ng = RG_DIV(b, a.m_bpi);
        const int rem = b - ng * a.bpi;
        const int by = RG_DIV(rem, a.m_regs_x);
        in_n0 = ng * a.TI;
        in_ty0 = by * a.THt;
        in_tx0 = (rem - by * a.regs_x) * a.TWt;
    };
    // the patch of (located block, chunk cc) -> this wave's LDS buffer; out-of-range rows / columns / images deliver zeros
    auto fetch_patch = [&](int cc) {
        const int ln = rg_lane(), dpx = ln >> 2;
        const int dq = (ln & 3) ^ ((((dpx >> a.sw_a) & a.sw_m) << a.sw_b) & 3);
        const int lcd = (((dpx - sh) >> sh) * a.Cin + 4 * dq) * 4;      // bytes from (row start + block column origin); arithmetic shift: -1 stays -1
        const unsigned gx = (unsigned)(2 * in_tx0 - 1 + dpx);
        const int lane_off = (gx < (unsigned)a.W) ? lcd + (((2 * in_tx0) >> sh) - 1 + sh) * a.Cin * 4 + cc * 64 : RG_OOB;
        if (dpx < PW) {
            int ti = 0, pr = 0;
            const int nrows = a.TI * PH;
#pragma unroll 2
            for (int r = 0; r < nrows; ++r) {
                const int n = in_n0 + ti, gy = 2 * in_ty0 - 1 + pr;
                const int urow = (n < a.N && (unsigned)gy < (unsigned)a.H) ? n * inimg + (gy >> sh) * inrow : RG_OOB;
                rg_dma(rs_in, ldsP + ti * a.imgP + pr * rowP, lane_off + urow);
                if (++pr == PH) { pr = 0; ++ti; }
            }
        }
    };
    // this wave's part of U chunk (cout tile co0, chunk cc) -> ring slot
    auto fetch_u = [&](int co0, int cc, int slot) {
        const int wbase = ((cc * ncot + (co0 >> 5)) * RG_WFL + wave * upw * 256) * 4;
        float* dst = ldsU + slot * RG_WFL + wave * upw * 256;
        const int voff = rg_lane() * 16;
#pragma unroll
        for (int k = 0; k < upw; ++k) rg_dma(rs_w, dst + k * 256, voff, wbase + k * 1024);
    };
    auto signal = [&](int slot, bool on) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMAs (U part and patch of the next chunk) have landed
        if (on && rg_lane() == 0) __hip_atomic_fetch_add(cnt + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto wait_full = [&](int slot, unsigned target) {
        int spins = 0;
        while (true) {
            const unsigned v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > RG_SPIN_LIMIT) {
                if (rg_lane() == 0) atomicAdd(&g_ring_timeouts, 1u);
                break;
            }
        }
        asm volatile("" ::: "memory");
    };

    int item = wg;
    if (item >= a.nitems) return;
    // the two waves of a SIMD (w and w + 4) would otherwise run their chunk heads (patch reads, DMA issue, transforms: no MFMA) at the
    // same time; delayed by about half a chunk, one's head runs beside the other's MFMAs (MI355X_MICROARCH.md, two waves per SIMD, item 9)
    if (wave >= 4)
        for (int k = 0; k < (a.flags & 15); ++k) __builtin_amdgcn_s_sleep(16);
    const int istep = G;
    locate(item);
    fetch_u(in_co0, in_c0, 0);
    if (in_active) fetch_patch(in_c0);
    signal(0, true);

    f32x4 acc[16][2];
    bool after_stores = false;                  // exactly 8 stores (and nothing else) were issued behind the current patch's DMAs
    int cc = in_c0, slot = 0;
    unsigned target = (unsigned)NW;             // arrivals that complete the chunk in `slot`
    while (true) {
        const bool active = in_active;
        const int cur_n0 = in_n0, cur_ty0 = in_ty0, cur_tx0 = in_tx0, cur_co0 = in_co0, cur_c0 = in_c0, cur_so = in_so;
        const bool last = cc + 1 == in_c1;
        const int nslot = slot == RG_SLOTS - 1 ? 0 : slot + 1;
        bool has_next = true;
        // request the next patch (next chunk, or chunk 0 of the next item) and this wave's part of the next U chunk; slot (q + 1) % 3
        // held chunk q - 2, which nobody reads any more once chunk q is complete (see the header)
        auto fetch_next = [&]() {
            if (last) {
                item += istep;
                has_next = item < a.nitems;
                if (has_next) {
                    locate(item);
                    fetch_u(in_co0, in_c0, nslot);
                    if (in_active) fetch_patch(in_c0);
                }
            } else {
                fetch_u(cur_co0, cc + 1, nslot);
                if (active) fetch_patch(cc + 1);
            }
        };
        if (!active) {
            // a wave without a block in this group (tail of the layer) only streams its share of the filter
            wait_full(slot, target);
            fetch_next();
            signal(nslot, has_next);
        } else {
            // ---- the 4x4 input pixels of this lane's tile, 4 channels each: own DMA, issued a whole chunk ago (the stores of an item
            // epilogue in between are younger: after_stores) ----
            if (after_stores) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            f32x4 t[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) t[i][j] = *(const f32x4*)(ldsP + offj[j] + i * rowP);
            wait_full(slot, target);                // the whole U chunk has landed (all NW parts)
            const int ln = rg_lane();
            const float* wb = ldsU + slot * RG_WFL + ((ln >> 4) * 32 + (ln & 15)) * 4;       // + xi * 512 + nb * 64
            const float* bl = ldsBias + cur_co0 + 4 * (ln >> 4);
            f32x4 wnx[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wnx[nb] = *(const f32x4*)(wb + nb * 64);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fetch_next();                           // the patch is in registers: its buffer is free
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
#define RG_V(i, j) ((j) == 0 ? aesr_sub4(t[i][0], t[i][2]) : (j) == 1 ? t[i][1] + t[i][2] : (j) == 2 ? aesr_sub4(t[i][2], t[i][1]) : aesr_sub4(t[i][1], t[i][3]))
            f32x4 vnx = RG_V(0, 0);
            auto positions = [&](auto firstc, auto X0c, auto X1c) {
                constexpr bool FIRST = decltype(firstc)::value;
#pragma unroll
                for (int xi = decltype(X0c)::value; xi < decltype(X1c)::value; ++xi) {
                    f32x4 wc[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) wc[nb] = wnx[nb];
                    const f32x4 vc = vnx;
                    if (xi + 1 < 16) {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) wnx[nb] = *(const f32x4*)(wb + (xi + 1) * 512 + nb * 64);
                        vnx = RG_V((xi + 1) >> 2, (xi + 1) & 3);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            f32x4 c = acc[xi][nb];
                            if (FIRST && r == 0) c = xi == 5 ? *(const f32x4*)(bl + nb * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
                            acc[xi][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nb][r], vc[r], c, 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            using X0 = std::integral_constant<int, 0>;
            using XS = std::integral_constant<int, RG_SIG>;
            using XE = std::integral_constant<int, 16>;
            // ONE branch per chunk (first chunk of an item or not), the signal inside both arms: a second branch between the two runs
            // of positions made the register allocator rotate ~200 registers through moves at the join
            if (cc == cur_c0) {
                positions(std::true_type{}, X0{}, XS{});
                signal(nslot, has_next);
                positions(std::true_type{}, XS{}, XE{});
            } else {
                positions(std::false_type{}, X0{}, XS{});
                signal(nslot, has_next);
                positions(std::false_type{}, XS{}, XE{});
            }
#undef RG_V
        }
        // chunk done: the next one sits in the next slot; a slot's arrivals grow by NW every time the ring comes round
        slot = nslot;
        if (slot == 0) target += (unsigned)NW;
        after_stores = false;
        if (!last) {
            ++cc;
            continue;
        }
        cc = in_c0;              // of the item located by fetch_next (unused after the last item)
        // ---- item finished: output transform Y = A^T M A, activation, (data gradient) derivative mask, store ----------
        if (active) {
            const int g = rg_lane() >> 4;
            const int n = cur_n0 + (tmap >> 16), y0 = 2 * (cur_ty0 + ((tmap >> 8) & 255)), x0 = 2 * (cur_tx0 + (tmap & 255));
            const bool okn = tmap >= 0 && n < a.N;
            int ob[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    ob[p][q] = (okn && y0 + p < a.H && x0 + q < a.W) ? ((n * a.H + y0 + p) * a.W + x0 + q) * a.Cout * 4 : RG_OOB;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = cur_co0 + nb * 16 + 4 * g;
                const int cob = co < a.Cout ? co * 4 : RG_OOB;
                f32x4 ys[2][2];
                if (MASK) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int q = 0; q < 2; ++q) ys[p][q] = rg_ld(rs_ys, ob[p][q] + cob);
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
                    const int obs = (okn && y0 < a.H && x0 < a.W) ? ((n * outH + (y0 >> 1)) * outW + (x0 >> 1)) * a.Cout * 4 : RG_OOB;
                    rg_st(rs_out, obs + cob + cur_so, s);
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
                            else rg_st(rs_out, ob[p][q] + cob, o * psc + psh);              // bn.hip bn_apply: v * scale + shift
                        } else {
                            rg_st(rs_out, ob[p][q] + cob + cur_so, o);
                        }
                    }
                    if (POST && a.post_pool) pm = p == 0 ? prow : pm + prow;
                }
                if (POST && a.post_pool) {
                    // AvgPool2d(2) of the activated tile, then the affine, in bn_apply's order; an odd image's last row / column has no window
                    const f32x4 m = pm * 0.25f;                     // ((o00 + o01) + (o10 + o11)) * 0.25
                    const int obs = (okn && y0 + 1 < a.H && x0 + 1 < a.W) ? ((n * outH + (y0 >> 1)) * outW + (x0 >> 1)) * a.Cout * 4 : RG_OOB;
                    rg_st(rs_out, obs + cob, m * psc + psh);
                }
            }
        }
        if (!has_next) break;
        after_stores = active && !MASK && !halfout;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no DMA may still be writing this workgroup's LDS when it is released
#undef RG_DIV
}

// ---- launcher: block shape, LDS layout, item list ----------------------------------------------------------------------------------

// LDS cycles the 16 ds_read_b128 of the input transform lose to bank conflicts (one phase = 16 lanes, 16-byte bank groups mod 16)
static int ring_conflicts(int TI, int THt, int TWt, int P16, int IP16, int sa, int sm, int sb) {
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59}, {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int TPI = THt * TWt, TP = TI * TPI;
    int tot = 0;
    for (int j = 0; j < 4; ++j)
        for (int gq = 0; gq < 4; ++gq) {
            int addr[16], n = 0;
            for (int k = 0; k < 16; ++k) {
                const int lane = groups[gq][k], g = lane >> 4, l15 = lane & 15;
                const int t = l15 < TP ? l15 : 0, ti = t / TPI, rem = t - ti * TPI, tr = rem / TWt, tc = rem - tr * TWt;
                const int px = 2 * tc + j;
                addr[n++] = ti * IP16 + 2 * tr * P16 + px * 4 + (g ^ ((((px >> sa) & sm) << sb) & 3));
            }
            for (int b = 0; b < 16; ++b) {
                int distinct = 0, seen[16];
                for (int k = 0; k < 16; ++k) {
                    if ((addr[k] & 15) != b) continue;
                    bool dup = false;
                    for (int s = 0; s < distinct; ++s) dup |= seen[s] == addr[k];
                    if (!dup) seen[distinct++] = addr[k];
                }
                if (distinct > 1) tot += distinct - 1;
            }
        }
    return tot * 4;         // the same for each of the four patch rows i
}

// the kernel's compile-time patch widths (pixel slots): the smallest that holds 2 TWt + 2 pixels
static int ring_pwt(int TWt) { const int pw = 2 * TWt + 2; return pw <= 8 ? 8 : pw <= 10 ? 10 : pw <= 12 ? 12 : 16; }

struct RingPlan { int TI, THt, TWt, PWT, imgP, sa, sm, sb, patch_fl, nfull, tail_k, ntail, conf, ksplit; double cost; };
static std::map<std::tuple<int, int, int, int, int, int, int>, RingPlan> g_ring_plans;
static std::mutex g_ring_mu;

static size_t ring_lds_bytes(int patch_fl, int CoutP) { return ((size_t)RG_SLOTS * RG_WFL + (size_t)8 * patch_fl + CoutP + 16) * sizeof(float); }

static void ring_layout(RingPlan& p) {
    // image stride (16-byte units) and channel-quad swizzle with the fewest conflicts at the kernel's row pitch
    const int PH = 2 * p.THt + 2, P16 = p.PWT * 4 + 1;
    int best = 1 << 30;
    for (int pad = 0; pad <= (p.TI > 1 ? 16 : 0); ++pad)
        for (int sa = 0; sa < 4; ++sa)
            for (int sm = 0; sm < 4; ++sm)
                for (int sb = 0; sb < 2; ++sb) {
                    if (sm == 0 && (sa || sb)) continue;
                    const int IP16 = PH * P16 + pad;
                    const int c = ring_conflicts(p.TI, p.THt, p.TWt, P16, IP16, sa, sm, sb);
                    if (c * 32 + pad < best) { best = c * 32 + pad; p.imgP = IP16 * 4; p.sa = sa; p.sm = sm; p.sb = sb; p.conf = c; }
                }
    // the last row's DMA writes only where lanes are active (PWT * 4 pieces of 16 bytes): the buffer ends with that row
    p.patch_fl = round_up((p.TI - 1) * p.imgP + (PH - 1) * (p.PWT * 16 + 4) + p.PWT * 16, 4);
}

// item list of a layer with B blocks on a grid of G workgroups: groups of 8 blocks for as many WHOLE rounds as there are, the rest dealt
// out tail_k <= 4 blocks per group over a whole round of its own (one wave per SIMD on every CU); a rest too large for that goes in 8s
static void ring_items(RingPlan& p, long B, int ncot, int G) {
    const long per_round = G % ncot == 0 ? (long)(G / ncot) : 0;       // groups per round
    p.nfull = (int)(B / 8);
    p.tail_k = (int)(B - 8L * p.nfull);
    p.ntail = p.tail_k ? 1 : 0;
    if (per_round > 0) {
        const long full_rounds = (B / 8) / per_round;
        const long rest = B - full_rounds * per_round * 8;
        if (rest > 0 && rest <= 4 * per_round) {
            p.nfull = (int)(full_rounds * per_round);
            p.tail_k = (int)((rest + per_round - 1) / per_round);
            p.ntail = (int)((rest + p.tail_k - 1) / p.tail_k);
        }
    }
}

constexpr int RG_KSPLIT_MAX = 16;

// smax: the largest channel split the caller's workspace allows (1: none)
static RingPlan plan_ring(const WinoArgs& a, int smax) {
    std::lock_guard<std::mutex> lk(g_ring_mu);
    if (smax > RG_KSPLIT_MAX) smax = RG_KSPLIT_MAX;
    if (smax < 1) smax = 1;
    int fks = 0;
    if (const char* e = getenv("AESR_RING_KSPLIT")) fks = atoi(e);                  // experiments / tests: force the channel split (where the workspace allows)
    const auto key = std::make_tuple(a.N, a.H, a.W, a.CinP, a.CoutP, smax, fks);
    auto it = g_ring_plans.find(key);
    if (it != g_ring_plans.end()) return it->second;
    const int Ht = ceil_div(a.H, 2), Wt = ceil_div(a.W, 2), ncot = a.CoutP / 32, nch = a.CinP / 16;
    RingPlan best;
    best.cost = 1e300;
    best.TI = 1; best.THt = 1; best.TWt = 1; best.ksplit = 1;
    const double out_bytes = (double)a.N * a.H * a.W * a.Cout * 4.0;
    auto consider = [&](int TI, int THt, int TWt) {
        RingPlan p;
        p.TI = TI; p.THt = THt; p.TWt = TWt;
        const int TP = TI * THt * TWt, nrows = TI * (2 * THt + 2);
        if (TP > 16 || 2 * TWt + 2 > 16 || nrows > 16) return;
        p.PWT = ring_pwt(TWt);
        ring_layout(p);
        if (ring_lds_bytes(p.patch_fl, a.CoutP) > (size_t)RG_LDS_MAX) return;
        const long B = (long)ceil_div(a.N, TI) * ceil_div(Ht, THt) * ceil_div(Wt, TWt);
        int sforce = 0;                       // forced split: the largest valid one <= min(fks, smax)
        for (int S = 1; fks > 0 && S <= smax && S <= fks; S *= 2)
            if (S == 1 || (ceil_div(nch, S) >= 2 && ceil_div(nch, ceil_div(nch, S)) == S)) sforce = S;
        for (int S = 1; S <= smax; S *= 2) {
            const int kch = ceil_div(nch, S);
            if (S > 1 && (kch < 2 || ceil_div(nch, kch) != S)) continue;        // at least two chunks per split, no empty split
            if (sforce > 0 && S != sforce) continue;
            ring_items(p, B, ncot * S, 256);
            // chunks the busiest SIMD runs: two waves per full round, one (tail_k <= 4) or two in the tail round; a wave's chunk = 128 MFMAs
            // + the transforms, LDS reads and DMA issue nothing overlaps when the wave is alone on its SIMD, about half of it when paired
            const double items = (double)p.nfull * ncot * S, tail_items = (double)p.ntail * ncot * S;
            const double rounds = items <= 4 * 256 ? (double)ceil_div((int)items, 256) : items / 256.0;
            const double ovh = 700.0 + 45.0 * (nrows + 4) + p.conf;
            const double tail = tail_items > 0 ? (double)ceil_div((int)tail_items, 256) * (p.tail_k <= 4 ? 4096.0 + ovh : 2.0 * 4096.0 + 1.2 * ovh) : 0.0;
            p.cost = kch * (rounds * (2.0 * 4096.0 + 1.2 * ovh) + tail) + 3000.0 * (rounds + (tail > 0)) + 3000.0;
            // the split-reduce launch: a kernel boundary + ramp (~3 us) and (S + 1) output-sized streams at ~2 KB per cycle
            if (S > 1) p.cost += 7000.0 + (S + 1) * out_bytes / 2048.0;
            p.ksplit = S;
            if (p.cost < best.cost) best = p;
        }
    };
    int fti = 0, fth = 0, ftw = 0;
    if (const char* e = getenv("AESR_RING_SHAPE")) (void)sscanf(e, "%d,%d,%d", &fti, &fth, &ftw);      // experiments: "TI,THt,TWt"
    if (fti > 0 && fth > 0 && ftw > 0) consider(fti, fth, ftw);
    if (best.cost > 1e299) {
        for (int TI = 1; TI <= 16 && TI <= a.N; ++TI)
            for (int THt = 1; THt <= 7 && THt <= Ht; ++THt)
                for (int TWt = 1; TWt <= 7 && TWt <= Wt; ++TWt) {
                    if (TI > 1 && (THt < Ht || TWt < Wt) && !(THt == 1 && TWt == Wt)) continue;       // several images: whole images or whole tile rows
                    consider(TI, THt, TWt);
                }
    }
    if (getenv("AESR_PLAN_DEBUG"))
        fprintf(stderr, "[aesr plan] ring N=%d %dx%d Cin=%d Cout=%d -> TI=%d THt=%d TWt=%d PWT=%d imgP=%d swizzle (%d,%d,%d) conflicts %d patch %d B; "
                "channel split %d (of <= %d); %d full groups + %d tail groups of %d; cost %.0f\n", a.N, a.H, a.W, a.CinP, a.CoutP, best.TI, best.THt, best.TWt,
                best.PWT, best.imgP, best.sa, best.sm, best.sb, best.conf, best.patch_fl * 4, best.ksplit, smax, best.nfull, best.ntail, best.tail_k, best.cost);
    g_ring_plans[key] = best;
    return best;
}

int aesr_wino_ring_mode() {
    // 2 (default since round 4): every streamed layer; 1: only where the cost estimate below is lower than the first streamed kernel's (the
    // default of round 3 -- it kept 81 x 81 and a few 40 x 40 layers on conv_wino_f32, measured 0.7-0.8 % slower per step at C2 / C4 / C5
    // than "always": profiles/r04_ring_always.txt); 0: never
    const char* e = getenv("AESR_WINO_RING");          // read per call: tests and A/B scripts switch it inside one process
    return e ? atoi(e) : 2;
}

// Which streamed kernel serves a layer: both planners estimate the cycles of the busiest SIMD with the same unit costs (fitted to
// the layer tables of profiles/r03_wino_layers.txt: 0.50-0.59 us per 1000 cycles for either); the ring kernel wins where 8 x 8-output
// blocks tile the image well, the first kernel's big shared patches where they do not (81 x 81: 87 % against 94 % of the tile slots
// used) and on layers too small for a round of ring items (VGG conv5)
// The channel split the workspace in ``a`` allows: the unconstrained plan's if its slabs fit, else none
static int ring_smax(const WinoArgs& a) {
    if (a.ws_floats == 0 || a.out_sum2) return 1;       // (queries pass ws_floats = SIZE_MAX without a buffer: "whatever the plan wants")
    const size_t out_floats = (size_t)a.N * a.H * a.W * a.Cout;
    // a store's 32-bit offset is (pixel or RG_OOB) + (channel or RG_OOB) + slab offset: with BOTH sentinels set, a slab offset of
    // 0x20000000 and more would wrap 2 RG_OOB + offset past 2^32 into a valid slab -- splits are for small layers, so larger ones get none
    if (out_floats * 4 * (size_t)RG_KSPLIT_MAX >= (size_t)0x20000000) return 1;
    const RingPlan p = plan_ring(a, RG_KSPLIT_MAX);
    return (size_t)p.ksplit * out_floats <= a.ws_floats ? RG_KSPLIT_MAX : 1;
}

bool aesr_wino_ring_takes(const WinoArgs& a) {
    const int mode = aesr_wino_ring_mode();
    if (mode <= 0 || a.CinP != a.Cin) return false;
    if (mode >= 2 || a.plan_cost <= 0.0) return true;
    return plan_ring(a, ring_smax(a)).cost < a.plan_cost;
}

size_t aesr_wino_ring_workspace_floats(const WinoArgs& a) {
    WinoArgs b = a;
    b.ws = nullptr;                       // a query, nothing is launched: "a workspace of any size" = the plan's wish
    b.ws_floats = ~(size_t)0;
    if (!aesr_wino_ring_takes(b)) return 0;
    const RingPlan p = plan_ring(b, ring_smax(b));
    return p.ksplit > 1 ? (size_t)p.ksplit * a.N * a.H * a.W * a.Cout : 0;
}

unsigned aesr_wino_ring_timeouts() {
    unsigned v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ring_timeouts), sizeof(v));
    return v;
}

template <int PWT, bool MASK, bool POST = false>
static int ring_launch_one(const WinoArgs& a, int grid, size_t shmem, hipStream_t st) {
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wino_ring_f32<PWT, MASK, POST>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_MAX);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wino_ring_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB - 256 B) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    hipLaunchKernelGGL((conv_wino_ring_f32<PWT, MASK, POST>), dim3(grid), dim3(512), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wino_ring_f32");
    return AESR_OK;
}

// called by aesr_launch_conv_wino (which has validated the arguments) for the layers the resident-filter kernel does not take
// out = act(sum of the S slabs + bias) (x the derivative mask of the data gradient): finishes a channel-split layer
__global__ __launch_bounds__(256) void wino_split_reduce_kernel(const float* __restrict__ part, int S, size_t n4, size_t stride4,
                                                                const float* __restrict__ bias, int Cout4, const float* __restrict__ ysave,
                                                                float* __restrict__ out, int act, float slope, int mask_act) {
    const f32x4* p4 = (const f32x4*)part;
    const float nslope = act == ACT_LRELU ? slope : (act == ACT_RELU ? 0.f : 1.f);
    const float mslope = mask_act == ACT_LRELU ? slope : (mask_act == ACT_RELU ? 0.f : 1.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = p4[i];
        for (int s_ = 1; s_ < S; ++s_) v += p4[s_ * stride4 + i];          // fixed order: deterministic
        if (bias) v += ((const f32x4*)bias)[i % Cout4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * nslope);
        if (act == ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
        }
        if (ysave) {
            const f32x4 y = ((const f32x4*)ysave)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= (y[e] > 0.f ? 1.f : mslope);
        }
        ((f32x4*)out)[i] = v;
    }
}

int aesr_launch_conv_wino_ring(const WinoArgs& a_in, hipStream_t st) {
    WinoArgs a = a_in;
    const RingPlan p = plan_ring(a, ring_smax(a));
    a.TI = p.TI; a.THt = p.THt; a.TWt = p.TWt; a.nw = 8;
    a.rowP = p.PWT * 16 + 4; a.imgP = p.imgP; a.sw_a = p.sa; a.sw_m = p.sm; a.sw_b = p.sb; a.patch_fl = p.patch_fl;
    const int Ht = ceil_div(a.H, 2), Wt = ceil_div(a.W, 2), ncot = a.CoutP / 32;
    a.regs_y = ceil_div(Ht, a.THt);
    a.regs_x = ceil_div(Wt, a.TWt);
    a.bpi = a.regs_y * a.regs_x;
    a.nblk = ceil_div(a.N, a.TI) * a.bpi;
    a.nfull = p.nfull; a.tail_k = p.tail_k;
    a.ksplit = p.ksplit;
    if (a.ksplit > 1 && !a.ws) {
        aesr_set_error("conv_wino_ring: a channel split of %d was planned without a workspace", a.ksplit);
        return AESR_ERR_ARG;
    }
    if (a.post_scale && (a.ksplit > 1 || a.ysave || a.out_sum2 || !a.post_shift)) {
        aesr_set_error("conv_wino_ring: the folded eval-mode BatchNorm is a forward epilogue without a channel split (pass no workspace)");
        return AESR_ERR_ARG;
    }
    a.kchunks = ceil_div(a.CinP / 16, p.ksplit);
    const bool halfout = a.out_sum2 || (a.post_scale && a.post_pool);
    const size_t out_floats = (size_t)a.N * (halfout ? a.H / 2 : a.H) * (halfout ? a.W / 2 : a.W) * a.Cout;
    if (out_floats * 4 * (size_t)a.ksplit >= (size_t)(a.ksplit > 1 ? 0x20000000 : RG_OOB)) {      // split: see ring_smax
        aesr_set_error("conv_wino_ring: %zu output bytes x %d channel splits exceed the kernel's 32-bit offsets", out_floats * 4, a.ksplit);
        return AESR_ERR_UNSUPPORTED;
    }
    a.split_bytes = (int)(out_floats * 4);
    a.nitems = (p.nfull + p.ntail) * ncot * a.ksplit;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + d - 1) / d); };
    a.m_ncot = magic(ncot); a.m_bpi = magic(a.bpi); a.m_regs_x = magic(a.regs_x); a.m_ksplit = magic(a.ksplit);
    const float* fin_bias = a.bias;
    const float* fin_mask = a.ysave;
    const int fin_act = a.act;
    float* const fin_out = a.out;
    if (a.ksplit > 1) {                 // the kernel writes raw partial sums into the slabs; bias, activation and mask in the reduce
        a.out = a.ws; a.bias = nullptr; a.ysave = nullptr; a.act = ACT_NONE;
    }
    if (((unsigned long long)a.nitems + 4096) * (unsigned)(ncot * a.ksplit + 1) >= ((unsigned long long)1 << 31) ||
        ((unsigned long long)a.nblk + 4096) * (unsigned)(a.bpi + a.regs_x) >= ((unsigned long long)1 << 31)) {
        aesr_set_error("conv_wino_ring: %d items exceed the exact range of the item decomposition", a.nitems);
        return AESR_ERR_UNSUPPORTED;
    }
    const size_t shmem = ring_lds_bytes(a.patch_fl, a.CoutP);
    if (shmem > (size_t)RG_LDS_MAX || a.CinP != a.Cin) {
        aesr_set_error("conv_wino_ring: block of %d x %d x %d tiles does not fit LDS (or Cin %d is not a multiple of 16)", a.TI, a.THt, a.TWt, a.Cin);
        return AESR_ERR_ARG;
    }
    int grid = 256;                                     // persistent: one workgroup per CU
    if (const char* e = getenv("AESR_WINO_GRID")) grid = atoi(e);
    if (grid > a.nitems) grid = a.nitems;
    int rc = AESR_ERR_UNSUPPORTED;
#define RG_CASE(pw)                                                                                                              \
    if (p.PWT == pw)                                                                                                             \
        rc = a.post_scale ? ring_launch_one<pw, false, true>(a, grid, shmem, st)                                                 \
                          : (a.ysave ? ring_launch_one<pw, true>(a, grid, shmem, st) : ring_launch_one<pw, false>(a, grid, shmem, st));
    RG_CASE(8) RG_CASE(10) RG_CASE(12) RG_CASE(16)
#undef RG_CASE
    if (rc == AESR_ERR_UNSUPPORTED) aesr_set_error("conv_wino_ring: no instantiation for a patch of %d pixel slots", p.PWT);
    if (rc != AESR_OK || a.ksplit == 1) return rc;
    const size_t n4 = out_floats / 4;
    hipLaunchKernelGGL(wino_split_reduce_kernel, dim3((unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048)), dim3(256), 0, st, a.ws, a.ksplit, n4, n4, fin_bias, a.Cout / 4, fin_mask,
                       fin_out, fin_act, a.slope, a.mask_act);
    AESR_LAUNCH_CHECK("wino_split_reduce");
    return AESR_OK;
}
