// Implicit-GEMM stride-1 convolution on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC.
//
//   out[n,y,x,co] = epilogue( bias[co] + sum_{ky,kx,ci} in[n,y+ky-pad,x+kx-pad,ci] * w[co,ci,ky,kx] )
//
// GEMM view: M = output pixels, N = Cout, K = KS*KS*Cin.  One workgroup (4 waves) owns a tile of
// TI images x TH x TW output pixels and 16*NB output channels.  The tile's pixels are flattened
// (image, row, col) and cut into M-blocks of 16 pixels -- any TH*TW works, so tiles are picked by the
// host to divide the odd sizes of this network (162, 81, 40 ...) without large edge waste.
//
//  * A operand (activations): the input patch (tile + halo) of 16 input channels at a time is staged
//    in LDS as [pixel][16 ci + 4 pad]; every one of the KS*KS taps re-reads it at a shifted offset
//    with one ds_read_b128 per M-block (4 MFMA k-steps: lane (pixel, g) holds ci = 4g..4g+3, MFMA r
//    contracts over ci = {r, 4+r, 8+r, 12+r}).
//  * weights: pre-packed on the device as [ci chunk][cout tile][tap][ci/4][TN][4], i.e. the slice one (chunk, cout tile)
//    needs is ONE contiguous block that is staged in LDS next to the patch (same register-prefetch pipeline); the
//    fragment of a tap is one conflict-free ds_read_b128 per lane.  The tap loop issues no global loads at all
//    (vmcnt retires in order: a weight load behind the patch prefetch would wait for HBM).
//  * Epilogue: bias + {none, LeakyReLU, ReLU, sigmoid}, or (dgrad use) multiply by the activation
//    derivative taken from the saved forward output.
//
// The same kernel is the data-gradient kernel: run it on dY with the flipped/transposed packing.
//
// Replaces the ATen/cuDNN conv2d calls behind networks/acai_vanilla.py:55-56,68,70,87-88,96,98 and
// lpips/pretrained_networks.py:107-116.
#include <stdio.h>
#include <stdlib.h>

#include "aesr_kernels.h"
#include "aesr_pack_dev.h"


constexpr int IG_S = 20;     // LDS floats per patch pixel: 16 channels + 4 pad (80 B keeps b128 alignment)
// threads per workgroup (template parameter NT): 512 = 8 waves, ONE workgroup per CU; 256 = 4 waves, TWO workgroups per CU
// that run out of phase (one's staging/barrier phases under the other's MFMAs) and give small layers twice the work items
constexpr int IG_MAXP = 6;   // staging pieces (16 B) per thread: patch pixels * 4 <= IG_NT * IG_MAXP

__device__ __forceinline__ f32x4 ig_ld(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0));
}
__device__ __forceinline__ void ig_st(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

// Persistent workgroups: each walks work items (spatial tile x cout tile) item, item+G, ...  The pixel->LDS maps are
// position independent, so every integer division happens once per kernel.  Software pipeline per 16-channel chunk:
//   barrier | registers -> LDS | barrier | issue global loads of the NEXT chunk (or next item) | 9 taps of MFMAs
// so HBM/L2 latency of the patch hides behind ~18k cycles of matrix work; weight fragments are prefetched one tap ahead.
// MFMA operand roles: A = weights (M = 16 couts), B = activations (N = 16 pixels): a lane's 4 accumulator registers are
// 4 CONSECUTIVE couts of one pixel -> bias/activation/mask/store of the epilogue are 16-byte wide.
template <int KS, int NB, int MBW, int NT, bool MASK>
__global__ __launch_bounds__(NT, 2) void conv_igemm_f32(IgemmArgs a) {
    constexpr int NW = NT / 64;                 // waves per workgroup: 8 (one workgroup per CU) or 4 (two per CU, out of phase)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    if ((a.dbg & 16) && wave >= 4) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for the younger half
    const int G = gridDim.x;
    const int nitems = a.nitems, ncot = a.CoutP / (16 * NB);

    const int PW = a.TW + KS - 1, PH = a.TH + KS - 1;
    const int PPI = PH * PW, PP = a.TI * PPI;
    const int TPI = a.TH * a.TW, TP = a.TI * TPI;
    // K-split (a.ksplit > 1): a work item is (tile, cout tile, k slice); slice ks walks chunks [ks*nchunks, (ks+1)*nchunks) of the
    // input channels and writes RAW partial sums to slab ks of the output buffer (the host then runs the fix-up kernel)
    const int ksplit = a.ksplit, nchunks = (a.CinP >> 4) / ksplit;
    const int slab_bytes = a.N * a.Ho * a.Wo * a.Cout * 4;
    constexpr int TN = 16 * NB;
    constexpr int NWP = KS * KS * 4 * TN;                 // 16-byte weight pieces per (chunk, cout tile)
    constexpr int WP = (NWP + NT - 1) / NT;         // ... per thread
    float* ldsW = lds + PP * IG_S;                        // [tap][ci/4][TN][4]
    float* ldsBias = ldsW + NWP * 4;                       // [CoutP] (zeros when there is no bias)
    for (int c = tid; c < a.CoutP; c += NT) ldsBias[c] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;

    // Global accesses of the hot loop go through buffer descriptors: an out-of-range byte offset makes a load return 0 and
    // drops a store, so halo / padding / tail pieces need no exec masking, no branches and no zero-initialised registers.
    // IG_OOB + any in-range offset stays >= 2^31 - 2^28 > the tensor size (the host refuses tensors of 2^29 floats or more).
    constexpr int IG_OOB = 0x70000000;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, 0, (int)((size_t)KS * KS * a.CinP * a.CoutP * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (a.dbg & 4) ? 0 : (int)((size_t)a.ksplit * a.N * a.Ho * a.Wo * a.Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ys = __builtin_amdgcn_make_buffer_rsrc((void*)(MASK ? a.ysave : a.out), 0, (int)((size_t)a.N * a.Ho * a.Wo * a.Cout * 4), 0x00020000);

    // ---- position-independent maps (computed once) ------------------------------------------------------------
    int a_off[MBW], pix[MBW];                 // pix = img<<20 | r<<10 | c of this lane's pixel of block i, or -1
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int t = (wave + NW * i) * 16 + l15;
        const bool valid = t < TP;
        const int tt = valid ? t : 0;
        const int img = tt / TPI;
        const int rem = tt - img * TPI;
        const int r = rem / a.TW;
        const int c = rem - r * a.TW;
        a_off[i] = (img * PPI + r * PW + c) * IG_S + 4 * g;
        pix[i] = valid ? ((img << 20) | (r << 10) | c) : -1;
    }
    int piece[IG_MAXP];                        // img<<20 | pr<<10 | pc of the patch pixel of staging piece j, or -1
#pragma unroll
    for (int j = 0; j < IG_MAXP; ++j) {
        const int q = tid + NT * j;
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
    const int part4 = (tid & 3) * 4;           // channel offset of this thread's pieces inside a chunk (q & 3 == tid & 3)

    // ---- load state: (l_item, l_cc) is the chunk whose patch is in flight into R ----
    int goff[IG_MAXP];
    f32x4 R[IG_MAXP], RW[WP];
    int l_item = blockIdx.x, l_cc = 0;
    if (l_item >= nitems) return;

#define IG_TILE_ORIGIN(item, n0, y0, x0, co0)                 \
    {                                                         \
        const int rest_ = (item) / ksplit;                    \
        int tile_ = rest_ / ncot;                             \
        co0 = (rest_ - tile_ * ncot) * (16 * NB);             \
        const int tx_ = tile_ % a.tiles_x;                    \
        tile_ /= a.tiles_x;                                   \
        const int ty_ = tile_ % a.tiles_y;                    \
        n0 = (tile_ / a.tiles_y) * a.TI;                      \
        y0 = ty_ * a.TH;                                      \
        x0 = tx_ * a.TW;                                      \
    }
#define IG_COMPUTE_GOFF(item)                                                                                     \
    {                                                                                                             \
        int n0_, y0_, x0_, co0_;                                                                                  \
        IG_TILE_ORIGIN(item, n0_, y0_, x0_, co0_)                                                                 \
        (void)co0_;                                                                                               \
        _Pragma("unroll") for (int j = 0; j < IG_MAXP; ++j) {                                                     \
            int go = IG_OOB;                                                                                      \
            if (piece[j] >= 0) {                                                                                  \
                const int n = n0_ + (piece[j] >> 20), gy = y0_ + ((piece[j] >> 10) & 1023) - a.pad;               \
                const int gx = x0_ + (piece[j] & 1023) - a.pad;                                                   \
                go = (n < a.N && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (((n * a.H + gy) * a.W + gx) * a.Cin + part4) * 4 : IG_OOB; \
            }                                                                                                     \
            goff[j] = go;                                                                                         \
        }                                                                                                         \
    }
#define IG_ISSUE_LOADS(item, cc)                                                                            \
    {                                                                                                       \
        const int gcc_ = ((item) % ksplit) * nchunks + (cc);                                                \
        const int wbase_ = (int)(((size_t)gcc_ * ncot + (((item) / ksplit) % ncot)) * (NWP * 4) * 4);       \
        _Pragma("unroll") for (int j = 0; j < WP; ++j) {                                                    \
            const int w_ = tid + NT * j;                                                                    \
            RW[j] = ig_ld(rs_w, (w_ < NWP && !(a.dbg & 2)) ? wbase_ + w_ * 16 : IG_OOB);                    \
        }                                                                                                   \
        const int coff_ = (gcc_ * 16 + part4 < a.Cin && !(a.dbg & 1)) ? gcc_ * 64 : IG_OOB;                 \
        _Pragma("unroll") for (int j = 0; j < IG_MAXP; ++j) R[j] = ig_ld(rs_in, goff[j] + coff_);           \
    }

// epilogue of one finished item: D layout of 16x16x4: column (pixel) = lane&15, row (cout) = 4*(lane>>4)+j
#define IG_EPILOGUE(en0, ey0, ex0, eco0)  \
    {  \
_Pragma("unroll")  \
        for (int i = 0; i < MBW; ++i) {  \
            if (pix[i] >= 0 && !(a.dbg & 4)) {  \
                const int n = en0 + (pix[i] >> 20), y = ey0 + ((pix[i] >> 10) & 1023), x = ex0 + (pix[i] & 1023);  \
                if (n < a.N && y < a.Ho && x < a.Wo) {  \
                    const int ob = ((n * a.Ho + y) * a.Wo + x) * a.Cout;  \
_Pragma("unroll")  \
                    for (int nb = 0; nb < NB; ++nb) {  \
                        const int co = eco0 + nb * 16 + 4 * g;  \
                        f32x4 v = acc[i][nb];  \
                        if (vec) {  \
                            if (co < a.Cout) {  \
                                v += *(const f32x4*)(ldsBias + co);  \
_Pragma("unroll")  \
                                for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], a.act, a.slope);  \
                                if (MASK) {  \
                                    const f32x4 ys = *(const f32x4*)(a.ysave + ob + co);  \
_Pragma("unroll")  \
                                    for (int e = 0; e < 4; ++e) v[e] *= act_grad_from_output(ys[e], a.mask_act, a.slope);  \
                                }  \
                                *(f32x4*)(a.out + ob + co) = v;  \
                            }  \
                        } else {  \
_Pragma("unroll")  \
                            for (int e = 0; e < 4; ++e) {  \
                                if (co + e < a.Cout) {  \
                                    float s = v[e];  \
                                    s += ldsBias[co + e];  \
                                    s = act_apply(s, a.act, a.slope);  \
                                    if (MASK) s *= act_grad_from_output(a.ysave[ob + co + e], a.mask_act, a.slope);  \
                                    a.out[ob + co + e] = s;  \
                                }  \
                            }  \
                        }  \
                    }  \
                }  \
            }  \
        }  \
    }

    IG_COMPUTE_GOFF(l_item)
    IG_ISSUE_LOADS(l_item, 0)

    // ---- compute state ----
    int c_item = blockIdx.x, cc = 0;
    int cn0, cy0, cx0, co0;
    IG_TILE_ORIGIN(c_item, cn0, cy0, cx0, co0)
    f32x4 acc[MBW][NB];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* wfrag = ldsW + (g * TN + l15) * 4;              // + tap * 4*TN*4 + nb*64
    // finished-item output waiting to be stored: the stores are TRICKLED through the next chunk's tap loop (one 16-byte
    // store per tap slot) so that the 8 waves never stall on the CU's ~10 B/clk store path all at once
    constexpr int NPIECE = MBW * NB;
    constexpr int LPT = (IG_MAXP + WP + KS * KS - 1) / (KS * KS) < 2 ? 2 : (IG_MAXP + WP + KS * KS - 1) / (KS * KS);   // loads per tap
    constexpr int PPS = (NPIECE + KS * KS * MBW - 1) / (KS * KS * MBW);       // pieces per (tap, block) slot
    f32x4 pend[MBW][NB], pmask[MBW][NB];
    int pend_ob[MBW], pend_cob[NB];          // byte offsets of the pending item's pixels / of this lane's cout quads (or IG_OOB)
    bool pend_valid = false;
    const bool vec = (a.Cout & 3) == 0;
    // data-gradient use (MASK): derivative of the activation that produced the forward input, from its saved output y:
    // y > 0 ? 1 : mslope with mslope = slope (LeakyReLU), 0 (ReLU) or 1 (none).  One compare + select per element.
    const float mslope = a.mask_act == ACT_LRELU ? a.slope : (a.mask_act == ACT_RELU ? 0.f : 1.f);
#define IG_MASK_OF(y) ((y) > 0.f ? 1.f : mslope)

    unsigned long long tphase[7] = {0, 0, 0, 0, 0, 0, 0}, tlast = 0;
    const bool stamp = (a.dbg & 8) && a.dbgbuf;
#define IG_STAMP(k)                                                        \
    if (stamp) {                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
        tphase[k] += t_ - tlast;                                           \
        tlast = t_;                                                        \
    }
    if (stamp) {
        tlast = __builtin_amdgcn_s_memtime();
        tphase[5] = tlast - t_entry;
    }
    bool first = true;
    while (true) {
        if (!first) __syncthreads();           // every wave has finished reading the previous chunk from LDS
        first = false;
        IG_STAMP(0)
#pragma unroll
        for (int j = 0; j < IG_MAXP; ++j)
            if (piece[j] >= 0) *(f32x4*)(lds + ((tid + NT * j) >> 2) * IG_S + part4) = R[j];
#pragma unroll
        for (int j = 0; j < WP; ++j)
            if (tid + NT * j < NWP) *(f32x4*)(ldsW + (tid + NT * j) * 4) = RW[j];
        IG_STAMP(1)
        __syncthreads();
        IG_STAMP(2)
        // advance the load state and put the next patch chunk in flight
        if (l_cc + 1 < nchunks) {
            ++l_cc;
        } else {
            l_item += G;
            l_cc = 0;
            if (l_item < nitems) IG_COMPUTE_GOFF(l_item)
        }
        // the pending item's mask values (data-gradient use) go in flight BEFORE the patch prefetch: vmcnt retires in
        // order, so they must not queue behind HBM loads
        if (MASK && pend_valid) {
#pragma unroll
            for (int i = 0; i < MBW; ++i)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    pmask[i][nb] = ig_ld(rs_ys, pend_ob[i] + pend_cob[nb]);      // out of range -> 0 -> the store is dropped too
                }
        }
        // the prefetch of the next chunk is TRICKLED through the first taps (IG_LOADS_PER_TAP pieces per tap): issued
        // as one burst right after the barrier, the 8 waves would queue on the CU's address path for ~2.5k cycles
        const bool do_load = l_item < nitems;
        const int l_gcc = (l_item % ksplit) * nchunks + l_cc;
        const int wbase = (int)(((size_t)l_gcc * ncot + ((l_item / ksplit) % ncot)) * (NWP * 4) * 4);
        const int coff = (l_gcc * 16 + part4 < a.Cin && !(a.dbg & 1)) ? l_gcc * 64 : IG_OOB;

        const bool last_chunk = (cc + 1 == nchunks);
        IG_STAMP(3)
        // explicit software pipeline, pinned with sched_barrier: the LDS reads of block i+1 (and of the next tap's
        // weight fragments) are issued BEFORE the 4*NB MFMAs of block i, so their latency hides under 128*NB cycles of
        // matrix work (left alone, the scheduler sinks the reads to just before their first use)
        f32x4 bnxt[NB], anxt;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bnxt[nb] = *(const f32x4*)(wfrag + nb * 64);
        anxt = *(const f32x4*)(lds + a_off[0]);
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            f32x4 bcur[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bcur[nb] = bnxt[nb];
            const int tap_off = ((tap / KS) * PW + (tap % KS)) * IG_S;
#pragma unroll
            for (int i = 0; i < MBW; ++i) {
                if (i == 0 && do_load) {
#pragma unroll
                    for (int u = 0; u < LPT; ++u) {
                        constexpr int zero = 0;
                        const int pj = tap * LPT + u + zero;
                        if (pj < IG_MAXP) {
                            R[pj < IG_MAXP ? pj : 0] = ig_ld(rs_in, goff[pj < IG_MAXP ? pj : 0] + coff);
                        } else if (pj - IG_MAXP < WP) {
                            const int wj = pj - IG_MAXP < WP ? pj - IG_MAXP : 0;
                            const int w_ = tid + NT * wj;
                            RW[wj] = ig_ld(rs_w, (w_ < NWP && !(a.dbg & 2)) ? wbase + w_ * 16 : IG_OOB);
                        }
                    }
                }
                const f32x4 acur = anxt;
                if (i + 1 < MBW) {
                    anxt = *(const f32x4*)(lds + a_off[i + 1 < MBW ? i + 1 : i] + tap_off);
                } else if (tap + 1 < KS * KS) {
                    const int noff = (((tap + 1) / KS) * PW + ((tap + 1) % KS)) * IG_S;
                    anxt = *(const f32x4*)(lds + a_off[0] + noff);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) bnxt[nb] = *(const f32x4*)(wfrag + (tap + 1) * (4 * TN * 4) + nb * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[nb][r], acur[r], acc[i][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (pend_valid) {
#pragma unroll
                    for (int pp = 0; pp < PPS; ++pp) {
                        constexpr int dummy = 0;
                        const int k = (tap * MBW + i) * PPS + pp + dummy;
                        if (k < NPIECE) {
                            const int pi = k / NB, pnb = k % NB;
                            f32x4 v = pend[pi][pnb];
                            if (MASK) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] *= IG_MASK_OF(pmask[pi][pnb][e]);
                            }
                            ig_st(rs_out, pend_ob[pi] + pend_cob[pnb], v);
                        }
                    }
                }
            }
        }
        pend_valid = false;

        IG_STAMP(4)
        if (!last_chunk) {
            ++cc;
            continue;
        }
        // item finished: bias + activation now, stores later (trickled through the next chunk / flushed after the loop)
        if (vec) {
#pragma unroll
            for (int i = 0; i < MBW; ++i) {
                pend_ob[i] = IG_OOB;
                if (pix[i] >= 0) {
                    const int n = cn0 + (pix[i] >> 20), y = cy0 + ((pix[i] >> 10) & 1023), x = cx0 + (pix[i] & 1023);
                    if (n < a.N && y < a.Ho && x < a.Wo) pend_ob[i] = ((n * a.Ho + y) * a.Wo + x) * a.Cout * 4 + (c_item % ksplit) * slab_bytes;
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 v = acc[i][nb] + *(const f32x4*)(ldsBias + co0 + nb * 16 + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], a.act, a.slope);
                    pend[i][nb] = v;
                    acc[i][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) pend_cob[nb] = (co0 + nb * 16 + 4 * g < a.Cout) ? (co0 + nb * 16 + 4 * g) * 4 : IG_OOB;
            pend_valid = true;
        } else {
            IG_EPILOGUE(cn0, cy0, cx0, co0)
#pragma unroll
            for (int i = 0; i < MBW; ++i)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[i][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        c_item += G;
        if (c_item >= nitems) break;
        cc = 0;
        IG_TILE_ORIGIN(c_item, cn0, cy0, cx0, co0)
    }
    if (pend_valid) {            // last item of this workgroup: flush
#pragma unroll
        for (int i = 0; i < MBW; ++i)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x4 v = pend[i][nb];
                if (MASK) {
                    const f32x4 ys = ig_ld(rs_ys, pend_ob[i] + pend_cob[nb]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= IG_MASK_OF(ys[e]);
                }
                ig_st(rs_out, pend_ob[i] + pend_cob[nb], v);
            }
    }
    if (stamp && tid == 0) {
        tphase[6] = __builtin_amdgcn_s_memtime() - tlast;
        for (int k = 0; k < 7; ++k) a.dbgbuf[blockIdx.x * 7 + k] = (float)tphase[k];
    }
#undef IG_MASK_OF
#undef IG_STAMP
#undef IG_EPILOGUE
#undef IG_TILE_ORIGIN
#undef IG_COMPUTE_GOFF
#undef IG_ISSUE_LOADS
}

// ---- weight packing: pack_elements lives in aesr_pack_dev.h (shared with the one-launch weight preparation, prep.hip) ----

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int KS,
                                    int KinP, int NoutP, int TN, int transpose) {
    pack_elements(w, p, Cout, Cin, KS, KinP, NoutP, TN, transpose, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                  (size_t)gridDim.x * blockDim.x);
}

// many filters, one launch: the block looks its job up in the by-value table
__global__ __launch_bounds__(256) void pack_many_kernel(PackTable t) {
    int j = 0;
    for (int k = 1; k < t.njobs; ++k)
        if ((int)blockIdx.x >= t.job[k].block0) j = k;
    const PackJob& jb = t.job[j];
    const int b1 = (j + 1 < t.njobs) ? t.job[j + 1].block0 : t.nblocks;
    pack_elements(jb.w, jb.p, jb.Cout, jb.Cin, jb.KS, jb.KinP, jb.NoutP, jb.TN, jb.transpose,
                  (size_t)(blockIdx.x - jb.block0) * 256 + threadIdx.x, (size_t)(b1 - jb.block0) * 256);
}

int aesr_launch_pack_weights(const float* w, float* p, int Cout, int Cin, int KS, int KinP, int NoutP, int TN, int transpose,
                             hipStream_t st) {
    const size_t total = (size_t)KS * KS * KinP * NoutP;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, st, w, p, Cout, Cin, KS, KinP, NoutP, TN, transpose);
    AESR_LAUNCH_CHECK("pack_weights");
    return AESR_OK;
}

int aesr_launch_pack_many(const PackTable& t, hipStream_t st) {
    hipLaunchKernelGGL(pack_many_kernel, dim3(t.nblocks), dim3(256), 0, st, t);
    AESR_LAUNCH_CHECK("pack_many");
    return AESR_OK;
}

template <int KS, int NB, int MBW, int NT, bool MASK>
static int launch_one(const IgemmArgs& a, hipStream_t st) {
    const int PP = a.TI * (a.TH + KS - 1) * (a.TW + KS - 1);
    const size_t shmem = ((size_t)PP * IG_S + (size_t)KS * KS * 4 * 16 * NB * 4 + a.CoutP) * sizeof(float);
    if (shmem > (size_t)160 * 1024 / (512 / NT)) {
        aesr_set_error("conv_igemm: tile needs %zu B of LDS", shmem);
        return AESR_ERR_ARG;
    }
    // opt in to the full 160 KB of dynamic LDS once per (instantiation, device); a failure is reported here, with its cause
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_igemm_f32<KS, NB, MBW, NT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_igemm_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    int grid = 256 * (512 / NT);              // persistent: 2 waves per SIMD (register budget), as one 8-wave or two 4-wave workgroups per CU
    if (const char* e = getenv("AESR_IGEMM_GRID")) grid = atoi(e);
    if (grid > a.nitems) grid = a.nitems;
    if (a.dbg & 8) {                          // debug: per-phase cycle stamps, printed after a host sync
        static float* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 1024 * 7 * sizeof(float));
        IgemmArgs b = a;
        b.dbgbuf = dbuf;
        hipLaunchKernelGGL((conv_igemm_f32<KS, NB, MBW, NT, MASK>), dim3(grid), dim3(NT), shmem, st, b);
        (void)hipStreamSynchronize(st);
        static float host[1024 * 7];
        (void)hipMemcpy(host, dbuf, grid * 7 * sizeof(float), hipMemcpyDeviceToHost);
        double s5[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < grid; ++i) for (int k = 0; k < 7; ++k) s5[k] += host[i * 7 + k];
        fprintf(stderr, "[igemm stamps] grid=%d items=%d per-WG kcycles: barrier1 %.1f | lds-write %.1f | barrier2 %.1f | loads+epilogue %.1f | mfma %.1f | setup %.1f | flush %.1f\n",
                grid, a.nitems, s5[0] / grid / 1e3, s5[1] / grid / 1e3, s5[2] / grid / 1e3, s5[3] / grid / 1e3, s5[4] / grid / 1e3,
                s5[5] / grid / 1e3, s5[6] / grid / 1e3);
        return AESR_OK;
    }
    hipLaunchKernelGGL((conv_igemm_f32<KS, NB, MBW, NT, MASK>), dim3(grid), dim3(NT), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_igemm_f32");
    return AESR_OK;
}

// K-split fix-up: out = act(sum_ks partial[ks] + bias) * act'(ysave), 16 bytes per thread
__global__ __launch_bounds__(256) void conv_ksplit_fixup_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                                const float* __restrict__ ysave, float* __restrict__ out, size_t n4,
                                                                int Cout, int ksplit, int act, int mask_act, float slope) {
    const size_t slab = n4 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = *(const f32x4*)(partial + i * 4);
        for (int k = 1; k < ksplit; ++k) v += *(const f32x4*)(partial + k * slab + i * 4);
        if (bias) v += *(const f32x4*)(bias + (i * 4) % Cout);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act, slope);
        if (ysave) {
            const f32x4 ys = *(const f32x4*)(ysave + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= act_grad_from_output(ys[e], mask_act, slope);
        }
        *(f32x4*)(out + i * 4) = v;
    }
}

int aesr_launch_conv_ksplit_fixup(const float* partial, const float* bias, const float* ysave, float* out, size_t nelem, int Cout,
                                  int ksplit, int act, int mask_act, float slope, hipStream_t st) {
    const size_t n4 = nelem / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv_ksplit_fixup_kernel, dim3(grid), dim3(256), 0, st, partial, bias, ysave, out, n4, Cout, ksplit, act, mask_act, slope);
    AESR_LAUNCH_CHECK("conv_ksplit_fixup");
    return AESR_OK;
}

int aesr_launch_conv_igemm(const IgemmArgs& a_in, int KS, int NB, int MBW, hipStream_t st) {
    IgemmArgs a = a_in;
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("AESR_IGEMM_DBG"); dbg = e ? atoi(e) : 0; }
    a.dbg = dbg;
    const int TP = a.TI * a.TH * a.TW;
    const int nblk = (TP + 15) / 16;
    const int NT = 512, NW = NT / 64;       // the 256-thread (two workgroups per CU) form measured slower: not instantiated
    if (nblk > NW * MBW) {
        aesr_set_error("conv_igemm: tile of %d pixels needs %d M-blocks > %d", TP, nblk, NW * MBW);
        return AESR_ERR_ARG;
    }
    MBW = (nblk + NW - 1) / NW;                // exact blocks per wave: the inner loop has no runtime block bound
    if (a.CoutP % (16 * NB) != 0 || a.CinP % 16 != 0 || a.Cin % 4 != 0) {
        aesr_set_error("conv_igemm: bad channel padding Cin=%d CinP=%d CoutP=%d NB=%d", a.Cin, a.CinP, a.CoutP, NB);
        return AESR_ERR_ARG;
    }
    const int PPc = a.TI * (a.TH + KS - 1) * (a.TW + KS - 1);
    if (PPc * 4 > NT * IG_MAXP) {
        aesr_set_error("conv_igemm: patch of %d pixels exceeds the staging capacity", PPc);
        return AESR_ERR_ARG;
    }
    // byte offsets are 32-bit buffer offsets and 0x70000000 marks "out of range": tensors must stay below that many bytes
    if ((size_t)a.N * a.H * a.W * a.Cin >= (size_t)0x1C000000 || (size_t)a.N * a.Ho * a.Wo * a.Cout >= (size_t)0x1C000000) {
        aesr_set_error("conv_igemm: tensors of 469M elements (1.75 GB) or more need 64-bit indexing (not built)");
        return AESR_ERR_UNSUPPORTED;
    }
    if (a.ysave && a.mask_act == ACT_SIGMOID) {
        aesr_set_error("conv_igemm: a sigmoid derivative mask is not fused into the data gradient (use aesr_act_bwd)");
        return AESR_ERR_UNSUPPORTED;
    }
    if (a.ksplit < 1) a.ksplit = 1;
    if (a.ksplit > 1 && ((a.CinP >> 4) % a.ksplit != 0 || a.bias || a.ysave || a.act != ACT_NONE || (a.Cout & 3) ||
                         (size_t)a.ksplit * a.N * a.Ho * a.Wo * a.Cout >= (size_t)0x1C000000)) {
        aesr_set_error("conv_igemm: invalid K-split launch (ksplit=%d)", a.ksplit);
        return AESR_ERR_ARG;
    }
    a.nitems = ceil_div(a.N, a.TI) * a.tiles_y * a.tiles_x * (a.CoutP / (16 * NB)) * a.ksplit;
    if (a.TH + KS - 1 > 1023 || a.TW + KS - 1 > 1023 || a.TI > 1023) {
        aesr_set_error("conv_igemm: tile dimensions exceed the packed-coordinate range");
        return AESR_ERR_ARG;
    }
#define IG_CASE(ks, nb, mbw)                                                          \
    if (KS == ks && NB == nb && MBW == mbw)                                           \
        return a.ysave ? launch_one<ks, nb, mbw, 512, true>(a, st) : launch_one<ks, nb, mbw, 512, false>(a, st);
    IG_CASE(3, 1, 1) IG_CASE(3, 1, 2) IG_CASE(3, 1, 3) IG_CASE(3, 1, 4) IG_CASE(3, 2, 1) IG_CASE(3, 2, 2) IG_CASE(3, 2, 3) IG_CASE(3, 2, 4)
    IG_CASE(3, 4, 1) IG_CASE(3, 4, 2)
    IG_CASE(1, 1, 1) IG_CASE(1, 1, 2) IG_CASE(1, 1, 3) IG_CASE(1, 1, 4) IG_CASE(1, 2, 1) IG_CASE(1, 2, 2) IG_CASE(1, 2, 3) IG_CASE(1, 2, 4)
    IG_CASE(1, 4, 1) IG_CASE(1, 4, 2)
#undef IG_CASE
    aesr_set_error("conv_igemm: no instantiation for KS=%d NB=%d MBW=%d", KS, NB, MBW);
    return AESR_ERR_UNSUPPORTED;
}
