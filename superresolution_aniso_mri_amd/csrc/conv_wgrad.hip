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
//
// Pipeline per pixel tile: [write the prefetched registers to LDS] barrier [issue the global loads of the NEXT tile into
// registers] [MFMA loop over LDS] barrier.  The loads of tile i+1 are in flight during the MFMA loop of tile i, so only the
// LDS write phase and the two barriers are exposed.
// Global-load lane map (chosen for the LDS side): a 32-lane group = 16 consecutive pixels x 2 adjacent channel quads, so
// the transposing ds_write_b32 (bank = address mod 32, plane stride = 4 mod 32) hit 32 distinct banks.
// float4 prefetch registers per thread: X patch (PP * 8 <= 256 * NX) and dY tile (TP * COT/4 <= 256 * ND); the 64-cout
// variant carries twice the accumulators, so it gets fewer slots (tiles up to 10x8) to stay at 3 waves per SIMD
#define WG_NX(nwco) ((nwco) == 4 ? 4 : 6)
#define WG_ND(nwco) ((nwco) == 4 ? 5 : 6)

template <int KS, int NWCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_wgrad_f32(WgradArgs a) {
    constexpr int CIBW = (NWCO == 2) ? 1 : 2;        // ci blocks per wave
    constexpr int CIT = 32, COT = 16 * NWCO;
    constexpr int XPAIRS = CIT / 8, DPAIRS = COT / 8;   // channel-quad pairs per pixel
    constexpr int NX = WG_NX(NWCO), ND = WG_ND(NWCO);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int wco = (NWCO == 2) ? (wave >> 1) : wave;
    const int wci = (NWCO == 2) ? (wave & 1) : 0;
    // Workgroup ids are dealt round-robin to the 8 XCDs.  The nchunks workgroups that walk the SAME pixel tiles (one per
    // (ci chunk, co chunk)) get ids that land on one XCD back to back, so that XCD's L2 serves their re-reads of X and dY.
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
    const int PH = a.TH + KS - 1;
    const int PWS = a.PWS, TWS = a.TWS, PSX = a.PSX, PSD = a.PSD;      // row / plane strides (floats), host-chosen
    float* ldsX = lds;                      // [CIT][PSX]
    float* ldsD = lds + CIT * PSX;          // [COT][PSD]
    const bool do_bias = (ciy == 0) && (wci == 0);

    f32x4 acc[KS * KS][CIBW];
    float accb = 0.f;
#pragma unroll
    for (int t = 0; t < KS * KS; ++t)
#pragma unroll
        for (int i = 0; i < CIBW; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tpi = a.tiles_y * a.tiles_x;
    const int PWp = a.TW + KS - 1;          // patch pixels per row actually staged
    const int PP = PH * PWp, TP = a.TH * a.TW;
    // this lane's fixed offsets: X plane of (ci block, l15) + 2g ; dY plane of (co block, l15) + 2g
    const int xbase0 = ((wci * CIBW) * 16 + l15) * PSX + 2 * g;
    const int dbase = (wco * 16 + l15) * PSD + 2 * g;

    // ---- per-thread staging slots (tile independent): slot k of this thread moves the float4 of (pixel, quad) ----
    // unit u = (k*256 + tid) >> 5 (a 32-lane group); pair j = u % PAIRS, pixel block = u / PAIRS; pixel = block*16 + (tid & 15),
    // quad = 2*j + ((tid >> 4) & 1)
    // Global loads go through buffer descriptors: an out-of-range byte offset returns 0, so halo / tail / padded-channel
    // slots need no exec masking and no zeroed registers; per slot the tile-independent part of the offset is kept.
    constexpr int WG_OOB = 0x70000000;       // + any in-range offset stays beyond the tensor (host: tensors < 1.75 GB)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)((size_t)a.N * a.Ho * a.Wo * a.Cout * 4), 0x00020000);
    int xrc[NX], drc[ND];                          // (row << 16 | col) of the slot's pixel, or -1
    int xrel[NX], drel[ND];                        // byte offset of the slot relative to the tile origin pixel, or WG_OOB
    int xlds[NX], dlds[ND];                        // LDS float offset of the slot's (quad, pixel), or -1
    const int qbit = (tid >> 4) & 1, u0 = tid >> 5;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int pix = ((k * 8 + u0) / XPAIRS) * 16 + (tid & 15);
        const int pr = pix / PWp, pc = pix - pr * PWp;
        const int ci = ci0 + (2 * ((k * 8 + u0) % XPAIRS) + qbit) * 4;
        xrc[k] = pix < PP ? ((pr << 16) | pc) : -1;
        xrel[k] = (pix < PP && ci < a.Cin) ? ((pr * a.W + pc) * a.Cin + ci) * 4 : WG_OOB;
        xlds[k] = pix < PP ? (2 * ((k * 8 + u0) % XPAIRS) + qbit) * 4 * PSX + pr * PWS + pc : -1;
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) {
        const int pix = ((k * 8 + u0) / DPAIRS) * 16 + (tid & 15);
        const int r = pix / a.TW, c = pix - r * a.TW;
        const int co = co0 + (2 * ((k * 8 + u0) % DPAIRS) + qbit) * 4;
        drc[k] = pix < TP ? ((r << 16) | c) : -1;
        drel[k] = (pix < TP && co < a.Cout) ? ((r * a.Wo + c) * a.Cout + co) * 4 : WG_OOB;
        dlds[k] = pix < TP ? (2 * ((k * 8 + u0) % DPAIRS) + qbit) * 4 * PSD + r * TWS + c : -1;
    }
    f32x4 RX[NX], RD[ND];
    auto issue_loads = [&](int tile) {
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        // tile origin (may lie in the padding: the per-slot row / column checks decide)
        const int xorg = ((n * a.H + y0 - a.pad) * a.W + (x0 - a.pad)) * a.Cin * 4;
        const int dorg = ((n * a.Ho + y0) * a.Wo + x0) * a.Cout * 4;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const unsigned gy = (unsigned)(y0 + (xrc[k] >> 16) - a.pad), gx = (unsigned)(x0 + (xrc[k] & 0xffff) - a.pad);
            const int off = (gy < (unsigned)a.H && gx < (unsigned)a.W) ? xorg + xrel[k] : WG_OOB;
            RX[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int gy = y0 + (drc[k] >> 16), gx = x0 + (drc[k] & 0xffff);
            const int off = (gy < a.Ho && gx < a.Wo) ? dorg + drel[k] : WG_OOB;
            RD[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, off, 0, 0));
        }
    };

    // debug stamps (wave 0 of each workgroup): cycles in [LDS write | barrier 1 | issue loads | MFMA loop | barrier 2]
    long long st[5] = {0, 0, 0, 0, 0}, t0 = 0, t1;
#define WG_STAMP(i)                                   \
    if (a.dbgbuf) {                                   \
        t1 = __builtin_amdgcn_s_memtime();            \
        st[i] += t1 - t0;                             \
        t0 = t1;                                      \
    }
    int tile = split;
    if (tile < a.ntiles) issue_loads(tile);
    if (a.dbgbuf) t0 = __builtin_amdgcn_s_memtime();
    while (tile < a.ntiles) {
        // ---- registers -> LDS, transposing to channel-major ----
#pragma unroll
        for (int k = 0; k < NX; ++k)
            if (xlds[k] >= 0) {
                float* d = ldsX + xlds[k];
                d[0] = RX[k][0];
                d[PSX] = RX[k][1];
                d[2 * PSX] = RX[k][2];
                d[3 * PSX] = RX[k][3];
            }
#pragma unroll
        for (int k = 0; k < ND; ++k)
            if (dlds[k] >= 0) {
                float* d = ldsD + dlds[k];
                d[0] = RD[k][0];
                d[PSD] = RD[k][1];
                d[2 * PSD] = RD[k][2];
                d[3 * PSD] = RD[k][3];
            }
        WG_STAMP(0)
        __syncthreads();
        WG_STAMP(1)
        const int next = tile + a.S;
        if (next < a.ntiles) issue_loads(next);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch loads ahead of the MFMA loop
        WG_STAMP(2)
        // ---- k loop: groups of 8 consecutive pixels of one row; lane (.,g) owns pixels c0+2g, c0+2g+1 (+ halo) ----
        // Explicit software pipeline, pinned with sched_barrier: the LDS reads of step s+1 (next tap row / ci block, or the
        // first step of the next pixel group) are issued BEFORE the 2*KS MFMAs of step s.
        {
            constexpr int NSTEP = KS * CIBW;
            const int gpr = a.TW >> 3, ngrp = a.TH * gpr;
            int r = 0, cg = 0;
            float2 dv = *(const float2*)(ldsD + dbase);
            float2 e01 = *(const float2*)(ldsX + xbase0), e23 = e01;
            if (KS > 1) e23 = *(const float2*)(ldsX + xbase0 + 2);
            for (int gi = 0; gi < ngrp; ++gi) {
                const int c0 = cg * 8;
                int rn = r, cgn = cg + 1;
                if (cgn == gpr) { cgn = 0; rn = r + 1; }
                if (gi + 1 == ngrp) { rn = r; cgn = cg; }             // last group: re-read it (value unused, address stays inside the tile)
                float2 dvn = dv;
#pragma unroll
                for (int st_ = 0; st_ < NSTEP; ++st_) {
                    const int ky = st_ / CIBW, i = st_ % CIBW;
                    const float* xn;
                    if (st_ + 1 < NSTEP) {
                        const int kyn = (st_ + 1) / CIBW, in_ = (st_ + 1) % CIBW;
                        xn = ldsX + xbase0 + in_ * 16 * PSX + (r + kyn) * PWS + c0;
                    } else {
                        xn = ldsX + xbase0 + rn * PWS + cgn * 8;
                        dvn = *(const float2*)(ldsD + dbase + rn * TWS + cgn * 8);
                    }
                    const float2 n01 = *(const float2*)xn;
                    float2 n23 = n01;
                    if (KS > 1) n23 = *(const float2*)(xn + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    const float e[4] = {e01.x, e01.y, e23.x, e23.y};
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)      // pixel sub-step 0, then sub-step 1: same accumulator KS MFMAs apart
                        acc[ky * KS + kx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[kx], dv.x, acc[ky * KS + kx][i], 0, 0, 0);
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx)
                        acc[ky * KS + kx][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(e[kx + 1], dv.y, acc[ky * KS + kx][i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    e01 = n01;
                    e23 = n23;
                }
                accb += dv.x + dv.y;                  // bias gradient: this lane's share of column sum co = l15 (VALU, free beside the MFMAs)
                dv = dvn;
                r = rn;
                cg = cgn;
            }
        }
        WG_STAMP(3)
        __syncthreads();
        WG_STAMP(4)
        tile = next;
    }
#undef WG_STAMP
    if (a.dbgbuf && tid == 0)
        for (int i = 0; i < 5; ++i) a.dbgbuf[blockIdx.x * 5 + i] = (float)st[i];

    // ---- write the partial slab: D layout col (co) = lane&15, row (ci) = 4*(lane>>4)+j ----
    const size_t plane = (size_t)a.CinP * a.CoutP;
    float* sl = a.slab + (size_t)split * (KS * KS + 1) * plane;
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
    accb += __shfl_xor(accb, 16, 64);
    accb += __shfl_xor(accb, 32, 64);
    if (do_bias && g == 0) sl[(KS * KS) * plane + co] = accb;
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

// the same for the convolutions of a whole backward pass in ONE launch: the block looks its job up in the by-value table
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(ReduceTable t) {
    __shared__ float red[4][64];
    int j = 0;
    for (int k = 1; k < t.njobs; ++k)
        if ((int)blockIdx.x >= t.job[k].block0) j = k;
    const ReduceJob& jb = t.job[j];
    const int o = ((int)blockIdx.x - jb.block0) * 64 + (threadIdx.x & 63);
    const int sl = threadIdx.x >> 6;
    const int nw = jb.KS2 * jb.Cin * jb.Cout;
    const int total = nw + (jb.db ? jb.Cout : 0);
    float s = 0.f;
    size_t off = 0;
    const bool live = o < total;
    int co = 0, ci = 0, tap = 0;
    if (live) {
        if (o < nw) {
            co = o % jb.Cout;
            const int rest = o / jb.Cout;
            ci = rest % jb.Cin;
            tap = rest / jb.Cin;
            off = ((size_t)tap * jb.CinP + ci) * jb.CoutP + co;
        } else {
            off = (size_t)jb.KS2 * jb.CinP * jb.CoutP + (o - nw);
        }
        const size_t stride = (size_t)(jb.KS2 + 1) * jb.CinP * jb.CoutP;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = sl;
        for (; k + 12 < jb.nslab; k += 16) {
            s0 += jb.slab[(size_t)k * stride + off];
            s1 += jb.slab[(size_t)(k + 4) * stride + off];
            s2 += jb.slab[(size_t)(k + 8) * stride + off];
            s3 += jb.slab[(size_t)(k + 12) * stride + off];
        }
        for (; k < jb.nslab; k += 4) s0 += jb.slab[(size_t)k * stride + off];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && live) {
        s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (o < nw) jb.dw[((size_t)co * jb.Cin + ci) * jb.KS2 + tap] = s;
        else jb.db[o - nw] = s;
    }
}

int aesr_launch_wgrad_reduce_many(const ReduceTable& t, hipStream_t st) {
    hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(t.nblocks), dim3(256), 0, st, t);
    AESR_LAUNCH_CHECK("wgrad_reduce_many");
    return AESR_OK;
}

template <int KS, int NWCO>
static int launch_wgrad(const WgradArgs& a, hipStream_t st) {
    constexpr int COT = 16 * NWCO;
    const size_t shmem = ((size_t)32 * a.PSX + (size_t)COT * a.PSD) * sizeof(float);
    if (shmem > 160 * 1024) {
        aesr_set_error("conv_wgrad: tile needs %zu B of LDS", shmem);
        return AESR_ERR_ARG;
    }
    if ((a.TH + KS - 1) * (a.TW + KS - 1) * 8 > 256 * WG_NX(NWCO) || a.TH * a.TW * (COT / 4) > 256 * WG_ND(NWCO)) {
        aesr_set_error("conv_wgrad: tile %dx%d does not fit the register prefetch slots", a.TH, a.TW);
        return AESR_ERR_ARG;
    }
    // opt in to the full 160 KB of dynamic LDS once per (instantiation, device); a failure is reported here, with its cause
    static bool attr_set[AESR_MAX_DEVICES] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= AESR_MAX_DEVICES) dev_ = 0;
    if (!attr_set[dev_]) {
        const hipError_t e_ = hipFuncSetAttribute((const void*)conv_wgrad_f32<KS, NWCO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e_ != hipSuccess) {
            aesr_set_error("conv_wgrad_f32: hipFuncSetAttribute(MaxDynamicSharedMemorySize = 160 KB) failed: %s", hipGetErrorString(e_));
            return AESR_ERR_HIP;
        }
        attr_set[dev_] = true;
    }
    dim3 grid(a.S * (a.CinP / 32) * (a.CoutP / COT));
    if (getenv("AESR_WGRAD_DBG")) {           // debug: per-phase cycle stamps, printed after a host sync
        static float* dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 5 * sizeof(float));
        WgradArgs b = a;
        b.dbgbuf = grid.x <= 4096 ? dbuf : nullptr;
        hipLaunchKernelGGL((conv_wgrad_f32<KS, NWCO>), grid, dim3(256), shmem, st, b);
        (void)hipStreamSynchronize(st);
        static float host[4096 * 5];
        (void)hipMemcpy(host, dbuf, grid.x * 5 * sizeof(float), hipMemcpyDeviceToHost);
        double s5[5] = {0, 0, 0, 0, 0};
        for (unsigned i = 0; i < grid.x; ++i) for (int k = 0; k < 5; ++k) s5[k] += host[i * 5 + k];
        fprintf(stderr, "[wgrad stamps] grid=%u tiles=%d tile=%dx%d per-WG kcycles: lds-write %.1f | barrier1 %.1f | issue-loads %.1f | mfma %.1f | barrier2 %.1f\n",
                grid.x, a.ntiles, a.TH, a.TW, s5[0] / grid.x / 1e3, s5[1] / grid.x / 1e3, s5[2] / grid.x / 1e3, s5[3] / grid.x / 1e3, s5[4] / grid.x / 1e3);
        return AESR_OK;
    }
    hipLaunchKernelGGL((conv_wgrad_f32<KS, NWCO>), grid, dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_wgrad_f32");
    return AESR_OK;
}

// variant: 0 -> 32 ci x 32 co per workgroup; 1 -> 32 ci x 64 co
int aesr_launch_conv_wgrad(const WgradArgs& a, int KS, int variant, hipStream_t st) {
    if ((size_t)a.N * a.H * a.W * a.Cin >= (size_t)0x1C000000 || (size_t)a.N * a.Ho * a.Wo * a.Cout >= (size_t)0x1C000000) {
        aesr_set_error("conv_wgrad: tensors of 469M elements (1.75 GB) or more need 64-bit indexing (not built)");
        return AESR_ERR_UNSUPPORTED;
    }
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
