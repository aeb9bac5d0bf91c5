// PROTOTYPE (round 6, not part of libaesr_hip): the resident-filter Winograd F(2x2,3x3) forward convolution for 32 -> 32 channels with the matrix
// work on the BF16 pipe by a three-term split (scripts/micro/bf16x3.hip: accuracy 2.5e-7, 1.44 x on the item-loop skeleton) -- a REAL kernel, checked
// against fp64 and timed against conv_wino_res_f32 by scripts/micro/wino_b3_proto.py, to find out what the skeleton's ratio is worth once the form
// that fits the chip is written down:
//   * 4 waves per workgroup, ONE per SIMD (the two-waves-per-SIMD form does not fit: three bf16 planes of U are 96 KB, a 32-channel patch 12.8 KB per
//     wave, a lane's 4 x 4 pixels x 8 channels 128 registers beside 128 accumulators);
//   * U = G g G^T split on the host once (x = hi + mid + lo, each a bf16, exact) and packed as the A operand of v_mfma_f32_16x16x32_bf16:
//     [position 16][cout block 2][plane 3][lane 64][8 bf16], copied to LDS once per workgroup (96 KB);
//   * a work item = 4 x 4 tiles x 32 couts of ONE wave, as in conv_wino_res.hip: the wave DMAs its own 10 x 10-pixel x 32-channel patch (two
//     instructions per row: 8 pixels x 8 channel quads, then 2 pixels on 16 lanes), single-buffered -- the next patch is requested as soon as this one is
//     in registers;
//   * per position: V = (B^T d B)[position] for the lane's tile and 8 channels, split into three packed bf16x8 operands (and, sub, and, sub per value;
//     v_perm_b32 packs pairs), 2 cout blocks x 6 products (mm, lh, hl, mh, hm, hh: small terms first) into the f32 accumulators;
//   * the epilogue (A^T M A, bias through position (1,1), LeakyReLU, stores) is conv_wino_res.hip's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef B3_SCHED
#define B3_SCHED 5            // vector instructions the scheduler is asked to place behind every MFMA (0: no request)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;
constexpr int PXB = 128;                     // bytes of a pixel: 32 channels fp32
constexpr int RPB = 10 * PXB + 32;           // patch row pitch: 82 x 16 B; with the channel-quad slot XORed by (pixel column >> 1) & 3 at the DMA source the 16
                                             // lanes of a ds_read_b128 phase (16 tiles, one channel quad) hit 16 different 16-byte bank groups (brute-force search)
constexpr int PFB = 10 * RPB;                // a wave's patch buffer
constexpr int UBYTES = 16 * 2 * 3 * 1024;    // 96 KB
constexpr int OOB = 0x70000000;

struct B3Args {
    const float* in; const void* upk; const float* bias; float* out;
    int N, H, W, nblk, bpi, regs_x;
    float slope;
    int variant;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, void* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

// three packed bf16x8 planes of 8 floats (two f32x4): truncation split, exact
struct Planes { u32x4 h, m, l; };
__device__ __forceinline__ Planes split8(f32x4 v0, f32x4 v1) {
    Planes p;
    float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    unsigned xr[8], rr[8], r2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        xr[k] = __float_as_uint(x[k]);
        const float r = x[k] - __uint_as_float(xr[k] & 0xffff0000u);
        rr[k] = __float_as_uint(r);
        const float q = r - __uint_as_float(rr[k] & 0xffff0000u);
        r2[k] = __float_as_uint(q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p.h[q] = __builtin_amdgcn_perm(xr[2 * q + 1], xr[2 * q], 0x07060302u);
        p.m[q] = __builtin_amdgcn_perm(rr[2 * q + 1], rr[2 * q], 0x07060302u);
        p.l[q] = __builtin_amdgcn_perm(r2[2 * q + 1], r2[2 * q], 0x07060302u);
    }
    return p;
}

#define MFMA_B(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a_), __builtin_bit_cast(bf16x8, b_), acc_, 0, 0, 0)

__global__ __launch_bounds__(NT, 1) void wino_b3_fwd(B3Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    char* const ldsU = lds;
    char* const ldsP = lds + UBYTES + wave * PFB;
    float* const ldsBias = (float*)(lds + UBYTES + 4 * PFB);

    const int inimg = a.H * a.W * PXB;
    const int obytes = __builtin_amdgcn_readfirstlane(a.N * a.H * a.W * 32 * 4);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void*)a.upk, 0, UBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, obytes, 0x00020000);

    // ---- prologue: U (96 KB) once per workgroup; bias ----
#pragma unroll
    for (int k = 0; k < UBYTES / 1024 / 4; ++k) {
        const int kb = wave + 4 * k;
        dma16(rs_u, ldsU + kb * 1024, kb * 1024 + lane * 16);
    }
    if (tid < 32) ldsBias[tid] = a.bias ? a.bias[tid] : 0.f;

    // ---- per-lane maps ----
    // DMA instruction d of a patch row: lane -> pixel 8 d + (lane >> 3), channel quad lane & 7
    // LDS slot s of pixel column c holds channel quad s ^ ((c >> 1) & 3)
    const int dp0 = lane >> 3, dp1 = 8 + (lane >> 3);
    const int dq0 = (lane & 7) ^ ((dp0 >> 1) & 3), dq1 = lane & 7;      // columns 8, 9: (c >> 1) & 3 == 0
    const int ty = l15 >> 2, tx = l15 & 3;
    // reads: pixel (2 ty + i, 2 tx + j), channel quads 2 g + h -> slot (2 g + h) ^ fx, fx = tx for j < 2, (tx + 1) & 3 for j >= 2
    const int rbase = (2 * ty) * RPB + (2 * tx) * PXB;
    const int fxa = tx, fxb = (tx + 1) & 3;
    const int rdA0 = rbase + (((2 * g) ^ fxa) << 4), rdA1 = rbase + (((2 * g + 1) ^ fxa) << 4);
    const int rdB0 = rbase + (((2 * g) ^ fxb) << 4), rdB1 = rbase + (((2 * g + 1) ^ fxb) << 4);

    int item = blockIdx.x * 4 + wave;
    const int stride = gridDim.x * 4;
    int in_n = 0, in_y0 = 0, in_x0 = 0;
    auto locate = [&](int it) {
        in_n = it / a.bpi;
        const int rem = it - in_n * a.bpi;
        const int by = rem / a.regs_x;
        in_y0 = by * 8;
        in_x0 = (rem - by * a.regs_x) * 8;
    };
    auto fetch = [&]() {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in + (size_t)in_n * inimg), 0, inimg, 0x00020000);
        const unsigned gx0 = (unsigned)(in_x0 - 1 + dp0), gx1 = (unsigned)(in_x0 - 1 + dp1);
        const int off0 = gx0 < (unsigned)a.W ? (int)gx0 * PXB + dq0 * 16 : OOB;
        const int off1 = (gx1 < (unsigned)a.W && lane < 16) ? (int)gx1 * PXB + dq1 * 16 : OOB;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const int urow = (in_y0 - 1 + r) * a.W * PXB;           // a row above the image: negative -> out of the image's resource
            dma16(rs_in, ldsP + r * RPB, off0 + urow);
            if (lane < 16) dma16(rs_in, ldsP + r * RPB + 8 * PXB, off1 + urow);
        }
    };
    if (item < a.nblk) {
        locate(item);
        fetch();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const float nslope = a.slope;
    f32x4 acc[16][2];
    bool after_stores = false;
    while (item < a.nblk) {
        // this item's patch: its DMAs are OLDER than the previous item's 8 stores, which may stay in flight
        if (after_stores) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 t[4][4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) t[i][j][h] = *(const f32x4*)(ldsP + (j < 2 ? (h ? rdA1 : rdA0) : (h ? rdB1 : rdB0)) + i * RPB + j * PXB);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int cur_n = in_n, cur_y0 = in_y0, cur_x0 = in_x0;
        item += stride;
        if (item < a.nblk) {
            locate(item);
            fetch();
        }
        // row half of the input transform
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 d0 = t[0][j][h], d1 = t[1][j][h], d2 = t[2][j][h], d3 = t[3][j][h];
                t[0][j][h] = d0 - d2;
                t[1][j][h] = d1 + d2;
                t[2][j][h] = d2 - d1;
                t[3][j][h] = d1 - d3;
            }
#define COLV(i, j, h) ((j) == 0 ? t[i][0][h] - t[i][2][h] : (j) == 1 ? t[i][1][h] + t[i][2][h] : (j) == 2 ? t[i][2][h] - t[i][1][h] : t[i][1][h] - t[i][3][h])
        const char* ub = ldsU + lane * 16;
        // software pipeline over the 16 positions: the six filter fragments and the split V of position xi + 1 are produced while the twelve MFMAs of
        // position xi run (a bf16 MFMA holds the vector issue for 8 of its 16 cycles: the other 8 carry the split)
        u32x4 ucur[2][3], unxt[2][3];
        Planes vcur, vnxt;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ucur[nb][pl] = *(const u32x4*)(ub + ((0 * 2 + nb) * 3 + pl) * 1024);
        vcur = split8(COLV(0, 0, 0), COLV(0, 0, 1));
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (xi + 1 < 16) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) unxt[nb][pl] = *(const u32x4*)(ub + (((xi + 1) * 2 + nb) * 3 + pl) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (xi + 1 < 16) vnxt = split8(COLV((xi + 1) >> 2, (xi + 1) & 3, 0), COLV((xi + 1) >> 2, (xi + 1) & 3, 1));
            f32x4 c0 = xi == 5 ? *(const f32x4*)(ldsBias + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 c1 = xi == 5 ? *(const f32x4*)(ldsBias + 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
            MFMA_B(c0, ucur[0][1], vcur.m); MFMA_B(c1, ucur[1][1], vcur.m);
            MFMA_B(c0, ucur[0][2], vcur.h); MFMA_B(c1, ucur[1][2], vcur.h);
            MFMA_B(c0, ucur[0][0], vcur.l); MFMA_B(c1, ucur[1][0], vcur.l);
            MFMA_B(c0, ucur[0][1], vcur.h); MFMA_B(c1, ucur[1][1], vcur.h);
            MFMA_B(c0, ucur[0][0], vcur.m); MFMA_B(c1, ucur[1][0], vcur.m);
            MFMA_B(c0, ucur[0][0], vcur.h); MFMA_B(c1, ucur[1][0], vcur.h);
            acc[xi][0] = c0;
            acc[xi][1] = c1;
#if B3_SCHED
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, B3_SCHED, 0);      // B3_SCHED vector instructions of the next position's split
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) ucur[nb][pl] = unxt[nb][pl];
            vcur = vnxt;
        }
#undef COLV
        // ---- output transform, LeakyReLU, stores (conv_wino_res.hip's epilogue) ----
        {
            const int y0 = cur_y0 + 2 * ty, x0 = cur_x0 + 2 * tx;
            int ob[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) ob[p][q] = (y0 + p < a.H && x0 + q < a.W) ? ((cur_n * a.H + y0 + p) * a.W + x0 + q) * 32 * 4 : OOB;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int cob = (nb * 16 + 4 * g) * 4;
                f32x4 P[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    P[0][j] = acc[0 + j][nb] + acc[4 + j][nb] + acc[8 + j][nb];
                    P[1][j] = acc[4 + j][nb] - acc[8 + j][nb] - acc[12 + j][nb];
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
                        st16(rs_out, ob[p][q] + cob, o);
                    }
                }
            }
        }
        after_stores = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

extern "C" int b3_lds_bytes() { return UBYTES + 4 * PFB + 32 * 4; }

extern "C" int b3_conv_fwd(const float* in, const void* upk, const float* bias, float* out, int N, int H, int W, float slope, int grid, void* stream) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wino_b3_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
        attr = true;
    }
    B3Args a;
    a.in = in; a.upk = upk; a.bias = bias; a.out = out;
    a.N = N; a.H = H; a.W = W;
    const int ry = (H + 7) / 8;
    a.regs_x = (W + 7) / 8;
    a.bpi = ry * a.regs_x;
    a.nblk = N * a.bpi;
    a.slope = slope;
    a.variant = 0;
    if (grid <= 0) grid = 256;
    if (grid > (a.nblk + 3) / 4) grid = (a.nblk + 3) / 4;
    hipLaunchKernelGGL(wino_b3_fwd, dim3(grid), dim3(NT), (size_t)b3_lds_bytes(), (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
