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
//  * B operand (weights): pre-packed on the device as [tap][ci/4][CoutP][4] so that the matching
//    fragment is ONE coalesced 16-byte global load per lane, straight to registers (L2-resident;
//    no LDS traffic, no staging barrier for weights).
//  * Epilogue: bias + {none, LeakyReLU, ReLU, sigmoid}, or (dgrad use) multiply by the activation
//    derivative taken from the saved forward output.
//
// The same kernel is the data-gradient kernel: run it on dY with the flipped/transposed packing.
//
// Replaces the ATen/cuDNN conv2d calls behind networks/acai_vanilla.py:55-56,68,70,87-88,96,98 and
// lpips/pretrained_networks.py:107-116.
#include "aesr_kernels.h"


constexpr int IG_S = 20;   // LDS floats per patch pixel: 16 channels + 4 pad (80 B keeps b128 alignment)

template <int KS, int NB, int MBW>
__global__ __launch_bounds__(256) void conv_igemm_f32(IgemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;

    int tile = blockIdx.x;
    const int tx = tile % a.tiles_x;
    tile /= a.tiles_x;
    const int ty = tile % a.tiles_y;
    const int ti = tile / a.tiles_y;
    const int n0 = ti * a.TI, y0 = ty * a.TH, x0 = tx * a.TW;
    const int co0 = blockIdx.y * (16 * NB);

    const int PW = a.TW + KS - 1, PH = a.TH + KS - 1;
    const int PPI = PH * PW, PP = a.TI * PPI;
    const int TPI = a.TH * a.TW, TP = a.TI * TPI;
    const int nblk = (TP + 15) >> 4;

    int a_off[MBW];
    int my_nblk = 0;
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        const int blk = wave + 4 * i;
        if (blk < nblk) my_nblk = i + 1;
        int t = blk * 16 + l15;
        if (t >= TP) t = 0;
        const int img = t / TPI;
        const int rem = t - img * TPI;
        const int r = rem / a.TW;
        const int c = rem - r * a.TW;
        a_off[i] = (img * PPI + r * PW + c) * IG_S + 4 * g;
    }

    f32x4 acc[MBW][NB];
#pragma unroll
    for (int i = 0; i < MBW; ++i)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[i][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinP >> 4;
    const size_t tapstride = (size_t)(a.CinP >> 2) * a.CoutP * 4;

    for (int cc = 0; cc < nchunks; ++cc) {
        // ---- stage the 16-channel slice of the input patch (zero-filled outside the image) ----
        for (int q = tid; q < PP * 4; q += 256) {
            const int p = q >> 2, part = q & 3;
            const int img = p / PPI;
            const int rem = p - img * PPI;
            const int pr = rem / PW;
            const int pc = rem - pr * PW;
            const int n = n0 + img, gy = y0 + pr - a.pad, gx = x0 + pc - a.pad;
            const int ci = cc * 16 + part * 4;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (n < a.N && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && ci < a.Cin)
                v = *(const f32x4*)(a.in + (((size_t)n * a.H + gy) * a.W + gx) * a.Cin + ci);
            *(f32x4*)(lds + p * IG_S + part * 4) = v;
        }
        __syncthreads();

        const float* wc = a.wpk + ((size_t)(cc * 4 + g) * a.CoutP + co0 + l15) * 4;
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int tap_off = ((tap / KS) * PW + (tap % KS)) * IG_S;
            f32x4 b[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = *(const f32x4*)(wc + (size_t)tap * tapstride + nb * 64);
#pragma unroll
            for (int i = 0; i < MBW; ++i) {
                if (i < my_nblk) {
                    const f32x4 av = *(const f32x4*)(lds + a_off[i] + tap_off);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[i][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], b[nb][r], acc[i][nb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 16x16x4: column (cout) = lane&15, row (pixel) = 4*(lane>>4)+j ----
#pragma unroll
    for (int i = 0; i < MBW; ++i) {
        if (i < my_nblk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = (wave + 4 * i) * 16 + g * 4 + j;
                if (t < TP) {
                    const int img = t / TPI;
                    const int rem = t - img * TPI;
                    const int r = rem / a.TW;
                    const int c = rem - r * a.TW;
                    const int n = n0 + img, y = y0 + r, x = x0 + c;
                    if (n < a.N && y < a.Ho && x < a.Wo) {
                        const size_t obase = (((size_t)n * a.Ho + y) * a.Wo + x) * a.Cout;
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int co = co0 + nb * 16 + l15;
                            if (co < a.Cout) {
                                float v = acc[i][nb][j];
                                if (a.bias) v += a.bias[co];
                                v = act_apply(v, a.act, a.slope);
                                if (a.ysave) v *= act_grad_from_output(a.ysave[obase + co], a.mask_act, a.slope);
                                a.out[obase + co] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- weight packing --------------------------------------------------------------------------------
// forward : P[tap][ci/4][co][ci%4]            = W[co][ci][ky][kx],                tap = ky*KS+kx
// dgrad   : P[tap'][co/4][ci][co%4]           = W[co][ci][ky][kx],  tap' = (KS-1-ky)*KS + (KS-1-kx)
//           (the "input channels" of the dgrad GEMM are the forward Cout and vice versa)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int KS,
                                    int KinP, int NoutP, int transpose) {
    // p has KS*KS * (KinP/4) * NoutP * 4 elements; K-side channel = (transpose ? co : ci)
    const size_t total = (size_t)KS * KS * KinP * NoutP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int r = idx & 3;
        size_t rest = idx >> 2;
        const int no = rest % NoutP;
        rest /= NoutP;
        const int kq = rest % (KinP / 4);
        const int tap = rest / (KinP / 4);
        const int kc = kq * 4 + r;
        float v = 0.f;
        if (!transpose) {
            if (kc < Cin && no < Cout) v = w[(((size_t)no * Cin + kc) * KS + tap / KS) * KS + tap % KS];
        } else {
            const int ky = KS - 1 - tap / KS, kx = KS - 1 - tap % KS;
            if (kc < Cout && no < Cin) v = w[(((size_t)kc * Cin + no) * KS + ky) * KS + kx];
        }
        p[idx] = v;
    }
}

int aesr_launch_pack_weights(const float* w, float* p, int Cout, int Cin, int KS, int KinP, int NoutP, int transpose,
                             hipStream_t st) {
    const size_t total = (size_t)KS * KS * KinP * NoutP;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, st, w, p, Cout, Cin, KS, KinP, NoutP, transpose);
    AESR_LAUNCH_CHECK("pack_weights");
    return AESR_OK;
}

template <int KS, int NB, int MBW>
static int launch_one(const IgemmArgs& a, int ntiles, hipStream_t st) {
    const int PP = a.TI * (a.TH + KS - 1) * (a.TW + KS - 1);
    const size_t shmem = (size_t)PP * IG_S * sizeof(float);
    if (shmem > 160 * 1024) {
        aesr_set_error("conv_igemm: tile needs %zu B of LDS", shmem);
        return AESR_ERR_ARG;
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_f32<KS, NB, MBW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    dim3 grid(ntiles, a.CoutP / (16 * NB));
    hipLaunchKernelGGL((conv_igemm_f32<KS, NB, MBW>), grid, dim3(256), shmem, st, a);
    AESR_LAUNCH_CHECK("conv_igemm_f32");
    return AESR_OK;
}

int aesr_launch_conv_igemm(const IgemmArgs& a, int KS, int NB, int MBW, hipStream_t st) {
    const int TP = a.TI * a.TH * a.TW;
    const int nblk = (TP + 15) / 16;
    if (nblk > 4 * MBW) {
        aesr_set_error("conv_igemm: tile of %d pixels needs %d M-blocks > 4*MBW=%d", TP, nblk, 4 * MBW);
        return AESR_ERR_ARG;
    }
    if (a.CoutP % (16 * NB) != 0 || a.CinP % 16 != 0 || a.Cin % 4 != 0) {
        aesr_set_error("conv_igemm: bad channel padding Cin=%d CinP=%d CoutP=%d NB=%d", a.Cin, a.CinP, a.CoutP, NB);
        return AESR_ERR_ARG;
    }
    const int ntiles = ceil_div(a.N, a.TI) * a.tiles_y * a.tiles_x;
#define IG_CASE(ks, nb, mbw) \
    if (KS == ks && NB == nb && MBW == mbw) return launch_one<ks, nb, mbw>(a, ntiles, st);
    IG_CASE(3, 1, 4) IG_CASE(3, 1, 8) IG_CASE(3, 2, 4) IG_CASE(3, 2, 8) IG_CASE(3, 4, 4) IG_CASE(3, 4, 8)
    IG_CASE(1, 1, 4) IG_CASE(1, 1, 8) IG_CASE(1, 2, 4) IG_CASE(1, 2, 8) IG_CASE(1, 4, 4) IG_CASE(1, 4, 8)
#undef IG_CASE
    aesr_set_error("conv_igemm: no instantiation for KS=%d NB=%d MBW=%d", KS, NB, MBW);
    return AESR_ERR_UNSUPPORTED;
}
