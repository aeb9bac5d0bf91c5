// Device-side bodies of the parameter-side preparation kernels (filter packing for the implicit GEMM, Winograd filter transform,
// stem folding, flipped filter of the single-output-channel convolution), shared by the stand-alone kernels (conv_igemm.hip,
// conv_wino.hip, conv_thin.hip) and by the ONE-launch preparation of all of a step's weights (prep.hip).
#pragma once
#include "aesr_common.h"

constexpr int PK_WN_TN = 32;        // = WN_TN of conv_wino.hip: output channels per Winograd work item

// ---- weight packing --------------------------------------------------------------------------------
// P[ci chunk][cout tile][tap][q(4)][col(TN)][r(4)]   K-side channel kc = chunk*16 + q*4 + r, N-side channel no = tile*TN + col
// forward : P = W[no][kc][ky][kx],                  tap = ky*KS+kx
// dgrad   : P = W[kc][no][KS-1-ky][KS-1-kx]         (the "input channels" of the dgrad GEMM are the forward Cout)
__device__ __forceinline__ void pack_elements(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int KS,
                                              int KinP, int NoutP, int TN, int transpose, size_t first, size_t stride) {
    const size_t total = (size_t)KS * KS * KinP * NoutP;
    const int ncot = NoutP / TN, KS2 = KS * KS;
    for (size_t idx = first; idx < total; idx += stride) {
        const int r = idx & 3;
        size_t rest = idx >> 2;
        const int col = rest % TN;
        rest /= TN;
        const int q = rest & 3;
        rest >>= 2;
        const int tap = rest % KS2;
        rest /= KS2;
        const int cot = rest % ncot;
        const int chunk = rest / ncot;
        const int kc = chunk * 16 + q * 4 + r, no = cot * TN + col;
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

// ---- weight transform + packing ----------------------------------------------------------------------------------------
// U[ci chunk][cout tile][xi = 4i+j][q(4)][col(32)][r(4)] = (G g G^T)[i][j] for K-side channel kc = chunk*16 + q*4 + r and
// N-side channel no = tile*32 + col.   forward: g = w[no][kc][.][.];   data gradient: g = flip(w[kc][no][.][.])
__device__ __forceinline__ void wino_pack_elements(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int KinP,
                                                   int NoutP, int transpose, size_t first, size_t stride) {
    // one thread per (chunk, tile, q, col, r) = per (kc, no) pair: reads the 9 taps, writes the 16 positions
    const size_t pairs = (size_t)KinP * NoutP;
    const int ncot = NoutP / PK_WN_TN;
    for (size_t idx = first; idx < pairs; idx += stride) {
        const int r = idx & 3;
        size_t rest = idx >> 2;
        const int col = rest % PK_WN_TN;
        rest /= PK_WN_TN;
        const int q = rest & 3;
        rest >>= 2;
        const int cot = rest % ncot;
        const int chunk = rest / ncot;
        const int kc = chunk * 16 + q * 4 + r, no = cot * PK_WN_TN + col;
        float gk[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float v = 0.f;
                if (!transpose) {
                    if (kc < Cin && no < Cout) v = w[(((size_t)no * Cin + kc) * 3 + ky) * 3 + kx];
                } else {
                    if (kc < Cout && no < Cin) v = w[(((size_t)kc * Cin + no) * 3 + (2 - ky)) * 3 + (2 - kx)];
                }
                gk[ky][kx] = v;
            }
        float Gg[4][3];                 // G g: rows g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            Gg[0][kx] = gk[0][kx];
            Gg[1][kx] = 0.5f * (gk[0][kx] + gk[1][kx] + gk[2][kx]);
            Gg[2][kx] = 0.5f * (gk[0][kx] - gk[1][kx] + gk[2][kx]);
            Gg[3][kx] = gk[2][kx];
        }
        float* dst = p + ((((size_t)chunk * ncot + cot) * 16) * 4 + q) * (PK_WN_TN * 4) + col * 4 + r;      // + xi * (4*TN*4)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float u0 = Gg[i][0], u3 = Gg[i][2];
            const float u1 = 0.5f * (Gg[i][0] + Gg[i][1] + Gg[i][2]);
            const float u2 = 0.5f * (Gg[i][0] - Gg[i][1] + Gg[i][2]);
            dst[(size_t)(i * 4 + 0) * (4 * PK_WN_TN * 4)] = u0;
            dst[(size_t)(i * 4 + 1) * (4 * PK_WN_TN * 4)] = u1;
            dst[(size_t)(i * 4 + 2) * (4 * PK_WN_TN * 4)] = u2;
            dst[(size_t)(i * 4 + 3) * (4 * PK_WN_TN * 4)] = u3;
        }
    }
}

// folded[0][t][co] = sum_c W1[co,c,t]*ws[c]   folded[1][t][co] = sum_c W1[co,c,t]*bs[c]   (double accumulation): the encoder stem
// folded into the first 3x3 convolution (conv_thin.hip)
__device__ __forceinline__ void stem_fold_elements(const float* __restrict__ ws, const float* __restrict__ bs, const float* __restrict__ w1,
                                                   float* __restrict__ folded, int Cs, int C1, int first, int stride) {
    for (int o = first; o < 9 * C1; o += stride) {
        const int t = o / C1, co = o - t * C1;
        double sw = 0.0, sb = 0.0;
        for (int c = 0; c < Cs; ++c) {
            const double wv = (double)w1[((size_t)co * Cs + c) * 9 + t];
            sw += wv * (double)ws[c];
            if (bs) sb += wv * (double)bs[c];
        }
        folded[o] = (float)sw;
        folded[9 * C1 + o] = (float)sb;
    }
}

// Cout == 1 conv: wexp[t][ci] = W[0,ci,8-t] (flipped filter for the data gradient)
__device__ __forceinline__ void cout1_flip_elements(const float* __restrict__ w, float* __restrict__ wexp, int Cin, int first, int stride) {
    for (int o = first; o < 9 * Cin; o += stride) {
        const int t = o / Cin, ci = o - t * Cin;
        wexp[o] = w[ci * 9 + (8 - t)];
    }
}
