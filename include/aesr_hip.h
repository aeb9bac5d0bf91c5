/* libaesr_hip.so -- C ABI of the MI355X (gfx950) kernels behind the ae_combined hot path.
 *
 * The reference (qurAI-amsterdam/SuperResolution_aniso_MRI) is pure Python on PyTorch: it has no FFI of
 * its own, every op below is an ATen/cuDNN call issued from the cited reference line.  This header is the
 * boundary a maintainer binds instead (ctypes stub: INTEGRATION.md).  Conventions:
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer owned by the caller (PyTorch's
 *     caching allocator in the shipped host code) unless named *_host; no allocation, no host sync and no
 *     global state inside the library besides a tile-plan cache; workspaces are caller-provided and sized
 *     by the *_workspace_* helpers;
 *   - activations are fp32 NHWC ("[N,H,W,C]"); weights cross the boundary in PyTorch's [Cout][Cin][KH][KW];
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all work is enqueued on it;
 *   - every entry returns 0 on success, non-zero on error (AESR_ERR_*), never throws; the message of the
 *     calling thread's last error is aesr_last_error_string();
 *   - activation codes: 0 none, 1 LeakyReLU(slope), 2 ReLU, 3 sigmoid;
 *   - BatchNorm "groups": a batch may consist of up to 4 consecutive sub-batches with independent batch
 *     statistics; group g covers images [nstart[g], nstart[g+1]) (nstart has G+1 entries).
 */
#ifndef AESR_HIP_H
#define AESR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AESR_ABI_VERSION 1
#define AESR_ERR_ARG 1
#define AESR_ERR_HIP 2
#define AESR_ERR_UNSUPPORTED 3

int aesr_version(void);
const char* aesr_last_error_string(void);

/* ---- convolution, stride 1 (nn.Conv2d: networks/acai_vanilla.py:51,55-56,68,70,87-88,96,98;
 *      lpips/pretrained_networks.py:107-116) ------------------------------------------------------------ */

/* Number of floats of the packed-weight buffer for a [Cout][Cin][KS][KS] filter.
 * transpose = 0: forward operand; 1: data-gradient operand (flipped taps, channels swapped). */
size_t aesr_conv2d_packed_floats(int Cout, int Cin, int KS, int transpose);
int aesr_conv2d_pack(const float* w, float* packed, int Cout, int Cin, int KS, int transpose, void* stream);
/* The same for many filters in one launch per 32 jobs (all the layers of a network after an optimizer step).  The job
 * array is read on the host during the call; the pointers inside travel as kernel arguments (graph-capture safe). */
typedef struct aesr_pack_job {
    const float* w;
    float* packed;
    int Cout, Cin, KS, transpose;
} aesr_pack_job;
int aesr_conv2d_pack_many(const aesr_pack_job* jobs_host, int njobs, void* stream);

/* ALL parameter-side preparation of a training step in ONE launch per 32 jobs (csrc/prep.hip): what aesr_conv2d_pack_many,
 * aesr_conv2d_wino_pack_many, aesr_stemconv_fold and the filter flip inside aesr_conv2d_cout1_dgrad do in four to five launches (at a
 * small data-parallel shard each is a ~5 us graph node).  Same code, bitwise the same results.  kind:
 *   AESR_PREP_PACK       w [Cout][Cin][KS][KS] -> out = the implicit-GEMM operand (aesr_conv2d_packed_floats), transpose as there
 *   AESR_PREP_WINO_PACK  ... -> out = U = G g G^T (aesr_conv2d_wino_packed_floats), KS = 3
 *   AESR_PREP_STEM_FOLD  w = W1 [C1][Cs][3][3], aux0 / aux1 = stem weight / bias [Cs] (aux1 may be NULL), Cout = C1, Cin = Cs
 *                        -> out = aesr_stemconv_folded_floats(C1) floats (as aesr_stemconv_fold)
 *   AESR_PREP_COUT1_FLIP w [1][Cin][3][3] -> out [9][Cin], the flipped filter aesr_conv2d_cout1_dgrad_pre takes
 * The job array is read on the host during the call; pointers are device pointers. */
#define AESR_PREP_PACK 0
#define AESR_PREP_WINO_PACK 1
#define AESR_PREP_STEM_FOLD 2
#define AESR_PREP_COUT1_FLIP 3
typedef struct aesr_prep_job {
    const float* w;
    const float* aux0;
    const float* aux1;
    float* out;
    int kind, Cout, Cin, KS, transpose;
} aesr_prep_job;
int aesr_weight_prep_many(const aesr_prep_job* jobs_host, int njobs, void* stream);

/* out = act(conv(in, w) + bias)                         [MFMA implicit GEMM; Cin % 4 == 0]
 * `packed` from aesr_conv2d_pack(transpose=0).  bias may be NULL.  Ho = H + 2*pad - KS + 1. */
int aesr_conv2d_fwd(const float* in, const float* packed, const float* bias, float* out, int N, int H, int W, int Cin,
                    int Cout, int KS, int pad, int act, float slope, void* stream);

/* The same with a caller-owned workspace that lets the planner split the input channels of under-filled layers (few, deep work
 * items: VGG conv4/5 at 20x20 / 10x10) into extra work items + a fix-up pass.  aesr_conv2d_workspace_floats returns the size the
 * plan of this shape wants (0: no split, workspace may be NULL).  For the data gradient pass the forward conv's N, H, W, Cin,
 * Cout, KS, pad to aesr_conv2d_dgrad_workspace_floats.  Results equal the plain entry points up to fp32 summation order. */
size_t aesr_conv2d_workspace_floats(int N, int H, int W, int Cin, int Cout, int KS, int pad);
size_t aesr_conv2d_dgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int KS, int pad);
int aesr_conv2d_fwd_ws(const float* in, const float* packed, const float* bias, float* out, float* workspace, int N, int H, int W,
                       int Cin, int Cout, int KS, int pad, int act, float slope, void* stream);
int aesr_conv2d_dgrad_ws(const float* dy, const float* packed_t, const float* x_saved, float* dx, float* workspace, int N, int H,
                         int W, int Cin, int Cout, int KS, int pad, int mask_act, float slope, void* stream);

/* dx = conv_transpose(dy, w) * act'(x_saved)            [same kernel on the transposed packing; Cout % 4 == 0]
 * dy is [N,Ho,Wo,Cout], dx is [N,H,W,Cin]; x_saved (may be NULL) is the saved OUTPUT of the activation that
 * produced the convolution's input, mask_act its code: fuses the LeakyReLU/ReLU backward of the previous layer. */
int aesr_conv2d_dgrad(const float* dy, const float* packed_t, const float* x_saved, float* dx, int N, int H, int W,
                      int Cin, int Cout, int KS, int pad, int mask_act, float slope, void* stream);

/* dw[Cout][Cin][KS][KS], db[Cout] (db may be NULL)      [MFMA split-K, fixed-order slab reduction]
 * workspace: aesr_conv2d_wgrad_workspace_floats(...) floats. */
size_t aesr_conv2d_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int KS, int pad);
int aesr_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W,
                      int Cin, int Cout, int KS, int pad, void* stream);

/* Bandwidth-bound special cases (K or N of the GEMM <= 4): weights in PyTorch layout, no packing.
 * small-Cin forward (stem networks/acai_vanilla.py:51; VGG conv1_1 with lpips/networks_basic.py:93-100 folded:
 * bcast != 0 means `in` has ONE channel and virtual channel c = ca[c]*in + cb[c]); transpose != 0 runs the data
 * gradient of a Cout<=4 conv (w given as [Cin_fwd... see DESIGN.md]); x_saved/mask_act as in aesr_conv2d_dgrad. */
int aesr_conv2d_smallcin_fwd(const float* in, const float* w, const float* bias, const float* y_saved, float* out, int N,
                             int H, int W, int Cin, int Cout, int KS, int pad, int act, int mask_act, float slope,
                             int transpose, int bcast, const float* ca_host, const float* cb_host, void* stream);
int aesr_conv2d_smallcin_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int KS,
                               int pad, int bcast, const float* ca_host, void* stream);
/* 1x1 small-Cin weight/bias gradient (the stem).  workspace: aesr_small_wgrad_workspace_floats(Cout*(Cin+1)). */
size_t aesr_small_wgrad_workspace_floats(int nout);
int aesr_conv2d_smallcin_wgrad(const float* in, const float* dout, float* dw, float* db, float* workspace, int N, int H,
                               int W, int Cin, int Cout, int pad, void* stream);
/* 3x3 pad-1 Cout==1 forward (output conv networks/acai_vanilla.py:98 + Sigmoid): out[N,H,W,1] = act(conv(x, w[1][Cin][3][3]) + bias). */
int aesr_conv2d_cout1_fwd(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int Cin, int act,
                          float slope, void* stream);
/* 3x3 pad-1 Cout==1 weight/bias gradient (output conv networks/acai_vanilla.py:98).  Cin = 4 * 2^k <= 256 (else a slower generic kernel when Cin divides 256);
 * workspace: aesr_conv2d_cout1_workspace_floats(Cin) floats. */
size_t aesr_conv2d_cout1_workspace_floats(int Cin);
int aesr_conv2d_cout1_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W,
                            int Cin, void* stream);
/* 3x3 pad-1 Cout==1 data gradient: dx[N,H,W,Cin] = conv^T(dy[N,H,W,1], w) * act'(y_saved) (y_saved = output of the
 * activation that produced the conv's input, or NULL).  Cin = 4 * 2^k <= 256.  workspace: >= 9*Cin floats (the flipped filter). */
int aesr_conv2d_cout1_dgrad(const float* dy, const float* w, const float* y_saved, float* dx, float* workspace, int N, int H,
                            int W, int Cin, int mask_act, float slope, void* stream);
/* The same with the flipped filter [9][Cin] prepared beforehand (AESR_PREP_COUT1_FLIP of aesr_weight_prep_many): one launch. */
int aesr_conv2d_cout1_dgrad_pre(const float* dy, const float* w_flipped, const float* y_saved, float* dx, int N, int H, int W, int Cin,
                                int mask_act, float slope, void* stream);

/* ---- encoder stem folded into the first 3x3 convolution (networks/acai_vanilla.py:51,55) ------------------------
 * Conv2d(1, Cs, 1, padding=stem_pad) followed directly (no non-linearity) by Conv2d(Cs, C1, 3, padding=1) + activation,
 * computed as ONE 1 -> C1 3x3 convolution of the single-channel image with the folded filter
 * weff[t][co] = sum_c w1[co,c,t]*w_stem[c] and the per-tap bias beff[t][co] = sum_c w1[co,c,t]*b_stem[c] (counted only for
 * taps inside the stem's (H+2*stem_pad)^2 output grid).  The Cs-channel stem tensor is never materialised.
 * C1 = 4 * 2^k <= 256.  folded: aesr_stemconv_folded_floats(C1) floats, refreshed by aesr_stemconv_fold whenever
 * the parameters change.  out: [N, H+2*stem_pad, W+2*stem_pad, C1]. */
size_t aesr_stemconv_folded_floats(int C1);
int aesr_stemconv_fold(const float* w_stem, const float* b_stem, const float* w1, float* folded, int Cs, int C1, void* stream);
int aesr_stemconv_fwd(const float* x, const float* folded, const float* b1, float* out, int N, int H, int W, int C1,
                      int stem_pad, int act, float slope, void* stream);
/* Gradients of all four parameter tensors from g = dL/d(pre-activation output) [N,H+2p,W+2p,C1]:
 * dw_stem[Cs], db_stem[Cs] (may be NULL with b_stem), dw1[C1][Cs][3][3], db1[C1] (may be NULL).
 * workspace: aesr_stemconv_workspace_floats(C1) floats. */
size_t aesr_stemconv_workspace_floats(int C1);
int aesr_stemconv_wgrad(const float* x, const float* g, const float* w_stem, const float* b_stem, const float* w1,
                        float* dw_stem, float* db_stem, float* dw1, float* db1, float* workspace, int N, int H, int W, int Cs,
                        int C1, int stem_pad, void* stream);

/* Stride-2 2x2 convolution (networks/acai_vanilla_strided.py:19) = space-to-depth + 1x1 MFMA conv:
 * out[n,y,x,(ky*2+kx)*C+c] = x[n,2y+ky,2x+kx,c], out is [N,H/2,W/2,4C]; the inverse scatters a [N,H/2,W/2,4C] gradient
 * back to [N,H,W,C] (zero in a dropped odd last row / column).  H, W are the FULL-resolution sizes in both calls. */
int aesr_space_to_depth2(const float* x, float* out, int N, int H, int W, int C, void* stream);
int aesr_depth_to_space2(const float* g, float* dx, int N, int H, int W, int C, void* stream);

/* ---- stand-alone x2 resampling (networks/acai_vanilla.py:59,92 without BatchNorm; networks/ae_standard.py:41,68) ---------
 * mode: AESR_RS_POOL AvgPool2d(2) (out [N,H/2,W/2,C]); AESR_RS_NEAREST / AESR_RS_BILINEAR Upsample(scale_factor=2)
 * (out [N,2H,2W,C]; bilinear = align_corners False).  C % 4 == 0.  H, W are the INPUT sizes of the forward op in both calls.
 * Backward: dx[N,H,W,C] from gout, times act'(x_saved) when x_saved (= the forward input, output of that activation) is given. */
#define AESR_RS_POOL 1
#define AESR_RS_NEAREST 2
#define AESR_RS_BILINEAR 3
int aesr_resample2_fwd(const float* x, float* out, int N, int H, int W, int C, int mode, void* stream);
int aesr_resample2_bwd(const float* gout, const float* x_saved, float* dx, int N, int H, int W, int C, int mode, int mask_act,
                       float slope, void* stream);

/* ---- BatchNorm2d (+AvgPool2d(2) / nearest Upsample x2) (networks/acai_vanilla.py:58-59,90-92) ------------ */
#define AESR_BN_NONE 0
#define AESR_BN_POOL 1
#define AESR_BN_UP 2
#define AESR_BN_NWG 512   /* partial rows per group of the stats / backward-reduce passes */

/* sums[G][2][C] (double): per group and channel sum(y), sum(y^2).  partial: G*AESR_BN_NWG*2*C floats. */
int aesr_bn_stats(const float* y, float* partial, double* sums, int HW, int C, int G, const int* nstart_host,
                  void* stream);
/* mean/invstd/scale/shift [G][C].  train: from sums and counts_host[G] (HOST doubles: elements per channel and group,
 * the GLOBAL count under data parallel); running_mean/var/num_batches_tracked updated group after group when update_running.
 * eval (train == 0): from the running buffers; sums/counts ignored. */
int aesr_bn_finalize(const double* sums, const double* counts_host, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                     float* scale, float* shift, int C, int G, float momentum, float eps, int train, int update_running,
                     void* stream);
/* aesr_bn_stats + aesr_bn_finalize(train) without the sums round trip (two launches): the single-process path, where no
 * SyncBN exchange sits between the two.  Same arithmetic, same outputs. */
int aesr_bn_stats_finalize(const float* y, float* partial, const double* counts_host, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                           float* scale, float* shift, int HW, int C, int G, const int* nstart_host, float momentum, float eps,
                           int update_running, void* stream);
/* out = scale[g]*f(y) + shift[g], f = identity / 2x2 mean (floor) / nearest x2. */
int aesr_bn_apply(const float* y, const float* scale, const float* shift, float* out, int N, int H, int W, int C, int mode,
                  int G, const int* nstart_host, void* stream);
/* Data parallel, train mode: finalize (from the all-reduced [G][2][C] sums) + apply in ONE launch -- every block derives scale / shift
 * itself, block 0 writes mean / invstd / scale / shift for the backward pass and updates the running statistics (same arithmetic as
 * aesr_bn_finalize + aesr_bn_apply).  aesr_bn_fused_supported: G * C <= 1024.  aesr_bn_bwd_apply fuses its finalize the same way. */
int aesr_bn_fused_supported(int C, int G);
int aesr_bn_finalize_apply(const double* sums, const double* counts_host, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                           const float* y, float* out, int N, int H, int W, int C, int mode, int G, const int* nstart_host, float momentum,
                           float eps, int update_running, void* stream);
/* backward, step 1: sums[G][2][C] (double) = sum(g), sum(g*xhat) with g the gradient w.r.t. the BN output seen
 * through the pool / upsample.  gout is [N,Ho,Wo,C] (pool: Ho=H/2; up: Ho=2H; none: Ho=H). */
int aesr_bn_bwd_reduce(const float* gout, const float* y, const float* mean, const float* invstd, float* partial,
                       double* sums, int N, int H, int W, int C, int mode, int G, const int* nstart_host, void* stream);
/* step 2: dpre = scale*(g - s1/M - xhat*s2/M) * act'(y);  dgamma = sum_g s2, dbeta = sum_g s1.
 * coef: G*2*C floats of scratch. */
int aesr_bn_bwd_apply(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale,
                      const double* sums, const double* counts_host, float* coef, float* dgamma, float* dbeta,
                      float* dpre, int N, int H, int W, int C, int mode, int act, float slope, int G,
                      const int* nstart_host, void* stream);
/* aesr_bn_bwd_reduce + aesr_bn_bwd_apply without the sums round trip (three launches), for the single-process path. */
int aesr_bn_bwd(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale, float* partial,
                const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N, int H, int W, int C,
                int mode, int act, float slope, int G, const int* nstart_host, void* stream);

/* ---- LPIPS-VGG (lpips/networks_basic.py:63-91, lpips/common.py:12-14, lpips/pretrained_networks.py:107-116) ---- */
/* ScalingLayer (lpips/networks_basic.py:93-100, with the 2x-1 of lpips/perceptual.py:29-31 folded in) of a 1-channel image,
 * materialised as 4 channels (c, c, c, 0) so VGG conv1_1 runs on the MFMA kernel: out4[p][c] = ca[c]*x[p] + cb[c];
 * backward: dx[p] = sum_c ca[c]*d4[p][c].  n = number of pixels. */
int aesr_scale_expand_fwd(const float* x, float* out4, size_t n, const float* ca_host, const float* cb_host, void* stream);
int aesr_scale_expand_bwd(const float* d4, float* dx, size_t n, const float* ca_host, void* stream);
/* nn.MaxPool2d(2): out [N,H/2,W/2,C]. */
int aesr_maxpool2_fwd(const float* x, float* out, int N, int H, int W, int C, void* stream);
/* dx = (scatter of gout to the FIRST maximum of each window + gadd) * (relu_mask ? x > 0 : 1); gadd may be NULL. */
int aesr_maxpool2_bwd(const float* gout, const float* x, const float* gadd, float* dx, int N, int H, int W, int C,
                      int relu_mask, void* stream);
#define AESR_LPIPS_NCH 64
/* One tap.  f is [2B,HW,C] (branch 0 = images 0..B-1, branch 1 = images B..2B-1), lin_w [C];
 * partial [B][AESR_LPIPS_NCH] = chunk sums over pixels of sum_c w_c (f0/(|f0|+1e-10) - f1/(|f1|+1e-10))^2.  C in {64,128,256,512}. */
int aesr_lpips_tap_fwd(const float* f, const float* lin_w, float* partial, int B, int HW, int C, void* stream);
/* gradient of d w.r.t. branch 0 of the tap: gf0 [B,HW,C]; gd [B] = dL/dd[n]. */
int aesr_lpips_tap_bwd(const float* f, const float* lin_w, const float* gd, float* gf0, int B, int HW, int C, void* stream);
/* d[n] = sum_k (1/hw[k]) * sum_chunk partials[k][n][chunk]; partials_host: ntaps DEVICE pointers in a host array. */
int aesr_lpips_finalize(const float* const* partials_host, const int* hw_host, int ntaps, float* d, int B, void* stream);

/* ---- latent lerp (kwatsch/cardiac/trainer_ae.py:173; kwatsch/brain/trainer_ae.py:264-266;
 *      generate_hr_volumes.py:88) ------------------------------------------------------------------------- */
/* z [2B][per], zmix [B][per]: zmix[b] = a_from[b]*z[b] + a_to[b]*z[B+b]; per % 4 == 0. */
int aesr_lerp_fwd(const float* z, const float* a_from, const float* a_to, float* zmix, int B, size_t per, void* stream);
int aesr_lerp_bwd(const float* dzmix, const float* a_from, const float* a_to, float* dz, int B, size_t per, void* stream);
/* The decoder input of the step in one pass (kwatsch/cardiac/trainer_ae.py:20-30: dec(z) and dec(z_mix) run as one batch here):
 * zcat[3B][per] = [z | a_from*z[:B] + a_to*z[B:]];  gradient dz[2B][per] = g[:2B] + (a_from, a_to) * g[2B:]. */
int aesr_lerp_cat_fwd(const float* z, const float* a_from, const float* a_to, float* zcat, int B, size_t per, void* stream);
/* Slice synthesis, all mixes of a volume in one launch (generate_hr_volumes.py:46-53,88; kwatsch/img_interpolation.py:57-89):
 * out[k][i] = act(alphas[k] * z[i + 1] + (1 - alphas[k]) * z[i]) for the Z - 1 pairs of neighbouring slices and n <= 16 coefficients
 * (host array).  z: [Z][per_slice] latents (act = AESR_ACT_NONE), or the pre-activations of the decoder's first convolution with that
 * layer's activation -- the convolution is linear, so it runs once per slice instead of once per synthesised slice. */
int aesr_lerp_multi(const float* z, float* out, int Z, size_t per_slice, const float* alphas_host, int n, int act, float slope,
                    void* stream);
int aesr_lerp_cat_bwd(const float* g, const float* a_from, const float* a_to, float* dz, int B, size_t per, void* stream);
/* generate_hr_volumes.py:57-67 (the volume is assembled slice by slice on the host there): out[(Z-1)(n+1)+1][per_slice], slot i (n+1) = orig[i],
 * slot i (n+1) + k + 1 = synth[k][i] (synth: [n][Z-1][per_slice], the order aesr_lerp_multi mixes in), every element clamped to [lo, hi]. */
int aesr_interleave_clamp(const float* orig, const float* synth, float* out, int Z, int n, size_t per_slice, float lo, float hi, void* stream);

/* ---- losses (kwatsch/base_trainer.py:177; kwatsch/cardiac/trainer_ae.py:181) ------------------------------ */
#define AESR_MSE_NPART 512
/* loss[0] = mean((a-b)^2); partial: AESR_MSE_NPART doubles. */
int aesr_mse_fwd(const float* a, const float* b, double* partial, float* loss, size_t n, void* stream);
/* da = 2*(a-b)*gloss[0]/n */
int aesr_mse_bwd(const float* a, const float* b, const float* gloss, float* da, size_t n, void* stream);
/* The three mean-squared errors of the ae_combined step (reconstruction kwatsch/base_trainer.py:177, synthesis
 * kwatsch/cardiac/trainer_ae.py:123 weighted by the device scalar lam[0], and the logged latent distance :181) in ONE launch:
 * out[0] = m1 + lam * m2, out[1] = m1, out[2] = lam * m2, out[3] = m3 with m_k = mean((a_k - b_k)^2) (a3 may be NULL: m3 = 0).
 * workspace: AESR_MSE3_WS doubles that the caller zeroes ONCE before the first call (the kernel leaves them consistent); the sums
 * are fp64 and added in a fixed order (bitwise reproducible).  aesr_mse3_bwd: the gradient of out[0] times gloss[0]:
 * d1 = 2 (a1 - b1) gloss / n1, d2 = 2 lam (a2 - b2) gloss / n2 (d1, d2 may be the two parts of one tensor). */
#define AESR_MSE3_NPART 256
#define AESR_MSE3_WS (3 * AESR_MSE3_NPART + 1)
int aesr_mse3_fwd(const float* a1, const float* b1, size_t n1, const float* a2, const float* b2, size_t n2, const float* a3,
                  const float* b3, size_t n3, const float* lam, double* workspace, float* out4, void* stream);
int aesr_mse3_bwd(const float* a1, const float* b1, size_t n1, const float* a2, const float* b2, size_t n2, const float* lam,
                  const float* gloss, float* d1, float* d2, void* stream);
/* loss[0] = mean |a-b| (F.l1_loss of kwatsch/lap_pyramid_loss.py:65); partial: AESR_MSE_NPART doubles.  da = sign(a-b)*gloss[0]/n. */
int aesr_l1_fwd(const float* a, const float* b, double* partial, float* loss, size_t n, void* stream);
int aesr_l1_bwd(const float* a, const float* b, const float* gloss, float* da, size_t n, void* stream);
/* Laplacian-pyramid pieces on planes [P][H][W] (kwatsch/lap_pyramid_loss.py:23-40):
 * blur5: out = (add ? add : 0) + gain * G(in), G = 5x5 binomial filter /256 with reflect padding (conv_gauss, :37-40); adjoint != 0
 *        applies the transposed operator (gradient of G).  upsample() of the reference = blur5(zero_insert2(x), gain 4).
 * down2: out[P][ceil(H/2)][ceil(W/2)] = in[:, ::2, ::2] (:23-24).  zero_insert2: its transpose, out[P][H][W] (:27-34). */
int aesr_lap_blur5(const float* in, const float* add, float* out, int P, int H, int W, float gain, int adjoint, void* stream);
int aesr_lap_down2(const float* in, float* out, int P, int H, int W, void* stream);
int aesr_lap_zero_insert2(const float* in, float* out, int P, int h, int w, int H, int W, void* stream);
/* Discriminator head (networks/acai_vanilla.py:146-150): out[n] = mean of the M elements of row n; dx[n][:] = g[n] / M. */
int aesr_row_mean_fwd(const float* x, float* out, int N, size_t M, void* stream);
int aesr_row_mean_bwd(const float* g, float* dx, int N, size_t M, void* stream);
/* dpre = dout * act'(y) from the saved activation output (sigmoid of networks/acai_vanilla.py:98). */
int aesr_act_bwd(const float* dout, const float* y, float* dpre, size_t n, int act, float slope, void* stream);

/* ---- Adam (torch.optim.Adam as used by kwatsch/trainer_ae.py:29-30) on a flat buffer ---------------------- */
/* state: 8 floats {steps done t, 1-beta1^(t+1), sqrt(1-beta2^(t+1)), ticket counter (keep at zero), beta1^(t+1) and beta2^(t+1) as two
 * doubles}, i.e. the bias corrections of the step about to run; fill it with aesr_adam_state_init (host, no launch) for a fresh optimizer
 * (t = 0) or a resumed one.  The step and the powers are advanced on the device by the workgroup that finishes last: one launch per
 * step, graph-replay safe.  The betas are doubles (their powers are, as in torch.optim.Adam's bias corrections; the element-wise
 * moments use them rounded to fp32).  zero_grad != 0 leaves g at zero (a caller that keeps no gradients between steps saves its memset). */
void aesr_adam_state_init(float* state_host8, double steps_done, double beta1, double beta2);
int aesr_adam_step(float* p, float* g, float* exp_avg, float* exp_avg_sq, float* state, size_t n, float lr,
                   double beta1, double beta2, float eps, float weight_decay, int zero_grad, void* stream);

/* ---- on-device batch assembly + augmentation of the training triplets (train_cardiac_aesr.py:83-96, datasets/
 * shared_transforms.py:48-120,224-254,297-363,366-447, datasets/ACDC/data4d_simple.py:327-355) ------------------------------
 * One descriptor per sample (HOST array, read during the call; B <= 64 per call): the slices z_from / z_to / z_between of the
 * [Z][H][W] volume at float offset vol_off inside `volumes` are cropped at (oy, ox) (may reach into the zero padding),
 * passed through 1/(1+exp(gain*(cutoff - v))), rotated by k*90 degrees (np.rot90) and written to image[b], image[B+b]
 * ([2B,1,width,width]) and between[b] ([B,1,width,width]). */
typedef struct aesr_triplet_desc {
    long long vol_off;
    int H, W, z_from, z_to, z_between, oy, ox, k;
    float gain, cutoff;
} aesr_triplet_desc;
int aesr_triplet_assemble(const float* volumes, const aesr_triplet_desc* desc_host, int B, int width, float* image,
                          float* between, void* stream);

/* ---- validation metrics on the device (evaluate/metrics.py:111-194: skimage structural_similarity / peak_signal_noise_ratio
 * slice by slice) -------------------------------------------------------------------------------------------------------
 * a, b: [Z][H][W] fp32.  ssim[Z], mse[Z]: fp64 device arrays (mean SSIM with a uniform win x win window, sample covariance,
 * C1 = (k1*data_range)^2, C2 = (k2*data_range)^2; mean squared difference).  win odd, 3..11, <= min(H, W).
 * workspace: aesr_ssim_workspace_doubles(Z, H, W) doubles. */
size_t aesr_ssim_workspace_doubles(int Z, int H, int W);
int aesr_ssim_mse(const float* a, const float* b, double* workspace, double* ssim, double* mse, int Z, int H, int W, int win,
                  double data_range, double k1, double k2, void* stream);

/* ---- BatchNorm2d (train) [+ AvgPool2d(2)] as ONE launch per direction, for SMALL batches (csrc/bn_fused.hip) -----------------------
 * The shard of a data-parallel rank (6 images) makes a BatchNorm call three ~5 us launches each way; a layer of that size fits the LDS
 * of the chip, so one launch of 256 workgroups keeps it there between the statistics and the normalisation, with one grid-wide barrier
 * in between (bounded waits; a wait that gives up counts in aesr_bn_fused1_timeouts and the host raises).  Same arithmetic per element
 * as aesr_bn_stats_finalize + aesr_bn_apply / aesr_bn_bwd; the partial sums are formed over other partitions of the pixels, so
 * statistics agree to fp64 rounding of the sums.  mode: AESR_BN_NONE or AESR_BN_POOL (the folded Upsample runs as NONE).
 * aesr_bn_fused1_supported: 1 when the layer fits (else take the three-launch entry points; the launchers return AESR_ERR_UNSUPPORTED).
 * workspace: aesr_bn_fused1_workspace_floats(C, G) floats.  barrier_state: aesr_bn_fused1_barrier_words() 32-bit words, zeroed ONCE by
 * the caller and then left alone; launches that share one state must be serialised on one stream (one state per network). */
int aesr_bn_fused1_supported(int N, int H, int W, int C, int mode, int G, int backward);
size_t aesr_bn_fused1_workspace_floats(int C, int G);
size_t aesr_bn_fused1_barrier_words(void);
unsigned int aesr_bn_fused1_timeouts(void);
int aesr_bn_fused1_fwd(const float* y, float* out, float* workspace, unsigned int* barrier_state, const double* counts_host, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                       float* scale, float* shift, int N, int H, int W, int C, int mode, int G, const int* nstart_host, float momentum, float eps,
                       int update_running, void* stream);
int aesr_bn_fused1_bwd(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale, float* workspace,
                       unsigned int* barrier_state, const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N, int H,
                       int W, int C, int mode, int act, float slope, int G, const int* nstart_host, void* stream);

/* ---- the same with the SyncBN exchange INSIDE the launch: data parallel over peer-mapped regions (csrc/p2p.hip, csrc/bn_fused.hip) ------
 * the [G][2][C] partial sums of a BatchNorm call are a few hundred bytes -- a one-shot write into every peer's memory
 * over xGMI beats a ring all-reduce, and it needs no launch of its own.  Every rank allocates ONE region (aesr_p2p_alloc, fine-grained
 * device memory of aesr_p2p_region_bytes(world) bytes), publishes its 64-byte IPC handle over any host channel and maps the others'
 * (aesr_p2p_open).  peers_host[r] = rank r's region as mapped into THIS process (peers_host[rank] = the own allocation).  slot: the
 * number of the BatchNorm call inside the step (call order, the same on every rank, < AESR_P2P_SLOTS).  gen_dev: a device word that
 * every rank advances ONCE per step with aesr_p2p_tick before the step's first BatchNorm call (zero-initialised).  Inside the launch
 * workgroup 0 writes the rank's totals into every region, every workgroup waits (bounded; a wait that gives up counts in
 * aesr_bn_fused1_timeouts) for all ranks' totals in its own region and adds them in rank order: identical bits on all ranks.
 * counts_host are the GLOBAL element counts of the groups.  A rank re-uses a slot only after the step's gradient all-reduce, which
 * every rank enters after its last slot: producers never wait for consumers.  OPT-IN (AESR_SYNCBN=p2p): covered on one device
 * (a group of one, two ranks through IPC); no multi-GPU box has run it yet.  world <= 8. */
#define AESR_P2P_HANDLE_BYTES 64
#define AESR_P2P_SLOTS 32
#define AESR_P2P_REC_BYTES (512 * 8 + 128)
int aesr_p2p_alloc(size_t bytes, void** region);
int aesr_p2p_free(void* region);
int aesr_p2p_get_handle(void* region, char* handle64);
int aesr_p2p_open(const char* handle64, void** peer_region);
int aesr_p2p_close(void* peer_region);
size_t aesr_p2p_region_bytes(int world);
int aesr_p2p_tick(unsigned int* gen_dev, void* stream);
int aesr_bn_fused1_fwd_p2p(const float* y, float* out, float* workspace, unsigned int* barrier_state, const double* counts_host, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                           float* scale, float* shift, int N, int H, int W, int C, int mode, int G, const int* nstart_host, float momentum, float eps,
                           int update_running, void* const* peers_host, int world, int rank, int slot, const unsigned int* gen_dev, void* stream);
int aesr_bn_fused1_bwd_p2p(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale, float* workspace,
                           unsigned int* barrier_state, const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N, int H,
                           int W, int C, int mode, int act, float slope, int G, const int* nstart_host, void* const* peers_host, int world, int rank,
                           int slot, const unsigned int* gen_dev, void* stream);

/* ---- VIF of the same protocol (evaluate/metrics.py:65-109 compute_vif_for_batch -> evaluate/vifvec.py:7-63 vifp_mscale, per slice) --
 * ref, dist: [Z][H][W] fp32 in [0, 1].  vif[Z]: fp64 device array, NaN where the denominator is 0 (a black reference slice).  The
 * arithmetic is the reference's on uint8 images (both volumes are converted with uint8(clip(x * 255, 0, 255)) before vifp_mscale is
 * called; scipy's Gaussian filter and numpy's products then stay in uint8: truncation after every filter pass, products modulo 256)
 * -- csrc/vif.hip, oracle/vif_oracle.py.  weights_host: the four 1-D Gaussian kernels back to back, 2 * radii_host[s] + 1 doubles each,
 * as scipy.ndimage computes them (sd = 3.4, 1.8, 1.0, 0.6; radius = int(4 sd + 0.5) <= 14): HOST arrays, read during the call.
 * sigma_nsq: the reference's default 2.0.  workspace: aesr_vif_workspace_bytes(Z, H, W) bytes, 8-byte aligned. */
size_t aesr_vif_workspace_bytes(int Z, int H, int W);
int aesr_vif_mscale(const float* ref, const float* dist, void* workspace, double* vif, int Z, int H, int W, const double* weights_host,
                    const int* radii_host, double sigma_nsq, void* stream);

/* ---- Winograd F(2x2,3x3) form of the same 3x3 / padding-1 convolutions (2.25x fewer matrix-core flops; csrc/conv_wino.hip) ------
 * Same results as aesr_conv2d_fwd / _dgrad up to fp32 rounding of the transforms (2-4e-7 relative).  Needs K-side channels % 16
 * == 0 and N-side channels % 32 == 0 (aesr_conv2d_wino_supported; transpose = 1 asks for the data-gradient roles).  The filter is
 * handed over pre-transformed: U = G g G^T, packed by aesr_conv2d_wino_pack_many (job.transpose: 0 forward, 1 data gradient;
 * job.KS must be 3) into aesr_conv2d_wino_packed_floats floats. */
int aesr_conv2d_wino_supported(int Cin, int Cout, int KS, int pad, int transpose);
/* Which kernel serves a layer of N images of H x W outputs (for profilers and benchmarks; the other arguments as aesr_conv2d_wino_supported):
 * 0 = not supported, 1 = conv_wino_f32 (16-channel chunks streamed through LDS, csrc/conv_wino.hip), 2 = conv_wino_res_f32 (the
 * transformed filter stays resident in LDS, independent waves, csrc/conv_wino_res.hip: K-side channels <= 32, or <= 64 where
 * 8 x 8-output blocks tile the image with <= 10 % padding; AESR_WINO_RES=0 disables it, =1 keeps it to <= 32 channels),
 * 3 = conv_wino_ring_f32 (filter chunks through a three-slot LDS ring, independent waves with per-wave patches,
 * csrc/conv_wino_ring.hip: all the other layers; AESR_WINO_RING=1: only where the launcher's cost estimate for it is below kernel 1's
 * -- layers whose image 8 x 8-output blocks tile well and that fill at least a round of work items -- =0: never).  The
 * LeakyReLU fused into these kernels is max(x, slope * x): slope must lie in [0, 1] (AESR_ERR_UNSUPPORTED otherwise). */
int aesr_conv2d_wino_kernel(int N, int H, int W, int Cin, int Cout, int KS, int pad, int transpose);
/* Watchdog of conv_wino_ring_f32's LDS arrival counters: how many waits gave up since the library was loaded (a wait that long
 * means a protocol bug; the results of that launch are garbage).  0 in correct operation; synchronises the device. */
unsigned int aesr_conv2d_wino_ring_timeouts(void);
size_t aesr_conv2d_wino_packed_floats(int Cout, int Cin, int transpose);
int aesr_conv2d_wino_pack_many(const aesr_pack_job* jobs_host, int njobs, void* stream);
int aesr_conv2d_wino_fwd(const float* in, const float* upacked, const float* bias, float* out, int N, int H, int W, int Cin,
                         int Cout, int act, float slope, void* stream);
int aesr_conv2d_wino_dgrad(const float* dy, const float* upacked_t, const float* x_saved, float* dx, int N, int H, int W, int Cin,
                           int Cout, int mask_act, float slope, void* stream);
/* Inference: conv + activation + EVAL-mode BatchNorm [+ AvgPool2d(2)] in one launch (networks/acai_vanilla.py:55-59,68-70,87-92 under
 * model.eval()): out = bn_scale[co] * [mean of each 2x2 window of] act(conv(in) + bias) + bn_shift[co]; pool: out is [N][H/2][W/2][Cout].
 * bn_scale / bn_shift: [Cout] device arrays (gamma / sqrt(running_var + eps), beta - running_mean * scale: aesr_bn_finalize with train = 0).
 * Bitwise what aesr_conv2d_wino_fwd followed by aesr_bn_apply gives.  Layers the resident-filter kernel serves only
 * (aesr_conv2d_wino_fwd_bn_supported; the others keep the two launches). */
int aesr_conv2d_wino_fwd_bn_supported(int N, int H, int W, int Cin, int Cout);
int aesr_conv2d_wino_fwd_bn(const float* in, const float* upacked, const float* bias, const float* bn_scale, const float* bn_shift, float* out, int N,
                            int H, int W, int Cin, int Cout, int act, float slope, int pool, void* stream);
/* The same two with a caller-owned workspace for the CHANNEL SPLIT of small layers with many K-side channels (a small shard of a
 * data-parallel step; the deep VGG layers of lpips/pretrained_networks.py:107-116 at any batch): a layer with fewer work items than
 * the chip has SIMDs streams its K side serially per item (512 channels = 32 chunks of ~3 us), so conv_wino_ring_f32 splits the
 * chunks over up to 16 items per (block group, cout tile), each writing raw partial sums into its own output-shaped slab of the
 * workspace, and wino_split_reduce_kernel adds the slabs in fixed order (+ bias, activation, derivative mask).
 * aesr_conv2d_wino_workspace_floats: the floats the library wants for a layer (0: no split; transpose = 1 for the data gradient).
 * A NULL or smaller workspace is legal: the layer then runs unsplit, exactly as through the functions above. */
size_t aesr_conv2d_wino_workspace_floats(int N, int H, int W, int Cin, int Cout, int transpose);
int aesr_conv2d_wino_fwd_ws(const float* in, const float* upacked, const float* bias, float* out, float* workspace, size_t workspace_floats,
                            int N, int H, int W, int Cin, int Cout, int act, float slope, void* stream);
int aesr_conv2d_wino_dgrad_ws(const float* dy, const float* upacked_t, const float* x_saved, float* dx, float* workspace, size_t workspace_floats,
                              int N, int H, int W, int Cin, int Cout, int mask_act, float slope, void* stream);

/* nearest-neighbour Upsample(x2) in front of a 3x3 convolution (Decoder: networks/acai_vanilla.py:92-96) folded into the Winograd
 * kernels: H, W are the convolution's (= the upsampled, even) size; `in_half` / `x_half` / `dx_half` are [N,H/2,W/2,C] tensors.  The
 * forward and the weight gradient read pixel (y, x) at (y/2, x/2) of the half-resolution tensor (the upsampled tensor never
 * exists); the data gradient stores the sum of every 2x2 block (the adjoint of the upsampling) -- no mask, the upsampled tensor
 * is not an activation output.  aesr_conv2d_wgrad_up2 needs aesr_conv2d_wgrad_up2_supported (Cin, Cout multiples of 32); its
 * workspace is that of aesr_conv2d_wgrad for the same N, H, W, Cin, Cout, KS = 3, pad = 1. */
int aesr_conv2d_wino_fwd_up2(const float* in_half, const float* upacked, const float* bias, float* out, int N, int H, int W, int Cin,
                             int Cout, int act, float slope, void* stream);
int aesr_conv2d_wino_dgrad_sum2(const float* dy, const float* upacked_t, float* dx_half, int N, int H, int W, int Cin, int Cout,
                                void* stream);
int aesr_conv2d_wgrad_up2_supported(int Cin, int Cout);
int aesr_conv2d_wgrad_up2(const float* x_half, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W, int Cin,
                          int Cout, void* stream);

/* Weight gradients of a whole backward pass with ONE reduction launch: aesr_conv2d_wgrad_partial writes only the partial slabs of
 * a layer into its workspace (same planner, same workspace size as aesr_conv2d_wgrad; x_up2 as aesr_conv2d_wgrad_up2), and
 * aesr_conv2d_wgrad_reduce_many sums the slabs of up to 16 layers per launch in the fixed order of aesr_conv2d_wgrad (bitwise the
 * same results).  The job array is read on the host during the call. */
typedef struct aesr_wgrad_reduce_job {
    const float* workspace;
    float* dw;
    float* db;       /* may be NULL */
    int N, H, W, Cin, Cout, KS, pad;
} aesr_wgrad_reduce_job;
int aesr_conv2d_wgrad_partial(const float* x, const float* dy, float* workspace, int N, int H, int W, int Cin, int Cout, int KS,
                              int pad, int x_up2, void* stream);
int aesr_conv2d_wgrad_reduce_many(const aesr_wgrad_reduce_job* jobs_host, int njobs, void* stream);

/* ---- data-parallel collectives: an RCCL communicator owned by the library (one process per GPU; new functionality -- the
 * reference's only multi-GPU code moves the loss to 'cuda:1', kwatsch/trainer_ae.py:43-44,84-86) --------------------------------
 * Collectives are plain enqueues on the caller's stream (capturable into a HIP graph; no watchdog thread, unlike
 * torch.distributed's ProcessGroupNCCL).  Bootstrap: rank 0 calls aesr_comm_unique_id and hands the 128 bytes to every rank
 * over any host channel (the shipped host code uses torch.distributed's gloo/TCP store); every rank then calls aesr_comm_init
 * with its device current (hipSetDevice / torch.cuda.set_device).  librccl is bound at run time: without it these entries
 * return AESR_ERR_UNSUPPORTED with a message.  dtype: 0 = f32, 1 = f64.  op: 0 = sum, 1 = max.  All buffers are device
 * pointers, reduced / broadcast IN PLACE. */
#define AESR_COMM_ID_BYTES 128
#define AESR_COMM_F32 0
#define AESR_COMM_F64 1
#define AESR_COMM_SUM 0
#define AESR_COMM_MAX 1
int aesr_comm_rccl_version(int* version_out);                      /* NCCL_VERSION_CODE of the bound librccl */
int aesr_comm_unique_id(void* id_host128);
int aesr_comm_init(const void* id_host128, int nranks, int rank, void** comm_out);
int aesr_comm_destroy(void* comm);
int aesr_comm_abort(void* comm);                                  /* tear down without waiting for pending collectives */
/* Flat gradient buffer (kwatsch/trainer_ae.py:93-96's backward/step pair, made data parallel) and SyncBN partial sums. */
int aesr_comm_allreduce(void* comm, void* buf, size_t count, int dtype, int op, void* stream);
/* Several buffers as ONE fused RCCL group launch (ncclGroupStart/End): the pointer / count arrays are host arrays. */
int aesr_comm_allreduce_many(void* comm, void* const* bufs_host, const size_t* counts_host, int nbufs, int dtype, int op,
                             void* stream);
int aesr_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AESR_HIP_H */
