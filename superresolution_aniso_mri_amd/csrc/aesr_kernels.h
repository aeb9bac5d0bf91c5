// Argument structs + host launchers of the aesr HIP kernels (shared by the kernel translation units and
// the C-ABI layer aesr_api.hip).
#pragma once
#include "aesr_common.h"

struct IgemmArgs {
    const float* in; const float* wpk; const float* bias; const float* ysave; float* out;
    int N, H, W, Cin, CinP, Cout, CoutP, Ho, Wo, pad;
    int TI, TH, TW, tiles_y, tiles_x, nitems, dbg;
    int NT;          // threads per workgroup: 512 or 256
    int ksplit;      // K-split slices (1 = off): raw partial sums go to slab ks of `out`, see conv_igemm.hip
    float* dbgbuf;   // debug stamps (nullptr in normal operation)
    int act, mask_act;
    float slope;
};
int aesr_launch_conv_igemm(const IgemmArgs& a, int KS, int NB, int MBW, hipStream_t st);
int aesr_launch_conv_ksplit_fixup(const float* partial, const float* bias, const float* ysave, float* out, size_t nelem, int Cout,
                                  int ksplit, int act, int mask_act, float slope, hipStream_t st);
int aesr_launch_pack_weights(const float* w, float* p, int Cout, int Cin, int KS, int KinP, int NoutP, int TN, int transpose, hipStream_t st);

// Winograd F(2x2,3x3) convolution (conv_wino.hip): 3x3, stride 1, padding 1
struct WinoArgs {
    const float* in; const float* upk; const float* bias; const float* ysave; float* out;
    int N, H, W, Cin, CinP, Cout, CoutP;
    int TI, THt, TWt;            // work item = TI images x THt x TWt Winograd tiles (2x2 outputs each) x 32 output channels
    int regs_y, regs_x, nitems;  // filled by the launcher
    unsigned m_ncot, m_regs_x, m_regs_y;     // ceil(2^32 / d) of the item decomposition's divisors (0: d == 1), filled by the launcher
    int act, mask_act;
    float slope;
    float* dbgbuf;               // per-wave phase stamps (AESR_WINO_DBG=1), nullptr in normal operation
    int flags;                   // experiment switches (AESR_WINO_FLAGS)
    int in_up2;                  // the input is stored at half resolution: pixel (y, x) reads (y/2, x/2) (nearest Upsample x2 folded in)
    int out_sum2;                // store the sum of every 2x2 output tile at half resolution (adjoint of that Upsample)
    int bpi, nblk;               // resident-filter kernel (conv_wino_res.hip): 8x8-output blocks per image / in all, filled by its launcher
    unsigned m_bpi;              // ceil(2^32 / bpi) (0: bpi == 1)
    // ring kernel (conv_wino_ring.hip), filled by its launcher: waves per workgroup, patch row pitch / image stride / buffer size in
    // floats, bank swizzle of the channel quads f(px) = (((px >> sw_a) & sw_m) << sw_b) & 3
    int nw, rowP, imgP, patch_fl, sw_a, sw_m, sw_b;
    int xcd_map;                 // resident-filter kernel: XCD-aware workgroup -> block map (set by its launcher)
    double plan_cost;            // cost estimate (cycles) of the first streamed kernel's plan for this layer: the ring kernel runs where its own is lower
    int nfull, tail_k;           // item list: groups of 8 blocks, then the last partial round in groups of tail_k <= 4 blocks
    // ring kernel, channel split (small layers with many K-side channels): ksplit workgroup items per (block group, cout tile), each
    // streaming kchunks 16-channel chunks into its own output-shaped slab of the workspace, split_bytes apart; summed by the split-reduce kernel
    int ksplit, kchunks, split_bytes;
    unsigned m_ksplit;           // ceil(2^32 / ksplit) (0: ksplit == 1)
    float* ws;                   // caller's workspace for the slabs (nullptr: no channel split)
    size_t ws_floats;
    // eval-mode BatchNorm (+ AvgPool2d(2)) behind the activation, folded into the epilogue (resident-filter kernel, forward only):
    // out = post_scale[co] * [mean of the 2x2 tile of] act(conv + bias) + post_shift[co]; post_pool: the output is [N][H/2][W/2][Cout]
    const float* post_scale; const float* post_shift;
    int post_pool;
};
int aesr_launch_conv_wino(const WinoArgs& a, hipStream_t st);
bool aesr_wino_res_ok(const WinoArgs& a);                        // few input channels: the filter stays resident, waves run on their own
int aesr_launch_conv_wino_res(const WinoArgs& a, hipStream_t st);
int aesr_wino_ring_mode();                                       // AESR_WINO_RING: 0 never, 1 where its cost estimate is lower, 2 (default) always
bool aesr_wino_ring_takes(const WinoArgs& a);                    // many K-side channels: filter chunks through an LDS ring, independent waves
int aesr_launch_conv_wino_ring(const WinoArgs& a, hipStream_t st);
size_t aesr_wino_ring_workspace_floats(const WinoArgs& a);      // floats of workspace the ring kernel's channel split wants for this layer (0: none)
unsigned aesr_wino_ring_timeouts();                              // protocol watchdog of the ring kernel (0 unless a spin gave up)
size_t aesr_wino_lds_bytes(int patch_pixels);

#define PACK_MAX_JOBS 32
struct PackJob { const float* w; float* p; int Cout, Cin, KS, KinP, NoutP, TN, transpose, block0; };
struct PackTable { int njobs, nblocks; PackJob job[PACK_MAX_JOBS]; };
int aesr_launch_pack_many(const PackTable& t, hipStream_t st);
// all parameter-side preparation of a step in one launch (prep.hip)
enum { PREP_PACK = 0, PREP_WINO_PACK = 1, PREP_STEM_FOLD = 2, PREP_COUT1_FLIP = 3 };
struct PrepJob { const float* w; const float* aux0; const float* aux1; float* out; int kind, Cout, Cin, KS, KinP, NoutP, TN, transpose, block0; };
struct PrepTable { int njobs, nblocks; PrepJob job[PACK_MAX_JOBS]; };
int aesr_launch_prep_many(const PrepTable& t, hipStream_t st);
int aesr_launch_wino_pack_many(const PackTable& t, hipStream_t st);

struct WgradArgs {
    const float* x; const float* dy; float* slab;
    int N, H, W, Cin, CinP, Cout, CoutP, Ho, Wo, pad;
    int TH, TW, tiles_y, tiles_x, ntiles;
    int S;
    int PWS, TWS, PSX, PSD;   // LDS row / plane strides in floats (even; planes = 4 mod 64)
    float* dbgbuf;            // debug phase stamps (AESR_WGRAD_DBG=1), nullptr in normal operation
    int x_up2;                // Winograd kernel only: x is stored at half resolution (nearest Upsample x2 folded into the loader)
};
int aesr_launch_conv_wgrad(const WgradArgs& a, int KS, int variant, hipStream_t st);
int aesr_launch_conv_wgrad_wino(const WgradArgs& a, hipStream_t st);
size_t aesr_wgrad_wino_lds_bytes(int TH, int TW);
bool aesr_wgrad_wino_tile_ok(int TH, int TW);
#define REDUCE_MAX_JOBS 16
struct ReduceJob { const float* slab; float* dw; float* db; int nslab, KS2, Cin, CinP, Cout, CoutP, block0; };
struct ReduceTable { int njobs, nblocks; ReduceJob job[REDUCE_MAX_JOBS]; };
int aesr_launch_wgrad_reduce_many(const ReduceTable& t, hipStream_t st);
int aesr_launch_wgrad_reduce(const float* slab, float* dw, float* db, int nslab, int KS, int Cin, int CinP, int Cout, int CoutP, hipStream_t st);

struct SmallArgs {
    const float* in; const float* w; const float* bias; const float* ysave; float* out;
    int N, H, W, Cin, Cout, Ho, Wo, KS, pad;
    int act, mask_act;
    float slope;
    int transpose, bcast;
    float ca[4], cb[4];
};
struct SmallDgradArgs {
    const float* dy; const float* w; float* dx;
    int N, H, W, Cin, Cout, Ho, Wo, KS, pad, bcast;
    float ca[4];
};
struct SmallWgradArgs {
    const float* in; const float* dout; float* partial;
    int N, H, W, Cin, Cout, Ho, Wo, pad;
};
struct Cout1WgradArgs {
    const float* x; const float* dy; float* partial;
    int N, H, W, Cin, TH, TW, tiles_y, tiles_x, ntiles;
};
struct Cout1FwdArgs {
    const float* x; const float* w; const float* bias; float* out;
    int N, H, W, Cin, TH, TW, tiles_y, tiles_x, act;
    float slope;
};
int aesr_launch_cout1_fwd(const Cout1FwdArgs& a, hipStream_t st);
int aesr_launch_smallcin_fwd(const SmallArgs& a, hipStream_t st);
int aesr_launch_smallcin_dgrad(const SmallDgradArgs& a, hipStream_t st);
int aesr_launch_sum_partials(const float* partial, int np, int n, float* out0, int n0, float* out1, hipStream_t st);
int aesr_launch_smallcin_wgrad(const SmallWgradArgs& a, int nwg, hipStream_t st);
int aesr_launch_cout1_wgrad(const Cout1WgradArgs& a, int nwg, hipStream_t st);

// "thin" 3x3 convolutions (one side single-channel): conv_thin.hip
struct ThinArgs {
    const float* s;        // single-channel image [N,Hs,Ws]
    const float* w;        // expand: [9][C] tap-major filter
    const float* be;       // expand: [9][C] per-tap bias counted for taps inside the grid, or nullptr
    const float* b;        // expand: [C] bias or nullptr
    const float* ysave;    // expand: saved activation output [N,Ho,Wo,C] for the derivative mask, or nullptr
    float* out;            // expand: [N,Ho,Wo,C]
    const float* t;        // reduce: [N,Ho,Wo,C]
    float* partial;        // reduce: [nwg][nrow][C]
    int N, Hs, Ws, Ho, Wo, C, ps;
    int act, mask_act;
    float slope;
    int with_be;
    int tiles_y, tiles_x, ntiles;
};
int aesr_launch_thin_expand(ThinArgs a, hipStream_t st);
int aesr_launch_thin_reduce(ThinArgs a, int nwg, hipStream_t st);
int aesr_launch_thin_collapse(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int C, int act,
                              float slope, hipStream_t st);
int aesr_launch_thin_stem_fold(const float* ws, const float* bs, const float* w1, float* folded, int Cs, int C1, hipStream_t st);
int aesr_launch_thin_stem_finish(const float* R, const float* ws, const float* bs, const float* w1, float* dws, float* dbs,
                                 float* dw1, float* db1, int Cs, int C1, hipStream_t st);
int aesr_launch_thin_cout1_flip(const float* w, float* wexp, int Cin, hipStream_t st);
int aesr_launch_thin_cout1_finish(const float* R, float* dw, float* db, int Cin, hipStream_t st);

struct BnGroups { int G; int nstart[5]; };
struct BnApplyArgs {
    const float* y; const float* scale; const float* shift; float* out;
    int N, H, W, C, Ho, Wo, mode;
    BnGroups gr;
};
struct BnBwdArgs {
    const float* gout; const float* y; const float* mean; const float* invstd; const float* scale; const float* coef;
    float* partial; float* dpre;
    int N, H, W, C, Ho, Wo, mode, act;
    float slope;
    BnGroups gr;
    unsigned long long m_hw, m_w;   // exact-division constants for H*W and W (filled by the bn_bwd_reduce launcher)
    int k_hw, k_w;
};
// one-launch BatchNorm for small batches (bn_fused.hip): activations resident in LDS, one grid barrier
struct BnFusedArgs;
struct BnP2P { int world, rank, slot; const unsigned* gen; void* peers[8]; };     // data parallel: peer-mapped exchange regions (p2p.hip)
int aesr_launch_p2p_tick(unsigned* gen, hipStream_t st);
bool aesr_bn_fused1_ok(int N, int H, int W, int C, int pool, int G, int backward);
unsigned aesr_bn_fused_timeouts_impl();
int aesr_bn_fused_run(const float* y, const float* gout, float* out, float* rec, unsigned* bar, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, long long* nbt, float* mean, float* invstd, float* scale, float* shift, float* coef,
                      float* dgamma, float* dbeta, int N, int H, int W, int C, int pool, int G, const int* nstart, const double* counts,
                      float momentum, float eps, int update_running, int act, float slope, int backward, const BnP2P* p2p, hipStream_t st);
int aesr_launch_bn_stats(const float* y, float* partial, int HW, int C, const BnGroups& gr, int nwg, hipStream_t st);
int aesr_launch_bn_reduce(const float* partial, double* sums, int nwg, int C, int G, hipStream_t st);
int aesr_launch_bn_finalize(const double* sums, const double* counts, const float* gamma, const float* beta, float* rm, float* rv,
                            long long* nbt, float* mean, float* invstd, float* scale, float* shift, int C, int G, float momentum,
                            float eps, int train, int update_running, hipStream_t st);
int aesr_launch_bn_reduce_finalize(const float* partial, int nwg, const double* counts, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, long long* nbt, float* mean, float* invstd,
                                   float* scale, float* shift, int C, int G, float momentum, float eps, int update_running,
                                   hipStream_t st);
int aesr_launch_bn_bwd_reduce_finalize(const float* partial, int nwg, const double* counts, float* coef, float* dgamma,
                                       float* dbeta, int C, int G, hipStream_t st);
int aesr_launch_bn_apply(const BnApplyArgs& a, hipStream_t st);
int aesr_launch_bn_bwd_reduce(const BnBwdArgs& a, int nwg, hipStream_t st);
int aesr_launch_bn_bwd_finalize(const double* sums, const double* counts, float* coef, float* dgamma, float* dbeta, int C, int G, hipStream_t st);
int aesr_launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t st);
// data parallel: finalize (from the all-reduced sums) + apply in one launch, forward and backward
bool aesr_bn_fused_ok(int C, int G);
int aesr_launch_bn_finalize_apply(const double* sums, const double* counts, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, long long* nbt, float* mean, float* invstd, float* scale, float* shift, float momentum,
                                  float eps, int update_running, int G, const BnApplyArgs& a, hipStream_t st);
int aesr_launch_bn_bwd_finalize_apply(const double* sums, const double* counts, float* coef, float* dgamma, float* dbeta, int G,
                                      const BnBwdArgs& a, hipStream_t st);

int aesr_launch_lerp_fwd(const float* z, const float* af, const float* at, float* zmix, int B, size_t per, hipStream_t st);
int aesr_launch_lerp_bwd(const float* dmix, const float* af, const float* at, float* dz, int B, size_t per, hipStream_t st);
int aesr_launch_lerp_cat_fwd(const float* z, const float* af, const float* at, float* zcat, int B, size_t per, hipStream_t st);
int aesr_launch_interleave_clamp(const float* orig, const float* synth, float* out, int Z, int n, size_t per, float lo, float hi, hipStream_t st);
int aesr_launch_lerp_multi(const float* z, float* out, int Z, size_t per, const float* alphas, int n, float nslope, hipStream_t st);
int aesr_launch_lerp_cat_bwd(const float* g, const float* af, const float* at, float* dz, int B, size_t per, hipStream_t st);
int aesr_launch_mse_fwd(const float* a, const float* b, double* partial, int np, float* out, size_t n, hipStream_t st);
int aesr_launch_mse3_fwd(const float* const* a, const float* const* b, const size_t* n, const float* lam, double* ws, float* out, hipStream_t st);
int aesr_launch_mse3_bwd(const float* a1, const float* b1, size_t n1, const float* a2, const float* b2, size_t n2, const float* lam,
                         const float* g, float* d1, float* d2, hipStream_t st);
int aesr_launch_mse_bwd(const float* a, const float* b, const float* g, float* da, size_t n, hipStream_t st);
int aesr_launch_act_bwd(const float* dout, const float* y, float* dpre, size_t n, int act, float slope, hipStream_t st);
int aesr_launch_adam(float* p, float* g, float* m, float* v, float* state, size_t n, float lr, double beta1, double beta2,
                     float eps, float wd, int zero_g, hipStream_t st);


int aesr_launch_maxpool2_fwd(const float* x, float* out, int N, int H, int W, int C, hipStream_t st);
int aesr_launch_maxpool2_bwd(const float* gout, const float* x, const float* gadd, float* dx, int N, int H, int W, int C, int relu, hipStream_t st);
int aesr_launch_lpips_tap_fwd(const float* f, const float* lin, float* partial, int B, int HW, int C, hipStream_t st);
int aesr_launch_lpips_tap_bwd(const float* f, const float* lin, const float* gd, float* gf0, int B, int HW, int C, hipStream_t st);
int aesr_launch_lpips_finalize(const float* const* partials, const int* hw, int ntaps, float* d, int B, hipStream_t st);
int aesr_launch_scale_expand(const float* x, float* out4, int n, const float* ca, const float* cb, int backward, hipStream_t st);
int aesr_launch_resample2(const float* x, const float* gout, const float* xsave, float* dst, int N, int H, int W, int C, int mode,
                          int backward, int mask_act, float slope, hipStream_t st);
size_t aesr_vif_workspace_bytes_impl(int Z, int H, int W);
int aesr_launch_vif_mscale(const float* ref, const float* dist, void* workspace, double* vif, int Z, int H, int W, const double* weights,
                           const int* radii, double sigma_nsq, hipStream_t st);
int aesr_launch_ssim_mse(const float* a, const float* b, double* partial, double* ssim, double* mse, int Z, int H, int W, int win,
                         double data_range, double k1, double k2, hipStream_t st);
#define TRIPLET_MAX 64
struct TripletDesc { long long vol_off; int H, W, z_from, z_to, z_between, oy, ox, k; float gain, cutoff; };
struct TripletTable { TripletDesc d[TRIPLET_MAX]; };
int aesr_launch_triplet_assemble(const float* vol, const TripletTable& t, int B, int W, float* image, float* between, hipStream_t st);
int aesr_launch_s2d(const float* x, float* out, int N, int H, int W, int C, int inverse, hipStream_t st);

// ---- Laplacian-pyramid loss (lap.hip) ----
int aesr_launch_lap_blur5(const float* in, const float* add, float* out, int P, int H, int W, float gain, int adjoint, hipStream_t st);
int aesr_launch_lap_down2(const float* in, float* out, int P, int H, int W, hipStream_t st);
int aesr_launch_lap_zero_insert2(const float* in, float* out, int P, int h, int w, int H, int W, hipStream_t st);
int aesr_launch_l1_fwd(const float* a, const float* b, double* partial, int np, float* out, size_t n, hipStream_t st);
int aesr_launch_l1_bwd(const float* a, const float* b, const float* g, float* da, size_t n, hipStream_t st);
int aesr_launch_row_mean_fwd(const float* x, float* out, int N, size_t M, hipStream_t st);
int aesr_launch_row_mean_bwd(const float* g, float* dx, int N, size_t M, hipStream_t st);
