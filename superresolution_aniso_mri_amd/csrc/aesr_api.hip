// extern "C" entry points of libaesr_hip.so (declared in include/aesr_hip.h) + the host-side tile planners.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/aesr_hip.h"
#include "aesr_kernels.h"

// ---- error string ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void aesr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- tile planning ------------------------------------------------------------------------------------------
static void cout_padding(int Cout, int* CoutP, int* NB) {
    if (Cout <= 16) { *CoutP = 16; *NB = 1; }
    else if (Cout <= 32) { *CoutP = 32; *NB = 2; }
    else { *CoutP = round_up(Cout, 64); *NB = 4; }
}

// Every convolution entry point and query decides FIRST, in 64 bits, whether the tensors stay inside the kernels' 32-bit element offsets (the
// launchers refuse 0x1C000000 elements and more): the planners below do their tile arithmetic in int and must never see sizes beyond that
// (found by the host-side sanitizer sweep of round 6, tests/test_host_sanitized.py: ceil_div(INT_MAX, 2) in plan_wino from a query).
static bool conv_dims_ok(int N, int H, int W, int Cin, int Cout) {
    if (N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return false;
    if (N > (1 << 24) || H > (1 << 15) || W > (1 << 15) || Cin > (1 << 15) || Cout > (1 << 15)) return false;
    const unsigned long long px = (unsigned long long)N * (unsigned)(H + 2) * (unsigned)(W + 2);       // <= 2^24 * 2^16 * 2^16: no overflow
    return px * (unsigned long long)(Cin > Cout ? Cin : Cout) < 0x1C000000ull;
}
#define AESR_CHECK_DIMS(who, N, H, W, Cin, Cout)                                                                                            \
    do {                                                                                                                                    \
        if (!conv_dims_ok(N, H, W, Cin, Cout)) {                                                                                            \
            aesr_set_error("%s: %d x %d x %d with %d -> %d channels is empty or beyond the kernels' 32-bit element offsets (469M elements)", who, N, H, \
                           W, Cin, Cout);                                                                                                   \
            return AESR_ERR_UNSUPPORTED;                                                                                                    \
        }                                                                                                                                   \
    } while (0)

struct ConvPlan { int TI, TH, TW, NB, MBW, CinP, CoutP, NT, ksplit; };
static std::mutex g_plan_mu;

static std::map<std::tuple<int, int, int, int, int, int>, ConvPlan> g_conv_plans;

// Pick the workgroup shape (512 threads = one workgroup per CU, or 256 = two per CU) and the output tile (TI images x TH x
// TW) that minimise estimated MFMA time: idle M-block slots at tile edges (the network's sizes are 162, 81, 40 ...) and the
// work-item quantisation over the CUs.
static ConvPlan plan_conv(int N, int Ho, int Wo, int Cin, int Cout, int KS) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    const auto key = std::make_tuple(N, Ho, Wo, Cin, Cout, KS);
    auto it = g_conv_plans.find(key);
    if (it != g_conv_plans.end()) return it->second;
    ConvPlan p;
    cout_padding(Cout, &p.CoutP, &p.NB);
    p.CinP = round_up(Cin, 16);
    p.MBW = (p.NB == 4) ? 2 : 4;                     // M-blocks per wave
    const int ncout = p.CoutP / (16 * p.NB);
    int force_nt = 0;
    if (const char* e = getenv("AESR_IGEMM_NT")) force_nt = atoi(e);
    double best = 1e300;
    p.TI = 1; p.TH = 1; p.TW = 1; p.NT = 512;
    for (int NT = 512; NT >= 256; NT -= 256) {
        // two 4-wave workgroups per CU measured 2-5 % slower than one 8-wave workgroup on all but the 32->32 160x160 layer
        // (profiles/r01_igemm_nt256.txt): kept as an experiment knob, not chosen by the planner
        (void)force_nt;
        if (NT != 512) continue;
        const int NW = NT / 64, wgs_per_cu = 512 / NT;
        const int maxpix = 16 * NW * p.MBW;
        // LDS per workgroup: patch (80 B per pixel) + the chunk's weights + bias; 6 staging pieces of 16 B per thread
        const int lds_budget = 160 * 1024 / wgs_per_cu - (KS * KS * 4 * 16 * p.NB * 16 + p.CoutP * 4);
        int maxpatch = lds_budget / 80;
        if (maxpatch > NT * 6 / 4) maxpatch = NT * 6 / 4;
        if (NT == 512 && maxpatch > 760) maxpatch = 760;
        auto consider = [&](int TI, int TH, int TW) {
            const int PP = TI * (TH + KS - 1) * (TW + KS - 1);
            const int TP = TI * TH * TW;
            if (TP > maxpix || PP > maxpatch) return;
            const int nblk = ceil_div(TP, 16);
            const long nwg = (long)ceil_div(N, TI) * ceil_div(Ho, TH) * ceil_div(Wo, TW) * ncout;
            // per CU and 16-channel chunk: the MFMA time of a SIMD (2 waves x blocks x NB x taps x 4 k-steps x 32 cycles)
            // plus the staging / barrier phases in which the matrix pipe idles (phase stamps: ~4.5k cycles with one
            // workgroup per CU)
            const double nch = p.CinP / 16;
            const double per = nch * (ceil_div(nblk, NW) * 2.0 * p.NB * KS * KS * 4 * 32 + 4500.0) + 3000.0;
            const double slots = 256.0 * wgs_per_cu;
            const double rounds = nwg <= 8 * slots ? (double)ceil_div((int)nwg, (int)slots) : (double)nwg / slots;
            const double t = per * rounds;
            if (t < best * 0.999 || (t < best * 1.001 && NT == p.NT && TP > p.TI * p.TH * p.TW)) {
                if (t < best) best = t;
                p.TI = TI; p.TH = TH; p.TW = TW; p.NT = NT;
            }
        };
        if (Ho * Wo <= maxpix && (Ho + KS - 1) * (Wo + KS - 1) <= maxpatch) {
            for (int TI = 1; TI <= N && TI * Ho * Wo <= maxpix; ++TI) consider(TI, Ho, Wo);
        }
        for (int TH = 1; TH <= Ho && TH <= 64; ++TH)
            for (int TW = 1; TW <= Wo && TW <= 64; ++TW) consider(1, TH, TW);
    }
    if (const char* e = getenv("AESR_IGEMM_TILE")) {          // experiments: force "TI,TH,TW"
        int ti, th, tw;
        if (sscanf(e, "%d,%d,%d", &ti, &th, &tw) == 3) { p.TI = ti; p.TH = th < Ho ? th : Ho; p.TW = tw < Wo ? tw : Wo; }
    }
    // K-split for layers whose work items under-fill the 256 persistent workgroups (VGG conv4/5 at 20x20 / 10x10): slices of
    // the input channels become extra work items that write raw partial sums, a fix-up pass adds them (+bias, activation,
    // mask).  Used only when the caller passes a workspace (aesr_conv2d_fwd_ws / _dgrad_ws).
    p.ksplit = 1;
    {
        const long items = (long)ceil_div(N, p.TI) * ceil_div(Ho, p.TH) * ceil_div(Wo, p.TW) * ncout;
        const int nch = p.CinP / 16, nblk = ceil_div(p.TI * p.TH * p.TW, 16);
        const double chunk = ceil_div(nblk, 8) * 2.0 * p.NB * KS * KS * 4 * 32 + 4500.0;
        const double out_bytes = (double)N * Ho * Wo * Cout * 4;
        double bestt = 1e300;
        for (int ks = 1; ks <= 4; ++ks) {
            if (nch % ks != 0 || nch / ks < 4 || (Cout & 3)) continue;
            if (ks > 1 && items * ks > 4 * 256) break;
            const double rounds = (double)ceil_div((int)(items * ks), 256);
            double t = rounds * ((nch / ks) * chunk + 3000.0);
            if (ks > 1) t += (ks + 2) * out_bytes / 2000.0 + 12000.0;        // fix-up traffic at ~4 TB/s + a launch
            if (t < bestt * 0.97) { bestt = t; p.ksplit = ks; }
        }
        if (const char* e = getenv("AESR_IGEMM_KSPLIT")) { const int k = atoi(e); if (k >= 1 && nch % k == 0) p.ksplit = k; }
    }
    if (getenv("AESR_PLAN_DEBUG"))
        fprintf(stderr, "[aesr plan] conv N=%d %dx%d Cin=%d Cout=%d KS=%d -> NT=%d TI=%d TH=%d TW=%d NB=%d nblk=%d items=%ld ksplit=%d\n", N, Ho, Wo,
                Cin, Cout, KS, p.NT, p.TI, p.TH, p.TW, p.NB, ceil_div(p.TI * p.TH * p.TW, 16),
                (long)ceil_div(N, p.TI) * ceil_div(Ho, p.TH) * ceil_div(Wo, p.TW) * ncout, p.ksplit);
    g_conv_plans[key] = p;
    return p;
}

// Winograd F(2x2,3x3) work-item shape: TI images x THt x TWt tiles (<= 128 tiles = 8 waves x 16), patch within LDS (two buffers)
// and within 5 staging pieces per thread.  Cost = work-item rounds over the 256 CUs x chunk time; a chunk costs one or two
// wave-passes per SIMD (waves whose 16 tiles are all invalid skip the arithmetic).
struct WinoPlan { int TI, THt, TWt, CinP, CoutP; double cost; };
static std::map<std::tuple<int, int, int, int, int>, WinoPlan> g_wino_plans;

static WinoPlan plan_wino(int N, int H, int W, int Cin, int Cout) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    const auto key = std::make_tuple(N, H, W, Cin, Cout);
    auto it = g_wino_plans.find(key);
    if (it != g_wino_plans.end()) return it->second;
    WinoPlan p;
    p.CinP = round_up(Cin, 16);
    p.CoutP = round_up(Cout, 32);
    const int Ht = ceil_div(H, 2), Wt = ceil_div(W, 2), ncot = p.CoutP / 32, nch = p.CinP / 16;
    p.TI = 1; p.THt = 1; p.TWt = 1; p.cost = 1e300;
    auto consider = [&](int TI, int THt, int TWt) {
        const int TP = TI * THt * TWt, PP = TI * (2 * THt + 2) * (2 * TWt + 2);
        if (TP > 128 || PP * 4 > 512 * 5 || aesr_wino_lds_bytes(PP) > (size_t)160 * 1024) return;
        const long items = (long)ceil_div(N, TI) * ceil_div(Ht, THt) * ceil_div(Wt, TWt) * ncot;
        const double passes = ceil_div(ceil_div(TP, 16), 4);
        const double per = nch * (passes * 128 * 32.0 + 1200.0) + 2500.0;
        const double rounds = items <= 8 * 256 ? (double)ceil_div((int)items, 256) : (double)items / 256.0;
        const double t = per * rounds;
        if (t < p.cost * 0.999 || (t < p.cost * 1.001 && TP > p.TI * p.THt * p.TWt)) {
            if (t < p.cost) p.cost = t;
            p.TI = TI; p.THt = THt; p.TWt = TWt;
        }
    };
    for (int TI = 1; TI <= N && TI * Ht * Wt <= 128; ++TI) consider(TI, Ht, Wt);
    for (int THt = 1; THt <= Ht && THt <= 64; ++THt)
        for (int TWt = 1; TWt <= Wt && TWt <= 64; ++TWt) consider(1, THt, TWt);
    if (const char* e = getenv("AESR_WINO_TILE")) {           // experiments: force "TI,THt,TWt"
        int ti, th, tw;
        if (sscanf(e, "%d,%d,%d", &ti, &th, &tw) == 3) { p.TI = ti; p.THt = th < Ht ? th : Ht; p.TWt = tw < Wt ? tw : Wt; }
    }
    if (getenv("AESR_PLAN_DEBUG"))
        fprintf(stderr, "[aesr plan] wino N=%d %dx%d Cin=%d Cout=%d -> TI=%d THt=%d TWt=%d (tiles %d, patch %d px) cost %.0f\n", N, H, W, Cin, Cout,
                p.TI, p.THt, p.TWt, p.TI * p.THt * p.TWt, p.TI * (2 * p.THt + 2) * (2 * p.TWt + 2), p.cost);
    g_wino_plans[key] = p;
    return p;
}

struct WgradPlan { int variant, COT, CinP, CoutP, TH, TW, S, nslab, PWS, TWS, PSX, PSD; size_t slab_floats; };
static std::map<std::tuple<int, int, int, int, int, int>, WgradPlan> g_wgrad_plans;

static int plane_stride(int n) { return round_up(n, 64) + 4; }    // = 4 (mod 64): 16 channel planes hit 16 distinct bank quads

// Winograd F(2x2,3x3) weight gradient (conv_wgrad_wino.hip, variant 2): 3x3 / padding 1 with both channel counts multiples of 32.
// One 4-wave workgroup per CU walks spatial tiles of TH x TW output pixels (TH even, TW a multiple of 8, within the register
// prefetch slots); S splits x (ci, co) chunks ~ 256 workgroups.  AESR_WGRAD_WINO=0 keeps the direct kernels.
static bool wgrad_wino_ok(int Cin, int Cout, int KS, int pad) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("AESR_WGRAD_WINO"); enabled = (e && atoi(e) == 0) ? 0 : 1; }
    return enabled && KS == 3 && pad == 1 && Cin % 32 == 0 && Cout % 32 == 0;
}

static WgradPlan plan_wgrad_wino(int N, int H, int W, int Cin, int Cout) {
    WgradPlan p;
    p.variant = 2;
    p.COT = 32;
    p.CinP = Cin;
    p.CoutP = Cout;
    const int nchunks = (Cin / 32) * (Cout / 32);
    int S = 256 / nchunks;
    if (S >= 8) S &= ~7;                       // multiple of 8: XCD-aware workgroup order
    if (S < 1) S = 1;
    double best = 1e300;
    p.TH = 16; p.TW = 8; p.S = S;
    for (int v = 0; v < 2; ++v) {
        const int TH = v ? 8 : 16, TW = v ? 16 : 8;                 // the kernel's two tiles (aesr_wgrad_wino_tile_ok): 8 k-steps, 2 per wave
        const int ntiles = N * ceil_div(H, TH) * ceil_div(W, TW);
        const int s = S < ntiles ? S : ntiles;
        // per visit: 128 MFMAs of a wave (4 096 cycles) + ~270 other instructions, which this chip does not overlap with them
        const double t = (double)ceil_div(ntiles, s) * (4096.0 + 1300.0);
        if (t < best) { best = t; p.TH = TH; p.TW = TW; p.S = s; }
    }
    if (const char* e = getenv("AESR_WGRAD_WINO_TILE")) {     // experiment knob: "TH,TW"
        int th = 0, tw = 0;
        if (sscanf(e, "%d,%d", &th, &tw) == 2 && aesr_wgrad_wino_tile_ok(th, tw)) {
            p.TH = th; p.TW = tw;
            const int ntiles = N * ceil_div(H, p.TH) * ceil_div(W, p.TW);
            p.S = S < ntiles ? S : ntiles;
        }
    }
    if (const char* e = getenv("AESR_WGRAD_WINO_S")) { const int s_ = atoi(e); if (s_ > 0) p.S = s_; }
    // experiment knob (round-4 verdict, next 3a): cap the slab count of every layer (fewer, longer-lived workgroups; fewer slabs to sum)
    if (const char* e = getenv("AESR_WGRAD_WINO_SMAX")) { const int s_ = atoi(e); if (s_ > 0 && p.S > s_) p.S = s_; }
    p.PWS = round_up(p.TW + 2, 4);             // LDS row strides in pixels (conv_wgrad_wino.hip)
    p.TWS = round_up(p.TW, 4);
    p.PSX = p.PSD = 0;
    p.nslab = p.S;
    p.slab_floats = (size_t)p.nslab * 10 * p.CinP * p.CoutP;
    if (getenv("AESR_PLAN_DEBUG"))
        fprintf(stderr, "[plan_wgrad] wino N=%d %dx%d %d->%d: tile %dx%d S=%d tiles=%d\n", N, H, W, Cin, Cout, p.TH, p.TW, p.S,
                N * ceil_div(H, p.TH) * ceil_div(W, p.TW));
    return p;
}

static WgradPlan plan_wgrad(int N, int Ho, int Wo, int Cin, int Cout, int KS, int pad = -1) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    const bool wino = wgrad_wino_ok(Cin, Cout, KS, pad);
    const auto key = std::make_tuple(N, Ho, Wo, Cin, Cout, wino ? -KS : KS);
    auto it = g_wgrad_plans.find(key);
    if (it != g_wgrad_plans.end()) return it->second;
    if (wino) {
        const WgradPlan pw = plan_wgrad_wino(N, Ho, Wo, Cin, Cout);
        g_wgrad_plans[key] = pw;
        return pw;
    }
    WgradPlan p;
    p.variant = Cout > 32 ? 1 : 0;
    p.COT = p.variant ? 64 : 32;
    const int cibw = p.variant ? 2 : 1;
    p.CinP = round_up(Cin, 32);
    p.CoutP = round_up(Cout, p.COT);
    const size_t max_lds = 52 * 1024;        // three workgroups per CU
    const int nchunks = (p.CinP / 32) * (p.CoutP / p.COT);
    const int S0 = round_up(768 / nchunks > 0 ? 768 / nchunks : 1, 8);      // multiple of 8: XCD-aware workgroup order
    double best = 1e300;
    p.TH = 1; p.TW = 8; p.S = 1;
    for (int TW = 8; TW <= 64; TW += 8) {
        if (TW - 8 >= Wo) break;
        for (int TH = 1; TH <= 32 && TH <= Ho; ++TH) {
            const int PH = TH + KS - 1, PWp = TW + KS - 1, PWS = PWp + (PWp & 1);
            // the kernel prefetches a whole tile into registers: WG_NX / WG_ND float4 slots per thread (conv_wgrad.hip)
            if (PH * PWp * 8 > 256 * (p.variant ? 4 : 6) || TH * TW * (p.COT / 4) > 256 * (p.variant ? 5 : 6)) break;
            const size_t ldsb = ((size_t)32 * plane_stride(PH * PWS) + (size_t)p.COT * plane_stride(TH * TW)) * 4;
            if (ldsb > max_lds) break;
            const int ntiles = N * ceil_div(Ho, TH) * ceil_div(Wo, TW);
            const int S = S0 < ntiles ? S0 : ntiles;
            const double rounds = (double)ceil_div(ntiles, S);
            // three workgroups share a SIMD's matrix pipe; per tile about 4.5k cycles of LDS-write phase, barriers and
            // address work are not hidden (fitted to the phase stamps of AESR_WGRAD_DBG on the layers of the AE)
            const double mf = 3.0 * (TH * TW / 8) * (2 * KS * KS * cibw) * 32.0;
            const double t = rounds * (mf + 4500.0);
            if (t < best) { best = t; p.TH = TH; p.TW = TW; p.S = S; }
        }
    }
    if (const char* e = getenv("AESR_WGRAD_TILE")) {          // experiment knob: "TH,TW"
        int th = 0, tw = 0;
        if (sscanf(e, "%d,%d", &th, &tw) == 2 && th > 0 && tw > 0 && tw % 8 == 0) {
            p.TH = th < Ho ? th : Ho; p.TW = tw;
            const int ntiles = N * ceil_div(Ho, p.TH) * ceil_div(Wo, p.TW);
            p.S = S0 < ntiles ? S0 : ntiles;
        }
    }
    if (const char* e = getenv("AESR_WGRAD_S")) {             // experiment knob: splits per (ci, co) chunk
        const int s_ = atoi(e);
        const int ntiles = N * ceil_div(Ho, p.TH) * ceil_div(Wo, p.TW);
        if (s_ > 0) p.S = s_ < ntiles ? s_ : ntiles;
    }
    p.PWS = p.TW + KS - 1 + ((p.TW + KS - 1) & 1);
    p.TWS = p.TW;
    p.PSX = plane_stride((p.TH + KS - 1) * p.PWS);
    p.PSD = plane_stride(p.TH * p.TW);
    p.nslab = p.S;
    if (getenv("AESR_PLAN_DEBUG"))
        fprintf(stderr, "[plan_wgrad] N=%d %dx%d %d->%d k%d: tile %dx%d S=%d tiles=%d rounds=%d lds=%zu\n", N, Ho, Wo, Cin, Cout, KS,
                p.TH, p.TW, p.S, N * ceil_div(Ho, p.TH) * ceil_div(Wo, p.TW), ceil_div(N * ceil_div(Ho, p.TH) * ceil_div(Wo, p.TW), p.S),
                ((size_t)32 * p.PSX + (size_t)p.COT * p.PSD) * 4);
    p.slab_floats = (size_t)p.nslab * (KS * KS + 1) * p.CinP * p.CoutP;
    g_wgrad_plans[key] = p;
    return p;
}

static bool fill_groups(BnGroups* gr, int G, const int* nstart_host) {
    if (G < 1 || G > 4 || !nstart_host) return false;
    gr->G = G;
    for (int i = 0; i <= G; ++i) gr->nstart[i] = nstart_host[i];
    for (int i = G + 1; i < 5; ++i) gr->nstart[i] = nstart_host[G];
    return true;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------
extern "C" {

int aesr_version(void) { return AESR_ABI_VERSION; }
const char* aesr_last_error_string(void) { return g_err; }

size_t aesr_conv2d_packed_floats(int Cout, int Cin, int KS, int transpose) {
    int NP, NB;
    const int kin = transpose ? Cout : Cin, nout = transpose ? Cin : Cout;
    cout_padding(nout, &NP, &NB);
    return (size_t)KS * KS * round_up(kin, 16) * NP;
}

int aesr_conv2d_pack(const float* w, float* packed, int Cout, int Cin, int KS, int transpose, void* stream) {
    AESR_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && (KS == 1 || KS == 3), "aesr_conv2d_pack: bad arguments");
    int NP, NB;
    const int kin = transpose ? Cout : Cin, nout = transpose ? Cin : Cout;
    cout_padding(nout, &NP, &NB);
    return aesr_launch_pack_weights(w, packed, Cout, Cin, KS, round_up(kin, 16), NP, 16 * NB, transpose, (hipStream_t)stream);
}

static int run_igemm(const float* in, const float* packed, const float* bias, const float* ysave, float* out, int N, int H,
                     int W, int Cin, int Cout, int KS, int pad, int act, int mask_act, float slope, float* workspace,
                     hipStream_t st) {
    AESR_CHECK_DIMS("aesr_conv2d (implicit GEMM)", N, H, W, Cin, Cout);
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    AESR_CHECK_ARG(Ho > 0 && Wo > 0, "aesr_conv2d: the %d x %d input is smaller than the %d x %d filter", H, W, KS, KS);
    const ConvPlan p = plan_conv(N, Ho, Wo, Cin, Cout, KS);
    IgemmArgs a;
    a.in = in; a.wpk = packed; a.bias = bias; a.ysave = ysave; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = p.CinP; a.Cout = Cout; a.CoutP = p.CoutP; a.Ho = Ho; a.Wo = Wo; a.pad = pad;
    a.TI = p.TI; a.TH = p.TH; a.TW = p.TW; a.tiles_y = ceil_div(Ho, p.TH); a.tiles_x = ceil_div(Wo, p.TW);
    a.act = act; a.mask_act = mask_act; a.slope = slope; a.dbgbuf = nullptr; a.NT = p.NT; a.ksplit = 1;
    if (workspace && p.ksplit > 1) {
        // raw partial sums of the k slices -> workspace, then the fix-up pass (bias, activation, derivative mask)
        a.ksplit = p.ksplit; a.out = workspace; a.bias = nullptr; a.ysave = nullptr; a.act = ACT_NONE; a.mask_act = ACT_NONE;
        if (int e = aesr_launch_conv_igemm(a, KS, p.NB, p.MBW, st)) return e;
        return aesr_launch_conv_ksplit_fixup(workspace, bias, ysave, out, (size_t)N * Ho * Wo * Cout, Cout, p.ksplit, act, mask_act,
                                             slope, st);
    }
    return aesr_launch_conv_igemm(a, KS, p.NB, p.MBW, st);
}

int aesr_conv2d_pack_many(const aesr_pack_job* jobs_host, int njobs, void* stream) {
    AESR_CHECK_ARG(jobs_host && njobs > 0, "aesr_conv2d_pack_many: no jobs");
    for (int j0 = 0; j0 < njobs; j0 += PACK_MAX_JOBS) {
        PackTable t;
        memset(&t, 0, sizeof(t));
        t.njobs = njobs - j0 < PACK_MAX_JOBS ? njobs - j0 : PACK_MAX_JOBS;
        int nb = 0;
        for (int k = 0; k < t.njobs; ++k) {
            const aesr_pack_job& jb = jobs_host[j0 + k];
            AESR_CHECK_ARG(jb.w && jb.packed && jb.Cout > 0 && jb.Cin > 0 && (jb.KS == 1 || jb.KS == 3),
                           "aesr_conv2d_pack_many: bad job %d", j0 + k);
            int NP, NB;
            const int kin = jb.transpose ? jb.Cout : jb.Cin, nout = jb.transpose ? jb.Cin : jb.Cout;
            cout_padding(nout, &NP, &NB);
            PackJob& o = t.job[k];
            o.w = jb.w; o.p = jb.packed; o.Cout = jb.Cout; o.Cin = jb.Cin; o.KS = jb.KS; o.KinP = round_up(kin, 16); o.NoutP = NP;
            o.TN = 16 * NB; o.transpose = jb.transpose; o.block0 = nb;
            const size_t total = (size_t)jb.KS * jb.KS * o.KinP * NP;
            int blocks = (int)((total + 1023) / 1024);         // 4 elements per thread
            if (blocks > 256) blocks = 256;
            nb += blocks;
        }
        t.nblocks = nb;
        if (int e = aesr_launch_pack_many(t, (hipStream_t)stream)) return e;
    }
    return AESR_OK;
}

int aesr_weight_prep_many(const aesr_prep_job* jobs_host, int njobs, void* stream) {
    AESR_CHECK_ARG(jobs_host && njobs > 0, "aesr_weight_prep_many: no jobs");
    for (int j0 = 0; j0 < njobs; j0 += PACK_MAX_JOBS) {
        PrepTable t;
        memset(&t, 0, sizeof(t));
        t.njobs = njobs - j0 < PACK_MAX_JOBS ? njobs - j0 : PACK_MAX_JOBS;
        int nb = 0;
        for (int k = 0; k < t.njobs; ++k) {
            const aesr_prep_job& jb = jobs_host[j0 + k];
            AESR_CHECK_ARG(jb.w && jb.out && jb.Cout > 0 && jb.Cin > 0, "aesr_weight_prep_many: bad job %d", j0 + k);
            PrepJob& o = t.job[k];
            o.w = jb.w; o.aux0 = jb.aux0; o.aux1 = jb.aux1; o.out = jb.out; o.kind = jb.kind; o.Cout = jb.Cout; o.Cin = jb.Cin; o.KS = jb.KS;
            o.transpose = jb.transpose; o.block0 = nb;
            const int kin = jb.transpose ? jb.Cout : jb.Cin, nout = jb.transpose ? jb.Cin : jb.Cout;
            size_t threads = 0;
            if (jb.kind == AESR_PREP_PACK) {                       // as aesr_conv2d_pack_many
                AESR_CHECK_ARG(jb.KS == 1 || jb.KS == 3, "aesr_weight_prep_many: job %d: KS=%d", j0 + k, jb.KS);
                int NP, NB;
                cout_padding(nout, &NP, &NB);
                o.KinP = round_up(kin, 16); o.NoutP = NP; o.TN = 16 * NB;
                threads = ((size_t)jb.KS * jb.KS * o.KinP * NP + 3) / 4;            // 4 elements per thread
            } else if (jb.kind == AESR_PREP_WINO_PACK) {           // as aesr_conv2d_wino_pack_many
                AESR_CHECK_ARG(jb.KS == 3, "aesr_weight_prep_many: job %d: the Winograd transform is for 3x3 filters", j0 + k);
                o.KinP = round_up(kin, 16); o.NoutP = round_up(nout, 32); o.TN = 32;
                threads = (size_t)o.KinP * o.NoutP;
            } else if (jb.kind == AESR_PREP_STEM_FOLD) {           // as aesr_stemconv_fold: Cout = C1, Cin = Cs, aux0 / aux1 = stem weight / bias
                AESR_CHECK_ARG(jb.aux0, "aesr_weight_prep_many: job %d: stem fold needs the stem weight", j0 + k);
                threads = (size_t)9 * jb.Cout;
            } else if (jb.kind == AESR_PREP_COUT1_FLIP) {
                threads = (size_t)9 * jb.Cin;
            } else {
                aesr_set_error("aesr_weight_prep_many: job %d: unknown kind %d", j0 + k, jb.kind);
                return AESR_ERR_ARG;
            }
            int blocks = (int)((threads + 255) / 256);
            if (blocks > 256) blocks = 256;
            nb += blocks;
        }
        t.nblocks = nb;
        if (int e = aesr_launch_prep_many(t, (hipStream_t)stream)) return e;
    }
    return AESR_OK;
}

int aesr_conv2d_fwd(const float* in, const float* packed, const float* bias, float* out, int N, int H, int W, int Cin,
                    int Cout, int KS, int pad, int act, float slope, void* stream) {
    AESR_CHECK_ARG(in && packed && out && N > 0 && H > 0 && W > 0, "aesr_conv2d_fwd: null pointer or empty shape");
    AESR_CHECK_ARG(Cin % 4 == 0 && Cin > 0 && Cout > 0, "aesr_conv2d_fwd: Cin=%d must be a positive multiple of 4", Cin);
    AESR_CHECK_ARG((KS == 1 || KS == 3) && pad >= 0 && pad < KS, "aesr_conv2d_fwd: unsupported KS=%d pad=%d", KS, pad);
    return run_igemm(in, packed, bias, nullptr, out, N, H, W, Cin, Cout, KS, pad, act, ACT_NONE, slope, nullptr, (hipStream_t)stream);
}

size_t aesr_conv2d_workspace_floats(int N, int H, int W, int Cin, int Cout, int KS, int pad) {
    if (!conv_dims_ok(N, H, W, Cin, Cout) || (KS != 1 && KS != 3) || pad < 0 || pad >= KS) return 0;
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const ConvPlan p = plan_conv(N, Ho, Wo, Cin, Cout, KS);
    return p.ksplit > 1 ? (size_t)p.ksplit * N * Ho * Wo * Cout : 0;
}

size_t aesr_conv2d_dgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int KS, int pad) {
    // the data gradient is the forward kernel on dy [N,Ho,Wo,Cout] with Cout and Cin swapped and padding KS-1-pad
    if (!conv_dims_ok(N, H, W, Cin, Cout) || (KS != 1 && KS != 3) || pad < 0 || pad >= KS) return 0;
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    return aesr_conv2d_workspace_floats(N, Ho, Wo, Cout, Cin, KS, KS - 1 - pad);
}

int aesr_conv2d_fwd_ws(const float* in, const float* packed, const float* bias, float* out, float* workspace, int N, int H, int W,
                       int Cin, int Cout, int KS, int pad, int act, float slope, void* stream) {
    AESR_CHECK_ARG(in && packed && out && N > 0 && H > 0 && W > 0, "aesr_conv2d_fwd_ws: null pointer or empty shape");
    AESR_CHECK_ARG(Cin % 4 == 0 && Cin > 0 && Cout > 0, "aesr_conv2d_fwd_ws: Cin=%d must be a positive multiple of 4", Cin);
    AESR_CHECK_ARG((KS == 1 || KS == 3) && pad >= 0 && pad < KS, "aesr_conv2d_fwd_ws: unsupported KS=%d pad=%d", KS, pad);
    return run_igemm(in, packed, bias, nullptr, out, N, H, W, Cin, Cout, KS, pad, act, ACT_NONE, slope, workspace, (hipStream_t)stream);
}

int aesr_conv2d_dgrad_ws(const float* dy, const float* packed_t, const float* x_saved, float* dx, float* workspace, int N, int H,
                         int W, int Cin, int Cout, int KS, int pad, int mask_act, float slope, void* stream) {
    AESR_CHECK_ARG(dy && packed_t && dx && N > 0 && H > 0 && W > 0, "aesr_conv2d_dgrad_ws: null pointer or empty shape");
    AESR_CHECK_ARG(Cout % 4 == 0 && Cin > 0 && Cout > 0, "aesr_conv2d_dgrad_ws: Cout=%d must be a positive multiple of 4", Cout);
    AESR_CHECK_ARG((KS == 1 || KS == 3) && pad >= 0 && pad < KS, "aesr_conv2d_dgrad_ws: unsupported KS=%d pad=%d", KS, pad);
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    return run_igemm(dy, packed_t, nullptr, x_saved, dx, N, Ho, Wo, Cout, Cin, KS, KS - 1 - pad, ACT_NONE, mask_act, slope, workspace,
                     (hipStream_t)stream);
}

int aesr_conv2d_dgrad(const float* dy, const float* packed_t, const float* x_saved, float* dx, int N, int H, int W, int Cin,
                      int Cout, int KS, int pad, int mask_act, float slope, void* stream) {
    AESR_CHECK_ARG(dy && packed_t && dx && N > 0 && H > 0 && W > 0, "aesr_conv2d_dgrad: null pointer or empty shape");
    AESR_CHECK_ARG(Cout % 4 == 0 && Cin > 0 && Cout > 0, "aesr_conv2d_dgrad: Cout=%d must be a positive multiple of 4", Cout);
    AESR_CHECK_ARG((KS == 1 || KS == 3) && pad >= 0 && pad < KS, "aesr_conv2d_dgrad: unsupported KS=%d pad=%d", KS, pad);
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    // dx = conv(dy [N,Ho,Wo,Cout], flipped w) with padding KS-1-pad -> output [N,H,W,Cin]
    return run_igemm(dy, packed_t, nullptr, x_saved, dx, N, Ho, Wo, Cout, Cin, KS, KS - 1 - pad, ACT_NONE, mask_act, slope, nullptr,
                     (hipStream_t)stream);
}

// ---- Winograd F(2x2,3x3) path --------------------------------------------------------------------------------------
int aesr_conv2d_wino_supported(int Cin, int Cout, int KS, int pad, int transpose) {
    const int kin = transpose ? Cout : Cin, nout = transpose ? Cin : Cout;
    return KS == 3 && pad == 1 && kin > 0 && nout > 0 && kin % 16 == 0 && nout % 32 == 0;
}

int aesr_conv2d_wino_kernel(int N, int H, int W, int Cin, int Cout, int KS, int pad, int transpose) {
    if (!aesr_conv2d_wino_supported(Cin, Cout, KS, pad, transpose) || !conv_dims_ok(N, H, W, Cin, Cout)) return 0;
    const int kin = transpose ? Cout : Cin, nout = transpose ? Cin : Cout;
    const WinoPlan p = plan_wino(N, H, W, kin, nout);
    WinoArgs a = {};
    a.N = N; a.H = H; a.W = W; a.Cin = kin; a.Cout = nout;
    a.CinP = p.CinP; a.CoutP = p.CoutP;
    a.plan_cost = p.cost;
    if (aesr_wino_res_ok(a)) return 2;
    a.ws = nullptr;                     // a query: as called with the workspace aesr_conv2d_wino_workspace_floats asks for
    a.ws_floats = ~(size_t)0;
    return aesr_wino_ring_takes(a) ? 3 : 1;
}

size_t aesr_conv2d_wino_workspace_floats(int N, int H, int W, int Cin, int Cout, int transpose) {
    if (!aesr_conv2d_wino_supported(Cin, Cout, 3, 1, transpose) || !conv_dims_ok(N, H, W, Cin, Cout)) return 0;
    const int kin = transpose ? Cout : Cin, nout = transpose ? Cin : Cout;
    const WinoPlan p = plan_wino(N, H, W, kin, nout);
    WinoArgs a = {};
    a.N = N; a.H = H; a.W = W; a.Cin = kin; a.Cout = nout;
    a.CinP = p.CinP; a.CoutP = p.CoutP;
    a.plan_cost = p.cost;
    if (aesr_wino_res_ok(a)) return 0;
    return aesr_wino_ring_workspace_floats(a);
}

unsigned int aesr_conv2d_wino_ring_timeouts(void) {
    (void)hipDeviceSynchronize();
    return aesr_wino_ring_timeouts();
}

size_t aesr_conv2d_wino_packed_floats(int Cout, int Cin, int transpose) {
    const int kin = transpose ? Cout : Cin, nout = transpose ? Cin : Cout;
    return (size_t)16 * round_up(kin, 16) * round_up(nout, 32);
}

int aesr_conv2d_wino_pack_many(const aesr_pack_job* jobs_host, int njobs, void* stream) {
    AESR_CHECK_ARG(jobs_host && njobs > 0, "aesr_conv2d_wino_pack_many: no jobs");
    for (int j0 = 0; j0 < njobs; j0 += PACK_MAX_JOBS) {
        PackTable t;
        memset(&t, 0, sizeof(t));
        t.njobs = njobs - j0 < PACK_MAX_JOBS ? njobs - j0 : PACK_MAX_JOBS;
        int nb = 0;
        for (int k = 0; k < t.njobs; ++k) {
            const aesr_pack_job& jb = jobs_host[j0 + k];
            AESR_CHECK_ARG(jb.w && jb.packed && jb.Cout > 0 && jb.Cin > 0 && jb.KS == 3, "aesr_conv2d_wino_pack_many: bad job %d", j0 + k);
            const int kin = jb.transpose ? jb.Cout : jb.Cin, nout = jb.transpose ? jb.Cin : jb.Cout;
            PackJob& o = t.job[k];
            o.w = jb.w; o.p = jb.packed; o.Cout = jb.Cout; o.Cin = jb.Cin; o.KS = 3; o.KinP = round_up(kin, 16); o.NoutP = round_up(nout, 32);
            o.TN = 32; o.transpose = jb.transpose; o.block0 = nb;
            const size_t pairs = (size_t)o.KinP * o.NoutP;
            int blocks = (int)((pairs + 255) / 256);
            if (blocks > 256) blocks = 256;
            nb += blocks;
        }
        t.nblocks = nb;
        if (int e = aesr_launch_wino_pack_many(t, (hipStream_t)stream)) return e;
    }
    return AESR_OK;
}

static int run_wino(const float* in, const float* upk, const float* bias, const float* ysave, float* out, int N, int H, int W,
                    int Cin, int Cout, int act, int mask_act, float slope, hipStream_t st, int in_up2 = 0, int out_sum2 = 0,
                    float* ws = nullptr, size_t ws_floats = 0) {
    AESR_CHECK_DIMS("aesr_conv2d_wino", N, H, W, Cin, Cout);
    const WinoPlan p = plan_wino(N, H, W, Cin, Cout);
    WinoArgs a = {};
    a.in = in; a.upk = upk; a.bias = bias; a.ysave = ysave; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = p.CinP; a.Cout = Cout; a.CoutP = p.CoutP;
    a.TI = p.TI; a.THt = p.THt; a.TWt = p.TWt; a.regs_y = a.regs_x = a.nitems = 0;
    a.plan_cost = p.cost;
    a.act = act; a.mask_act = mask_act; a.slope = slope; a.dbgbuf = nullptr; a.flags = 0; a.in_up2 = in_up2; a.out_sum2 = out_sum2;
    a.ws = ws_floats ? ws : nullptr; a.ws_floats = ws ? ws_floats : 0;
    a.ksplit = 1;
    return aesr_launch_conv_wino(a, st);
}

int aesr_conv2d_wino_fwd(const float* in, const float* upacked, const float* bias, float* out, int N, int H, int W, int Cin,
                         int Cout, int act, float slope, void* stream) {
    AESR_CHECK_ARG(in && upacked && out && N > 0 && H > 0 && W > 0, "aesr_conv2d_wino_fwd: null pointer or empty shape");
    AESR_CHECK_ARG(aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 0), "aesr_conv2d_wino_fwd: needs Cin %% 16 == 0 and Cout %% 32 == 0 (got %d -> %d)", Cin, Cout);
    return run_wino(in, upacked, bias, nullptr, out, N, H, W, Cin, Cout, act, ACT_NONE, slope, (hipStream_t)stream);
}

int aesr_conv2d_wino_dgrad(const float* dy, const float* upacked_t, const float* x_saved, float* dx, int N, int H, int W, int Cin,
                           int Cout, int mask_act, float slope, void* stream) {
    AESR_CHECK_ARG(dy && upacked_t && dx && N > 0 && H > 0 && W > 0, "aesr_conv2d_wino_dgrad: null pointer or empty shape");
    AESR_CHECK_ARG(aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 1), "aesr_conv2d_wino_dgrad: needs Cout %% 16 == 0 and Cin %% 32 == 0 (got %d -> %d)", Cin, Cout);
    // dx = conv(dy [N,H,W,Cout], flipped / transposed filter), padding 1 -> [N,H,W,Cin]
    return run_wino(dy, upacked_t, nullptr, x_saved, dx, N, H, W, Cout, Cin, ACT_NONE, mask_act, slope, (hipStream_t)stream);
}

int aesr_conv2d_wino_fwd_ws(const float* in, const float* upacked, const float* bias, float* out, float* workspace, size_t workspace_floats,
                            int N, int H, int W, int Cin, int Cout, int act, float slope, void* stream) {
    AESR_CHECK_ARG(in && upacked && out && N > 0 && H > 0 && W > 0, "aesr_conv2d_wino_fwd_ws: null pointer or empty shape");
    AESR_CHECK_ARG(aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 0), "aesr_conv2d_wino_fwd_ws: needs Cin %% 16 == 0 and Cout %% 32 == 0 (got %d -> %d)", Cin, Cout);
    return run_wino(in, upacked, bias, nullptr, out, N, H, W, Cin, Cout, act, ACT_NONE, slope, (hipStream_t)stream, 0, 0, workspace, workspace_floats);
}

int aesr_conv2d_wino_dgrad_ws(const float* dy, const float* upacked_t, const float* x_saved, float* dx, float* workspace, size_t workspace_floats,
                              int N, int H, int W, int Cin, int Cout, int mask_act, float slope, void* stream) {
    AESR_CHECK_ARG(dy && upacked_t && dx && N > 0 && H > 0 && W > 0, "aesr_conv2d_wino_dgrad_ws: null pointer or empty shape");
    AESR_CHECK_ARG(aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 1), "aesr_conv2d_wino_dgrad_ws: needs Cout %% 16 == 0 and Cin %% 32 == 0 (got %d -> %d)", Cin, Cout);
    return run_wino(dy, upacked_t, nullptr, x_saved, dx, N, H, W, Cout, Cin, ACT_NONE, mask_act, slope, (hipStream_t)stream, 0, 0, workspace, workspace_floats);
}

/* conv + activation + eval-mode BatchNorm (per-channel scale / shift) [+ AvgPool2d(2)] as one launch: the resident-filter kernel, and the
   ring kernel where it takes the layer WITHOUT a workspace (no channel split: the epilogue has to see the finished sums) */
int aesr_conv2d_wino_fwd_bn_supported(int N, int H, int W, int Cin, int Cout) {
    if (!aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 0) || !conv_dims_ok(N, H, W, Cin, Cout)) return 0;
    const WinoPlan p = plan_wino(N, H, W, Cin, Cout);
    WinoArgs a = {};
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.CinP = p.CinP; a.CoutP = p.CoutP;
    a.plan_cost = p.cost;
    if (aesr_wino_res_ok(a)) return 1;
    return aesr_wino_ring_takes(a) ? 1 : 0;          /* a.ws_floats = 0: as the call below runs it */
}

int aesr_conv2d_wino_fwd_bn(const float* in, const float* upacked, const float* bias, const float* bn_scale, const float* bn_shift, float* out, int N,
                            int H, int W, int Cin, int Cout, int act, float slope, int pool, void* stream) {
    AESR_CHECK_ARG(in && upacked && bn_scale && bn_shift && out && N > 0 && H > 0 && W > 0, "aesr_conv2d_wino_fwd_bn: null pointer or empty shape");
    AESR_CHECK_ARG(!pool || (H >= 2 && W >= 2), "aesr_conv2d_wino_fwd_bn: pooling needs H, W >= 2");
    AESR_CHECK_ARG(aesr_conv2d_wino_fwd_bn_supported(N, H, W, Cin, Cout), "aesr_conv2d_wino_fwd_bn: %d -> %d at %d x %d x %d is not a resident-filter or ring-kernel layer "
                   "(aesr_conv2d_wino_fwd_bn_supported)", Cin, Cout, N, H, W);
    const WinoPlan p = plan_wino(N, H, W, Cin, Cout);
    WinoArgs a = {};
    a.in = in; a.upk = upacked; a.bias = bias; a.ysave = nullptr; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = p.CinP; a.Cout = Cout; a.CoutP = p.CoutP;
    a.TI = p.TI; a.THt = p.THt; a.TWt = p.TWt;
    a.plan_cost = p.cost;
    a.act = act; a.mask_act = ACT_NONE; a.slope = slope; a.ksplit = 1;
    a.post_scale = bn_scale; a.post_shift = bn_shift; a.post_pool = pool ? 1 : 0;
    return aesr_launch_conv_wino(a, (hipStream_t)stream);
}

/* nearest Upsample(x2) in front of the convolution folded into the kernels (H, W = the convolution's = upsampled size, even) */
int aesr_conv2d_wino_fwd_up2(const float* in_half, const float* upacked, const float* bias, float* out, int N, int H, int W, int Cin,
                             int Cout, int act, float slope, void* stream) {
    AESR_CHECK_ARG(in_half && upacked && out && N > 0 && H > 0 && W > 0 && !((H | W) & 1), "aesr_conv2d_wino_fwd_up2: null pointer, empty or odd shape");
    AESR_CHECK_ARG(aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 0), "aesr_conv2d_wino_fwd_up2: needs Cin %% 16 == 0 and Cout %% 32 == 0 (got %d -> %d)", Cin, Cout);
    return run_wino(in_half, upacked, bias, nullptr, out, N, H, W, Cin, Cout, act, ACT_NONE, slope, (hipStream_t)stream, 1, 0);
}

int aesr_conv2d_wino_dgrad_sum2(const float* dy, const float* upacked_t, float* dx_half, int N, int H, int W, int Cin, int Cout,
                                void* stream) {
    AESR_CHECK_ARG(dy && upacked_t && dx_half && N > 0 && H > 0 && W > 0 && !((H | W) & 1), "aesr_conv2d_wino_dgrad_sum2: null pointer, empty or odd shape");
    AESR_CHECK_ARG(aesr_conv2d_wino_supported(Cin, Cout, 3, 1, 1), "aesr_conv2d_wino_dgrad_sum2: needs Cout %% 16 == 0 and Cin %% 32 == 0 (got %d -> %d)", Cin, Cout);
    return run_wino(dy, upacked_t, nullptr, nullptr, dx_half, N, H, W, Cout, Cin, ACT_NONE, ACT_NONE, 0.f, (hipStream_t)stream, 0, 1);
}

size_t aesr_conv2d_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int KS, int pad) {
    if (!conv_dims_ok(N, H, W, Cin, Cout) || (KS != 1 && KS != 3) || pad < 0 || pad >= KS) return 0;
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    return plan_wgrad(N, Ho, Wo, Cin, Cout, KS, pad).slab_floats;
}

static int wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W, int Cin,
                      int Cout, int KS, int pad, void* stream, int x_up2);

int aesr_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W, int Cin,
                      int Cout, int KS, int pad, void* stream) {
    return wgrad_impl(x, dy, dw, db, workspace, N, H, W, Cin, Cout, KS, pad, stream, 0);
}

int aesr_conv2d_wgrad_partial(const float* x, const float* dy, float* workspace, int N, int H, int W, int Cin, int Cout, int KS,
                              int pad, int x_up2, void* stream) {
    AESR_CHECK_ARG(!x_up2 || (wgrad_wino_ok(Cin, Cout, KS, pad) && !((H | W) & 1)), "aesr_conv2d_wgrad_partial: the folded upsampling needs "
                   "the Winograd weight-gradient kernel (3x3, padding 1, Cin, Cout multiples of 32) and an even size");
    return wgrad_impl(x, dy, nullptr, nullptr, workspace, N, H, W, Cin, Cout, KS, pad, stream, x_up2);
}

int aesr_conv2d_wgrad_reduce_many(const aesr_wgrad_reduce_job* jobs_host, int njobs, void* stream) {
    AESR_CHECK_ARG(jobs_host && njobs > 0, "aesr_conv2d_wgrad_reduce_many: no jobs");
    for (int j0 = 0; j0 < njobs; j0 += REDUCE_MAX_JOBS) {
        ReduceTable t;
        memset(&t, 0, sizeof(t));
        t.njobs = njobs - j0 < REDUCE_MAX_JOBS ? njobs - j0 : REDUCE_MAX_JOBS;
        int nb = 0;
        for (int k = 0; k < t.njobs; ++k) {
            const aesr_wgrad_reduce_job& jb = jobs_host[j0 + k];
            AESR_CHECK_ARG(jb.workspace && jb.dw && jb.N > 0 && (jb.KS == 1 || jb.KS == 3), "aesr_conv2d_wgrad_reduce_many: bad job %d", j0 + k);
            AESR_CHECK_ARG(conv_dims_ok(jb.N, jb.H, jb.W, jb.Cin, jb.Cout) && jb.pad >= 0 && jb.pad < jb.KS && jb.H + 2 * jb.pad >= jb.KS && jb.W + 2 * jb.pad >= jb.KS,
                           "aesr_conv2d_wgrad_reduce_many: job %d: bad shape", j0 + k);
            const int Ho = jb.H + 2 * jb.pad - jb.KS + 1, Wo = jb.W + 2 * jb.pad - jb.KS + 1;
            const WgradPlan p = plan_wgrad(jb.N, Ho, Wo, jb.Cin, jb.Cout, jb.KS, jb.pad);      // the plan the partial launch used
            ReduceJob& o = t.job[k];
            o.slab = jb.workspace; o.dw = jb.dw; o.db = jb.db; o.nslab = p.nslab; o.KS2 = jb.KS * jb.KS; o.Cin = jb.Cin; o.CinP = p.CinP;
            o.Cout = jb.Cout; o.CoutP = p.CoutP; o.block0 = nb;
            nb += ceil_div(o.KS2 * jb.Cin * jb.Cout + (jb.db ? jb.Cout : 0), 64);
        }
        t.nblocks = nb;
        if (int e = aesr_launch_wgrad_reduce_many(t, (hipStream_t)stream)) return e;
    }
    return AESR_OK;
}

int aesr_conv2d_wgrad_up2_supported(int Cin, int Cout) { return wgrad_wino_ok(Cin, Cout, 3, 1) ? 1 : 0; }

int aesr_conv2d_wgrad_up2(const float* x_half, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W, int Cin,
                          int Cout, void* stream) {
    AESR_CHECK_ARG(wgrad_wino_ok(Cin, Cout, 3, 1) && !((H | W) & 1), "aesr_conv2d_wgrad_up2: needs the Winograd weight-gradient kernel "
                   "(Cin, Cout multiples of 32; got %d -> %d) and an even size", Cin, Cout);
    return wgrad_impl(x_half, dy, dw, db, workspace, N, H, W, Cin, Cout, 3, 1, stream, 1);
}

static int wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W, int Cin,
                      int Cout, int KS, int pad, void* stream, int x_up2) {
    AESR_CHECK_ARG(x && dy && workspace && N > 0, "aesr_conv2d_wgrad: null pointer or empty shape");
    AESR_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0, "aesr_conv2d_wgrad: Cin=%d, Cout=%d must be multiples of 4", Cin, Cout);
    AESR_CHECK_ARG((KS == 1 || KS == 3) && pad >= 0 && pad < KS, "aesr_conv2d_wgrad: unsupported KS=%d pad=%d", KS, pad);
    AESR_CHECK_DIMS("aesr_conv2d_wgrad", N, H, W, Cin, Cout);
    const int Ho = H + 2 * pad - KS + 1, Wo = W + 2 * pad - KS + 1;
    AESR_CHECK_ARG(Ho > 0 && Wo > 0, "aesr_conv2d_wgrad: the %d x %d input is smaller than the %d x %d filter", H, W, KS, KS);
    const WgradPlan p = plan_wgrad(N, Ho, Wo, Cin, Cout, KS, pad);
    WgradArgs a;
    a.x = x; a.dy = dy; a.slab = workspace;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.CinP = p.CinP; a.Cout = Cout; a.CoutP = p.CoutP; a.Ho = Ho; a.Wo = Wo; a.pad = pad;
    a.TH = p.TH; a.TW = p.TW; a.tiles_y = ceil_div(Ho, p.TH); a.tiles_x = ceil_div(Wo, p.TW);
    a.ntiles = N * a.tiles_y * a.tiles_x; a.S = p.S;
    a.PWS = p.PWS; a.TWS = p.TWS; a.PSX = p.PSX; a.PSD = p.PSD; a.dbgbuf = nullptr; a.x_up2 = x_up2;
    if (p.variant == 2) {
        if (int e = aesr_launch_conv_wgrad_wino(a, (hipStream_t)stream)) return e;
    } else if (int e = aesr_launch_conv_wgrad(a, KS, p.variant, (hipStream_t)stream)) {
        return e;
    }
    if (!dw) return AESR_OK;            // partial-slab form (aesr_conv2d_wgrad_partial): the caller reduces later
    return aesr_launch_wgrad_reduce(workspace, dw, db, p.nslab, KS, Cin, p.CinP, Cout, p.CoutP, (hipStream_t)stream);
}

int aesr_conv2d_smallcin_fwd(const float* in, const float* w, const float* bias, const float* y_saved, float* out, int N,
                             int H, int W, int Cin, int Cout, int KS, int pad, int act, int mask_act, float slope,
                             int transpose, int bcast, const float* ca_host, const float* cb_host, void* stream) {
    AESR_CHECK_ARG(in && w && out && N > 0 && Cin >= 1 && Cin <= 4 && Cout > 0, "aesr_conv2d_smallcin_fwd: need 1 <= Cin <= 4");
    AESR_CHECK_ARG(!bcast || (ca_host && cb_host), "aesr_conv2d_smallcin_fwd: bcast needs ca/cb");
    SmallArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = w; a.bias = bias; a.ysave = y_saved; a.out = out;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KS = KS; a.pad = pad;
    a.Ho = H + 2 * pad - KS + 1; a.Wo = W + 2 * pad - KS + 1;
    a.act = act; a.mask_act = mask_act; a.slope = slope; a.transpose = transpose; a.bcast = bcast;
    for (int i = 0; i < Cin && bcast; ++i) { a.ca[i] = ca_host[i]; a.cb[i] = cb_host[i]; }
    return aesr_launch_smallcin_fwd(a, (hipStream_t)stream);
}

int aesr_conv2d_smallcin_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int KS,
                               int pad, int bcast, const float* ca_host, void* stream) {
    AESR_CHECK_ARG(dy && w && dx && N > 0 && Cin >= 1 && Cin <= 4 && Cout > 0, "aesr_conv2d_smallcin_dgrad: need 1 <= Cin <= 4");
    AESR_CHECK_ARG(!bcast || ca_host, "aesr_conv2d_smallcin_dgrad: bcast needs ca");
    SmallDgradArgs a;
    memset(&a, 0, sizeof(a));
    a.dy = dy; a.w = w; a.dx = dx; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.KS = KS; a.pad = pad; a.bcast = bcast;
    a.Ho = H + 2 * pad - KS + 1; a.Wo = W + 2 * pad - KS + 1;
    for (int i = 0; i < Cin && bcast; ++i) a.ca[i] = ca_host[i];
    return aesr_launch_smallcin_dgrad(a, (hipStream_t)stream);
}

#define SMALL_WGRAD_NWG 512
size_t aesr_small_wgrad_workspace_floats(int nout) { return (size_t)SMALL_WGRAD_NWG * nout; }

int aesr_conv2d_smallcin_wgrad(const float* in, const float* dout, float* dw, float* db, float* workspace, int N, int H,
                               int W, int Cin, int Cout, int pad, void* stream) {
    AESR_CHECK_ARG(in && dout && dw && db && workspace && Cin >= 1 && Cin <= 4, "aesr_conv2d_smallcin_wgrad: need 1 <= Cin <= 4");
    AESR_CHECK_ARG(Cout >= 4 && Cout % 4 == 0 && 256 % (Cout / 4) == 0 && Cout <= 256,
                   "aesr_conv2d_smallcin_wgrad: Cout=%d must be a multiple of 4 with Cout/4 dividing 256", Cout);
    SmallWgradArgs a;
    a.in = in; a.dout = dout; a.partial = workspace; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.pad = pad;
    a.Ho = H + 2 * pad; a.Wo = W + 2 * pad;
    if (int e = aesr_launch_smallcin_wgrad(a, SMALL_WGRAD_NWG, (hipStream_t)stream)) return e;
    return aesr_launch_sum_partials(workspace, SMALL_WGRAD_NWG, Cout * (Cin + 1), dw, Cout * Cin, db, (hipStream_t)stream);
}

int aesr_conv2d_cout1_fwd(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int Cin, int act,
                          float slope, void* stream) {
    AESR_CHECK_ARG(x && w && out && N > 0 && H > 0 && W > 0, "aesr_conv2d_cout1_fwd: null pointer or empty shape");
    AESR_CHECK_ARG(Cin > 0 && Cin % 4 == 0 && Cin <= 256, "aesr_conv2d_cout1_fwd: Cin=%d must be a multiple of 4 (<= 256)", Cin);
    if (((Cin / 4) & (Cin / 4 - 1)) == 0)
        return aesr_launch_thin_collapse(x, w, bias, out, N, H, W, Cin, act, slope, (hipStream_t)stream);
    Cout1FwdArgs a;
    a.x = x; a.w = w; a.bias = bias; a.out = out; a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.act = act; a.slope = slope;
    a.TH = H < 16 ? H : 16; a.TW = W < 16 ? W : 16;
    while (((size_t)(a.TH + 2) * (a.TW + 2) * (Cin + 4) + 9 * Cin) * 4 > 64 * 1024 && a.TH > 1) a.TH /= 2;
    a.tiles_y = ceil_div(H, a.TH); a.tiles_x = ceil_div(W, a.TW);
    return aesr_launch_cout1_fwd(a, (hipStream_t)stream);
}

#define THIN_NWG 512
static bool thin_channels_ok(int C) { return C >= 4 && C <= 256 && C % 4 == 0 && ((C / 4) & (C / 4 - 1)) == 0; }

size_t aesr_conv2d_cout1_workspace_floats(int Cin) {
    const size_t legacy = (size_t)SMALL_WGRAD_NWG * (Cin * 9 + 1);
    const size_t thin = (size_t)(THIN_NWG + 1) * 10 * Cin;
    return legacy > thin ? legacy : thin;
}

int aesr_conv2d_cout1_wgrad(const float* x, const float* dy, float* dw, float* db, float* workspace, int N, int H, int W,
                            int Cin, void* stream) {
    AESR_CHECK_ARG(x && dy && dw && db && workspace, "aesr_conv2d_cout1_wgrad: null pointer");
    AESR_CHECK_ARG(N > 0 && H > 0 && W > 0, "aesr_conv2d_cout1_wgrad: empty shape");
    if (thin_channels_ok(Cin)) {
        // dW[0,ci,ky,kx] = sum_u X[u,ci] * dy[u - (ky-1, kx-1)]  ->  thin reduce of X against dy, taps flipped
        ThinArgs a;
        memset(&a, 0, sizeof(a));
        a.s = dy; a.t = x; a.partial = workspace;
        a.N = N; a.Hs = H; a.Ws = W; a.Ho = H; a.Wo = W; a.C = Cin; a.ps = 0; a.with_be = 0;
        float* R = workspace + (size_t)THIN_NWG * 10 * Cin;
        if (int e = aesr_launch_thin_reduce(a, THIN_NWG, (hipStream_t)stream)) return e;
        if (int e = aesr_launch_sum_partials(workspace, THIN_NWG, 10 * Cin, R, 10 * Cin, nullptr, (hipStream_t)stream)) return e;
        return aesr_launch_thin_cout1_finish(R, dw, db, Cin, (hipStream_t)stream);
    }
    AESR_CHECK_ARG(Cin >= 4 && Cin <= 128 && 256 % Cin == 0, "aesr_conv2d_cout1_wgrad: Cin=%d must divide 256 (and be >= 4)", Cin);
    Cout1WgradArgs a;
    a.x = x; a.dy = dy; a.partial = workspace; a.N = N; a.H = H; a.W = W; a.Cin = Cin;
    a.TH = H < 16 ? H : 16; a.TW = W < 16 ? W : 16;
    while ((size_t)(a.TH + 2) * (a.TW + 2) * (Cin + 1) * 4 > 60 * 1024 && a.TH > 1) a.TH /= 2;
    a.tiles_y = ceil_div(H, a.TH); a.tiles_x = ceil_div(W, a.TW); a.ntiles = N * a.tiles_y * a.tiles_x;
    int nwg = a.ntiles < SMALL_WGRAD_NWG ? a.ntiles : SMALL_WGRAD_NWG;
    if (int e = aesr_launch_cout1_wgrad(a, nwg, (hipStream_t)stream)) return e;
    return aesr_launch_sum_partials(workspace, nwg, Cin * 9 + 1, dw, Cin * 9, db, (hipStream_t)stream);
}

int aesr_conv2d_cout1_dgrad(const float* dy, const float* w, const float* y_saved, float* dx, float* workspace, int N, int H,
                            int W, int Cin, int mask_act, float slope, void* stream) {
    AESR_CHECK_ARG(dy && w && dx && workspace && N > 0 && H > 0 && W > 0, "aesr_conv2d_cout1_dgrad: null pointer or empty shape");
    AESR_CHECK_ARG(thin_channels_ok(Cin), "aesr_conv2d_cout1_dgrad: Cin=%d must be 4 times a power of two (4..256)", Cin);
    if (int e = aesr_launch_thin_cout1_flip(w, workspace, Cin, (hipStream_t)stream)) return e;
    ThinArgs a;
    memset(&a, 0, sizeof(a));
    a.s = dy; a.w = workspace; a.ysave = y_saved; a.out = dx;
    a.N = N; a.Hs = H; a.Ws = W; a.Ho = H; a.Wo = W; a.C = Cin; a.ps = 0;
    a.act = ACT_NONE; a.mask_act = y_saved ? mask_act : ACT_NONE; a.slope = slope;
    return aesr_launch_thin_expand(a, (hipStream_t)stream);
}

int aesr_conv2d_cout1_dgrad_pre(const float* dy, const float* w_flipped, const float* y_saved, float* dx, int N, int H, int W, int Cin,
                                int mask_act, float slope, void* stream) {
    AESR_CHECK_ARG(dy && w_flipped && dx && N > 0 && H > 0 && W > 0, "aesr_conv2d_cout1_dgrad_pre: null pointer or empty shape");
    AESR_CHECK_ARG(thin_channels_ok(Cin), "aesr_conv2d_cout1_dgrad_pre: Cin=%d must be 4 times a power of two (4..256)", Cin);
    ThinArgs a;
    memset(&a, 0, sizeof(a));
    a.s = dy; a.w = w_flipped; a.ysave = y_saved; a.out = dx;
    a.N = N; a.Hs = H; a.Ws = W; a.Ho = H; a.Wo = W; a.C = Cin; a.ps = 0;
    a.act = ACT_NONE; a.mask_act = y_saved ? mask_act : ACT_NONE; a.slope = slope;
    return aesr_launch_thin_expand(a, (hipStream_t)stream);
}

size_t aesr_stemconv_folded_floats(int C1) { return (size_t)2 * 9 * C1; }

int aesr_stemconv_fold(const float* w_stem, const float* b_stem, const float* w1, float* folded, int Cs, int C1, void* stream) {
    AESR_CHECK_ARG(w_stem && w1 && folded && Cs > 0 && C1 > 0, "aesr_stemconv_fold: bad arguments");
    return aesr_launch_thin_stem_fold(w_stem, b_stem, w1, folded, Cs, C1, (hipStream_t)stream);
}

int aesr_stemconv_fwd(const float* x, const float* folded, const float* b1, float* out, int N, int H, int W, int C1,
                      int stem_pad, int act, float slope, void* stream) {
    AESR_CHECK_ARG(x && folded && out && N > 0 && H > 0 && W > 0 && stem_pad >= 0, "aesr_stemconv_fwd: bad arguments");
    AESR_CHECK_ARG(thin_channels_ok(C1), "aesr_stemconv_fwd: C1=%d must be 4 times a power of two (4..256)", C1);
    ThinArgs a;
    memset(&a, 0, sizeof(a));
    a.s = x; a.w = folded; a.be = folded + (size_t)9 * C1; a.b = b1; a.out = out;
    a.N = N; a.Hs = H; a.Ws = W; a.Ho = H + 2 * stem_pad; a.Wo = W + 2 * stem_pad; a.C = C1; a.ps = stem_pad;
    a.act = act; a.mask_act = ACT_NONE; a.slope = slope;
    return aesr_launch_thin_expand(a, (hipStream_t)stream);
}

size_t aesr_stemconv_workspace_floats(int C1) { return (size_t)(THIN_NWG + 1) * 19 * C1; }

int aesr_stemconv_wgrad(const float* x, const float* g, const float* w_stem, const float* b_stem, const float* w1,
                        float* dw_stem, float* db_stem, float* dw1, float* db1, float* workspace, int N, int H, int W, int Cs,
                        int C1, int stem_pad, void* stream) {
    AESR_CHECK_ARG(x && g && w_stem && w1 && dw_stem && dw1 && workspace, "aesr_stemconv_wgrad: null pointer");
    AESR_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cs > 0 && stem_pad >= 0, "aesr_stemconv_wgrad: bad shape");
    AESR_CHECK_ARG(thin_channels_ok(C1), "aesr_stemconv_wgrad: C1=%d must be 4 times a power of two (4..256)", C1);
    AESR_CHECK_ARG(!db_stem || b_stem, "aesr_stemconv_wgrad: db_stem needs b_stem");
    ThinArgs a;
    memset(&a, 0, sizeof(a));
    a.s = x; a.t = g; a.partial = workspace;
    a.N = N; a.Hs = H; a.Ws = W; a.Ho = H + 2 * stem_pad; a.Wo = W + 2 * stem_pad; a.C = C1; a.ps = stem_pad; a.with_be = 1;
    float* R = workspace + (size_t)THIN_NWG * 19 * C1;
    if (int e = aesr_launch_thin_reduce(a, THIN_NWG, (hipStream_t)stream)) return e;
    if (int e = aesr_launch_sum_partials(workspace, THIN_NWG, 19 * C1, R, 19 * C1, nullptr, (hipStream_t)stream)) return e;
    return aesr_launch_thin_stem_finish(R, w_stem, b_stem, w1, dw_stem, db_stem, dw1, db1, Cs, C1, (hipStream_t)stream);
}

int aesr_resample2_fwd(const float* x, float* out, int N, int H, int W, int C, int mode, void* stream) {
    AESR_CHECK_ARG(x && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "aesr_resample2_fwd: bad arguments (C %% 4 == 0)");
    AESR_CHECK_ARG(mode >= AESR_RS_POOL && mode <= AESR_RS_BILINEAR, "aesr_resample2_fwd: unknown mode %d", mode);
    AESR_CHECK_ARG(mode != AESR_RS_POOL || (H >= 2 && W >= 2), "aesr_resample2_fwd: pooling needs H, W >= 2");
    return aesr_launch_resample2(x, nullptr, nullptr, out, N, H, W, C, mode, 0, ACT_NONE, 0.f, (hipStream_t)stream);
}

int aesr_resample2_bwd(const float* gout, const float* x_saved, float* dx, int N, int H, int W, int C, int mode, int mask_act,
                       float slope, void* stream) {
    AESR_CHECK_ARG(gout && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "aesr_resample2_bwd: bad arguments (C %% 4 == 0)");
    AESR_CHECK_ARG(mode >= AESR_RS_POOL && mode <= AESR_RS_BILINEAR, "aesr_resample2_bwd: unknown mode %d", mode);
    AESR_CHECK_ARG(mode != AESR_RS_POOL || (H >= 2 && W >= 2), "aesr_resample2_bwd: pooling needs H, W >= 2");
    return aesr_launch_resample2(nullptr, gout, x_saved, dx, N, H, W, C, mode, 1, x_saved ? mask_act : ACT_NONE, slope,
                                 (hipStream_t)stream);
}

int aesr_bn_stats(const float* y, float* partial, double* sums, int HW, int C, int G, const int* nstart_host, void* stream) {
    BnGroups gr;
    AESR_CHECK_ARG(y && partial && sums && fill_groups(&gr, G, nstart_host), "aesr_bn_stats: bad arguments");
    if (int e = aesr_launch_bn_stats(y, partial, HW, C, gr, AESR_BN_NWG, (hipStream_t)stream)) return e;
    return aesr_launch_bn_reduce(partial, sums, AESR_BN_NWG, C, G, (hipStream_t)stream);
}

int aesr_bn_finalize(const double* sums, const double* counts_host, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                     int C, int G, float momentum, float eps, int train, int update_running, void* stream) {
    AESR_CHECK_ARG(gamma && beta && mean && invstd && scale && shift && G >= 1 && G <= 4, "aesr_bn_finalize: bad arguments");
    AESR_CHECK_ARG(!train || (sums && counts_host), "aesr_bn_finalize: train mode needs sums and counts");
    AESR_CHECK_ARG(train || (running_mean && running_var), "aesr_bn_finalize: eval mode needs running stats");
    return aesr_launch_bn_finalize(sums, counts_host, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, mean,
                                   invstd, scale, shift, C, G, momentum, eps, train, update_running && running_mean && running_var,
                                   (hipStream_t)stream);
}

int aesr_bn_stats_finalize(const float* y, float* partial, const double* counts_host, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                           float* scale, float* shift, int HW, int C, int G, const int* nstart_host, float momentum, float eps,
                           int update_running, void* stream) {
    BnGroups gr;
    AESR_CHECK_ARG(y && partial && counts_host && fill_groups(&gr, G, nstart_host), "aesr_bn_stats_finalize: bad arguments");
    AESR_CHECK_ARG(gamma && beta && mean && invstd && scale && shift, "aesr_bn_stats_finalize: null pointer");
    if (int e = aesr_launch_bn_stats(y, partial, HW, C, gr, AESR_BN_NWG, (hipStream_t)stream)) return e;
    return aesr_launch_bn_reduce_finalize(partial, AESR_BN_NWG, counts_host, gamma, beta, running_mean, running_var,
                                          (long long*)num_batches_tracked, mean, invstd, scale, shift, C, G, momentum, eps,
                                          update_running && running_mean && running_var, (hipStream_t)stream);
}

static void bn_out_dims(int H, int W, int mode, int* Ho, int* Wo) {
    if (mode == AESR_BN_POOL) { *Ho = H / 2; *Wo = W / 2; }
    else if (mode == AESR_BN_UP) { *Ho = 2 * H; *Wo = 2 * W; }
    else { *Ho = H; *Wo = W; }
}

int aesr_bn_apply(const float* y, const float* scale, const float* shift, float* out, int N, int H, int W, int C, int mode,
                  int G, const int* nstart_host, void* stream) {
    BnApplyArgs a;
    AESR_CHECK_ARG(y && scale && shift && out && fill_groups(&a.gr, G, nstart_host), "aesr_bn_apply: bad arguments");
    a.y = y; a.scale = scale; a.shift = shift; a.out = out; a.N = N; a.H = H; a.W = W; a.C = C; a.mode = mode;
    bn_out_dims(H, W, mode, &a.Ho, &a.Wo);
    AESR_CHECK_ARG(a.Ho > 0 && a.Wo > 0, "aesr_bn_apply: empty output");
    return aesr_launch_bn_apply(a, (hipStream_t)stream);
}

int aesr_bn_fused_supported(int C, int G) { return aesr_bn_fused_ok(C, G) ? 1 : 0; }

int aesr_bn_fused1_supported(int N, int H, int W, int C, int mode, int G, int backward) {
    if (mode != AESR_BN_NONE && mode != AESR_BN_POOL) return 0;
    return aesr_bn_fused1_ok(N, H, W, C, mode == AESR_BN_POOL, G, backward) ? 1 : 0;
}

size_t aesr_bn_fused1_workspace_floats(int C, int G) { return (size_t)256 * G * 2 * C; }
size_t aesr_bn_fused1_barrier_words(void) { return 16 * 32; }
unsigned int aesr_bn_fused1_timeouts(void) { return aesr_bn_fused_timeouts_impl(); }

int aesr_bn_fused1_fwd(const float* y, float* out, float* workspace, unsigned int* barrier_state, const double* counts_host, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                       float* scale, float* shift, int N, int H, int W, int C, int mode, int G, const int* nstart_host, float momentum, float eps,
                       int update_running, void* stream) {
    BnGroups gr;
    AESR_CHECK_ARG(y && out && workspace && barrier_state && counts_host && gamma && beta && mean && invstd && scale && shift &&
                       fill_groups(&gr, G, nstart_host) && gr.nstart[G] == N, "aesr_bn_fused1_fwd: bad arguments");
    AESR_CHECK_ARG(mode == AESR_BN_NONE || mode == AESR_BN_POOL, "aesr_bn_fused1_fwd: mode %d (the un-folded Upsample takes the three-launch path)", mode);
    return aesr_bn_fused_run(y, nullptr, out, workspace, barrier_state, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, mean,
                             invstd, scale, shift, nullptr, nullptr, nullptr, N, H, W, C, mode == AESR_BN_POOL, G, gr.nstart, counts_host, momentum, eps,
                             update_running && running_mean && running_var, ACT_NONE, 0.f, 0, nullptr, (hipStream_t)stream);
}

int aesr_bn_fused1_bwd(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale, float* workspace,
                       unsigned int* barrier_state, const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N, int H,
                       int W, int C, int mode, int act, float slope, int G, const int* nstart_host, void* stream) {
    BnGroups gr;
    AESR_CHECK_ARG(gout && y && mean && invstd && scale && workspace && barrier_state && counts_host && coef && dgamma && dbeta && dpre &&
                       fill_groups(&gr, G, nstart_host) && gr.nstart[G] == N, "aesr_bn_fused1_bwd: bad arguments");
    AESR_CHECK_ARG(mode == AESR_BN_NONE || mode == AESR_BN_POOL, "aesr_bn_fused1_bwd: mode %d (the un-folded Upsample takes the three-launch path)", mode);
    return aesr_bn_fused_run(y, gout, dpre, workspace, barrier_state, nullptr, nullptr, nullptr, nullptr, nullptr, (float*)mean, (float*)invstd,
                             (float*)scale, nullptr, coef, dgamma, dbeta, N, H, W, C, mode == AESR_BN_POOL, G, gr.nstart, counts_host, 0.f, 0.f, 0, act, slope,
                             1, nullptr, (hipStream_t)stream);
}

/* ---- the same two with the SyncBN exchange inside: data parallel over peer-mapped regions ---- */
size_t aesr_p2p_region_bytes(int world) { return world > 0 && world <= 8 ? (size_t)AESR_P2P_SLOTS * 2 * world * AESR_P2P_REC_BYTES : 0; }

int aesr_p2p_tick(unsigned int* gen_dev, void* stream) {
    AESR_CHECK_ARG(gen_dev, "aesr_p2p_tick: null pointer");
    return aesr_launch_p2p_tick(gen_dev, (hipStream_t)stream);
}

static int fill_p2p(BnP2P* p, void* const* peers_host, int world, int rank, int slot, const unsigned int* gen_dev, const char* who) {
    if (!peers_host || !gen_dev || world < 1 || world > 8 || rank < 0 || rank >= world || slot < 0 || slot >= AESR_P2P_SLOTS) {
        aesr_set_error("%s: bad exchange arguments (world %d, rank %d, slot %d of %d)", who, world, rank, slot, AESR_P2P_SLOTS);
        return AESR_ERR_ARG;
    }
    p->world = world; p->rank = rank; p->slot = slot; p->gen = gen_dev;
    for (int r = 0; r < 8; ++r) p->peers[r] = r < world ? peers_host[r] : nullptr;
    for (int r = 0; r < world; ++r)
        if (!p->peers[r]) {
            aesr_set_error("%s: the region of rank %d is not mapped", who, r);
            return AESR_ERR_ARG;
        }
    return AESR_OK;
}

int aesr_bn_fused1_fwd_p2p(const float* y, float* out, float* workspace, unsigned int* barrier_state, const double* counts_host, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd,
                           float* scale, float* shift, int N, int H, int W, int C, int mode, int G, const int* nstart_host, float momentum, float eps,
                           int update_running, void* const* peers_host, int world, int rank, int slot, const unsigned int* gen_dev, void* stream) {
    BnGroups gr;
    BnP2P p2p;
    AESR_CHECK_ARG(y && out && workspace && barrier_state && counts_host && gamma && beta && mean && invstd && scale && shift &&
                       fill_groups(&gr, G, nstart_host) && gr.nstart[G] == N, "aesr_bn_fused1_fwd_p2p: bad arguments");
    AESR_CHECK_ARG(mode == AESR_BN_NONE || mode == AESR_BN_POOL, "aesr_bn_fused1_fwd_p2p: mode %d", mode);
    if (int e = fill_p2p(&p2p, peers_host, world, rank, slot, gen_dev, "aesr_bn_fused1_fwd_p2p")) return e;
    return aesr_bn_fused_run(y, nullptr, out, workspace, barrier_state, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, mean,
                             invstd, scale, shift, nullptr, nullptr, nullptr, N, H, W, C, mode == AESR_BN_POOL, G, gr.nstart, counts_host, momentum, eps,
                             update_running && running_mean && running_var, ACT_NONE, 0.f, 0, &p2p, (hipStream_t)stream);
}

int aesr_bn_fused1_bwd_p2p(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale, float* workspace,
                           unsigned int* barrier_state, const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N, int H,
                           int W, int C, int mode, int act, float slope, int G, const int* nstart_host, void* const* peers_host, int world, int rank,
                           int slot, const unsigned int* gen_dev, void* stream) {
    BnGroups gr;
    BnP2P p2p;
    AESR_CHECK_ARG(gout && y && mean && invstd && scale && workspace && barrier_state && counts_host && coef && dgamma && dbeta && dpre &&
                       fill_groups(&gr, G, nstart_host) && gr.nstart[G] == N, "aesr_bn_fused1_bwd_p2p: bad arguments");
    AESR_CHECK_ARG(mode == AESR_BN_NONE || mode == AESR_BN_POOL, "aesr_bn_fused1_bwd_p2p: mode %d", mode);
    if (int e = fill_p2p(&p2p, peers_host, world, rank, slot, gen_dev, "aesr_bn_fused1_bwd_p2p")) return e;
    return aesr_bn_fused_run(y, gout, dpre, workspace, barrier_state, nullptr, nullptr, nullptr, nullptr, nullptr, (float*)mean, (float*)invstd,
                             (float*)scale, nullptr, coef, dgamma, dbeta, N, H, W, C, mode == AESR_BN_POOL, G, gr.nstart, counts_host, 0.f, 0.f, 0, act, slope,
                             1, &p2p, (hipStream_t)stream);
}

int aesr_bn_finalize_apply(const double* sums, const double* counts_host, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                           const float* y, float* out, int N, int H, int W, int C, int mode, int G, const int* nstart_host, float momentum,
                           float eps, int update_running, void* stream) {
    BnApplyArgs a;
    AESR_CHECK_ARG(sums && counts_host && gamma && beta && mean && invstd && scale && shift && y && out && fill_groups(&a.gr, G, nstart_host),
                   "aesr_bn_finalize_apply: bad arguments");
    AESR_CHECK_ARG(aesr_bn_fused_ok(C, G), "aesr_bn_finalize_apply: %d groups x %d channels exceed the kernel's tables (aesr_bn_fused_supported)", G, C);
    a.y = y; a.scale = scale; a.shift = shift; a.out = out; a.N = N; a.H = H; a.W = W; a.C = C; a.mode = mode;
    bn_out_dims(H, W, mode, &a.Ho, &a.Wo);
    AESR_CHECK_ARG(a.Ho > 0 && a.Wo > 0, "aesr_bn_finalize_apply: empty output");
    return aesr_launch_bn_finalize_apply(sums, counts_host, gamma, beta, running_mean, running_var, (long long*)num_batches_tracked, mean, invstd,
                                         scale, shift, momentum, eps, update_running && running_mean && running_var, G, a, (hipStream_t)stream);
}

int aesr_bn_bwd_reduce(const float* gout, const float* y, const float* mean, const float* invstd, float* partial, double* sums,
                       int N, int H, int W, int C, int mode, int G, const int* nstart_host, void* stream) {
    BnBwdArgs a;
    memset(&a, 0, sizeof(a));
    AESR_CHECK_ARG(gout && y && mean && invstd && partial && sums && fill_groups(&a.gr, G, nstart_host), "aesr_bn_bwd_reduce: bad arguments");
    a.gout = gout; a.y = y; a.mean = mean; a.invstd = invstd; a.partial = partial;
    a.N = N; a.H = H; a.W = W; a.C = C; a.mode = mode;
    AESR_CHECK_ARG((double)N * H * W < 2147483648.0, "aesr_bn_bwd_reduce: more than 2^31 pixels");
    bn_out_dims(H, W, mode, &a.Ho, &a.Wo);
    if (int e = aesr_launch_bn_bwd_reduce(a, AESR_BN_NWG, (hipStream_t)stream)) return e;
    return aesr_launch_bn_reduce(partial, sums, AESR_BN_NWG, C, G, (hipStream_t)stream);
}

int aesr_bn_bwd_apply(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale,
                      const double* sums, const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N,
                      int H, int W, int C, int mode, int act, float slope, int G, const int* nstart_host, void* stream) {
    BnBwdArgs a;
    memset(&a, 0, sizeof(a));
    AESR_CHECK_ARG(gout && y && mean && invstd && scale && sums && counts_host && coef && dgamma && dbeta && dpre &&
                       fill_groups(&a.gr, G, nstart_host), "aesr_bn_bwd_apply: bad arguments");
    a.gout = gout; a.y = y; a.mean = mean; a.invstd = invstd; a.scale = scale; a.coef = coef; a.dpre = dpre;
    a.N = N; a.H = H; a.W = W; a.C = C; a.mode = mode; a.act = act; a.slope = slope;
    bn_out_dims(H, W, mode, &a.Ho, &a.Wo);
    if (aesr_bn_fused_ok(C, G))          // coef / dgamma / dbeta from the sums in the apply kernel's prologue: one launch
        return aesr_launch_bn_bwd_finalize_apply(sums, counts_host, coef, dgamma, dbeta, G, a, (hipStream_t)stream);
    if (int e = aesr_launch_bn_bwd_finalize(sums, counts_host, coef, dgamma, dbeta, C, G, (hipStream_t)stream)) return e;
    return aesr_launch_bn_bwd_apply(a, (hipStream_t)stream);
}

int aesr_bn_bwd(const float* gout, const float* y, const float* mean, const float* invstd, const float* scale, float* partial,
                const double* counts_host, float* coef, float* dgamma, float* dbeta, float* dpre, int N, int H, int W, int C,
                int mode, int act, float slope, int G, const int* nstart_host, void* stream) {
    BnBwdArgs a;
    memset(&a, 0, sizeof(a));
    AESR_CHECK_ARG(gout && y && mean && invstd && scale && partial && counts_host && coef && dgamma && dbeta && dpre &&
                       fill_groups(&a.gr, G, nstart_host), "aesr_bn_bwd: bad arguments");
    AESR_CHECK_ARG((double)N * H * W < 2147483648.0, "aesr_bn_bwd: more than 2^31 pixels");
    a.gout = gout; a.y = y; a.mean = mean; a.invstd = invstd; a.scale = scale; a.coef = coef; a.dpre = dpre; a.partial = partial;
    a.N = N; a.H = H; a.W = W; a.C = C; a.mode = mode; a.act = act; a.slope = slope;
    bn_out_dims(H, W, mode, &a.Ho, &a.Wo);
    if (int e = aesr_launch_bn_bwd_reduce(a, AESR_BN_NWG, (hipStream_t)stream)) return e;
    if (int e = aesr_launch_bn_bwd_reduce_finalize(partial, AESR_BN_NWG, counts_host, coef, dgamma, dbeta, C, G, (hipStream_t)stream))
        return e;
    return aesr_launch_bn_bwd_apply(a, (hipStream_t)stream);
}

int aesr_maxpool2_fwd(const float* x, float* out, int N, int H, int W, int C, void* stream) {
    AESR_CHECK_ARG(x && out && N > 0 && H >= 2 && W >= 2 && C % 4 == 0, "aesr_maxpool2_fwd: bad arguments (C %% 4 == 0, H,W >= 2)");
    return aesr_launch_maxpool2_fwd(x, out, N, H, W, C, (hipStream_t)stream);
}

int aesr_maxpool2_bwd(const float* gout, const float* x, const float* gadd, float* dx, int N, int H, int W, int C, int relu_mask,
                      void* stream) {
    AESR_CHECK_ARG(gout && x && dx && N > 0 && H >= 2 && W >= 2 && C % 4 == 0, "aesr_maxpool2_bwd: bad arguments");
    return aesr_launch_maxpool2_bwd(gout, x, gadd, dx, N, H, W, C, relu_mask, (hipStream_t)stream);
}

int aesr_scale_expand_fwd(const float* x, float* out4, size_t n, const float* ca_host, const float* cb_host, void* stream) {
    AESR_CHECK_ARG(x && out4 && ca_host && cb_host && n > 0 && n < ((size_t)1 << 29), "aesr_scale_expand_fwd: bad arguments");
    return aesr_launch_scale_expand(x, out4, (int)n, ca_host, cb_host, 0, (hipStream_t)stream);
}

int aesr_scale_expand_bwd(const float* d4, float* dx, size_t n, const float* ca_host, void* stream) {
    AESR_CHECK_ARG(d4 && dx && ca_host && n > 0 && n < ((size_t)1 << 29), "aesr_scale_expand_bwd: bad arguments");
    return aesr_launch_scale_expand(d4, dx, (int)n, ca_host, nullptr, 1, (hipStream_t)stream);
}

int aesr_lpips_tap_fwd(const float* f, const float* lin_w, float* partial, int B, int HW, int C, void* stream) {
    AESR_CHECK_ARG(f && lin_w && partial && B > 0 && HW > 0, "aesr_lpips_tap_fwd: bad arguments");
    return aesr_launch_lpips_tap_fwd(f, lin_w, partial, B, HW, C, (hipStream_t)stream);
}

int aesr_lpips_tap_bwd(const float* f, const float* lin_w, const float* gd, float* gf0, int B, int HW, int C, void* stream) {
    AESR_CHECK_ARG(f && lin_w && gd && gf0 && B > 0 && HW > 0, "aesr_lpips_tap_bwd: bad arguments");
    return aesr_launch_lpips_tap_bwd(f, lin_w, gd, gf0, B, HW, C, (hipStream_t)stream);
}

int aesr_lpips_finalize(const float* const* partials_host, const int* hw_host, int ntaps, float* d, int B, void* stream) {
    AESR_CHECK_ARG(partials_host && hw_host && d && ntaps >= 1 && ntaps <= 8 && B > 0, "aesr_lpips_finalize: bad arguments");
    return aesr_launch_lpips_finalize(partials_host, hw_host, ntaps, d, B, (hipStream_t)stream);
}

int aesr_space_to_depth2(const float* x, float* out, int N, int H, int W, int C, void* stream) {
    AESR_CHECK_ARG(x && out && N > 0 && H >= 2 && W >= 2 && C % 4 == 0, "aesr_space_to_depth2: bad arguments (C %% 4 == 0, H,W >= 2)");
    return aesr_launch_s2d(x, out, N, H, W, C, 0, (hipStream_t)stream);
}

int aesr_depth_to_space2(const float* g, float* dx, int N, int H, int W, int C, void* stream) {
    AESR_CHECK_ARG(g && dx && N > 0 && H >= 2 && W >= 2 && C % 4 == 0, "aesr_depth_to_space2: bad arguments (C %% 4 == 0, H,W >= 2)");
    return aesr_launch_s2d(g, dx, N, H, W, C, 1, (hipStream_t)stream);
}

int aesr_lerp_fwd(const float* z, const float* a_from, const float* a_to, float* zmix, int B, size_t per, void* stream) {
    AESR_CHECK_ARG(z && a_from && a_to && zmix && B > 0 && per % 4 == 0, "aesr_lerp_fwd: bad arguments (per %% 4 == 0)");
    return aesr_launch_lerp_fwd(z, a_from, a_to, zmix, B, per, (hipStream_t)stream);
}

int aesr_lerp_bwd(const float* dzmix, const float* a_from, const float* a_to, float* dz, int B, size_t per, void* stream) {
    AESR_CHECK_ARG(dzmix && a_from && a_to && dz && B > 0 && per % 4 == 0, "aesr_lerp_bwd: bad arguments (per %% 4 == 0)");
    return aesr_launch_lerp_bwd(dzmix, a_from, a_to, dz, B, per, (hipStream_t)stream);
}

int aesr_lerp_cat_fwd(const float* z, const float* a_from, const float* a_to, float* zcat, int B, size_t per, void* stream) {
    AESR_CHECK_ARG(z && a_from && a_to && zcat && z != zcat && B > 0 && per % 4 == 0, "aesr_lerp_cat_fwd: bad arguments (per %% 4 == 0)");
    return aesr_launch_lerp_cat_fwd(z, a_from, a_to, zcat, B, per, (hipStream_t)stream);
}

int aesr_lerp_multi(const float* z, float* out, int Z, size_t per_slice, const float* alphas_host, int n, int act, float slope,
                    void* stream) {
    AESR_CHECK_ARG(z && out && alphas_host && Z >= 2 && per_slice > 0 && per_slice % 4 == 0, "aesr_lerp_multi: bad arguments (two slices or more, per_slice %% 4 == 0)");
    AESR_CHECK_ARG(n >= 1 && n <= 16, "aesr_lerp_multi: %d mixing coefficients per call (1..16)", n);
    AESR_CHECK_ARG(act == ACT_NONE || act == ACT_RELU || (act == ACT_LRELU && slope >= 0.f && slope <= 1.f),
                   "aesr_lerp_multi: activation %d (none, ReLU, or LeakyReLU with a slope in [0, 1])", act);
    const float nslope = act == ACT_LRELU ? slope : (act == ACT_RELU ? 0.f : 1.f);
    return aesr_launch_lerp_multi(z, out, Z, per_slice, alphas_host, n, nslope, (hipStream_t)stream);
}

int aesr_interleave_clamp(const float* orig, const float* synth, float* out, int Z, int n, size_t per_slice, float lo, float hi, void* stream) {
    AESR_CHECK_ARG(orig && out && Z >= 1 && n >= 0 && per_slice > 0 && per_slice % 4 == 0 && (synth || n == 0 || Z == 1) && orig != out && synth != out,
                   "aesr_interleave_clamp: bad arguments (per_slice %% 4 == 0, out apart from its inputs)");
    AESR_CHECK_ARG(lo <= hi, "aesr_interleave_clamp: empty range [%g, %g]", (double)lo, (double)hi);
    return aesr_launch_interleave_clamp(orig, synth, out, Z, Z > 1 ? n : 0, per_slice, lo, hi, (hipStream_t)stream);
}

int aesr_lerp_cat_bwd(const float* g, const float* a_from, const float* a_to, float* dz, int B, size_t per, void* stream) {
    AESR_CHECK_ARG(g && a_from && a_to && dz && g != dz && B > 0 && per % 4 == 0, "aesr_lerp_cat_bwd: bad arguments (per %% 4 == 0)");
    return aesr_launch_lerp_cat_bwd(g, a_from, a_to, dz, B, per, (hipStream_t)stream);
}

int aesr_mse_fwd(const float* a, const float* b, double* partial, float* loss, size_t n, void* stream) {
    AESR_CHECK_ARG(a && b && partial && loss && n > 0, "aesr_mse_fwd: bad arguments");
    return aesr_launch_mse_fwd(a, b, partial, AESR_MSE_NPART, loss, n, (hipStream_t)stream);
}

int aesr_mse3_fwd(const float* a1, const float* b1, size_t n1, const float* a2, const float* b2, size_t n2, const float* a3,
                  const float* b3, size_t n3, const float* lam, double* workspace, float* out4, void* stream) {
    AESR_CHECK_ARG(a1 && b1 && n1 > 0 && a2 && b2 && n2 > 0 && (!a3 || (b3 && n3 > 0)) && lam && workspace && out4, "aesr_mse3_fwd: bad arguments");
    const float* a[3] = {a1, a2, a3};
    const float* b[3] = {b1, b2, a3 ? b3 : nullptr};
    const size_t n[3] = {n1, n2, a3 ? n3 : 1};
    return aesr_launch_mse3_fwd(a, b, n, lam, workspace, out4, (hipStream_t)stream);
}

int aesr_mse3_bwd(const float* a1, const float* b1, size_t n1, const float* a2, const float* b2, size_t n2, const float* lam,
                  const float* gloss, float* d1, float* d2, void* stream) {
    AESR_CHECK_ARG(a1 && b1 && n1 > 0 && a2 && b2 && n2 > 0 && lam && gloss && d1 && d2, "aesr_mse3_bwd: bad arguments");
    return aesr_launch_mse3_bwd(a1, b1, n1, a2, b2, n2, lam, gloss, d1, d2, (hipStream_t)stream);
}

int aesr_mse_bwd(const float* a, const float* b, const float* gloss, float* da, size_t n, void* stream) {
    AESR_CHECK_ARG(a && b && gloss && da && n > 0, "aesr_mse_bwd: bad arguments");
    return aesr_launch_mse_bwd(a, b, gloss, da, n, (hipStream_t)stream);
}

int aesr_l1_fwd(const float* a, const float* b, double* partial, float* loss, size_t n, void* stream) {
    AESR_CHECK_ARG(a && b && partial && loss && n > 0, "aesr_l1_fwd: bad arguments");
    return aesr_launch_l1_fwd(a, b, partial, AESR_MSE_NPART, loss, n, (hipStream_t)stream);
}

int aesr_l1_bwd(const float* a, const float* b, const float* gloss, float* da, size_t n, void* stream) {
    AESR_CHECK_ARG(a && b && gloss && da && n > 0, "aesr_l1_bwd: bad arguments");
    return aesr_launch_l1_bwd(a, b, gloss, da, n, (hipStream_t)stream);
}

int aesr_row_mean_fwd(const float* x, float* out, int N, size_t M, void* stream) {
    AESR_CHECK_ARG(x && out && N > 0 && M > 0, "aesr_row_mean_fwd: bad arguments");
    return aesr_launch_row_mean_fwd(x, out, N, M, (hipStream_t)stream);
}

int aesr_row_mean_bwd(const float* g, float* dx, int N, size_t M, void* stream) {
    AESR_CHECK_ARG(g && dx && N > 0 && M > 0, "aesr_row_mean_bwd: bad arguments");
    return aesr_launch_row_mean_bwd(g, dx, N, M, (hipStream_t)stream);
}

int aesr_lap_blur5(const float* in, const float* add, float* out, int P, int H, int W, float gain, int adjoint, void* stream) {
    AESR_CHECK_ARG(in && out && P > 0 && H >= 3 && W >= 3, "aesr_lap_blur5: bad arguments (reflect padding by 2 needs H, W >= 3)");
    AESR_CHECK_ARG(in != out, "aesr_lap_blur5: in-place filtering is not supported");
    return aesr_launch_lap_blur5(in, add, out, P, H, W, gain, adjoint ? 1 : 0, (hipStream_t)stream);
}

int aesr_lap_down2(const float* in, float* out, int P, int H, int W, void* stream) {
    AESR_CHECK_ARG(in && out && P > 0 && H > 0 && W > 0, "aesr_lap_down2: bad arguments");
    return aesr_launch_lap_down2(in, out, P, H, W, (hipStream_t)stream);
}

int aesr_lap_zero_insert2(const float* in, float* out, int P, int h, int w, int H, int W, void* stream) {
    AESR_CHECK_ARG(in && out && P > 0 && h > 0 && w > 0 && (H + 1) / 2 == h && (W + 1) / 2 == w,
                   "aesr_lap_zero_insert2: need h == ceil(H/2), w == ceil(W/2)");
    return aesr_launch_lap_zero_insert2(in, out, P, h, w, H, W, (hipStream_t)stream);
}

int aesr_act_bwd(const float* dout, const float* y, float* dpre, size_t n, int act, float slope, void* stream) {
    AESR_CHECK_ARG(dout && y && dpre && n > 0, "aesr_act_bwd: bad arguments");
    return aesr_launch_act_bwd(dout, y, dpre, n, act, slope, (hipStream_t)stream);
}

void aesr_adam_state_init(float* state_host8, double steps_done, double beta1, double beta2) {
    // the same chain of double multiplications the kernel runs step after step (NOT pow): an optimizer resumed from a checkpoint
    // continues bit for bit like the one that was never stopped
    double b1p = beta1, b2p = beta2;
    for (long k = 0; k < (long)steps_done; ++k) {
        b1p *= beta1;
        b2p *= beta2;
    }
    state_host8[0] = (float)steps_done;
    state_host8[1] = (float)(1.0 - b1p);
    state_host8[2] = (float)sqrt(1.0 - b2p);
    state_host8[3] = 0.f;
    memcpy(state_host8 + 4, &b1p, sizeof(double));
    memcpy(state_host8 + 6, &b2p, sizeof(double));
}

int aesr_adam_step(float* p, float* g, float* exp_avg, float* exp_avg_sq, float* state, size_t n, float lr, double beta1,
                   double beta2, float eps, float weight_decay, int zero_grad, void* stream) {
    AESR_CHECK_ARG(p && g && exp_avg && exp_avg_sq && state && n > 0, "aesr_adam_step: bad arguments");
    return aesr_launch_adam(p, g, exp_avg, exp_avg_sq, state, n, lr, beta1, beta2, eps, weight_decay, zero_grad, (hipStream_t)stream);
}

int aesr_triplet_assemble(const float* volumes, const aesr_triplet_desc* desc_host, int B, int width, float* image,
                          float* between, void* stream) {
    AESR_CHECK_ARG(volumes && desc_host && image && between && width > 0, "aesr_triplet_assemble: null pointer or empty shape");
    AESR_CHECK_ARG(B >= 1 && B <= TRIPLET_MAX, "aesr_triplet_assemble: B=%d must be 1..%d per call", B, TRIPLET_MAX);
    TripletTable t;
    memset(&t, 0, sizeof(t));
    for (int b = 0; b < B; ++b) {
        const aesr_triplet_desc& s = desc_host[b];
        AESR_CHECK_ARG(s.H > 0 && s.W > 0 && s.vol_off >= 0 && s.z_from >= 0 && s.z_to >= 0 && s.z_between >= 0 && s.k >= 0 && s.k <= 3,
                       "aesr_triplet_assemble: bad descriptor %d", b);
        TripletDesc& d = t.d[b];
        d.vol_off = s.vol_off; d.H = s.H; d.W = s.W; d.z_from = s.z_from; d.z_to = s.z_to; d.z_between = s.z_between;
        d.oy = s.oy; d.ox = s.ox; d.k = s.k; d.gain = s.gain; d.cutoff = s.cutoff;
    }
    return aesr_launch_triplet_assemble(volumes, t, B, width, image, between, (hipStream_t)stream);
}

size_t aesr_ssim_workspace_doubles(int Z, int H, int W) { return (size_t)Z * ceil_div(H, 16) * ceil_div(W, 16) * 2; }

int aesr_ssim_mse(const float* a, const float* b, double* workspace, double* ssim, double* mse, int Z, int H, int W, int win,
                  double data_range, double k1, double k2, void* stream) {
    AESR_CHECK_ARG(a && b && workspace && ssim && mse && Z > 0 && H > 0 && W > 0, "aesr_ssim_mse: null pointer or empty shape");
    AESR_CHECK_ARG(win >= 3 && win <= 11 && (win & 1) && win <= H && win <= W,
                   "aesr_ssim_mse: win=%d must be odd, 3..11 and not larger than the image (%dx%d)", win, H, W);
    AESR_CHECK_ARG(data_range > 0.0, "aesr_ssim_mse: data_range must be positive");
    return aesr_launch_ssim_mse(a, b, workspace, ssim, mse, Z, H, W, win, data_range, k1, k2, (hipStream_t)stream);
}

// (the shapes aesr_vif_mscale takes; anything else has no workspace: the layout arithmetic is int -- sanitizer sweep, round 6)
size_t aesr_vif_workspace_bytes(int Z, int H, int W) {
    return (Z > 0 && Z <= 65535 && H > 0 && W > 0 && (size_t)H * W < ((size_t)1 << 30)) ? aesr_vif_workspace_bytes_impl(Z, H, W) : 0;
}

int aesr_vif_mscale(const float* ref, const float* dist, void* workspace, double* vif, int Z, int H, int W, const double* weights_host,
                    const int* radii_host, double sigma_nsq, void* stream) {
    AESR_CHECK_ARG(ref && dist && workspace && vif && weights_host && radii_host, "aesr_vif_mscale: null pointer");
    AESR_CHECK_ARG(Z > 0 && Z <= 65535 && H > 0 && W > 0 && (size_t)H * W < ((size_t)1 << 30), "aesr_vif_mscale: unsupported shape %d x %d x %d", Z, H, W);
    AESR_CHECK_ARG(sigma_nsq > 0.0, "aesr_vif_mscale: sigma_nsq must be positive");
    return aesr_launch_vif_mscale(ref, dist, workspace, vif, Z, H, W, weights_host, radii_host, sigma_nsq, (hipStream_t)stream);
}

}  // extern "C"
