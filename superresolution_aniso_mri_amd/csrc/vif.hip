// VIF (visual information fidelity, pixel domain, four scales) of the model-selection / evaluation protocol on the device:
// evaluate/metrics.py:65-109 ``compute_vif_for_batch`` -> evaluate/vifvec.py:7-63 ``vifp_mscale`` of the reference, per slice of two
// volumes [Z][H][W].  The reference converts BOTH volumes to uint8 first (metrics.py:72-73) and numpy / scipy.ndimage then keep that
// dtype, so the score its result files carry is the one of this arithmetic, which the kernels follow operation by operation
// (restated on the CPU in oracle/vif_oracle.py, pinned by tests/golden/vif.npz = outputs of the reference's own functions):
//   q(x)      = (uint8) trunc(clamp(x * 255.f, 0, 255))                                   [fp32 product, C cast]
//   G_sd(u)   = scipy.ndimage.gaussian_filter on uint8: along axis 0, then axis 1, each pass in double
//                 tmp = u[l] w[r];  for ii = -r .. -1:  tmp += (u[l + ii] + u[l - ii]) w[ii + r]        (this summation order)
//               with the 'reflect' boundary (d c b a | a b c d | d c b a), and each pass stored back as uint8 by TRUNCATION
//   scale s   : sd = (2^(5-s) + 1) / 5, r = int(4 sd + 0.5); s > 1: ref, dist <- G_sd(.)[::2, ::2]
//               mu1 = G(ref), mu2 = G(dist), s11 = G(ref ref), s22 = G(dist dist), s12 = G(ref dist)      products modulo 256
//               sigma1_sq = s11 - mu1 mu1, sigma2_sq = s22 - mu2 mu2, sigma12 = s12 - mu1 mu2              modulo 256
//               g, sv_sq and the masks of vifvec.py:39-53 in double;  num += sum log10(1 + g g sigma1_sq / (sv_sq + nsq)),
//               den += sum log10(1 + sigma1_sq / nsq)
//   vif[z]    = num / den  (NaN when den == 0)
// Everything in front of the logarithms is integer-exact, so the result differs from the reference's only by the rounding of log10
// and of the fp64 sums (1e-15 relative).  The filter weights arrive from the HOST (computed there with numpy exactly as scipy's
// _gaussian_kernel1d does): a truncation is sensitive to the last bit of a weight wherever the image is constant.
// Bandwidth-trivial (a 160 x 160 slice is 25 KB as uint8; everything stays in L2): plain one-thread-per-output kernels, fixed-order
// block sums, no atomics (bitwise reproducible).
#include "aesr_kernels.h"

#define VIF_MAXR 14
struct VifW { double w[2 * VIF_MAXR + 1]; int r; };

__device__ __forceinline__ int vif_reflect(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

__global__ __launch_bounds__(256) void vif_quant_kernel(const float* __restrict__ a, const float* __restrict__ b, unsigned char* __restrict__ qa,
                                                        unsigned char* __restrict__ qb, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = fminf(fmaxf(a[i] * 255.f, 0.f), 255.f), y = fminf(fmaxf(b[i] * 255.f, 0.f), 255.f);
        qa[i] = (unsigned char)(int)x;
        qb[i] = (unsigned char)(int)y;
    }
}

// Pass along axis 0 (rows).  MOM: out = 5 planes [5][Z][h][w] of G0(ref), G0(dist), G0(ref ref), G0(dist dist), G0(ref dist) at every
// row; else out = 2 planes [2][Z][ho][w] of G0(ref), G0(dist) at the EVEN rows (the rows ``[::2]`` keeps).
template <bool MOM>
__global__ __launch_bounds__(256) void vif_vpass_kernel(const unsigned char* __restrict__ ref, const unsigned char* __restrict__ dist,
                                                        unsigned char* __restrict__ out, int Z, int h, int w, int ho, VifW fw) {
    const size_t per = (size_t)ho * w, n = (size_t)Z * per;
    const int r = fw.r;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int z = (int)(i / per);
        const int rem = (int)(i - (size_t)z * per);
        const int yo = rem / w, x = rem - yo * w;
        const int y = MOM ? yo : 2 * yo;
        const unsigned char* pr = ref + (size_t)z * h * w + x;
        const unsigned char* pd = dist + (size_t)z * h * w + x;
        const unsigned cr = pr[(size_t)y * w], cd = pd[(size_t)y * w];
        double t0 = (double)cr * fw.w[r], t1 = (double)cd * fw.w[r], t2 = 0.0, t3 = 0.0, t4 = 0.0;
        if (MOM) {
            t2 = (double)((cr * cr) & 255u) * fw.w[r];
            t3 = (double)((cd * cd) & 255u) * fw.w[r];
            t4 = (double)((cr * cd) & 255u) * fw.w[r];
        }
        for (int ii = -r; ii < 0; ++ii) {
            const int ya = vif_reflect(y + ii, h), yb = vif_reflect(y - ii, h);
            const unsigned ra = pr[(size_t)ya * w], rb = pr[(size_t)yb * w], da = pd[(size_t)ya * w], db = pd[(size_t)yb * w];
            const double wk = fw.w[ii + r];
            t0 += ((double)ra + (double)rb) * wk;
            t1 += ((double)da + (double)db) * wk;
            if (MOM) {
                t2 += ((double)((ra * ra) & 255u) + (double)((rb * rb) & 255u)) * wk;
                t3 += ((double)((da * da) & 255u) + (double)((db * db) & 255u)) * wk;
                t4 += ((double)((ra * da) & 255u) + (double)((rb * db) & 255u)) * wk;
            }
        }
        out[i] = (unsigned char)(int)t0;
        out[n + i] = (unsigned char)(int)t1;
        if (MOM) {
            out[2 * n + i] = (unsigned char)(int)t2;
            out[3 * n + i] = (unsigned char)(int)t3;
            out[4 * n + i] = (unsigned char)(int)t4;
        }
    }
}

__device__ __forceinline__ unsigned char vif_hfilter(const unsigned char* __restrict__ row, int x, int w, const VifW& fw) {
    const int r = fw.r;
    double t = (double)row[x] * fw.w[r];
    for (int ii = -r; ii < 0; ++ii)
        t += ((double)row[vif_reflect(x + ii, w)] + (double)row[vif_reflect(x - ii, w)]) * fw.w[ii + r];
    return (unsigned char)(int)t;
}

// Pass along axis 1 of the down-sampling filter: in = 2 planes [2][Z][ho][w] -> ref', dist' [Z][ho][wo] at the even columns.
__global__ __launch_bounds__(256) void vif_hdown_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ ref, unsigned char* __restrict__ dist,
                                                        int Z, int ho, int w, int wo, VifW fw) {
    const size_t n = (size_t)Z * ho * wo, plane = (size_t)Z * ho * w;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t zy = i / wo;
        const int xo = (int)(i - zy * wo);
        ref[i] = vif_hfilter(in + zy * w, 2 * xo, w, fw);
        dist[i] = vif_hfilter(in + plane + zy * w, 2 * xo, w, fw);
    }
}

// Pass along axis 1 of the five moment planes + the per-pixel terms of vifvec.py:31-57, summed per block (fixed order) into
// partial[(z * nblk + block) * 2 + {0: numerator, 1: denominator}].  grid = (nblk, Z), a block covers 256 consecutive pixels of a slice.
__global__ __launch_bounds__(256) void vif_hstat_kernel(const unsigned char* __restrict__ v5, double* __restrict__ partial, int Z, int h, int w,
                                                        double sigma_nsq, VifW fw) {
    __shared__ double red[2][256];
    const int z = blockIdx.y, px = blockIdx.x * 256 + threadIdx.x;
    const size_t plane = (size_t)Z * h * w;
    double num = 0.0, den = 0.0;
    if (px < h * w) {
        const int y = px / w, x = px - y * w;
        const unsigned char* row = v5 + ((size_t)z * h + y) * w;
        const unsigned m1 = vif_hfilter(row, x, w, fw), m2 = vif_hfilter(row + plane, x, w, fw);
        const unsigned s11 = vif_hfilter(row + 2 * plane, x, w, fw), s22 = vif_hfilter(row + 3 * plane, x, w, fw), s12 = vif_hfilter(row + 4 * plane, x, w, fw);
        const double eps = 1e-10;
        const double s1 = (double)((s11 - m1 * m1) & 255u), s2 = (double)((s22 - m2 * m2) & 255u), c = (double)((s12 - m1 * m2) & 255u);
        double g = c / (s1 + eps);
        double sv = s2 - g * c;
        if (s1 < eps) { g = 0.0; sv = s2; }
        if (s2 < eps) { g = 0.0; sv = 0.0; }
        if (g < 0.0) { sv = s2; g = 0.0; }
        if (sv <= eps) sv = eps;
        num = log10(1.0 + g * g * s1 / (sv + sigma_nsq));
        den = log10(1.0 + s1 / sigma_nsq);
    }
    red[0][threadIdx.x] = num;
    red[1][threadIdx.x] = den;
    __syncthreads();
    for (int hh = 128; hh > 0; hh >>= 1) {
        if ((int)threadIdx.x < hh) {
            red[0][threadIdx.x] += red[0][threadIdx.x + hh];
            red[1][threadIdx.x] += red[1][threadIdx.x + hh];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[((size_t)z * gridDim.x + blockIdx.x) * 2 + 0] = red[0][0];
        partial[((size_t)z * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
    }
}

struct VifFinish { size_t off[4]; int nblk[4]; };     // per scale: first double of its partial sums, blocks per slice

__global__ __launch_bounds__(256) void vif_finish_kernel(const double* __restrict__ partial, double* __restrict__ vif, VifFinish f) {
    __shared__ double red[2][256];
    const int z = blockIdx.x;
    double num = 0.0, den = 0.0;
    for (int s = 0; s < 4; ++s) {          // scale after scale, as the reference accumulates
        double a = 0.0, b = 0.0;
        for (int t = threadIdx.x; t < f.nblk[s]; t += 256) {
            a += partial[f.off[s] + ((size_t)z * f.nblk[s] + t) * 2 + 0];
            b += partial[f.off[s] + ((size_t)z * f.nblk[s] + t) * 2 + 1];
        }
        red[0][threadIdx.x] = a;
        red[1][threadIdx.x] = b;
        __syncthreads();
        for (int hh = 128; hh > 0; hh >>= 1) {
            if ((int)threadIdx.x < hh) {
                red[0][threadIdx.x] += red[0][threadIdx.x + hh];
                red[1][threadIdx.x] += red[1][threadIdx.x + hh];
            }
            __syncthreads();
        }
        num += red[0][0];
        den += red[1][0];
        __syncthreads();
    }
    if (threadIdx.x == 0) vif[z] = den != 0.0 ? num / den : __builtin_nan("");
}

// workspace (bytes, 8-aligned pieces): ref / dist uint8 pyramids, the filter intermediate (5 planes at full size), the partial sums
struct VifLayout { size_t q[4][2], tmp, partial, total; VifFinish fin; int h[4], w[4]; };

static VifLayout vif_layout(int Z, int H, int W) {
    VifLayout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 7) & ~(size_t)7; return o; };
    int h = H, w = W;
    for (int s = 0; s < 4; ++s) {
        L.h[s] = h; L.w[s] = w;
        L.q[s][0] = take((size_t)Z * h * w);
        L.q[s][1] = take((size_t)Z * h * w);
        h = (h + 1) / 2; w = (w + 1) / 2;
    }
    L.tmp = take((size_t)5 * Z * H * W);
    L.partial = take(0);
    size_t d = 0;
    for (int s = 0; s < 4; ++s) {
        L.fin.nblk[s] = ceil_div(L.h[s] * L.w[s], 256);
        L.fin.off[s] = d;
        d += (size_t)Z * L.fin.nblk[s] * 2;
    }
    take(d * sizeof(double));
    L.total = off;
    return L;
}

size_t aesr_vif_workspace_bytes_impl(int Z, int H, int W) { return vif_layout(Z, H, W).total; }

int aesr_launch_vif_mscale(const float* ref, const float* dist, void* workspace, double* vif, int Z, int H, int W, const double* weights,
                           const int* radii, double sigma_nsq, hipStream_t st) {
    const VifLayout L = vif_layout(Z, H, W);
    unsigned char* ws = (unsigned char*)workspace;
    VifW fw[4];
    const double* wp = weights;
    for (int s = 0; s < 4; ++s) {
        if (radii[s] < 0 || radii[s] > VIF_MAXR) {
            aesr_set_error("aesr_vif_mscale: filter radius %d of scale %d outside 0..%d", radii[s], s + 1, VIF_MAXR);
            return AESR_ERR_ARG;
        }
        fw[s].r = radii[s];
        for (int k = 0; k < 2 * radii[s] + 1; ++k) fw[s].w[k] = wp[k];
        for (int k = 2 * radii[s] + 1; k < 2 * VIF_MAXR + 1; ++k) fw[s].w[k] = 0.0;
        wp += 2 * radii[s] + 1;
    }
    auto grid = [](size_t n) { const size_t g = (n + 255) / 256; return dim3((unsigned)(g < 4096 ? (g ? g : 1) : 4096)); };
    const size_t n0 = (size_t)Z * H * W;
    hipLaunchKernelGGL(vif_quant_kernel, grid(n0), dim3(256), 0, st, ref, dist, ws + L.q[0][0], ws + L.q[0][1], n0);
    AESR_LAUNCH_CHECK("vif_quant");
    double* partial = (double*)(ws + L.partial);
    for (int s = 0; s < 4; ++s) {
        const int h = L.h[s], w = L.w[s];
        if (s > 0) {
            const int hp = L.h[s - 1], wpv = L.w[s - 1];
            hipLaunchKernelGGL(vif_vpass_kernel<false>, grid((size_t)Z * h * wpv), dim3(256), 0, st, ws + L.q[s - 1][0], ws + L.q[s - 1][1], ws + L.tmp, Z, hp,
                               wpv, h, fw[s]);
            AESR_LAUNCH_CHECK("vif_vpass(down)");
            hipLaunchKernelGGL(vif_hdown_kernel, grid((size_t)Z * h * w), dim3(256), 0, st, ws + L.tmp, ws + L.q[s][0], ws + L.q[s][1], Z, h, wpv, w, fw[s]);
            AESR_LAUNCH_CHECK("vif_hdown");
        }
        hipLaunchKernelGGL(vif_vpass_kernel<true>, grid((size_t)Z * h * w), dim3(256), 0, st, ws + L.q[s][0], ws + L.q[s][1], ws + L.tmp, Z, h, w, h, fw[s]);
        AESR_LAUNCH_CHECK("vif_vpass(moments)");
        hipLaunchKernelGGL(vif_hstat_kernel, dim3(L.fin.nblk[s], Z), dim3(256), 0, st, ws + L.tmp, partial + L.fin.off[s], Z, h, w, sigma_nsq, fw[s]);
        AESR_LAUNCH_CHECK("vif_hstat");
    }
    hipLaunchKernelGGL(vif_finish_kernel, dim3(Z), dim3(256), 0, st, partial, vif, L.fin);
    AESR_LAUNCH_CHECK("vif_finish");
    return AESR_OK;
}
