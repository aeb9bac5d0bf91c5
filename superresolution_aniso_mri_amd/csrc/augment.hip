// On-device batch assembly + augmentation of the ae_combined training triplets: replaces the
// per-sample numpy pipeline of the reference's DataLoader workers (train_cardiac_aesr.py:83-96 ->
// datasets/shared_transforms.py AdjustToPatchSize, CenterCrop, RandomCrop, RandomIntensity, RandomRotation; batch layout of
// datasets/ACDC/data4d_simple.py:327-355) with ONE launch over a device-resident volume cache:
//   out[b][s][i][j] = sigm( gain_b * (src_b[z_s][oy_b + u][ox_b + v] - cutoff_b) ),   (u, v) = rot90^-k (i, j),
// src = 0 outside the slice (the zero padding of AdjustToPatchSize goes through the intensity curve like in the reference),
// s = from / to / between;  `image` receives all "from" slices then all "to" slices ([2B,1,W,W]), `between` [B,1,W,W].
// The random numbers (crop origin, gain, cutoff, k) are drawn on the host in the reference's order and travel as a by-value table.
#include "aesr_kernels.h"

__global__ __launch_bounds__(256) void triplet_assemble_kernel(const float* __restrict__ vol, TripletTable t, int B, int W,
                                                               float* __restrict__ image, float* __restrict__ between) {
    const int b = blockIdx.z, s = blockIdx.y;
    const TripletDesc d = t.d[b];
    const int z = s == 0 ? d.z_from : (s == 1 ? d.z_to : d.z_between);
    const float* src = vol + d.vol_off + (size_t)z * d.H * d.W;
    float* dst = (s == 2 ? between + (size_t)b * W * W : image + (size_t)(s * B + b) * W * W);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < W * W; p += gridDim.x * 256) {
        const int i = p / W, j = p - i * W;
        int u, v;                                   // np.rot90(X, k, axes=(1, 2)): out[i][j] = X[u][v]
        if (d.k == 0) { u = i; v = j; }
        else if (d.k == 1) { u = j; v = W - 1 - i; }
        else if (d.k == 2) { u = W - 1 - i; v = W - 1 - j; }
        else { u = W - 1 - j; v = i; }
        const int y = d.oy + u, x = d.ox + v;
        float val = 0.f;
        if (y >= 0 && y < d.H && x >= 0 && x < d.W) val = src[(size_t)y * d.W + x];
        dst[p] = 1.f / (1.f + expf(d.gain * (d.cutoff - val)));
    }
}

int aesr_launch_triplet_assemble(const float* vol, const TripletTable& t, int B, int W, float* image, float* between,
                                 hipStream_t st) {
    int gx = ceil_div(W * W, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(triplet_assemble_kernel, dim3(gx, 3, B), dim3(256), 0, st, vol, t, B, W, image, between);
    AESR_LAUNCH_CHECK("triplet_assemble");
    return AESR_OK;
}
