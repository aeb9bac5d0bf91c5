// Micro-benchmark: what a write-only kernel reaches on MI355X as a function of WHICH bytes a workgroup writes (HBM-cold: a 600 MB scrub between
// repetitions).  Buffer = N x H x W x 32 floats (NHWC, 128 B per pixel) = the output of the thin "expand" kernels (conv_thin.hip).
//   pattern 0: linear -- workgroup b writes 64 KB contiguous chunks, grid-stride (what a fill does)
//   pattern 1: conv_thin.hip's tiles -- a 256-thread workgroup owns 16 rows x 32 columns: 16 pieces of 4 KB, one image row (W x 128 B) apart
//   pattern 2: whole rows -- a workgroup owns 2 image rows (2 x W x 128 B contiguous) and walks them in 4 KB pieces
//   pattern 3: as 1 with 8-row tiles      pattern 4: as 1, but the workgroups of a tile ROW are issued with consecutive block indices last
// build + run:  hipcc -O3 --offload-arch=gfx950 scripts/micro/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float* out, int N, int H, int W, int pattern, float* scr, size_t nscr) {
    const int tid = threadIdx.x, c4 = tid & 7, pl = tid >> 3;
    const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
    if (pattern == 99) {            // scrub
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < nscr / 4; i += (size_t)gridDim.x * 256) ((f32x4*)scr)[i] = v;
        return;
    }
    const size_t total4 = (size_t)N * H * W * 8;
    if (pattern == 0) {
        const size_t chunk4 = 4096;          // 64 KB
        for (size_t c0 = (size_t)blockIdx.x * chunk4; c0 < total4; c0 += (size_t)gridDim.x * chunk4)
            for (size_t i = c0 + tid; i < c0 + chunk4 && i < total4; i += 256) ((f32x4*)out)[i] = v;
        return;
    }
    if (pattern == 2) {
        const int rows = N * H, r0 = blockIdx.x * 2;
        for (int r = r0; r < r0 + 2 && r < rows; ++r)
            for (int x0 = 0; x0 < W; x0 += 32) {
                const int x = x0 + pl;
                if (x < W) ((f32x4*)out)[((size_t)r * W + x) * 8 + c4] = v;
            }
        return;
    }
    const int TH = pattern == 3 ? 8 : 16;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH;
    int tile = blockIdx.x;
    int tx, ty, n;
    if (pattern == 4) { n = tile % N; tile /= N; tx = tile % tiles_x; ty = tile / tiles_x; }
    else { tx = tile % tiles_x; tile /= tiles_x; ty = tile % tiles_y; n = tile / tiles_y; }
    const int x = tx * 32 + pl;
    if (x >= W) return;
    for (int r = 0; r < TH; ++r) {
        const int y = ty * TH + r;
        if (y >= H) break;
        ((f32x4*)out)[(((size_t)n * H + y) * W + x) * 8 + c4] = v;
    }
}

int main() {
    const int N = 36, H = 162, W = 162;
    const size_t bytes = (size_t)N * H * W * 32 * 4, nscr = 150u * 1000 * 1000;
    float *out, *scr;
    hipMalloc(&out, bytes);
    hipMalloc(&scr, nscr * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int pattern : {0, 1, 2, 3, 4, 0, 1}) {
        int grid;
        if (pattern == 0) grid = 2048;
        else if (pattern == 2) grid = (N * H + 1) / 2;
        else grid = N * ((H + (pattern == 3 ? 7 : 15)) / (pattern == 3 ? 8 : 16)) * ((W + 31) / 32);
        std::vector<float> ts;
        for (int rep = 0; rep < 9; ++rep) {
            hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, out, N, H, W, 99, scr, nscr);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, N, H, W, pattern, scr, nscr);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        printf("pattern %d  grid %5d  %6.1f us  %5.2f TB/s written (%.0f MB)\n", pattern, grid, ts[4], bytes / ts[4] / 1e6, bytes / 1e6);
    }
    return 0;
}
