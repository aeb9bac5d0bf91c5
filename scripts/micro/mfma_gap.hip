// Micro-benchmark: ONE wave per SIMD, 64 independent v_mfma_f32_16x16x4_f32 accumulators (256 AGPRs) issued round robin, NF filler
// instructions of one kind behind every MFMA -- how many hide in the 32-cycle gap of this MFMA?
//   kind 0: independent v_add_f32     kind 1: dependent v_add_f32 chain     kind 2: ds_read_b32 (waited for once per 64 MFMAs)
//   kind 3: s_add_i32 (scalar)        kind 4: v_mul_i32_i24                 kind 5: v_sub_f32 reading the previous ds_read (lgkmcnt waits)
// Each gap's fillers stand in ONE asm statement: between separate asm statements the compiler puts an s_nop (4 more cycles).
// build + run:  hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_gap.hip -o /tmp/mfma_gap && /tmp/mfma_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NF, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a0 = lane * 0.001f, b0 = 1.0f + lane * 0.002f;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    int si = 1, vi = lane;
    const unsigned lp = (unsigned)(unsigned long long)(lds + lane);     // LDS byte address (low half of the flat address)
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a0), "v"(b0));
            if constexpr (KIND == 0 && NF > 0) {
                asm volatile(".rept %8\n\tv_add_f32 %0, %6, %7\n\tv_add_f32 %1, %6, %7\n\t.endr\n\t.rept %9\n\tv_add_f32 %2, %6, %7\n\t.endr"
                             : "=v"(f[0]), "=v"(f[1]), "=v"(f[2]), "=v"(f[3]), "=v"(f[4]), "=v"(f[5]) : "v"(a0), "v"(b0), "i"(NF / 2), "i"(NF & 1));
            }
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                if constexpr (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[0]) : "v"(b0));
                if constexpr (KIND == 2) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[n]) : "v"(lp), "i"(n * 256));
                if constexpr (KIND == 3) asm volatile("s_add_i32 %0, %0, 1" : "+s"(si));
                if constexpr (KIND == 4) asm volatile("v_mul_i32_i24 %0, %1, %1" : "=v"(vi) : "v"(lane));
                if constexpr (KIND == 5) {
                    if (n & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\tv_sub_f32 %0, %1, %2" : "=v"(f[n]) : "v"(f[n - 1]), "v"(b0));
                    else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[n]) : "v"(lp), "i"(n * 256));
                }
            }
        }
        if constexpr (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) s += acc[j][0];
    for (int n = 0; n < 8; ++n) s += f[n];
    out[blockIdx.x * 256 + threadIdx.x] = s + si + vi;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// G MFMAs back to back, then G * NF fillers back to back (same totals as k<NF, KIND>): does grouping change what a filler costs?
//   kind 0: v_add_f32    kind 6: v_pk_add_f32 (two adds per instruction)    kind 7: no MFMA at all, fillers only (their own rate)
template <int NF, int KIND, int G>
__global__ __launch_bounds__(256, 1) void kg(float* out, long long* cyc, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    f32x4 acc[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a0 = lane * 0.001f, b0 = 1.0f + lane * 0.002f;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}}, pa = {a0, b0};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j0 = 0; j0 < 64; j0 += G) {
            if constexpr (KIND != 7) {
#pragma unroll
                for (int j = j0; j < j0 + G; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a0), "v"(b0));
            }
            if constexpr (KIND == 0 || KIND == 7)
                asm volatile(".rept %4\n\tv_add_f32 %0, %2, %3\n\tv_add_f32 %1, %2, %3\n\t.endr" : "=v"(f[0]), "=v"(f[1]) : "v"(a0), "v"(b0), "i"(NF * G / 2));
            if constexpr (KIND == 6)
                asm volatile(".rept %3\n\tv_pk_add_f32 %0, %2, %2\n\tv_pk_add_f32 %1, %2, %2\n\t.endr" : "=v"(p[0]), "=v"(p[1]) : "v"(pa), "i"(NF * G / 2));
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 64; ++j) s += acc[j][0];
    for (int n = 0; n < 8; ++n) s += f[n];
    for (int n = 0; n < 4; ++n) s += p[n][0] + p[n][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NF, int KIND, int G>
static void rung(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipFuncSetAttribute((const void*)kg<NF, KIND, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 400;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((kg<NF, KIND, G>), dim3(256), dim3(256), 160 * 1024, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    const double per = s / 256 / iters / 64;
    printf("%-24s groups of %2d MFMA + %3d fillers: %.1f cycles per MFMA (+%.1f per filler)\n", name, G, NF * G, per, (per - (KIND == 7 ? 0 : 32.1)) / NF);
    (void)hipFree(out); (void)hipFree(cyc);
}

// TWO waves per SIMD (512 threads, 32 accumulators each): do one wave's fillers run beside the OTHER wave's MFMA?
template <int NF>
__global__ __launch_bounds__(512, 1) void k2(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a0 = lane * 0.001f, b0 = 1.0f + lane * 0.002f;
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(a0), "v"(b0));
            if constexpr (NF > 0)
                asm volatile(".rept %4\n\tv_add_f32 %0, %2, %3\n\t.endr\n\t.rept %5\n\tv_add_f32 %1, %2, %3\n\t.endr" : "=v"(f[0]), "=v"(f[1]) : "v"(a0), "v"(b0), "i"((NF + 1) / 2), "i"(NF / 2));
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += acc[j][0];
    for (int n = 0; n < 8; ++n) s += f[n];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NF>
static void run2() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipFuncSetAttribute((const void*)k2<NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 800;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k2<NF>), dim3(256), dim3(512), 160 * 1024, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    static long long h[2048]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0, smax = 0;
    for (int i = 0; i < 256; ++i) { long long m = 0; for (int w = 0; w < 8; ++w) { s += h[i * 8 + w]; if (h[i * 8 + w] > m) m = h[i * 8 + w]; } smax += m; }
    const double per = s / 2048 / iters / 32, pmax = smax / 256 / iters / 32;
    printf("two waves per SIMD, %d v_add_f32 per gap: mean %.1f, slowest wave %.1f cycles per MFMA of a wave = %.1f per MFMA of the SIMD (%.0f%% of the pipe); one wave alone would take %.1f\n",
           NF, per, pmax, pmax / 2, 6400.0 / pmax, 32.0 + (NF ? 4.3 + 4.0 * NF : 0.0));
    (void)hipFree(out); (void)hipFree(cyc);
}

template <int NF, int KIND>
static void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    (void)hipFuncSetAttribute((const void*)k<NF, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 400;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NF, KIND>), dim3(256), dim3(256), 160 * 1024, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, KIND>), dim3(256), dim3(256), 160 * 1024, 0, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    const double per = s / 256 / iters / 64;
    printf("%-28s NF=%d: %.1f cycles per MFMA (%.0f%% of the pipe), %.3f ms, clock %.2f GHz\n", name, NF, per, 3200.0 / per, ms, s / 256 / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0, 0>("no filler");
    run<2, 0>("independent v_add_f32"); run<4, 0>("independent v_add_f32"); run<5, 0>("independent v_add_f32"); run<6, 0>("independent v_add_f32"); run<7, 0>("independent v_add_f32"); run<8, 0>("independent v_add_f32");
    run<4, 1>("dependent v_add_f32 chain"); run<6, 1>("dependent v_add_f32 chain");
    run<2, 2>("ds_read_b32"); run<4, 2>("ds_read_b32");
    run<4, 3>("s_add_i32"); run<8, 3>("s_add_i32");
    run<4, 4>("v_mul_i32_i24"); run<6, 4>("v_mul_i32_i24");
    run<2, 5>("ds_read + waited v_sub"); run<4, 5>("ds_read + waited v_sub");
    run2<0>(); run2<1>(); run2<2>(); run2<3>(); run2<4>(); run2<6>();
    rung<4, 7, 1>("v_add_f32 alone");
    rung<4, 0, 1>("v_add_f32"); rung<4, 0, 2>("v_add_f32"); rung<4, 0, 4>("v_add_f32"); rung<4, 0, 8>("v_add_f32"); rung<4, 0, 16>("v_add_f32"); rung<4, 0, 64>("v_add_f32");
    rung<2, 0, 4>("v_add_f32"); rung<2, 0, 16>("v_add_f32");
    rung<2, 6, 1>("v_pk_add_f32"); rung<2, 6, 4>("v_pk_add_f32"); rung<2, 6, 16>("v_pk_add_f32"); rung<4, 6, 16>("v_pk_add_f32");
    return 0;
}
