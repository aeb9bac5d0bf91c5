// Micro-benchmark: what does a v_mfma_f32_16x16x4_f32 stream of the igemm kernel's shape sustain?
//   variant 0: pure MFMA, 8 independent accumulators
//   variant 1: 2 accumulators alternating (the NB=2 inner pattern), operands in registers
//   variant 2: variant 1 + one ds_read_b128 per 8 MFMAs (software-pipelined like the kernel)
//   variant 3: variant 2 with 16x16 MFMA replaced by 32x32x2 (64-cycle) for comparison
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    float a0 = lane * 0.001f, b0 = 1.0f + lane * 0.002f;
    if constexpr (V == 0) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[j], 0, 0, 0);
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if constexpr (V == 1) {
        f32x4 acc[2] = {(f32x4){0, 0, 0, 0}, (f32x4){0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[j], 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1];
    } else if constexpr (V == 2) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        const float* base = lds + lane * 20;
        f32x4 acur = *(const f32x4*)base;
        f32x4 bq = (f32x4){b0, b0 + 1, b0 + 2, b0 + 3};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                f32x4 anxt = *(const f32x4*)(base + ((it * 4 + blk + 1) & 7) * 1280);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[blk * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[r], acur[r], acc[blk * 2 + j], 0, 0, 0);
                acur = anxt;
            }
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if constexpr (V == 4) {
        // igemm-like chunk structure: [barrier, 6 x ds_write_b128, barrier, 36 block-taps x (ds_read_b128 + 8 MFMA)]
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        const float* base = lds + lane * 20;
        f32x4 bq = (f32x4){b0, b0 + 1, b0 + 2, b0 + 3};
        f32x4 R = (f32x4){a0, a0, a0, a0};
        for (int it = 0; it < iters / 9; ++it) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 6; ++j) *(f32x4*)(lds + ((threadIdx.x + 512 * j) & 4095) * 4) = R;
            __syncthreads();
            f32x4 acur = *(const f32x4*)base;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    f32x4 anxt = *(const f32x4*)(base + ((tap * 4 + blk + 1) & 7) * 1280);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[blk * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[r], acur[r], acc[blk * 2 + j], 0, 0, 0);
                    acur = anxt;
                }
            }
            R = acc[0];
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else if constexpr (V >= 5 && V <= 9) {
        // igemm-like chunk (NB = 4, MBW = 2): [barrier, 11 ds_write_b128, barrier, 18 block-taps x (ds_read_b128 + 16 MFMA + EXTRA)]
        // EXTRA per block-tap: V5 none, V6 12 VALU, V7 12 VALU + 6 SALU + 1 uniform not-taken branch, V8 24 VALU, V9 like V7 + buffer load/store pair
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
        const float* base = lds + lane * 20;
        f32x4 bq[4];
        for (int j = 0; j < 4; ++j) bq[j] = (f32x4){b0 + j, b0 + 1, b0 + 2, b0 + 3};
        f32x4 R = (f32x4){a0, a0, a0, a0};
        int x0 = lane, x1 = lane * 3;
        for (int it = 0; it < iters / 9; ++it) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 11; ++j) *(f32x4*)(lds + ((threadIdx.x + 512 * j) & 4095) * 4) = R;
            __syncthreads();
            f32x4 acur = *(const f32x4*)base;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    f32x4 anxt = *(const f32x4*)(base + ((tap * 2 + blk + 1) & 7) * 1280);
                    if (V == 6 || V == 7 || V == 8 || V == 9) {
#pragma unroll
                        for (int e = 0; e < (V == 8 ? 24 : 12); ++e) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                    }
                    if (V == 7 || V == 9) {
                        asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
                        if (iters == 12345) x1 += 1;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[blk * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[j][r], acur[r], acc[blk * 4 + j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    acur = anxt;
                }
            }
            R = acc[0];
            R[0] += (float)x0;
        }
        float s = 0;
        for (int j = 0; j < 8; ++j) s += acc[j][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x16 acc[2];
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[j][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[j], 0, 0, 0);
        }
        out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1];
    }
}

template <int V>
void run(const char* name, double flop_per_iter_per_wave, int grid = 1024) {
    float* out;
    hipMalloc(&out, 1024 * 512 * 4);
    const int iters = 4000;       // grid 1024: 2 x 8-wave workgroups per CU -> 4 waves / SIMD; 256: 2 waves / SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<V><<<grid, 512>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<V><<<grid, 512>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = flop_per_iter_per_wave * (V >= 4 ? (iters / 9) * 9 : iters) * grid * 8;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, fl / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    run<0>("pure MFMA 16x16x4, 8 accumulators", 32.0 * 2048);
    run<1>("pure MFMA 16x16x4, 2 alternating accumulators", 32.0 * 2048);
    run<2>("2 alternating acc + ds_read_b128 per 8 MFMA", 32.0 * 2048);
    run<3>("pure MFMA 32x32x2, 2 alternating accumulators", 16.0 * 4096);
    run<2>("V2 at 2 waves/SIMD (grid 256)", 32.0 * 2048, 256);
    run<2>("V2 at 1 wave-pair... grid 512 (2 WG/CU)", 32.0 * 2048, 512);
    run<4>("chunked: barriers + 6 ds_write + 288 MFMA, 4 waves/SIMD", 32.0 * 2048, 1024);
    run<4>("chunked: barriers + 6 ds_write + 288 MFMA, 2 waves/SIMD", 32.0 * 2048, 256);
    run<5>("igemm-like NB4 MBW2 chunk, no extra, 2 waves/SIMD", 32.0 * 2048, 256);
    run<6>("  + 12 VALU per block-tap", 32.0 * 2048, 256);
    run<7>("  + 12 VALU + 6 s_nop + branch per block-tap", 32.0 * 2048, 256);
    run<8>("  + 24 VALU per block-tap", 32.0 * 2048, 256);
    return 0;
}
