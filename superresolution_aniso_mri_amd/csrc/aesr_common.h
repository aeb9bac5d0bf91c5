// Shared device/host helpers for the aesr HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// a - b on four floats as TWO v_pk_add_f32 with the negate modifiers: the compiler turns every vector subtraction into four
// scalar v_sub_f32 (it only packs additions), and beside f32 MFMAs every vector instruction costs matrix-pipe time
// (scripts/micro/mfma_gap.hip).  Bit-identical to the scalar subtraction.
__device__ __forceinline__ f32x4 aesr_sub4(f32x4 a, f32x4 b) {
#ifdef __HIP_DEVICE_COMPILE__
    const f32x2_t alo = a.xy, ahi = a.zw, blo = b.xy, bhi = b.zw;
    f32x2_t rlo, rhi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rlo) : "v"(alo), "v"(blo));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rhi) : "v"(ahi), "v"(bhi));
    return (f32x4){rlo.x, rlo.y, rhi.x, rhi.y};
#else
    return a - b;           // host pass of the single-source compile: never executed
#endif
}

#define AESR_OK 0
#define AESR_ERR_ARG 1
#define AESR_ERR_HIP 2
#define AESR_ERR_UNSUPPORTED 3
#define AESR_MAX_DEVICES 16      // per-device one-time state (function attributes)

// activation codes shared by every epilogue
enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };

void aesr_set_error(const char* fmt, ...);

#define AESR_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            aesr_set_error(__VA_ARGS__);     \
            return AESR_ERR_ARG;             \
        }                                    \
    } while (0)

#define AESR_LAUNCH_CHECK(name)                                                     \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            aesr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return AESR_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// derivative of the activation expressed through its saved OUTPUT y (sign(y) == sign(pre-activation))
__device__ __forceinline__ float act_grad_from_output(float y, int act, float slope) {
    if (act == ACT_LRELU) return y > 0.f ? 1.f : slope;
    if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

static inline int ceil_div(int a, int b) { return a / b + (a % b != 0); }       // (a, b > 0; no a + b - 1 to overflow)
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
