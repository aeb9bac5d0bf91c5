// Checks the v_pk_add_f32 source-select / negate modifier forms and the ds_read2st64_b32 pairing the Winograd weight-gradient
// kernel (conv_wgrad_wino.hip) relies on.   hipcc -O3 --offload-arch=gfx950 scripts/micro/pk_mods.hip -o /tmp/pk_mods && /tmp/pk_mods
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k(float* out) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)i;
    __syncthreads();
    f32x2 p01 = {1.f, 10.f}, p23 = {100.f, 1000.f}, r[6];
    // (d0 - d2, d1 + d2)
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r[0]) : "v"(p01), "v"(p23));
    // (d2 - d1, d1 - d3)
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r[1]) : "v"(p01), "v"(p23));
    // a - b (both halves)
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r[2]) : "v"(p01), "v"(p23));
    // (a.lo + a.hi, a.lo - a.hi) of ONE pair
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r[3]) : "v"(p01));
    // (a.lo + b.lo, a.lo - b.lo) and (a.hi + b.hi, a.hi - b.hi) of two pairs
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r[4]) : "v"(p01), "v"(p23));
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r[5]) : "v"(p01), "v"(p23));
    const unsigned addr = (unsigned)(unsigned long long)(lds + threadIdx.x);
    f32x2 q;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:1 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr));
    if (threadIdx.x == 5) {
        for (int i = 0; i < 6; ++i) { out[2 * i] = r[i][0]; out[2 * i + 1] = r[i][1]; }
        out[12] = q[0]; out[13] = q[1];
    }
}

int main() {
    float* d; (void)hipMalloc(&d, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[14]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const float want[14] = {1 - 100, 10 + 100, 100 - 10, 10 - 1000, 1 - 100, 10 - 1000, 1 + 10, 1 - 10, 1 + 100, 1 - 100, 10 + 1000, 10 - 1000, 64 + 5, 192 + 5};
    int bad = 0;
    for (int i = 0; i < 14; ++i) { printf("%d: got %g want %g%s\n", i, h[i], want[i], h[i] == want[i] ? "" : "   <-- MISMATCH"); bad += h[i] != want[i]; }
    printf(bad ? "FAILED\n" : "all forms as expected\n");
    return bad != 0;
}
