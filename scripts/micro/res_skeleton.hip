// Micro-benchmark (round 5): would ONE wave per SIMD with a software-pipelined chunk loop beat TWO waves per SIMD for conv_wino_res_f32?
// (round-4 verdict, next 2: "one wave per SIMD, the NEXT chunk's patch in a second register set, its reads / transform / DMA issues dealt out
//  between the MFMA runs, epilogue of item k under the first chunk of item k + 1 ... if the stamped prototype does not beat 95.8 -> <= 88 us, record
//  it as rejected with the stamps and stop".)
//
// Timing-only SKELETONS with the instruction totals of the shipped kernel's item loop (scripts/isa_bound.py on conv_wino_res_f32<32, false>):
// per item 256 v_mfma_f32_16x16x4_f32, 164 v_pk_add_f32, 324 other vector instructions, 96 ds_read_b128 (32 patch + 64 filter), 20 LDS-DMAs of 1 KB
// (buffer_load_dwordx4 ... lds from an L2-resident source), 8 buffer_store_dwordx4.  No data dependence between the fillers and the MFMAs except the
// ones a real kernel cannot avoid (the filter reads feed the MFMAs one position later; a patch is read only after its DMAs have landed):
//   mode 0  "seq2":  512 threads = two waves per SIMD, 128 accumulators each, the shipped kernel's ORDER: per chunk [wait patch, 16 patch reads, wait,
//                    10 DMAs, row transform], 16 positions x [2 filter reads, column transform, 8 MFMAs], epilogue behind the second chunk
//   mode 1  "pipe1": 256 threads = ONE wave per SIMD, 256 accumulators (two items' worth), everything besides the MFMAs dealt out evenly between the
//                    32 position groups of an item -- the proposed rewrite in its most favourable form (nothing ever waits)
//   mode 2  "seq1":  one wave per SIMD in the shipped order (what dropping a wave without re-pipelining costs)
//   mode 3  "pong2": two waves per SIMD in PING-PONG: waves 0..3 run a chunk's 128 MFMAs (+ its 32 filter reads) while waves 4..7 run everything
//                    else of THEIR next chunk (patch wait + reads, DMA issue, the whole input transform, and behind an item's last chunk its
//                    epilogue), one s_barrier, roles swap -- the matrix pipe of a SIMD always belongs to exactly one wave
// Output: cycles of SIMD time per item = what the matrix pipe would need is 256 x 32 = 8 192.
// build + run:  hipcc -O3 --offload-arch=gfx950 scripts/micro/res_skeleton.hip -o /tmp/res_skeleton && /tmp/res_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define PK(n) asm volatile(".rept %3\n\tv_pk_add_f32 %0, %2, %2\n\tv_pk_add_f32 %1, %2, %2\n\t.endr" : "=v"(p0), "=v"(p1) : "v"(pa), "i"((n) / 2))
#define VA(n) asm volatile(".rept %4\n\tv_add_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3\n\t.endr" : "=v"(f0), "=v"(f1) : "v"(a0), "v"(b0), "i"((n) / 2))
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(n))
#define VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n))

__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t rs, float* lds_wave_base, int byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, byte_off, 0, 0, 0);
}
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, int byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 0);
}

constexpr int PFL = 2600;            // floats of a wave's patch buffer (10 rows x 260, as in conv_wino_res.hip)

template <int MODE>
__global__ __launch_bounds__((MODE == 0 || MODE == 3) ? 512 : 256, (MODE == 0 || MODE == 3) ? 2 : 1) void skel(const float* src, float* out, long long* cyc, int items, unsigned src_bytes, unsigned out_bytes, int stream) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NW = (MODE == 0 || MODE == 3) ? 8 : 4, NBUF = (MODE == 0 || MODE == 3) ? 1 : 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384 + NW * NBUF * PFL; i += blockDim.x) lds[i] = (float)(i & 15) * 0.0625f;
    __syncthreads();
    float* const ldsP = lds + 16384 + wave * NBUF * PFL;                   // patch buffer(s) of this wave
    const unsigned uaddr = (unsigned)(unsigned long long)(lds + lane * 4);            // filter reads: 64 lanes x 16 B, conflict free
    const unsigned paddr = (unsigned)(unsigned long long)(ldsP + lane * 4);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    // stream == 0: sources L2-resident (a 16 MB window re-read), results into a 64 MB window (absorbed by the Infinity Cache);
    // stream == 1: every wave reads and writes bytes of its OWN that nobody touched before (as a layer of 121 MB in, 121 MB out does): HBM both ways
    const unsigned wg = (unsigned)(blockIdx.x * NW + wave), ipw = (unsigned)items;
    const bool rs_ = (stream & 1) != 0, ws_ = (stream & 2) != 0;           // bit 0: reads stream from HBM, bit 1: writes stream to HBM
    const unsigned dmask = rs_ ? 0xffffffffu : ((16u << 20) / 2 - 1), omask = ws_ ? 0xffffffffu : ((64u << 20) - 1);
    unsigned doff = (rs_ ? wg * ipw * 20u * 1024u : wg * 20u * 1024u) + lane * 16;
    doff &= dmask;
    unsigned ooff = ((ws_ ? wg * ipw * 8u * 1024u : wg * 8u * 1024u) + lane * 16) & omask;
    constexpr int NACC = MODE == 1 ? 64 : 32;
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, pa = {0.5f, 0.25f};
    float f0 = 1.f, f1 = 2.f, a0 = lane * 0.001f, b0 = 1.f + lane * 0.002f;
    f32x4 uA0 = {1.f, 1.f, 1.f, 1.f}, uA1 = uA0, uB0 = uA0, uB1 = uA0, t0 = uA0;
    const f32x4 sv = {1.f, 2.f, 3.f, 4.f};
    const long long c0 = (long long)__builtin_amdgcn_s_memtime();
    if constexpr (MODE == 3) {
        // ---- ping-pong: step s = (chunk index of group A); group B is half a step behind ----
        const bool grpB = wave >= 4;
        auto other = [&](bool with_epilogue) {
            VM(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) DSR(t0, paddr, (r & 7) * 1040);
            LGKM(0);
#pragma unroll
            for (int r = 0; r < 10; ++r) dma(rs_in, ldsP + r * 260, (int)(doff + r * 1024));
            doff = (doff + 10 * 1024) & dmask;
            PK(64);                                                     // row AND column half of the input transform: V of all 16 positions
            VA(24);
            if (with_epilogue) {
                PK(20);
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_) {
                    PK(2);
                    VA(34);
                    st(rs_out, (int)(ooff + s_ * 1024), sv);
                }
                VA(4);
                if (stream & 2) ooff += 8 * 1024;
            }
        };
        auto mfmas = [&]() {
            DSR(uA0, uaddr, 0);
            DSR(uA1, uaddr, 1024);
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                DSR(uB0, uaddr, ((g + 1) & 15) * 2048);
                DSR(uB1, uaddr, ((g + 1) & 15) * 2048 + 1024);
                LGKM(2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    MFMA(acc[2 * g], uA0[r], b0);
                    MFMA(acc[2 * g + 1], uA1[r], b0);
                }
                if (g + 2 < 16) {
                    DSR(uA0, uaddr, ((g + 2) & 15) * 2048);
                    DSR(uA1, uaddr, ((g + 2) & 15) * 2048 + 1024);
                    LGKM(2);
                } else {
                    LGKM(0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    MFMA(acc[2 * g + 2], uB0[r], b0);
                    MFMA(acc[2 * g + 3], uB1[r], b0);
                }
            }
        };
        // every wave runs the SAME loop [everything else | barrier | MFMAs | barrier]; waves 4..7 enter it one barrier late, so between two
        // barriers one wave of every SIMD is in its MFMA segment and the other in its everything-else segment
#define BAR asm volatile("s_barrier" ::: "memory")      /* a BARE barrier: no vmcnt / lgkmcnt drain (each wave owns its patch buffer) */
        if (grpB) BAR;
#pragma unroll 1
        for (int stp = 0; stp < 2 * items; ++stp) {                     // 2 chunks per item
            other((stp & 1) == 0 && stp > 0);                           // behind an item's second chunk: its epilogue
            BAR;
            mfmas();
            BAR;
        }
        if (!grpB) BAR;
    } else if constexpr (MODE != 1) {
        // ---- the shipped order ----
#pragma unroll 1
        for (int it = 0; it < items; ++it) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                VM(0);                                                  // this chunk's patch (requested a chunk ago) has landed
#pragma unroll
                for (int r = 0; r < 16; ++r) DSR(t0, paddr, (r & 7) * 1040);
                DSR(uA0, uaddr, 0);
                DSR(uA1, uaddr, 1024);
                LGKM(0);
#pragma unroll
                for (int r = 0; r < 10; ++r) dma(rs_in, ldsP + r * 260, (int)(doff + r * 1024));          // next patch into the (single) buffer
                doff = (doff + 10 * 1024) & dmask;
                PK(32);                                                 // row half of the input transform
                VA(24);
#pragma unroll
                for (int g = 0; g < 16; g += 2) {
                    // position g on set A while set B is fetched, then position g + 1 on set B while set A is fetched (no register copies)
                    DSR(uB0, uaddr, ((g + 1) & 15) * 2048);
                    DSR(uB1, uaddr, ((g + 1) & 15) * 2048 + 1024);
                    PK(2);                                              // column half, one position ahead
                    LGKM(2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        MFMA(acc[2 * g], uA0[r], b0);
                        MFMA(acc[2 * g + 1], uA1[r], b0);
                    }
                    if (g + 2 < 16) {
                        DSR(uA0, uaddr, ((g + 2) & 15) * 2048);
                        DSR(uA1, uaddr, ((g + 2) & 15) * 2048 + 1024);
                        LGKM(2);
                    } else {
                        LGKM(0);
                    }
                    PK(2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        MFMA(acc[2 * g + 2], uB0[r], b0);
                        MFMA(acc[2 * g + 3], uB1[r], b0);
                    }
                }
            }
            // epilogue: output transform, activation, 8 stores
            PK(20);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                PK(2);
                VA(34);
                st(rs_out, (int)(ooff + s * 1024), sv);
            }
            VA(4);
            if (stream & 2) ooff += 8 * 1024;
        }
    } else {
        // ---- one wave per SIMD, everything dealt out between the 32 position groups of an item; accumulator set (it & 1) ----
        // per group: 8 MFMAs + 2 filter reads + 1 patch read + 5 v_pk_add + 10 other vector instructions; a DMA in 20 of the 32 groups, a store
        // (of the PREVIOUS item's results) in 8 of them.  5 x 32 = 160 (+ 4) packed, 10 x 32 = 320 (+ 4) others.
#pragma unroll 1
        for (int it = 0; it < items; it += 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#define GROUP(G, UC0, UC1, UN0, UN1)                                                                                   \
    {                                                                                                                  \
        constexpr int g = (G);                                                                                         \
        DSR(UN0, uaddr, ((g + 1) & 15) * 2048);                                                                        \
        DSR(UN1, uaddr, ((g + 1) & 15) * 2048 + 1024);                                                                 \
        DSR(t0, paddr, (g & 7) * 1040 + (g & 16 ? PFL * 4 : 0));                                                       \
        if constexpr ((g & 15) >= 3 && (g & 15) < 13) { /* 10 DMAs per chunk, into the buffer NOT being read */        \
            dma(rs_in, ldsP + (g & 16 ? 0 : PFL) + ((g & 15) - 3) * 260, (int)doff);                                        \
            doff = (doff + 1024) & dmask;                                                                \
        }                                                                                                              \
        PK(4);                                                                                                         \
        VA(10);                                                                                                        \
        if constexpr ((g & 3) == 1) st(rs_out, (int)(ooff + (g >> 2) * 1024), sv);                                            \
        if constexpr ((g & 15) == 15) VM(2); /* the patch the next chunk reads: requested 12+ groups ago */            \
        asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(p0) : "v"(pa));                                                  \
        LGKM(3);                                                                                                       \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                                \
            MFMA(acc[half * 32 + (g & 15) * 2], UC0[r], b0);                                                           \
            MFMA(acc[half * 32 + (g & 15) * 2 + 1], UC1[r], b0);                                                       \
        }                                                                                                              \
    }
#define GROUP2(G) GROUP(G, uA0, uA1, uB0, uB1) GROUP((G) + 1, uB0, uB1, uA0, uA1)
#define GROUP8(G) GROUP2(G) GROUP2((G) + 2) GROUP2((G) + 4) GROUP2((G) + 6)
                GROUP8(0) GROUP8(8) GROUP8(16) GROUP8(24)
                PK(4);
                VA(4);
                if (stream & 2) ooff += 8 * 1024;
            }
        }
    }
    VM(0);
    LGKM(0);
    const long long c1 = (long long)__builtin_amdgcn_s_memtime();
    float s = f0 + f1 + p0[0] + p1[1] + t0[0] + uA0[0] + uA1[1] + uB0[2] + uB1[3];
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0];
    if (s == 12345.678f) out[tid] = s;
    if (lane == 0) cyc[blockIdx.x * NW + wave] = c1 - c0;
}

template <int MODE>
static void run(const char* name, const float* src, float* out, long long* cyc, unsigned src_bytes, unsigned out_bytes, int stream) {
    constexpr int NW = (MODE == 0 || MODE == 3) ? 8 : 4, NT = (MODE == 0 || MODE == 3) ? 512 : 256;
    const size_t shmem = (16384 + NW * ((MODE == 0 || MODE == 3) ? 1 : 2) * PFL) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)skel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // the same number of items per SIMD in every mode: 64 per SIMD = 32 per wave with two waves, 64 with one
    const int items = (MODE == 0 || MODE == 3) ? 32 : 64;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((skel<MODE>), dim3(256), dim3(NT), shmem, 0, src, out, cyc, items, src_bytes, out_bytes, stream);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    static long long h[256 * 8];
    (void)hipMemcpy(h, cyc, 256 * NW * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int i = 0; i < 256 * NW; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
    const double per_wave = sum / (256 * NW) / items;                   // cycles per item of ONE wave
    const double per_simd = per_wave / ((MODE == 0 || MODE == 3) ? 2 : 1);             // two waves of a SIMD finish two items in that time
    printf("%-6s %s %d waves/SIMD: %.0f cycles per item of a wave = %.0f cycles of SIMD time per item -> matrix pipe %.1f %% busy by the mean wave, %.1f %% by the slowest (%.0f per item)\n", name,
           stream == 0 ? "reads cached, writes cached" : stream == 1 ? "reads HBM,    writes cached" : stream == 2 ? "reads cached, writes HBM   " : "reads HBM,    writes HBM   ", MODE == 0 ? 2 : 1, per_wave, per_simd, 100.0 * 8192.0 / per_simd, 100.0 * 8192.0 / (mx / items / ((MODE == 0 || MODE == 3) ? 2 : 1)), mx / items);
}

int main() {
    // stream mode: 2 048 (1 024) waves x 32 (64) items x 20 KB read = 1.34 GB, x 8 KB written = 0.54 GB; offsets stay below 2^31 (buffer offsets are 32-bit)
    const unsigned src_bytes = 1408u << 20, out_bytes = 576u << 20;
    float *src, *out;
    long long* cyc;
    if (hipMalloc(&src, src_bytes) != hipSuccess || hipMalloc(&out, out_bytes) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    (void)hipMemset(src, 0, src_bytes);
    printf("per item: 256 MFMA (8 192 cycles), 164 v_pk_add_f32, 324 other vector instructions, 96 ds_read_b128, 20 LDS-DMAs, 8 x 16-byte stores\n");
    for (int rep = 0; rep < 2; ++rep)
        for (int stream = 0; stream < 4; ++stream) {
            const unsigned sb = src_bytes, ob = out_bytes;
            run<0>("seq2", src, out, cyc, sb, ob, stream);
            run<1>("pipe1", src, out, cyc, sb, ob, stream);
            run<2>("seq1", src, out, cyc, sb, ob, stream);
            run<3>("pong2", src, out, cyc, sb, ob, stream);
        }
    return 0;
}
