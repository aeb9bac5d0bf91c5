// BatchNorm2d (train) + AvgPool2d(2) as ONE launch per direction for SMALL batches (the shard of a data-parallel rank): the layer's
// activations stay in LDS between the statistics pass and the normalisation, and the workgroups meet at one grid-wide barrier.
//
// Why: at 2 triplets per rank (6 images) a BatchNorm call of the ae_combined step is 3 launches -- statistics, reduce + finalize, apply
// (backward: reduce, reduce + finalize, apply) -- of 5-8 us each, of which the middle one is pure latency and the third reads the
// activations a second time; 8 calls per step = 24 graph nodes, 138 of the step's ~870 us (profiles/r04_small_shard_budget.txt).
// A 6-image layer is 3-20 MB: it FITS in the chip's LDS (256 CUs x 160 KB = 40 MB).  So:
//   phase 1  workgroup b (of 256, one per CU, 512 threads) loads its units (a unit = RU rows of one image: 2 rows where the pooling
//            follows, so that every 2x2 window is local) into LDS, accumulating per-channel sum / sum of squares per statistic group
//            on the way (backward: sum g, sum g * xhat, with the gathered gradient in LDS too), and writes ONE record [G][2][C] of
//            partial sums;
//   barrier  grid-wide (below);
//   phase 2  EVERY workgroup adds the 256 records in a fixed order (fp64) and finalizes scale / shift (backward: the two coefficients)
//            into LDS -- 128 KB of L2 reads per workgroup instead of a third launch; workgroup 0 also writes mean / invstd / scale /
//            shift for the backward pass and updates the running statistics, group after group (nn.BatchNorm2d semantics,
//            bn.hip: bn_finalize_vals);
//   phase 3  out = scale * pool(y) + shift from LDS (backward: dpre = scale (g - k1 - xhat k2) act'(y)).
// The arithmetic per element is that of bn.hip; the partial sums are formed over other partitions of the pixels, so batch
// statistics agree with the three-launch path to fp64 rounding of the sums (~1e-16), i.e. to the last bit of the fp32 results
// except on rounding ties.  The launcher refuses (returns AESR_ERR_UNSUPPORTED, callers take the three-launch path) when the layer
// does not fit the LDS of 256 workgroups, for the un-folded Upsample mode, and on devices with fewer than 256 CUs.
//
// Grid barrier: 256 workgroups arrive on 8 shard counters (shard = blockIdx % 8: the XCD under round-robin placement -- speed only,
// the populations follow from blockIdx alone), the last arriver of a shard arrives on the top counter, the last of those bumps the
// generation word everybody else polls (relaxed sc1 loads + s_sleep).  Counters are MONOTONIC (an episode adds exactly 32 to a shard
// and 8 to the top; "last" = the count is a multiple), so nothing is ever reset and no two atomics need ordering; the state lives in
// caller-owned memory that is zeroed once (one per network runner: launches that share a state must be serialised on one stream).
// Publication follows MI355X_MICROARCH.md (hand-offs with sc1 accesses, first row): the records are stored write-through (sc1), every
// storing wave waits vmcnt(0), workgroup barrier, one lane arrives; after the poll matched, workgroup barrier, then sc1 loads of the
// records -- no agent-scope fences (the first form of this kernel had both: 3.4 us of its 17-20).  EVERY spin is bounded: a wait that gives up (a workgroup that never became resident: the device is shared with
// another grid-barrier kernel) counts in g_bn_fused_timeouts and the kernel finishes on garbage; the host raises at the next
// boundary (_hip.check_device_watchdogs), as for the ring kernel.
#include <stdlib.h>

#include "aesr_kernels.h"

#define BF_NB_MAX 256             // workgroups of a launch: one per CU (AESR_BN_FUSED_NB = 64 / 128 for rehearsals of several ranks on ONE device)
#define BF_NT 512
#define BF_RED_FL (BF_NT * 8)     // floats of the reduction scratch: [512 / C4][2][C] floats = [row-lanes][G 2 C] doubles at most = 16 KB
#define BF_SPIN_LIMIT (1 << 20)

__device__ unsigned int g_bn_fused_timeouts = 0;

unsigned aesr_bn_fused_timeouts_impl() {
    unsigned v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bn_fused_timeouts), sizeof(v));
    return v;
}

// words of the barrier state (each on a 128-byte line of its own)
#define GB_SHARD(s) ((s) * 32)
#define GB_TOP (8 * 32)
#define GB_GEN (9 * 32)

// sc1 (write-through / L1-bypassing) 16-byte accesses for the records that cross the barrier: MI355X_MICROARCH.md, "Hand-offs measured
// with sc1 loads in place of the acquire", first row -- EVERY store and EVERY load of the handed-off bytes is one of these, every
// storing wave waits vmcnt(0) and the workgroup barriers before its one lane arrives, the polling lane's workgroup barriers before
// anybody loads.  That saves the agent-scope release (buffer_wbl2) and acquire (buffer_inv) fences: ~1.7 us each on the critical path
__device__ __forceinline__ void bf_store_sc1(float* base, size_t nbytes, int byte_off, f32x4 v) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nbytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), rs, byte_off, 0, 16);
}
__device__ __forceinline__ f32x4 bf_load_sc1(const __amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}

// all BF_NB workgroups of the grid; every thread of the workgroup calls it.  Publishes NOTHING but sc1-stored bytes (see above).
__device__ __forceinline__ void bf_grid_barrier(unsigned* bar, int nb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's (write-through) stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        // the generation cannot move before this workgroup has arrived: read it first
        const unsigned my_gen = __hip_atomic_load(bar + GB_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool last = false;
        const unsigned a = __hip_atomic_fetch_add(bar + GB_SHARD(blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if ((a & (unsigned)(nb / 8 - 1)) == 0u) {          // nb / 8 is a power of two
            const unsigned t = __hip_atomic_fetch_add(bar + GB_TOP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            last = (t & 7u) == 0u;
        }
        if (last) {
            __hip_atomic_store(bar + GB_GEN, my_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(bar + GB_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
                __builtin_amdgcn_s_sleep(2);
                // a wait gives up after BF_SPIN_LIMIT polls (counted; the host raises) -- and, once ANY wait of this process has given up, after
                // 1 024: the results are lost already, the remaining launches of the step must not each spin for seconds on top
                if (++spins > BF_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(&g_bn_fused_timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    atomicAdd(&g_bn_fused_timeouts, 1u);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

struct BnFusedArgs {
    // forward: y -> out;  backward: (gout, y) -> dpre
    const float* y; const float* gout; float* out;
    float* rec;                     // [BF_NB][G][2][C] floats: the workgroups' partial sums
    unsigned* bar;                  // grid-barrier state (zeroed once by the owner)
    const float* gamma; const float* beta; float* running_mean; float* running_var; long long* nbt;
    float* mean; float* invstd; float* scale; float* shift;       // [G][C]: written forward, read backward
    float* coef; float* dgamma; float* dbeta;                     // backward
    int N, H, W, C, Ho, Wo, pool;   // pool: AvgPool2d(2) follows (out / gout are [N][H/2][W/2][C])
    int RU, upi, nunits, unit_fl, gunit_fl;      // rows per unit, units per image, units, floats of a unit of y / of the gathered gradient
    int G, update_running, act;
    int nb;                         // workgroups of the launch (64, 128 or 256)
    // data parallel (SyncBN over peer-mapped regions, p2p.hip): world > 0
    int world, rank, slot, p2p_spins;      // p2p_spins: polls before a wait for a peer gives up (AESR_P2P_SPINS; ~4 us each after the first 4096)
    const unsigned* gen;            // device word: the step generation (same on every rank; aesr_p2p_tick advances it once per step)
    unsigned char* peers[8];        // every rank's exchange region as mapped into this process (peers[rank] = this rank's own)
    float momentum, eps, slope;
    double counts[4];
    int nstart[5];
};

__device__ __forceinline__ int bf_group_of(const BnFusedArgs& a, int n) {
    int g = 0;
    for (int k = 1; k < a.G; ++k)
        if (n >= a.nstart[k]) g = k;
    return g;
}

// block reduction of the threads' (s, q) quads that share a channel quad (tid % C4) -> tot[g][2][C] (floats), fixed order
__device__ __forceinline__ void bf_flush(float* red, float* tot, int g, int C, f32x4 s, f32x4 q) {
    const int C4 = C >> 2, PL = BF_NT / C4;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
    __syncthreads();
    *(f32x4*)(red + (pl * 2 + 0) * C + c4 * 4) = s;
    *(f32x4*)(red + (pl * 2 + 1) * C + c4 * 4) = q;
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += BF_NT) {
        float t = 0.f;
        for (int k = 0; k < PL; ++k) t += red[k * 2 * C + o];
        tot[g * 2 * C + o] = t;
    }
    __syncthreads();
}

// totd[o] (fp64, LDS) = sum over the BF_NB records of column o, o < GC2 = G * 2 * C (a multiple of 8, <= 512): fixed order
__device__ __forceinline__ void bf_totals(const float* __restrict__ rec, int GC2, float* red, double* totd, int BF_NB) {
    const int nq = GC2 >> 2, RL = BF_NT / nq;                 // column quads, row-lanes per quad (>= 4)
    const int col = threadIdx.x % nq, rl = threadIdx.x / nq;
    double* redd = (double*)red;                              // [RL][GC2] doubles <= 16 KB
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, BF_NB * GC2 * 4, 0x00020000);
    if (rl < RL) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        // 8 records in flight per thread: the records of other XCDs come across the fabric (1-2 us a round trip); one load per trip
        // made this reduction 16-64 dependent round trips.  The loads are unconditional (clamped row, the value dropped afterwards): a
        // bounds branch in front of each load serialises them
        for (int b0 = rl; b0 < BF_NB; b0 += 8 * RL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = bf_load_sc1(rs, (min(b0 + u * RL, BF_NB - 1) * GC2 + col * 4) * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b0 + u * RL < BF_NB) {
                    s0 += (double)v[u][0];
                    s1 += (double)v[u][1];
                    s2 += (double)v[u][2];
                    s3 += (double)v[u][3];
                }
        }
        redd[rl * GC2 + col * 4 + 0] = s0;
        redd[rl * GC2 + col * 4 + 1] = s1;
        redd[rl * GC2 + col * 4 + 2] = s2;
        redd[rl * GC2 + col * 4 + 3] = s3;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < GC2; o += BF_NT) {
        double t = 0.0;
        for (int r = 0; r < RL; ++r) t += redd[r * GC2 + o];
        totd[o] = t;
    }
    __syncthreads();
}

// ---- data parallel: one-shot exchange of the totals over peer-mapped regions (p2p.hip) ----------------------------------------------
// Region layout (bytes): record of rank r for slot k, parity q at ((k * 2 + q) * world + r) * P2P_REC: [512 doubles | flag line of 128 B].
// A slot is one BatchNorm call of the step (numbered in call order, the same on every rank); the parity is the generation's low bit.
// Workgroup 0 writes THIS rank's totals into every rank's region (system-scope stores over xGMI), fences, and stores the generation into
// its flag there; EVERY workgroup then waits until all `world` flags of the slot in ITS OWN region show the generation and adds the
// records in rank order (fixed order: the same bits on every rank).  Producers never wait for consumers: a rank reaches slot k of step
// t + 1 only behind the gradient all-reduce of step t, which every rank enters after it has consumed all slots of step t.
#define P2P_REC (512 * 8 + 128)
#define P2P_SPIN_LIMIT (1 << 21)          // ~7 s at 3.5 us per poll (round 4: 1 << 23 = 30 s per call, 16 calls per step; AESR_P2P_SPINS overrides)
__device__ __forceinline__ void bf_exchange(const BnFusedArgs& a, int GC2, double* totd) {
    const int tid = threadIdx.x, W = a.world;
    const unsigned gen = *a.gen;
    const size_t base = ((size_t)(a.slot * 2 + (int)(gen & 1u)) * W) * P2P_REC;
    if (blockIdx.x == 0) {
        for (int p = 0; p < W; ++p) {
            double* dst = (double*)(a.peers[p] + base + (size_t)a.rank * P2P_REC);
            for (int o = tid; o < GC2; o += BF_NT) __hip_atomic_store(dst + o, totd[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's (write-through, system-scope) stores have been performed
        __syncthreads();
        if (tid < W) {                                            // ONE wave fences and signals (eight system-scope fences cost eight times one)
            unsigned* flag = (unsigned*)(a.peers[tid] + base + (size_t)a.rank * P2P_REC + 512 * 8);
            __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    unsigned char* mine = a.peers[a.rank] + base;
    if (tid < W) {
        const unsigned* flag = (const unsigned*)(mine + (size_t)tid * P2P_REC + 512 * 8);
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gen) {
            if (spins < 4096) __builtin_amdgcn_s_sleep(4);
            else __builtin_amdgcn_s_sleep(127);
            // a peer that never arrives: give up (counted; the host raises), never hang; after the FIRST give-up of this process every
            // later exchange gives up within 1 024 polls (the step runs on partial sums already; _bounded_sync ends the rank on the host)
            if (++spins > a.p2p_spins || ((spins & 1023) == 0 && __hip_atomic_load(&g_bn_fused_timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                atomicAdd(&g_bn_fused_timeouts, 1u);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    for (int o = tid; o < GC2; o += BF_NT) {
        double t = 0.0;
        for (int r = 0; r < W; ++r) t += __hip_atomic_load((const double*)(mine + (size_t)r * P2P_REC) + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        totd[o] = t;
    }
    __syncthreads();
}

// LDS: [data: maxu * (unit_fl + gunit_fl)] [red: BF_RED_FL] [tot: 512 floats] [totd: 512 doubles] [tabA: 512] [tabB: 512]
__global__ __launch_bounds__(BF_NT, 2) void bn_fused_fwd_kernel(BnFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = a.C, C4 = C >> 2, tid = threadIdx.x, b = blockIdx.x, BF_NB = a.nb;
    const int u0 = (int)((long long)b * a.nunits / BF_NB), u1 = (int)((long long)(b + 1) * a.nunits / BF_NB);
    const int maxu = (a.nunits + BF_NB - 1) / BF_NB;
    float* data = lds;
    float* red = lds + (size_t)maxu * a.unit_fl;
    float* tot = red + BF_RED_FL;
    double* totd = (double*)(tot + 512);
    float* s_sc = (float*)(totd + 512);
    float* s_sh = s_sc + 512;
    const int GC2 = a.G * 2 * C;
    for (int o = tid; o < GC2; o += BF_NT) tot[o] = 0.f;
    // ---- phase 1: units -> LDS, statistics ----
    int cur_g = -1;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    for (int u = u0; u < u1; ++u) {
        const int n = u / a.upi, r0 = (u - n * a.upi) * a.RU;
        const int nr = min(a.RU, a.H - r0);
        const int g = bf_group_of(a, n);
        if (g != cur_g) {
            if (cur_g >= 0) bf_flush(red, tot, cur_g, C, s, q);
            cur_g = g;
            s = q = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const f32x4* src = (const f32x4*)(a.y + ((size_t)n * a.H + r0) * a.W * C);
        f32x4* dst = (f32x4*)(data + (size_t)(u - u0) * a.unit_fl);
        const int cnt4 = nr * a.W * C4;
        // 8 loads in flight per thread (one workgroup per CU: the latency is hidden by the loads of one wave, not by other waves)
        for (int e0 = tid; e0 < cnt4; e0 += 8 * BF_NT) {
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = src[min(e0 + k * BF_NT, cnt4 - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (e0 + k * BF_NT < cnt4) {
                    dst[e0 + k * BF_NT] = v[k];
                    s += v[k];
                    q += v[k] * v[k];
                }
        }
    }
    if (cur_g >= 0) bf_flush(red, tot, cur_g, C, s, q);
    else __syncthreads();
    for (int o = tid; o < (GC2 >> 2); o += BF_NT) bf_store_sc1(a.rec, (size_t)BF_NB * GC2 * 4, (b * GC2 + o * 4) * 4, *(const f32x4*)(tot + o * 4));
    bf_grid_barrier(a.bar, BF_NB);
    // ---- phase 2: totals of all workgroups, finalize (bn.hip: bn_finalize_vals / bn_finalize_apply_kernel) ----
    bf_totals(a.rec, GC2, red, totd, BF_NB);
    if (a.world > 0) bf_exchange(a, GC2, totd);
    const int GC = a.G * C;
    for (int i = tid; i < GC; i += BF_NT) {
        const int g = i / C, c = i - g * C;
        const double M = a.counts[g];
        const double mu = totd[(g * 2 + 0) * C + c] / M;
        double var = totd[(g * 2 + 1) * C + c] / M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float m = (float)mu, iv = (float)(1.0 / sqrt(var + (double)a.eps));
        const float sc = a.gamma[c] * iv, sh = a.beta[c] - m * sc;
        s_sc[i] = sc;
        s_sh[i] = sh;
        if (b == 0) {
            a.mean[i] = m;
            a.invstd[i] = iv;
            a.scale[i] = sc;
            a.shift[i] = sh;
        }
    }
    if (b == 0 && a.update_running) {
        if (tid == 0 && a.nbt) *a.nbt += a.G;
        for (int c = tid; c < C; c += BF_NT) {
            float rm = a.running_mean[c], rv = a.running_var[c];
            for (int g = 0; g < a.G; ++g) {          // group after group, as the reference's successive calls
                const double M = a.counts[g];
                const double mu = totd[(g * 2 + 0) * C + c] / M;
                double var = totd[(g * 2 + 1) * C + c] / M - mu * mu;
                if (var < 0.0) var = 0.0;
                const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
                rm = (1.f - a.momentum) * rm + a.momentum * (float)mu;
                rv = (1.f - a.momentum) * rv + a.momentum * (float)unb;
            }
            a.running_mean[c] = rm;
            a.running_var[c] = rv;
        }
    }
    __syncthreads();
    // ---- phase 3: normalise (+ 2x2 mean) out of LDS ----
    const int c4 = tid % C4;
    for (int u = u0; u < u1; ++u) {
        const int n = u / a.upi, r0 = (u - n * a.upi) * a.RU;
        const int nr = min(a.RU, a.H - r0);
        const int g = bf_group_of(a, n);
        const f32x4 sc = *(const f32x4*)(s_sc + g * C + c4 * 4), sh = *(const f32x4*)(s_sh + g * C + c4 * 4);
        const f32x4* src = (const f32x4*)(data + (size_t)(u - u0) * a.unit_fl);
        if (!a.pool) {
            f32x4* dst = (f32x4*)(a.out + ((size_t)n * a.H + r0) * a.W * C);
            const int cnt4 = nr * a.W * C4;
            for (int e = tid; e < cnt4; e += BF_NT) dst[e] = src[e] * sc + sh;
        } else {
            const int p = r0 >> 1;                    // RU == 2: the unit is one row pair
            if (nr == 2 && p < a.Ho) {
                f32x4* dst = (f32x4*)(a.out + ((size_t)n * a.Ho + p) * a.Wo * C);
                for (int o = tid; o < a.Wo * C4; o += BF_NT) {
                    const int xo = o / C4;
                    const f32x4 v00 = src[(2 * xo) * C4 + c4], v01 = src[(2 * xo + 1) * C4 + c4];
                    const f32x4 v10 = src[(a.W + 2 * xo) * C4 + c4], v11 = src[(a.W + 2 * xo + 1) * C4 + c4];
                    dst[o] = (((v00 + v01) + (v10 + v11)) * 0.25f) * sc + sh;
                }
            }
        }
    }
}

__global__ __launch_bounds__(BF_NT, 2) void bn_fused_bwd_kernel(BnFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int C = a.C, C4 = C >> 2, tid = threadIdx.x, b = blockIdx.x, BF_NB = a.nb;
    const int u0 = (int)((long long)b * a.nunits / BF_NB), u1 = (int)((long long)(b + 1) * a.nunits / BF_NB);
    const int maxu = (a.nunits + BF_NB - 1) / BF_NB;
    float* data = lds;                                              // per unit: y [unit_fl] then the gathered gradient [gunit_fl]
    const int ustride = a.unit_fl + a.gunit_fl;
    float* red = lds + (size_t)maxu * ustride;
    float* tot = red + BF_RED_FL;
    double* totd = (double*)(tot + 512);
    float* s_k = (float*)(totd + 512);                             // [G][2][C]: s1 / M, s2 / M
    const int GC2 = a.G * 2 * C;
    const int c4 = tid % C4;
    for (int o = tid; o < GC2; o += BF_NT) tot[o] = 0.f;
    // ---- phase 1: y and the gradient of the BatchNorm output -> LDS; s1 = sum g, s2 = sum g xhat ----
    int cur_g = -1;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, mu = s1, iv = s1;
    for (int u = u0; u < u1; ++u) {
        const int n = u / a.upi, r0 = (u - n * a.upi) * a.RU;
        const int nr = min(a.RU, a.H - r0);
        const int g = bf_group_of(a, n);
        if (g != cur_g) {
            if (cur_g >= 0) bf_flush(red, tot, cur_g, C, s1, s2);
            cur_g = g;
            s1 = s2 = (f32x4){0.f, 0.f, 0.f, 0.f};
            mu = *(const f32x4*)(a.mean + g * C + c4 * 4);
            iv = *(const f32x4*)(a.invstd + g * C + c4 * 4);
        }
        const f32x4* src = (const f32x4*)(a.y + ((size_t)n * a.H + r0) * a.W * C);
        f32x4* dy = (f32x4*)(data + (size_t)(u - u0) * ustride);
        f32x4* dg = dy + (a.unit_fl >> 2);
        const int cnt4 = nr * a.W * C4;
        if (!a.pool) {
            const f32x4* gsrc = (const f32x4*)(a.gout + ((size_t)n * a.H + r0) * a.W * C);
            for (int e0 = tid; e0 < cnt4; e0 += 4 * BF_NT) {
                f32x4 v[4], gv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = src[min(e0 + k * BF_NT, cnt4 - 1)];
                    gv[k] = gsrc[min(e0 + k * BF_NT, cnt4 - 1)];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (e0 + k * BF_NT < cnt4) {
                        dy[e0 + k * BF_NT] = v[k];
                        dg[e0 + k * BF_NT] = gv[k];
                        s1 += gv[k];
                        s2 += gv[k] * ((v[k] - mu) * iv);
                    }
            }
        } else {
            // the pooled row p = r0 / 2 of the gradient (x 0.25: every pixel of a 2x2 window gets a quarter); rows / columns the pooling
            // left out (odd H / W) get 0
            const int p = r0 >> 1;
            const bool prow = p < a.Ho;
            if (prow) {
                const f32x4* gsrc = (const f32x4*)(a.gout + ((size_t)n * a.Ho + p) * a.Wo * C);
                for (int o = tid; o < a.Wo * C4; o += BF_NT) dg[o] = gsrc[o] * 0.25f;
            }
            __syncthreads();                             // the pooled gradient row is in LDS
            for (int e0 = tid; e0 < cnt4; e0 += 8 * BF_NT) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = src[min(e0 + k * BF_NT, cnt4 - 1)];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + k * BF_NT;
                    if (e < cnt4) {
                        dy[e] = v[k];
                        const int x = (e / C4) % a.W, xo = x >> 1;
                        if (prow && xo < a.Wo) {
                            const f32x4 gg = dg[xo * C4 + c4];
                            s1 += gg;
                            s2 += gg * ((v[k] - mu) * iv);
                        }
                    }
                }
            }
        }
    }
    if (cur_g >= 0) bf_flush(red, tot, cur_g, C, s1, s2);
    else __syncthreads();
    for (int o = tid; o < (GC2 >> 2); o += BF_NT) bf_store_sc1(a.rec, (size_t)BF_NB * GC2 * 4, (b * GC2 + o * 4) * 4, *(const f32x4*)(tot + o * 4));
    bf_grid_barrier(a.bar, BF_NB);
    // ---- phase 2: coefficients (bn.hip: bn_bwd_finalize_apply_kernel) ----
    bf_totals(a.rec, GC2, red, totd, BF_NB);
    if (a.world > 0) bf_exchange(a, GC2, totd);
    for (int i = tid; i < GC2; i += BF_NT) {
        const int g = i / (2 * C);
        const float k = (float)(totd[i] / a.counts[g]);
        s_k[i] = k;
        if (b == 0) a.coef[i] = k;
    }
    if (b == 0)
        for (int c = tid; c < C; c += BF_NT) {
            double dgm = 0.0, dbt = 0.0;
            for (int g = 0; g < a.G; ++g) {
                dbt += totd[(g * 2 + 0) * C + c];
                dgm += totd[(g * 2 + 1) * C + c];
            }
            a.dgamma[c] = (float)dgm;
            a.dbeta[c] = (float)dbt;
        }
    __syncthreads();
    // ---- phase 3: dpre = scale (g - k1 - xhat k2) act'(y) ----
    for (int u = u0; u < u1; ++u) {
        const int n = u / a.upi, r0 = (u - n * a.upi) * a.RU;
        const int nr = min(a.RU, a.H - r0);
        const int g = bf_group_of(a, n);
        const f32x4 m = *(const f32x4*)(a.mean + g * C + c4 * 4), v_iv = *(const f32x4*)(a.invstd + g * C + c4 * 4);
        const f32x4 sc = *(const f32x4*)(a.scale + g * C + c4 * 4);
        const f32x4 k1 = *(const f32x4*)(s_k + (g * 2 + 0) * C + c4 * 4), k2 = *(const f32x4*)(s_k + (g * 2 + 1) * C + c4 * 4);
        const f32x4* dy = (const f32x4*)(data + (size_t)(u - u0) * ustride);
        const f32x4* dg = dy + (a.unit_fl >> 2);
        f32x4* dst = (f32x4*)(a.out + ((size_t)n * a.H + r0) * a.W * C);
        const int cnt4 = nr * a.W * C4;
        const bool prow = (r0 >> 1) < a.Ho;
        for (int e = tid; e < cnt4; e += BF_NT) {
            const f32x4 yv = dy[e];
            f32x4 gg = {0.f, 0.f, 0.f, 0.f};
            if (!a.pool) {
                gg = dg[e];
            } else {
                const int x = (e / C4) % a.W, xo = x >> 1;
                if (prow && xo < a.Wo) gg = dg[xo * C4 + c4];
            }
            const f32x4 xh = (yv - m) * v_iv;
            f32x4 d = sc * (gg - k1 - xh * k2);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] *= act_grad_from_output(yv[k], a.act, a.slope);
            dst[e] = d;
        }
    }
}

// ---- launcher ----------------------------------------------------------------------------------------------------------------
// workgroups per launch: 256 = one per CU; AESR_BN_FUSED_NB = 64 / 128 lets several ranks that are REHEARSED on one device be resident
// together (their kernels wait for each other's records).  Read once: a barrier state must always meet the same number.
static int bf_nb() {
    static int nb = 0;
    if (nb == 0) {
        const char* e = getenv("AESR_BN_FUSED_NB");
        const int v = e ? atoi(e) : BF_NB_MAX;
        nb = (v == 64 || v == 128) ? v : BF_NB_MAX;
    }
    return nb;
}

static bool bf_plan(BnFusedArgs& a, int backward, size_t* shmem) {
    const int C4 = a.C / 4;
    if (a.C % 4 != 0 || C4 <= 0 || BF_NT % C4 != 0 || a.G < 1 || a.G > 4 || a.G * 2 * a.C > 512) return false;
    if ((double)a.N * a.H * a.W * a.C >= 2147483648.0) return false;
    a.RU = a.pool ? 2 : 1;
    a.upi = (a.H + a.RU - 1) / a.RU;
    a.nunits = a.N * a.upi;
    a.unit_fl = a.RU * a.W * a.C;
    a.gunit_fl = backward ? (a.pool ? ((a.Wo * a.C + 3) & ~3) : a.unit_fl) : 0;
    a.nb = bf_nb();
    const size_t maxu = (size_t)(a.nunits + a.nb - 1) / a.nb;
    *shmem = (maxu * (size_t)(a.unit_fl + a.gunit_fl) + BF_RED_FL + 512 + 1024 + 1024) * sizeof(float);
    return *shmem <= (size_t)150 * 1024;
}

static int bf_device_ok() {
    static int ok[AESR_MAX_DEVICES] = {};          // 0 unknown, 1 yes, -1 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= AESR_MAX_DEVICES) return 0;
    if (ok[dev] == 0) {
        int cus = 0;
        ok[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= BF_NB_MAX) ? 1 : -1;
        if (ok[dev] == 1) {
            const hipError_t e1 = hipFuncSetAttribute((const void*)bn_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            const hipError_t e2 = hipFuncSetAttribute((const void*)bn_fused_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            if (e1 != hipSuccess || e2 != hipSuccess) ok[dev] = -1;
        }
    }
    return ok[dev] == 1;
}

bool aesr_bn_fused1_ok(int N, int H, int W, int C, int pool, int G, int backward) {
    BnFusedArgs a = {};
    a.N = N; a.H = H; a.W = W; a.C = C; a.pool = pool; a.G = G; a.Ho = pool ? H / 2 : H; a.Wo = pool ? W / 2 : W;
    size_t sh;
    return N > 0 && H > 0 && W > 0 && (!pool || (H >= 2 && W >= 2)) && bf_plan(a, backward, &sh) && bf_device_ok();
}

int aesr_launch_bn_fused(BnFusedArgs a, int backward, hipStream_t st) {
    size_t shmem = 0;
    if (!bf_plan(a, backward, &shmem) || !bf_device_ok()) {
        aesr_set_error("bn_fused: %d x %d x %d x %d (%d groups) does not fit the one-launch form (aesr_bn_fused1_supported)", a.N, a.H, a.W, a.C, a.G);
        return AESR_ERR_UNSUPPORTED;
    }
    if (backward) hipLaunchKernelGGL(bn_fused_bwd_kernel, dim3(a.nb), dim3(BF_NT), shmem, st, a);
    else hipLaunchKernelGGL(bn_fused_fwd_kernel, dim3(a.nb), dim3(BF_NT), shmem, st, a);
    AESR_LAUNCH_CHECK(backward ? "bn_fused_bwd" : "bn_fused_fwd");
    return AESR_OK;
}

int aesr_bn_fused_run(const float* y, const float* gout, float* out, float* rec, unsigned* bar, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, long long* nbt, float* mean, float* invstd, float* scale, float* shift, float* coef,
                      float* dgamma, float* dbeta, int N, int H, int W, int C, int pool, int G, const int* nstart, const double* counts,
                      float momentum, float eps, int update_running, int act, float slope, int backward, const BnP2P* p2p, hipStream_t st) {
    BnFusedArgs a = {};
    if (p2p) {
        const char* e = getenv("AESR_P2P_SPINS");             // tests shorten the wait for a peer that never comes (default P2P_SPIN_LIMIT: ~7 s -- the peer-skew tolerance of AESR_SYNCBN=p2p)
        a.p2p_spins = e && atoi(e) > 0 ? atoi(e) : P2P_SPIN_LIMIT;
        a.world = p2p->world; a.rank = p2p->rank; a.slot = p2p->slot; a.gen = p2p->gen;
        for (int r = 0; r < 8; ++r) a.peers[r] = r < p2p->world ? (unsigned char*)p2p->peers[r] : nullptr;
    }
    a.y = y; a.gout = gout; a.out = out; a.rec = rec; a.bar = bar; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean;
    a.running_var = running_var; a.nbt = nbt; a.mean = mean; a.invstd = invstd; a.scale = scale; a.shift = shift; a.coef = coef;
    a.dgamma = dgamma; a.dbeta = dbeta; a.N = N; a.H = H; a.W = W; a.C = C; a.pool = pool; a.Ho = pool ? H / 2 : H; a.Wo = pool ? W / 2 : W;
    a.G = G; a.update_running = update_running; a.act = act; a.momentum = momentum; a.eps = eps; a.slope = slope;
    for (int g = 0; g < 4; ++g) a.counts[g] = g < G ? counts[g] : 1.0;
    for (int g = 0; g <= 4; ++g) a.nstart[g] = g <= G ? nstart[g] : nstart[G];
    return aesr_launch_bn_fused(a, backward, st);
}

// the step generation of the peer exchange: advanced once per step on every rank (a graph node like any other kernel)
__global__ void p2p_tick_kernel(unsigned* gen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned g = *gen + 1u;
        if (g == 0u) g = 2u;          // 0 is the "never written" value of the flags; 2 keeps the slot parity alternating across the wrap (... 0xffffffff, 2, 3 ...)
        *gen = g;
    }
}

int aesr_launch_p2p_tick(unsigned* gen, hipStream_t st) {
    hipLaunchKernelGGL(p2p_tick_kernel, dim3(1), dim3(64), 0, st, gen);
    AESR_LAUNCH_CHECK("p2p_tick");
    return AESR_OK;
}
