// GroupNorm32 (+FiLM, +SiLU) with ONE read of every input: workgroup-cooperative, register-resident.   (bf16, NHWC)
//   reference: tools/nn.py:17-19,93-100 (GroupNorm32), models/unet.py:236-256 (the FiLM / SiLU around it)
//
// The streaming kernels above read x twice in the forward (sums, apply) and x and dout twice in the backward: 3 and 5
// HBM passes.  A sample's group statistics need all of its HW rows, and a [HW, C] slab (1.5 MB at 64 x 64 x 192) is
// more than one CU holds -- but not more than a few of them do: here an ITEM is one chunk of R rows of one sample,
// sized so that its x (and dout, and dx_add) rows sit in the registers of ONE workgroup (12 x 16 B per lane and stream),
// and the nch workgroups that hold one sample's chunks exchange their partial sums through L2:
//   phase 1   load the chunk (every load of the item in flight at once), per-channel sums in registers, cross-thread
//             fold through LDS, per-(chunk, group) partials to global, release, arrive on the sample's counter
//   wait      one lane spins (acquire, bounded: a trap, not a hang, if the partners never arrive) until nch arrivals
//   phase 2   every workgroup folds the sample's nch partials in the same fixed order (so all of them hold identical
//             statistics), applies them to the rows it still holds and stores y / dx
// 2 and 3 passes (+ dx_add).  Co-residency is by construction: the grid is one resident round of workgroups (occupancy x
// CUs, never more), items are dealt round-robin in sample-major order and every workgroup walks its items in increasing
// order, so the partners of a sample are nch consecutive workgroups that reach it after finishing only EARLIER samples
// (needs nch <= grid and nch <= 64; anything else takes the streaming kernels).
// Thread mapping: a thread owns one channel OCTET (16 B of bf16) and every rpi-th row; nt = a multiple of C/8 threads
// are live, so a workgroup's loads of one step are nt x 16 contiguous bytes and all per-channel constants stay in
// registers.  Summation order is fixed (no float atomics): rows within a thread, threads of an octet in LDS order,
// channels of a group, chunks.
#pragma once

#define GNC_NV_FWD 6       // 16-byte vectors per lane and stream held in registers: forward (one stream)
#define GNC_NV_BWD 8       // backward (x, dout, dx_add)
#define GNC_NT 512
#define GNC_CMAX 2048
#ifndef GNC_NT_STORE
#define GNC_NT_STORE 1
#endif

#ifndef GNC_PROF
#define GNC_PROF 0
#endif
#if GNC_PROF
__device__ unsigned long long gnc_prof_buf[4096];
#define GNC_STAMP(slot)                                                                                    \
    do {                                                                                                   \
        if (threadIdx.x == 64 && blockIdx.x == 7 && prof_n < 4000) gnc_prof_buf[prof_n++] = ((unsigned long long)(slot) << 56) | (wall_clock64() & 0xffffffffffffffull); \
    } while (0)
#else
#define GNC_STAMP(slot) do {} while (0)
#endif

struct GncGeom {
    int C8, nt, rpi, R, nch, items;
};
static inline GncGeom gnc_geom(int B, int HW, int C, int nv) {
    GncGeom g;
    g.C8 = C / 8;
    g.nt = (GNC_NT / g.C8) * g.C8;
    g.rpi = g.nt / g.C8;
    const int rmax = g.rpi * nv;
    g.nch = (HW + rmax - 1) / rmax;
    g.R = (HW + g.nch - 1) / g.nch;
    g.items = B * g.nch;
    return g;
}
static inline bool gnc_shape_ok(vaw_dtype dt, int B, int HW, int C, int G) {
    return dt == VAW_BF16 && C % 8 == 0 && C / 8 <= GNC_NT && C <= GNC_CMAX && G <= 32 && C % G == 0 &&
           (int64_t)B * HW * C < ((int64_t)1 << 40);
}
// floats of workspace: part1 [B][nch][2][64] | part2 [B][nch][2][C] | sums [4][B][C] | counters [B]
static inline int64_t gnc_workspace_floats(int B, int HW, int C) {
    if (C % 8 != 0 || C / 8 > GNC_NT) return 0;
    const GncGeom gb = gnc_geom(B, HW, C, GNC_NV_BWD), gf = gnc_geom(B, HW, C, GNC_NV_FWD);
    const int64_t bwd = (int64_t)B * gb.nch * 2 * (64 + C) + (int64_t)4 * B * C + B + 64;      // part1 | part2 | sums | counters
    const int64_t fwd = (int64_t)B * gf.nch * 128 + 64;                                          // part1 only
    return bwd > fwd ? bwd : fwd;
}

typedef unsigned gnc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gnc_u32x4 gnc_ld(const bf16_t* p) { return __builtin_nontemporal_load(reinterpret_cast<const gnc_u32x4*>(p)); }
__device__ __forceinline__ void gnc_st(bf16_t* p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)f[j];
#if GNC_NT_STORE
    __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
#else
    *reinterpret_cast<bf16x8*>(p) = v;
#endif
}
// eight bf16 of a held vector as f32 (a shift / a mask each: cheaper to redo than to keep 8 registers per vector alive)
__device__ __forceinline__ void gnc_unpack(gnc_u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = __uint_as_float(v[k] << 16);
        f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u);
    }
}
// SiLU through v_rcp_f32 as in the streaming kernels (gn_silu): with 2-3 waves per SIMD the arithmetic of a chunk is on the
// critical path of the item
#define gnc_silu gn_silu
#define gnc_silu_grad gn_silu_grad
// the held vectors are re-read in phase 2: keep the compiler from carrying their unpacked (or activated) forms across the wait
#define GNC_FORGET(v) asm volatile("" : "+v"(v))
#define GNC_SENTINEL 0xffffffffu
__device__ __forceinline__ float gnc_not_sentinel(float v) { return __float_as_uint(v) == GNC_SENTINEL ? __uint_as_float(0x7fc00000u) : v; }
__device__ __forceinline__ float gnc_peek(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gnc_poke(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// per-thread sums of 8 channels x 2 quantities -> this chunk's per-channel sums in chs[2][C]
__device__ __forceinline__ void gnc_fold_chunk(const float (&s)[8], const float (&q)[8], float* red, float (*chs)[GNC_CMAX], int C, int rpi,
                                               int rl, int c0, bool live) {
    if (live) {
        float* r0 = red + (int64_t)rl * C + c0;
        float* r1 = red + (int64_t)(rpi + rl) * C + c0;
        *reinterpret_cast<f32x4*>(r0) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(r0 + 4) = f32x4{s[4], s[5], s[6], s[7]};
        *reinterpret_cast<f32x4*>(r1) = f32x4{q[0], q[1], q[2], q[3]};
        *reinterpret_cast<f32x4*>(r1 + 4) = f32x4{q[4], q[5], q[6], q[7]};
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
        const int k = c >= C ? 1 : 0, cc = c - k * C;
        const float* p = red + (int64_t)k * rpi * C + cc;
        float t = 0.f;
        for (int r = 0; r < rpi; ++r) t += p[(int64_t)r * C];
        chs[k][cc] = t;
    }
    __syncthreads();
}

// Hand-off of the partials between the workgroups of a sample (MI355X_MICROARCH.md, inter-workgroup visibility: the per-XCD
// L2s are not coherent and an agent-scope release / acquire costs a write-back / invalidate of a cache full of streaming
// rows, so neither is used): every partial is stored `sc1` (write-through) by ONE wave, whole 128-byte lines per store
// instruction; that wave drains its stores (vmcnt(0)), the workgroup barrier orders the drain before lane 0's relaxed
// agent-scope add on the sample's counter; lane 0 polls the counter with relaxed sc1 loads (bounded: a trap, not a hang, if
// the partners never arrive), a workgroup barrier follows the poll, and every load of the partials is an sc1 load.
__device__ __forceinline__ void gnc_publish_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void gnc_arrive(unsigned* counter) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gnc_wait(const unsigned* counter, int nch) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nch) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) __builtin_trap();
        }
    }
}
// the same for a wave that does nothing else (lane = index within the wave)
__device__ __forceinline__ void gnc_arrive_lane(unsigned* counter, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gnc_wait_lane(const unsigned* counter, int nch, int lane) {
    if (lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) __builtin_trap();
        }
    }
}
__device__ __forceinline__ void gnc_arrive_and_wait(unsigned* counter, int nch) {
    gnc_arrive(counter);
    gnc_wait(counter, nch);
}
// the sample's partials [nch][2][64] -> LDS (one sc1 load per lane and step, all in flight together)
__device__ __forceinline__ void gnc_fetch_partials(const float* part_b, int nch, int G, float* buf) {
    for (int i = threadIdx.x; i < nch * 128; i += blockDim.x)
        if ((i & 63) < G) buf[i] = gnc_peek(part_b + i);
    __syncthreads();
}

// Forward: 8 data waves + 1 SYNC wave per workgroup (blockDim = GNC_NT + 64), four register sets in a ring.
// Iteration j of the data waves:   BAR_C | phase 2 of item j-2 (stores) | loads of item j+2 into the registers just freed |
//                                  sums of item j+1 -> LDS | BAR_A | column sums -> chs | BAR_B
// so a chunk's loads have a whole iteration to land and its statistics two to come back.  The sync wave holds no rows: after
// BAR_B it folds chs into the group partials of item j+1 and stores them (sc1, no wait); between BAR_C and BAR_A of iteration j it
// polls the partials of item j-1 -- its own were stored an iteration ago, the partners' most likely too: one round trip -- and
// writes the statistics to st[(j-1) & 1] for the phase 2 of iteration j+1, while the data waves compute and stream.  Every wave
// executes the same barrier sequence.  Order: a workgroup publishes item j only after its poll for item j-2 has returned, and
// items j-1, j-2 of a workgroup lie in strictly earlier samples than item j (grid >= nch), so the induction of the header
// (a sample's partials need only polls on earlier samples) holds.
__device__ __forceinline__ void gnc_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NV, bool SILU, bool FILM>
__global__ void __launch_bounds__(GNC_NT + 64)
gnc_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
               const float* __restrict__ scale, const float* __restrict__ shift, int64_t film_ld, bf16_t* __restrict__ y,
               float* __restrict__ mean, float* __restrict__ rstd, int HW, int C, int G, float eps, GncGeom gm,
               float* __restrict__ part /* [B][nch][2][64] */, unsigned* __restrict__ counter) {
    __shared__ __attribute__((aligned(16))) float red[2 * GNC_NT * 8];
    __shared__ float chs[2][GNC_CMAX];
    __shared__ float st_mu[2][64], st_rs[2][64];
#if GNC_PROF
    int prof_n = 0;
#endif
    const int t = threadIdx.x;
    const int cg = C / G;
    const int n = (gm.items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;       // items of this workgroup (>= 1)
    auto rows = [&](int j, int& b, int& ch, int& r0, int& r1) {
        const int item = blockIdx.x + j * gridDim.x;
        b = item / gm.nch;
        ch = item - b * gm.nch;
        r0 = ch * gm.R;
        r1 = r0 + gm.R < HW ? r0 + gm.R : HW;
    };
    if (t >= GNC_NT) {                                       // ---- the sync wave
        const int lane = t - GNC_NT, k = lane >> 5, g = lane & 31;
        auto publish = [&](int j) __attribute__((always_inline)) {          // two whole 128-byte lines, one store instruction
            int b, ch, r0, r1;
            rows(j, b, ch, r0, r1);
            float acc = 0.f;
            if (g < G)
                for (int jj = 0; jj < cg; ++jj) acc += chs[k][g * cg + jj];
            gnc_poke(part + ((int64_t)b * gm.nch + ch) * 128 + k * 64 + g, gnc_not_sentinel(acc));
        };
        gnc_bar();
        gnc_bar();
        publish(0);
        for (int j = 0; j <= n + 1; ++j) {
            gnc_bar();                                       // BAR_C
            if (j >= 1 && j - 1 < n) {                       // statistics of item j-1 (published an iteration ago), for phase 2 in j+1
                const int jj = j - 1;
                int b, ch, r0, r1;
                rows(jj, b, ch, r0, r1);
                // the partials are their own flags: the host filled the region with GNC_SENTINEL words before the launch; poll the
                // nch words of this lane's (quantity, group) column until none is a sentinel (all of a batch's loads in flight together)
                const float* p = part + (int64_t)b * gm.nch * 128 + k * 64 + g;
                double acc = 0.0;
                for (int c = 0; c < gm.nch; c += 32) {
                    float v[32];
                    unsigned spins = 0;
                    for (;;) {
                        bool ready = true;
#pragma unroll
                        for (int u = 0; u < 32; ++u) {
                            v[u] = c + u < gm.nch ? gnc_peek(p + (c + u) * 128) : 0.f;
                            ready = ready && __float_as_uint(v[u]) != GNC_SENTINEL;
                        }
                        if (__all(ready)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) __builtin_trap();
                    }
#pragma unroll
                    for (int u = 0; u < 32; ++u) acc += (double)v[u];
                }
                const double other = __shfl_xor(acc, 32, 64);
                if (lane < G) {
                    const double nn = (double)cg * HW, m = acc / nn;
                    double var = other / nn - m * m;
                    if (var < 0.0) var = 0.0;
                    const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
                    st_mu[jj & 1][lane] = mu;
                    st_rs[jj & 1][lane] = rs;
                    if (ch == 0) {
                        mean[b * G + lane] = mu;
                        rstd[b * G + lane] = rs;
                    }
                }
            }
            if (j + 1 < n) {
                gnc_bar();                                   // BAR_A
                gnc_bar();                                   // BAR_B
                publish(j + 1);
            }
        }
        return;
    }
    // ---- the data waves
    const bool live = t < gm.nt;
    const int oct = live ? t % gm.C8 : 0, rl = live ? t / gm.C8 : 0;
    const int c0_ = oct * 8, rl_ = rl;
    // per-lane addresses derived from (c0, rl) are loop invariants the compiler would keep in registers (and then spill: a spill
    // reload is a memory instruction that queues behind the streaming loads); every stage re-derives them from an opaque copy
#define GNC_LOCALS        \
    int c0 = c0_, rl = rl_; \
    asm volatile("" : "+v"(c0), "+v"(rl));
    auto load_item = [&](int j, gnc_u32x4 (&xv)[NV]) __attribute__((always_inline)) {
        GNC_LOCALS
        int b, ch, r0, r1;
        rows(j, b, ch, r0, r1);
        const bf16_t* xs = x + (int64_t)b * HW * C;          // uniform; per-lane offsets within a sample fit 32 bits
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r = r0 + rl + i * gm.rpi;
            r = r < r1 ? r : r1 - 1;
            xv[i] = gnc_ld(xs + (unsigned)(r * C + c0));
        }
    };
    auto sums = [&](int j, gnc_u32x4 (&xv)[NV]) __attribute__((always_inline)) {
        GNC_LOCALS
        int b, ch, r0, r1;
        rows(j, b, ch, r0, r1);
        float s[8], q[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) s[jj] = q[jj] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = live && r0 + rl + i * gm.rpi < r1;
            float xf[8];
            gnc_unpack(xv[i], xf);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float v = ok ? xf[jj] : 0.f;
                s[jj] += v;
                q[jj] += v * v;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) GNC_FORGET(xv[i]);
        if (live) {
            float* w0 = red + rl * C + c0;
            float* w1 = red + (gm.rpi + rl) * C + c0;
            *reinterpret_cast<f32x4*>(w0) = f32x4{s[0], s[1], s[2], s[3]};
            *reinterpret_cast<f32x4*>(w0 + 4) = f32x4{s[4], s[5], s[6], s[7]};
            *reinterpret_cast<f32x4*>(w1) = f32x4{q[0], q[1], q[2], q[3]};
            *reinterpret_cast<f32x4*>(w1 + 4) = f32x4{q[4], q[5], q[6], q[7]};
        }
    };
    auto colsums = [&]() __attribute__((always_inline)) {
        for (int c = t; c < 2 * C; c += GNC_NT) {
            const int k = c >= C ? 1 : 0, cc = c - k * C;
            const float* p = red + k * gm.rpi * C + cc;
            float acc = 0.f;
            for (int r = 0; r < gm.rpi; ++r) acc += p[r * C];
            chs[k][cc] = acc;
        }
    };
    auto phase2 = [&](int j, gnc_u32x4 (&xv)[NV]) __attribute__((always_inline)) {
        GNC_LOCALS
        int b, ch, r0, r1;
        rows(j, b, ch, r0, r1);
        bf16_t* ys = y + (int64_t)b * HW * C;
        float a1[8], b1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int g = (c0 + jj) / cg;
            const float mu = st_mu[j & 1][g], rs = st_rs[j & 1][g], gaj = gamma[c0 + jj];
            a1[jj] = rs * gaj;
            b1[jj] = beta[c0 + jj] - mu * rs * gaj;
            if (FILM) {                                      // (x a + b)(1 + scale) + shift as one multiply-add
                const float sc = 1.f + scale[(int64_t)b * film_ld + c0 + jj];
                a1[jj] *= sc;
                b1[jj] = b1[jj] * sc + shift[(int64_t)b * film_ld + c0 + jj];
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = r0 + rl + i * gm.rpi;
            float xf[8], o[8];
            gnc_unpack(xv[i], xf);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                float v = xf[jj] * a1[jj] + b1[jj];
                if (SILU) v = gnc_silu(v);
                o[jj] = v;
            }
            if (live && r < r1) gnc_st(ys + (unsigned)(r * C + c0), o);
        }
    };
    gnc_u32x4 S0[NV], S1[NV], S2[NV], S3[NV];
    load_item(0, S0);
    if (1 < n) load_item(1, S1);
    sums(0, S0);
    gnc_bar();
    colsums();
    gnc_bar();
#define GNC_IT(j, SP, S1N)                       \
    {                                            \
        gnc_bar();                               \
        GNC_STAMP(1);                            \
        if ((j) >= 2) phase2((j)-2, SP);         \
        GNC_STAMP(2);                            \
        if ((j) + 2 < n) load_item((j) + 2, SP); \
        GNC_STAMP(3);                            \
        if ((j) + 1 < n) {                       \
            sums((j) + 1, S1N);                  \
            GNC_STAMP(4);                        \
            gnc_bar();                           \
            colsums();                           \
            gnc_bar();                           \
            GNC_STAMP(5);                        \
        }                                        \
        if ((j) >= n + 1) break;                 \
    }
    for (int j = 0;; j += 4) {
        GNC_IT(j, S2, S1)
        GNC_IT(j + 1, S3, S2)
        GNC_IT(j + 2, S0, S3)
        GNC_IT(j + 3, S1, S0)
    }
#undef GNC_IT
#undef GNC_LOCALS
}

// Backward.  With d = dout * act'(n2) the four per-channel sums of the streaming kernels are functions of two:
//   sd = sum d,  sx = sum d * (x - mu):   Bs = (1+scale) sd,  A = (1+scale) rstd sx,  DH = sd,  DS = gamma rstd sx + beta sd
// and the per-row formula dx = rstd (dn1 gamma - S1/N - xhat S2/N) is  P d - K2 - K3 (x - mu)  with
//   P = rstd gamma (1+scale),  K2 = rstd S1/N,  K3 = rstd^2 S2/N,  n2 = x P + Q,  Q = (beta - mu rstd gamma)(1+scale) + shift
template <int NV, bool SILU, bool FILM, bool ADD>
__global__ void __launch_bounds__(GNC_NT)
gnc_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x, const float* __restrict__ mean,
               const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
               const float* __restrict__ scale, const float* __restrict__ shift, int64_t film_ld,
               const bf16_t* __restrict__ dx_add, bf16_t* __restrict__ dx, int HW, int C, int G, GncGeom gm,
               float* __restrict__ part1 /* [B][nch][2][64] */, float* __restrict__ part2 /* [B][nch][2][C] */,
               unsigned* __restrict__ counter) {
    __shared__ __attribute__((aligned(16))) float red[2 * GNC_NT * 8];
    __shared__ float chs[2][GNC_CMAX];
    __shared__ float st[2][64];
    const int t = threadIdx.x;
    const bool live = t < gm.nt;
    const int oct = live ? t % gm.C8 : 0, rl = live ? t / gm.C8 : 0;
    const int c0 = oct * 8, cg = C / G;
    const float invn = 1.f / ((float)cg * HW);
    for (int item = blockIdx.x; item < gm.items; item += gridDim.x) {
        const int b = item / gm.nch, ch = item - b * gm.nch;
        const int r0 = ch * gm.R, r1 = r0 + gm.R < HW ? r0 + gm.R : HW;
        const int64_t sample = (int64_t)b * HW * C;          // uniform; per-lane offsets within a sample fit 32 bits
        const bf16_t *xs = x + sample, *ds = dout + sample, *as = dx_add + sample;
        bf16_t* os = dx + sample;
        gnc_u32x4 xv[NV], dv[NV], av[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r = r0 + rl + i * gm.rpi;
            r = r < r1 ? r : r1 - 1;
            xv[i] = gnc_ld(xs + (unsigned)(r * C + c0));
            dv[i] = gnc_ld(ds + (unsigned)(r * C + c0));
        }
        float P[8], Q[8], mu[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (c0 + j) / cg;
            mu[j] = mean[b * G + g];
            const float rsj = rstd[b * G + g];
            const float gaj = gamma[c0 + j], bej = beta[c0 + j];
            const float s1 = FILM ? 1.f + scale[(int64_t)b * film_ld + c0 + j] : 1.f;
            const float shj = FILM ? shift[(int64_t)b * film_ld + c0 + j] : 0.f;
            P[j] = rsj * gaj * s1;
            Q[j] = (bej - mu[j] * rsj * gaj) * s1 + shj;
        }
        float sd[8], sx[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sd[j] = sx[j] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool ok = live && r0 + rl + i * gm.rpi < r1;
            float xf[8], df[8];
            gnc_unpack(xv[i], xf);
            gnc_unpack(dv[i], df);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = df[j];
                if (SILU) d *= gnc_silu_grad(xf[j] * P[j] + Q[j]);
                d = ok ? d : 0.f;
                sd[j] += d;
                sx[j] += d * (xf[j] - mu[j]);
            }
        }
        gnc_fold_chunk(sd, sx, red, chs, C, gm.rpi, rl, c0, live);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            GNC_FORGET(xv[i]);
            GNC_FORGET(dv[i]);
        }
        float* p2 = part2 + ((int64_t)b * gm.nch + ch) * 2 * C;
        for (int c = t; c < 2 * C; c += blockDim.x) p2[c] = chs[c >= C ? 1 : 0][c >= C ? c - C : c];
        float* mine = part1 + ((int64_t)b * gm.nch + ch) * 128;
        if (t < 64) {                                        // wave 0: two whole 128-byte lines, one store instruction
            const int k = t >> 5, g = t & 31;
            float acc = 0.f;
            if (g < G) {
                const float f = k ? rstd[b * G + g] : 1.f;
                for (int j = 0; j < cg; ++j) {
                    const int c = g * cg + j;
                    const float w = gamma[c] * (FILM ? 1.f + scale[(int64_t)b * film_ld + c] : 1.f);
                    acc += w * (f * chs[k][c]);
                }
            }
            gnc_poke(mine + k * 64 + g, acc);
        }
        gnc_publish_drain();
        __syncthreads();
        if (ADD) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                int r = r0 + rl + i * gm.rpi;
                r = r < r1 ? r : r1 - 1;
                av[i] = gnc_ld(as + (unsigned)(r * C + c0));
            }
        }
        if (gm.nch > 1) gnc_arrive_and_wait(counter + b, gm.nch);
        __syncthreads();
        gnc_fetch_partials(part1 + (int64_t)b * gm.nch * 128, gm.nch, G, red);
        if (t < 2 * G) {
            const int k = t >= G ? 1 : 0, g = t - k * G;
            float acc = 0.f;
            for (int c = 0; c < gm.nch; ++c) acc += red[c * 128 + k * 64 + g];
            st[k][g] = acc * invn;
        }
        __syncthreads();
        float K2[8], K3[8];                                  // dx = P d - K3 x - K2  (K2 takes the mean term: mu is dead from here)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (c0 + j) / cg;
            const float rsj = rstd[b * G + g], muj = mean[b * G + g];
            K3[j] = rsj * rsj * st[1][g];
            K2[j] = rsj * st[0][g] - K3[j] * muj;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = r0 + rl + i * gm.rpi;
            float xf[8], df[8], af[8], o[8];
            gnc_unpack(xv[i], xf);
            gnc_unpack(dv[i], df);
            if (ADD) gnc_unpack(av[i], af);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = df[j];
                if (SILU) d *= gnc_silu_grad(xf[j] * P[j] + Q[j]);
                float v = P[j] * d - K3[j] * xf[j] - K2[j];
                if (ADD) v += af[j];
                o[j] = v;
            }
            if (live && r < r1) gnc_st(os + (unsigned)(r * C + c0), o);
        }
        __syncthreads();
    }
}

// chunk partials -> the [4][B][C] per-(sample, channel) sums gn_bwd_group_kernel folds into dgamma / dbeta / FiLM rows
__global__ void gnc_bwd_fold_kernel(const float* __restrict__ part2, int nch, int B, int C, int G, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ scale, int64_t film_ld, float* __restrict__ sums) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, BC = (int64_t)B * C;
    if (i >= BC) return;
    const int b = (int)(i / C), c = (int)(i - (int64_t)b * C);
    const float* p = part2 + (int64_t)b * nch * 2 * C + c;
    float sd = 0.f, sx = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        sd += p[(int64_t)ch * 2 * C];
        sx += p[(int64_t)ch * 2 * C + C];
    }
    const float rs = rstd[b * G + c / (C / G)];
    const float s1 = scale ? 1.f + scale[(int64_t)b * film_ld + c] : 1.f;
    sums[i] = s1 * (rs * sx);
    sums[BC + i] = s1 * sd;
    sums[2 * BC + i] = gamma[c] * (rs * sx) + beta[c] * sd;
    sums[3 * BC + i] = sd;
}

static int gnc_grid(const void* kernel, int block, int items) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, block, 0) != hipSuccess || occ < 1) return 0;
    int dev = 0, cus = 0;
    hipDeviceProp_t prop;
    static int n_cus = 0;
    if (!n_cus) {
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cus = prop.multiProcessorCount;
        if (n_cus <= 0) n_cus = 256;
    }
    cus = n_cus;
    const int64_t g = (int64_t)occ * cus;
    return (int)(g < items ? g : items);
}
