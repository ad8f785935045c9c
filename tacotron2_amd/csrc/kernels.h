// HIP kernels of the Tacotron2 mel-generation path for gfx950 (MI355X, CDNA4).
//
// Data layouts
//  * "plain": row-major fp32, innermost dimension contiguous.
//  * "frag16": activation matrix X[rows, K] stored in the operand order of
//    v_mfma_f32_16x16x4_f32 so that one wave-instruction loads one contiguous 1 KiB piece:
//        float index = (((kb*MB + rb)*4 + kg)*16 + rr)*4 + j,   k = kb*16 + kg*4 + j,  row = rb*16 + rr
//    (MB = row blocks of 16 batch rows; every K segment padded to a multiple of 64; pad rows / pad k are zero).
//    Lane l = kg*16 + rr reads the float4 at ((kb*MB + rb)*64 + l)*4.
//  * packed weights for the skinny (batch <= 64) GEMMs: per workgroup g a contiguous slab
//        [kb][lane = kg*16 + c][j]  = W[k = kb*16 + kg*4 + j][col(g, c)]
//    so the weight stream of a workgroup is one linear run of 1 KiB wave-loads, read exactly once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef T2_SKINNY_WAVES
#define T2_SKINNY_WAVES 8
#endif
#define T2_SKINNY_THREADS (64 * T2_SKINNY_WAVES)
#define T2_MAX_SEG 3
#ifndef T2_W_NT
#define T2_W_NT 0   // plain loads measured faster than nontemporal ones here (tools/lstm_bench.hip)
#endif
#if T2_W_NT
#define T2_WLOAD(p) __builtin_nontemporal_load(p)
#else
#define T2_WLOAD(p) (*(p))
#endif

// ctl block (device ints): [0] base step, [1] nsteps_final (0 = not finished), [2] max steps,
// [3] min_iters, [4] stop rule enabled
#define CTL_BASE 0
#define CTL_FINAL 1
#define CTL_MAX 2
#define CTL_MINIT 3
#define CTL_STOP_EN 4

__host__ __device__ inline size_t frag_idx(int row, int k, int MB) {
    const int kb = k >> 4, kg = (k >> 2) & 3, j = k & 3, rb = row >> 4, rr = row & 15;
    return ((size_t)(kb * MB + rb) * 64 + kg * 16 + rr) * 4 + j;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Counter-based generator of the sampled training paths (PreNet / conv dropout, zoneout): 24 uniform bits per
// (seed, stream, step, row, column).  TensorFlow's stateful random ops cannot be reproduced; the oracle uses this same
// function (oracle/t2_oracle.py:rng_u01), TF's arithmetic around it.
__host__ __device__ inline uint32_t t2_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline float t2_u01(uint32_t seed, uint32_t stream, uint32_t t, uint32_t row, uint32_t col) {
    uint32_t h = t2_mix32(seed ^ (stream * 0x9E3779B9u));
    h = t2_mix32(h ^ (t * 0x85EBCA6Bu + 0x165667B1u));
    h = t2_mix32(h ^ (row * 0xC2B2AE35u + 0x27D4EB2Fu));
    h = t2_mix32(h ^ (col * 0x9E3779B1u + 0x85EBCA77u));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// tf.nn.dropout(x, keep_prob) of TF 1.11/1.12: x / keep_prob * floor(keep_prob + uniform)
__device__ __forceinline__ float dropout_train(float x, float rate, float u) { return (u >= rate) ? x / (1.0f - rate) : 0.0f; }
#define T2_STREAM_PRENET 1
#define T2_STREAM_ZONE_ATT 10
#define T2_STREAM_ZONE_L1 12
#define T2_STREAM_ZONE_L2 14

// returns the step index of this launch or -1 when the launch must do nothing
__device__ __forceinline__ int step_index(const int* __restrict__ ctl, int step_off) {
    if (ctl == nullptr) return step_off;
    const int t = ctl[CTL_BASE] + step_off;
    const int fin = ctl[CTL_FINAL];
    if (t >= ctl[CTL_MAX] || (fin != 0 && t >= fin)) return -1;
    return t;
}

// A-operand source: frag16 buffer with time slots.  address = base + ((t + dt) % nslots) * slot_stride
struct FragBuf {
    float* base;
    int slot_stride;     // floats
    int nslots;          // 1: single buffer, 2: ping-pong state, T: time-indexed input
    int dt;              // slot = (t + dt) % nslots
    int koff;            // first k (multiple of 16 for sources; any for destinations)
};

__device__ __forceinline__ float* frag_slot(const FragBuf& f, int t) {
    return f.base + (size_t)((t + f.dt) % f.nslots) * f.slot_stride;
}

struct SkinnyCore {
    const float* w[2];        // packed weights per direction (blockIdx.y)
    const float* bias[2];     // LSTM: packed [n_wg][16]; dense: [N] (may be null)
    FragBuf seg[2][2];        // per direction: k-blocks [0,kb0) from seg[.][0], [kb0,kbtot) from seg[.][1]
    int kb0;                  // k-blocks (of this launch) served by seg[.][0]
    int kbtot;                // k-blocks of this launch
    int kb_w0;                // first k-block of the weight slab used by this launch
    int kb_full;              // k-blocks per workgroup slab (slab stride)
    const int* ctl;
    int step_off;
    int t_hint;               // >= 0: step index known to select the same operand slots as the real one (lets the
                              // activation loads start before ctl arrives); -1: wait for ctl
    int B;                    // valid rows
};

struct LstmEpi {
    int H;                    // hidden units
    float* c[2];              // plain [MB*16][Hld]
    int Hld;
    FragBuf hprev[2];         // previous state h (for the zoneout blend), per direction
    FragBuf hstate[2];        // new state h destination
    FragBuf hraw[2];          // raw output destination (base may be null)
    float zc, zh, fb;
    const int* lens;          // optional sequence lengths (encoder), device int32 [B]
    float* plain_out;         // optional plain output [B][T][ld]
    int plain_bs, plain_ts;   // batch / time strides (floats)
    int plain_off[2];         // column offset per direction
    int reverse[2];           // direction writes to position len-1-t
    const float* zpart;       // optional partial pre-activations of the recurrent K range (partial_step_kernel), D-fragment layout
    const float* zpart2;      // optional second partial (a second hoisted K range), same layout
    float* y_all;             // optional plain [S][B][H] store of the raw output of every step (deferred projection)
    // fused partial query (attention LSTM only): qpart[wg][row][u] = sum_uu h[row][4wg+uu] * wq[4wg+uu][u]
    const float* wq;          // [A][U] or null
    float* qpart;             // [n_wg][MB*16][U]
    int U;
    // training-mode forward (ZoneoutLSTMCell, modules.py:274-276) and the tape of the backward pass (all may be null):
    int train;                // 1: sampled zoneout masks from t2_u01(seed, stream | stream + 1, t, row, unit)
    uint32_t seed;
    int stream;
    float* tp_z;              // [S][B][4H] gate pre-activations (i | j | f | o), bias included
    float* tp_cprev;          // [S][B][H] cell state before the step
    float* tp_cnew;           // [S][B][H] new cell value before zoneout
    float* tp_hraw;           // [S][B][H] raw output h
    float* tp_hstate;         // [S][B][H] zoned state h
};

struct DenseEpi {
    int N;                    // valid columns
    int relu;
    FragBuf out_frag;         // frag16 destination (k = koff + col - frag_col0), base may be null
    int frag_col0;
    float* out_plain;         // plain [row][ld] (+ t * plain_t_stride), may be null
    int plain_ld;
    long long plain_t_stride;
    int plain_ncol;           // columns < plain_ncol go to out_plain
    // stop token handling (projection kernel): column stop_col holds the stop logit
    int n_wg;               // workgroups (16-column tiles) of this dense layer
    int stop_col;             // -1: none
    float* out_stop;          // [row * stop_ld + t * stop_t_mul]
    int stop_ld;
    int stop_t_mul;
    int* ctl_w;               // writable ctl (same block as core.ctl)
    float drop_rate;          // > 0: tf.layers.dropout after the activation (PreNet, modules.py:90), training mode
    uint32_t seed;
    int stream;
    // fused first PreNet layer of the NEXT step (free-running decode): columns [fuse_col0, fuse_col0 + fuse_n) hold
    // y . (W_mel[:, fed frames] . W_prenet0) + (b_mel[fed frames] . W_prenet0 + b_prenet0); relu, frag16 destination
    int fuse_col0, fuse_n;
    FragBuf fuse_frag;
};

// ------------------------------------------------------------------------------------------------
// skinny GEMM core: acc[rb] (16x16 tile, rows = batch rows of block rb, cols = this workgroup's 16
// columns) += X[:, k-range of this wave] . W[k-range, cols].  The K range is split over the 8 waves
// of the workgroup; every wave streams its part of the packed weight slab once, straight to
// registers (no LDS round trip: the operand is not shared between waves), in register-double-
// buffered groups of G k-blocks so that the next group's loads are in flight under the MFMAs.
// ------------------------------------------------------------------------------------------------
template <int MB, int G>
struct KGroup {
    f32x4 b[G];
    f32x4 a[G][MB];
};

template <int MB, int G>
__device__ __forceinline__ void kg_load_w(KGroup<MB, G>& g, const f32x4* __restrict__ wp, int kb) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
#ifdef T2_ABL_NOW   // ablation builds (tools/lstm_bench.hip) only
        g.b[u] = f32x4{0.01f, 0.02f, 0.03f, 0.04f};
#else
        g.b[u] = T2_WLOAD(wp + (size_t)(kb + u) * 64);
#endif
    }
}

template <int MB, int G>
__device__ __forceinline__ void kg_load_a(KGroup<MB, G>& g, const f32x4* __restrict__ ap, int kba) {
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) {
#ifdef T2_ABL_NOA
            g.a[u][rb] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
#else
            g.a[u][rb] = ap[(size_t)((kba + u) * MB + rb) * 64];
#endif
        }
}

template <int MB, int G>
__device__ __forceinline__ void kg_mma(const KGroup<MB, G>& g, f32x4 (&acc)[MB]) {
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) {
#if defined(T2_ABL_NOMMA)
                asm volatile("" ::"v"(g.a[u][rb][j]), "v"(g.b[u][j]));
#elif defined(T2_ABL_DECOUPLE)   // MFMAs on constants, loads kept alive: do loads and MFMAs overlap at all?
                asm volatile("" ::"v"(g.a[u][rb][j]), "v"(g.b[u][j]));
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(0.5f + j, 0.25f + u, acc[rb], 0, 0, 0);
#else
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.a[u][rb][j], g.b[u][j], acc[rb], 0, 0, 0);
#endif
            }
}

// The wave's share of the K range is [lo, hi) in groups of G k-blocks (kb0 and kbtot are multiples of G:
// the host pads every K segment to 64).  A ring of NBUF register buffers keeps NBUF groups of loads in
// flight under the MFMAs.  The first weight loads are issued before the step index is known (they do
// not depend on it); the walk through the range starts at a workgroup-dependent rotation so that the
// 256 workgroups do not all request the same activation lines from L2 at the same moment.
// Returns the step index, or -1 when this launch must do nothing (uniform over the workgroup).
// NPW > 0: every wave owns exactly NPW or NPW + 1 groups (host-checked); the walk over them is then
// fully unrolled straight-line code, so the compiler emits exact counted s_waitcnt vmcnt(N) and a wave
// only ever waits for its OLDEST group in flight.  (With a runtime trip count hipcc drains vmcnt(0) at
// the loop header and the loads stop overlapping the MFMAs: measured 22.8 vs ~13 us.)
#ifndef T2_FENCE
#define T2_FENCE 0
#endif
#if T2_FENCE
#define T2_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define T2_SCHED_FENCE() do { } while (0)
#endif
template <int NBUF, int N, class Buf, class Mma, class LoadW, class LoadA, class GIdx>
__device__ __forceinline__ void ring_static(Buf (&buf)[NBUF], Mma&& mma, LoadW&& load_w, LoadA&& load_a, GIdx&& gidx) {
#pragma unroll
    for (int g = 0; g < N; ++g) {
        mma(buf[g % NBUF]);
        T2_SCHED_FENCE();
        if (g + NBUF < N) {
            const int gi = gidx(g + NBUF);
            load_w(buf[g % NBUF], gi);
            load_a(buf[g % NBUF], gi);
            T2_SCHED_FENCE();
        }
    }
}

#ifndef T2_G4
#define T2_G4 2      // k-blocks per group at MB = 4
#endif
#ifndef T2_NBUF
#define T2_NBUF 3    // groups in flight per wave (3 measured best: tools/lstm_bench.hip)
#endif
template <int MB, int NPW, class Pre>
__device__ __forceinline__ int skinny_core(const SkinnyCore& a, int dir, int wg, f32x4 (&acc)[MB], Pre&& pre) {
    constexpr int G = (MB >= 4) ? T2_G4 : 4;
    constexpr int NBUF = T2_NBUF;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ng = a.kbtot / G, ng0 = a.kb0 / G;
    const int base = ng / T2_SKINNY_WAVES, rem = ng % T2_SKINNY_WAVES;
    const int lo = wave * base + min(wave, rem);
    const int n = base + (wave < rem ? 1 : 0);
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.w[dir]) + ((size_t)wg * a.kb_full + a.kb_w0) * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rot = n > 0 ? (int)(((unsigned)wg * 7u + (unsigned)dir * 3u) % (unsigned)n) : 0;
    auto gidx = [&](int i) { const int j = i + rot; return lo + (j >= n ? j - n : j); };
    auto load_w = [&](KGroup<MB, G>& b, int g) { kg_load_w<MB, G>(b, wp, g * G); };
    KGroup<MB, G> buf[NBUF];
    if (n > 0) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) load_w(buf[i], gidx(min(i, n - 1)));
    }
    // activation prologue: with a slot hint it is issued before ctl is back
    const f32x4* ap0 = nullptr;
    const f32x4* ap1 = nullptr;
    auto set_ap = [&](int ts) {
        ap0 = reinterpret_cast<const f32x4*>(frag_slot(a.seg[dir][0], ts)) + (size_t)(a.seg[dir][0].koff >> 4) * MB * 64 + lane;
        ap1 = reinterpret_cast<const f32x4*>(frag_slot(a.seg[dir][1], ts)) + (size_t)(a.seg[dir][1].koff >> 4) * MB * 64 + lane;
    };
    auto load_a = [&](KGroup<MB, G>& b, int g) {
        const f32x4* ap = g < ng0 ? ap0 + (size_t)g * G * MB * 64 : ap1 + (size_t)(g - ng0) * G * MB * 64;
        kg_load_a<MB, G>(b, ap, 0);
    };
    auto prologue_a = [&]() {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) load_a(buf[i], gidx(min(i, n - 1)));
    };
    const bool hinted = a.t_hint >= 0;
    if (hinted && n > 0) { set_ap(a.t_hint); prologue_a(); }
    const int t = step_index(a.ctl, a.step_off);
    if (t < 0) return -1;
    pre(t);                      // the caller's epilogue-operand prefetch
    if (n > 0) {
        const bool same = hinted && frag_slot(a.seg[dir][0], t) == frag_slot(a.seg[dir][0], a.t_hint) &&
                          frag_slot(a.seg[dir][1], t) == frag_slot(a.seg[dir][1], a.t_hint);
        if (!same) { set_ap(t); prologue_a(); }     // no hint, or (host bug) a hint that selects other slots
        bool done = false;
        if constexpr (NPW > 0) {
            if (base == NPW) {      // checked on the device too: a wrong NPW must never walk out of the slab
                auto mma = [&](const KGroup<MB, G>& b) { kg_mma<MB, G>(b, acc); };
                if (n == NPW) ring_static<NBUF, NPW>(buf, mma, load_w, load_a, gidx);
                else ring_static<NBUF, NPW + 1>(buf, mma, load_w, load_a, gidx);
                done = true;
            }
        }
        if (!done) {
            for (int g = 0; g < n; g += NBUF) {
#pragma unroll
                for (int i = 0; i < NBUF; ++i) {
                    if (g + i < n) {
                        kg_mma<MB, G>(buf[i], acc);
                        if (g + i + NBUF < n) {
                            const int gi = gidx(g + i + NBUF);
                            load_w(buf[i], gi);
                            load_a(buf[i], gi);
                        }
                    }
                }
            }
        }
    }
    return t;
}

// sum of the 8 waves' partial tiles for element (row, col c) of the workgroup's [MB*16 x 16] tile.
// D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg.
template <int MB>
__device__ __forceinline__ float red_sum(const f32x4* red, int row, int c) {
    const int rb = row >> 4, rr = row & 15;
    const int src_lane = (rr >> 2) * 16 + c, reg = rr & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < T2_SKINNY_WAVES; ++w) s += red[(w * MB + rb) * 64 + src_lane][reg];
    return s;
}

// ------------------------------------------------------------------------------------------------
// LSTM step: gates = [x | h_prev] . kernel + bias for 4 hidden units per workgroup (columns
// gate*4 + u), fused with the LSTMCell pointwise math, the inference-mode
// zoneout blend of ZoneoutLSTMCell (reference tacotron/modules.py:261-283) and, for the attention
// LSTM, this workgroup's share of the attention query projection (tacotron_v2.py:122).
// ------------------------------------------------------------------------------------------------
template <int MB, bool ENC, int NPW = 0>   // ENC: encoder BiLSTM instance (own name in profiles; same code)
__global__ __launch_bounds__(T2_SKINNY_THREADS) void lstm_step_kernel(SkinnyCore a, LstmEpi e) {
    const int dir = blockIdx.y;
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    __shared__ float s_h[MB * 16 * 4];
    f32x4 acc[MB];
    // epilogue operands of thread (row, unit): requested as soon as the step index is known, used after the GEMM
    const int idx = threadIdx.x;
    const int row = idx >> 2, uu = idx & 3;
    const int unit = blockIdx.x * 4 + uu;
    const bool owner = idx < MB * 16 * 4 && row < a.B && unit < e.H;
    float cp = 0.f, hp = 0.f, zp4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
    int len = 0;
    const int t = skinny_core<MB, NPW>(a, dir, blockIdx.x, acc, [&](int tt) {
        if (owner) {
            cp = e.c[dir][(size_t)row * e.Hld + unit];
            hp = frag_slot(e.hprev[dir], tt)[frag_idx(row, e.hprev[dir].koff + unit, MB)];
            const float* bias = a.bias[dir] + (size_t)blockIdx.x * 16;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) b4[g4] = bias[g4 * 4 + uu];
            if (e.zpart != nullptr) {
                const float* zp = e.zpart + ((size_t)(blockIdx.x * MB + (row >> 4)) * 64 + ((row & 15) >> 2) * 16) * 4 + (row & 3);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) zp4[g4] = zp[(g4 * 4 + uu) * 4];
                if (e.zpart2 != nullptr) {
                    const float* zq = e.zpart2 + (zp - e.zpart);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) zp4[g4] += zq[(g4 * 4 + uu) * 4];
                }
            }
            if (e.lens != nullptr) len = e.lens[row];
        }
    });
    if (t < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    __syncthreads();
    if (idx < MB * 16 * 4) {
        float hn = 0.f;
        if (owner) {
            const float zi = red_sum<MB>(red, row, 0 * 4 + uu) + b4[0] + zp4[0];
            const float zj = red_sum<MB>(red, row, 1 * 4 + uu) + b4[1] + zp4[1];
            const float zf = red_sum<MB>(red, row, 2 * 4 + uu) + b4[2] + zp4[2];
            const float zo = red_sum<MB>(red, row, 3 * 4 + uu) + b4[3] + zp4[3];
            const float cn = sigmoidf_(zf + e.fb) * cp + sigmoidf_(zi) * tanhf(zj);
            hn = sigmoidf_(zo) * tanhf(cn);
            float cs = (1.0f - e.zc) * cn + e.zc * cp;
            float hs = (1.0f - e.zh) * hn + e.zh * hp;
            if (e.train) {                                    // modules.py:274-276
                cs = (1.0f - e.zc) * dropout_train(cn - cp, e.zc, t2_u01(e.seed, e.stream + 2 * dir, t, row, unit)) + cp;
                hs = (1.0f - e.zh) * dropout_train(hn - hp, e.zh, t2_u01(e.seed, e.stream + 2 * dir + 1, t, row, unit)) + hp;
            }
            bool live = true;
            if (e.lens != nullptr) {
                live = t < len;
                if (!live) { cs = cp; hs = hp; hn = 0.f; }   // dynamic_rnn: zero output, state copied through
            }
            if (e.tp_z != nullptr) {
                const size_t ti = ((size_t)t * a.B + row);
                float* tz = e.tp_z + ti * 4 * e.H;
                tz[unit] = zi; tz[e.H + unit] = zj; tz[2 * e.H + unit] = zf; tz[3 * e.H + unit] = zo;
                e.tp_cprev[ti * e.H + unit] = cp; e.tp_cnew[ti * e.H + unit] = cn;
                if (e.tp_hraw != nullptr) e.tp_hraw[ti * e.H + unit] = hn;
                e.tp_hstate[ti * e.H + unit] = hs;
            }
            e.c[dir][(size_t)row * e.Hld + unit] = cs;
            frag_slot(e.hstate[dir], t)[frag_idx(row, e.hstate[dir].koff + unit, MB)] = hs;
            if (e.hraw[dir].base != nullptr) frag_slot(e.hraw[dir], t)[frag_idx(row, e.hraw[dir].koff + unit, MB)] = hn;
            if (e.y_all != nullptr) e.y_all[((size_t)t * a.B + row) * e.H + unit] = hn;
            if (e.plain_out != nullptr && live) {
                const int tpos = e.reverse[dir] ? (len - 1 - t) : t;
                e.plain_out[(size_t)row * e.plain_bs + (size_t)tpos * e.plain_ts + e.plain_off[dir] + unit] = hn;
            }
        }
        if (!ENC) s_h[idx] = hn;
    }
    if (!ENC) {
        if (e.wq == nullptr) return;
        __syncthreads();
        // thread -> (row group, unit u): the 4 query-layer rows of this workgroup's hidden units stay in
        // registers; only the valid batch rows are produced (the attention kernel reads rows < B)
        const int U = e.U;
        const int RG = max(1, T2_SKINNY_THREADS / U);
        if ((int)threadIdx.x < RG * U) {
            const int rg = threadIdx.x / U, u = threadIdx.x - rg * U;
            const float* wq = e.wq + (size_t)blockIdx.x * 4 * U + u;
            const int nvalid = min(4, e.H - (int)blockIdx.x * 4);
            float wv[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) wv[uu] = uu < nvalid ? wq[uu * U] : 0.f;
            float* qp = e.qpart + (size_t)blockIdx.x * MB * 16 * U + u;
            for (int row = rg; row < a.B; row += RG) {
                float sq = 0.f;
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) sq = fmaf(s_h[row * 4 + uu], wv[uu], sq);
                qp[(size_t)row * U] = sq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Recurrent-half pre-activations: z_part = [state operands of step t] . kernel[their K range].  These
// operands are known as soon as the previous step's producer has run, so this launch sits on a side
// branch of the step graph and overlaps the attention / prenet / projection chain; the LSTM step then
// only multiplies the operands produced in its own step and adds z_part (LstmEpi::zpart).
// Output: the summed 16x16 tiles in D-fragment layout, [wg][rb][lane] float4.
// ------------------------------------------------------------------------------------------------
template <int MB, int NPW>
__device__ __forceinline__ void partial_body(const SkinnyCore& a, float* __restrict__ zout, int wg, f32x4* red) {
    f32x4 acc[MB];
    const int t = skinny_core<MB, NPW>(a, 0, wg, acc, [](int) {});
    if (t < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    __syncthreads();
    for (int i = threadIdx.x; i < MB * 64; i += T2_SKINNY_THREADS) {
        f32x4 sacc = red[i];
#pragma unroll
        for (int w = 1; w < T2_SKINNY_WAVES; ++w) sacc += red[w * MB * 64 + i];
        reinterpret_cast<f32x4*>(zout)[(size_t)wg * MB * 64 + i] = sacc;
    }
}

template <int MB, int NPW = 0>
__global__ __launch_bounds__(T2_SKINNY_THREADS) void partial_step_kernel(SkinnyCore a, float* __restrict__ zout) {
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    partial_body<MB, NPW>(a, zout, blockIdx.x, red);
}

// Up to three recurrent-half jobs that ride in front of a small kernel's grid (horizontal fusion): the
// attention step (one CU per batch row) and the projection step (a dozen workgroups) leave most of the
// chip idle, and kernels of different streams / graph branches were measured not to overlap on this
// stack, so the independent GEMM work is put into the SAME launch: the host kernel's own workgroups
// come first (they are the critical path), then n0 workgroups of job 0 and n1 of job 1.
struct PartialJobs {
    SkinnyCore core[3];
    float* z[3];
    int n[3];
    int off[3];      // first column tile of the job served by this launch (a job may be spread over several launches)
    int skip;        // dense_step_kernel: workgroups below skip take no riding tile (the dense role's own workgroups)
};

// ------------------------------------------------------------------------------------------------
// Split-K LSTM step for batches > 16 rows.  With 4 units (16 columns) per workgroup every CU has to
// pull the whole activation matrix through its L1 (B*K*4 bytes, ~0.65 MB at B=64), which, not the
// MFMA pipe, bounds the kernel.  Here a workgroup owns 8 units (two 16-column tiles) and HALF of K:
// activation bytes per CU halve, the weight bytes stay.  The two K-halves of a tile are combined
// inside the launch: both workgroups store their partial gate pre-activations write-through
// (sc1), draw a ticket with a returning agent-scope atomic, and the workgroup whose ticket is last
// reads the other half with sc1 loads and runs the LSTM epilogue (cdna guide, "in-launch split-K
// reduction", sc1 form; every store and every load of the handed-off bytes is an agent-scope
// relaxed atomic access = global_store/load ... sc1).  The sum is always half0 + half1, so the
// result does not depend on which workgroup arrives last.
// ------------------------------------------------------------------------------------------------
template <int MB, int G>
struct KGroup2 {
    f32x4 b0[G], b1[G];
    f32x4 a[G][MB];
};

template <int MB, int NPW = 0>
__global__ __launch_bounds__(T2_SKINNY_THREADS) void lstm_step_splitk_kernel(SkinnyCore a, LstmEpi e, int n_wg4,
                                                                              unsigned* tickets, unsigned long long* slabs) {
    constexpr int G = 2, NBUF = 3;
    const int tile = blockIdx.x >> 1, half = blockIdx.x & 1;
    __shared__ f32x4 red[T2_SKINNY_WAVES * 2 * MB * 64];
    __shared__ float s_h[MB * 16 * 8];
    __shared__ unsigned s_ticket;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ng = a.kbtot / G, ng0 = a.kb0 / G;
    const int h_lo = half == 0 ? 0 : (ng + 1) / 2, h_n = half == 0 ? (ng + 1) / 2 : ng / 2;
    const int base = h_n / T2_SKINNY_WAVES, rem = h_n % T2_SKINNY_WAVES;
    const int lo = h_lo + wave * base + min(wave, rem);
    const int n = base + (wave < rem ? 1 : 0);
    const int t0i = 2 * tile, t1i = min(2 * tile + 1, n_wg4 - 1);
    const f32x4* __restrict__ wp0 = reinterpret_cast<const f32x4*>(a.w[0]) + ((size_t)t0i * a.kb_full + a.kb_w0) * 64 + lane;
    const f32x4* __restrict__ wp1 = reinterpret_cast<const f32x4*>(a.w[0]) + ((size_t)t1i * a.kb_full + a.kb_w0) * 64 + lane;
    f32x4 acc0[MB], acc1[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) { acc0[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int rot = n > 0 ? (int)((tile * 5u) % (unsigned)n) : 0;
    auto gidx = [&](int i) { const int j = i + rot; return lo + (j >= n ? j - n : j); };
    KGroup2<MB, G> buf[NBUF];
    auto load_w = [&](KGroup2<MB, G>& b, int g) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            b.b0[u] = T2_WLOAD(wp0 + (size_t)(g * G + u) * 64);
            b.b1[u] = T2_WLOAD(wp1 + (size_t)(g * G + u) * 64);
        }
    };
    if (n > 0) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i) load_w(buf[i], gidx(min(i, n - 1)));
    }
    const int t = step_index(a.ctl, a.step_off);
    if (t < 0) return;
    if (n > 0) {
        const f32x4* ap0 = reinterpret_cast<const f32x4*>(frag_slot(a.seg[0][0], t)) + (size_t)(a.seg[0][0].koff >> 4) * MB * 64 + lane;
        const f32x4* ap1 = reinterpret_cast<const f32x4*>(frag_slot(a.seg[0][1], t)) + (size_t)(a.seg[0][1].koff >> 4) * MB * 64 + lane;
        auto load_a = [&](KGroup2<MB, G>& b, int g) {
            const f32x4* ap = g < ng0 ? ap0 + (size_t)g * G * MB * 64 : ap1 + (size_t)(g - ng0) * G * MB * 64;
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) b.a[u][rb] = ap[(size_t)(u * MB + rb) * 64];
        };
        auto mma = [&](const KGroup2<MB, G>& b) {
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb) {
                        acc0[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.a[u][rb][j], b.b0[u][j], acc0[rb], 0, 0, 0);
                        acc1[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.a[u][rb][j], b.b1[u][j], acc1[rb], 0, 0, 0);
                    }
        };
#pragma unroll
        for (int i = 0; i < NBUF; ++i) load_a(buf[i], gidx(min(i, n - 1)));
        bool done = false;
        if constexpr (NPW > 0) {
            if (base == NPW) {
                if (n == NPW) ring_static<NBUF, NPW>(buf, mma, load_w, load_a, gidx);
                else ring_static<NBUF, NPW + 1>(buf, mma, load_w, load_a, gidx);
                done = true;
            }
        }
        if (!done) {
            for (int g = 0; g < n; g += NBUF) {
#pragma unroll
                for (int i = 0; i < NBUF; ++i) {
                    if (g + i < n) {
                        mma(buf[i]);
                        if (g + i + NBUF < n) {
                            const int gi = gidx(g + i + NBUF);
                            load_w(buf[i], gi);
                            load_a(buf[i], gi);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) {
        red[((wave * 2 + 0) * MB + rb) * 64 + lane] = acc0[rb];
        red[((wave * 2 + 1) * MB + rb) * 64 + lane] = acc1[rb];
    }
    __syncthreads();
    // this half's partial pre-activations: thread -> (row, unit-in-tile), 4 gates
    const int idx = threadIdx.x;
    const bool owner = idx < MB * 16 * 8;
    const int row = idx >> 3, u8 = idx & 7, ct = u8 >> 2, uu = u8 & 3;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
        const int rb = row >> 4, rr = row & 15, reg = rr & 3;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
            const int src_lane = (rr >> 2) * 16 + gate * 4 + uu;
            float sacc = 0.f;
#pragma unroll
            for (int w = 0; w < T2_SKINNY_WAVES; ++w) sacc += red[((w * 2 + ct) * MB + rb) * 64 + src_lane][reg];
            z[gate] = sacc;
        }
    }
    unsigned long long* my = slabs + ((size_t)(tile * 2 + half) * T2_SKINNY_THREADS + idx) * 2;
    unsigned long long* other = slabs + ((size_t)(tile * 2 + (half ^ 1)) * T2_SKINNY_THREADS + idx) * 2;
    if (owner) {
        __hip_atomic_store(my, ((unsigned long long)__float_as_uint(z[1]) << 32) | __float_as_uint(z[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(my + 1, ((unsigned long long)__float_as_uint(z[3]) << 32) | __float_as_uint(z[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket == 0u) return;                             // the other half finishes this tile
    if (threadIdx.x == 0) __hip_atomic_store(tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    float hn = 0.f;
    if (owner) {
        const unsigned long long o0 = __hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long o1 = __hip_atomic_load(other + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float oz[4] = {__uint_as_float((unsigned)o0), __uint_as_float((unsigned)(o0 >> 32)),
                             __uint_as_float((unsigned)o1), __uint_as_float((unsigned)(o1 >> 32))};
        float zt[4];
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) zt[gate] = half == 0 ? z[gate] + oz[gate] : oz[gate] + z[gate];   // always half0 + half1
        const int wg4 = 2 * tile + ct;
        const int unit = wg4 * 4 + uu;
        if (row < a.B && unit < e.H) {
            const float* bias = a.bias[0] + (size_t)wg4 * 16;
            const float zi = zt[0] + bias[0 * 4 + uu], zj = zt[1] + bias[1 * 4 + uu];
            const float zf = zt[2] + bias[2 * 4 + uu], zo = zt[3] + bias[3 * 4 + uu];
            float* cptr = e.c[0] + (size_t)row * e.Hld + unit;
            const float cp = *cptr;
            const float hp = frag_slot(e.hprev[0], t)[frag_idx(row, e.hprev[0].koff + unit, MB)];
            const float cn = sigmoidf_(zf + e.fb) * cp + sigmoidf_(zi) * tanhf(zj);
            hn = sigmoidf_(zo) * tanhf(cn);
            *cptr = (1.0f - e.zc) * cn + e.zc * cp;
            frag_slot(e.hstate[0], t)[frag_idx(row, e.hstate[0].koff + unit, MB)] = (1.0f - e.zh) * hn + e.zh * hp;
            if (e.hraw[0].base != nullptr) frag_slot(e.hraw[0], t)[frag_idx(row, e.hraw[0].koff + unit, MB)] = hn;
        }
        s_h[idx] = hn;
    }
    if (e.wq == nullptr) return;
    __syncthreads();
    const int U = e.U;
    const float* wq = e.wq + (size_t)tile * 8 * U;
    const int nvalid = min(8, e.H - tile * 8);
    float* qp = e.qpart + (size_t)tile * MB * 16 * U;
    for (int i = threadIdx.x; i < MB * 16 * U; i += T2_SKINNY_THREADS) {
        const int r2 = i / U, u = i - r2 * U;
        float sq = 0.f;
        for (int k = 0; k < nvalid; ++k) sq = fmaf(s_h[r2 * 8 + k], wq[k * U + u], sq);
        qp[i] = sq;
    }
}

// ------------------------------------------------------------------------------------------------
// Dense step: y = act(x . W + b) for 16 columns per workgroup.  Used for the PreNet layers
// (modules.py:88-91) and the mel / stop projections (rnn_wrappers.py:105-109) together with the
// stop rule of StopTokenBasedInferenceHelper (helpers.py:90-94) evaluated on the device.
// ------------------------------------------------------------------------------------------------
template <int MB>
__device__ __forceinline__ void dense_body(const SkinnyCore& a, const DenseEpi& e, int wg, f32x4* red, int* s_notstop) {
    f32x4 acc[MB];
    const int t = skinny_core<MB, 0>(a, 0, wg, acc, [](int) {});
    if (t < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    if (threadIdx.x == 0) *s_notstop = 0;
    __syncthreads();
    const bool has_stop = e.stop_col >= 0 && (e.stop_col >> 4) == wg;
    float* ofrag = e.out_frag.base ? frag_slot(e.out_frag, t) : nullptr;
    for (int idx = threadIdx.x; idx < MB * 16 * 16; idx += T2_SKINNY_THREADS) {
        const int row = idx >> 4, c = idx & 15;
        const int col = wg * 16 + c;
        if (row >= a.B) continue;
        const bool fused_col = e.fuse_n > 0 && col >= e.fuse_col0 && col < e.fuse_col0 + e.fuse_n;
        if (col >= e.N && !fused_col) continue;
        float v = red_sum<MB>(red, row, c);
        if (a.bias[0] != nullptr) v += a.bias[0][col];
        if (fused_col) {
            frag_slot(e.fuse_frag, t)[frag_idx(row, e.fuse_frag.koff + col - e.fuse_col0, MB)] = fmaxf(v, 0.f);
            continue;
        }
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.drop_rate > 0.f) v = dropout_train(v, e.drop_rate, t2_u01(e.seed, e.stream, t, row, col));
        if (col == e.stop_col) {
            e.out_stop[(size_t)row * e.stop_ld + (size_t)t * e.stop_t_mul] = v;
            if (!(v > 0.f)) atomicOr(s_notstop, 1);       // sigmoid(v) > 0.5  <=>  v > 0
            continue;
        }
        if (e.out_plain != nullptr && col < e.plain_ncol)
            e.out_plain[(size_t)row * e.plain_ld + (size_t)t * e.plain_t_stride + col] = v;
        if (ofrag != nullptr && col >= e.frag_col0) ofrag[frag_idx(row, e.out_frag.koff + col - e.frag_col0, MB)] = v;
    }
    if (has_stop) {
        __syncthreads();
        // finished = reduce_all(stop & (time > min_iters)); outputs of the finishing step are kept,
        // so the decode emits t + 1 steps.  Readers at step t see 0 or t + 1: both let step t run.
        if (threadIdx.x == 0 && e.ctl_w != nullptr && e.ctl_w[CTL_STOP_EN] != 0 && *s_notstop == 0 &&
            t > e.ctl_w[CTL_MINIT] && e.ctl_w[CTL_FINAL] == 0)
            e.ctl_w[CTL_FINAL] = t + 1;
    }
}

// Grid = max(e.n_wg, pj.n[0]).  Workgroup i first runs the kernel's own column tile i (the critical
// path), then tile i of the riding job: the skinny kernels hold one workgroup per CU (register file), so
// extra workgroups beyond 256 would run as a second round behind the job's.
template <int MB, int NPW>      // NPW: static ring of the riding partial job (0: runtime loop)
__global__ __launch_bounds__(T2_SKINNY_THREADS) void dense_step_kernel(SkinnyCore a, DenseEpi e, PartialJobs pj) {
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    __shared__ int s_notstop;
    if ((int)blockIdx.x < e.n_wg) dense_body<MB>(a, e, blockIdx.x, red, &s_notstop);
    const int j = (int)blockIdx.x - pj.skip;
    if (j < 0 || j >= pj.n[0] + pj.n[1] + pj.n[2]) return;
    if ((int)blockIdx.x < e.n_wg) __syncthreads();
    if (j >= pj.n[0] + pj.n[1]) partial_body<MB, NPW>(pj.core[2], pj.z[2], pj.off[2] + j - pj.n[0] - pj.n[1], red);
    else if (j >= pj.n[0]) partial_body<MB, NPW>(pj.core[1], pj.z[1], pj.off[1] + j - pj.n[0], red);
    else partial_body<MB, NPW>(pj.core[0], pj.z[0], pj.off[0] + j, red);
}

// ------------------------------------------------------------------------------------------------
// Location-sensitive attention step for one batch row per workgroup
// (reference tacotron/tacotron_v2.py:118-141, _location_sensitive_score :66-78; BahdanauAttention
// masking / softmax and AttentionWrapper context: SURVEY.md App. A5/A6).
// Phases: (0) previous (cumulative) alignments, conv kernel and the processed query (sum of the
// attention LSTM workgroups' partial products) into LDS; (1) location conv with a register sliding
// window; (2) energies: one wave per position, lanes over attention units, location-layer columns
// held in registers, wavefront-reduction over units; (3) masked softmax; (4) context.
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* keys;      // [B][T][U]
    const float* values;    // [B][T][E] (length-masked memory)
    const int* lens;        // [B] (device int32)
    float* att_state;       // [B][T] in/out: cumulative alignments (or previous alignments)
    float* alignments;      // [B][T] out
    float* history;         // [cap][B][T] or null
    const float* qpart;     // [nparts][MB*16][U] partial processed queries
    int nparts;
    FragBuf ctx_dst[2];     // frag16 destinations of the context (base may be null)
    float* ctx_plain;       // [B][E] out (state readout), may be null
    const float* conv_w;    // [K][F]
    const float* conv_b;    // [F]
    const float* wloc;      // [F][U]
    const float* v_a;       // [U]
    const float* b_a;       // [U]
    int B, T, U, E, F, K, MB, cumulative;
    const int* ctl;
    int step_off;
    // tape of the backward pass (may be null): processed query + bias [S][B][U], attention state before the step [S][B][T],
    // context [S][B][E]
    float* tp_q; float* tp_state_prev; float* tp_ctx;
};

#define T2_ATTN_THREADS 512

// kernel variants: <FMAX, UPL> = <32, 2> (F <= 32, U <= 128: Tacotron2), <32, 4> (F <= 32, U <= 256: the V1 decoder's
// plain Bahdanau score runs as F = 1 with a zero location layer), <64, 4> (F <= 64, U <= 256)
__host__ __device__ inline int attn_upl(int U, int F) { return (F <= 32 && U <= 128) ? 2 : 4; }
__host__ __device__ inline int attn_fmax(int F) { return F <= 32 ? 32 : 64; }

// Positions whose location features are held in LDS at a time: inputs longer than this are swept in chunks (features of a
// chunk, then its energies), so that only 8 bytes per encoder position scale with T_in (previous alignments, energies) and
// the 120 KB budget reaches T_in of several thousand instead of ~600.
#define T2_ATTN_CHUNK 256
__host__ __device__ inline int attn_lds_floats(int T, int U, int F, int K) {
    const int FP = (F + 3) & ~3, KP = (K + 3) & ~3, UP = (U + 3) & ~3;
    const int FMAX = attn_fmax(F);
    const int TC = T < T2_ATTN_CHUNK ? T : T2_ATTN_CHUNK;
    return ((T + KP + 16 + 3) & ~3) + KP * FP + ((TC + 8) * FMAX) + ((T + 3) & ~3) + 17 * UP + 64 + 2048 + FMAX * 64 * attn_upl(U, F);
}

// tanh via v_exp_f32 / v_rcp_f32: abs error < 1e-6 (the energies feed a softmax; tolerance budget 1e-3)
__device__ __forceinline__ float fast_tanhf(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// 1 / (1 + e^-x) on the hardware exp2 / rcp (about 1e-6 relative; e^-x = inf for x < -88 gives exactly 0)
__device__ __forceinline__ float fast_sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

#ifdef T2_ATTN_STAMPS   // diagnostic builds (tools/attn_bench.hip) only
__device__ unsigned long long* g_attn_stamps;
#define T2_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_attn_stamps[i] = clock64(); } while (0)
#else
#define T2_STAMP(i) do { } while (0)
#endif

template <int N> struct VecOf;
template <> struct VecOf<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef float type __attribute__((ext_vector_type(4))); };

// The kernel is a chain of short phases on ONE workgroup per batch row, so its time is load latency
// and instruction issue on one CU.  Every global load that does not depend on this step's LDS work
// (query partials, keys, later the values) is issued early, in the order it is needed, with wide
// loads; the location-layer matrix goes through LDS once instead of once per wave; the per-position
// sums over attention units are reduced eight positions at a time (transposed butterfly).
// Lane l owns attention units UPL*l .. UPL*l + UPL-1.
template <int FMAX, int UPL, int MB, int NPW>     // MB / NPW: geometry of the riding partial jobs
__global__ __launch_bounds__(T2_ATTN_THREADS) void attention_step_kernel(AttnArgs a, PartialJobs pj) {
    typedef typename VecOf<UPL>::type vecu;
    static_assert(T2_ATTN_THREADS == T2_SKINNY_THREADS, "the riding jobs use the skinny GEMM geometry");
    {
        __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
        const int j = (int)blockIdx.x - a.B;           // the attention rows come first: they are the critical path
        if (j >= pj.n[0] + pj.n[1]) { partial_body<MB, NPW>(pj.core[2], pj.z[2], pj.off[2] + j - pj.n[0] - pj.n[1], red); return; }
        if (j >= pj.n[0]) { partial_body<MB, NPW>(pj.core[1], pj.z[1], pj.off[1] + j - pj.n[0], red); return; }
        if (j >= 0) { partial_body<MB, NPW>(pj.core[0], pj.z[0], pj.off[0] + j, red); return; }
    }
    T2_STAMP(0);
    const int t_step = step_index(a.ctl, a.step_off);
    if (t_step < 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int T = a.T, U = a.U, F = a.F, K = a.K, E = a.E;
    const int FP = (F + 3) & ~3, KP = (K + 3) & ~3;
    const int pad_l = (K - 1) / 2;
    const int UP = (U + 3) & ~3;
    constexpr int UW = 64 * UPL;                        // padded unit count of the register layout
    float* s_prev = smem;                               // [T + KP + 16]: index i <-> position i - pad_l
    float* s_cw = s_prev + ((T + KP + 16 + 3) & ~3);    // [KP][FP] conv kernel, zero padded
    float* s_f = s_cw + KP * FP;                        // [min(T, T2_ATTN_CHUNK) + 8][FMAX] of the current chunk, columns >= F are zero
    float* s_e = s_f + (min(T, T2_ATTN_CHUNK) + 8) * FMAX;   // [T]
    float* s_q = s_e + ((T + 3) & ~3);                  // [16][UP] partial sums, then [16*UP ..] = q + b_a
    float* s_red = s_q + 17 * UP;                       // [64]
    float* s_ctx = s_red + 64;                          // [<= 2048] context partials
    float* s_wl = s_ctx + 2048;                         // [FMAX][UW] location layer, zero padded
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = T2_ATTN_THREADS / 64;
    const int len = a.lens[b];
    float* state = a.att_state + (size_t)b * T;

    // ---- early loads, in the order they are consumed ----
    for (int i = tid; i < T + KP + 16; i += T2_ATTN_THREADS) {
        const int p = i - pad_l;
        s_prev[i] = (p >= 0 && p < T) ? state[p] : 0.f;
    }
    for (int i = tid; i < KP * FP; i += T2_ATTN_THREADS) {
        const int j = i / FP, c = i - j * FP;
        s_cw[i] = (j < K && c < F) ? a.conv_w[j * F + c] : 0.f;
    }
    // location layer [F][U]: loaded to registers now (small, issued before the big loads), staged to LDS
    // [FMAX][UW] (zero padded) after the conv phase, once per workgroup instead of once per wave
    constexpr int WLN = FMAX * UW / 4 / T2_ATTN_THREADS;
    f32x4 wlreg[WLN];
    {
        const bool w_vec = (U & 3) == 0;
#pragma unroll
        for (int k = 0; k < WLN; ++k) {
            const int i4 = tid + k * T2_ATTN_THREADS;
            const int c = i4 / (UW / 4), u4 = (i4 - c * (UW / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < F) {
                if (w_vec && u4 + 3 < U) v = *reinterpret_cast<const f32x4*>(a.wloc + (size_t)c * U + u4);
                else {
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) if (u4 + e4 < U) v[e4] = a.wloc[(size_t)c * U + u4 + e4];
                }
            }
            wlreg[k] = v;
        }
    }
    // query partials: thread -> (unit quad, part group), float4 loads, up to 16 in flight per thread
    constexpr int QN = 16;
    const bool q_vec = (U & 3) == 0;
    const int QV = q_vec ? (U >> 2) : U;                // columns of the partial matrix in load units
    const int PG = min(16, T2_ATTN_THREADS / QV);       // part groups
    const bool q_active = tid < PG * QV;
    const int qg = q_active ? tid / QV : 0, qc = q_active ? tid - qg * QV : 0;
    const size_t pstride = (size_t)a.MB * 16 * U;
    const float* qbase = a.qpart + (size_t)b * U;
    f32x4 qv[QN];
    if (q_vec) {
#pragma unroll
        for (int i = 0; i < QN; ++i)
            qv[i] = *reinterpret_cast<const f32x4*>(qbase + (size_t)min(qg + i * PG, a.nparts - 1) * pstride + 4 * qc);
    } else {
#pragma unroll
        for (int i = 0; i < QN; ++i) {
            const float v = qbase[(size_t)min(qg + i * PG, a.nparts - 1) * pstride + qc];
            qv[i] = f32x4{v, 0.f, 0.f, 0.f};
        }
    }
    // keys of this wave's positions tb, tb + NW, ... (8 per batch)
    const float* keys = a.keys + (size_t)b * T * U;
    const bool k_vec = (U % UPL) == 0;
    vecu kvA[8], kvB[8];
    auto load_keys = [&](vecu (&kv)[8], int tb) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int tt = min(tb + r * NW, T - 1);
            if (k_vec) {
                kv[r] = *reinterpret_cast<const vecu*>(keys + (size_t)tt * U + min(UPL * lane, U - UPL));
            } else {
#pragma unroll
                for (int i = 0; i < UPL; ++i) kv[r][i] = keys[(size_t)tt * U + min(UPL * lane + i, U - 1)];
            }
        }
    };
    load_keys(kvA, wave);
    load_keys(kvB, wave + 8 * NW);
    float va[UPL], kq[UPL];
#pragma unroll
    for (int i = 0; i < UPL; ++i) { const int u = UPL * lane + i; va[i] = (u < U) ? a.v_a[u] : 0.f; }   // 0 masks u >= U
    T2_STAMP(1);
    __syncthreads();
    T2_STAMP(2);
    // ---- phase 1: location features f[t][c] = sum_j w[j][c] * prev[t + j - pad_l] + b[c]   (:128-130) of positions
    // c0 .. c0 + cn - 1 (row t - c0 of s_f)
    auto features = [&](int c0, int cn) {
        const int nblk = (cn + 7) >> 3;
        const float* s_prev_c = s_prev + c0;
        for (int item = tid; item < nblk * FMAX; item += T2_ATTN_THREADS) {
            const int c = item % FMAX, t0 = (item / FMAX) << 3;
            if (c >= F) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s_f[(t0 + i) * FMAX + c] = 0.f;
                continue;
            }
            float accv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) accv[i] = 0.f;
            float win[12];
            {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_prev_c + t0);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_prev_c + t0 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { win[i] = w0[i]; win[4 + i] = w1[i]; }
            }
            for (int j0 = 0; j0 < KP; j0 += 4) {
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(s_prev_c + t0 + j0 + 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) win[8 + i] = w2[i];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float w = s_cw[(j0 + jj) * FP + c];
#pragma unroll
                    for (int i = 0; i < 8; ++i) accv[i] = fmaf(w, win[i + jj], accv[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) win[i] = win[i + 4];
            }
            const float bc = a.conv_b[c];
#pragma unroll
            for (int i = 0; i < 8; ++i) s_f[(t0 + i) * FMAX + c] = accv[i] + bc;
        }
    };
    const int cn0 = min(T, T2_ATTN_CHUNK);
    features(0, cn0);
#pragma unroll
    for (int k = 0; k < WLN; ++k) *reinterpret_cast<f32x4*>(s_wl + 4 * (tid + k * T2_ATTN_THREADS)) = wlreg[k];
    T2_STAMP(3);
    // processed query (tacotron_v2.py:122): sum over the attention-LSTM workgroups' partial products
    {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < QN; ++i) s0 += (qg + i * PG < a.nparts) ? qv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = qg + QN * PG; p < a.nparts; p += PG) {          // only if nparts > 16 * PG
            if (q_vec) s0 += *reinterpret_cast<const f32x4*>(qbase + (size_t)p * pstride + 4 * qc);
            else s0[0] += qbase[(size_t)p * pstride + qc];
        }
        if (q_active) {
            if (q_vec) *reinterpret_cast<f32x4*>(s_q + qg * UP + 4 * qc) = s0;
            else s_q[qg * UP + qc] = s0[0];
        }
    }
    __syncthreads();
    if (tid < U) {
        float sq = a.b_a[tid];
        for (int g = 0; g < PG; ++g) sq += s_q[g * UP + tid];
        s_q[16 * UP + tid] = sq;          // q[u] + b_a[u]
        if (a.tp_q != nullptr) a.tp_q[((size_t)t_step * a.B + b) * U + tid] = sq;
    }
    __syncthreads();
    // location-layer columns of this lane's units: LDS -> registers
    float wl[FMAX][UPL];
#pragma unroll
    for (int c = 0; c < FMAX; ++c) {
        const vecu w = *reinterpret_cast<const vecu*>(s_wl + c * UW + UPL * lane);
#pragma unroll
        for (int i = 0; i < UPL; ++i) wl[c][i] = w[i];
    }
    T2_STAMP(4);
    // ---- phase 2: energies   (:131-133, :66-78); one wave per position, 8 positions per batch (64 positions per sweep of
    // the workgroup; T2_ATTN_CHUNK is a multiple of it, so a batch never straddles two feature chunks)
    {
        static_assert(T2_ATTN_CHUNK % (8 * (T2_ATTN_THREADS / 64)) == 0 && T2_ATTN_CHUNK >= 16 * (T2_ATTN_THREADS / 64), "chunk = whole position batches");
#pragma unroll
        for (int i = 0; i < UPL; ++i) { const int u = UPL * lane + i; kq[i] = (u < U) ? s_q[16 * UP + u] : 0.f; }
        int c0 = 0;                                            // first position of the feature chunk in s_f
        auto energies = [&](const vecu (&kv)[8], int tb) {
            float part[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int tt = min(tb + r * NW, T - 1);           // clamped: branch-free, invalid ones are not stored
                float pf[UPL];
#pragma unroll
                for (int i = 0; i < UPL; ++i) pf[i] = 0.f;
                const float* ft = s_f + (tt - c0) * FMAX;
#pragma unroll
                for (int c4 = 0; c4 < FMAX; c4 += 4) {
                    const f32x4 fv = *reinterpret_cast<const f32x4*>(ft + c4);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                        for (int i = 0; i < UPL; ++i) pf[i] = fmaf(fv[cc], wl[c4 + cc][i], pf[i]);
                }
                float p = 0.f;
#pragma unroll
                for (int i = 0; i < UPL; ++i) p += va[i] * fast_tanhf(kv[r][i] + kq[i] + pf[i]);
                part[r] = p;
            }
            // transposed butterfly: 8 sums over 64 lanes with 10 shuffles; position r ends on lanes with
            // (lane>>5, (lane>>4)&1, (lane>>3)&1) = bits of r, lane & 7 == 0
            float p4[4], p2[2], p1;
            const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float send = h5 ? part[k] : part[k + 4];
                const float keep = h5 ? part[k + 4] : part[k];
                p4[k] = keep + __shfl_xor(send, 32);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float send = h4 ? p4[k] : p4[k + 2];
                const float keep = h4 ? p4[k + 2] : p4[k];
                p2[k] = keep + __shfl_xor(send, 16);
            }
            {
                const float send = h3 ? p2[0] : p2[1];
                const float keep = h3 ? p2[1] : p2[0];
                p1 = keep + __shfl_xor(send, 8);
            }
            p1 += __shfl_xor(p1, 4);
            p1 += __shfl_xor(p1, 2);
            p1 += __shfl_xor(p1, 1);
            if ((lane & 7) == 0) {
                const int r = (h5 ? 4 : 0) + (h4 ? 2 : 0) + (h3 ? 1 : 0);
                const int tt = tb + r * NW;
                if (tt < T) s_e[tt] = p1;
            }
        };
        energies(kvA, wave);
        if (wave + 8 * NW < T) energies(kvB, wave + 8 * NW);
        for (int tb = wave + 16 * NW; tb < cn0; tb += 8 * NW) {    // only if T > 128
            load_keys(kvA, tb);
            energies(kvA, tb);
        }
        for (c0 = T2_ATTN_CHUNK; c0 < T; c0 += T2_ATTN_CHUNK) {    // long inputs: the next chunks' features, then their energies
            const int cn = min(T - c0, T2_ATTN_CHUNK);
            load_keys(kvA, c0 + wave);
            __syncthreads();                                       // every wave is past its reads of the previous chunk's features
            features(c0, cn);
            __syncthreads();
            for (int tb = c0 + wave; tb < c0 + cn; tb += 8 * NW) {
                if (tb != c0 + wave) load_keys(kvA, tb);
                energies(kvA, tb);
            }
        }
    }
    T2_STAMP(5);
    // ---- values for the context: issued before the softmax, consumed after it
    const float* vals = a.values + (size_t)b * T * E;
    const int EC = E >> 2;
    const bool vec_ctx = (E & 3) == 0 && EC <= T2_ATTN_THREADS;
    constexpr int VN = 32;
    const int NG = vec_ctx ? T2_ATTN_THREADS / EC : 1;
    const int vg = vec_ctx ? tid / EC : 0, vcq = vec_ctx ? tid - vg * EC : 0;
    const bool v_active = vec_ctx && vg < NG;
    f32x4 vv[VN];
    if (v_active) {
        const f32x4* vp = reinterpret_cast<const f32x4*>(vals) + vcq;
#pragma unroll
        for (int r = 0; r < VN; ++r) vv[r] = vp[(size_t)min(vg + r * NG, len - 1) * EC];
    }
    __syncthreads();
    T2_STAMP(6);
    // ---- phase 3: masked softmax over positions < len (score mask value -inf -> probability exactly 0)
    float m = -INFINITY;
    for (int i = tid; i < len; i += T2_ATTN_THREADS) m = fmaxf(m, s_e[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, s_red[w]);
    float sum = 0.f;
    for (int i = tid; i < T; i += T2_ATTN_THREADS) {
        const float ex = (i < len) ? expf(s_e[i] - m) : 0.f;
        s_e[i] = ex;
        sum += ex;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) s_red[32 + wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < NW; ++w) sum += s_red[32 + w];
    float* al = a.alignments + (size_t)b * T;
    float* hist = a.history ? a.history + ((size_t)t_step * a.B + b) * T : nullptr;
    for (int i = tid; i < T; i += T2_ATTN_THREADS) {
        const float p = s_e[i] / sum;
        s_e[i] = p;
        al[i] = p;
        if (hist) hist[i] = p;
        if (a.tp_state_prev != nullptr) a.tp_state_prev[((size_t)t_step * a.B + b) * T + i] = s_prev[i + pad_l];
        state[i] = a.cumulative ? (p + s_prev[i + pad_l]) : p;      // :137-140
    }
    __syncthreads();
    T2_STAMP(7);
    // ---- phase 4: context = alignments . values   (AttentionWrapper._compute_attention)
    float* d0 = a.ctx_dst[0].base ? frag_slot(a.ctx_dst[0], t_step) : nullptr;
    float* d1 = a.ctx_dst[1].base ? frag_slot(a.ctx_dst[1], t_step) : nullptr;
    auto emit = [&](int c, float v) {
        if (d0) d0[frag_idx(b, a.ctx_dst[0].koff + c, a.MB)] = v;
        if (d1) d1[frag_idx(b, a.ctx_dst[1].koff + c, a.MB)] = v;
        if (a.ctx_plain) a.ctx_plain[(size_t)b * E + c] = v;
        if (a.tp_ctx) a.tp_ctx[((size_t)t_step * a.B + b) * E + c] = v;
    };
    if (vec_ctx) {
        if (v_active) {
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < VN; ++r) {
                const int i = vg + r * NG;
                const float p = (i < len) ? s_e[i] : 0.f;
                acc4 += p * vv[r];
            }
            const f32x4* vp = reinterpret_cast<const f32x4*>(vals) + vcq;
            for (int i = vg + VN * NG; i < len; i += NG) acc4 += s_e[i] * vp[(size_t)i * EC];    // only if len > VN*NG
            *reinterpret_cast<f32x4*>(s_ctx + (size_t)vg * E + 4 * vcq) = acc4;
        }
        __syncthreads();
        for (int c = tid; c < E; c += T2_ATTN_THREADS) {
            float sc = 0.f;
            for (int gg = 0; gg < NG; ++gg) sc += s_ctx[gg * E + c];
            emit(c, sc);
        }
    } else {
        for (int c = tid; c < E; c += T2_ATTN_THREADS) {
            float s0 = 0.f, s1 = 0.f;
            int i = 0;
            for (; i + 1 < len; i += 2) {
                s0 = fmaf(s_e[i], vals[(size_t)i * E + c], s0);
                s1 = fmaf(s_e[i + 1], vals[(size_t)(i + 1) * E + c], s1);
            }
            if (i < len) s0 = fmaf(s_e[i], vals[(size_t)i * E + c], s0);
            emit(c, s0 + s1);
        }
    }
    T2_STAMP(8);
}

// ------------------------------------------------------------------------------------------------
// small data movement kernels
// ------------------------------------------------------------------------------------------------
// plain [B][ld] (+ offset) -> frag16; used to feed decoder inputs (helpers.py:86,182,187-188).
struct FeedArgs {
    const float* src;         // plain base
    long long row_stride;     // floats between batch rows
    long long t_stride;       // floats between steps: source of step t is src + (t-1) * t_stride
    FragBuf dst;              // frag16 destination
    int B, K, MB;
    int go_zero;              // 1: step 0 writes zeros (go frame), steps t >= 1 read step t-1's slice
    const int* ctl;
    int step_off;
};

__global__ void feed_frames_kernel(FeedArgs a) {
    const int t = step_index(a.ctl, a.step_off);
    if (t < 0) return;
    const int KP = (a.K + 15) & ~15;
    const int n = a.MB * 16 * KP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int row = i / KP, k = i - row * KP;
        float v = 0.f;
        if (row < a.B && k < a.K) {
            if (a.go_zero) {
                if (t > 0) v = a.src[(size_t)row * a.row_stride + (size_t)(t - 1) * a.t_stride + k];
            } else {
                v = a.src[(size_t)row * a.row_stride + (size_t)t * a.t_stride + k];
            }
        }
        frag_slot(a.dst, t)[frag_idx(row, a.dst.koff + k, a.MB)] = v;
    }
}

__global__ void advance_ctl_kernel(int* ctl, int n) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctl[CTL_BASE] += n;
}

// Embedding lookup (modules.py:67): out[m][:] = table[ids[m]][:]
__global__ void embedding_kernel(const int* __restrict__ ids, const float* __restrict__ table, float* __restrict__ out,
                                 int n_rows, int dim) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)n_rows * dim;
         i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / dim), c = (int)(i - (long long)m * dim);
        out[i] = table[(size_t)ids[m] * dim + c];
    }
}

// values = memory * sequence_mask (BahdanauAttention._prepare_memory)
__global__ void mask_memory_kernel(const float* __restrict__ mem, const int* __restrict__ lens, float* __restrict__ out,
                                   int B, int T, int E) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)B * T * E;
         i += (long long)gridDim.x * blockDim.x) {
        const long long bt = i / E;
        const int b = (int)(bt / T), tt = (int)(bt - (long long)b * T);
        out[i] = (tt < lens[b]) ? mem[i] : 0.f;
    }
}

// conv stack output [B][T][C] -> per-step frag16 inputs of the BiLSTM:
// dst[dir][s] (frag16 [C]) = x[b][s] (fw) or x[b][len_b-1-s] (bw, tf.reverse_sequence), 0 beyond len.
__global__ void seq_to_frag_kernel(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ dst,
                                   int B, int T, int C, int MB, int KP) {
    const long long per_step = (long long)MB * 16 * KP;
    const long long n = 2LL * T * per_step;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int dir = (int)(i / (T * per_step));
        long long r = i - (long long)dir * T * per_step;
        const int s = (int)(r / per_step);
        r -= (long long)s * per_step;
        const int row = (int)(r / KP), k = (int)(r - (long long)row * KP);
        float v = 0.f;
        if (row < B && k < C) {
            const int len = lens[row];
            const int tt = dir == 0 ? s : (len - 1 - s);
            if (tt >= 0 && tt < T && (dir == 0 || s < len)) v = x[((size_t)row * T + tt) * C + k];
        }
        dst[((size_t)dir * T + s) * per_step + frag_idx(row, k, MB)] = v;
    }
}

// frag16 -> plain [B][K] (state readout for tests / RNNCell-protocol callers)
__global__ void frag_to_plain_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int K, int MB, int koff) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * K; i += gridDim.x * blockDim.x) {
        const int row = i / K, k = i - row * K;
        dst[i] = src[frag_idx(row, koff + k, MB)];
    }
}

// ------------------------------------------------------------------------------------------------
// Tiled fp32 MFMA implicit-GEMM conv1d (SAME, stride 1) with fused per-channel affine (inference
// batch norm or bias), activation and optional residual:
//   Y[m][n] = act( (sum_{j,i} X[m + j - pad_l][i] * W[j][i][n]) * scale[n] + shift[n] ) (+ R[m][n])
// m = b*T + t; taps reaching outside [0,T) of the same utterance read zero.  k = 1 gives a plain
// Dense.  Used for Conv1d (modules.py:149-157), PostNetV2's projection + residual
// (tacotron_v2.py:202-203) and the attention memory layer.
// 128x128 tile, BK = 16, 256 threads = 4 waves, each wave 2x2 tiles of v_mfma_f32_32x32x2_f32.
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;  int ldx;      // input rows [Mtot][ldx], Cin valid channels
    const float* w;                // [ktaps*Cin][N]
    const float* scale;            // [N] or null (1)
    const float* shift;            // [N] or null (0)
    const float* res; int ldr;     // residual [Mtot][ldr] or null
    float* y; int ldy;
    int Mtot, T, Cin, N, ktaps, pad_l;
    int act;                       // 0 none, 1 relu, 2 tanh
    int vec_a, vec_b;              // float4 paths usable
    int trans_a;                   // 1: weight gradient dW = im2col(X)^T . dY: output row mo = (tap j, channel i) of ktaps*Cin rows,
                                   //    reduction over the `red` rows rr of X / dY (w = dY [red][N]); element = X[rr + j - pad_l][i] within the utterance
    int red;
    int splits;                    // > 1: split-K over gridDim.z; raw partial sums go to part[z][Mtot][N] (no epilogue), summed in
    float* part;                   //      split order by a second kernel (deterministic)
    const float* wt;               // forward fast path: the kernel transposed, [N][ktaps*Cin] (null: not available)
};

#define CG_BM 128
#define CG_BN 128
#define CG_BK 16
#define CG_LD (CG_BM + 4)

__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
    __shared__ float As[2][CG_BK][CG_LD];
    __shared__ float Bs[2][CG_BK][CG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * CG_BM, n0 = blockIdx.x * CG_BN;
    const int Ktot = a.trans_a ? a.red : a.ktaps * a.Cin;
    const int nk = (Ktot + CG_BK - 1) / CG_BK;
    const int nz = a.splits > 1 ? a.splits : 1, z = a.splits > 1 ? (int)blockIdx.z : 0;
    const int kt0 = (int)((long long)nk * z / nz), kt1 = (int)((long long)nk * (z + 1) / nz);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A tile loader: thread -> rows (tid >> 2) and (tid >> 2) + 64, k-quad (tid & 3)
    const int a_kq = tid & 3;
    const int a_r0 = tid >> 2;
    // B tile loader: thread -> k rows (tid >> 5) and +8, n-quad (tid & 31)
    const int b_nq = tid & 31;
    const int b_k0 = tid >> 5;

    float4 ra[2], rbv[2];

    auto load_tiles = [&](int kt) {
        const int kbase = kt * CG_BK;
        if (a.trans_a) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int rr = kbase + b_k0 + p * 8;                 // reduction row
                const int mo = m0 + b_nq * 4;                        // output rows mo .. mo + 3
                float tmp[4] = {0.f, 0.f, 0.f, 0.f};
                if (rr < Ktot && a.vec_a) {                         // Cin % 4 == 0: the four rows share the tap
                    if (mo < a.Mtot) {
                        const int j = mo / a.Cin, i = mo - j * a.Cin;
                        const int ts = rr % a.T + j - a.pad_l;
                        if (ts >= 0 && ts < a.T) {
                            const float4 v = *reinterpret_cast<const float4*>(a.x + (size_t)(rr + j - a.pad_l) * a.ldx + i);
                            tmp[0] = v.x; tmp[1] = v.y; tmp[2] = v.z; tmp[3] = v.w;
                        }
                    }
                } else if (rr < Ktot) {
                    const int tt = rr % a.T;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (mo + c < a.Mtot) {
                            const int j = (mo + c) / a.Cin, i = (mo + c) - j * a.Cin;
                            const int ts = tt + j - a.pad_l;
                            if (ts >= 0 && ts < a.T) tmp[c] = a.x[(size_t)(rr + j - a.pad_l) * a.ldx + i];
                        }
                    }
                }
                ra[p] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
            }
        } else
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int m = m0 + a_r0 + p * 64;
            const int kk = kbase + a_kq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < a.Mtot) {
                const int tt = m % a.T;
                if (a.vec_a) {
                    if (kk < Ktot) {
                        const int j = kk / a.Cin, i = kk - j * a.Cin;
                        const int ts = tt + j - a.pad_l;
                        if (ts >= 0 && ts < a.T)
                            v = *reinterpret_cast<const float4*>(a.x + (size_t)(m + j - a.pad_l) * a.ldx + i);
                    }
                } else {
                    float tmp[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k2 = kk + c;
                        float s = 0.f;
                        if (k2 < Ktot) {
                            const int j = k2 / a.Cin, i = k2 - j * a.Cin;
                            const int ts = tt + j - a.pad_l;
                            if (ts >= 0 && ts < a.T) s = a.x[(size_t)(m + j - a.pad_l) * a.ldx + i];
                        }
                        tmp[c] = s;
                    }
                    v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int kk = kbase + b_k0 + p * 8;
            const int n = n0 + b_nq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kk < Ktot) {
                if (a.vec_b) {
                    if (n < a.N) v = *reinterpret_cast<const float4*>(a.w + (size_t)kk * a.N + n);
                } else {
                    float tmp[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) tmp[c] = (n + c < a.N) ? a.w[(size_t)kk * a.N + n + c] : 0.f;
                    v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
            }
            rbv[p] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        if (a.trans_a) {
#pragma unroll
            for (int p = 0; p < 2; ++p) *reinterpret_cast<float4*>(&As[buf][b_k0 + p * 8][b_nq * 4]) = ra[p];
        } else
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = a_r0 + p * 64;
            As[buf][a_kq * 4 + 0][r] = ra[p].x;
            As[buf][a_kq * 4 + 1][r] = ra[p].y;
            As[buf][a_kq * 4 + 2][r] = ra[p].z;
            As[buf][a_kq * 4 + 3][r] = ra[p].w;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
            *reinterpret_cast<float4*>(&Bs[buf][b_k0 + p * 8][b_nq * 4]) = rbv[p];
    };

    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles(0);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < CG_BK; ks += 2) {
            const float a0 = As[buf][ks + lh][wm * 64 + l31];
            const float a1 = As[buf][ks + lh][wm * 64 + 32 + l31];
            const float b0 = Bs[buf][ks + lh][wn * 64 + l31];
            const float b1 = Bs[buf][ks + lh][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tiles(buf ^ 1);
        __syncthreads();
    }
    if (a.splits > 1) {
        float* part = a.part + (size_t)z * a.Mtot * a.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (n >= a.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < a.Mtot) part[(size_t)m * a.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    // epilogue: D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= a.N) continue;
            const float sc = a.scale ? a.scale[n] : 1.f;
            const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.Mtot) continue;
                float v = acc[i][j][r] * sc + sh;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = tanhf(v);
                if (a.res) v += a.res[(size_t)m * a.ldr + n];
                a.y[(size_t)m * a.ldy + n] = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Forward fast path of conv_gemm_kernel: float4 loads on both operands, Cin a multiple of the k-tile
// (so a k-tile lies inside ONE tap: the tap index is uniform and advances with the loop, no integer
// division or divergent control flow in the main loop), no split-K, no transposed operand.  Same
// tiles, same MFMA loop, same summation order and epilogue as the generic kernel (bit-identical results).
// ------------------------------------------------------------------------------------------------
// SPLIT: the k-tiles are dealt to gridDim.z slices (a.splits); a slice writes its raw partial sums to a.part[z][Mtot][N] and
// conv_split_finish_kernel adds them in slice order and applies the epilogue (few output tiles against a long reduction, e.g.
// the CBHG bank's data gradient: 70 tiles x 2048 k-tiles).
template <int MI, bool SPLIT = false>      // M sub-tiles of 32 rows per wave: workgroup tile 64 * MI rows x 128 columns (MI = 1 for small M: more workgroups)
__global__ __launch_bounds__(256) void conv_gemm_fast_kernel(ConvArgs a) {
    constexpr int BM = 64 * MI;
    __shared__ float As[2][CG_BK][BM + 4];
    __shared__ float Bs[2][CG_BK][CG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * CG_BN;
    const int nk_all = (a.ktaps * a.Cin) / CG_BK;
    const int kper = SPLIT ? (nk_all + a.splits - 1) / a.splits : nk_all;
    const int kt0 = SPLIT ? (int)blockIdx.z * kper : 0;
    const int nk = SPLIT ? min(nk_all, kt0 + kper) : nk_all;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_kq = tid & 3, a_r0 = tid >> 2;
    const int b_nq = tid & 31, b_k0 = tid >> 5;
    // per-thread row state of the A loader: position inside the utterance (or far outside for rows >= Mtot)
    int tt[MI];
    const float* xrow[MI];
#pragma unroll
    for (int p = 0; p < MI; ++p) {
        const int m = m0 + a_r0 + p * 64;
        tt[p] = m < a.Mtot ? m % a.T : -(1 << 28);
        xrow[p] = a.x + (size_t)min(m, a.Mtot - 1) * a.ldx + a_kq * 4;
    }
    const int n_b = n0 + b_nq * 4;
    const bool b_ok = n_b < a.N;
    const float* wcol = a.w + (size_t)b_k0 * a.N + (b_ok ? n_b : 0);

    float4 ra[MI], rbv[2];
    int tap = (kt0 * CG_BK) / a.Cin, ci = (kt0 * CG_BK) % a.Cin;      // k-tile kt covers channels [ci, ci + 16) of tap `tap`
    auto load_tiles = [&](int kt) {
        const int dt = tap - a.pad_l;
#pragma unroll
        for (int p = 0; p < MI; ++p) {
            const int ts = tt[p] + dt;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < a.T) v = *reinterpret_cast<const float4*>(xrow[p] + (long long)dt * a.ldx + ci);
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_ok) v = *reinterpret_cast<const float4*>(wcol + (size_t)(kt * CG_BK + p * 8) * a.N);
            rbv[p] = v;
        }
        ci += CG_BK;
        if (ci >= a.Cin) { ci = 0; ++tap; }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int p = 0; p < MI; ++p) {
            const int r = a_r0 + p * 64;
            As[buf][a_kq * 4 + 0][r] = ra[p].x;
            As[buf][a_kq * 4 + 1][r] = ra[p].y;
            As[buf][a_kq * 4 + 2][r] = ra[p].z;
            As[buf][a_kq * 4 + 3][r] = ra[p].w;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) *reinterpret_cast<float4*>(&Bs[buf][b_k0 + p * 8][b_nq * 4]) = rbv[p];
    };

    if (kt0 < nk) {
        load_tiles(kt0);
        store_tiles(kt0 & 1);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = kt0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < CG_BK; ks += 2) {
            const float b0 = Bs[buf][ks + lh][wn * 64 + l31];
            const float b1 = Bs[buf][ks + lh][wn * 64 + 32 + l31];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const float av = As[buf][ks + lh][wm * 32 * MI + i * 32 + l31];
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[i][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= a.N) continue;
            const float sc = a.scale ? a.scale[n] : 1.f;
            const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.Mtot) continue;
                if constexpr (SPLIT) {
                    a.part[((size_t)blockIdx.z * a.Mtot + m) * a.N + n] = acc[i][j][r];
                } else {
                    float v = acc[i][j][r] * sc + sh;
                    if (a.act == 1) v = fmaxf(v, 0.f);
                    else if (a.act == 2) v = tanhf(v);
                    if (a.res) v += a.res[(size_t)m * a.ldr + n];
                    a.y[(size_t)m * a.ldy + n] = v;
                }
            }
        }
}
// y = epilogue(sum_z part[z]) of a split conv_gemm_fast_kernel launch (slice order: deterministic)
__global__ __launch_bounds__(256) void conv_split_finish_kernel(ConvArgs a) {
    const long long n_el = (long long)a.Mtot * a.N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_el; i += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(i / a.N), n = (int)(i - (long long)m * a.N);
        float s = 0.f;
        for (int z = 0; z < a.splits; ++z) s += a.part[(size_t)z * n_el + i];
        float v = s * (a.scale ? a.scale[n] : 1.f) + (a.shift ? a.shift[n] : 0.f);
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = tanhf(v);
        if (a.res) v += a.res[(size_t)m * a.ldr + n];
        a.y[(size_t)m * a.ldy + n] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 3 forward fast path (conv_gemm_kt_kernel): K-contiguous LDS tiles.  Both operands are read from memory as float4 along K
// (the activations are K-contiguous anyway; the kernel comes transposed, ConvArgs::wt [N][ktaps*Cin]) and go to LDS as ONE
// 128-bit store per thread and operand row: tile [rows][16 k], the 16-byte chunk c (k = 4c .. 4c + 3) of row r at chunk slot
// c ^ ((r >> 1) & 3), which makes both the stores and the reads conflict-free.  A lane reads its 8 k of a k-tile as two
// 128-bit LDS loads: lane half lh holds k = 8 lh .. 8 lh + 7, MFMA step s multiplies the k pair (s, 8 + s) -- 8 LDS reads per
// 32 MFMAs instead of 32, no transposing scalar stores.  (The summation order inside a k-tile differs from conv_gemm_kernel's:
// results agree to fp32 rounding, not bit for bit.)
// Tile shapes: NS sub-tiles of 32 rows per wave.  WN = 2: waves 2 x 2, tile 64 NS x 128 (NS = 1, 2: the 64- and 128-row tiles
// of conv_gemm_fast_kernel).  WN = 4: the four waves side by side, each 32 NS rows x 32 columns -- NS = 5 gives a 160 x 128
// tile, for output shapes whose 128-row tiling leaves a ragged last round on the 256 CUs (PostNetV2 at B = 64: 1600 tiles =
// 6.25 per CU -> some CUs run 7; 160 rows: 1280 tiles = 5 per CU; launch_conv picks by that quotient).  Measured at that
// shape: 1.30 -> 1.19 ms (layout) -> 1.06 ms (160-row tiles, three workgroups per CU) per layer.
// ------------------------------------------------------------------------------------------------
template <int NS, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WN == 4 ? 3 : 2, WN == 4 ? 3 : 4)))
void conv_gemm_kt_kernel(ConvArgs a) {
    static_assert(WN == 2 || WN == 4, "waves across the 128 columns");
    constexpr int WM = 4 / WN;                       // waves down the rows
    constexpr int NJ = 4 / WN;                       // 32-column sub-tiles per wave (WN = 2: 2, WN = 4: 1)
    constexpr int BM = 32 * NS * WM;
    constexpr int NP = (BM + 63) / 64;               // loader passes over the rows of the A tile
    __shared__ __attribute__((aligned(16))) float As[2][BM][CG_BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][CG_BN][CG_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * CG_BN;
    const int Ktot = a.ktaps * a.Cin;
    const int nk = Ktot / CG_BK;
    f32x16 acc[NS][NJ];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kq = tid & 3, r0 = tid >> 2;           // loader: 16-byte chunk kq of tile rows r0, r0 + 64, ...
    int tt[NP];
    const float* xrow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int m = m0 + r0 + p * 64;
        tt[p] = (m < a.Mtot && r0 + p * 64 < BM) ? m % a.T : -(1 << 28);
        xrow[p] = a.x + (size_t)min(m, a.Mtot - 1) * a.ldx + kq * 4;
    }
    const float* wrow[2];
    bool w_ok[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int n = n0 + r0 + p * 64;
        w_ok[p] = n < a.N;
        wrow[p] = a.wt + (size_t)min(n, a.N - 1) * Ktot + kq * 4;
    }
    float4 ra[NP], rb[2];
    int tap = 0, ci = 0;                             // k-tile kt covers channels [ci, ci + 16) of tap `tap`
    auto load_tiles = [&](int kt) {
        const int dt = tap - a.pad_l;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int ts = tt[p] + dt;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < a.T) v = *reinterpret_cast<const float4*>(xrow[p] + (long long)dt * a.ldx + ci);
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w_ok[p]) v = *reinterpret_cast<const float4*>(wrow[p] + kt * CG_BK);
            rb[p] = v;
        }
        ci += CG_BK;
        if (ci >= a.Cin) { ci = 0; ++tap; }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = r0 + p * 64;
            if (BM % 64 == 0 || r < BM) *reinterpret_cast<float4*>(&As[buf][r][(kq ^ ((r >> 1) & 3)) * 4]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = r0 + p * 64;
            *reinterpret_cast<float4*>(&Bs[buf][r][(kq ^ ((r >> 1) & 3)) * 4]) = rb[p];
        }
    };
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        f32x4 av[NS][2], bv[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = wn * 32 * NJ + j * 32 + l31;
#pragma unroll
            for (int h = 0; h < 2; ++h) bv[j][h] = *reinterpret_cast<const f32x4*>(&Bs[buf][r][((lh * 2 + h) ^ ((r >> 1) & 3)) * 4]);
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int r = wm * 32 * NS + i * 32 + l31;
#pragma unroll
            for (int h = 0; h < 2; ++h) av[i][h] = *reinterpret_cast<const f32x4*>(&As[buf][r][((lh * 2 + h) ^ ((r >> 1) & 3)) * 4]);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s >> 2][s & 3], bv[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * 32 * NJ + j * 32 + l31;
        if (n >= a.N) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * NS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.Mtot) continue;
                float v = acc[i][j][r] * sc + sh;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = tanhf(v);
                if (a.res) v += a.res[(size_t)m * a.ldr + n];
                a.y[(size_t)m * a.ldy + n] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight-gradient fast path (trans_a mode of conv_gemm_kernel with float4 operands): the output rows of a thread
// (tap j, channels i .. i + 3) never change, and the position of a reduction row inside its utterance advances by the
// k-tile, so the main loop holds no integer division.  Same tiles, MFMA loop, split-K scheme and summation order as
// the generic kernel (bit-identical results).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_gemm_wgrad_fast_kernel(ConvArgs a) {
    __shared__ float As[2][CG_BK][CG_LD];
    __shared__ float Bs[2][CG_BK][CG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * CG_BM, n0 = blockIdx.x * CG_BN;
    const int Ktot = a.red;
    const int nk = (Ktot + CG_BK - 1) / CG_BK;
    const int nz = a.splits > 1 ? a.splits : 1, z = a.splits > 1 ? (int)blockIdx.z : 0;
    const int kt0 = (int)((long long)nk * z / nz), kt1 = (int)((long long)nk * (z + 1) / nz);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int b_nq = tid & 31, b_k0 = tid >> 5;
    const int mo = m0 + b_nq * 4;                       // output rows mo .. mo + 3 = (tap, channels): fixed per thread
    const bool mo_ok = mo < a.Mtot;
    const int tap = mo_ok ? mo / a.Cin : 0, ch = mo_ok ? mo - tap * a.Cin : 0;
    const int dt = tap - a.pad_l;
    const bool taps = a.ktaps > 1;                      // plain GEMM (one tap): no utterance boundary to respect
    int tt[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) tt[p] = taps ? (kt0 * CG_BK + b_k0 + p * 8) % a.T : 0;
    const int n_b = n0 + b_nq * 4;
    const bool b_ok = n_b < a.N;

    float4 ra[2], rbv[2];
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int rr = kt * CG_BK + b_k0 + p * 8;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (rr < Ktot) {
                const int ts = tt[p] + dt;
                if (mo_ok && (!taps || (ts >= 0 && ts < a.T)))
                    va = *reinterpret_cast<const float4*>(a.x + (long long)(rr + dt) * a.ldx + ch);
                if (b_ok) vb = *reinterpret_cast<const float4*>(a.w + (size_t)rr * a.N + n_b);
            }
            ra[p] = va; rbv[p] = vb;
            if (taps) { tt[p] += CG_BK; while (tt[p] >= a.T) tt[p] -= a.T; }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<float4*>(&As[buf][b_k0 + p * 8][b_nq * 4]) = ra[p];
            *reinterpret_cast<float4*>(&Bs[buf][b_k0 + p * 8][b_nq * 4]) = rbv[p];
        }
    };
    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles(0);
    }
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tiles(kt + 1);
#pragma unroll
        for (int ks = 0; ks < CG_BK; ks += 2) {
            const float a0 = As[buf][ks + lh][wm * 64 + l31];
            const float a1 = As[buf][ks + lh][wm * 64 + 32 + l31];
            const float b0 = Bs[buf][ks + lh][wn * 64 + l31];
            const float b1 = Bs[buf][ks + lh][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < kt1) store_tiles(buf ^ 1);
        __syncthreads();
    }
    float* dst = a.splits > 1 ? a.part + (size_t)z * a.Mtot * a.N : a.y;
    const int ld = a.splits > 1 ? a.N : a.ldy;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < a.Mtot) dst[(size_t)m * ld + n] = acc[i][j][r];
            }
        }
}

// Deferred projection of a teacher-forced decode: rows (t, b) of the batched GEMM output [S*B][ld] -> mel [b][t][M*r]
// (row stride mel_ld, step stride M*r) and the stop logit [b][t] (row stride stop_ld).
__global__ __launch_bounds__(256) void proj_scatter_kernel(const float* __restrict__ y, int ld, int S, int B, int ncol, float* __restrict__ mel,
                                                           long long mel_ld, float* __restrict__ stop, int stop_ld) {
    const long long n = (long long)S * B * (ncol + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (ncol + 1));
        const long long tb = i / (ncol + 1);
        const int b = (int)(tb % B), t = (int)(tb / B);
        const float v = y[tb * ld + c];
        if (c < ncol) mel[(long long)b * mel_ld + (long long)t * ncol + c] = v;
        else stop[(long long)b * stop_ld + t] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Teacher-forced PreNets of ALL steps at once (TrainingHelper / ValidationHelper with teacher forcing feed known
// frames, helpers.py:86,182): inputs gathered for every step, the Dense layers as conv_gemm_kernel GEMMs over S*B rows,
// then the pass below (dropout in training mode, frag16 operands of the attention LSTM per step).
// ------------------------------------------------------------------------------------------------
// xin[t][b][:] = targets[b][t - 1][-M*nf:] (t >= 1), zeros at t = 0: inputs of PreNet layer 0 over the whole tape
__global__ __launch_bounds__(256) void feed_gather_kernel(const float* __restrict__ targets, float* __restrict__ xin, int B, int S, int M, int r, int nf) {
    const int W = M * nf;
    const long long n = (long long)S * B * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const long long tb = i / W;
        const int b = (int)(tb % B), t = (int)(tb / B);
        xin[i] = t == 0 ? 0.f : targets[((size_t)b * S + (t - 1)) * r * M + (size_t)M * (r - nf) + c];
    }
}

// PreNet dropout over all teacher-forced steps at once: x [S][B][P] in place (x = relu(dense) already), optionally also as the
// per-step frag16 operands of the attention LSTM (slot t at frag + t * frag_stride)
__global__ __launch_bounds__(256) void prenet_dropout_all_kernel(float* __restrict__ x, float* __restrict__ frag, long long frag_stride, int S, int B, int P,
                                                                 int MB, float rate, uint32_t seed, int stream) {
    const long long n = (long long)S * B * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % P);
        const long long tb = i / P;
        const int b = (int)(tb % B), t = (int)(tb / B);
        float v = x[i];
        if (rate > 0.f) v = dropout_train(v, rate, t2_u01(seed, stream, t, b, c));
        x[i] = v;
        if (frag != nullptr) frag[(size_t)t * frag_stride + frag_idx(b, c, MB)] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Tacotron v1 cells
// ------------------------------------------------------------------------------------------------
// TF GRUCell step in two launches (reference tacotron/rnn_impl.py:38): the reset gate
// multiplies h BEFORE the candidate matmul, so the candidate GEMM needs every r of the step:
//   phase 1: [r | u] = sigmoid([x | h] . Wg + bg)   -> writes r*h (frag16 operand of phase 2) and u
//   phase 2: c = tanh([x | r*h] . Wc + bc);  h' = u*h + (1-u)*c   -> state, output (h' or x_res + h':
//            tf.nn.rnn_cell.ResidualWrapper, tacotron_v1.py:72-73), optional plain sequence output with
//            dynamic_rnn's sequence_length semantics (zero output / state copied through past the end).
struct GruEpi {
    int H;
    FragBuf hprev;            // h_{t-1}
    FragBuf rh;               // r*h destination (phase 1)
    float* u;                 // plain [MB*16][H]
    FragBuf hstate;           // h_t destination (phase 2)
    FragBuf out;              // output destination, base may be null
    FragBuf xres;             // residual input, base null: none
    const int* lens;          // optional [B]
    float* plain_out;         // optional [B][T][ld]
    int plain_bs, plain_ts, plain_off, reverse;
    float* y_all;             // optional plain [S][B][H] store of the output of every step (deferred projection)
};

template <int MB, int PHASE>
__global__ __launch_bounds__(T2_SKINNY_THREADS) void gru_step_kernel(SkinnyCore a, GruEpi e) {
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    f32x4 acc[MB];
    const int wg = blockIdx.x;
    const int t = skinny_core<MB, 0>(a, 0, wg, acc, [](int) {});
    if (t < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    __syncthreads();
    const int ncol = PHASE == 1 ? 2 * e.H : e.H;
    const float* hprev = frag_slot(e.hprev, t);
    for (int idx = threadIdx.x; idx < MB * 16 * 16; idx += T2_SKINNY_THREADS) {
        const int row = idx >> 4, c = idx & 15;
        const int col = wg * 16 + c;
        if (row >= a.B || col >= ncol) continue;
        const float v = red_sum<MB>(red, row, c) + a.bias[0][col];
        if (PHASE == 1) {
            const float g = sigmoidf_(v);
            if (col < e.H) frag_slot(e.rh, t)[frag_idx(row, e.rh.koff + col, MB)] = g * hprev[frag_idx(row, e.hprev.koff + col, MB)];
            else e.u[(size_t)row * e.H + col - e.H] = g;
        } else {
            const float cand = tanhf(v);
            const float u = e.u[(size_t)row * e.H + col];
            const float hp = hprev[frag_idx(row, e.hprev.koff + col, MB)];
            float hn = u * hp + (1.0f - u) * cand;
            float outv = hn;
            bool live = true;
            int len = 0;
            if (e.lens != nullptr) {
                len = e.lens[row];
                live = t < len;
                if (!live) { hn = hp; outv = 0.f; }
            }
            frag_slot(e.hstate, t)[frag_idx(row, e.hstate.koff + col, MB)] = hn;
            if (e.xres.base != nullptr) outv += frag_slot(e.xres, t)[frag_idx(row, e.xres.koff + col, MB)];
            if (e.out.base != nullptr) frag_slot(e.out, t)[frag_idx(row, e.out.koff + col, MB)] = outv;
            if (e.y_all != nullptr) e.y_all[((size_t)t * a.B + row) * e.H + col] = outv;
            if (e.plain_out != nullptr && live) {
                const int tpos = (e.reverse && e.lens != nullptr) ? (len - 1 - t) : t;
                e.plain_out[(size_t)row * e.plain_bs + (size_t)tpos * e.plain_ts + e.plain_off + col] = outv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Whole-sequence GRU of the CBHG's bidirectional_dynamic_rnn (modules.py:222-229) as ONE persistent launch: a workgroup
// owns 16 batch rows of one direction and walks all T steps by itself (rows are independent, so no cross-workgroup
// synchronisation exists).  The recurrent halves of the kernels ([H, 2H] gates, [H, H] candidate; H = 128: 192 KB)
// stay in REGISTERS for the whole sequence as MFMA B fragments (wave w: gate column tiles 2w, 2w + 1 and candidate
// tile w = 96 VGPRs); the input halves were applied to all positions beforehand by one GEMM (xz, biases included).
// h, r*h and u live in LDS.  Per step: gates MFMAs -> sigmoid -> r*h, u -> barrier -> candidate MFMAs -> tanh, blend,
// sequence-length masking (state copied through, no output past the end), per-row reversed positions for the backward
// direction -> barrier.  TF GRUCell arithmetic as gru_step_kernel.
// ------------------------------------------------------------------------------------------------
struct GruSeqArgs {
    const float* xz[2];        // per direction [B*T][3H]: x . [Wg_x | Wc_x] + [bg | bc] at every position
    const float* whg[2];       // [H][2H] recurrent rows of the gates kernel
    const float* whc[2];       // [H][H] recurrent rows of the candidate kernel
    const int* lens;           // [B]
    float* out;                // [B][T][2H]: forward half | backward half
    int B, T;
    float* tape_ru[2];         // training forward (csrc/v1_train.h): per direction [B*T][2H] gates r | u and [B*T][H] candidate c at
    float* tape_c[2];          // every live position (the backward recurrence reads them); null: nothing is taped
};

template <int H>
__global__ __launch_bounds__(512) void gru_seq_kernel(GruSeqArgs a) {
    static_assert(H == 128, "register plan below: 8 waves x (2 gate tiles + 1 candidate tile) of 16 columns");
    constexpr int KS = H / 4, LDH = H + 4;
    __shared__ __attribute__((aligned(16))) float hs[16][LDH];
    __shared__ __attribute__((aligned(16))) float rhs[16][LDH];
    __shared__ float us[16][H];
    const int dir = blockIdx.y, row0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 4, c16 = lane & 15;
    const float* __restrict__ whg = a.whg[dir];
    const float* __restrict__ whc = a.whc[dir];
    const float* __restrict__ xz = a.xz[dir];
    float* const tape_ru = a.tape_ru[dir];
    float* const tape_c = a.tape_c[dir];
    float wg0[KS], wg1[KS], wc[KS];
    // MFMA step ks takes k = 16 (ks / 4) + 4 kg + ks % 4 from lane group kg: a lane's four k of one k-block are contiguous,
    // so the A operand of four steps is ONE 16-byte LDS read
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 16 * (ks >> 2) + 4 * kg + (ks & 3);
        wg0[ks] = whg[(size_t)k * 2 * H + (2 * wave) * 16 + c16];
        wg1[ks] = whg[(size_t)k * 2 * H + (2 * wave + 1) * 16 + c16];
        wc[ks] = whc[(size_t)k * H + wave * 16 + c16];
    }
    for (int i = threadIdx.x; i < 16 * LDH; i += 512) { (&hs[0][0])[i] = 0.f; (&rhs[0][0])[i] = 0.f; }
    // this lane's four output rows (D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg)
    int len4[4];
    long long base4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = row0 + 4 * kg + r;
        len4[r] = b < a.B ? a.lens[b] : 0;
        base4[r] = (long long)min(b, a.B - 1) * a.T;
    }
    const int cg0 = (2 * wave) * 16 + c16, cg1 = cg0 + 16, cc = wave * 16 + c16;     // gate columns (r | u), candidate column
    auto pos_of = [&](int r, int t) { return dir ? len4[r] - 1 - t : t; };
    float xg0[4], xg1[4], xc[4];
    // branch-free prefetch: dead steps (t >= len) read a clamped, valid position and their values are never used (the state is
    // copied through).  A conditional load here made the compiler keep x in scratch memory and wait for every load at once.
    auto fetch = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pos = min(max(pos_of(r, t), 0), a.T - 1);
            const float* p = xz + (size_t)(base4[r] + pos) * 3 * H;
            xg0[r] = p[cg0]; xg1[r] = p[cg1]; xc[r] = p[2 * H + cc];
        }
    };
    fetch(0);
    __syncthreads();
    for (int t = 0; t < a.T; ++t) {
        // ---- gates: [r | u] = sigmoid(x . Wg_x + bg + h . Wg_h)
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 av[KS / 4];
#pragma unroll
        for (int kb = 0; kb < KS / 4; ++kb) av[kb] = *reinterpret_cast<const f32x4*>(&hs[c16][16 * kb + 4 * kg]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            g0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][ks & 3], wg0[ks], g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][ks & 3], wg1[ks], g1, 0, 0, 0);
        }
        // all of the prefetched x is waited for HERE (the loads have had the whole gates phase): the vector-memory counter is
        // in order, so a wait for xc placed after this step's stores would wait for those stores too
        asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kg + r;
            const float s0 = fast_sigmoidf(g0[r] + xg0[r]), s1 = fast_sigmoidf(g1[r] + xg1[r]);
            if (cg0 < H) rhs[row][cg0] = s0 * hs[row][cg0]; else us[row][cg0 - H] = s0;
            if (cg1 < H) rhs[row][cg1] = s1 * hs[row][cg1]; else us[row][cg1 - H] = s1;
            if (tape_ru != nullptr && t < len4[r]) {
                float* tp = tape_ru + (size_t)(base4[r] + pos_of(r, t)) * 2 * H;
                tp[cg0] = s0; tp[cg1] = s1;
            }
        }
        __syncthreads();
        // ---- candidate and blend: c = tanh(x . Wc_x + bc + (r*h) . Wc_h);  h' = u*h + (1 - u)*c
        f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = ca;
#pragma unroll
        for (int kb = 0; kb < KS / 4; ++kb) av[kb] = *reinterpret_cast<const f32x4*>(&rhs[c16][16 * kb + 4 * kg]);
#pragma unroll
        for (int ks = 0; ks < KS; ks += 2) {
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][ks & 3], wc[ks], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][(ks & 3) + 1], wc[ks + 1], cb, 0, 0, 0);
        }
        ca += cb;
        float hn4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kg + r;
            const float cand = fast_tanhf(ca[r] + xc[r]);
            const float u = us[row][cc], hp = hs[row][cc];
            const bool live = t < len4[r];
            hn4[r] = live ? u * hp + (1.0f - u) * cand : hp;
            if (live) a.out[(size_t)(base4[r] + pos_of(r, t)) * 2 * H + dir * H + cc] = hn4[r];
            if (live && tape_c != nullptr) tape_c[(size_t)(base4[r] + pos_of(r, t)) * H + cc] = cand;
        }
        if (t + 1 < a.T) fetch(t + 1);
        // own columns only: no other wave reads them before the barrier (the gates of step t are done, the blend of the
        // other waves touches their own columns)
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[4 * kg + r][cc] = hn4[r];
        __syncthreads();           // h of step t complete; every wave is past its reads of r*h
    }
}

// ------------------------------------------------------------------------------------------------
// Reverse-time mirror of gru_seq_kernel: BPTT through the whole-sequence GRU of one direction (training step of the Tacotron-1
// PostNet model, csrc/v1_train.h).  Same ownership: a workgroup walks 16 batch rows of one direction alone; the TRANSPOSED
// recurrent halves of the two kernels stay in registers as MFMA B fragments (wave w owns hidden units 16 w .. 16 w + 15:
// 32 registers of Wc_h^T, 64 of Wg_h^T); the gradient of the state is carried in registers (D layout of the MFMA).
// TF GRUCell: [r | u] = sigmoid(zg), c = tanh(zc), zc = x.Wc_x + (r*h).Wc_h + bc, h' = u*h + (1 - u)*c.
// Per step, last step first:  dh = d out[pos] + carry;  du = dh (h - c);  dc = dh (1 - u);  carry = dh u;
//   dzc = dc (1 - c^2);  d(r*h) = dzc . Wc_h^T;  dr = d(r*h) h;  carry += d(r*h) r;  dzr = dr r (1 - r);  dzu = du u (1 - u);
//   carry += [dzr | dzu] . Wg_h^T.   dzg = [dzr | dzu] and dzc are written per position: they are d(x . W_x + b), so every
// weight / input gradient is a tape-wide GEMM afterwards.  Dead steps (t >= len) pass the carry through.
// ------------------------------------------------------------------------------------------------
struct GruSeqBwdArgs {
    const float* dout;         // [B][T][2H] gradient of the concatenated outputs
    const float* out;          // [B][T][2H] forward outputs (h of every live position)
    const float* tape_ru[2];   // [B*T][2H] r | u
    const float* tape_c[2];    // [B*T][H]
    const float* whg[2];       // [H][2H] recurrent rows of the gates kernel
    const float* whc[2];       // [H][H] recurrent rows of the candidate kernel
    const int* lens;
    float* dzg[2];             // [B*T][2H] (zeroed by the caller: dead positions stay zero)
    float* dzc[2];             // [B*T][H]
    int B, T;
};

template <int H>
__global__ __launch_bounds__(512) void gru_seq_bwd_kernel(GruSeqBwdArgs a) {
    static_assert(H == 128, "register plan: 8 waves x 16 hidden units");
    constexpr int KC = H / 4, KG = 2 * H / 4, LDC = H + 4, LDG = 2 * H + 4;
    __shared__ __attribute__((aligned(16))) float dzc_s[16][LDC];
    __shared__ __attribute__((aligned(16))) float dzg_s[16][LDG];
    const int dir = blockIdx.y, row0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kg = lane >> 4, c16 = lane & 15;
    const int cc = wave * 16 + c16;                       // this lane's hidden unit
    const float* __restrict__ whg = a.whg[dir];
    const float* __restrict__ whc = a.whc[dir];
    // B fragments of the transposed kernels: MFMA step ks takes k = 16 (ks / 4) + 4 kg + ks % 4; B[k][n = cc] = W[cc][k]
    float wcT[KC], wgT[KG];
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) wcT[ks] = whc[(size_t)cc * H + 16 * (ks >> 2) + 4 * kg + (ks & 3)];
#pragma unroll
    for (int ks = 0; ks < KG; ++ks) wgT[ks] = whg[(size_t)cc * 2 * H + 16 * (ks >> 2) + 4 * kg + (ks & 3)];
    int len4[4];
    long long base4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = row0 + 4 * kg + r;
        len4[r] = b < a.B ? a.lens[b] : 0;
        base4[r] = (long long)min(b, a.B - 1) * a.T;
    }
    auto pos_of = [&](int r, int t) { return dir ? len4[r] - 1 - t : t; };
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    float rr4[4], uu4[4], cn4[4], hp4[4], go4[4];          // step t
    float nrr[4], nuu[4], ncn[4], nhp[4], ngo[4];          // step t - 1, in flight during step t
    // branch-free prefetch (see gru_seq_kernel): clamped positions, then a select -- the tape of dead positions was never
    // written, so its contents must not reach the arithmetic.  Issued right after the first phase of step t and consumed at
    // the top of step t - 1: two MFMA phases hide the latency.
    const float* __restrict__ tru = a.tape_ru[dir];
    const float* __restrict__ tcn = a.tape_c[dir];
    auto fetch = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = t >= 0 && t < len4[r];
            const size_t p = (size_t)(base4[r] + min(max(pos_of(r, t), 0), a.T - 1));
            const size_t pp = (size_t)(base4[r] + min(max(pos_of(r, t - 1), 0), a.T - 1));
            const float v_r = tru[p * 2 * H + cc], v_u = tru[p * 2 * H + H + cc], v_c = tcn[p * H + cc];
            const float v_g = a.dout[p * 2 * H + dir * H + cc], v_h = a.out[pp * 2 * H + dir * H + cc];
            nrr[r] = live ? v_r : 0.f; nuu[r] = live ? v_u : 0.f; ncn[r] = live ? v_c : 0.f; ngo[r] = live ? v_g : 0.f;
            nhp[r] = (live && t > 0) ? v_h : 0.f;
        }
    };
    fetch(a.T - 1);
    for (int t = a.T - 1; t >= 0; --t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { rr4[r] = nrr[r]; uu4[r] = nuu[r]; cn4[r] = ncn[r]; hp4[r] = nhp[r]; go4[r] = ngo[r]; }
        float dh4[4], du4[4];
        size_t p4[4];
        bool live4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kg + r;
            live4[r] = t < len4[r];
            p4[r] = (size_t)(base4[r] + (live4[r] ? pos_of(r, t) : 0));
            dh4[r] = live4[r] ? go4[r] + carry[r] : 0.f;
            du4[r] = dh4[r] * (hp4[r] - cn4[r]);
            const float dzc = dh4[r] * (1.0f - uu4[r]) * (1.0f - cn4[r] * cn4[r]);
            const float dzu = du4[r] * uu4[r] * (1.0f - uu4[r]);
            dzc_s[row][cc] = dzc;
            dzg_s[row][H + cc] = dzu;
            if (live4[r]) { a.dzc[dir][p4[r] * H + cc] = dzc; a.dzg[dir][p4[r] * 2 * H + H + cc] = dzu; }
        }
        fetch(t - 1);
        __syncthreads();
        // d(r*h) = dzc . Wc_h^T
        f32x4 da = {0.f, 0.f, 0.f, 0.f}, db = da;
        {
            f32x4 av[KC / 4];
#pragma unroll
            for (int kb = 0; kb < KC / 4; ++kb) av[kb] = *reinterpret_cast<const f32x4*>(&dzc_s[c16][16 * kb + 4 * kg]);
#pragma unroll
            for (int ks = 0; ks < KC; ks += 2) {
                da = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][ks & 3], wcT[ks], da, 0, 0, 0);
                db = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][(ks & 3) + 1], wcT[ks + 1], db, 0, 0, 0);
            }
            da += db;
        }
        float keep4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kg + r;
            const float drh = da[r];
            const float dzr = drh * hp4[r] * rr4[r] * (1.0f - rr4[r]);
            keep4[r] = dh4[r] * uu4[r] + drh * rr4[r];
            dzg_s[row][cc] = dzr;
            if (live4[r]) a.dzg[dir][p4[r] * 2 * H + cc] = dzr;
        }
        __syncthreads();
        // carry += [dzr | dzu] . Wg_h^T
        f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = ga;
        {
            f32x4 av[KG / 4];
#pragma unroll
            for (int kb = 0; kb < KG / 4; ++kb) av[kb] = *reinterpret_cast<const f32x4*>(&dzg_s[c16][16 * kb + 4 * kg]);
#pragma unroll
            for (int ks = 0; ks < KG; ks += 2) {
                ga = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][ks & 3], wgT[ks], ga, 0, 0, 0);
                gb = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks >> 2][(ks & 3) + 1], wgT[ks + 1], gb, 0, 0, 0);
            }
            ga += gb;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) carry[r] = live4[r] ? keep4[r] + ga[r] : carry[r];
        __syncthreads();           // every wave is past its reads of dzc_s / dzg_s before the next step rewrites them
    }
}

// tf.layers.MaxPooling1D(pool_size=2, strides=1, padding="SAME") (modules.py:182): y[t] = max(x[t], x[t+1]),
// the last frame of every utterance passes through.  x, y: [B*T][C]
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ x, float* __restrict__ y, long long Mtot, int T, int C) {
    const long long n = Mtot * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        float v = x[i];
        if ((int)(m % T) + 1 < T) v = fmaxf(v, x[i + C]);
        y[i] = v;
    }
}

// HighwayNet.call (modules.py:124-127): ht = [x.Wh + bh | x.Wt + bt] (one GEMM);  y = relu(h)*sigmoid(t) + x*(1 - sigmoid(t))
__global__ __launch_bounds__(256) void highway_kernel(const float* __restrict__ ht, const float* __restrict__ x, float* __restrict__ y,
                                                      long long Mtot, int H) {
    const long long n = Mtot * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / H;
        const int c = (int)(i - m * H);
        const float h = fmaxf(ht[m * 2 * H + c], 0.f);
        const float tg = sigmoidf_(ht[m * 2 * H + H + c]);
        y[i] = h * tg + x[i] * (1.0f - tg);
    }
}
