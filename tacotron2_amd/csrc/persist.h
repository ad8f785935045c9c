// Persistent free-running decoder: the whole dynamic_decode loop of the V2 cell stack as ONE launch.
//
// Replaces, for the paper-size dimension class (last PreNet 256, encoder 512, attention / decoder LSTMs 1024,
// attention dim 128), the per-step launch sequence of launch_step():
//   OutputAndStopTokenWrapper.call (rnn_wrappers.py:105-109) <- DecoderRNNV2.call (tacotron_v2.py:155-161,177-178)
//   <- AttentionRNN.call (rnn_wrappers.py:118-127,143-144) with LocationSensitiveAttention.__call__
//   (tacotron_v2.py:118-141), driven by StopTokenBasedInferenceHelper / ValidationHelper(teacher_forcing=False)
//   (helpers.py:83-94, :131-138) under dynamic_decode (tacotron_v1.py:149-150).
//
// Structure.  256 workgroups (one per CU) stay resident for the whole decode.  Workgroup g owns hidden units
// 4g..4g+3 (16 gate columns) of all three ZoneoutLSTM cells: their cell / hidden state lives in registers for
// the whole decode, the gate GEMM of a layer is [rows x K] . [K x 16] on v_mfma_f32_16x16x4_f32 with K split
// over the 8 waves.  A decoder step is six all-to-all exchanges ("hops"):
//   P6 last PreNet layer -> H6 -> P1 attention LSTM -> H1 -> P2 attention -> H2 -> P3 LSTM 1 -> H3 ->
//   P4 LSTM 2 -> H4 -> P5 mel / stop projection + first PreNet layer of the next step -> H5 -> stop rule.
// Every K range whose operand is already known (recurrent halves, the attention cell output into LSTM 1) is
// multiplied in the gaps between "arrive" and "wait" of a hop and accumulates in the per-wave MFMA accumulators
// that the critical part of the layer continues, so nothing hoisted ever goes through memory.  Weights of the
// critical K ranges are loaded to registers BEFORE the wait (they depend on nothing).
// The attention of one batch row is split over C workgroups by encoder position; partners exchange their energies
// (one small group hop), each normalises the full softmax redundantly and produces its own slice of the context.
//
// Hand-off protocol (MI355X_MICROARCH.md, "Valid forms", sc1 row 1): every byte that crosses workgroups is
// stored write-through (buffer_store ... sc1), every storing wave drains (s_waitcnt vmcnt(0)), the workgroup
// barriers, ONE lane adds to a monotonic counter shard with an agent-scope atomic; consumers poll all shards with
// sc1 loads from one wave, barrier, then read the payload with sc1 buffer loads only (L1 is bypassed, so no
// acquire invalidate is needed).  Nothing depends on dispatch order or XCD placement.  Every spin is bounded by a
// wall-clock limit; a timeout sets an error word and makes every workgroup leave the loop.
#pragma once
#include "kernels.h"

namespace pz {

constexpr int NWG = 256;            // workgroups = hidden units / 4
constexpr int NT = 512;             // threads per workgroup
constexpr int NWAVE = 8;
constexpr int KBX = 16;             // k-blocks (of 16) of the last PreNet layer's output (256)
constexpr int KBE = 32;             // encoder width 512
constexpr int KBH = 64;             // LSTM width 1024
constexpr int UA = 128;             // attention dim
constexpr int EW = 512;             // encoder width
constexpr int NSHARD = 8;
constexpr unsigned SPIN_LIMIT_TICKS = 25u * 1000u * 1000u;   // s_memrealtime ticks (100 MHz): 250 ms per wait

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct PArgs {
    const float *wA, *bA, *w1, *b1, *w2, *b2;   // packed LSTM slabs [256][kbtot][64][4] (pack_skinny), biases [256][16]
    const float *wP, *bP;                       // projection + composed first PreNet layer: [28][64][64][4], bias [448]
    const float *wN, *bN;                       // last PreNet layer: [16][16][64][4], bias [256]
    const float *wq, *conv_w, *conv_b, *wloc, *v_a, *b_a;
    const float *keys, *values;                 // [B][T][128], [B][T][512]
    const int* lens;
    float* xs;                                  // exchange slab (zeroed before the launch)
    unsigned xs_bytes;
    unsigned oX1, oX2, oHAr, oHA, oCTX, oY1, oH1, oY2, oH2, oQP, oEN, oCUM, oSTOP;   // byte offsets into xs
    unsigned* bar;                              // [NSHARD][32] hop counters, then [64][32] attention group counters (zeroed)
    int* ctl;
    float *mel, *stop, *hist;
    long long mel_ld, mel_ts;
    int stop_ld;
    int B, T, C, CH, F, KA, MR, max_steps, min_iters, stop_en, cumulative;
    float zc, zh, fb;
};

__host__ __device__ inline int lds_floats(int MB, int T, int CH) {
    const int CHP = (CH + 15) & ~15, TP = (T + 3) & ~3;
    const int red = NWAVE * MB * 64 * 4;
    return red + MB * 64 + 32 * 32 + 32 + (CHP + 32) + CHP * 36 + 16 * UA + UA + NWAVE * CHP + 2 * TP + 2048 + 64 + 16 * 16 + 16 + 64;
}

// ---- exchange-slab accessors: write-through stores, L1-bypassing loads ----
__device__ __forceinline__ f32x4 ld16(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 16));
}
__device__ __forceinline__ void st16(rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, 16);
}
__device__ __forceinline__ float ld4(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 16));
}
__device__ __forceinline__ void st4(rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)off, 0, 16);
}

// ---- hops ----
// arrive: every wave has drained its stores; one lane adds to this workgroup's counter shard
__device__ __forceinline__ void hop_arrive(unsigned* bar) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar + (blockIdx.x & (NSHARD - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: wave 0 polls the NSHARD shards (one lane each) until each has seen `per_shard` arrivals.  Returns false on
// timeout or when another wait of this workgroup already failed (s_abort).
__device__ __forceinline__ bool hop_wait(unsigned* bar, unsigned per_shard, int* s_abort) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool ok = false, bad = false;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            unsigned v = per_shard;
            if (lane < NSHARD) v = __hip_atomic_load(bar + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = __all((int)(v - per_shard) >= 0);
            if (ok) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { bad = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (bad && lane == 0) *s_abort = 1;
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    return *s_abort == 0;
}
// one word, one poller (attention group counters)
__device__ __forceinline__ bool word_wait(unsigned* w, unsigned target, int* s_abort) {
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - target) >= 0) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { *s_abort = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    asm volatile("" ::: "memory");
    __syncthreads();
    return *s_abort == 0;
}

// ---- GEMM pieces ----
// Every operand address is (descriptor, per-lane byte offset lane * 16 in ONE VGPR, wave-uniform byte offset in an SGPR):
// the address arithmetic stays on the scalar unit and hipcc has no loop-invariant pointer VGPRs to keep (or spill).
__device__ __forceinline__ f32x4 ldw(rsrc_t r, unsigned l16, unsigned soff) {       // weights: read-only, cached
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)l16, (int)soff, 0));
}
__device__ __forceinline__ f32x4 ldx(rsrc_t r, unsigned l16, unsigned soff) {       // exchange slab: sc1
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)l16, (int)soff, 16));
}

// this wave's NKB weight k-blocks starting at byte offset soff of a packed slab
template <int NKB>
__device__ __forceinline__ void wload(f32x4 (&b)[NKB], rsrc_t w, unsigned soff, unsigned l16) {
#pragma unroll
    for (int i = 0; i < NKB; ++i) b[i] = ldw(w, l16, soff + (unsigned)i * 1024u);
}

// acc[rb] += X[rows of block rb, NKB k-blocks from byte offset xoff (k-block stride MB KiB)] . b   for all MB row blocks.
// A ring of NSLOT register slots of U k-blocks (16 registers each) keeps ~16 KiB of operand loads in flight per wave;
// the scheduling fences keep hipcc from hoisting every load of a job (and of the next job) to the top, which spills.
template <int MB, int NKB>
__device__ __forceinline__ void seg_mma(f32x4 (&acc)[MB], const f32x4 (&b)[NKB], rsrc_t xs, unsigned xoff, unsigned l16) {
    constexpr int U0 = MB == 4 ? 1 : (MB == 2 ? 2 : 4);
    constexpr int U = U0 < NKB ? U0 : NKB;
    constexpr int NU = NKB / U;
    constexpr int NSLOT = NU < 4 ? NU : 4;
    static_assert(NKB % U == 0, "k-blocks per wave must be a multiple of the slot size");
    f32x4 a[NSLOT][U][MB];
    auto load = [&](int i) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int rb = 0; rb < MB; ++rb)
                a[i % NSLOT][u][rb] = ldx(xs, l16, xoff + (unsigned)(((i * U + u) * MB + rb) * 1024));
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) load(i);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rb = 0; rb < MB; ++rb)
                    acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i % NSLOT][u][rb][j], b[i * U + u][j], acc[rb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (i + NSLOT < NU) load(i + NSLOT);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// one row block only (projection / PreNet tiles): xoff = byte offset of (first k-block, row block), k-block stride MB KiB
template <int NKB>
__device__ __forceinline__ void tile_mma(f32x4& acc, const f32x4 (&b)[NKB], rsrc_t xs, unsigned xoff, int MB, unsigned l16) {
    f32x4 a[NKB];
#pragma unroll
    for (int u = 0; u < NKB; ++u) a[u] = ldx(xs, l16, xoff + (unsigned)(u * MB * 1024));
#pragma unroll
    for (int u = 0; u < NKB; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], b[u][j], acc, 0, 0, 0);
}

// frag16 byte offset of the float4 that holds k = 4*q4 .. 4*q4 + 3 of `row`
__device__ __forceinline__ unsigned frag_off16(int row, int q4, int MB) {
    const int kb = q4 >> 2, kg = q4 & 3;
    return (unsigned)((((kb * MB + (row >> 4)) * 64) + kg * 16 + (row & 15)) * 16);
}

template <int MB>
__global__ __launch_bounds__(NT) void decode_kernel(PArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // Thread coordinates are re-derived from an opaque copy of the thread index at the start of every phase (fresh()):
    // otherwise hipcc hoists every step-invariant address computation of every phase out of the step loop and spills them.
    int tid = threadIdx.x, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int B = a.B, T = a.T, C = a.C, CH = a.CH;
    const int CHP = (CH + 15) & ~15, TP = (T + 3) & ~3;
    // ---- LDS carve ----
    f32x4* red = reinterpret_cast<f32x4*>(smem);                 // [8][MB][64] float4
    float* s_h = smem + NWAVE * MB * 64 * 4;                     // [MB*16][4]
    float* s_cw = s_h + MB * 64;                                 // [32 taps][32 filters]
    float* s_cb = s_cw + 32 * 32;                                // [32]
    float* s_cum = s_cb + 32;                                    // [CHP + 32]
    float* s_f = s_cum + CHP + 32;                               // [CHP][36]
    float* s_qp = s_f + CHP * 36;                                // [16][128]
    float* s_q = s_qp + 16 * UA;                                 // [128]
    float* s_ep = s_q + UA;                                      // [8][CHP]
    float* s_e = s_ep + NWAVE * CHP;                             // [TP]
    float* s_a = s_e + TP;                                       // [TP]
    float* s_cx = s_a + TP;                                      // [2048]
    float* s_rw = s_cx + 2048;                                   // [64]
    float* s_tile = s_rw + 64;                                   // [16][16]
    int* s_flag = reinterpret_cast<int*>(s_tile + 256);          // [0] abort, [1] finished
    float* s_bias = s_tile + 256 + 16;                           // [3][16] gate biases of this workgroup's units

    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(a.xs, 0, (int)a.xs_bytes, 0x00020000);
    unsigned* grp = a.bar + NSHARD * 32;

    // ---- roles ----
    const bool isAtt = wg < B * C;
    const int ab = isAtt ? wg / C : 0, ac = isAtt ? wg - ab * C : 0;
    const int jP = NWG - 1 - wg;                                  // projection tile job (27 column tiles x MB row blocks)
    const bool isP = jP < 27 * MB;
    const int pt27 = jP / MB, prb = jP - pt27 * MB;
    const int ptile = pt27 < 11 ? pt27 : pt27 + 1;                // tile 11 (columns 176..191) is padding
    const int jN = NWG - 1 - 27 * MB - wg;                        // last-PreNet tile job (16 column tiles x MB row blocks)
    const bool isN = jN >= 0 && jN < 16 * MB;
    const int ntile = isN ? jN / MB : 0, nrb = isN ? jN - ntile * MB : 0;

    // ---- one-time set-up ----
    if (tid == 0) { s_flag[0] = 0; s_flag[1] = 0; }
    for (int i = tid; i < 32 * 32; i += NT) {
        const int j = i >> 5, f = i & 31;
        s_cw[i] = (j < a.KA && f < a.F) ? a.conv_w[j * a.F + f] : 0.f;
    }
    if (tid < 32) s_cb[tid] = tid < a.F ? a.conv_b[tid] : 0.f;
    // location layer as MFMA B fragments of this wave's 16 attention units; step s uses k = 16 (s >> 2) + 4 kg + (s & 3)
    int kg = lane >> 4, c16 = lane & 15;
    float wlB[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = 16 * (s >> 2) + 4 * kg + (s & 3);
        wlB[s] = k < a.F ? a.wloc[(size_t)k * UA + wave * 16 + c16] : 0.f;
    }
    const float va_l = a.v_a[wave * 16 + c16];
    // per-thread LSTM state (threads < MB * 64: row = tid >> 2, unit-in-workgroup = tid & 3)
    int erow = tid >> 2, euu = tid & 3;
    bool eown = tid < MB * 64 && erow < B;
    float cA = 0.f, hA = 0.f, c1 = 0.f, h1 = 0.f, c2 = 0.f, h2 = 0.f;
    if (tid < 48) s_bias[tid] = (tid < 16 ? a.bA : tid < 32 ? a.b1 : a.b2)[(size_t)wg * 16 + (tid & 15)];
    const rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wA), 0, NWG * (KBX + KBE + KBH) * 1024, 0x00020000);
    const rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1), 0, NWG * (KBH + KBE + KBH) * 1024, 0x00020000);
    const rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2), 0, NWG * (KBH + KBH) * 1024, 0x00020000);
    const rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wP), 0, 28 * KBH * 1024, 0x00020000);
    const rsrc_t rN = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wN), 0, 16 * KBX * 1024, 0x00020000);
    const unsigned slabA = (unsigned)wg * (KBX + KBE + KBH) * 1024u;      // byte offsets of this workgroup's slabs
    const unsigned slab1 = (unsigned)wg * (KBH + KBE + KBH) * 1024u;
    const unsigned slab2 = (unsigned)wg * (KBH + KBH) * 1024u;
    unsigned l16 = (unsigned)lane * 16u;
    auto fresh = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        kg = lane >> 4; c16 = lane & 15; erow = tid >> 2; euu = tid & 3;
        eown = tid < MB * 64 && erow < B; l16 = (unsigned)lane * 16u;
    };
    const int len_b = isAtt ? a.lens[ab] : 0;

    f32x4 accA[MB], acc1[MB], acc2[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) { accA[rb] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[rb] = accA[rb]; acc2[rb] = accA[rb]; }
    __syncthreads();

    // gate sums of this thread's (row, unit) from the waves' partial tiles; D layout: col = lane & 15, row = 4 (lane >> 4) + reg
    auto gate4 = [&](float (&z)[4]) {
        const int rb = erow >> 4, rr = erow & 15;
        const int reg = rr & 3, lbase = (rr >> 2) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) s += red[(w * MB + rb) * 64 + lbase + g * 4 + euu][reg];
            z[g] = s;
        }
    };
    auto put_red = [&](f32x4 (&acc)[MB]) {
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    };
    // LSTMCell + inference zoneout (modules.py:261-283): returns raw output, updates (c, h) state
    auto cell = [&](const float (&z)[4], const float* bv, float& c, float& h) -> float {
        const float zi = z[0] + bv[euu], zj = z[1] + bv[4 + euu], zf = z[2] + bv[8 + euu], zo = z[3] + bv[12 + euu];
        const float cn = sigmoidf_(zf + a.fb) * c + sigmoidf_(zi) * tanhf(zj);
        const float hn = sigmoidf_(zo) * tanhf(cn);
        c = (1.0f - a.zc) * cn + a.zc * c;
        h = (1.0f - a.zh) * hn + a.zh * h;
        return hn;
    };
    // the four units of a row sit in four adjacent lanes: lane uu == 0 stores them as one float4 of the frag16 operand
    auto publish4 = [&](float v, unsigned off_buf) {
        f32x4 q;
        q[0] = __shfl(v, (lane & ~3) + 0); q[1] = __shfl(v, (lane & ~3) + 1);
        q[2] = __shfl(v, (lane & ~3) + 2); q[3] = __shfl(v, (lane & ~3) + 3);
        if (eown && euu == 0) st16(xs, off_buf + frag_off16(erow, wg, MB), q);
    };

    unsigned epoch = 0;        // hops completed so far (identical in every workgroup)
    int t = 0, final_steps = 0;
    bool alive = true;
    for (; t < a.max_steps; ++t) {
        fresh();
        // ================= P6: last PreNet layer  x2 = relu(x1 . Wn + bn)   (modules.py:88-91) =================
        if (isN) {
            f32x4 b[2];
            wload<2>(b, rN, (unsigned)((ntile * KBX + wave * 2) * 1024), l16);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma<2>(acc, b, xs, a.oX1 + (unsigned)((wave * 2 * MB + nrb) * 1024), MB, l16);
            red[wave * 64 + lane] = acc;
            __syncthreads();
            if (tid < 256) {
                const int r16 = tid >> 4, c = tid & 15;
                const int sl = (r16 >> 2) * 16 + c, reg = r16 & 3;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) s += red[w * 64 + sl][reg];
                s_tile[r16 * 16 + c] = fmaxf(s + a.bN[ntile * 16 + c], 0.f);
            }
            __syncthreads();
            if (tid < 64) {
                const int r16 = tid >> 2, q = tid & 3, row = nrb * 16 + r16;
                if (row < B) st16(xs, a.oX2 + frag_off16(row, ntile * 4 + q, MB), *reinterpret_cast<const f32x4*>(s_tile + r16 * 16 + q * 4));
            }
        }
        hop_arrive(a.bar); ++epoch; fresh();
        // critical weights of P1 (the PreNet rows of the attention LSTM kernel) while the hop completes
        f32x4 bx[2];
        wload<2>(bx, rA, slabA + (unsigned)(wave * 2 * 1024), l16);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh();

        // ================= P1: attention LSTM (AttentionWrapper cell call, App. A6)  =================
        seg_mma<MB, 2>(accA, bx, xs, a.oX2 + (unsigned)(wave * 2 * MB * 1024), l16);
        put_red(accA);
        __syncthreads();
        {
            float hn = 0.f;
            if (tid < MB * 64) {
                float z[4];
                gate4(z);
                hn = cell(z, s_bias, cA, hA);
                publish4(hn, a.oHAr);
                publish4(hA, a.oHA);
                s_h[tid] = eown ? hn : 0.f;
            }
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) accA[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            // this workgroup's share of the processed query (tacotron_v2.py:122): qpart[wg][row][u] = sum_uu h[row][uu] wq[4 wg + uu][u]
            const int u4 = tid & 31, r16 = tid >> 5;
            f32x4 wqv[4];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) wqv[uu] = *reinterpret_cast<const f32x4*>(a.wq + (size_t)(wg * 4 + uu) * UA + u4 * 4);
#pragma unroll
            for (int ps = 0; ps < MB; ++ps) {
                const int row = ps * 16 + r16;
                if (row < B) {
                    f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu) q += s_h[row * 4 + uu] * wqv[uu];
                    st16(xs, a.oQP + (unsigned)((((wg * MB * 16) + row) * UA + u4 * 4) * 4), q);
                }
            }
        }
        hop_arrive(a.bar); ++epoch; fresh();
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh();

        // ================= P2: location-sensitive attention (tacotron_v2.py:118-141) =================
        if (isAtt) {
            const int p0 = ac * CH;
            const int n = max(0, min(CH, T - p0));
            const int pl = (a.KA - 1) >> 1;
            // (a) processed query: 256 partial rows of 128, 16 in flight per thread
            {
                const int u4 = tid & 31, pg = tid >> 5;
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                f32x4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    v[i] = ld16(xs, a.oQP + (unsigned)(((((pg + 16 * i) * MB * 16) + ab) * UA + u4 * 4) * 4));
#pragma unroll
                for (int i = 0; i < 16; ++i) s += v[i];
                *reinterpret_cast<f32x4*>(s_qp + pg * UA + u4 * 4) = s;
            }
            // (b) previous (cumulative) alignments of this chunk with the conv halo
            {
                const unsigned cum_prev = a.oCUM + (unsigned)(((t + 1) & 1) * B * T * 4);
                for (int i = tid; i < CHP + 32; i += NT) {
                    const int p = p0 - pl + i;
                    float v = 0.f;
                    if (t > 0 && p >= 0 && p < T && i < n + a.KA - 1) v = ld4(xs, cum_prev + (unsigned)((ab * T + p) * 4));
                    s_cum[i] = v;
                }
            }
            // keys of this lane's (position, unit) pairs for the first four row tiles
            float kreg[4][4];
            const int NRT = (n + 15) >> 4;
            const float* keys_b = a.keys + (size_t)ab * T * UA + wave * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p = min(p0 + 16 * r + 4 * kg + g, T - 1);
                    kreg[r][g] = r < NRT ? keys_b[(size_t)p * UA] : 0.f;
                }
            __syncthreads();
            if (tid < UA) {
                float q = a.b_a[tid];
#pragma unroll
                for (int g = 0; g < 16; ++g) q += s_qp[g * UA + tid];
                s_q[tid] = q;
            }
            // (c) location features f[p][c] = sum_j w[j][c] cum[p + j - pad_l] + b[c]   (:128-130)
            for (int item = tid; item < CHP * 32; item += NT) {
                const int pp = item >> 5, f = item & 31;
                float acc = 0.f;
                if (pp < n) {
                    acc = s_cb[f];
                    for (int j = 0; j < a.KA; ++j) acc = fmaf(s_cw[j * 32 + f], s_cum[pp + j], acc);
                }
                s_f[pp * 36 + f] = acc;
            }
            __syncthreads();
            // (d) processed location features on the matrix pipe + energies (:131-133, :66-78)
            const float q_l = s_q[wave * 16 + c16];
            auto energy_tile = [&](int r, const float (&kk)[4]) {
                f32x4 pf = {0.f, 0.f, 0.f, 0.f};
                const float* fr = s_f + (16 * r + c16) * 36 + 4 * kg;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(fr), a1 = *reinterpret_cast<const f32x4*>(fr + 16);
#pragma unroll
                for (int s = 0; s < 4; ++s) pf = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], wlB[s], pf, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) pf = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], wlB[4 + s], pf, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float e = va_l * fast_tanhf(kk[g] + q_l + pf[g]);
                    e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4); e += __shfl_xor(e, 8);
                    if (c16 == 0) s_ep[wave * CHP + 16 * r + 4 * kg + g] = e;
                }
            };
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < NRT) energy_tile(r, kreg[r]);
            for (int r = 4; r < NRT; ++r) {          // only for chunks of more than 64 positions
                float kk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) kk[g] = keys_b[(size_t)min(p0 + 16 * r + 4 * kg + g, T - 1) * UA];
                energy_tile(r, kk);
            }
            __syncthreads();
            if (tid < n) {
                float e = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) e += s_ep[w * CHP + tid];
                st4(xs, a.oEN + (unsigned)((ab * T + p0 + tid) * 4), e);
            }
            // (e) group hop: the C workgroups of this row exchange their energies
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(grp + ab * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // values of the context slice: in flight during the group hop, consumed after the softmax
            const int ES = EW / C, nc4 = ES >> 2, NPG = NT / nc4;
            const int c4 = tid % nc4, pg = tid / nc4;
            const f32x4* vp = reinterpret_cast<const f32x4*>(a.values + (size_t)ab * T * EW + ac * ES) + c4;
            constexpr int VN = 10;
            f32x4 vv[VN];
#pragma unroll
            for (int i = 0; i < VN; ++i) vv[i] = vp[(size_t)min(pg + i * NPG, len_b - 1) * (EW / 4)];
            if (!word_wait(grp + ab * 32, (unsigned)C * (unsigned)(t + 1), s_flag)) { alive = false; break; }
            for (int i = tid; i < T; i += NT) s_e[i] = ld4(xs, a.oEN + (unsigned)((ab * T + i) * 4));
            __syncthreads();
            // (f) masked softmax over positions < len (score mask -inf -> probability exactly 0; App. A5)
            float m = -INFINITY;
            for (int i = tid; i < len_b; i += NT) m = fmaxf(m, s_e[i]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if (lane == 0) s_rw[wave] = m;
            __syncthreads();
            m = s_rw[0];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) m = fmaxf(m, s_rw[w]);
            float sum = 0.f;
            for (int i = tid; i < T; i += NT) {
                const float ex = (i < len_b) ? expf(s_e[i] - m) : 0.f;
                s_a[i] = ex;
                sum += ex;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            if (lane == 0) s_rw[32 + wave] = sum;
            __syncthreads();
            sum = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) sum += s_rw[32 + w];
            for (int i = tid; i < T; i += NT) s_a[i] = s_a[i] / sum;
            __syncthreads();
            // (g) this chunk's alignments: history row and the next attention state (:137-140)
            if (tid < n) {
                const float p = s_a[p0 + tid];
                a.hist[((size_t)t * B + ab) * T + p0 + tid] = p;
                st4(xs, a.oCUM + (unsigned)((t & 1) * B * T * 4) + (unsigned)((ab * T + p0 + tid) * 4), a.cumulative ? p + s_cum[tid + pl] : p);
            }
            // (h) context slice = alignments . values[:, slice]   (AttentionWrapper._compute_attention)
            {
                f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < VN; ++i) {
                    const int p = pg + i * NPG;
                    acc4 += (p < len_b ? s_a[p] : 0.f) * vv[i];
                }
                for (int p = pg + VN * NPG; p < len_b; p += NPG) acc4 += s_a[p] * vp[(size_t)p * (EW / 4)];
                *reinterpret_cast<f32x4*>(s_cx + pg * ES + c4 * 4) = acc4;
            }
            __syncthreads();
            if (tid < nc4) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                for (int g = 0; g < NPG; ++g) s += *reinterpret_cast<const f32x4*>(s_cx + g * ES + tid * 4);
                st16(xs, a.oCTX + frag_off16(ab, (ac * ES >> 2) + tid, MB), s);
            }
        }
        hop_arrive(a.bar); ++epoch; fresh();
        // gap: attention cell output into LSTM 1 (rnn_wrappers.py:76-78 concat, first K segment)
        {
            f32x4 b[8];
            wload<8>(b, r1, slab1 + (unsigned)(wave * 8 * 1024), l16);
            seg_mma<MB, 8>(acc1, b, xs, a.oHAr + (unsigned)(wave * 8 * MB * 1024), l16);
        }
        f32x4 bc1[4];
        wload<4>(bc1, r1, slab1 + (unsigned)((KBH + wave * 4) * 1024), l16);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh();

        // ================= P3: decoder LSTM 1 (tacotron_v2.py:155-161) =================
        seg_mma<MB, 4>(acc1, bc1, xs, a.oCTX + (unsigned)(wave * 4 * MB * 1024), l16);
        put_red(acc1);
        __syncthreads();
        if (tid < MB * 64) {
            float z[4];
            gate4(z);
            const float hn = cell(z, s_bias + 16, c1, h1);
            publish4(hn, a.oY1);
            publish4(h1, a.oH1);
        }
        hop_arrive(a.bar); ++epoch; fresh();
        // gap: the attention LSTM's recurrent K ranges for the NEXT step: [ctx(t) | h_att(t)]
        {
            f32x4 b[8];
            wload<8>(b, rA, slabA + (unsigned)((KBX + KBE + wave * 8) * 1024), l16);
            seg_mma<MB, 8>(accA, b, xs, a.oHA + (unsigned)(wave * 8 * MB * 1024), l16);
            f32x4 bc[4];
            wload<4>(bc, rA, slabA + (unsigned)((KBX + wave * 4) * 1024), l16);
            seg_mma<MB, 4>(accA, bc, xs, a.oCTX + (unsigned)(wave * 4 * MB * 1024), l16);
        }
        f32x4 by[8];
        wload<8>(by, r2, slab2 + (unsigned)(wave * 8 * 1024), l16);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh();

        // ================= P4: decoder LSTM 2 =================
        seg_mma<MB, 8>(acc2, by, xs, a.oY1 + (unsigned)(wave * 8 * MB * 1024), l16);
        put_red(acc2);
        __syncthreads();
        if (tid < MB * 64) {
            float z[4];
            gate4(z);
            const float hn = cell(z, s_bias + 32, c2, h2);
            publish4(hn, a.oY2);
            publish4(h2, a.oH2);
        }
        hop_arrive(a.bar); ++epoch; fresh();
        // gap: LSTM 1's recurrent K range for the next step
        {
            f32x4 b[8];
            wload<8>(b, r1, slab1 + (unsigned)((KBH + KBE + wave * 8) * 1024), l16);
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) acc1[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            seg_mma<MB, 8>(acc1, b, xs, a.oH1 + (unsigned)(wave * 8 * MB * 1024), l16);
        }
        f32x4 bp[8];
        if (isP) wload<8>(bp, rP, (unsigned)((ptile * KBH + wave * 8) * 1024), l16);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh();

        // ================= P5: mel / stop projections (rnn_wrappers.py:105-109) + composed first PreNet layer of step t+1 =================
        if (isP) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            tile_mma<8>(acc, bp, xs, a.oY2 + (unsigned)((wave * 8 * MB + prb) * 1024), MB, l16);
            red[wave * 64 + lane] = acc;
            __syncthreads();
            if (tid < 256) {
                const int r16 = tid >> 4, c = tid & 15;
                const int sl = (r16 >> 2) * 16 + c, reg = r16 & 3;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) s += red[w * 64 + sl][reg];
                const int col = ptile * 16 + c, row = prb * 16 + r16;
                s += a.bP[col];
                if (ptile >= 12) s_tile[r16 * 16 + c] = fmaxf(s, 0.f);
                else if (row < B) {
                    if (col < a.MR) a.mel[(size_t)row * a.mel_ld + (size_t)t * a.mel_ts + col] = s;
                    else if (col == a.MR) {
                        a.stop[(size_t)row * a.stop_ld + t] = s;
                        st4(xs, a.oSTOP + (unsigned)(row * 4), s);
                    }
                }
            }
            __syncthreads();
            if (ptile >= 12 && tid < 64) {
                const int r16 = tid >> 2, q = tid & 3, row = prb * 16 + r16;
                if (row < B) st16(xs, a.oX1 + frag_off16(row, (ptile - 12) * 4 + q, MB), *reinterpret_cast<const f32x4*>(s_tile + r16 * 16 + q * 4));
            }
        }
        hop_arrive(a.bar); ++epoch; fresh();
        // gap: LSTM 2's recurrent K range for the next step
        {
            f32x4 b[8];
            wload<8>(b, r2, slab2 + (unsigned)((KBH + wave * 8) * 1024), l16);
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) acc2[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            seg_mma<MB, 8>(acc2, b, xs, a.oH2 + (unsigned)(wave * 8 * MB * 1024), l16);
        }
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh();
        // ---- stop rule of StopTokenBasedInferenceHelper (helpers.py:90-94): every row says stop and time > min_iters;
        //      the finishing step's outputs are kept, so the decode emits t + 1 steps.  Every workgroup evaluates it on the
        //      same published logits, so all of them leave the loop at the same step.
        if (tid < 64) {
            const float v = lane < B ? ld4(xs, a.oSTOP + (unsigned)(lane * 4)) : 1.0f;
            const bool all_stop = __all(v > 0.f);
            if (lane == 0) s_flag[1] = (a.stop_en && all_stop && t > a.min_iters) ? 1 : 0;
        }
        __syncthreads();
        if (s_flag[1]) { final_steps = t + 1; break; }
        __syncthreads();
    }
    if (wg == 0 && tid == 0) {
        a.ctl[CTL_FINAL] = final_steps;
        if (!alive) a.ctl[5] = 1;
    }
    if (!alive && tid == 0) atomicOr(reinterpret_cast<unsigned*>(a.ctl) + 6, 1u);
}

// x1 of step 0: the go frame is zeros (helpers.py:187-188), so the first PreNet layer's output is relu(bias)
__global__ void init_x1_kernel(float* x1, const float* __restrict__ bias, int B, int MB, int P) {
    const int n = B * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int row = i / P, k = i - row * P;
        x1[frag_idx(row, k, MB)] = fmaxf(bias[k], 0.f);
    }
}

}   // namespace pz
