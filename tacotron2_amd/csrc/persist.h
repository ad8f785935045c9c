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
// over the 8 waves.  A decoder step is five all-to-all exchanges ("hops") and three group hops:
//   P6 last PreNet layer -> H6 (+ stop rule) -> P1 attention LSTM -> H1 -> P2a query tiles -> granules -> P2 attention
//   (energy granules inside) -> H2 -> P3 LSTM 1 -> H3 -> P4 LSTM 2 -> H4 -> P5 mel / stop projection + first PreNet layer
//   of the next step -> row-block counter -> P6.
// Teacher-forced decodes (TrainingHelper / ValidationHelper(teacher_forcing=True), helpers.py:131-136,176-184) feed nothing
// back: the PreNet outputs of all steps are computed beforehand (teacher_prenets), the projections afterwards
// (teacher_projection), and a step is P1 -> H1 -> P2 -> H2 -> P3 -> H3 -> P4 -> H4.
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
#ifndef PZ_POLL_SLEEP
#define PZ_POLL_SLEEP 1      // s_sleep between polls of a hop counter (0 measured no faster: tools A/B)
#endif
constexpr unsigned SPIN_LIMIT_TICKS = 25u * 1000u * 1000u;   // s_memrealtime ticks (100 MHz): 250 ms per wait

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// Training forward (csrc/train_step.h): sampled zoneout (modules.py:274-276) from the counter-based generator and the tape
// the backward pass reads, written by the workgroups that own the values.  Layout as TrainTape (plain [S][rows][.]).
struct PTape {
    float *z[3], *cprev[3], *cnew[3], *hraw[3], *hstate[3];    // attention LSTM, LSTM 1, LSTM 2
    float *q, *state_prev, *ctx;
    unsigned seed;
    int on;
};

struct PArgs {
    const float *wA, *bA, *w1, *b1, *w2, *b2;   // packed LSTM slabs [256][kbtot][64][4] (pack_skinny), biases [256][16]
    const float *wP, *bP;                       // projection + composed first PreNet layer: [28][64][64][4], bias [448]
    const float *wN, *bN;                       // last PreNet layer: [16][16][64][4], bias [256]
    const float *wQ;                            // query layer packed like a dense layer: [8 column tiles][64][64][4]
    const float *conv_w, *conv_b, *wloc, *v_a, *b_a;
    const float *keys, *values;                 // [B][T][128], [B][T][512]
    const int* lens;
    float* xs;                                  // exchange slab (zeroed before the launch)
    unsigned xs_bytes;
    unsigned oX1, oX2, oHAr, oHA, oCTX, oY1, oH1, oY2, oH2, oQP, oEN, oCUM, oSTOP;   // byte offsets into xs
    unsigned* bar;                              // [NSHARD][32] hop counters, [64][32] attention row groups, [4][32] + [4][32] row-block groups (zeroed)
    int* ctl;
    float *mel, *stop, *hist;
    float *cA_out, *c1_out, *c2_out;            // cell states [rows][1024] stored when the loop ends (final_state of dynamic_decode)
    long long mel_ld, mel_ts;
    int stop_ld;
    int B, T, C, CH, F, KA, MR, max_steps, min_iters, stop_en, cumulative;
    float zc, zh, fb;
};

// What only the teacher-forced modes read: a second kernel argument, so that the free-running instantiation's argument block
// (and its scalar-register budget) stays what it was.
struct PExtra {
    const float* x2_all;                        // last PreNet layer's outputs of every step, frag16 per step
    unsigned x2_bytes, x2_stride;               // size of x2_all, bytes per step
    float* y_all;                               // LSTM 2's raw outputs [S][rows][1024] for the deferred projection (null: the tape has them)
    unsigned oB1[4];                            // two-chain schedule (MODE 3 / 4): second bank of the h_att raw, h_att, h1, h2 slots (odd steps)
    PTape tp;
};

__host__ __device__ inline int lds_floats(int MB, int T, int CH) {
    const int CHP = (CH + 15) & ~15, TP = (T + 3) & ~3;
    const int red = NWAVE * MB * 64 * 4;
    return red + MB * 64 + 32 * 32 + 32 + (CHP + 48) + CHP * 36 + 16 * UA + UA + NWAVE * CHP + (NWAVE + 1) * TP + 2048 + 64 + 16 * 16 + 16 + 64;
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
// peek: issue the shard loads early (in the middle of gap work) so that their latency is hidden; the value goes to hop_wait
__device__ __forceinline__ unsigned hop_peek(unsigned* bar) {
    unsigned v = 0u;
    if (threadIdx.x < NSHARD) v = __hip_atomic_load(bar + threadIdx.x * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
// wait: wave 0 polls the NSHARD shards (one lane each) until each has seen `per_shard` arrivals; the first check uses the
// peeked values.  Returns false on timeout or when another wait of this workgroup already failed (s_abort).
__device__ __forceinline__ bool hop_wait(unsigned* bar, unsigned per_shard, int* s_abort, unsigned peek = 0u) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool bad = false;
        unsigned v = lane < NSHARD ? peek : per_shard;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            if (__all((int)(v - per_shard) >= 0)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { bad = true; break; }
#if PZ_POLL_SLEEP
            __builtin_amdgcn_s_sleep(1);
#endif
            if (lane < NSHARD) v = __hip_atomic_load(bar + lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
#ifdef PZ_NO_WEIGHT_STREAM   // diagnostic builds only (tools/build_noweights.sh): every weight load reads the slab's first KiB, which
    soff = 0u;               // stays in L1 -- the step time without any weight traffic = the most on-chip weight residency could give
#endif
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

// Diagnostic builds only (-DPZ_STAMPS, tools/persist_stamps.py): thread 0 of every workgroup sums the wall-clock time
// (s_memrealtime, 10 ns ticks) between consecutive stamp points over all steps; the product build has no stamps.
#ifdef PZ_STAMPS
__device__ unsigned long long g_pz_stamps[NWG][24];
#define PZ_STAMP(i) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); st_acc[i] += now_ - st_prev; st_prev = now_; } } while (0)
#else
#define PZ_STAMP(i) do { } while (0)
#endif

// MODE 0: free-running decode (feeds its own output back, stop rule); 1: teacher-forced decode; 2: teacher-forced training
// forward (sampled zoneout + tape).  A template parameter, not a runtime switch: the free-running instantiation carries none
// of the other modes' code (a runtime switch cost it 2.4 us per step at B = 1).
template <int MB, int MODE>
__global__ __launch_bounds__(NT) void decode_kernel(PArgs a, PExtra x) {
    constexpr bool TEACH = MODE >= 1, TAPE = MODE == 2 || MODE == 4, TWO = MODE >= 3;
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
    float* s_f = s_cum + CHP + 48;                               // [CHP][36]
    float* s_qp = s_f + CHP * 36;                                // [16][128]
    float* s_q = s_qp + 16 * UA;                                 // [128]
    float* s_ep = s_q + UA;                                      // [8][CHP]
    float* s_e = s_ep + NWAVE * CHP;                             // [8][TP]: every wave's copy of the energies
    float* s_a = s_e + NWAVE * TP;                               // [TP]
    float* s_cx = s_a + TP;                                      // [2048]
    float* s_rw = s_cx + 2048;                                   // [64]
    float* s_tile = s_rw + 64;                                   // [16][16]
    int* s_flag = reinterpret_cast<int*>(s_tile + 256);          // [0] abort, [1] finished
    float* s_bias = s_tile + 256 + 16;                           // [3][16] gate biases of this workgroup's units

    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(a.xs, 0, (int)a.xs_bytes, 0x00020000);
    unsigned* grp = a.bar + NSHARD * 32;                          // [64] attention row groups
    unsigned* grpx = grp + 64 * 32;                                // [4] row-block groups: projection tiles -> last PreNet layer

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

    constexpr int KSQ = 8;                                        // K splits of the query layer product
    const bool isQ = wg < 8 * KSQ * MB;                           // query tile job: column tile, K split, row block
    const int qct = wg & 7, qks = (wg >> 3) & (KSQ - 1), qrb = wg >> 6;
    unsigned long long* qgr = reinterpret_cast<unsigned long long*>(a.xs + a.oQP / 4);     // [8 K splits][rows][128] query partial granules
    const rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wQ), 0, 8 * KBH * 1024, 0x00020000);

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
    const rsrc_t rX2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(TEACH ? x.x2_all : a.wN), 0, TEACH ? (int)x.x2_bytes : 0, 0x00020000);
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

    int t = 0;                 // decoder step
    int tc = 0;                // the step a cell call belongs to (tape rows, mask counters): t, or the chain's own step (TWO)
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
    auto cell = [&](const float (&z)[4], const float* bv, float& c, float& h, int layer) -> float {
        const float zi = z[0] + bv[euu], zj = z[1] + bv[4 + euu], zf = z[2] + bv[8 + euu], zo = z[3] + bv[12 + euu];
        const float cn = sigmoidf_(zf + a.fb) * c + sigmoidf_(zi) * tanhf(zj);
        const float hn = sigmoidf_(zo) * tanhf(cn);
        if constexpr (!TAPE) {
            c = (1.0f - a.zc) * cn + a.zc * c;
            h = (1.0f - a.zh) * hn + a.zh * h;
        } else if (eown) {
            const int unit = wg * 4 + euu, H = NWG * 4;
            const unsigned stream = (unsigned)(T2_STREAM_ZONE_ATT + 2 * layer);
            const float cp = c, hp = h;
            c = (1.0f - a.zc) * dropout_train(cn - cp, a.zc, t2_u01(x.tp.seed, stream, (unsigned)tc, (unsigned)erow, (unsigned)unit)) + cp;
            h = (1.0f - a.zh) * dropout_train(hn - hp, a.zh, t2_u01(x.tp.seed, stream + 1u, (unsigned)tc, (unsigned)erow, (unsigned)unit)) + hp;
            const size_t ti = (size_t)tc * B + erow;
            float* tz = x.tp.z[layer] + ti * 4 * H;
            tz[unit] = zi; tz[H + unit] = zj; tz[2 * H + unit] = zf; tz[3 * H + unit] = zo;
            x.tp.cprev[layer][ti * H + unit] = cp; x.tp.cnew[layer][ti * H + unit] = cn;
            if (x.tp.hraw[layer] != nullptr) x.tp.hraw[layer][ti * H + unit] = hn;
            x.tp.hstate[layer][ti * H + unit] = h;
        }
        return hn;
    };
    // the four units of a row sit in four adjacent lanes: lane uu == 0 stores them as one float4 of the frag16 operand
    auto publish4 = [&](float v, unsigned off_buf) {
        f32x4 q;
        q[0] = __shfl(v, (lane & ~3) + 0); q[1] = __shfl(v, (lane & ~3) + 1);
        q[2] = __shfl(v, (lane & ~3) + 2); q[3] = __shfl(v, (lane & ~3) + 3);
        if (eown && euu == 0) st16(xs, off_buf + frag_off16(erow, wg, MB), q);
    };

#ifdef PZ_STAMPS
    unsigned long long st_acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) st_acc[i] = 0;
    unsigned long long st_prev = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned epoch = 0;        // hops completed so far (identical in every workgroup)
    int final_steps = 0;
    bool alive = true;
    // attention pre-work of the NEXT step, done in a gap: everything of the energies that does not need the query.
    // pre[r][g] = keys[p][u] + (f . W_loc)[p][u] for this lane's (position 16 r + 4 kg + g of the chunk, unit 16 wave + c16)
    float pre[4][4];
    const int att_p0 = ac * CH;
    const int att_n = isAtt ? max(0, min(CH, T - att_p0)) : 0;
    const int att_pl = (a.KA - 1) >> 1;
    auto attention_pre = [&](int tt) {
        // previous (cumulative) alignments of this chunk with the conv halo: s_cum[i] <-> position p0 - pad_l + i
        const unsigned cum_prev = a.oCUM + (unsigned)(((tt + 1) & 1) * B * T * 4);
        for (int i = tid; i < CHP + 48; i += NT) {
            const int p = att_p0 - att_pl + i;
            float v = 0.f;
            if (tt > 0 && p >= 0 && p < T && i < att_n + a.KA - 1) v = ld4(xs, cum_prev + (unsigned)((ab * T + p) * 4));
            s_cum[i] = v;
        }
        const int NRT = (att_n + 15) >> 4;
        const float* keys_b = a.keys + (size_t)ab * T * UA + wave * 16 + c16;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int p = min(att_p0 + 16 * r + 4 * kg + g, T - 1);
                pre[r][g] = r < NRT ? keys_b[(size_t)p * UA] : 0.f;
            }
        __syncthreads();
        // location features f[p][c] = sum_j w[j][c] cum[p + j - pad_l] + b[c]   (tacotron_v2.py:128-130): one thread = one
        // filter x four positions, the 32 (zero-padded) taps unrolled over a register window
        {
            const int f = tid & 31, pq = tid >> 5;
            if (4 * pq < CHP) {
                float win[36];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_cum + 4 * pq + 4 * i);
                    win[4 * i] = w4[0]; win[4 * i + 1] = w4[1]; win[4 * i + 2] = w4[2]; win[4 * i + 3] = w4[3];
                }
                const float cb = s_cb[f];
                float acc[4] = {cb, cb, cb, cb};
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const float wj = s_cw[j * 32 + f];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(wj, win[i + j], acc[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s_f[(4 * pq + i) * 36 + f] = (4 * pq + i < att_n) ? acc[i] : 0.f;
            }
        }
        __syncthreads();
        // processed location features on the matrix pipe (tacotron_v2.py:131): [16 positions x 32 filters] . [32 x 16 units]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < NRT) {
                f32x4 pf = {0.f, 0.f, 0.f, 0.f};
                const float* fr = s_f + (16 * r + c16) * 36 + 4 * kg;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(fr), a1 = *reinterpret_cast<const f32x4*>(fr + 16);
#pragma unroll
                for (int s = 0; s < 4; ++s) pf = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], wlB[s], pf, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) pf = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], wlB[4 + s], pf, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[r][g] += pf[g];
            }
        }
    };
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[r][g] = 0.f;
    if (TEACH && !TWO && isAtt) attention_pre(0);

    // ============================================================================================================================
    // Teacher-forced modes, two-chain schedule (MODE 3 / 4).  With known inputs the attention chain of step i (attention LSTM ->
    // query -> attention) needs only ctx and h_att of step i - 1, and the decoder chain of step i - 1 (LSTM 1 -> LSTM 2) only
    // h_att, ctx of its own step: the two run side by side and a decoder step costs TWO hops instead of four --
    //   X(i): attention LSTM (i)  +  LSTM 1 (i - 1)      critical: ctx(i - 1) against both kernels' context rows
    //   Y(i): query + attention (i)  +  LSTM 2 (i - 1)   critical: y1(i - 1) . W2[0:1024]
    // gap of X's hop: h2(i - 2) . W2[h rows], the query-independent part of attention (i); gap of Y's hop: everything of X(i + 1)
    // that is known (x2, h_att, h_att raw, h1 products).  Slots whose readers sit in the gap right before their next write
    // (h_att raw, h_att, h1, h2) are double-banked by step parity (x.oB1).
    // ============================================================================================================================
    if constexpr (TWO) {
        const int S = a.max_steps;
        float hnA_last = 0.f;
        {   // accA = x2(0) . W_A[x rows]
            f32x4 bx[2];
            wload<2>(bx, rA, slabA + (unsigned)(wave * 2 * 1024), l16);
            seg_mma<MB, 2>(accA, bx, rX2, (unsigned)(wave * 2 * MB * 1024), l16);
        }
        f32x4 bcA[4], bc1[4];
        wload<4>(bcA, rA, slabA + (unsigned)((KBX + wave * 4) * 1024), l16);
        wload<4>(bc1, r1, slab1 + (unsigned)((KBH + wave * 4) * 1024), l16);
        for (int i = 0; i <= S; ++i) {
            fresh();
            const int ta = i, td = i - 1;
            const bool actA = ta < S, actB = td >= 0;
            const unsigned oHAr_a = (ta & 1) ? x.oB1[0] : a.oHAr, oHA_a = (ta & 1) ? x.oB1[1] : a.oHA;
            const unsigned oH1_d = (td & 1) ? x.oB1[2] : a.oH1, oH2_d = (td & 1) ? x.oB1[3] : a.oH2;       // (td = -1: unused)
            // ================= X(i): attention LSTM of step ta, LSTM 1 of step td =================
            if (actA) {
                if (i > 0) seg_mma<MB, 4>(accA, bcA, xs, a.oCTX + (unsigned)(wave * 4 * MB * 1024), l16);
                put_red(accA);
                __syncthreads();
                if (tid < MB * 64) {
                    float z[4];
                    gate4(z);
                    tc = ta;
                    const float hn = cell(z, s_bias, cA, hA, 0);
                    hnA_last = hn;
                    publish4(hn, oHAr_a);
                    publish4(hA, oHA_a);
                }
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) accA[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (actB) {
                seg_mma<MB, 4>(acc1, bc1, xs, a.oCTX + (unsigned)(wave * 4 * MB * 1024), l16);
                if (actA) __syncthreads();                     // the attention LSTM's partial tiles have been read
                put_red(acc1);
                __syncthreads();
                if (tid < MB * 64) {
                    float z[4];
                    gate4(z);
                    tc = td;
                    const float hn = cell(z, s_bias + 16, c1, h1, 1);
                    publish4(hn, a.oY1);
                    publish4(h1, oH1_d);
                }
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) acc1[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            hop_arrive(a.bar); ++epoch; fresh();
            // gap: LSTM 2's recurrent K range (h2 of step td - 1), the query-independent part of attention (ta)
            if (td > 0) {
                const unsigned oH2_p = ((td - 1) & 1) ? x.oB1[3] : a.oH2;
                f32x4 b[8];
                wload<8>(b, r2, slab2 + (unsigned)((KBH + wave * 8) * 1024), l16);
                seg_mma<MB, 8>(acc2, b, xs, oH2_p + (unsigned)(wave * 8 * MB * 1024), l16);
            }
            if (isAtt && actA) { attention_pre(ta); fresh(); }
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (isQ) bq = ldw(rQ, l16, (unsigned)((qct * KBH + qks * 8 + wave) * 1024));
            const unsigned pkx = hop_peek(a.bar);
            f32x4 by[8];
            wload<8>(by, r2, slab2 + (unsigned)(wave * 8 * 1024), l16);
            if (!hop_wait(a.bar, 32u * epoch, s_flag, pkx)) { alive = false; break; }
            fresh();
            // ================= Y(i): query + attention of step ta, LSTM 2 of step td =================
            if (isQ && actA) {
                const f32x4 av = ldx(xs, l16, oHAr_a + (unsigned)(((qks * 8 + wave) * MB + qrb) * 1024));
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[j], acc, 0, 0, 0);
                red[wave * 64 + lane] = acc;
                __syncthreads();
                // partial tile -> tagged 8-byte granules {value, tag = step + 1}: the data is the flag (no drain, no counter)
                if (tid < 256) {
                    const int r16 = tid >> 4, c = tid & 15;
                    const int sl = (r16 >> 2) * 16 + c, reg = r16 & 3;
                    float sacc = 0.f;
    #pragma unroll
                    for (int w = 0; w < NWAVE; ++w) sacc += red[w * 64 + sl][reg];
                    const int row = qrb * 16 + r16;
                    if (row < B)
                        __hip_atomic_store(qgr + ((size_t)(qks * MB * 16 + row) * UA + qct * 16 + c),
                                           ((unsigned long long)(unsigned)(ta + 1) << 32) | __builtin_bit_cast(unsigned, sacc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }

            if (actB) {
                seg_mma<MB, 8>(acc2, by, xs, a.oY1 + (unsigned)(wave * 8 * MB * 1024), l16);
                if (isQ && actA) __syncthreads();              // the query tile's partial tiles have been read
                put_red(acc2);
                __syncthreads();
                if (tid < MB * 64) {
                    float z[4];
                    gate4(z);
                    tc = td;
                    const float hn = cell(z, s_bias + 32, c2, h2, 2);
                    publish4(hn, a.oY2);
                    publish4(h2, oH2_d);
                    if (x.y_all != nullptr && eown) x.y_all[((size_t)td * B + erow) * (NWG * 4) + wg * 4 + euu] = hn;
                }
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) acc2[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (isAtt && actA) {
                const int p0 = att_p0, n = att_n, pl = att_pl;
                const int NRT = (n + 15) >> 4;
                // (a) processed query of this row: bias + the 8 K-split partials of its row block's query tiles (granule sweep by
                //     the two waves that own the 128 units)
                if (tid < UA) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    unsigned long long g[KSQ];
                    for (;;) {
                        bool ok = true;
    #pragma unroll
                        for (int k = 0; k < KSQ; ++k) {
                            g[k] = __hip_atomic_load(qgr + ((size_t)(k * MB * 16 + ab) * UA + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok &= (unsigned)(g[k] >> 32) == (unsigned)(ta + 1);
                        }
                        if (__all(ok)) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { s_flag[0] = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    float q = a.b_a[tid];
    #pragma unroll
                    for (int k = 0; k < KSQ; ++k) q += __builtin_bit_cast(float, (unsigned)g[k]);
                    s_q[tid] = q;
                    if (TAPE && ac == 0) x.tp.q[((size_t)ta * B + ab) * UA + tid] = q;
                }
                __syncthreads();
                if (s_flag[0]) { alive = false; break; }
                // (b) energies e[p] = sum_u v[u] tanh(keys + q + processed location)[p][u]   (:66-78, :133)
                const float q_l = s_q[wave * 16 + c16];
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r < NRT) {
    #pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float e = va_l * fast_tanhf(pre[r][g] + q_l);
                            e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4); e += __shfl_xor(e, 8);
                            if (c16 == 0) s_ep[wave * CHP + 16 * r + 4 * kg + g] = e;
                        }
                    }
                }
                __syncthreads();
                // (c) group exchange of the energies as tagged 8-byte granules {value, tag = step + 1} (the data is the flag:
                //     one aligned 8-byte sc1 store per position; readers re-read until every tag matches)
                unsigned long long* eng = reinterpret_cast<unsigned long long*>(a.xs + a.oEN / 4) + (size_t)ab * T;
                if (tid < n) {
                    float e = 0.f;
    #pragma unroll
                    for (int w = 0; w < NWAVE; ++w) e += s_ep[w * CHP + tid];
                    __hip_atomic_store(eng + p0 + tid, ((unsigned long long)(unsigned)(ta + 1) << 32) | __builtin_bit_cast(unsigned, e), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                // values of the context slice: in flight during the exchange, consumed after the softmax
                const int ES = EW / C, nc4 = ES >> 2, NPG = NT / nc4;
                const int c4 = tid % nc4, pg = tid / nc4;
                const f32x4* vp = reinterpret_cast<const f32x4*>(a.values + (size_t)ab * T * EW + ac * ES) + c4;
                constexpr int VN = 10;
                f32x4 vv[VN];
    #pragma unroll
                for (int i = 0; i < VN; ++i) vv[i] = vp[(size_t)min(pg + i * NPG, len_b - 1) * (EW / 4)];
                // (d) every wave sweeps all T granules and computes the masked softmax redundantly (score mask -inf -> probability
                //     exactly 0; App. A5): no workgroup barrier between the exchange and the alignments
                {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    bool bad = false;
                    float m = -INFINITY;
                    for (int i0 = 0; i0 < T; i0 += 64) {
                        const int i = i0 + lane;
                        float e = -INFINITY;
                        for (;;) {
                            unsigned long long g = (unsigned long long)(unsigned)(ta + 1) << 32;
                            if (i < T) g = __hip_atomic_load(eng + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (__all((unsigned)(g >> 32) == (unsigned)(ta + 1))) { if (i < len_b) e = __builtin_bit_cast(float, (unsigned)g); break; }
                            if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { bad = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (bad) break;
                        if (i < T) s_e[wave * TP + i] = e;
                        m = fmaxf(m, e);
                    }
                    if (bad) s_flag[0] = 1;
    #pragma unroll
                    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                    float sum = 0.f;
                    for (int i = lane; i < T; i += 64) {
                        const float ex = (i < len_b) ? expf(s_e[wave * TP + i] - m) : 0.f;
                        s_e[wave * TP + i] = ex;
                        sum += ex;
                    }
    #pragma unroll
                    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                    // wave w publishes the probabilities of positions i with (i >> 6) % 8 == w ... every wave holds all of them; each
                    // writes the same values, so one copy per position is enough: wave (i / 64) % 8 writes block i / 64
                    for (int i0 = 0; i0 < T; i0 += 64)
                        if (((i0 >> 6) & (NWAVE - 1)) == wave && i0 + lane < T) s_a[i0 + lane] = s_e[wave * TP + i0 + lane] / sum;
                }
                __syncthreads();
                if (s_flag[0]) { alive = false; break; }
                // (e) this chunk's alignments: history row and the next attention state (:137-140)
                if (tid < n) {
                    const float p = s_a[p0 + tid];
                    a.hist[((size_t)ta * B + ab) * T + p0 + tid] = p;
                    if constexpr (TAPE) x.tp.state_prev[((size_t)ta * B + ab) * T + p0 + tid] = s_cum[tid + pl];
                    st4(xs, a.oCUM + (unsigned)((ta & 1) * B * T * 4) + (unsigned)((ab * T + p0 + tid) * 4), a.cumulative ? p + s_cum[tid + pl] : p);
                }
                // (f) context slice = alignments . values[:, slice]   (AttentionWrapper._compute_attention)
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
                    if constexpr (TAPE) *reinterpret_cast<f32x4*>(x.tp.ctx + ((size_t)ta * B + ab) * EW + ac * ES + tid * 4) = s;
                }
            }
            hop_arrive(a.bar); ++epoch; fresh();
            // gap: everything of X(i + 1) that is known
            if (ta + 1 < S) {
                f32x4 bx[2];
                wload<2>(bx, rA, slabA + (unsigned)(wave * 2 * 1024), l16);
                seg_mma<MB, 2>(accA, bx, rX2, (unsigned)(ta + 1) * x.x2_stride + (unsigned)(wave * 2 * MB * 1024), l16);
                f32x4 b[8];
                wload<8>(b, rA, slabA + (unsigned)((KBX + KBE + wave * 8) * 1024), l16);
                seg_mma<MB, 8>(accA, b, xs, oHA_a + (unsigned)(wave * 8 * MB * 1024), l16);
            }
            if (actA) {
                {
                    f32x4 b[8];
                    wload<8>(b, r1, slab1 + (unsigned)(wave * 8 * 1024), l16);
                    seg_mma<MB, 8>(acc1, b, xs, oHAr_a + (unsigned)(wave * 8 * MB * 1024), l16);
                }
                if (ta >= 1) {
                    const unsigned oH1_p = ((ta - 1) & 1) ? x.oB1[2] : a.oH1;
                    f32x4 b[8];
                    wload<8>(b, r1, slab1 + (unsigned)((KBH + KBE + wave * 8) * 1024), l16);
                    seg_mma<MB, 8>(acc1, b, xs, oH1_p + (unsigned)(wave * 8 * MB * 1024), l16);
                }
            }
            const unsigned pky = hop_peek(a.bar);
            wload<4>(bcA, rA, slabA + (unsigned)((KBX + wave * 4) * 1024), l16);
            wload<4>(bc1, r1, slab1 + (unsigned)((KBH + wave * 4) * 1024), l16);
            if (!hop_wait(a.bar, 32u * epoch, s_flag, pky)) { alive = false; break; }
        }
        // final_state reads bank 0: the last step's values once more (they live in registers)
        fresh();
        if (alive && tid < MB * 64) { publish4(hnA_last, a.oHAr); publish4(hA, a.oHA); publish4(h1, a.oH1); publish4(h2, a.oH2); }
    }
    if constexpr (!TWO)
    for (; t < a.max_steps; ++t) {
        fresh();
        tc = t;
        float stop_v = 1.0f;
        if constexpr (!TEACH) {
        // gap work of the P5 -> P6 chain: the query-independent part of this step's attention.  The owners of a last-PreNet
        // tile do it BEFORE their tile (they are waiting for the projection tiles anyway), everybody else after arriving.
        auto gap6 = [&]() {
            if (isAtt) attention_pre(t);
        };
        // ================= P6: last PreNet layer  x2 = relu(x1 . Wn + bn)   (modules.py:88-91) =================
        if (isN) {
            gap6(); fresh();
            f32x4 b[2];
            wload<2>(b, rN, (unsigned)((ntile * KBX + wave * 2) * 1024), l16);
            const float bias_n = a.bN[ntile * 16 + (tid & 15)];
            // x1 of this row block comes from the 16 projection tiles of the previous step (row-block group hop)
            if (t > 0 && !word_wait(grpx + nrb * 32, 16u * (unsigned)t, s_flag)) { alive = false; break; }
            fresh();
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
                s_tile[r16 * 16 + c] = fmaxf(s + bias_n, 0.f);
            }
            __syncthreads();
            if (tid < 64) {
                const int r16 = tid >> 2, q = tid & 3, row = nrb * 16 + r16;
                if (row < B) st16(xs, a.oX2 + frag_off16(row, ntile * 4 + q, MB), *reinterpret_cast<const f32x4*>(s_tile + r16 * 16 + q * 4));
            }
        }
        PZ_STAMP(0); hop_arrive(a.bar); ++epoch; fresh();
        if (!isN) { gap6(); fresh(); }
        }
        // critical weights of P1 (the PreNet rows of the attention LSTM kernel) while the hop completes
        f32x4 bx[2];
        wload<2>(bx, rA, slabA + (unsigned)(wave * 2 * 1024), l16);
        if constexpr (!TEACH) {
        PZ_STAMP(1);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh(); PZ_STAMP(2);
        // ---- stop rule of StopTokenBasedInferenceHelper (helpers.py:90-94) for the step just finished: every row says stop
        //      and time > min_iters; the finishing step's outputs are kept, so the decode emits t steps.  Every workgroup
        //      evaluates it on the same published logits (stored before their owner arrived at this hop), so all of them
        //      leave the loop together.  The logits are fetched here and judged behind P1's product (its barrier carries the
        //      verdict), before the cell commits anything of step t: the load's latency is off the chain.
        if (t > 0 && tid < 64) stop_v = lane < B ? ld4(xs, a.oSTOP + (unsigned)(lane * 4)) : 1.0f;
        }

        // ================= P1: attention LSTM (AttentionWrapper cell call, App. A6)  =================
        if constexpr (!TEACH) seg_mma<MB, 2>(accA, bx, xs, a.oX2 + (unsigned)(wave * 2 * MB * 1024), l16);
        else seg_mma<MB, 2>(accA, bx, rX2, (unsigned)t * x.x2_stride + (unsigned)(wave * 2 * MB * 1024), l16);
        put_red(accA);
        if constexpr (!TEACH) {
            if (t > 0 && tid < 64) {
                const bool all_stop = __all(stop_v > 0.f);
                if (lane == 0) s_flag[1] = (a.stop_en && all_stop && t - 1 > a.min_iters) ? 1 : 0;
            }
        }
        __syncthreads();
        if constexpr (!TEACH) {
            if (s_flag[1]) { final_steps = t; break; }
        }
        if (tid < MB * 64) {
            float z[4];
            gate4(z);
            const float hn = cell(z, s_bias, cA, hA, 0);
            publish4(hn, a.oHAr);
            publish4(hA, a.oHA);
        }
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) accA[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        PZ_STAMP(3); hop_arrive(a.bar); ++epoch; fresh();
        // gap: first half of LSTM 2's recurrent K range (h2 of the previous step)
        if (t > 0) {
            f32x4 b[4];
            wload<4>(b, r2, slab2 + (unsigned)((KBH + wave * 8) * 1024), l16);
            seg_mma<MB, 4>(acc2, b, xs, a.oH2 + (unsigned)(wave * 8 * MB * 1024), l16);
        }
        f32x4 bq = {0.f, 0.f, 0.f, 0.f};
        if (isQ) bq = ldw(rQ, l16, (unsigned)((qct * KBH + qks * 8 + wave) * 1024));
        PZ_STAMP(4);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh(); PZ_STAMP(5);

        // ================= P2a: processed query (tacotron_v2.py:122) as 8 column tiles x 8 K splits per row block =================
        if (isQ) {
            const f32x4 av = ldx(xs, l16, a.oHAr + (unsigned)(((qks * 8 + wave) * MB + qrb) * 1024));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[j], acc, 0, 0, 0);
            red[wave * 64 + lane] = acc;
            __syncthreads();
            // partial tile -> tagged 8-byte granules {value, tag = step + 1}: the data is the flag (no drain, no counter)
            if (tid < 256) {
                const int r16 = tid >> 4, c = tid & 15;
                const int sl = (r16 >> 2) * 16 + c, reg = r16 & 3;
                float sacc = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) sacc += red[w * 64 + sl][reg];
                const int row = qrb * 16 + r16;
                if (row < B)
                    __hip_atomic_store(qgr + ((size_t)(qks * MB * 16 + row) * UA + qct * 16 + c),
                                       ((unsigned long long)(unsigned)(t + 1) << 32) | __builtin_bit_cast(unsigned, sacc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // ================= P2: location-sensitive attention (tacotron_v2.py:118-141) =================
        if (isAtt) {
            const int p0 = att_p0, n = att_n, pl = att_pl;
            const int NRT = (n + 15) >> 4;
            // (a) processed query of this row: bias + the 8 K-split partials of its row block's query tiles (granule sweep by
            //     the two waves that own the 128 units)
            if (tid < UA) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned long long g[KSQ];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < KSQ; ++k) {
                        g[k] = __hip_atomic_load(qgr + ((size_t)(k * MB * 16 + ab) * UA + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (unsigned)(g[k] >> 32) == (unsigned)(t + 1);
                    }
                    if (__all(ok)) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { s_flag[0] = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                float q = a.b_a[tid];
#pragma unroll
                for (int k = 0; k < KSQ; ++k) q += __builtin_bit_cast(float, (unsigned)g[k]);
                s_q[tid] = q;
                if (TAPE && ac == 0) x.tp.q[((size_t)t * B + ab) * UA + tid] = q;
            }
            __syncthreads();
            if (s_flag[0]) { alive = false; break; }
            // (b) energies e[p] = sum_u v[u] tanh(keys + q + processed location)[p][u]   (:66-78, :133)
            const float q_l = s_q[wave * 16 + c16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < NRT) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float e = va_l * fast_tanhf(pre[r][g] + q_l);
                        e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4); e += __shfl_xor(e, 8);
                        if (c16 == 0) s_ep[wave * CHP + 16 * r + 4 * kg + g] = e;
                    }
                }
            }
            __syncthreads();
            // (c) group exchange of the energies as tagged 8-byte granules {value, tag = step + 1} (the data is the flag:
            //     one aligned 8-byte sc1 store per position; readers re-read until every tag matches)
            unsigned long long* eng = reinterpret_cast<unsigned long long*>(a.xs + a.oEN / 4) + (size_t)ab * T;
            if (tid < n) {
                float e = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) e += s_ep[w * CHP + tid];
                __hip_atomic_store(eng + p0 + tid, ((unsigned long long)(unsigned)(t + 1) << 32) | __builtin_bit_cast(unsigned, e), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            PZ_STAMP(18);
            // values of the context slice: in flight during the exchange, consumed after the softmax
            const int ES = EW / C, nc4 = ES >> 2, NPG = NT / nc4;
            const int c4 = tid % nc4, pg = tid / nc4;
            const f32x4* vp = reinterpret_cast<const f32x4*>(a.values + (size_t)ab * T * EW + ac * ES) + c4;
            constexpr int VN = 10;
            f32x4 vv[VN];
#pragma unroll
            for (int i = 0; i < VN; ++i) vv[i] = vp[(size_t)min(pg + i * NPG, len_b - 1) * (EW / 4)];
            // (d) every wave sweeps all T granules and computes the masked softmax redundantly (score mask -inf -> probability
            //     exactly 0; App. A5): no workgroup barrier between the exchange and the alignments
            {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool bad = false;
                float m = -INFINITY;
                for (int i0 = 0; i0 < T; i0 += 64) {
                    const int i = i0 + lane;
                    float e = -INFINITY;
                    for (;;) {
                        unsigned long long g = (unsigned long long)(unsigned)(t + 1) << 32;
                        if (i < T) g = __hip_atomic_load(eng + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all((unsigned)(g >> 32) == (unsigned)(t + 1))) { if (i < len_b) e = __builtin_bit_cast(float, (unsigned)g); break; }
                        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT_TICKS) { bad = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (bad) break;
                    if (i < T) s_e[wave * TP + i] = e;
                    m = fmaxf(m, e);
                }
                if (bad) s_flag[0] = 1;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                float sum = 0.f;
                for (int i = lane; i < T; i += 64) {
                    const float ex = (i < len_b) ? expf(s_e[wave * TP + i] - m) : 0.f;
                    s_e[wave * TP + i] = ex;
                    sum += ex;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                // wave w publishes the probabilities of positions i with (i >> 6) % 8 == w ... every wave holds all of them; each
                // writes the same values, so one copy per position is enough: wave (i / 64) % 8 writes block i / 64
                for (int i0 = 0; i0 < T; i0 += 64)
                    if (((i0 >> 6) & (NWAVE - 1)) == wave && i0 + lane < T) s_a[i0 + lane] = s_e[wave * TP + i0 + lane] / sum;
            }
            __syncthreads();
            if (s_flag[0]) { alive = false; break; }
            PZ_STAMP(20);
            // (e) this chunk's alignments: history row and the next attention state (:137-140)
            if (tid < n) {
                const float p = s_a[p0 + tid];
                a.hist[((size_t)t * B + ab) * T + p0 + tid] = p;
                if constexpr (TAPE) x.tp.state_prev[((size_t)t * B + ab) * T + p0 + tid] = s_cum[tid + pl];
                st4(xs, a.oCUM + (unsigned)((t & 1) * B * T * 4) + (unsigned)((ab * T + p0 + tid) * 4), a.cumulative ? p + s_cum[tid + pl] : p);
            }
            // (f) context slice = alignments . values[:, slice]   (AttentionWrapper._compute_attention)
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
                if constexpr (TAPE) *reinterpret_cast<f32x4*>(x.tp.ctx + ((size_t)t * B + ab) * EW + ac * ES + tid * 4) = s;
            }
        }
        PZ_STAMP(6); hop_arrive(a.bar); ++epoch; fresh();
        // gap: attention cell output into LSTM 1 (rnn_wrappers.py:76-78 concat, first K segment)
        {
            f32x4 b[8];
            wload<8>(b, r1, slab1 + (unsigned)(wave * 8 * 1024), l16);
            seg_mma<MB, 8>(acc1, b, xs, a.oHAr + (unsigned)(wave * 8 * MB * 1024), l16);
        }
        const unsigned pk2 = hop_peek(a.bar);
        // ... and the second half of LSTM 2's recurrent K range
        if (t > 0) {
            f32x4 b[4];
            wload<4>(b, r2, slab2 + (unsigned)((KBH + wave * 8 + 4) * 1024), l16);
            seg_mma<MB, 4>(acc2, b, xs, a.oH2 + (unsigned)((wave * 8 + 4) * MB * 1024), l16);
        }
        f32x4 bc1[4];
        wload<4>(bc1, r1, slab1 + (unsigned)((KBH + wave * 4) * 1024), l16);
        PZ_STAMP(7);
        if (!hop_wait(a.bar, 32u * epoch, s_flag, pk2)) { alive = false; break; }
        fresh(); PZ_STAMP(8);

        // ================= P3: decoder LSTM 1 (tacotron_v2.py:155-161) =================
        seg_mma<MB, 4>(acc1, bc1, xs, a.oCTX + (unsigned)(wave * 4 * MB * 1024), l16);
        put_red(acc1);
        __syncthreads();
        if (tid < MB * 64) {
            float z[4];
            gate4(z);
            const float hn = cell(z, s_bias + 16, c1, h1, 1);
            publish4(hn, a.oY1);
            publish4(h1, a.oH1);
        }
        PZ_STAMP(9); hop_arrive(a.bar); ++epoch; fresh();
        // gap: the attention LSTM's recurrent K ranges for the NEXT step: [ctx(t) | h_att(t)]
        unsigned pk3 = 0u;
        {
            f32x4 b[8];
            wload<8>(b, rA, slabA + (unsigned)((KBX + KBE + wave * 8) * 1024), l16);
            seg_mma<MB, 8>(accA, b, xs, a.oHA + (unsigned)(wave * 8 * MB * 1024), l16);
            pk3 = hop_peek(a.bar);
            f32x4 bc[4];
            wload<4>(bc, rA, slabA + (unsigned)((KBX + wave * 4) * 1024), l16);
            seg_mma<MB, 4>(accA, bc, xs, a.oCTX + (unsigned)(wave * 4 * MB * 1024), l16);
        }
        f32x4 by[8];
        wload<8>(by, r2, slab2 + (unsigned)(wave * 8 * 1024), l16);
        PZ_STAMP(10);
        if (!hop_wait(a.bar, 32u * epoch, s_flag, pk3)) { alive = false; break; }
        fresh(); PZ_STAMP(11);

        // ================= P4: decoder LSTM 2 =================
        seg_mma<MB, 8>(acc2, by, xs, a.oY1 + (unsigned)(wave * 8 * MB * 1024), l16);
        put_red(acc2);
        __syncthreads();
        if (tid < MB * 64) {
            float z[4];
            gate4(z);
            const float hn = cell(z, s_bias + 32, c2, h2, 2);
            publish4(hn, a.oY2);
            publish4(h2, a.oH2);
            if (TEACH && x.y_all != nullptr && eown) x.y_all[((size_t)t * B + erow) * (NWG * 4) + wg * 4 + euu] = hn;
        }
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) acc2[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        PZ_STAMP(12); hop_arrive(a.bar); ++epoch; fresh();
        // gap: LSTM 1's recurrent K range for the next step
        unsigned pk4 = 0u;
        {
            f32x4 b[8];
            wload<8>(b, r1, slab1 + (unsigned)((KBH + KBE + wave * 8) * 1024), l16);
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) acc1[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 blo[4] = {b[0], b[1], b[2], b[3]}, bhi[4] = {b[4], b[5], b[6], b[7]};
            seg_mma<MB, 4>(acc1, blo, xs, a.oH1 + (unsigned)(wave * 8 * MB * 1024), l16);
            pk4 = hop_peek(a.bar);
            seg_mma<MB, 4>(acc1, bhi, xs, a.oH1 + (unsigned)((wave * 8 + 4) * MB * 1024), l16);
        }
        if (TEACH && isAtt) { attention_pre(t + 1); fresh(); }
        f32x4 bp[8];
        float bias_p = 0.f;
        if (isP && !TEACH) { wload<8>(bp, rP, (unsigned)((ptile * KBH + wave * 8) * 1024), l16); bias_p = a.bP[ptile * 16 + (tid & 15)]; }
        PZ_STAMP(13);
        if (!hop_wait(a.bar, 32u * epoch, s_flag, pk4)) { alive = false; break; }
        fresh(); PZ_STAMP(14);

        // ================= P5: mel / stop projections (rnn_wrappers.py:105-109) + composed first PreNet layer of step t+1 =================
        if (isP && !TEACH) {
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
                s += bias_p;
                if (ptile >= 12) s_tile[r16 * 16 + c] = fmaxf(s, 0.f);
                else if (row < B) {
                    if (col < a.MR) a.mel[(size_t)row * a.mel_ld + (size_t)t * a.mel_ts + col] = s;
                    else if (col == a.MR) {
                        a.stop[(size_t)row * a.stop_ld + t] = s;
                        st4(xs, a.oSTOP + (unsigned)(row * 4), s);
                    }
                }
            }
            if (ptile >= 12) {
                __syncthreads();
                if (tid < 64) {
                    const int r16 = tid >> 2, q = tid & 3, row = prb * 16 + r16;
                    if (row < B) st16(xs, a.oX1 + frag_off16(row, (ptile - 12) * 4 + q, MB), *reinterpret_cast<const f32x4*>(s_tile + r16 * 16 + q * 4));
                }
                // row-block group hop towards the last PreNet layer's tiles of the next step
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(grpx + prb * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        PZ_STAMP(15);
    }
    // (the last step's stop decision changes nothing: the decode ends at max_steps either way -- dynamic_decode's maximum_iterations)
    // final_state: the hidden states are in the exchange slab already (last published values); the cell states live in registers
    fresh();
    if (eown) {
        a.cA_out[(size_t)erow * (NWG * 4) + wg * 4 + euu] = cA;
        a.c1_out[(size_t)erow * (NWG * 4) + wg * 4 + euu] = c1;
        a.c2_out[(size_t)erow * (NWG * 4) + wg * 4 + euu] = c2;
    }
#ifdef PZ_STAMPS
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 24; ++i) g_pz_stamps[wg][i] = st_acc[i];
    }
#endif
    if (wg == 0 && tid == 0) {
        a.ctl[CTL_FINAL] = final_steps;
        if (!alive) a.ctl[5] = 1;
    }
    if (!alive && tid == 0) atomicOr(reinterpret_cast<unsigned*>(a.ctl) + 6, 1u);
}

// ------------------------------------------------------------------------------------------------------------------------
// EncoderV2's bidirectional ZoneoutLSTM (tacotron_v2.py:53-63: bidirectional_dynamic_rnn with sequence_length) as ONE launch.
// 2 directions x 64 workgroups; workgroup g of a direction owns hidden units 4g .. 4g+3 (16 gate columns) and keeps its whole
// weight slab in registers for the sequence (K = 512 inputs + 256 recurrent: 48 KB per workgroup, 24 registers per lane) --
// the recurrence the register file can hold.  Per step: the input half x_s . Wx is multiplied in the gap of the previous
// step's hop (it depends on nothing), the recurrent half h . Wh after it; cell, inference zoneout, dynamic_rnn's length
// masking (dead steps copy the state through and emit zeros) and the plain output row follow in the same workgroup; the new
// h goes to the other 63 workgroups of the direction through the sc1 hand-off of the decoder kernel (hop_arrive / hop_wait,
// one counter set per direction).  The backward direction reads the per-row reversed operands seq_to_frag_kernel made and
// writes position len - 1 - s.
// ------------------------------------------------------------------------------------------------------------------------
struct EArgs {
    const float* w[2];          // packed slabs [64][48][64][4] per direction
    const float* bias[2];       // [64][16]
    const float* xf;            // [2][T][frag16 MB x 512] inputs per step (reversed per row for direction 1)
    unsigned xf_bytes, xf_step; // bytes of xf, bytes per step
    float* hx;                  // [2 directions][2 slots][frag16 MB x 256] state exchange (zeroed)
    unsigned hx_bytes;
    unsigned* bar;              // [2][NSHARD][32] hop counters (zeroed)
    int* err;                   // set to 1 when a wait ran out of time
    const int* lens;
    float* out;                 // [B][T][512]
    int B, T;
    float zc, zh, fb;
    // training forward (TRAIN instantiation): sampled zoneout (modules.py:274-276) and the tape of the backward pass,
    // per direction [T][B][.] in step order (train_step.h EncTape)
    float *tz[2], *tcprev[2], *tcnew[2], *thstate[2];
    unsigned seed; int stream;
};
constexpr int ENC_G = 64;       // workgroups per direction (256 hidden units / 4)

template <int MB, bool TRAIN = false>
__global__ __launch_bounds__(NT) void encoder_kernel(EArgs a) {
    __shared__ f32x4 red[NWAVE * MB * 64];
    __shared__ float s_bias[16];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / ENC_G, g = blockIdx.x - dir * ENC_G;
    const int B = a.B, T = a.T;
    const unsigned l16 = (unsigned)lane * 16u;
    if (tid < 16) s_bias[tid] = a.bias[dir][(size_t)g * 16 + tid];
    if (tid == 0) s_abort = 0;
    const rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w[dir]), 0, ENC_G * 48 * 1024, 0x00020000);
    const rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xf), 0, (int)a.xf_bytes, 0x00020000);
    const rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(a.hx, 0, (int)a.hx_bytes, 0x00020000);
    unsigned* bar = a.bar + dir * NSHARD * 32;
    // the whole slab of this workgroup, once: 4 input k-blocks and 2 recurrent k-blocks per wave
    f32x4 bx[4], bh[2];
    wload<4>(bx, rW, (unsigned)((g * 48 + wave * 4) * 1024), l16);
    wload<2>(bh, rW, (unsigned)((g * 48 + 32 + wave * 2) * 1024), l16);
    const int erow = tid >> 2, euu = tid & 3;
    const bool eown = tid < MB * 64 && erow < B;
    const int len = eown ? a.lens[erow] : 0;
    float c = 0.f, h = 0.f;
    const unsigned hslot = (unsigned)(MB * 16 * 256 * 4), hdir = (unsigned)dir * 2u * hslot;
    const unsigned xdir = (unsigned)dir * (unsigned)T * a.xf_step;
    f32x4 acc[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    seg_mma<MB, 4>(acc, bx, rX, xdir + (unsigned)(wave * 4 * MB * 1024), l16);
    unsigned epoch = 0;
    bool alive = true;
    for (int s = 0; s < T; ++s) {
        if (s > 0) {
            if (!hop_wait(bar, (unsigned)(ENC_G / NSHARD) * epoch, &s_abort)) { alive = false; break; }
            seg_mma<MB, 2>(acc, bh, rH, hdir + (unsigned)(s & 1) * hslot + (unsigned)(wave * 2 * MB * 1024), l16);
        }
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
        __syncthreads();
        if (tid < MB * 64) {
            const int rb = erow >> 4, rr = erow & 15;
            const int reg = rr & 3, lbase = (rr >> 2) * 16;
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) sum += red[(w * MB + rb) * 64 + lbase + q * 4 + euu][reg];
                z[q] = sum + s_bias[q * 4 + euu];
            }
            // LSTMCell + inference zoneout (modules.py:261-283); dynamic_rnn: past the row's length the state is copied through
            const float cn = sigmoidf_(z[2] + a.fb) * c + sigmoidf_(z[0]) * tanhf(z[1]);
            const float hn = sigmoidf_(z[3]) * tanhf(cn);
            const bool live = s < len;
            const float cp = c;
            if (live) {
                if constexpr (TRAIN) {
                    const unsigned stream = (unsigned)(a.stream + 2 * dir), unit = (unsigned)(g * 4 + euu);
                    c = (1.0f - a.zc) * dropout_train(cn - c, a.zc, t2_u01(a.seed, stream, (unsigned)s, (unsigned)erow, unit)) + c;
                    h = (1.0f - a.zh) * dropout_train(hn - h, a.zh, t2_u01(a.seed, stream + 1u, (unsigned)s, (unsigned)erow, unit)) + h;
                } else {
                    c = (1.0f - a.zc) * cn + a.zc * c;
                    h = (1.0f - a.zh) * hn + a.zh * h;
                }
                const int tpos = dir ? (len - 1 - s) : s;
                a.out[((size_t)erow * T + tpos) * 512 + dir * 256 + g * 4 + euu] = hn;
            }
            if (TRAIN && eown) {
                const size_t ti = (size_t)s * B + erow;
                const int unit = g * 4 + euu;
                float* tz = a.tz[dir] + ti * 1024;
                tz[unit] = z[0]; tz[256 + unit] = z[1]; tz[512 + unit] = z[2]; tz[768 + unit] = z[3];
                a.tcprev[dir][ti * 256 + unit] = cp; a.tcnew[dir][ti * 256 + unit] = cn; a.thstate[dir][ti * 256 + unit] = h;
            }
            f32x4 q4;
            q4[0] = __shfl(h, (lane & ~3) + 0); q4[1] = __shfl(h, (lane & ~3) + 1);
            q4[2] = __shfl(h, (lane & ~3) + 2); q4[3] = __shfl(h, (lane & ~3) + 3);
            if (eown && euu == 0) st16(rH, hdir + (unsigned)((s + 1) & 1) * hslot + frag_off16(erow, g, MB), q4);
        }
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s + 1 < T) {
            hop_arrive(bar); ++epoch;
            seg_mma<MB, 4>(acc, bx, rX, xdir + (unsigned)(s + 1) * a.xf_step + (unsigned)(wave * 4 * MB * 1024), l16);
        }
    }
    if (!alive && tid == 0) atomicOr(reinterpret_cast<unsigned*>(a.err), 1u);
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
