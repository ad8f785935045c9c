// Persistent backward recurrence of the teacher-forced training step: the whole reverse-time loop of the V2 cell stack
// (ZoneoutLSTMCell x 3, rnn_wrappers.py:118-127,143-144 AttentionRNN, tacotron_v2.py:118-141 location-sensitive attention,
// :155-161 DecoderRNNV2) as ONE launch, for the paper-size dimension class of persist.h.
//
// Structure (mirrors pz::decode_kernel).  256 workgroups stay resident for all S steps.  Workgroup g owns hidden units
// 4g..4g+3 of the three LSTM cells: the gradients of their cell / hidden states are carried in registers across steps, the
// ZoneoutLSTMCell backward (modules.py:261-283, sampled masks from the counter-based generator) runs where the units live, and
// the gate gradients dz [rows x 4H] are published to the exchange slab as frag16 operands.  The data-gradient products
// d x = dz . W^T (K = 4H = 4096) are 2-D decomposed: only the columns somebody consumes are computed (a workgroup's units' d h,
// its two context columns, its PreNet column: 7..10 columns per owner, packed densely by pack_bwd_tiles2d_kernel); four
// consecutive owners form a group, and workgroup (group, ks) multiplies K slice ks (1024 of the 4096 rows of W^T) by the
// group's 28..40 columns -- a quarter of dz instead of all of it through every CU, and full MFMA tiles instead of one half-empty
// tile.  The partial tiles go to the slab; every consumer of a product sits behind a later hop anyway and adds the four
// partials when it loads them (cells: their own columns; attention: d ctx).  A step is four all-to-all hops:
//   LSTM 2 backward of step t-1 and LSTM 1 backward of step t (dz2, dz1 published)               -> hop A ->
//   GEMM 1 (dz1 -> d h_att | d h1_prev | d ctx partial tiles)                                    -> hop B ->
//   attention backward (row split over C workgroups by encoder position, partial d q / d state)  -> hop C ->
//   query-layer product (sum of the C partial d q), attention LSTM backward (dz_att published)   -> hop D.
// Everything else fills the time between a hop's arrive and wait: GEMM 0 of step t+1 (dz_att -> d prenet | d ctx_prev |
// d h_att_prev) around hop A, GEMM 2 of step t-1 (dz2 -> d y1 | d h2_prev; LSTM 2's chain runs one step ahead: dz2 of step
// t-1 needs only GEMM 2 of step t) around hops B and C, the attention's query-independent work in the gap of hop B, its
// d f / partial d state_prev around hop D.  At 32 / 64 rows a piece of that work also runs just AHEAD of the arrive, over the
// drain of the stores the phase issued.  Hand-off protocol: persist.h (sc1 stores, drain, barrier, one agent-scope add per
// workgroup; sc1 loads).  Everything the weight-gradient GEMMs after the loop read (dz, dq, du / f / df, d ctx, d prenet) is
// written to the same plain tape buffers as the launch-per-phase backward writes, so the code after the loop is shared.
#pragma once
#include "persist.h"
#include <type_traits>

namespace pz {

struct BArgs {
    const float *wT2, *wT1, *wT0, *wTQ;        // tiles of W^T: 2-D [256][64][4][4 NC][4] (pack_bwd_tiles2d_kernel); query layer 1-D [256][8][4][NC][4]
    const float *conv_w, *conv_b, *wloc, *v_a;
    const float *keys, *values;                // [B][T][128], [B][T][512]
    const int* lens;
    // forward tape (TrainTape) and loss gradient
    const float *z[3], *cprev[3], *cnew[3];    // attention LSTM, LSTM 1, LSTM 2
    const float *q, *state_prev, *ctx, *hist;  // [S][B][128], [S][B][T], [S][B][512], [S][B][T]
    const float* dy2_all;                      // [S][B][1024] d (LSTM 2 raw output) from the projections
    const float* xp_last;                      // [S][B][256] last PreNet layer's output after dropout
    // backward tape (read by the weight-gradient GEMMs after the loop)
    float* dz[3];                              // [S][B][4096]
    float *dq, *de_all, *dctx_all, *du_all, *f_all, *df_all, *dpre, *pva_part;
    long long dctx_row_ld;                     // floats between batch rows of dctx_all (S * 512)
    float* xs; unsigned xs_bytes;
    unsigned oDZ2, oDZ1, oDZ0, oDCTX, oDQ, oDS; // byte offsets into xs; DZ2 and DS hold two step parities (oDCTX: unused)
    unsigned oP2, oP1, oP0;                     // partial product tiles [256 workgroups][rows][4 NC]
    unsigned dz_bytes, dq_bytes;               // bytes of one dz operand / one d q partial operand
    unsigned* bar; int* err;
    int B, T, S, C, CH, F, KA, cumulative;
    unsigned seed;
    float zc, zh, fb, inv_keep;
};

constexpr int NC2 = 8, NC1 = 10, NC0 = 7, NCQ = 4;     // columns a workgroup owns in the four products
constexpr int KBZ = 256;                                // k-blocks of a dz operand (4H / 16)
constexpr int DUS = 132;                                // LDS row stride of the du tile

__host__ __device__ inline int bwd_lds_floats(int MB, int T, int CH) {
    const int CHP = (CH + 15) & ~15, TP = (T + 3) & ~3;
    return NWAVE * 3 * MB * 64 * 4 + 512 + 2 * TP + (CHP + 48) + 2 * CHP * 36 + CHP * DUS + 32 * 33 + 32 + 2 * UA + 2 * CHP + 32 + 16;
}

// Source: plain [N][Kc] row-major (a TF kernel [inputs][outputs]: d x[n] = sum_k dz[k] W[n][k]).  Tile of workgroup g: column c
// is source row n(g, c) = base[s] + per_g[s] * g + (c - c0[s]) of the segment s that holds c.  Output [g][kb][kg][c][j] with
// k = 16 kb + 4 kg + j: one k-block is 4 * NC float4, the B operand of lane (kg, c) of v_mfma_f32_16x16x4_f32.
struct TileMap { int nseg, NC, c0[3], base[3], per_g[3]; };
__global__ __launch_bounds__(256) void pack_bwd_tiles_kernel(const float* __restrict__ src, int Kc, int nwg, TileMap mp, float* __restrict__ dst) {
    const int KB = Kc / 16;
    const long long n4 = (long long)nwg * KB * 4 * mp.NC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % mp.NC);
        const int kg = (int)((i / mp.NC) & 3);
        const int kb = (int)((i / (4 * mp.NC)) % KB);
        const int g = (int)(i / ((long long)4 * mp.NC * KB));
        int s = 0;
        for (int q = 1; q < mp.nseg; ++q) if (c >= mp.c0[q]) s = q;
        const int n = mp.base[s] + mp.per_g[s] * g + (c - mp.c0[s]);
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)n * Kc + kb * 16 + kg * 4);
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

template <int NKB>
__device__ __forceinline__ void wloadc(f32x4 (&b)[NKB], rsrc_t w, unsigned soff, unsigned loff, unsigned kbs) {
#pragma unroll
    for (int i = 0; i < NKB; ++i) b[i] = ldw(w, loff, soff + (unsigned)i * kbs);
}

// acc += dz[rows x 4096] . tile: this wave's 32 k-blocks in four rounds, the next round's weights in flight during a round
template <int MB>
__device__ __forceinline__ void gemm_dz(f32x4 (&acc)[MB], rsrc_t rW, unsigned wsoff, unsigned kbs, unsigned loff, rsrc_t xs, unsigned xoff,
                                        unsigned l16) {
    f32x4 b0[8], b1[8];
    wloadc<8>(b0, rW, wsoff, loff, kbs);
    wloadc<8>(b1, rW, wsoff + 8u * kbs, loff, kbs);
    seg_mma<MB, 8>(acc, b0, xs, xoff, l16);
    wloadc<8>(b0, rW, wsoff + 16u * kbs, loff, kbs);
    seg_mma<MB, 8>(acc, b1, xs, xoff + (unsigned)(8 * MB * 1024), l16);
    wloadc<8>(b1, rW, wsoff + 24u * kbs, loff, kbs);
    seg_mma<MB, 8>(acc, b0, xs, xoff + (unsigned)(16 * MB * 1024), l16);
    seg_mma<MB, 8>(acc, b1, xs, xoff + (unsigned)(24 * MB * 1024), l16);
}


// 2-D tiles: workgroup w = 4 group + ks holds k-blocks [64 ks, 64 ks + 64) x the 4 NC columns of owners 4 group .. 4 group + 3
// (column cc = owner_local * NC + c).  Output [w][kb_local][kg][cc][j], k = 16 (64 ks + kb_local) + 4 kg + j.
__global__ __launch_bounds__(256) void pack_bwd_tiles2d_kernel(const float* __restrict__ src, int Kc, TileMap mp, float* __restrict__ dst) {
    const int NC4 = 4 * mp.NC;
    const long long n4 = (long long)NWG * 64 * 4 * NC4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % NC4);
        const int kg = (int)((i / NC4) & 3);
        const int kbl = (int)((i / (4 * NC4)) & 63);
        const int w = (int)(i / ((long long)4 * NC4 * 64));
        const int g = (w & ~3) + cc / mp.NC, c = cc % mp.NC, ks = w & 3;
        int s = 0;
        for (int q = 1; q < mp.nseg; ++q) if (c >= mp.c0[q]) s = q;
        const int n = mp.base[s] + mp.per_g[s] * g + (c - mp.c0[s]);
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)n * Kc + (64 * ks + kbl) * 16 + kg * 4);
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// acc[ct][rb] += dz[rows, NKB k-blocks at xoff] . tile columns 16 ct ..: software-pipelined in pieces of two k-blocks (the
// next piece's loads are in flight while a piece multiplies)
template <int MB, int CT, int NKB>
__device__ __forceinline__ void gemm2d_part(f32x4 (&acc)[CT][MB], rsrc_t rW, unsigned wsoff, unsigned kbs, const unsigned (&loff)[CT], rsrc_t xs,
                                            unsigned xoff, unsigned l16) {
    constexpr int STEP = MB == 4 ? 1 : 2, NP = NKB / STEP;      // 64 rows: one k-block per piece (register budget)
    static_assert(NKB % STEP == 0, "whole pieces");
    f32x4 av[2][STEP][MB], bv[2][STEP][CT];
    auto load = [&](int p) {
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) av[p & 1][i][rb] = ldx(xs, l16, xoff + (unsigned)(((p * STEP + i) * MB + rb) * 1024));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bv[p & 1][i][ct] = ldw(rW, loff[ct], wsoff + (unsigned)(p * STEP + i) * kbs);
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    load(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        __builtin_amdgcn_sched_barrier(0);
        if (p + 1 < NP) load(p + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEP; ++i)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb)
                        acc[ct][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[p & 1][i][rb][j], bv[p & 1][i][ct][j], acc[ct][rb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}
constexpr int CT2 = 2, CT1 = 3, CT0 = 2;              // MFMA column tiles of a group's 4 NC columns

template <int MB>
__global__ __launch_bounds__(NT) void backward_kernel(BArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int tid = threadIdx.x, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int B = a.B, T = a.T, S = a.S, C = a.C, CH = a.CH;
    const int CHP = (CH + 15) & ~15, TP = (T + 3) & ~3, NIP = CH + 32;
    // ---- LDS carve ----
    f32x4* red = reinterpret_cast<f32x4*>(smem);                 // [8][3 column tiles][MB][64] float4
    float* s_dctx = smem + NWAVE * 3 * MB * 64 * 4;              // [512]
    float* s_al = s_dctx + 512;                                  // [TP] alignments of the step
    float* s_gs = s_al + TP;                                     // [TP] d / d attention state after the step (carried)
    float* s_sp = s_gs + TP;                                     // [CHP + 48] attention state before the step, with the conv halo
    float* s_f = s_sp + CHP + 48;                                // [CHP][36]
    float* s_df = s_f + CHP * 36;                                // [CHP][36]
    float* s_du = s_df + CHP * 36;                               // [CHP][DUS]
    float* s_cw = s_du + CHP * DUS;                              // [32 taps][33]: 32 filters, odd stride (conflict-free by tap and by filter)
    float* s_cb = s_cw + 32 * 33;                                // [32]
    float* s_q = s_cb + 32;                                      // [128]
    float* s_dq = s_q + UA;                                      // [128]
    float* s_de = s_dq + UA;                                     // [CHP]
    float* s_da = s_de + CHP;                                    // [CHP]
    float* s_rd = s_da + CHP;                                    // [32] block reductions
    int* s_flag = reinterpret_cast<int*>(s_rd + 32);             // [0] abort

    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(a.xs, 0, (int)a.xs_bytes, 0x00020000);
    const rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wT2), 0, NWG * KBZ * 64 * NC2, 0x00020000);
    const rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wT1), 0, NWG * KBZ * 64 * NC1, 0x00020000);
    const rsrc_t rW0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wT0), 0, NWG * KBZ * 64 * NC0, 0x00020000);
    const rsrc_t rWQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wTQ), 0, NWG * 8 * 64 * NCQ, 0x00020000);

    // ---- roles ----
    const bool isAtt = wg < B * C;
    const int ab = isAtt ? wg / C : 0, ac = isAtt ? wg - ab * C : 0;
    const int p0 = ac * CH, pl = (a.KA - 1) >> 1;
    const int n_pos = isAtt ? max(0, min(CH, T - p0)) : 0;
    const int NRT = (n_pos + 15) >> 4;
    const int len_b = isAtt ? min(a.lens[ab], T) : 0;

    if (tid == 0) s_flag[0] = 0;
    for (int i = tid; i < 32 * 32; i += NT) {
        const int j = i >> 5, f = i & 31;
        s_cw[j * 33 + f] = (j < a.KA && f < a.F) ? a.conv_w[j * a.F + f] : 0.f;
    }
    if (tid < 32) s_cb[tid] = tid < a.F ? a.conv_b[tid] : 0.f;
    for (int i = tid; i < TP; i += NT) s_gs[i] = 0.f;
    int kg = lane >> 4, c16 = lane & 15;
    // location layer as B fragments (forward orientation: k = filter, column = this wave's 16 attention units)
    float wlB[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int k = 16 * (s >> 2) + 4 * kg + (s & 3);
        wlB[s] = k < a.F ? a.wloc[(size_t)k * UA + wave * 16 + c16] : 0.f;
    }
    // ... and transposed (k = attention unit, column = filter 16 (wave & 1) + c16) for d f = d u . Wloc^T
    f32x4 wdf[8];
    {
        const int filt = 16 * (wave & 1) + c16;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) wdf[kb][j] = filt < a.F ? a.wloc[(size_t)filt * UA + 16 * kb + 4 * kg + j] : 0.f;
    }
    const float va_l = a.v_a[wave * 16 + c16];
    float dva_acc = 0.f;

    int erow = tid >> 2, euu = tid & 3;
    bool eown = tid < MB * 64 && erow < B;
    unsigned l16 = (unsigned)lane * 16u;
    auto fresh = [&]() {
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        kg = lane >> 4; c16 = lane & 15; erow = tid >> 2; euu = tid & 3;
        eown = tid < MB * 64 && erow < B; l16 = (unsigned)lane * 16u;
    };
    // carried gradients of this thread's (row, unit): cell state, the part of d h_prev that bypasses the cell, and the parts
    // of next step's d state that this workgroup's own product columns produced
    float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, ghp0 = 0.f, ghp1 = 0.f, ghp2 = 0.f;
    float dh2p = 0.f, dh1p = 0.f, dhAp = 0.f;

    auto colsum = [&](int c) -> float {
        const int rb = erow >> 4, rr = erow & 15;
        const int reg = rr & 3, lbase = (rr >> 2) * 16;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NWAVE; ++w) s += red[(w * MB + rb) * 64 + lbase + c][reg];
        return s;
    };
    auto put_red = [&](f32x4 (&acc)[MB]) {
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    };
    struct TapeRow { float zi, zj, zf, zo, cp, cn; };
    auto tape_load = [&](int layer, int tt) -> TapeRow {
        TapeRow r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (eown && tt >= 0) {
            const int unit = wg * 4 + euu, H = NWG * 4;
            const size_t ti = (size_t)tt * B + erow;
            const float* tz = a.z[layer] + ti * 4 * H;
            r.zi = tz[unit]; r.zj = tz[H + unit]; r.zf = tz[2 * H + unit]; r.zo = tz[3 * H + unit];
            r.cp = a.cprev[layer][ti * H + unit]; r.cn = a.cnew[layer][ti * H + unit];
        }
        return r;
    };
    // ZoneoutLSTMCell backward of one (row, unit) (train_step.h lstm_bwd_elem): writes dz to the tape and the exchange slab
    auto cell_bwd = [&](int layer, int tt, const TapeRow& r, float g_out, float g_hs, float& gc, float& ghp, unsigned off_dz) {
        const int unit = wg * 4 + euu, H = NWG * 4;
        const unsigned stream = (unsigned)(T2_STREAM_ZONE_ATT + 2 * layer);
        const float gi = sigmoidf_(r.zi), gj = tanhf(r.zj), gf = sigmoidf_(r.zf + a.fb), go = sigmoidf_(r.zo);
        const float mc = t2_u01(a.seed, stream, (unsigned)tt, (unsigned)erow, (unsigned)unit) >= a.zc ? 1.f : 0.f;
        const float mh = t2_u01(a.seed, stream + 1u, (unsigned)tt, (unsigned)erow, (unsigned)unit) >= a.zh ? 1.f : 0.f;
        const float tc = tanhf(r.cn);
        const float dhn = g_out + mh * g_hs;
        const float dcn = mc * gc + dhn * go * (1.f - tc * tc);
        gc = (1.f - mc) * gc + dcn * gf;
        ghp = (1.f - mh) * g_hs;
        float dz4[4] = {dcn * gj * gi * (1.f - gi), dcn * gi * (1.f - gj * gj), dcn * r.cp * gf * (1.f - gf), dhn * tc * go * (1.f - go)};
        if (!eown) { dz4[0] = 0.f; dz4[1] = 0.f; dz4[2] = 0.f; dz4[3] = 0.f; }
        float* dzt = a.dz[layer] + ((size_t)tt * B + erow) * 4 * H + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (eown) dzt[g * H] = dz4[g];
            f32x4 q;
            q[0] = __shfl(dz4[g], (lane & ~3) + 0); q[1] = __shfl(dz4[g], (lane & ~3) + 1);
            q[2] = __shfl(dz4[g], (lane & ~3) + 2); q[3] = __shfl(dz4[g], (lane & ~3) + 3);
            if (eown && euu == 0) st16(xs, off_dz + frag_off16(erow, g * NWG + wg, MB), q);
        }
    };

#ifdef PZ_STAMPS
    unsigned long long st_acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) st_acc[i] = 0;
    unsigned long long st_prev = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned epoch = 0;
    bool alive = true;
    f32x4 acc[MB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    __syncthreads();

    // ---- 2-D products: this workgroup = (group of four owners, K slice) ----
    const int ks = wg & 3, grp = wg >> 2, own_j = wg & 3;        // as an owner, this workgroup is member own_j of its group
    // reduce the waves' partial tiles and publish this workgroup's [rows][4 NC] partial product
    auto publish_tiles = [&](auto& accv, auto CTc, unsigned oP, int NC4) {
        constexpr int CT = decltype(CTc)::value;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) red[((wave * CT + ct) * MB + rb) * 64 + lane] = accv[ct][rb];
        __syncthreads();
        for (int idx = tid; idx < MB * 16 * CT * 16; idx += NT) {
            const int row = idx / (CT * 16), cc = idx - row * (CT * 16);
            if (cc < NC4) {
                const int ct = cc >> 4, c = cc & 15, rb = row >> 4, rr = row & 15;
                const int reg = rr & 3, lbase = (rr >> 2) * 16;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NWAVE; ++w) sum += red[((w * CT + ct) * MB + rb) * 64 + lbase + c][reg];
                st4(xs, oP + (unsigned)(((wg * MB * 16 + row) * NC4 + cc) * 4), sum);
            }
        }
        __syncthreads();          // the next product overwrites the reduction tiles
    };
    // column c of this owner's product: the sum of the group's four K-slice partials
    auto pread = [&](unsigned oP, int NC, int c) -> float {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld4(xs, oP + (unsigned)(((((grp * 4 + k) * MB * 16) + erow) * 4 * NC + own_j * NC + c) * 4));
        return (v[0] + v[1]) + (v[2] + v[3]);
    };
    unsigned loff2[CT2], loff1[CT1], loff0[CT0];
#pragma unroll
    for (int ct = 0; ct < CT2; ++ct) loff2[ct] = (unsigned)((kg * 4 * NC2 + min(16 * ct + c16, 4 * NC2 - 1)) * 16);
#pragma unroll
    for (int ct = 0; ct < CT1; ++ct) loff1[ct] = (unsigned)((kg * 4 * NC1 + min(16 * ct + c16, 4 * NC1 - 1)) * 16);
#pragma unroll
    for (int ct = 0; ct < CT0; ++ct) loff0[ct] = (unsigned)((kg * 4 * NC0 + min(16 * ct + c16, 4 * NC0 - 1)) * 16);
    f32x4 acc2[CT2][MB];       // GEMM 2 of the step ahead, accumulated piecewise in the hop gaps
#pragma unroll
    for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) acc2[ct][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // k-blocks [k0, k0 + 4) of this wave's 8 of GEMM 2's K slice over dz2(tt)
    auto gemm2_part = [&](int tt, int k0) {
        gemm2d_part<MB, CT2, 4>(acc2, rW2, (unsigned)((wg * 64 + wave * 8 + k0) * 256 * NC2), 256u * NC2, loff2, xs,
                                a.oDZ2 + (unsigned)(tt & 1) * a.dz_bytes + (unsigned)((ks * 64 + wave * 8 + k0) * MB * 1024), l16);
    };
    auto gemm2_publish = [&]() {
        publish_tiles(acc2, std::integral_constant<int, CT2>{}, a.oP2, 4 * NC2);
#pragma unroll
        for (int ct = 0; ct < CT2; ++ct)
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) acc2[ct][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // GEMM 0 of a step (d [prenet out | ctx_prev | h_att_prev] = dz_att . W0^T): this workgroup's partial tile, in two halves
    // (the first runs before a hop's arrive and overlaps the drain of the stores just issued, the second in the hop's gap)
    f32x4 acc0[CT0][MB];
    auto gemm0_half0 = [&]() {
#pragma unroll
        for (int ct = 0; ct < CT0; ++ct)
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) acc0[ct][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm2d_part<MB, CT0, 4>(acc0, rW0, (unsigned)((wg * 64 + wave * 8) * 256 * NC0), 256u * NC0, loff0, xs,
                                a.oDZ0 + (unsigned)((ks * 64 + wave * 8) * MB * 1024), l16);
    };
    // (64 rows: the accumulators would not survive the hop without spilling; both halves run in the gap)
    // (16 rows: the step is hop-latency bound, work ahead of an arrive only delays it -- everything stays in the gaps)
    auto gemm0_first = [&]() { if constexpr (MB == 2) gemm0_half0(); };
    auto gemm0_second = [&]() {
        if constexpr (MB != 2) gemm0_half0();
        gemm2d_part<MB, CT0, 4>(acc0, rW0, (unsigned)((wg * 64 + wave * 8 + 4) * 256 * NC0), 256u * NC0, loff0, xs,
                                a.oDZ0 + (unsigned)((ks * 64 + wave * 8 + 4) * MB * 1024), l16);
        publish_tiles(acc0, std::integral_constant<int, CT0>{}, a.oP0, 4 * NC0);
    };
    // ---- attention backward, part 1 (hop gap): everything that does not need d ctx of the step ----
    float pre[4][4];
    float ctx_v = 0.f, part_gs = 0.f;
    constexpr int VN = 5;       // values rows per wave kept in registers across the hop (positions wave, wave + 8, ...)
    float vv[VN][8];
    auto att_pre = [&](int t) {
        const size_t tb = (size_t)t * B + ab;
        ctx_v = a.ctx[tb * EW + tid];
        for (int p = tid; p < T; p += NT) s_al[p] = a.hist[tb * T + p];
        // d state after this step = (cumulative: d state after step t + 1) + the C chunks' partials of step t + 1
        if (t < S - 1) {
            const unsigned ds_prev = a.oDS + (unsigned)((((t + 1) & 1) * B + ab) * C * NIP * 4);
            for (int p = tid; p < T; p += NT) {
                float s = a.cumulative ? s_gs[p] : 0.f;
                // chunks whose window [c CH - pad_l, c CH - pad_l + NIP) holds p; four loads in flight per round
                const int c_hi = min(C - 1, (p + pl) / CH);
                const int c_lo = max(0, (p + pl - NIP + CH) / CH);
                for (int c0 = c_lo; c0 <= c_hi; c0 += 4) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c2 = min(c0 + k, c_hi);
                        v[k] = ld4(xs, ds_prev + (unsigned)((c2 * NIP + p - (c2 * CH - pl)) * 4));
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (c0 + k <= c_hi) s += v[k];
                }
                s_gs[p] = s;
            }
        }
        for (int i = tid; i < CHP + 48; i += NT) {
            const int p = p0 - pl + i;
            s_sp[i] = (p >= 0 && p < T && i < n_pos + a.KA - 1) ? a.state_prev[tb * T + p] : 0.f;
        }
        if (tid < UA) s_q[tid] = a.q[tb * UA + tid];
        {
            const float* keys_b = a.keys + (size_t)ab * T * UA + wave * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p = min(p0 + 16 * r + 4 * kg + g, T - 1);
                    pre[r][g] = r < NRT ? keys_b[(size_t)p * UA] : 0.f;
                }
            const float* vals = a.values + (size_t)ab * T * EW;
#pragma unroll
            for (int jj = 0; jj < VN; ++jj) {
                const int jp = min(p0 + wave + jj * NWAVE, T - 1);      // rows past the chunk are loaded (in bounds) and not used
#pragma unroll
                for (int i = 0; i < 8; ++i) vv[jj][i] = vals[(size_t)jp * EW + lane + 64 * i];
            }
        }
        __syncthreads();
        // a . d state part of the softmax-backward dot (every chunk computes the row's whole dot)
        part_gs = 0.f;
        for (int p = tid; p < T; p += NT) part_gs = fmaf(s_al[p], s_gs[p], part_gs);
        // location features of the chunk (as persist.h attention_pre)
        {
            const int f = tid & 31, pq = tid >> 5;
            if (4 * pq < CHP) {
                float win[36];
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_sp + 4 * pq + 4 * i);
                    win[4 * i] = w4[0]; win[4 * i + 1] = w4[1]; win[4 * i + 2] = w4[2]; win[4 * i + 3] = w4[3];
                }
                const float cb = s_cb[f];
                float fa[4] = {cb, cb, cb, cb};
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const float wj = s_cw[j * 33 + f];
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i] = fmaf(wj, win[i + j], fa[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) s_f[(4 * pq + i) * 36 + f] = (4 * pq + i < n_pos) ? fa[i] : 0.f;
            }
        }
        __syncthreads();
        // processed location features on the matrix pipe, then pre = keys + f . Wloc (the query joins in part 2)
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
    // ---- attention backward, part 2 (after hop B): train_step.h attention_bwd_kernel's arithmetic from d ctx on ----
    auto att_main = [&](int t) {
        const size_t tb = (size_t)t * B + ab;
        // d context of step t: LSTM 1's input part (GEMM 1 of this step) + the attention LSTM's input part of step t + 1
        // (GEMM 0 of step t + 1; its buffer is still zero at the last step), four K-slice partials each
        float dcx;
        {
            const int og = tid >> 1, ogrp = og >> 2, oj = og & 3, e2 = tid & 1;
            float v1[4], v0[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v1[k] = ld4(xs, a.oP1 + (unsigned)(((((ogrp * 4 + k) * MB * 16) + ab) * 4 * NC1 + oj * NC1 + 8 + e2) * 4));
                v0[k] = ld4(xs, a.oP0 + (unsigned)(((((ogrp * 4 + k) * MB * 16) + ab) * 4 * NC0 + oj * NC0 + 4 + e2) * 4));
            }
            dcx = ((v1[0] + v1[1]) + (v1[2] + v1[3])) + ((v0[0] + v0[1]) + (v0[2] + v0[3]));
        }
        s_dctx[tid] = dcx;
        if (ac == 0) a.dctx_all[(size_t)ab * a.dctx_row_ld + (size_t)t * EW + tid] = dcx;
        // softmax-backward dot: d ctx . ctx_t + a . d state
        float part = fmaf(dcx, ctx_v, part_gs);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) s_rd[wave] = part;
        __syncthreads();
        PZ_STAMP(12);
        // d alignments: da_j = d ctx . values_j + d state_j; wave w takes positions w, w + 8, ...
        {
            float dv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) dv[i] = s_dctx[lane + 64 * i];
#pragma unroll
            for (int jj = 0; jj < VN; ++jj) {
                const int j = wave + jj * NWAVE;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(dv[i], vv[jj][i], s);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
                if (lane == 0 && j < n_pos) s_da[j] = s + s_gs[p0 + j];
            }
            const float* vals = a.values + ((size_t)ab * T + p0) * EW;
            for (int j = wave + VN * NWAVE; j < n_pos; j += NWAVE) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s = fmaf(dv[i], vals[(size_t)j * EW + lane + 64 * i], s);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
                if (lane == 0) s_da[j] = s + s_gs[p0 + j];
            }
        }
        __syncthreads();
        if (tid < CHP) {
            float dot = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) dot += s_rd[w];
            const float de = (tid < n_pos && p0 + tid < len_b) ? s_al[p0 + tid] * (s_da[tid] - dot) : 0.f;
            s_de[tid] = de;
            if (tid < n_pos) a.de_all[tb * T + p0 + tid] = de;
        }
        __syncthreads();
        PZ_STAMP(13);
        // d u = d e * v * (1 - tanh^2), d v_a += d e * tanh, d q = sum over positions of d u
        {
            const float q_l = s_q[wave * 16 + c16];
            float dqs = 0.f;
            float* du_t = a.du_all + (tb * T + p0) * UA + wave * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < NRT) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int pos = 16 * r + 4 * kg + g;
                        const float de = s_de[pos];
                        const float th = tanhf(pre[r][g] + q_l);
                        const float du = de * va_l * (1.f - th * th);
                        dva_acc = fmaf(de, th, dva_acc);
                        dqs += du;
                        s_du[pos * DUS + wave * 16 + c16] = du;
                        if (pos < n_pos) du_t[(size_t)pos * UA] = du;
                    }
                }
            }
            dqs += __shfl_xor(dqs, 16); dqs += __shfl_xor(dqs, 32);
            if (kg == 0) s_dq[wave * 16 + c16] = dqs;
        }
        __syncthreads();
        // this chunk's partial d q as a frag16 operand of the query-layer product
        if (tid < UA / 4) st16(xs, a.oDQ + (unsigned)ac * a.dq_bytes + frag_off16(ab, tid, MB), *reinterpret_cast<const f32x4*>(s_dq + 4 * tid));
    };
    // ---- attention backward, part 3: what only the next step and the tape need (d f before hop D's arrive, over the drain of the
    //      dz stores; the partial d state in the hop's gap) ----
    auto att_df = [&](int t) {
        const size_t tb = (size_t)t * B + ab;
        // d f = d u . Wloc^T on the matrix pipe: wave w -> positions 16 (w >> 1) .., filters 16 (w & 1) ..
        {
            const int r = wave >> 1, ct = wave & 1;
            if (r < NRT) {
                f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
                const float* ur = s_du + (16 * r + c16) * DUS + 4 * kg;
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ur + 16 * kb);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], wdf[kb][j], d4, 0, 0, 0);
                }
                const int filt = 16 * ct + c16;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int pos = 16 * r + 4 * kg + g;
                    s_df[pos * 36 + filt] = d4[g];
                    if (pos < n_pos && filt < a.F) {
                        a.df_all[(tb * T + p0 + pos) * a.F + filt] = d4[g];
                        a.f_all[(tb * T + p0 + pos) * a.F + filt] = s_f[pos * 36 + filt];
                    }
                }
            }
        }
    };
    auto att_ds = [&](int t) {
        // partial d state_prev over the chunk + conv halo: ds[i] <-> position p0 - pad_l + i
        {
            const int i = tid >> 2, cg = tid & 3;
            float s = 0.f;
            if (i < NIP) {
                const int ja = max(0, i - a.KA + 1), jb = min(n_pos - 1, i);
                for (int j = ja; j <= jb; ++j)
#pragma unroll
                    for (int c = 0; c < 8; ++c) s = fmaf(s_cw[(i - j) * 33 + cg + 4 * c], s_df[j * 36 + cg + 4 * c], s);
            }
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
            if (i < NIP && cg == 0) st4(xs, a.oDS + (unsigned)(((((t & 1) * B + ab) * C + ac) * NIP + i) * 4), s);
        }
    };

    // ---- LSTM 2 backward of the last step: its raw-output gradient comes from the projections only ----
    if (tid < MB * 64) {
        const TapeRow r2 = tape_load(2, S - 1);
        const float g_out = eown ? a.dy2_all[((size_t)(S - 1) * B + erow) * (NWG * 4) + wg * 4 + euu] : 0.f;
        cell_bwd(2, S - 1, r2, g_out, 0.f, gc2, ghp2, a.oDZ2 + (unsigned)((S - 1) & 1) * a.dz_bytes);
    }
    hop_arrive(a.bar); ++epoch;
    if (!hop_wait(a.bar, 32u * epoch, s_flag)) alive = false;
    if (alive) { gemm2_part(S - 1, 0); gemm2_part(S - 1, 4); gemm2_publish(); }
    hop_arrive(a.bar); ++epoch;
    if (alive && !hop_wait(a.bar, 32u * epoch, s_flag)) alive = false;
    TapeRow r1{}, r2{};
    float dy2p = 0.f;
    if (tid < MB * 64) {
        r2 = tape_load(2, S - 2); r1 = tape_load(1, S - 1);
        dy2p = (eown && S > 1) ? a.dy2_all[((size_t)(S - 2) * B + erow) * (NWG * 4) + wg * 4 + euu] : 0.f;
    }

#ifdef PZ_STAMPS
    st_prev = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = S - 1; alive && t >= 0; --t) {
        fresh();
        // ================= GEMM 2 of this step and GEMM 1 of step t + 1 are complete and published: cell backwards =================
        TapeRow r0{};
        if (tid < MB * 64) {
            r0 = tape_load(0, t);
            const float dy1 = pread(a.oP2, NC2, euu);
            dh2p = pread(a.oP2, NC2, 4 + euu);
            dh1p = pread(a.oP1, NC1, 4 + euu);               // GEMM 1 of step t + 1 (zero at the last step)
            if (t > 0) cell_bwd(2, t - 1, r2, dy2p, ghp2 + dh2p, gc2, ghp2, a.oDZ2 + (unsigned)((t - 1) & 1) * a.dz_bytes);
            cell_bwd(1, t, r1, dy1, ghp1 + dh1p, gc1, ghp1, a.oDZ1);
        }
        // GEMM 0 of step t + 1 (its consumers -- d ctx in the attention, d h_att_prev in the query phase -- sit behind hop B):
        // first half here, over the drain of the dz stores; second half in the gap of hop A
        if (t < S - 1) { gemm0_first(); fresh(); }
        PZ_STAMP(1); hop_arrive(a.bar); ++epoch; fresh();
        if (t < S - 1) { gemm0_second(); fresh(); }
        PZ_STAMP(0);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh(); PZ_STAMP(2);

        // ================= GEMM 1: d [h_att | h1_prev | ctx] = dz1(t) . W1^T, this workgroup's partial tile =================
        {
            f32x4 acc1[CT1][MB];
#pragma unroll
            for (int ct = 0; ct < CT1; ++ct)
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) acc1[ct][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm2d_part<MB, CT1, 8>(acc1, rW1, (unsigned)((wg * 64 + wave * 8) * 256 * NC1), 256u * NC1, loff1, xs,
                                    a.oDZ1 + (unsigned)((ks * 64 + wave * 8) * MB * 1024), l16);
            publish_tiles(acc1, std::integral_constant<int, CT1>{}, a.oP1, 4 * NC1);
        }
        fresh();
        // first half of GEMM 2 of step t - 1 (dz2(t - 1) was published before hop A) over the drain of the partial-tile stores
        if (MB >= 2 && t > 0) { gemm2_part(t - 1, 0); fresh(); }
        PZ_STAMP(3); hop_arrive(a.bar); ++epoch; fresh();
        // gap B: attention pre-work
        if (MB < 2 && t > 0) { gemm2_part(t - 1, 0); fresh(); }
        // (dead values for the workgroups without an attention role: keeps these registers from being live around the loop)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[r][g] = 0.f;
#pragma unroll
        for (int jj = 0; jj < VN; ++jj)
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[jj][i] = 0.f;
        ctx_v = 0.f; part_gs = 0.f;
        if (isAtt) { att_pre(t); fresh(); }
        PZ_STAMP(9);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh(); PZ_STAMP(4);

        // this owner's columns of GEMM 1 (step t) and GEMM 0 (step t + 1): in flight during the attention
        float dhA1 = 0.f, dprev = 0.f;
        if (tid < MB * 64) {
            dhA1 = pread(a.oP1, NC1, euu);
            dhAp = pread(a.oP0, NC0, euu);
            if (euu == 2 && eown && t < S - 1) dprev = pread(a.oP0, NC0, 6);
        }
        // ================= attention backward =================
        if (isAtt) { att_main(t); fresh(); }
        PZ_STAMP(5); hop_arrive(a.bar); ++epoch; fresh();
        // gap C: second half of GEMM 2 of step t - 1, its partial tile published (read at the top of the next step, behind hop D)
        if (t > 0) { gemm2_part(t - 1, 4); gemm2_publish(); fresh(); }
        const f32x4 bq = ldw(rWQ, (unsigned)((kg * NCQ + min(c16, NCQ - 1)) * 16), (unsigned)((wg * 8 + wave) * 64 * NCQ));
        PZ_STAMP(10);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh(); PZ_STAMP(6);

        // ================= query layer: d h_att += (sum of the C partial d q) . Wq^T; attention LSTM backward =================
        {
            zero_acc();
            // d q of the row (tape of the query layer's weight gradient): workgroup b < B sums row b's partials (C is 4, 8 or 16)
            float dq_row = 0.f;
            const bool dq_owner = wg < B && tid < UA;
            const unsigned dq_off = frag_off16(min(wg, B - 1), (tid & (UA - 1)) >> 2, MB) + (unsigned)((tid & 3) * 4);
            for (int c0 = 0; c0 < C; c0 += 4) {
                f32x4 av[4][MB];
                float dv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb) av[k][rb] = ldx(xs, l16, a.oDQ + (unsigned)(c0 + k) * a.dq_bytes + (unsigned)((wave * MB + rb) * 1024));
                    if (dq_owner) dv[k] = ld4(xs, a.oDQ + (unsigned)(c0 + k) * a.dq_bytes + dq_off);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][rb][j], bq[j], acc[rb], 0, 0, 0);
                    dq_row += dv[k];
                }
            }
            put_red(acc);
            if (dq_owner) a.dq[((size_t)t * B + wg) * UA + tid] = dq_row;
            __syncthreads();
            fresh();
            if (tid < MB * 64) {
                const float dhq = colsum(euu);
                cell_bwd(0, t, r0, dhA1 + dhq, ghp0 + dhAp, gc0, ghp0, a.oDZ0);
                if (euu == 2 && eown && t < S - 1) {
                    // PreNet backward of step t + 1 through dropout(relu(.)) (modules.py:88-91): d pre = (y != 0) ? d y / keep : 0
                    const size_t pi = ((size_t)(t + 1) * B + erow) * 256 + wg;
                    a.dpre[pi] = a.xp_last[pi] != 0.f ? dprev * a.inv_keep : 0.f;
                }
            }
        }
        if (MB >= 2 && isAtt) { att_df(t); fresh(); }
        PZ_STAMP(7); hop_arrive(a.bar); ++epoch; fresh();
        if (MB < 2 && isAtt) { att_df(t); __syncthreads(); fresh(); }
        // gap D: the attention's partial d state_prev (read by the next step's pre-work, behind hop A);
        // tape rows of the next step's LSTM 2 / LSTM 1 backwards, in flight across the hop
        if (isAtt) { att_ds(t); fresh(); }
        PZ_STAMP(15);
        if (tid < MB * 64 && t > 0) {
            r2 = tape_load(2, t - 2); r1 = tape_load(1, t - 1);
            dy2p = (eown && t > 1) ? a.dy2_all[((size_t)(t - 2) * B + erow) * (NWG * 4) + wg * 4 + euu] : 0.f;
        }
        PZ_STAMP(11);
        if (!hop_wait(a.bar, 32u * epoch, s_flag)) { alive = false; break; }
        fresh(); PZ_STAMP(8);
    }
    // the last product: d prenet of step 0
    if (alive) {
        gemm0_first(); gemm0_second();
        hop_arrive(a.bar); ++epoch;
        if (hop_wait(a.bar, 32u * epoch, s_flag)) {
            fresh();
            if (tid < MB * 64 && euu == 2 && eown) {
                const size_t pi = (size_t)erow * 256 + wg;
                a.dpre[pi] = a.xp_last[pi] != 0.f ? pread(a.oP0, NC0, 6) * a.inv_keep : 0.f;
            }
        } else alive = false;
    }
#ifdef PZ_STAMPS
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 24; ++i) g_pz_stamps[wg][i] = st_acc[i];
    }
#endif
    fresh();
    // d v_a: this workgroup's sum over all steps of its chunk (rows of pva_part are added in order after the launch)
    {
        float s = dva_acc;
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        if (kg == 0) a.pva_part[(size_t)wg * UA + wave * 16 + c16] = isAtt ? s : 0.f;
    }
    if (!alive && tid == 0) atomicOr(reinterpret_cast<unsigned*>(a.err), 1u);
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward of EncoderV2's bidirectional ZoneoutLSTM (tacotron_v2.py:53-63, training branch modules.py:274-276) as ONE launch:
// the reverse-time mirror of pz::encoder_kernel.  2 directions x 64 workgroups; workgroup g of a direction owns hidden units
// 4g..4g+3: their state gradients stay in registers, the cell backward (with dynamic_rnn's length masking: dead steps pass the
// state gradient through) runs where the units live, dz [rows x 1024] goes to the other 63 workgroups as a frag16 operand, and
// the product d [x | h_prev] = dz . W^T is column-owner: this workgroup's 4 recurrent columns + 8 of the 512 input columns,
// the 48 KB tile resident in registers for the whole sequence.  One hop per step.
// ------------------------------------------------------------------------------------------------------------------------
struct EBArgs {
    const float* wT[2];                        // per direction [64][64 k-blocks][4][12][4] (pack_bwd_tiles_kernel)
    const float *z[2], *cprev[2], *cnew[2];    // forward tape per direction, step order: [T][B][1024], [T][B][256] x 2
    const float* dmem;                         // [B][T][512] d loss / d memory
    const int* lens;
    float* dzp[2];                             // [T][B][1024] gate gradients (weight-gradient GEMMs after the launch)
    float* dX[2];                              // [T][B][768]: the 512 input columns are written
    float* xs; unsigned xs_bytes;              // [2 directions][2 slots] dz operands, frag16 MB x 1024 (zeroed)
    unsigned* bar; int* err;                   // [2][NSHARD][32] hop counters (zeroed), error word
    int B, T, stream;
    unsigned seed;
    float zc, zh, fb;
};
constexpr int NCE = 12;

template <int MB>
__global__ __launch_bounds__(NT) void encoder_bwd_kernel(EBArgs a) {
    __shared__ f32x4 red[NWAVE * MB * 64];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / ENC_G, g = blockIdx.x - dir * ENC_G;
    const int B = a.B, T = a.T;
    const unsigned l16 = (unsigned)lane * 16u;
    const int kg = lane >> 4, c16 = lane & 15;
    if (tid == 0) s_abort = 0;
    const rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wT[dir]), 0, ENC_G * 64 * 64 * NCE, 0x00020000);
    const rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(a.xs, 0, (int)a.xs_bytes, 0x00020000);
    unsigned* bar = a.bar + dir * NSHARD * 32;
    f32x4 b[8];                                 // this wave's 8 k-blocks of the tile, resident
    wloadc<8>(b, rW, (unsigned)((g * 64 + wave * 8) * 64 * NCE), (unsigned)((kg * NCE + min(c16, NCE - 1)) * 16), 64u * NCE);
    const int erow = tid >> 2, euu = tid & 3;
    const bool eown = tid < MB * 64 && erow < B;
    const int len = eown ? min(a.lens[erow], T) : 0;
    const int unit = g * 4 + euu;
    const unsigned dzslot = (unsigned)(MB * 16 * 1024 * 4), ddir = (unsigned)dir * 2u * dzslot;
    float gc = 0.f, ghp = 0.f;
    struct Row { float zi, zj, zf, zo, cp, cn, go; };
    auto tape_load = [&](int s) -> Row {
        Row r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (eown && s >= 0 && s < len) {
            const size_t ti = (size_t)s * B + erow;
            const float* tz = a.z[dir] + ti * 1024;
            r.zi = tz[unit]; r.zj = tz[256 + unit]; r.zf = tz[512 + unit]; r.zo = tz[768 + unit];
            r.cp = a.cprev[dir][ti * 256 + unit]; r.cn = a.cnew[dir][ti * 256 + unit];
            const int pos = dir ? (len - 1 - s) : s;
            r.go = a.dmem[((size_t)erow * T + pos) * 512 + dir * 256 + unit];
        }
        return r;
    };
    // ZoneoutLSTMCell backward of step s for this thread's (row, unit); dhp: d h_prev of step s + 1 from this workgroup's columns
    auto cell = [&](int s, const Row& r, float dhp) {
        float dz4[4] = {0.f, 0.f, 0.f, 0.f};
        const float g_hs = ghp + dhp;
        if (eown && s < len) {
            const unsigned stream = (unsigned)(a.stream + 2 * dir);
            const float gi = sigmoidf_(r.zi), gj = tanhf(r.zj), gf = sigmoidf_(r.zf + a.fb), go = sigmoidf_(r.zo);
            const float mc = t2_u01(a.seed, stream, (unsigned)s, (unsigned)erow, (unsigned)unit) >= a.zc ? 1.f : 0.f;
            const float mh = t2_u01(a.seed, stream + 1u, (unsigned)s, (unsigned)erow, (unsigned)unit) >= a.zh ? 1.f : 0.f;
            const float tc = tanhf(r.cn);
            const float dhn = r.go + mh * g_hs;
            const float dcn = mc * gc + dhn * go * (1.f - tc * tc);
            gc = (1.f - mc) * gc + dcn * gf;
            ghp = (1.f - mh) * g_hs;
            dz4[0] = dcn * gj * gi * (1.f - gi); dz4[1] = dcn * gi * (1.f - gj * gj); dz4[2] = dcn * r.cp * gf * (1.f - gf); dz4[3] = dhn * tc * go * (1.f - go);
        } else {
            ghp = g_hs;                         // dead step (dynamic_rnn copied the state through): the state gradient passes
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (eown) a.dzp[dir][((size_t)s * B + erow) * 1024 + q * 256 + unit] = dz4[q];
            f32x4 v;
            v[0] = __shfl(dz4[q], (lane & ~3) + 0); v[1] = __shfl(dz4[q], (lane & ~3) + 1);
            v[2] = __shfl(dz4[q], (lane & ~3) + 2); v[3] = __shfl(dz4[q], (lane & ~3) + 3);
            if (eown && euu == 0) st16(xs, ddir + (unsigned)(s & 1) * dzslot + frag_off16(erow, q * ENC_G + g, MB), v);
        }
    };
    __syncthreads();
    unsigned epoch = 0;
    bool alive = true;
    if (tid < MB * 64) cell(T - 1, tape_load(T - 1), 0.f);
    hop_arrive(bar); ++epoch;
    for (int s = T - 1; s >= 0; --s) {
        Row r{};
        if (tid < MB * 64) r = tape_load(s - 1);            // in flight across the hop
        if (!hop_wait(bar, (unsigned)(ENC_G / NSHARD) * epoch, &s_abort)) { alive = false; break; }
        f32x4 acc[MB];
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        seg_mma<MB, 8>(acc, b, xs, ddir + (unsigned)(s & 1) * dzslot + (unsigned)(wave * 8 * MB * 1024), l16);
#pragma unroll
        for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
        __syncthreads();
        if (tid < MB * 64) {
            const int rb = erow >> 4, rr = erow & 15;
            const int reg = rr & 3, lbase = (rr >> 2) * 16;
            float dhp = 0.f, dx0 = 0.f, dx1 = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) {
                const f32x4* rw = red + (w * MB + rb) * 64 + lbase;
                dhp += rw[euu][reg]; dx0 += rw[4 + 2 * euu][reg]; dx1 += rw[5 + 2 * euu][reg];
            }
            if (eown) {
                float* dx = a.dX[dir] + ((size_t)s * B + erow) * 768 + g * 8 + 2 * euu;
                dx[0] = dx0; dx[1] = dx1;
            }
            if (s > 0) cell(s - 1, r, dhp);
        }
        if (s > 0) { hop_arrive(bar); ++epoch; }
    }
    if (!alive && tid == 0) atomicOr(reinterpret_cast<unsigned*>(a.err), 1u);
}

}  // namespace pz
