// HIP kernels of the Tacotron2 mel-generation path for gfx950 (MI355X, CDNA4).
//
// Data layouts
//  * "plain": row-major fp32, innermost dimension contiguous.
//  * "frag16": activation matrix X[rows, K] stored in the operand order of
//    v_mfma_f32_16x16x4_f32 so that one wave-instruction loads one contiguous 1 KiB piece:
//        float index = (((kb*MB + rb)*4 + kg)*16 + rr)*4 + j,   k = kb*16 + kg*4 + j,  row = rb*16 + rr
//    (MB = row blocks of 16 batch rows; K padded to a multiple of 16; pad rows / pad k are zero).
//    Lane l = kg*16 + rr reads the float4 at ((kb*MB + rb)*64 + l)*4.
//  * packed weights for the skinny (batch <= 64) GEMMs: per workgroup g a contiguous slab
//        [kb][lane = kg*16 + c][j]  = W[k = kb*16 + kg*4 + j][col(g, c)]
//    so the weight stream of a workgroup is one linear run of 1 KiB wave-loads, read exactly once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T2_SKINNY_THREADS 512
#define T2_SKINNY_WAVES 8
#define T2_MAX_SEG 3

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

// returns the step index of this launch or -1 when the launch must do nothing
__device__ __forceinline__ int step_index(const int* __restrict__ ctl, int step_off) {
    if (ctl == nullptr) return step_off;
    const int t = ctl[CTL_BASE] + step_off;
    const int fin = ctl[CTL_FINAL];
    if (t >= ctl[CTL_MAX] || (fin != 0 && t >= fin)) return -1;
    return t;
}

struct SegDesc {
    const float* base;   // frag16 buffer
    int kb;              // k-blocks of 16
    int pp_stride;       // floats; buffer used = base + (t & 1) * pp_stride   (ping-pong state)
    int t_stride;        // floats; buffer used += t * t_stride                (time-indexed input)
    int pp_flip;         // 1: use ((t+1)&1) instead of (t&1)
};

struct SkinnyCore {
    const float* w[2];        // packed weights per direction (blockIdx.y)
    const float* bias[2];     // packed bias [n_wg][16] (may be null)
    SegDesc seg[2][T2_MAX_SEG];
    int nseg;
    int kbtot;
    const int* ctl;
    int step_off;
    int B;                    // valid rows
};

struct LstmEpi {
    int H;                    // hidden units
    float* c[2];              // plain [MB*16][Hld]
    int Hld;
    float* hstate[2];         // frag16 ping-pong: read (t&1), write ((t+1)&1)
    int hstate_pp_stride;
    float* hraw[2];           // frag16 raw output (layer output), may be null
    float zc, zh, fb;
    const int* lens;          // optional sequence lengths (encoder), device int32 [B]
    float* plain_out;         // optional plain output [B][T][ld]
    int plain_bs, plain_ts;   // batch / time strides (floats)
    int plain_off[2];         // column offset per direction
    int reverse[2];           // direction writes to position len-1-t
};

struct DenseEpi {
    int N;                    // valid columns
    int relu;
    float* out_frag;          // frag16 output (k = col - frag_col0), may be null
    int frag_col0;
    float* out_plain;         // plain [row][ld] (+ t * plain_t_stride), may be null
    int plain_ld;
    long long plain_t_stride;
    int plain_ncol;           // columns < plain_ncol go to out_plain
    // stop token handling (projection kernel): column stop_col holds the stop logit
    int stop_col;             // -1: none
    float* out_stop;          // [row * stop_ld + t]
    int stop_ld;
    int stop_t_mul;           // 1: index row*stop_ld + t, 0: index row*stop_ld
    int* ctl_w;               // writable ctl (same block as core.ctl)
};

// ------------------------------------------------------------------------------------------------
// skinny GEMM core: acc[rb] (16x16 tile, rows = batch rows of block rb, cols = this workgroup's 16
// columns) += X[:, k-range of this wave] . W[k-range, cols].  The K range is split over the 8 waves
// of the workgroup; every wave streams its part of the packed weight slab once, straight to
// registers (no LDS round trip: the operand is not shared between waves).
// ------------------------------------------------------------------------------------------------
template <int MB>
__device__ __forceinline__ void skinny_core(const SkinnyCore& a, int dir, int t, f32x4 (&acc)[MB]) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int per = (a.kbtot + T2_SKINNY_WAVES - 1) / T2_SKINNY_WAVES;
    const int lo = wave * per;
    const int hi = min(a.kbtot, lo + per);
    const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.w[dir]) + (size_t)blockIdx.x * a.kbtot * 64 + lane;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kb0 = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const SegDesc& sd = a.seg[dir][s];
        const int s_lo = max(lo, kb0), s_hi = min(hi, kb0 + sd.kb);
        if (s_lo < s_hi) {
            const int par = sd.pp_flip ? ((t + 1) & 1) : (t & 1);
            const f32x4* __restrict__ ap = reinterpret_cast<const f32x4*>(
                sd.base + (size_t)par * sd.pp_stride + (size_t)t * sd.t_stride) + lane;
            int kb = s_lo;
            constexpr int UN = (MB <= 2) ? 8 : 4;
            for (; kb + UN <= s_hi; kb += UN) {
                f32x4 b[UN];
                f32x4 av[UN][MB];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    b[u] = __builtin_nontemporal_load(wp + (size_t)(kb + u) * 64);
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb) av[u][rb] = ap[(size_t)((kb + u - kb0) * MB + rb) * 64];
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int rb = 0; rb < MB; ++rb)
                            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][rb][j], b[u][j], acc[rb], 0, 0, 0);
            }
            for (; kb < s_hi; ++kb) {
                const f32x4 b = __builtin_nontemporal_load(wp + (size_t)kb * 64);
                f32x4 av[MB];
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) av[rb] = ap[(size_t)((kb - kb0) * MB + rb) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < MB; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rb][j], b[j], acc[rb], 0, 0, 0);
            }
        }
        kb0 += sd.kb;
    }
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
// LSTM step: gates = [x-segments | h_prev] . kernel + bias for 4 hidden units per workgroup (columns
// gate*4 + u), fused with the LSTMCell pointwise math and the inference-mode
// zoneout blend of ZoneoutLSTMCell (reference tacotron/modules.py:261-283).
// ------------------------------------------------------------------------------------------------
template <int MB, bool ENC>   // ENC: encoder BiLSTM instance (own name in profiles; same code)
__global__ __launch_bounds__(T2_SKINNY_THREADS) void lstm_step_kernel(SkinnyCore a, LstmEpi e) {
    const int t = step_index(a.ctl, a.step_off);
    if (t < 0) return;
    const int dir = blockIdx.y;
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    f32x4 acc[MB];
    skinny_core<MB>(a, dir, t, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    __syncthreads();
    const int idx = threadIdx.x;
    if (idx >= MB * 16 * 4) return;
    const int row = idx >> 2, uu = idx & 3;
    const int unit = blockIdx.x * 4 + uu;
    if (row >= a.B || unit >= e.H) return;
    const float* bias = a.bias[dir] + (size_t)blockIdx.x * 16;
    const float zi = red_sum<MB>(red, row, 0 * 4 + uu) + bias[0 * 4 + uu];
    const float zj = red_sum<MB>(red, row, 1 * 4 + uu) + bias[1 * 4 + uu];
    const float zf = red_sum<MB>(red, row, 2 * 4 + uu) + bias[2 * 4 + uu];
    const float zo = red_sum<MB>(red, row, 3 * 4 + uu) + bias[3 * 4 + uu];
    float* cptr = e.c[dir] + (size_t)row * e.Hld + unit;
    const size_t fi = frag_idx(row, unit, MB);
    const float* hin = e.hstate[dir] + (size_t)(t & 1) * e.hstate_pp_stride;
    float* hout = e.hstate[dir] + (size_t)((t + 1) & 1) * e.hstate_pp_stride;
    const float cp = *cptr;
    const float hp = hin[fi];
    const float cn = sigmoidf_(zf + e.fb) * cp + sigmoidf_(zi) * tanhf(zj);
    float hn = sigmoidf_(zo) * tanhf(cn);
    float cs = (1.0f - e.zc) * cn + e.zc * cp;
    float hs = (1.0f - e.zh) * hn + e.zh * hp;
    bool live = true;
    int len = 0;
    if (e.lens != nullptr) {
        len = e.lens[row];
        live = t < len;
        if (!live) { cs = cp; hs = hp; hn = 0.f; }   // dynamic_rnn: zero output, state copied through
    }
    *cptr = cs;
    hout[fi] = hs;
    if (e.hraw[dir] != nullptr) e.hraw[dir][fi] = hn;
    if (e.plain_out != nullptr && live) {
        const int tpos = e.reverse[dir] ? (len - 1 - t) : t;
        e.plain_out[(size_t)row * e.plain_bs + (size_t)tpos * e.plain_ts + e.plain_off[dir] + unit] = hn;
    }
}

// ------------------------------------------------------------------------------------------------
// Dense step: y = act(x . W + b) for 16 columns per workgroup.  Used for the PreNet layers
// (modules.py:88-91), the attention query layer (tacotron_v2.py:122) and the mel / stop projections
// (rnn_wrappers.py:105-109) together with the stop rule of StopTokenBasedInferenceHelper
// (helpers.py:90-94) evaluated on the device.
// ------------------------------------------------------------------------------------------------
template <int MB>
__global__ __launch_bounds__(T2_SKINNY_THREADS) void dense_step_kernel(SkinnyCore a, DenseEpi e) {
    const int t = step_index(a.ctl, a.step_off);
    if (t < 0) return;
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    __shared__ int s_notstop;
    f32x4 acc[MB];
    skinny_core<MB>(a, 0, t, acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    if (threadIdx.x == 0) s_notstop = 0;
    __syncthreads();
    const bool has_stop = e.stop_col >= 0 && (e.stop_col >> 4) == (int)blockIdx.x;
    for (int idx = threadIdx.x; idx < MB * 16 * 16; idx += T2_SKINNY_THREADS) {
        const int row = idx >> 4, c = idx & 15;
        const int col = blockIdx.x * 16 + c;
        if (row >= a.B || col >= e.N) continue;
        float v = red_sum<MB>(red, row, c);
        if (a.bias[0] != nullptr) v += a.bias[0][col];
        if (e.relu) v = fmaxf(v, 0.f);
        if (col == e.stop_col) {
            e.out_stop[(size_t)row * e.stop_ld + (size_t)t * e.stop_t_mul] = v;
            if (!(v > 0.f)) atomicOr(&s_notstop, 1);       // sigmoid(v) > 0.5  <=>  v > 0
            continue;
        }
        if (e.out_plain != nullptr && col < e.plain_ncol)
            e.out_plain[(size_t)row * e.plain_ld + (size_t)t * e.plain_t_stride + col] = v;
        if (e.out_frag != nullptr && col >= e.frag_col0) e.out_frag[frag_idx(row, col - e.frag_col0, MB)] = v;
    }
    if (has_stop) {
        __syncthreads();
        // finished = reduce_all(stop & (time > min_iters)); outputs of the finishing step are kept,
        // so the decode emits t + 1 steps.  Readers at step t see 0 or t + 1: both let step t run.
        if (threadIdx.x == 0 && e.ctl_w != nullptr && e.ctl_w[CTL_STOP_EN] != 0 && s_notstop == 0 &&
            t > e.ctl_w[CTL_MINIT] && e.ctl_w[CTL_FINAL] == 0)
            e.ctl_w[CTL_FINAL] = t + 1;
    }
}

// ------------------------------------------------------------------------------------------------
// Location-sensitive attention step for one batch row per workgroup
// (reference tacotron/tacotron_v2.py:118-141, _location_sensitive_score :66-78; BahdanauAttention
// masking / softmax and AttentionWrapper context: SURVEY.md App. A5/A6).
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* keys;      // [B][T][U]
    const float* values;    // [B][T][E] (length-masked memory)
    const int* lens;        // [B] (device int32)
    float* att_state;       // [B][T] in/out: cumulative alignments (or previous alignments)
    float* alignments;      // [B][T] out
    float* history;         // [cap][B][T] or null
    const float* q;         // [B][U] processed query
    float* ctx_frag;        // frag16 [E] out
    float* ctx_plain;       // [B][E] out (state readout), may be null
    const float* conv_w;    // [K][F]
    const float* conv_b;    // [F]
    const float* wloc;      // [F][U]
    const float* v_a;       // [U]
    const float* b_a;       // [U]
    int B, T, U, E, F, K, MB, cumulative;
    const int* ctl;
    int step_off;
};

#define T2_ATTN_THREADS 512

__global__ __launch_bounds__(T2_ATTN_THREADS) void attention_step_kernel(AttnArgs a) {
    const int t_step = step_index(a.ctl, a.step_off);
    if (t_step < 0) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int T = a.T, U = a.U, F = a.F, K = a.K, E = a.E;
    const int pad_l = (K - 1) / 2;
    // LDS carve
    float* s_prev = smem;                       // [T + K] with halo (index i <-> position i - pad_l)
    float* s_f = s_prev + ((T + K + 3) & ~3);   // [T][F]
    float* s_wloc = s_f + ((T * F + 3) & ~3);   // [F][U]
    float* s_e = s_wloc + ((F * U + 3) & ~3);   // [T] energies -> alignments
    float* s_red = s_e + ((T + 3) & ~3);        // [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NW = T2_ATTN_THREADS / 64;
    const int len = a.lens[b];
    float* state = a.att_state + (size_t)b * T;
    for (int i = tid; i < T + K; i += T2_ATTN_THREADS) {
        const int p = i - pad_l;
        s_prev[i] = (p >= 0 && p < T) ? state[p] : 0.f;
    }
    for (int i = tid; i < F * U; i += T2_ATTN_THREADS) s_wloc[i] = a.wloc[i];
    __syncthreads();
    // location features f[t][c] = sum_j w[j][c] * prev[t + j - pad_l] + b[c]   (:128-130)
    for (int i = tid; i < T * F; i += T2_ATTN_THREADS) {
        const int tt = i / F, c = i - tt * F;
        float s = 0.f;
        for (int j = 0; j < K; ++j) s = fmaf(a.conv_w[j * F + c], s_prev[tt + j], s);
        s_f[i] = s + a.conv_b[c];
    }
    __syncthreads();
    // energies: one wave per position, lanes over attention units   (:131-133, :66-78)
    const float* keys = a.keys + (size_t)b * T * U;
    const float* q = a.q + (size_t)b * U;
    for (int tt = wave; tt < T; tt += NW) {
        float part = 0.f;
        const float* ft = s_f + tt * F;
        for (int u = lane; u < U; u += 64) {
            float pf = 0.f;
            for (int c = 0; c < F; ++c) pf = fmaf(ft[c], s_wloc[c * U + u], pf);
            part += a.v_a[u] * tanhf(keys[(size_t)tt * U + u] + q[u] + pf + a.b_a[u]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if (lane == 0) s_e[tt] = part;
    }
    __syncthreads();
    // masked softmax over positions < len (score mask value -inf -> probability exactly 0)
    float m = -INFINITY;
    for (int i = tid; i < len; i += T2_ATTN_THREADS) m = fmaxf(m, s_e[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = s_red[0];
    for (int w = 1; w < NW; ++w) m = fmaxf(m, s_red[w]);
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < T; i += T2_ATTN_THREADS) {
        const float ex = (i < len) ? expf(s_e[i] - m) : 0.f;
        s_e[i] = ex;
        sum += ex;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) s_red[wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < NW; ++w) sum += s_red[w];
    float* al = a.alignments + (size_t)b * T;
    float* hist = a.history ? a.history + ((size_t)t_step * a.B + b) * T : nullptr;
    for (int i = tid; i < T; i += T2_ATTN_THREADS) {
        const float p = s_e[i] / sum;
        s_e[i] = p;
        al[i] = p;
        if (hist) hist[i] = p;
        state[i] = a.cumulative ? (p + s_prev[i + pad_l]) : p;      // :137-140
    }
    __syncthreads();
    // context = alignments . values   (AttentionWrapper._compute_attention)
    const float* vals = a.values + (size_t)b * T * E;
    for (int c = tid; c < E; c += T2_ATTN_THREADS) {
        float s0 = 0.f, s1 = 0.f;
        int i = 0;
        for (; i + 1 < len; i += 2) {
            s0 = fmaf(s_e[i], vals[(size_t)i * E + c], s0);
            s1 = fmaf(s_e[i + 1], vals[(size_t)(i + 1) * E + c], s1);
        }
        if (i < len) s0 = fmaf(s_e[i], vals[(size_t)i * E + c], s0);
        const float s = s0 + s1;
        a.ctx_frag[frag_idx(b, c, a.MB)] = s;
        if (a.ctx_plain) a.ctx_plain[(size_t)b * E + c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// small data movement kernels
// ------------------------------------------------------------------------------------------------
// plain [B][ld] (+ offset) -> frag16; used to feed decoder inputs (helpers.py:86,182,187-188).
struct FeedArgs {
    const float* src;         // plain base
    long long row_stride;     // floats between batch rows
    long long t_stride;       // floats between steps: source of step t is src + (t-1) * t_stride
    float* dst;               // frag16 [K]
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
        a.dst[frag_idx(row, k, a.MB)] = v;
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
                                   int B, int T, int C, int MB) {
    const int KP = (C + 15) & ~15;
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
__global__ void frag_to_plain_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int K, int MB) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * K; i += gridDim.x * blockDim.x) {
        const int row = i / K, k = i - row * K;
        dst[i] = src[frag_idx(row, k, MB)];
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
    const int Ktot = a.ktaps * a.Cin;
    const int nk = (Ktot + CG_BK - 1) / CG_BK;

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

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
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
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
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
