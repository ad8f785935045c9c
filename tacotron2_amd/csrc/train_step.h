// Training step of the V2 path (BASELINE.json configs[3]): the teacher-forced decoder forward in training-mode
// arithmetic (PreNet dropout, sampled zoneout) that records a tape, and the backward pass over it.
// Included by t2mi.hip.
#pragma once

// ------------------------------------------------------------------------------------------------
// backward kernels (plain [B][.] gradients; GEMM operands additionally in frag16)
// ------------------------------------------------------------------------------------------------
// d(loss)/d(mel, stop) of models.py:58-61: loss = spec_loss + binary_loss, each sum(l * w) / count_nonzero(w broadcast).
// Writes dY[t][b][M*r + 1] (plain, the tape of the projection's weight gradient) and the same rows as per-step frag16 operands.  n_spec / n_bin: device scalars
// (out3[1] of t2_spec_loss / t2_binary_loss).
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ mel, const float* __restrict__ target,
                                                        const float* __restrict__ spec_mask, const float* __restrict__ stop,
                                                        const float* __restrict__ done, const float* __restrict__ bin_mask,
                                                        const float* __restrict__ n_spec, const float* __restrict__ n_bin,
                                                        float* __restrict__ dY, float* __restrict__ dYf, long long frag_stride, int MB,
                                                        int B, int S, int M, int r) {
    const int N = M * r + 1;
    const long long n = (long long)S * B * N;
    const float is = n_spec[0] > 0.f ? 1.0f / n_spec[0] : 0.f, ib = n_bin[0] > 0.f ? 1.0f / n_bin[0] : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % N);
        const long long tb = i / N;
        const int b = (int)(tb % B), t = (int)(tb / B);
        float g;
        if (c < M * r) {
            const int frame = t * r + c / M;
            const size_t idx = ((size_t)b * S * r + frame) * M + c % M;
            const float diff = mel[idx] - target[idx];
            g = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * spec_mask[(size_t)b * S * r + frame] * is;
        } else {
            const float x = stop[(size_t)b * S + t];
            g = (sigmoidf_(x) - done[(size_t)b * S + t]) * bin_mask[(size_t)b * S + t] * ib;
        }
        dY[i] = g;
        dYf[(size_t)t * frag_stride + frag_idx(b, c, MB)] = g;     // frag16 operand of the step's projection backward
    }
}

// plain [B][K] (row stride ld) -> frag16 operand (rows >= B and k >= K stay zero: buffer zeroed once)
__global__ __launch_bounds__(256) void plain_to_frag_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, int B, int K, int MB) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * K; i += gridDim.x * blockDim.x) {
        const int row = i / K, k = i - row * K;
        dst[frag_idx(row, k, MB)] = src[(size_t)row * ld + k];
    }
}

// ZoneoutLSTMCell (training branch) backward, one thread per (row, unit).
//   g_out = g_out0 + g_out1 : gradient of the raw output h
//   g_hs  = g_hs0 + g_hs1   : gradient of the zoned state h AFTER this step (from step t + 1), null at the last step
//   g_c                      : gradient of the zoned cell state AFTER this step, in / out (becomes that of step t - 1)
// Writes dz (plain tape [B][4H] + frag16 operand), g_hp (the part of d/dh_prev that bypasses the cell: (1 - mask_h) g_hs).
struct LstmBwdArgs {
    const float *z, *cprev, *cnew;            // tape slices of step t: [B][4H], [B][H], [B][H]
    const float* g_out0; int ld_out0;
    const float* g_out1; int ld_out1;
    const float* g_hs0; int ld_hs0;
    const float* g_hs1; int ld_hs1;
    float* g_c;                               // [B][H]
    float* g_hp;                              // [B][H]
    float* dz_plain;                          // [B][4H]
    float* dz_frag;
    int B, H, MB, t, stream;
    uint32_t seed;
    float zc, zh, fb;
    int first;                                // 1: last time step (no gradient from the future)
    const int* lens;                          // optional [B]: rows with t >= len copied their state through (dynamic_rnn)
};

// one (row, unit) element; g_extra: a part of the raw-output gradient the caller holds in a register
__device__ __forceinline__ void lstm_bwd_elem(const LstmBwdArgs& a, int row, int u, float g_extra, bool extra_is_state = false) {
    const int i = row * a.H + u;
    const float* z = a.z + (size_t)row * 4 * a.H;
    const float gi = sigmoidf_(z[u]), gj = tanhf(z[a.H + u]), gf = sigmoidf_(z[2 * a.H + u] + a.fb), go = sigmoidf_(z[3 * a.H + u]);
    const float cp = a.cprev[i], cn = a.cnew[i];
    const float mc = t2_u01(a.seed, a.stream, a.t, row, u) >= a.zc ? 1.f : 0.f;
    const float mh = t2_u01(a.seed, a.stream + 1, a.t, row, u) >= a.zh ? 1.f : 0.f;
    float g_out = extra_is_state ? 0.f : g_extra, g_hs = 0.f;
    if (a.g_out0) g_out += a.g_out0[(size_t)row * a.ld_out0 + u];
    if (a.g_out1) g_out += a.g_out1[(size_t)row * a.ld_out1 + u];
    if (!a.first) {
        if (a.g_hs0) g_hs += a.g_hs0[(size_t)row * a.ld_hs0 + u];
        if (a.g_hs1) g_hs += a.g_hs1[(size_t)row * a.ld_hs1 + u];
        if (extra_is_state) g_hs += g_extra;
    }
    const float g_c = a.first ? 0.f : a.g_c[i];
    if (a.lens != nullptr && a.t >= a.lens[row]) {          // dead step: state passes through, no output, no gate gradient
        a.g_c[i] = g_c;
        a.g_hp[i] = g_hs;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            a.dz_plain[(size_t)row * 4 * a.H + g * a.H + u] = 0.f;
            a.dz_frag[frag_idx(row, g * a.H + u, a.MB)] = 0.f;
        }
        return;
    }
    const float tc = tanhf(cn);
    const float dhn = g_out + mh * g_hs;
    const float dcn = mc * g_c + dhn * go * (1.f - tc * tc);
    a.g_c[i] = (1.f - mc) * g_c + dcn * gf;
    a.g_hp[i] = (1.f - mh) * g_hs;
    const float dz4[4] = {dcn * gj * gi * (1.f - gi), dcn * gi * (1.f - gj * gj), dcn * cp * gf * (1.f - gf), dhn * tc * go * (1.f - go)};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        a.dz_plain[(size_t)row * 4 * a.H + g * a.H + u] = dz4[g];
        a.dz_frag[frag_idx(row, g * a.H + u, a.MB)] = dz4[g];
    }
}

__global__ __launch_bounds__(256) void lstm_bwd_kernel(LstmBwdArgs a) {
    const int n = a.B * a.H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int row = i / a.H;
        lstm_bwd_elem(a, row, i - row * a.H, 0.f);
    }
}

// Data-gradient GEMM of the backward recurrence, y[B][N] = x_frag[B][K] . W^T (skinny core, 16 columns per workgroup), with
// the elementwise stage that consumes y fused into the epilogue:
//   mode 1: ZoneoutLSTMCell backward of the layer whose raw-output gradient the columns [0, lb.H) are (g_extra = y);
//   mode 3: ZoneoutLSTMCell backward of the SAME layer's previous step, whose state gradient the columns [col0, col0 + lb.H)
//           are (encoder BiLSTM: d h_prev of step s is d state of step s - 1);
//   mode 2: PreNet backward through dropout(relu(.)) on the columns [0, pn_P): d = (out != 0) ? y / keep : 0.
struct BwdEpi {
    int N;
    float* out_plain; int plain_ld;          // y itself, may be null
    int mode, col0;
    LstmBwdArgs lb;
    int mode_b, col0_b;                      // mode_b == 3: a second fused cell backward, of the layer whose state gradient the
    LstmBwdArgs lb2;                         // columns [col0_b, col0_b + lb2.H) are (decoder LSTM 2's previous step)
    const float* pn_y; float pn_inv_keep; float *pn_plain, *pn_frag; int pn_P;
};

// NPW > 0: the statically unrolled walk of skinny_core (counted vmcnt waits).
// SPLIT: the K range is cut in two halves over blockIdx.y so that the K = 4H products of the recurrence (N / 16 = 112..160
// column tiles) occupy the whole chip: both workgroups of a tile store their partial tile write-through (sc1), drain, and take
// a ticket (one agent-scope atomic add); the one whose add came last reads the other half with sc1 loads, runs the epilogue
// on the sum (two addends: order-independent, deterministic) and clears the ticket for the next launch.
// (MI355X_MICROARCH.md, "Valid forms", sc1 table row 1: one unsharded counter, the last adder told by the returned value.)
#define T2_SK_MAX_TILES 256
template <int MB, int NPW = 0, bool SPLIT = false>
__global__ __launch_bounds__(T2_SKINNY_THREADS) void bwd_dense_kernel(SkinnyCore a, BwdEpi e, float* sk_part, unsigned* sk_ticket) {
    __shared__ f32x4 red[T2_SKINNY_WAVES * MB * 64];
    __shared__ __attribute__((aligned(16))) float s_tile[SPLIT ? MB * 256 : 4];
    __shared__ unsigned s_old;
    f32x4 acc[MB];
    const int wg = blockIdx.x;
    if constexpr (SPLIT) {
        const int hk = a.kbtot / 2, half = (int)blockIdx.y;
        a.kbtot = hk; a.kb_w0 += half * hk; a.seg[0][1].koff += half * hk * 16;
    }
    const int t = skinny_core<MB, NPW>(a, 0, wg, acc, [](int) {});
    if (t < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    __syncthreads();
    if constexpr (SPLIT) {
        const int half = (int)blockIdx.y;
        for (int idx = threadIdx.x; idx < MB * 256; idx += T2_SKINNY_THREADS) s_tile[idx] = red_sum<MB>(red, idx >> 4, idx & 15);
        __syncthreads();
        const pz::rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(sk_part + (size_t)wg * 2 * MB * 256, 0, 2 * MB * 256 * 4, 0x00020000);
        if (threadIdx.x < MB * 64) pz::st16(rp, (unsigned)((half * MB * 256 + threadIdx.x * 4) * 4), *reinterpret_cast<const f32x4*>(s_tile + threadIdx.x * 4));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_old = __hip_atomic_fetch_add(sk_ticket + wg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_old == 0u) return;                           // the other half has not arrived yet: it will finish the tile
        if (threadIdx.x == 0) __hip_atomic_store(sk_ticket + wg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < MB * 64) {
            const f32x4 o = pz::ld16(rp, (unsigned)(((half ^ 1) * MB * 256 + threadIdx.x * 4) * 4));
            f32x4* mine = reinterpret_cast<f32x4*>(s_tile + threadIdx.x * 4);
            *mine = *mine + o;
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < MB * 16 * 16; idx += T2_SKINNY_THREADS) {
        const int row = idx >> 4, c = idx & 15;
        const int col = wg * 16 + c;
        if (row >= a.B || col >= e.N) continue;
        float v;
        if constexpr (SPLIT) v = s_tile[idx]; else v = red_sum<MB>(red, row, c);
        if (e.out_plain != nullptr) e.out_plain[(size_t)row * e.plain_ld + col] = v;
        if (e.mode_b == 3 && col >= e.col0_b && col < e.col0_b + e.lb2.H) lstm_bwd_elem(e.lb2, row, col - e.col0_b, v, true);
        if (e.mode == 1) {
            if (col < e.lb.H) lstm_bwd_elem(e.lb, row, col, v);
        } else if (e.mode == 3) {
            if (col >= e.col0 && col < e.col0 + e.lb.H) lstm_bwd_elem(e.lb, row, col - e.col0, v, true);
        } else if (e.mode == 2) {
            if (col < e.pn_P) {
                const size_t i = (size_t)row * e.pn_P + col;
                const float dd = e.pn_y[i] != 0.f ? v * e.pn_inv_keep : 0.f;
                e.pn_plain[i] = dd;
                e.pn_frag[frag_idx(row, col, MB)] = dd;
            }
        }
    }
}

// PreNet backward through dropout(relu(.)): y = relu(pre) / keep * mask, so dy/dpre = (y != 0) / keep.
// g = g0 (row stride ld0); writes plain tape [B][N] and the frag16 operand of the next GEMM.
__global__ __launch_bounds__(256) void prenet_bwd_kernel(const float* __restrict__ g0, int ld0, const float* __restrict__ y,
                                                         float inv_keep, float* __restrict__ d_plain, float* __restrict__ d_frag,
                                                         int B, int N, int MB) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * N; i += gridDim.x * blockDim.x) {
        const int row = i / N, c = i - row * N;
        const float v = y[i] != 0.f ? g0[(size_t)row * ld0 + c] * inv_keep : 0.f;
        d_plain[i] = v;
        if (d_frag != nullptr) d_frag[frag_idx(row, c, MB)] = v;
    }
}

// column sums of a [rows][N] matrix (bias gradients), fixed order: stage 1 sums gridDim.y row ranges (64 columns x 4 row
// lanes per block, double accumulators), stage 2 (batch_sum_kernel) adds the range partials in order
#define T2_CS_CHUNKS 64        // row ranges of a wide matrix; narrow ones get more (col_sum)
#define T2_CS_PART_FLOATS (64 * 4096)
#define T2_BN_CHUNKS 32      // row ranges of the batch-norm reductions (bn_stats_kernel, bn_bwd_reduce_kernel)
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, long long rows, int N, float* __restrict__ part) {
    __shared__ double s1[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const long long per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = rows < r0 + per ? rows : r0 + per;
    double s = 0.0;
    if (c < N)
        for (long long r = r0 + rl; r < r1; r += 4) s += (double)x[r * N + c];
    s1[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < N) part[(size_t)blockIdx.y * N + c] = (float)(s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl]);
}

// sum over the batch of per-row partial gradients part[b][n] -> out[n]
// (the rows are added in increasing order; eight loads are in flight per round: a serial chain of dependent loads made the
// tall, narrow sums -- 2048 partials of 128 columns -- take 0.4 ms each)
__global__ __launch_bounds__(256) void batch_sum_kernel(const float* __restrict__ part, int B, long long n, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int b = 0;
        for (; b + 8 <= B; b += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(b + k) * n + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; b < B; ++b) s += part[(size_t)b * n + i];
        out[i] = s;
    }
}


// d (location convolution kernel)[k][f] = sum over all (step, row) pairs and positions j of state_prev[j - pad_l + k] * df[j][f]
// (tacotron_v2.py:128-130 backward; taps outside [0, T) read zero).  Thread = one filter x four consecutive taps: the attention
// state slides through a four-register window, so a position costs two LDS reads for four FMAs.  The blocks stride over the
// S * B tape rows and stage a row's attention state (with the halo) and its df tile in LDS, 128 positions at a time;
// part[block][K][F] is then summed in block order (batch_sum_kernel): deterministic.
#define T2_LCW_POS 128
__global__ __launch_bounds__(256) void loc_conv_wgrad_kernel(const float* __restrict__ state_prev, const float* __restrict__ df, long long n_rows,
                                                              int T, int K, int F, float* __restrict__ part) {
    __shared__ float s_sp[T2_LCW_POS + 36];
    __shared__ float s_dfl[T2_LCW_POS * 33];
    const int tid = threadIdx.x, kq = tid >> 5, f = tid & 31, pad_l = (K - 1) / 2, k0 = 4 * kq;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const float* sp = state_prev + r * T;
        const float* dr = df + r * (long long)T * F;
        for (int j0 = 0; j0 < T; j0 += T2_LCW_POS) {
            const int nj = min(T2_LCW_POS, T - j0);
            __syncthreads();
            for (int i = tid; i < nj + 35; i += 256) { const int p = j0 - pad_l + i; s_sp[i] = (p >= 0 && p < T) ? sp[p] : 0.f; }
            for (int i = tid; i < nj * F; i += 256) { const int j = i / F, c = i - j * F; s_dfl[j * 33 + c] = dr[(long long)j0 * F + i]; }
            __syncthreads();
            if (f < F) {
                float w0 = s_sp[k0], w1 = s_sp[k0 + 1], w2 = s_sp[k0 + 2];
#pragma unroll 4
                for (int j = 0; j < nj; ++j) {
                    const float w3 = s_sp[j + k0 + 3], dv = s_dfl[j * 33 + f];
                    acc[0] = fmaf(w0, dv, acc[0]); acc[1] = fmaf(w1, dv, acc[1]); acc[2] = fmaf(w2, dv, acc[2]); acc[3] = fmaf(w3, dv, acc[3]);
                    w0 = w1; w1 = w2; w2 = w3;
                }
            }
        }
    }
    if (f < F)
#pragma unroll
        for (int q = 0; q < 4; ++q) if (k0 + q < K) part[(size_t)blockIdx.x * K * F + (k0 + q) * F + f] = acc[q];
}

// d values[b][j][:] = sum_t a_t[b][j] * d ctx_t[b][:] (AttentionWrapper._compute_attention backward): block = (row b, 16 positions),
// thread = context column; the alignments of 64 steps x 16 positions are staged in LDS, steps are added in order
__global__ __launch_bounds__(512) void dvalues_kernel(const float* __restrict__ hist, const float* __restrict__ dctx, int B, int T, int S, int E,
                                                      float* __restrict__ dvalues) {
    __shared__ float s_a[64 * 16];
    const int b = blockIdx.y, j0 = blockIdx.x * 16, nj = min(16, T - j0), e = blockIdx.z * 512 + threadIdx.x;
    float acc[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) acc[jj] = 0.f;
    const float* dc = dctx + (size_t)b * S * E + e;
    for (int t0 = 0; t0 < S; t0 += 64) {
        const int nt = min(64, S - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 16; i += 512) {
            const int tt = i >> 4, jj = i & 15;
            s_a[i] = (tt < nt && jj < nj) ? hist[((size_t)(t0 + tt) * B + b) * T + j0 + jj] : 0.f;
        }
        __syncthreads();
        if (e < E)
            for (int tt = 0; tt < nt; ++tt) {
                const float d = dc[(size_t)(t0 + tt) * E];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) acc[jj] = fmaf(s_a[tt * 16 + jj], d, acc[jj]);
            }
    }
    if (e < E)
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) if (jj < nj) dvalues[((size_t)b * T + j0 + jj) * E + e] = acc[jj];
}

// dmemory = (dvalues + dkeys . Wmem^T is added by a GEMM beforehand) masked by the sequence lengths (values = memory * mask)
__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, const int* __restrict__ lens, int B, int T, int E) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)B * T * E; i += (long long)gridDim.x * blockDim.x) {
        const int pos = (int)((i / E) % T), b = (int)(i / ((long long)E * T));
        if (pos >= lens[b]) x[i] = 0.f;
    }
}

// ---- location-sensitive attention backward (reference forward: tacotron_v2.py:118-141) -------------------------------
// The recurrence only needs dq (-> attention LSTM) and d(state_prev) per step; everything else is deferred to tape-wide
// kernels.  Per step: attention_bwd_kernel, one workgroup per (row, chunk of T2_AB_J positions), recomputes
// f = conv(state_prev) + b_c, pre = keys + (q + b_a) + f . Wloc, tanh(pre) for its chunk and writes
//   de [B][T] (d energies, kept tape-wide), dctx [B][E] (kept tape-wide: d values = sum_t a_t (x) dctx_t),
//   per-chunk partial dq [U] and partial d(state_prev) over the chunk + conv halo;
// the row's last chunk to arrive sums the partials in chunk order (deterministic) into dq (plain + frag16) and g_state.
// The softmax-backward dot sum_j a_j da_j equals dctx . ctx_t + a . g_state (ctx_t = sum_j a_j values_j is on the tape),
// which lets every chunk work without the other chunks' da.
// The kernel also keeps du [B][T][U], f [B][T][F] and df [B][T][F] of the step tape-wide, so that after the loop
//   d Wloc = f^T . du and d conv kernel = im2col(state_prev)^T . df are two split-K GEMMs over all S*B*T rows,
//   d keys = sum_t du, d conv bias = column sums of df, and d v_a accumulates per step in the combine.
// Global-memory latency (about 1000 cycles a round trip) dominated the first version of these kernels: every load batch is
// now issued before its first use, the parameters come as one prebuilt LDS image (attention_image_kernel) copied with
// float4 loads.
#define T2_AB_J 16
#define T2_AB_THREADS 256

// workgroup barrier that orders LDS traffic only: global loads issued before it (prefetches) stay in flight
__device__ __forceinline__ void ab_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct AbLds {
    float *cw, *cb, *wlT, *v;                 // parameter image: conv kernel [K][FP], conv bias [F], Wloc^T [U][FP], v_a [U]
    float *sp, *f, *du, *df, *dfp, *q, *de;
    int FP, US;
};

__host__ __device__ inline int ab_us(int F) {             // u-splits of the df reduction so that all threads work
    const int items = (T2_AB_J / 4) * F;
    const int us = T2_AB_THREADS / items;
    return us < 1 ? 1 : (us > 8 ? 8 : us);
}
__host__ __device__ inline int ab_r4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int ab_image_floats(int U, int F, int K) { return ab_r4(K * (F + 1)) + ab_r4(F) + ab_r4(U * (F + 1)) + ab_r4(U); }
__host__ __device__ inline int ab_common_floats(int U, int F, int K) {
    const int FP = F + 1;
    return ab_image_floats(U, F, K) + ab_r4(T2_AB_J + K + 8) + 2 * ab_r4(T2_AB_J * FP) + ab_r4(T2_AB_J * U) + ab_r4(ab_us(F) * T2_AB_J * FP) +
           ab_r4(U) + ab_r4(T2_AB_J);
}
__host__ __device__ inline float* ab_carve(float* sm, int U, int F, int K, AbLds& L) {
    const int FP = F + 1;
    L.FP = FP; L.US = ab_us(F);
    float* p = sm;
    L.cw = p; p += ab_r4(K * FP);
    L.cb = p; p += ab_r4(F);
    L.wlT = p; p += ab_r4(U * FP);
    L.v = p; p += ab_r4(U);
    L.sp = p; p += ab_r4(T2_AB_J + K + 8);
    L.f = p; p += ab_r4(T2_AB_J * FP);
    L.df = p; p += ab_r4(T2_AB_J * FP);
    L.du = p; p += ab_r4(T2_AB_J * U);
    L.dfp = p; p += ab_r4(L.US * T2_AB_J * FP);
    L.q = p; p += ab_r4(U);
    L.de = p; p += ab_r4(T2_AB_J);
    return p;
}

// the parameter image in global memory, laid out exactly as the head of the kernels' LDS (one block)
__global__ __launch_bounds__(256) void attention_image_kernel(const float* __restrict__ conv_w, const float* __restrict__ conv_b,
                                                              const float* __restrict__ wloc, const float* __restrict__ v_a, float* __restrict__ img,
                                                              int U, int F, int K) {
    AbLds L;
    ab_carve(img, U, F, K, L);
    const int FP = L.FP, tid = threadIdx.x;
    for (int i = tid; i < ab_image_floats(U, F, K); i += 256) img[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < K * F; i += 256) { const int k = i / F, c = i - k * F; L.cw[k * FP + c] = conv_w[i]; }
    for (int i = tid; i < F; i += 256) L.cb[i] = conv_b[i];
    for (int i = tid; i < F * U; i += 256) { const int c = i / U, u = i - c * U; L.wlT[u * FP + c] = wloc[i]; }
    for (int i = tid; i < U; i += 256) L.v[i] = v_a[i];
}

// image -> LDS, eight float4 loads in flight per thread
__device__ inline void ab_load_image(float* sm, const float* __restrict__ img, int n4, int tid) {
    const float4* src = reinterpret_cast<const float4*>(img);
    float4* dst = reinterpret_cast<float4*>(sm);
    for (int base = tid; base < n4; base += 8 * T2_AB_THREADS) {
        float4 w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int idx = base + i * T2_AB_THREADS; w[i] = idx < n4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int idx = base + i * T2_AB_THREADS; if (idx < n4) dst[idx] = w[i]; }
    }
}

// keys of the chunk in the register tile of ab_du (zero past the valid positions)
__device__ inline void ab_load_keys(const float* __restrict__ keys_chunk, int jn, int U, int tid, float (*kq)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int idx = tid + m * T2_AB_THREADS;
        const int jq = idx / U, u = idx - jq * U;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) kq[m][jj] = (idx < (T2_AB_J / 4) * U && jq * 4 + jj < jn) ? keys_chunk[(size_t)(jq * 4 + jj) * U + u] : 0.f;
    }
}

__device__ inline void ab_features(const AbLds& L, int F, int K, int tid) {
    const int FP = L.FP;
    for (int i = tid; i < T2_AB_J * F; i += T2_AB_THREADS) {
        const int j = i / F, c = i - j * F;
        float acc = L.cb[c];
#pragma unroll 8
        for (int k = 0; k < K; ++k) acc = fmaf(L.cw[k * FP + c], L.sp[j + k], acc);
        L.f[j * FP + c] = acc;
    }
}

// du[j][u] = de[j] * v[u] * (1 - tanh(pre)^2) for the jn valid positions of the chunk (zero beyond); thread tile: one unit x
// four positions; s_dva[(position group, unit)] = this tile's part of d v_a = sum_j de[j] * tanh(pre).
__device__ inline void ab_du(const AbLds& L, const float (*kq)[4], int jn, int U, int F, int tid, float* s_dva) {
    const int FP = L.FP;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int idx = tid + m * T2_AB_THREADS;
        if (idx < (T2_AB_J / 4) * U) {
            const int jq = idx / U, u = idx - jq * U, jb = jq * 4;
            float pre[4];
            const float qu = L.q[u];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) pre[jj] = kq[m][jj] + qu;
#pragma unroll 8
            for (int c = 0; c < F; ++c) {
                const float w = L.wlT[u * FP + c];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) pre[jj] = fmaf(L.f[(jb + jj) * FP + c], w, pre[jj]);
            }
            const float vu = L.v[u];
            float dva = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float d = 0.f;
                if (jb + jj < jn) {
                    const float th = tanhf(pre[jj]), de = L.de[jb + jj];
                    d = de * vu * (1.f - th * th);
                    dva = fmaf(de, th, dva);
                }
                L.du[(jb + jj) * U + u] = d;
            }
            s_dva[idx] = dva;
        }
    }
}

// df[j][c] = sum_u du[j][u] * wloc[c][u]; thread tile: one filter x four positions x one u-range, partials summed in range order
__device__ inline void ab_df(const AbLds& L, int U, int F, int tid) {
    const int FP = L.FP, US = L.US, items = (T2_AB_J / 4) * F, ur = (U + US - 1) / US;
    for (int idx = tid; idx < items * US; idx += T2_AB_THREADS) {
        const int us = idx / items, r = idx - us * items, jq = r / F, c = r - jq * F, jb = jq * 4;
        const int u0 = us * ur, u1 = min(U, u0 + ur);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int u = u0; u < u1; ++u) {
            const float w = L.wlT[u * FP + c];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = fmaf(L.du[(jb + jj) * U + u], w, acc[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) L.dfp[(us * T2_AB_J + jb + jj) * FP + c] = acc[jj];
    }
    ab_lds_barrier();
    for (int i = tid; i < T2_AB_J * F; i += T2_AB_THREADS) {
        const int j = i / F, c = i - j * F;
        float s = 0.f;
        for (int us = 0; us < US; ++us) s += L.dfp[(us * T2_AB_J + j) * FP + c];
        L.df[j * FP + c] = s;
    }
}

struct AttnBwdArgs {
    const float *keys, *values;               // [B][T][U], [B][T][E]
    const int* lens;
    const float *q, *state_prev, *align, *ctx; // tape slices of step t: [B][U] (bias included), [B][T], [B][T], [B][E]
    const float* dctx0; int ld_dctx0;         // d context: sum of two sources (LSTM-1 input part, attention-LSTM input of t + 1)
    const float* dctx1; int ld_dctx1;
    const float* g_state;                     // [B][T] d / d attention state after this step
    const float* image;                       // attention_image_kernel's output
    float* de;                                // [B][T] this step's slice of the tape-wide d energies
    float* dctx_sum; long long ld_dctx_sum;   // row b at dctx_sum + b * ld: this step's slice of the tape-wide d context
    float *part_dq, *part_dva, *part_ds;      // [B][NJ][U], [B][NJ][U], [B][NJ][J + K - 1]
    float *du_out, *f_out, *df_out;           // this step's slices of the tape-wide du [B][T][U], f [B][T][F], df [B][T][F]
    int B, T, U, E, F, K, NJ, first;
    // the row's last chunk to arrive sums the partials (round 1 did this in a launch of its own)
    unsigned* ticket;                         // [B], zero between launches
    float *dq_plain, *dq_frag, *g_state_out, *p_va;
    int MB, cumulative;
};

__host__ __device__ inline int attn_bwd_lds_floats(int U, int E, int F, int K) {
    return ab_common_floats(U, F, K) + ab_r4(E) + 64 + ab_r4((T2_AB_J / 4) * U);
}

__global__ __launch_bounds__(T2_AB_THREADS) void attention_bwd_kernel(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = T2_AB_THREADS / 64, J = T2_AB_J, JW = J / NW;
    const int T = a.T, U = a.U, E = a.E, F = a.F, K = a.K;
    const int len = a.lens[b], j0 = ch * J;
    const int jt = min(J, T - j0);            // rows of the chunk that exist
    if (ch > 0 && j0 >= len) {                // masked chunk: its partials stay zero (workspace zeroed per call), tape rows are zero
        for (int i = tid; i < jt * U; i += T2_AB_THREADS) a.du_out[((size_t)b * T + j0) * U + i] = 0.f;
        for (int i = tid; i < jt * F; i += T2_AB_THREADS) { a.f_out[((size_t)b * T + j0) * F + i] = 0.f; a.df_out[((size_t)b * T + j0) * F + i] = 0.f; }
        return;
    }
    const int jn = max(0, min(J, len - j0)), pad_l = (K - 1) / 2, NI = J + K - 1;
    AbLds L;
    float* s_dctx = ab_carve(sm, U, F, K, L);
    float* s_red = s_dctx + ab_r4(E);         // [NW] dot partials, [16..32) da
    float* s_dva = s_red + 64;                // [(J / 4)][U]
    // ---- every global load of the kernel is issued here
    float r_dctx[2], r_ctx[2];                // first 512 columns; the rest (E > 512) goes through the slow loop below
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * T2_AB_THREADS;
        float v = 0.f;
        if (c < E) {
            if (a.dctx0) v += a.dctx0[(size_t)b * a.ld_dctx0 + c];
            if (a.dctx1) v += a.dctx1[(size_t)b * a.ld_dctx1 + c];
        }
        r_dctx[i] = v;
        r_ctx[i] = c < E ? a.ctx[(size_t)b * E + c] : 0.f;
    }
    float r_sp = 0.f, r_q[2] = {0.f, 0.f};
    if (tid < NI) { const int p = j0 - pad_l + tid; r_sp = (p >= 0 && p < T) ? a.state_prev[(size_t)b * T + p] : 0.f; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int u = tid + i * T2_AB_THREADS; if (u < U) r_q[i] = a.q[(size_t)b * U + u]; }
    float r_al[2] = {0.f, 0.f}, r_gs[2] = {0.f, 0.f};       // first 512 positions of the row (dot); the rest: slow loop
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = tid + i * T2_AB_THREADS;
        if (j < len && !a.first) { r_al[i] = a.align[(size_t)b * T + j]; r_gs[i] = a.g_state[(size_t)b * T + j]; }
    }
    const float my_al = tid < jn ? a.align[(size_t)b * T + j0 + tid] : 0.f;
    const float my_gs = (tid < jn && !a.first) ? a.g_state[(size_t)b * T + j0 + tid] : 0.f;
    const float* vals = a.values + ((size_t)b * T + j0) * E;
    float vv[JW][8];                          // values rows of this wave's positions, first 512 columns
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
        const int j = wave + jj * NW;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int c = lane + 64 * i; vv[jj][i] = (j < jn && c < E) ? vals[(size_t)j * E + c] : 0.f; }
    }
    float kq[4][4];
    ab_load_keys(a.keys + ((size_t)b * T + j0) * U, jn, U, tid, kq);
    ab_load_image(sm, a.image, ab_image_floats(U, F, K) / 4, tid);
    // ---- stage the step's inputs
    float part = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * T2_AB_THREADS;
        if (c < E) {
            s_dctx[c] = r_dctx[i];
            if (ch == 0) a.dctx_sum[(size_t)b * a.ld_dctx_sum + c] = r_dctx[i];
            part = fmaf(r_dctx[i], r_ctx[i], part);
        }
        part = fmaf(r_al[i], r_gs[i], part);
    }
    for (int c = tid + 2 * T2_AB_THREADS; c < E; c += T2_AB_THREADS) {
        float v = 0.f;
        if (a.dctx0) v += a.dctx0[(size_t)b * a.ld_dctx0 + c];
        if (a.dctx1) v += a.dctx1[(size_t)b * a.ld_dctx1 + c];
        s_dctx[c] = v;
        if (ch == 0) a.dctx_sum[(size_t)b * a.ld_dctx_sum + c] = v;
        part = fmaf(v, a.ctx[(size_t)b * E + c], part);
    }
    if (!a.first)
        for (int j = tid + 2 * T2_AB_THREADS; j < len; j += T2_AB_THREADS) part = fmaf(a.align[(size_t)b * T + j], a.g_state[(size_t)b * T + j], part);
    if (tid < NI) L.sp[tid] = r_sp;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int u = tid + i * T2_AB_THREADS; if (u < U) L.q[u] = r_q[i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lane == 0) s_red[wave] = part;
    ab_lds_barrier();
    ab_features(L, F, K, tid);
    // d alignments of the chunk: da_j = dctx . values[j] + d state after this step
    {
        float acc[JW];
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) acc[jj] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            const float dv = c < E ? s_dctx[c] : 0.f;
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) acc[jj] = fmaf(dv, vv[jj][i], acc[jj]);
        }
        for (int c = 512 + lane; c < E; c += 64)
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) {
                const int j = wave + jj * NW;
                if (j < jn) acc[jj] = fmaf(s_dctx[c], vals[(size_t)j * E + c], acc[jj]);
            }
#pragma unroll
        for (int jj = 0; jj < JW; ++jj) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[jj] += __shfl_xor(acc[jj], off, 64);
            if (lane == 0) s_red[16 + wave + jj * NW] = acc[jj];
        }
    }
    ab_lds_barrier();
    if (tid < J) {
        float dot = 0.f;
        for (int w = 0; w < NW; ++w) dot += s_red[w];
        const float de = tid < jn ? my_al * (s_red[16 + tid] + my_gs - dot) : 0.f;
        L.de[tid] = de;
        if (j0 + tid < T) a.de[(size_t)b * T + j0 + tid] = de;
    }
    ab_lds_barrier();
    ab_du(L, kq, jn, U, F, tid, s_dva);
    ab_lds_barrier();
    for (int u = tid; u < U; u += T2_AB_THREADS) {
        float s = 0.f, sv = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) s += L.du[j * U + u];
#pragma unroll
        for (int jq = 0; jq < J / 4; ++jq) sv += s_dva[jq * U + u];
        __hip_atomic_store(a.part_dq + ((size_t)b * a.NJ + ch) * U + u, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1):
        __hip_atomic_store(a.part_dva + ((size_t)b * a.NJ + ch) * U + u, sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // read inside this launch
    }
    for (int i = tid; i < jt * U; i += T2_AB_THREADS) a.du_out[((size_t)b * T + j0) * U + i] = L.du[i];
    ab_df(L, U, F, tid);
    ab_lds_barrier();
    for (int i = tid; i < jt * F; i += T2_AB_THREADS) {
        const int j = i / F, c = i - j * F;
        a.f_out[((size_t)b * T + j0) * F + i] = L.f[j * L.FP + c];
        a.df_out[((size_t)b * T + j0) * F + i] = L.df[j * L.FP + c];
    }
    // partial d state_prev over the chunk + halo: ds[i] = sum_{j, c} w[i - j][c] * df[j][c], i <-> position j0 - pad_l + i
    const int FP = L.FP;
    for (int i0 = 0; i0 < NI; i0 += T2_AB_THREADS / 4) {
        const int i = i0 + (tid >> 2), cg = tid & 3;
        float acc = 0.f;
        if (i < NI) {
            const int ja = max(0, i - K + 1), jb = min(jn - 1, i);
            for (int j = ja; j <= jb; ++j)
#pragma unroll 8
                for (int c = cg; c < F; c += 4) acc = fmaf(L.cw[(i - j) * FP + c], L.df[j * FP + c], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (i < NI && cg == 0) __hip_atomic_store(a.part_ds + ((size_t)b * a.NJ + ch) * NI + i, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- in-launch hand-off to the row's last chunk (MI355X_MICROARCH.md "Valid forms", sc1 table row 1: sc1 partial stores,
    //      every wave drains, barrier, one agent-scope atomic add on ONE counter per row; the workgroup whose add came last reads
    //      all partials with sc1 loads).  Chunks past the row's length never run this far and are not counted.
    __shared__ unsigned s_old;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_old = __hip_atomic_fetch_add(a.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned n_active = (unsigned)max(1, (min(len, T) + J - 1) / J);
    if (s_old != n_active - 1u) return;
    if (tid == 0) __hip_atomic_store(a.ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int u = tid; u < U; u += T2_AB_THREADS) {
        float s = 0.f;
        float sv = a.first ? 0.f : a.p_va[(size_t)b * U + u];
        for (int c2 = 0; c2 < a.NJ; ++c2) {
            s += __hip_atomic_load(a.part_dq + ((size_t)b * a.NJ + c2) * U + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sv += __hip_atomic_load(a.part_dva + ((size_t)b * a.NJ + c2) * U + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a.dq_plain[(size_t)b * U + u] = s;
        a.dq_frag[frag_idx(b, u, a.MB)] = s;
        a.p_va[(size_t)b * U + u] = sv;
    }
    for (int p = tid; p < T; p += T2_AB_THREADS) {
        float s = (a.cumulative && !a.first) ? a.g_state[(size_t)b * T + p] : 0.f;    // state_t = state_{t-1} + a_t
        for (int c2 = 0; c2 < a.NJ; ++c2) {
            const int i = p - (c2 * J - pad_l);
            if (i >= 0 && i < NI) s += __hip_atomic_load(a.part_ds + ((size_t)b * a.NJ + c2) * NI + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a.g_state_out[(size_t)b * T + p] = s;
    }
}

struct BwdOperand {             // skinny dense operand holding W^T of a TF-layout [rows][cols] variable: K = cols, N = rows
    Packed p;
};

struct t2_trainer {
    t2_model* m = nullptr;
    int B = 0, T = 0, S = 0;
    void* slab = nullptr;
    size_t slab_bytes = 0;
    TrainTape tape{};
    t2_state* state = nullptr;          // decoder state + staging owned by the trainer (never evicted by other decodes)
    bool have_forward = false;
    // backward pass
    std::vector<void*> allocs;
    Packed projT, lstmT[3], wqT, prenetT[T2_MAX_PRENET_LAYERS];
    float* wmemT = nullptr;             // [U][E] memory layer transposed (plain)
    float* prenetTp[T2_MAX_PRENET_LAYERS] = {};    // PreNet layer i >= 1 transposed (plain [P_i][P_{i-1}]): its data gradient runs once, after the loop
    float* dpre_tmp = nullptr;          // [S*B][max PreNet width]
    float* compose_tmp = nullptr;       // [D + 1][N2] composed projection operand before packing (parameter refresh)
    float* projTp = nullptr;            // [M*r + 1][D] projections transposed (plain): d y2 of all steps is one GEMM before the loop
    float* dy2_all = nullptr;           // [S*B][D]
    float* dzf2b = nullptr;             // second frag16 operand of LSTM 2's dz (ping-pong with dzf[2] by step parity)
    // persistent backward recurrence (persist_bwd.h): column-owner tiles of the transposed kernels, exchange slab, counters
    float* pzb_w[4] = {};               // LSTM 2, LSTM 1, attention LSTM, query layer
    char* pzb_slab = nullptr; size_t pzb_slab_bytes = 0;
    unsigned pzb_off[9] = {}, pzb_dz_bytes = 0, pzb_dq_bytes = 0, pzb_xs_bytes = 0;
    unsigned* pzb_bar = nullptr; float* pzb_pva = nullptr;
    void* bslab = nullptr; size_t bslab_bytes = 0;
    float *dY = nullptr, *dYf = nullptr, *dz[3] = {}, *dzf[3] = {}, *dq = nullptr, *dqf = nullptr;
    float *dpre[T2_MAX_PRENET_LAYERS] = {}, *dpref[T2_MAX_PRENET_LAYERS] = {}, *xin = nullptr;
    float *dX[3] = {}, *dy2 = nullptr, *dhq = nullptr, *dxp = nullptr;
    float *g_c[3] = {}, *g_hp[3] = {}, *g_state = nullptr;
    float *dkeys = nullptr, *dvalues = nullptr, *p_va = nullptr;
    float* ab_image = nullptr;
    float *de_all = nullptr, *dctx_all = nullptr, *part_dq = nullptr, *part_dva = nullptr, *part_ds = nullptr;
    float *du_all = nullptr, *f_all = nullptr, *df_all = nullptr;      // own allocation: every row is written each call, never zero-filled
    int NJ = 0;                         // attention backward: position chunks per row
    float *tmp = nullptr, *loss3 = nullptr;
    float* cs_part = nullptr;                      // column-sum partials [chunks][N], T2_CS_PART_FLOATS
    double* bn_part = nullptr; int bn_cols = 0;    // batch-norm reduction partials [T2_BN_CHUNKS][2][bn_cols]
    float* wg_part = nullptr;                      // split-K partials of the weight-gradient GEMMs [splits][rows][cols]
    float* sk_part = nullptr; unsigned* sk_ticket = nullptr;   // split-K halves of the recurrence's data-gradient GEMMs (bwd_dense_kernel SPLIT)
    std::map<std::string, long long> offsets;      // flat-gradient layout given by the caller
    const float* glob_counts = nullptr;            // dev [2]: non-zero loss weights of the whole data-parallel batch (null: this rank's own)
    unsigned long long gen = 0;                    // parameter generation (t2_model::param_gen) the trainer-owned operands were built from
};

static int trainer_pack_T(t2_trainer* tr, Packed* out, const std::string& name, int rows, int cols, int col0, int ncols) {
    // W^T of the column range [col0, col0 + ncols) of variable `name` ([rows][cols]) as a skinny dense operand:
    // reduction over the ncols source columns, one output column per source row
    t2_model* m = tr->m;
    const HostWeight* w;
    T2_TRY(get_w(m, name, &w, {rows, cols}));
    std::vector<float> t((size_t)ncols * rows);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < ncols; ++c) t[(size_t)c * rows + r] = w->data[(size_t)r * cols + col0 + c];
    const size_t before = m->allocs.size();
    T2_TRY(pack_skinny(m, out, t.data(), rows, {{0, ncols}}, 0, rows, nullptr));
    for (size_t i = before; i < m->allocs.size(); ++i) tr->allocs.push_back(m->allocs[i]);     // owned by the trainer
    m->allocs.resize(before);
    return T2_OK;
}


// ---- persistent backward recurrence (persist_bwd.h): operands and workspace ----------------------------------------------
// which: 0 LSTM 2 (d y1 | d h2_prev), 1 LSTM 1 (d h_att | d h1_prev | d ctx), 2 attention LSTM (d h_att_prev | d ctx_prev | d prenet),
// 3 query layer (d h_att).  Column c of workgroup g's tile = row n of the TF kernel [inputs][outputs].
static pz::TileMap pzb_map(int which) {
    pz::TileMap mp{};
    auto seg = [&](int i, int c0, int base, int per_g) { mp.c0[i] = c0; mp.base[i] = base; mp.per_g[i] = per_g; };
    if (which == 0) { mp.nseg = 2; mp.NC = pz::NC2; seg(0, 0, 0, 4); seg(1, 4, pz::KBH * 16, 4); }
    else if (which == 1) { mp.nseg = 3; mp.NC = pz::NC1; seg(0, 0, 0, 4); seg(1, 4, pz::KBH * 16 + pz::EW, 4); seg(2, 8, pz::KBH * 16, 2); }
    else if (which == 2) { mp.nseg = 3; mp.NC = pz::NC0; seg(0, 0, pz::KBX * 16 + pz::EW, 4); seg(1, 4, pz::KBX * 16, 2); seg(2, 6, 0, 1); }
    else { mp.nseg = 1; mp.NC = pz::NCQ; seg(0, 0, 0, 4); }
    return mp;
}
static inline size_t pzb_tile_bytes(int which) {
    const pz::TileMap mp = pzb_map(which);
    return (size_t)pz::NWG * (which == 3 ? pz::UA / 16 : pz::KBZ) * 64 * mp.NC;
}
static int pzb_pack(t2_trainer* tr, int which, const float* w_dev) {
    if (!tr->pzb_w[which]) return T2_OK;
    const pz::TileMap mp = pzb_map(which);
    const int Kc = which == 3 ? pz::UA : pz::KBZ * 16;
    const long long n4 = (long long)pzb_tile_bytes(which) / 16;
    const dim3 pgrid((unsigned)std::min<long long>(4096, (n4 + 255) / 256));
    if (which == 3) hipLaunchKernelGGL(pz::pack_bwd_tiles_kernel, pgrid, dim3(256), 0, tr->m->ctx->stream, w_dev, Kc, pz::NWG, mp, tr->pzb_w[which]);
    else hipLaunchKernelGGL(pz::pack_bwd_tiles2d_kernel, pgrid, dim3(256), 0, tr->m->ctx->stream, w_dev, Kc, mp, tr->pzb_w[which]);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}
static int pzb_setup(t2_trainer* tr) {
    t2_model* m = tr->m;
    const int B = tr->B, T = tr->T, MB = mb_for(B);
    if (!persist_dims_ok(m, B, T)) return T2_OK;
    const int C = persist_chunks(B), CH = (T + C - 1) / C;
    if ((size_t)pz::bwd_lds_floats(MB, T, CH) * 4 > 150 * 1024) return T2_OK;
    hipStream_t st = m->ctx->stream;
    const char* names[4] = {"decoder/lstm_2/kernel", "decoder/lstm_1/kernel", "decoder/attention_lstm/kernel", "attention/query_layer/kernel"};
    const int rows[4] = {2 * pz::KBH * 16, 2 * pz::KBH * 16 + pz::EW, pz::KBX * 16 + pz::EW + pz::KBH * 16, pz::KBH * 16};
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)rows[1] * pz::KBZ * 16 * 4));
    for (int i = 0; i < 4; ++i) {
        const int Kc = i == 3 ? pz::UA : pz::KBZ * 16;
        const HostWeight* w;
        int rc = get_w(m, names[i], &w, {rows[i], Kc});
        hipError_t e = rc == T2_OK ? hipMalloc((void**)&tr->pzb_w[i], pzb_tile_bytes(i)) : hipErrorUnknown;
        if (e == hipSuccess) { tr->allocs.push_back(tr->pzb_w[i]); e = hipMemcpyAsync(tmp, w->data.data(), (size_t)rows[i] * Kc * 4, hipMemcpyHostToDevice, st); }
        if (e == hipSuccess) rc = pzb_pack(tr, i, tmp);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (rc != T2_OK || e != hipSuccess) { (void)hipFree(tmp); return rc != T2_OK ? rc : fail(T2_EHIP, "persistent backward operands: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(hipFree(tmp));
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (unsigned)o; };
    tr->pzb_dz_bytes = (unsigned)((size_t)MB * 16 * pz::KBZ * 16 * 4);
    tr->pzb_dq_bytes = (unsigned)((size_t)MB * 16 * pz::UA * 4);
    tr->pzb_off[0] = carve(2 * (size_t)tr->pzb_dz_bytes); tr->pzb_off[1] = carve(tr->pzb_dz_bytes); tr->pzb_off[2] = carve(tr->pzb_dz_bytes);
    tr->pzb_off[3] = carve((size_t)MB * 16 * pz::EW * 4);
    tr->pzb_off[4] = carve((size_t)C * tr->pzb_dq_bytes);
    tr->pzb_off[5] = carve(2 * (size_t)B * C * (CH + 32) * 4);
    tr->pzb_off[6] = carve((size_t)pz::NWG * MB * 16 * 4 * pz::NC2 * 4);          // partial product tiles of GEMM 2, 1, 0
    tr->pzb_off[7] = carve((size_t)pz::NWG * MB * 16 * 4 * pz::NC1 * 4);
    tr->pzb_off[8] = carve((size_t)pz::NWG * MB * 16 * 4 * pz::NC0 * 4);
    tr->pzb_xs_bytes = (unsigned)off;
    const size_t o_bar = carve((size_t)(pz::NSHARD + 1) * 32 * 4);
    tr->pzb_slab_bytes = off;
    const size_t o_pva = carve((size_t)pz::NWG * pz::UA * 4);
    HIP_TRY(hipMalloc((void**)&tr->pzb_slab, off));
    tr->allocs.push_back(tr->pzb_slab);
    tr->pzb_bar = (unsigned*)(tr->pzb_slab + o_bar);
    tr->pzb_pva = (float*)(tr->pzb_slab + o_pva);
    HIP_TRY(hipFuncSetAttribute((const void*)pz::backward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)pz::backward_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)pz::backward_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    return T2_OK;
}
// the whole reverse-time recurrence as one launch; fills the same tape buffers as the launch-per-phase loop
static int pzb_launch(t2_trainer* tr, bool* timed_out) {
    t2_model* m = tr->m;
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    t2_state* s = tr->state;
    const TrainTape& tp = tr->tape;
    const int B = tr->B, T = tr->T, S = tr->S, MB = mb_for(B), np = d.num_prenet_layers;
    HIP_TRY(hipMemsetAsync(tr->pzb_slab, 0, tr->pzb_slab_bytes, st));
    pz::BArgs a{};
    a.wT2 = tr->pzb_w[0]; a.wT1 = tr->pzb_w[1]; a.wT0 = tr->pzb_w[2]; a.wTQ = tr->pzb_w[3];
    a.conv_w = m->conv_w; a.conv_b = m->conv_b; a.wloc = m->wloc; a.v_a = m->v_a;
    a.keys = s->keys; a.values = s->values; a.lens = s->lens;
    for (int l = 0; l < 3; ++l) { a.z[l] = tp.z[l]; a.cprev[l] = tp.cprev[l]; a.cnew[l] = tp.cnew[l]; a.dz[l] = tr->dz[l]; }
    a.q = tp.q; a.state_prev = tp.state_prev; a.ctx = tp.ctx; a.hist = s->o_hist;
    a.dy2_all = tr->dy2_all; a.xp_last = tp.xp[np - 1];
    a.dq = tr->dq; a.de_all = tr->de_all; a.dctx_all = tr->dctx_all; a.du_all = tr->du_all; a.f_all = tr->f_all; a.df_all = tr->df_all;
    a.dpre = tr->dpre[np - 1]; a.pva_part = tr->pzb_pva; a.dctx_row_ld = (long long)S * pz::EW;
    a.xs = (float*)tr->pzb_slab; a.xs_bytes = tr->pzb_xs_bytes;
    a.oDZ2 = tr->pzb_off[0]; a.oDZ1 = tr->pzb_off[1]; a.oDZ0 = tr->pzb_off[2]; a.oDCTX = tr->pzb_off[3]; a.oDQ = tr->pzb_off[4]; a.oDS = tr->pzb_off[5];
    a.oP2 = tr->pzb_off[6]; a.oP1 = tr->pzb_off[7]; a.oP0 = tr->pzb_off[8];
    a.dz_bytes = tr->pzb_dz_bytes; a.dq_bytes = tr->pzb_dq_bytes;
    a.bar = tr->pzb_bar; a.err = (int*)(tr->pzb_bar + pz::NSHARD * 32);
    a.B = B; a.T = T; a.S = S; a.C = persist_chunks(B); a.CH = (T + a.C - 1) / a.C;
    a.F = d.attention_filters; a.KA = d.attention_kernel; a.cumulative = d.cumulative_weights;
    a.seed = tp.seed; a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias;
    a.inv_keep = 1.0f / (1.0f - tp.prenet_drop_rate);
    const size_t lds = (size_t)pz::bwd_lds_floats(MB, T, a.CH) * 4;
    const dim3 grid(pz::NWG), blk(pz::NT);
    {
        XwgGate gate;
        T2_TRY(gate.begin(m->ctx));
        if (MB == 1) hipLaunchKernelGGL((pz::backward_kernel<1>), grid, blk, lds, st, a);
        else if (MB == 2) hipLaunchKernelGGL((pz::backward_kernel<2>), grid, blk, lds, st, a);
        else hipLaunchKernelGGL((pz::backward_kernel<4>), grid, blk, lds, st, a);
        HIP_TRY(hipGetLastError());
    }
    T2_TRY(inject_timeout(m, a.err));
    ++m->n_bwd_persist;
    // a hop that ran out of time leaves garbage gradients: the caller then runs the launch-per-phase recurrence instead (it
    // rewrites every tape buffer this kernel touched); the caller reads the loss after this call anyway, so the
    // synchronisation costs nothing extra
    int h_err = 0;
    HIP_TRY(hipMemcpyAsync(&h_err, a.err, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *timed_out = h_err != 0;
    if (h_err) note_persist_timeout(m);
    return T2_OK;
}

static void enc_tape_release(t2_trainer* tr);

extern "C" int t2_trainer_destroy(t2_trainer* tr) {
    if (!tr) return T2_OK;
    (void)hipSetDevice(tr->m->ctx->device);
    (void)hipStreamSynchronize(tr->m->ctx->stream);
    enc_tape_release(tr);
    if (tr->state && tr->m->last_loop == tr->state) tr->m->last_loop = nullptr;
    if (tr->slab) (void)hipFree(tr->slab);
    if (tr->bslab) (void)hipFree(tr->bslab);
    if (tr->cs_part) (void)hipFree(tr->cs_part);
    if (tr->wg_part) (void)hipFree(tr->wg_part);
    if (tr->bn_part) (void)hipFree(tr->bn_part);
    if (tr->du_all) (void)hipFree(tr->du_all);
    for (void* p : tr->allocs) (void)hipFree(p);
    if (tr->state) state_free(tr->state);
    delete tr;
    return T2_OK;
}

extern "C" int t2_trainer_create(t2_model* m, int B, int T_in, int S, t2_trainer** out) {
    if (!m || !out) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    T2_TRY(check_batch(B, T_in));
    if (S < 1) return fail(T2_EINVAL, "S must be >= 1");
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    t2_trainer* tr = new t2_trainer();
    tr->m = m; tr->B = B; tr->T = T_in; tr->S = S;
    const int H[3] = {d.attention_lstm_units, d.decoder_lstm_units, d.decoder_lstm_units};
    size_t off = 0;
    auto carve = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
    const size_t SB = (size_t)S * B;
    size_t o_xp[T2_MAX_PRENET_LAYERS], o_z[3], o_cp[3], o_cn[3], o_hr[3], o_hs[3];
    for (int i = 0; i < d.num_prenet_layers; ++i) o_xp[i] = carve(SB * d.prenet_units[i]);
    for (int l = 0; l < 3; ++l) {
        o_z[l] = carve(SB * 4 * H[l]); o_cp[l] = carve(SB * H[l]); o_cn[l] = carve(SB * H[l]);
        o_hr[l] = carve(SB * H[l]); o_hs[l] = carve(SB * H[l]);
    }
    const size_t o_q = carve(SB * d.attention_dim), o_sp = carve(SB * T_in), o_ctx = carve(SB * d.enc_units);
    tr->slab_bytes = off;
    hipError_t e = hipMalloc(&tr->slab, off);
    if (e != hipSuccess) { delete tr; return fail(T2_ENOMEM, "hipMalloc(trainer tape, %zu bytes) failed: %s", off, hipGetErrorString(e)); }
    char* p = (char*)tr->slab;
    for (int i = 0; i < d.num_prenet_layers; ++i) tr->tape.xp[i] = (float*)(p + o_xp[i]);
    for (int l = 0; l < 3; ++l) {
        tr->tape.z[l] = (float*)(p + o_z[l]); tr->tape.cprev[l] = (float*)(p + o_cp[l]); tr->tape.cnew[l] = (float*)(p + o_cn[l]);
        tr->tape.hraw[l] = (float*)(p + o_hr[l]); tr->tape.hstate[l] = (float*)(p + o_hs[l]);
    }
    tr->tape.q = (float*)(p + o_q); tr->tape.state_prev = (float*)(p + o_sp); tr->tape.ctx = (float*)(p + o_ctx);
    if (attn_lds_bytes(m->d, T_in) > 120 * 1024) { t2_trainer_destroy(tr); return fail(T2_EINVAL, "T_in %d too long for the attention kernel's LDS", T_in); }
    int rc = state_alloc(m, B, T_in, S, true, &tr->state);
    if (rc != T2_OK) { t2_trainer_destroy(tr); return rc; }
    *out = tr;
    return T2_OK;
}

// DecoderV1.call(is_training=True) with the V2 cells (tacotron_v1.py:134-150): TrainingHelper, S = T_out / r steps for every
// row, PreNet dropout with prenet_drop_rate (modules.py:90,96-98), ZoneoutLSTMCell training branch (modules.py:274-276).
// Outputs as t2_decode_teacher; the tape stays in the trainer for t2_train_decoder_backward.
extern "C" int t2_train_decoder_forward(t2_trainer* tr, const float* memory, const int64_t* lens, const float* targets,
                                        float prenet_drop_rate, uint32_t seed, float* mel, float* stop, float* align) {
    if (!tr || !memory || !targets || !mel || !stop || !align) return fail(T2_EINVAL, "null argument");
    if (!(prenet_drop_rate >= 0.f && prenet_drop_rate < 1.f)) return fail(T2_EINVAL, "drop rate must be in [0, 1)");
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = tr->B, T_in = tr->T, S = tr->S, M = d.num_mels, r = d.outputs_per_step;
    t2_state* s = tr->state;
    holdoff_tick(m);
    T2_TRY(state_init(s, memory, lens));
    HIP_TRY(hipMemcpyAsync(s->i_targets, targets, (size_t)B * S * r * M * 4, hipMemcpyDeviceToDevice, st));
    tr->tape.seed = seed; tr->tape.prenet_drop_rate = prenet_drop_rate;
    // one pass; a persistent launch whose hand-off waits ran out of time is followed by a second pass, one launch per phase
    for (int pass = 0; pass < 2; ++pass) {
    T2_TRY(state_zero(s, S, 0, 0));
    // TrainingHelper feeds known frames, so the PreNets of all S steps run as batched GEMMs over S*B rows (tape rows
    // xp[i][t][b][:]), followed by the sampled dropout; the last layer also lands as the per-step frag16 operands.
    T2_TRY(teacher_prenets(s, prenet_drop_rate, seed, tr->tape.xp));
    if (m->opt_tfproj) T2_TRY(teacher_projection_buffers(s, false));
    StepIO io = loop_io(s, 0);
    io.train = &tr->tape;
    io.y_all = nullptr;                            // the tape keeps LSTM 2's raw output of every step (hraw[2])
    io.slot_hint = 1;                              // the control block's base stays 0: step t runs as step_off = t
    // the whole teacher-forced recurrence as ONE persistent launch (persist.h, teacher mode + tape), else one launch per phase
    const bool persist = m->opt_persist && m->persist_holdoff == 0 && m->opt_tfproj && s->pz_xs != nullptr && s->tf_frag != nullptr &&
                         s->proj_tmp != nullptr && !m->opt_lstm_only && tr->tape.hraw[2] != nullptr;
    m->last_loop = s; m->last_persist = persist;   // t2_decode_final_state reads the trainer's state until the trainer goes
    if (persist) T2_TRY(persist_decode(s, S, 0, 0, true, &tr->tape));
    else {
        for (int t = 0; t < S; ++t) T2_TRY(launch_step(s, t, io));        // eager launches: seed and rates are per-call kernel arguments
        // the steps ran with explicit offsets; leave the step count where t2_state_get looks for the parity of the h slots
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(s->ctl + CTL_BASE), S, 1, st));
    }
    if (persist || io.defer_proj) T2_TRY(teacher_projection(s, tr->tape.hraw[2], s->o_mel, (long long)s->cap * r * M, s->o_stop, s->cap));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mel, s->o_mel, (size_t)B * S * r * M * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)B * S * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(align, s->o_hist, (size_t)S * B * T_in * 4, hipMemcpyDeviceToDevice, st));
    if (!persist) break;
    bool timed_out = false;
    T2_TRY(persist_timed_out(s, &timed_out));      // the backward pass must not consume a tape a timed-out launch left behind
    if (!timed_out) break;
    }
    tr->have_forward = true;
    return T2_OK;
}

// ---- backward pass -------------------------------------------------------------------------------------------------------
static int trainer_sync_params(t2_trainer* tr);
static int trainer_setup_backward(t2_trainer* tr) {
    t2_model* m = tr->m;
    const t2_dims& d = m->d;
    const int B = tr->B, T = tr->T, S = tr->S, MB = mb_for(B), R = MB * 16;
    const int E = d.enc_units, U = d.attention_dim, A = d.attention_lstm_units, D = d.decoder_lstm_units;
    const int M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, np = d.num_prenet_layers, N = M * r + 1;
    const int P_last = d.prenet_units[np - 1];
    // transposed operands of the per-step data-gradient GEMMs
    {
        const HostWeight *wm, *ws;
        T2_TRY(get_w(m, "decoder/out_projection/kernel", &wm, {D, M * r}));
        T2_TRY(get_w(m, "decoder/stop_token_projection/kernel", &ws, {D, 1}));
        std::vector<float> t((size_t)N * D);
        for (int k = 0; k < D; ++k) {
            for (int c = 0; c < M * r; ++c) t[(size_t)c * D + k] = wm->data[(size_t)k * M * r + c];
            t[(size_t)(M * r) * D + k] = ws->data[k];
        }
        HIP_TRY(hipMalloc((void**)&tr->projTp, t.size() * 4));
        tr->allocs.push_back(tr->projTp);
        HIP_TRY(hipMemcpy(tr->projTp, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        const size_t before = m->allocs.size();
        T2_TRY(pack_skinny(m, &tr->projT, t.data(), D, {{0, N}}, 0, D, nullptr));
        for (size_t i = before; i < m->allocs.size(); ++i) tr->allocs.push_back(m->allocs[i]);
        m->allocs.resize(before);
    }
    T2_TRY(trainer_pack_T(tr, &tr->lstmT[0], "decoder/attention_lstm/kernel", P_last + E + A, 4 * A, 0, 4 * A));
    T2_TRY(trainer_pack_T(tr, &tr->lstmT[1], "decoder/lstm_1/kernel", A + E + D, 4 * D, 0, 4 * D));
    T2_TRY(trainer_pack_T(tr, &tr->lstmT[2], "decoder/lstm_2/kernel", D + D, 4 * D, 0, 4 * D));
    T2_TRY(trainer_pack_T(tr, &tr->wqT, "attention/query_layer/kernel", A, U, 0, U));
    for (int i = 1; i < np; ++i)
        T2_TRY(trainer_pack_T(tr, &tr->prenetT[i], "decoder/prenet_" + std::to_string(i) + "/dense/kernel", d.prenet_units[i - 1],
                              d.prenet_units[i], 0, d.prenet_units[i]));
    {
        const HostWeight* w;
        T2_TRY(get_w(m, "attention/memory_layer/kernel", &w, {E, U}));
        std::vector<float> t((size_t)U * E);
        for (int e = 0; e < E; ++e) for (int u = 0; u < U; ++u) t[(size_t)u * E + e] = w->data[(size_t)e * U + u];
        HIP_TRY(hipMalloc((void**)&tr->wmemT, t.size() * 4));
        tr->allocs.push_back(tr->wmemT);
        HIP_TRY(hipMemcpy(tr->wmemT, t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    for (int i = 1; i < np; ++i) {
        const HostWeight* w;
        const int Pi = d.prenet_units[i], Pp = d.prenet_units[i - 1];
        T2_TRY(get_w(m, "decoder/prenet_" + std::to_string(i) + "/dense/kernel", &w, {Pp, Pi}));
        std::vector<float> t((size_t)Pi * Pp);
        for (int k = 0; k < Pp; ++k) for (int c = 0; c < Pi; ++c) t[(size_t)c * Pp + k] = w->data[(size_t)k * Pi + c];
        HIP_TRY(hipMalloc((void**)&tr->prenetTp[i], t.size() * 4));
        tr->allocs.push_back(tr->prenetTp[i]);
        HIP_TRY(hipMemcpy(tr->prenetTp[i], t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    // workspaces
    const int H[3] = {A, D, D};
    const int KX[3] = {P_last + E + A, A + E + D, D + D};
    const size_t SB = (size_t)S * B;
    size_t off = 0;
    auto carve = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
    const size_t o_dY = carve(SB * N), o_dYf = carve((size_t)S * R * pad16(N));
    size_t o_dz[3], o_dzf[3], o_dX[3], o_gc[3], o_ghp[3], o_dpre[T2_MAX_PRENET_LAYERS], o_dpref[T2_MAX_PRENET_LAYERS];
    for (int l = 0; l < 3; ++l) {
        o_dz[l] = carve(SB * 4 * H[l]); o_dzf[l] = carve((size_t)R * pad16(4 * H[l])); o_dX[l] = carve((size_t)B * KX[l]);
        o_gc[l] = carve((size_t)B * H[l]); o_ghp[l] = carve((size_t)B * H[l]);
    }
    const size_t o_dq = carve(SB * U), o_dqf = carve((size_t)R * pad16(U));
    int pmax = 0;
    for (int i = 0; i < np; ++i) { o_dpre[i] = carve(SB * d.prenet_units[i]); o_dpref[i] = carve((size_t)R * pad16(d.prenet_units[i])); pmax = std::max(pmax, d.prenet_units[i]); }
    const size_t o_xin = carve(SB * M * nf), o_dy2 = carve((size_t)B * D), o_dhq = carve((size_t)B * A), o_dxp = carve((size_t)B * pmax);
    const size_t o_gs = carve((size_t)B * T), o_dk = carve((size_t)B * T * U), o_dv = carve((size_t)B * T * E);
    const int NJ = ceil_div(T, T2_AB_J);
    tr->NJ = NJ;
    const size_t o_pva = carve((size_t)B * U);
    const size_t o_de = carve(SB * T), o_dctx = carve(SB * E), o_pdq = carve((size_t)B * NJ * U), o_pdva = carve((size_t)B * NJ * U),
                 o_pds = carve((size_t)B * NJ * (T2_AB_J + d.attention_kernel - 1)),
                 o_img = carve((size_t)ab_image_floats(U, d.attention_filters, d.attention_kernel));
    if (T2_AB_J + d.attention_kernel - 1 > T2_AB_THREADS || U > 256)
        return fail(T2_EINVAL, "attention kernel size / attention_dim too large for the attention backward kernels");
    const size_t o_tmp = carve((size_t)std::max(D * N, 4096)), o_loss = carve(16), o_dpt = carve(SB * pmax), o_dy2a = carve(SB * D),
                 o_dzf2b = carve((size_t)R * pad16(4 * D));
    tr->bslab_bytes = off;
    HIP_TRY(hipMalloc(&tr->bslab, off));
    {
        const size_t part_bytes = (size_t)T2_SK_MAX_TILES * 2 * MB * 256 * 4;
        char* q = nullptr;
        HIP_TRY(hipMalloc((void**)&q, part_bytes + (T2_SK_MAX_TILES + 64) * 4));      // + one ticket per batch row (attention_bwd_kernel)
        tr->allocs.push_back(q);
        HIP_TRY(hipMemset(q, 0, part_bytes + (T2_SK_MAX_TILES + 64) * 4));
        tr->sk_part = (float*)q; tr->sk_ticket = (unsigned*)(q + part_bytes);
    }
    char* p = (char*)tr->bslab;
    tr->dY = (float*)(p + o_dY); tr->dYf = (float*)(p + o_dYf);
    for (int l = 0; l < 3; ++l) {
        tr->dz[l] = (float*)(p + o_dz[l]); tr->dzf[l] = (float*)(p + o_dzf[l]); tr->dX[l] = (float*)(p + o_dX[l]);
        tr->g_c[l] = (float*)(p + o_gc[l]); tr->g_hp[l] = (float*)(p + o_ghp[l]);
    }
    tr->dq = (float*)(p + o_dq); tr->dqf = (float*)(p + o_dqf);
    for (int i = 0; i < np; ++i) { tr->dpre[i] = (float*)(p + o_dpre[i]); tr->dpref[i] = (float*)(p + o_dpref[i]); }
    tr->xin = (float*)(p + o_xin); tr->dy2 = (float*)(p + o_dy2); tr->dhq = (float*)(p + o_dhq); tr->dxp = (float*)(p + o_dxp);
    tr->g_state = (float*)(p + o_gs); tr->dkeys = (float*)(p + o_dk); tr->dvalues = (float*)(p + o_dv);
    tr->p_va = (float*)(p + o_pva); tr->part_dva = (float*)(p + o_pdva);
    tr->tmp = (float*)(p + o_tmp); tr->loss3 = (float*)(p + o_loss); tr->dpre_tmp = (float*)(p + o_dpt);
    tr->dy2_all = (float*)(p + o_dy2a); tr->dzf2b = (float*)(p + o_dzf2b);
    tr->de_all = (float*)(p + o_de); tr->dctx_all = (float*)(p + o_dctx); tr->part_dq = (float*)(p + o_pdq); tr->part_ds = (float*)(p + o_pds);
    tr->ab_image = (float*)(p + o_img);
    {   // tape-wide du / f / df of the attention backward: written row by row every call, so outside the zero-filled slab
        const size_t rows = SB * T, nU = align_up(rows * U * 4, 256), nF = align_up(rows * d.attention_filters * 4, 256);
        char* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, nU + 2 * nF);
        if (e != hipSuccess) return fail(T2_ENOMEM, "hipMalloc(attention backward tape, %zu bytes) failed: %s", nU + 2 * nF, hipGetErrorString(e));
        tr->du_all = (float*)q; tr->f_all = (float*)(q + nU); tr->df_all = (float*)(q + nU + nF);
        if (rows > (size_t)INT_MAX) return fail(T2_EINVAL, "S * B * T_in too large");
    }
    const size_t lds = (size_t)attn_bwd_lds_floats(U, E, d.attention_filters, d.attention_kernel) * 4;
    if (lds > 150 * 1024) return fail(T2_EINVAL, "attention dimensions too large for the attention backward kernels' LDS");
    HIP_TRY(hipFuncSetAttribute((const void*)attention_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    T2_TRY(pzb_setup(tr));
    return T2_OK;
}

extern "C" int t2_trainer_bind_gradient(t2_trainer* tr, const char* name, int64_t offset) {
    if (!tr || !name || offset < 0 || (offset & 3) != 0) return fail(T2_EINVAL, "bad argument (offsets are multiples of 4 floats)");
    tr->offsets[name] = offset;
    return T2_OK;
}

static int bn_part_reserve(t2_trainer* tr, int C) {
    if (C <= tr->bn_cols) return T2_OK;
    if (tr->bn_part) { HIP_TRY(hipStreamSynchronize(tr->m->ctx->stream)); (void)hipFree(tr->bn_part); tr->bn_part = nullptr; }
    tr->bn_cols = std::max(C, 512);
    HIP_TRY(hipMalloc((void**)&tr->bn_part, (size_t)T2_BN_CHUNKS * 2 * tr->bn_cols * sizeof(double)));
    return T2_OK;
}

// out[N] = column sums of x [rows][N]
static int col_sum(t2_trainer* tr, const float* x, long long rows, int N, float* out) {
    if (N > T2_CS_PART_FLOATS / T2_CS_CHUNKS) return fail(T2_EINVAL, "internal: col_sum of %d columns", N);
    if (!tr->cs_part) HIP_TRY(hipMalloc((void**)&tr->cs_part, (size_t)T2_CS_PART_FLOATS * 4));
    const int cb = ceil_div(N, 64);
    const int chunks = (int)std::max<long long>(1, std::min<long long>({(long long)std::max(T2_CS_CHUNKS, 2048 / cb), (long long)T2_CS_PART_FLOATS / N, (rows + 63) / 64}));
    hipStream_t st = tr->m->ctx->stream;
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3(cb, chunks), dim3(256), 0, st, x, rows, N, tr->cs_part);
    hipLaunchKernelGGL(batch_sum_kernel, dim3(std::max(1, std::min(1024, ceil_div(N, 256)))), dim3(256), 0, st, tr->cs_part, chunks, (long long)N, out);
    return T2_OK;
}

// y[B][N] = x_frag[B][K] . W^T through the skinny core (no control block: t = 0), epilogue per BwdEpi
static void bwd_dense_epi(t2_trainer* tr, const Packed& wT, float* x_frag, BwdEpi e) {
    const int MB = mb_for(tr->B);
    SkinnyCore c{};
    c.w[0] = c.w[1] = wT.w; c.bias[0] = c.bias[1] = nullptr;
    c.kb0 = 0; c.kbtot = wT.kbtot; c.kb_w0 = 0; c.kb_full = wT.kbtot; c.ctl = nullptr; c.step_off = 0; c.t_hint = 0; c.B = tr->B;
    c.seg[0][1] = FragBuf{x_frag, 0, 1, 0, 0};
    hipStream_t st = tr->m->ctx->stream;
    const int G = MB >= 4 ? T2_G4 : 4;
    const bool split = tr->sk_part != nullptr && wT.kbtot >= 128 && wT.kbtot % (2 * G) == 0 && wT.n_wg <= T2_SK_MAX_TILES && tr->m->opt_bwd_splitk;
    const int per_wave = ((split ? wT.kbtot / 2 : wT.kbtot) / G) / T2_SKINNY_WAVES;
    const dim3 grid(wT.n_wg, split ? 2 : 1), blk(T2_SKINNY_THREADS);
    float* sp = tr->sk_part; unsigned* tk = tr->sk_ticket;
    if (split) {
        switch (MB) {
            case 1:
                if (per_wave == 4) hipLaunchKernelGGL((bwd_dense_kernel<1, 4, true>), grid, blk, 0, st, c, e, sp, tk);
                else hipLaunchKernelGGL((bwd_dense_kernel<1, 0, true>), grid, blk, 0, st, c, e, sp, tk);
                break;
            case 2:
                if (per_wave == 4) hipLaunchKernelGGL((bwd_dense_kernel<2, 4, true>), grid, blk, 0, st, c, e, sp, tk);
                else hipLaunchKernelGGL((bwd_dense_kernel<2, 0, true>), grid, blk, 0, st, c, e, sp, tk);
                break;
            default:
                if (per_wave == 8) hipLaunchKernelGGL((bwd_dense_kernel<4, 8, true>), grid, blk, 0, st, c, e, sp, tk);
                else hipLaunchKernelGGL((bwd_dense_kernel<4, 0, true>), grid, blk, 0, st, c, e, sp, tk);
                break;
        }
        return;
    }
    switch (MB) {
        case 1:
            if (per_wave == 8) hipLaunchKernelGGL((bwd_dense_kernel<1, 8>), grid, blk, 0, st, c, e, sp, tk);
            else hipLaunchKernelGGL((bwd_dense_kernel<1, 0>), grid, blk, 0, st, c, e, sp, tk);
            break;
        case 2:
            if (per_wave == 8) hipLaunchKernelGGL((bwd_dense_kernel<2, 8>), grid, blk, 0, st, c, e, sp, tk);
            else hipLaunchKernelGGL((bwd_dense_kernel<2, 0>), grid, blk, 0, st, c, e, sp, tk);
            break;
        default:
            if (per_wave == 16) hipLaunchKernelGGL((bwd_dense_kernel<4, 16>), grid, blk, 0, st, c, e, sp, tk);
            else hipLaunchKernelGGL((bwd_dense_kernel<4, 0>), grid, blk, 0, st, c, e, sp, tk);
            break;
    }
}

static void bwd_dense(t2_trainer* tr, const Packed& wT, float* x_frag, float* y, int N) {
    BwdEpi e{};
    e.N = N; e.out_plain = y; e.plain_ld = N; e.mode = 0;
    bwd_dense_epi(tr, wT, x_frag, e);
}

// Weight-gradient GEMM (trans_a mode of conv_gemm_kernel).  The reduction runs over all S*B (or B*T) tape rows while the
// output is small, so the reduction is split over gridDim.z until about four workgroups per CU are in flight; the partial
// products are added in split order by batch_sum_kernel.
#define T2_WG_PART_FLOATS (16u << 20)
static int launch_wgrad(t2_trainer* tr, ConvArgs a) {
    hipStream_t st = tr->m->ctx->stream;
    const int gx = ceil_div(a.N, CG_BN), gy = ceil_div(a.Mtot, CG_BM), nk = ceil_div(a.red, CG_BK);
    long long z = std::min<long long>({(long long)ceil_div(1024, gx * gy), (long long)nk / 8, (long long)(T2_WG_PART_FLOATS / ((size_t)a.Mtot * a.N)), 1024LL});
    if (a.ldy != a.N || z < 2) z = 1;
    a.splits = (int)z; a.part = nullptr;
    if (z > 1) {
        if (!tr->wg_part) HIP_TRY(hipMalloc((void**)&tr->wg_part, (size_t)T2_WG_PART_FLOATS * 4));
        a.part = tr->wg_part;
    }
    if (a.trans_a && a.vec_a && a.vec_b && !a.scale && !a.shift && !a.res && a.act == 0 && !tr->m->opt_conv_generic)
        hipLaunchKernelGGL(conv_gemm_wgrad_fast_kernel, dim3(gx, gy, (unsigned)z), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(conv_gemm_kernel, dim3(gx, gy, (unsigned)z), dim3(256), 0, st, a);
    if (z > 1) {
        const long long n = (long long)a.Mtot * a.N;
        hipLaunchKernelGGL(batch_sum_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))), dim3(256), 0, st,
                           tr->wg_part, (int)z, n, a.y);
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// dst[rows][ncols] (row stride ldd) = A^T . Bm with A [red][lda] (rows columns used), Bm [red][ldb]: weight gradients
static int bwd_wgrad(t2_trainer* tr, const float* A, int lda, int rows, const float* Bm, int ldb, int ncols, long long red, float* dst, int ldd) {
    ConvArgs a{};
    a.x = A; a.ldx = lda; a.w = Bm; a.scale = nullptr; a.shift = nullptr; a.res = nullptr; a.ldr = 0; a.y = dst; a.ldy = ldd;
    a.Mtot = rows; a.T = (int)red; a.Cin = rows; a.N = ncols; a.ktaps = 1; a.pad_l = 0; a.act = 0; a.trans_a = 1; a.red = (int)red;
    a.vec_a = (rows % 4 == 0) && (lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
    a.vec_b = (ncols % 4 == 0) && (ldb == ncols) && (((uintptr_t)Bm) % 16 == 0);
    if (ldb != ncols) return fail(T2_EINVAL, "internal: wgrad right operand must be dense");
    return launch_wgrad(tr, a);
}

static int grad_dst(t2_trainer* tr, float* flat, const std::string& name, float** out) {
    auto it = tr->offsets.find(name);
    if (it == tr->offsets.end()) return fail(T2_ESTATE, "no gradient offset bound for '%s' (t2_trainer_bind_gradient)", name.c_str());
    *out = flat + it->second;
    return T2_OK;
}

// Backward of t2_train_decoder_forward + the two losses of models.py:58-61.  done / bin_mask dev [B][S], spec_mask dev
// [B][S*r].  Writes the gradient of every decoder / attention variable into flat_grads at the bound offsets (TF layouts),
// d(loss)/d(memory) into dmemory [B][T_in][E] and {mel_loss, done_loss, loss} into loss_out (dev [3]).
extern "C" int t2_train_decoder_backward(t2_trainer* tr, const float* done, const float* spec_mask, const float* bin_mask,
                                         float* flat_grads, float* dmemory, float* loss_out) {
    if (!tr || !done || !spec_mask || !bin_mask || !flat_grads || !dmemory || !loss_out) return fail(T2_EINVAL, "null argument");
    if (!tr->have_forward) return fail(T2_ESTATE, "t2_train_decoder_forward has not run");
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    if (!tr->bslab) T2_TRY(trainer_setup_backward(tr));
    T2_TRY(trainer_sync_params(tr));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    t2_state* s = tr->state;
    const TrainTape& tp = tr->tape;
    const int B = tr->B, T = tr->T, S = tr->S, MB = mb_for(B);
    const int E = d.enc_units, U = d.attention_dim, A = d.attention_lstm_units, D = d.decoder_lstm_units;
    const int M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, np = d.num_prenet_layers, N = M * r + 1;
    const int F = d.attention_filters, K = d.attention_kernel;
    const int P_last = d.prenet_units[np - 1];
    const int H[3] = {A, D, D};
    const int KX[3] = {P_last + E + A, A + E + D, D + D};
    const int stream_of[3] = {T2_STREAM_ZONE_ATT, T2_STREAM_ZONE_L1, T2_STREAM_ZONE_L2};
    const size_t SB = (size_t)S * B;
    HIP_TRY(hipMemsetAsync(tr->bslab, 0, tr->bslab_bytes, st));
    // arrival tickets of the in-launch combines (bwd_dense_kernel SPLIT, attention_bwd_kernel): the last arriver clears its own,
    // this clears whatever a failed launch may have left
    HIP_TRY(hipMemsetAsync(tr->sk_ticket, 0, (T2_SK_MAX_TILES + 64) * 4, st));
    // losses (T1, T2) and their gradients
    float* l3 = tr->loss3;
    T2_TRY(t2_spec_loss(m->ctx, s->o_mel, s->i_targets, spec_mask, B, S * r, M, l3));
    T2_TRY(t2_binary_loss(m->ctx, s->o_stop, done, bin_mask, B, S, l3 + 4));
    hipLaunchKernelGGL(loss_grad_kernel, dim3(std::min<long long>(2048, (SB * N + 255) / 256)), dim3(256), 0, st, s->o_mel, s->i_targets,
                       spec_mask, s->o_stop, done, bin_mask, tr->glob_counts ? tr->glob_counts : l3 + 1, tr->glob_counts ? tr->glob_counts + 1 : l3 + 5, tr->dY, tr->dYf, (long long)MB * 16 * pad16(N), MB, B, S, M, r);
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(1024, (n + 255) / 256))); };
    const size_t lds = (size_t)attn_bwd_lds_floats(U, E, F, K) * 4;
    const int NJ = tr->NJ;
    hipLaunchKernelGGL(attention_image_kernel, dim3(1), dim3(256), 0, st, m->conv_w, m->conv_b, m->wloc, m->v_a, tr->ab_image, U, F, K);
    // Nothing of the loss gradient depends on the recurrence: d y2 (projection part) of all steps is one GEMM
    // (rnn_wrappers.py:105-109 backward), and LSTM 2's cell backward of step t - 1 rides in the epilogue of step t's
    // d [y1 | h2_prev] product (its columns [D, 2D) are the state gradient it waits for); the last step's runs on its own.
    {
        ConvLayer L;
        L.w = tr->projTp; L.k = 1; L.cin = N; L.cout = D;
        T2_TRY(launch_conv(m, L, tr->dY, N, tr->dy2_all, D, (int)SB, (int)SB, 0, nullptr, 0));
    }
    float* const dzf2[2] = {tr->dzf[2], tr->dzf2b};
    auto lstm_args_at = [&](int l, int tt, int first_, const float* go0, int ld0, const float* gh1, int ldh1, float* dz_frag) {
        LstmBwdArgs a{};
        a.z = tp.z[l] + (size_t)tt * B * 4 * H[l]; a.cprev = tp.cprev[l] + (size_t)tt * B * H[l]; a.cnew = tp.cnew[l] + (size_t)tt * B * H[l];
        a.g_out0 = go0; a.ld_out0 = ld0; a.g_out1 = nullptr; a.ld_out1 = 0;
        a.g_hs0 = tr->g_hp[l]; a.ld_hs0 = H[l]; a.g_hs1 = gh1; a.ld_hs1 = ldh1;
        a.g_c = tr->g_c[l]; a.g_hp = tr->g_hp[l];
        a.dz_plain = tr->dz[l] + (size_t)tt * B * 4 * H[l]; a.dz_frag = dz_frag;
        a.B = B; a.H = H[l]; a.MB = MB; a.t = tt; a.stream = stream_of[l]; a.seed = tp.seed;
        a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias; a.first = first_;
        return a;
    };
    // the reverse-time recurrence: ONE persistent launch at the paper-size class (persist_bwd.h), else 5 launches per step
    bool pzb = m->opt_bwd_persist && m->persist_holdoff == 0 && tr->pzb_slab != nullptr;
    if (pzb) {
        bool timed_out = false;
        T2_TRY(pzb_launch(tr, &timed_out));
        if (timed_out) pzb = false;
    }
    if (!pzb) hipLaunchKernelGGL(lstm_bwd_kernel, grid1((long long)B * D), dim3(256), 0, st,
                            lstm_args_at(2, S - 1, 1, tr->dy2_all + (size_t)(S - 1) * B * D, D, nullptr, 0, dzf2[(S - 1) & 1]));
    for (int t = S - 1; !pzb && t >= 0; --t) {
        const int first = (t == S - 1) ? 1 : 0;
        auto lstm_args = [&](int l, const float* go0, int ld0, const float* gh1, int ldh1) {
            return lstm_args_at(l, t, first, go0, ld0, gh1, ldh1, tr->dzf[l]);
        };
        // Every elementwise stage rides in the epilogue of the GEMM that produces its input gradient.
        // d [y1 | h2_prev] -> LSTM 1 backward (input [h_att | ctx | h1_prev]) and LSTM 2 backward of step t - 1
        {
            BwdEpi e{};
            e.N = KX[2]; e.out_plain = tr->dX[2]; e.plain_ld = KX[2]; e.mode = 1;
            e.lb = lstm_args(1, nullptr, 0, tr->dX[1] + A + E, KX[1]);
            if (t > 0) {
                e.mode_b = 3; e.col0_b = D;
                e.lb2 = lstm_args_at(2, t - 1, 0, tr->dy2_all + (size_t)(t - 1) * B * D, D, nullptr, 0, dzf2[(t - 1) & 1]);
            }
            bwd_dense_epi(tr, tr->lstmT[2], dzf2[t & 1], e);
        }
        bwd_dense(tr, tr->lstmT[1], tr->dzf[1], tr->dX[1], KX[1]);
        // attention
        {
            AttnBwdArgs a{};
            a.keys = s->keys; a.values = s->values; a.lens = s->lens;
            a.q = tp.q + (size_t)t * B * U; a.state_prev = tp.state_prev + (size_t)t * B * T; a.align = s->o_hist + (size_t)t * B * T;
            a.ctx = tp.ctx + (size_t)t * B * E;
            a.dctx0 = tr->dX[1] + A; a.ld_dctx0 = KX[1];
            a.dctx1 = first ? nullptr : tr->dX[0] + P_last; a.ld_dctx1 = KX[0];
            a.g_state = tr->g_state; a.image = tr->ab_image;
            a.de = tr->de_all + (size_t)t * B * T; a.dctx_sum = tr->dctx_all + (size_t)t * E; a.ld_dctx_sum = (long long)S * E;
            a.part_dq = tr->part_dq; a.part_dva = tr->part_dva; a.part_ds = tr->part_ds;
            a.du_out = tr->du_all + (size_t)t * B * T * U; a.f_out = tr->f_all + (size_t)t * B * T * F; a.df_out = tr->df_all + (size_t)t * B * T * F;
            a.B = B; a.T = T; a.U = U; a.E = E; a.F = F; a.K = K; a.NJ = NJ; a.first = first;
            a.ticket = tr->sk_ticket + T2_SK_MAX_TILES; a.dq_plain = tr->dq + (size_t)t * B * U; a.dq_frag = tr->dqf; a.g_state_out = tr->g_state;
            a.p_va = tr->p_va; a.MB = MB; a.cumulative = d.cumulative_weights;
            hipLaunchKernelGGL(attention_bwd_kernel, dim3(NJ, B), dim3(T2_AB_THREADS), lds, st, a);
        }
        // query layer -> d h_att (second source), attention LSTM backward (input [prenet out | ctx_prev | h_att_prev]; its raw
        // output feeds LSTM 1 and the query layer)
        {
            BwdEpi e{};
            e.N = A; e.out_plain = nullptr; e.mode = 1;
            e.lb = lstm_args(0, tr->dX[1], KX[1], tr->dX[0] + P_last + E, KX[0]);
            bwd_dense_epi(tr, tr->wqT, tr->dqf, e);
        }
        // d [prenet out | ctx_prev | h_att_prev], then the PreNets
        const float inv_keep = 1.0f / (1.0f - tp.prenet_drop_rate);
        auto prenet_epi = [&](BwdEpi& e, int i) {
            const int P = d.prenet_units[i];
            e.mode = 2; e.pn_y = tp.xp[i] + (size_t)t * B * P; e.pn_inv_keep = inv_keep;
            e.pn_plain = tr->dpre[i] + (size_t)t * B * P; e.pn_frag = tr->dpref[i]; e.pn_P = P;
        };
        {
            BwdEpi e{};
            e.N = KX[0]; e.out_plain = tr->dX[0]; e.plain_ld = KX[0];
            prenet_epi(e, np - 1);
            bwd_dense_epi(tr, tr->lstmT[0], tr->dzf[0], e);
        }
    }
    HIP_TRY(hipGetLastError());
    // the PreNet layers below the last one do not feed the recurrence: their data gradients are one GEMM over all S*B tape rows
    // each, d x_{i-1} = dropout'(relu'( d pre_i . W_i^T )) (modules.py:88-91 backward)
    for (int i = np - 1; i >= 1; --i) {
        const int Pi = d.prenet_units[i], Pp = d.prenet_units[i - 1];
        ConvLayer L;
        L.w = tr->prenetTp[i]; L.k = 1; L.cin = Pi; L.cout = Pp;
        T2_TRY(launch_conv(m, L, tr->dpre[i], Pi, tr->dpre_tmp, Pp, (int)SB, (int)SB, 0, nullptr, 0));
        hipLaunchKernelGGL(prenet_bwd_kernel, grid1((long long)SB * Pp), dim3(256), 0, st, tr->dpre_tmp, Pp, tp.xp[i - 1], 1.0f / (1.0f - tp.prenet_drop_rate),
                           tr->dpre[i - 1], (float*)nullptr, (int)SB, Pp, MB);
    }
    // ---- weight gradients: one GEMM per K segment over all S*B rows of the tape
    float* g = nullptr;
    const long long rows1 = (long long)(S - 1) * B;
    {   // projections
        T2_TRY(bwd_wgrad(tr, tp.hraw[2], D, D, tr->dY, N, N, SB, tr->tmp, N));
        T2_TRY(grad_dst(tr, flat_grads, "decoder/out_projection/kernel", &g));
        HIP_TRY(hipMemcpy2DAsync(g, (size_t)M * r * 4, tr->tmp, (size_t)N * 4, (size_t)M * r * 4, D, hipMemcpyDeviceToDevice, st));
        T2_TRY(grad_dst(tr, flat_grads, "decoder/stop_token_projection/kernel", &g));
        HIP_TRY(hipMemcpy2DAsync(g, 4, tr->tmp + M * r, (size_t)N * 4, 4, D, hipMemcpyDeviceToDevice, st));
        T2_TRY(col_sum(tr, tr->dY, (long long)SB, N, tr->dYf));
        T2_TRY(grad_dst(tr, flat_grads, "decoder/out_projection/bias", &g));
        HIP_TRY(hipMemcpyAsync(g, tr->dYf, (size_t)M * r * 4, hipMemcpyDeviceToDevice, st));
        T2_TRY(grad_dst(tr, flat_grads, "decoder/stop_token_projection/bias", &g));
        HIP_TRY(hipMemcpyAsync(g, tr->dYf + M * r, 4, hipMemcpyDeviceToDevice, st));
    }
    const char* lname[3] = {"decoder/attention_lstm", "decoder/lstm_1", "decoder/lstm_2"};
    for (int l = 0; l < 3; ++l) {
        const int n4 = 4 * H[l];
        T2_TRY(grad_dst(tr, flat_grads, std::string(lname[l]) + "/kernel", &g));
        const float* dz = tr->dz[l];
        const float* dz1 = dz + (size_t)B * n4;            // rows of steps 1 .. S-1 (operands taken from step t - 1)
        if (l == 0) {
            T2_TRY(bwd_wgrad(tr, tp.xp[np - 1], P_last, P_last, dz, n4, n4, SB, g, n4));
            if (S > 1) {
                T2_TRY(bwd_wgrad(tr, tp.ctx, E, E, dz1, n4, n4, rows1, g + (size_t)P_last * n4, n4));
                T2_TRY(bwd_wgrad(tr, tp.hstate[0], A, A, dz1, n4, n4, rows1, g + (size_t)(P_last + E) * n4, n4));
            }
        } else if (l == 1) {
            T2_TRY(bwd_wgrad(tr, tp.hraw[0], A, A, dz, n4, n4, SB, g, n4));
            T2_TRY(bwd_wgrad(tr, tp.ctx, E, E, dz, n4, n4, SB, g + (size_t)A * n4, n4));
            if (S > 1) T2_TRY(bwd_wgrad(tr, tp.hstate[1], D, D, dz1, n4, n4, rows1, g + (size_t)(A + E) * n4, n4));
        } else {
            T2_TRY(bwd_wgrad(tr, tp.hraw[1], D, D, dz, n4, n4, SB, g, n4));
            if (S > 1) T2_TRY(bwd_wgrad(tr, tp.hstate[2], D, D, dz1, n4, n4, rows1, g + (size_t)D * n4, n4));
        }
        T2_TRY(grad_dst(tr, flat_grads, std::string(lname[l]) + "/bias", &g));
        T2_TRY(col_sum(tr, dz, (long long)SB, n4, g));
    }
    // query layer, attention parameters
    T2_TRY(grad_dst(tr, flat_grads, "attention/query_layer/kernel", &g));
    T2_TRY(bwd_wgrad(tr, tp.hraw[0], A, A, tr->dq, U, U, SB, g, U));
    T2_TRY(grad_dst(tr, flat_grads, "attention/attention_bias", &g));
    T2_TRY(col_sum(tr, tr->dq, (long long)SB, U, g));
    {   // tape-wide attention gradients from the taped du / f / df (rows (t, b, j))
        const long long rows = (long long)SB * T;
        hipLaunchKernelGGL(batch_sum_kernel, grid1((long long)B * T * U), dim3(256), 0, st, tr->du_all, S, (long long)B * T * U, tr->dkeys);
        T2_TRY(grad_dst(tr, flat_grads, "attention/attention_variable", &g));
        if (pzb) hipLaunchKernelGGL(batch_sum_kernel, grid1(U), dim3(256), 0, st, tr->pzb_pva, pz::NWG, (long long)U, g);
        else hipLaunchKernelGGL(batch_sum_kernel, grid1(U), dim3(256), 0, st, tr->p_va, B, (long long)U, g);
        T2_TRY(grad_dst(tr, flat_grads, "attention/location_features_layer/kernel", &g));
        T2_TRY(bwd_wgrad(tr, tr->f_all, F, F, tr->du_all, U, U, rows, g, U));
        T2_TRY(grad_dst(tr, flat_grads, "attention/location_features_convolution/kernel", &g));
        if (K <= 32 && F <= 32) {
            // [K][1][F] = im2col(state_prev)^T . df on the dedicated kernel (the generic weight-gradient GEMM took 0.8 ms here)
            const int nblk = 1024;
            if (!tr->wg_part) HIP_TRY(hipMalloc((void**)&tr->wg_part, (size_t)T2_WG_PART_FLOATS * 4));
            hipLaunchKernelGGL(loc_conv_wgrad_kernel, dim3(nblk), dim3(256), 0, st, tp.state_prev, tr->df_all, (long long)SB, T, K, F, tr->wg_part);
            hipLaunchKernelGGL(batch_sum_kernel, dim3(ceil_div(K * F, 256)), dim3(256), 0, st, tr->wg_part, nblk, (long long)K * F, g);
        } else {   // [K][1][F] = im2col(state_prev)^T . df, taps reaching outside a row's [0, T) read zero
            ConvArgs a{};
            a.x = tp.state_prev; a.ldx = 1; a.w = tr->df_all; a.y = g; a.ldy = F;
            a.Mtot = K; a.T = T; a.Cin = 1; a.N = F; a.ktaps = K; a.pad_l = (K - 1) / 2; a.act = 0; a.trans_a = 1; a.red = (int)rows;
            a.vec_a = 0; a.vec_b = (F % 4 == 0) && (((uintptr_t)tr->df_all) % 16 == 0);
            T2_TRY(launch_wgrad(tr, a));
        }
        T2_TRY(grad_dst(tr, flat_grads, "attention/location_features_convolution/bias", &g));
        T2_TRY(col_sum(tr, tr->df_all, rows, F, g));
        // d values[b][j][:] = sum_t a_t[b][j] * dctx_t[b][:]: one [T x S] . [S x E] product per row
        hipLaunchKernelGGL(dvalues_kernel, dim3(ceil_div(T, 16), B, ceil_div(E, 512)), dim3(512), 0, st, s->o_hist, tr->dctx_all, B, T, S, E, tr->dvalues);
    }
    // PreNets: inputs of layer 0 are the teacher-forced frames (go frame at t = 0)
    {
        hipLaunchKernelGGL(feed_gather_kernel, grid1((long long)SB * M * nf), dim3(256), 0, st, s->i_targets, tr->xin, B, S, M, r, nf);
    }
    for (int i = 0; i < np; ++i) {
        const int P = d.prenet_units[i];
        const int Kin = i == 0 ? M * nf : d.prenet_units[i - 1];
        const float* x = i == 0 ? tr->xin : tp.xp[i - 1];
        const std::string pn = "decoder/prenet_" + std::to_string(i) + "/dense";
        T2_TRY(grad_dst(tr, flat_grads, pn + "/kernel", &g));
        T2_TRY(bwd_wgrad(tr, x, Kin, Kin, tr->dpre[i], P, P, SB, g, P));
        T2_TRY(grad_dst(tr, flat_grads, pn + "/bias", &g));
        T2_TRY(col_sum(tr, tr->dpre[i], (long long)SB, P, g));
    }
    // memory layer and d(memory): keys = values . Wmem, values = memory * mask
    T2_TRY(grad_dst(tr, flat_grads, "attention/memory_layer/kernel", &g));
    T2_TRY(bwd_wgrad(tr, s->values, E, E, tr->dkeys, U, U, (long long)B * T, g, U));
    {
        ConvLayer L;
        L.w = tr->wmemT; L.k = 1; L.cin = U; L.cout = E;
        T2_TRY(launch_conv(m, L, tr->dkeys, U, dmemory, E, B * T, B * T, 0, tr->dvalues, E));
        hipLaunchKernelGGL(mask_rows_kernel, grid1((long long)B * T * E), dim3(256), 0, st, dmemory, s->lens, B, T, E);
    }
    // {mel_loss, done_loss, loss}
    HIP_TRY(hipMemcpyAsync(loss_out, l3 + 2, 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(loss_out + 1, l3 + 6, 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(batch_sum_kernel, dim3(1), dim3(256), 0, st, loss_out, 2, 1LL, loss_out + 2);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// ==========================================================================================================================
// Encoder: training-mode forward (batch-statistics batch norm, conv dropout, zoneout-sampled BiLSTM) and backward
// ==========================================================================================================================
#define T2_STREAM_ENC_CONV 20
#define T2_STREAM_ZONE_ENC 30

// per-channel batch statistics over all M rows (tf.layers.batch_normalization(training=True): padding rows included),
// biased variance.  Two stages, fixed summation order: blocks of 64 channels x 4 row lanes over T2_BN_CHUNKS row ranges write
// double partial sums part[chunk][2][C]; the final kernel adds the ranges in order.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int M, int C, double* __restrict__ part) {
    __shared__ double s1[4][64], s2[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int per = (M + T2_BN_CHUNKS - 1) / T2_BN_CHUNKS, m0 = blockIdx.y * per, m1 = min(M, m0 + per);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int m = m0 + rl; m < m1; m += 4) { const double v = x[(size_t)m * C + c]; a += v; b += v * v; }
    s1[rl][cl] = a; s2[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl];
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = s2[0][cl] + s2[1][cl] + s2[2][cl] + s2[3][cl];
    }
}

__global__ __launch_bounds__(256) void bn_stats_final_kernel(const double* __restrict__ part, int M, int C, float eps, float* __restrict__ mean,
                                                             float* __restrict__ invstd, float* __restrict__ var_out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double sum = 0.0, sq = 0.0;
    for (int k = 0; k < T2_BN_CHUNKS; ++k) { sum += part[((size_t)k * 2 + 0) * C + c]; sq += part[((size_t)k * 2 + 1) * C + c]; }
    const double mu = sum / M, var = fmax(sq / M - mu * mu, 0.0);
    mean[c] = (float)mu; var_out[c] = (float)var; invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
}

// Conv1d.call after the convolution (modules.py:152-156), training mode: batch norm with batch statistics -> relu -> dropout
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float rate, uint32_t seed,
                                                       int stream, float* __restrict__ out, long long M, int C, int act = 1) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int c = (int)(i - m * C);
        float v = (y[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        if (rate > 0.f) v = dropout_train(v, rate, t2_u01(seed, stream, 0, (uint32_t)m, c));
        out[i] = v;
    }
}

// d gamma / d beta of the layer above: g = d_out * (out != 0) / keep (relu and dropout), xhat = (y - mean) * invstd
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ d_out, const float* __restrict__ out, const float* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, float inv_keep,
                                                            int M, int C, double* __restrict__ part) {
    __shared__ double s1[4][64], s2[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int per = (M + T2_BN_CHUNKS - 1) / T2_BN_CHUNKS, m0 = blockIdx.y * per, m1 = min(M, m0 + per);
    double a = 0.0, b = 0.0;
    if (c < C) {
        const float mu = mean[c], is = invstd[c];
        for (int m = m0 + rl; m < m1; m += 4) {
            const size_t i = (size_t)m * C + c;
            const float g = (!out || out[i] != 0.f) ? d_out[i] * inv_keep : 0.f;      // out == null: identity activation
            a += (double)g * (double)((y[i] - mu) * is);
            b += (double)g;
        }
    }
    s1[rl][cl] = a; s2[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < C) {
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = s1[0][cl] + s1[1][cl] + s1[2][cl] + s1[3][cl];
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = s2[0][cl] + s2[1][cl] + s2[2][cl] + s2[3][cl];
    }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const double* __restrict__ part, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < T2_BN_CHUNKS; ++k) { a += part[((size_t)k * 2 + 0) * C + c]; b += part[((size_t)k * 2 + 1) * C + c]; }
    dgamma[c] = (float)a; dbeta[c] = (float)b;
}

// dy = gamma * invstd * (g - dbeta / M - xhat * dgamma / M): gradient of the convolution output through the batch statistics
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ d_out, const float* __restrict__ out, const float* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta, float inv_keep,
                                                           long long M, int C, float* __restrict__ dy) {
    const float invM = 1.0f / (float)M;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float g = (!out || out[i] != 0.f) ? d_out[i] * inv_keep : 0.f;
        const float xh = (y[i] - mean[c]) * invstd[c];
        dy[i] = gamma[c] * invstd[c] * (g - dbeta[c] * invM - xh * dgamma[c] * invM);
    }
}

// x [B][T][C] -> time-major per-step rows xs[s][b][:] in the order direction `reverse` consumes them (zeros past the length)
__global__ __launch_bounds__(256) void seq_gather_kernel(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ xs,
                                                         int B, int T, int C, int reverse) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)T * B * C; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), b = (int)((i / C) % B), s = (int)(i / ((long long)C * B));
        const int len = lens[b];
        const int pos = reverse ? len - 1 - s : s;
        xs[i] = (s < len) ? x[((size_t)b * T + pos) * C + c] : 0.f;
    }
}

// gradient of the raw output of every step of one direction: g[s][b][u] = dmemory[b][pos][off + u] for live rows
__global__ __launch_bounds__(256) void enc_gout_kernel(const float* __restrict__ dmem, const int* __restrict__ lens, float* __restrict__ g,
                                                       int B, int T, int E, int H, int off, int reverse) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)T * B * H; i += (long long)gridDim.x * blockDim.x) {
        const int u = (int)(i % H), b = (int)((i / H) % B), s = (int)(i / ((long long)H * B));
        const int len = lens[b];
        const int pos = reverse ? len - 1 - s : s;
        g[i] = (s < len) ? dmem[((size_t)b * T + pos) * E + off + u] : 0.f;
    }
}

// d(conv stack output)[b][pos][:] += dX[s][b][:C] over all steps s (live rows; every (b, pos) has one step per direction)
__global__ __launch_bounds__(256) void enc_dx_scatter_kernel(const float* __restrict__ dX, int ld, const int* __restrict__ lens, float* __restrict__ dxc,
                                                             int B, int T, int C, int reverse) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)T * B * C; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), b = (int)((i / C) % B), s = (int)(i / ((long long)C * B));
        const int len = lens[b];
        if (s >= len) continue;
        const int pos = reverse ? len - 1 - s : s;
        dxc[((size_t)b * T + pos) * C + c] += dX[((size_t)s * B + b) * ld + c];
    }
}

// d embedding[sym][:] = sum over the rows m with ids[m] == sym of dx[m][:]: one block per symbol.  The block first compacts the
// matching row numbers into LDS in row order (ballot + prefix counts per 256 rows), then every thread sums its columns over
// that list: the rows are added in increasing order whatever the block's scheduling (deterministic).
__global__ __launch_bounds__(256) void embedding_grad_kernel(const int* __restrict__ ids, const float* __restrict__ dx, int M, int C, float* __restrict__ demb) {
    constexpr int CAP = 2048;
    __shared__ int s_list[CAP + 256];
    __shared__ int s_wave[4];
    __shared__ int s_n;
    const int sym = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int CMAX = 8;                              // columns per thread: C <= 2048
    float acc[CMAX];
#pragma unroll
    for (int k = 0; k < CMAX; ++k) acc[k] = 0.f;
    if (tid == 0) s_n = 0;
    __syncthreads();
    auto flush = [&]() {
        const int n = s_n;
        int i = 0;
        for (; i + 4 <= n; i += 4) {                 // four rows in flight, added in list order
            float v[4][CMAX];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* row = dx + (size_t)s_list[i + j] * C;
#pragma unroll
                for (int k = 0; k < CMAX; ++k) { const int c = tid + k * 256; v[j][k] = c < C ? row[c] : 0.f; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < CMAX; ++k) acc[k] += v[j][k];
        }
        for (; i < n; ++i) {
            const float* row = dx + (size_t)s_list[i] * C;
#pragma unroll
            for (int k = 0; k < CMAX; ++k) { const int c = tid + k * 256; if (c < C) acc[k] += row[c]; }
        }
    };
    for (int m0 = 0; m0 < M; m0 += 256) {
        const int m = m0 + tid;
        const bool hit = m < M && ids[m] == sym;
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int base = s_n;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        if (hit) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = m;
        __syncthreads();
        if (tid == 0) s_n += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        if (s_n >= CAP) { flush(); __syncthreads(); if (tid == 0) s_n = 0; __syncthreads(); }
    }
    flush();
#pragma unroll
    for (int k = 0; k < CMAX; ++k) { const int c = tid + k * 256; if (c < C) demb[(size_t)sym * C + c] = acc[k]; }
}

struct EncTape {
    void* slab = nullptr; size_t bytes = 0;
    int* ids = nullptr; int* lens = nullptr;
    std::vector<float*> x;            // x[0] = embedding output [M][emb]; x[i + 1] = output of conv layer i after dropout [M][E]
    std::vector<float*> yraw, mean, invstd, var;
    float *z[2] = {}, *cprev[2] = {}, *cnew[2] = {}, *hstate[2] = {};     // BiLSTM tape per direction: [T][B][.]
    float *xf = nullptr, *hfrag = nullptr, *cst = nullptr; int* ctl = nullptr;
    float drop_rate = 0.f; uint32_t seed = 0;
    // backward
    std::vector<ConvLayer> convT;     // flipped / transposed conv kernels of the data gradient
    Packed lstmT[2];
    float *xs = nullptr, *gout = nullptr, *dzp = nullptr, *dzf = nullptr, *dX = nullptr, *g_c = nullptr, *g_hp = nullptr;
    float *dxa = nullptr, *dxb = nullptr, *dgamma = nullptr, *dbeta = nullptr;
    bool have_forward = false, have_bwd = false;
    // persistent forward / backward recurrences (persist.h encoder_kernel<., true>, persist_bwd.h encoder_bwd_kernel)
    unsigned* pz_bar = nullptr;       // forward: [2][NSHARD][32] counters + error word (in the tape slab)
    float* pze_w[2] = {};             // backward: column-owner tiles of the transposed kernels
    char* pze_slab = nullptr; size_t pze_slab_bytes = 0; unsigned pze_xs_bytes = 0; unsigned* pze_bar = nullptr;
    float *dzp2 = nullptr, *dX2 = nullptr;
};

static std::map<t2_trainer*, EncTape*> g_enc_tapes;      // trainer -> encoder tape (kept outside t2_trainer's layout)

static void enc_tape_release(t2_trainer* tr) {
    auto it = g_enc_tapes.find(tr);
    if (it == g_enc_tapes.end()) return;
    if (it->second->slab) (void)hipFree(it->second->slab);
    delete it->second;
    g_enc_tapes.erase(it);
}

static EncTape* enc_tape_of(t2_trainer* tr) {
    auto it = g_enc_tapes.find(tr);
    if (it != g_enc_tapes.end()) return it->second;
    EncTape* e = new EncTape();
    g_enc_tapes[tr] = e;
    return e;
}

static int enc_tape_alloc(t2_trainer* tr, EncTape* et) {
    t2_model* m = tr->m;
    const t2_dims& d = m->d;
    const int B = tr->B, T = tr->T, MB = mb_for(B), E = d.enc_units, He = E / 2, emb = d.embedding_dim, L = d.enc_conv_layers;
    const size_t Mtot = (size_t)B * T;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_ids = carve(Mtot * 4), o_lens = carve((size_t)B * 4);
    std::vector<size_t> o_x(L + 1), o_y(L), o_mu(L), o_is(L), o_var(L);
    o_x[0] = carve(Mtot * emb * 4);
    for (int i = 0; i < L; ++i) { o_x[i + 1] = carve(Mtot * E * 4); o_y[i] = carve(Mtot * E * 4); o_mu[i] = carve((size_t)E * 4); o_is[i] = carve((size_t)E * 4); o_var[i] = carve((size_t)E * 4); }
    size_t o_z[2], o_cp[2], o_cn[2], o_hs[2];
    for (int dir = 0; dir < 2; ++dir) {
        o_z[dir] = carve((size_t)T * B * 4 * He * 4); o_cp[dir] = carve((size_t)T * B * He * 4); o_cn[dir] = carve((size_t)T * B * He * 4);
        o_hs[dir] = carve((size_t)T * B * He * 4);
    }
    const int Cx = L > 0 ? E : emb;
    const int KPc = pad16(Cx), KPh = pad16(He);
    const size_t per_step = (size_t)MB * 16 * KPc;
    const size_t o_xf = carve(2 * (size_t)T * per_step * 4), o_h = carve(2 * 2 * (size_t)MB * 16 * KPh * 4), o_c = carve(2 * (size_t)MB * 16 * He * 4);
    const size_t o_ctl = carve(64);
    const size_t o_pzbar = carve((size_t)(2 * pz::NSHARD * 32 + 64) * 4);
    et->bytes = off;
    HIP_TRY(hipMalloc(&et->slab, off));
    char* p = (char*)et->slab;
    et->ids = (int*)(p + o_ids); et->lens = (int*)(p + o_lens);
    et->x.resize(L + 1); et->yraw.resize(L); et->mean.resize(L); et->invstd.resize(L); et->var.resize(L);
    et->x[0] = (float*)(p + o_x[0]);
    for (int i = 0; i < L; ++i) {
        et->x[i + 1] = (float*)(p + o_x[i + 1]); et->yraw[i] = (float*)(p + o_y[i]); et->mean[i] = (float*)(p + o_mu[i]);
        et->invstd[i] = (float*)(p + o_is[i]); et->var[i] = (float*)(p + o_var[i]);
    }
    for (int dir = 0; dir < 2; ++dir) {
        et->z[dir] = (float*)(p + o_z[dir]); et->cprev[dir] = (float*)(p + o_cp[dir]); et->cnew[dir] = (float*)(p + o_cn[dir]);
        et->hstate[dir] = (float*)(p + o_hs[dir]);
    }
    et->xf = (float*)(p + o_xf); et->hfrag = (float*)(p + o_h); et->cst = (float*)(p + o_c); et->ctl = (int*)(p + o_ctl);
    et->pz_bar = (unsigned*)(p + o_pzbar);
    return T2_OK;
}

// Embedding.call + EncoderV2.call with is_training = True (tacotron_v2.py:53-63; Conv1d.call modules.py:149-157):
// conv -> batch norm on the batch statistics of all B*T positions -> relu -> dropout(drop_rate); BiLSTM of ZoneoutLSTMCells in
// their training branch with sequence lengths.  ids host int64 [B][T_in]; memory_out dev [B][T_in][enc_units].
// bn_stats_out (dev, may be null): per conv layer [mean | biased variance], 2 * enc_units floats each (the moving-statistics
// update of the step, models.py:75, is the caller's: moving = 0.99 * moving + 0.01 * batch).
static int train_encoder_forward_impl(t2_trainer* tr, const int64_t* ids, const float* emb_in, const int64_t* lens, float drop_rate, uint32_t seed,
                                      float* memory_out, float* bn_stats_out) {
    if (!tr || (!ids && !emb_in) || !memory_out) return fail(T2_EINVAL, "null argument");
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return fail(T2_EINVAL, "drop rate must be in [0, 1)");
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = tr->B, T = tr->T, MB = mb_for(B), E = d.enc_units, He = E / 2, emb = d.embedding_dim, L = d.enc_conv_layers;
    const int Mtot = B * T;
    EncTape* et = enc_tape_of(tr);
    if (!et->slab) T2_TRY(enc_tape_alloc(tr, et));
    std::vector<int> h_lens(B), h_ids(Mtot);
    for (int b = 0; b < B; ++b) {
        const int64_t l = lens ? lens[b] : T;
        if (l < 0 || l > T) return fail(T2_EINVAL, "sequence length %lld outside 0..%d", (long long)l, T);
        h_lens[b] = (int)l;
    }
    for (int i = 0; ids && i < Mtot; ++i) {
        if (ids[i] < 0 || ids[i] >= d.num_symbols) return fail(T2_ERANGE, "embedding id %lld outside [0, %d) (modules.py:65-66)", (long long)ids[i], d.num_symbols);
        h_ids[i] = (int)ids[i];
    }
    HIP_TRY(hipMemcpyAsync(et->lens, h_lens.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(et->ids, h_ids.data(), (size_t)Mtot * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    et->drop_rate = drop_rate; et->seed = seed;
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); };
    if (ids) hipLaunchKernelGGL(embedding_kernel, grid1((long long)Mtot * emb), dim3(256), 0, st, et->ids, m->embedding, et->x[0], Mtot, emb);
    else HIP_TRY(hipMemcpyAsync(et->x[0], emb_in, (size_t)Mtot * emb * 4, hipMemcpyDeviceToDevice, st));
    int cin = emb;
    for (int i = 0; i < L; ++i) {
        ConvLayer raw = m->enc_convs[i];
        raw.scale = nullptr; raw.shift = nullptr;               // the batch statistics replace the folded moving statistics
        T2_TRY(launch_conv(m, raw, et->x[i], cin, et->yraw[i], E, Mtot, T, 0, nullptr, 0));
        T2_TRY(bn_part_reserve(tr, E));
        hipLaunchKernelGGL(bn_stats_kernel, dim3(ceil_div(E, 64), T2_BN_CHUNKS), dim3(256), 0, st, et->yraw[i], Mtot, E, tr->bn_part);
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, st, tr->bn_part, Mtot, E, d.bn_eps, et->mean[i], et->invstd[i], et->var[i]);
        hipLaunchKernelGGL(bn_apply_kernel, grid1((long long)Mtot * E), dim3(256), 0, st, et->yraw[i], et->mean[i], et->invstd[i],
                           m->enc_bn_gamma[i], m->enc_bn_beta[i], drop_rate, seed, T2_STREAM_ENC_CONV + i, et->x[i + 1], (long long)Mtot, E, 1);
        if (bn_stats_out) {
            HIP_TRY(hipMemcpyAsync(bn_stats_out + (size_t)i * 2 * E, et->mean[i], (size_t)E * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(bn_stats_out + (size_t)i * 2 * E + E, et->var[i], (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        }
        cin = E;
    }
    const int Cx = cin, KPc = pad16(Cx), KPh = pad16(He);
    const size_t per_step = (size_t)MB * 16 * KPc;
    hipLaunchKernelGGL(seq_to_frag_kernel, grid1(2LL * T * per_step), dim3(256), 0, st, et->x[L], et->lens, et->xf, B, T, Cx, MB, KPc);
    HIP_TRY(hipMemsetAsync(memory_out, 0, (size_t)Mtot * E * 4, st));
    HIP_TRY(hipMemsetAsync(et->hfrag, 0, 2 * 2 * (size_t)MB * 16 * KPh * 4, st));
    HIP_TRY(hipMemsetAsync(et->cst, 0, 2 * (size_t)MB * 16 * He * 4, st));
    const int h_ctl[8] = {0, 0, T, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(et->ctl, h_ctl, sizeof(h_ctl), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int hsz = MB * 16 * KPh;
    // the whole recurrence as ONE launch (persist.h encoder_kernel, training instantiation) at the paper-size class
    const bool enc_persist = m->opt_enc_persist && m->persist_holdoff == 0 && Cx == 512 && He == 256 && KPh == 256 && m->n_cu >= 2 * pz::ENC_G &&
                             m->enc_lstm[0].n_wg == pz::ENC_G && m->enc_lstm[0].kbtot == 48 && 2LL * T * (long long)per_step * 4 < 0x7fffffffLL;
    if (enc_persist) {
        HIP_TRY(hipMemsetAsync(et->pz_bar, 0, (size_t)(2 * pz::NSHARD * 32 + 64) * 4, st));
        pz::EArgs a{};
        for (int dir = 0; dir < 2; ++dir) {
            a.w[dir] = m->enc_lstm[dir].w; a.bias[dir] = m->enc_lstm[dir].bias;
            a.tz[dir] = et->z[dir]; a.tcprev[dir] = et->cprev[dir]; a.tcnew[dir] = et->cnew[dir]; a.thstate[dir] = et->hstate[dir];
        }
        a.xf = et->xf; a.xf_step = (unsigned)(per_step * 4); a.xf_bytes = (unsigned)(2LL * T * per_step * 4);
        a.hx = et->hfrag; a.hx_bytes = (unsigned)(2 * 2 * (size_t)hsz * 4);
        a.bar = et->pz_bar; a.err = (int*)(et->pz_bar + 2 * pz::NSHARD * 32);
        a.lens = et->lens; a.out = memory_out; a.B = B; a.T = T;
        a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias;
        a.seed = seed; a.stream = T2_STREAM_ZONE_ENC;
        const dim3 grid(2 * pz::ENC_G), blk(pz::NT);
        {
            XwgGate gate;
            T2_TRY(gate.begin(m->ctx));
            if (MB == 1) hipLaunchKernelGGL((pz::encoder_kernel<1, true>), grid, blk, 0, st, a);
            else if (MB == 2) hipLaunchKernelGGL((pz::encoder_kernel<2, true>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((pz::encoder_kernel<4, true>), grid, blk, 0, st, a);
            HIP_TRY(hipGetLastError());
        }
        T2_TRY(inject_timeout(m, a.err));
        ++m->n_enc_persist;
        int h_err = 0;
        HIP_TRY(hipMemcpyAsync(&h_err, a.err, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!h_err) { et->have_forward = ids != nullptr; return T2_OK; }
        // a hand-off wait ran out of time: the same recurrence again, one launch per step and direction
        note_persist_timeout(m);
        HIP_TRY(hipMemsetAsync(memory_out, 0, (size_t)Mtot * E * 4, st));
        HIP_TRY(hipMemsetAsync(et->hfrag, 0, 2 * 2 * (size_t)MB * 16 * KPh * 4, st));
        HIP_TRY(hipMemsetAsync(et->cst, 0, 2 * (size_t)MB * 16 * He * 4, st));
        HIP_TRY(hipMemcpyAsync(et->ctl, h_ctl, sizeof(h_ctl), hipMemcpyHostToDevice, st));
    }
    for (int s = 0; s < T; ++s)
        for (int dir = 0; dir < 2; ++dir) {
            SkinnyCore c{};
            LstmEpi e{};
            c.w[0] = c.w[1] = m->enc_lstm[dir].w; c.bias[0] = c.bias[1] = m->enc_lstm[dir].bias;
            c.seg[0][0] = FragBuf{et->xf + (size_t)dir * T * per_step, (int)per_step, T, 0, 0};
            c.seg[0][1] = FragBuf{et->hfrag + (size_t)dir * 2 * hsz, hsz, 2, 0, 0};
            c.kb0 = KPc / 16; c.kbtot = m->enc_lstm[dir].kbtot; c.kb_w0 = 0; c.kb_full = c.kbtot; c.ctl = et->ctl; c.step_off = s; c.t_hint = s; c.B = B;
            e.c[0] = et->cst + (size_t)dir * MB * 16 * He;
            e.hprev[0] = FragBuf{et->hfrag + (size_t)dir * 2 * hsz, hsz, 2, 0, 0};
            e.hstate[0] = FragBuf{et->hfrag + (size_t)dir * 2 * hsz, hsz, 2, 1, 0};
            e.hraw[0] = FragBuf{nullptr, 0, 1, 0, 0};
            e.plain_off[0] = dir * He; e.reverse[0] = dir;
            e.H = He; e.Hld = He; e.zc = d.zoneout_cell; e.zh = d.zoneout_output; e.fb = d.forget_bias;
            e.lens = et->lens; e.plain_out = memory_out; e.plain_bs = T * E; e.plain_ts = E;
            e.train = 1; e.seed = seed; e.stream = T2_STREAM_ZONE_ENC + 2 * dir;
            e.tp_z = et->z[dir]; e.tp_cprev = et->cprev[dir]; e.tp_cnew = et->cnew[dir]; e.tp_hraw = nullptr; e.tp_hstate = et->hstate[dir];
            launch_lstm<true>(MB, dim3(m->enc_lstm[dir].n_wg, 1), st, c, e);
        }
    HIP_TRY(hipGetLastError());
    et->have_forward = ids != nullptr;      // the backward pass needs the ids of the embedding lookup
    return T2_OK;
}

extern "C" int t2_train_encoder_forward(t2_trainer* tr, const int64_t* ids, const int64_t* lens, float drop_rate, uint32_t seed,
                                        float* memory_out, float* bn_stats_out) {
    if (!ids) return fail(T2_EINVAL, "ids is null");
    return train_encoder_forward_impl(tr, ids, nullptr, lens, drop_rate, seed, memory_out, bn_stats_out);
}

extern "C" int t2_train_encoder_forward_embedded(t2_trainer* tr, const float* emb, const int64_t* lens, float drop_rate, uint32_t seed,
                                                 float* memory_out, float* bn_stats_out) {
    if (!emb) return fail(T2_EINVAL, "emb is null");
    return train_encoder_forward_impl(tr, nullptr, emb, lens, drop_rate, seed, memory_out, bn_stats_out);
}

// PostNetV2.call with is_training=True (tacotron_v2.py:181-204 over Conv1d.call, modules.py:149-157): every convolution is
// followed by batch normalisation on the BATCH statistics (all B*T_out positions), its activation (tanh, none on the last
// layer) and dropout sampled from the counter-based generator (stream T2_STREAM_POST_CONV + layer); then Dense(out_units) and
// the residual.  Forward only: the reference's loss (models.py:58-61) does not contain the PostNet, so no gradient is needed.
// bn_stats_out (optional): dev [layers][2][channels] batch mean | biased variance.
#define T2_STREAM_POST_CONV 40
extern "C" int t2_train_postnet_forward(t2_model* m, const float* mel, int B, int T_out, float drop_rate, uint32_t seed, float* mel_post,
                                        float* bn_stats_out) {
    if (!m || !mel || !mel_post) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    if (B < 1 || T_out < 1) return fail(T2_EINVAL, "bad shape");
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return fail(T2_EINVAL, "drop rate must be in [0, 1)");
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int Mtot = B * T_out, C = d.postnet_channels, M = d.num_mels, L = d.postnet_layers;
    const size_t one = align_up((size_t)Mtot * C * 4, 256), stats = align_up((size_t)3 * C * 4, 256),
                 part = align_up((size_t)T2_BN_CHUNKS * 2 * C * sizeof(double), 256);
    T2_TRY(ensure_ws(&m->post_ws, &m->post_ws_bytes, 3 * one + stats + part));
    char* ws = (char*)m->post_ws;
    float* bufs[2] = {(float*)ws, (float*)(ws + one)};
    float* yraw = (float*)(ws + 2 * one);
    float *mean = (float*)(ws + 3 * one), *invstd = mean + C, *var = invstd + C;
    double* bn_part = (double*)(ws + 3 * one + stats);
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); };
    const float* cur = mel;
    int cur_c = M;
    for (int i = 0; i < L; ++i) {
        ConvLayer raw = m->post_convs[i];
        raw.scale = nullptr; raw.shift = nullptr;               // the batch statistics replace the folded moving statistics
        T2_TRY(launch_conv(m, raw, cur, cur_c, yraw, C, Mtot, T_out, 0, nullptr, 0));
        hipLaunchKernelGGL(bn_stats_kernel, dim3(ceil_div(C, 64), T2_BN_CHUNKS), dim3(256), 0, st, yraw, Mtot, C, bn_part);
        hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, bn_part, Mtot, C, d.bn_eps, mean, invstd, var);
        hipLaunchKernelGGL(bn_apply_kernel, grid1((long long)Mtot * C), dim3(256), 0, st, yraw, mean, invstd, m->post_bn_gamma[i], m->post_bn_beta[i],
                           drop_rate, seed, T2_STREAM_POST_CONV + i, bufs[i & 1], (long long)Mtot, C, (i + 1 < L) ? 2 : 0);
        if (bn_stats_out) {
            HIP_TRY(hipMemcpyAsync(bn_stats_out + (size_t)i * 2 * C, mean, (size_t)C * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(bn_stats_out + (size_t)i * 2 * C + C, var, (size_t)C * 4, hipMemcpyDeviceToDevice, st));
        }
        cur = bufs[i & 1];
        cur_c = C;
    }
    T2_TRY(launch_conv(m, m->post_proj, cur, C, mel_post, M, Mtot, Mtot, 0, mel, M));
    HIP_TRY(hipGetLastError());
    return T2_OK;
}


// column-owner tiles of an encoder direction's transposed kernel (persist_bwd.h encoder_bwd_kernel): workgroup g's columns are
// d h_prev of its units 4g..4g+3 and the input columns 8g..8g+7; source: the TF kernel [Cx + He][4 He] on the device
static int pze_pack(t2_trainer* tr, EncTape* et, int dir, const float* w_dev) {
    if (!et->pze_w[dir]) return T2_OK;
    pz::TileMap mp{};
    mp.nseg = 2; mp.NC = pz::NCE;
    mp.c0[0] = 0; mp.base[0] = 512; mp.per_g[0] = 4;
    mp.c0[1] = 4; mp.base[1] = 0; mp.per_g[1] = 8;
    const long long n4 = (long long)pz::ENC_G * 64 * 4 * pz::NCE;
    hipLaunchKernelGGL(pz::pack_bwd_tiles_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, tr->m->ctx->stream, w_dev, 1024, pz::ENC_G, mp, et->pze_w[dir]);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}
static int pze_setup(t2_trainer* tr, EncTape* et) {
    t2_model* m = tr->m;
    const t2_dims& d = m->d;
    const int B = tr->B, T = tr->T, MB = mb_for(B), E = d.enc_units, He = E / 2, L = d.enc_conv_layers;
    const int Cx = L > 0 ? E : d.embedding_dim;
    if (Cx != 512 || He != 256 || m->n_cu < 2 * pz::ENC_G) return T2_OK;
    hipStream_t st = m->ctx->stream;
    const char* dirs[2] = {"fw", "bw"};
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, (size_t)768 * 1024 * 4));
    for (int dir = 0; dir < 2; ++dir) {
        const HostWeight* w;
        int rc = get_w(m, std::string("encoder/bidirectional_rnn/") + dirs[dir] + "/lstm_cell/kernel", &w, {768, 1024});
        hipError_t e = rc == T2_OK ? hipMalloc((void**)&et->pze_w[dir], (size_t)pz::ENC_G * 64 * 64 * pz::NCE) : hipErrorUnknown;
        if (e == hipSuccess) { tr->allocs.push_back(et->pze_w[dir]); e = hipMemcpyAsync(tmp, w->data.data(), (size_t)768 * 1024 * 4, hipMemcpyHostToDevice, st); }
        if (e == hipSuccess) rc = pze_pack(tr, et, dir, tmp);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (rc != T2_OK || e != hipSuccess) { (void)hipFree(tmp); return rc != T2_OK ? rc : fail(T2_EHIP, "persistent encoder backward operands: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(hipFree(tmp));
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    et->pze_xs_bytes = (unsigned)carve(2 * 2 * (size_t)MB * 16 * 1024 * 4);
    et->pze_xs_bytes = (unsigned)off;
    const size_t o_bar = carve((size_t)(2 * pz::NSHARD * 32 + 64) * 4);
    et->pze_slab_bytes = off;
    const size_t o_dz2 = carve((size_t)T * B * 1024 * 4), o_dx2 = carve((size_t)T * B * 768 * 4);
    HIP_TRY(hipMalloc((void**)&et->pze_slab, off));
    tr->allocs.push_back(et->pze_slab);
    HIP_TRY(hipMemset(et->pze_slab, 0, off));
    et->pze_bar = (unsigned*)(et->pze_slab + o_bar);
    et->dzp2 = (float*)(et->pze_slab + o_dz2); et->dX2 = (float*)(et->pze_slab + o_dx2);
    return T2_OK;
}

static int enc_setup_backward(t2_trainer* tr, EncTape* et) {
    t2_model* m = tr->m;
    const t2_dims& d = m->d;
    const int B = tr->B, T = tr->T, MB = mb_for(B), R = MB * 16, E = d.enc_units, He = E / 2, emb = d.embedding_dim, L = d.enc_conv_layers;
    const int Cx = L > 0 ? E : emb;
    const char* dirs[2] = {"fw", "bw"};
    for (int dir = 0; dir < 2; ++dir)
        T2_TRY(trainer_pack_T(tr, &et->lstmT[dir], std::string("encoder/bidirectional_rnn/") + dirs[dir] + "/lstm_cell/kernel", Cx + He, 4 * He, 0, 4 * He));
    et->convT.resize(L);
    int cin = emb;
    for (int i = 0; i < L; ++i) {
        const HostWeight* w;
        const int k = d.enc_kernel;
        T2_TRY(get_w(m, "encoder/conv1d_" + std::to_string(i) + "/kernel", &w, {k, cin, E}));
        std::vector<float> t((size_t)k * E * cin);
        for (int j = 0; j < k; ++j)
            for (int ci = 0; ci < cin; ++ci)
                for (int co = 0; co < E; ++co) t[((size_t)(k - 1 - j) * E + co) * cin + ci] = w->data[((size_t)j * cin + ci) * E + co];
        ConvLayer& Lc = et->convT[i];
        Lc.k = k; Lc.cin = E; Lc.cout = cin;
        HIP_TRY(hipMalloc((void**)&Lc.w, t.size() * 4));
        tr->allocs.push_back(Lc.w);
        HIP_TRY(hipMemcpy(Lc.w, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        cin = E;
    }
    const size_t Mtot = (size_t)B * T;
    const int cmax = std::max(E, emb);
    size_t off = 0;
    auto carve = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
    const size_t o_xs = carve((size_t)T * B * Cx), o_go = carve((size_t)T * B * He), o_dzp = carve((size_t)T * B * 4 * He),
                 o_dzf = carve((size_t)2 * R * pad16(4 * He));          // ping-pong: a step's GEMM reads one while its epilogue fills the other
    const size_t o_dX = carve((size_t)T * B * (Cx + He)), o_gc = carve((size_t)B * He), o_ghp = carve((size_t)B * He);
    const size_t o_a = carve(Mtot * cmax), o_b = carve(Mtot * cmax), o_dg = carve(E), o_db = carve(E);
    void* p0 = nullptr;
    HIP_TRY(hipMalloc(&p0, off));
    tr->allocs.push_back(p0);
    HIP_TRY(hipMemset(p0, 0, off));
    char* p = (char*)p0;
    et->xs = (float*)(p + o_xs); et->gout = (float*)(p + o_go); et->dzp = (float*)(p + o_dzp); et->dzf = (float*)(p + o_dzf);
    et->dX = (float*)(p + o_dX); et->g_c = (float*)(p + o_gc); et->g_hp = (float*)(p + o_ghp);
    et->dxa = (float*)(p + o_a); et->dxb = (float*)(p + o_b); et->dgamma = (float*)(p + o_dg); et->dbeta = (float*)(p + o_db);
    T2_TRY(pze_setup(tr, et));
    et->have_bwd = true;
    return T2_OK;
}

// Backward of t2_train_encoder_forward: dmemory dev [B][T_in][enc_units] (d loss / d memory, e.g. from
// t2_train_decoder_backward) -> gradients of embedding, encoder/conv1d_i/{kernel, batch_normalization/gamma, beta} and the
// BiLSTM kernels / biases, written into flat_grads at the bound offsets.
extern "C" int t2_train_encoder_backward(t2_trainer* tr, const float* dmemory, float* flat_grads) {
    if (!tr || !dmemory || !flat_grads) return fail(T2_EINVAL, "null argument");
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    EncTape* et = enc_tape_of(tr);
    if (!et->have_forward) return fail(T2_ESTATE, "t2_train_encoder_forward has not run");
    if (!et->have_bwd) T2_TRY(enc_setup_backward(tr, et));
    T2_TRY(trainer_sync_params(tr));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = tr->B, T = tr->T, MB = mb_for(B), R = MB * 16, E = d.enc_units, He = E / 2, emb = d.embedding_dim, L = d.enc_conv_layers;
    const int Cx = L > 0 ? E : emb, Mtot = B * T;
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); };
    float* g = nullptr;
    HIP_TRY(hipMemsetAsync(et->dxa, 0, (size_t)Mtot * Cx * 4, st));
    const char* dirs[2] = {"fw", "bw"};
    // both directions' reverse-time recurrences as ONE persistent launch at the paper-size class (persist_bwd.h), else one launch per step
    bool pze = m->opt_bwd_persist && m->persist_holdoff == 0 && et->pze_slab != nullptr;
    if (pze) {
        HIP_TRY(hipMemsetAsync(et->pze_slab, 0, et->pze_slab_bytes, st));
        pz::EBArgs a{};
        for (int dir = 0; dir < 2; ++dir) { a.wT[dir] = et->pze_w[dir]; a.z[dir] = et->z[dir]; a.cprev[dir] = et->cprev[dir]; a.cnew[dir] = et->cnew[dir]; }
        a.dmem = dmemory; a.lens = et->lens;
        a.dzp[0] = et->dzp; a.dzp[1] = et->dzp2; a.dX[0] = et->dX; a.dX[1] = et->dX2;
        a.xs = (float*)et->pze_slab; a.xs_bytes = et->pze_xs_bytes;
        a.bar = et->pze_bar; a.err = (int*)(et->pze_bar + 2 * pz::NSHARD * 32);
        a.B = B; a.T = T; a.stream = T2_STREAM_ZONE_ENC; a.seed = et->seed;
        a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias;
        const dim3 grid(2 * pz::ENC_G), blk(pz::NT);
        {
            XwgGate gate;
            T2_TRY(gate.begin(m->ctx));
            if (MB == 1) hipLaunchKernelGGL((pz::encoder_bwd_kernel<1>), grid, blk, 0, st, a);
            else if (MB == 2) hipLaunchKernelGGL((pz::encoder_bwd_kernel<2>), grid, blk, 0, st, a);
            else hipLaunchKernelGGL((pz::encoder_bwd_kernel<4>), grid, blk, 0, st, a);
            HIP_TRY(hipGetLastError());
        }
        T2_TRY(inject_timeout(m, a.err));
        ++m->n_bwd_persist;
        int h_err = 0;
        HIP_TRY(hipMemcpyAsync(&h_err, a.err, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_err) {                       // a hand-off wait ran out of time: the launch-per-step recurrence below runs instead
            note_persist_timeout(m);
            HIP_TRY(hipMemsetAsync(et->dxa, 0, (size_t)Mtot * Cx * 4, st));
            pze = false;
        }
    }
    for (int dir = 0; dir < 2; ++dir) {
        hipLaunchKernelGGL(seq_gather_kernel, grid1((long long)T * B * Cx), dim3(256), 0, st, et->x[L], et->lens, et->xs, B, T, Cx, dir);
        float* const dzp = (pze && dir == 1) ? et->dzp2 : et->dzp;
        float* const dXd = (pze && dir == 1) ? et->dX2 : et->dX;
        const size_t fsz = (size_t)R * pad16(4 * He);
        if (!pze) {
        HIP_TRY(hipMemsetAsync(et->dzf, 0, 2 * fsz * 4, st));
        hipLaunchKernelGGL(enc_gout_kernel, grid1((long long)T * B * He), dim3(256), 0, st, dmemory, et->lens, et->gout, B, T, E, He, dir * He, dir);
        auto lstm_args = [&](int s) {
            LstmBwdArgs a{};
            a.z = et->z[dir] + (size_t)s * B * 4 * He; a.cprev = et->cprev[dir] + (size_t)s * B * He; a.cnew = et->cnew[dir] + (size_t)s * B * He;
            a.g_out0 = et->gout + (size_t)s * B * He; a.ld_out0 = He; a.g_out1 = nullptr;
            a.g_hs0 = et->g_hp; a.ld_hs0 = He; a.g_hs1 = nullptr; a.ld_hs1 = 0;      // d h_prev of step s + 1 arrives as the epilogue's own value
            a.g_c = et->g_c; a.g_hp = et->g_hp; a.dz_plain = et->dzp + (size_t)s * B * 4 * He; a.dz_frag = et->dzf + (size_t)(s & 1) * fsz;
            a.B = B; a.H = He; a.MB = MB; a.t = s; a.stream = T2_STREAM_ZONE_ENC + 2 * dir; a.seed = et->seed;
            a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias; a.first = (s == T - 1) ? 1 : 0; a.lens = et->lens;
            return a;
        };
        hipLaunchKernelGGL(lstm_bwd_kernel, grid1((long long)B * He), dim3(256), 0, st, lstm_args(T - 1));
        for (int s = T - 1; s >= 0; --s) {            // one launch per step: d [x | h_prev] of step s, cell backward of step s - 1 in the epilogue
            BwdEpi e{};
            e.N = Cx + He; e.out_plain = et->dX + (size_t)s * B * (Cx + He); e.plain_ld = Cx + He;
            if (s > 0) { e.mode = 3; e.col0 = Cx; e.lb = lstm_args(s - 1); }
            bwd_dense_epi(tr, et->lstmT[dir], et->dzf + (size_t)(s & 1) * fsz, e);
        }
        }
        hipLaunchKernelGGL(enc_dx_scatter_kernel, grid1((long long)T * B * Cx), dim3(256), 0, st, dXd, Cx + He, et->lens, et->dxa, B, T, Cx, dir);
        const std::string pn = std::string("encoder/bidirectional_rnn/") + dirs[dir] + "/lstm_cell";
        const int n4 = 4 * He;
        T2_TRY(grad_dst(tr, flat_grads, pn + "/kernel", &g));
        T2_TRY(bwd_wgrad(tr, et->xs, Cx, Cx, dzp, n4, n4, (long long)T * B, g, n4));
        if (T > 1) T2_TRY(bwd_wgrad(tr, et->hstate[dir], He, He, dzp + (size_t)B * n4, n4, n4, (long long)(T - 1) * B, g + (size_t)Cx * n4, n4));
        else HIP_TRY(hipMemsetAsync(g + (size_t)Cx * n4, 0, (size_t)He * n4 * 4, st));
        T2_TRY(grad_dst(tr, flat_grads, pn + "/bias", &g));
        T2_TRY(col_sum(tr, dzp, (long long)T * B, n4, g));
    }
    // conv stack: dropout / relu / batch-statistics batch norm / convolution, last layer first
    float* d_out = et->dxa;
    float* other = et->dxb;
    const float inv_keep = 1.0f / (1.0f - et->drop_rate);
    for (int i = L - 1; i >= 0; --i) {
        const int cin = i == 0 ? emb : E, k = d.enc_kernel, pad_l = (k - 1) / 2;
        const std::string pn = "encoder/conv1d_" + std::to_string(i);
        T2_TRY(bn_part_reserve(tr, E));
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ceil_div(E, 64), T2_BN_CHUNKS), dim3(256), 0, st, d_out, et->x[i + 1], et->yraw[i], et->mean[i], et->invstd[i],
                           inv_keep, Mtot, E, tr->bn_part);
        hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, st, tr->bn_part, E, et->dgamma, et->dbeta);
        T2_TRY(grad_dst(tr, flat_grads, pn + "/batch_normalization/gamma", &g));
        HIP_TRY(hipMemcpyAsync(g, et->dgamma, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        T2_TRY(grad_dst(tr, flat_grads, pn + "/batch_normalization/beta", &g));
        HIP_TRY(hipMemcpyAsync(g, et->dbeta, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(bn_bwd_apply_kernel, grid1((long long)Mtot * E), dim3(256), 0, st, d_out, et->x[i + 1], et->yraw[i], et->mean[i], et->invstd[i],
                           m->enc_bn_gamma[i], et->dgamma, et->dbeta, inv_keep, (long long)Mtot, E, other);
        // weight gradient: im2col(x_i)^T . dy
        T2_TRY(grad_dst(tr, flat_grads, pn + "/kernel", &g));
        {
            ConvArgs a{};
            a.x = et->x[i]; a.ldx = cin; a.w = other; a.y = g; a.ldy = E;
            a.Mtot = k * cin; a.T = T; a.Cin = cin; a.N = E; a.ktaps = k; a.pad_l = pad_l; a.act = 0; a.trans_a = 1; a.red = Mtot;
            a.vec_a = (cin % 4 == 0) && (((uintptr_t)et->x[i]) % 16 == 0);
            a.vec_b = (E % 4 == 0) && (((uintptr_t)other) % 16 == 0);
            T2_TRY(launch_wgrad(tr, a));
        }
        // data gradient: correlation of dy with the flipped, transposed kernel
        {
            const ConvLayer& Lc = et->convT[i];
            if (k % 2 == 1) {                 // symmetric padding: the MFMA convolution of the forward pass serves (launch_conv)
                T2_TRY(launch_conv(m, Lc, other, E, d_out, cin, Mtot, T, 0, nullptr, 0));
                continue;
            }
            ConvArgs a{};
            a.x = other; a.ldx = E; a.w = Lc.w; a.y = d_out; a.ldy = cin;
            a.Mtot = Mtot; a.T = T; a.Cin = E; a.N = cin; a.ktaps = k; a.pad_l = k - 1 - pad_l; a.act = 0;
            a.vec_a = (E % 4 == 0) && (((uintptr_t)other) % 16 == 0);
            a.vec_b = (cin % 4 == 0) && (((uintptr_t)Lc.w) % 16 == 0);
            hipLaunchKernelGGL(conv_gemm_kernel, dim3(ceil_div(cin, CG_BN), ceil_div(Mtot, CG_BM)), dim3(256), 0, st, a);
        }
        // d_out now holds d x_i ([M][cin]); `other` is free again
    }
    T2_TRY(grad_dst(tr, flat_grads, "embedding", &g));
    if (emb > 2048) return fail(T2_EINVAL, "embedding_dim > 2048 is not supported by the embedding gradient kernel");
    hipLaunchKernelGGL(embedding_grad_kernel, dim3(d.num_symbols), dim3(256), 0, st, et->ids, d_out, Mtot, emb, g);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// ==========================================================================================================================
// Parameters on the device: after the optimizer has updated the flat fp32 parameter buffer, every kernel-native copy of the
// weights (packed skinny operands, folded batch-norm affines, the transposed operands of the backward pass) is rebuilt from it
// on the GPU -- the device-side mirror of t2_model_finalize / trainer_setup_backward.
// ==========================================================================================================================
struct PackDesc {
    const float* src; long long sk, sc;       // element (k, col) of the operand = src[k * sk + col * sc]
    int nseg, seg_first[T2_MAX_SEG], seg_rows[T2_MAX_SEG], seg_kb[T2_MAX_SEG];
    int kbtot, n_wg, lstm_units, ncols;
};

__global__ __launch_bounds__(256) void pack_skinny_dev_kernel(float* __restrict__ dst, PackDesc d) {
    const long long n = (long long)d.n_wg * d.kbtot * 256;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        const long long gk = idx >> 8;
        const int kb = (int)(gk % d.kbtot), g = (int)(gk / d.kbtot);
        const int kg = lane >> 4, c = lane & 15;
        int col;
        if (d.lstm_units > 0) {
            const int gate = c >> 2, u = 4 * g + (c & 3);
            col = u < d.lstm_units ? gate * d.lstm_units + u : -1;
        } else {
            col = 16 * g + c;
            if (col >= d.ncols) col = -1;
        }
        int s = 0, kb0 = 0;
        while (s + 1 < d.nseg && kb >= kb0 + d.seg_kb[s]) { kb0 += d.seg_kb[s]; ++s; }
        const int k = (kb - kb0) * 16 + kg * 4 + j;
        float v = 0.f;
        if (col >= 0 && k < d.seg_rows[s]) v = d.src[(long long)(d.seg_first[s] + k) * d.sk + (long long)col * d.sc];
        dst[idx] = v;
    }
}

__global__ __launch_bounds__(256) void lstm_bias_pack_kernel(float* __restrict__ dst, const float* __restrict__ bias, int n_wg, int H) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_wg * 16; i += gridDim.x * blockDim.x) {
        const int g = i >> 4, c = i & 15, gate = c >> 2, u = 4 * g + (c & 3);
        dst[i] = u < H ? bias[gate * H + u] : 0.f;
    }
}

// dst[r][c] (row stride ldd) = src[c][r] (row stride lds): rows x cols of dst
__global__ __launch_bounds__(256) void transpose_kernel(float* __restrict__ dst, int ldd, const float* __restrict__ src, int lds, int rows, int cols) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)rows * cols; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        dst[(size_t)r * ldd + c] = src[(size_t)c * lds + r];
    }
}

// the transposed copy of a convolution kernel (ConvLayer::wt, [cout][k*cin]) after w was rewritten on the device
static void conv_refresh_wt(hipStream_t st, const ConvLayer& L) {
    if (L.wt == nullptr) return;
    const int K = L.k * L.cin;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(4096, ((long long)K * L.cout + 255) / 256))), dim3(256), 0, st,
                       L.wt, K, L.w, L.cout, L.cout, K);
}

// inference batch norm folded to scale / shift (make_conv)
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                                                      float* scale, float* shift, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc; shift[i] = beta[i] - mean[i] * sc;
}

// conv kernel [k][cin][cout] -> [k][cout][cin] with the taps reversed (data gradient of a SAME convolution)
__global__ __launch_bounds__(256) void conv_flip_kernel(float* __restrict__ dst, const float* __restrict__ src, int k, int cin, int cout) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)k * cin * cout; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout), ci = (int)((i / cout) % cin), j = (int)(i / ((long long)cout * cin));
        dst[((size_t)(k - 1 - j) * cout + co) * cin + ci] = src[i];
    }
}

// tf.layers.batch_normalization moving statistics (the UPDATE_OPS of models.py:75): moving = momentum * moving + (1 - momentum) *
// batch, the variance with Bessel's correction as TF's fused kernel does
__global__ __launch_bounds__(256) void bn_moving_update_kernel(float* mov_mean, float* mov_var, const float* mean, const float* var, float momentum,
                                                               float bessel, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    mov_mean[i] = momentum * mov_mean[i] + (1.f - momentum) * mean[i];
    mov_var[i] = momentum * mov_var[i] + (1.f - momentum) * var[i] * bessel;
}

static int flat_src(t2_trainer* tr, const float* flat, const std::string& name, const float** out) {
    auto it = tr->offsets.find(name);
    if (it == tr->offsets.end()) return fail(T2_ESTATE, "no offset bound for '%s' (t2_trainer_bind_gradient)", name.c_str());
    *out = flat + it->second;
    return T2_OK;
}

// The composed projection operand of the free-running decode (t2_model_finalize builds it on the host), rebuilt on the
// device after a parameter refresh: out [D + 1][N2], row D = bias.  Columns [0, N) = [W_mel | w_stop], [N, N0) zero,
// [N0, N0 + P0) = W_mel[:, fed frames] . W_prenet0 (+ b_mel[fed] . W_prenet0 + b_prenet0 in the bias row), summed in double
// and rounded once, like the host version.
__global__ __launch_bounds__(256) void compose_proj_kernel(const float* __restrict__ wm, const float* __restrict__ ws, const float* __restrict__ bm,
                                                           const float* __restrict__ bs, const float* __restrict__ wp0, const float* __restrict__ bp0,
                                                           float* __restrict__ out, int D, int MR, int c0, int nfm, int P0, int N0, int N2) {
    const long long n = (long long)(D + 1) * N2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / N2), c = (int)(i - (long long)k * N2);
        const bool bias = k == D;
        float v = 0.f;
        if (c < MR) v = bias ? bm[c] : wm[(size_t)k * MR + c];
        else if (c == MR) v = bias ? bs[0] : ws[k];
        else if (c >= N0 && c < N0 + P0) {
            const int q = c - N0;
            double acc = bias ? (double)bp0[q] : 0.0;
            const float* row = bias ? bm : wm + (size_t)k * MR;
            for (int j = 0; j < nfm; ++j) acc += (double)row[c0 + j] * (double)wp0[(size_t)j * P0 + q];
            v = (float)acc;
        }
        out[i] = v;
    }
}

static void repack(hipStream_t st, const Packed& p, const float* src, long long sk, long long sc, int ncols) {
    PackDesc d{};
    d.src = src; d.sk = sk; d.sc = sc; d.nseg = p.nseg; d.kbtot = p.kbtot; d.n_wg = p.n_wg; d.lstm_units = p.lstm_units; d.ncols = ncols;
    for (int s = 0; s < p.nseg; ++s) { d.seg_first[s] = p.seg_first[s]; d.seg_rows[s] = p.seg_rows[s]; d.seg_kb[s] = p.seg_kb[s]; }
    const long long n = (long long)p.n_wg * p.kbtot * 256;
    hipLaunchKernelGGL(pack_skinny_dev_kernel, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, st, p.w, d);
}

// Rebuild every device copy of the weights from the flat parameter buffer (layout = the offsets bound with
// t2_trainer_bind_gradient; TF layouts).  Called after the optimizer step; inference entry points of the same model see the
// new weights too.
// model_part: the copies every entry point of the model reads; trainer_part: the transposed / flipped operands of THIS trainer's
// backward pass.  A model can have several trainers (one per batch shape); each refreshes its own operands when it lags behind
// the model's parameter generation (trainer_sync_params).
static int load_params_impl(t2_trainer* tr, const float* flat, bool model_part, bool trainer_part) {
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    if (!tr->bslab) T2_TRY(trainer_setup_backward(tr));
    EncTape* et = enc_tape_of(tr);
    if (!et->slab) T2_TRY(enc_tape_alloc(tr, et));
    if (!et->have_bwd) T2_TRY(enc_setup_backward(tr, et));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int E = d.enc_units, He = E / 2, U = d.attention_dim, A = d.attention_lstm_units, D = d.decoder_lstm_units, emb = d.embedding_dim;
    const int F = d.attention_filters, K = d.attention_kernel, M = d.num_mels, r = d.outputs_per_step, np = d.num_prenet_layers, L = d.enc_conv_layers;
    const float *w, *b;
    auto copy = [&](float* dst, const float* src, size_t n) -> int { HIP_TRY(hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, st)); return T2_OK; };
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); };
    T2_TRY(flat_src(tr, flat, "embedding", &w));
    if (model_part) T2_TRY(copy(m->embedding, w, (size_t)d.num_symbols * emb));
    int cin = emb;
    for (int i = 0; i < L; ++i) {
        const std::string pn = "encoder/conv1d_" + std::to_string(i);
        const float *g, *be, *mu, *var;
        T2_TRY(flat_src(tr, flat, pn + "/kernel", &w));
        if (model_part) { T2_TRY(copy(m->enc_convs[i].w, w, (size_t)d.enc_kernel * cin * E)); conv_refresh_wt(st, m->enc_convs[i]); }
        if (trainer_part) hipLaunchKernelGGL(conv_flip_kernel, grid1((long long)d.enc_kernel * cin * E), dim3(256), 0, st, et->convT[i].w, w, d.enc_kernel, cin, E);
        T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/gamma", &g));
        T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/beta", &be));
        T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/moving_mean", &mu));
        T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/moving_variance", &var));
        if (model_part) T2_TRY(copy(m->enc_bn_gamma[i], g, E));
        if (model_part) T2_TRY(copy(m->enc_bn_beta[i], be, E));
        if (model_part) hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, st, g, be, mu, var, d.bn_eps, m->enc_convs[i].scale, m->enc_convs[i].shift, E);
        cin = E;
    }
    const char* dirs[2] = {"fw", "bw"};
    for (int dir = 0; dir < 2; ++dir) {
        const std::string pn = std::string("encoder/bidirectional_rnn/") + dirs[dir] + "/lstm_cell";
        T2_TRY(flat_src(tr, flat, pn + "/kernel", &w));
        T2_TRY(flat_src(tr, flat, pn + "/bias", &b));
        if (model_part) {
            repack(st, m->enc_lstm[dir], w, 4 * He, 1, 0);
            hipLaunchKernelGGL(lstm_bias_pack_kernel, grid1(m->enc_lstm[dir].n_wg * 16), dim3(256), 0, st, m->enc_lstm[dir].bias, b, m->enc_lstm[dir].n_wg, He);
        }
        if (trainer_part) repack(st, et->lstmT[dir], w, 1, 4 * He, cin + He);
        if (trainer_part) T2_TRY(pze_pack(tr, et, dir, w));
    }
    T2_TRY(flat_src(tr, flat, "attention/memory_layer/kernel", &w));
    if (model_part) T2_TRY(copy(m->memory_layer.w, w, (size_t)E * U));
    if (trainer_part) hipLaunchKernelGGL(transpose_kernel, grid1((long long)U * E), dim3(256), 0, st, tr->wmemT, E, w, U, U, E);
    T2_TRY(flat_src(tr, flat, "attention/query_layer/kernel", &w));
    if (model_part) { T2_TRY(copy(m->wq, w, (size_t)A * U)); repack(st, m->wq_pk, w, U, 1, U); }
    if (trainer_part) repack(st, tr->wqT, w, 1, U, A);
    if (trainer_part) T2_TRY(pzb_pack(tr, 3, w));
    if (model_part) {
        T2_TRY(flat_src(tr, flat, "attention/location_features_convolution/kernel", &w)); T2_TRY(copy(m->conv_w, w, (size_t)K * F));
        T2_TRY(flat_src(tr, flat, "attention/location_features_convolution/bias", &w)); T2_TRY(copy(m->conv_b, w, F));
        T2_TRY(flat_src(tr, flat, "attention/location_features_layer/kernel", &w)); T2_TRY(copy(m->wloc, w, (size_t)F * U));
        T2_TRY(flat_src(tr, flat, "attention/attention_variable", &w)); T2_TRY(copy(m->v_a, w, U));
        T2_TRY(flat_src(tr, flat, "attention/attention_bias", &w)); T2_TRY(copy(m->b_a, w, U));
    }
    int pin = M * d.n_feed_frame;
    for (int i = 0; i < np; ++i) {
        const std::string pn = "decoder/prenet_" + std::to_string(i) + "/dense";
        T2_TRY(flat_src(tr, flat, pn + "/kernel", &w));
        T2_TRY(flat_src(tr, flat, pn + "/bias", &b));
        if (model_part) {
            repack(st, m->prenet[i], w, d.prenet_units[i], 1, d.prenet_units[i]);
            T2_TRY(copy(m->prenet[i].bias, b, d.prenet_units[i]));
            T2_TRY(copy(m->prenet_plain[i].w, w, (size_t)m->prenet_plain[i].cin * d.prenet_units[i]));
            T2_TRY(copy(m->prenet_plain[i].shift, b, d.prenet_units[i]));
        }
        if (trainer_part && i > 0) repack(st, tr->prenetT[i], w, 1, d.prenet_units[i], d.prenet_units[i - 1]);
        if (trainer_part && i > 0 && tr->prenetTp[i])
            hipLaunchKernelGGL(transpose_kernel, grid1((long long)d.prenet_units[i] * d.prenet_units[i - 1]), dim3(256), 0, st, tr->prenetTp[i],
                               d.prenet_units[i - 1], w, d.prenet_units[i], d.prenet_units[i], d.prenet_units[i - 1]);
        pin = d.prenet_units[i];
    }
    const char* lname[3] = {"decoder/attention_lstm", "decoder/lstm_1", "decoder/lstm_2"};
    Packed* lp[3] = {&m->att_lstm, &m->lstm1, &m->lstm2};
    const int Hh[3] = {A, D, D};
    const int KXs[3] = {pin + E + A, A + E + D, D + D};
    for (int l = 0; l < 3; ++l) {
        T2_TRY(flat_src(tr, flat, std::string(lname[l]) + "/kernel", &w));
        T2_TRY(flat_src(tr, flat, std::string(lname[l]) + "/bias", &b));
        if (model_part) {
            repack(st, *lp[l], w, 4 * Hh[l], 1, 0);
            hipLaunchKernelGGL(lstm_bias_pack_kernel, grid1(lp[l]->n_wg * 16), dim3(256), 0, st, lp[l]->bias, b, lp[l]->n_wg, Hh[l]);
        }
        if (trainer_part) repack(st, tr->lstmT[l], w, 1, 4 * Hh[l], KXs[l]);
        if (trainer_part) T2_TRY(pzb_pack(tr, 2 - l, w));
    }
    {   // projections: concatenated [D][M*r + 1] operand and its transpose
        const int N = M * r + 1;
        const float *wm, *bm, *ws, *bs;
        T2_TRY(flat_src(tr, flat, "decoder/out_projection/kernel", &wm));
        T2_TRY(flat_src(tr, flat, "decoder/out_projection/bias", &bm));
        T2_TRY(flat_src(tr, flat, "decoder/stop_token_projection/kernel", &ws));
        T2_TRY(flat_src(tr, flat, "decoder/stop_token_projection/bias", &bs));
        float* wc = tr->tmp;                                      // [D][N]
        HIP_TRY(hipMemcpy2DAsync(wc, (size_t)N * 4, wm, (size_t)M * r * 4, (size_t)M * r * 4, D, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpy2DAsync(wc + M * r, (size_t)N * 4, ws, 4, 4, D, hipMemcpyDeviceToDevice, st));
        if (model_part) {
            repack(st, m->proj, wc, N, 1, N);
            HIP_TRY(hipMemcpy2DAsync(m->proj_plain.w, (size_t)m->proj_plain.cout * 4, wc, (size_t)N * 4, (size_t)N * 4, D, hipMemcpyDeviceToDevice, st));
            T2_TRY(copy(m->proj_plain.shift, bm, (size_t)M * r));
            T2_TRY(copy(m->proj_plain.shift + M * r, bs, 1));
            if (m->proj_fused_ok) {        // the composed projection + PreNet-0 operand (free-running decode): recomposed here
                const int nfm = M * d.n_feed_frame, c0 = M * (r - d.n_feed_frame), P0 = d.prenet_units[0], N0 = pad16(N), N2 = N0 + P0;
                if (!tr->compose_tmp) {
                    HIP_TRY(hipMalloc((void**)&tr->compose_tmp, (size_t)(D + 1) * N2 * 4));
                    tr->allocs.push_back(tr->compose_tmp);
                }
                const float *wp0, *bp0;
                T2_TRY(flat_src(tr, flat, "decoder/prenet_0/dense/kernel", &wp0));
                T2_TRY(flat_src(tr, flat, "decoder/prenet_0/dense/bias", &bp0));
                hipLaunchKernelGGL(compose_proj_kernel, grid1((long long)(D + 1) * N2), dim3(256), 0, st, wm, ws, bm, bs, wp0, bp0, tr->compose_tmp,
                                   D, M * r, c0, nfm, P0, N0, N2);
                repack(st, m->proj_fused, tr->compose_tmp, N2, 1, N2);
                T2_TRY(copy(m->proj_fused.bias, tr->compose_tmp + (size_t)D * N2, (size_t)N2));
            }
            T2_TRY(copy(m->proj.bias, bm, (size_t)M * r));
            T2_TRY(copy(m->proj.bias + M * r, bs, 1));
        }
        if (trainer_part) repack(st, tr->projT, wc, 1, N, D);
        if (trainer_part && tr->projTp) hipLaunchKernelGGL(transpose_kernel, grid1((long long)N * D), dim3(256), 0, st, tr->projTp, D, wc, N, N, D);
    }
    if (model_part) {
        cin = M;
        for (int i = 1; i <= d.postnet_layers; ++i) {
            const std::string pn = "postnet/conv1d_" + std::to_string(i);
            const float *g, *be, *mu, *var;
            ConvLayer& Lc = m->post_convs[i - 1];
            T2_TRY(flat_src(tr, flat, pn + "/kernel", &w));
            T2_TRY(copy(Lc.w, w, (size_t)d.postnet_kernel * cin * d.postnet_channels));
            conv_refresh_wt(st, Lc);
            T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/gamma", &g));
            T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/beta", &be));
            T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/moving_mean", &mu));
            T2_TRY(flat_src(tr, flat, pn + "/batch_normalization/moving_variance", &var));
            T2_TRY(copy(m->post_bn_gamma[i - 1], g, (size_t)d.postnet_channels));
            T2_TRY(copy(m->post_bn_beta[i - 1], be, (size_t)d.postnet_channels));
            hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(d.postnet_channels, 256)), dim3(256), 0, st, g, be, mu, var, d.bn_eps, Lc.scale, Lc.shift, d.postnet_channels);
            cin = d.postnet_channels;
        }
        T2_TRY(flat_src(tr, flat, "postnet/projection/kernel", &w));
        T2_TRY(copy(m->post_proj.w, w, (size_t)d.postnet_channels * M));
        T2_TRY(flat_src(tr, flat, "postnet/projection/bias", &b));
        T2_TRY(copy(m->post_proj.shift, b, M));
    }
    // cached decode graphs hold no weight pointers that moved: buffers are updated in place
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// Data-parallel exactness: SUM_BY_NONZERO_WEIGHTS divides by the number of non-zero weights of the WHOLE batch.
// With counts = {spec, stop} totals over all ranks (dev [2], kept alive by the caller), the backward pass differentiates
// num_rank / count_total, so the SUM of the ranks' gradients is the gradient of the single-device loss of the whole batch.
// counts == NULL restores the single-device normalisation.
extern "C" int t2_trainer_set_loss_counts(t2_trainer* tr, const float* counts) {
    if (!tr) return fail(T2_EINVAL, "null argument");
    tr->glob_counts = counts;
    return T2_OK;
}

extern "C" int t2_trainer_load_params(t2_trainer* tr, const float* flat) {
    if (!tr || !flat) return fail(T2_EINVAL, "null argument");
    T2_TRY(load_params_impl(tr, flat, true, true));
    tr->m->param_gen += 1;
    tr->m->last_flat = flat;
    tr->gen = tr->m->param_gen;
    return T2_OK;
}

// A trainer whose own operands were built from an older parameter generation (another trainer of the same model loaded newer
// parameters, or this trainer was created after a refresh: trainer_setup_backward packs from the INITIAL host weights) rebuilds
// them from the flat buffer of the last t2_trainer_load_params, which the caller keeps alive and unchanged between the refresh
// and the next backward pass (training.Trainer.train_step does).
static int trainer_sync_params(t2_trainer* tr) {
    t2_model* m = tr->m;
    if (tr->gen == m->param_gen) return T2_OK;
    if (m->last_flat == nullptr) return fail(T2_ESTATE, "internal: parameters were refreshed without a flat buffer");
    T2_TRY(load_params_impl(tr, m->last_flat, false, true));
    tr->gen = m->param_gen;
    return T2_OK;
}

// The UPDATE_OPS of the training step (models.py:75): moving statistics of the encoder's batch norms from the last
// t2_train_encoder_forward, in place in the flat parameter buffer.
extern "C" int t2_trainer_update_moving_stats(t2_trainer* tr, float* flat, float momentum) {
    if (!tr || !flat) return fail(T2_EINVAL, "null argument");
    t2_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->ctx->device));
    EncTape* et = enc_tape_of(tr);
    if (!et->have_forward) return fail(T2_ESTATE, "t2_train_encoder_forward has not run");
    const t2_dims& d = m->d;
    const int E = d.enc_units;
    const long long n = (long long)tr->B * tr->T;
    const float bessel = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
    for (int i = 0; i < d.enc_conv_layers; ++i) {
        const std::string pn = "encoder/conv1d_" + std::to_string(i) + "/batch_normalization/";
        const float *mu, *var;
        T2_TRY(flat_src(tr, flat, pn + "moving_mean", &mu));
        T2_TRY(flat_src(tr, flat, pn + "moving_variance", &var));
        hipLaunchKernelGGL(bn_moving_update_kernel, dim3(ceil_div(E, 256)), dim3(256), 0, m->ctx->stream, (float*)mu, (float*)var, et->mean[i], et->var[i],
                           momentum, bessel, E);
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}
