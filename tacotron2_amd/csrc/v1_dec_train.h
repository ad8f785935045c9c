// Training step of the Tacotron-1 decoder: DecoderV1.call(is_training=True) (tacotron/tacotron_v1.py:94-153) =
// OutputAndStopTokenWrapper(DecoderRNNV1(OutputProjectionWrapper(AttentionRNN(GRUCell, PreNets with dropout, BahdanauAttention)),
// 2 x ResidualWrapper(GRUCell))) (tacotron_v1.py:56-91, rnn_wrappers.py:81-186) under TrainingHelper (helpers.py:141-184), the
// two losses of models.py:58-61, and the hand-written backward pass (BPTT) of all of it.  Included by t2mi.hip after v1_train.h.
// With t2v1_encoder_train_* this is SingleSpeakerTacotronV1Model.model_fn(mode=TRAIN) (models.py:40-76).
//
// Design: plain row-major operands, weights read in place from the caller's flat parameter buffer.
// (1) Every operand a cell multiplies is kept CONCATENATED the way TF's kernels expect it ([ctx | h], [ctx | r*h], [h_att | ctx],
// [y | h]), written in place by the stages that produce the pieces, so a GRU cell is two products and the per-step slices double
// as the tape of the weight-gradient GEMMs.  (2) Everything that does not feed the recurrence is batched over all S*B rows:
// PreNets and their share of the attention GRU's pre-activations before the loop, the mel / stop projections after it; in the
// backward pass the projection gradient before the loop, every weight gradient, the PreNets, d values and the memory layer after
// it.  (3) A stage of the recurrence = one skinny product with the elementwise step that consumes it in its epilogue (sk_body):
// nine stages per decoder step forward, nine backward.  (4) At the reference widths both recurrences are ONE persistent launch
// each (v1d_fwd_persist_kernel / v1d_bwd_persist_kernel): two independent chains of stages side by side, five hops per step, a
// row's attention on four workgroups; other shapes run the same stages as launches inside a hipGraph (the parity twin and the
// fallback after a hop timeout).  Teacher-forced inference decodes reuse the forward launch (v1_teacher_via_trainer).
#pragma once

#define T2_STREAM_V1_DEC_PRENET 60

struct V1Dst { float* p; int ld; };

// x[m][:] of step t = m / B, row b = m % B: the last nf target frames of the previous group (TrainingHelper.next_inputs), zeros at t = 0
__global__ __launch_bounds__(256) void v1d_feed_kernel(const float* __restrict__ target, float* __restrict__ xin, int B, int S, int r, int nf, int M) {
    const long long n = (long long)S * B * nf * M;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (nf * M)), b = (int)((i / (nf * M)) % B), t = (int)(i / ((long long)nf * M * B));
        xin[i] = t == 0 ? 0.f : target[((size_t)b * S * r + (size_t)t * r - nf) * M + c];
    }
}
// PreNet dropout over the batched rows m = t * B + b: masks keyed on (seed, stream, step t, row b, unit)
__global__ __launch_bounds__(256) void v1d_dropout_kernel(float* __restrict__ x, float rate, uint32_t seed, int stream, int B, long long rows, int C) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * C; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        x[i] = dropout_train(x[i], rate, t2_u01(seed, stream, (uint32_t)(m / B), (uint32_t)(m % B), (uint32_t)(i - m * C)));
    }
}
// GRUCell gates: [r | u] = sigmoid(zg); the candidate's operand [x | r * h] from the gates' operand [x | h]
__global__ __launch_bounds__(256) void v1d_gate_kernel(const float* __restrict__ zg, const float* __restrict__ cat, int X, int H, float* __restrict__ ru,
                                                       float* __restrict__ catr, int B) {
    const int W = X + H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * (W + H); i += gridDim.x * blockDim.x) {
        const int b = i / (W + H), c = i - b * (W + H);
        if (c < X) catr[(size_t)b * W + c] = cat[(size_t)b * W + c];
        else if (c < W) {
            const int u = c - X;
            const float r = sigmoidf_(zg[(size_t)b * 2 * H + u]);
            ru[(size_t)b * 2 * H + u] = r;
            catr[(size_t)b * W + c] = r * cat[(size_t)b * W + c];
        } else {
            const int u = c - W;
            ru[(size_t)b * 2 * H + H + u] = sigmoidf_(zg[(size_t)b * 2 * H + H + u]);
        }
    }
}
// c = tanh(zc); h' = u h + (1 - u) c -> up to two destinations; ResidualWrapper: y = x + h' -> up to two destinations (X == H)
__global__ __launch_bounds__(256) void v1d_blend_kernel(const float* __restrict__ zc, const float* __restrict__ ru, const float* __restrict__ cat, int X, int H,
                                                        float* __restrict__ cnd, V1Dst h0, V1Dst h1, V1Dst y0, V1Dst y1, int B) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * H; i += gridDim.x * blockDim.x) {
        const int b = i / H, u = i - b * H;
        const float c = tanhf(zc[i]), uu = ru[(size_t)b * 2 * H + H + u], h = cat[(size_t)b * (X + H) + X + u];
        const float hn = uu * h + (1.0f - uu) * c;
        cnd[i] = c;
        if (h0.p) h0.p[(size_t)b * h0.ld + u] = hn;
        if (h1.p) h1.p[(size_t)b * h1.ld + u] = hn;
        if (y0.p || y1.p) {
            const float y = cat[(size_t)b * (X + H) + u] + hn;
            if (y0.p) y0.p[(size_t)b * y0.ld + u] = y;
            if (y1.p) y1.p[(size_t)b * y1.ld + u] = y;
        }
    }
}
// Accessors of data that crosses workgroups.  COH = false: one launch per stage, plain loads and stores.  COH = true: the stages of
// a whole recurrence run inside ONE launch (v1d_*_persist_kernel below) and hand their results to other workgroups between hops
// (persist.h's protocol): such words are stored write-through and loaded past the non-coherent caches (agent-scope relaxed atomics
// = global_store / global_load ... sc1 on gfx950).
template <bool COH> __device__ __forceinline__ float v1_ld(const float* p) {
    if constexpr (COH) return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return *p;
}
template <bool COH> __device__ __forceinline__ void v1_st(float* p, float v) {
    if constexpr (COH) __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// BahdanauAttention.__call__ + AttentionWrapper._compute_attention (App. A5, A6) of one batch row per workgroup of 16 waves:
// score_j = v . tanh(keys_j + q), softmax over the live positions (-inf mask), context = sum_j a_j values_j.  Wave w owns the
// positions j = w (mod 16) in both sweeps, a lane the features lane + 64 i; five positions are in flight per wave (the loads of a
// sweep are independent: issued together, clamped instead of branched), partial contexts of the 16 waves are added in wave order.
// tanh on the hardware exp / rcp (fast_tanhf, ~1e-6); the backward kernel uses the same function.
#define V1D_ATT_THREADS 1024
#define V1D_ATT_U 5
template <bool COH>
__device__ __forceinline__ void v1d_attn_fwd_body(const float* __restrict__ keys, const float* __restrict__ values, const float* q,
                                                  const float* __restrict__ v, const int* __restrict__ lens, int T, int A, int E,
                                                  float* __restrict__ align, V1Dst c0, V1Dst c1, int b, float* sm, float* s_red) {
    constexpr int NW = V1D_ATT_THREADS / 64, U = V1D_ATT_U;
    float *s_q = sm, *s_v = sm + A, *s_e = sm + 2 * A, *s_part = sm + 2 * A + T;        // s_part [NW][E]
    int tid = threadIdx.x;
    if constexpr (COH) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int len = lens[b];
    for (int u = tid; u < A; u += V1D_ATT_THREADS) { s_q[u] = v1_ld<COH>(q + (size_t)b * A + u); s_v[u] = v[u]; }
    __syncthreads();
    for (int j0 = wave; j0 < len; j0 += NW * U) {
        float p[U];
#pragma unroll
        for (int i = 0; i < U; ++i) p[i] = 0.f;
        for (int u0 = 0; u0 < A; u0 += 256) {
            float kk[U][4];
#pragma unroll
            for (int i = 0; i < U; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) kk[i][c] = keys[((size_t)b * T + min(j0 + i * NW, len - 1)) * A + min(u0 + lane + 64 * c, A - 1)];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int u = u0 + lane + 64 * c;
                const float qv = s_q[min(u, A - 1)], vv = u < A ? s_v[u] : 0.f;
#pragma unroll
                for (int i = 0; i < U; ++i) p[i] += vv * fast_tanhf(kk[i][c] + qv);
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p[i] += __shfl_xor(p[i], off);
            if (lane == 0 && j0 + i * NW < len) s_e[j0 + i * NW] = p[i];
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < len; j += V1D_ATT_THREADS) mx = fmaxf(mx, s_e[j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, s_red[w]);
    float sum = 0.f;
    for (int j = tid; j < T; j += V1D_ATT_THREADS) { const float e = j < len ? expf(s_e[j] - mx) : 0.f; s_e[j] = e; sum += e; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) s_red[NW + wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < NW; ++w) sum += s_red[NW + w];
    for (int j = tid; j < T; j += V1D_ATT_THREADS) { const float p = s_e[j] / sum; s_e[j] = p; align[(size_t)b * T + j] = p; }
    __syncthreads();
    for (int e0 = 0; e0 < E; e0 += 256) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = wave; j0 < len; j0 += NW * U) {
            float vv[U][4];
#pragma unroll
            for (int i = 0; i < U; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) vv[i][c] = values[((size_t)b * T + min(j0 + i * NW, len - 1)) * E + min(e0 + lane + 64 * c, E - 1)];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const float a = j0 + i * NW < len ? s_e[j0 + i * NW] : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = fmaf(a, vv[i][c], acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (e0 + lane + 64 * c < E) s_part[wave * E + e0 + lane + 64 * c] = acc[c];
    }
    __syncthreads();
    for (int e = tid; e < E; e += V1D_ATT_THREADS) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += s_part[w * E + e];
        if (c0.p) v1_st<COH>(c0.p + (size_t)b * c0.ld + e, acc);
        if (c1.p) v1_st<COH>(c1.p + (size_t)b * c1.ld + e, acc);
    }
}
__global__ __launch_bounds__(V1D_ATT_THREADS) void v1d_attn_fwd_kernel(const float* __restrict__ keys, const float* __restrict__ values, const float* __restrict__ q,
                                                                       const float* __restrict__ v, const int* __restrict__ lens, int T, int A, int E,
                                                                       float* __restrict__ align, V1Dst c0, V1Dst c1) {
    extern __shared__ float sm[];
    __shared__ float s_red[32];
    v1d_attn_fwd_body<false>(keys, values, q, v, lens, T, A, E, align, c0, c1, blockIdx.x, sm, s_red);
}
// its backward for one row: d a_j = d ctx . values_j; softmax backward; d q = sum_j g_j, d keys_j += g_j, d v += sum_j d e_j tanh_j
// with g_ju = d e_j v_u (1 - tanh_ju^2).  (d values is a tape-wide product afterwards; the Bahdanau state is not fed back.)
// Same ownership as the forward kernel; partial d q / d v of the 16 waves are added in wave order.
template <bool COH>
__device__ __forceinline__ void v1d_attn_bwd_body(const float* __restrict__ keys, const float* __restrict__ values, const float* __restrict__ q,
                                                  const float* __restrict__ v, const int* __restrict__ lens, const float* __restrict__ align,
                                                  const float* dctx, long long dctx_row_stride, int T, int A, int E,
                                                  float* dq, float* dkeys, float* dvpart, int b, float* sm, float* s_red) {
    constexpr int NW = V1D_ATT_THREADS / 64, U = V1D_ATT_U;
    float *s_dc = sm, *s_de = sm + E, *s_pq = sm + E + T, *s_pv = s_pq + NW * A;          // partial d q / d v [NW][A]
    int tid = threadIdx.x;
    if constexpr (COH) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int len = lens[b];
    for (int e = tid; e < E; e += V1D_ATT_THREADS) s_dc[e] = v1_ld<COH>(dctx + (size_t)b * dctx_row_stride + e);
    __syncthreads();
    for (int j0 = wave; j0 < len; j0 += NW * U) {
        float p[U];
#pragma unroll
        for (int i = 0; i < U; ++i) p[i] = 0.f;
        for (int e0 = 0; e0 < E; e0 += 256) {
            float vv[U][4];
#pragma unroll
            for (int i = 0; i < U; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) vv[i][c] = values[((size_t)b * T + min(j0 + i * NW, len - 1)) * E + min(e0 + lane + 64 * c, E - 1)];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = e0 + lane + 64 * c;
                const float dc = e < E ? s_dc[e] : 0.f;
#pragma unroll
                for (int i = 0; i < U; ++i) p[i] += dc * vv[i][c];
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p[i] += __shfl_xor(p[i], off);
            if (lane == 0 && j0 + i * NW < len) s_de[j0 + i * NW] = p[i];
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < len; j += V1D_ATT_THREADS) s += align[(size_t)b * T + j] * s_de[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) s_red[wave] = s;
    __syncthreads();
    s = 0.f;
    for (int w = 0; w < NW; ++w) s += s_red[w];
    __syncthreads();
    for (int j = tid; j < len; j += V1D_ATT_THREADS) s_de[j] = align[(size_t)b * T + j] * (s_de[j] - s);
    __syncthreads();
    for (int u0 = 0; u0 < A; u0 += 256) {
        float aq[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f}, qu[4], vu[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int u = min(u0 + lane + 64 * c, A - 1); qu[c] = q[(size_t)b * A + u]; vu[c] = v[u]; }
        for (int j0 = wave; j0 < len; j0 += NW * U) {
            float kk[U][4], dk[U][4];
#pragma unroll
            for (int i = 0; i < U; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const size_t ki = ((size_t)b * T + min(j0 + i * NW, len - 1)) * A + min(u0 + lane + 64 * c, A - 1);
                    kk[i][c] = keys[ki]; dk[i][c] = dkeys[ki];
                }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const bool live = j0 + i * NW < len;
                const float de = live ? s_de[j0 + i * NW] : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float th = fast_tanhf(kk[i][c] + qu[c]);
                    const float gg = de * vu[c] * (1.0f - th * th);
                    aq[c] += gg; av[c] += de * th;
                    if (live && u0 + lane + 64 * c < A) dkeys[((size_t)b * T + j0 + i * NW) * A + u0 + lane + 64 * c] = dk[i][c] + gg;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (u0 + lane + 64 * c < A) { s_pq[wave * A + u0 + lane + 64 * c] = aq[c]; s_pv[wave * A + u0 + lane + 64 * c] = av[c]; }
    }
    __syncthreads();
    for (int u = tid; u < A; u += V1D_ATT_THREADS) {
        float aq = 0.f, av = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { aq += s_pq[w * A + u]; av += s_pv[w * A + u]; }
        v1_st<COH>(dq + (size_t)b * A + u, aq);
        dvpart[(size_t)b * A + u] += av;              // a row's d keys / d v words belong to the same thread in every step
    }
}
__global__ __launch_bounds__(V1D_ATT_THREADS) void v1d_attn_bwd_kernel(const float* __restrict__ keys, const float* __restrict__ values, const float* __restrict__ q,
                                                                       const float* __restrict__ v, const int* __restrict__ lens, const float* __restrict__ align,
                                                                       const float* __restrict__ dctx, long long dctx_row_stride, int T, int A, int E,
                                                                       float* __restrict__ dq, float* __restrict__ dkeys, float* __restrict__ dvpart) {
    extern __shared__ float sm[];
    __shared__ float s_red[32];
    v1d_attn_bwd_body<false>(keys, values, q, v, lens, align, dctx, dctx_row_stride, T, A, E, dq, dkeys, dvpart, blockIdx.x, sm, s_red);
}
// ---- the attention of a row on V1D_ATT_C workgroups (persistent launches only) ----------------------------------------------------
// One workgroup per row is bound by its own vector pipe: T * A tanh per row and step = 3.3 us of a CU at T = 140, A = 256.  Inside
// the persistent launches workgroup (b, c) takes a quarter of the row.  Forward, two stages: (1) scores of positions chunk c ->
// en [b][j]; (2) every workgroup of the row normalises the whole row (redundantly) and adds up columns chunk c of the context
// (no partial sums cross workgroups).  Backward, one stage: positions chunk c; the row-wide term of the softmax backward is taken
// from the taped context (sum_j a_j da_j = d ctx . ctx, as in csrc/train_step.h), so the chunks need nothing from each other; their
// partial d q go to dq4 [b][c][A] and the query layer's backward product sums them as K = C * A against C stacked copies of W_q^T.
#define V1D_ATT_C 4
__device__ __forceinline__ void v1d_attn_scores_body(const float* __restrict__ keys, const float* q, const float* __restrict__ v, const int* __restrict__ lens,
                                                     int T, int A, float* en, int b, int c, float* sm) {
    constexpr int NW = V1D_ATT_THREADS / 64, U = 3;
    float *s_q = sm, *s_v = sm + A;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int len = lens[b], Tc = (T + V1D_ATT_C - 1) / V1D_ATT_C, j_lo = c * Tc, j_hi = min(len, j_lo + Tc);
    if (j_lo >= j_hi) return;                                   // (uniform over the workgroup)
    for (int u = tid; u < A; u += V1D_ATT_THREADS) { s_q[u] = v1_ld<true>(q + (size_t)b * A + u); s_v[u] = v[u]; }
    __syncthreads();
    for (int j0 = j_lo + wave; j0 < j_hi; j0 += NW * U) {
        float p[U];
#pragma unroll
        for (int i = 0; i < U; ++i) p[i] = 0.f;
        for (int u0 = 0; u0 < A; u0 += 256) {
            float kk[U][4];
#pragma unroll
            for (int i = 0; i < U; ++i)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) kk[i][cc] = keys[((size_t)b * T + min(j0 + i * NW, j_hi - 1)) * A + min(u0 + lane + 64 * cc, A - 1)];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int u = u0 + lane + 64 * cc;
                const float qv = s_q[min(u, A - 1)], vv = u < A ? s_v[u] : 0.f;
#pragma unroll
                for (int i = 0; i < U; ++i) p[i] += vv * fast_tanhf(kk[i][cc] + qv);
            }
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p[i] += __shfl_xor(p[i], off);
            if (lane == 0 && j0 + i * NW < j_hi) v1_st<true>(en + (size_t)b * T + j0 + i * NW, p[i]);
        }
    }
}
__device__ __forceinline__ void v1d_attn_context_body(const float* __restrict__ values, const float* en, const int* __restrict__ lens, int T, int E,
                                                      float* __restrict__ align, V1Dst c0, V1Dst c1, int b, int c, float* sm, float* s_red) {
    constexpr int NW = V1D_ATT_THREADS / 64, U = V1D_ATT_U;
    const int Ec = (E + V1D_ATT_C - 1) / V1D_ATT_C, e_lo = c * Ec, e_hi = min(E, e_lo + Ec);
    float *s_e = sm, *s_part = sm + T;                           // s_part [NW][Ec]
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int len = lens[b];
    float mx = -INFINITY;
    for (int j = tid; j < len; j += V1D_ATT_THREADS) { const float e = v1_ld<true>(en + (size_t)b * T + j); s_e[j] = e; mx = fmaxf(mx, e); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, s_red[w]);
    float sum = 0.f;
    for (int j = tid; j < T; j += V1D_ATT_THREADS) { const float e = j < len ? expf(s_e[j] - mx) : 0.f; s_e[j] = e; sum += e; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) s_red[NW + wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < NW; ++w) sum += s_red[NW + w];
    for (int j = tid; j < T; j += V1D_ATT_THREADS) { const float p = s_e[j] / sum; s_e[j] = p; if (c == 0) align[(size_t)b * T + j] = p; }
    __syncthreads();
    for (int e0 = e_lo; e0 < e_hi; e0 += 64) {                   // a lane per column of the chunk, a wave per position (mod 16)
        const int e = min(e0 + lane, e_hi - 1);
        float acc = 0.f;
        for (int j0 = wave; j0 < len; j0 += NW * U) {
            float vv[U];
#pragma unroll
            for (int i = 0; i < U; ++i) vv[i] = values[((size_t)b * T + min(j0 + i * NW, len - 1)) * E + e];
#pragma unroll
            for (int i = 0; i < U; ++i) acc = fmaf(j0 + i * NW < len ? s_e[j0 + i * NW] : 0.f, vv[i], acc);
        }
        if (e0 + lane < e_hi) s_part[wave * Ec + e0 - e_lo + lane] = acc;
    }
    __syncthreads();
    for (int i = tid; i < e_hi - e_lo; i += V1D_ATT_THREADS) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += s_part[w * Ec + i];
        if (c0.p) v1_st<true>(c0.p + (size_t)b * c0.ld + e_lo + i, acc);
        if (c1.p) v1_st<true>(c1.p + (size_t)b * c1.ld + e_lo + i, acc);
    }
}
__device__ __forceinline__ void v1d_attn_bwd_chunk_body(const float* __restrict__ keys, const float* __restrict__ values, const float* __restrict__ q,
                                                        const float* __restrict__ v, const int* __restrict__ lens, const float* __restrict__ align, const float* dctx,
                                                        long long dctx_row_stride, const float* __restrict__ ctx, int ldctx, int T, int A, int E, float* dq4,
                                                        float* dkeys, float* dvpart, int b, int c, float* sm, float* s_red) {
    // A, E <= 256 (dt_persist_ok): a lane holds four features of a position, a wave three positions per batch.  Everything a batch
    // reads (values, keys, d keys, alignments) depends on nothing of this step: the first batch is requested before the row-wide
    // term s is reduced, so the stage pays one memory round trip, not three.
    constexpr int NW = V1D_ATT_THREADS / 64, U = 3;
    const int Tc = (T + V1D_ATT_C - 1) / V1D_ATT_C;
    float *s_dc = sm, *s_pq = sm + E, *s_pv = s_pq + NW * A;                              // partial d q / d v [NW][A]
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int len = lens[b], j_lo = c * Tc, j_hi = min(len, j_lo + Tc);
    float dcv = 0.f, cxv = 0.f;
    if (tid < E) { dcv = v1_ld<true>(dctx + (size_t)b * dctx_row_stride + tid); cxv = ctx[(size_t)b * ldctx + tid]; }
    float vv[U][4], kk[U][4], dk[U][4], al[U], qu[4], vu[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) { const int u = min(lane + 64 * cc, A - 1); qu[cc] = q[(size_t)b * A + u]; vu[cc] = v[u]; }
    auto load = [&](int j0) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const size_t row = (size_t)b * T + min(j0 + i * NW, j_hi - 1);
            al[i] = align[row];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                vv[i][cc] = values[row * E + min(lane + 64 * cc, E - 1)];
                kk[i][cc] = keys[row * A + min(lane + 64 * cc, A - 1)];
                dk[i][cc] = dkeys[row * A + min(lane + 64 * cc, A - 1)];
            }
        }
    };
    const int jw = j_lo + wave;
    if (jw < j_hi) load(jw);
    // s = sum_j a_j da_j over the whole row = d ctx . ctx
    float s = dcv * cxv;
    if (tid < E) s_dc[tid] = dcv;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) s_red[wave] = s;
    __syncthreads();
    s = 0.f;
    for (int w = 0; w < NW; ++w) s += s_red[w];
    float dcl[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) dcl[cc] = lane + 64 * cc < E ? s_dc[lane + 64 * cc] : 0.f;
    float aq[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = jw; j0 < j_hi; j0 += NW * U) {
        if (j0 != jw) load(j0);
#pragma unroll
        for (int i = 0; i < U; ++i) {
            float p = 0.f;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) p += dcl[cc] * vv[i][cc];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
            const bool live = j0 + i * NW < j_hi;
            const float de = live ? al[i] * (p - s) : 0.f;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float th = fast_tanhf(kk[i][cc] + qu[cc]);
                const float gg = de * vu[cc] * (1.0f - th * th);
                aq[cc] += gg; av[cc] += de * th;
                if (live && lane + 64 * cc < A) dkeys[((size_t)b * T + j0 + i * NW) * A + lane + 64 * cc] = dk[i][cc] + gg;
            }
        }
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
        if (lane + 64 * cc < A) { s_pq[wave * A + lane + 64 * cc] = aq[cc]; s_pv[wave * A + lane + 64 * cc] = av[cc]; }
    __syncthreads();
    for (int u = tid; u < A; u += V1D_ATT_THREADS) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { a1 += s_pq[w * A + u]; a2 += s_pv[w * A + u]; }
        v1_st<true>(dq4 + ((size_t)b * V1D_ATT_C + c) * A + u, a1);
        dvpart[((size_t)b * V1D_ATT_C + c) * A + u] += a2;      // a chunk's d keys / d v words belong to the same thread in every step
    }
}
// dq [m][u] = sum_c dq4 [m][c][u] (after a persistent backward launch: the tape the query layer's weight gradient reads)
__global__ __launch_bounds__(256) void v1d_sum_chunks_kernel(const float* __restrict__ dq4, float* __restrict__ dq, long long rows, int A) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * A; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / A; const int u = (int)(i - m * A);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < V1D_ATT_C; ++c) sum += dq4[((size_t)m * V1D_ATT_C + c) * A + u];
        dq[i] = sum;
    }
}
// GRU cell backward, stage 1: d h' = d0 (+ d1) -> d zc, d zu, and the part of d h that does not pass a GEMM (d h' u)
__global__ __launch_bounds__(256) void v1d_bwd1_kernel(const float* __restrict__ d0, int ld0, const float* __restrict__ d1, int ld1, const float* __restrict__ cat,
                                                       const float* __restrict__ ru, const float* __restrict__ cnd, int X, int H, float* __restrict__ dzc,
                                                       float* __restrict__ dzg, float* __restrict__ keep, int B) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * H; i += gridDim.x * blockDim.x) {
        const int b = i / H, u = i - b * H;
        float dh = d0[(size_t)b * ld0 + u];
        if (d1) dh += d1[(size_t)b * ld1 + u];
        const float uu = ru[(size_t)b * 2 * H + H + u], c = cnd[i], h = cat[(size_t)b * (X + H) + X + u];
        dzc[i] = dh * (1.0f - uu) * (1.0f - c * c);
        dzg[(size_t)b * 2 * H + H + u] = dh * (h - c) * uu * (1.0f - uu);
        keep[i] = dh * uu;
    }
}
// stage 2, after d [x | r h] = d zc . Wc^T: d r = d(r h) h -> d zr; keep += d(r h) r
__global__ __launch_bounds__(256) void v1d_bwd2_kernel(const float* __restrict__ dcatr, const float* __restrict__ cat, const float* __restrict__ ru, int X, int H,
                                                       float* __restrict__ dzg, float* __restrict__ keep, int B) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * H; i += gridDim.x * blockDim.x) {
        const int b = i / H, u = i - b * H;
        const float drh = dcatr[(size_t)b * (X + H) + X + u], r = ru[(size_t)b * 2 * H + u], h = cat[(size_t)b * (X + H) + X + u];
        dzg[(size_t)b * 2 * H + u] = drh * h * r * (1.0f - r);
        keep[i] += drh * r;
    }
}
// stage 3, after d [x | h] = d zg . Wg^T: the state gradient handed to the step before, and d x (+ the residual's d y)
__global__ __launch_bounds__(256) void v1d_bwd3_kernel(const float* __restrict__ dcatr, const float* __restrict__ dcat, const float* __restrict__ keep,
                                                       const float* __restrict__ dres, int ldres, int X, int H, float* __restrict__ carry,
                                                       float* __restrict__ dx, int lddx, int B) {
    const int W = X + H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * W; i += gridDim.x * blockDim.x) {
        const int b = i / W, c = i - b * W;
        if (c < X) {
            float g = dcatr[(size_t)b * W + c] + dcat[(size_t)b * W + c];
            if (dres) g += dres[(size_t)b * ldres + c];
            dx[(size_t)b * lddx + c] = g;
        } else {
            carry[(size_t)b * H + c - X] = keep[(size_t)b * H + c - X] + dcat[i];
        }
    }
}
// d ctx_t = (output projection's share) + (the share carried from step t + 1's attention GRU) -> the tape [B][S][E]
__global__ __launch_bounds__(256) void v1d_dctx_kernel(const float* __restrict__ dcato, int A, int E, const float* __restrict__ carry, float* __restrict__ dctx_t,
                                                       long long row_stride, int B) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * E; i += gridDim.x * blockDim.x) {
        const int b = i / E, e = i - b * E;
        dctx_t[(size_t)b * row_stride + e] = dcato[(size_t)b * (A + E) + A + e] + carry[i];
    }
}
// mel [S][B][r*M] -> [B][S*r][M]
__global__ __launch_bounds__(256) void v1d_sb_to_bt_kernel(const float* __restrict__ sb, float* __restrict__ bt, int B, int S, int rM) {
    const long long n = (long long)S * B * rM;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % rM), b = (int)((i / rM) % B), t = (int)(i / ((long long)rM * B));
        bt[((size_t)b * S + t) * rM + c] = sb[i];
    }
}
// {mel_loss, done_loss, loss} from the two {sum, count, ratio} triples
__global__ void v1d_loss_out_kernel(const float* __restrict__ l6, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = l6[2]; out[1] = l6[5]; out[2] = l6[2] + l6[5]; }
}
// d loss / d mel (masked L1, SUM_BY_NONZERO_WEIGHTS) in the [S][B][r*M] layout, and d loss / d stop logit (sigmoid cross entropy) [S][B]
__global__ __launch_bounds__(256) void v1d_loss_grad_kernel(const float* __restrict__ mel_sb, const float* __restrict__ stop_sb, const float* __restrict__ target,
                                                            const float* __restrict__ done, const float* __restrict__ m1, const float* __restrict__ m2,
                                                            const float* __restrict__ n1, const float* __restrict__ n2, int B, int S, int r, int M,
                                                            float* __restrict__ dmel_sb, float* __restrict__ dstop_sb) {
    const int rM = r * M;
    const float i1 = n1[0] > 0.f ? 1.0f / n1[0] : 0.f, i2 = n2[0] > 0.f ? 1.0f / n2[0] : 0.f;
    const long long n = (long long)S * B * (rM + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (rM + 1)), b = (int)((i / (rM + 1)) % B), t = (int)(i / ((long long)(rM + 1) * B));
        if (c < rM) {
            const int frame = t * r + c / M;
            const float d = mel_sb[((size_t)t * B + b) * rM + c] - target[((size_t)b * S * r + frame) * M + c % M];
            dmel_sb[((size_t)t * B + b) * rM + c] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m1[(size_t)b * S * r + frame] * i1;
        } else {
            const float z = stop_sb[(size_t)t * B + b];
            dstop_sb[(size_t)t * B + b] = (sigmoidf_(z) - done[(size_t)b * S + t]) * m2[(size_t)b * S + t] * i2;
        }
    }
}
__global__ __launch_bounds__(256) void v1d_stop_to_bs_kernel(const float* __restrict__ stop_sb, float* __restrict__ stop_bs, int B, int S) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * S; i += gridDim.x * blockDim.x) {
        const int t = i / B, b = i - t * B;
        stop_bs[(size_t)b * S + t] = stop_sb[i];
    }
}

// y [rows][N] = x [rows][K] . w [K][N] (+ bias) (+ res) for rows <= 64 (the per-step products of the recurrence: 32 batch rows against
// 256..640 x 256..512 kernels in TF layout).  conv_gemm_*'s 64 x 128 tiles leave 2-4 workgroups with a serial K loop for such a
// product (31 us each, measured); here a workgroup owns 16 output columns and all rows, its eight waves split K (k-blocks of 16
// round robin) on v_mfma_f32_16x16x4_f32 and add their partial tiles through LDS: N / 16 workgroups, ~3 us.
// MFMA step j of a k-block takes k = 4 kg + j from lane group kg (A: one float4 of x per lane and row block; B: W[k][n0 + c16]).
// The elementwise stage that consumes a product rides in its epilogue (EPI): a launch of a few microseconds each, six per step
// forward and seven backward.
enum { SK_PLAIN = 0, SK_GATE = 1, SK_BLEND = 2, SK_BWD2 = 3, SK_BWD3 = 4, SK_DCTX = 5, SK_BWD1 = 6 };
// stage 1 of a GRU cell's backward (v1d_bwd1_kernel) on a value the epilogue has just produced: d h' = value + add
struct SkBwd1 {
    const float* add = nullptr; int ldadd = 0; const float* cat = nullptr; const float* ru = nullptr; const float* cnd = nullptr; int X = 0, H = 0;
    float* dzc = nullptr; float* dzg = nullptr; float* keep = nullptr;
};
__device__ __forceinline__ void sk_bwd1(const SkBwd1& q, int row, int u, float dh) {
    if (q.add) dh += q.add[(size_t)row * q.ldadd + u];
    const float uu = q.ru[(size_t)row * 2 * q.H + q.H + u], c = q.cnd[(size_t)row * q.H + u], h = q.cat[(size_t)row * (q.X + q.H) + q.X + u];
    q.dzc[(size_t)row * q.H + u] = dh * (1.0f - uu) * (1.0f - c * c);
    q.dzg[(size_t)row * 2 * q.H + q.H + u] = dh * (h - c) * uu * (1.0f - uu);
    q.keep[(size_t)row * q.H + u] = dh * uu;
}
struct SkEpi {
    const float* bias = nullptr; const float* res = nullptr; int ldr = 0;    // v = product + bias[col] + res[row][col]
    float* y = nullptr; int ldy = 0;                                          // v itself (null: not stored)
    const float* cat = nullptr; int X = 0, H = 0;                             // the cell's operand [x | h], row stride X + H
    float* ru = nullptr;                                                      // [r | u], row stride 2H (written by GATE, read by the others)
    float* catr = nullptr;                                                    // GATE: the candidate's operand [x | r h]
    float* cnd = nullptr; V1Dst h0{nullptr, 0}, h1{nullptr, 0}, y0{nullptr, 0}, y1{nullptr, 0};   // BLEND (as v1d_blend_kernel)
    float* dzg = nullptr; float* keep = nullptr;                              // BWD2 (as v1d_bwd2_kernel)
    const float* dcatr = nullptr; const float* dres = nullptr; int ldres = 0; float* carry = nullptr; float* dx = nullptr; int lddx = 0;   // BWD3
    const float* cin = nullptr; float* dctx = nullptr; long long row_stride = 0; int A = 0;      // DCTX (as v1d_dctx_kernel)
    SkBwd1 b1x, b1c;   // BWD3: stage 1 of the cell fed by d x (same step) / of this cell one step earlier (fed by the carry); BWD1: b1x on v
};
#ifdef V1D_STAMPS
__device__ unsigned long long g_v1d_body[8];                // diagnostic: ticks of a product stage up to [0] loads issued, [1] K loop done, [2] reduced (barrier), [3] end
#endif
// what an output element's epilogue reads besides the product, fetched before the K loop so that its latency hides behind it
struct SkPre { float f[6]; };
// Which optional parts of SkEpi a stage uses: SKF_DYN = look at the pointers (the one-launch-per-stage kernels), otherwise a
// compile-time set (the persistent kernels: no branches around the loads, so that all of them are in flight together).
enum { SKF_DYN = 1 << 15, SKF_BIAS = 1, SKF_RES = 2, SKF_Y = 4, SKF_H0 = 8, SKF_H1 = 16, SKF_Y0 = 32, SKF_Y1 = 64, SKF_DRES = 128, SKF_B1X = 256, SKF_B1C = 512,
       SKF_ADD = 1024 };
template <int FL, int F> __device__ __forceinline__ bool sk_has(const void* p) {
    if constexpr ((FL & SKF_DYN) != 0) return p != nullptr;
    else return (FL & F) != 0;
}
template <bool COH, int FL>
__device__ __forceinline__ void sk_bwd1_load(const SkBwd1& q, int row, int u, float* o) {
    o[0] = sk_has<FL, SKF_ADD>(q.add) ? v1_ld<COH>(q.add + (size_t)row * q.ldadd + u) : 0.f;
    o[1] = v1_ld<COH>(q.ru + (size_t)row * 2 * q.H + q.H + u); o[2] = v1_ld<COH>(q.cnd + (size_t)row * q.H + u);
    o[3] = v1_ld<COH>(q.cat + (size_t)row * (q.X + q.H) + q.X + u);
}
template <bool COH>
__device__ __forceinline__ void sk_bwd1_apply(const SkBwd1& q, int row, int u, float dh, const float* o) {
    dh += o[0];
    const float uu = o[1], c = o[2], h = o[3];
    v1_st<COH>(q.dzc + (size_t)row * q.H + u, dh * (1.0f - uu) * (1.0f - c * c));
    v1_st<COH>(q.dzg + (size_t)row * 2 * q.H + q.H + u, dh * (h - c) * uu * (1.0f - uu));
    v1_st<COH>(q.keep + (size_t)row * q.H + u, dh * uu);
}
template <int EPI, bool COH, int FL>
__device__ __forceinline__ SkPre sk_pre(const SkEpi& e, int row, int col, int N) {
    SkPre p;
#pragma unroll
    for (int i = 0; i < 6; ++i) p.f[i] = 0.f;
    if (sk_has<FL, SKF_BIAS>(e.bias)) p.f[0] = e.bias[col];
    if (sk_has<FL, SKF_RES>(e.res)) p.f[0] += v1_ld<COH>(e.res + (size_t)row * e.ldr + col);
    const int X = e.X, H = e.H, Wd = X + H;
    if constexpr (EPI == SK_GATE) {
        if (col < H) p.f[1] = v1_ld<COH>(e.cat + (size_t)row * Wd + X + col);
    } else if constexpr (EPI == SK_BLEND) {
        p.f[1] = v1_ld<COH>(e.ru + (size_t)row * 2 * H + H + col); p.f[2] = v1_ld<COH>(e.cat + (size_t)row * Wd + X + col);
        if (sk_has<FL, SKF_Y0>(e.y0.p) || sk_has<FL, SKF_Y1>(e.y1.p)) p.f[3] = v1_ld<COH>(e.cat + (size_t)row * Wd + col);
    } else if constexpr (EPI == SK_BWD2) {
        if (col >= X) {
            p.f[1] = v1_ld<COH>(e.ru + (size_t)row * 2 * H + col - X); p.f[2] = v1_ld<COH>(e.cat + (size_t)row * Wd + col);
            p.f[3] = v1_ld<COH>(e.keep + (size_t)row * H + col - X);
        }
    } else if constexpr (EPI == SK_BWD3) {
        if (col < X) {
            p.f[1] = v1_ld<COH>(e.dcatr + (size_t)row * Wd + col);
            if (sk_has<FL, SKF_DRES>(e.dres)) p.f[1] += v1_ld<COH>(e.dres + (size_t)row * e.ldres + col);
            if (sk_has<FL, SKF_B1X>(e.b1x.dzc)) sk_bwd1_load<COH, FL>(e.b1x, row, col, p.f + 2);
        } else {
            p.f[1] = v1_ld<COH>(e.keep + (size_t)row * H + col - X);
            if (sk_has<FL, SKF_B1C>(e.b1c.dzc)) sk_bwd1_load<COH, FL>(e.b1c, row, col - X, p.f + 2);
        }
    } else if constexpr (EPI == SK_DCTX) {
        if (col >= e.A) p.f[1] = v1_ld<COH>(e.cin + (size_t)row * (N - e.A) + col - e.A);
    } else if constexpr (EPI == SK_BWD1) {
        sk_bwd1_load<COH, FL>(e.b1x, row, col, p.f + 2);
    }
    return p;
}
template <int EPI, bool COH, int FL>
__device__ __forceinline__ void sk_post(const SkEpi& e, int row, int col, float v, const SkPre& p) {
    v += p.f[0];
    if (sk_has<FL, SKF_Y>(e.y)) v1_st<COH>(e.y + (size_t)row * e.ldy + col, v);
    const int X = e.X, H = e.H, Wd = X + H;
    if constexpr (EPI == SK_GATE) {                   // N = 2H: [r | u] = sigmoid(v); [x | r h]
        const float sg = sigmoidf_(v);
        v1_st<COH>(e.ru + (size_t)row * 2 * H + col, sg);
        if (col < H) v1_st<COH>(e.catr + (size_t)row * Wd + X + col, sg * p.f[1]);
    } else if constexpr (EPI == SK_BLEND) {           // N = H: c = tanh(v); h' = u h + (1 - u) c; residual y = x + h'
        const float cc = tanhf(v), uu = p.f[1], h = p.f[2];
        const float hn = uu * h + (1.0f - uu) * cc;
        v1_st<COH>(e.cnd + (size_t)row * H + col, cc);
        if (sk_has<FL, SKF_H0>(e.h0.p)) v1_st<COH>(e.h0.p + (size_t)row * e.h0.ld + col, hn);
        if (sk_has<FL, SKF_H1>(e.h1.p)) v1_st<COH>(e.h1.p + (size_t)row * e.h1.ld + col, hn);
        if (sk_has<FL, SKF_Y0>(e.y0.p) || sk_has<FL, SKF_Y1>(e.y1.p)) {
            const float yy = p.f[3] + hn;
            if (sk_has<FL, SKF_Y0>(e.y0.p)) v1_st<COH>(e.y0.p + (size_t)row * e.y0.ld + col, yy);
            if (sk_has<FL, SKF_Y1>(e.y1.p)) v1_st<COH>(e.y1.p + (size_t)row * e.y1.ld + col, yy);
        }
    } else if constexpr (EPI == SK_BWD2) {            // N = X + H: v = d [x | r h]; d zr, keep += d(r h) r
        if (col >= X) {
            const float rg = p.f[1], h = p.f[2];
            v1_st<COH>(e.dzg + (size_t)row * 2 * H + col - X, v * h * rg * (1.0f - rg));
            v1_st<COH>(e.keep + (size_t)row * H + col - X, p.f[3] + v * rg);
        }
    } else if constexpr (EPI == SK_BWD3) {            // N = X + H: v = d [x | h] of the gates' product
        if (col < X) {
            const float g = p.f[1] + v;
            v1_st<COH>(e.dx + (size_t)row * e.lddx + col, g);
            if (sk_has<FL, SKF_B1X>(e.b1x.dzc)) sk_bwd1_apply<COH>(e.b1x, row, col, g, p.f + 2);
        } else {
            const float cr = p.f[1] + v;
            v1_st<COH>(e.carry + (size_t)row * H + col - X, cr);
            if (sk_has<FL, SKF_B1C>(e.b1c.dzc)) sk_bwd1_apply<COH>(e.b1c, row, col - X, cr, p.f + 2);
        }
    } else if constexpr (EPI == SK_DCTX) {            // N = A + E: v = d [h_att | ctx] of the output projection
        if (col >= e.A) v1_st<COH>(e.dctx + (size_t)row * e.row_stride + col - e.A, v + p.f[1]);
    } else if constexpr (EPI == SK_BWD1) {
        sk_bwd1_apply<COH>(e.b1x, row, col, v, p.f + 2);
    }
}
// one product + epilogue by workgroup `wg` of `nwg` (16 columns each) with NWV waves; red: LDS [NWV][MB][64] float4
template <int MB, int EPI, int NWV, bool COH, int FL = SKF_DYN, int UKX = 0>
__device__ __forceinline__ void sk_body(const float* x, int ldx, int K, const float* __restrict__ w, int N, const SkEpi& e, int rows, int wg, int nwg,
                                        f32x4* red, int wmod = 0) {        // wmod: w holds wmod rows, used by every K segment of that length
    constexpr int NT_ = 64 * NWV, UK = UKX ? UKX : (NWV >= 16 ? 2 : 4), EL = (MB * 256 + NT_ - 1) / NT_;       // UK k-blocks in flight per wave: K <= 512 is one batch
    int tid = threadIdx.x;
    // inside the persistent loops: an opaque copy of the thread index, or every stage's step-invariant addresses are hoisted out of
    // the step loop, stay live across all stages and push the K loop's loads into a wait-per-load schedule (persist.h, fresh())
    if constexpr (COH) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), kg = lane >> 4, c16 = lane & 15;
    const int n0 = wg * 16, n = min(n0 + c16, N - 1);
    // epilogue ownership: element i = tid + NT_ * el -> (row, column) with the column fastest
#ifdef V1D_STAMPS
    const bool stamp_ = COH && wg == 0 && threadIdx.x == 0;
    unsigned long long tb_ = __builtin_amdgcn_s_memrealtime();
#define SKB_STAMP(i) do { if (stamp_) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); g_v1d_body[i] += n_ - tb_; tb_ = n_; } } while (0)
#else
#define SKB_STAMP(i) do { } while (0)
#endif
    SkPre pre[EL];
#pragma unroll
    for (int el = 0; el < EL; ++el) {
        const int i = tid + NT_ * el, row = i >> 4, col = n0 + (i & 15);
        if (i < MB * 256 && row < rows && col < N) pre[el] = sk_pre<EPI, COH, FL>(e, row, col, N);
    }
    f32x4 acc[MB];
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkb = K >> 4;
    [[maybe_unused]] const pz::rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7ffffff0, 0x00020000);
    for (int kb0 = wave; kb0 < nkb; kb0 += NWV * UK) {
        f32x4 a[UK][MB];
        float bw[UK][4];
#pragma unroll
        for (int i = 0; i < UK; ++i) {
            const int kb = min(kb0 + NWV * i, nkb - 1), k0 = kb * 16 + 4 * kg;
#pragma unroll
            for (int rb = 0; rb < MB; ++rb) {
                const size_t off = (size_t)min(rb * 16 + c16, rows - 1) * ldx + k0;
                if constexpr (COH) a[i][rb] = pz::ld16(xr, (unsigned)(off * 4));
                else a[i][rb] = *reinterpret_cast<const f32x4*>(x + off);
            }
            const int kw = wmod ? k0 % wmod : k0;
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[i][j] = w[(size_t)(kw + j) * N + n];
        }
        if constexpr (COH) __builtin_amdgcn_sched_barrier(0);     // every load of the batch is in flight before the first product waits
        SKB_STAMP(0);
#pragma unroll
        for (int i = 0; i < UK; ++i) {
            const bool live = kb0 + NWV * i < nkb;           // a clamped (repeated) k-block contributes nothing
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bj = live ? bw[i][j] : 0.f;
#pragma unroll
                for (int rb = 0; rb < MB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][rb][j], bj, acc[rb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < MB; ++rb) red[(wave * MB + rb) * 64 + lane] = acc[rb];
    SKB_STAMP(1);
    if constexpr (EPI == SK_GATE) {
        // the x half of the candidate's operand is a copy: every workgroup moves its share of the rows * X words
        const int total = rows * e.X, per = (total + nwg - 1) / nwg, Wd = e.X + e.H;
        for (int i = wg * per + tid; i < min(total, (wg + 1) * per); i += NT_) {
            const int b = i / e.X, c = i - b * e.X;
            v1_st<COH>(e.catr + (size_t)b * Wd + c, v1_ld<COH>(e.cat + (size_t)b * Wd + c));
        }
    }
    __syncthreads();
    SKB_STAMP(2);
    // D layout: reg r of lane (kg, c16) = row 4 kg + r, column c16
#pragma unroll
    for (int el = 0; el < EL; ++el) {
        const int i = tid + NT_ * el, c = i & 15, rr = i >> 4, rb = rr >> 4, l = ((rr >> 2) & 3) * 16 + c, r = rr & 3;
        const int row = rr, col = n0 + c;
        if (i >= MB * 256 || row >= rows || col >= N) continue;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < NWV; ++wv) v += red[(wv * MB + rb) * 64 + l][r];
        sk_post<EPI, COH, FL>(e, row, col, v, pre[el]);
    }
    SKB_STAMP(3);
}
#define V1D_SK_WAVES 8
template <int MB, int EPI>
__global__ __launch_bounds__(64 * V1D_SK_WAVES) void v1d_skinny_gemm_kernel(const float* __restrict__ x, int ldx, int K, const float* __restrict__ w, int N,
                                                                            SkEpi e, int rows) {
    __shared__ f32x4 red[V1D_SK_WAVES * MB * 64];
    sk_body<MB, EPI, V1D_SK_WAVES, false>(x, ldx, K, w, N, e, rows, blockIdx.x, gridDim.x, red);
}

// ---- both recurrences as ONE launch each -------------------------------------------------------------------------------------
// The stages of a decoder step (the launches of the loops below, same device code) run inside one resident grid of G workgroups x
// 16 waves, in two independent chains on the two halves of the grid; a stage hands its results to the next through persist.h's hop (write-through stores, drained, one
// counter add per workgroup; polled shards), ~2 us instead of a launch boundary plus the fixed cost of a launch.  Workgroup g
// owns output columns 16 g .. 16 g + 15 of a product (or batch row g of the attention); workgroups beyond a stage's width only
// hop.  Every spin is bounded (persist.h SPIN_LIMIT_TICKS): a timeout leaves the loop with the error word set and the caller
// re-runs the recurrence on the launch-per-stage schedule.
// Diagnostic builds only (-DV1D_STAMPS, tools/v1_train_stamps.py): thread 0 of the first workgroup of each half sums the time
// (s_memrealtime, 10 ns ticks) spent in stage bodies / arrive / wait per stage slot; the product build has no stamps.
#ifdef V1D_STAMPS
__device__ unsigned long long g_v1d_stamps[2][2][16];       // [kernel][half][3 * slot + {body, arrive, wait}]
#define V1D_STAMP(kern, i) do { if (threadIdx.x == 0 && lw == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    g_v1d_stamps[kern][grp][i] += now_ - st_prev; st_prev = now_; } } while (0)
#else
#define V1D_STAMP(kern, i) do { } while (0)
#endif
template <int F> using FLG = std::integral_constant<int, F>;
struct V1PFwd {
    int B, T, S, A, E, D;
    const float *Wga, *Wca;                    // rows [P1, P1 + E + A) of the attention GRU's kernels (the PreNet rows are batched)
    const float *Wq, *va, *Wo, *bo, *Wg1, *bg1, *Wc1, *bc1, *Wg2, *bg2, *Wc2, *bc2;
    const float *zgp, *zcp, *keys, *values;
    const int* lens;
    float *CATA, *CATAR, *RUa, *Ca, *Q, *ALIGN, *CATO, *CAT1, *CAT1R, *RU1, *C1, *CAT2, *CAT2R, *RU2, *C2, *Y2;
    float* en;                                 // scores of the step in flight [B][T]
    int nd;                                    // workgroups of the decoder chain (the rest: attention chain)
    unsigned* bar; int* err;
};
// Forward: two chains run side by side on the two halves of the grid, one stage per hop --
//   attention chain of step i      : attention GRU gates | candidate | query layer | scores | softmax + context   (needs ctx, h_att of step i - 1)
//   decoder chain of step i - 1    : output projection   | GRU 1 gates | candidate | GRU 2 gates | candidate   (needs [h_att | ctx] of step i - 1)
// so a decoder step costs five hops instead of nine.
template <int MB>
__global__ __launch_bounds__(V1D_ATT_THREADS) void v1d_fwd_persist_kernel(V1PFwd a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    __shared__ float s_red[32];
    __shared__ int s_abort;
    constexpr int NWV = V1D_ATT_THREADS / 64;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    const int B = a.B, T = a.T, S = a.S, A = a.A, E = a.E, D = a.D, XA = E + A;
    const int half = a.nd, grp = (int)blockIdx.x >= half ? 1 : 0, lw = blockIdx.x - grp * half;       // a.nd workgroups for the decoder chain, the rest for the attention chain
    const unsigned per = gridDim.x / pz::NSHARD;
    unsigned epoch = 0;
    f32x4* red = reinterpret_cast<f32x4*>(dsm);
    [[maybe_unused]] unsigned long long st_prev = __builtin_amdgcn_s_memrealtime();
    int slot = 0;
    auto hop = [&]() -> bool {
        V1D_STAMP(0, 3 * slot); pz::hop_arrive(a.bar); V1D_STAMP(0, 3 * slot + 1); ++epoch;
        const bool r = pz::hop_wait(a.bar, per * epoch, &s_abort);
        V1D_STAMP(0, 3 * slot + 2); slot = slot == 4 ? 0 : slot + 1;
        return r;
    };
    auto gates = [&](auto fl, const float* cat, float* catr, int X, int H, const float* Wg, const float* bg, const float* zgp, float* ru) {
        constexpr int FL = decltype(fl)::value;
        const int Wd = X + H, ng = (2 * H + 15) / 16;
        if (lw >= ng) return;
        SkEpi g; g.bias = bg; g.res = zgp; g.ldr = 2 * H; g.cat = cat; g.X = X; g.H = H; g.ru = ru; g.catr = catr;
        sk_body<MB, SK_GATE, NWV, true, FL>(cat, Wd, Wd, Wg, 2 * H, g, B, lw, ng, red);
    };
    auto blend = [&](auto fl, const float* cat, const float* catr, int X, int H, const float* Wc, const float* bc, const float* zcp, float* ru, float* cnd, V1Dst h0, V1Dst h1,
                     V1Dst y0) {
        constexpr int FL = decltype(fl)::value;
        const int Wd = X + H, nc = (H + 15) / 16;
        if (lw >= nc) return;
        SkEpi c; c.bias = bc; c.res = zcp; c.ldr = H; c.cat = cat; c.X = X; c.H = H; c.ru = ru; c.cnd = cnd; c.h0 = h0; c.h1 = h1; c.y0 = y0;
        sk_body<MB, SK_BLEND, NWV, true, FL>(catr, Wd, Wd, Wc, H, c, B, lw, nc, red);
    };
    auto plain = [&](auto fl, const float* x, int ldx, int K, const float* w, int N, const float* bias, float* y, int ldy) {
        constexpr int FL = decltype(fl)::value;
        const int nw = (N + 15) / 16;
        if (lw >= nw) return;
        SkEpi e; e.bias = bias; e.y = y; e.ldy = ldy;
        sk_body<MB, SK_PLAIN, NWV, true, FL>(x, ldx, K, w, N, e, B, lw, nw, red);
    };
    bool ok = true;
    for (int i = 0; i <= S; ++i) {
        const int t = grp ? i : i - 1;                        // this half's step
        const bool act = grp ? i < S : i >= 1;
        const size_t tt = act ? (size_t)t : 0;
        float *cata = a.CATA + tt * B * XA, *cata_n = cata + (size_t)B * XA, *catar = a.CATAR + tt * B * XA;
        float* cato = a.CATO + tt * B * (A + E);
        float *cat1 = a.CAT1 + tt * B * 2 * D, *cat1_n = cat1 + (size_t)B * 2 * D, *cat1r = a.CAT1R + tt * B * 2 * D;
        float *cat2 = a.CAT2 + tt * B * 2 * D, *cat2_n = cat2 + (size_t)B * 2 * D, *cat2r = a.CAT2R + tt * B * 2 * D;
        float *rua = a.RUa + tt * B * 2 * A, *ru1 = a.RU1 + tt * B * 2 * D, *ru2 = a.RU2 + tt * B * 2 * D, *q = a.Q + tt * B * A;
        if (act) {
            if (grp) gates(FLG<SKF_RES>{}, cata, catar, E, A, a.Wga, nullptr, a.zgp + tt * B * 2 * A, rua);
            else plain(FLG<SKF_BIAS | SKF_Y>{}, cato, A + E, A + E, a.Wo, D, a.bo, cat1, 2 * D);
        }
        if (!(ok = hop())) break;
        if (act) {
            if (grp) blend(FLG<SKF_RES | SKF_H0 | SKF_H1>{}, cata, catar, E, A, a.Wca, nullptr, a.zcp + tt * B * A, rua, a.Ca + tt * B * A, V1Dst{cato, A + E}, V1Dst{cata_n + E, XA}, V1Dst{nullptr, 0});
            else gates(FLG<SKF_BIAS>{}, cat1, cat1r, D, D, a.Wg1, a.bg1, nullptr, ru1);
        }
        if (!(ok = hop())) break;
        if (act) {
            if (grp) plain(FLG<SKF_Y>{}, cato, A + E, A, a.Wq, A, nullptr, q, A);
            else blend(FLG<SKF_BIAS | SKF_H0 | SKF_Y0>{}, cat1, cat1r, D, D, a.Wc1, a.bc1, nullptr, ru1, a.C1 + tt * B * D, V1Dst{cat1_n + D, 2 * D}, V1Dst{nullptr, 0}, V1Dst{cat2, 2 * D});
        }
        if (!(ok = hop())) break;
        if (act) {
            if (grp) {
                if (lw < B * V1D_ATT_C) v1d_attn_scores_body(a.keys, q, a.va, a.lens, T, A, a.en, lw / V1D_ATT_C, lw % V1D_ATT_C, dsm);
            } else gates(FLG<SKF_BIAS>{}, cat2, cat2r, D, D, a.Wg2, a.bg2, nullptr, ru2);
        }
        if (!(ok = hop())) break;
        if (act && grp) {
            if (lw < B * V1D_ATT_C)
                v1d_attn_context_body(a.values, a.en, a.lens, T, E, a.ALIGN + tt * B * T, V1Dst{cato + A, A + E}, V1Dst{cata_n, XA}, lw / V1D_ATT_C, lw % V1D_ATT_C, dsm, s_red);
        }
        if (act && !grp) blend(FLG<SKF_BIAS | SKF_H0 | SKF_Y0>{}, cat2, cat2r, D, D, a.Wc2, a.bc2, nullptr, ru2, a.C2 + tt * B * D, V1Dst{cat2_n + D, 2 * D}, V1Dst{nullptr, 0}, V1Dst{a.Y2 + tt * B * D, D});
        if (!(ok = hop())) break;
    }
    if (!ok && threadIdx.x == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct V1PBwd {
    int B, T, S, A, E, D;
    const float *Wc2T, *Wg2T, *Wc1T, *Wg1T, *WoT, *WqT, *WcaT, *WgaT, *va, *keys, *values;
    const int* lens;
    const float *CATA, *CATO, *RUa, *Ca, *Q, *ALIGN, *CAT1, *RU1, *C1, *CAT2, *RU2, *C2, *DY2;
    float *DZC2, *DZG2, *DZC1, *DZG1, *DY0, *DQ4, *DZCa, *DZGa, *DCTX, *dkeys, *dvpart;     // DQ4 [S][B][C][A], dvpart [B][C][A]
    int nd;
    float *cHA, *cCTX, *cH1, *cH2, *keep, *keep1, *keepA, *dcatr, *dcatrA, *dy1, *dcato, *sink;
    unsigned* bar; int* err;
};
// Backward, last step first: the same split --
//   decoder chain of step S - 1 - i  : GRU 2 stages 2, 3 | GRU 1 stages 2, 3                          (needs only its own carries and d y2)
//   attention chain of step S - i    : output projection | attention | query layer + stage 1 | attention GRU stages 2, 3   (needs d y0 of its step)
template <int MB>
__global__ __launch_bounds__(V1D_ATT_THREADS) void v1d_bwd_persist_kernel(V1PBwd a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    __shared__ float s_red[32];
    __shared__ int s_abort;
    constexpr int NWV = V1D_ATT_THREADS / 64;
    if (threadIdx.x == 0) s_abort = 0;
    __syncthreads();
    const int B = a.B, T = a.T, S = a.S, A = a.A, E = a.E, D = a.D, XA = E + A;
    const int half = a.nd, grp = (int)blockIdx.x >= half ? 1 : 0, lw = blockIdx.x - grp * half;
    const unsigned per = gridDim.x / pz::NSHARD;
    unsigned epoch = 0;
    f32x4* red = reinterpret_cast<f32x4*>(dsm);
    [[maybe_unused]] unsigned long long st_prev = __builtin_amdgcn_s_memrealtime();
    int slot = 0;
    auto hop = [&]() -> bool {
        V1D_STAMP(1, 3 * slot); pz::hop_arrive(a.bar); V1D_STAMP(1, 3 * slot + 1); ++epoch;
        const bool r = pz::hop_wait(a.bar, per * epoch, &s_abort);
        V1D_STAMP(1, 3 * slot + 2); slot = slot == 4 ? 0 : slot + 1;
        return r;
    };
    // stages 2 and 3 of a GRU cell's backward (stage 1 rode in the epilogue that produced the cell's d h')
    auto stage2 = [&](const float* cat, const float* ru, int X, int H, const float* WcT, const float* dzc, float* dzg, float* keep, float* dcatr) {
        const int Wd = X + H, nw = (Wd + 15) / 16;
        if (lw >= nw) return;
        SkEpi e2; e2.y = dcatr; e2.ldy = Wd; e2.cat = cat; e2.X = X; e2.H = H; e2.ru = const_cast<float*>(ru); e2.dzg = dzg; e2.keep = keep;
        sk_body<MB, SK_BWD2, NWV, true, SKF_Y>(dzc, H, H, WcT, Wd, e2, B, lw, nw, red);
    };
    auto stage3 = [&](auto fl, int X, int H, const float* WgT, const float* dzg, const float* dres, int ldres, float* carry, float* dx, int lddx, float* keep, const float* dcatr,
                      const SkBwd1* b1x, const SkBwd1* b1c) {
        constexpr int FL = decltype(fl)::value;
        const int Wd = X + H, nw = (Wd + 15) / 16;
        if (lw >= nw) return;
        SkEpi e3; e3.X = X; e3.H = H; e3.dcatr = dcatr; e3.dres = dres; e3.ldres = ldres; e3.keep = keep; e3.carry = carry; e3.dx = dx; e3.lddx = lddx;
        if (b1x) e3.b1x = *b1x;
        if (b1c) e3.b1c = *b1c;
        sk_body<MB, SK_BWD3, NWV, true, FL>(dzg, 2 * H, 2 * H, WgT, Wd, e3, B, lw, nw, red);
    };
    bool ok = true;
    for (int i = 0; i <= S; ++i) {
        const int t = grp ? S - i : S - 1 - i;
        const bool act = grp ? i >= 1 : i < S;
        const size_t tt = act ? (size_t)t : 0;
        const float* cata = a.CATA + tt * B * XA;
        const float *cat1 = a.CAT1 + tt * B * 2 * D, *cat2 = a.CAT2 + tt * B * 2 * D;
        const float* dy2 = a.DY2 + tt * B * D;
        float *dy0 = a.DY0 + tt * B * D, *dctx_t = a.DCTX + tt * E;
        // stage 0: GRU 2 stage 2 | output projection backward (+ d ctx)
        if (act) {
            if (!grp) stage2(cat2, a.RU2 + tt * B * 2 * D, D, D, a.Wc2T, a.DZC2 + tt * B * D, a.DZG2 + tt * B * 2 * D, a.keep, a.dcatr);
            else if (lw < (A + E + 15) / 16) {
                SkEpi e; e.y = a.dcato; e.ldy = A + E; e.cin = a.cCTX; e.dctx = dctx_t; e.row_stride = (long long)S * E; e.A = A;
                sk_body<MB, SK_DCTX, NWV, true, SKF_Y>(dy0, D, D, a.WoT, A + E, e, B, lw, (A + E + 15) / 16, red);
            }
        }
        if (!(ok = hop())) break;
        // stage 1: GRU 2 stage 3 (+ stage 1 of GRU 1 at this step and of GRU 2 one step earlier) | attention backward
        if (act) {
            if (!grp) {
                SkBwd1 s1; s1.add = a.cH1; s1.ldadd = D; s1.cat = cat1; s1.ru = a.RU1 + tt * B * 2 * D; s1.cnd = a.C1 + tt * B * D; s1.X = D; s1.H = D;
                s1.dzc = a.DZC1 + tt * B * D; s1.dzg = a.DZG1 + tt * B * 2 * D; s1.keep = a.keep1;
                SkBwd1 s2;                                       // GRU 2 one step earlier; at t = 0 the same loads, results into the sink
                const size_t u = t > 0 ? tt - 1 : 0;
                s2.add = a.DY2 + u * B * D; s2.ldadd = D; s2.cat = a.CAT2 + u * B * 2 * D; s2.ru = a.RU2 + u * B * 2 * D; s2.cnd = a.C2 + u * B * D; s2.X = D; s2.H = D;
                s2.dzc = t > 0 ? a.DZC2 + u * B * D : a.sink; s2.dzg = t > 0 ? a.DZG2 + u * B * 2 * D : a.sink; s2.keep = t > 0 ? a.keep : a.sink;
                stage3(FLG<SKF_DRES | SKF_B1X | SKF_B1C | SKF_ADD>{}, D, D, a.Wg2T, a.DZG2 + tt * B * 2 * D, dy2, D, a.cH2, a.dy1, D, a.keep, a.dcatr, &s1, &s2);
            } else if (lw < B * V1D_ATT_C) {
                v1d_attn_bwd_chunk_body(a.keys, a.values, a.Q + tt * B * A, a.va, a.lens, a.ALIGN + tt * B * T, dctx_t, (long long)S * E, a.CATO + tt * B * (A + E) + A,
                                        A + E, T, A, E, a.DQ4 + tt * B * V1D_ATT_C * A, a.dkeys, a.dvpart, lw / V1D_ATT_C, lw % V1D_ATT_C, dsm, s_red);
            }
        }
        if (!(ok = hop())) break;
        // stage 2: GRU 1 stage 2 | query layer backward + stage 1 of the attention GRU
        if (act) {
            if (!grp) stage2(cat1, a.RU1 + tt * B * 2 * D, D, D, a.Wc1T, a.DZC1 + tt * B * D, a.DZG1 + tt * B * 2 * D, a.keep1, a.dcatr);
            else if (lw < (A + 15) / 16) {
                SkEpi e; e.res = a.dcato; e.ldr = A + E;
                e.b1x.add = a.cHA; e.b1x.ldadd = A; e.b1x.cat = cata; e.b1x.ru = a.RUa + tt * B * 2 * A; e.b1x.cnd = a.Ca + tt * B * A; e.b1x.X = E; e.b1x.H = A;
                e.b1x.dzc = a.DZCa + tt * B * A; e.b1x.dzg = a.DZGa + tt * B * 2 * A; e.b1x.keep = a.keepA;
                sk_body<MB, SK_BWD1, NWV, true, SKF_RES | SKF_ADD, 4>(a.DQ4 + tt * B * V1D_ATT_C * A, V1D_ATT_C * A, V1D_ATT_C * A, a.WqT, A, e, B, lw, (A + 15) / 16, red, A);
            }
        }
        if (!(ok = hop())) break;
        // stage 3: GRU 1 stage 3 -> d y0 of this step | attention GRU stage 2
        if (act) {
            if (!grp) stage3(FLG<SKF_DRES>{}, D, D, a.Wg1T, a.DZG1 + tt * B * 2 * D, a.dy1, D, a.cH1, dy0, D, a.keep1, a.dcatr, nullptr, nullptr);
            else stage2(cata, a.RUa + tt * B * 2 * A, E, A, a.WcaT, a.DZCa + tt * B * A, a.DZGa + tt * B * 2 * A, a.keepA, a.dcatrA);
        }
        if (!(ok = hop())) break;
        // stage 4: - | attention GRU stage 3 -> the carries d h_att, d ctx of the step before
        if (act && grp) stage3(FLG<0>{}, E, A, a.WgaT, a.DZGa + tt * B * 2 * A, nullptr, 0, a.cHA, a.cCTX, E, a.keepA, a.dcatrA, nullptr, nullptr);
        if (!(ok = hop())) break;
    }
    if (!ok && threadIdx.x == 0) __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct t2v1_dtrainer {
    t2v1_model* m = nullptr;
    int B = 0, T = 0, S = 0;
    t2_trainer ws;                       // workspace carrier of the shared helpers (split-K partials, column sums, offsets)
    char* slab = nullptr;
    bool have_forward = false, forward_only = false;
    const float* params = nullptr;
    const float* glob_counts = nullptr;  // dev [2] or null
    float drop_rate = 0.f;
    int* lens = nullptr;
    float *values, *keys, *target;
    // tapes ([S] or [S + 1] slices of [B][.])
    float *xin, *p0, *p1, *zgp, *zcp, *CATA, *CATAR, *RUa, *Ca, *Q, *ALIGN, *CATO, *CAT1, *CAT1R, *RU1, *C1, *CAT2, *CAT2R, *RU2, *C2, *Y2;
    float *zg, *zc, *mel_sb, *stop_sb, *mel_bt, *stop_bs, *loss6;
    // backward
    float *dmel_sb, *dstop_sb, *DY2, *DZG2, *DZC2, *DZG1, *DZC1, *DY0, *DQ, *DZGa, *DZCa, *DCTX, *dkeys, *dvpart, *dvalues, *DQ4, *en;
    float *cHA, *cCTX, *cH1, *cH2, *keep, *keep1, *keepA, *sink, *dcatr, *dcat, *dy1, *dy0, *dcato, *dha, *dP1, *dP0;
    float *WmT, *WsT, *Wg2T, *Wc2T, *Wg1T, *Wc1T, *WoT, *WqT, *WgaT, *WcaT, *WgpT, *WcpT, *Wp1T, *WmemT;
    // the two recurrence loops as hipGraphs (S x 15 / S x 17 launches each): every pointer in them is the trainer's own or lies in
    // the flat parameter buffer, so a graph stays valid as long as that buffer's address does
    unsigned* pbar = nullptr;           // hop counters + error word of the persistent recurrences
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    const float* gparams[2] = {nullptr, nullptr};
};

template <typename F>
static int dt_run_loop(t2v1_dtrainer* tr, int which, F&& body) {
    hipStream_t st = tr->m->base.ctx->stream;
    if (!tr->m->opt_graph) return body();
    if (tr->gexec[which] == nullptr || tr->gparams[which] != tr->params) {
        if (tr->gexec[which]) { (void)hipGraphExecDestroy(tr->gexec[which]); tr->gexec[which] = nullptr; }
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int r = body();
        const hipError_t e = hipStreamEndCapture(st, &g);
        if (r != T2_OK) { if (g) (void)hipGraphDestroy(g); return r; }
        if (e != hipSuccess) return fail(T2_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        const hipError_t e2 = hipGraphInstantiate(&tr->gexec[which], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) { tr->gexec[which] = nullptr; return fail(T2_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e2)); }
        tr->gparams[which] = tr->params;
    }
    HIP_TRY(hipGraphLaunch(tr->gexec[which], st));
    return T2_OK;
}

static const float* dt_w(t2v1_dtrainer* tr, const std::string& name, int* err) {
    auto it = tr->ws.offsets.find(name);
    if (it == tr->ws.offsets.end()) { *err = fail(T2_ESTATE, "no offset bound for '%s' (t2v1_decoder_trainer_bind)", name.c_str()); return nullptr; }
    return tr->params + it->second;
}
static inline dim3 dt_grid(long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); }

extern "C" int t2v1_decoder_trainer_destroy(t2v1_dtrainer* tr) {
    if (!tr) return T2_OK;
    (void)hipSetDevice(tr->m->base.ctx->device);
    (void)hipStreamSynchronize(tr->m->base.ctx->stream);
    for (int i = 0; i < 2; ++i) if (tr->gexec[i]) (void)hipGraphExecDestroy(tr->gexec[i]);
    if (tr->slab) (void)hipFree(tr->slab);
    if (tr->ws.cs_part) (void)hipFree(tr->ws.cs_part);
    if (tr->ws.wg_part) (void)hipFree(tr->ws.wg_part);
    if (tr->ws.bn_part) (void)hipFree(tr->ws.bn_part);
    delete tr;
    return T2_OK;
}

static int dt_create(t2v1_model* m, int B, int T_in, int S, bool forward_only, t2v1_dtrainer** out);
extern "C" int t2v1_decoder_trainer_create(t2v1_model* m, int B, int T_in, int S, t2v1_dtrainer** out) {
    return dt_create(m, B, T_in, S, false, out);
}
// forward_only: the engine's own instance behind teacher-forced decodes (v1_teacher_via_trainer): no backward buffers
static int dt_create(t2v1_model* m, int B, int T_in, int S, bool forward_only, t2v1_dtrainer** out) {
    T2_TRY(v1_ready(m));
    if (!out || S < 1) return fail(T2_EINVAL, "bad argument");
    T2_TRY(check_batch(B, T_in));
    const t2v1_dims& d = m->d;
    {   // the attention kernels keep a row's query, scores and the 16 waves' partial sums in LDS (64 KB without opting in to more)
        const size_t NW = V1D_ATT_THREADS / 64, A = d.attention_out_units, E = d.cbhg_out_units;
        const size_t lds = std::max((2 * A + T_in + NW * E) * 4, (E + T_in + 2 * NW * A) * 4);
        if (lds > 64 * 1024) return fail(T2_EINVAL, "T_in = %d with attention width %d / memory width %d needs %zu bytes of LDS per attention workgroup (limit 65536)",
                                         T_in, (int)A, (int)E, lds);
    }
    HIP_TRY(hipSetDevice(m->base.ctx->device));
    t2v1_dtrainer* tr = new t2v1_dtrainer();
    tr->m = m; tr->B = B; tr->T = T_in; tr->S = S;
    tr->ws.m = &m->base;
    const size_t P0 = d.dec_prenet_units[0], P1 = d.dec_prenet_units[1], A = d.attention_out_units, E = d.cbhg_out_units, D = d.decoder_out_units;
    const size_t M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, rM = r * M, SB = (size_t)S * B, BT = (size_t)B * T_in;
    Carver cv;
    std::vector<std::pair<float**, size_t>> plan;
    auto want = [&](float** p, size_t floats) { plan.push_back({p, cv.take(floats * 4)}); };
    want(&tr->values, BT * E); want(&tr->keys, BT * A); want(&tr->target, (size_t)B * S * rM);
    want(&tr->xin, SB * nf * M); want(&tr->p0, SB * P0); want(&tr->p1, SB * P1); want(&tr->zgp, SB * 2 * A); want(&tr->zcp, SB * A);
    want(&tr->CATA, (SB + B) * (E + A)); want(&tr->CATAR, SB * (E + A)); want(&tr->RUa, SB * 2 * A); want(&tr->Ca, SB * A); want(&tr->Q, SB * A);
    want(&tr->ALIGN, SB * T_in); want(&tr->CATO, SB * (A + E));
    want(&tr->CAT1, (SB + B) * 2 * D); want(&tr->CAT1R, SB * 2 * D); want(&tr->RU1, SB * 2 * D); want(&tr->C1, SB * D);
    want(&tr->CAT2, (SB + B) * 2 * D); want(&tr->CAT2R, SB * 2 * D); want(&tr->RU2, SB * 2 * D); want(&tr->C2, SB * D); want(&tr->Y2, SB * D);
    const size_t Hmax = std::max(A, D), Wmax = std::max(E + A, 2 * D);
    want(&tr->zg, (size_t)B * 2 * Hmax); want(&tr->zc, (size_t)B * Hmax);
    want(&tr->en, BT);
    want(&tr->mel_sb, SB * rM); want(&tr->stop_sb, SB); want(&tr->mel_bt, SB * rM); want(&tr->stop_bs, SB); want(&tr->loss6, 8);
    tr->forward_only = forward_only;
    if (!forward_only) {
    want(&tr->dmel_sb, SB * rM); want(&tr->dstop_sb, SB); want(&tr->DY2, SB * D); want(&tr->DZG2, SB * 2 * D); want(&tr->DZC2, SB * D);
    want(&tr->DZG1, SB * 2 * D); want(&tr->DZC1, SB * D); want(&tr->DY0, SB * D); want(&tr->DQ, SB * A); want(&tr->DZGa, SB * 2 * A); want(&tr->DZCa, SB * A);
    want(&tr->DCTX, SB * E); want(&tr->dkeys, BT * A); want(&tr->dvpart, (size_t)B * V1D_ATT_C * A); want(&tr->DQ4, SB * V1D_ATT_C * A); want(&tr->dvalues, BT * E);
    want(&tr->cHA, (size_t)B * A); want(&tr->cCTX, (size_t)B * E); want(&tr->cH1, (size_t)B * D); want(&tr->cH2, (size_t)B * D);
    want(&tr->keep, (size_t)B * Hmax); want(&tr->keep1, (size_t)B * Hmax); want(&tr->keepA, (size_t)B * Hmax); want(&tr->sink, (size_t)B * 2 * Hmax); want(&tr->dcatr, (size_t)B * Wmax); want(&tr->dcat, (size_t)B * Wmax);
    want(&tr->dy1, (size_t)B * D); want(&tr->dy0, (size_t)B * D); want(&tr->dcato, (size_t)B * (A + E)); want(&tr->dha, (size_t)B * A);
    want(&tr->dP1, SB * P1); want(&tr->dP0, SB * P0);
    want(&tr->WmT, rM * D); want(&tr->WsT, D); want(&tr->Wg2T, 2 * D * 2 * D); want(&tr->Wc2T, D * 2 * D); want(&tr->Wg1T, 2 * D * 2 * D); want(&tr->Wc1T, D * 2 * D);
    want(&tr->WoT, D * (A + E)); want(&tr->WqT, A * A); want(&tr->WgaT, 2 * A * (E + A)); want(&tr->WcaT, A * (E + A)); want(&tr->WgpT, 2 * A * P1);
    want(&tr->WcpT, A * P1); want(&tr->Wp1T, P1 * P0); want(&tr->WmemT, A * E);
    }
    const size_t o_lens = cv.take((size_t)B * 4), o_bar = cv.take((size_t)(pz::NSHARD * 32 + 64) * 4);
    hipError_t e = hipMalloc((void**)&tr->slab, cv.off);
    if (e != hipSuccess) { delete tr; return fail(T2_ENOMEM, "hipMalloc(V1 decoder trainer, %zu bytes) failed: %s", cv.off, hipGetErrorString(e)); }
    for (auto& pr : plan) *pr.first = (float*)(tr->slab + pr.second);
    tr->lens = (int*)(tr->slab + o_lens);
    tr->pbar = (unsigned*)(tr->slab + o_bar);
    *out = tr;
    return T2_OK;
}

extern "C" int t2v1_decoder_trainer_bind(t2v1_dtrainer* tr, const char* name, int64_t offset) {
    if (!tr || !name || offset < 0 || (offset & 3) != 0) return fail(T2_EINVAL, "bad argument (offsets are multiples of 4 floats)");
    tr->ws.offsets[name] = offset;
    return T2_OK;
}
// counts dev [2] = non-zero loss weights (spec x num_mels, stop) of a larger data-parallel batch, or null
extern "C" int t2v1_decoder_trainer_set_loss_counts(t2v1_dtrainer* tr, const float* counts) {
    if (!tr) return fail(T2_EINVAL, "null trainer");
    tr->glob_counts = counts;
    return T2_OK;
}

static inline bool dt_skinny_ok(t2v1_dtrainer* tr, const float* x, int ldx, int K, int rows) {
    return rows <= 64 && K % 16 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && tr->m->opt_skinny;
}
// The recurrences as one launch each (v1d_*_persist_kernel): B <= 48 (LDS of the 16-wave reduction), every per-step product on the
// skinny path, a whole chip (the grid must be resident: one 1024-thread workgroup per CU, G <= 64 of them)
static size_t dt_persist_lds(t2v1_dtrainer* tr) {
    const t2v1_dims& d = tr->m->d;
    const size_t NW = V1D_ATT_THREADS / 64, A = d.attention_out_units, E = d.cbhg_out_units, MB = ceil_div(tr->B, 16);
    return std::max({NW * MB * 64 * 16, (2 * A + tr->T + NW * E) * 4, (E + tr->T + 2 * NW * A) * 4});
}
static int dt_persist_nd(t2v1_dtrainer* tr) {                 // workgroups of the decoder chain (a multiple of the counter shards)
    const t2v1_dims& d = tr->m->d;
    return ceil_div(ceil_div(2 * d.decoder_out_units, 16), pz::NSHARD) * pz::NSHARD;
}
static int dt_persist_grid(t2v1_dtrainer* tr) {
    const t2v1_dims& d = tr->m->d;
    const int A = d.attention_out_units, E = d.cbhg_out_units;
    return dt_persist_nd(tr) + ceil_div(std::max({ceil_div(2 * A, 16), ceil_div(A + E, 16), tr->B * V1D_ATT_C}), pz::NSHARD) * pz::NSHARD;
}
static bool dt_persist_ok(t2v1_dtrainer* tr) {
    t2v1_model* m = tr->m;
    const t2v1_dims& d = m->d;
    const int A = d.attention_out_units, E = d.cbhg_out_units, D = d.decoder_out_units, B = tr->B;
    return m->opt_train_persist && m->opt_skinny && m->base.persist_holdoff == 0 && B <= 48 && A % 16 == 0 && E % 16 == 0 && D % 16 == 0 && A <= 256 && E <= 256 &&
           dt_persist_lds(tr) + 256 <= 64 * 1024 && m->base.n_cu >= dt_persist_grid(tr);
}
// launches kernel `which` (0 forward, 1 backward) of the persistent pair; *timed_out: the caller re-runs the loop launch by launch
template <typename ARGS>
static int dt_persist_launch(t2v1_dtrainer* tr, int which, ARGS& a, bool* timed_out) {
    t2_model* b = &tr->m->base;
    hipStream_t st = b->ctx->stream;
    HIP_TRY(hipMemsetAsync(tr->pbar, 0, (size_t)(pz::NSHARD * 32 + 64) * 4, st));
    a.bar = tr->pbar; a.err = (int*)(tr->pbar + pz::NSHARD * 32);
    const dim3 grid(dt_persist_grid(tr)), blk(V1D_ATT_THREADS);
    const size_t lds = dt_persist_lds(tr);
    const int MB = ceil_div(tr->B, 16);
    {
        XwgGate gate;
        T2_TRY(gate.begin(b->ctx));
        if constexpr (std::is_same<ARGS, V1PFwd>::value) {
            if (MB == 1) hipLaunchKernelGGL(v1d_fwd_persist_kernel<1>, grid, blk, lds, st, a);
            else if (MB == 2) hipLaunchKernelGGL(v1d_fwd_persist_kernel<2>, grid, blk, lds, st, a);
            else hipLaunchKernelGGL(v1d_fwd_persist_kernel<3>, grid, blk, lds, st, a);
        } else {
            if (MB == 1) hipLaunchKernelGGL(v1d_bwd_persist_kernel<1>, grid, blk, lds, st, a);
            else if (MB == 2) hipLaunchKernelGGL(v1d_bwd_persist_kernel<2>, grid, blk, lds, st, a);
            else hipLaunchKernelGGL(v1d_bwd_persist_kernel<3>, grid, blk, lds, st, a);
        }
        HIP_TRY(hipGetLastError());
    }
    (void)which;
    T2_TRY(inject_timeout(b, a.err));
    ++tr->m->n_train_persist;
    int h_err = 0;
    HIP_TRY(hipMemcpyAsync(&h_err, a.err, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *timed_out = h_err != 0;
    if (h_err) note_persist_timeout(b);
    return T2_OK;
}
template <int EPI>
static void dt_skinny(t2v1_dtrainer* tr, const float* x, int ldx, int K, const float* w, int N, const SkEpi& e, int rows) {
    hipStream_t st = tr->m->base.ctx->stream;
    const dim3 grid(ceil_div(N, 16)), blk(64 * V1D_SK_WAVES);
    switch (ceil_div(rows, 16)) {
        case 1: hipLaunchKernelGGL((v1d_skinny_gemm_kernel<1, EPI>), grid, blk, 0, st, x, ldx, K, w, N, e, rows); break;
        case 2: hipLaunchKernelGGL((v1d_skinny_gemm_kernel<2, EPI>), grid, blk, 0, st, x, ldx, K, w, N, e, rows); break;
        case 3: hipLaunchKernelGGL((v1d_skinny_gemm_kernel<3, EPI>), grid, blk, 0, st, x, ldx, K, w, N, e, rows); break;
        default: hipLaunchKernelGGL((v1d_skinny_gemm_kernel<4, EPI>), grid, blk, 0, st, x, ldx, K, w, N, e, rows); break;
    }
}
// y [B][N] (row stride ldy) = x [B][K] (row stride ldx) . w [K][N] (+ bias) (+ res): the small per-step products
static int dt_gemm(t2v1_dtrainer* tr, const float* x, int ldx, int K, const float* w, int N, const float* bias, float* y, int ldy, const float* res, int ldr,
                   int rows) {
    if (dt_skinny_ok(tr, x, ldx, K, rows)) {
        SkEpi e; e.bias = bias; e.res = res; e.ldr = ldr; e.y = y; e.ldy = ldy;
        dt_skinny<SK_PLAIN>(tr, x, ldx, K, w, N, e, rows);
        return T2_OK;
    }
    ConvLayer L{};
    L.w = const_cast<float*>(w); L.shift = const_cast<float*>(bias); L.k = 1; L.cin = K; L.cout = N;
    return launch_conv(&tr->m->base, L, x, ldx, y, ldy, rows, rows, 0, res, ldr);
}
// GRUCell forward of one step on the operand cat = [x | h]: gates product (+ bias, + the batched share zgp) ->
// [r | u] tape, catr = [x | r h]; candidate product -> c tape, h' (and the residual y = x + h') to their destinations.  Two launches
// on the skinny path, four otherwise.
static int dt_gru_fwd(t2v1_dtrainer* tr, const float* cat, float* catr, int X, int H, const float* Wg, const float* bg, const float* zgp, const float* Wc,
                      const float* bc, const float* zcp, float* ru, float* cnd, V1Dst h0, V1Dst h1, V1Dst y0, V1Dst y1) {
    hipStream_t st = tr->m->base.ctx->stream;
    const int B = tr->B, Wd = X + H;
    if (dt_skinny_ok(tr, cat, Wd, Wd, B) && dt_skinny_ok(tr, catr, Wd, Wd, B)) {
        SkEpi g; g.bias = bg; g.res = zgp; g.ldr = 2 * H; g.cat = cat; g.X = X; g.H = H; g.ru = ru; g.catr = catr;
        dt_skinny<SK_GATE>(tr, cat, Wd, Wd, Wg, 2 * H, g, B);
        SkEpi c; c.bias = bc; c.res = zcp; c.ldr = H; c.cat = cat; c.X = X; c.H = H; c.ru = ru; c.cnd = cnd; c.h0 = h0; c.h1 = h1; c.y0 = y0; c.y1 = y1;
        dt_skinny<SK_BLEND>(tr, catr, Wd, Wd, Wc, H, c, B);
        return T2_OK;
    }
    T2_TRY(dt_gemm(tr, cat, Wd, Wd, Wg, 2 * H, bg, tr->zg, 2 * H, zgp, 2 * H, B));
    hipLaunchKernelGGL(v1d_gate_kernel, dt_grid((long long)B * (Wd + H)), dim3(256), 0, st, tr->zg, cat, X, H, ru, catr, B);
    T2_TRY(dt_gemm(tr, catr, Wd, Wd, Wc, H, bc, tr->zc, H, zcp, H, B));
    hipLaunchKernelGGL(v1d_blend_kernel, dt_grid((long long)B * H), dim3(256), 0, st, tr->zc, ru, cat, X, H, cnd, h0, h1, y0, y1, B);
    return T2_OK;
}
static void dt_transpose(t2v1_dtrainer* tr, const float* src, float* dst, int R, int C) {      // src [R][C] -> dst [C][R]
    hipLaunchKernelGGL(transpose_kernel, dt_grid((long long)R * C), dim3(256), 0, tr->m->base.ctx->stream, dst, R, src, C, C, R);
}

#ifdef V1D_STAMPS
extern "C" int t2v1_debug_stamps(unsigned long long* out, int reset) {     // diagnostic builds only: [2][2][16] + [8] tick sums (10 ns)
    if (out) {
        HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_v1d_stamps), sizeof(unsigned long long) * 64));
        HIP_TRY(hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(g_v1d_body), sizeof(unsigned long long) * 8));
    }
    if (reset) {
        unsigned long long z[64] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_v1d_stamps), z, sizeof(z)));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_v1d_body), z, sizeof(unsigned long long) * 8));
    }
    return T2_OK;
}
#endif
extern "C" int t2v1_model_get_stat(t2v1_model* m, const char* name, double* out) {
    if (!m || !name || !out) return fail(T2_EINVAL, "null argument");
    if (!strcmp(name, "train_persist_launches")) { *out = (double)m->n_train_persist; return T2_OK; }
    if (!strcmp(name, "persist_timeouts")) { *out = (double)m->base.n_persist_timeouts; return T2_OK; }
    return fail(T2_EINVAL, "unknown stat %s", name);
}

// memory dev [B][T_in][E]; lens host [B] or null; target dev [B][S*r][M] (labels.mel); mel_out dev [B][S*r][M], stop_out dev [B][S],
// align_out dev [S][B][T_in] (alignment_history.stack())
extern "C" int t2v1_train_decoder_forward(t2v1_dtrainer* tr, const float* flat_params, const float* memory, const int64_t* lens, const float* target,
                                          float drop_rate, uint32_t seed, float* mel_out, float* stop_out, float* align_out) {
    if (!tr || !flat_params || !memory || !target || !mel_out || !stop_out || !align_out) return fail(T2_EINVAL, "null argument");
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return fail(T2_EINVAL, "drop_rate %g outside [0, 1)", (double)drop_rate);
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const t2v1_dims& d = m->d;
    const int B = tr->B, T = tr->T, S = tr->S, P0 = d.dec_prenet_units[0], P1 = d.dec_prenet_units[1], A = d.attention_out_units, E = d.cbhg_out_units,
              D = d.decoder_out_units, M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, rM = r * M, SB = S * B;
    tr->params = flat_params; tr->have_forward = false; tr->drop_rate = drop_rate;
    T2_TRY(v1_lens_to_dev(m, lens, B, T, 1, tr->lens));
    int err = T2_OK;
    auto W = [&](const std::string& n) { return dt_w(tr, n, &err); };
    const std::string pd = "v1/decoder/";
    const float *Wp0 = W(pd + "prenet_0/dense/kernel"), *bp0 = W(pd + "prenet_0/dense/bias"), *Wp1 = W(pd + "prenet_1/dense/kernel"), *bp1 = W(pd + "prenet_1/dense/bias");
    const float *Wga = W(pd + "attention_gru/gates/kernel"), *bga = W(pd + "attention_gru/gates/bias"), *Wca = W(pd + "attention_gru/candidate/kernel"),
                *bca = W(pd + "attention_gru/candidate/bias");
    const float *Wmem = W("v1/attention/memory_layer/kernel"), *Wq = W("v1/attention/query_layer/kernel"), *va = W("v1/attention/attention_v");
    const float *Wo = W(pd + "output_projection/kernel"), *bo = W(pd + "output_projection/bias");
    const float *Wg1 = W(pd + "gru_1/gates/kernel"), *bg1 = W(pd + "gru_1/gates/bias"), *Wc1 = W(pd + "gru_1/candidate/kernel"), *bc1 = W(pd + "gru_1/candidate/bias");
    const float *Wg2 = W(pd + "gru_2/gates/kernel"), *bg2 = W(pd + "gru_2/gates/bias"), *Wc2 = W(pd + "gru_2/candidate/kernel"), *bc2 = W(pd + "gru_2/candidate/bias");
    const float *Wm = W(pd + "out_projection/kernel"), *bm = W(pd + "out_projection/bias"), *Wst = W(pd + "stop_token_projection/kernel"),
                *bst = W(pd + "stop_token_projection/bias");
    if (err != T2_OK) return err;
    // attention memory: values = memory * mask, keys = values . W_mem (BahdanauAttention constructor)
    hipLaunchKernelGGL(mask_memory_kernel, dt_grid((long long)B * T * E), dim3(256), 0, st, memory, tr->lens, tr->values, B, T, E);
    T2_TRY(dt_gemm(tr, tr->values, E, E, Wmem, A, nullptr, tr->keys, A, nullptr, 0, B * T));
    HIP_TRY(hipMemcpyAsync(tr->target, target, (size_t)B * S * rM * 4, hipMemcpyDeviceToDevice, st));
    // PreNets of all steps, and their share of the attention GRU's pre-activations
    hipLaunchKernelGGL(v1d_feed_kernel, dt_grid((long long)SB * nf * M), dim3(256), 0, st, tr->target, tr->xin, B, S, r, nf, M);
    {
        ConvLayer L{};
        L.w = const_cast<float*>(Wp0); L.shift = const_cast<float*>(bp0); L.k = 1; L.cin = nf * M; L.cout = P0;
        T2_TRY(launch_conv(b, L, tr->xin, nf * M, tr->p0, P0, SB, SB, 1, nullptr, 0));
        if (drop_rate > 0.f) hipLaunchKernelGGL(v1d_dropout_kernel, dt_grid((long long)SB * P0), dim3(256), 0, st, tr->p0, drop_rate, seed, T2_STREAM_V1_DEC_PRENET, B, (long long)SB, P0);
        L.w = const_cast<float*>(Wp1); L.shift = const_cast<float*>(bp1); L.cin = P0; L.cout = P1;
        T2_TRY(launch_conv(b, L, tr->p0, P0, tr->p1, P1, SB, SB, 1, nullptr, 0));
        if (drop_rate > 0.f) hipLaunchKernelGGL(v1d_dropout_kernel, dt_grid((long long)SB * P1), dim3(256), 0, st, tr->p1, drop_rate, seed, T2_STREAM_V1_DEC_PRENET + 1, B, (long long)SB, P1);
    }
    T2_TRY(dt_gemm(tr, tr->p1, P1, P1, Wga, 2 * A, bga, tr->zgp, 2 * A, nullptr, 0, SB));
    T2_TRY(dt_gemm(tr, tr->p1, P1, P1, Wca, A, bca, tr->zcp, A, nullptr, 0, SB));
    // zero states: [ctx | h_att] of step 0, the h halves of [y | h] of step 0
    HIP_TRY(hipMemsetAsync(tr->CATA, 0, (size_t)B * (E + A) * 4, st));
    HIP_TRY(hipMemsetAsync(tr->CAT1, 0, (size_t)B * 2 * D * 4, st));
    HIP_TRY(hipMemsetAsync(tr->CAT2, 0, (size_t)B * 2 * D * 4, st));
    const int XA = E + A;
    const size_t attn_lds = (size_t)(2 * A + T + (V1D_ATT_THREADS / 64) * E) * 4;
    bool looped = false;
    holdoff_tick(b);
    if (dt_persist_ok(tr)) {
        V1PFwd a{};
        a.B = B; a.T = T; a.S = S; a.A = A; a.E = E; a.D = D;
        a.Wga = Wga + (size_t)P1 * 2 * A; a.Wca = Wca + (size_t)P1 * A; a.Wq = Wq; a.va = va; a.Wo = Wo; a.bo = bo;
        a.Wg1 = Wg1; a.bg1 = bg1; a.Wc1 = Wc1; a.bc1 = bc1; a.Wg2 = Wg2; a.bg2 = bg2; a.Wc2 = Wc2; a.bc2 = bc2;
        a.zgp = tr->zgp; a.zcp = tr->zcp; a.keys = tr->keys; a.values = tr->values; a.lens = tr->lens;
        a.CATA = tr->CATA; a.CATAR = tr->CATAR; a.RUa = tr->RUa; a.Ca = tr->Ca; a.Q = tr->Q; a.ALIGN = tr->ALIGN; a.CATO = tr->CATO;
        a.CAT1 = tr->CAT1; a.CAT1R = tr->CAT1R; a.RU1 = tr->RU1; a.C1 = tr->C1; a.CAT2 = tr->CAT2; a.CAT2R = tr->CAT2R; a.RU2 = tr->RU2; a.C2 = tr->C2; a.Y2 = tr->Y2;
        a.en = tr->en; a.nd = dt_persist_nd(tr);
        bool timed_out = false;
        T2_TRY(dt_persist_launch(tr, 0, a, &timed_out));
        looped = !timed_out;          // every word of the tapes is rewritten by the launch-per-stage loop: nothing to reset
    }
    if (!looped) T2_TRY(dt_run_loop(tr, 0, [&]() -> int {
    for (int t = 0; t < S; ++t) {
        float *cata = tr->CATA + (size_t)t * B * XA, *cata_n = cata + (size_t)B * XA, *catar = tr->CATAR + (size_t)t * B * XA;
        float *cato = tr->CATO + (size_t)t * B * (A + E);
        float *cat1 = tr->CAT1 + (size_t)t * B * 2 * D, *cat1_n = cat1 + (size_t)B * 2 * D, *cat1r = tr->CAT1R + (size_t)t * B * 2 * D;
        float *cat2 = tr->CAT2 + (size_t)t * B * 2 * D, *cat2_n = cat2 + (size_t)B * 2 * D, *cat2r = tr->CAT2R + (size_t)t * B * 2 * D;
        // attention GRU on [p1_t | ctx_{t-1}] (AttentionWrapper, App. A6): the p1 rows of the kernels were applied above
        T2_TRY(dt_gru_fwd(tr, cata, catar, E, A, Wga + (size_t)P1 * 2 * A, nullptr, tr->zgp + (size_t)t * B * 2 * A, Wca + (size_t)P1 * A, nullptr,
                          tr->zcp + (size_t)t * B * A, tr->RUa + (size_t)t * B * 2 * A, tr->Ca + (size_t)t * B * A, V1Dst{cato, A + E}, V1Dst{cata_n + E, XA},
                          V1Dst{nullptr, 0}, V1Dst{nullptr, 0}));
        // Bahdanau attention
        float* q = tr->Q + (size_t)t * B * A;
        T2_TRY(dt_gemm(tr, cato, A + E, A, Wq, A, nullptr, q, A, nullptr, 0, B));
        hipLaunchKernelGGL(v1d_attn_fwd_kernel, dim3(B), dim3(V1D_ATT_THREADS), attn_lds, st, tr->keys, tr->values, q, va, tr->lens, T, A, E, tr->ALIGN + (size_t)t * B * T,
                           V1Dst{cato + A, A + E}, V1Dst{cata_n, XA});
        // OutputProjectionWrapper on [h_att | ctx] -> y0 = the x half of GRU 1's operand
        T2_TRY(dt_gemm(tr, cato, A + E, A + E, Wo, D, bo, cat1, 2 * D, nullptr, 0, B));
        // ResidualWrapper(GRUCell) x 2
        T2_TRY(dt_gru_fwd(tr, cat1, cat1r, D, D, Wg1, bg1, nullptr, Wc1, bc1, nullptr, tr->RU1 + (size_t)t * B * 2 * D, tr->C1 + (size_t)t * B * D,
                          V1Dst{cat1_n + D, 2 * D}, V1Dst{nullptr, 0}, V1Dst{cat2, 2 * D}, V1Dst{nullptr, 0}));
        T2_TRY(dt_gru_fwd(tr, cat2, cat2r, D, D, Wg2, bg2, nullptr, Wc2, bc2, nullptr, tr->RU2 + (size_t)t * B * 2 * D, tr->C2 + (size_t)t * B * D,
                          V1Dst{cat2_n + D, 2 * D}, V1Dst{nullptr, 0}, V1Dst{tr->Y2 + (size_t)t * B * D, D}, V1Dst{nullptr, 0}));
    }
    return T2_OK;
    }));
    // mel / stop projections of all steps (OutputAndStopTokenWrapper, rnn_wrappers.py:105-109)
    T2_TRY(dt_gemm(tr, tr->Y2, D, D, Wm, rM, bm, tr->mel_sb, rM, nullptr, 0, SB));
    T2_TRY(dt_gemm(tr, tr->Y2, D, D, Wst, 1, bst, tr->stop_sb, 1, nullptr, 0, SB));
    hipLaunchKernelGGL(v1d_sb_to_bt_kernel, dt_grid((long long)SB * rM), dim3(256), 0, st, tr->mel_sb, tr->mel_bt, B, S, rM);
    hipLaunchKernelGGL(v1d_stop_to_bs_kernel, dt_grid((long long)SB), dim3(256), 0, st, tr->stop_sb, tr->stop_bs, B, S);
    HIP_TRY(hipMemcpyAsync(mel_out, tr->mel_bt, (size_t)SB * rM * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop_out, tr->stop_bs, (size_t)SB * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(align_out, tr->ALIGN, (size_t)SB * T * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    tr->have_forward = true;
    return T2_OK;
}

// one GRU cell backward (three elementwise stages around two GEMMs); dz tapes written for the weight gradients
static int dt_gru_bwd(t2v1_dtrainer* tr, const float* d0, int ld0, const float* d1, int ld1, const float* cat, const float* ru, const float* cnd, int X, int H,
                      const float* WcT, const float* WgT, float* dzc, float* dzg, const float* dres, int ldres, float* carry, float* dx, int lddx,
                      float* keep = nullptr, bool stage1_done = false, const SkBwd1* b1x = nullptr, const SkBwd1* b1c = nullptr) {
    hipStream_t st = tr->m->base.ctx->stream;
    const int B = tr->B, Wd = X + H;
    if (!keep) keep = tr->keep;
    if (!stage1_done) hipLaunchKernelGGL(v1d_bwd1_kernel, dt_grid((long long)B * H), dim3(256), 0, st, d0, ld0, d1, ld1, cat, ru, cnd, X, H, dzc, dzg, keep, B);
    if (dt_skinny_ok(tr, dzc, H, H, B) && dt_skinny_ok(tr, dzg, 2 * H, 2 * H, B)) {          // stages 2 and 3 in the products' epilogues
        SkEpi e2; e2.y = tr->dcatr; e2.ldy = Wd; e2.cat = cat; e2.X = X; e2.H = H; e2.ru = const_cast<float*>(ru); e2.dzg = dzg; e2.keep = keep;
        dt_skinny<SK_BWD2>(tr, dzc, H, H, WcT, Wd, e2, B);
        SkEpi e3; e3.X = X; e3.H = H; e3.dcatr = tr->dcatr; e3.dres = dres; e3.ldres = ldres; e3.keep = keep; e3.carry = carry; e3.dx = dx; e3.lddx = lddx;
        if (b1x) e3.b1x = *b1x;
        if (b1c) e3.b1c = *b1c;
        dt_skinny<SK_BWD3>(tr, dzg, 2 * H, 2 * H, WgT, Wd, e3, B);
        return T2_OK;
    }
    if (stage1_done || b1x || b1c) return fail(T2_ESTATE, "internal: fused GRU backward stages outside the skinny path");
    T2_TRY(dt_gemm(tr, dzc, H, H, WcT, Wd, nullptr, tr->dcatr, Wd, nullptr, 0, B));
    hipLaunchKernelGGL(v1d_bwd2_kernel, dt_grid((long long)B * H), dim3(256), 0, st, tr->dcatr, cat, ru, X, H, dzg, keep, B);
    T2_TRY(dt_gemm(tr, dzg, 2 * H, 2 * H, WgT, Wd, nullptr, tr->dcat, Wd, nullptr, 0, B));
    hipLaunchKernelGGL(v1d_bwd3_kernel, dt_grid((long long)B * Wd), dim3(256), 0, st, tr->dcatr, tr->dcat, keep, dres, ldres, X, H, carry, dx, lddx, B);
    return T2_OK;
}

// done dev [B][S], spec_mask dev [B][S*r], bin_mask dev [B][S].  Writes every v1/decoder/ and v1/attention/ gradient into flat_grads,
// d loss / d memory into dmemory (dev [B][T_in][E]) and {mel_loss, done_loss, loss} into loss_out (dev [3]).
extern "C" int t2v1_train_decoder_backward(t2v1_dtrainer* tr, const float* done, const float* spec_mask, const float* bin_mask, float* flat_grads,
                                           float* dmemory, float* loss_out) {
    if (!tr || !done || !spec_mask || !bin_mask || !flat_grads || !dmemory || !loss_out) return fail(T2_EINVAL, "null argument");
    if (!tr->have_forward) return fail(T2_ESTATE, "t2v1_train_decoder_forward has not run");
    if (tr->forward_only) return fail(T2_ESTATE, "internal: a forward-only decoder trainer has no backward pass");
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const t2v1_dims& d = m->d;
    const int B = tr->B, T = tr->T, S = tr->S, P0 = d.dec_prenet_units[0], P1 = d.dec_prenet_units[1], A = d.attention_out_units, E = d.cbhg_out_units,
              D = d.decoder_out_units, M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, rM = r * M, SB = S * B, XA = E + A;
    int err = T2_OK;
    auto W = [&](const std::string& n) { return dt_w(tr, n, &err); };
    auto G = [&](const std::string& n) -> float* {
        auto it = tr->ws.offsets.find(n);
        if (it == tr->ws.offsets.end()) { err = fail(T2_ESTATE, "no offset bound for '%s'", n.c_str()); return nullptr; }
        return flat_grads + it->second;
    };
    const std::string pd = "v1/decoder/";
    const float *Wp1 = W(pd + "prenet_1/dense/kernel");
    const float *Wga = W(pd + "attention_gru/gates/kernel"), *Wca = W(pd + "attention_gru/candidate/kernel");
    const float *Wmem = W("v1/attention/memory_layer/kernel"), *Wq = W("v1/attention/query_layer/kernel"), *va = W("v1/attention/attention_v");
    const float *Wo = W(pd + "output_projection/kernel"), *Wg1 = W(pd + "gru_1/gates/kernel"), *Wc1 = W(pd + "gru_1/candidate/kernel");
    const float *Wg2 = W(pd + "gru_2/gates/kernel"), *Wc2 = W(pd + "gru_2/candidate/kernel");
    const float *Wm = W(pd + "out_projection/kernel"), *Wst = W(pd + "stop_token_projection/kernel");
    if (err != T2_OK) return err;
    // losses (models.py:58-61) and their gradients
    T2_TRY(t2_spec_loss(b->ctx, tr->mel_bt, tr->target, spec_mask, B, S * r, M, tr->loss6));
    T2_TRY(t2_binary_loss(b->ctx, tr->stop_bs, done, bin_mask, B, S, tr->loss6 + 3));
    const float *n1 = tr->glob_counts ? tr->glob_counts : tr->loss6 + 1, *n2 = tr->glob_counts ? tr->glob_counts + 1 : tr->loss6 + 4;
    hipLaunchKernelGGL(v1d_loss_grad_kernel, dt_grid((long long)SB * (rM + 1)), dim3(256), 0, st, tr->mel_sb, tr->stop_sb, tr->target, done, spec_mask, bin_mask, n1, n2,
                       B, S, r, M, tr->dmel_sb, tr->dstop_sb);
    hipLaunchKernelGGL(v1d_loss_out_kernel, dim3(1), dim3(64), 0, st, tr->loss6, loss_out);
    // transposed operands of the data-gradient products
    dt_transpose(tr, Wm, tr->WmT, D, rM); dt_transpose(tr, Wst, tr->WsT, D, 1);
    dt_transpose(tr, Wg2, tr->Wg2T, 2 * D, 2 * D); dt_transpose(tr, Wc2, tr->Wc2T, 2 * D, D);
    dt_transpose(tr, Wg1, tr->Wg1T, 2 * D, 2 * D); dt_transpose(tr, Wc1, tr->Wc1T, 2 * D, D);
    dt_transpose(tr, Wo, tr->WoT, A + E, D); dt_transpose(tr, Wq, tr->WqT, A, A);
    dt_transpose(tr, Wga + (size_t)P1 * 2 * A, tr->WgaT, XA, 2 * A); dt_transpose(tr, Wca + (size_t)P1 * A, tr->WcaT, XA, A);
    dt_transpose(tr, Wga, tr->WgpT, P1, 2 * A); dt_transpose(tr, Wca, tr->WcpT, P1, A);
    dt_transpose(tr, Wp1, tr->Wp1T, P0, P1); dt_transpose(tr, Wmem, tr->WmemT, E, A);
    // d y2 of all steps: the projections do not depend on the recurrence
    T2_TRY(dt_gemm(tr, tr->dmel_sb, rM, rM, tr->WmT, D, nullptr, tr->DY2, D, nullptr, 0, SB));
    T2_TRY(dt_gemm(tr, tr->dstop_sb, 1, 1, tr->WsT, D, nullptr, tr->DY2, D, tr->DY2, D, SB));
    auto zero_carries = [&]() -> int {
        HIP_TRY(hipMemsetAsync(tr->cHA, 0, (size_t)B * A * 4, st)); HIP_TRY(hipMemsetAsync(tr->cCTX, 0, (size_t)B * E * 4, st));
        HIP_TRY(hipMemsetAsync(tr->cH1, 0, (size_t)B * D * 4, st)); HIP_TRY(hipMemsetAsync(tr->cH2, 0, (size_t)B * D * 4, st));
        HIP_TRY(hipMemsetAsync(tr->dkeys, 0, (size_t)B * T * A * 4, st)); HIP_TRY(hipMemsetAsync(tr->dvpart, 0, (size_t)B * V1D_ATT_C * A * 4, st));
        return T2_OK;
    };
    T2_TRY(zero_carries());
    const size_t attn_lds = (size_t)(E + T + 2 * (V1D_ATT_THREADS / 64) * A) * 4;
    // every per-step product on the skinny path (B <= 64, widths multiples of 16): stage 1 of the cells fused as well
    const bool fused = dt_skinny_ok(tr, tr->DZC2, D, D, B) && dt_skinny_ok(tr, tr->DZG2, 2 * D, 2 * D, B) && dt_skinny_ok(tr, tr->DQ, A, A, B) &&
                       dt_skinny_ok(tr, tr->DZCa, A, A, B) && dt_skinny_ok(tr, tr->DZGa, 2 * A, 2 * A, B) && (B * D) % 4 == 0 && (B * A) % 4 == 0;
    bool looped = false;
    if (fused && dt_persist_ok(tr)) {
        const size_t u = (size_t)(S - 1);            // stage 1 of GRU 2 at the last step; every other stage 1 rides in an epilogue
        hipLaunchKernelGGL(v1d_bwd1_kernel, dt_grid((long long)B * D), dim3(256), 0, st, tr->DY2 + u * B * D, D, tr->cH2, D, tr->CAT2 + u * B * 2 * D,
                           tr->RU2 + u * B * 2 * D, tr->C2 + u * B * D, D, D, tr->DZC2 + u * B * D, tr->DZG2 + u * B * 2 * D, tr->keep, B);
        V1PBwd a{};
        a.B = B; a.T = T; a.S = S; a.A = A; a.E = E; a.D = D;
        a.Wc2T = tr->Wc2T; a.Wg2T = tr->Wg2T; a.Wc1T = tr->Wc1T; a.Wg1T = tr->Wg1T; a.WoT = tr->WoT; a.WqT = tr->WqT; a.WcaT = tr->WcaT; a.WgaT = tr->WgaT;
        a.va = va; a.keys = tr->keys; a.values = tr->values; a.lens = tr->lens;
        a.CATA = tr->CATA; a.CATO = tr->CATO; a.RUa = tr->RUa; a.Ca = tr->Ca; a.Q = tr->Q; a.ALIGN = tr->ALIGN; a.CAT1 = tr->CAT1; a.RU1 = tr->RU1; a.C1 = tr->C1;
        a.CAT2 = tr->CAT2; a.RU2 = tr->RU2; a.C2 = tr->C2; a.DY2 = tr->DY2;
        a.DZC2 = tr->DZC2; a.DZG2 = tr->DZG2; a.DZC1 = tr->DZC1; a.DZG1 = tr->DZG1; a.DY0 = tr->DY0; a.DQ4 = tr->DQ4; a.nd = dt_persist_nd(tr); a.DZCa = tr->DZCa; a.DZGa = tr->DZGa;
        a.DCTX = tr->DCTX; a.dkeys = tr->dkeys; a.dvpart = tr->dvpart;
        a.cHA = tr->cHA; a.cCTX = tr->cCTX; a.cH1 = tr->cH1; a.cH2 = tr->cH2; a.keep = tr->keep; a.keep1 = tr->keep1; a.keepA = tr->keepA;
        a.dcatr = tr->dcatr; a.dcatrA = tr->dcat; a.sink = tr->sink; a.dy1 = tr->dy1; a.dcato = tr->dcato;
        bool timed_out = false;
        T2_TRY(dt_persist_launch(tr, 1, a, &timed_out));
        looped = !timed_out;
        if (timed_out) T2_TRY(zero_carries());       // the accumulators start again; the tapes are rewritten
        else hipLaunchKernelGGL(v1d_sum_chunks_kernel, dt_grid((long long)SB * A), dim3(256), 0, st, tr->DQ4, tr->DQ, (long long)SB, A);
    }
    if (!looped) T2_TRY(dt_run_loop(tr, 1, [&]() -> int {
    for (int t = S - 1; t >= 0; --t) {
        const float* cata = tr->CATA + (size_t)t * B * XA;
        const float *cat1 = tr->CAT1 + (size_t)t * B * 2 * D, *cat2 = tr->CAT2 + (size_t)t * B * 2 * D;
        const float* dy2 = tr->DY2 + (size_t)t * B * D;
        // y2 = y1 + h2': d h2' = d y2 + carry; d y1 = d y2 + d x
        float* dy0 = tr->DY0 + (size_t)t * B * D;
        if (fused) {
            // stage 1 of a cell rides in the epilogue that produces its d h': GRU 2 of step t - 1 behind this step's carry,
            // GRU 1 behind this step's d y1, the attention GRU behind the query layer's product
            if (t == S - 1)
                hipLaunchKernelGGL(v1d_bwd1_kernel, dt_grid((long long)B * D), dim3(256), 0, st, dy2, D, tr->cH2, D, cat2, tr->RU2 + (size_t)t * B * 2 * D,
                                   tr->C2 + (size_t)t * B * D, D, D, tr->DZC2 + (size_t)t * B * D, tr->DZG2 + (size_t)t * B * 2 * D, tr->keep, B);
            SkBwd1 s1; s1.add = tr->cH1; s1.ldadd = D; s1.cat = cat1; s1.ru = tr->RU1 + (size_t)t * B * 2 * D; s1.cnd = tr->C1 + (size_t)t * B * D; s1.X = D; s1.H = D;
            s1.dzc = tr->DZC1 + (size_t)t * B * D; s1.dzg = tr->DZG1 + (size_t)t * B * 2 * D; s1.keep = tr->keep1;
            SkBwd1 s2;
            if (t > 0) {
                const size_t u = (size_t)(t - 1);
                s2.add = tr->DY2 + u * B * D; s2.ldadd = D; s2.cat = tr->CAT2 + u * B * 2 * D; s2.ru = tr->RU2 + u * B * 2 * D; s2.cnd = tr->C2 + u * B * D; s2.X = D; s2.H = D;
                s2.dzc = tr->DZC2 + u * B * D; s2.dzg = tr->DZG2 + u * B * 2 * D; s2.keep = tr->keep;
            }
            T2_TRY(dt_gru_bwd(tr, nullptr, 0, nullptr, 0, cat2, tr->RU2 + (size_t)t * B * 2 * D, tr->C2 + (size_t)t * B * D, D, D, tr->Wc2T, tr->Wg2T,
                              tr->DZC2 + (size_t)t * B * D, tr->DZG2 + (size_t)t * B * 2 * D, dy2, D, tr->cH2, tr->dy1, D, tr->keep, true, &s1, t > 0 ? &s2 : nullptr));
            T2_TRY(dt_gru_bwd(tr, nullptr, 0, nullptr, 0, cat1, tr->RU1 + (size_t)t * B * 2 * D, tr->C1 + (size_t)t * B * D, D, D, tr->Wc1T, tr->Wg1T,
                              tr->DZC1 + (size_t)t * B * D, tr->DZG1 + (size_t)t * B * 2 * D, tr->dy1, D, tr->cH1, dy0, D, tr->keep1, true));
        } else {
        T2_TRY(dt_gru_bwd(tr, dy2, D, tr->cH2, D, cat2, tr->RU2 + (size_t)t * B * 2 * D, tr->C2 + (size_t)t * B * D, D, D, tr->Wc2T, tr->Wg2T,
                          tr->DZC2 + (size_t)t * B * D, tr->DZG2 + (size_t)t * B * 2 * D, dy2, D, tr->cH2, tr->dy1, D));
        T2_TRY(dt_gru_bwd(tr, tr->dy1, D, tr->cH1, D, cat1, tr->RU1 + (size_t)t * B * 2 * D, tr->C1 + (size_t)t * B * D, D, D, tr->Wc1T, tr->Wg1T,
                          tr->DZC1 + (size_t)t * B * D, tr->DZG1 + (size_t)t * B * 2 * D, tr->dy1, D, tr->cH1, dy0, D));
        }
        // output projection: d [h_att | ctx]
        float* dctx_t = tr->DCTX + (size_t)t * E;                       // tape [B][S][E]
        if (dt_skinny_ok(tr, dy0, D, D, B)) {
            SkEpi e; e.y = tr->dcato; e.ldy = A + E; e.cin = tr->cCTX; e.dctx = dctx_t; e.row_stride = (long long)S * E; e.A = A;
            dt_skinny<SK_DCTX>(tr, dy0, D, D, tr->WoT, A + E, e, B);
        } else {
            T2_TRY(dt_gemm(tr, dy0, D, D, tr->WoT, A + E, nullptr, tr->dcato, A + E, nullptr, 0, B));
            hipLaunchKernelGGL(v1d_dctx_kernel, dt_grid((long long)B * E), dim3(256), 0, st, tr->dcato, A, E, tr->cCTX, dctx_t, (long long)S * E, B);
        }
        float* dq = tr->DQ + (size_t)t * B * A;
        hipLaunchKernelGGL(v1d_attn_bwd_kernel, dim3(B), dim3(V1D_ATT_THREADS), attn_lds, st, tr->keys, tr->values, tr->Q + (size_t)t * B * A, va, tr->lens,
                           tr->ALIGN + (size_t)t * B * T, dctx_t, (long long)S * E, T, A, E, dq, tr->dkeys, tr->dvpart);
        // d h_att = output projection's share + d q . Wq^T (+ the carry, added in stage 1)
        // attention GRU on [p1 | ctx_{t-1}]: d ctx_{t-1} -> carry; the p1 share is batched after the loop
        if (fused) {
            SkEpi e; e.res = tr->dcato; e.ldr = A + E;
            e.b1x.add = tr->cHA; e.b1x.ldadd = A; e.b1x.cat = cata; e.b1x.ru = tr->RUa + (size_t)t * B * 2 * A; e.b1x.cnd = tr->Ca + (size_t)t * B * A; e.b1x.X = E; e.b1x.H = A;
            e.b1x.dzc = tr->DZCa + (size_t)t * B * A; e.b1x.dzg = tr->DZGa + (size_t)t * B * 2 * A; e.b1x.keep = tr->keepA;
            dt_skinny<SK_BWD1>(tr, dq, A, A, tr->WqT, A, e, B);
            T2_TRY(dt_gru_bwd(tr, nullptr, 0, nullptr, 0, cata, tr->RUa + (size_t)t * B * 2 * A, tr->Ca + (size_t)t * B * A, E, A, tr->WcaT, tr->WgaT,
                              tr->DZCa + (size_t)t * B * A, tr->DZGa + (size_t)t * B * 2 * A, nullptr, 0, tr->cHA, tr->cCTX, E, tr->keepA, true));
        } else {
        T2_TRY(dt_gemm(tr, dq, A, A, tr->WqT, A, nullptr, tr->dha, A, tr->dcato, A + E, B));
        T2_TRY(dt_gru_bwd(tr, tr->dha, A, tr->cHA, A, cata, tr->RUa + (size_t)t * B * 2 * A, tr->Ca + (size_t)t * B * A, E, A, tr->WcaT, tr->WgaT,
                          tr->DZCa + (size_t)t * B * A, tr->DZGa + (size_t)t * B * 2 * A, nullptr, 0, tr->cHA, tr->cCTX, E));
        }
    }
    return T2_OK;
    }));
    HIP_TRY(hipGetLastError());
    // ---- weight gradients: tape-wide GEMMs ----
    float *gWm = G(pd + "out_projection/kernel"), *gbm = G(pd + "out_projection/bias"), *gWs = G(pd + "stop_token_projection/kernel"), *gbs = G(pd + "stop_token_projection/bias");
    float *gWg2 = G(pd + "gru_2/gates/kernel"), *gbg2 = G(pd + "gru_2/gates/bias"), *gWc2 = G(pd + "gru_2/candidate/kernel"), *gbc2 = G(pd + "gru_2/candidate/bias");
    float *gWg1 = G(pd + "gru_1/gates/kernel"), *gbg1 = G(pd + "gru_1/gates/bias"), *gWc1 = G(pd + "gru_1/candidate/kernel"), *gbc1 = G(pd + "gru_1/candidate/bias");
    float *gWo = G(pd + "output_projection/kernel"), *gbo = G(pd + "output_projection/bias");
    float *gWq = G("v1/attention/query_layer/kernel"), *gva = G("v1/attention/attention_v"), *gWmem = G("v1/attention/memory_layer/kernel");
    float *gWga = G(pd + "attention_gru/gates/kernel"), *gbga = G(pd + "attention_gru/gates/bias"), *gWca = G(pd + "attention_gru/candidate/kernel"),
          *gbca = G(pd + "attention_gru/candidate/bias");
    float *gWp1 = G(pd + "prenet_1/dense/kernel"), *gbp1 = G(pd + "prenet_1/dense/bias"), *gWp0 = G(pd + "prenet_0/dense/kernel"), *gbp0 = G(pd + "prenet_0/dense/bias");
    if (err != T2_OK) return err;
    t2_trainer* ws = &tr->ws;
    T2_TRY(bwd_wgrad(ws, tr->Y2, D, D, tr->dmel_sb, rM, rM, SB, gWm, rM)); T2_TRY(col_sum(ws, tr->dmel_sb, SB, rM, gbm));
    T2_TRY(bwd_wgrad(ws, tr->Y2, D, D, tr->dstop_sb, 1, 1, SB, gWs, 1)); T2_TRY(col_sum(ws, tr->dstop_sb, SB, 1, gbs));
    T2_TRY(bwd_wgrad(ws, tr->CAT2, 2 * D, 2 * D, tr->DZG2, 2 * D, 2 * D, SB, gWg2, 2 * D)); T2_TRY(col_sum(ws, tr->DZG2, SB, 2 * D, gbg2));
    T2_TRY(bwd_wgrad(ws, tr->CAT2R, 2 * D, 2 * D, tr->DZC2, D, D, SB, gWc2, D)); T2_TRY(col_sum(ws, tr->DZC2, SB, D, gbc2));
    T2_TRY(bwd_wgrad(ws, tr->CAT1, 2 * D, 2 * D, tr->DZG1, 2 * D, 2 * D, SB, gWg1, 2 * D)); T2_TRY(col_sum(ws, tr->DZG1, SB, 2 * D, gbg1));
    T2_TRY(bwd_wgrad(ws, tr->CAT1R, 2 * D, 2 * D, tr->DZC1, D, D, SB, gWc1, D)); T2_TRY(col_sum(ws, tr->DZC1, SB, D, gbc1));
    T2_TRY(bwd_wgrad(ws, tr->CATO, A + E, A + E, tr->DY0, D, D, SB, gWo, D)); T2_TRY(col_sum(ws, tr->DY0, SB, D, gbo));
    T2_TRY(bwd_wgrad(ws, tr->CATO, A + E, A, tr->DQ, A, A, SB, gWq, A));
    T2_TRY(col_sum(ws, tr->dvpart, B * V1D_ATT_C, A, gva));          // [B][C][A]: the launch-per-stage loop fills chunk-row 0 .. B - 1 only
    T2_TRY(bwd_wgrad(ws, tr->p1, P1, P1, tr->DZGa, 2 * A, 2 * A, SB, gWga, 2 * A));
    T2_TRY(bwd_wgrad(ws, tr->CATA, XA, XA, tr->DZGa, 2 * A, 2 * A, SB, gWga + (size_t)P1 * 2 * A, 2 * A)); T2_TRY(col_sum(ws, tr->DZGa, SB, 2 * A, gbga));
    T2_TRY(bwd_wgrad(ws, tr->p1, P1, P1, tr->DZCa, A, A, SB, gWca, A));
    T2_TRY(bwd_wgrad(ws, tr->CATAR, XA, XA, tr->DZCa, A, A, SB, gWca + (size_t)P1 * A, A)); T2_TRY(col_sum(ws, tr->DZCa, SB, A, gbca));
    // PreNets, all steps at once (dropout and relu masks from the taped outputs)
    const float inv_keep = 1.0f / (1.0f - tr->drop_rate);
    T2_TRY(dt_gemm(tr, tr->DZGa, 2 * A, 2 * A, tr->WgpT, P1, nullptr, tr->dP1, P1, nullptr, 0, SB));
    T2_TRY(dt_gemm(tr, tr->DZCa, A, A, tr->WcpT, P1, nullptr, tr->dP1, P1, tr->dP1, P1, SB));
    hipLaunchKernelGGL(v1_relu_dropout_bwd_kernel, dt_grid((long long)SB * P1), dim3(256), 0, st, tr->dP1, tr->p1, inv_keep, tr->dP1, (long long)SB * P1);
    T2_TRY(bwd_wgrad(ws, tr->p0, P0, P0, tr->dP1, P1, P1, SB, gWp1, P1)); T2_TRY(col_sum(ws, tr->dP1, SB, P1, gbp1));
    T2_TRY(dt_gemm(tr, tr->dP1, P1, P1, tr->Wp1T, P0, nullptr, tr->dP0, P0, nullptr, 0, SB));
    hipLaunchKernelGGL(v1_relu_dropout_bwd_kernel, dt_grid((long long)SB * P0), dim3(256), 0, st, tr->dP0, tr->p0, inv_keep, tr->dP0, (long long)SB * P0);
    T2_TRY(bwd_wgrad(ws, tr->xin, nf * M, nf * M, tr->dP0, P0, P0, SB, gWp0, P0)); T2_TRY(col_sum(ws, tr->dP0, SB, P0, gbp0));
    // attention memory: d values = sum_t a_t (x) d ctx_t + d keys . W_mem^T; d W_mem = values^T . d keys; d memory = d values * mask
    hipLaunchKernelGGL(dvalues_kernel, dim3(ceil_div(T, 16), B, ceil_div(E, 512)), dim3(512), 0, st, tr->ALIGN, tr->DCTX, B, T, S, E, tr->dvalues);
    T2_TRY(bwd_wgrad(ws, tr->values, E, E, tr->dkeys, A, A, (long long)B * T, gWmem, A));
    T2_TRY(dt_gemm(tr, tr->dkeys, A, A, tr->WmemT, E, nullptr, dmemory, E, tr->dvalues, E, B * T));
    hipLaunchKernelGGL(mask_rows_kernel, dt_grid((long long)B * T * E), dim3(256), 0, st, dmemory, tr->lens, B, T, E);
    HIP_TRY(hipGetLastError());
    tr->have_forward = false;
    return T2_OK;
}

// ---- teacher-forced inference decodes on the training forward ------------------------------------------------------------------
// t2v1_decode_teacher(teacher_forcing = 1) (TrainingHelper / ValidationHelper(teacher_forcing=True), helpers.py:97-184) is the
// training forward with drop_rate 0 (tacotron_v1.py:94-153): the engine keeps a forward-only trainer per shape and a flat copy of
// the decoder's variables (taken at finalize), so such a decode runs as the ONE persistent launch above instead of ~10 launches per
// step.  Shapes outside the persistent class keep the launch-per-phase schedule of v1.h (the parity twin, option "tf_persist" 0).
static int v1_teacher_via_trainer(t2v1_model* m, const float* memory, const int64_t* lens, const float* targets, int B, int T_in, int S, float* mel, float* stop,
                                  float* align, bool* done) {
    *done = false;
    if (!m->opt_tf_persist || !m->opt_tfbatch || m->tf_flat == nullptr) return T2_OK;
    t2v1_dtrainer* tr = nullptr;
    for (t2v1_dtrainer* c : m->tf_trainers)
        if (c->B == B && c->T == T_in && c->S == S) { tr = c; break; }
    if (!tr) {
        {   // eligibility before anything is allocated (the conditions of dt_persist_ok that do not need a trainer)
            const t2v1_dims& d = m->d;
            const int A = d.attention_out_units, E = d.cbhg_out_units, D = d.decoder_out_units;
            if (!(m->opt_train_persist && m->opt_skinny && B <= 48 && A % 16 == 0 && E % 16 == 0 && D % 16 == 0 && m->base.n_cu >= 64)) return T2_OK;
            const size_t NW = V1D_ATT_THREADS / 64;
            if (std::max({NW * (size_t)ceil_div(B, 16) * 64 * 16, (2 * (size_t)A + T_in + NW * E) * 4, ((size_t)E + T_in + 2 * NW * A) * 4}) + 256 > 64 * 1024) return T2_OK;
        }
        if (m->tf_trainers.size() >= 4) {                          // a few shapes at most (each holds its tape)
            HIP_TRY(hipStreamSynchronize(m->base.ctx->stream));
            for (t2v1_dtrainer* c : m->tf_trainers) (void)t2v1_decoder_trainer_destroy(c);
            m->tf_trainers.clear();
        }
        T2_TRY(dt_create(m, B, T_in, S, true, &tr));
        tr->ws.offsets = m->tf_offsets;
        m->tf_trainers.push_back(tr);
    }
    if (!dt_persist_ok(tr)) return T2_OK;                          // e.g. held off after a timeout: the launch-per-phase schedule serves
    T2_TRY(t2v1_train_decoder_forward(tr, m->tf_flat, memory, lens, targets, 0.0f, 0u, mel, stop, align));
    tr->have_forward = false;
    *done = true;
    return T2_OK;
}
static void v1_tf_trainers_free(t2v1_model* m) {
    for (t2v1_dtrainer* c : m->tf_trainers) (void)t2v1_decoder_trainer_destroy(c);
    m->tf_trainers.clear();
}
