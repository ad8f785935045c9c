// Training step of the Tacotron-1 PostNet model (BASELINE.json configs[4], "train_postnet"): TacotronV1PostNetModel.model_fn
// with mode == TRAIN (tacotron/models.py:190-233) -- PostNet(is_training=True) (tacotron/tacotron_v1.py:156-186) = CBHG with
// batch-statistics batch norm (tacotron/modules.py:163-232, Conv1d.call :149-157) -> Dense(num_freq), the masked L1 loss
// (models.py:265-274), and the hand-written backward pass of all of it.  Included by t2mi.hip after v1.h / train_step.h.
//
// The trainer reads every weight straight from the caller's flat fp32 parameter buffer (TF layouts, offsets bound by name):
// nothing of the model's inference copies is used, so a step needs no refresh; t2v1_postnet_trainer_export rebuilds the
// inference copies of the PostNet from the buffer when the caller wants to predict with the trained weights.
// Layout tricks: the conv bank (K kernels of width 1..K) runs as ONE convolution of width K whose kernel holds the K kernels
// at their SAME-padding offsets (zeros elsewhere): one GEMM, one batch norm over K*C channels, one weight-gradient GEMM whose
// rows are scattered back to the K variables.  The GRU recurrences are the whole-sequence kernels of kernels.h
// (gru_seq_kernel with a tape, gru_seq_bwd_kernel); every weight gradient is a tape-wide GEMM on the conv GEMM kernels.
#pragma once

#define T2V1_MAX_BANK 16

struct BankPtrs { const float* p[T2V1_MAX_BANK]; };
struct BankDst { float* p[T2V1_MAX_BANK]; };

// unified bank kernel wu [KU*cin][K*C] from the K TF kernels [k][cin][C]: tap j of kernel k sits at unified tap j + PU - (k-1)/2
__global__ __launch_bounds__(256) void bank_pack_kernel(BankPtrs src, float* __restrict__ wu, int K, int cin, int C) {
    const int KC = K * C, PU = (K - 1) / 2;
    const long long n = (long long)K * cin * KC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(i % KC), ci = (int)((i / KC) % cin), ju = (int)(i / ((long long)KC * cin));
        const int kk = col / C + 1, c = col - (kk - 1) * C;
        const int j = ju - PU + (kk - 1) / 2;
        wu[i] = (j >= 0 && j < kk) ? src.p[kk - 1][((size_t)j * cin + ci) * C + c] : 0.f;
    }
}
// ... and the way back for the gradient: dst_k[j][ci][c] = dwu[j + PU - (k-1)/2][ci][(k-1) C + c]
__global__ __launch_bounds__(256) void bank_unpack_kernel(const float* __restrict__ dwu, BankDst dst, int K, int cin, int C) {
    const int KC = K * C, PU = (K - 1) / 2;
    const long long n = (long long)K * cin * KC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(i % KC), ci = (int)((i / KC) % cin), ju = (int)(i / ((long long)KC * cin));
        const int kk = col / C + 1, c = col - (kk - 1) * C;
        const int j = ju - PU + (kk - 1) / 2;
        if (j >= 0 && j < kk) dst.p[kk - 1][((size_t)j * cin + ci) * C + c] = dwu[i];
    }
}
// per-channel vectors of the K bank layers <-> one vector of K*C channels
__global__ void bank_vec_gather_kernel(BankPtrs src, float* __restrict__ dst, int K, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * C) dst[i] = src.p[i / C][i % C];
}
// The inference CBHG's unified bank layer (CbhgW::bank_u) from its K folded layers: kernel, scale, shift (+ the transposed copy
// of the fast path).  Called at finalize and after a trainer's export.
static int v1_bank_unify(t2_model* b, CbhgW* c) {
    const int K = c->K, cin = c->cin, C = c->C, KC = K * C;
    if (K > T2V1_MAX_BANK) return T2_OK;
    hipStream_t st = b->ctx->stream;
    ConvLayer& U = c->bank_u;
    if (U.w == nullptr) {
        U.k = K; U.cin = cin; U.cout = KC;
        T2_TRY(dev_alloc(b, (void**)&U.w, (size_t)K * cin * KC * 4));
        T2_TRY(dev_alloc(b, (void**)&U.scale, (size_t)KC * 4));
        T2_TRY(dev_alloc(b, (void**)&U.shift, (size_t)KC * 4));
        if (cin % 16 == 0) T2_TRY(dev_alloc(b, (void**)&U.wt, (size_t)K * cin * KC * 4));
    }
    BankPtrs w{}, sc{}, sh{};
    for (int k = 0; k < K; ++k) { w.p[k] = c->bank[k].w; sc.p[k] = c->bank[k].scale; sh.p[k] = c->bank[k].shift; }
    hipLaunchKernelGGL(bank_pack_kernel, dim3((unsigned)std::min<long long>(4096, ((long long)K * cin * KC + 255) / 256)), dim3(256), 0, st, w, U.w, K, cin, C);
    hipLaunchKernelGGL(bank_vec_gather_kernel, dim3(ceil_div(KC, 256)), dim3(256), 0, st, sc, U.scale, K, C);
    hipLaunchKernelGGL(bank_vec_gather_kernel, dim3(ceil_div(KC, 256)), dim3(256), 0, st, sh, U.shift, K, C);
    conv_refresh_wt(st, U);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}
__global__ void bank_vec_scatter_kernel(const float* __restrict__ src, BankDst dst, int K, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K * C) dst.p[i / C][i % C] = src[i];
}
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}
// HighwayNet.call (modules.py:124-127) from the two pre-activations: y = relu(zh) sigmoid(zt) + x (1 - sigmoid(zt))
__global__ __launch_bounds__(256) void highway_train_kernel(const float* __restrict__ zh, const float* __restrict__ zt, const float* __restrict__ x,
                                                            float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float tg = sigmoidf_(zt[i]);
        y[i] = fmaxf(zh[i], 0.f) * tg + x[i] * (1.0f - tg);
    }
}
// its backward: dzh = dy t [zh > 0], dzt = dy (relu(zh) - x) t (1 - t), dx (direct part) = dy (1 - t)
__global__ __launch_bounds__(256) void highway_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ zh, const float* __restrict__ zt,
                                                          const float* __restrict__ x, float* __restrict__ dzh, float* __restrict__ dzt,
                                                          float* __restrict__ dx, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float tg = sigmoidf_(zt[i]), h = fmaxf(zh[i], 0.f), g = dy[i];
        dzh[i] = zh[i] > 0.f ? g * tg : 0.f;
        dzt[i] = g * (h - x[i]) * tg * (1.0f - tg);
        dx[i] = g * (1.0f - tg);
    }
}
// MaxPooling1D(2, 1, SAME) backward (App. A3: y[t] = max(x[t], x[t + 1]), the last frame passes through): the gradient goes to the
// position that holds the maximum (the first one on a tie, as TensorFlow's max-pool gradient does)
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                           long long Mtot, int T, int C) {
    const long long n = Mtot * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        const int t = (int)(m % T);
        const float v = x[i];
        float g = 0.f;
        if (t + 1 >= T || v >= x[i + C]) g += dy[i];                 // window starting here
        if (t > 0 && v > x[i - C]) g += dy[i - C];                   // window starting one frame earlier
        dx[i] = g;
    }
}
// d loss / d spec of the masked L1 loss (models.py:265-274; SUM_BY_NONZERO_WEIGHTS): sign(y_hat - y) mask[row] / count
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* y_hat, const float* __restrict__ y, const float* __restrict__ mask,
                                                      const float* __restrict__ count, long long rows, int F, float* g) {
    const float ic = count[0] > 0.f ? 1.0f / count[0] : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * F; i += (long long)gridDim.x * blockDim.x) {
        const float d = y_hat[i] - y[i];
        g[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * mask[i / F] * ic;
    }
}
// previous state and r * previous state of every position of one direction (operands of the recurrent weight gradients)
__global__ __launch_bounds__(256) void gru_prev_kernel(const float* __restrict__ out, const float* __restrict__ tape_ru, const int* __restrict__ lens,
                                                       float* __restrict__ hprev, float* __restrict__ rh, int B, int T, int H, int dir) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)B * T * H; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % H), pos = (int)((i / H) % T), b = (int)(i / ((long long)H * T));
        const int len = lens[b];
        float h = 0.f, r = 0.f;
        if (pos < len) {
            const int pp = dir ? pos + 1 : pos - 1;                 // the position the previous step wrote
            if (pp >= 0 && pp < len) h = out[((size_t)b * T + pp) * 2 * H + dir * H + c];
            r = tape_ru[((size_t)b * T + pos) * 2 * H + c];
        }
        hprev[i] = h; rh[i] = r * h;
    }
}

struct t2v1_ptrainer {
    t2v1_model* m = nullptr;
    int which = 1;                       // 1: the PostNet (its CBHG, Dense(num_freq), L1 loss); 0: EncoderV1 (embedding, PreNets with dropout, its CBHG)
    const CbhgW* c = nullptr;            // dimensions of that CBHG
    std::string pfx;                     // its variable prefix
    int B = 0, T = 0;
    t2_trainer ws;                       // workspace carrier of the shared helpers (split-K partials, column sums, BN partial sums, offsets)
    char* slab = nullptr;
    bool have_forward = false;           // a tape the backward pass can consume
    bool have_stats = false;             // batch statistics of the last forward (they outlive the backward pass)
    const float* params = nullptr;       // flat parameter buffer of the last forward
    int* lens = nullptr;
    // tape
    float *x0, *bank_raw, *bank_act, *pooled, *p1raw, *p1, *p2raw, *p2, *hw0, *a[9], *zh[8], *zt[8], *xz, *out, *tru[2], *tc[2], *spec;
    float *mean[3], *invstd[3], *var[3], *gam_u, *bet_u, *dgam[3], *dbet[3];
    // per-step packed operands
    float *wbank, *proj1T, *proj2T, *wlT, *wgxT[2], *wcxT[2], *whT[8], *wtT[8], *wadjT;
    // backward temporaries
    float *dout, *dzg[2], *dzc[2], *hprev, *rh, *da[2], *dzh, *dzt, *d80a, *d80b, *dp1a, *dp1b, *dkc_a, *dkc_b, *dwbank, *loss3;
    // EncoderV1 only: token ids, embedded input, PreNet outputs (after relu and dropout), their gradients, transposed PreNet kernels,
    // the flipped unified bank kernel and the gradient of the CBHG's input
    int* ids = nullptr;
    float *emb = nullptr, *pre[2] = {nullptr, nullptr}, *dpre[2] = {nullptr, nullptr}, *wpreT[2] = {nullptr, nullptr}, *wbankF = nullptr, *dx0 = nullptr, *dxpart = nullptr;
    float drop_rate = 0.f;
    uint32_t seed = 0;
    bool have_ids = false;               // the last forward looked the input up in the table (its gradient goes to the table)
};

static const float* pt_w(t2v1_ptrainer* tr, const std::string& name, int* err) {
    auto it = tr->ws.offsets.find(name);
    if (it == tr->ws.offsets.end()) { *err = fail(T2_ESTATE, "no offset bound for '%s' (t2v1_postnet_trainer_bind)", name.c_str()); return nullptr; }
    return tr->params + it->second;
}
static inline dim3 pt_grid(long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(4096, (n + 255) / 256))); }

extern "C" int t2v1_postnet_trainer_destroy(t2v1_ptrainer* tr) {
    if (!tr) return T2_OK;
    (void)hipSetDevice(tr->m->base.ctx->device);
    (void)hipStreamSynchronize(tr->m->base.ctx->stream);
    if (tr->slab) (void)hipFree(tr->slab);
    if (tr->ws.cs_part) (void)hipFree(tr->ws.cs_part);
    if (tr->ws.wg_part) (void)hipFree(tr->ws.wg_part);
    if (tr->ws.bn_part) (void)hipFree(tr->ws.bn_part);
    delete tr;
    return T2_OK;
}

static int cbhg_trainer_create(t2v1_model* m, int which, int B, int T, t2v1_ptrainer** out) {
    T2_TRY(v1_ready(m));
    if (!out) return fail(T2_EINVAL, "null argument");
    T2_TRY(check_batch(B, T));
    const CbhgW& c = which == 1 ? m->post_cbhg : m->enc_cbhg;
    const int nhw = (int)c.highway.size();
    if (c.half != 128) return fail(T2_EINVAL, "the CBHG training step is built for cbhg_out_units = 256 (whole-sequence GRU kernels of 128 units)");
    if (c.K > T2V1_MAX_BANK || nhw > 8) return fail(T2_EINVAL, "at most %d bank kernels and 8 highway layers", T2V1_MAX_BANK);
    HIP_TRY(hipSetDevice(m->base.ctx->device));
    t2v1_ptrainer* tr = new t2v1_ptrainer();
    tr->m = m; tr->B = B; tr->T = T; tr->which = which; tr->c = &c;
    tr->pfx = which == 1 ? "v1/postnet/cbhg" : "v1/encoder/cbhg";
    tr->ws.m = &m->base;
    const size_t M = (size_t)B * T, KC = (size_t)c.K * c.C, H = c.half, F = which == 1 ? m->d.num_freq : 4, cin = c.cin, P1 = c.P1, P2 = c.P2;
    Carver cv;
    std::vector<std::pair<float**, size_t>> plan;
    auto want = [&](float** p, size_t floats) { plan.push_back({p, cv.take(floats * 4)}); };
    want(&tr->x0, M * cin); want(&tr->bank_raw, M * KC); want(&tr->bank_act, M * KC); want(&tr->pooled, M * KC);
    want(&tr->p1raw, M * P1); want(&tr->p1, M * P1); want(&tr->p2raw, M * P2); want(&tr->p2, M * P2); want(&tr->hw0, M * P2);
    for (int i = 0; i <= nhw; ++i) want(&tr->a[i], M * H);
    for (int i = 0; i < nhw; ++i) { want(&tr->zh[i], M * H); want(&tr->zt[i], M * H); }
    want(&tr->xz, 2 * M * 3 * H); want(&tr->out, M * 2 * H);
    for (int dir = 0; dir < 2; ++dir) { want(&tr->tru[dir], M * 2 * H); want(&tr->tc[dir], M * H); }
    want(&tr->spec, M * F);
    const size_t chan[3] = {KC, P1, P2};
    for (int l = 0; l < 3; ++l) { want(&tr->mean[l], chan[l]); want(&tr->invstd[l], chan[l]); want(&tr->var[l], chan[l]); want(&tr->dgam[l], chan[l]); want(&tr->dbet[l], chan[l]); }
    want(&tr->gam_u, KC); want(&tr->bet_u, KC);
    want(&tr->wbank, (size_t)c.K * cin * KC); want(&tr->dwbank, (size_t)c.K * cin * KC);
    want(&tr->proj1T, 3 * P1 * KC); want(&tr->proj2T, 3 * P2 * P1); want(&tr->wlT, F * 2 * H);
    for (int dir = 0; dir < 2; ++dir) { want(&tr->wgxT[dir], 2 * H * H); want(&tr->wcxT[dir], H * H); }
    for (int i = 0; i < nhw; ++i) { want(&tr->whT[i], H * H); want(&tr->wtT[i], H * H); }
    want(&tr->wadjT, H * P2);
    want(&tr->dout, M * 2 * H);
    for (int dir = 0; dir < 2; ++dir) { want(&tr->dzg[dir], M * 2 * H); want(&tr->dzc[dir], M * H); }
    want(&tr->hprev, M * H); want(&tr->rh, M * H); want(&tr->da[0], M * H); want(&tr->da[1], M * H); want(&tr->dzh, M * H); want(&tr->dzt, M * H);
    want(&tr->d80a, M * P2); want(&tr->d80b, M * P2); want(&tr->dp1a, M * P1); want(&tr->dp1b, M * P1); want(&tr->dkc_a, M * KC); want(&tr->dkc_b, M * KC);
    want(&tr->loss3, 8);
    size_t o_ids = 0;
    if (which == 0) {
        const size_t E0 = m->d.embedding_dim, U0 = m->d.enc_prenet_units[0], U1 = m->d.enc_prenet_units[1];
        want(&tr->emb, M * E0); want(&tr->pre[0], M * U0); want(&tr->pre[1], M * U1); want(&tr->dpre[0], M * std::max(U0, E0)); want(&tr->dpre[1], M * U1);
        want(&tr->wpreT[0], E0 * U0); want(&tr->wpreT[1], U0 * U1); want(&tr->wbankF, (size_t)c.K * cin * KC); want(&tr->dx0, M * cin); want(&tr->dxpart, M * cin * 16);
        o_ids = cv.take(M * 4);
    }
    const size_t o_lens = cv.take((size_t)B * 4);
    hipError_t e = hipMalloc((void**)&tr->slab, cv.off);
    if (e != hipSuccess) { delete tr; return fail(T2_ENOMEM, "hipMalloc(PostNet trainer, %zu bytes) failed: %s", cv.off, hipGetErrorString(e)); }
    for (auto& pr : plan) *pr.first = (float*)(tr->slab + pr.second);
    tr->lens = (int*)(tr->slab + o_lens);
    if (which == 0) tr->ids = (int*)(tr->slab + o_ids);
    int rc = v1_lens_to_dev(m, nullptr, B, T, 0, tr->lens);
    if (rc != T2_OK) { t2v1_postnet_trainer_destroy(tr); return rc; }
    *out = tr;
    return T2_OK;
}
extern "C" int t2v1_postnet_trainer_create(t2v1_model* m, int B, int T, t2v1_ptrainer** out) { return cbhg_trainer_create(m, 1, B, T, out); }
extern "C" int t2v1_encoder_trainer_create(t2v1_model* m, int B, int T, t2v1_ptrainer** out) { return cbhg_trainer_create(m, 0, B, T, out); }

// where variable `name` lives in the flat parameter / gradient buffers (floats from the start)
extern "C" int t2v1_postnet_trainer_bind(t2v1_ptrainer* tr, const char* name, int64_t offset) {
    if (!tr || !name || offset < 0 || (offset & 3) != 0) return fail(T2_EINVAL, "bad argument (offsets are multiples of 4 floats)");
    tr->ws.offsets[name] = offset;
    return T2_OK;
}

static int pt_bn_forward(t2v1_ptrainer* tr, const float* raw, int l, int C, const float* gamma, const float* beta, int act, float* outp) {
    hipStream_t st = tr->m->base.ctx->stream;
    const int M = tr->B * tr->T;
    T2_TRY(bn_part_reserve(&tr->ws, C));
    hipLaunchKernelGGL(bn_stats_kernel, dim3(ceil_div(C, 64), T2_BN_CHUNKS), dim3(256), 0, st, raw, M, C, tr->ws.bn_part);
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, tr->ws.bn_part, M, C, tr->m->base.d.bn_eps, tr->mean[l], tr->invstd[l], tr->var[l]);
    hipLaunchKernelGGL(bn_apply_kernel, pt_grid((long long)M * C), dim3(256), 0, st, raw, tr->mean[l], tr->invstd[l], gamma, beta, 0.f, 0u, 0, outp, (long long)M, C, act);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// CBHG.call(is_training=True) (modules.py:205-229) on x dev [B][T][cin] with the lengths in tr->lens: leaves the output in
// tr->out [B][T][2H] and the tape of the backward pass in the trainer
static int cbhg_train_forward_impl(t2v1_ptrainer* tr, const float* x) {
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    hipStream_t st = b->ctx->stream;
    const CbhgW& c = *tr->c;
    const int B = tr->B, T = tr->T, M = B * T, K = c.K, C = c.C, KC = K * C, H = c.half, cin = c.cin, P1 = c.P1, P2 = c.P2;
    const int nhw = (int)c.highway.size();
    int err = T2_OK;
    const std::string p = tr->pfx;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    if (x != tr->x0) HIP_TRY(hipMemcpyAsync(tr->x0, x, (size_t)M * cin * 4, hipMemcpyDeviceToDevice, st));
    // conv bank as one convolution of width K
    BankPtrs bk{}, bg{}, bb{};
    for (int k = 1; k <= K; ++k) {
        const std::string q = p + "/conv1d_K" + std::to_string(k);
        bk.p[k - 1] = W(q + "/kernel"); bg.p[k - 1] = W(q + "/batch_normalization/gamma"); bb.p[k - 1] = W(q + "/batch_normalization/beta");
    }
    if (err != T2_OK) return err;
    hipLaunchKernelGGL(bank_pack_kernel, pt_grid((long long)K * cin * KC), dim3(256), 0, st, bk, tr->wbank, K, cin, C);
    hipLaunchKernelGGL(bank_vec_gather_kernel, dim3(ceil_div(KC, 256)), dim3(256), 0, st, bg, tr->gam_u, K, C);
    hipLaunchKernelGGL(bank_vec_gather_kernel, dim3(ceil_div(KC, 256)), dim3(256), 0, st, bb, tr->bet_u, K, C);
    ConvLayer L{};
    L.w = tr->wbank; L.k = K; L.cin = cin; L.cout = KC;
    T2_TRY(launch_conv(b, L, tr->x0, cin, tr->bank_raw, KC, M, T, 0, nullptr, 0));
    T2_TRY(pt_bn_forward(tr, tr->bank_raw, 0, KC, tr->gam_u, tr->bet_u, 1, tr->bank_act));
    hipLaunchKernelGGL(maxpool2_kernel, pt_grid((long long)M * KC), dim3(256), 0, st, tr->bank_act, tr->pooled, (long long)M, T, KC);
    // projections
    L = ConvLayer{}; L.w = const_cast<float*>(W(p + "/proj1/kernel")); L.k = 3; L.cin = KC; L.cout = P1;
    const float *g1 = W(p + "/proj1/batch_normalization/gamma"), *b1 = W(p + "/proj1/batch_normalization/beta");
    if (err != T2_OK) return err;
    if (tr->dxpart) { L.part = tr->dxpart; L.part_floats = (size_t)M * tr->c->cin * 16; }     // encoder sizes: 70 output tiles x 384 k-tiles
    T2_TRY(launch_conv(b, L, tr->pooled, KC, tr->p1raw, P1, M, T, 0, nullptr, 0));
    T2_TRY(pt_bn_forward(tr, tr->p1raw, 1, P1, g1, b1, 1, tr->p1));
    L = ConvLayer{}; L.w = const_cast<float*>(W(p + "/proj2/kernel")); L.k = 3; L.cin = P1; L.cout = P2;
    const float *g2 = W(p + "/proj2/batch_normalization/gamma"), *b2 = W(p + "/proj2/batch_normalization/beta");
    if (err != T2_OK) return err;
    T2_TRY(launch_conv(b, L, tr->p1, P1, tr->p2raw, P2, M, T, 0, nullptr, 0));
    T2_TRY(pt_bn_forward(tr, tr->p2raw, 2, P2, g2, b2, 0, tr->p2));
    hipLaunchKernelGGL(add_kernel, pt_grid((long long)M * P2), dim3(256), 0, st, tr->p2, tr->x0, tr->hw0, (long long)M * P2);      // residual (modules.py:214)
    // adjustment layer (modules.py:216-217)
    if (c.has_adj) {
        L = ConvLayer{}; L.w = const_cast<float*>(W(p + "/adjustment/kernel")); L.shift = const_cast<float*>(W(p + "/adjustment/bias")); L.k = 1; L.cin = P2; L.cout = H;
        if (err != T2_OK) return err;
        T2_TRY(launch_conv(b, L, tr->hw0, P2, tr->a[0], H, M, M, 0, nullptr, 0));
    } else {
        HIP_TRY(hipMemcpyAsync(tr->a[0], tr->hw0, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    }
    for (int i = 0; i < nhw; ++i) {
        const std::string hp = p + "/highway_" + std::to_string(i + 1);
        ConvLayer Lh{}, Lt{};
        Lh.w = const_cast<float*>(W(hp + "/H/kernel")); Lh.shift = const_cast<float*>(W(hp + "/H/bias")); Lh.k = 1; Lh.cin = H; Lh.cout = H;
        Lt.w = const_cast<float*>(W(hp + "/T/kernel")); Lt.shift = const_cast<float*>(W(hp + "/T/bias")); Lt.k = 1; Lt.cin = H; Lt.cout = H;
        if (err != T2_OK) return err;
        T2_TRY(launch_conv(b, Lh, tr->a[i], H, tr->zh[i], H, M, M, 0, nullptr, 0));
        T2_TRY(launch_conv(b, Lt, tr->a[i], H, tr->zt[i], H, M, M, 0, nullptr, 0));
        hipLaunchKernelGGL(highway_train_kernel, pt_grid((long long)M * H), dim3(256), 0, st, tr->zh[i], tr->zt[i], tr->a[i], tr->a[i + 1], (long long)M * H);
    }
    // BiGRU over all positions (modules.py:222-229; the PostNet passes no lengths)
    const char* dirs[2] = {"fw", "bw"};
    GruSeqArgs ga{};
    for (int dir = 0; dir < 2; ++dir) {
        const std::string gp = p + "/bidirectional_rnn/" + dirs[dir] + "/gru_cell";
        const float *wg = W(gp + "/gates/kernel"), *bgt = W(gp + "/gates/bias"), *wc = W(gp + "/candidate/kernel"), *bc = W(gp + "/candidate/bias");
        if (err != T2_OK) return err;
        float* xd = tr->xz + (size_t)dir * M * 3 * H;
        ConvLayer Lg{}, Lc{};
        Lg.w = const_cast<float*>(wg); Lg.shift = const_cast<float*>(bgt); Lg.k = 1; Lg.cin = H; Lg.cout = 2 * H;       // rows [0, H) of [2H][2H]: the input half
        Lc.w = const_cast<float*>(wc); Lc.shift = const_cast<float*>(bc); Lc.k = 1; Lc.cin = H; Lc.cout = H;
        T2_TRY(launch_conv(b, Lg, tr->a[nhw], H, xd, 3 * H, M, M, 0, nullptr, 0));
        T2_TRY(launch_conv(b, Lc, tr->a[nhw], H, xd + 2 * H, 3 * H, M, M, 0, nullptr, 0));
        ga.xz[dir] = xd; ga.whg[dir] = wg + (size_t)H * 2 * H; ga.whc[dir] = wc + (size_t)H * H;
        ga.tape_ru[dir] = tr->tru[dir]; ga.tape_c[dir] = tr->tc[dir];
    }
    ga.lens = tr->lens; ga.out = tr->out; ga.B = B; ga.T = T;
    HIP_TRY(hipMemsetAsync(tr->out, 0, (size_t)M * 2 * H * 4, st));
    hipLaunchKernelGGL(gru_seq_kernel<128>, dim3(ceil_div(B, 16), 2), dim3(512), 0, st, ga);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// PostNet(is_training=True)(mel) (tacotron_v1.py:180-183).  mel dev [B][T][num_mels]; flat_params dev: the parameter buffer
// (offsets bound with t2v1_postnet_trainer_bind); spec_out dev [B][T][num_freq].
extern "C" int t2v1_postnet_train_forward(t2v1_ptrainer* tr, const float* flat_params, const float* mel, float* spec_out) {
    if (!tr || !flat_params || !mel || !spec_out) return fail(T2_EINVAL, "null argument");
    if (tr->which != 1) return fail(T2_EINVAL, "not a PostNet trainer");
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const int M = tr->B * tr->T, H = tr->c->half, F = m->d.num_freq;
    tr->params = flat_params;
    tr->have_forward = false;
    tr->have_stats = false;
    int err = T2_OK;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    T2_TRY(cbhg_train_forward_impl(tr, mel));
    // Dense(num_freq) (tacotron_v1.py:176,182)
    ConvLayer L{}; L.w = const_cast<float*>(W("v1/postnet/linear/kernel")); L.shift = const_cast<float*>(W("v1/postnet/linear/bias")); L.k = 1; L.cin = 2 * H; L.cout = F;
    if (err != T2_OK) return err;
    T2_TRY(launch_conv(b, L, tr->out, 2 * H, tr->spec, F, M, M, 0, nullptr, 0));
    HIP_TRY(hipMemcpyAsync(spec_out, tr->spec, (size_t)M * F * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    tr->have_forward = true;
    tr->have_stats = true;
    return T2_OK;
}

// CBHG.call(is_training=True) (modules.py:205-229) on its own: the trainer's CBHG (the PostNet's for a PostNet trainer, the
// encoder's for an encoder trainer) with batch statistics.  x dev [B][T][in_units]; lens host [B] or null; out dev [B][T][out_units].
// Block-level mirror only: no backward pass follows this entry (the model-level forwards above keep their tapes).
extern "C" int t2v1_cbhg_train_forward(t2v1_ptrainer* tr, const float* flat_params, const float* x, const int64_t* lens, float* out) {
    if (!tr || !flat_params || !x || !out) return fail(T2_EINVAL, "null argument");
    t2v1_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->base.ctx->device));
    hipStream_t st = m->base.ctx->stream;
    tr->params = flat_params;
    tr->have_forward = false;
    tr->have_stats = false;
    T2_TRY(v1_lens_to_dev(m, lens, tr->B, tr->T, 0, tr->lens));
    T2_TRY(cbhg_train_forward_impl(tr, x));
    HIP_TRY(hipMemcpyAsync(out, tr->out, (size_t)tr->B * tr->T * 2 * tr->c->half * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// batch-norm backward of one layer: d_out = gradient of the layer's output, act_out = its output when the activation is relu
// (the mask), null for the identity; raw = the convolution output; -> d raw (dy), dgamma / dbeta in tr->dgam[l] / dbet[l]
static int pt_bn_backward(t2v1_ptrainer* tr, const float* d_out, const float* act_out, const float* raw, int l, int C, const float* gamma, float* dy) {
    hipStream_t st = tr->m->base.ctx->stream;
    const int M = tr->B * tr->T;
    T2_TRY(bn_part_reserve(&tr->ws, C));
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ceil_div(C, 64), T2_BN_CHUNKS), dim3(256), 0, st, d_out, act_out, raw, tr->mean[l], tr->invstd[l], 1.0f, M, C, tr->ws.bn_part);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, tr->ws.bn_part, C, tr->dgam[l], tr->dbet[l]);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, pt_grid((long long)M * C), dim3(256), 0, st, d_out, act_out, raw, tr->mean[l], tr->invstd[l], gamma, tr->dgam[l], tr->dbet[l], 1.0f,
                       (long long)M, C, dy);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}
// dW [k*cin][cout] = im2col(x)^T . dy of a SAME convolution
static int pt_conv_wgrad(t2v1_ptrainer* tr, const float* x, int cin, const float* dy, int cout, int k, float* g) {
    ConvArgs a{};
    a.x = x; a.ldx = cin; a.w = dy; a.y = g; a.ldy = cout;
    a.Mtot = k * cin; a.T = tr->T; a.Cin = cin; a.N = cout; a.ktaps = k; a.pad_l = (k - 1) / 2; a.act = 0; a.trans_a = 1; a.red = tr->B * tr->T;
    a.vec_a = (cin % 4 == 0) && (((uintptr_t)x) % 16 == 0);
    a.vec_b = (cout % 4 == 0) && (((uintptr_t)dy) % 16 == 0);
    return launch_wgrad(&tr->ws, a);
}
static int pt_transpose(t2v1_ptrainer* tr, const float* src, float* dst, int R, int C) {
    hipLaunchKernelGGL(transpose_kernel, pt_grid((long long)R * C), dim3(256), 0, tr->m->base.ctx->stream, dst, R, src, C, C, R);      // src [R][C] -> dst [C][R]
    return T2_OK;
}
// y [M][N] = x [M][K] . wT^T, wT given as [K][N] row-major (a transposed weight), + res
static int pt_dgrad_dense(t2v1_ptrainer* tr, const float* x, int K, float* wT, int N, float* y, const float* res) {
    ConvLayer L{};
    L.w = wT; L.k = 1; L.cin = K; L.cout = N;
    const int M = tr->B * tr->T;
    return launch_conv(&tr->m->base, L, x, K, y, N, M, M, 0, res, N);
}

// Backward of t2v1_postnet_train_forward + the L1 loss of models.py:208-211.  spec_target dev [B][T][num_freq], mask dev [B][T]
// (labels.spec_loss_mask), glob_count dev [1] or null (the non-zero weight count of a larger data-parallel batch).  Writes the
// gradient of every trainable PostNet variable into flat_grads at the bound offsets and {sum, count, loss} into loss_out (dev [3]).
static int cbhg_train_backward_impl(t2v1_ptrainer* tr, float* flat_grads, bool want_dx);
extern "C" int t2v1_postnet_train_backward(t2v1_ptrainer* tr, const float* spec_target, const float* mask, const float* glob_count,
                                           float* flat_grads, float* loss_out) {
    if (!tr || !spec_target || !mask || !flat_grads || !loss_out) return fail(T2_EINVAL, "null argument");
    if (tr->which != 1) return fail(T2_EINVAL, "not a PostNet trainer");
    if (!tr->have_forward) return fail(T2_ESTATE, "t2v1_postnet_train_forward has not run");
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const int B = tr->B, T = tr->T, M = B * T, H = tr->c->half, F = m->d.num_freq;
    int err = T2_OK;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    auto G = [&](const std::string& n) -> float* {
        auto it = tr->ws.offsets.find(n);
        if (it == tr->ws.offsets.end()) { err = fail(T2_ESTATE, "no offset bound for '%s'", n.c_str()); return nullptr; }
        return flat_grads + it->second;
    };
    // loss and d spec
    T2_TRY(t2_spec_loss(b->ctx, tr->spec, spec_target, mask, B, T, F, tr->loss3));
    HIP_TRY(hipMemcpyAsync(loss_out, tr->loss3, 3 * 4, hipMemcpyDeviceToDevice, st));
    float* dspec = tr->spec;                                   // in place: the forward output is not needed again
    hipLaunchKernelGGL(l1_grad_kernel, pt_grid((long long)M * F), dim3(256), 0, st, tr->spec, spec_target, mask, glob_count ? glob_count : tr->loss3 + 1, (long long)M, F, dspec);
    // Dense(num_freq)
    {
        const float* wl = W("v1/postnet/linear/kernel");
        float *gw = G("v1/postnet/linear/kernel"), *gb = G("v1/postnet/linear/bias");
        if (err != T2_OK) return err;
        T2_TRY(bwd_wgrad(&tr->ws, tr->out, 2 * H, 2 * H, dspec, F, F, M, gw, F));
        T2_TRY(col_sum(&tr->ws, dspec, M, F, gb));
        T2_TRY(pt_transpose(tr, wl, tr->wlT, 2 * H, F));
        T2_TRY(pt_dgrad_dense(tr, dspec, F, tr->wlT, 2 * H, tr->dout, nullptr));
    }
    T2_TRY(cbhg_train_backward_impl(tr, flat_grads, false));
    tr->have_forward = false;                                  // the tape is consumed (d spec overwrote the forward output)
    return T2_OK;
}

// Backward of cbhg_train_forward_impl from tr->dout (the gradient of the CBHG's output): every gradient of the CBHG's variables
// into flat_grads; want_dx: the gradient of its input into tr->dx0 (residual branch + the bank's data gradient)
static int cbhg_train_backward_impl(t2v1_ptrainer* tr, float* flat_grads, bool want_dx) {
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    hipStream_t st = b->ctx->stream;
    const CbhgW& c = *tr->c;
    const int B = tr->B, T = tr->T, M = B * T, K = c.K, C = c.C, KC = K * C, H = c.half, cin = c.cin, P1 = c.P1, P2 = c.P2;
    const int nhw = (int)c.highway.size();
    int err = T2_OK;
    const std::string p = tr->pfx;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    auto G = [&](const std::string& n) -> float* {
        auto it = tr->ws.offsets.find(n);
        if (it == tr->ws.offsets.end()) { err = fail(T2_ESTATE, "no offset bound for '%s'", n.c_str()); return nullptr; }
        return flat_grads + it->second;
    };
    // BiGRU: reverse-time recurrences, then tape-wide weight gradients
    const char* dirs[2] = {"fw", "bw"};
    GruSeqBwdArgs ga{};
    ga.dout = tr->dout; ga.out = tr->out; ga.lens = tr->lens; ga.B = B; ga.T = T;
    for (int dir = 0; dir < 2; ++dir) {
        const std::string gp = p + "/bidirectional_rnn/" + dirs[dir] + "/gru_cell";
        const float *wg = W(gp + "/gates/kernel"), *wc = W(gp + "/candidate/kernel");
        if (err != T2_OK) return err;
        ga.tape_ru[dir] = tr->tru[dir]; ga.tape_c[dir] = tr->tc[dir];
        ga.whg[dir] = wg + (size_t)H * 2 * H; ga.whc[dir] = wc + (size_t)H * H;
        ga.dzg[dir] = tr->dzg[dir]; ga.dzc[dir] = tr->dzc[dir];
        HIP_TRY(hipMemsetAsync(tr->dzg[dir], 0, (size_t)M * 2 * H * 4, st));
        HIP_TRY(hipMemsetAsync(tr->dzc[dir], 0, (size_t)M * H * 4, st));
    }
    hipLaunchKernelGGL(gru_seq_bwd_kernel<128>, dim3(ceil_div(B, 16), 2), dim3(512), 0, st, ga);
    HIP_TRY(hipGetLastError());
    float* da_cur = tr->da[0];
    const float* res = nullptr;
    for (int dir = 0; dir < 2; ++dir) {
        const std::string gp = p + "/bidirectional_rnn/" + dirs[dir] + "/gru_cell";
        const float *wg = W(gp + "/gates/kernel"), *wc = W(gp + "/candidate/kernel");
        float *gwg = G(gp + "/gates/kernel"), *gbg = G(gp + "/gates/bias"), *gwc = G(gp + "/candidate/kernel"), *gbc = G(gp + "/candidate/bias");
        if (err != T2_OK) return err;
        hipLaunchKernelGGL(gru_prev_kernel, pt_grid((long long)M * H), dim3(256), 0, st, tr->out, tr->tru[dir], tr->lens, tr->hprev, tr->rh, B, T, H, dir);
        // TF kernels: rows [inputs; state] (App. A2)
        T2_TRY(bwd_wgrad(&tr->ws, tr->a[nhw], H, H, tr->dzg[dir], 2 * H, 2 * H, M, gwg, 2 * H));
        T2_TRY(bwd_wgrad(&tr->ws, tr->hprev, H, H, tr->dzg[dir], 2 * H, 2 * H, M, gwg + (size_t)H * 2 * H, 2 * H));
        T2_TRY(col_sum(&tr->ws, tr->dzg[dir], M, 2 * H, gbg));
        T2_TRY(bwd_wgrad(&tr->ws, tr->a[nhw], H, H, tr->dzc[dir], H, H, M, gwc, H));
        T2_TRY(bwd_wgrad(&tr->ws, tr->rh, H, H, tr->dzc[dir], H, H, M, gwc + (size_t)H * H, H));
        T2_TRY(col_sum(&tr->ws, tr->dzc[dir], M, H, gbc));
        // d (GRU input) += dzg . Wg_x^T + dzc . Wc_x^T
        T2_TRY(pt_transpose(tr, wg, tr->wgxT[dir], H, 2 * H));
        T2_TRY(pt_transpose(tr, wc, tr->wcxT[dir], H, H));
        float* nxt = (da_cur == tr->da[0]) ? tr->da[1] : tr->da[0];
        T2_TRY(pt_dgrad_dense(tr, tr->dzg[dir], 2 * H, tr->wgxT[dir], H, da_cur, res));
        T2_TRY(pt_dgrad_dense(tr, tr->dzc[dir], H, tr->wcxT[dir], H, nxt, da_cur));
        res = nxt; da_cur = (nxt == tr->da[0]) ? tr->da[1] : tr->da[0];
    }
    const float* dcur = res;                                   // d a[nhw]
    float* dfree = da_cur;
    // highway layers, last first
    for (int i = nhw - 1; i >= 0; --i) {
        const std::string hp = p + "/highway_" + std::to_string(i + 1);
        const float *wh = W(hp + "/H/kernel"), *wt = W(hp + "/T/kernel");
        float *gwh = G(hp + "/H/kernel"), *gbh = G(hp + "/H/bias"), *gwt = G(hp + "/T/kernel"), *gbt = G(hp + "/T/bias");
        if (err != T2_OK) return err;
        hipLaunchKernelGGL(highway_bwd_kernel, pt_grid((long long)M * H), dim3(256), 0, st, dcur, tr->zh[i], tr->zt[i], tr->a[i], tr->dzh, tr->dzt, dfree, (long long)M * H);
        T2_TRY(bwd_wgrad(&tr->ws, tr->a[i], H, H, tr->dzh, H, H, M, gwh, H));
        T2_TRY(bwd_wgrad(&tr->ws, tr->a[i], H, H, tr->dzt, H, H, M, gwt, H));
        T2_TRY(col_sum(&tr->ws, tr->dzh, M, H, gbh));
        T2_TRY(col_sum(&tr->ws, tr->dzt, M, H, gbt));
        T2_TRY(pt_transpose(tr, wh, tr->whT[i], H, H));
        T2_TRY(pt_transpose(tr, wt, tr->wtT[i], H, H));
        float* other = const_cast<float*>(dcur);               // d a[i + 1] is consumed: its buffer takes the sum
        T2_TRY(pt_dgrad_dense(tr, tr->dzh, H, tr->whT[i], H, other, dfree));
        T2_TRY(pt_dgrad_dense(tr, tr->dzt, H, tr->wtT[i], H, dfree, other));
        dcur = dfree; dfree = other;
    }
    // adjustment layer -> d (p2 + x); the residual branch continues into d x at the end (want_dx)
    const float* d80 = nullptr;
    if (c.has_adj) {
        const float* wa = W(p + "/adjustment/kernel");
        float *gwa = G(p + "/adjustment/kernel"), *gba = G(p + "/adjustment/bias");
        if (err != T2_OK) return err;
        T2_TRY(bwd_wgrad(&tr->ws, tr->hw0, P2, P2, dcur, H, H, M, gwa, H));
        T2_TRY(col_sum(&tr->ws, dcur, M, H, gba));
        T2_TRY(pt_transpose(tr, wa, tr->wadjT, P2, H));
        T2_TRY(pt_dgrad_dense(tr, dcur, H, tr->wadjT, P2, tr->d80a, nullptr));
        d80 = tr->d80a;
    } else {
        d80 = dcur;
    }
    // proj2: batch norm (identity activation), convolution
    {
        const float* gam = W(p + "/proj2/batch_normalization/gamma");
        const float* wk = W(p + "/proj2/kernel");
        float *gg = G(p + "/proj2/batch_normalization/gamma"), *gb = G(p + "/proj2/batch_normalization/beta"), *gk = G(p + "/proj2/kernel");
        if (err != T2_OK) return err;
        T2_TRY(pt_bn_backward(tr, d80, nullptr, tr->p2raw, 2, P2, gam, tr->d80b));
        HIP_TRY(hipMemcpyAsync(gg, tr->dgam[2], (size_t)P2 * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(gb, tr->dbet[2], (size_t)P2 * 4, hipMemcpyDeviceToDevice, st));
        T2_TRY(pt_conv_wgrad(tr, tr->p1, P1, tr->d80b, P2, 3, gk));
        hipLaunchKernelGGL(conv_flip_kernel, pt_grid(3LL * P1 * P2), dim3(256), 0, st, tr->proj2T, wk, 3, P1, P2);
        ConvLayer L{};
        L.w = tr->proj2T; L.k = 3; L.cin = P2; L.cout = P1;
        T2_TRY(launch_conv(b, L, tr->d80b, P2, tr->dp1a, P1, M, T, 0, nullptr, 0));
    }
    // proj1: relu, batch norm, convolution
    {
        const float* gam = W(p + "/proj1/batch_normalization/gamma");
        const float* wk = W(p + "/proj1/kernel");
        float *gg = G(p + "/proj1/batch_normalization/gamma"), *gb = G(p + "/proj1/batch_normalization/beta"), *gk = G(p + "/proj1/kernel");
        if (err != T2_OK) return err;
        T2_TRY(pt_bn_backward(tr, tr->dp1a, tr->p1, tr->p1raw, 1, P1, gam, tr->dp1b));
        HIP_TRY(hipMemcpyAsync(gg, tr->dgam[1], (size_t)P1 * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(gb, tr->dbet[1], (size_t)P1 * 4, hipMemcpyDeviceToDevice, st));
        T2_TRY(pt_conv_wgrad(tr, tr->pooled, KC, tr->dp1b, P1, 3, gk));
        hipLaunchKernelGGL(conv_flip_kernel, pt_grid(3LL * KC * P1), dim3(256), 0, st, tr->proj1T, wk, 3, KC, P1);
        ConvLayer L{};
        L.w = tr->proj1T; L.k = 3; L.cin = P1; L.cout = KC;
        T2_TRY(launch_conv(b, L, tr->dp1b, P1, tr->dkc_a, KC, M, T, 0, nullptr, 0));
    }
    // max-pool, then the bank: relu, batch norm over the K*C channels, ONE weight-gradient GEMM scattered to the K kernels
    hipLaunchKernelGGL(maxpool2_bwd_kernel, pt_grid((long long)M * KC), dim3(256), 0, st, tr->bank_act, tr->dkc_a, tr->dkc_b, (long long)M, T, KC);
    {
        T2_TRY(pt_bn_backward(tr, tr->dkc_b, tr->bank_act, tr->bank_raw, 0, KC, tr->gam_u, tr->dkc_a));
        T2_TRY(pt_conv_wgrad(tr, tr->x0, cin, tr->dkc_a, KC, K, tr->dwbank));
        BankDst dk{}, dg{}, db{};
        for (int k = 1; k <= K; ++k) {
            const std::string q = p + "/conv1d_K" + std::to_string(k);
            dk.p[k - 1] = G(q + "/kernel"); dg.p[k - 1] = G(q + "/batch_normalization/gamma"); db.p[k - 1] = G(q + "/batch_normalization/beta");
        }
        if (err != T2_OK) return err;
        hipLaunchKernelGGL(bank_unpack_kernel, pt_grid((long long)K * cin * KC), dim3(256), 0, st, tr->dwbank, dk, K, cin, C);
        hipLaunchKernelGGL(bank_vec_scatter_kernel, dim3(ceil_div(KC, 256)), dim3(256), 0, st, tr->dgam[0], dg, K, C);
        hipLaunchKernelGGL(bank_vec_scatter_kernel, dim3(ceil_div(KC, 256)), dim3(256), 0, st, tr->dbet[0], db, K, C);
    }
    if (want_dx) {
        // d x = d (p2 + x) + the bank's data gradient: the convolution of d (bank output) with the flipped unified kernel; for an
        // even width K the SAME padding is asymmetric (pad_l = (K - 1) / 2 in front), so the transposed convolution pads K - 1 - pad_l
        hipLaunchKernelGGL(conv_flip_kernel, pt_grid((long long)K * cin * KC), dim3(256), 0, st, tr->wbankF, tr->wbank, K, cin, KC);
        ConvLayer L{};
        L.w = tr->wbankF; L.k = K; L.cin = KC; L.cout = cin; L.pad_l = K - 1 - (K - 1) / 2;
        L.part = tr->dxpart; L.part_floats = (size_t)M * cin * 16;       // 70 output tiles x 2048 k-tiles at the reference sizes: split over gridDim.z
        T2_TRY(launch_conv(b, L, tr->dkc_a, KC, tr->dx0, cin, M, T, 0, d80, P2));
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// UPDATE_OPS of tf.layers.batch_normalization (models.py:226-228 runs them with the train op): the moving statistics of the ten
// batch-norm layers in flat_params, from the batch statistics of the last forward
extern "C" int t2v1_postnet_trainer_update_moving_stats(t2v1_ptrainer* tr, float* flat_params, float momentum) {
    if (!tr || !flat_params) return fail(T2_EINVAL, "null argument");
    if (!tr->have_stats) return fail(T2_ESTATE, "t2v1_postnet_train_forward has not run");
    hipStream_t st = tr->m->base.ctx->stream;
    const CbhgW& c = *tr->c;
    const int M = tr->B * tr->T;
    const float bessel = M > 1 ? (float)M / (float)(M - 1) : 1.0f;
    auto upd = [&](const std::string& q, const float* mean, const float* var, int C) -> int {
        auto im = tr->ws.offsets.find(q + "/batch_normalization/moving_mean"), iv = tr->ws.offsets.find(q + "/batch_normalization/moving_variance");
        if (im == tr->ws.offsets.end() || iv == tr->ws.offsets.end()) return fail(T2_ESTATE, "no offset bound for the moving statistics of '%s'", q.c_str());
        hipLaunchKernelGGL(bn_moving_update_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, st, flat_params + im->second, flat_params + iv->second, mean, var, momentum, bessel, C);
        return T2_OK;
    };
    const std::string p = tr->pfx;
    for (int k = 1; k <= c.K; ++k) T2_TRY(upd(p + "/conv1d_K" + std::to_string(k), tr->mean[0] + (size_t)(k - 1) * c.C, tr->var[0] + (size_t)(k - 1) * c.C, c.C));
    T2_TRY(upd(p + "/proj1", tr->mean[1], tr->var[1], c.P1));
    T2_TRY(upd(p + "/proj2", tr->mean[2], tr->var[2], c.P2));
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// The PostNet's inference copies (t2v1_postnet, TacotronV1PostNetModel PREDICT / EVAL) rebuilt on the device from the flat
// parameter buffer: conv kernels, folded batch norm, concatenated highway / GRU operands, the linear layer.
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows, int cols) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)rows * cols; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), cc = (int)(i - (long long)r * cols);
        dst[(size_t)r * ldd + cc] = src[(size_t)r * lds + cc];
    }
}
extern "C" int t2v1_postnet_trainer_export(t2v1_ptrainer* tr, const float* flat_params) {
    if (!tr || !flat_params) return fail(T2_EINVAL, "null argument");
    t2v1_model* m = tr->m;
    HIP_TRY(hipSetDevice(m->base.ctx->device));
    hipStream_t st = m->base.ctx->stream;
    CbhgW& c = tr->which == 1 ? m->post_cbhg : m->enc_cbhg;
    const int H = c.half;
    tr->params = flat_params;
    int err = T2_OK;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    const std::string p = tr->pfx;
    auto conv = [&](ConvLayer& L, const std::string& q) -> int {
        const float *k = W(q + "/kernel"), *g = W(q + "/batch_normalization/gamma"), *be = W(q + "/batch_normalization/beta"),
                    *mu = W(q + "/batch_normalization/moving_mean"), *va = W(q + "/batch_normalization/moving_variance");
        if (err != T2_OK) return err;
        HIP_TRY(hipMemcpyAsync(L.w, k, (size_t)L.k * L.cin * L.cout * 4, hipMemcpyDeviceToDevice, st));
        conv_refresh_wt(st, L);
        hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(L.cout, 256)), dim3(256), 0, st, g, be, mu, va, m->base.d.bn_eps, L.scale, L.shift, L.cout);
        return T2_OK;
    };
    for (int k = 1; k <= c.K; ++k) T2_TRY(conv(c.bank[k - 1], p + "/conv1d_K" + std::to_string(k)));
    T2_TRY(v1_bank_unify(&m->base, &c));
    T2_TRY(conv(c.proj1, p + "/proj1"));
    T2_TRY(conv(c.proj2, p + "/proj2"));
    auto cp2 = [&](const float* src, int lds, float* dst, int ldd, int rows, int cols) {
        hipLaunchKernelGGL(copy2d_kernel, pt_grid((long long)rows * cols), dim3(256), 0, st, src, lds, dst, ldd, rows, cols);
    };
    if (c.has_adj) {
        const float *k = W(p + "/adjustment/kernel"), *bi = W(p + "/adjustment/bias");
        if (err != T2_OK) return err;
        cp2(k, H, c.adj.w, H, c.P2, H); cp2(bi, H, c.adj.shift, H, 1, H);
    }
    for (size_t i = 0; i < c.highway.size(); ++i) {
        const std::string hp = p + "/highway_" + std::to_string(i + 1);
        const float *wh = W(hp + "/H/kernel"), *wt = W(hp + "/T/kernel"), *bh = W(hp + "/H/bias"), *bt = W(hp + "/T/bias");
        if (err != T2_OK) return err;
        cp2(wh, H, c.highway[i].w, 2 * H, H, H); cp2(wt, H, c.highway[i].w + H, 2 * H, H, H);
        cp2(bh, H, c.highway[i].shift, 2 * H, 1, H); cp2(bt, H, c.highway[i].shift + H, 2 * H, 1, H);
    }
    const char* dirs[2] = {"fw", "bw"};
    for (int dir = 0; dir < 2; ++dir) {
        const std::string gp = p + "/bidirectional_rnn/" + dirs[dir] + "/gru_cell";
        const float *wg = W(gp + "/gates/kernel"), *bg = W(gp + "/gates/bias"), *wc = W(gp + "/candidate/kernel"), *bc = W(gp + "/candidate/bias");
        if (err != T2_OK) return err;
        cp2(wg, 2 * H, c.gru_x[dir].w, 3 * H, H, 2 * H); cp2(wc, H, c.gru_x[dir].w + 2 * H, 3 * H, H, H);
        cp2(bg, 2 * H, c.gru_x[dir].shift, 3 * H, 1, 2 * H); cp2(bc, H, c.gru_x[dir].shift + 2 * H, 3 * H, 1, H);
        cp2(wg + (size_t)H * 2 * H, 2 * H, c.gru_hg[dir], 2 * H, H, 2 * H); cp2(wc + (size_t)H * H, H, c.gru_hc[dir], H, H, H);
    }
    if (tr->which == 1) {
        const float *k = W("v1/postnet/linear/kernel"), *bi = W("v1/postnet/linear/bias");
        if (err != T2_OK) return err;
        const int F = m->d.num_freq;
        cp2(k, F, m->post_linear.w, F, 2 * H, F); cp2(bi, F, m->post_linear.shift, F, 1, F);
    } else {
        const float* emb = W("embedding");
        if (err != T2_OK) return err;
        HIP_TRY(hipMemcpyAsync(m->embedding, emb, (size_t)m->d.num_symbols * m->d.embedding_dim * 4, hipMemcpyDeviceToDevice, st));
        int in = m->d.embedding_dim;
        for (int i = 0; i < 2; ++i) {
            const std::string q = "v1/encoder/prenet_" + std::to_string(i) + "/dense";
            const float *k = W(q + "/kernel"), *bi = W(q + "/bias");
            if (err != T2_OK) return err;
            const int u = m->d.enc_prenet_units[i];
            cp2(k, u, m->enc_prenet[i].w, u, in, u); cp2(bi, u, m->enc_prenet[i].shift, u, 1, u);
            in = u;
        }
    }
    m->gru_packed_stale = true;           // the launch-per-step GRU operands (option gruseq = 0) still hold the old values
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// ==========================================================================================================================
// EncoderV1(is_training=True) (tacotron_v1.py:18-49) with the embedding in front (models.py:40-47): Embedding -> two PreNets
// (Dense + relu + dropout, modules.py:80-91; masks from the counter-based generator, streams T2_STREAM_V1_ENC_PRENET + i, step 0,
// row = batch position, column = unit) -> CBHG with batch-statistics batch norm -- forward with a tape, and the backward pass
// from the gradient of the encoder output down to the embedding table: the encoder half of SingleSpeakerTacotronV1Model's
// training step (the decoder half: v1_dec_train.h).
// ==========================================================================================================================
#define T2_STREAM_V1_ENC_PRENET 50

__global__ __launch_bounds__(256) void v1_dropout_kernel(float* __restrict__ x, float rate, uint32_t seed, int stream, long long M, int C) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / C;
        x[i] = dropout_train(x[i], rate, t2_u01(seed, stream, 0, (uint32_t)m, (uint32_t)(i - m * C)));
    }
}
// through the dropout and the relu of a PreNet: out is what the layer handed on (0 where either of them cut)
__global__ __launch_bounds__(256) void v1_relu_dropout_bwd_kernel(const float* dy, const float* __restrict__ out, float inv_keep, float* dz, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dz[i] = out[i] != 0.f ? dy[i] * inv_keep : 0.f;
}

// ids host [B][T_in] (range-checked: T2_ERANGE), lens host [B] or null (the V1 model_fn passes none, models.py:46-47),
// memory_out dev [B][T_in][cbhg_out_units]
static int encoder_train_forward_impl(t2v1_ptrainer* tr, const float* flat_params, const int64_t* ids, const float* emb_in, const int64_t* lens,
                                      float drop_rate, uint32_t seed, float* memory_out) {
    if (!tr || !flat_params || (!ids && !emb_in) || !memory_out) return fail(T2_EINVAL, "null argument");
    if (tr->which != 0) return fail(T2_EINVAL, "not an encoder trainer");
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return fail(T2_EINVAL, "drop_rate %g outside [0, 1)", (double)drop_rate);
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const t2v1_dims& d = m->d;
    const int B = tr->B, T = tr->T, M = B * T, H = tr->c->half;
    tr->params = flat_params; tr->have_forward = false; tr->have_stats = false; tr->drop_rate = drop_rate; tr->seed = seed;
    int err = T2_OK;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    tr->have_ids = ids != nullptr;
    if (ids) {
        std::vector<int> h_ids(M);
        for (int i = 0; i < M; ++i) {
            if (ids[i] < 0 || ids[i] >= d.num_symbols) return fail(T2_ERANGE, "embedding id %lld outside [0, %d) (modules.py:65-66)", (long long)ids[i], d.num_symbols);
            h_ids[i] = (int)ids[i];
        }
        HIP_TRY(hipMemcpyAsync(tr->ids, h_ids.data(), (size_t)M * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));                    // h_ids is a local
        const float* table = W("embedding");
        if (err != T2_OK) return err;
        hipLaunchKernelGGL(embedding_kernel, dim3(std::min(1024, ceil_div(M * d.embedding_dim, 256))), dim3(256), 0, st, tr->ids, table, tr->emb, M, d.embedding_dim);
    } else {
        HIP_TRY(hipMemcpyAsync(tr->emb, emb_in, (size_t)M * d.embedding_dim * 4, hipMemcpyDeviceToDevice, st));
    }
    T2_TRY(v1_lens_to_dev(m, lens, B, T, 0, tr->lens));
    const float* cur = tr->emb;
    int in = d.embedding_dim;
    for (int i = 0; i < 2; ++i) {
        const std::string q = "v1/encoder/prenet_" + std::to_string(i) + "/dense";
        ConvLayer L{};
        L.w = const_cast<float*>(W(q + "/kernel")); L.shift = const_cast<float*>(W(q + "/bias")); L.k = 1; L.cin = in; L.cout = d.enc_prenet_units[i];
        if (err != T2_OK) return err;
        T2_TRY(launch_conv(b, L, cur, in, tr->pre[i], L.cout, M, M, 1, nullptr, 0));
        if (drop_rate > 0.f)
            hipLaunchKernelGGL(v1_dropout_kernel, pt_grid((long long)M * L.cout), dim3(256), 0, st, tr->pre[i], drop_rate, seed, T2_STREAM_V1_ENC_PRENET + i, (long long)M, L.cout);
        cur = tr->pre[i]; in = L.cout;
    }
    T2_TRY(cbhg_train_forward_impl(tr, tr->pre[1]));
    HIP_TRY(hipMemcpyAsync(memory_out, tr->out, (size_t)M * 2 * H * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    tr->have_forward = true; tr->have_stats = true;
    return T2_OK;
}
extern "C" int t2v1_encoder_train_forward(t2v1_ptrainer* tr, const float* flat_params, const int64_t* ids, const int64_t* lens, float drop_rate,
                                          uint32_t seed, float* memory_out) {
    if (!ids) return fail(T2_EINVAL, "ids is null");
    return encoder_train_forward_impl(tr, flat_params, ids, nullptr, lens, drop_rate, seed, memory_out);
}
// EncoderV1.call(is_training=True) on an embedded input emb dev [B][T_in][embedding_dim] (the block on its own, tacotron_v1.py:46-49)
extern "C" int t2v1_encoder_train_forward_embedded(t2v1_ptrainer* tr, const float* flat_params, const float* emb, const int64_t* lens, float drop_rate,
                                                   uint32_t seed, float* memory_out) {
    if (!emb) return fail(T2_EINVAL, "emb is null");
    return encoder_train_forward_impl(tr, flat_params, nullptr, emb, lens, drop_rate, seed, memory_out);
}

// dmemory dev [B][T_in][cbhg_out_units]: the gradient of the loss w.r.t. the encoder output.  Writes the gradients of the CBHG,
// the PreNets and (after a forward from ids) the embedding table into flat_grads at the bound offsets; demb_out (dev
// [B][T_in][embedding_dim], may be null): the gradient of the embedded input.
extern "C" int t2v1_encoder_train_backward(t2v1_ptrainer* tr, const float* dmemory, float* flat_grads, float* demb_out) {
    if (!tr || !dmemory || !flat_grads) return fail(T2_EINVAL, "null argument");
    if (tr->which != 0) return fail(T2_EINVAL, "not an encoder trainer");
    if (!tr->have_forward) return fail(T2_ESTATE, "t2v1_encoder_train_forward has not run");
    t2v1_model* m = tr->m;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    hipStream_t st = b->ctx->stream;
    const t2v1_dims& d = m->d;
    const int M = tr->B * tr->T, H = tr->c->half;
    HIP_TRY(hipMemcpyAsync(tr->dout, dmemory, (size_t)M * 2 * H * 4, hipMemcpyDeviceToDevice, st));
    T2_TRY(cbhg_train_backward_impl(tr, flat_grads, true));
    int err = T2_OK;
    auto W = [&](const std::string& n) { return pt_w(tr, n, &err); };
    auto G = [&](const std::string& n) -> float* {
        auto it = tr->ws.offsets.find(n);
        if (it == tr->ws.offsets.end()) { err = fail(T2_ESTATE, "no offset bound for '%s'", n.c_str()); return nullptr; }
        return flat_grads + it->second;
    };
    const float inv_keep = 1.0f / (1.0f - tr->drop_rate);
    const float* dy = tr->dx0;                                 // gradient of PreNet 1's output (= the CBHG's input)
    for (int i = 1; i >= 0; --i) {
        const std::string q = "v1/encoder/prenet_" + std::to_string(i) + "/dense";
        const int u = d.enc_prenet_units[i], in = i == 0 ? d.embedding_dim : d.enc_prenet_units[0];
        const float* x = i == 0 ? tr->emb : tr->pre[0];
        const float* wk = W(q + "/kernel");
        float *gk = G(q + "/kernel"), *gb = G(q + "/bias");
        if (err != T2_OK) return err;
        hipLaunchKernelGGL(v1_relu_dropout_bwd_kernel, pt_grid((long long)M * u), dim3(256), 0, st, dy, tr->pre[i], inv_keep, tr->dpre[i], (long long)M * u);
        T2_TRY(bwd_wgrad(&tr->ws, x, in, in, tr->dpre[i], u, u, M, gk, u));
        T2_TRY(col_sum(&tr->ws, tr->dpre[i], M, u, gb));
        T2_TRY(pt_transpose(tr, wk, tr->wpreT[i], in, u));
        if (i == 1) {
            T2_TRY(pt_dgrad_dense(tr, tr->dpre[1], u, tr->wpreT[1], in, tr->dpre[0], nullptr));
            dy = tr->dpre[0];                                  // masked in place by the next round
        } else {
            // d embedded rows -> the table: the rows of every symbol are added in row order (embedding_grad_kernel)
            T2_TRY(pt_dgrad_dense(tr, tr->dpre[0], u, tr->wpreT[0], in, tr->emb, nullptr));      // the embedded input is consumed: its buffer takes d emb
            if (demb_out) HIP_TRY(hipMemcpyAsync(demb_out, tr->emb, (size_t)M * in * 4, hipMemcpyDeviceToDevice, st));
            if (tr->have_ids) {
                float* ge = G("embedding");
                if (err != T2_OK) return err;
                hipLaunchKernelGGL(embedding_grad_kernel, dim3(d.num_symbols), dim3(256), 0, st, tr->ids, tr->emb, M, in, ge);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    tr->have_forward = false;
    return T2_OK;
}
