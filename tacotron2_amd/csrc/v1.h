// Tacotron v1 path (BASELINE.json configs[4]) on the kernels of
// kernels.h: CBHG (conv bank, max-pool, projections, highway, BiGRU), EncoderV1, the V1 decoder cell
// stack (PreNets -> GRU + plain Bahdanau attention -> OutputProjectionWrapper -> 2 residual GRUs ->
// mel / stop projections) and the mel -> linear PostNet.  Included by t2mi.hip.
#pragma once

struct GruPacked {
    Packed gates, cand;
    int H = 0;
};

struct CbhgW {
    std::vector<ConvLayer> bank;
    ConvLayer bank_u;                     // the K bank kernels as ONE convolution of width K (zero-padded taps, columns (k - 1) C + c): v1_bank_unify
    ConvLayer proj1, proj2, adj;
    bool has_adj = false;
    std::vector<ConvLayer> highway;       // one GEMM per layer: [Wh | Wt], shift = [bh | bt]
    GruPacked gru[2];
    // whole-sequence GRU (gru_seq_kernel, half == 128): input halves of both kernels as one GEMM operand [half][3 half] + biases,
    // recurrent halves in plain layout
    ConvLayer gru_x[2];
    float *gru_hg[2] = {nullptr, nullptr}, *gru_hc[2] = {nullptr, nullptr};
    int cin = 0, out_units = 0, half = 0, C = 0, K = 0, P1 = 0, P2 = 0;
};

struct t2v1_state;

struct t2v1_model {
    t2_model base;                        // ctx, named host weights, device allocations, bn_eps
    t2v1_dims d{};
    float* embedding = nullptr;
    ConvLayer enc_prenet[2];
    CbhgW enc_cbhg, post_cbhg;
    ConvLayer post_linear, memory_layer;
    Packed dec_prenet[2], query, outproj, proj;
    ConvLayer dec_prenet_plain[2], proj_plain;   // teacher-forced decodes: PreNets and projections of all steps as batched GEMMs
    int opt_tfbatch = 1;
    int opt_gruseq = 1;                   // CBHG BiGRU of 128 units per direction as one persistent launch (gru_seq_kernel)
    int opt_bank_unified = 1;             // CBHG inference: the conv bank as one width-K convolution when the batch is short
    int opt_skinny = 1;                   // training recurrence: per-step products on v1d_skinny_gemm_kernel (0: conv_gemm tiles)
    int opt_train_persist = 1;            // decoder trainer: each recurrence as one persistent launch (v1_dec_train.h)
    long long n_train_persist = 0;        // such launches so far (t2v1_model_get_stat "train_persist_launches")
    bool gru_packed_stale = false;        // a trainer exported new variables: only the whole-sequence GRU operands were rebuilt
    GruPacked att_gru, gru1, gru2;
    float* v_a = nullptr;
    float* zeros = nullptr;               // stands in for the location terms the plain Bahdanau score lacks
    void* ws = nullptr; size_t ws_bytes = 0;
    std::vector<t2v1_state*> states;
    int* h_ctl = nullptr;
    int opt_graph = 1, opt_chunk = 16;
    // teacher-forced decodes on the training forward's persistent launch (v1_dec_train.h, v1_teacher_via_trainer)
    int opt_tf_persist = 1;
    float* tf_flat = nullptr;                              // flat copy of the v1/decoder/ and v1/attention/ variables (TF layouts)
    std::map<std::string, long long> tf_offsets;
    std::vector<struct t2v1_dtrainer*> tf_trainers;
};
static int v1_teacher_via_trainer(t2v1_model* m, const float* memory, const int64_t* lens, const float* targets, int B, int T_in, int S, float* mel, float* stop,
                                  float* align, bool* done);
static void v1_tf_trainers_free(t2v1_model* m);

struct t2v1_state {
    t2v1_model* m = nullptr;
    int B = 0, T = 0, MB = 0, cap = 1;
    void *fixed = nullptr, *slab = nullptr, *outbuf = nullptr;
    size_t slab_bytes = 0;
    float *values = nullptr, *keys = nullptr;
    int* lens = nullptr;
    int* ctl = nullptr;
    float *att_state = nullptr, *alignments = nullptr, *ctx_plain = nullptr, *q = nullptr;
    float *u_a = nullptr, *u1 = nullptr, *u2 = nullptr;
    float *xf = nullptr, *xp[2] = {nullptr, nullptr};
    float *XA = nullptr, *XR = nullptr, *X1 = nullptr, *RH1 = nullptr, *X2 = nullptr, *RH2 = nullptr, *XY = nullptr;
    int XA_sz = 0, X1_sz = 0, KBe = 0, KBa = 0, KBd = 0;
    float *o_mel = nullptr, *o_stop = nullptr, *o_hist = nullptr, *i_targets = nullptr;
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    int gchunk[2] = {0, 0};
    // teacher-forced decodes (lazy): gathered inputs, PreNet outputs, per-step frag16 operands, GRU 2 outputs, projection GEMM output
    float *tf_buf = nullptr, *tf_xin = nullptr, *tf_x[2] = {nullptr, nullptr}, *tf_frag = nullptr, *y_all = nullptr, *proj_tmp = nullptr;
    long long tf_stride = 0;
};

// ---- weights ---------------------------------------------------------------------------------------
static int v1_dense_conv(t2_model* b, ConvLayer* L, const std::string& prefix, int cin, int cout, bool bias) {
    const HostWeight* w;
    T2_TRY(get_w(b, prefix + "/kernel", &w, {cin, cout}));
    L->k = 1; L->cin = cin; L->cout = cout;
    T2_TRY(upload(b, &L->w, w->data));
    if (bias) {
        const HostWeight* bb;
        T2_TRY(get_w(b, prefix + "/bias", &bb, {cout}));
        T2_TRY(upload(b, &L->shift, bb->data));
    }
    return T2_OK;
}

static int v1_pack_dense(t2_model* b, Packed* p, const std::string& prefix, const std::vector<std::pair<int, int>>& segs, int rows,
                         int cols, bool bias) {
    const HostWeight *w, *bb = nullptr;
    T2_TRY(get_w(b, prefix + "/kernel", &w, {rows, cols}));
    if (bias) T2_TRY(get_w(b, prefix + "/bias", &bb, {cols}));
    return pack_skinny(b, p, w->data.data(), cols, segs, 0, cols, bb ? bb->data.data() : nullptr);
}

static int v1_pack_gru(t2_model* b, GruPacked* g, const std::string& prefix, const std::vector<std::pair<int, int>>& segs, int rows, int H) {
    g->H = H;
    T2_TRY(v1_pack_dense(b, &g->gates, prefix + "/gates", segs, rows, 2 * H, true));
    T2_TRY(v1_pack_dense(b, &g->cand, prefix + "/candidate", segs, rows, H, true));
    return T2_OK;
}

static int v1_make_cbhg(t2_model* b, CbhgW* c, const std::string& p, int cin, int out_units, int C, int K, int P1, int P2, int nhw) {
    c->cin = cin; c->out_units = out_units; c->half = out_units / 2; c->C = C; c->K = K; c->P1 = P1; c->P2 = P2;
    c->bank.resize(K);
    for (int k = 1; k <= K; ++k) T2_TRY(make_conv(b, &c->bank[k - 1], p + "/conv1d_K" + std::to_string(k), k, cin, C, true));
    T2_TRY(make_conv(b, &c->proj1, p + "/proj1", 3, C * K, P1, true));
    T2_TRY(make_conv(b, &c->proj2, p + "/proj2", 3, P1, P2, true));
    if (P2 != cin) return fail(T2_EINVAL, "CBHG residual needs projection2_out_channels == input width (modules.py:214)");
    const int half = c->half;
    c->has_adj = P2 != half;
    if (c->has_adj) T2_TRY(v1_dense_conv(b, &c->adj, p + "/adjustment", P2, half, true));
    c->highway.resize(nhw);
    for (int i = 1; i <= nhw; ++i) {
        const std::string hp = p + "/highway_" + std::to_string(i);
        const HostWeight *wh, *wt, *bh, *bt;
        T2_TRY(get_w(b, hp + "/H/kernel", &wh, {half, half}));
        T2_TRY(get_w(b, hp + "/T/kernel", &wt, {half, half}));
        T2_TRY(get_w(b, hp + "/H/bias", &bh, {half}));
        T2_TRY(get_w(b, hp + "/T/bias", &bt, {half}));
        std::vector<float> w((size_t)half * 2 * half), sh(2 * half);
        for (int r = 0; r < half; ++r)
            for (int cc = 0; cc < half; ++cc) {
                w[(size_t)r * 2 * half + cc] = wh->data[(size_t)r * half + cc];
                w[(size_t)r * 2 * half + half + cc] = wt->data[(size_t)r * half + cc];
            }
        for (int cc = 0; cc < half; ++cc) { sh[cc] = bh->data[cc]; sh[half + cc] = bt->data[cc]; }
        ConvLayer& L = c->highway[i - 1];
        L.k = 1; L.cin = half; L.cout = 2 * half;
        T2_TRY(upload(b, &L.w, w));
        T2_TRY(upload(b, &L.shift, sh));
    }
    const char* dirs[2] = {"fw", "bw"};
    for (int dir = 0; dir < 2; ++dir) {
        T2_TRY(v1_pack_gru(b, &c->gru[dir], p + "/bidirectional_rnn/" + dirs[dir] + "/gru_cell", {{0, half}, {half, half}}, 2 * half, half));
        {
            const std::string gp = p + "/bidirectional_rnn/" + dirs[dir] + "/gru_cell";
            const HostWeight *wg, *bg, *wcd, *bc;
            T2_TRY(get_w(b, gp + "/gates/kernel", &wg, {2 * half, 2 * half}));
            T2_TRY(get_w(b, gp + "/gates/bias", &bg, {2 * half}));
            T2_TRY(get_w(b, gp + "/candidate/kernel", &wcd, {2 * half, half}));
            T2_TRY(get_w(b, gp + "/candidate/bias", &bc, {half}));
            std::vector<float> wx((size_t)half * 3 * half), bx(3 * half), hg((size_t)half * 2 * half), hc((size_t)half * half);
            for (int k = 0; k < half; ++k) {
                for (int n = 0; n < 2 * half; ++n) {
                    wx[(size_t)k * 3 * half + n] = wg->data[(size_t)k * 2 * half + n];
                    hg[(size_t)k * 2 * half + n] = wg->data[(size_t)(half + k) * 2 * half + n];
                }
                for (int n = 0; n < half; ++n) {
                    wx[(size_t)k * 3 * half + 2 * half + n] = wcd->data[(size_t)k * half + n];
                    hc[(size_t)k * half + n] = wcd->data[(size_t)(half + k) * half + n];
                }
            }
            for (int n = 0; n < 2 * half; ++n) bx[n] = bg->data[n];
            for (int n = 0; n < half; ++n) bx[2 * half + n] = bc->data[n];
            c->gru_x[dir].k = 1; c->gru_x[dir].cin = half; c->gru_x[dir].cout = 3 * half;
            T2_TRY(upload(b, &c->gru_x[dir].w, wx));
            T2_TRY(upload(b, &c->gru_x[dir].shift, bx));
            T2_TRY(upload(b, &c->gru_hg[dir], hg));
            T2_TRY(upload(b, &c->gru_hc[dir], hc));
        }
    }
    return T2_OK;
}

extern "C" int t2v1_model_create(t2_ctx* ctx, const t2v1_dims* dims, t2v1_model** out) {
    if (!ctx || !dims || !out) return fail(T2_EINVAL, "null argument");
    const t2v1_dims& d = *dims;
    if (d.n_feed_frame < 1 || d.n_feed_frame > d.outputs_per_step) return fail(T2_EINVAL, "n_feed_frame must be in 1..outputs_per_step (helpers.py:55)");
    if (d.cbhg_out_units % 2 != 0 || d.post_cbhg_out_units % 2 != 0) return fail(T2_EINVAL, "CBHG out_units must be even (modules.py:169)");
    if (d.enc_prenet_units[0] != d.embedding_dim) return fail(T2_EINVAL, "prenet_out_units[0] must equal the embedding width (tacotron_v1.py:41-44)");
    if (d.decoder_out_units < 1) return fail(T2_EINVAL, "Parameter output_size must be > 0: %d. (rnn_wrappers.py:155-156)", d.decoder_out_units);
    const int dims_pos[] = {d.num_symbols, d.embedding_dim, d.enc_prenet_units[1], d.cbhg_out_units, d.conv_channels, d.max_filter_width,
                            d.proj1_channels, d.proj2_channels, d.dec_prenet_units[0], d.dec_prenet_units[1], d.attention_out_units,
                            d.num_mels, d.outputs_per_step, d.post_cbhg_out_units, d.post_conv_channels, d.post_max_filter_width,
                            d.post_proj1_channels, d.post_proj2_channels, d.num_freq};
    for (int v : dims_pos) if (v < 1) return fail(T2_EINVAL, "every dimension must be >= 1");
    if (d.num_highway < 0 || d.post_num_highway < 0) return fail(T2_EINVAL, "num_highway must be >= 0");
    if (d.attention_out_units > 256) return fail(T2_EINVAL, "attention units > 256 not supported by the attention kernel");
    t2v1_model* m = new t2v1_model();
    m->base.ctx = ctx;
    m->base.d.bn_eps = d.bn_eps;
    m->d = d;
    *out = m;
    return T2_OK;
}

extern "C" int t2v1_model_set_weight(t2v1_model* m, const char* name, const float* host_ptr, const int64_t* shape, int rank) {
    if (!m) return fail(T2_EINVAL, "null model");
    return t2_model_set_weight(&m->base, name, host_ptr, shape, rank);
}

static void v1_state_free(t2v1_state* s);
static int v1_bank_unify(t2_model* b, CbhgW* c);       // v1_train.h (next to the trainer's bank kernels)

extern "C" int t2v1_model_set_option(t2v1_model* m, const char* name, int value) {
    if (!m || !name) return fail(T2_EINVAL, "null argument");
    if (!strcmp(name, "graph")) m->opt_graph = value;
    else if (!strcmp(name, "chunk")) { if (value < 1 || value > 64) return fail(T2_EINVAL, "chunk 1..64"); m->opt_chunk = value; }
    else if (!strcmp(name, "skinny")) m->opt_skinny = value;
    else if (!strcmp(name, "bank_unified")) m->opt_bank_unified = value;
    else if (!strcmp(name, "tf_persist")) m->opt_tf_persist = value;
    else if (!strcmp(name, "train_persist")) { m->opt_train_persist = value; m->base.persist_holdoff = 0; }
    else if (!strcmp(name, "inject_timeout")) m->base.opt_inject_timeout = value;
    else if (!strcmp(name, "gruseq")) {
        if (!value && m->gru_packed_stale) return fail(T2_ESTATE, "gruseq = 0 after a trainer export: the launch-per-step GRU operands hold the variables of before the training");
        m->opt_gruseq = value;
    }
    else if (!strcmp(name, "tfbatch")) {      // teacher-forced decodes: batched PreNets / projections; states are shaped by it
        if (value != m->opt_tfbatch) {
            HIP_TRY(hipStreamSynchronize(m->base.ctx->stream));
            for (t2v1_state* s : m->states) v1_state_free(s);
            m->states.clear();
        }
        m->opt_tfbatch = value;
    }
    else return fail(T2_EINVAL, "unknown option %s", name);
    return T2_OK;
}

extern "C" int t2v1_model_finalize(t2v1_model* m) {
    if (!m) return fail(T2_EINVAL, "null model");
    if (m->base.finalized) return T2_OK;
    t2_model* b = &m->base;
    HIP_TRY(hipSetDevice(b->ctx->device));
    const t2v1_dims& d = m->d;
    const HostWeight* w;
    T2_TRY(get_w(b, "embedding", &w, {d.num_symbols, d.embedding_dim}));
    T2_TRY(upload(b, &m->embedding, w->data));
    T2_TRY(v1_dense_conv(b, &m->enc_prenet[0], "v1/encoder/prenet_0/dense", d.embedding_dim, d.enc_prenet_units[0], true));
    T2_TRY(v1_dense_conv(b, &m->enc_prenet[1], "v1/encoder/prenet_1/dense", d.enc_prenet_units[0], d.enc_prenet_units[1], true));
    T2_TRY(v1_make_cbhg(b, &m->enc_cbhg, "v1/encoder/cbhg", d.enc_prenet_units[1], d.cbhg_out_units, d.conv_channels, d.max_filter_width,
                        d.proj1_channels, d.proj2_channels, d.num_highway));
    T2_TRY(v1_make_cbhg(b, &m->post_cbhg, "v1/postnet/cbhg", d.num_mels, d.post_cbhg_out_units, d.post_conv_channels,
                        d.post_max_filter_width, d.post_proj1_channels, d.post_proj2_channels, d.post_num_highway));
    T2_TRY(v1_dense_conv(b, &m->post_linear, "v1/postnet/linear", d.post_cbhg_out_units, d.num_freq, true));
    T2_TRY(v1_bank_unify(b, &m->enc_cbhg));
    T2_TRY(v1_bank_unify(b, &m->post_cbhg));
    const int E = d.cbhg_out_units, A = d.attention_out_units, D = d.decoder_out_units, M = d.num_mels, r = d.outputs_per_step;
    const int P0 = d.dec_prenet_units[0], P1 = d.dec_prenet_units[1];
    T2_TRY(v1_dense_conv(b, &m->memory_layer, "v1/attention/memory_layer", E, A, false));
    T2_TRY(v1_pack_dense(b, &m->dec_prenet[0], "v1/decoder/prenet_0/dense", {{0, M * d.n_feed_frame}}, M * d.n_feed_frame, P0, true));
    T2_TRY(v1_pack_dense(b, &m->dec_prenet[1], "v1/decoder/prenet_1/dense", {{0, P0}}, P0, P1, true));
    T2_TRY(v1_dense_conv(b, &m->dec_prenet_plain[0], "v1/decoder/prenet_0/dense", M * d.n_feed_frame, P0, true));
    T2_TRY(v1_dense_conv(b, &m->dec_prenet_plain[1], "v1/decoder/prenet_1/dense", P0, P1, true));
    T2_TRY(v1_pack_gru(b, &m->att_gru, "v1/decoder/attention_gru", {{0, P1}, {P1, E}, {P1 + E, A}}, P1 + E + A, A));
    T2_TRY(v1_pack_dense(b, &m->query, "v1/attention/query_layer", {{0, A}}, A, A, false));
    // OutputProjectionWrapper input = concat(cell output, attention) (rnn_wrappers.py:76-78); the operand buffer holds [ctx | h]
    T2_TRY(v1_pack_dense(b, &m->outproj, "v1/decoder/output_projection", {{A, E}, {0, A}}, A + E, D, true));
    T2_TRY(v1_pack_gru(b, &m->gru1, "v1/decoder/gru_1", {{0, D}, {D, D}}, 2 * D, D));
    T2_TRY(v1_pack_gru(b, &m->gru2, "v1/decoder/gru_2", {{0, D}, {D, D}}, 2 * D, D));
    {   // mel and stop projections as one [D, M*r + 1] operand, like the V2 path
        const HostWeight *wm, *bm, *wst, *bst;
        T2_TRY(get_w(b, "v1/decoder/out_projection/kernel", &wm, {D, M * r}));
        T2_TRY(get_w(b, "v1/decoder/out_projection/bias", &bm, {M * r}));
        T2_TRY(get_w(b, "v1/decoder/stop_token_projection/kernel", &wst, {D, 1}));
        T2_TRY(get_w(b, "v1/decoder/stop_token_projection/bias", &bst, {1}));
        const int N = M * r + 1;
        std::vector<float> wcat((size_t)D * N), bcat(N);
        for (int k = 0; k < D; ++k) {
            for (int c = 0; c < M * r; ++c) wcat[(size_t)k * N + c] = wm->data[(size_t)k * M * r + c];
            wcat[(size_t)k * N + M * r] = wst->data[k];
        }
        for (int c = 0; c < M * r; ++c) bcat[c] = bm->data[c];
        bcat[M * r] = bst->data[0];
        T2_TRY(pack_skinny(b, &m->proj, wcat.data(), N, {{0, D}}, 0, N, bcat.data()));
        const int Np = (N + 3) & ~3;             // plain copy, columns padded to a multiple of 4 (batched projection)
        std::vector<float> wp((size_t)D * Np, 0.f), bp(Np, 0.f);
        for (int k = 0; k < D; ++k) for (int c = 0; c < N; ++c) wp[(size_t)k * Np + c] = wcat[(size_t)k * N + c];
        for (int c = 0; c < N; ++c) bp[c] = bcat[c];
        m->proj_plain.k = 1; m->proj_plain.cin = D; m->proj_plain.cout = Np;
        T2_TRY(upload(b, &m->proj_plain.w, wp));
        T2_TRY(upload(b, &m->proj_plain.shift, bp));
    }
    T2_TRY(get_w(b, "v1/attention/attention_v", &w, {A}));
    T2_TRY(upload(b, &m->v_a, w->data));
    T2_TRY(upload(b, &m->zeros, std::vector<float>((size_t)std::max(A, 16) + 16, 0.f)));
    HIP_TRY(hipHostMalloc((void**)&m->h_ctl, 8 * sizeof(int), hipHostMallocDefault));
    T2_TRY(set_attention_func_attributes());
    {   // the decoder's variables once more as one flat buffer in TF layouts: what the training forward reads (teacher-forced decodes)
        std::vector<float> flat;
        for (const auto& kv : b->hw) {
            if (kv.first.rfind("v1/decoder/", 0) != 0 && kv.first.rfind("v1/attention/", 0) != 0) continue;
            m->tf_offsets[kv.first] = (long long)flat.size();
            flat.insert(flat.end(), kv.second.data.begin(), kv.second.data.end());
            flat.resize((flat.size() + 3) & ~(size_t)3, 0.f);
        }
        T2_TRY(upload(b, &m->tf_flat, flat));
    }
    b->hw.clear();
    b->finalized = true;
    return T2_OK;
}

static void v1_state_free(t2v1_state* s) {
    if (!s) return;
    for (int i = 0; i < 2; ++i) if (s->gexec[i]) hipGraphExecDestroy(s->gexec[i]);
    if (s->fixed) hipFree(s->fixed);
    if (s->slab) hipFree(s->slab);
    if (s->outbuf) hipFree(s->outbuf);
    if (s->tf_buf) hipFree(s->tf_buf);
    delete s;
}

extern "C" int t2v1_model_destroy(t2v1_model* m) {
    if (!m) return T2_OK;
    (void)hipSetDevice(m->base.ctx->device);
    (void)hipStreamSynchronize(m->base.ctx->stream);
    for (t2v1_state* s : m->states) v1_state_free(s);
    v1_tf_trainers_free(m);
    for (void* p : m->base.allocs) (void)hipFree(p);
    if (m->ws) (void)hipFree(m->ws);
    if (m->h_ctl) (void)hipHostFree(m->h_ctl);
    delete m;
    return T2_OK;
}

// ---- GRU launches ------------------------------------------------------------------------------------
template <int PHASE>
static inline void launch_gru(int MB, int n_wg, hipStream_t st, const SkinnyCore& c, const GruEpi& e) {
    const dim3 grid(n_wg), blk(T2_SKINNY_THREADS);
    if (MB == 1) hipLaunchKernelGGL((gru_step_kernel<1, PHASE>), grid, blk, 0, st, c, e);
    else if (MB == 2) hipLaunchKernelGGL((gru_step_kernel<2, PHASE>), grid, blk, 0, st, c, e);
    else hipLaunchKernelGGL((gru_step_kernel<4, PHASE>), grid, blk, 0, st, c, e);
}

static SkinnyCore v1_core(const Packed& p, const int* ctl, int step_off, int t_hint, int B) {
    SkinnyCore c{};
    c.w[0] = c.w[1] = p.w; c.bias[0] = c.bias[1] = p.bias;
    c.kb0 = 0; c.kbtot = p.kbtot; c.kb_w0 = 0; c.kb_full = p.kbtot; c.ctl = ctl; c.step_off = step_off; c.t_hint = t_hint; c.B = B;
    return c;
}

// One GRU step = gates launch + candidate launch.  x: operand of the input part (kbx k-blocks); hbuf: 2-slot state
// buffer with h at koff_h (may be the same buffer as x); rh: single buffer that receives r*h at koff_rh; when the
// input part of the candidate operand lives next to r*h (attention GRU: [ctx | r*h]) x2 / kbx2 name the extra segment.
struct GruStep {
    const GruPacked* g;
    FragBuf x; int kbx;               // first operand segment (k-blocks [0, kbx)); kbx = 0: none
    FragBuf gates_rest;               // second segment of the gates operand (contains h)
    FragBuf cand_rest;                // second segment of the candidate operand (contains r*h)
    FragBuf hprev, rh, hstate, out, xres;
    float* u;
    const int* lens; float* plain_out; int plain_bs, plain_ts, plain_off, reverse;
    float* y_all;
};

static void v1_gru_step(int MB, int B, hipStream_t st, const int* ctl, int step_off, int t_hint, const GruStep& gs) {
    GruEpi e{};
    e.H = gs.g->H; e.hprev = gs.hprev; e.rh = gs.rh; e.u = gs.u; e.hstate = gs.hstate; e.out = gs.out; e.xres = gs.xres;
    e.lens = gs.lens; e.plain_out = gs.plain_out; e.plain_bs = gs.plain_bs; e.plain_ts = gs.plain_ts; e.plain_off = gs.plain_off;
    e.reverse = gs.reverse; e.y_all = gs.y_all;
    SkinnyCore c = v1_core(gs.g->gates, ctl, step_off, t_hint, B);
    c.kb0 = gs.kbx; c.seg[0][0] = gs.x; c.seg[0][1] = gs.gates_rest;
    launch_gru<1>(MB, gs.g->gates.n_wg, st, c, e);
    SkinnyCore c2 = v1_core(gs.g->cand, ctl, step_off, t_hint, B);
    c2.kb0 = gs.kbx; c2.seg[0][0] = gs.x; c2.seg[0][1] = gs.cand_rest;
    launch_gru<2>(MB, gs.g->cand.n_wg, st, c2, e);
}

// ---- CBHG ---------------------------------------------------------------------------------------------
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; }
};

// split-K workspace of the first projection (launch_conv splits only when the output tiles fill less than half of a 256-CU chip)
static size_t cbhg_p1part_floats(const CbhgW& c, size_t Mtot) {
    return (ceil_div((long long)Mtot, 64) * ceil_div(c.P1, CG_BN) * 2 <= 256) ? Mtot * c.P1 * 16 : 0;
}
static size_t cbhg_ws_bytes(const CbhgW& c, int B, int T) {
    const int MB = mb_for(B);
    const size_t Mtot = (size_t)B * T;
    Carver cv;
    cv.take(Mtot * c.K * c.C * 4); cv.take(Mtot * c.K * c.C * 4); cv.take(Mtot * c.P1 * 4); cv.take(cbhg_p1part_floats(c, Mtot) * 4); cv.take(Mtot * c.P2 * 4);
    cv.take(Mtot * c.half * 4); cv.take(Mtot * c.half * 4); cv.take(Mtot * 2 * c.half * 4);
    const int KPh = pad16(c.half);
    cv.take(2 * (size_t)T * MB * 16 * KPh * 4);                 // per-step inputs, both directions
    cv.take(2 * 2 * (size_t)MB * 16 * KPh * 4);                 // h: 2 directions x 2 slots
    cv.take(2 * (size_t)MB * 16 * KPh * 4);                     // r*h
    cv.take(2 * (size_t)MB * 16 * c.half * 4);                  // u
    cv.take(64);                                                // ctl
    cv.take(2 * Mtot * 3 * c.half * 4);                         // whole-sequence GRU: x-halves at every position, both directions
    return cv.off;
}

// x: dev [B*T][cin]; d_lens: dev int32 [B] (T for every row when the caller gave no lengths); out: dev [B][T][out_units]
static int cbhg_forward(t2v1_model* m, const CbhgW& c, const float* x, int B, int T, const int* d_lens, float* out, char* ws) {
    t2_model* b = &m->base;
    hipStream_t st = b->ctx->stream;
    const int MB = mb_for(B), Mtot = B * T, half = c.half, KC = c.K * c.C;
    Carver cv;
    float* bank = (float*)(ws + cv.take((size_t)Mtot * KC * 4));
    float* pooled = (float*)(ws + cv.take((size_t)Mtot * KC * 4));
    float* p1 = (float*)(ws + cv.take((size_t)Mtot * c.P1 * 4));
    const size_t p1part_floats = cbhg_p1part_floats(c, (size_t)Mtot);
    float* p1part = (float*)(ws + cv.take(p1part_floats * 4));
    float* p2 = (float*)(ws + cv.take((size_t)Mtot * c.P2 * 4));
    float* hwa = (float*)(ws + cv.take((size_t)Mtot * half * 4));
    float* hwb = (float*)(ws + cv.take((size_t)Mtot * half * 4));
    float* ht = (float*)(ws + cv.take((size_t)Mtot * 2 * half * 4));
    const int KPh = pad16(half);
    const size_t per_step = (size_t)MB * 16 * KPh;
    float* xf = (float*)(ws + cv.take(2 * (size_t)T * per_step * 4));
    const size_t zero_from = cv.off;
    float* hst = (float*)(ws + cv.take(2 * 2 * per_step * 4));
    float* rh = (float*)(ws + cv.take(2 * per_step * 4));
    float* ug = (float*)(ws + cv.take(2 * (size_t)MB * 16 * half * 4));
    int* ctl = (int*)(ws + cv.take(64));
    HIP_TRY(hipMemsetAsync(ws + zero_from, 0, cv.off - zero_from, st));
    // conv bank, concatenated along channels (modules.py:206)
    // a short batch leaves each bank kernel (N = C columns) a fraction of the chip: the K kernels as ONE convolution of width K then
    // (zero-padded taps: ~2x the products, K x the workgroups); long batches keep the K exact convolutions
    if (c.bank_u.w != nullptr && m->opt_bank_unified && ceil_div(Mtot, 64) * ceil_div(c.C, CG_BN) * 2 <= b->n_cu)
        T2_TRY(launch_conv(b, c.bank_u, x, c.cin, bank, KC, Mtot, T, 1, nullptr, 0));
    else
    for (int k = 1; k <= c.K; ++k)
        T2_TRY(launch_conv(b, c.bank[k - 1], x, c.cin, bank + (size_t)(k - 1) * c.C, KC, Mtot, T, 1, nullptr, 0));
    {
        const long long n = (long long)Mtot * KC;
        hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, st, bank, pooled,
                           (long long)Mtot, T, KC);
    }
    {   // few output tiles (N = P1) against K = 3 K C: split over gridDim.z where the batch is short (launch_conv decides)
        ConvLayer L = c.proj1;
        if (p1part_floats) { L.part = p1part; L.part_floats = p1part_floats; }
        T2_TRY(launch_conv(b, L, pooled, KC, p1, c.P1, Mtot, T, 1, nullptr, 0));
    }
    T2_TRY(launch_conv(b, c.proj2, p1, c.P1, p2, c.P2, Mtot, T, 0, x, c.cin));          // + residual (modules.py:214)
    const float* cur = p2;
    float* nxt = hwa;
    if (c.has_adj) { T2_TRY(launch_conv(b, c.adj, p2, c.P2, hwa, half, Mtot, Mtot, 0, nullptr, 0)); cur = hwa; nxt = hwb; }
    for (size_t i = 0; i < c.highway.size(); ++i) {
        T2_TRY(launch_conv(b, c.highway[i], cur, half, ht, 2 * half, Mtot, Mtot, 0, nullptr, 0));
        const long long n = (long long)Mtot * half;
        hipLaunchKernelGGL(highway_kernel, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, st, ht, cur, nxt,
                           (long long)Mtot, half);
        cur = nxt;
        nxt = (cur == hwa) ? hwb : hwa;
    }
    // BiGRU (modules.py:222-229)
    if (half == 128 && m->opt_gruseq) {
        // one GEMM per direction for the input halves at every position, then ONE persistent launch for the recurrence
        float* xz = (float*)(ws + cv.take(2 * (size_t)Mtot * 3 * half * 4));
        GruSeqArgs ga{};
        for (int dir = 0; dir < 2; ++dir) {
            float* xd = xz + (size_t)dir * Mtot * 3 * half;
            T2_TRY(launch_conv(b, c.gru_x[dir], cur, half, xd, 3 * half, Mtot, Mtot, 0, nullptr, 0));
            ga.xz[dir] = xd; ga.whg[dir] = c.gru_hg[dir]; ga.whc[dir] = c.gru_hc[dir];
        }
        ga.lens = d_lens; ga.out = out; ga.B = B; ga.T = T;
        HIP_TRY(hipMemsetAsync(out, 0, (size_t)Mtot * c.out_units * 4, st));
        hipLaunchKernelGGL(gru_seq_kernel<128>, dim3(ceil_div(B, 16), 2), dim3(512), 0, st, ga);
        HIP_TRY(hipGetLastError());
        return T2_OK;
    }
    {
        const long long n = 2LL * T * per_step;
        hipLaunchKernelGGL(seq_to_frag_kernel, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, st,
                           cur, d_lens, xf, B, T, half, MB, KPh);
    }
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)Mtot * c.out_units * 4, st));
    const int h_ctl[8] = {0, 0, T, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(ctl, h_ctl, sizeof(h_ctl), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));        // h_ctl is a stack temporary
    const int hsz = (int)per_step;
    for (int s = 0; s < T; ++s)
        for (int dir = 0; dir < 2; ++dir) {
            GruStep gs{};
            gs.g = &c.gru[dir];
            gs.x = FragBuf{xf + (size_t)dir * T * per_step, hsz, T, 0, 0}; gs.kbx = KPh / 16;
            float* hb = hst + (size_t)dir * 2 * per_step;
            gs.gates_rest = FragBuf{hb, hsz, 2, 0, 0};
            gs.cand_rest = FragBuf{rh + (size_t)dir * per_step, 0, 1, 0, 0};
            gs.hprev = FragBuf{hb, hsz, 2, 0, 0};
            gs.rh = FragBuf{rh + (size_t)dir * per_step, 0, 1, 0, 0};
            gs.hstate = FragBuf{hb, hsz, 2, 1, 0};
            gs.out = FragBuf{nullptr, 0, 1, 0, 0}; gs.xres = FragBuf{nullptr, 0, 1, 0, 0};
            gs.u = ug + (size_t)dir * MB * 16 * half;
            gs.lens = d_lens; gs.plain_out = out; gs.plain_bs = T * c.out_units; gs.plain_ts = c.out_units; gs.plain_off = dir * half;
            gs.reverse = dir;
            v1_gru_step(MB, B, st, ctl, s, s, gs);
        }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int v1_lens_to_dev(t2v1_model* m, const int64_t* lens, int B, int T, int min_len, int* d_lens) {
    std::vector<int> h(B);
    for (int bb = 0; bb < B; ++bb) {
        const int64_t l = lens ? lens[bb] : T;
        if (l < min_len || l > T) return fail(T2_EINVAL, "sequence length %lld outside %d..%d", (long long)l, min_len, T);
        h[bb] = (int)l;
    }
    hipStream_t st = m->base.ctx->stream;
    HIP_TRY(hipMemcpyAsync(d_lens, h.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return T2_OK;
}

static int v1_ready(t2v1_model* m) {
    if (!m) return fail(T2_EINVAL, "null model");
    if (!m->base.finalized) return fail(T2_ESTATE, "model not finalized");
    HIP_TRY(hipSetDevice(m->base.ctx->device));
    return T2_OK;
}

extern "C" int t2v1_cbhg(t2v1_model* m, int which, const float* x, const int64_t* lens, int B, int T, float* out) {
    T2_TRY(v1_ready(m));
    if (!x || !out || (which != 0 && which != 1)) return fail(T2_EINVAL, "bad argument");
    T2_TRY(check_batch(B, T));
    const CbhgW& c = which == 0 ? m->enc_cbhg : m->post_cbhg;
    Carver cv;
    const size_t o_lens = cv.take((size_t)B * 4), o_c = cv.take(cbhg_ws_bytes(c, B, T));
    T2_TRY(ensure_ws(&m->ws, &m->ws_bytes, cv.off));
    char* ws = (char*)m->ws;
    T2_TRY(v1_lens_to_dev(m, lens, B, T, 0, (int*)(ws + o_lens)));
    return cbhg_forward(m, c, x, B, T, (int*)(ws + o_lens), out, ws + o_c);
}

static int v1_encode_impl(t2v1_model* m, const int64_t* ids, const float* emb_in, const int64_t* lens, int B, int T, float* memory_out) {
    T2_TRY(v1_ready(m));
    if (!memory_out) return fail(T2_EINVAL, "null argument");
    T2_TRY(check_batch(B, T));
    const t2v1_dims& d = m->d;
    t2_model* b = &m->base;
    hipStream_t st = b->ctx->stream;
    const int Mtot = B * T;
    std::vector<int> h_ids;
    if (ids) {
        h_ids.resize(Mtot);
        for (int i = 0; i < Mtot; ++i) {
            if (ids[i] < 0 || ids[i] >= d.num_symbols)
                return fail(T2_ERANGE, "embedding id %lld outside [0, %d) (modules.py:65-66)", (long long)ids[i], d.num_symbols);
            h_ids[i] = (int)ids[i];
        }
    }
    Carver cv;
    const size_t o_ids = cv.take((size_t)Mtot * 4), o_lens = cv.take((size_t)B * 4);
    const size_t o_e = cv.take((size_t)Mtot * d.embedding_dim * 4), o_p0 = cv.take((size_t)Mtot * d.enc_prenet_units[0] * 4);
    const size_t o_p1 = cv.take((size_t)Mtot * d.enc_prenet_units[1] * 4), o_c = cv.take(cbhg_ws_bytes(m->enc_cbhg, B, T));
    T2_TRY(ensure_ws(&m->ws, &m->ws_bytes, cv.off));
    char* ws = (char*)m->ws;
    T2_TRY(v1_lens_to_dev(m, lens, B, T, 0, (int*)(ws + o_lens)));
    const float* emb = emb_in;
    if (ids) {
        HIP_TRY(hipMemcpyAsync(ws + o_ids, h_ids.data(), (size_t)Mtot * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        hipLaunchKernelGGL(embedding_kernel, dim3(std::min(1024, ceil_div(Mtot * d.embedding_dim, 256))), dim3(256), 0, st,
                           (const int*)(ws + o_ids), m->embedding, (float*)(ws + o_e), Mtot, d.embedding_dim);
        emb = (const float*)(ws + o_e);
    }
    // EncoderV1.call (tacotron_v1.py:46-49): PreNets (dropout off), then CBHG
    T2_TRY(launch_conv(b, m->enc_prenet[0], emb, d.embedding_dim, (float*)(ws + o_p0), d.enc_prenet_units[0], Mtot, Mtot, 1, nullptr, 0));
    T2_TRY(launch_conv(b, m->enc_prenet[1], (float*)(ws + o_p0), d.enc_prenet_units[0], (float*)(ws + o_p1), d.enc_prenet_units[1], Mtot,
                       Mtot, 1, nullptr, 0));
    return cbhg_forward(m, m->enc_cbhg, (float*)(ws + o_p1), B, T, (int*)(ws + o_lens), memory_out, ws + o_c);
}

extern "C" int t2v1_encode(t2v1_model* m, const int64_t* ids, const int64_t* lens, int B, int T_in, float* memory_out) {
    if (!ids) return fail(T2_EINVAL, "ids is null");
    return v1_encode_impl(m, ids, nullptr, lens, B, T_in, memory_out);
}

extern "C" int t2v1_encode_embedded(t2v1_model* m, const float* emb, const int64_t* lens, int B, int T_in, float* memory_out) {
    if (!emb) return fail(T2_EINVAL, "emb is null");
    return v1_encode_impl(m, nullptr, emb, lens, B, T_in, memory_out);
}

extern "C" int t2v1_postnet(t2v1_model* m, const float* mel, int B, int T_out, float* spec_out) {
    T2_TRY(v1_ready(m));
    if (!mel || !spec_out) return fail(T2_EINVAL, "null argument");
    T2_TRY(check_batch(B, T_out));
    const t2v1_dims& d = m->d;
    Carver cv;
    const size_t o_lens = cv.take((size_t)B * 4), o_y = cv.take((size_t)B * T_out * d.post_cbhg_out_units * 4);
    const size_t o_c = cv.take(cbhg_ws_bytes(m->post_cbhg, B, T_out));
    T2_TRY(ensure_ws(&m->ws, &m->ws_bytes, cv.off));
    char* ws = (char*)m->ws;
    T2_TRY(v1_lens_to_dev(m, nullptr, B, T_out, 0, (int*)(ws + o_lens)));
    T2_TRY(cbhg_forward(m, m->post_cbhg, mel, B, T_out, (int*)(ws + o_lens), (float*)(ws + o_y), ws + o_c));
    return launch_conv(&m->base, m->post_linear, (float*)(ws + o_y), d.post_cbhg_out_units, spec_out, d.num_freq, B * T_out, B * T_out, 0,
                       nullptr, 0);
}

// ---- decoder -------------------------------------------------------------------------------------------
static int v1_state_alloc(t2v1_model* m, int B, int T, int cap, t2v1_state** out) {
    const t2v1_dims& d = m->d;
    t2v1_state* s = new t2v1_state();
    s->m = m; s->B = B; s->T = T; s->MB = mb_for(B); s->cap = cap;
    const int MB = s->MB, R = MB * 16, blk = MB * 256;
    const int E = d.cbhg_out_units, A = d.attention_out_units, D = d.decoder_out_units, M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame;
    Carver cv;
    const size_t o_val = cv.take((size_t)B * T * E * 4), o_keys = cv.take((size_t)B * T * A * 4), o_lens = cv.take((size_t)B * 4);
    if (hipMalloc(&s->fixed, cv.off) != hipSuccess) { v1_state_free(s); return fail(T2_ENOMEM, "hipMalloc(state) failed"); }
    s->values = (float*)((char*)s->fixed + o_val); s->keys = (float*)((char*)s->fixed + o_keys); s->lens = (int*)((char*)s->fixed + o_lens);
    s->KBe = pad16(E) / 16; s->KBa = pad16(A) / 16; s->KBd = pad16(D) / 16;
    s->XA_sz = (s->KBe + s->KBa) * blk; s->X1_sz = 2 * s->KBd * blk;
    cv = Carver();
    const size_t o_ctl = cv.take(64), o_as = cv.take((size_t)B * T * 4), o_al = cv.take((size_t)B * T * 4), o_cp = cv.take((size_t)B * E * 4);
    const size_t o_q = cv.take((size_t)R * A * 4), o_ua = cv.take((size_t)R * A * 4), o_u1 = cv.take((size_t)R * D * 4), o_u2 = cv.take((size_t)R * D * 4);
    const size_t o_xf = cv.take((size_t)R * pad16(M * nf) * 4), o_xp0 = cv.take((size_t)R * pad16(d.dec_prenet_units[0]) * 4),
                 o_xp1 = cv.take((size_t)R * pad16(d.dec_prenet_units[1]) * 4);
    const size_t o_XA = cv.take(2 * (size_t)s->XA_sz * 4), o_XR = cv.take((size_t)s->XA_sz * 4);
    const size_t o_X1 = cv.take(2 * (size_t)s->X1_sz * 4), o_R1 = cv.take((size_t)s->KBd * blk * 4);
    const size_t o_X2 = cv.take(2 * (size_t)s->X1_sz * 4), o_R2 = cv.take((size_t)s->KBd * blk * 4), o_XY = cv.take((size_t)s->KBd * blk * 4);
    s->slab_bytes = cv.off;
    if (hipMalloc(&s->slab, cv.off) != hipSuccess) { v1_state_free(s); return fail(T2_ENOMEM, "hipMalloc(state) failed"); }
    char* p = (char*)s->slab;
    s->ctl = (int*)(p + o_ctl); s->att_state = (float*)(p + o_as); s->alignments = (float*)(p + o_al); s->ctx_plain = (float*)(p + o_cp);
    s->q = (float*)(p + o_q); s->u_a = (float*)(p + o_ua); s->u1 = (float*)(p + o_u1); s->u2 = (float*)(p + o_u2);
    s->xf = (float*)(p + o_xf); s->xp[0] = (float*)(p + o_xp0); s->xp[1] = (float*)(p + o_xp1);
    s->XA = (float*)(p + o_XA); s->XR = (float*)(p + o_XR); s->X1 = (float*)(p + o_X1); s->RH1 = (float*)(p + o_R1);
    s->X2 = (float*)(p + o_X2); s->RH2 = (float*)(p + o_R2); s->XY = (float*)(p + o_XY);
    cv = Carver();
    const size_t o_mel = cv.take((size_t)B * cap * r * M * 4), o_stop = cv.take((size_t)B * cap * 4), o_hist = cv.take((size_t)cap * B * T * 4),
                 o_tgt = cv.take((size_t)B * cap * r * M * 4);
    if (hipMalloc(&s->outbuf, cv.off) != hipSuccess) { v1_state_free(s); return fail(T2_ENOMEM, "hipMalloc(state) failed"); }
    p = (char*)s->outbuf;
    s->o_mel = (float*)(p + o_mel); s->o_stop = (float*)(p + o_stop); s->o_hist = (float*)(p + o_hist); s->i_targets = (float*)(p + o_tgt);
    *out = s;
    return T2_OK;
}

static size_t v1_attn_lds(const t2v1_dims& d, int T) { return (size_t)attn_lds_floats(T, d.attention_out_units, 1, 1) * 4; }

static int v1_get_state(t2v1_model* m, int B, int T, int cap, t2v1_state** out) {
    for (t2v1_state* s : m->states)
        if (s->B == B && s->T == T && s->cap == cap) { *out = s; return T2_OK; }
    if (v1_attn_lds(m->d, T) > 120 * 1024) return fail(T2_EINVAL, "T_in %d too long for the attention kernel's LDS", T);
    if (m->states.size() >= 4) {
        HIP_TRY(hipStreamSynchronize(m->base.ctx->stream));
        for (t2v1_state* s : m->states) v1_state_free(s);
        m->states.clear();
    }
    t2v1_state* s = nullptr;
    T2_TRY(v1_state_alloc(m, B, T, cap, &s));
    m->states.push_back(s);
    *out = s;
    return T2_OK;
}

// BahdanauAttention constructor side effects (App. A5) + zero state
static int v1_state_init(t2v1_state* s, const float* memory, const int64_t* lens, int max_steps, int min_iters, int stop_enabled) {
    t2v1_model* m = s->m;
    hipStream_t st = m->base.ctx->stream;
    const int B = s->B, T = s->T, E = m->d.cbhg_out_units;
    T2_TRY(v1_lens_to_dev(m, lens, B, T, 1, s->lens));
    const long long n = (long long)B * T * E;
    hipLaunchKernelGGL(mask_memory_kernel, dim3((unsigned)std::min<long long>(2048, (n + 255) / 256)), dim3(256), 0, st, memory, s->lens,
                       s->values, B, T, E);
    T2_TRY(launch_conv(&m->base, m->memory_layer, s->values, E, s->keys, m->d.attention_out_units, B * T, B * T, 0, nullptr, 0));
    HIP_TRY(hipMemsetAsync(s->slab, 0, s->slab_bytes, st));
    int* h = m->h_ctl;
    HIP_TRY(hipStreamSynchronize(st));
    h[0] = 0; h[1] = 0; h[2] = max_steps; h[3] = min_iters; h[4] = stop_enabled; h[5] = h[6] = h[7] = 0;
    HIP_TRY(hipMemcpyAsync(s->ctl, h, 8 * sizeof(int), hipMemcpyHostToDevice, st));
    return T2_OK;
}

// One step of OutputAndStopTokenWrapper(DecoderRNNV1(...)) (tacotron_v1.py:63-91,126-130): launch sequence
static int v1_launch_step(t2v1_state* s, int step_off, int mode, int slot_hint) {
    t2v1_model* m = s->m;
    const t2v1_dims& d = m->d;
    hipStream_t st = m->base.ctx->stream;
    const int B = s->B, MB = s->MB, T = s->T;
    const int E = d.cbhg_out_units, A = d.attention_out_units, D = d.decoder_out_units, M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame;
    const int hint = slot_hint ? step_off : -1;
    auto single = [&](float* base, int koff) { return FragBuf{base, 0, 1, 0, koff}; };
    auto pp = [&](float* base, int sz, int dt, int koff) { return FragBuf{base, sz, 2, dt, koff}; };
    const FragBuf none{nullptr, 0, 1, 0, 0};
    const int KBe = s->KBe, KBd = s->KBd;
    // teacher forcing with batched PreNets / projections (v1_teacher_prenets, v1_teacher_projection): the step keeps only
    // what the recurrence needs
    const bool batched = mode == 0 && s->tf_frag != nullptr;
    // 1. decoder input frames (teacher forcing only; otherwise the projection of the previous step wrote them)
    if (mode == 0 && !batched) {
        FeedArgs f{};
        f.src = s->i_targets + (size_t)M * (r - nf); f.row_stride = (long long)s->cap * r * M; f.t_stride = (long long)r * M;
        f.dst = single(s->xf, 0); f.B = B; f.K = M * nf; f.MB = MB; f.go_zero = 1; f.ctl = s->ctl; f.step_off = step_off;
        hipLaunchKernelGGL(feed_frames_kernel, dim3(ceil_div(MB * 16 * pad16(M * nf), 256)), dim3(256), 0, st, f);
    }
    // 2. PreNets (rnn_wrappers.py:51-53)
    for (int i = 0; i < 2 && !batched; ++i) {
        SkinnyCore c = v1_core(m->dec_prenet[i], s->ctl, step_off, hint, B);
        c.seg[0][1] = single(i == 0 ? s->xf : s->xp[0], 0);
        DenseEpi e{};
        e.N = d.dec_prenet_units[i]; e.relu = 1; e.out_frag = single(s->xp[i], 0); e.stop_col = -1;
        launch_dense(MB, dim3(m->dec_prenet[i].n_wg), st, c, e);
    }
    // 3. attention GRU on concat(prenet out, previous context) (AttentionWrapper, App. A6)
    {
        GruStep gs{};
        gs.g = &m->att_gru;
        gs.x = batched ? FragBuf{s->tf_frag, (int)s->tf_stride, s->cap, 0, 0} : single(s->xp[1], 0); gs.kbx = m->att_gru.gates.seg_kb[0];
        gs.gates_rest = pp(s->XA, s->XA_sz, 0, 0);
        gs.cand_rest = single(s->XR, 0);
        gs.hprev = pp(s->XA, s->XA_sz, 0, KBe * 16); gs.rh = single(s->XR, KBe * 16); gs.hstate = pp(s->XA, s->XA_sz, 1, KBe * 16);
        gs.out = none; gs.xres = none; gs.u = s->u_a;
        v1_gru_step(MB, B, st, s->ctl, step_off, hint, gs);
    }
    // 4. processed query = h_att . query_layer (no bias)
    {
        SkinnyCore c = v1_core(m->query, s->ctl, step_off, hint, B);
        c.seg[0][1] = pp(s->XA, s->XA_sz, 1, KBe * 16);
        DenseEpi e{};
        e.N = A; e.out_frag = none; e.out_plain = s->q; e.plain_ld = A; e.plain_t_stride = 0; e.plain_ncol = A; e.stop_col = -1;
        launch_dense(MB, dim3(m->query.n_wg), st, c, e);
    }
    // 5. plain Bahdanau score, masked softmax, context: the location-sensitive kernel with a zero location term
    {
        AttnArgs a{};
        a.keys = s->keys; a.values = s->values; a.lens = s->lens; a.att_state = s->att_state; a.alignments = s->alignments;
        a.history = s->o_hist; a.qpart = s->q; a.nparts = 1;
        a.ctx_dst[0] = pp(s->XA, s->XA_sz, 1, 0); a.ctx_dst[1] = single(s->XR, 0); a.ctx_plain = s->ctx_plain;
        a.conv_w = m->zeros; a.conv_b = m->zeros; a.wloc = m->zeros; a.v_a = m->v_a; a.b_a = m->zeros;
        a.B = B; a.T = T; a.U = A; a.E = E; a.F = 1; a.K = 1; a.MB = MB; a.cumulative = 0; a.ctl = s->ctl; a.step_off = step_off;
        launch_attention(MB, B, v1_attn_lds(d, T), st, a, nullptr);
    }
    // 6. OutputProjectionWrapper on concat(h_att, context) (rnn_wrappers.py:180-186)
    {
        SkinnyCore c = v1_core(m->outproj, s->ctl, step_off, hint, B);
        c.seg[0][1] = pp(s->XA, s->XA_sz, 1, 0);
        DenseEpi e{};
        e.N = D; e.out_frag = pp(s->X1, s->X1_sz, 0, 0); e.stop_col = -1;
        launch_dense(MB, dim3(m->outproj.n_wg), st, c, e);
    }
    // 7./8. two ResidualWrapper(GRUCell) layers (tacotron_v1.py:72-73)
    for (int layer = 0; layer < 2; ++layer) {
        float* X = layer == 0 ? s->X1 : s->X2;
        float* RH = layer == 0 ? s->RH1 : s->RH2;
        GruStep gs{};
        gs.g = layer == 0 ? &m->gru1 : &m->gru2;
        gs.x = pp(X, s->X1_sz, 0, 0); gs.kbx = KBd;
        gs.gates_rest = pp(X, s->X1_sz, 0, KBd * 16);
        gs.cand_rest = single(RH, 0);
        gs.hprev = pp(X, s->X1_sz, 0, KBd * 16); gs.rh = single(RH, 0); gs.hstate = pp(X, s->X1_sz, 1, KBd * 16);
        gs.out = layer == 0 ? pp(s->X2, s->X1_sz, 0, 0) : single(s->XY, 0);
        gs.xres = pp(X, s->X1_sz, 0, 0);
        gs.u = layer == 0 ? s->u1 : s->u2;
        gs.y_all = (batched && layer == 1) ? s->y_all : nullptr;
        v1_gru_step(MB, B, st, s->ctl, step_off, hint, gs);
    }
    // 9. mel / stop projections + stop rule (rnn_wrappers.py:105-109, helpers.py:90-94)
    if (!batched) {
        SkinnyCore c = v1_core(m->proj, s->ctl, step_off, hint, B);
        c.seg[0][1] = single(s->XY, 0);
        DenseEpi e{};
        e.N = M * r + 1;
        e.out_plain = s->o_mel; e.plain_ld = s->cap * r * M; e.plain_t_stride = (long long)r * M; e.plain_ncol = M * r;
        e.out_frag = mode == 1 ? single(s->xf, 0) : none; e.frag_col0 = M * (r - nf);
        e.stop_col = M * r; e.out_stop = s->o_stop; e.stop_ld = s->cap; e.stop_t_mul = 1; e.ctl_w = s->ctl;
        launch_dense(MB, dim3(m->proj.n_wg), st, c, e);
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// Teacher forcing feeds known frames (helpers.py:134-136,182) and nothing of a step's output is fed back: the PreNets of all
// steps run as two GEMMs over S*B rows before the loop, the projections as one GEMM after it (the same scheme as the V2 path).
static int v1_teacher_prenets(t2v1_state* s) {
    t2v1_model* m = s->m;
    const t2v1_dims& d = m->d;
    hipStream_t st = m->base.ctx->stream;
    const int B = s->B, S = s->cap, MB = s->MB, M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, D = d.decoder_out_units;
    const long long SB = (long long)S * B;
    if (!s->tf_buf) {
        const int pmax = std::max(d.dec_prenet_units[0], d.dec_prenet_units[1]);
        s->tf_stride = (long long)MB * 16 * pad16(d.dec_prenet_units[1]);
        const size_t n_xin = align_up((size_t)SB * M * nf, 64), n_x = align_up((size_t)SB * pmax, 64), n_frag = align_up((size_t)S * s->tf_stride, 64),
                     n_y = align_up((size_t)SB * D, 64), n_p = align_up((size_t)SB * m->proj_plain.cout, 64);
        hipError_t e = hipMalloc((void**)&s->tf_buf, (n_xin + 2 * n_x + n_frag + n_y + n_p) * 4);
        if (e != hipSuccess) { s->tf_buf = nullptr; return fail(T2_ENOMEM, "hipMalloc(teacher-forced buffers) failed: %s", hipGetErrorString(e)); }
        HIP_TRY(hipMemsetAsync(s->tf_buf, 0, (n_xin + 2 * n_x + n_frag) * 4, st));
        s->tf_xin = s->tf_buf; s->tf_x[0] = s->tf_xin + n_xin; s->tf_x[1] = s->tf_x[0] + n_x; s->tf_frag = s->tf_x[1] + n_x;
        s->y_all = s->tf_frag + n_frag; s->proj_tmp = s->y_all + n_y;
    }
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); };
    hipLaunchKernelGGL(feed_gather_kernel, grid1(SB * M * nf), dim3(256), 0, st, s->i_targets, s->tf_xin, B, S, M, r, nf);
    const float* x = s->tf_xin;
    for (int i = 0; i < 2; ++i) {
        const ConvLayer& L = m->dec_prenet_plain[i];
        T2_TRY(launch_conv(&m->base, L, x, L.cin, s->tf_x[i], L.cout, (int)SB, (int)SB, 1, nullptr, 0));
        if (i == 1)
            hipLaunchKernelGGL(prenet_dropout_all_kernel, grid1(SB * L.cout), dim3(256), 0, st, s->tf_x[i], s->tf_frag, s->tf_stride, S, B, L.cout, MB,
                               0.f, 0u, 0);
        x = s->tf_x[i];
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int v1_teacher_projection(t2v1_state* s) {
    t2v1_model* m = s->m;
    const t2v1_dims& d = m->d;
    const int S = s->cap, B = s->B, Np = m->proj_plain.cout, ncol = d.num_mels * d.outputs_per_step;
    T2_TRY(launch_conv(&m->base, m->proj_plain, s->y_all, d.decoder_out_units, s->proj_tmp, Np, S * B, S * B, 0, nullptr, 0));
    const long long n = (long long)S * B * (ncol + 1);
    hipLaunchKernelGGL(proj_scatter_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))), dim3(256), 0,
                       m->base.ctx->stream, s->proj_tmp, Np, S, B, ncol, s->o_mel, (long long)s->cap * ncol, s->o_stop, s->cap);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int v1_enqueue_chunk(t2v1_state* s, int mode, int chunk) {
    const int slot_hint = (chunk % 2 == 0) ? 1 : 0;
    for (int i = 0; i < chunk; ++i) T2_TRY(v1_launch_step(s, i, mode, slot_hint));
    hipLaunchKernelGGL(advance_ctl_kernel, dim3(1), dim3(64), 0, s->m->base.ctx->stream, s->ctl, chunk);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int v1_run_chunk(t2v1_state* s, int mode) {
    t2v1_model* m = s->m;
    hipStream_t st = m->base.ctx->stream;
    const int chunk = m->opt_chunk;
    if (!m->opt_graph) return v1_enqueue_chunk(s, mode, chunk);
    if (s->gexec[mode] == nullptr || s->gchunk[mode] != chunk) {
        if (s->gexec[mode]) { hipGraphExecDestroy(s->gexec[mode]); s->gexec[mode] = nullptr; }
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int r = v1_enqueue_chunk(s, mode, chunk);
        hipError_t e = hipStreamEndCapture(st, &g);
        if (r != T2_OK) { if (g) hipGraphDestroy(g); return r; }
        if (e != hipSuccess) return fail(T2_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&s->gexec[mode], g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
        if (e != hipSuccess) { s->gexec[mode] = nullptr; return fail(T2_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
        s->gchunk[mode] = chunk;
    }
    HIP_TRY(hipGraphLaunch(s->gexec[mode], st));
    return T2_OK;
}

extern "C" int t2v1_decode_teacher(t2v1_model* m, const float* memory, const int64_t* lens, const float* targets, int B, int T_in,
                                   int T_out, int teacher_forcing, float* mel, float* stop, float* align) {
    T2_TRY(v1_ready(m));
    if (!memory || !targets || !mel || !stop || !align) return fail(T2_EINVAL, "null argument");
    T2_TRY(check_batch(B, T_in));
    const int M = m->d.num_mels, r = m->d.outputs_per_step;
    if (T_out < r || T_out % r != 0) return fail(T2_EINVAL, "T_out must be a positive multiple of outputs_per_step");
    hipStream_t st = m->base.ctx->stream;
    const int S = T_out / r;
    if (teacher_forcing) {                          // the training forward's persistent launch where the shape allows
        bool done = false;
        T2_TRY(v1_teacher_via_trainer(m, memory, lens, targets, B, T_in, S, mel, stop, align, &done));
        if (done) return T2_OK;
    }
    t2v1_state* s = nullptr;
    T2_TRY(v1_get_state(m, B, T_in, S, &s));
    T2_TRY(v1_state_init(s, memory, lens, S, 0, 0));
    HIP_TRY(hipMemcpyAsync(s->i_targets, targets, (size_t)B * T_out * M * 4, hipMemcpyDeviceToDevice, st));
    const bool batched = teacher_forcing && m->opt_tfbatch;
    if (batched) T2_TRY(v1_teacher_prenets(s));
    for (int done = 0; done < S; done += m->opt_chunk) T2_TRY(v1_run_chunk(s, teacher_forcing ? 0 : 1));
    if (batched) T2_TRY(v1_teacher_projection(s));
    HIP_TRY(hipMemcpyAsync(mel, s->o_mel, (size_t)B * T_out * M * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)B * S * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(align, s->o_hist, (size_t)S * B * T_in * 4, hipMemcpyDeviceToDevice, st));
    return T2_OK;
}

extern "C" int t2v1_decode_free(t2v1_model* m, const float* memory, const int64_t* lens, int B, int T_in, int max_iters, int min_iters,
                                float* mel, float* stop, float* align, int* steps_out) {
    T2_TRY(v1_ready(m));
    if (!memory || !mel || !stop || !align || !steps_out) return fail(T2_EINVAL, "null argument");
    T2_TRY(check_batch(B, T_in));
    if (max_iters < 1) return fail(T2_EINVAL, "max_iters must be >= 1");
    hipStream_t st = m->base.ctx->stream;
    const t2v1_dims& d = m->d;
    t2v1_state* s = nullptr;
    T2_TRY(v1_get_state(m, B, T_in, max_iters, &s));
    T2_TRY(v1_state_init(s, memory, lens, max_iters, min_iters, 1));
    // steps after the stop step are no-ops on the device; the host looks at the flag every other chunk
    int* h = m->h_ctl;
    for (int done = 0, n = 0; done < max_iters; done += m->opt_chunk, ++n) {
        T2_TRY(v1_run_chunk(s, 1));
        if ((n & 1) == 1) {
            HIP_TRY(hipMemcpyAsync(h, s->ctl, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (h[CTL_FINAL] != 0) break;
        }
    }
    const size_t cap = s->cap;
    HIP_TRY(hipMemcpyAsync(mel, s->o_mel, (size_t)B * cap * d.outputs_per_step * d.num_mels * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)B * cap * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(align, s->o_hist, cap * B * T_in * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(h, s->ctl, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *steps_out = h[CTL_FINAL] != 0 ? h[CTL_FINAL] : h[CTL_MAX];
    return T2_OK;
}
