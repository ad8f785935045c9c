// libt2mi.so -- host side of the C-ABI declared in include/t2mi.h.
// Weight repacking, workspaces, launch sequences (eager or hipGraph-chunked) for the kernels in
// kernels.h.  No CPU fallback: every entry point runs HIP kernels or fails.
#include "../../include/t2mi.h"
#include "kernels.h"
#include "persist.h"
#include "persist_bwd.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? T2_ENOMEM : T2_EHIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

#define T2_TRY(expr)              \
    do {                          \
        int r_ = (expr);          \
        if (r_ != T2_OK) return r_; \
    } while (0)

extern "C" int t2_version(void) { return T2_ABI_VERSION; }
extern "C" const char* t2_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
struct t2_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    float* red_ws = nullptr;   // partial sums of the loss reductions (train.h)
    void* voc_ws = nullptr;    // Griffin-Lim workspace (vocoder.h), grown on demand
    size_t voc_ws_bytes = 0;
};

struct HostWeight {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct Packed {            // skinny-GEMM operand
    float* w = nullptr;    // [n_wg][kbtot][64][4]
    float* bias = nullptr; // LSTM: [n_wg][16]; dense: [N]
    int n_wg = 0, kbtot = 0, nseg = 0;
    int seg_kb[T2_MAX_SEG] = {0, 0, 0};
    int seg_first[T2_MAX_SEG] = {0, 0, 0}, seg_rows[T2_MAX_SEG] = {0, 0, 0};   // source rows of each K segment
    int lstm_units = 0;    // > 0: LSTM column order (4 units x 4 gates per workgroup)
    int N = 0;             // LSTM: hidden units; dense: columns
};

struct ConvLayer {
    float* w = nullptr;     // [k*cin][cout]
    float* scale = nullptr; // [cout] or null
    float* shift = nullptr; // [cout] or null
    int k = 0, cin = 0, cout = 0;
    float* wt = nullptr;    // [cout][k*cin]: the same kernel transposed (conv_gemm_kt_kernel's operand) or null; whoever rewrites
                            // w rewrites wt (conv_refresh_wt)
    int pad_l = -1;         // zero positions in front; -1: TF's SAME rule (k - 1) / 2.  The data gradient of an even-width SAME
                            // convolution is the flipped kernel with k - 1 - (k - 1) / 2 in front
    float* part = nullptr;  // optional workspace of part_floats floats: lets launch_conv split a long reduction with few output
    size_t part_floats = 0; // tiles over gridDim.z (conv_gemm_fast_kernel<1, true> + conv_split_finish_kernel)
};

struct t2_model {
    t2_ctx* ctx = nullptr;
    t2_dims d{};
    std::map<std::string, HostWeight> hw;
    bool finalized = false;
    std::vector<void*> allocs;     // everything hipMalloc'ed for weights
    // device weights
    float* embedding = nullptr;
    std::vector<ConvLayer> enc_convs, post_convs;
    std::vector<float*> enc_bn_gamma, enc_bn_beta;   // raw batch-norm scale / offset of the encoder convs (training-mode forward)
    std::vector<float*> post_bn_gamma, post_bn_beta; // the same of the PostNet convs (t2_train_postnet_forward)
    ConvLayer post_proj, memory_layer;
    Packed enc_lstm[2];
    Packed prenet[T2_MAX_PRENET_LAYERS];
    ConvLayer prenet_plain[T2_MAX_PRENET_LAYERS];     // the same layers in plain layout (batched teacher-forced PreNets)
    Packed att_lstm, lstm1, lstm2, proj;
    Packed proj_fused;                  // proj + the first PreNet layer of the next step composed into it (free-running decode)
    bool proj_fused_ok = false;         // built at t2_model_finalize (host, in double), rebuilt by compose_proj_kernel after a device-side parameter refresh
    int opt_projfuse = 1;
    int opt_conv_generic = 0;           // tuning / tests: 1 forces the generic conv_gemm_kernel
    ConvLayer proj_plain;               // mel | stop projection in plain layout [D][pad4(M*r + 1)] (deferred, batched projection)
    int opt_tfproj = 1;                 // teacher-forced decode: projections of all steps as one GEMM after the loop
    float *wq = nullptr;                // query layer kernel [A][U], plain
    Packed wq_pk;                       // the same as a skinny dense operand (persistent decoder)
    float *conv_w = nullptr, *conv_b = nullptr, *wloc = nullptr, *v_a = nullptr, *b_a = nullptr;
    // cached workspaces
    std::vector<t2_state*> states;      // decode states owned by the model (cache)
    t2_state* pending = nullptr;        // free-run decode in flight
    t2_state* last_loop = nullptr;      // state of the last decode loop (final_state of dynamic_decode: t2_decode_final_state)
    bool last_persist = false; int last_steps_cap = 0;
    float* pend_mel = nullptr; float* pend_stop = nullptr; float* pend_align = nullptr;
    void* enc_ws = nullptr; size_t enc_ws_bytes = 0;
    void* post_ws = nullptr; size_t post_ws_bytes = 0;
    int* h_ctl = nullptr;               // pinned host mirror of a ctl block
    int* h_flag = nullptr;              // pinned mirrors of the ctl block, two chunks in flight (t2_decode_free)
    hipEvent_t flag_ev[2] = {nullptr, nullptr};
    int opt_graph = 1;                  // use hipGraph chunks for decode loops
    int opt_chunk = 16;                 // steps per graph
    int opt_splitk = 0;                 // batches > 16 rows: split-K LSTM step with in-launch reduction
    int opt_hoist = 1;                  // recurrent halves of the LSTM GEMMs ride in the attention / projection launches
    int opt_att3 = 1;                   // B <= 16: all three riders in the attention launch
    int opt_spread = 2;                 // riders spread as tile ranges over the PreNet / attention / projection launches (1: only B > 16)
    int opt_sp_att = -1;                // tuning: rider rounds the attention launch holds (-1: 3)
    int n_cu = 256;                     // compute units of the device (rider capacity of a launch)
    int opt_lstm_only = 0;              // measurement: decode steps launch only the three LSTM kernels
    int opt_time_lstm = 0;              // record HIP events around decoder LSTM launches (eager mode)
    unsigned long long param_gen = 0;   // bumped by t2_trainer_load_params; trainers compare it with their own generation
    const float* last_flat = nullptr;   // flat parameter buffer of the last refresh (owned by the caller)
    long long n_persist = 0;            // persistent decodes launched (stat "persist_launches")
    int opt_enc_persist = 1;            // EncoderV2's BiLSTM as one persistent launch (persist.h encoder_kernel) at the paper-size class
    unsigned long long n_enc_persist = 0;
    int opt_conv_mi = 0;                // measurement: 1 / 2 / 5 force 64- / 128- / 160-row conv GEMM tiles
    int opt_conv_kt = 1;                // forward convolutions on conv_gemm_kt_kernel (K-contiguous LDS tiles, transposed kernel); 0: round 2's kernel
    int opt_bwd_persist = 1;            // training backward: the reverse-time recurrence as ONE persistent launch (persist_bwd.h)
    unsigned long long n_bwd_persist = 0;
    int opt_bwd_splitk = 1;             // training backward: K = 4H data-gradient GEMMs split in two halves over the grid (train_step.h)
    int opt_persist = 1;                // free-running decodes of the paper-size dimension class run as ONE persistent launch (persist.h)
    // a persistent launch whose hand-off waits ran out of time (its workgroups were not all resident: another process holds
    // part of the GPU) is re-run with the launch-per-phase schedule; the next persist_holdoff decodes skip the persistent kernels
    long long n_persist_timeouts = 0;   // stat "persist_timeouts"
    int persist_holdoff = 0;
    int opt_tf2 = 1;                    // teacher-forced persistent decodes: two-chain schedule (two hops per step; persist.h MODE 3 / 4)
    int opt_inject_timeout = 0;         // tests: the next N persistent launches get their error word set after the launch
    // arguments of the free-running decode in flight (t2_decode_free_launch -> t2_decode_free_finish may have to re-run it)
    int pend_max_iters = 0, pend_min_iters = 0;
};

struct t2_state {
    t2_model* m = nullptr;
    int B = 0, T = 0, MB = 0, cap = 1;
    bool owned_by_model = false;
    void* slab = nullptr;        // zeroable state slab
    size_t slab_bytes = 0;
    void* fixed = nullptr;       // values, keys, lens
    // fixed
    float *values = nullptr, *keys = nullptr;
    int* lens = nullptr;
    // zeroable
    float *att_state = nullptr, *alignments = nullptr, *ctx_plain = nullptr, *qpart = nullptr;
    float *c_att = nullptr, *c1 = nullptr, *c2 = nullptr;
    // frag16 operand buffers (see launch_step): XA = [ctx | h_att] x2 slots, X1 = [h_att_raw | ctx | h1] x2,
    // X2 = [y1_raw | h2] x2, XY = y2_raw, xf = decoder input frames, xp[i] = prenet outputs
    float *XA = nullptr, *X1 = nullptr, *X2 = nullptr, *XY = nullptr, *xf = nullptr;
    float* xp[T2_MAX_PRENET_LAYERS] = {nullptr, nullptr, nullptr, nullptr};
    int XA_sz = 0, X1_sz = 0, X2_sz = 0;         // floats per slot
    int KBe = 0, KBa = 0, KBd = 0;
    float *ZA = nullptr, *Z1 = nullptr, *Z2 = nullptr;   // recurrent-half pre-activations (partial_step_kernel)
    float* Z1x = nullptr;                        // LSTM 1: product of the attention cell's raw output (spread schedule)
    float *y_all = nullptr, *proj_tmp = nullptr; // deferred projection of a teacher-forced decode: y2 of every step, GEMM output (lazy)
    unsigned* tickets = nullptr;                 // split-K arrival counters, 3 layers x tiles (zeroed with the slab)
    unsigned long long* sk_slabs = nullptr;      // split-K partial slabs [tiles][2][512][2]
    int sk_tiles = 0;
    int* ctl = nullptr;
    // persistent decoder (persist.h): exchange slab and hop counters, carved from the zeroable slab; null when not eligible
    float* pz_xs = nullptr; unsigned pz_bytes = 0; unsigned* pz_bar = nullptr;
    unsigned pz_off[17] = {0};
    // output staging (graph-stable addresses)
    float *o_mel = nullptr, *o_stop = nullptr, *o_hist = nullptr, *i_targets = nullptr;
    float *tf_buf = nullptr, *tf_xin = nullptr, *tf_x[2] = {nullptr, nullptr}, *tf_frag = nullptr;   // batched teacher-forced PreNets (lazy)
    long long tf_stride = 0;
    void* outbuf = nullptr;
    // cached graphs: [mode] 0 = teacher forcing, 1 = feed own output (free / validation)
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    int gchunk[2] = {0, 0};
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// Kernels whose workgroups wait for each other (persist.h, persist_bwd.h) need ALL their workgroups resident at once.  One
// such kernel owns the chip (one workgroup per CU), but two of them dispatched at the same moment from two streams could
// each get part of the CUs and spin until the hand-off timeout.  Within this process they are therefore serialised per
// device ON THE DEVICE: every such launch waits (hipStreamWaitEvent, no host synchronisation) for the previous one's
// completion event when that ran on another stream.  Other processes on the same GPU cannot be seen from here: that case
// is caught by the kernels' bounded waits and the launch-per-phase re-run (persist_timed_out).
// ------------------------------------------------------------------------------------------------
struct XwgGateDev { hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool armed = false; };
static std::mutex g_xwg_mu;
static XwgGateDev g_xwg[64];
struct XwgGate {                 // RAII: begin() before the launch, the destructor records the launch's completion event
    t2_ctx* c = nullptr;
    bool held = false;
    int begin(t2_ctx* ctx) {
        c = ctx;
        g_xwg_mu.lock();
        held = true;
        XwgGateDev& g = g_xwg[c->device & 63];
        if (!g.ev) HIP_TRY(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
        if (g.armed && g.last != c->stream) HIP_TRY(hipStreamWaitEvent(c->stream, g.ev, 0));
        return T2_OK;
    }
    ~XwgGate() {
        if (!held) return;
        XwgGateDev& g = g_xwg[c->device & 63];
        if (g.ev && hipEventRecord(g.ev, c->stream) == hipSuccess) { g.last = c->stream; g.armed = true; }
        g_xwg_mu.unlock();
    }
};

struct LensArg { int v[64]; };
__global__ void set_lens_kernel(LensArg a, int* __restrict__ lens, int B) {
    if ((int)threadIdx.x < B) lens[threadIdx.x] = a.v[threadIdx.x];
}
__global__ void set_ctl_kernel(int* __restrict__ ctl, int max_steps, int min_iters, int stop_enabled) {
    if (threadIdx.x < 8) ctl[threadIdx.x] = threadIdx.x == 2 ? max_steps : threadIdx.x == 3 ? min_iters : threadIdx.x == 4 ? stop_enabled : 0;
}

// tests: option inject_timeout = N marks the next N persistent launches as timed out (sets the word the kernel would set)
static int inject_timeout(t2_model* m, int* err_word);
static void note_persist_timeout(t2_model* m);

static inline void launch_lstm_splitk(int MB, int n_wg4, hipStream_t st, const SkinnyCore& c, const LstmEpi& e, unsigned* tickets,
                                      unsigned long long* slabs) {
    const dim3 grid(2 * ceil_div(n_wg4, 2));
    if (MB == 2) hipLaunchKernelGGL((lstm_step_splitk_kernel<2>), grid, dim3(T2_SKINNY_THREADS), 0, st, c, e, n_wg4, tickets, slabs);
    else hipLaunchKernelGGL((lstm_step_splitk_kernel<4>), grid, dim3(T2_SKINNY_THREADS), 0, st, c, e, n_wg4, tickets, slabs);
}

// optional HIP-event timing of the decoder LSTM launches (eager mode only; bench.py's roofline leg)
struct LstmTimer {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    bool on = false;
};
static LstmTimer g_timer;

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// every K segment of a skinny operand is padded to 64 (4 k-blocks of 16) so that the kernels' groups of
// 2 or 4 k-blocks never straddle a segment
static inline int pad16(int k) { return (k + 63) & ~63; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int dev_alloc(t2_model* m, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    m->allocs.push_back(*p);
    return T2_OK;
}

static int upload(t2_model* m, float** dst, const std::vector<float>& v) {
    T2_TRY(dev_alloc(m, (void**)dst, v.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return T2_OK;
}

static int get_w(t2_model* m, const std::string& name, const HostWeight** out, std::vector<int64_t> shape) {
    auto it = m->hw.find(name);
    if (it == m->hw.end()) return fail(T2_ESTATE, "weight '%s' was never set", name.c_str());
    if (it->second.shape != shape) return fail(T2_EINVAL, "weight '%s' has the wrong shape", name.c_str());
    *out = &it->second;
    return T2_OK;
}

// Pack W (TF layout [rows][ncol_src]) for the skinny kernels.  segs = (first source row, rows) of each
// K segment (each padded to a multiple of 16 with zero rows).  lstm_units > 0: workgroup g owns
// hidden units 4g..4g+3, column c = gate*4 + u  <->  source column gate*H + 4g + u.
// lstm_units == 0: workgroup g owns source columns 16g..16g+15.
static int pack_skinny(t2_model* m, Packed* p, const float* W, int ncol_src, const std::vector<std::pair<int, int>>& segs,
                       int lstm_units, int ncols, const float* bias) {
    const bool lstm = lstm_units > 0;
    p->N = lstm ? lstm_units : ncols;
    p->n_wg = lstm ? ceil_div(lstm_units, 4) : ceil_div(ncols, 16);
    p->nseg = (int)segs.size();
    p->kbtot = 0;
    p->lstm_units = lstm_units;
    for (int s = 0; s < p->nseg; ++s) {
        p->seg_kb[s] = pad16(segs[s].second) / 16;
        p->seg_first[s] = segs[s].first; p->seg_rows[s] = segs[s].second;
        p->kbtot += p->seg_kb[s];
    }
    std::vector<float> out((size_t)p->n_wg * p->kbtot * 256, 0.f);
    auto colmap = [&](int g, int c) -> int {
        if (lstm) {
            const int gate = c >> 2, u = 4 * g + (c & 3);
            return u < lstm_units ? gate * lstm_units + u : -1;
        }
        const int col = 16 * g + c;
        return col < ncols ? col : -1;
    };
    for (int g = 0; g < p->n_wg; ++g) {
        int kb0 = 0;
        for (int s = 0; s < p->nseg; ++s) {
            for (int kbl = 0; kbl < p->seg_kb[s]; ++kbl)
                for (int kg = 0; kg < 4; ++kg)
                    for (int c = 0; c < 16; ++c) {
                        const int col = colmap(g, c);
                        if (col < 0) continue;
                        for (int j = 0; j < 4; ++j) {
                            const int k = kbl * 16 + kg * 4 + j;
                            if (k >= segs[s].second) continue;
                            out[(((size_t)g * p->kbtot + kb0 + kbl) * 64 + kg * 16 + c) * 4 + j] =
                                W[(size_t)(segs[s].first + k) * ncol_src + col];
                        }
                    }
            kb0 += p->seg_kb[s];
        }
    }
    T2_TRY(upload(m, &p->w, out));
    if (bias != nullptr) {
        if (lstm) {
            std::vector<float> pb((size_t)p->n_wg * 16, 0.f);
            for (int g = 0; g < p->n_wg; ++g)
                for (int c = 0; c < 16; ++c) {
                    const int col = colmap(g, c);
                    if (col >= 0) pb[(size_t)g * 16 + c] = bias[col];
                }
            T2_TRY(upload(m, &p->bias, pb));
        } else {
            std::vector<float> pb(bias, bias + ncols);
            T2_TRY(upload(m, &p->bias, pb));
        }
    }
    return T2_OK;
}

static int make_conv(t2_model* m, ConvLayer* L, const std::string& prefix, int k, int cin, int cout, bool bn) {
    const HostWeight* w;
    T2_TRY(get_w(m, prefix + "/kernel", &w, {k, cin, cout}));
    L->k = k; L->cin = cin; L->cout = cout;
    T2_TRY(upload(m, &L->w, w->data));
    if (cin % CG_BK == 0) {
        const size_t K = (size_t)k * cin;
        std::vector<float> t(K * cout);
        for (size_t kk = 0; kk < K; ++kk)
            for (int n = 0; n < cout; ++n) t[(size_t)n * K + kk] = w->data[kk * cout + n];
        T2_TRY(upload(m, &L->wt, t));
    }
    if (bn) {
        const HostWeight *g, *b, *mu, *var;
        const std::string p = prefix + "/batch_normalization/";
        T2_TRY(get_w(m, p + "gamma", &g, {cout}));
        T2_TRY(get_w(m, p + "beta", &b, {cout}));
        T2_TRY(get_w(m, p + "moving_mean", &mu, {cout}));
        T2_TRY(get_w(m, p + "moving_variance", &var, {cout}));
        std::vector<float> sc(cout), sh(cout);
        for (int i = 0; i < cout; ++i) {
            // inference batch norm folded to one affine: y = x*scale + shift
            sc[i] = g->data[i] / sqrtf(var->data[i] + m->d.bn_eps);
            sh[i] = b->data[i] - mu->data[i] * sc[i];
        }
        T2_TRY(upload(m, &L->scale, sc));
        T2_TRY(upload(m, &L->shift, sh));
    }
    return T2_OK;
}

// ------------------------------------------------------------------------------------------------
// ctx / model
// ------------------------------------------------------------------------------------------------
static int noalloc_setup(int device);     // train.h
extern "C" int t2_ctx_create(int device, void* hip_stream, t2_ctx** out) {
    if (!out) return fail(T2_EINVAL, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(T2_EINVAL, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(T2_EHIP, "device %d is %s; libt2mi is built for gfx950 only", device, prop.gcnArchName);
    T2_TRY(noalloc_setup(device));        // per-device address of the allocation-free entry point's scratch array (train.h)
    t2_ctx* c = new t2_ctx();
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(T2_EHIP, "hipStreamCreate failed: %s", hipGetErrorString(e)); }
        c->own_stream = true;
    }
    *out = c;
    return T2_OK;
}

extern "C" int t2_ctx_destroy(t2_ctx* c) {
    if (!c) return T2_OK;
    if (c->red_ws) (void)hipFree(c->red_ws);
    if (c->voc_ws) (void)hipFree(c->voc_ws);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
    return T2_OK;
}

extern "C" int t2_ctx_synchronize(t2_ctx* c) {
    if (!c) return fail(T2_EINVAL, "ctx is null");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return T2_OK;
}

extern "C" int t2_model_create(t2_ctx* ctx, const t2_dims* d, t2_model** out) {
    if (!ctx || !d || !out) return fail(T2_EINVAL, "null argument");
    // same checks the reference constructors make
    if (d->enc_units <= 0 || d->enc_units % 2 != 0) return fail(T2_EINVAL, "enc_units must be even (tacotron_v2.py:40)");
    if (d->n_feed_frame < 1 || d->n_feed_frame > d->outputs_per_step) return fail(T2_EINVAL, "n_feed_frame <= r required (helpers.py:55)");
    const float zm = std::min(d->zoneout_cell, d->zoneout_output), zs = std::max(d->zoneout_cell, d->zoneout_output);
    if (zm < 0.f || zs > 1.f) return fail(T2_EINVAL, "One/both provided Zoneout factors are not in [0, 1]");
    if (d->num_prenet_layers < 1 || d->num_prenet_layers > T2_MAX_PRENET_LAYERS) return fail(T2_EINVAL, "1..%d prenet layers supported", T2_MAX_PRENET_LAYERS);
    const int pos[] = {d->num_symbols, d->embedding_dim, d->enc_kernel, d->attention_lstm_units, d->decoder_lstm_units,
                       d->attention_dim, d->attention_filters, d->attention_kernel, d->num_mels, d->outputs_per_step,
                       d->postnet_kernel, d->postnet_channels};
    for (int v : pos) if (v < 1) return fail(T2_EINVAL, "every size must be >= 1");
    if (d->enc_conv_layers < 0 || d->postnet_layers < 1) return fail(T2_EINVAL, "bad layer count");
    for (int i = 0; i < d->num_prenet_layers; ++i) if (d->prenet_units[i] < 1) return fail(T2_EINVAL, "bad prenet size");
    t2_model* m = new t2_model();
    m->ctx = ctx;
    m->d = *d;
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && n_cu > 0) m->n_cu = n_cu;
    *out = m;
    return T2_OK;
}

extern "C" int t2_model_set_weight(t2_model* m, const char* name, const float* host_ptr, const int64_t* shape, int rank) {
    if (!m || !name || !host_ptr || !shape || rank < 1 || rank > 4) return fail(T2_EINVAL, "bad argument");
    if (m->finalized) return fail(T2_ESTATE, "model already finalized");
    HostWeight w;
    size_t n = 1;
    for (int i = 0; i < rank; ++i) {
        if (shape[i] < 1) return fail(T2_EINVAL, "bad shape");
        w.shape.push_back(shape[i]);
        n *= (size_t)shape[i];
    }
    w.data.assign(host_ptr, host_ptr + n);
    m->hw[name] = std::move(w);
    return T2_OK;
}

extern "C" int t2_model_set_option(t2_model* m, const char* name, int value) {
    if (!m || !name) return fail(T2_EINVAL, "null argument");
    if (!strcmp(name, "graph")) m->opt_graph = value;
    else if (!strcmp(name, "chunk")) { if (value < 1 || value > 64) return fail(T2_EINVAL, "chunk 1..64"); m->opt_chunk = value; }
    else if (!strcmp(name, "splitk")) {
        m->opt_splitk = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "hoist")) {
        m->opt_hoist = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "att3")) {
        m->opt_att3 = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "conv_generic")) m->opt_conv_generic = value;
    else if (!strcmp(name, "tfproj")) {
        m->opt_tfproj = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "projfuse")) {
        m->opt_projfuse = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "sp_att")) {
        m->opt_sp_att = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "spread")) {
        m->opt_spread = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "lstm_only")) {
        m->opt_lstm_only = value;
        for (t2_state* s : m->states) for (int i = 0; i < 2; ++i) if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
    }
    else if (!strcmp(name, "persist")) { m->opt_persist = value; m->persist_holdoff = 0; }
    else if (!strcmp(name, "tf2")) m->opt_tf2 = value;
    else if (!strcmp(name, "inject_timeout")) m->opt_inject_timeout = value;
    else if (!strcmp(name, "bwd_splitk")) m->opt_bwd_splitk = value;
    else if (!strcmp(name, "bwd_persist")) m->opt_bwd_persist = value;
    else if (!strcmp(name, "conv_mi")) m->opt_conv_mi = value;
    else if (!strcmp(name, "conv_kt")) m->opt_conv_kt = value;
    else if (!strcmp(name, "enc_persist")) m->opt_enc_persist = value;
    else if (!strcmp(name, "time_lstm")) { m->opt_time_lstm = value; g_timer.used = 0; }
    else return fail(T2_EINVAL, "unknown option %s", name);
    return T2_OK;
}

// the attention kernel's dynamic LDS goes beyond the 64 KiB default
static int set_attention_func_attributes() {
    const void* fns[] = {
            (const void*)attention_step_kernel<32, 2, 1, 0>, (const void*)attention_step_kernel<32, 2, 1, 2>, (const void*)attention_step_kernel<32, 2, 1, 4>,
            (const void*)attention_step_kernel<32, 2, 2, 0>, (const void*)attention_step_kernel<32, 2, 2, 2>, (const void*)attention_step_kernel<32, 2, 2, 4>,
            (const void*)attention_step_kernel<32, 2, 4, 0>, (const void*)attention_step_kernel<32, 2, 4, 2>, (const void*)attention_step_kernel<32, 2, 4, 4>,
            (const void*)attention_step_kernel<64, 4, 1, 0>, (const void*)attention_step_kernel<64, 4, 2, 0>, (const void*)attention_step_kernel<64, 4, 4, 0>,
            (const void*)attention_step_kernel<32, 4, 1, 0>, (const void*)attention_step_kernel<32, 4, 2, 0>, (const void*)attention_step_kernel<32, 4, 4, 0>};
        for (const void* f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    return T2_OK;
}

extern "C" int t2_model_finalize(t2_model* m) {
    if (!m) return fail(T2_EINVAL, "null model");
    if (m->finalized) return T2_OK;
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    const HostWeight* w;
    const HostWeight* b;
    T2_TRY(get_w(m, "embedding", &w, {d.num_symbols, d.embedding_dim}));
    T2_TRY(upload(m, &m->embedding, w->data));
    int cin = d.embedding_dim;
    m->enc_convs.resize(d.enc_conv_layers);
    for (int i = 0; i < d.enc_conv_layers; ++i) {
        T2_TRY(make_conv(m, &m->enc_convs[i], "encoder/conv1d_" + std::to_string(i), d.enc_kernel, cin, d.enc_units, true));
        {
            const HostWeight *g, *bb;
            const std::string bp = "encoder/conv1d_" + std::to_string(i) + "/batch_normalization/";
            T2_TRY(get_w(m, bp + "gamma", &g, {d.enc_units}));
            T2_TRY(get_w(m, bp + "beta", &bb, {d.enc_units}));
            float *dg = nullptr, *db = nullptr;
            T2_TRY(upload(m, &dg, g->data));
            T2_TRY(upload(m, &db, bb->data));
            m->enc_bn_gamma.push_back(dg); m->enc_bn_beta.push_back(db);
        }
        cin = d.enc_units;
    }
    const int He = d.enc_units / 2;
    const char* dirs[2] = {"fw", "bw"};
    for (int dir = 0; dir < 2; ++dir) {
        const std::string p = std::string("encoder/bidirectional_rnn/") + dirs[dir] + "/lstm_cell/";
        T2_TRY(get_w(m, p + "kernel", &w, {cin + He, 4 * He}));
        T2_TRY(get_w(m, p + "bias", &b, {4 * He}));
        T2_TRY(pack_skinny(m, &m->enc_lstm[dir], w->data.data(), 4 * He, {{0, cin}, {cin, He}}, He, 0, b->data.data()));
    }
    const int E = d.enc_units, U = d.attention_dim, A = d.attention_lstm_units, D = d.decoder_lstm_units;
    const int F = d.attention_filters, K = d.attention_kernel, M = d.num_mels, r = d.outputs_per_step;
    T2_TRY(get_w(m, "attention/memory_layer/kernel", &w, {E, U}));
    m->memory_layer.k = 1; m->memory_layer.cin = E; m->memory_layer.cout = U;
    T2_TRY(upload(m, &m->memory_layer.w, w->data));
    T2_TRY(get_w(m, "attention/query_layer/kernel", &w, {A, U}));
    T2_TRY(upload(m, &m->wq, w->data));
    T2_TRY(pack_skinny(m, &m->wq_pk, w->data.data(), U, {{0, A}}, 0, U, nullptr));
    T2_TRY(get_w(m, "attention/location_features_convolution/kernel", &w, {K, 1, F}));
    T2_TRY(upload(m, &m->conv_w, w->data));
    T2_TRY(get_w(m, "attention/location_features_convolution/bias", &w, {F}));
    T2_TRY(upload(m, &m->conv_b, w->data));
    T2_TRY(get_w(m, "attention/location_features_layer/kernel", &w, {F, U}));
    T2_TRY(upload(m, &m->wloc, w->data));
    T2_TRY(get_w(m, "attention/attention_variable", &w, {U}));
    T2_TRY(upload(m, &m->v_a, w->data));
    T2_TRY(get_w(m, "attention/attention_bias", &w, {U}));
    T2_TRY(upload(m, &m->b_a, w->data));
    int pin = M * d.n_feed_frame;
    for (int i = 0; i < d.num_prenet_layers; ++i) {
        const std::string p = "decoder/prenet_" + std::to_string(i) + "/dense/";
        T2_TRY(get_w(m, p + "kernel", &w, {pin, d.prenet_units[i]}));
        T2_TRY(get_w(m, p + "bias", &b, {d.prenet_units[i]}));
        T2_TRY(pack_skinny(m, &m->prenet[i], w->data.data(), d.prenet_units[i], {{0, pin}}, 0, d.prenet_units[i], b->data.data()));
        ConvLayer& Lp = m->prenet_plain[i];
        Lp.k = 1; Lp.cin = pin; Lp.cout = d.prenet_units[i];
        T2_TRY(upload(m, &Lp.w, w->data));
        T2_TRY(upload(m, &Lp.shift, b->data));
        pin = d.prenet_units[i];
    }
    T2_TRY(get_w(m, "decoder/attention_lstm/kernel", &w, {pin + E + A, 4 * A}));
    T2_TRY(get_w(m, "decoder/attention_lstm/bias", &b, {4 * A}));
    T2_TRY(pack_skinny(m, &m->att_lstm, w->data.data(), 4 * A, {{0, pin}, {pin, E}, {pin + E, A}}, A, 0, b->data.data()));
    T2_TRY(get_w(m, "decoder/lstm_1/kernel", &w, {A + E + D, 4 * D}));
    T2_TRY(get_w(m, "decoder/lstm_1/bias", &b, {4 * D}));
    T2_TRY(pack_skinny(m, &m->lstm1, w->data.data(), 4 * D, {{0, A}, {A, E}, {A + E, D}}, D, 0, b->data.data()));
    T2_TRY(get_w(m, "decoder/lstm_2/kernel", &w, {D + D, 4 * D}));
    T2_TRY(get_w(m, "decoder/lstm_2/bias", &b, {4 * D}));
    T2_TRY(pack_skinny(m, &m->lstm2, w->data.data(), 4 * D, {{0, D}, {D, D}}, D, 0, b->data.data()));
    {   // mel and stop projections share one kernel: columns [0, M*r) mel, column M*r the stop logit
        const HostWeight *wm, *bm, *wsn, *bs;
        T2_TRY(get_w(m, "decoder/out_projection/kernel", &wm, {D, M * r}));
        T2_TRY(get_w(m, "decoder/out_projection/bias", &bm, {M * r}));
        T2_TRY(get_w(m, "decoder/stop_token_projection/kernel", &wsn, {D, 1}));
        T2_TRY(get_w(m, "decoder/stop_token_projection/bias", &bs, {1}));
        const int N = M * r + 1;
        std::vector<float> wc((size_t)D * N), bc(N);
        for (int k = 0; k < D; ++k) {
            for (int c = 0; c < M * r; ++c) wc[(size_t)k * N + c] = wm->data[(size_t)k * M * r + c];
            wc[(size_t)k * N + M * r] = wsn->data[k];
        }
        for (int c = 0; c < M * r; ++c) bc[c] = bm->data[c];
        bc[M * r] = bs->data[0];
        T2_TRY(pack_skinny(m, &m->proj, wc.data(), N, {{0, D}}, 0, N, bc.data()));
        {   // the same operand in plain layout, columns padded to a multiple of 4 (batched projection of teacher-forced decodes)
            const int Np = (N + 3) & ~3;
            std::vector<float> wp((size_t)D * Np, 0.f), bp(Np, 0.f);
            for (int k = 0; k < D; ++k) for (int c = 0; c < N; ++c) wp[(size_t)k * Np + c] = wc[(size_t)k * N + c];
            for (int c = 0; c < N; ++c) bp[c] = bc[c];
            m->proj_plain.k = 1; m->proj_plain.cin = D; m->proj_plain.cout = Np;
            T2_TRY(upload(m, &m->proj_plain.w, wp));
            T2_TRY(upload(m, &m->proj_plain.shift, bp));
        }
        // Free-running decode: the first PreNet layer of step t+1 reads only the fed-back frames of step t's mel output
        // (helpers.py:86), so relu((y.Wm + bm)[fed] . Wp + bp) = relu(y . (Wm[:, fed] . Wp) + (bm[fed] . Wp + bp)): the
        // composed columns are appended to the projection (16-column tile aligned) and the PreNet-0 launch disappears.
        // Composition in double, rounded once.
        {
            const HostWeight *wp, *bp;
            const int nfm = M * d.n_feed_frame, c0 = M * (r - d.n_feed_frame), P0 = d.prenet_units[0], N0 = pad16(N), N2 = N0 + P0;
            T2_TRY(get_w(m, "decoder/prenet_0/dense/kernel", &wp, {nfm, P0}));
            T2_TRY(get_w(m, "decoder/prenet_0/dense/bias", &bp, {P0}));
            std::vector<float> w2((size_t)D * N2, 0.f), b2(N2, 0.f);
            std::vector<double> acc(P0);
            for (int k = 0; k < D; ++k) {
                for (int c = 0; c < N; ++c) w2[(size_t)k * N2 + c] = wc[(size_t)k * N + c];
                std::fill(acc.begin(), acc.end(), 0.0);
                for (int j = 0; j < nfm; ++j) {
                    const double a = wm->data[(size_t)k * M * r + c0 + j];
                    const float* wr = wp->data.data() + (size_t)j * P0;
                    for (int q = 0; q < P0; ++q) acc[q] += a * (double)wr[q];
                }
                for (int q = 0; q < P0; ++q) w2[(size_t)k * N2 + N0 + q] = (float)acc[q];
            }
            for (int c = 0; c < N; ++c) b2[c] = bc[c];
            for (int q = 0; q < P0; ++q) {
                double a = bp->data[q];
                for (int j = 0; j < nfm; ++j) a += (double)bm->data[c0 + j] * (double)wp->data[(size_t)j * P0 + q];
                b2[N0 + q] = (float)a;
            }
            T2_TRY(pack_skinny(m, &m->proj_fused, w2.data(), N2, {{0, D}}, 0, N2, b2.data()));
            m->proj_fused_ok = true;
        }
    }
    cin = M;
    m->post_convs.resize(d.postnet_layers);
    for (int i = 1; i <= d.postnet_layers; ++i) {
        T2_TRY(make_conv(m, &m->post_convs[i - 1], "postnet/conv1d_" + std::to_string(i), d.postnet_kernel, cin, d.postnet_channels, true));
        {
            const HostWeight *g, *bb;
            const std::string bp = "postnet/conv1d_" + std::to_string(i) + "/batch_normalization/";
            T2_TRY(get_w(m, bp + "gamma", &g, {d.postnet_channels}));
            T2_TRY(get_w(m, bp + "beta", &bb, {d.postnet_channels}));
            float *dg = nullptr, *db = nullptr;
            T2_TRY(upload(m, &dg, g->data));
            T2_TRY(upload(m, &db, bb->data));
            m->post_bn_gamma.push_back(dg); m->post_bn_beta.push_back(db);
        }
        cin = d.postnet_channels;
    }
    T2_TRY(get_w(m, "postnet/projection/kernel", &w, {d.postnet_channels, M}));
    T2_TRY(get_w(m, "postnet/projection/bias", &b, {M}));
    m->post_proj.k = 1; m->post_proj.cin = d.postnet_channels; m->post_proj.cout = M;
    T2_TRY(upload(m, &m->post_proj.w, w->data));
    T2_TRY(upload(m, &m->post_proj.shift, b->data));
    HIP_TRY(hipHostMalloc((void**)&m->h_ctl, 8 * sizeof(int), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&m->h_flag, 2 * 8 * sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&m->flag_ev[i], hipEventDisableTiming));
    if (d.attention_filters > 64 || U > 256) return fail(T2_EINVAL, "attention_filters <= 64 and attention_dim <= 256 supported");
    T2_TRY(set_attention_func_attributes());
    {
        const void* pk[] = {(const void*)pz::decode_kernel<1, 0>, (const void*)pz::decode_kernel<2, 0>, (const void*)pz::decode_kernel<4, 0>,
                            (const void*)pz::decode_kernel<1, 1>, (const void*)pz::decode_kernel<2, 1>, (const void*)pz::decode_kernel<4, 1>,
                            (const void*)pz::decode_kernel<1, 2>, (const void*)pz::decode_kernel<2, 2>, (const void*)pz::decode_kernel<4, 2>,
                            (const void*)pz::decode_kernel<1, 3>, (const void*)pz::decode_kernel<2, 3>, (const void*)pz::decode_kernel<4, 3>,
                            (const void*)pz::decode_kernel<1, 4>, (const void*)pz::decode_kernel<2, 4>, (const void*)pz::decode_kernel<4, 4>};
        for (const void* f : pk) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
    // the host copies stay: a trainer packs the transposed operands of the backward pass from them
    m->finalized = true;
    return T2_OK;
}

static void state_free(t2_state* s);

extern "C" int t2_model_destroy(t2_model* m) {
    if (!m) return T2_OK;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    for (t2_state* s : m->states) state_free(s);
    for (void* p : m->allocs) hipFree(p);
    if (m->enc_ws) hipFree(m->enc_ws);
    if (m->post_ws) hipFree(m->post_ws);
    if (m->h_ctl) hipHostFree(m->h_ctl);
    if (m->h_flag) hipHostFree(m->h_flag);
    for (int i = 0; i < 2; ++i) if (m->flag_ev[i]) hipEventDestroy(m->flag_ev[i]);
    delete m;
    return T2_OK;
}

// ------------------------------------------------------------------------------------------------
// conv / dense on large M
// ------------------------------------------------------------------------------------------------
static int launch_conv(t2_model* m, const ConvLayer& L, const float* x, int ldx, float* y, int ldy, int Mtot, int T,
                       int act, const float* res, int ldr) {
    ConvArgs a{};
    a.x = x; a.ldx = ldx; a.w = L.w; a.scale = L.scale; a.shift = L.shift; a.res = res; a.ldr = ldr; a.y = y; a.ldy = ldy;
    a.Mtot = Mtot; a.T = T; a.Cin = L.cin; a.N = L.cout; a.ktaps = L.k; a.pad_l = L.pad_l >= 0 ? L.pad_l : (L.k - 1) / 2; a.act = act;
    a.vec_a = (L.cin % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x) % 16 == 0);
    a.vec_b = (L.cout % 4 == 0) && (((uintptr_t)L.w) % 16 == 0);
    dim3 grid(ceil_div(L.cout, CG_BN), ceil_div(Mtot, CG_BM));
    if (a.vec_a && a.vec_b && L.cin % CG_BK == 0 && !m->opt_conv_generic) {
        const long long t64 = (long long)grid.x * ceil_div(Mtot, 64);
        const int nk = L.k * L.cin / CG_BK;
        if (L.part != nullptr && 2 * t64 <= m->n_cu && nk >= 128) {
            // few output tiles against a long reduction: the k-tiles over gridDim.z slices, partial sums added in slice order
            int z = (int)std::min<long long>({16LL, (4LL * m->n_cu + t64 - 1) / t64, (long long)nk / 32,
                                              (long long)(L.part_floats / ((size_t)Mtot * L.cout))});
            if (z >= 2) {
                a.splits = z; a.part = L.part;
                grid.y = ceil_div(Mtot, 64); grid.z = z;
                hipLaunchKernelGGL((conv_gemm_fast_kernel<1, true>), grid, dim3(256), 0, m->ctx->stream, a);
                hipLaunchKernelGGL(conv_split_finish_kernel, dim3(std::min(2048, ceil_div(Mtot * L.cout, 256))), dim3(256), 0, m->ctx->stream, a);
                HIP_TRY(hipGetLastError());
                return T2_OK;
            }
        }
    }
    if (a.vec_a && L.wt != nullptr && L.cin % CG_BK == 0 && !m->opt_conv_generic && m->opt_conv_kt) {
        // K-contiguous LDS tiles on the transposed kernel.  Tile height: 64 rows while 128-row tiles would leave fewer than ~3
        // workgroups per CU; otherwise 128 or 160 rows, whichever leaves the fuller last round on the CUs
        a.wt = L.wt;
        const long long t128 = (long long)grid.x * grid.y, t160 = (long long)grid.x * ceil_div(Mtot, 160);
        auto fill = [&](long long tiles) { const double q = (double)tiles / m->n_cu; return q / std::ceil(q); };
        if (t128 < 3LL * m->n_cu || m->opt_conv_mi == 1) {
            grid.y = ceil_div(Mtot, 64);
            hipLaunchKernelGGL((conv_gemm_kt_kernel<1, 2>), grid, dim3(256), 0, m->ctx->stream, a);
        } else if (m->opt_conv_mi != 2 && (m->opt_conv_mi == 5 || fill(t160) > fill(t128) + 0.02)) {
            grid.y = ceil_div(Mtot, 160);
            hipLaunchKernelGGL((conv_gemm_kt_kernel<5, 4>), grid, dim3(256), 0, m->ctx->stream, a);
        } else {
            hipLaunchKernelGGL((conv_gemm_kt_kernel<2, 2>), grid, dim3(256), 0, m->ctx->stream, a);
        }
    } else if (a.vec_a && a.vec_b && L.cin % CG_BK == 0 && !m->opt_conv_generic) {
        // fewer than ~3 workgroups of 128 rows per CU: 64-row tiles (twice the workgroups) keep more waves per SIMD
        if ((long long)grid.x * grid.y < 3LL * m->n_cu || m->opt_conv_mi == 1) {
            grid.y = ceil_div(Mtot, 64);
            hipLaunchKernelGGL(conv_gemm_fast_kernel<1>, grid, dim3(256), 0, m->ctx->stream, a);
        } else {
            hipLaunchKernelGGL(conv_gemm_fast_kernel<2>, grid, dim3(256), 0, m->ctx->stream, a);
        }
    } else
        hipLaunchKernelGGL(conv_gemm_kernel, grid, dim3(256), 0, m->ctx->stream, a);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// ------------------------------------------------------------------------------------------------
// stand-alone cells of the reference's RNNCell / Layer protocol (API parity; the decode loops never use these)
// ------------------------------------------------------------------------------------------------
// LSTMCell pointwise math + the inference zoneout blend on the gate pre-activations z [B][4H] (i | j | f | o; bias included)
__global__ __launch_bounds__(256) void lstm_cell_pointwise_kernel(const float* __restrict__ z, const float* __restrict__ c_prev,
                                                                  const float* __restrict__ h_prev, int B, int H, float zc, float zh, float fb,
                                                                  float* __restrict__ h_raw, float* __restrict__ c_new, float* __restrict__ h_new) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * H; i += gridDim.x * blockDim.x) {
        const int b = i / H, u = i - b * H;
        const float* zr = z + (size_t)b * 4 * H;
        const float cp = c_prev[i], hp = h_prev[i];
        const float cn = sigmoidf_(zr[2 * H + u] + fb) * cp + sigmoidf_(zr[u]) * tanhf(zr[H + u]);
        const float hn = sigmoidf_(zr[3 * H + u]) * tanhf(cn);
        h_raw[i] = hn;
        c_new[i] = (1.0f - zc) * cn + zc * cp;
        h_new[i] = (1.0f - zh) * hn + zh * hp;
    }
}

// the sampled branch of ZoneoutLSTMCell (modules.py:274-276): binary keep masks from t2_u01(seed, stream | stream + 1, t, row, unit)
__global__ __launch_bounds__(256) void lstm_cell_pointwise_train_kernel(const float* __restrict__ z, const float* __restrict__ c_prev,
                                                                        const float* __restrict__ h_prev, int B, int H, float zc, float zh, float fb,
                                                                        uint32_t seed, int stream, int t, float* __restrict__ h_raw,
                                                                        float* __restrict__ c_new, float* __restrict__ h_new) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * H; i += gridDim.x * blockDim.x) {
        const int b = i / H, u = i - b * H;
        const float* zr = z + (size_t)b * 4 * H;
        const float cp = c_prev[i], hp = h_prev[i];
        const float cn = sigmoidf_(zr[2 * H + u] + fb) * cp + sigmoidf_(zr[u]) * tanhf(zr[H + u]);
        const float hn = sigmoidf_(zr[3 * H + u]) * tanhf(cn);
        h_raw[i] = hn;
        c_new[i] = (1.0f - zc) * dropout_train(cn - cp, zc, t2_u01(seed, stream, t, b, u)) + cp;
        h_new[i] = (1.0f - zh) * dropout_train(hn - hp, zh, t2_u01(seed, stream + 1, t, b, u)) + hp;
    }
}

// tf.layers.dropout(training=True) with the library's counter-based generator: y = x / keep * [u01(seed, stream, t, row, col) >= rate]
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, long long rows, int cols, float rate, uint32_t seed, int stream, int t,
                                                      float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols;
        y[i] = dropout_train(x[i], rate, t2_u01(seed, stream, t, (uint32_t)r, (uint32_t)(i - r * cols)));
    }
}

// y = act(x . kernel + bias) on plain operands (tf.layers.Dense: PreNet.call modules.py:88-91 without dropout; any [rows][K] input)
static int dense_plain(t2_ctx* ctx, const float* x, const float* kernel, const float* bias, int rows, int K, int N, int act, float* y) {
    ConvArgs a{};
    a.x = x; a.ldx = K; a.w = kernel; a.scale = nullptr; a.shift = bias; a.res = nullptr; a.ldr = 0; a.y = y; a.ldy = N;
    a.Mtot = rows; a.T = rows; a.Cin = K; a.N = N; a.ktaps = 1; a.pad_l = 0; a.act = act;
    a.vec_a = (K % 4 == 0) && (((uintptr_t)x) % 16 == 0);
    a.vec_b = (N % 4 == 0) && (((uintptr_t)kernel) % 16 == 0);
    hipLaunchKernelGGL(conv_gemm_kernel, dim3(ceil_div(N, CG_BN), ceil_div(rows, CG_BM)), dim3(256), 0, ctx->stream, a);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_dense(t2_ctx* ctx, const float* x, const float* kernel, const float* bias, int rows, int K, int N, int activation, float* y) {
    if (!ctx || !x || !kernel || !y) return fail(T2_EINVAL, "null argument");
    if (rows < 1 || K < 1 || N < 1 || activation < 0 || activation > 2) return fail(T2_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return dense_plain(ctx, x, kernel, bias, rows, K, N, activation, y);
}

extern "C" int t2_dropout(t2_ctx* ctx, const float* x, long long rows, int cols, float rate, uint32_t seed, int stream, int t, float* y) {
    if (!ctx || !x || !y) return fail(T2_EINVAL, "null argument");
    if (rows < 1 || cols < 1 || !(rate >= 0.f && rate < 1.f)) return fail(T2_EINVAL, "bad argument (rate must be in [0, 1))");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (rows * cols + 255) / 256))), dim3(256), 0, ctx->stream,
                       x, rows, cols, rate, seed, stream, t, y);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_zoneout_lstm_cell_train(t2_ctx* ctx, const float* xh, const float* c_prev, const float* h_prev, const float* kernel,
                                          const float* bias, int B, int K, int H, float zoneout_cell, float zoneout_output, float forget_bias,
                                          uint32_t seed, int stream, int t, float* z_ws, float* h_raw, float* c_new, float* h_new) {
    if (!ctx || !xh || !c_prev || !h_prev || !kernel || !bias || !z_ws || !h_raw || !c_new || !h_new) return fail(T2_EINVAL, "null argument");
    if (B < 1 || H < 1 || K <= H) return fail(T2_EINVAL, "bad shape: the kernel has K = inputs + H rows");
    const float zm = std::min(zoneout_cell, zoneout_output), zs = std::max(zoneout_cell, zoneout_output);
    if (zm < 0.f || zs >= 1.f) return fail(T2_EINVAL, "sampled zoneout factors must be in [0, 1)");
    HIP_TRY(hipSetDevice(ctx->device));
    T2_TRY(dense_plain(ctx, xh, kernel, bias, B, K, 4 * H, 0, z_ws));
    hipLaunchKernelGGL(lstm_cell_pointwise_train_kernel, dim3(std::max(1, std::min(1024, ceil_div(B * H, 256)))), dim3(256), 0, ctx->stream, z_ws, c_prev,
                       h_prev, B, H, zoneout_cell, zoneout_output, forget_bias, seed, stream, t, h_raw, c_new, h_new);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_zoneout_lstm_cell(t2_ctx* ctx, const float* xh, const float* c_prev, const float* h_prev, const float* kernel,
                                    const float* bias, int B, int K, int H, float zoneout_cell, float zoneout_output, float forget_bias,
                                    float* z_ws, float* h_raw, float* c_new, float* h_new) {
    if (!ctx || !xh || !c_prev || !h_prev || !kernel || !bias || !z_ws || !h_raw || !c_new || !h_new) return fail(T2_EINVAL, "null argument");
    if (B < 1 || H < 1 || K <= H) return fail(T2_EINVAL, "bad shape: the kernel has K = inputs + H rows");
    const float zm = std::min(zoneout_cell, zoneout_output), zs = std::max(zoneout_cell, zoneout_output);
    if (zm < 0.f || zs > 1.f) return fail(T2_EINVAL, "One/both provided Zoneout factors are not in [0, 1]");
    HIP_TRY(hipSetDevice(ctx->device));
    T2_TRY(dense_plain(ctx, xh, kernel, bias, B, K, 4 * H, 0, z_ws));
    hipLaunchKernelGGL(lstm_cell_pointwise_kernel, dim3(std::max(1, std::min(1024, ceil_div(B * H, 256)))), dim3(256), 0, ctx->stream, z_ws, c_prev, h_prev, B,
                       H, zoneout_cell, zoneout_output, forget_bias, h_raw, c_new, h_new);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int ensure_ws(void** ws, size_t* have, size_t need) {
    if (*have >= need) return T2_OK;
    if (*ws) HIP_TRY(hipFree(*ws));
    *ws = nullptr; *have = 0;
    HIP_TRY(hipMalloc(ws, need));
    *have = need;
    return T2_OK;
}

static int check_batch(int B, int T) {
    if (B < 1 || B > T2_MAX_BATCH) return fail(T2_EINVAL, "batch %d outside 1..%d", B, T2_MAX_BATCH);
    if (T < 1) return fail(T2_EINVAL, "sequence length must be >= 1");
    return T2_OK;
}

static inline int mb_for(int B) { const int mb = ceil_div(B, 16); return mb == 3 ? 4 : mb; }

// NPW (groups per wave, fully unrolled ring) instantiations; anything else runs the runtime loop.
template <int MB, bool ENC>
static inline void launch_lstm_mb(dim3 grid, hipStream_t st, const SkinnyCore& c, const LstmEpi& e) {
    constexpr int G = (MB >= 4) ? T2_G4 : 4;
    const int base = (c.kbtot / G) / T2_SKINNY_WAVES;
    const dim3 blk(T2_SKINNY_THREADS);
    if (ENC) { hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 0>), grid, blk, 0, st, c, e); return; }
    switch (base) {
        case 2: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 2>), grid, blk, 0, st, c, e); break;
        case 3: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 3>), grid, blk, 0, st, c, e); break;
        case 4: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 4>), grid, blk, 0, st, c, e); break;
        case 5: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 5>), grid, blk, 0, st, c, e); break;
        case 6: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 6>), grid, blk, 0, st, c, e); break;
        case 7: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 7>), grid, blk, 0, st, c, e); break;
        case 8: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 8>), grid, blk, 0, st, c, e); break;
        case 10: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 10>), grid, blk, 0, st, c, e); break;
        default: hipLaunchKernelGGL((lstm_step_kernel<MB, ENC, 0>), grid, blk, 0, st, c, e); break;
    }
}
template <bool ENC>
static inline void launch_lstm(int MB, dim3 grid, hipStream_t st, const SkinnyCore& c, const LstmEpi& e) {
    if (MB == 1) launch_lstm_mb<1, ENC>(grid, st, c, e);
    else if (MB == 2) launch_lstm_mb<2, ENC>(grid, st, c, e);
    else launch_lstm_mb<4, ENC>(grid, st, c, e);
}

template <int MB>
static inline void launch_partial_mb(dim3 grid, hipStream_t st, const SkinnyCore& c, float* z) {
    constexpr int G = (MB >= 4) ? T2_G4 : 4;
    const int base = (c.kbtot / G) / T2_SKINNY_WAVES;
    const dim3 blk(T2_SKINNY_THREADS);
    switch (base) {
        case 2: hipLaunchKernelGGL((partial_step_kernel<MB, 2>), grid, blk, 0, st, c, z); break;
        case 3: hipLaunchKernelGGL((partial_step_kernel<MB, 3>), grid, blk, 0, st, c, z); break;
        case 4: hipLaunchKernelGGL((partial_step_kernel<MB, 4>), grid, blk, 0, st, c, z); break;
        case 6: hipLaunchKernelGGL((partial_step_kernel<MB, 6>), grid, blk, 0, st, c, z); break;
        default: hipLaunchKernelGGL((partial_step_kernel<MB, 0>), grid, blk, 0, st, c, z); break;
    }
}
static inline void launch_partial(int MB, dim3 grid, hipStream_t st, const SkinnyCore& c, float* z) {
    if (MB == 1) launch_partial_mb<1>(grid, st, c, z);
    else if (MB == 2) launch_partial_mb<2>(grid, st, c, z);
    else launch_partial_mb<4>(grid, st, c, z);
}

static void timed_lstm(bool timing, int MB, int n_wg4, hipStream_t st, const SkinnyCore& c, const LstmEpi& e, unsigned* tickets,
                       unsigned long long* slabs) {
    const bool splitk = tickets != nullptr && MB >= 2;
    auto go = [&]() {
        if (splitk) launch_lstm_splitk(MB, n_wg4, st, c, e, tickets, slabs);
        else launch_lstm<false>(MB, dim3(n_wg4, 1), st, c, e);
    };
    if (timing) {
        if (g_timer.used + 2 > g_timer.ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) return; g_timer.ev.push_back(ev); }
        }
        (void)hipEventRecord(g_timer.ev[g_timer.used], st);
        go();
        (void)hipEventRecord(g_timer.ev[g_timer.used + 1], st);
        g_timer.used += 2;
    } else {
        go();
    }
}
template <int MB>
static inline void launch_dense_mb(dim3 grid, hipStream_t st, const SkinnyCore& c, const DenseEpi& e, const PartialJobs& pj) {
    constexpr int G = (MB >= 4) ? T2_G4 : 4;
    int base = -1;      // groups per wave common to every riding job present (else the runtime loop)
    for (int i = 0; i < 3; ++i) {
        if (pj.n[i] <= 0) continue;
        const int bi = (pj.core[i].kbtot / G) / T2_SKINNY_WAVES;
        base = (base < 0 || base == bi) ? bi : 0;
    }
    const dim3 blk(T2_SKINNY_THREADS);
    switch (base) {
        case 2: hipLaunchKernelGGL((dense_step_kernel<MB, 2>), grid, blk, 0, st, c, e, pj); break;
        case 3: hipLaunchKernelGGL((dense_step_kernel<MB, 3>), grid, blk, 0, st, c, e, pj); break;
        case 4: hipLaunchKernelGGL((dense_step_kernel<MB, 4>), grid, blk, 0, st, c, e, pj); break;
        case 6: hipLaunchKernelGGL((dense_step_kernel<MB, 6>), grid, blk, 0, st, c, e, pj); break;
        default: hipLaunchKernelGGL((dense_step_kernel<MB, 0>), grid, blk, 0, st, c, e, pj); break;
    }
}
// grid.x = workgroups of the dense role; the grid is widened to the riding job's workgroup count
static inline void launch_dense(int MB, dim3 grid, hipStream_t st, const SkinnyCore& c, const DenseEpi& e, const PartialJobs* jobs = nullptr) {
    PartialJobs pj{};
    if (jobs) pj = *jobs;
    DenseEpi e2 = e;
    e2.n_wg = grid.x;
    grid.x = std::max<unsigned>(grid.x, pj.skip + pj.n[0] + pj.n[1] + pj.n[2]);
    if (MB == 1) launch_dense_mb<1>(grid, st, c, e2, pj);
    else if (MB == 2) launch_dense_mb<2>(grid, st, c, e2, pj);
    else launch_dense_mb<4>(grid, st, c, e2, pj);
}

template <int MB>
static inline void launch_attention_mb(int B, size_t lds, hipStream_t st, const AttnArgs& a, const PartialJobs& pj) {
    const bool small = a.F <= 32 && a.U <= 128;
    constexpr int G = (MB >= 4) ? T2_G4 : 4;
    const int base = pj.n[0] > 0 ? (pj.core[0].kbtot / G) / T2_SKINNY_WAVES : 0;
    const int base1 = pj.n[1] > 0 ? (pj.core[1].kbtot / G) / T2_SKINNY_WAVES : base;
    const int base2 = pj.n[2] > 0 ? (pj.core[2].kbtot / G) / T2_SKINNY_WAVES : base;
    const dim3 grid(B + pj.n[0] + pj.n[1] + pj.n[2]), blk(T2_ATTN_THREADS);
    if (!small) {
        if (a.F <= 32) hipLaunchKernelGGL((attention_step_kernel<32, 4, MB, 0>), grid, blk, lds, st, a, pj);
        else hipLaunchKernelGGL((attention_step_kernel<64, 4, MB, 0>), grid, blk, lds, st, a, pj);
        return;
    }
    const int npw = (base == base1 && base == base2) ? base : 0;
    switch (npw) {
        case 2: hipLaunchKernelGGL((attention_step_kernel<32, 2, MB, 2>), grid, blk, lds, st, a, pj); break;
        case 4: hipLaunchKernelGGL((attention_step_kernel<32, 2, MB, 4>), grid, blk, lds, st, a, pj); break;
        default: hipLaunchKernelGGL((attention_step_kernel<32, 2, MB, 0>), grid, blk, lds, st, a, pj); break;
    }
}
static inline void launch_attention(int MB, int B, size_t lds, hipStream_t st, const AttnArgs& a, const PartialJobs* jobs) {
    PartialJobs pj{};
    if (jobs) pj = *jobs;
    if (MB == 1) launch_attention_mb<1>(B, lds, st, a, pj);
    else if (MB == 2) launch_attention_mb<2>(B, lds, st, a, pj);
    else launch_attention_mb<4>(B, lds, st, a, pj);
}

// ------------------------------------------------------------------------------------------------
// encoder
// ------------------------------------------------------------------------------------------------
static int encode_impl(t2_model* m, const int64_t* ids, const float* emb_in, const int64_t* lens, int B, int T, float* memory_out) {
    if (!m || !memory_out) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    T2_TRY(check_batch(B, T));
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int MB = mb_for(B), E = d.enc_units, He = E / 2, emb = d.embedding_dim;
    const int Mtot = B * T;
    std::vector<int> h_lens(B), h_ids;
    for (int b = 0; b < B; ++b) {
        const int64_t l = lens ? lens[b] : T;
        if (l < 0 || l > T) return fail(T2_EINVAL, "sequence length %lld outside 0..%d", (long long)l, T);
        h_lens[b] = (int)l;
    }
    if (ids) {
        h_ids.resize(Mtot);
        for (int i = 0; i < Mtot; ++i) {
            if (ids[i] < 0 || ids[i] >= d.num_symbols)
                return fail(T2_ERANGE, "embedding id %lld outside [0, %d) (modules.py:65-66)", (long long)ids[i], d.num_symbols);
            h_ids[i] = (int)ids[i];
        }
    }
    const int cw = std::max(emb, E);
    const int KPx = pad16(std::max(emb, E)), KPh = pad16(He);
    const size_t per_step = (size_t)MB * 16 * KPx;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_ids = carve((size_t)Mtot * 4), o_lens = carve((size_t)B * 4);
    const size_t o_x0 = carve((size_t)Mtot * cw * 4), o_x1 = carve((size_t)Mtot * cw * 4);
    const size_t o_xf = carve(2 * (size_t)T * per_step * 4);
    const size_t o_zero = off;
    const size_t o_c = carve(2 * (size_t)MB * 16 * He * 4);
    const size_t o_h = carve(2 * 2 * (size_t)MB * 16 * KPh * 4);
    const size_t o_ctl = carve(8 * 4);
    const size_t o_ebar = carve((size_t)2 * pz::NSHARD * 32 * 4 + 256);      // persistent BiLSTM: hop counters per direction + error word
    const size_t zero_bytes = off - o_zero;
    T2_TRY(ensure_ws(&m->enc_ws, &m->enc_ws_bytes, off));
    char* ws = (char*)m->enc_ws;
    int* d_ids = (int*)(ws + o_ids);
    int* d_lens = (int*)(ws + o_lens);
    float* x0 = (float*)(ws + o_x0);
    float* x1 = (float*)(ws + o_x1);
    float* xf = (float*)(ws + o_xf);
    float* cst = (float*)(ws + o_c);
    float* hst = (float*)(ws + o_h);
    int* ctl = (int*)(ws + o_ctl);
    // stream-ordered uploads from pageable memory are synchronous w.r.t. the host copy
    HIP_TRY(hipMemcpyAsync(d_lens, h_lens.data(), B * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(ws + o_zero, 0, zero_bytes, st));
    const float* cur;
    int cur_c;
    if (ids) {
        HIP_TRY(hipMemcpyAsync(d_ids, h_ids.data(), (size_t)Mtot * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(embedding_kernel, dim3(std::min(1024, ceil_div(Mtot * emb, 256))), dim3(256), 0, st, d_ids, m->embedding, x0, Mtot, emb);
        cur = x0;
    } else {
        cur = emb_in;
    }
    cur_c = emb;
    float* bufs[2] = {x1, x0};
    for (int i = 0; i < d.enc_conv_layers; ++i) {
        float* dst = bufs[i & 1];
        T2_TRY(launch_conv(m, m->enc_convs[i], cur, cur_c, dst, E, Mtot, T, 1, nullptr, 0));
        cur = dst;
        cur_c = E;
    }
    const int Cx = cur_c;   // BiLSTM input width
    const int KPc = pad16(Cx);
    if (KPc > KPx) return fail(T2_EINVAL, "internal: lstm input wider than workspace");
    const size_t per_step_c = (size_t)MB * 16 * KPc;
    {
        const long long n = 2LL * T * per_step_c;
        hipLaunchKernelGGL(seq_to_frag_kernel, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, st,
                           cur, d_lens, xf, B, T, Cx, MB, KPc);
    }
    HIP_TRY(hipMemsetAsync(memory_out, 0, (size_t)Mtot * E * 4, st));
    const int h_ctl[8] = {0, 0, T, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(ctl, h_ctl, sizeof(h_ctl), hipMemcpyHostToDevice, st));
    const int hsz = MB * 16 * KPh;
    // the whole recurrence as ONE launch (persist.h encoder_kernel): paper-size class only (512 inputs, 256 units per direction)
    const bool enc_persist = m->opt_enc_persist && m->persist_holdoff == 0 && Cx == 512 && He == 256 && KPh == 256 && m->n_cu >= 2 * pz::ENC_G &&
                             m->enc_lstm[0].n_wg == pz::ENC_G && m->enc_lstm[0].kbtot == 48 && 2LL * T * per_step_c * 4 < 0x7fffffffLL;
    if (enc_persist) {
        pz::EArgs a{};
        for (int dir = 0; dir < 2; ++dir) { a.w[dir] = m->enc_lstm[dir].w; a.bias[dir] = m->enc_lstm[dir].bias; }
        a.xf = xf; a.xf_step = (unsigned)(per_step_c * 4); a.xf_bytes = (unsigned)(2LL * T * per_step_c * 4);
        a.hx = hst; a.hx_bytes = (unsigned)(2 * 2 * (size_t)hsz * 4);
        a.bar = (unsigned*)(ws + o_ebar); a.err = (int*)(ws + o_ebar + (size_t)2 * pz::NSHARD * 32 * 4);
        a.lens = d_lens; a.out = memory_out; a.B = B; a.T = T;
        a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias;
        const dim3 grid(2 * pz::ENC_G), blk(pz::NT);
        {
            XwgGate gate;
            T2_TRY(gate.begin(m->ctx));
            if (MB == 1) hipLaunchKernelGGL(pz::encoder_kernel<1>, grid, blk, 0, st, a);
            else if (MB == 2) hipLaunchKernelGGL(pz::encoder_kernel<2>, grid, blk, 0, st, a);
            else hipLaunchKernelGGL(pz::encoder_kernel<4>, grid, blk, 0, st, a);
            HIP_TRY(hipGetLastError());
        }
        T2_TRY(inject_timeout(m, a.err));
        ++m->n_enc_persist;
        // the error word decides between "done" and "run it again launch per step": one host synchronisation per encode
        int h_err = 0;
        HIP_TRY(hipMemcpyAsync(&h_err, a.err, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!h_err) return T2_OK;
        // a hand-off wait ran out of time (the 128 workgroups were not all resident): same recurrence, one launch per step
        note_persist_timeout(m);
        HIP_TRY(hipMemsetAsync(ws + o_zero, 0, zero_bytes, st));
        HIP_TRY(hipMemsetAsync(memory_out, 0, (size_t)Mtot * E * 4, st));
        HIP_TRY(hipMemcpyAsync(ctl, h_ctl, sizeof(h_ctl), hipMemcpyHostToDevice, st));
    }
    for (int s = 0; s < T; ++s) {
        SkinnyCore c{};
        LstmEpi e{};
        for (int dir = 0; dir < 2; ++dir) {
            c.w[dir] = m->enc_lstm[dir].w;
            c.bias[dir] = m->enc_lstm[dir].bias;
            c.seg[dir][0] = FragBuf{xf + (size_t)dir * T * per_step_c, (int)per_step_c, T, 0, 0};
            c.seg[dir][1] = FragBuf{hst + (size_t)dir * 2 * hsz, hsz, 2, 0, 0};
            e.c[dir] = cst + (size_t)dir * MB * 16 * He;
            e.hprev[dir] = FragBuf{hst + (size_t)dir * 2 * hsz, hsz, 2, 0, 0};
            e.hstate[dir] = FragBuf{hst + (size_t)dir * 2 * hsz, hsz, 2, 1, 0};
            e.hraw[dir] = FragBuf{nullptr, 0, 1, 0, 0};
            e.plain_off[dir] = dir * He;
            e.reverse[dir] = dir;
        }
        c.kb0 = KPc / 16; c.kbtot = m->enc_lstm[0].kbtot; c.kb_w0 = 0; c.kb_full = c.kbtot; c.ctl = ctl; c.step_off = s; c.t_hint = s; c.B = B;
        e.H = He; e.Hld = He; e.zc = d.zoneout_cell; e.zh = d.zoneout_output; e.fb = d.forget_bias;
        e.lens = d_lens; e.plain_out = memory_out; e.plain_bs = T * E; e.plain_ts = E;
        launch_lstm<true>(MB, dim3(m->enc_lstm[0].n_wg, 2), st, c, e);
    }
    HIP_TRY(hipGetLastError());
    // h_lens / h_ids are host temporaries: make sure the async copies consumed them
    HIP_TRY(hipStreamSynchronize(st));
    return T2_OK;
}

extern "C" int t2_encode(t2_model* m, const int64_t* ids, const int64_t* lens, int B, int T_in, float* memory_out) {
    if (!ids) return fail(T2_EINVAL, "ids is null");
    return encode_impl(m, ids, nullptr, lens, B, T_in, memory_out);
}

extern "C" int t2_encode_embedded(t2_model* m, const float* emb, const int64_t* lens, int B, int T_in, float* memory_out) {
    if (!emb) return fail(T2_EINVAL, "emb is null");
    return encode_impl(m, nullptr, emb, lens, B, T_in, memory_out);
}

// ------------------------------------------------------------------------------------------------
// postnet
// ------------------------------------------------------------------------------------------------
extern "C" int t2_postnet(t2_model* m, const float* mel, int B, int T_out, float* mel_post) {
    if (!m || !mel || !mel_post) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    if (B < 1 || T_out < 1) return fail(T2_EINVAL, "bad shape");
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    const int Mtot = B * T_out, C = d.postnet_channels, M = d.num_mels;
    const size_t one = align_up((size_t)Mtot * C * 4, 256);
    T2_TRY(ensure_ws(&m->post_ws, &m->post_ws_bytes, 2 * one));
    float* bufs[2] = {(float*)m->post_ws, (float*)((char*)m->post_ws + one)};
    const float* cur = mel;
    int cur_c = M;
    for (int i = 0; i < d.postnet_layers; ++i) {
        float* dst = bufs[i & 1];
        const int act = (i + 1 < d.postnet_layers) ? 2 : 0;   // tanh, last layer linear (tacotron_v2.py:187-196)
        T2_TRY(launch_conv(m, m->post_convs[i], cur, cur_c, dst, C, Mtot, T_out, act, nullptr, 0));
        cur = dst;
        cur_c = C;
    }
    // Dense(out_units) + residual on the raw input (tacotron_v2.py:202-203)
    T2_TRY(launch_conv(m, m->post_proj, cur, C, mel_post, M, Mtot, Mtot, 0, mel, M));
    return T2_OK;
}

// ------------------------------------------------------------------------------------------------
// decoder state
// ------------------------------------------------------------------------------------------------
static void state_free(t2_state* s) {
    if (!s) return;
    for (int i = 0; i < 2; ++i) if (s->gexec[i]) hipGraphExecDestroy(s->gexec[i]);
    if (s->slab) hipFree(s->slab);
    if (s->fixed) hipFree(s->fixed);
    if (s->outbuf) hipFree(s->outbuf);
    if (s->tf_buf) hipFree(s->tf_buf);
    if (s->proj_tmp) hipFree(s->proj_tmp);
    delete s;
}

// The persistent decoder serves one dimension class (persist.h): the Tacotron2-paper sizes of SURVEY.md section 8d.
static bool persist_dims_ok(const t2_model* m, int B, int T) {
    const t2_dims& d = m->d;
    if (d.num_prenet_layers != 2 || d.prenet_units[0] != pz::KBX * 16 || d.prenet_units[1] != pz::KBX * 16) return false;
    if (d.enc_units != pz::EW || d.attention_lstm_units != pz::NWG * 4 || d.decoder_lstm_units != pz::NWG * 4) return false;
    if (d.attention_dim != pz::UA || d.attention_filters > 32 || d.attention_kernel > 32) return false;
    if (d.num_mels * d.outputs_per_step != 160) return false;
    if (m->n_cu < pz::NWG || B < 1 || B > 64 || T < 1) return false;
    int c = 16;
    while (c > 4 && B * c > pz::NWG) c >>= 1;
    return (T + c - 1) / c <= 64;          // the attention keeps four 16-position tiles per workgroup in registers
}
static inline int persist_chunks(int B) { int c = 16; while (c > 4 && B * c > pz::NWG) c >>= 1; return c; }
static inline size_t persist_lds_bytes(int B, int T) {
    const int C = persist_chunks(B), CH = (T + C - 1) / C, mb = (B + 15) / 16;
    return (size_t)pz::lds_floats(mb == 3 ? 4 : mb, T, CH) * 4;
}

static int state_alloc(t2_model* m, int B, int T, int cap, bool with_targets, t2_state** out) {
    const t2_dims& d = m->d;
    t2_state* s = new t2_state();
    s->m = m; s->B = B; s->T = T; s->MB = mb_for(B); s->cap = cap;
    const int MB = s->MB, R = MB * 16;
    const int E = d.enc_units, U = d.attention_dim, A = d.attention_lstm_units, D = d.decoder_lstm_units;
    const int M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    // fixed part
    const size_t o_val = carve((size_t)B * T * E * 4), o_keys = carve((size_t)B * T * U * 4), o_lens = carve((size_t)B * 4);
    hipError_t e1 = hipMalloc(&s->fixed, off);
    if (e1 != hipSuccess) { state_free(s); return fail(T2_ENOMEM, "hipMalloc(state fixed) failed: %s", hipGetErrorString(e1)); }
    s->values = (float*)((char*)s->fixed + o_val);
    s->keys = (float*)((char*)s->fixed + o_keys);
    s->lens = (int*)((char*)s->fixed + o_lens);
    // zeroable part
    off = 0;
    s->KBe = pad16(E) / 16; s->KBa = pad16(A) / 16; s->KBd = pad16(D) / 16;
    const int blk = MB * 256;                      // floats per k-block
    s->XA_sz = (s->KBe + s->KBa) * blk; s->X1_sz = (s->KBa + s->KBe + s->KBd) * blk; s->X2_sz = 2 * s->KBd * blk;
    const size_t o_ctl = carve(8 * 4);
    const size_t o_as = carve((size_t)B * T * 4), o_al = carve((size_t)B * T * 4);
    const size_t o_cp = carve((size_t)B * E * 4), o_q = carve((size_t)m->att_lstm.n_wg * R * U * 4);
    const size_t o_ca = carve((size_t)R * A * 4), o_c1 = carve((size_t)R * D * 4), o_c2 = carve((size_t)R * D * 4);
    const size_t o_XA = carve(2 * (size_t)s->XA_sz * 4), o_X1 = carve(2 * (size_t)s->X1_sz * 4), o_X2 = carve(2 * (size_t)s->X2_sz * 4);
    const size_t o_XY = carve((size_t)s->KBd * blk * 4);
    s->sk_tiles = std::max(ceil_div(m->att_lstm.n_wg, 2), ceil_div(m->lstm1.n_wg, 2));
    const size_t o_tk = carve((size_t)3 * s->sk_tiles * 4);
    const size_t o_sk = carve((size_t)s->sk_tiles * 2 * T2_SKINNY_THREADS * 2 * 8);
    const size_t o_xf = carve((size_t)R * pad16(M * nf) * 4);
    const size_t o_ZA = carve((size_t)m->att_lstm.n_wg * MB * 256 * 4), o_Z1 = carve((size_t)m->lstm1.n_wg * MB * 256 * 4),
                 o_Z2 = carve((size_t)m->lstm2.n_wg * MB * 256 * 4), o_Z1x = carve((size_t)m->lstm1.n_wg * MB * 256 * 4);
    size_t o_xp[T2_MAX_PRENET_LAYERS];
    for (int i = 0; i < d.num_prenet_layers; ++i) o_xp[i] = carve((size_t)R * pad16(d.prenet_units[i]) * 4);
    const bool pz_ok = persist_dims_ok(m, B, T) && persist_lds_bytes(B, T) <= 150 * 1024;
    size_t o_pz = 0, o_pzbar = 0, pz_total = 0;
    if (pz_ok) {
        auto pcarve = [&](size_t bytes) { size_t o = pz_total; pz_total = align_up(pz_total + bytes, 256); return (unsigned)o; };
        const size_t h = (size_t)R * 1024 * 4, x = (size_t)R * 256 * 4;
        s->pz_off[0] = pcarve(x); s->pz_off[1] = pcarve(x);                       // X1, X2
        s->pz_off[2] = pcarve(h); s->pz_off[3] = pcarve(h);                       // h_att raw, h_att state
        s->pz_off[4] = pcarve((size_t)R * 512 * 4);                               // context
        s->pz_off[5] = pcarve(h); s->pz_off[6] = pcarve(h);                       // y1 raw, h1 state
        s->pz_off[7] = pcarve(h); s->pz_off[8] = pcarve(h);                       // y2 raw, h2 state
        s->pz_off[9] = pcarve((size_t)8 * R * pz::UA * 8);                        // processed query: 8 K-split partials as tagged granules
        s->pz_off[10] = pcarve((size_t)B * T * 8);                                // energies as tagged 8-byte granules
        s->pz_off[11] = pcarve(2 * (size_t)B * T * 4);                            // attention state, two step parities
        s->pz_off[12] = pcarve(256);                                              // stop logits
        for (int i = 13; i < 17; ++i) s->pz_off[i] = pcarve(h);                   // second bank of h_att raw, h_att, h1, h2 (two-chain teacher schedule)
        o_pz = carve(pz_total);
        o_pzbar = carve((size_t)(pz::NSHARD + 64 + 4) * 32 * 4);
    }
    s->slab_bytes = off;
    e1 = hipMalloc(&s->slab, off);
    if (e1 != hipSuccess) { state_free(s); return fail(T2_ENOMEM, "hipMalloc(state slab) failed: %s", hipGetErrorString(e1)); }
    char* p = (char*)s->slab;
    s->ctl = (int*)(p + o_ctl);
    s->att_state = (float*)(p + o_as); s->alignments = (float*)(p + o_al);
    s->ctx_plain = (float*)(p + o_cp); s->qpart = (float*)(p + o_q);
    s->c_att = (float*)(p + o_ca); s->c1 = (float*)(p + o_c1); s->c2 = (float*)(p + o_c2);
    s->XA = (float*)(p + o_XA); s->X1 = (float*)(p + o_X1); s->X2 = (float*)(p + o_X2); s->XY = (float*)(p + o_XY);
    s->xf = (float*)(p + o_xf);
    s->ZA = (float*)(p + o_ZA); s->Z1 = (float*)(p + o_Z1); s->Z2 = (float*)(p + o_Z2); s->Z1x = (float*)(p + o_Z1x);
    s->tickets = (unsigned*)(p + o_tk); s->sk_slabs = (unsigned long long*)(p + o_sk);
    for (int i = 0; i < d.num_prenet_layers; ++i) s->xp[i] = (float*)(p + o_xp[i]);
    if (pz_ok) { s->pz_xs = (float*)(p + o_pz); s->pz_bytes = (unsigned)pz_total; s->pz_bar = (unsigned*)(p + o_pzbar); }
    // output staging
    off = 0;
    const size_t o_mel = carve((size_t)B * cap * r * M * 4), o_stop = carve((size_t)B * cap * 4);
    const size_t o_hist = carve((size_t)cap * B * T * 4);
    const size_t o_tgt = with_targets ? carve((size_t)B * cap * r * M * 4) : 0;
    e1 = hipMalloc(&s->outbuf, off);
    if (e1 != hipSuccess) { state_free(s); return fail(T2_ENOMEM, "hipMalloc(state out) failed: %s", hipGetErrorString(e1)); }
    p = (char*)s->outbuf;
    s->o_mel = (float*)(p + o_mel); s->o_stop = (float*)(p + o_stop); s->o_hist = (float*)(p + o_hist);
    s->i_targets = with_targets ? (float*)(p + o_tgt) : nullptr;
    *out = s;
    return T2_OK;
}

static size_t attn_lds_bytes(const t2_dims& d, int T) {
    return (size_t)attn_lds_floats(T, d.attention_dim, d.attention_filters, d.attention_kernel) * 4;
}

// attention mechanism constructor side effects + zero state
static int state_init(t2_state* s, const float* memory, const int64_t* lens) {
    t2_model* m = s->m;
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = s->B, T = s->T, E = d.enc_units;
    LensArg la{};                        // B <= 64 (check_batch): the lengths travel as a kernel argument, no host round trip
    if (B > 64) return fail(T2_EINVAL, "internal: batch of %d rows", B);
    for (int b = 0; b < B; ++b) {
        const int64_t l = lens ? lens[b] : T;
        if (l < 1 || l > T) return fail(T2_EINVAL, "memory_sequence_length %lld outside 1..%d", (long long)l, T);
        la.v[b] = (int)l;
    }
    hipLaunchKernelGGL(set_lens_kernel, dim3(1), dim3(64), 0, st, la, s->lens, B);
    const long long n = (long long)B * T * E;
    hipLaunchKernelGGL(mask_memory_kernel, dim3((unsigned)std::min<long long>(2048, (n + 255) / 256)), dim3(256), 0, st,
                       memory, s->lens, s->values, B, T, E);
    T2_TRY(launch_conv(m, m->memory_layer, s->values, E, s->keys, d.attention_dim, B * T, B * T, 0, nullptr, 0));
    return T2_OK;
}

static int state_zero(t2_state* s, int max_steps, int min_iters, int stop_enabled) {
    hipStream_t st = s->m->ctx->stream;
    HIP_TRY(hipMemsetAsync(s->slab, 0, s->slab_bytes, st));
    hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(64), 0, st, s->ctl, max_steps, min_iters, stop_enabled);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_state_create(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in, t2_state** out) {
    if (!m || !memory || !out) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    T2_TRY(check_batch(B, T_in));
    HIP_TRY(hipSetDevice(m->ctx->device));
    if (attn_lds_bytes(m->d, T_in) > 120 * 1024) return fail(T2_EINVAL, "T_in %d too long for the attention kernel's LDS", T_in);
    t2_state* s = nullptr;
    T2_TRY(state_alloc(m, B, T_in, 1, false, &s));
    int r = state_init(s, memory, lens);
    if (r == T2_OK) r = state_zero(s, 0x7fffffff, 0, 0);
    if (r != T2_OK) { state_free(s); return r; }
    *out = s;
    return T2_OK;
}

extern "C" int t2_state_reset(t2_state* s) {
    if (!s) return fail(T2_EINVAL, "null state");
    return state_zero(s, 0x7fffffff, 0, 0);
}

extern "C" int t2_state_destroy(t2_state* s) {
    if (!s) return T2_OK;
    if (s->owned_by_model) return fail(T2_EINVAL, "state is owned by the model");
    hipStreamSynchronize(s->m->ctx->stream);
    state_free(s);
    return T2_OK;
}

static int current_parity(t2_state* s, int* par) {
    hipStream_t st = s->m->ctx->stream;
    int h[8];
    HIP_TRY(hipMemcpyAsync(h, s->ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    int t = h[CTL_BASE];
    if (h[CTL_FINAL] != 0) t = std::min(t, h[CTL_FINAL]);
    t = std::min(t, h[CTL_MAX]);
    *par = t & 1;
    return T2_OK;
}

extern "C" int t2_state_get(t2_state* s, const char* name, float* out) {
    if (!s || !name || !out) return fail(T2_EINVAL, "null argument");
    const t2_dims& d = s->m->d;
    hipStream_t st = s->m->ctx->stream;
    const int B = s->B, T = s->T;
    auto copy_plain = [&](const float* src, size_t rows, size_t ld_src, size_t cols) -> int {
        HIP_TRY(hipMemcpy2DAsync(out, cols * 4, src, ld_src * 4, cols * 4, rows, hipMemcpyDeviceToDevice, st));
        return T2_OK;
    };
    auto copy_frag = [&](const float* src, int K, int koff) -> int {
        hipLaunchKernelGGL(frag_to_plain_kernel, dim3(ceil_div(B * K, 256)), dim3(256), 0, st, src, out, B, K, s->MB, koff);
        HIP_TRY(hipGetLastError());
        return T2_OK;
    };
    int par = 0;
    if (!strcmp(name, "attention_c")) return copy_plain(s->c_att, B, d.attention_lstm_units, d.attention_lstm_units);
    if (!strcmp(name, "lstm1_c")) return copy_plain(s->c1, B, d.decoder_lstm_units, d.decoder_lstm_units);
    if (!strcmp(name, "lstm2_c")) return copy_plain(s->c2, B, d.decoder_lstm_units, d.decoder_lstm_units);
    if (!strcmp(name, "attention_h")) { T2_TRY(current_parity(s, &par)); return copy_frag(s->XA + (size_t)par * s->XA_sz, d.attention_lstm_units, s->KBe * 16); }
    if (!strcmp(name, "lstm1_h")) { T2_TRY(current_parity(s, &par)); return copy_frag(s->X1 + (size_t)par * s->X1_sz, d.decoder_lstm_units, (s->KBa + s->KBe) * 16); }
    if (!strcmp(name, "lstm2_h")) { T2_TRY(current_parity(s, &par)); return copy_frag(s->X2 + (size_t)par * s->X2_sz, d.decoder_lstm_units, s->KBd * 16); }
    if (!strcmp(name, "attention")) return copy_plain(s->ctx_plain, B, d.enc_units, d.enc_units);
    if (!strcmp(name, "alignments")) return copy_plain(s->alignments, B, T, T);
    if (!strcmp(name, "attention_state")) return copy_plain(s->att_state, B, T, T);
    if (!strcmp(name, "keys")) return copy_plain(s->keys, (size_t)B * T, d.attention_dim, d.attention_dim);
    // raw cell outputs of the LAST step (what the inner RNNCell wrappers return: rnn_wrappers.py:51-53, 76-78; tacotron_v2.py:177-178)
    if (!strcmp(name, "attention_output")) { T2_TRY(current_parity(s, &par)); return copy_frag(s->X1 + (size_t)(par ^ 1) * s->X1_sz, d.attention_lstm_units, 0); }
    if (!strcmp(name, "lstm1_output")) { T2_TRY(current_parity(s, &par)); return copy_frag(s->X2 + (size_t)(par ^ 1) * s->X2_sz, d.decoder_lstm_units, 0); }
    if (!strcmp(name, "decoder_output")) return copy_frag(s->XY, d.decoder_lstm_units, 0);
    return fail(T2_EINVAL, "unknown state tensor '%s'", name);
}

// ------------------------------------------------------------------------------------------------
// one decoder step = the launch sequence below
// ------------------------------------------------------------------------------------------------
struct StepIO {
    int feed_mode;              // 0: teacher (targets), 1: own output (proj writes xin), 2: caller frames (API),
                                // 3: teacher, PreNets of every step already computed (xp_all: frag16 operand per step)
    float* xp_all; int xp_all_stride; int xp_all_slots;
    const float* feed_src;      // mode 0: targets + M*(r-nf) ; mode 2: prev_frames
    long long feed_row_stride;  // floats
    long long feed_t_stride;
    float* mel; int mel_ld; long long mel_t_stride;
    float* stop; int stop_ld; int stop_t_mul;
    float* hist;
    int slot_hint;              // 1: ctl base is even (chunked loops with an even chunk), so t and step_off select the same slots
    const struct TrainTape* train;   // non-null: training-mode arithmetic (sampled dropout / zoneout) + tape writes
    int defer_proj;             // 1: teacher-forced decode, no projection launch per step (teacher_projection after the loop)
    float* y_all;               // where LSTM 2 keeps its raw output of every step [S][B][D] (null: the training tape has it)
};

// Tape of one teacher-forced training forward (csrc/train_step.h): plain [S][B][.] arrays read by the backward pass.
struct TrainTape {
    uint32_t seed;
    float prenet_drop_rate;
    float* xp[T2_MAX_PRENET_LAYERS];          // PreNet outputs after dropout
    float *z[3], *cprev[3], *cnew[3], *hraw[3], *hstate[3];   // attention LSTM, LSTM 1, LSTM 2
    float *q, *state_prev, *ctx;              // attention: q + b_a, state before the step, context
    int inference_zoneout;                    // 1: only the PreNet dropout is sampled (apply_dropout_on_inference, modules.py:96-98); the cells blend
};

static void persist_fill_tape(pz::PTape& tp, const TrainTape* tape) {
    for (int l = 0; l < 3; ++l) { tp.z[l] = tape->z[l]; tp.cprev[l] = tape->cprev[l]; tp.cnew[l] = tape->cnew[l]; tp.hraw[l] = tape->hraw[l]; tp.hstate[l] = tape->hstate[l]; }
    tp.q = tape->q; tp.state_prev = tape->state_prev; tp.ctx = tape->ctx; tp.seed = tape->seed; tp.on = 1;
}

// which launch schedule a decoder step uses (see launch_step)
static inline bool spread_mode(const t2_model* m, int MB, int feed_mode, bool train) {
    return m->opt_hoist && m->opt_spread && (MB >= 2 || m->opt_spread >= 2) && !m->opt_lstm_only && feed_mode != 3 &&
           m->d.num_prenet_layers >= 2 && !train;
}
static inline bool fused_mode(const t2_model* m, int MB, int feed_mode, bool train) {
    return m->opt_hoist && !m->opt_lstm_only && !train && feed_mode == 1 && m->opt_projfuse && m->proj_fused_ok;
}

static int launch_step(t2_state* s, int step_off, const StepIO& io) {
    t2_model* m = s->m;
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = s->B, MB = s->MB;
    const int E = d.enc_units, U = d.attention_dim, A = d.attention_lstm_units, D = d.decoder_lstm_units;
    const int M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame;
    auto core = [&](const Packed& p) {
        SkinnyCore c{};
        c.w[0] = c.w[1] = p.w; c.bias[0] = c.bias[1] = p.bias;
        c.kb0 = 0; c.kbtot = p.kbtot; c.kb_w0 = 0; c.kb_full = p.kbtot; c.ctl = s->ctl; c.step_off = step_off; c.B = B;
        c.t_hint = io.slot_hint ? step_off : -1;
        return c;
    };
    auto single = [&](float* base, int koff) { return FragBuf{base, 0, 1, 0, koff}; };
    auto pp = [&](float* base, int sz, int dt, int koff) { return FragBuf{base, sz, 2, dt, koff}; };
    const FragBuf none{nullptr, 0, 1, 0, 0};
    const int KBe = s->KBe, KBa = s->KBa;
    const int np = d.num_prenet_layers;
    const bool all = !m->opt_lstm_only;
    const FragBuf xp_last = io.feed_mode == 3 ? FragBuf{io.xp_all, io.xp_all_stride, io.xp_all_slots, 0, 0} : single(s->xp[np - 1], 0);
    // 1. decoder input frames (helpers.py:86 / :182 / go frame :187-188)
    if (all && (io.feed_mode == 0 || io.feed_mode == 2)) {
        FeedArgs f{};
        f.src = io.feed_src; f.row_stride = io.feed_row_stride; f.t_stride = io.feed_t_stride; f.dst = single(s->xf, 0);
        f.B = B; f.K = M * nf; f.MB = MB; f.go_zero = io.feed_mode == 0; f.ctl = s->ctl; f.step_off = step_off;
        hipLaunchKernelGGL(feed_frames_kernel, dim3(ceil_div(MB * 16 * pad16(M * nf), 256)), dim3(256), 0, st, f);
    }
    const bool hoist = m->opt_hoist != 0;
    // recurrent half of an LSTM layer: [state operands] . kernel[their K range] -> z (partial_body)
    auto job = [&](const Packed& p, int kb_w0, int kb_n, FragBuf src, int off) {
        SkinnyCore c = core(p);
        c.step_off = step_off + off;
        c.t_hint = io.slot_hint ? step_off + off : -1;
        c.kb_w0 = kb_w0; c.kbtot = kb_n; c.kb0 = 0; c.seg[0][1] = src;
        return c;
    };
    // Spread schedule (more than one row block, PreNet launches present).  Four hoisted products of K = one state width
    // each -- Z1 = h1.W, Z2 = h2.W (state of the previous step), Z1x = h_att_raw.W (this step's attention cell output),
    // ZA = h_att.W for the NEXT step -- are cut into column-tile ranges and placed on the CUs that the small launches of
    // the step leave idle.  The attention launch lasts ~25 us whenever anything rides in it and holds three rider rounds
    // on its n_cu - B free CUs in that time, so it takes Z1x, Z2 and the tail of Z1; the projection takes ZA of the next
    // step (as many tiles as it has free CUs); what is left (head of Z1, the last tiles of ZA) rides in the LAST PreNet
    // launch and, only if that is full, in the one before.  The riding workgroups are separate from the dense role's own,
    // so a small launch lasts max(own, one rider round) instead of own + rider.  The attention LSTM then multiplies
    // [prenet | ctx], LSTM 1 only the new context.
    const bool spread = spread_mode(m, MB, io.feed_mode, io.train != nullptr);
    const int nA = m->att_lstm.n_wg, n1 = m->lstm1.n_wg, n2 = m->lstm2.n_wg;
    // fused: the projection launch also computes the first PreNet layer of the next step (composed weights), so the
    // PreNet-0 launch is skipped (step 0 gets it from decode_prologue)
    const bool fused = fused_mode(m, MB, io.feed_mode, io.train != nullptr);
    const Packed& projP = fused ? m->proj_fused : m->proj;
    struct Piece { int job, off, n; };       // job: 0 = Z1, 1 = Z2, 2 = ZA (this step's)
    int za_proj = 0, z2_att = 0, z1_att = 0;
    Piece pre_pieces[2][3] = {};             // [0]: last PreNet launch, [1]: the one before
    int pre_np[2] = {0, 0};
    if (spread) {
        const int att_cap = (m->opt_sp_att >= 0 ? m->opt_sp_att : 3) * std::max(1, m->n_cu - B);
        za_proj = std::min(nA, std::max(0, m->n_cu - projP.n_wg));
        z2_att = std::min(n2, std::max(0, att_cap - n1));
        z1_att = std::min(n1, std::max(0, att_cap - n1 - z2_att));
        const Piece todo[3] = {{2, za_proj, nA - za_proj}, {1, z2_att, n2 - z2_att}, {0, 0, n1 - z1_att}};
        int slot = 0, room = fused ? (1 << 30) : std::max(0, m->n_cu - m->prenet[np - 1].n_wg);     // fused: no launch before the last PreNet
        for (const Piece& q : todo) {
            int off = q.off, n = q.n;
            while (n > 0) {
                if (slot == 0 && (room == 0 || pre_np[0] == 3)) { slot = 1; room = 1 << 30; }     // the rest: a second round there
                const int k = std::min(n, room);
                if (pre_np[slot] == 3) return fail(T2_ESTATE, "internal: rider plan needs more than three tile ranges in one launch");
                pre_pieces[slot][pre_np[slot]++] = Piece{q.job, off, k};
                off += k; n -= k; room -= k;
            }
        }
    }
    auto job_Z1 = [&]() { return job(m->lstm1, KBa + KBe, s->KBd, pp(s->X1, s->X1_sz, 0, (KBa + KBe) * 16), 0); };
    auto job_Z2 = [&]() { return job(m->lstm2, s->KBd, s->KBd, pp(s->X2, s->X2_sz, 0, s->KBd * 16), 0); };
    auto job_ZAh = [&](int off) { return job(m->att_lstm, m->att_lstm.seg_kb[0] + KBe, KBa, pp(s->XA, s->XA_sz, 0, KBe * 16), off); };
    // 2. prenets (rnn_wrappers.py:51-53)
    for (int i = fused ? 1 : 0; all && io.feed_mode != 3 && i < np; ++i) {
        SkinnyCore c = core(m->prenet[i]);
        c.seg[0][1] = single(i == 0 ? s->xf : s->xp[i - 1], 0);
        DenseEpi e{};
        e.N = d.prenet_units[i]; e.relu = 1; e.out_frag = single(s->xp[i], 0); e.frag_col0 = 0; e.stop_col = -1;
        if (io.train) {
            e.drop_rate = io.train->prenet_drop_rate; e.seed = io.train->seed; e.stream = T2_STREAM_PRENET + i;
            e.out_plain = io.train->xp[i]; e.plain_ld = d.prenet_units[i]; e.plain_t_stride = (long long)B * d.prenet_units[i];
            e.plain_ncol = d.prenet_units[i];
        }
        PartialJobs pj{};
        const int slot = np - 1 - i;        // 0: last PreNet launch
        if (spread && slot < 2) {
            pj.skip = m->prenet[i].n_wg;
            for (int k = 0; k < pre_np[slot]; ++k) {
                const Piece& q = pre_pieces[slot][k];
                pj.core[k] = q.job == 0 ? job_Z1() : q.job == 1 ? job_Z2() : job_ZAh(0);
                pj.z[k] = q.job == 0 ? s->Z1 : q.job == 1 ? s->Z2 : s->ZA;
                pj.off[k] = q.off; pj.n[k] = q.n;
            }
        }
        launch_dense(MB, dim3(m->prenet[i].n_wg), st, c, e, (spread && slot < 2 && pre_np[slot] > 0) ? &pj : nullptr);
    }
    // One row block (B <= 16): the attention launch leaves > 240 CUs idle for ~19 us, so it takes all three state halves
    // (h1, h2 of this step and h_att for the NEXT step); the context part of the attention LSTM's K range stays in the
    // LSTM launch and the projection launch carries no rider.  At B = 64 the same placement costs a fourth rider round.
    const bool att3 = hoist && m->opt_att3 && (MB == 1 || io.defer_proj) && !m->opt_lstm_only && !spread;   // deferred projection: its rider moves here
    const bool timing = m->opt_time_lstm && !m->opt_graph;
    auto taped = [&](LstmEpi& e, int layer, int stream) {
        if (!io.train || io.train->inference_zoneout) return;
        e.train = 1; e.seed = io.train->seed; e.stream = stream;
        e.tp_z = io.train->z[layer]; e.tp_cprev = io.train->cprev[layer]; e.tp_cnew = io.train->cnew[layer];
        e.tp_hraw = io.train->hraw[layer]; e.tp_hstate = io.train->hstate[layer];
    };
    // 3. attention LSTM on concat(prenet out, previous context)  (AttentionWrapper, App. A6) + this step's
    //    partial query products (tacotron_v2.py:122)
    {
        SkinnyCore c = core(m->att_lstm);
        LstmEpi e{};
        if (att3 || spread) {
            c.kb0 = m->att_lstm.seg_kb[0];
            c.kbtot = m->att_lstm.seg_kb[0] + KBe;
            c.seg[0][0] = xp_last;
            c.seg[0][1] = pp(s->XA, s->XA_sz, 0, 0);
            e.zpart = s->ZA;
        } else if (hoist) {
            c.kbtot = m->att_lstm.seg_kb[0];
            c.seg[0][1] = xp_last;
            e.zpart = s->ZA;
        } else {
            c.kb0 = m->att_lstm.seg_kb[0];
            c.seg[0][0] = xp_last;
            c.seg[0][1] = pp(s->XA, s->XA_sz, 0, 0);
        }
        e.H = A; e.c[0] = s->c_att; e.Hld = A;
        e.hprev[0] = pp(s->XA, s->XA_sz, 0, KBe * 16); e.hstate[0] = pp(s->XA, s->XA_sz, 1, KBe * 16);
        e.hraw[0] = pp(s->X1, s->X1_sz, 0, 0);
        e.zc = d.zoneout_cell; e.zh = d.zoneout_output; e.fb = d.forget_bias;
        e.wq = m->wq; e.qpart = s->qpart; e.U = U;
        taped(e, 0, T2_STREAM_ZONE_ATT);
        timed_lstm(timing, MB, m->att_lstm.n_wg, st, c, e, (m->opt_splitk && !hoist) ? s->tickets : nullptr, s->sk_slabs);
    }
    // 4./5. location-sensitive attention + context
    if (all) {
        AttnArgs a{};
        a.keys = s->keys; a.values = s->values; a.lens = s->lens; a.att_state = s->att_state; a.alignments = s->alignments;
        a.history = io.hist; a.qpart = s->qpart;
        a.nparts = (m->opt_splitk && !hoist && MB >= 2) ? ceil_div(m->att_lstm.n_wg, 2) : m->att_lstm.n_wg;
        a.ctx_dst[0] = pp(s->XA, s->XA_sz, 1, 0); a.ctx_dst[1] = pp(s->X1, s->X1_sz, 0, KBa * 16); a.ctx_plain = s->ctx_plain;
        a.conv_w = m->conv_w; a.conv_b = m->conv_b; a.wloc = m->wloc; a.v_a = m->v_a; a.b_a = m->b_a;
        a.B = B; a.T = s->T; a.U = U; a.E = E; a.F = d.attention_filters; a.K = d.attention_kernel; a.MB = MB;
        a.cumulative = d.cumulative_weights; a.ctl = s->ctl; a.step_off = step_off;
        if (io.train) { a.tp_q = io.train->q; a.tp_state_prev = io.train->state_prev; a.tp_ctx = io.train->ctx; }
        const size_t lds = attn_lds_bytes(d, s->T);
        // the recurrent halves of the two decoder LSTMs of THIS step ride behind the attention grid
        PartialJobs pj{};
        if (spread) {
            pj.core[0] = job(m->lstm1, 0, KBa, pp(s->X1, s->X1_sz, 0, 0), 0); pj.z[0] = s->Z1x; pj.n[0] = n1;
            pj.core[1] = job_Z2(); pj.z[1] = s->Z2; pj.n[1] = z2_att;
            pj.core[2] = job_Z1(); pj.z[2] = s->Z1; pj.off[2] = n1 - z1_att; pj.n[2] = z1_att;
        } else if (hoist) {
            pj.core[0] = job(m->lstm1, KBa + KBe, s->KBd, pp(s->X1, s->X1_sz, 0, (KBa + KBe) * 16), 0); pj.z[0] = s->Z1; pj.n[0] = m->lstm1.n_wg;
            pj.core[1] = job(m->lstm2, s->KBd, s->KBd, pp(s->X2, s->X2_sz, 0, s->KBd * 16), 0); pj.z[1] = s->Z2; pj.n[1] = m->lstm2.n_wg;
            if (att3) {
                pj.core[2] = job(m->att_lstm, m->att_lstm.seg_kb[0] + KBe, KBa, pp(s->XA, s->XA_sz, 0, KBe * 16), 1);
                pj.z[2] = s->ZA; pj.n[2] = m->att_lstm.n_wg;
            }
        }
        launch_attention(MB, B, lds, st, a, hoist ? &pj : nullptr);
    } else if (hoist) {      // lstm_only timing runs: the riding jobs as launches of their own
        launch_partial(MB, dim3(m->lstm1.n_wg), st, job(m->lstm1, KBa + KBe, s->KBd, pp(s->X1, s->X1_sz, 0, (KBa + KBe) * 16), 0), s->Z1);
        launch_partial(MB, dim3(m->lstm2.n_wg), st, job(m->lstm2, s->KBd, s->KBd, pp(s->X2, s->X2_sz, 0, s->KBd * 16), 0), s->Z2);
    }
    // 6./7. decoder LSTMs on concat(attention cell output, new context)  (rnn_wrappers.py:76-78, tacotron_v2.py:155-161)
    {
        SkinnyCore c = core(m->lstm1);
        LstmEpi e{};
        c.seg[0][1] = pp(s->X1, s->X1_sz, 0, 0);
        if (spread) { c.kb_w0 = KBa; c.kbtot = KBe; c.seg[0][1] = pp(s->X1, s->X1_sz, 0, KBa * 16); e.zpart = s->Z1; e.zpart2 = s->Z1x; }
        else if (hoist) { c.kbtot = KBa + KBe; e.zpart = s->Z1; }
        e.H = D; e.c[0] = s->c1; e.Hld = D;
        e.hprev[0] = pp(s->X1, s->X1_sz, 0, (KBa + KBe) * 16); e.hstate[0] = pp(s->X1, s->X1_sz, 1, (KBa + KBe) * 16);
        e.hraw[0] = pp(s->X2, s->X2_sz, 0, 0);
        e.zc = d.zoneout_cell; e.zh = d.zoneout_output; e.fb = d.forget_bias;
        taped(e, 1, T2_STREAM_ZONE_L1);
        timed_lstm(timing, MB, m->lstm1.n_wg, st, c, e, (m->opt_splitk && !hoist) ? s->tickets + s->sk_tiles : nullptr, s->sk_slabs);
    }
    {
        SkinnyCore c = core(m->lstm2);
        LstmEpi e{};
        c.seg[0][1] = pp(s->X2, s->X2_sz, 0, 0);
        if (hoist) { c.kbtot = s->KBd; e.zpart = s->Z2; }
        e.H = D; e.c[0] = s->c2; e.Hld = D;
        e.hprev[0] = pp(s->X2, s->X2_sz, 0, s->KBd * 16); e.hstate[0] = pp(s->X2, s->X2_sz, 1, s->KBd * 16);
        e.hraw[0] = single(s->XY, 0);
        e.zc = d.zoneout_cell; e.zh = d.zoneout_output; e.fb = d.forget_bias;
        taped(e, 2, T2_STREAM_ZONE_L2);
        e.y_all = io.defer_proj ? io.y_all : nullptr;
        if (io.defer_proj && !io.y_all && !io.train) return fail(T2_ESTATE, "internal: deferred projection without a store for LSTM 2's outputs");
        timed_lstm(timing, MB, m->lstm2.n_wg, st, c, e, (m->opt_splitk && !hoist) ? s->tickets + 2 * s->sk_tiles : nullptr, s->sk_slabs);
    }
    // 8. mel / stop projections + stop rule  (rnn_wrappers.py:105-109, helpers.py:90-94)
    if (all && io.defer_proj) {
        // teacher forcing: nothing of this step's output is fed back (helpers.py:134-136,182), so the projections of all steps
        // run as one GEMM after the loop (teacher_projection)
    } else if (all) {
        SkinnyCore c = core(projP);
        c.seg[0][1] = single(s->XY, 0);
        DenseEpi e{};
        e.N = M * r + 1; e.relu = 0;
        if (fused) { e.fuse_col0 = pad16(M * r + 1); e.fuse_n = d.prenet_units[0]; e.fuse_frag = single(s->xp[0], 0); }
        e.out_plain = io.mel; e.plain_ld = io.mel_ld; e.plain_t_stride = io.mel_t_stride; e.plain_ncol = M * r;
        e.out_frag = io.feed_mode == 1 ? single(s->xf, 0) : none; e.frag_col0 = M * (r - nf);
        e.stop_col = M * r; e.out_stop = io.stop; e.stop_ld = io.stop_ld; e.stop_t_mul = io.stop_t_mul; e.ctl_w = s->ctl;
        // the recurrent half of the attention LSTM of the NEXT step rides behind the projection grid
        PartialJobs pj{};
        if (spread) { pj.skip = projP.n_wg; pj.core[0] = job_ZAh(1); pj.z[0] = s->ZA; pj.n[0] = za_proj; }
        else if (hoist && !att3) { pj.core[0] = job(m->att_lstm, m->att_lstm.seg_kb[0], KBe + KBa, pp(s->XA, s->XA_sz, 0, 0), 1); pj.z[0] = s->ZA; pj.n[0] = m->att_lstm.n_wg; }
        launch_dense(MB, dim3(projP.n_wg), st, c, e, (hoist && !att3) ? &pj : nullptr);
    } else if (hoist) {
        launch_partial(MB, dim3(m->att_lstm.n_wg), st, job(m->att_lstm, m->att_lstm.seg_kb[0], KBe + KBa, pp(s->XA, s->XA_sz, 0, 0), 1), s->ZA);
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int read_steps_quiet(t2_state* s, int* steps_out) {
    hipStream_t st = s->m->ctx->stream;
    int h[8];
    HIP_TRY(hipMemcpyAsync(h, s->ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *steps_out = h[CTL_FINAL] != 0 ? h[CTL_FINAL] : h[CTL_MAX];
    return T2_OK;
}

// final_state of dynamic_decode (tacotron_v1.py:149-153: DecoderV1.call returns it; models.py:53 reads its alignment history):
// the cell stack's state after the last decode loop of this model (t2_decode_teacher / t2_decode_free*).  Names and shapes as
// t2_state_get.  The persistent decoder keeps the LSTM states in registers during the loop and stores them when it ends.
extern "C" int t2_decode_final_state(t2_model* m, const char* name, float* out) {
    if (!m || !name || !out) return fail(T2_EINVAL, "null argument");
    t2_state* s = m->last_loop;
    if (!s) return fail(T2_ESTATE, "no decode loop has run (or its state was evicted)");
    if (m->pending) return fail(T2_ESTATE, "a free-running decode is in flight; call t2_decode_free_finish first");
    if (!m->last_persist) return t2_state_get(s, name, out);
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = s->B, T = s->T;
    int steps = 0;
    T2_TRY(read_steps_quiet(s, &steps));
    auto frag = [&](unsigned off, int K) -> int {
        hipLaunchKernelGGL(frag_to_plain_kernel, dim3(ceil_div(B * K, 256)), dim3(256), 0, st, s->pz_xs + off / 4, out, B, K, s->MB, 0);
        HIP_TRY(hipGetLastError());
        return T2_OK;
    };
    auto plain = [&](const float* src, size_t n) -> int { HIP_TRY(hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToDevice, st)); return T2_OK; };
    const int A = d.attention_lstm_units, D = d.decoder_lstm_units;
    if (!strcmp(name, "attention_c")) return plain(s->c_att, (size_t)B * A);
    if (!strcmp(name, "lstm1_c")) return plain(s->c1, (size_t)B * D);
    if (!strcmp(name, "lstm2_c")) return plain(s->c2, (size_t)B * D);
    if (!strcmp(name, "attention_h")) return frag(s->pz_off[3], A);
    if (!strcmp(name, "lstm1_h")) return frag(s->pz_off[6], D);
    if (!strcmp(name, "lstm2_h")) return frag(s->pz_off[8], D);
    if (!strcmp(name, "attention_output")) return frag(s->pz_off[2], A);
    if (!strcmp(name, "lstm1_output")) return frag(s->pz_off[5], D);
    if (!strcmp(name, "decoder_output")) return frag(s->pz_off[7], D);
    if (!strcmp(name, "attention")) return frag(s->pz_off[4], d.enc_units);
    if (!strcmp(name, "alignments")) { if (steps < 1) return fail(T2_ESTATE, "no step has run"); return plain(s->o_hist + (size_t)(steps - 1) * B * T, (size_t)B * T); }
    if (!strcmp(name, "attention_state")) {
        if (steps < 1) return fail(T2_ESTATE, "no step has run");
        return plain(s->pz_xs + s->pz_off[11] / 4 + (size_t)((steps - 1) & 1) * B * T, (size_t)B * T);
    }
    if (!strcmp(name, "keys")) return plain(s->keys, (size_t)B * T * d.attention_dim);
    return fail(T2_EINVAL, "unknown state tensor '%s'", name);
}

static int decode_step_impl(t2_model* m, t2_state* s, const float* prev_frames, float* mel_r, float* stop, const TrainTape* train);
extern "C" int t2_decode_step(t2_model* m, t2_state* s, const float* prev_frames, float* mel_r, float* stop) {
    return decode_step_impl(m, s, prev_frames, mel_r, stop, nullptr);
}

// One cell call with the training-mode sampling of the reference (rnn_wrappers.py:51-53 -> modules.py:90,96-98 PreNet dropout;
// modules.py:274-276 ZoneoutLSTMCell's sampled branch): masks u01(seed, stream, time, row, unit) of the shared counter-based
// generator, time = the state's step count -- the same masks t2_train_decoder_forward draws at that step.  Nothing is taped.
extern "C" int t2_decode_step_train(t2_model* m, t2_state* s, const float* prev_frames, float prenet_drop_rate, int sample_zoneout,
                                    uint32_t seed, float* mel_r, float* stop) {
    if (!(prenet_drop_rate >= 0.f && prenet_drop_rate < 1.f)) return fail(T2_EINVAL, "drop rate must be in [0, 1)");
    TrainTape tape{};
    tape.seed = seed; tape.prenet_drop_rate = prenet_drop_rate; tape.inference_zoneout = sample_zoneout ? 0 : 1;
    return decode_step_impl(m, s, prev_frames, mel_r, stop, &tape);       // the launches copy what they need: the tape may die here
}

static int decode_step_impl(t2_model* m, t2_state* s, const float* prev_frames, float* mel_r, float* stop, const TrainTape* train) {
    if (!m || !s || !prev_frames || !mel_r || !stop) return fail(T2_EINVAL, "null argument");
    if (s->m != m) return fail(T2_EINVAL, "state belongs to another model");
    HIP_TRY(hipSetDevice(m->ctx->device));
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    StepIO io{};
    io.train = train;
    io.feed_mode = 2; io.feed_src = prev_frames; io.feed_row_stride = d.num_mels * d.n_feed_frame; io.feed_t_stride = 0;
    io.mel = s->o_mel; io.mel_ld = d.num_mels * d.outputs_per_step; io.mel_t_stride = 0;
    io.stop = s->o_stop; io.stop_ld = 1; io.stop_t_mul = 0; io.hist = nullptr;
    T2_TRY(launch_step(s, 0, io));
    hipLaunchKernelGGL(advance_ctl_kernel, dim3(1), dim3(64), 0, st, s->ctl, 1);
    const size_t nmel = (size_t)s->B * d.num_mels * d.outputs_per_step;
    HIP_TRY(hipMemcpyAsync(mel_r, s->o_mel, nmel * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)s->B * 4, hipMemcpyDeviceToDevice, st));
    return T2_OK;
}

// ------------------------------------------------------------------------------------------------
// decode loops
// ------------------------------------------------------------------------------------------------
static int get_loop_state(t2_model* m, int B, int T, int cap, t2_state** out) {
    for (t2_state* s : m->states)
        if (s->B == B && s->T == T && s->cap == cap) { *out = s; return T2_OK; }
    if (attn_lds_bytes(m->d, T) > 120 * 1024) return fail(T2_EINVAL, "T_in %d too long for the attention kernel's LDS", T);
    // keep the cache small: drop everything when it grows
    if (m->states.size() >= 4) {
        if (m->pending) return fail(T2_ESTATE, "a free-running decode is in flight; call t2_decode_free_finish first");
        HIP_TRY(hipStreamSynchronize(m->ctx->stream));
        for (t2_state* s : m->states) state_free(s);
        m->states.clear();
        m->pending = nullptr;
        m->last_loop = nullptr;
    }
    t2_state* s = nullptr;
    T2_TRY(state_alloc(m, B, T, cap, true, &s));
    s->owned_by_model = true;
    m->states.push_back(s);
    *out = s;
    return T2_OK;
}

static StepIO loop_io(t2_state* s, int mode) {
    const t2_dims& d = s->m->d;
    const int M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame;
    StepIO io{};
    io.feed_mode = mode;
    if (mode == 0 && s->tf_frag != nullptr) { io.feed_mode = 3; io.xp_all = s->tf_frag; io.xp_all_stride = (int)s->tf_stride; io.xp_all_slots = s->cap; }
    if (io.feed_mode == 3 && s->m->opt_tfproj && s->m->opt_hoist && s->m->opt_att3 && s->proj_tmp != nullptr) { io.defer_proj = 1; io.y_all = s->y_all; }
    io.feed_src = s->i_targets ? s->i_targets + (size_t)M * (r - nf) : nullptr;
    io.feed_row_stride = (long long)s->cap * r * M;
    io.feed_t_stride = (long long)r * M;
    io.mel = s->o_mel; io.mel_ld = s->cap * r * M; io.mel_t_stride = (long long)r * M;
    io.stop = s->o_stop; io.stop_ld = s->cap; io.stop_t_mul = 1;
    io.hist = s->o_hist;
    io.slot_hint = (s->m->opt_chunk % 2 == 0) ? 1 : 0;
    return io;
}

// PreNets of all cap steps of a teacher-forced decode from s->i_targets (batched GEMMs); the per-step frag16 operands land
// in s->tf_frag, which loop_io hands to launch_step as feed mode 3.  tape_xp (training): where layer i's outputs are kept.
static int teacher_prenets(t2_state* s, float drop_rate, uint32_t seed, float* const* tape_xp = nullptr) {
    t2_model* m = s->m;
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    const int B = s->B, S = s->cap, MB = s->MB, M = d.num_mels, r = d.outputs_per_step, nf = d.n_feed_frame, np = d.num_prenet_layers;
    const long long SB = (long long)S * B;
    if (!s->tf_buf) {
        int pmax = 0;
        for (int i = 0; i < np; ++i) pmax = std::max(pmax, d.prenet_units[i]);
        s->tf_stride = (long long)MB * 16 * pad16(d.prenet_units[np - 1]);
        const size_t n_xin = align_up((size_t)SB * M * nf, 64), n_x = align_up((size_t)SB * pmax, 64), n_frag = (size_t)S * s->tf_stride;
        hipError_t e = hipMalloc((void**)&s->tf_buf, (n_xin + 2 * n_x + n_frag) * 4);
        if (e != hipSuccess) { s->tf_buf = nullptr; return fail(T2_ENOMEM, "hipMalloc(teacher-forced PreNet buffers) failed: %s", hipGetErrorString(e)); }
        HIP_TRY(hipMemsetAsync(s->tf_buf, 0, (n_xin + 2 * n_x + n_frag) * 4, st));
        s->tf_xin = s->tf_buf; s->tf_x[0] = s->tf_xin + n_xin; s->tf_x[1] = s->tf_x[0] + n_x; s->tf_frag = s->tf_x[1] + n_x;
    }
    auto grid1 = [](long long n) { return dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))); };
    hipLaunchKernelGGL(feed_gather_kernel, grid1(SB * M * nf), dim3(256), 0, st, s->i_targets, s->tf_xin, B, S, M, r, nf);
    const float* x = s->tf_xin;
    for (int i = 0; i < np; ++i) {
        const int P = d.prenet_units[i];
        const ConvLayer& L = m->prenet_plain[i];
        float* y = tape_xp ? tape_xp[i] : s->tf_x[i & 1];
        T2_TRY(launch_conv(m, L, x, L.cin, y, P, (int)SB, (int)SB, 1, nullptr, 0));
        hipLaunchKernelGGL(prenet_dropout_all_kernel, grid1(SB * P), dim3(256), 0, st, y, i == np - 1 ? s->tf_frag : nullptr, s->tf_stride, S, B, P, MB,
                           drop_rate, seed, T2_STREAM_PRENET + i);
        x = y;
    }
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// Enqueue one chunk of `chunk` steps followed by the base-step advance.
// Deferred projection of a teacher-forced decode (rnn_wrappers.py:105-109 for all S steps at once): y_all [S*B][D] .
// [D][M*r + 1] + bias as one GEMM, then rows (t, b) scattered to mel [B][S*r][M] / stop [B][S].
// with_y: also the [S][B][D] store of LSTM 2's raw outputs (the training tape has its own)
static int teacher_projection_buffers(t2_state* s, bool with_y) {
    if (s->proj_tmp) return T2_OK;
    const t2_dims& d = s->m->d;
    const size_t SB = (size_t)s->cap * s->B, n_t = align_up(SB * s->m->proj_plain.cout, 64), n_y = with_y ? SB * d.decoder_lstm_units : 0;
    hipError_t e = hipMalloc((void**)&s->proj_tmp, (n_t + n_y) * 4);
    if (e != hipSuccess) { s->proj_tmp = nullptr; return fail(T2_ENOMEM, "hipMalloc(deferred projection buffers) failed: %s", hipGetErrorString(e)); }
    s->y_all = with_y ? s->proj_tmp + n_t : nullptr;
    return T2_OK;
}
static int teacher_projection(t2_state* s, const float* y_all, float* mel, long long mel_ld, float* stop, int stop_ld) {
    t2_model* m = s->m;
    const t2_dims& d = m->d;
    const int S = s->cap, B = s->B, Np = m->proj_plain.cout, ncol = d.num_mels * d.outputs_per_step;
    T2_TRY(launch_conv(m, m->proj_plain, y_all, d.decoder_lstm_units, s->proj_tmp, Np, S * B, S * B, 0, nullptr, 0));
    const long long n = (long long)S * B * (ncol + 1);
    hipLaunchKernelGGL(proj_scatter_kernel, dim3((unsigned)std::max<long long>(1, std::min<long long>(2048, (n + 255) / 256))), dim3(256), 0, m->ctx->stream,
                       s->proj_tmp, Np, S, B, ncol, mel, mel_ld, stop, stop_ld);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int enqueue_chunk(t2_state* s, int mode, int chunk) {
    hipStream_t st = s->m->ctx->stream;
    const StepIO io = loop_io(s, mode);
    for (int i = 0; i < chunk; ++i) T2_TRY(launch_step(s, i, io));
    hipLaunchKernelGGL(advance_ctl_kernel, dim3(1), dim3(64), 0, st, s->ctl, chunk);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// Fused schedule: step t's projection launch produces the first PreNet layer of step t + 1; step 0 takes it from here
// (go frame = zeros in xf after state_zero: helpers.py:187-188).
static int decode_prologue(t2_state* s, int mode) {
    t2_model* m = s->m;
    const StepIO io = loop_io(s, mode);
    if (!fused_mode(m, s->MB, io.feed_mode, false)) return T2_OK;
    SkinnyCore c{};
    const Packed& p = m->prenet[0];
    c.w[0] = c.w[1] = p.w; c.bias[0] = c.bias[1] = p.bias;
    c.kb0 = 0; c.kbtot = p.kbtot; c.kb_w0 = 0; c.kb_full = p.kbtot; c.ctl = s->ctl; c.step_off = 0; c.B = s->B; c.t_hint = -1;
    c.seg[0][1] = FragBuf{s->xf, 0, 1, 0, 0};
    DenseEpi e{};
    e.N = m->d.prenet_units[0]; e.relu = 1; e.out_frag = FragBuf{s->xp[0], 0, 1, 0, 0}; e.frag_col0 = 0; e.stop_col = -1;
    launch_dense(s->MB, dim3(p.n_wg), m->ctx->stream, c, e);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

static int run_chunk(t2_state* s, int mode) {
    t2_model* m = s->m;
    hipStream_t st = m->ctx->stream;
    const int chunk = m->opt_chunk;
    if (!m->opt_graph) return enqueue_chunk(s, mode, chunk);
    if (s->gexec[mode] == nullptr || s->gchunk[mode] != chunk) {
        if (s->gexec[mode]) { hipGraphExecDestroy(s->gexec[mode]); s->gexec[mode] = nullptr; }
        hipGraph_t g = nullptr;
        HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int r = enqueue_chunk(s, mode, chunk);
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

// Free-running decode (the helper feeds the model's own output back: helpers.py:86, :135-136) as ONE persistent launch.
static inline bool persist_mode(const t2_state* s) {
    const t2_model* m = s->m;
    return m->opt_persist && m->persist_holdoff == 0 && s->pz_xs != nullptr && m->proj_fused_ok && m->opt_projfuse && !m->opt_lstm_only;
}
// Teacher-forced decode as one persistent launch: PreNets of all steps before it (teacher_prenets), projections after it
// (teacher_projection); needs neither the composed projection nor the stop rule.
static inline bool persist_mode_teacher(const t2_state* s) {
    const t2_model* m = s->m;
    return m->opt_persist && m->persist_holdoff == 0 && m->opt_tfproj && s->pz_xs != nullptr && s->tf_frag != nullptr && s->y_all != nullptr &&
           !m->opt_lstm_only;
}
struct TrainTape;
static void persist_fill_tape(pz::PTape& tp, const TrainTape* tape);
static int persist_decode(t2_state* s, int max_steps, int min_iters, int stop_en, bool teacher = false, const TrainTape* tape = nullptr) {
    t2_model* m = s->m;
    const t2_dims& d = m->d;
    hipStream_t st = m->ctx->stream;
    pz::PArgs a{};
    pz::PExtra x{};
    a.wA = m->att_lstm.w; a.bA = m->att_lstm.bias; a.w1 = m->lstm1.w; a.b1 = m->lstm1.bias; a.w2 = m->lstm2.w; a.b2 = m->lstm2.bias;
    a.wP = m->proj_fused.w; a.bP = m->proj_fused.bias; a.wN = m->prenet[1].w; a.bN = m->prenet[1].bias;
    a.wQ = m->wq_pk.w; a.conv_w = m->conv_w; a.conv_b = m->conv_b; a.wloc = m->wloc; a.v_a = m->v_a; a.b_a = m->b_a;
    a.keys = s->keys; a.values = s->values; a.lens = s->lens;
    a.xs = s->pz_xs; a.xs_bytes = s->pz_bytes;
    a.oX1 = s->pz_off[0]; a.oX2 = s->pz_off[1]; a.oHAr = s->pz_off[2]; a.oHA = s->pz_off[3]; a.oCTX = s->pz_off[4];
    a.oY1 = s->pz_off[5]; a.oH1 = s->pz_off[6]; a.oY2 = s->pz_off[7]; a.oH2 = s->pz_off[8]; a.oQP = s->pz_off[9];
    a.oEN = s->pz_off[10]; a.oCUM = s->pz_off[11]; a.oSTOP = s->pz_off[12];
    a.bar = s->pz_bar; a.ctl = s->ctl;
    a.cA_out = s->c_att; a.c1_out = s->c1; a.c2_out = s->c2;
    const int M = d.num_mels, r = d.outputs_per_step;
    a.mel = s->o_mel; a.mel_ld = (long long)s->cap * r * M; a.mel_ts = (long long)r * M;
    a.stop = s->o_stop; a.stop_ld = s->cap; a.hist = s->o_hist;
    a.B = s->B; a.T = s->T; a.C = persist_chunks(s->B); a.CH = (s->T + a.C - 1) / a.C;
    a.F = d.attention_filters; a.KA = d.attention_kernel; a.MR = M * r;
    a.max_steps = max_steps; a.min_iters = min_iters; a.stop_en = stop_en; a.cumulative = d.cumulative_weights;
    a.zc = d.zoneout_cell; a.zh = d.zoneout_output; a.fb = d.forget_bias;
    if (m->att_lstm.n_wg != pz::NWG || m->att_lstm.kbtot != pz::KBX + pz::KBE + pz::KBH || m->lstm1.kbtot != 2 * pz::KBH + pz::KBE ||
        m->lstm2.kbtot != 2 * pz::KBH || m->wq_pk.n_wg != 8 || m->wq_pk.kbtot != pz::KBH ||
        (!teacher && (m->proj_fused.n_wg != 28 || m->proj_fused.kbtot != pz::KBH || m->prenet[1].n_wg != 16 || m->prenet[1].kbtot != pz::KBX)))
        return fail(T2_ESTATE, "internal: packed operands do not have the persistent decoder's geometry");
    if (teacher) {
        if (s->tf_stride != (long long)s->MB * 16 * pz::KBX * 16 || (long long)max_steps * s->tf_stride * 4 > 0x7fffffffLL)
            return fail(T2_ESTATE, "internal: teacher-forced PreNet operands do not have the persistent decoder's geometry");
        x.x2_all = s->tf_frag; x.x2_stride = (unsigned)(s->tf_stride * 4); x.x2_bytes = (unsigned)((long long)max_steps * s->tf_stride * 4);
        x.y_all = tape ? nullptr : s->y_all;            // the training tape keeps LSTM 2's raw outputs itself (hraw[2])
        for (int i = 0; i < 4; ++i) x.oB1[i] = s->pz_off[13 + i];
        if (tape) persist_fill_tape(x.tp, tape);
    } else {
        // x1 of step 0 = relu(bias of the first PreNet layer): the go frame is zeros (helpers.py:187-188)
        hipLaunchKernelGGL(pz::init_x1_kernel, dim3(ceil_div(s->B * d.prenet_units[0], 256)), dim3(256), 0, st, s->pz_xs + a.oX1 / 4,
                           m->prenet[0].bias, s->B, s->MB, d.prenet_units[0]);
    }
    const size_t lds = persist_lds_bytes(s->B, s->T);
    const dim3 grid(pz::NWG), blk(pz::NT);
    const int mode = teacher ? (m->opt_tf2 ? (tape ? 4 : 3) : (tape ? 2 : 1)) : 0;        // pz::decode_kernel's MODE (3 / 4: two-chain teacher schedule)
    XwgGate gate;
    T2_TRY(gate.begin(m->ctx));
#define T2_PZ_LAUNCH(MBv) do { \
        if (mode == 0) hipLaunchKernelGGL((pz::decode_kernel<MBv, 0>), grid, blk, lds, st, a, x); \
        else if (mode == 1) hipLaunchKernelGGL((pz::decode_kernel<MBv, 1>), grid, blk, lds, st, a, x); \
        else if (mode == 2) hipLaunchKernelGGL((pz::decode_kernel<MBv, 2>), grid, blk, lds, st, a, x); \
        else if (mode == 3) hipLaunchKernelGGL((pz::decode_kernel<MBv, 3>), grid, blk, lds, st, a, x); \
        else hipLaunchKernelGGL((pz::decode_kernel<MBv, 4>), grid, blk, lds, st, a, x); } while (0)
    if (s->MB == 1) T2_PZ_LAUNCH(1);
    else if (s->MB == 2) T2_PZ_LAUNCH(2);
    else T2_PZ_LAUNCH(4);
#undef T2_PZ_LAUNCH
    HIP_TRY(hipGetLastError());
    ++m->n_persist;
    return inject_timeout(m, s->ctl + 5);
}

static int inject_timeout(t2_model* m, int* err_word) {
    if (m->opt_inject_timeout <= 0) return T2_OK;
    --m->opt_inject_timeout;
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)err_word, 1, 1, m->ctx->stream));
    return T2_OK;
}

// After a persistent launch: did one of its hand-off waits run out of time (ctl[5] / ctl[6], set by every workgroup that gave
// up)?  Synchronises the stream.  A timeout is counted and switches the next 16 decodes of this model to the launch-per-phase
// schedule; the caller re-runs the decode it just tried with that schedule, so no invalid buffer is ever handed back.
static void note_persist_timeout(t2_model* m) { ++m->n_persist_timeouts; m->persist_holdoff = 16; }
static inline void holdoff_tick(t2_model* m) { if (m->persist_holdoff > 0) --m->persist_holdoff; }
static int persist_timed_out(t2_state* s, bool* timed_out) {
    hipStream_t st = s->m->ctx->stream;
    int* h = s->m->h_ctl;
    HIP_TRY(hipMemcpyAsync(h, s->ctl, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *timed_out = h[5] != 0 || h[6] != 0;
    if (*timed_out) note_persist_timeout(s->m);
    return T2_OK;
}

extern "C" int t2_decode_teacher(t2_model* m, const float* memory, const int64_t* lens, const float* targets,
                                 int B, int T_in, int T_out, int teacher_forcing,
                                 float* mel, float* stop, float* align) {
    if (!m || !memory || !targets || !mel || !stop || !align) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    T2_TRY(check_batch(B, T_in));
    const t2_dims& d = m->d;
    const int M = d.num_mels, r = d.outputs_per_step;
    if (T_out < r || T_out % r != 0) return fail(T2_EINVAL, "T_out must be a positive multiple of outputs_per_step");
    // the loop states are shared with t2_decode_free_launch: zeroing or evicting the state of a decode in flight would corrupt
    // its step count (t2_decode_free_finish reads the state's control block)
    if (m->pending) return fail(T2_ESTATE, "a free-running decode is in flight; call t2_decode_free_finish first");
    HIP_TRY(hipSetDevice(m->ctx->device));
    hipStream_t st = m->ctx->stream;
    const int S = T_out / r;
    t2_state* s = nullptr;
    holdoff_tick(m);
    T2_TRY(get_loop_state(m, B, T_in, S, &s));
    T2_TRY(state_init(s, memory, lens));
    HIP_TRY(hipMemcpyAsync(s->i_targets, targets, (size_t)B * T_out * M * 4, hipMemcpyDeviceToDevice, st));
    const int mode = teacher_forcing ? 0 : 1;
    // one pass of the loop; a persistent launch that timed out is followed by a second pass on the launch-per-phase schedule
    for (int pass = 0; pass < 2; ++pass) {
        T2_TRY(state_zero(s, S, 0, 0));
        if (teacher_forcing) T2_TRY(teacher_prenets(s, 0.f, 0));
        if (teacher_forcing && m->opt_tfproj) T2_TRY(teacher_projection_buffers(s, true));
        const bool persist_tf = teacher_forcing && persist_mode_teacher(s);
        const bool persist = persist_tf || (!teacher_forcing && persist_mode(s));
        m->last_loop = s; m->last_persist = persist;
        if (persist) T2_TRY(persist_decode(s, S, 0, 0, persist_tf));
        else T2_TRY(decode_prologue(s, mode));
        const bool deferred = persist_tf || (!persist && loop_io(s, mode).defer_proj != 0);
        for (int done = 0; !persist && done < S; done += m->opt_chunk) T2_TRY(run_chunk(s, mode));
        if (deferred) T2_TRY(teacher_projection(s, s->y_all, s->o_mel, (long long)s->cap * r * M, s->o_stop, s->cap));
        HIP_TRY(hipMemcpyAsync(mel, s->o_mel, (size_t)B * T_out * M * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)B * S * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(align, s->o_hist, (size_t)S * B * T_in * 4, hipMemcpyDeviceToDevice, st));
        if (!persist) break;
        bool timed_out = false;
        T2_TRY(persist_timed_out(s, &timed_out));          // the one host synchronisation of this call (persistent schedule only)
        if (!timed_out) break;
    }
    return T2_OK;
}

static int free_prepare(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in, int max_iters, int min_iters,
                        float* mel, float* stop, float* align, t2_state** out) {
    if (!m || !memory || !mel || !stop || !align) return fail(T2_EINVAL, "null argument");
    if (!m->finalized) return fail(T2_ESTATE, "model not finalized");
    T2_TRY(check_batch(B, T_in));
    if (max_iters < 1) return fail(T2_EINVAL, "max_iters must be >= 1");
    if (m->pending) return fail(T2_ESTATE, "a free-running decode is already in flight; call t2_decode_free_finish");
    HIP_TRY(hipSetDevice(m->ctx->device));
    t2_state* s = nullptr;
    holdoff_tick(m);
    T2_TRY(get_loop_state(m, B, T_in, max_iters, &s));
    T2_TRY(state_init(s, memory, lens));
    T2_TRY(state_zero(s, max_iters, min_iters, 1));
    m->last_loop = s; m->last_persist = persist_mode(s);
    *out = s;
    return T2_OK;
}

static int free_copy_out(t2_state* s, float* mel, float* stop, float* align) {
    const t2_dims& d = s->m->d;
    hipStream_t st = s->m->ctx->stream;
    const size_t cap = s->cap;
    HIP_TRY(hipMemcpyAsync(mel, s->o_mel, (size_t)s->B * cap * d.outputs_per_step * d.num_mels * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(stop, s->o_stop, (size_t)s->B * cap * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(align, s->o_hist, cap * s->B * s->T * 4, hipMemcpyDeviceToDevice, st));
    return T2_OK;
}

static int read_steps(t2_state* s, int* steps_out, bool* timed_out) {
    hipStream_t st = s->m->ctx->stream;
    int* h = s->m->h_ctl;
    HIP_TRY(hipMemcpyAsync(h, s->ctl, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *steps_out = h[CTL_FINAL] != 0 ? h[CTL_FINAL] : h[CTL_MAX];
    *timed_out = h[5] != 0 || h[6] != 0;
    if (*timed_out) note_persist_timeout(s->m);
    return T2_OK;
}

// The free-running loop on the launch-per-phase schedule: hipGraph chunks enqueued two ahead of the host's view of the finished
// flag (chunks after the stop step are no-ops on the device, so over-enqueueing is harmless and the host never stalls the GPU).
static int free_loop_launches(t2_state* s, int max_iters, float* mel, float* stop, float* align, int* steps_out) {
    t2_model* m = s->m;
    hipStream_t st = m->ctx->stream;
    hipEvent_t* ev = m->flag_ev;          // created once per model (t2_model_finalize), as is the pinned mirror
    int* h_flag = m->h_flag;
    int rc = decode_prologue(s, 1), n = 0;
    bool finished = false;
    for (int done = 0; rc == T2_OK && done < max_iters && !finished; done += m->opt_chunk, ++n) {
        if (n >= 2) {
            hipEventSynchronize(ev[n & 1]);
            if (h_flag[(n & 1) * 8 + CTL_FINAL] != 0) { finished = true; break; }
        }
        rc = run_chunk(s, 1);
        if (rc != T2_OK) break;
        hipMemcpyAsync(h_flag + (n & 1) * 8, s->ctl, 8 * sizeof(int), hipMemcpyDeviceToHost, st);
        hipEventRecord(ev[n & 1], st);
    }
    bool unused = false;
    if (rc == T2_OK) rc = free_copy_out(s, mel, stop, align);
    if (rc == T2_OK) rc = read_steps(s, steps_out, &unused);
    hipStreamSynchronize(st);
    return rc;
}

// A persistent free-running decode whose hand-off waits ran out of time: the same decode again, launch per phase (the
// attention constructor's side effects -- values, keys, lengths -- are still in the state; only the loop state is re-zeroed).
static int free_rerun_after_timeout(t2_state* s, int max_iters, int min_iters, float* mel, float* stop, float* align, int* steps_out) {
    T2_TRY(state_zero(s, max_iters, min_iters, 1));
    s->m->last_persist = false;
    return free_loop_launches(s, max_iters, mel, stop, align, steps_out);
}

extern "C" int t2_decode_free_launch(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in,
                                     int max_iters, int min_iters, float* mel, float* stop, float* align) {
    t2_state* s = nullptr;
    T2_TRY(free_prepare(m, memory, lens, B, T_in, max_iters, min_iters, mel, stop, align, &s));
    if (persist_mode(s)) T2_TRY(persist_decode(s, max_iters, min_iters, 1));
    else {
        T2_TRY(decode_prologue(s, 1));
        for (int done = 0; done < max_iters; done += m->opt_chunk) T2_TRY(run_chunk(s, 1));
    }
    T2_TRY(free_copy_out(s, mel, stop, align));
    m->pending = s;
    m->pend_mel = mel; m->pend_stop = stop; m->pend_align = align; m->pend_max_iters = max_iters; m->pend_min_iters = min_iters;
    return T2_OK;
}

extern "C" int t2_decode_free_finish(t2_model* m, int* steps_out) {
    if (!m || !steps_out) return fail(T2_EINVAL, "null argument");
    if (!m->pending) return fail(T2_ESTATE, "no free-running decode in flight");
    t2_state* s = m->pending;
    m->pending = nullptr;
    bool timed_out = false;
    T2_TRY(read_steps(s, steps_out, &timed_out));
    if (timed_out) return free_rerun_after_timeout(s, m->pend_max_iters, m->pend_min_iters, m->pend_mel, m->pend_stop, m->pend_align, steps_out);
    return T2_OK;
}

extern "C" int t2_decode_free(t2_model* m, const float* memory, const int64_t* lens, int B, int T_in,
                              int max_iters, int min_iters, float* mel, float* stop, float* align, int* steps_out) {
    if (!steps_out) return fail(T2_EINVAL, "steps_out is null");
    t2_state* s = nullptr;
    T2_TRY(free_prepare(m, memory, lens, B, T_in, max_iters, min_iters, mel, stop, align, &s));
    if (persist_mode(s)) {
        T2_TRY(persist_decode(s, max_iters, min_iters, 1));
        T2_TRY(free_copy_out(s, mel, stop, align));
        bool timed_out = false;
        T2_TRY(read_steps(s, steps_out, &timed_out));
        if (timed_out) return free_rerun_after_timeout(s, max_iters, min_iters, mel, stop, align, steps_out);
        return T2_OK;
    }
    return free_loop_launches(s, max_iters, mel, stop, align, steps_out);
}

#ifdef PZ_STAMPS
extern "C" int t2_debug_stamps(unsigned long long* out) {     // diagnostic builds only: [256][24] tick sums (10 ns)
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(pz::g_pz_stamps), sizeof(unsigned long long) * pz::NWG * 24));
    return T2_OK;
}
#endif

extern "C" int t2_model_get_stat(t2_model* m, const char* name, double* out) {
    if (!m || !name || !out) return fail(T2_EINVAL, "null argument");
    if (!strcmp(name, "persist_launches")) { *out = (double)m->n_persist; return T2_OK; }
    if (!strcmp(name, "persist_timeouts")) { *out = (double)m->n_persist_timeouts; return T2_OK; }
    if (!strcmp(name, "enc_persist_launches")) { *out = (double)m->n_enc_persist; return T2_OK; }
    if (!strcmp(name, "bwd_persist_launches")) { *out = (double)m->n_bwd_persist; return T2_OK; }
    if (!strcmp(name, "lstm_launches")) { *out = (double)(g_timer.used / 2); return T2_OK; }
    if (!strcmp(name, "lstm_ms")) {
        HIP_TRY(hipStreamSynchronize(m->ctx->stream));
        double sum = 0;
        for (size_t i = 0; i + 1 < g_timer.used; i += 2) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, g_timer.ev[i], g_timer.ev[i + 1]));
            sum += ms;
        }
        *out = sum;
        return T2_OK;
    }
    return fail(T2_EINVAL, "unknown stat %s", name);
}

// ------------------------------------------------------------------------------------------------
// losses and optimizer block
// ------------------------------------------------------------------------------------------------
#include "train.h"

// ------------------------------------------------------------------------------------------------
// Tacotron v1 path
// ------------------------------------------------------------------------------------------------
#include "v1.h"

// ------------------------------------------------------------------------------------------------
// training step: training-mode forward with a tape, backward pass (BASELINE.json configs[3])
// ------------------------------------------------------------------------------------------------
#include "train_step.h"

// ------------------------------------------------------------------------------------------------
// training step of the Tacotron-1 PostNet model (BASELINE.json configs[4])
// ------------------------------------------------------------------------------------------------
#include "v1_train.h"
#include "v1_dec_train.h"

// ------------------------------------------------------------------------------------------------
// spectrogram -> waveform (Griffin-Lim)
// ------------------------------------------------------------------------------------------------
#include "vocoder.h"
