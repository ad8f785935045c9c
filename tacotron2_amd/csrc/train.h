// Training-step math of the path: the two losses with
// TensorFlow's SUM_BY_NONZERO_WEIGHTS reduction and the optimizer block (global-norm clipping +
// Adam, TF semantics) over a flat fp32 parameter / gradient buffer.  All of it is streaming,
// HBM-bound work: float4 grid-stride loops, per-block partials, a fixed-order final reduction in
// double (bit-reproducible run to run), no atomics.  Included by t2mi.hip (uses its fail/HIP_TRY).
#pragma once

#define T2_RED_THREADS 256
#define T2_RED_MAX_BLOCKS 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// block-wide sums of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b) {
    __shared__ float sa[T2_RED_THREADS / 64], sb[T2_RED_THREADS / 64];
    a = wave_sum(a); b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sa[wave] = a; sb[wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < T2_RED_THREADS / 64; ++w) { ta += sa[w]; tb += sb[w]; }
        a = ta; b = tb;
    }
}

// T1: sum over [B*T][M] of |y_hat - y| * mask[row], and the number of non-zero broadcast weights
// (reference models.py:136-144; tf.losses.compute_weighted_loss default reduction).
__global__ __launch_bounds__(T2_RED_THREADS) void spec_loss_partial_kernel(const float* __restrict__ y_hat, const float* __restrict__ y,
                                                                           const float* __restrict__ mask, long long rows, int M,
                                                                           float* __restrict__ partial) {
    float s = 0.f, c = 0.f;
    const long long n = rows * M;
    if ((M & 3) == 0) {
        const long long n4 = n >> 2;
        const int M4 = M >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            const float w = mask[i / M4];
            const float4 a = reinterpret_cast<const float4*>(y_hat)[i], b = reinterpret_cast<const float4*>(y)[i];
            s += (fabsf(a.x - b.x) + fabsf(a.y - b.y) + fabsf(a.z - b.z) + fabsf(a.w - b.w)) * w;
            c += (w != 0.f) ? 4.f : 0.f;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
            const float w = mask[i / M];
            s += fabsf(y_hat[i] - y[i]) * w;
            c += (w != 0.f) ? 1.f : 0.f;
        }
    }
    block_sum2(s, c);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = c; }
}

// T2: sigmoid cross entropy of the stop logits, max(x,0) - x*z + log(1 + exp(-|x|)), weighted by
// mask (reference models.py:146-148; tf.losses.sigmoid_cross_entropy).
__global__ __launch_bounds__(T2_RED_THREADS) void binary_loss_partial_kernel(const float* __restrict__ logit, const float* __restrict__ done,
                                                                             const float* __restrict__ mask, long long n,
                                                                             float* __restrict__ partial) {
    float s = 0.f, c = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = logit[i], z = done[i], w = mask[i];
        s += (fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)))) * w;
        c += (w != 0.f) ? 1.f : 0.f;
    }
    block_sum2(s, c);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s; partial[2 * blockIdx.x + 1] = c; }
}

// fixed-order parallel sum of the per-block partials in double: thread i adds partials i, i + 256, ...,
// then a shared-memory tree; the order never depends on timing, so results are bit-reproducible
__device__ __forceinline__ double final_sum(const float* __restrict__ partial, int nblocks, int stride, double* sh) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += T2_RED_THREADS) s += (double)partial[(size_t)i * stride];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = T2_RED_THREADS / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// out[0] = sum, out[1] = count, out[2] = sum / count (0 when count == 0: TF's _safe_div)
__global__ __launch_bounds__(T2_RED_THREADS) void loss_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
    __shared__ double sh[T2_RED_THREADS];
    const double s = final_sum(partial, nblocks, 2, sh);
    const double c = final_sum(partial + 1, nblocks, 2, sh);
    if (threadIdx.x == 0) { out[0] = (float)s; out[1] = (float)c; out[2] = c > 0.0 ? (float)(s / c) : 0.f; }
}

// T4a: sum of squares of the gradient buffer (tf.clip_by_global_norm -> tf.global_norm)
__global__ __launch_bounds__(T2_RED_THREADS) void sumsq_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
    float s = 0.f, unused = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(g)[i];
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += g[i] * g[i];
    block_sum2(s, unused);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// norm_out[0] = grad_scale * sqrt(sum of partials): the global norm of grad_scale * g
__global__ __launch_bounds__(T2_RED_THREADS) void norm_final_kernel(const float* __restrict__ partial, int nblocks, float grad_scale, float* __restrict__ norm_out) {
    __shared__ double sh[T2_RED_THREADS];
    const double s = final_sum(partial, nblocks, 1, sh);
    if (threadIdx.x == 0) norm_out[0] = (float)((double)grad_scale * sqrt(s));
}

// T4b: clip_by_global_norm + tf.train.AdamOptimizer._apply_dense (reference models.py:67-76):
//   g' = grad_scale * g * clip / max(norm, clip)          (clip <= 0: no clipping)
//   m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;  p -= lr_t * m / (sqrt(v) + eps)
// with lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the host (epsilon OUTSIDE the sqrt and the
// bias correction folded into lr_t: TF, not the Kingma-Ba / torch form).
struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    const float* norm;      // device scalar written by norm_final_kernel
    long long n;
    float lr_t, b1, b2, eps, clip, grad_scale;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float gs) {
    g *= gs;
    m = a.b1 * m + (1.f - a.b1) * g;
    v = a.b2 * v + (1.f - a.b2) * g * g;
    p -= a.lr_t * m / (sqrtf(v) + a.eps);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    float gs = a.grad_scale;
    if (a.clip > 0.f) gs *= a.clip / fmaxf(a.norm[0], a.clip);
    const long long n4 = a.n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4*>(a.p)[i], m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        adam_one(p.x, g.x, m.x, v.x, a, gs); adam_one(p.y, g.y, m.y, v.y, a, gs);
        adam_one(p.z, g.z, m.z, v.z, a, gs); adam_one(p.w, g.w, m.w, v.w, a, gs);
        reinterpret_cast<float4*>(a.p)[i] = p; reinterpret_cast<float4*>(a.m)[i] = m; reinterpret_cast<float4*>(a.v)[i] = v;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x)
        adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a, gs);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct t2_optim {
    t2_ctx* ctx = nullptr;
    long long n = 0;
    float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    long long step = 0;          // number of apply calls so far (Adam's t - 1)
    float* m = nullptr; float* v = nullptr;
    float* partial = nullptr;    // [T2_RED_MAX_BLOCKS]
    float* norm = nullptr;       // [1]
};

static int red_blocks(long long work_items) {
    long long b = (work_items + T2_RED_THREADS * 4 - 1) / (T2_RED_THREADS * 4);
    if (b < 1) b = 1;
    if (b > T2_RED_MAX_BLOCKS) b = T2_RED_MAX_BLOCKS;
    return (int)b;
}

static int loss_ws(t2_ctx* ctx, float** ws) {
    if (ctx->red_ws == nullptr) HIP_TRY(hipMalloc(&ctx->red_ws, (size_t)T2_RED_MAX_BLOCKS * 2 * sizeof(float)));
    *ws = ctx->red_ws;
    return T2_OK;
}

extern "C" int t2_spec_loss(t2_ctx* ctx, const float* y_hat, const float* y, const float* mask, int B, int T, int M, float* out3) {
    if (!ctx || !y_hat || !y || !mask || !out3) return fail(T2_EINVAL, "null argument");
    if (B < 1 || T < 1 || M < 1) return fail(T2_EINVAL, "bad shape [%d,%d,%d]", B, T, M);
    HIP_TRY(hipSetDevice(ctx->device));
    float* ws;
    T2_TRY(loss_ws(ctx, &ws));
    const long long rows = (long long)B * T;
    if ((M & 3) == 0 && ((((uintptr_t)y_hat) | ((uintptr_t)y)) & 15) != 0) return fail(T2_EINVAL, "y_hat / y must be 16-byte aligned");
    const int nb = red_blocks(rows * M / 4 + 1);
    hipLaunchKernelGGL(spec_loss_partial_kernel, dim3(nb), dim3(T2_RED_THREADS), 0, ctx->stream, y_hat, y, mask, rows, M, ws);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(T2_RED_THREADS), 0, ctx->stream, ws, nb, out3);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

// ---- allocation-free variant of t2_spec_loss (no reference counterpart; for callers that cannot allocate: code under HIP
// graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same two kernels; the per-block partials go to a
// file-scope device array instead of the lazily hipMalloc'ed ctx workspace.  Each GPU holds its own copy of the array at its
// own address, looked up once per device in t2_ctx_create (the library's set-up call), with that device current.
#define T2_NOALLOC_MAX_ELEMS (1LL << 28)            /* B*T*M accepted by t2_spec_loss_noalloc (1 GiB per tensor) */
static __device__ __align__(256) float scratch[T2_RED_MAX_BLOCKS * 2];      // 16 KiB per GPU, held until the process ends
static float* g_scratch_ptr[64] = {nullptr};        // device address of `scratch` on each GPU (host code never uses the name itself)

static int noalloc_setup(int device) {              // called with `device` current
    if (device < 0 || device >= 64) return fail(T2_EINVAL, "device index %d beyond the per-device table", device);
    if (g_scratch_ptr[device] != nullptr) return T2_OK;
    float* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = ptr;
    return T2_OK;
}

extern "C" int t2_spec_loss_noalloc(t2_ctx* ctx, const float* y_hat, const float* y, const float* mask, int B, int T, int M, float* out3) {
    if (!ctx || !y_hat || !y || !mask || !out3) return fail(T2_EINVAL, "null argument");
    if (B < 1 || T < 1 || M < 1) return fail(T2_EINVAL, "bad shape [%d,%d,%d]", B, T, M);
    if ((long long)B * T * M > T2_NOALLOC_MAX_ELEMS)
        return fail(T2_EINVAL, "t2_spec_loss_noalloc: %lld elements exceed the maximum of %lld", (long long)B * T * M, (long long)T2_NOALLOC_MAX_ELEMS);
    float* ws = (ctx->device >= 0 && ctx->device < 64) ? g_scratch_ptr[ctx->device] : nullptr;
    if (ws == nullptr) return fail(T2_ESTATE, "t2_spec_loss_noalloc: the scratch array of device %d was not set up", ctx->device);
    HIP_TRY(hipSetDevice(ctx->device));
    const long long rows = (long long)B * T;
    if ((M & 3) == 0 && ((((uintptr_t)y_hat) | ((uintptr_t)y)) & 15) != 0) return fail(T2_EINVAL, "y_hat / y must be 16-byte aligned");
    const int nb = red_blocks(rows * M / 4 + 1);
    // every partial the final kernel reads (2 * nb floats) is written by this call's first kernel: nothing has to start at zero
    hipLaunchKernelGGL(spec_loss_partial_kernel, dim3(nb), dim3(T2_RED_THREADS), 0, ctx->stream, y_hat, y, mask, rows, M, ws);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(T2_RED_THREADS), 0, ctx->stream, ws, nb, out3);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_binary_loss(t2_ctx* ctx, const float* logits, const float* done, const float* mask, int B, int S, float* out3) {
    if (!ctx || !logits || !done || !mask || !out3) return fail(T2_EINVAL, "null argument");
    if (B < 1 || S < 1) return fail(T2_EINVAL, "bad shape [%d,%d]", B, S);
    HIP_TRY(hipSetDevice(ctx->device));
    float* ws;
    T2_TRY(loss_ws(ctx, &ws));
    const long long n = (long long)B * S;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(binary_loss_partial_kernel, dim3(nb), dim3(T2_RED_THREADS), 0, ctx->stream, logits, done, mask, n, ws);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(T2_RED_THREADS), 0, ctx->stream, ws, nb, out3);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_optim_destroy(t2_optim* o) {
    if (!o) return T2_OK;
    (void)hipSetDevice(o->ctx->device);
    if (o->m) (void)hipFree(o->m);
    if (o->v) (void)hipFree(o->v);
    if (o->partial) (void)hipFree(o->partial);
    if (o->norm) (void)hipFree(o->norm);
    delete o;
    return T2_OK;
}

extern "C" int t2_optim_create(t2_ctx* ctx, int64_t n_params, float beta1, float beta2, float epsilon, t2_optim** out) {
    if (!ctx || !out) return fail(T2_EINVAL, "null argument");
    if (n_params < 1) return fail(T2_EINVAL, "n_params must be >= 1");
    if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(epsilon > 0.f))
        return fail(T2_EINVAL, "Adam needs 0 <= beta < 1 and epsilon > 0");
    HIP_TRY(hipSetDevice(ctx->device));
    t2_optim* o = new t2_optim();
    o->ctx = ctx; o->n = n_params; o->b1 = beta1; o->b2 = beta2; o->eps = epsilon;
    auto bail = [&](hipError_t e, const char* what) {
        t2_optim_destroy(o);
        return fail(e == hipErrorOutOfMemory ? T2_ENOMEM : T2_EHIP, "%s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc(&o->m, (size_t)n_params * 4)) != hipSuccess) return bail(e, "hipMalloc(m)");
    if ((e = hipMalloc(&o->v, (size_t)n_params * 4)) != hipSuccess) return bail(e, "hipMalloc(v)");
    if ((e = hipMalloc(&o->partial, (size_t)T2_RED_MAX_BLOCKS * 4)) != hipSuccess) return bail(e, "hipMalloc(partial)");
    if ((e = hipMalloc(&o->norm, 16)) != hipSuccess) return bail(e, "hipMalloc(norm)");
    if ((e = hipMemsetAsync(o->m, 0, (size_t)n_params * 4, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(o->v, 0, (size_t)n_params * 4, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    *out = o;
    return T2_OK;
}

static int optim_norm(t2_optim* o, const float* grads, float grad_scale) {
    if ((((uintptr_t)grads) & 15) != 0) return fail(T2_EINVAL, "gradient buffer must be 16-byte aligned");
    const int nb = red_blocks(o->n / 4 + 1);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(T2_RED_THREADS), 0, o->ctx->stream, grads, o->n, o->partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(T2_RED_THREADS), 0, o->ctx->stream, o->partial, nb, grad_scale, o->norm);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_optim_global_norm(t2_optim* o, const float* grads, float grad_scale, float* norm_out) {
    if (!o || !grads || !norm_out) return fail(T2_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(o->ctx->device));
    T2_TRY(optim_norm(o, grads, grad_scale));
    HIP_TRY(hipMemcpyAsync(norm_out, o->norm, 4, hipMemcpyDeviceToDevice, o->ctx->stream));
    return T2_OK;
}

extern "C" int t2_optim_step(t2_optim* o, float* params, const float* grads, float learning_rate, float clip_norm, float grad_scale) {
    if (!o || !params || !grads) return fail(T2_EINVAL, "null argument");
    if ((((uintptr_t)params) & 15) != 0) return fail(T2_EINVAL, "parameter buffer must be 16-byte aligned");
    HIP_TRY(hipSetDevice(o->ctx->device));
    T2_TRY(optim_norm(o, grads, grad_scale));
    o->step += 1;
    const double t = (double)o->step;
    AdamArgs a{};
    a.p = params; a.g = grads; a.m = o->m; a.v = o->v; a.norm = o->norm; a.n = o->n;
    a.lr_t = (float)((double)learning_rate * std::sqrt(1.0 - std::pow((double)o->b2, t)) / (1.0 - std::pow((double)o->b1, t)));
    a.b1 = o->b1; a.b2 = o->b2; a.eps = o->eps; a.clip = clip_norm; a.grad_scale = grad_scale;
    long long nb = (o->n / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 256 * 16) nb = 256 * 16;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, o->ctx->stream, a);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}

extern "C" int t2_optim_get_state(t2_optim* o, float* m_out, float* v_out, int64_t* step_out) {
    if (!o) return fail(T2_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(o->ctx->device));
    if (m_out) HIP_TRY(hipMemcpyAsync(m_out, o->m, (size_t)o->n * 4, hipMemcpyDeviceToDevice, o->ctx->stream));
    if (v_out) HIP_TRY(hipMemcpyAsync(v_out, o->v, (size_t)o->n * 4, hipMemcpyDeviceToDevice, o->ctx->stream));
    if (step_out) *step_out = o->step;
    return T2_OK;
}

extern "C" int t2_optim_set_state(t2_optim* o, const float* m_in, const float* v_in, int64_t step) {
    if (!o || step < 0) return fail(T2_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(o->ctx->device));
    if (m_in) HIP_TRY(hipMemcpyAsync(o->m, m_in, (size_t)o->n * 4, hipMemcpyDeviceToDevice, o->ctx->stream));
    if (v_in) HIP_TRY(hipMemcpyAsync(o->v, v_in, (size_t)o->n * 4, hipMemcpyDeviceToDevice, o->ctx->stream));
    o->step = step;
    return T2_OK;
}
