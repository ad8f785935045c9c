// Spectrogram -> waveform: the reference's Griffin-Lim (util/audio.py:51-82,88-100,108-115) as HIP kernels behind
// t2_griffin_lim.  Included by t2mi.hip.
//
// One workgroup per STFT frame; the frame's transform is a radix-2 FFT of n_fft = 2 (num_freq - 1) complex points in
// LDS (bit-reversed load, log2(n_fft) in-place stages, twiddles tabulated in LDS with sincospif), used both for
// tf.contrib.signal.inverse_stft (Hermitian extension of the num_freq bins -> inverse FFT -> real part; the imaginary
// parts of the DC and Nyquist bins are dropped as a C2R transform does) and for tf.contrib.signal.stft (windowed frame,
// zero padded at the end to n_fft -> FFT -> bins 0 .. n_fft / 2).  The overlap-add of inverse_stft is a gather (every
// output sample sums its <= ceil(win / hop) frames in frame order: deterministic, no atomics) fused into the loader of
// the next STFT, so one Griffin-Lim iteration is two launches.
#pragma once

#define T2_VOC_THREADS 256
#define T2_VOC_MAX_FFT 4096

struct VocArgs {
    const float* spectrogram;   // [B][T][F] normalised (inv_amp input), or null once S is built
    float* S;                   // [B][T][F] magnitudes ** power
    float2* Z;                  // [B][T][F] complex spectrum to invert (S * angles)
    float* frames;              // [B][T][win] windowed synthesis frames
    float* wav;                 // [B][L]
    int B, T, F, n_fft, log_n, win, hop, L;
    float min_level_db, ref_level_db, power;
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// exp(-2 pi i k / n), k < n / 2, into LDS
__device__ __forceinline__ void voc_twiddles(float2* tw, int n) {
    for (int k = threadIdx.x; k < n / 2; k += T2_VOC_THREADS) {
        float s, c;
        sincospif(2.0f * (float)k / (float)n, &s, &c);
        tw[k] = make_float2(c, -s);
    }
}

// in-place radix-2 decimation-in-time stages over bit-reversed input; inverse: conjugated twiddles (no 1/n scale)
template <bool INVERSE>
__device__ __forceinline__ void voc_fft(float2* buf, const float2* tw, int n, int log_n) {
    for (int s = 1; s <= log_n; ++s) {
        const int half = 1 << (s - 1), tstep = n >> s;
        __syncthreads();
        for (int j = threadIdx.x; j < n / 2; j += T2_VOC_THREADS) {
            const int pos = j & (half - 1), i0 = ((j >> (s - 1)) << s) + pos, i1 = i0 + half;
            float2 w = tw[pos * tstep];
            if (INVERSE) w.y = -w.y;
            const float2 t = cmul(w, buf[i1]), u = buf[i0];
            buf[i0] = make_float2(u.x + t.x, u.y + t.y);
            buf[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int voc_brev(int i, int log_n) { return (int)(__brev((unsigned)i) >> (32 - log_n)); }
__device__ __forceinline__ float voc_hann(int k, int win) { return 0.5f - 0.5f * cospif(2.0f * (float)k / (float)win); }

// inv_amp + pow (util/audio.py:51-59): S = (10 ** ((clip(x, 0, 1) * -min_db + min_db + ref_db) * 0.05)) ** power; Z = S + 0i
__global__ void voc_amp_kernel(VocArgs a) {
    const long long n = (long long)a.B * a.T * a.F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float x = fminf(fmaxf(a.spectrogram[i], 0.f), 1.f) * -a.min_level_db + a.min_level_db;
        const float s = powf(powf(10.0f, (x + a.ref_level_db) * 0.05f), a.power);
        a.S[i] = s;
        a.Z[i] = make_float2(s, 0.f);
    }
}

// inverse_stft of frame (blockIdx.x, utterance blockIdx.y) without the overlap-add: frames[t][k] = irfft(Z[t])[k] * hann[k]
__global__ __launch_bounds__(T2_VOC_THREADS) void voc_istft_kernel(VocArgs a) {
    extern __shared__ float2 voc_smem[];
    float2* buf = voc_smem;
    float2* tw = voc_smem + a.n_fft;
    const int t = blockIdx.x, b = blockIdx.y, n = a.n_fft;
    const float2* z = a.Z + ((size_t)b * a.T + t) * a.F;
    voc_twiddles(tw, n);
    for (int k = threadIdx.x; k < n; k += T2_VOC_THREADS) {
        float2 v = k < a.F ? z[k] : z[n - k];
        if (k >= a.F) v.y = -v.y;                       // X[n - k] = conj(X[k])
        if (k == 0 || k == n / 2) v.y = 0.f;            // a C2R transform ignores these
        buf[voc_brev(k, a.log_n)] = v;
    }
    voc_fft<true>(buf, tw, n, a.log_n);
    float* fr = a.frames + ((size_t)b * a.T + t) * a.win;
    const float inv_n = 1.0f / (float)n;
    for (int k = threadIdx.x; k < a.win; k += T2_VOC_THREADS) fr[k] = buf[k].x * inv_n * voc_hann(k, a.win);
}

// overlap_and_add sample i of utterance b: frames t with t * hop <= i < t * hop + win, summed in frame order
__device__ __forceinline__ float voc_ola(const float* frames, int T, int win, int hop, int i) {
    const int t_hi = min(T - 1, i / hop);
    const int t_lo = max(0, (i - win + hop) / hop);     // smallest t with t * hop + win > i
    float s = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) s += frames[(size_t)t * win + (i - t * hop)];
    return s;
}

// stft of frame t of y = overlap_and_add(frames), then Z[t][f] = S[t][f] * est / max(1e-8, |est|)   (util/audio.py:75-78)
__global__ __launch_bounds__(T2_VOC_THREADS) void voc_stft_kernel(VocArgs a) {
    extern __shared__ float2 voc_smem[];
    float2* buf = voc_smem;
    float2* tw = voc_smem + a.n_fft;
    const int t = blockIdx.x, b = blockIdx.y, n = a.n_fft;
    const float* frames = a.frames + (size_t)b * a.T * a.win;
    voc_twiddles(tw, n);
    for (int k = threadIdx.x; k < n; k += T2_VOC_THREADS) {
        float v = 0.f;
        if (k < a.win) v = voc_ola(frames, a.T, a.win, a.hop, t * a.hop + k) * voc_hann(k, a.win);
        buf[voc_brev(k, a.log_n)] = make_float2(v, 0.f);
    }
    voc_fft<false>(buf, tw, n, a.log_n);
    const size_t row = ((size_t)b * a.T + t) * a.F;
    for (int f = threadIdx.x; f < a.F; f += T2_VOC_THREADS) {
        const float2 e = buf[f];
        const float m = fmaxf(1e-8f, sqrtf(e.x * e.x + e.y * e.y));
        const float s = a.S[row + f];
        a.Z[row + f] = make_float2(s * (e.x / m), s * (e.y / m));
    }
}

__global__ void voc_ola_kernel(VocArgs a) {
    const long long n = (long long)a.B * a.L;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.L), k = (int)(i - (long long)b * a.L);
        a.wav[i] = voc_ola(a.frames + (size_t)b * a.T * a.win, a.T, a.win, a.hop, k);
    }
}

extern "C" int t2_griffin_lim(t2_ctx* ctx, const float* spectrogram, int B, int T, int num_freq, int win_length, int hop_length,
                              float min_level_db, float ref_level_db, float power, int iters, float* wav) {
    if (!ctx || !spectrogram || !wav) return fail(T2_EINVAL, "null argument");
    if (B < 1 || T < 1 || num_freq < 2) return fail(T2_EINVAL, "bad shape [%d,%d,%d]", B, T, num_freq);
    const int n_fft = (num_freq - 1) * 2;                 // util/audio.py:97
    int log_n = 0;
    while ((1 << log_n) < n_fft) ++log_n;
    if ((1 << log_n) != n_fft || n_fft > T2_VOC_MAX_FFT || n_fft < 4)
        return fail(T2_EINVAL, "n_fft = 2 (num_freq - 1) = %d must be a power of two in 4..%d", n_fft, T2_VOC_MAX_FFT);
    if (win_length < 1 || win_length > n_fft || hop_length < 1 || hop_length > win_length)
        return fail(T2_EINVAL, "need 1 <= hop_length <= win_length <= n_fft");
    if (iters < 0) return fail(T2_EINVAL, "iters must be >= 0");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    VocArgs a{};
    a.B = B; a.T = T; a.F = num_freq; a.n_fft = n_fft; a.log_n = log_n; a.win = win_length; a.hop = hop_length;
    a.L = (T - 1) * hop_length + win_length;
    a.min_level_db = min_level_db; a.ref_level_db = ref_level_db; a.power = power;
    const size_t nS = align_up((size_t)B * T * num_freq * 4, 256), nZ = align_up((size_t)B * T * num_freq * 8, 256),
                 nF = align_up((size_t)B * T * win_length * 4, 256);
    T2_TRY(ensure_ws(&ctx->voc_ws, &ctx->voc_ws_bytes, nS + nZ + nF));
    char* p = (char*)ctx->voc_ws;
    a.spectrogram = spectrogram; a.S = (float*)p; a.Z = (float2*)(p + nS); a.frames = (float*)(p + nS + nZ); a.wav = wav;
    const long long n = (long long)B * T * num_freq;
    hipLaunchKernelGGL(voc_amp_kernel, dim3((unsigned)std::min<long long>(4096, (n + 255) / 256)), dim3(256), 0, st, a);
    const size_t lds = (size_t)(n_fft + n_fft / 2) * sizeof(float2);
    const dim3 grid(T, B), blk(T2_VOC_THREADS);
    hipLaunchKernelGGL(voc_istft_kernel, grid, blk, lds, st, a);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(voc_stft_kernel, grid, blk, lds, st, a);
        hipLaunchKernelGGL(voc_istft_kernel, grid, blk, lds, st, a);
    }
    const long long nw = (long long)B * a.L;
    hipLaunchKernelGGL(voc_ola_kernel, dim3((unsigned)std::min<long long>(4096, (nw + 255) / 256)), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return T2_OK;
}
