// Micro-benchmark for attention_step_kernel (not part of the library).
#include "../tacotron2_amd/csrc/kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static float* dalloc(size_t n, float scale) {
    float* p; CK(hipMalloc(&p, n * 4));
    std::vector<float> h(n);
    for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * scale;
    CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    return p;
}

static void run(int B, int T, int iters) {
    const int U = 128, E = 512, F = 32, K = 31, MB = (B + 15) / 16 == 3 ? 4 : (B + 15) / 16, nparts = 256;
    AttnArgs a{};
    a.keys = dalloc((size_t)B * T * U, 1.f); a.values = dalloc((size_t)B * T * E, 1.f);
    int* lens; CK(hipMalloc(&lens, B * 4)); std::vector<int> hl(B, T); CK(hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice));
    a.lens = lens; a.att_state = dalloc((size_t)B * T, 0.f); a.alignments = dalloc((size_t)B * T, 0.f); a.history = nullptr;
    a.qpart = dalloc((size_t)nparts * MB * 16 * U, 0.1f); a.nparts = nparts;
    const int xsz = (E / 16) * MB * 256;
    float* X = dalloc(2 * (size_t)xsz, 0.f);
    a.ctx_dst[0] = FragBuf{X, xsz, 2, 1, 0}; a.ctx_dst[1] = FragBuf{nullptr, 0, 1, 0, 0}; a.ctx_plain = nullptr;
    a.conv_w = dalloc(K * F, 0.5f); a.conv_b = dalloc(F, 0.1f); a.wloc = dalloc(F * U, 0.5f); a.v_a = dalloc(U, 0.5f); a.b_a = dalloc(U, 0.1f);
    a.B = B; a.T = T; a.U = U; a.E = E; a.F = F; a.K = K; a.MB = MB; a.cumulative = 1; a.ctl = nullptr; a.step_off = 0;
    const size_t lds = (size_t)attn_lds_floats(T, U, F, K) * 4;
#ifdef T2_ATTN_STAMPS
    unsigned long long* st; CK(hipMalloc(&st, 16 * 8)); CK(hipMemset(st, 0, 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &st, sizeof(st)));
#endif
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) { a.step_off = i; hipLaunchKernelGGL((attention_step_kernel<32, 2, 4, 0>), dim3(B), dim3(T2_ATTN_THREADS), lds, s, a, PartialJobs{}); }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("B=%d T=%d: %.2f us/launch (incl. boundary)\n", B, T, best * 1e3 / iters);
#ifdef T2_ATTN_STAMPS
    unsigned long long h[16]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    const char* names[] = {"start->early loads issued+LDS fill", "barrier", "conv", "q reduce", "energies", "value loads issue", "softmax", "context"};
    for (int i = 0; i < 8; ++i) printf("   %-38s %8llu cycles\n", names[i], h[i + 1] - h[i]);
    printf("   total %llu cycles\n", h[8] - h[0]);
#endif
}

int main() { run(1, 100, 100); run(64, 140, 100); return 0; }
