// Micro-benchmark / ablation harness for lstm_step_kernel (not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DT2_ABL_NOA] [-DT2_ABL_NOW] [-DT2_ABL_NOMMA] tools/lstm_bench.hip -o /tmp/lstm_bench
#include "../tacotron2_amd/csrc/kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MB, int NPW = 0>
static void run(int K, int H, int B, int iters, int rotate) {
    const int kbtot = K / 16, n_wg = H / 4;
    float *w, *bias, *X, *c;
    int* ctl;
    const size_t wsz = (size_t)n_wg * kbtot * 256, xsz = (size_t)kbtot * MB * 256;
    const int nrot = rotate ? 3 : 1;   // rotate between weight copies so that they cannot stay cache resident
    CK(hipMalloc(&w, wsz * 4 * nrot)); CK(hipMalloc(&bias, n_wg * 16 * 4)); CK(hipMalloc(&X, 2 * xsz * 4));
    CK(hipMalloc(&c, (size_t)MB * 16 * H * 4)); CK(hipMalloc(&ctl, 32));
    std::vector<float> hw(wsz);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (int r = 0; r < nrot; ++r) CK(hipMemcpy(w + r * wsz, hw.data(), wsz * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(2 * xsz);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(X, hx.data(), 2 * xsz * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, n_wg * 16 * 4)); CK(hipMemset(c, 0, (size_t)MB * 16 * H * 4));
    int hctl[8] = {0, 0, 1 << 30, 0, 0, 0, 0, 0};
    CK(hipMemcpy(ctl, hctl, 32, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, tot = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) {
            SkinnyCore a{}; LstmEpi e{};
            a.w[0] = w + (size_t)(i % nrot) * wsz; a.bias[0] = bias; a.kb0 = 0; a.kbtot = kbtot; a.kb_w0 = 0; a.kb_full = kbtot; a.ctl = ctl; a.step_off = i; a.t_hint = i; a.B = B;
            a.seg[0][1] = FragBuf{X, (int)xsz, 2, 0, 0};
            e.H = H; e.c[0] = c; e.Hld = H;
            const int hoff = K - H;
            e.hprev[0] = FragBuf{X, (int)xsz, 2, 0, hoff}; e.hstate[0] = FragBuf{X, (int)xsz, 2, 1, hoff};
            e.hraw[0] = FragBuf{nullptr, 0, 1, 0, 0};
            e.zc = 0.1f; e.zh = 0.1f; e.fb = 1.f;
            hipLaunchKernelGGL((lstm_step_kernel<MB, false, NPW>), dim3(n_wg, 1), dim3(T2_SKINNY_THREADS), 0, st, a, e);
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) { tot += ms; if (ms < best) best = ms; }
    }
    const double us = best * 1e3 / iters;
    const double flops = 2.0 * B * K * 4.0 * H, wbytes = (double)K * 4 * H * 4;
    printf("NPW=%d MB=%d B=%d K=%d H=%d rot=%d: %.2f us/launch (incl. boundary)  %.1f TFLOP/s  W-stream %.2f TB/s\n", NPW, MB, B, K, H, rotate, us,
           flops / us / 1e6, wbytes / us / 1e6);
    hipFree(w); hipFree(bias); hipFree(X); hipFree(c); hipFree(ctl);
}

template <int MB, int NPW>
static void run_splitk(int K, int H, int B, int iters) {
    const int kbtot = K / 16, n_wg = H / 4;
    float *w, *bias, *X, *c; int* ctl; unsigned* tickets; unsigned long long* slabs;
    const size_t wsz = (size_t)n_wg * kbtot * 256, xsz = (size_t)kbtot * MB * 256;
    CK(hipMalloc(&w, wsz * 4 * 3)); CK(hipMalloc(&bias, n_wg * 16 * 4)); CK(hipMalloc(&X, 2 * xsz * 4));
    CK(hipMalloc(&c, (size_t)MB * 16 * H * 4)); CK(hipMalloc(&ctl, 32));
    CK(hipMalloc(&tickets, n_wg * 4)); CK(hipMemset(tickets, 0, n_wg * 4)); CK(hipMalloc(&slabs, (size_t)n_wg * 512 * 16));
    std::vector<float> hw(wsz);
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (int r = 0; r < 3; ++r) CK(hipMemcpy(w + r * wsz, hw.data(), wsz * 4, hipMemcpyHostToDevice));
    std::vector<float> hx(2 * xsz);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMemcpy(X, hx.data(), 2 * xsz * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, n_wg * 16 * 4)); CK(hipMemset(c, 0, (size_t)MB * 16 * H * 4));
    int hctl[8] = {0, 0, 1 << 30, 0, 0, 0, 0, 0};
    CK(hipMemcpy(ctl, hctl, 32, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) {
            SkinnyCore a{}; LstmEpi e{};
            a.w[0] = w + (size_t)(i % 3) * wsz; a.bias[0] = bias; a.kb0 = 0; a.kbtot = kbtot; a.kb_w0 = 0; a.kb_full = kbtot; a.ctl = ctl; a.step_off = i; a.t_hint = i; a.B = B;
            a.seg[0][1] = FragBuf{X, (int)xsz, 2, 0, 0};
            e.H = H; e.c[0] = c; e.Hld = H;
            const int hoff = K - H;
            e.hprev[0] = FragBuf{X, (int)xsz, 2, 0, hoff}; e.hstate[0] = FragBuf{X, (int)xsz, 2, 1, hoff};
            e.hraw[0] = FragBuf{nullptr, 0, 1, 0, 0};
            e.zc = 0.1f; e.zh = 0.1f; e.fb = 1.f;
            hipLaunchKernelGGL((lstm_step_splitk_kernel<MB, NPW>), dim3(n_wg), dim3(T2_SKINNY_THREADS), 0, st, a, e, n_wg, tickets, slabs);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
    }
    const double us = best * 1e3 / iters;
    printf("splitk NPW=%d MB=%d B=%d K=%d: %.2f us/launch  %.1f TFLOP/s\n", NPW, MB, B, K, us, 2.0 * B * K * 4.0 * H / us / 1e6);
}

int main(int argc, char** argv) {
#if T2_SKINNY_WAVES == 8
    if (argc <= 1) run_splitk<4, 0>(2560, 1024, 64, 200);
    if (argc <= 1) run_splitk<4, 5>(2560, 1024, 64, 200);
#endif
    if (argc > 1) {      // short list for A/B builds
        run<4, 160 / T2_G4 / T2_SKINNY_WAVES>(2560, 1024, 64, 200, 1);
        run<4, 96 / T2_G4 / T2_SKINNY_WAVES>(1536, 1024, 64, 200, 1);
        run<4, 64 / T2_G4 / T2_SKINNY_WAVES>(1024, 1024, 64, 200, 1);
        run<1, 40 / T2_SKINNY_WAVES>(2560, 1024, 1, 200, 1);
        return 0;
    }
    run<4>(2560, 1024, 64, 200, 1);
    run<4, 160 / T2_G4 / T2_SKINNY_WAVES>(2560, 1024, 64, 200, 1);
    run<4, 128 / T2_G4 / T2_SKINNY_WAVES>(2048, 1024, 64, 200, 1);
    run<4, 112 / T2_G4 / T2_SKINNY_WAVES>(1792, 1024, 64, 200, 1);
    run<1, 40 / T2_SKINNY_WAVES>(2560, 1024, 1, 200, 1);
    run<4>(2048, 1024, 64, 200, 1);
    run<4>(1792, 1024, 64, 200, 1);
    run<4>(2560, 1024, 64, 200, 0);
    run<2>(2560, 1024, 32, 200, 1);
    run<1>(2560, 1024, 16, 200, 1);
    run<1>(2560, 1024, 1, 200, 1);
    run<1>(2560, 1024, 1, 200, 0);
    return 0;
}
