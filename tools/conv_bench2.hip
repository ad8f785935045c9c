// Experiment bench of the conv GEMM forward fast path (not part of the library): variants of conv_gemm_kt_kernel (kernels.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench2.hip -o tools/bin/conv_bench2
#include "../tacotron2_amd/csrc/kernels.h"
template <int NS, int WN, int ORDER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WN == 4 ? 3 : 2, WN == 4 ? 3 : 4)))
void conv_kt_exp_kernel(ConvArgs a) {
    static_assert(WN == 2 || WN == 4, "waves across the 128 columns");
    constexpr int WM = 4 / WN;                       // waves down the rows
    constexpr int NJ = 4 / WN;                       // 32-column sub-tiles per wave (WN = 2: 2, WN = 4: 1)
    constexpr int BM = 32 * NS * WM;
    constexpr int NP = (BM + 63) / 64;               // loader passes over the rows of the A tile
    __shared__ __attribute__((aligned(16))) float As[2][BM][CG_BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][CG_BN][CG_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * CG_BN;
    const int Ktot = a.ktaps * a.Cin;
    const int nk = Ktot / CG_BK;
    f32x16 acc[NS][NJ];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kq = tid & 3, r0 = tid >> 2;           // loader: 16-byte chunk kq of tile rows r0, r0 + 64, ...
    int tt[NP];
    const float* xrow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int m = m0 + r0 + p * 64;
        tt[p] = (m < a.Mtot && r0 + p * 64 < BM) ? m % a.T : -(1 << 28);
        xrow[p] = a.x + (size_t)min(m, a.Mtot - 1) * a.ldx + kq * 4;
    }
    const float* wrow[2];
    bool w_ok[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int n = n0 + r0 + p * 64;
        w_ok[p] = n < a.N;
        wrow[p] = a.wt + (size_t)min(n, a.N - 1) * Ktot + kq * 4;
    }
    float4 ra[NP], rb[2];
    int tap = 0, ci = 0;                             // k-tile kt covers channels [ci, ci + 16) of tap `tap`
    auto load_tiles = [&](int kt) {
        const int dt = tap - a.pad_l;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int ts = tt[p] + dt;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ts >= 0 && ts < a.T) v = *reinterpret_cast<const float4*>(xrow[p] + (long long)dt * a.ldx + ci);
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w_ok[p]) v = *reinterpret_cast<const float4*>(wrow[p] + kt * CG_BK);
            rb[p] = v;
        }
        ci += CG_BK;
        if (ci >= a.Cin) { ci = 0; ++tap; }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int r = r0 + p * 64;
            if (BM % 64 == 0 || r < BM) *reinterpret_cast<float4*>(&As[buf][r][(kq ^ ((r >> 1) & 3)) * 4]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = r0 + p * 64;
            *reinterpret_cast<float4*>(&Bs[buf][r][(kq ^ ((r >> 1) & 3)) * 4]) = rb[p];
        }
    };
    load_tiles(0);
    store_tiles(0);
    if (ORDER == 1 && nk > 1) load_tiles(1);
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (ORDER == 0) { if (kt + 1 < nk) load_tiles(kt + 1); }
        else {
            if (kt + 1 < nk) store_tiles(buf ^ 1);
            if (kt + 2 < nk) load_tiles(kt + 2);
        }
        f32x4 av[NS][2], bv[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = wn * 32 * NJ + j * 32 + l31;
#pragma unroll
            for (int h = 0; h < 2; ++h) bv[j][h] = *reinterpret_cast<const f32x4*>(&Bs[buf][r][((lh * 2 + h) ^ ((r >> 1) & 3)) * 4]);
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int r = wm * 32 * NS + i * 32 + l31;
#pragma unroll
            for (int h = 0; h < 2; ++h) av[i][h] = *reinterpret_cast<const f32x4*>(&As[buf][r][((lh * 2 + h) ^ ((r >> 1) & 3)) * 4]);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NS; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s >> 2][s & 3], bv[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        if (ORDER == 0 && kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * 32 * NJ + j * 32 + l31;
        if (n >= a.N) continue;
        const float sc = a.scale ? a.scale[n] : 1.f;
        const float sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * NS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.Mtot) continue;
                float v = acc[i][j][r] * sc + sh;
                if (a.act == 1) v = fmaxf(v, 0.f);
                else if (a.act == 2) v = tanhf(v);
                if (a.res) v += a.res[(size_t)m * a.ldr + n];
                a.y[(size_t)m * a.ldy + n] = v;
            }
    }
}


#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename F> static float timeit(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 10; ++it) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 1 && ms < best) best = ms;
    }
    return best;
}
int main(int argc, char** argv) {
    const int B = argc > 2 ? atoi(argv[2]) : 64, T = 800, Cin = argc > 1 ? atoi(argv[1]) : 512, N = 512, ktaps = 5, Mtot = B * T, K = ktaps * Cin;
    float *x, *w, *wT, *sc, *sh, *y, *y2;
    CK(hipMalloc(&x, (size_t)Mtot * Cin * 4)); CK(hipMalloc(&w, (size_t)K * N * 4)); CK(hipMalloc(&wT, (size_t)K * N * 4)); CK(hipMalloc(&sc, N * 4)); CK(hipMalloc(&sh, N * 4));
    CK(hipMalloc(&y, (size_t)Mtot * N * 4)); CK(hipMalloc(&y2, (size_t)Mtot * N * 4));
    std::vector<float> hx((size_t)Mtot * Cin), hw((size_t)K * N), hwT((size_t)K * N), hs(N, 1.f);
    for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
    for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) hwT[(size_t)n * K + k] = hw[(size_t)k * N + n];
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wT, hwT.data(), hwT.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hs.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(sh, 0, N * 4));
    ConvArgs a{};
    a.x = x; a.ldx = Cin; a.w = w; a.wt = wT; a.scale = sc; a.shift = sh; a.y = y; a.ldy = N; a.Mtot = Mtot; a.T = T; a.Cin = Cin; a.N = N; a.ktaps = ktaps;
    a.pad_l = 2; a.act = 2; a.vec_a = 1; a.vec_b = 1;
    ConvArgs r = a; r.y = y2;
    const dim3 g128(N / 128, (Mtot + 127) / 128), g160(N / 128, (Mtot + 159) / 160);
    const double fl = 2.0 * Mtot * (double)K * N;
    hipLaunchKernelGGL(conv_gemm_fast_kernel<2>, g128, dim3(256), 0, 0, r);
    float t;
    t = timeit([&] { hipLaunchKernelGGL((conv_kt_exp_kernel<2, 2, 0>), g128, dim3(256), 0, 0, a); }); printf("128 rows order 0: %.3f ms (%.3f)\n", t, fl / t / 1e9 / 157.3);
    t = timeit([&] { hipLaunchKernelGGL((conv_kt_exp_kernel<2, 2, 1>), g128, dim3(256), 0, 0, a); }); printf("128 rows order 1: %.3f ms (%.3f)\n", t, fl / t / 1e9 / 157.3);
    t = timeit([&] { hipLaunchKernelGGL((conv_kt_exp_kernel<5, 4, 0>), g160, dim3(256), 0, 0, a); }); printf("160 rows order 0: %.3f ms (%.3f)\n", t, fl / t / 1e9 / 157.3);
    t = timeit([&] { hipLaunchKernelGGL((conv_kt_exp_kernel<5, 4, 1>), g160, dim3(256), 0, 0, a); }); printf("160 rows order 1: %.3f ms (%.3f)\n", t, fl / t / 1e9 / 157.3);
    CK(hipGetLastError());
    std::vector<float> h1((size_t)Mtot * N), h2((size_t)Mtot * N);
    CK(hipMemcpy(h1.data(), y, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), y2, h2.size() * 4, hipMemcpyDeviceToHost));
    double md = 0; for (size_t i = 0; i < h1.size(); ++i) md = fmax(md, fabs((double)h1[i] - h2[i]));
    printf("max |diff| against conv_gemm_fast_kernel: %.2e\n", md);
    return md < 1e-4 ? 0 : 2;
}
