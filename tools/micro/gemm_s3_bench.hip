// micro-benchmark + check of wino_gemm_s3.hip (the split-bf16 batched GEMM) outside the library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I object_tracking_amd/csrc tools/micro/gemm_s3_bench.hip \
//         object_tracking_amd/csrc/wino_gemm_s3.hip -o tools/micro/gemm_s3_bench && tools/micro/gemm_s3_bench
// Prints, per shape, the error against a float64 product of the fp32 inputs (also for a plain fp32 fmaf chain, as the
// yardstick) and the executed-bf16 / fp32-equivalent TFLOP/s.
#pragma clang diagnostic ignored "-Wunused-value"
#include "dt_internal.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <random>
#include <cstring>
#include <algorithm>

int dt_fail(dt_ctx *, int rc, const char *, ...) { return rc; }

static void run(int P, int Mt, int K, int N, int check_rows, int iters)
{
    const int Mp = (Mt + 255) / 256 * 256, Np = (N + 255) / 256 * 256, KB = K / 16;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    const int Pc = 2;      // planes with real data for the check (the rest reuse them through the same buffers)
    std::vector<float> V((size_t)Pc * Mt * K), U((size_t)Pc * N * K);
    for (auto &x : V) x = nd(rng) * std::exp(nd(rng));
    for (auto &x : U) x = nd(rng) * 0.05f;
    // S3_NT=2: the fp16 form (two terms, scaled operands).  S3_SMALLROWS=k: every 5th row of V is 2^-k of the others (its lo terms are
    // fp16 subnormals from k ~ 18: the probe of whether the MFMA keeps them)
    const int NT = getenv("S3_NT") ? atoi(getenv("S3_NT")) : 3;
    const int smallk = getenv("S3_SMALLROWS") ? atoi(getenv("S3_SMALLROWS")) : 0;
    if (smallk)
        for (int p = 0; p < Pc; ++p)
            for (int m = 0; m < Mt; m += 5)
                for (int k = 0; k < K; ++k) V[((size_t)p * Mt + m) * K + k] = std::ldexp(V[((size_t)p * Mt + m) * K + k], -smallk);
    auto bits = [](float x) { unsigned u; memcpy(&u, &x, 4); return u & 0x7fffffffu; };
    float vmax = 0;
    for (auto &x : V) vmax = std::max(vmax, std::fabs(x));
    const float vbase = dt_h2_base(bits(vmax));
    std::vector<float> uscale(P), pscale(P);
    for (int p = 0; p < P; ++p) {
        float um = 0;
        for (size_t i = 0; i < (size_t)N * K; ++i) um = std::max(um, std::fabs(U[(size_t)(p % Pc) * N * K + i]));
        uscale[p] = dt_h2_base(bits(um));
        pscale[p] = 1.0f / uscale[p];
    }
    std::vector<unsigned short> Vs((size_t)P * NT * KB * Mp * 16, 0), Us((size_t)P * NT * KB * Np * 16, 0);
    for (int p = 0; p < P; ++p)
        for (int m = 0; m < Mt; ++m)
            for (int k = 0; k < K; ++k) {
                unsigned short t[3];
                const float x = V[((size_t)(p % Pc) * Mt + m) * K + k];
                if (NT == 2) wino_h2_split_host(x * vbase, t); else wino_s3_split_host(x, t);
                for (int t3 = 0; t3 < NT; ++t3) Vs[((((size_t)p * NT + t3) * KB + (k >> 4)) * Mp + m) * 16 + (k & 15)] = t[t3];
            }
    {
        std::vector<float> Upad((size_t)P * Np * K, 0.0f);
        for (int p = 0; p < P; ++p)
            for (int n = 0; n < N; ++n)
                for (int k = 0; k < K; ++k) Upad[((size_t)p * Np + n) * K + k] = U[((size_t)(p % Pc) * N + n) * K + k];
        if (NT == 2) wino_h2_pack_weights(Upad.data(), P, Np, K, uscale.data(), Us.data());
        else wino_s3_pack_weights(Upad.data(), P, Np, K, Us.data());
    }
    float *dPs = nullptr; unsigned *dAm = nullptr;
    {
        unsigned am[DT_AMAX_SUB] = {};
        am[3] = bits(vmax);      // (any sub-slot)
        hipMalloc(&dPs, P * 4); hipMalloc(&dAm, sizeof(am));
        hipMemcpy(dPs, pscale.data(), P * 4, hipMemcpyHostToDevice); hipMemcpy(dAm, am, sizeof(am), hipMemcpyHostToDevice);
    }
    unsigned short *dV, *dU;
    float *dC;
    hipMalloc(&dV, Vs.size() * 2); hipMalloc(&dU, Us.size() * 2); hipMalloc(&dC, (size_t)P * Mt * N * 4);
    hipMemcpy(dV, Vs.data(), Vs.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dU, Us.data(), Us.size() * 2, hipMemcpyHostToDevice);
    hipMemset(dC, 0xff, (size_t)P * Mt * N * 4);
    GemmS3Args a = {};
    a.a = dV; a.b = dU; a.c = dC; a.c_ps = (long long)Mt * N; a.P = P; a.Mt = Mt; a.Mp = Mp; a.N = N; a.Np = Np; a.K = K; a.ldc = N;
    a.dbg = nullptr;
    if (NT == 2) { a.nt = 2; a.pscale = dPs; a.amax = dAm; }
    a.waves = getenv("S3_WAVES") ? atoi(getenv("S3_WAVES")) : 0;
    a.half = getenv("S3_HALF") ? atoi(getenv("S3_HALF")) : 0;
    a.act = getenv("S3_ACT") ? 1 : 0; a.slope = 0.1f;      // timing only (the check below expects the plain product)
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
#ifdef S3_TIMING
    hipMalloc(&a.dbg, 256 * 8 * 5 * 8);
    hipMemset(a.dbg, 0, 256 * 8 * 5 * 8);
#endif
    int rc = launch_wino_gemm_s3(0, a, prop.multiProcessorCount);
    if (rc || hipDeviceSynchronize() != hipSuccess) { printf("launch failed rc=%d %s\n", rc, hipGetErrorString(hipGetLastError())); exit(1); }
    std::vector<float> C((size_t)P * Mt * N);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    // check: planes 0, 1 and the last; rows spread over the tile rows
    double e_s3 = 0, e_f32 = 0, ref_rms = 0; long long cnt = 0; double worst = 0;
    double e_small = 0; long long cnt_small = 0;
    const int planes[3] = {0, 1 % P, P - 1};
    for (int pi = 0; pi < 3; ++pi) {
        const int p = planes[pi];
        for (int r = 0; r < check_rows; ++r) {
            const int m = (int)(((long long)r * 2654435761u) % Mt);
            for (int n = 0; n < N; n += 7) {
                const float *v = &V[((size_t)(p % Pc) * Mt + m) * K], *u = &U[((size_t)(p % Pc) * N + n) * K];
                double ref = 0; float f = 0; double mag = 0;
                for (int k = 0; k < K; ++k) { ref += (double)v[k] * (double)u[k]; f = fmaf(v[k], u[k], f); mag += fabs((double)v[k] * u[k]); }
                const double got = C[((size_t)p * Mt + m) * N + n];
                const double d = fabs(got - ref) / mag, d32 = fabs((double)f - ref) / mag;
                if (smallk && m % 5 == 0) { e_small += d * d; ++cnt_small; continue; }
                e_s3 += d * d; e_f32 += d32 * d32; ref_rms += 1; ++cnt;
                if (d > worst) worst = d;
            }
        }
    }
    const char *fill = getenv("S3_FILL");      // timing-only probes of the clock: "zero" = all-zero operands, "const" = every element 1.0 (no
    if (fill && fill[0]) {                      // bit toggles between successive operands): what the same instruction stream draws without data activity
        const int byte = fill[0] == 'z' ? 0x00 : 0x3f;      // 0x3f3f = bf16 0.746: constant, non-zero
        hipMemset(dV, byte, Vs.size() * 2); hipMemset(dU, byte, Us.size() * 2);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_wino_gemm_s3(0, a, prop.multiProcessorCount);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch_wino_gemm_s3(0, a, prop.multiProcessorCount);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    const double eq = 2.0 * P * (double)Mt * K * N / (ms * 1e-3) / 1e12;
    printf("NT=%d P=%d Mt=%d K=%d N=%d: %.3f ms  %.1f TF/s fp32-equivalent (%.0f executed 16-bit)  rel.err/|u||v|: split rms %.3g max %.3g, fp32 fmaf rms %.3g (%lld samples)",
           NT, P, Mt, K, N, ms, eq, eq * (NT == 2 ? 3 : 6), sqrt(e_s3 / cnt), worst, sqrt(e_f32 / cnt), cnt);
    if (cnt_small) printf("  rows 2^-%d: rms %.3g (%lld)", smallk, sqrt(e_small / cnt_small), cnt_small);
    printf("\n");
#ifdef S3_TIMING
    {
        std::vector<unsigned long long> h(256 * 8 * 5);
        hipMemcpy(h.data(), a.dbg, h.size() * 8, hipMemcpyDeviceToHost);
        double s[5] = {0, 0, 0, 0, 0}; int n = 0;
        for (int w = 0; w < 256 * 8; ++w) if (h[w * 5 + 3]) { for (int k = 0; k < 5; ++k) s[k] += (double)h[w * 5 + k]; ++n; }
        if (n) printf("   timing (mean per wave over %d waves, clock64 ticks): total %.0f; per stage: lgkm wait %.1f, DMA wait %.1f, barrier %.1f, whole stage %.1f (%0.f stages)\n",
                      n, s[4] / n, s[0] / s[3], s[1] / s[3], s[2] / s[3], s[4] / s[3], s[3] / n);
        hipFree(a.dbg);
    }
#endif
    hipFree(dV); hipFree(dU); hipFree(dC); hipFree(dPs); hipFree(dAm);
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == '2') { run(64, 7840, 1024, 1024, 1, 5); return 0; }      // PMC runs: one shape
    if (argc > 1 && argv[1][0] == '3') { run(64, 30240, 256, 512, 8, 5); run(36, 588, 512, 2048, 32, 20); run(64, 7840, 1024, 1024, 4, 5); return 0; }   // half-tile A/B
    if (argc > 1) { run(64, 7840, 1024, 1024, 1, 5); run(64, 30240, 256, 512, 1, 5); run(64, 116640, 128, 256, 1, 3); return 0; }   // probes: timing only
    run(4, 300, 64, 256, 64, 3);          // ragged Mt, short K
    run(3, 700, 128, 128, 64, 3);         // BN = 128
    run(64, 7840, 1024, 1024, 16, 5);     // conv_19 / conv_20 at 1440 frames (mosaic g = 3)
    run(64, 7840, 512, 1024, 8, 5);       // conv_14 / 16 / 18
    run(64, 30240, 256, 512, 8, 5);       // conv_9 / 11 / 13   (26x26: 21 tiles per frame)
    run(64, 116640, 128, 256, 4, 3);      // conv_6 / conv_8   (52x52: 81 tiles per frame)
    run(36, 588, 512, 2048, 32, 20);      // the ConvLSTM recurrent step at 48 clips (F(4x4): 36 positions)
    return 0;
}
