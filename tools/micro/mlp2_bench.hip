// Standalone timing harness for k_rollout_mlp2<3, true, SRC_PHILOX> at BASELINE configs[3] (K=65536, H=64): one TU, one
// instantiation, so that timing-only ablations (-DMPPI_MLP2_ABL=bits, see mppi_mlp2.hip.h) build in seconds.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include [-DMPPI_MLP2_ABL=n] [-DMPPI_MLP2_STAMP]
//         tools/micro/mlp2_bench.hip -o build/mlp2_bench[_n]
// Prints one line: kernel ms, TFLOP/s; with -DMPPI_MLP2_STAMP also the in-kernel clock (s_memtime / s_memrealtime) and
// the median cycles per workgroup.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../mppi-tf_amd/csrc/mppi_kernels.hip.h"
using namespace mppi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    constexpr int A = 3, S = 6;
    const int K = argc > 1 ? atoi(argv[1]) : 65536, H = argc > 2 ? atoi(argv[2]) : 64, reps = argc > 3 ? atoi(argv[3]) : 5;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> un(-1.f, 1.f);
    auto dev = [&](size_t n, float scale, float **p) {
        std::vector<float> h(n);
        for (auto &v : h) v = un(rng) * scale;
        if (hipMalloc((void **)p, n * 4) != hipSuccess) return false;
        return hipMemcpy(*p, h.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess;
    };
    float *W1, *b1, *W2, *b2, *W3, *b3, *x, *U, *cost, *part;
    if (!dev(9 * 256, 1 / 3.f, &W1) || !dev(256, 1 / 3.f, &b1) || !dev(256 * 256, 1 / 16.f, &W2) || !dev(256, 1 / 16.f, &b2) ||
        !dev(256 * S, 0.1f / 16.f, &W3) || !dev(S, 0.1f / 16.f, &b3) || !dev(S, 0.1f, &x) || !dev((H + 1) * A, 0.1f, &U)) return 1;
    const int nb = (K + kMlp2R - 1) / kMlp2R;
    const int grid = argc > 4 ? atoi(argv[4]) : std::min(nb, 256); // one workgroup per CU walks over the tiles
    CK(hipMalloc((void **)&cost, K * 4));
    CK(hipMalloc((void **)&part, (size_t)nb * (2 + H * A) * 4));
    MlpDev m{};
    m.W1 = W1; m.b1 = b1; m.W2 = W2; m.b2 = b2; m.W3 = W3; m.b3 = b3;
    for (int i = 0; i < S + A; ++i) { m.xmean[i] = 0.01f * i; m.xstd[i] = 1.0f + 0.1f * i; }
    for (int i = 0; i < S; ++i) { m.ymean[i] = 0.0f; m.ystd[i] = 1.0f; }
    DevConsts c{};
    c.K_local = K; c.H = H; c.s = S; c.a = A; c.model_kind = 1; c.lambda = 1.f; c.neg_inv_lambda = -1.f; c.gamma = 1.f; c.upsilon = 1.f;
    c.dt = 0.1f; c.seed = 1;
    for (int i = 0; i < S; ++i) { c.goal[i] = 0.5f; c.qdiag[i] = 1.f; }
    for (int i = 0; i < A; ++i) { c.sigma[i * kMaxA + i] = 0.25f; c.sigma_inv[i * kMaxA + i] = 4.f; }
    DevConsts *dC; MlpDev *dM; unsigned long long *step;
    CK(hipMalloc((void **)&dC, sizeof(c))); CK(hipMemcpy(dC, &c, sizeof(c), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&dM, sizeof(m))); CK(hipMemcpy(dM, &m, sizeof(m), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&step, 8)); CK(hipMemset(step, 0, 8));
    auto kern = k_rollout_mlp2<A, true, SRC_PHILOX>;
    const size_t lds = mlp2_lds_floats(S, A, H) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps + 2; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kMlp2Threads), lds, 0, dC, dM, x, U, (const float *)nullptr, step, cost, part, (int)MODE_ROLLOUT, 1, nb);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { best = std::min(best, ms); sum += ms; }
    }
    const double fl = 2.0 * (9 * 256 + 256 * 256 + 256 * 6) * (double)K * H;
    std::printf("abl=%d K=%d H=%d: kernel %.3f ms avg, %.3f best -> %.1f TFLOP/s (avg)", MPPI_MLP2_ABL, K, H, sum / reps, best, fl / (sum / reps * 1e-3) / 1e12);
#ifdef MPPI_MLP2_STAMP
    std::vector<unsigned long long> st(4 * 4096);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_mlp2_stamp), st.size() * 8));
    std::vector<double> cyc, clk;
    for (int b = 0; b < std::min(grid, 4096); ++b) {
        const double dc = (double)(st[4 * b + 1] - st[4 * b + 0]), dr = (double)(st[4 * b + 3] - st[4 * b + 2]);
        cyc.push_back(dc); clk.push_back(dc / dr * 100.0);
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    std::printf(" | workgroup cycles median %.0f (%.1f per tile-step), clock median %.0f MHz", cyc[cyc.size() / 2], cyc[cyc.size() / 2] / H, clk[clk.size() / 2]);
#endif
    std::printf("\n");
#ifdef MPPI_MLP2_TRACE
    {
        std::vector<unsigned long long> tr(4 * 32);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_mlp2_trace), tr.size() * 8));
        for (int w = 0; w < 4; ++w) {
            std::printf("wave %d: ", w);
            for (int i = 1; i < kMlp2TraceN; ++i)
                std::printf("kp%d..%d:%llu(%.0f/kp) ", kMlp2TraceKp[i - 1], kMlp2TraceKp[i], tr[w * 32 + i] - tr[w * 32 + i - 1],
                            (double)(tr[w * 32 + i] - tr[w * 32 + i - 1]) / (kMlp2TraceKp[i] - kMlp2TraceKp[i - 1]));
            std::printf("\n");
        }
    }
#endif
    return 0;
}
