// tools/micro/sync_breakdown.hip — where the host-synchronous control step's time goes.
// mppi_next(x) -> u is what a real control loop pays per step (it needs u before it can step the plant): 26.9 us from Python against
// 19.9 us pipelined in r03, with nothing under profiles/ saying what the 7 us are. This program measures, on the box it runs on:
//   null kernel round trips   an EMPTY kernel that stores one word into pinned, device-mapped host memory:
//                             launch call alone | launch -> host sees the word (spin) | launch -> hipStreamSynchronize returns
//                             = enqueue + doorbell + dispatch + PCIe write visibility, with NO work in between
//   the control step          mppi_next_device enqueue alone (empty queue) | pipelined step (200 steps, one synchronisation) |
//                             mppi_next with the host watching the pinned u slot (MPPI_TUNE_SYNC_SPIN 1, default) |
//                             mppi_next waiting on the stream (MPPI_TUNE_SYNC_SPIN 0)
// and prints one JSON object. sync - pipelined is then compared with the null round trip: what is left is the dependent-dispatch
// gap the pipelined loop hides (the next rollout's dispatch overlapping the previous finish).
//   hipcc -O2 -std=c++17 -I include tools/micro/sync_breakdown.hip -o build/sync_breakdown -L mppi-tf_amd -lmppi_hip -Wl,-rpath,$PWD/mppi-tf_amd
//   build/sync_breakdown [K=65536] [H=64] [a=3]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mppi_c.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CKM(x) do { mppi_status s_ = (x); if (s_ != MPPI_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, mppi_status_string(s_), mppi_last_error(h)); return 3; } } while (0)

__global__ void k_flag(volatile unsigned *flag, unsigned v) { *flag = v; }

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double p95(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[(size_t)(0.95 * (v.size() - 1))]; }

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 65536, H = argc > 2 ? atoi(argv[2]) : 64, a = argc > 3 ? atoi(argv[3]) : 3, s = 2 * a;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *h_flag = nullptr, *d_flag = nullptr;
    CK(hipHostMalloc((void **)&h_flag, 64, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    *h_flag = 0;
    const int N = 400;
    std::vector<double> t_launch, t_spin, t_wait;
    for (int i = 0; i < N + 20; ++i) { // launch -> the host sees the store
        const unsigned v = 2 * i + 1;
        auto t0 = clk::now();
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, d_flag, v);
        auto t1 = clk::now();
        while (*(volatile unsigned *)h_flag != v) { }
        auto t2 = clk::now();
        CK(hipStreamSynchronize(st));
        if (i >= 20) { t_launch.push_back(us(t0, t1)); t_spin.push_back(us(t0, t2)); }
    }
    for (int i = 0; i < N + 20; ++i) { // launch -> hipStreamSynchronize returns
        auto t0 = clk::now();
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, d_flag, 2u * i);
        CK(hipStreamSynchronize(st));
        auto t1 = clk::now();
        if (i >= 20) t_wait.push_back(us(t0, t1));
    }

    mppi_config cfg;
    mppi_handle *h = nullptr;
    CKM(mppi_config_init(&cfg, K, H, 0.1f, 1.0f, s, a));
    std::vector<float> sig(a * a, 0.f), goal(s, 0.f);
    for (int i = 0; i < a; ++i) { sig[i * a + i] = 0.25f; goal[2 * i] = 1.0f - 0.25f * i; }
    cfg.sigma = sig.data(); cfg.goal = goal.data();
    CKM(mppi_create(&cfg, &h));
    float *x_dev = nullptr, *u_dev = nullptr;
    CK(hipMalloc((void **)&x_dev, sizeof(float) * s));
    CK(hipMalloc((void **)&u_dev, sizeof(float) * a));
    CK(hipMemset(x_dev, 0, sizeof(float) * s));
    for (int i = 0; i < 50; ++i) CKM(mppi_next_device(h, x_dev, u_dev, st));
    CK(hipStreamSynchronize(st));
    // pipelined
    auto p0 = clk::now();
    for (int i = 0; i < 2000; ++i) CKM(mppi_next_device(h, x_dev, u_dev, st));
    CK(hipStreamSynchronize(st));
    const double pipelined = us(p0, clk::now()) / 2000;
    // enqueue alone: 8 steps into an empty queue, repeated
    std::vector<double> t_enq;
    for (int r = 0; r < 50; ++r) {
        auto t0 = clk::now();
        for (int i = 0; i < 8; ++i) CKM(mppi_next_device(h, x_dev, u_dev, st));
        t_enq.push_back(us(t0, clk::now()) / 8);
        CK(hipStreamSynchronize(st));
    }
    // host-synchronous steps, plant on the host
    std::vector<float> x(s, 0.f), u(a, 0.f);
    auto closed_loop = [&](std::vector<double> &ts) -> int {
        for (int i = 0; i < N + 20; ++i) {
            auto t0 = clk::now();
            CKM(mppi_next(h, x.data(), s, u.data(), a));
            auto t1 = clk::now();
            if (i >= 20) ts.push_back(us(t0, t1));
            for (int j = 0; j < a; ++j) { x[2 * j] += 0.1f * x[2 * j + 1] + 0.005f * u[j]; x[2 * j + 1] += 0.1f * u[j]; }
        }
        return 0;
    };
    std::vector<double> t_sync_spin, t_sync_wait;
    if (int rc = closed_loop(t_sync_spin)) return rc;
    CKM(mppi_set_tuning(h, MPPI_TUNE_SYNC_SPIN, 0));
    if (int rc = closed_loop(t_sync_wait)) return rc;
    float roll_ms = 0, fin_ms = 0; int n = 0;
    CKM(mppi_set_tuning(h, MPPI_TUNE_SYNC_SPIN, 1));
    CKM(mppi_profile_begin(h, 200));
    for (int i = 0; i < 200; ++i) CKM(mppi_next_device(h, x_dev, u_dev, st));
    CKM(mppi_profile_end(h, &roll_ms, &fin_ms, &n));
    printf("{\"K\": %d, \"H\": %d, \"a\": %d, "
           "\"null_kernel_us\": {\"launch_call\": %.2f, \"launch_to_host_sees_pinned_store\": %.2f, \"launch_to_stream_synchronize_returns\": %.2f}, "
           "\"step_us\": {\"enqueue_call_mppi_next_device\": %.2f, \"pipelined\": %.2f, \"rollout_kernel\": %.2f, \"finish_kernel\": %.2f, "
           "\"sync_spin_median\": %.2f, \"sync_spin_p95\": %.2f, \"sync_wait_median\": %.2f, \"sync_wait_p95\": %.2f}}\n",
           K, H, a, median(t_launch), median(t_spin), median(t_wait), median(t_enq), pipelined, 1e3 * roll_ms, 1e3 * fin_ms,
           median(t_sync_spin), p95(t_sync_spin), median(t_sync_wait), p95(t_sync_wait));
    mppi_destroy(h);
    return 0;
}
