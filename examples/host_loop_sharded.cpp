// examples/host_loop_sharded.cpp — the K-sharded control step from a native C++ host, RCCL called directly:
// one mppi handle per GPU (shard_rank/shard_count), per control step
//     mppi_shard_partial (rollouts + local soft-min -> one 2+tau*a float record)
//     ncclAllGather of the records (the ONE collective of the step)
//     mppi_shard_finish  (fixed-order combine, U' = U + V/eta, shift; replicated on every GPU)
// all stream-ordered, no host synchronisation inside the step. Uses every visible GPU (1 on a one-GPU box).
// This is the single-process flavour (ncclCommInitAll); bench.py uses one process per GPU through torch.distributed.
// With a 5th argument "p2p" the exchange is the direct one of include/mppi_c.h instead (mppi_shard_p2p_*: the finish
// kernel stores the record into every GPU's inbox and spins for the others; peer access inside this one process).
// With "step" the three calls become ONE: mppi_shard_step(h, x, u, &coll, stream) with coll = {ncclAllGather, ncclAllReduce, comm} —
// the library calls RCCL's own entry points between its kernels (it links no collective library itself). One host thread per
// GPU then (a single thread driving several communicators would need ncclGroupStart/End around calls that also enqueue kernels).
// Prints the pipelined time per step and the HOST time of one step's enqueue (what a Python host pays several times over).
//   usage: host_loop_sharded [k_per_gpu=65536] [tau=64] [a_dim=3] [steps=100] [rccl|p2p|step]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "mppi_c.h"

#define CK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 3; } } while (0)
#define CK_MPPI(x, h) do { mppi_status s_ = (x); if (s_ != MPPI_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, mppi_status_string(s_), mppi_last_error(h)); return 4; } } while (0)

int main(int argc, char **argv)
{
    const int kper = argc > 1 ? atoi(argv[1]) : 65536, tau = argc > 2 ? atoi(argv[2]) : 64;
    const int a = argc > 3 ? atoi(argv[3]) : 3, steps = argc > 4 ? atoi(argv[4]) : 100, s = 2 * a;
    const bool p2p = argc > 5 && std::string(argv[5]) == "p2p";
    const bool one_call = argc > 5 && std::string(argv[5]) == "step";
    int ndev = 0;
    CK_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
    std::vector<int> devs(ndev);
    for (int d = 0; d < ndev; ++d) devs[d] = d;
    std::vector<ncclComm_t> comm(ndev);
    CK_NCCL(ncclCommInitAll(comm.data(), ndev, devs.data()));

    const float sigma[16] = {0.25f, 0, 0, 0, 0, 0.25f, 0, 0, 0, 0, 0.25f, 0, 0, 0, 0, 0.25f};
    std::vector<float> sig(a * a, 0.f), goal(s, 0.f);
    for (int i = 0; i < a; ++i) { sig[i * a + i] = sigma[0]; goal[2 * i] = 1.0f - 0.25f * i; }
    std::vector<mppi_handle *> h(ndev, nullptr);
    std::vector<hipStream_t> st(ndev);
    std::vector<float *> x_dev(ndev), u_dev(ndev), rec(ndev), recs(ndev);
    int nrec = 0;
    for (int d = 0; d < ndev; ++d) {
        mppi_config cfg;
        CK_MPPI(mppi_config_init(&cfg, kper * ndev, tau, 0.1f, 1.0f, s, a), nullptr);
        cfg.sigma = sig.data(); cfg.goal = goal.data(); cfg.device = d; cfg.shard_rank = d; cfg.shard_count = ndev;
        CK_MPPI(mppi_create(&cfg, &h[d]), nullptr);
        nrec = mppi_record_size(h[d]);
        CK_HIP(hipSetDevice(d));
        CK_HIP(hipStreamCreate(&st[d]));
        CK_HIP(hipMalloc((void **)&x_dev[d], sizeof(float) * s));
        CK_HIP(hipMalloc((void **)&u_dev[d], sizeof(float) * a));
        CK_HIP(hipMalloc((void **)&rec[d], sizeof(float) * nrec));
        CK_HIP(hipMalloc((void **)&recs[d], sizeof(float) * nrec * ndev));
        CK_HIP(hipMemset(x_dev[d], 0, sizeof(float) * s));
    }
    if (p2p) {
        std::vector<void *> inbox(ndev);
        for (int d = 0; d < ndev; ++d) CK_MPPI(mppi_shard_p2p_export(h[d], nullptr, &inbox[d]), h[d]);
        for (int d = 0; d < ndev; ++d) {
            CK_HIP(hipSetDevice(d));
            for (int e = 0; e < ndev; ++e) if (e != d) {
                hipError_t pe = hipDeviceEnablePeerAccess(e, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { fprintf(stderr, "no peer access %d -> %d\n", d, e); return 6; }
            }
            CK_MPPI(mppi_shard_p2p_attach(h[d], inbox.data(), ndev, 1000), h[d]);
        }
    }
    std::vector<float> x(s, 0.f), u(a, 0.f), u_other(a, 0.f);
    std::vector<mppi_collectives> coll(ndev);
    for (int d = 0; d < ndev; ++d) // RCCL's entry points as they are: the signatures mppi_collectives declares are theirs
        coll[d] = {reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, void *)>(&ncclAllGather),
                   reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, void *)>(&ncclAllReduce), (void *)comm[d]};
    auto steps_of = [&](int d, int n) -> int { // n whole steps of GPU d, one C call each
        for (int it = 0; it < n; ++it) CK_MPPI(mppi_shard_step(h[d], x_dev[d], u_dev[d], &coll[d], st[d]), h[d]);
        return 0;
    };
    auto step = [&]() -> int {
        if (one_call) {
            if (ndev == 1) return steps_of(0, 1);
            std::vector<int> rc(ndev, 0);
            std::vector<std::thread> th;
            for (int d = 0; d < ndev; ++d) th.emplace_back([&, d] { rc[d] = steps_of(d, 1); });
            for (auto &t : th) t.join();
            for (int d = 0; d < ndev; ++d) if (rc[d]) return rc[d];
            return 0;
        }
        if (p2p) {
            for (int d = 0; d < ndev; ++d) CK_MPPI(mppi_shard_p2p_step(h[d], x_dev[d], u_dev[d], st[d]), h[d]);
            return 0;
        }
        for (int d = 0; d < ndev; ++d) CK_MPPI(mppi_shard_partial(h[d], x_dev[d], rec[d], st[d]), h[d]);
        CK_NCCL(ncclGroupStart());
        for (int d = 0; d < ndev; ++d) CK_NCCL(ncclAllGather(rec[d], recs[d], nrec, ncclFloat, comm[d], st[d]));
        CK_NCCL(ncclGroupEnd());
        for (int d = 0; d < ndev; ++d) CK_MPPI(mppi_shard_finish(h[d], recs[d], ndev, u_dev[d], st[d]), h[d]);
        return 0;
    };
    // closed loop: the plant (same point mass) runs on the host
    for (int it = 0; it < steps; ++it) {
        for (int d = 0; d < ndev; ++d) { CK_HIP(hipSetDevice(d)); CK_HIP(hipMemcpyAsync(x_dev[d], x.data(), sizeof(float) * s, hipMemcpyHostToDevice, st[d])); }
        if (int rc = step()) return rc;
        CK_HIP(hipSetDevice(0));
        CK_HIP(hipMemcpyAsync(u.data(), u_dev[0], sizeof(float) * a, hipMemcpyDeviceToHost, st[0]));
        for (int d = 0; d < ndev; ++d) { CK_HIP(hipSetDevice(d)); CK_HIP(hipStreamSynchronize(st[d])); }
        if (ndev > 1) { // every GPU must hold the same control, bit for bit
            CK_HIP(hipSetDevice(ndev - 1));
            CK_HIP(hipMemcpy(u_other.data(), u_dev[ndev - 1], sizeof(float) * a, hipMemcpyDeviceToHost));
            for (int j = 0; j < a; ++j) if (u_other[j] != u[j]) { fprintf(stderr, "replicated controls differ\n"); return 5; }
        }
        for (int j = 0; j < a; ++j) {
            x[2 * j] = x[2 * j] + 0.1f * x[2 * j + 1] + 0.005f * u[j];
            x[2 * j + 1] = x[2 * j + 1] + 0.1f * u[j];
        }
    }
    float d2 = 0, d2_init = 0;
    for (int i = 0; i < s; ++i) { d2 += (x[i] - goal[i]) * (x[i] - goal[i]); d2_init += goal[i] * goal[i]; }
    // pipelined timing: steps enqueued back to back, one synchronisation at the end
    auto t0 = std::chrono::high_resolution_clock::now();
    if (one_call && ndev > 1) { // one thread per GPU runs its 200 steps
        std::vector<int> rc(ndev, 0);
        std::vector<std::thread> th;
        for (int d = 0; d < ndev; ++d) th.emplace_back([&, d] { rc[d] = steps_of(d, 200); });
        for (auto &t : th) t.join();
        for (int d = 0; d < ndev; ++d) if (rc[d]) return rc[d];
    } else
        for (int it = 0; it < 200; ++it) if (int rc = step()) return rc;
    for (int d = 0; d < ndev; ++d) { CK_HIP(hipSetDevice(d)); CK_HIP(hipStreamSynchronize(st[d])); }
    const double el = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count() / 200;
    // host time of one step's enqueue: 20 steps into an EMPTY queue (nothing blocks on the GPU), timed before the synchronisation
    auto h0 = std::chrono::high_resolution_clock::now();
    for (int it = 0; it < 20; ++it) if (int rc = step()) return rc;
    const double host_us = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - h0).count() / 20 * 1e6;
    for (int d = 0; d < ndev; ++d) { CK_HIP(hipSetDevice(d)); CK_HIP(hipStreamSynchronize(st[d])); }
    for (int d = 0; d < ndev && p2p; ++d) {
        int late = 0;
        CK_MPPI(mppi_shard_p2p_status(h[d], &late), h[d]);
        if (late) { fprintf(stderr, "GPU %d: a packet missed its deadline\n", d); return 7; }
    }
    printf("%s exchange, %d GPU(s), K=%d per GPU, tau=%d: |x-goal|^2 %g -> %g after %d closed-loop steps; %.1f us per sharded control step = %.3g rollouts/s; host enqueue %.1f us per step\n",
           p2p ? "direct" : (one_call ? "RCCL all-gather (mppi_shard_step)" : "RCCL all-gather"), ndev, kper, tau, d2_init, d2, steps, el * 1e6, (double)kper * ndev / el, host_us);
    for (int d = 0; d < ndev; ++d) { mppi_destroy(h[d]); ncclCommDestroy(comm[d]); }
    return d2 < 0.5f * d2_init ? 0 : 1;
}
