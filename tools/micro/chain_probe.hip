// chain_probe.hip — what a control step costs when its kernels are chained by DATA instead of by dispatch order (r05).
// A step = tile workgroups (stand-ins for the rollout tiles: wait for the nominal sequence U of the previous step, work, publish a
// record) + column workgroups (the finish: wait for every tile's record, combine, publish U'). Everything that crosses a workgroup
// is an 8-byte {value, sequence} granule written by ONE write-through store and polled with L1-bypassing loads
// (MI355X_MICROARCH.md, handoff-1to1): no flag, no fence, no atomic. Because the dependency travels with the data, consecutive
// steps may be launched on DIFFERENT streams: step i+1's workgroups are placed and armed while step i still runs, and the fixed cost
// of a dependent dispatch leaves the critical path. Measured here, per grid shape:
//   fused    one kernel per step (tiles + columns in one grid), steps round-robin over 1 / 2 / 3 streams;
//   split    a tile kernel and a column kernel per step (the big shape: the tile kernel fills the chip by LDS, 4 per CU),
//            tile kernels round-robin over 1 / 2 streams, column kernels over 1 / 2 streams of their own;
//   plain    the same work as ordinary dependent launches on one stream (what the library does today): no polling needed.
// and the host round trip of an ARMED kernel: the kernel is resident and polls a word in pinned host memory; the host stores
// {x, seq} there and watches a pinned reply word (what mppi_next would pay instead of launch + dispatch).
// Every spin has a wall-clock deadline (s_memrealtime) and raises a status word on expiry: a missing producer is a timeout, not a hang.
//   hipcc -O2 --offload-arch=gfx950 tools/micro/chain_probe.hip -o build/chain_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("\n%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ long long wall_clock() { return (long long)__builtin_amdgcn_s_memrealtime(); } // 100 MHz
__device__ __forceinline__ void gr_store(u64 *p, float v, unsigned seq)
{
    __hip_atomic_store(p, ((u64)seq << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// poll until the granule carries `seq`; false on deadline / dead flag
__device__ __forceinline__ bool gr_wait(const u64 *p, unsigned seq, float *v, long long ticks, unsigned *status)
{
    const long long t0 = wall_clock();
    for (unsigned it = 0;; ++it) {
        const u64 g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(g >> 32) == seq) { *v = __uint_as_float((unsigned)g); return true; }
        if ((it & 15u) == 15u) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (wall_clock() - t0 > ticks) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

__device__ __forceinline__ float spin_work(float x, int iters)
{
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f); // dependent chain
    return x;
}

// record granule (col, b) at recs[col * nbp + b]; U granule c at U[c]
// tile role: needs all HA values of U(seq-1); publishes 2+HA record granules tagged seq
__device__ __forceinline__ void tile_role(int b, int nb, int nbp, int HA, unsigned seq, int iters, const u64 *Uin, u64 *recs,
                                          long long ticks, unsigned *status, int poll)
{
    __shared__ float U_s[512];
    __shared__ int bad_s;
    const int tid = threadIdx.x;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    float u = 0.f;
    if (tid < HA) {
        if (poll) { if (!gr_wait(Uin + tid, seq - 1u, &u, ticks, status)) bad_s = 1; }
        else u = __uint_as_float((unsigned)Uin[tid]);
        U_s[tid] = u;
    }
    __syncthreads();
    if (bad_s) return;
    float acc = spin_work(U_s[(tid + b) % HA], iters);
    // record: beta = b-dependent, eta = 1, V[c] = small function of U[c] and b
    for (int c = tid; c < 2 + HA; c += blockDim.x) {
        float v;
        if (c == 0) v = (float)(b % 7) + 0.0f * acc;
        else if (c == 1) v = 1.0f;
        else v = 0.001f * (float)((b + c) % 5) - 0.002f + 0.0f * acc + 1e-4f * U_s[c - 2];
        gr_store(recs + (size_t)c * nbp + b, v, seq);
    }
}

// column role: U'[c] = U[c] + sum_b r_b V_b[c] / sum_b r_b eta_b, r_b = exp(-(beta_b - beta))
__device__ __forceinline__ void column_role(int c, int nb, int nbp, int HA, unsigned seq, const u64 *Uin, u64 *Uout, const u64 *recs,
                                            long long ticks, unsigned *status, int poll)
{
    __shared__ float red[3][8];
    __shared__ int bad_c;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) bad_c = 0;
    __syncthreads();
    float bb[4], ee[4], vv[4];
    bool ok = true;
    if (poll) {
        // SENTINEL first (r05a measured the naive form — every thread of every column workgroup sweeping from the start — at +3 us
        // per step against plain launches: 128 x 256 x 12 polling loads per sweep beside the working tiles): ONE lane polls ONE
        // granule (a tile's beta, a different tile per column) with a sleep between polls; only when that tile has published do
        // all threads sweep. Tiles finish within a fraction of a microsecond of each other, so a sweep or two completes it.
        const long long t0 = wall_clock();
        if (tid == 0) {
            float dummy;
            if (!gr_wait(recs + (c % nb), seq, &dummy, ticks, status)) bad_c = 1;
        }
        __syncthreads();
        if (bad_c) return;
        for (unsigned it = 0;; ++it) {
            u64 g[12];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = min(tid + i * 256, nb - 1);
                g[3 * i + 0] = __hip_atomic_load(recs + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g[3 * i + 1] = __hip_atomic_load(recs + (size_t)nbp + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g[3 * i + 2] = __hip_atomic_load(recs + (size_t)(2 + c) * nbp + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool all = true;
#pragma unroll
            for (int j = 0; j < 12; ++j) all = all && (unsigned)(g[j] >> 32) == seq;
            if (all) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = tid + i * 256 < nb;
                    bb[i] = in ? __uint_as_float((unsigned)g[3 * i]) : 3e38f;
                    ee[i] = in ? __uint_as_float((unsigned)g[3 * i + 1]) : 0.f;
                    vv[i] = in ? __uint_as_float((unsigned)g[3 * i + 2]) : 0.f;
                }
                break;
            }
            if ((it & 7u) == 7u) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                if (wall_clock() - t0 > ticks) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { for (int i = 0; i < 4; ++i) { bb[i] = 3e38f; ee[i] = 0.f; vv[i] = 0.f; } }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = tid + i * 256;
            bb[i] = 3e38f; ee[i] = 0.f; vv[i] = 0.f;
            if (b < nb) {
                bb[i] = __uint_as_float((unsigned)recs[b]);
                ee[i] = __uint_as_float((unsigned)recs[(size_t)nbp + b]);
                vv[i] = __uint_as_float((unsigned)recs[(size_t)(2 + c) * nbp + b]);
            }
        }
    }
    if (!ok) bad_c = 1;
    float bmin = fminf(fminf(bb[0], bb[1]), fminf(bb[2], bb[3]));
    for (int o = 32; o; o >>= 1) bmin = fminf(bmin, __shfl_xor(bmin, o));
    if (lane == 0) red[0][w] = bmin;
    __syncthreads();
    if (bad_c) return;
    float beta = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    float se = 0.f, sv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (tid + i * 256 < nb) { const float r = __expf(-(bb[i] - beta)); se += r * ee[i]; sv += r * vv[i]; }
    }
    for (int o = 32; o; o >>= 1) { se += __shfl_xor(se, o); sv += __shfl_xor(sv, o); }
    if (lane == 0) { red[1][w] = se; red[2][w] = sv; }
    __syncthreads();
    if (tid == 0) {
        const float eta = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float V = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
        float u;
        if (poll) { if (!gr_wait(Uin + c, seq - 1u, &u, ticks, status)) return; }
        else u = __uint_as_float((unsigned)Uin[c]);
        gr_store(Uout + c, u + V / eta, seq);
    }
}

__global__ __launch_bounds__(256) void k_fused(int nb, int nbp, int HA, unsigned seq, int iters, const u64 *Uin, u64 *Uout, u64 *recs,
                                               long long ticks, unsigned *status, int poll)
{
    if ((int)blockIdx.x < nb) tile_role(blockIdx.x, nb, nbp, HA, seq, iters, Uin, recs, ticks, status, poll);
    else column_role(blockIdx.x - nb, nb, nbp, HA, seq, Uin, Uout, recs, ticks, status, poll);
}
__global__ __launch_bounds__(256) void k_tiles(int nb, int nbp, int HA, unsigned seq, int iters, const u64 *Uin, u64 *recs,
                                               long long ticks, unsigned *status, int poll)
{
    extern __shared__ float dyn[];
    if (iters < 0) dyn[threadIdx.x] = 0.f; // keep the dynamic LDS
    tile_role(blockIdx.x, nb, nbp, HA, seq, iters, Uin, recs, ticks, status, poll);
}
__global__ __launch_bounds__(256) void k_columns(int nb, int nbp, int HA, unsigned seq, const u64 *Uin, u64 *Uout, const u64 *recs,
                                                 long long ticks, unsigned *status, int poll)
{
    column_role(blockIdx.x, nb, nbp, HA, seq, Uin, Uout, recs, ticks, status, poll);
}
__global__ void k_empty(float *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0.f; }

// armed kernel: lane 0 polls a host word for seq, answers with a granule in pinned host memory
__global__ void k_armed(const u64 *host_word, u64 *reply, unsigned seq, long long ticks, unsigned *status)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock();
    for (;;) {
        const u64 g = __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(g >> 32) == seq) {
            __hip_atomic_store(reply, ((u64)seq << 32) | (g & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (wall_clock() - t0 > ticks) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __builtin_amdgcn_s_sleep(1);
    }
}
// launched AFTER the host word is written: what today's path pays (launch + dispatch + reply)
__global__ void k_reply(const u64 *host_word, u64 *reply, unsigned seq)
{
    if (threadIdx.x != 0) return;
    const u64 g = __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reply, ((u64)seq << 32) | (g & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static sigjmp_buf g_jb;
static void on_segv(int) { siglongjmp(g_jb, 1); }

struct Shape { const char *name; int nb, HA, iters; };

// host reference of the chain (float, same order is not needed: tolerance 1e-4)
static void host_chain(int nb, int HA, int steps, std::vector<float> &U)
{
    U.assign(HA, 0.f);
    for (int s = 0; s < steps; ++s) {
        std::vector<float> Un(HA);
        float beta = 3e38f;
        for (int b = 0; b < nb; ++b) beta = std::min(beta, (float)(b % 7));
        double eta = 0.0;
        for (int b = 0; b < nb; ++b) eta += std::exp(-((double)(b % 7) - beta));
        for (int c = 0; c < HA; ++c) {
            double V = 0.0;
            for (int b = 0; b < nb; ++b)
                V += std::exp(-((double)(b % 7) - beta)) * (0.001 * ((b + c + 2) % 5) - 0.002 + 1e-4 * U[c]);
            Un[c] = U[c] + (float)(V / eta);
        }
        U = Un;
    }
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    hipStream_t st[6];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int nbp_max = 1024, HA_max = 256;
    u64 *U[2], *recs;
    unsigned *status;
    CK(hipMalloc((void **)&U[0], sizeof(u64) * HA_max));
    CK(hipMalloc((void **)&U[1], sizeof(u64) * HA_max));
    CK(hipMalloc((void **)&recs, sizeof(u64) * (size_t)(2 + HA_max) * nbp_max * 2)); // two parities
    CK(hipMalloc((void **)&status, 64));
    const long long ticks = 20 * 100000ll; // 20 ms
    printf("{\n \"steps\": %d,\n", steps);

    // ---- 0. plain dependent launches of an empty kernel: wall per kernel without events -------------------------------
    {
        float *buf; CK(hipMalloc((void **)&buf, 4096));
        for (int grid : {1, 256, 1024}) {
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st[0], buf);
            CK(hipStreamSynchronize(st[0]));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st[0], buf);
            CK(hipStreamSynchronize(st[0]));
            printf(" \"empty_chain_wall_us_grid%d\": %.2f,\n", grid, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / steps);
        }
        CK(hipFree(buf));
    }

    const Shape shapes[] = {
        {"pm2d_K4096 (64 tiles, 128 columns)", 64, 128, 200},
        {"pm2d_K4096 (64 tiles, 128 columns), short tiles", 64, 128, 100},
        {"pm3d_K3000_H50 (47 tiles, 150 columns)", 47, 150, 200},
        {"K8192 (128 tiles, 192 columns)", 128, 192, 200},
        {"pm3d_K65536 (1024 tiles, 192 columns)", 1024, 192, 600},
    };
    printf(" \"shapes\": [\n");
    bool first_shape = true;
    for (const Shape &sh : shapes) {
        const int nb = sh.nb, HA = sh.HA, nbp = nbp_max;
        std::vector<float> Uref;
        host_chain(nb, HA, 8, Uref); // validate 8 steps
        printf("%s  {\"shape\": \"%s\"", first_shape ? "" : ",\n", sh.name);
        first_shape = false;
        // isolated durations of the two roles (events)
        {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipMemset(U[0], 0, sizeof(u64) * HA_max));
            float ms_t = 0.f, ms_c = 0.f;
            for (int r = 0; r < 3; ++r) {
                hipExtLaunchKernelGGL(k_tiles, dim3(nb), dim3(256), 0, st[0], e0, e1, 0, nb, nbp, HA, 1u, sh.iters, U[0], recs, ticks, status, 0);
                CK(hipStreamSynchronize(st[0])); CK(hipEventElapsedTime(&ms_t, e0, e1));
                hipExtLaunchKernelGGL(k_columns, dim3(HA), dim3(256), 0, st[0], e0, e1, 0, nb, nbp, HA, 1u, U[0], U[1], recs, ticks, status, 0);
                CK(hipStreamSynchronize(st[0])); CK(hipEventElapsedTime(&ms_c, e0, e1));
            }
            printf(", \"tile_kernel_event_us\": %.2f, \"column_kernel_event_us\": %.2f", ms_t * 1e3, ms_c * 1e3);
        }
        auto reset = [&]() -> int {
            CK(hipMemset(U[0], 0, sizeof(u64) * HA_max)); // value 0, tag 0 = "step 0 done"
            CK(hipMemset(U[1], 0xff, sizeof(u64) * HA_max));
            CK(hipMemset(recs, 0xff, sizeof(u64) * (size_t)(2 + HA_max) * nbp_max * 2));
            CK(hipMemset(status, 0, 64));
            CK(hipDeviceSynchronize());
            return 0;
        };
        auto check = [&](int nsteps, const char *tag) -> int {
            CK(hipDeviceSynchronize());
            unsigned stt = 0; CK(hipMemcpy(&stt, status, 4, hipMemcpyDeviceToHost));
            std::vector<u64> got(HA);
            CK(hipMemcpy(got.data(), U[nsteps & 1], sizeof(u64) * HA, hipMemcpyDeviceToHost));
            double md = 0.0; bool tags = true;
            for (int c = 0; c < HA; ++c) {
                unsigned bits = (unsigned)got[c]; float v; memcpy(&v, &bits, 4);
                md = std::max(md, (double)std::fabs(v - Uref[c]));
                tags = tags && (unsigned)(got[c] >> 32) == (unsigned)nsteps;
            }
            printf(", \"%s_check\": {\"timed_out\": %u, \"tags_ok\": %s, \"max_abs_diff_vs_host_8_steps\": %.2e}", tag, stt, tags ? "true" : "false", md);
            return 0;
        };
        // ---- plain: dependent launches on one stream, no polling ---------------------------------------------------
        for (int fused = 1; fused >= 0; --fused) {
            for (int pass = 0; pass < 2; ++pass) { // pass 0: 8 validated steps; pass 1: timed
                const int n = pass == 0 ? 8 : steps;
                if (reset()) return 1;
                const auto t0 = std::chrono::steady_clock::now();
                for (int s = 1; s <= n; ++s) {
                    u64 *Ui = U[(s - 1) & 1], *Uo = U[s & 1];
                    if (fused) { // without polling a fused grid cannot order tiles before columns: two launches of the fused kernel's roles
                        hipLaunchKernelGGL(k_tiles, dim3(nb), dim3(256), 0, st[0], nb, nbp, HA, (unsigned)s, sh.iters, Ui, recs, ticks, status, 0);
                        hipLaunchKernelGGL(k_columns, dim3(HA), dim3(256), 0, st[0], nb, nbp, HA, (unsigned)s, Ui, Uo, recs, ticks, status, 0);
                    } else {
                        hipLaunchKernelGGL(k_tiles, dim3(nb), dim3(256), 40 * 1024, st[0], nb, nbp, HA, (unsigned)s, sh.iters, Ui, recs, ticks, status, 0);
                        hipLaunchKernelGGL(k_columns, dim3(HA), dim3(256), 0, st[0], nb, nbp, HA, (unsigned)s, Ui, Uo, recs, ticks, status, 0);
                    }
                }
                CK(hipDeviceSynchronize());
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
                if (pass == 0) { if (fused && check(8, "plain")) return 1; }
                else printf(", \"plain_two_launches%s_us_per_step\": %.2f", fused ? "" : "_40KB_lds", us);
            }
        }
        // ---- fused: one polled kernel per step over ns streams -------------------------------------------------------
        for (int ns : {1, 2, 3}) {
            for (int pass = 0; pass < 2; ++pass) {
                const int n = pass == 0 ? 8 : steps;
                if (reset()) return 1;
                const auto t0 = std::chrono::steady_clock::now();
                for (int s = 1; s <= n; ++s)
                    hipLaunchKernelGGL(k_fused, dim3(nb + HA), dim3(256), 0, st[s % ns], nb, nbp, HA, (unsigned)s, sh.iters, U[(s - 1) & 1], U[s & 1],
                                       recs + (size_t)(s & 1) * (2 + HA_max) * nbp_max, ticks, status, 1);
                CK(hipDeviceSynchronize());
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
                char tag[64]; snprintf(tag, sizeof tag, "fused_%dstream", ns);
                if (pass == 0) { if (check(8, tag)) return 1; }
                else {
                    unsigned stt = 0; CK(hipMemcpy(&stt, status, 4, hipMemcpyDeviceToHost));
                    printf(", \"fused_%dstream_us_per_step\": %.2f, \"fused_%dstream_timed_out\": %u", ns, us, ns, stt);
                }
            }
        }
        // ---- split: tile kernel (40 KB LDS: 4 per CU) + column kernel, each kind on its own streams -----------------
        for (int nts : {1, 2}) for (int ncs : {1, 2}) {
            for (int pass = 0; pass < 2; ++pass) {
                const int n = pass == 0 ? 8 : steps;
                if (reset()) return 1;
                const auto t0 = std::chrono::steady_clock::now();
                for (int s = 1; s <= n; ++s) {
                    u64 *rc = recs + (size_t)(s & 1) * (2 + HA_max) * nbp_max;
                    hipLaunchKernelGGL(k_tiles, dim3(nb), dim3(256), 40 * 1024, st[s % nts], nb, nbp, HA, (unsigned)s, sh.iters, U[(s - 1) & 1], rc, ticks, status, 1);
                    hipLaunchKernelGGL(k_columns, dim3(HA), dim3(256), 0, st[2 + s % ncs], nb, nbp, HA, (unsigned)s, U[(s - 1) & 1], U[s & 1], rc, ticks, status, 1);
                }
                CK(hipDeviceSynchronize());
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
                char tag[64]; snprintf(tag, sizeof tag, "split_t%d_c%d", nts, ncs);
                if (pass == 0) { if (check(8, tag)) return 1; }
                else {
                    unsigned stt = 0; CK(hipMemcpy(&stt, status, 4, hipMemcpyDeviceToHost));
                    printf(", \"%s_us_per_step\": %.2f, \"%s_timed_out\": %u", tag, us, tag, stt);
                }
            }
        }
        printf("}");
        fflush(stdout);
    }
    printf("\n ],\n");

    // ---- host round trip: armed kernel vs launch-after-write --------------------------------------------------------
    {
        u64 *h_word, *h_reply, *d_word, *d_reply; unsigned *h_stat, *d_stat;
        CK(hipHostMalloc((void **)&h_word, 4096, hipHostMallocMapped));
        CK(hipHostGetDevicePointer((void **)&d_word, h_word, 0));
        h_reply = h_word + 64; d_reply = d_word + 64;
        h_stat = (unsigned *)(h_word + 128); d_stat = (unsigned *)(d_word + 128);
        memset(h_word, 0, 4096);
        const int n = 300;
        for (int mode = 0; mode < 3; ++mode) { // 0: launch after write (today); 1: armed, spin on reply; 2: armed on a second stream pair (next one armed before this one is answered)
            std::vector<double> rt;
            if (mode >= 1) hipLaunchKernelGGL(k_armed, dim3(1), dim3(64), 0, st[0], d_word, d_reply, 1000u * (mode + 1) + 1u, 100 * 100000ll, d_stat);
            for (int i = 1; i <= n; ++i) {
                const unsigned seq = 1000u * (mode + 1) + (unsigned)i;
                const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(30); // let the armed kernel get resident
                while (std::chrono::steady_clock::now() < spin_until) {}
                const auto t0 = std::chrono::steady_clock::now();
                __atomic_store_n(h_word, ((u64)seq << 32) | (u64)i, __ATOMIC_RELEASE);
                if (mode == 0) hipLaunchKernelGGL(k_reply, dim3(1), dim3(64), 0, st[0], d_word, d_reply, seq);
                else if (i < n) hipLaunchKernelGGL(k_armed, dim3(1), dim3(64), 0, st[mode == 2 ? (i & 1) : 0], d_word, d_reply, seq + 1u, 100 * 100000ll, d_stat); // arm the next one
                while ((unsigned)(__atomic_load_n(h_reply, __ATOMIC_ACQUIRE) >> 32) != seq) {
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { printf(" \"armed_mode%d_error\": \"no reply at i=%d\",\n", mode, i); goto done_mode; }
                }
                rt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
        done_mode:
            CK(hipDeviceSynchronize());
            if (!rt.empty()) {
                std::sort(rt.begin(), rt.end());
                printf(" \"host_roundtrip_%s_us\": {\"median\": %.2f, \"p10\": %.2f, \"p90\": %.2f, \"timed_out\": %u},\n",
                       mode == 0 ? "launch_after_write" : mode == 1 ? "armed_next_launched_before_spin" : "armed_two_streams",
                       rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10], *h_stat);
            }
        }
        // can the host store straight into device memory (large BAR)? then the armed kernel polls local memory
        u64 *fg = nullptr;
        int host_can_write = 0;
        if (hipExtMallocWithFlags((void **)&fg, 4096, hipDeviceMallocFinegrained) == hipSuccess) {
            struct sigaction sa{}, old{};
            sa.sa_handler = on_segv; sigemptyset(&sa.sa_mask);
            sigaction(SIGSEGV, &sa, &old);
            struct sigaction oldb{}; sigaction(SIGBUS, &sa, &oldb);
            if (sigsetjmp(g_jb, 1) == 0) { *(volatile u64 *)fg = 42ull; host_can_write = 1; }
            sigaction(SIGSEGV, &old, nullptr); sigaction(SIGBUS, &oldb, nullptr);
        }
        printf(" \"host_can_store_to_finegrained_device_memory\": %d", host_can_write);
        if (host_can_write) {
            CK(hipMemset(fg, 0, 4096)); CK(hipDeviceSynchronize());
            std::vector<double> rt;
            hipLaunchKernelGGL(k_armed, dim3(1), dim3(64), 0, st[0], fg, d_reply, 9001u, 100 * 100000ll, d_stat);
            for (int i = 1; i <= n; ++i) {
                const unsigned seq = 9000u + (unsigned)i;
                const auto spin_until = std::chrono::steady_clock::now() + std::chrono::microseconds(30);
                while (std::chrono::steady_clock::now() < spin_until) {}
                const auto t0 = std::chrono::steady_clock::now();
                __atomic_store_n((u64 *)fg, ((u64)seq << 32) | (u64)i, __ATOMIC_RELEASE);
                if (i < n) hipLaunchKernelGGL(k_armed, dim3(1), dim3(64), 0, st[0], fg, d_reply, seq + 1u, 100 * 100000ll, d_stat);
                bool ok = true;
                while ((unsigned)(__atomic_load_n(h_reply, __ATOMIC_ACQUIRE) >> 32) != seq)
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { ok = false; break; }
                if (!ok) break;
                rt.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            CK(hipDeviceSynchronize());
            if (!rt.empty()) { std::sort(rt.begin(), rt.end()); printf(",\n \"host_roundtrip_armed_word_in_device_memory_us\": {\"median\": %.2f, \"p10\": %.2f, \"p90\": %.2f, \"n\": %zu}", rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10], rt.size()); }
        }
        printf("\n}\n");
    }
    return 0;
}
