// mppi_learner.hip — the learner of the learned model_base: LearnerBase._train_step / train
// (scripts/src/learners/learner_base.py:324-358, 469-496): full-batch Adam on the mean squared error of the normalised
// state delta, for the reference's small Dense networks (nn_model.py:54-60: Dense(32, relu) x3 + Dense(13); widths <= 32,
// up to 4 layers). It supplies the weights k_rollout_gen / k_rollout_mlp32 roll out. One train step = three launches:
//   k_learn_fwd_bwd  one sample per lane: forward, loss term, backward; activations A_l and deltas D_l go to HBM as [n][32]
//   k_learn_grad     dW_l = A_{l-1}^T D_l is a GEMM whose k dimension is the BATCH: v_mfma_f32_32x32x2_f32 (exact fp32), every
//                    wave accumulates a 32x32 tile over its slice of the samples (the bias gradient rides along as a second MFMA
//                    against a row of ones), fixed-order reduction over the waves, one partial tile per workgroup
//   k_learn_adam     partial tiles summed in fixed order, Adam (Keras form: w -= lr sqrt(1-b2^t)/(1-b1^t) m/(sqrt(v)+eps)),
//                    the step counter (advanced by the forward pass) and the loss live on the device: a train loop needs no host round trip
// Deterministic: no float atomics anywhere. All dimensions are padded to 32 (zero weights / activations in the padding).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "mppi_c.h"

namespace {

constexpr int W32 = 32;            // padded width of every layer
constexpr int kMaxLayers = 4;
constexpr int kChunk = 512;        // samples per workgroup of k_learn_grad (r03: 4096 left 8 workgroups on 256 CUs: 180 us per step at n = 8192)
constexpr int kGradThreads = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Net { // device pointers; weights padded: W[l] is [32][32] ([in][out]), b[l] is [32]
    float *W[kMaxLayers], *b[kMaxLayers];
    int n_layers, width[kMaxLayers + 1]; // width[0] = inputs, width[l+1] = outputs of layer l
};

// one sample per thread: forward, squared error, backward. A[l] ([n][32], l = 0..n_layers-1: the INPUT of layer l) and
// D[l] ([n][32]: dLoss/d(pre-activation of layer l)) are what the gradient GEMMs read. loss_part[block] = sum of squared errors.
__global__ __launch_bounds__(256) void k_learn_fwd_bwd(const Net net, const float *__restrict__ X, const float *__restrict__ Y, int n,
                                                       float *__restrict__ A, float *__restrict__ D, float *__restrict__ loss_part,
                                                       float *__restrict__ pred_out, int *__restrict__ step, int apply)
{
    __shared__ float w_s[kMaxLayers][W32 * W32];
    __shared__ float b_s[kMaxLayers][W32];
    __shared__ float red_s[4];
    const int tid = threadIdx.x, L = net.n_layers;
    for (int l = 0; l < L; ++l) {
        for (int i = tid; i < W32 * W32; i += 256) w_s[l][i] = net.W[l][i];
        if (tid < W32) b_s[l][tid] = net.b[l][tid];
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + tid;
    const bool valid = i < n;
    const int ii = valid ? i : n - 1;
    const int nin = net.width[0], nout = net.width[L];
    float a[kMaxLayers + 1][W32]; // a[l] = input of layer l; a[L] = the prediction
#pragma unroll
    for (int j = 0; j < W32; ++j) a[0][j] = j < nin ? X[(size_t)ii * nin + j] : 0.0f;
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l < L) {
#pragma unroll
            for (int o = 0; o < W32; ++o) {
                float acc = b_s[l][o];
#pragma unroll
                for (int j = 0; j < W32; ++j) acc = __builtin_fmaf(a[l][j], w_s[l][j * W32 + o], acc);
                a[l + 1][o] = (l + 1 < L && acc < 0.0f) ? 0.0f : acc; // relu on the hidden layers
            }
        }
    }
    // loss = mean over samples and outputs of (pred - Y)^2 (tf.reduce_mean(squared_difference), learner_base.py:474-476)
    float d[W32], se = 0.0f;
    const float scale = 2.0f / ((float)n * (float)nout);
#pragma unroll
    for (int o = 0; o < W32; ++o) {
        float pred = 0.0f;
#pragma unroll
        for (int l = 1; l <= kMaxLayers; ++l) pred = l == L ? a[l][o] : pred;
        const float e = (o < nout && valid) ? pred - Y[(size_t)ii * nout + o] : 0.0f;
        se += e * e;
        d[o] = scale * e;
        if (pred_out != nullptr && valid && o < nout) pred_out[(size_t)i * nout + o] = pred;
    }
    // backward: D_l = d ; d_{l-1}[j] = (sum_o W_l[j][o] d[o]) * [a_l[j] > 0]
#pragma unroll
    for (int l = kMaxLayers - 1; l >= 0; --l) {
        if (l < L) {
            if (valid) {
                float4 *Dp = reinterpret_cast<float4 *>(D + ((size_t)l * n + i) * W32), *Ap = reinterpret_cast<float4 *>(A + ((size_t)l * n + i) * W32);
#pragma unroll
                for (int q = 0; q < W32 / 4; ++q) {
                    Dp[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
                    Ap[q] = make_float4(a[l][4 * q], a[l][4 * q + 1], a[l][4 * q + 2], a[l][4 * q + 3]);
                }
            }
            if (l > 0) {
                float dn[W32];
#pragma unroll
                for (int j = 0; j < W32; ++j) {
                    float acc = 0.0f;
#pragma unroll
                    for (int o = 0; o < W32; ++o) acc = __builtin_fmaf(w_s[l][j * W32 + o], d[o], acc);
                    dn[j] = a[l][j] > 0.0f ? acc : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < W32; ++j) d[j] = dn[j];
            }
        }
    }
    // block sum of the squared errors, fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off, 64);
    if ((tid & 63) == 0) red_s[tid >> 6] = se;
    __syncthreads();
    if (tid == 0) loss_part[blockIdx.x] = ((red_s[0] + red_s[1]) + red_s[2]) + red_s[3];
    // the optimiser step this pass belongs to: advanced HERE, two launches ahead of the Adam kernel whose workgroups all read it
    if (apply && blockIdx.x == 0 && tid == 0) step[0] = step[0] + 1;
}

// dW_l = A_l^T D_l and db_l = 1^T D_l over the samples of one chunk: grid (chunks, layers), 4 waves per workgroup.
// v_mfma_f32_32x32x2_f32: D[m][n] += sum_{k<2} Aop[m][k] Bop[k][n]; lane (r = lane & 31, k = lane >> 5) supplies Aop[r][k] and Bop[k][r].
// Here m = input unit, n = output unit, k = sample: Aop[r][k] = A_l[sample k][r], Bop[k][r] = D_l[sample k][r] — both are one
// coalesced 256-byte read of two consecutive [32]-rows. part is [chunks][layers][33][32]: rows 0..31 = dW, row 32 = db.
__global__ __launch_bounds__(kGradThreads) void k_learn_grad(const float *__restrict__ A, const float *__restrict__ D, int n, float *__restrict__ part)
{
    __shared__ float tile_s[4][33 * W32];
    const int chunk = blockIdx.x, l = blockIdx.y, L = gridDim.y;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, r = lane & 31, k = lane >> 5;
    const float *Al = A + (size_t)l * n * W32, *Dl = D + (size_t)l * n * W32;
    const int s0 = chunk * kChunk, s1 = min(n, s0 + kChunk);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accb = acc;
    const float one_row0 = r == 0 ? 1.0f : 0.0f; // Aop of the bias MFMA: row 0 is all ones -> row 0 of the result = column sums of D
    for (int s = s0 + 2 * w; s < s1; s += 8) {   // wave w takes sample pairs (s, s+1), s = s0 + 2w, +8, ...
        const int sk = s + k;
        const float av = sk < s1 ? Al[(size_t)sk * W32 + r] : 0.0f;
        const float dv = sk < s1 ? Dl[(size_t)sk * W32 + r] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, dv, acc, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x2f32(one_row0, dv, accb, 0, 0, 0);
    }
    // accumulator register q of lane (r, k) is element [row = 8 (q >> 2) + 4 k + (q & 3)][col = r]
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = 8 * (q >> 2) + 4 * k + (q & 3);
        tile_s[w][row * W32 + r] = acc[q];
        if (row == 0) tile_s[w][32 * W32 + r] = accb[q];
    }
    __syncthreads();
    float *out = part + ((size_t)chunk * L + l) * 33 * W32;
    for (int e = tid; e < 33 * W32; e += kGradThreads) out[e] = ((tile_s[0][e] + tile_s[1][e]) + tile_s[2][e]) + tile_s[3][e];
}

struct AdamState { float *m[kMaxLayers], *v[kMaxLayers], *mb[kMaxLayers], *vb[kMaxLayers]; };

// gradient = fixed-order sum of the chunks' partial tiles; Adam as tf.keras.optimizers.Adam applies it (learner_base.py:31,
// :149): m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps). One workgroup (4 x 1056 parameters).
// loss_out = the loss of THIS step's forward pass (before the update); grads_out optional.
__global__ __launch_bounds__(256) void k_learn_adam(const Net net, AdamState st, const float *__restrict__ part, int chunks,
                                                    const float *__restrict__ loss_part, int loss_blocks, int n,
                                                    float lr, float b1, float b2, float eps, const int *__restrict__ step, float *__restrict__ loss_out,
                                                    float *__restrict__ grads_out, int apply)
{
    // grid (layers, ceil(33 * 32 / 256)): one thread per weight / bias. The step counter was advanced by this step's forward pass.
    const int L = net.n_layers, l = blockIdx.x, tid = threadIdx.x, e = blockIdx.y * 256 + tid;
    const int t = apply ? step[0] : step[0] + 1;
    const float bc1 = 1.0f - powf(b1, (float)t), bc2 = 1.0f - powf(b2, (float)t);
    const float lr_t = lr * sqrtf(bc2) / bc1;
    if (e < 33 * W32) {
        float g = 0.0f;
        for (int c = 0; c < chunks; ++c) g += part[((size_t)c * L + l) * 33 * W32 + e]; // fixed order: deterministic
        const int row = e / W32, col = e % W32;
        const bool is_b = row == 32;
        const bool live = col < net.width[l + 1] && (is_b || row < net.width[l]); // padding stays exactly zero
        if (!live) g = 0.0f;
        if (grads_out != nullptr) grads_out[(size_t)l * 33 * W32 + e] = g;
        if (apply && live) {
            float *w = is_b ? net.b[l] + col : net.W[l] + row * W32 + col;
            float *m = is_b ? st.mb[l] + col : st.m[l] + row * W32 + col;
            float *v = is_b ? st.vb[l] + col : st.v[l] + row * W32 + col;
            const float mn = b1 * *m + (1.0f - b1) * g, vn = b2 * *v + (1.0f - b2) * g * g;
            *m = mn; *v = vn;
            *w = *w - lr_t * mn / (sqrtf(vn) + eps);
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        float s = 0.0f;
        for (int q = 0; q < loss_blocks; ++q) s += loss_part[q];
        loss_out[0] = s / ((float)n * (float)net.width[L]);
    }
}

} // namespace

struct mppi_learner {
    int device = 0;
    hipStream_t stream = nullptr;
    Net net{};
    AdamState adam{};
    std::vector<float *> bufs;
    float *dX = nullptr, *dY = nullptr, *dA = nullptr, *dD = nullptr, *d_part = nullptr, *d_loss_part = nullptr, *d_loss = nullptr, *d_grads = nullptr;
    int *d_step = nullptr;
    int n = 0, cap = 0, chunks = 0, blocks = 0;
    std::string err;
};

static thread_local std::string g_learner_err;
static mppi_status lfail(mppi_learner *l, mppi_status st, const std::string &msg)
{
    if (l) l->err = msg; else g_learner_err = msg;
    return st;
}
#define L_TRY(l, expr)                                                                            \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return lfail((l), MPPI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

extern "C" const char *mppi_learner_last_error(const mppi_learner *l) { return l ? l->err.c_str() : g_learner_err.c_str(); }

extern "C" void mppi_learner_destroy(mppi_learner *l)
{
    if (!l) return;
    (void)hipSetDevice(l->device);
    if (l->stream) (void)hipStreamSynchronize(l->stream);
    for (float *p : l->bufs) if (p) (void)hipFree(p);
    float *more[] = {l->dX, l->dY, l->dA, l->dD, l->d_part, l->d_loss_part, l->d_loss, l->d_grads};
    for (float *p : more) if (p) (void)hipFree(p);
    if (l->d_step) (void)hipFree(l->d_step);
    if (l->stream) (void)hipStreamDestroy(l->stream);
    delete l;
}

extern "C" mppi_status mppi_learner_set_weights(mppi_learner *l, const float *const *W, const float *const *b)
{
    if (!l || !W || !b) return l ? lfail(l, MPPI_ERR_INVALID_ARG, "NULL weights") : MPPI_ERR_INVALID_ARG;
    L_TRY(l, hipSetDevice(l->device));
    for (int k = 0; k < l->net.n_layers; ++k) {
        const int in = l->net.width[k], out = l->net.width[k + 1];
        std::vector<float> pad(W32 * W32, 0.0f), pb(W32, 0.0f);
        for (int i = 0; i < in; ++i) for (int o = 0; o < out; ++o) pad[i * W32 + o] = W[k][(size_t)i * out + o];
        for (int o = 0; o < out; ++o) pb[o] = b[k][o];
        L_TRY(l, hipMemcpy(l->net.W[k], pad.data(), sizeof(float) * W32 * W32, hipMemcpyHostToDevice));
        L_TRY(l, hipMemcpy(l->net.b[k], pb.data(), sizeof(float) * W32, hipMemcpyHostToDevice));
    }
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_reset_optimizer(mppi_learner *l)
{
    if (!l) return MPPI_ERR_INVALID_ARG;
    L_TRY(l, hipSetDevice(l->device));
    for (int k = 0; k < l->net.n_layers; ++k) {
        L_TRY(l, hipMemset(l->adam.m[k], 0, sizeof(float) * W32 * W32)); L_TRY(l, hipMemset(l->adam.v[k], 0, sizeof(float) * W32 * W32));
        L_TRY(l, hipMemset(l->adam.mb[k], 0, sizeof(float) * W32)); L_TRY(l, hipMemset(l->adam.vb[k], 0, sizeof(float) * W32));
    }
    L_TRY(l, hipMemset(l->d_step, 0, sizeof(int)));
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_create(int n_layers, const int32_t *widths, const float *const *W, const float *const *b, int device,
                                           mppi_learner **out)
{
    if (!out || !widths || !W || !b) return lfail(nullptr, MPPI_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (n_layers < 1 || n_layers > kMaxLayers) return lfail(nullptr, MPPI_ERR_UNSUPPORTED, "learner: 1 to 4 Dense layers");
    for (int k = 0; k <= n_layers; ++k) if (widths[k] < 1 || widths[k] > W32) return lfail(nullptr, MPPI_ERR_UNSUPPORTED, "learner: layer widths 1..32 (nn_model.py:54-60)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return lfail(nullptr, MPPI_ERR_NO_DEVICE, "no HIP device visible: the learner has no CPU path"); }
    if (device < 0 || device >= ndev) return lfail(nullptr, MPPI_ERR_NO_DEVICE, "device ordinal out of range");
    mppi_learner *l = new (std::nothrow) mppi_learner();
    if (!l) return lfail(nullptr, MPPI_ERR_ALLOC, "out of host memory");
    l->device = device;
    l->net.n_layers = n_layers;
    for (int k = 0; k <= n_layers; ++k) l->net.width[k] = widths[k];
    auto body = [&]() -> mppi_status {
        L_TRY(l, hipSetDevice(device));
        L_TRY(l, hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
        auto alloc = [&](float **p, size_t nfl) -> hipError_t { hipError_t e = hipMalloc((void **)p, sizeof(float) * nfl); if (e == hipSuccess) l->bufs.push_back(*p); return e; };
        for (int k = 0; k < n_layers; ++k) {
            L_TRY(l, alloc(&l->net.W[k], W32 * W32)); L_TRY(l, alloc(&l->net.b[k], W32));
            L_TRY(l, alloc(&l->adam.m[k], W32 * W32)); L_TRY(l, alloc(&l->adam.v[k], W32 * W32));
            L_TRY(l, alloc(&l->adam.mb[k], W32)); L_TRY(l, alloc(&l->adam.vb[k], W32));
        }
        L_TRY(l, hipMalloc((void **)&l->d_step, sizeof(int)));
        L_TRY(l, hipMalloc((void **)&l->d_loss, sizeof(float)));
        L_TRY(l, hipMalloc((void **)&l->d_grads, sizeof(float) * kMaxLayers * 33 * W32));
        mppi_status s = mppi_learner_set_weights(l, W, b);
        if (s != MPPI_OK) return s;
        return mppi_learner_reset_optimizer(l);
    };
    mppi_status s = body();
    if (s != MPPI_OK) { g_learner_err = l->err; mppi_learner_destroy(l); return s; }
    *out = l;
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_set_data(mppi_learner *l, const float *X, const float *Y, int n)
{
    if (!l || !X || !Y || n <= 0) return l ? lfail(l, MPPI_ERR_INVALID_ARG, "X is [n, in], Y is [n, out], n > 0") : MPPI_ERR_INVALID_ARG;
    L_TRY(l, hipSetDevice(l->device));
    L_TRY(l, hipStreamSynchronize(l->stream));
    const int nin = l->net.width[0], nout = l->net.width[l->net.n_layers], L = l->net.n_layers;
    if (n > l->cap) {
        float **ps[] = {&l->dX, &l->dY, &l->dA, &l->dD, &l->d_part, &l->d_loss_part};
        for (float **p : ps) if (*p) { L_TRY(l, hipFree(*p)); *p = nullptr; }
        const int chunks = (n + kChunk - 1) / kChunk, blocks = (n + 255) / 256;
        L_TRY(l, hipMalloc((void **)&l->dX, sizeof(float) * (size_t)n * nin));
        L_TRY(l, hipMalloc((void **)&l->dY, sizeof(float) * (size_t)n * nout));
        L_TRY(l, hipMalloc((void **)&l->dA, sizeof(float) * (size_t)L * n * W32));
        L_TRY(l, hipMalloc((void **)&l->dD, sizeof(float) * (size_t)L * n * W32));
        L_TRY(l, hipMalloc((void **)&l->d_part, sizeof(float) * (size_t)chunks * L * 33 * W32));
        L_TRY(l, hipMalloc((void **)&l->d_loss_part, sizeof(float) * blocks));
        l->cap = n;
    }
    l->n = n; l->chunks = (n + kChunk - 1) / kChunk; l->blocks = (n + 255) / 256;
    L_TRY(l, hipMemcpy(l->dX, X, sizeof(float) * (size_t)n * nin, hipMemcpyHostToDevice));
    L_TRY(l, hipMemcpy(l->dY, Y, sizeof(float) * (size_t)n * nout, hipMemcpyHostToDevice));
    return MPPI_OK;
}

static mppi_status enqueue_step(mppi_learner *l, float lr, float b1, float b2, float eps, int apply, float *pred_dev)
{
    const int L = l->net.n_layers;
    hipLaunchKernelGGL(k_learn_fwd_bwd, dim3(l->blocks), dim3(256), 0, l->stream, l->net, (const float *)l->dX, (const float *)l->dY, l->n, l->dA, l->dD,
                       l->d_loss_part, pred_dev, l->d_step, apply);
    L_TRY(l, hipGetLastError());
    hipLaunchKernelGGL(k_learn_grad, dim3(l->chunks, L), dim3(kGradThreads), 0, l->stream, (const float *)l->dA, (const float *)l->dD, l->n, l->d_part);
    L_TRY(l, hipGetLastError());
    hipLaunchKernelGGL(k_learn_adam, dim3(L, (33 * W32 + 255) / 256), dim3(256), 0, l->stream, l->net, l->adam, (const float *)l->d_part, l->chunks,
                       (const float *)l->d_loss_part, l->blocks, l->n, lr, b1, b2, eps, (const int *)l->d_step, l->d_loss, l->d_grads, apply);
    L_TRY(l, hipGetLastError());
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_train(mppi_learner *l, int steps, float lr, float beta1, float beta2, float eps, float *loss_first, float *loss_last)
{
    if (!l) return MPPI_ERR_INVALID_ARG;
    if (l->n <= 0) return lfail(l, MPPI_ERR_INVALID_ARG, "no data: call mppi_learner_set_data first");
    if (steps <= 0 || !(lr > 0.0f) || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f) || !(eps >= 0.0f))
        return lfail(l, MPPI_ERR_INVALID_ARG, "steps > 0, lr > 0, 0 <= beta < 1, eps >= 0");
    L_TRY(l, hipSetDevice(l->device));
    float first = 0.0f, last = 0.0f;
    for (int s = 0; s < steps; ++s) {
        mppi_status st = enqueue_step(l, lr, beta1, beta2, eps, 1, nullptr);
        if (st != MPPI_OK) return st;
        if (s == 0 && loss_first) L_TRY(l, hipMemcpyAsync(&first, l->d_loss, sizeof(float), hipMemcpyDeviceToHost, l->stream));
    }
    if (loss_last) L_TRY(l, hipMemcpyAsync(&last, l->d_loss, sizeof(float), hipMemcpyDeviceToHost, l->stream));
    L_TRY(l, hipStreamSynchronize(l->stream));
    if (loss_first) *loss_first = first;
    if (loss_last) *loss_last = last;
    return MPPI_OK;
}

// loss (and optionally the gradients [n_layers][33][32]: rows 0..31 dW padded, row 32 db; and the predictions [n, out]) of the CURRENT weights; no update
extern "C" mppi_status mppi_learner_evaluate(mppi_learner *l, float *loss_out, float *grads_out, float *pred_out)
{
    if (!l) return MPPI_ERR_INVALID_ARG;
    if (l->n <= 0) return lfail(l, MPPI_ERR_INVALID_ARG, "no data: call mppi_learner_set_data first");
    L_TRY(l, hipSetDevice(l->device));
    const int nout = l->net.width[l->net.n_layers];
    float *dp = nullptr;
    if (pred_out) L_TRY(l, hipMalloc((void **)&dp, sizeof(float) * (size_t)l->n * nout));
    mppi_status st = enqueue_step(l, 1.0f, 0.9f, 0.999f, 1e-7f, 0, dp);
    if (st == MPPI_OK) {
        hipError_t e = hipSuccess;
        if (loss_out) e = hipMemcpyAsync(loss_out, l->d_loss, sizeof(float), hipMemcpyDeviceToHost, l->stream);
        if (e == hipSuccess && grads_out) e = hipMemcpyAsync(grads_out, l->d_grads, sizeof(float) * l->net.n_layers * 33 * W32, hipMemcpyDeviceToHost, l->stream);
        if (e == hipSuccess && pred_out) e = hipMemcpyAsync(pred_out, dp, sizeof(float) * (size_t)l->n * nout, hipMemcpyDeviceToHost, l->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(l->stream);
        if (e != hipSuccess) st = lfail(l, MPPI_ERR_HIP, hipGetErrorString(e));
    }
    if (dp) (void)hipFree(dp);
    return st;
}

extern "C" mppi_status mppi_learner_get_weights(mppi_learner *l, float *const *W, float *const *b)
{
    if (!l || !W || !b) return l ? lfail(l, MPPI_ERR_INVALID_ARG, "NULL output") : MPPI_ERR_INVALID_ARG;
    L_TRY(l, hipSetDevice(l->device));
    L_TRY(l, hipStreamSynchronize(l->stream));
    for (int k = 0; k < l->net.n_layers; ++k) {
        const int in = l->net.width[k], out = l->net.width[k + 1];
        std::vector<float> pad(W32 * W32), pb(W32);
        L_TRY(l, hipMemcpy(pad.data(), l->net.W[k], sizeof(float) * W32 * W32, hipMemcpyDeviceToHost));
        L_TRY(l, hipMemcpy(pb.data(), l->net.b[k], sizeof(float) * W32, hipMemcpyDeviceToHost));
        for (int i = 0; i < in; ++i) for (int o = 0; o < out; ++o) W[k][(size_t)i * out + o] = pad[i * W32 + o];
        for (int o = 0; o < out; ++o) b[k][o] = pb[o];
    }
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_get_step(mppi_learner *l, int *step)
{
    if (!l || !step) return MPPI_ERR_INVALID_ARG;
    L_TRY(l, hipSetDevice(l->device));
    L_TRY(l, hipStreamSynchronize(l->stream));
    L_TRY(l, hipMemcpy(step, l->d_step, sizeof(int), hipMemcpyDeviceToHost));
    return MPPI_OK;
}

// ---- persistence (reference: nn_model.py:137-142 save_params / load_params, learner_base.py:66-68) ----------------
// One flat little-endian file, documented in include/mppi_c.h: magic "MPPILRN1", n_layers, widths[5], Adam step, has_norm, then per layer
// W b mW mb vW vb (fp32, compact [in][out]), then the optional fp64 normalisation (xmean, xstd, ymean, ystd).
namespace {
struct LearnerFileHeader { char magic[8]; int32_t n_layers; int32_t widths[5]; int32_t step; int32_t has_norm; };
static_assert(sizeof(LearnerFileHeader) == 40, "file header layout");
const char kLearnerMagic[8] = {'M', 'P', 'P', 'I', 'L', 'R', 'N', '1'};

bool read_header(FILE *f, LearnerFileHeader &hd)
{
    if (fread(&hd, sizeof(hd), 1, f) != 1 || memcmp(hd.magic, kLearnerMagic, 8) != 0) return false;
    if (hd.n_layers < 1 || hd.n_layers > kMaxLayers) return false;
    for (int k = 0; k <= hd.n_layers; ++k) if (hd.widths[k] < 1 || hd.widths[k] > W32) return false;
    return hd.step >= 0 && (hd.has_norm == 0 || hd.has_norm == 1);
}
} // namespace

extern "C" mppi_status mppi_learner_peek(const char *filename, int *n_layers, int32_t *widths)
{
    if (!filename || !n_layers || !widths) return lfail(nullptr, MPPI_ERR_INVALID_ARG, "NULL argument");
    FILE *f = fopen(filename, "rb");
    if (!f) return lfail(nullptr, MPPI_ERR_IO, std::string("cannot open ") + filename);
    LearnerFileHeader hd;
    const bool ok = read_header(f, hd);
    fclose(f);
    if (!ok) return lfail(nullptr, MPPI_ERR_IO, std::string(filename) + ": not a learner file (MPPILRN1)");
    *n_layers = hd.n_layers;
    for (int k = 0; k < 5; ++k) widths[k] = k <= hd.n_layers ? hd.widths[k] : 0;
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_save(mppi_learner *l, const char *filename, const double *xmean, const double *xstd,
                                         const double *ymean, const double *ystd)
{
    if (!l || !filename) return l ? lfail(l, MPPI_ERR_INVALID_ARG, "NULL filename") : MPPI_ERR_INVALID_ARG;
    const bool norm = xmean || xstd || ymean || ystd;
    if (norm && !(xmean && xstd && ymean && ystd)) return lfail(l, MPPI_ERR_INVALID_ARG, "normalisation: all four of xmean, xstd, ymean, ystd or none");
    L_TRY(l, hipSetDevice(l->device));
    L_TRY(l, hipStreamSynchronize(l->stream));
    const int L = l->net.n_layers;
    LearnerFileHeader hd{};
    memcpy(hd.magic, kLearnerMagic, 8);
    hd.n_layers = L;
    for (int k = 0; k <= L; ++k) hd.widths[k] = l->net.width[k];
    L_TRY(l, hipMemcpy(&hd.step, l->d_step, sizeof(int), hipMemcpyDeviceToHost));
    hd.has_norm = norm ? 1 : 0;
    std::vector<float> body;
    std::vector<float> pad(W32 * W32), pb(W32);
    for (int k = 0; k < L; ++k) {
        const int in = l->net.width[k], out = l->net.width[k + 1];
        const float *mats[3] = {l->net.W[k], l->adam.m[k], l->adam.v[k]}, *vecs[3] = {l->net.b[k], l->adam.mb[k], l->adam.vb[k]};
        for (int q = 0; q < 3; ++q) {
            L_TRY(l, hipMemcpy(pad.data(), mats[q], sizeof(float) * W32 * W32, hipMemcpyDeviceToHost));
            L_TRY(l, hipMemcpy(pb.data(), vecs[q], sizeof(float) * W32, hipMemcpyDeviceToHost));
            for (int i = 0; i < in; ++i) for (int o = 0; o < out; ++o) body.push_back(pad[i * W32 + o]);
            for (int o = 0; o < out; ++o) body.push_back(pb[o]);
        }
    }
    // written beside the target and renamed over it: a reader (or a crash) never sees half a file
    const std::string tmp = std::string(filename) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return lfail(l, MPPI_ERR_IO, "cannot open " + tmp);
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1 && fwrite(body.data(), sizeof(float), body.size(), f) == body.size();
    if (ok && norm) {
        const size_t nx = (size_t)l->net.width[0], ny = (size_t)l->net.width[L];
        ok = fwrite(xmean, sizeof(double), nx, f) == nx && fwrite(xstd, sizeof(double), nx, f) == nx &&
             fwrite(ymean, sizeof(double), ny, f) == ny && fwrite(ystd, sizeof(double), ny, f) == ny;
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), filename) != 0) { (void)remove(tmp.c_str()); return lfail(l, MPPI_ERR_IO, std::string("cannot write ") + filename); }
    return MPPI_OK;
}

extern "C" mppi_status mppi_learner_load(mppi_learner *l, const char *filename, double *xmean, double *xstd, double *ymean, double *ystd, int *has_norm)
{
    if (!l || !filename) return l ? lfail(l, MPPI_ERR_INVALID_ARG, "NULL filename") : MPPI_ERR_INVALID_ARG;
    FILE *f = fopen(filename, "rb");
    if (!f) return lfail(l, MPPI_ERR_IO, std::string("cannot open ") + filename);
    LearnerFileHeader hd;
    const int L = l->net.n_layers;
    bool ok = read_header(f, hd);
    if (ok && hd.n_layers != L) ok = false;
    for (int k = 0; ok && k <= L; ++k) ok = hd.widths[k] == l->net.width[k];
    if (!ok) { fclose(f); return lfail(l, MPPI_ERR_IO, std::string(filename) + ": not a learner file of this network's layer widths"); }
    size_t nfl = 0;
    for (int k = 0; k < L; ++k) nfl += (size_t)3 * (l->net.width[k] + 1) * l->net.width[k + 1];
    std::vector<float> body(nfl);
    ok = fread(body.data(), sizeof(float), nfl, f) == nfl;
    const size_t nx = (size_t)l->net.width[0], ny = (size_t)l->net.width[L];
    std::vector<double> nrm(2 * (nx + ny));
    if (ok && hd.has_norm) ok = fread(nrm.data(), sizeof(double), nrm.size(), f) == nrm.size();
    fclose(f);
    if (!ok) return lfail(l, MPPI_ERR_IO, std::string(filename) + ": truncated");
    L_TRY(l, hipSetDevice(l->device));
    L_TRY(l, hipStreamSynchronize(l->stream));
    const float *p = body.data();
    std::vector<float> pad(W32 * W32), pb(W32);
    for (int k = 0; k < L; ++k) {
        const int in = l->net.width[k], out = l->net.width[k + 1];
        float *mats[3] = {l->net.W[k], l->adam.m[k], l->adam.v[k]}, *vecs[3] = {l->net.b[k], l->adam.mb[k], l->adam.vb[k]};
        for (int q = 0; q < 3; ++q) {
            std::fill(pad.begin(), pad.end(), 0.0f); std::fill(pb.begin(), pb.end(), 0.0f); // the padding stays exactly zero
            for (int i = 0; i < in; ++i) for (int o = 0; o < out; ++o) pad[i * W32 + o] = *p++;
            for (int o = 0; o < out; ++o) pb[o] = *p++;
            L_TRY(l, hipMemcpy(mats[q], pad.data(), sizeof(float) * W32 * W32, hipMemcpyHostToDevice));
            L_TRY(l, hipMemcpy(vecs[q], pb.data(), sizeof(float) * W32, hipMemcpyHostToDevice));
        }
    }
    L_TRY(l, hipMemcpy(l->d_step, &hd.step, sizeof(int), hipMemcpyHostToDevice));
    if (has_norm) *has_norm = hd.has_norm;
    if (hd.has_norm && xmean && xstd && ymean && ystd) {
        memcpy(xmean, nrm.data(), sizeof(double) * nx); memcpy(xstd, nrm.data() + nx, sizeof(double) * nx);
        memcpy(ymean, nrm.data() + 2 * nx, sizeof(double) * ny); memcpy(ystd, nrm.data() + 2 * nx + ny, sizeof(double) * ny);
    }
    return MPPI_OK;
}
