// mfma_bf16_shadow.hip — what hides in the shadow of v_mfma_f32_32x32x16_bf16 on gfx950? (generated from mfma_f32_shadow.hip:
// same harness, the bf16 MFMA of k_rollout_mlp_bx3 in place of the f32 one; 32 cycles = the MFMA alone.)
// This measures whether other instructions of the
// SAME wave (and of a second wave on the SIMD) overlap with it: a loop of 32 MFMAs on two alternating accumulators,
// with N filler instructions after each MFMA, one asm statement per loop body. Reported: cycles per MFMA (s_memtime
// over the loop / MFMAs issued by the wave), for W = 1 and 2 waves per SIMD.
//   64 cycles = the MFMA alone; 64 + N*c = fillers serialise with it; 64 flat up to some N = they hide.
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_bf16_shadow.hip -o build/mfma_bf16_shadow
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// accumulators v[64:79], v[80:95]; A operand a0 / v2; B operand v3; filler destinations v16..v31 (sources v1, v2)
#define MF0 "v_mfma_f32_32x32x16_bf16 v[64:79], a[0:3], v[100:103], v[64:79]\n\t"
#define MF1 "v_mfma_f32_32x32x16_bf16 v[80:95], a[4:7], v[100:103], v[80:95]\n\t"
#define F_FMA(n) "v_fma_f32 v" #n ", v" #n ", v1, v2\n\t"
#define F_MAX(n) "v_max_f32_e32 v" #n ", v1, v" #n "\n\t"
#define F_MOV(n) "v_mov_b32_e32 v" #n ", v1\n\t"
#define F_PKFMA(n) "v_pk_fma_f32 v[32:33], v[32:33], v[4:5], v[6:7]\n\t"
#define F_DSR(n) "ds_read_b32 v" #n ", v8 offset:" #n "*256\n\t"
#define F_DSR128(n) "ds_read_b128 v[40:43], v9 offset:" #n "*1024\n\t"
#define F_DSW(n) "ds_write_b128 v9, v[4:7] offset:" #n "*1024\n\t"
#define F_SNOP(n) "s_nop 0\n\t"
#define F_SALU(n) "s_add_u32 s40, s40, 1\n\t"
#define F_CVT(n) "v_cvt_pk_bf16_f32 v" #n ", v1, v2\n\t"
#define F_EXP(n) "v_exp_f32_e32 v" #n ", v" #n "\n\t"
#define F_MULLO(n) "v_mul_lo_u32 v" #n ", v" #n ", v3\n\t"
// dependent chains: one accumulator pair (every instruction waits for the previous one) and three rotating pairs
#define F_PKDEP1(n) "v_pk_fma_f32 v[32:33], v[4:5], v[6:7], v[32:33]\n\t"
#define F_PKDEP3(n) "v_pk_fma_f32 v[32:33], v[4:5], v[6:7], v[32:33]\n\tv_pk_fma_f32 v[34:35], v[4:5], v[6:7], v[34:35]\n\tv_pk_fma_f32 v[36:37], v[4:5], v[6:7], v[36:37]\n\t"
#define F_FMADEP1(n) "v_fma_f32 v16, v1, v2, v16\n\t"
#define F_FMADEP3(n) "v_fma_f32 v16, v1, v2, v16\n\tv_fma_f32 v17, v1, v2, v17\n\tv_fma_f32 v18, v1, v2, v18\n\t"
// the layer-3 row of k_rollout_mlp2: relu of an accumulator register, then 3 packed fmas that broadcast it
#define F_L3ROW(n) "v_max_f32 v16, 0, v96\n\ts_nop 0\n\tv_pk_fma_f32 v[32:33], v[16:17], v[4:5], v[32:33] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[34:35], v[16:17], v[6:7], v[34:35] op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 v[36:37], v[16:17], v[4:5], v[36:37] op_sel_hi:[0,1,1]\n\t"
// in-place relu of registers that no MFMA in flight touches (k_rollout_mlp2's relu lump), 16 distinct registers
#define F_RELU(n) "v_max_f32 v" #n ", 0, v" #n "\n\t"
#define N0(F)
#define N1(F) F(16)
#define N2(F) F(16) F(17)
#define N4(F) F(16) F(17) F(18) F(19)
#define N8(F) F(16) F(17) F(18) F(19) F(20) F(21) F(22) F(23)
#define N12(F) N8(F) F(24) F(25) F(26) F(27)
#define N16(F) N12(F) F(28) F(29) F(30) F(31)
#define N32(F) N16(F) N16(F)
// 8 MFMAs per asm body, fillers after each
#define BODY(NF, F) MF0 NF(F) MF1 NF(F) MF0 NF(F) MF1 NF(F) MF0 NF(F) MF1 NF(F) MF0 NF(F) MF1 NF(F)
// the rollout kernel's shape: both MFMAs of a k pair (each behind its s_nop 1), then the lump
#define BODY2(NF, F) "s_nop 1\n\t" MF0 "s_nop 1\n\t" MF1 NF(F) "s_nop 1\n\t" MF0 "s_nop 1\n\t" MF1 NF(F) "s_nop 1\n\t" MF0 "s_nop 1\n\t" MF1 NF(F) "s_nop 1\n\t" MF0 "s_nop 1\n\t" MF1 NF(F)
#define CLOB "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", \
    "v32", "v33", "v34", "v35", "v36", "v37", "v96", "v40", "v41", "v42", "v43", "s40", "scc", "memory", \
    "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
    "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "v100", "v101", "v102", "v103"

constexpr int ITERS = 256;
#define KERNEL(NAME, NF, F) KERNELB(NAME, BODY(NF, F))
#define KERNEL2(NAME, NF, F) KERNELB(NAME, BODY2(NF, F))
#define KERNELB(NAME, THEBODY)                                                                                            \
    __global__ __launch_bounds__(512) void NAME(unsigned long long *out, float seed)                                    \
    {                                                                                                                  \
        extern __shared__ float lds[];                                                                                 \
        lds[threadIdx.x] = seed;                                                                                       \
        __syncthreads();                                                                                               \
        asm volatile("v_mov_b32 v1, %0\n\tv_mov_b32 v2, 0x3f7fff00\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\t"  \
                     "v_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\tv_and_b32 v8, 63, %1\n\tv_lshlrev_b32 v8, 2, v8\n\tv_lshlrev_b32 v9, 2, v8\n\t"    \
                     "v_accvgpr_write_b32 a0, v2\n\tv_accvgpr_write_b32 a1, v1\n\tv_accvgpr_write_b32 a2, v2\n\tv_accvgpr_write_b32 a3, v1\n\tv_accvgpr_write_b32 a4, v2\n\tv_accvgpr_write_b32 a5, v1\n\tv_accvgpr_write_b32 a6, v2\n\tv_accvgpr_write_b32 a7, v1\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"                                      \
                     "v_mov_b32 v16, v1\n\tv_mov_b32 v17, v1\n\tv_mov_b32 v18, v1\n\tv_mov_b32 v19, v1\n\tv_mov_b32 v20, v1\n\tv_mov_b32 v21, v1\n\t" \
                     "v_mov_b32 v22, v1\n\tv_mov_b32 v23, v1\n\tv_mov_b32 v24, v1\n\tv_mov_b32 v25, v1\n\tv_mov_b32 v26, v1\n\tv_mov_b32 v27, v1\n\t" \
                     "v_mov_b32 v32, v1\n\tv_mov_b32 v33, v1\n\tv_mov_b32 v34, v1\n\tv_mov_b32 v35, v1\n\tv_mov_b32 v36, v1\n\tv_mov_b32 v37, v1\n\tv_mov_b32 v96, v1\n\ts_mov_b32 s40, 0\n\t" ::"v"(seed), "v"(threadIdx.x) : CLOB);     \
        for (int i = 64; i < 96; ++i) asm volatile("" ::: "memory");                                                    \
        asm volatile("v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\tv_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\t" \
                     "v_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\tv_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\t" \
                     "v_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0\n\tv_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\t" \
                     "v_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\t" \
                     "v_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\tv_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\t" \
                     "v_mov_b32 v94, 0\n\tv_mov_b32 v95, 0\n\t" ::: CLOB);                                               \
        __syncthreads();                                                                                               \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                    \
        for (int it = 0; it < ITERS; ++it) asm volatile(THEBODY "s_waitcnt lgkmcnt(0)\n\t" ::: CLOB);               \
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                    \
        float r;                                                                                                       \
        asm volatile("v_add_f32 %0, v64, v80\n\tv_add_f32 %0, %0, v16\n\tv_add_f32 %0, %0, v32\n\tv_add_f32 %0, %0, v40" : "=v"(r)::CLOB); \
        if (r == 12345.678f) out[4096] = 1;                                                                            \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                                \
    }

KERNEL(k_none, N0, F_FMA)
KERNEL(k_fma1, N1, F_FMA) KERNEL(k_fma2, N2, F_FMA) KERNEL(k_fma4, N4, F_FMA) KERNEL(k_fma8, N8, F_FMA) KERNEL(k_fma12, N12, F_FMA)
KERNEL(k_max4, N4, F_MAX) KERNEL(k_mov4, N4, F_MOV) KERNEL(k_mov8, N8, F_MOV)
KERNEL(k_pk1, N1, F_PKFMA) KERNEL(k_pk2, N2, F_PKFMA) KERNEL(k_pk4, N4, F_PKFMA)
KERNEL(k_dsr1, N1, F_DSR) KERNEL(k_dsr2, N2, F_DSR) KERNEL(k_dsr4, N4, F_DSR)
KERNEL(k_dsr128_1, N1, F_DSR128) KERNEL(k_dsr128_2, N2, F_DSR128)
KERNEL(k_dsw2, N2, F_DSW) KERNEL(k_dsw4, N4, F_DSW)
KERNEL(k_snop4, N4, F_SNOP) KERNEL(k_snop8, N8, F_SNOP) KERNEL(k_salu4, N4, F_SALU) KERNEL(k_salu8, N8, F_SALU)
KERNEL(k_pkd1_4, N4, F_PKDEP1) KERNEL(k_pkd1_8, N8, F_PKDEP1) KERNEL(k_pkd3_1, N1, F_PKDEP3) KERNEL(k_pkd3_4, N4, F_PKDEP3)
KERNEL(k_fd1_4, N4, F_FMADEP1) KERNEL(k_fd1_8, N8, F_FMADEP1) KERNEL(k_fd3_4, N4, F_FMADEP3)
KERNEL(k_l3_1, N1, F_L3ROW) KERNEL(k_l3_2, N2, F_L3ROW) KERNEL(k_l3_4, N4, F_L3ROW)
KERNEL(k_relu16, N16, F_RELU) KERNEL(k_relu32, N32, F_RELU) KERNEL(k_fma16, N16, F_FMA) KERNEL(k_fma32, N32, F_FMA)
KERNEL2(k2_none, N0, F_FMA) KERNEL2(k2_l3x2, N2, F_L3ROW) KERNEL2(k2_l3x4, N4, F_L3ROW) KERNEL2(k2_relu32, N32, F_RELU) KERNEL2(k2_fma8, N8, F_FMA) KERNEL2(k2_fma1, N1, F_FMA)
KERNEL(k_cvt4, N4, F_CVT) KERNEL(k_exp2, N2, F_EXP) KERNEL(k_exp4, N4, F_EXP) KERNEL(k_mullo4, N4, F_MULLO)

struct Case { const char *name; void (*k)(unsigned long long *, float); int nf; };
int main()
{
    Case cases[] = {{"none", k_none, 0}, {"v_fma_f32 x1", k_fma1, 1}, {"v_fma_f32 x2", k_fma2, 2}, {"v_fma_f32 x4", k_fma4, 4}, {"v_fma_f32 x8", k_fma8, 8},
                    {"v_fma_f32 x12", k_fma12, 12}, {"v_max_f32 x4", k_max4, 4}, {"v_mov_b32 x4", k_mov4, 4}, {"v_mov_b32 x8", k_mov8, 8},
                    {"v_pk_fma_f32 x1", k_pk1, 1}, {"v_pk_fma_f32 x2", k_pk2, 2}, {"v_pk_fma_f32 x4", k_pk4, 4},
                    {"ds_read_b32 x1", k_dsr1, 1}, {"ds_read_b32 x2", k_dsr2, 2}, {"ds_read_b32 x4", k_dsr4, 4},
                    {"ds_read_b128 x1", k_dsr128_1, 1}, {"ds_read_b128 x2", k_dsr128_2, 2}, {"ds_write_b128 x2", k_dsw2, 2}, {"ds_write_b128 x4", k_dsw4, 4},
                    {"s_nop 0 x4", k_snop4, 4}, {"s_nop 0 x8", k_snop8, 8}, {"s_add_u32 x4", k_salu4, 4}, {"s_add_u32 x8", k_salu8, 8},
                    {"v_pk_fma_f32 dependent x4", k_pkd1_4, 4}, {"v_pk_fma_f32 dependent x8", k_pkd1_8, 8}, {"v_pk_fma_f32 3 chains x3", k_pkd3_1, 3}, {"v_pk_fma_f32 3 chains x12", k_pkd3_4, 12},
                    {"v_fma_f32 dependent x4", k_fd1_4, 4}, {"v_fma_f32 dependent x8", k_fd1_8, 8}, {"v_fma_f32 3 chains x12", k_fd3_4, 12},
                    {"layer-3 row (v_max + 3 v_pk_fma) x1", k_l3_1, 4}, {"layer-3 row x2", k_l3_2, 8}, {"layer-3 row x4", k_l3_4, 16},
                    {"v_max_f32 in place x16", k_relu16, 16}, {"v_max_f32 in place x32", k_relu32, 32}, {"v_fma_f32 x16", k_fma16, 16}, {"v_fma_f32 x32", k_fma32, 32},
                    {"[2 MFMAs, lump] none", k2_none, 0}, {"[2 MFMAs, lump] layer-3 row x2", k2_l3x2, 8}, {"[2 MFMAs, lump] layer-3 row x4", k2_l3x4, 16}, {"[2 MFMAs, lump] v_max in place x32", k2_relu32, 32}, {"[2 MFMAs, lump] v_fma_f32 x8", k2_fma8, 8}, {"[2 MFMAs, lump] v_fma_f32 x1", k2_fma1, 1},
                    {"v_cvt_pk_bf16_f32 x4", k_cvt4, 4}, {"v_exp_f32 x2", k_exp2, 2}, {"v_exp_f32 x4", k_exp4, 4}, {"v_mul_lo_u32 x4", k_mullo4, 4}};
    unsigned long long *d;
    CK(hipMalloc((void **)&d, 8 * 8192));
    printf("{\"what\": \"cycles per v_mfma_f32_32x32x16_bf16 with N fillers after each, per wave (median over waves); W waves per SIMD\", \"rows\": [\n");
    bool first = true;
    for (const Case &c : cases) {
        for (int W = 1; W <= 2; ++W) {
            const int threads = 256 * W, blocks = 256;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(c.k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(c.k, dim3(blocks), dim3(threads), 96 * 1024, 0, d, 1.0f);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(blocks * 8);
            CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> v;
            for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * W; ++w) v.push_back((double)h[b * 8 + w] / (ITERS * 8.0));
            std::sort(v.begin(), v.end());
            printf("%s {\"filler\": \"%s\", \"n\": %d, \"waves_per_simd\": %d, \"cyc_per_mfma_wave\": %.1f, \"cyc_per_mfma_simd\": %.1f}", first ? "" : ",\n", c.name, c.nf, W,
                   v[v.size() / 2], v[v.size() / 2] / W);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
