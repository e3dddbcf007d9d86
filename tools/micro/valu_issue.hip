// valu_issue.hip — ISA-verified instruction issue-rate micro-benchmark for gfx950 (MI355X).
//
// Every timed loop is ONE inline-asm statement: the instructions between the two s_memtime stamps are exactly the
// ones written here (hipcc neither packs, fuses nor re-schedules the inside of an asm string). `make_excerpt.sh`
// (tools/micro/valu_issue_isa.sh) disassembles the code object and commits the loop bodies as evidence.
//
// Geometry: 256-thread workgroups = 4 wavefronts; W workgroups per CU are asked for through the dynamic-LDS size and a
// grid of 256 CUs x W. Each wave issues ITERS x 32 instructions on 16 independent destination registers (no dependent
// pair closer than 16 instructions).
// Every wave also records the SIMD it ran on (HW_ID, XCC_ID). Reported per op and W, per SIMD:
//   cyc/inst/SIMD = (last stamp - first stamp over the waves that ran on the SIMD) / (instructions they issued)
// (median over SIMDs; correct whatever the placement was, also when the workgroups came in several rounds), the number
// of waves that shared a SIMD, the per-wave cost (s_memtime delta / instructions of one wave), and the shader clock
// from s_memtime / s_memrealtime (100 MHz).
//
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/valu_issue.hip -o build/valu_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// 16 destination registers v16..v31 (v32..v63 as 16 pairs for the 64-bit ops); sources v1, v2, v3; s[40:41] carry-out
#define R16(F) F(16) F(17) F(18) F(19) F(20) F(21) F(22) F(23) F(24) F(25) F(26) F(27) F(28) F(29) F(30) F(31)
#define P16(F) F(32, 33) F(34, 35) F(36, 37) F(38, 39) F(40, 41) F(42, 43) F(44, 45) F(46, 47) \
               F(48, 49) F(50, 51) F(52, 53) F(54, 55) F(56, 57) F(58, 59) F(60, 61) F(62, 63)

#define I_FMA(n) "v_fma_f32 v" #n ", v" #n ", v1, v2\n\t"
#define I_MUL(n) "v_mul_f32_e32 v" #n ", v1, v" #n "\n\t"
#define I_ADD(n) "v_add_f32_e32 v" #n ", v2, v" #n "\n\t"
#define I_MOV(n) "v_mov_b32_e32 v" #n ", v1\n\t"
#define I_CND(n) "v_cndmask_b32_e32 v" #n ", v1, v" #n ", vcc\n\t"
#define I_CND_NODEP(n) "v_cndmask_b32_e32 v" #n ", v1, v2, vcc\n\t"
#define I_CND_E64(n) "v_cndmask_b32_e64 v" #n ", v1, v" #n ", s[44:45]\n\t"
#define I_CND_ONES(n) "v_cndmask_b32_e64 v" #n ", v1, v" #n ", s[46:47]\n\t"
#define I_MAXF(n) "v_max_f32_e32 v" #n ", v1, v" #n "\n\t"
#define I_PERM(n) "ds_bpermute_b32 v" #n ", v8, v" #n "\n\t"
#define I_READLANE(n) "v_readlane_b32 s43, v" #n ", 3\n\t"
#define I_XOR(n) "v_xor_b32_e32 v" #n ", v3, v" #n "\n\t"
#define I_BITOP3(n) "v_bitop3_b32 v" #n ", v" #n ", v3, v1 bitop3:0x96\n\t"
#define I_CVT(n) "v_cvt_f32_u32_e32 v" #n ", v" #n "\n\t"
#define I_LOG(n) "v_log_f32_e32 v" #n ", v" #n "\n\t"
#define I_SQRT(n) "v_sqrt_f32_e32 v" #n ", v" #n "\n\t"
#define I_SIN(n) "v_sin_f32_e32 v" #n ", v" #n "\n\t"
#define I_EXP(n) "v_exp_f32_e32 v" #n ", v" #n "\n\t"
#define I_RCP(n) "v_rcp_f32_e32 v" #n ", v" #n "\n\t"
#define I_MULLO(n) "v_mul_lo_u32 v" #n ", v" #n ", v3\n\t"
#define I_MULHI(n) "v_mul_hi_u32 v" #n ", v" #n ", v3\n\t"
#define I_DPPADD(n) "v_add_f32_dpp v" #n ", v" #n ", v" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPMOV(n) "v_mov_b32_dpp v" #n ", v" #n " row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define I_FMAMK(n) "v_fmamk_f32 v" #n ", v" #n ", 0x3f800347, v2\n\t"
#define I_PKMUL(a, b) "v_pk_mul_f32 v[" #a ":" #b "], v[" #a ":" #b "], v[4:5]\n\t"
#define I_PKADD(a, b) "v_pk_add_f32 v[" #a ":" #b "], v[" #a ":" #b "], v[4:5]\n\t"
#define I_PKFMA(a, b) "v_pk_fma_f32 v[" #a ":" #b "], v[" #a ":" #b "], v[4:5], v[6:7]\n\t"
#define I_MAD64(a, b) "v_mad_u64_u32 v[" #a ":" #b "], s[40:41], v" #a ", v3, 0\n\t"
#define I_MAD64C(a, b) "v_mad_u64_u32 v[" #a ":" #b "], s[40:41], v" #a ", v3, v[" #a ":" #b "]\n\t"
#define I_ADDF64(a, b) "v_add_f64 v[" #a ":" #b "], v[" #a ":" #b "], v[4:5]\n\t"
#define I_DSW(n) "ds_write_b32 v8, v" #n " offset:" #n "*256\n\t"
#define I_DSR(n) "ds_read_b32 v" #n ", v8 offset:" #n "*256\n\t"
#define I_SNOP(n) "s_nop 0\n\t"
#define I_SNOP7(n) "s_nop 7\n\t"
#define I_SNOP15(n) "s_nop 15\n\t"
// the Philox4x32-10 round as the rollout kernel issues it: 2 x v_mad_u64_u32 + 2 x v_bitop3_b32, each round dependent on
// the previous one (8 independent blocks in flight per wave here; the kernel has A = 3 per horizon group)
#define I_PHILOX(a, b) "v_mad_u64_u32 v[" #a ":" #b "], s[40:41], v" #a ", v3, 0\n\t" \
                       "v_bitop3_b32 v" #a ", v" #b ", v1, v2 bitop3:0x96\n\t"

#define CLOBBERS                                                                                                       \
    "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", \
        "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", \
        "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", \
        "v61", "v62", "v63", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc", "vcc", "memory"

// init: positive finite floats (log/sqrt stay finite: values collapse to fixed points, the issue rate does not depend
// on the data), LDS address v8 = lane*4 (conflict-free ds_read/write_b32), vcc = alternating lanes
#define PREAMBLE                                                                                                       \
    "v_mov_b32 v1, 0x3f800347\n\tv_mov_b32 v2, 0x3a83126f\n\tv_mov_b32 v3, 0xD2511F53\n\t"                              \
    "v_mov_b32 v4, 0x3f800347\n\tv_mov_b32 v5, 0x3f7fff58\n\tv_mov_b32 v6, 0x3a83126f\n\tv_mov_b32 v7, 0x3a83126f\n\t"  \
    "v_mbcnt_lo_u32_b32 v8, -1, 0\n\tv_mbcnt_hi_u32_b32 v8, -1, v8\n\tv_lshlrev_b32 v8, 2, v8\n\t"                      \
    "s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\t"                                                 \
    "s_mov_b32 s44, 0x55555555\n\ts_mov_b32 s45, 0x55555555\n\ts_mov_b64 s[46:47], -1\n\t"                            \
    "v_mov_b32 v16, 2.0\n\tv_mov_b32 v17, 2.0\n\tv_mov_b32 v18, 2.0\n\tv_mov_b32 v19, 2.0\n\t"                          \
    "v_mov_b32 v20, 2.0\n\tv_mov_b32 v21, 2.0\n\tv_mov_b32 v22, 2.0\n\tv_mov_b32 v23, 2.0\n\t"                          \
    "v_mov_b32 v24, 2.0\n\tv_mov_b32 v25, 2.0\n\tv_mov_b32 v26, 2.0\n\tv_mov_b32 v27, 2.0\n\t"                          \
    "v_mov_b32 v28, 2.0\n\tv_mov_b32 v29, 2.0\n\tv_mov_b32 v30, 2.0\n\tv_mov_b32 v31, 2.0\n\t"                          \
    "v_mov_b32 v32, 2.0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 2.0\n\tv_mov_b32 v35, 0\n\t"                              \
    "v_mov_b32 v36, 2.0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 2.0\n\tv_mov_b32 v39, 0\n\t"                              \
    "v_mov_b32 v40, 2.0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 2.0\n\tv_mov_b32 v43, 0\n\t"                              \
    "v_mov_b32 v44, 2.0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 2.0\n\tv_mov_b32 v47, 0\n\t"                              \
    "v_mov_b32 v48, 2.0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 2.0\n\tv_mov_b32 v51, 0\n\t"                              \
    "v_mov_b32 v52, 2.0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 2.0\n\tv_mov_b32 v55, 0\n\t"                              \
    "v_mov_b32 v56, 2.0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 2.0\n\tv_mov_b32 v59, 0\n\t"                              \
    "v_mov_b32 v60, 2.0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 2.0\n\tv_mov_b32 v63, 0\n\t"

// BODY = one sweep of 16 instructions; the loop body holds two sweeps (32 instructions per iteration)
#define DEFINE_KERNEL(NAME, BODY)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, int iters)                                    \
    {                                                                                                                  \
        extern __shared__ float lds_[];                                                                                \
        unsigned long long t0, t1, r0, r1;                                                                             \
        asm volatile(PREAMBLE                                                                                          \
                     "s_mov_b32 s42, %4\n\t"                                                                           \
                     "s_barrier\n\t"                                                                                   \
                     "s_memrealtime %2\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)\n"                                       \
                     ".Lloop_" #NAME "_%=:\n\t" BODY BODY                                                               \
                     "s_sub_u32 s42, s42, 1\n\ts_cmp_lg_u32 s42, 0\n\ts_cbranch_scc1 .Lloop_" #NAME "_%=\n\t"            \
                     "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                               \
                     "s_memtime %1\n\ts_memrealtime %3\n\ts_waitcnt lgkmcnt(0)\n\t"                                     \
                     : "=&s"(t0), "=&s"(t1), "=&s"(r0), "=&s"(r1)                                                      \
                     : "s"(iters)                                                                                      \
                     : CLOBBERS);                                                                                      \
        if ((threadIdx.x & 63) == 0) {                                                                                 \
            unsigned long long *o = out + 5ull * (blockIdx.x * 4 + (threadIdx.x >> 6));                                \
            o[0] = t0; o[1] = t1; o[2] = r0; o[3] = r1;                                                                 \
            /* where the wave ran: HW_REG_HW_ID[15:0] (wave, simd, pipe, cu, sh, se) | HW_REG_XCC_ID[3:0] << 16 */      \
            o[4] = __builtin_amdgcn_s_getreg((15 << 11) | 4) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16);       \
        }                                                                                                              \
        if (iters < 0) lds_[threadIdx.x] = 0.f; /* keep the dynamic LDS allocation referenced */                       \
    }

DEFINE_KERNEL(k_fma, R16(I_FMA))
DEFINE_KERNEL(k_mul, R16(I_MUL))
DEFINE_KERNEL(k_add, R16(I_ADD))
DEFINE_KERNEL(k_mov, R16(I_MOV))
DEFINE_KERNEL(k_cndmask, R16(I_CND))
DEFINE_KERNEL(k_cndmask_nodep, R16(I_CND_NODEP))
DEFINE_KERNEL(k_cndmask_e64, R16(I_CND_E64))
DEFINE_KERNEL(k_cndmask_ones, R16(I_CND_ONES))
DEFINE_KERNEL(k_max, R16(I_MAXF))
DEFINE_KERNEL(k_bpermute, R16(I_PERM))
DEFINE_KERNEL(k_readlane, R16(I_READLANE))
DEFINE_KERNEL(k_xor, R16(I_XOR))
DEFINE_KERNEL(k_bitop3, R16(I_BITOP3))
DEFINE_KERNEL(k_cvt_f32_u32, R16(I_CVT))
DEFINE_KERNEL(k_log, R16(I_LOG))
DEFINE_KERNEL(k_sqrt, R16(I_SQRT))
DEFINE_KERNEL(k_sin, R16(I_SIN))
DEFINE_KERNEL(k_exp, R16(I_EXP))
DEFINE_KERNEL(k_rcp, R16(I_RCP))
DEFINE_KERNEL(k_mul_lo_u32, R16(I_MULLO))
DEFINE_KERNEL(k_mul_hi_u32, R16(I_MULHI))
DEFINE_KERNEL(k_add_f32_dpp, R16(I_DPPADD))
DEFINE_KERNEL(k_mov_dpp, R16(I_DPPMOV))
DEFINE_KERNEL(k_fmamk, R16(I_FMAMK))
DEFINE_KERNEL(k_pk_mul, P16(I_PKMUL))
DEFINE_KERNEL(k_pk_add, P16(I_PKADD))
DEFINE_KERNEL(k_pk_fma, P16(I_PKFMA))
DEFINE_KERNEL(k_mad_u64_u32, P16(I_MAD64))
DEFINE_KERNEL(k_mad_u64_u32_c, P16(I_MAD64C))
DEFINE_KERNEL(k_add_f64, P16(I_ADDF64))
DEFINE_KERNEL(k_ds_write_b32, R16(I_DSW))
DEFINE_KERNEL(k_ds_read_b32, R16(I_DSR))
DEFINE_KERNEL(k_s_nop, R16(I_SNOP))
DEFINE_KERNEL(k_s_nop7, R16(I_SNOP7))
DEFINE_KERNEL(k_s_nop15, R16(I_SNOP15))
// 16 x (mad64 + bitop3) = 32 instructions per sweep: counted as 64 per BODY pair below (insts_per_body = 32)
DEFINE_KERNEL(k_philox_round, "v_mad_u64_u32 v[32:33], s[40:41], v32, v3, 0\n\tv_mad_u64_u32 v[34:35], s[40:41], v34, v3, 0\n\t"
                              "v_mad_u64_u32 v[36:37], s[40:41], v36, v3, 0\n\tv_mad_u64_u32 v[38:39], s[40:41], v38, v3, 0\n\t"
                              "v_mad_u64_u32 v[40:41], s[40:41], v40, v3, 0\n\tv_mad_u64_u32 v[42:43], s[40:41], v42, v3, 0\n\t"
                              "v_bitop3_b32 v32, v35, v1, v2 bitop3:0x96\n\tv_bitop3_b32 v34, v33, v1, v2 bitop3:0x96\n\t"
                              "v_bitop3_b32 v36, v39, v1, v2 bitop3:0x96\n\tv_bitop3_b32 v38, v37, v1, v2 bitop3:0x96\n\t"
                              "v_bitop3_b32 v40, v43, v1, v2 bitop3:0x96\n\tv_bitop3_b32 v42, v41, v1, v2 bitop3:0x96\n\t"
                              "v_mov_b32 v33, v32\n\tv_mov_b32 v37, v36\n\tv_mov_b32 v41, v40\n\tv_mov_b32 v35, v34\n\t")

struct Op { const char *name; void (*fn)(unsigned long long *, int); int insts_per_body; const char *cls; };

int main(int argc, char **argv)
{
    const Op ops[] = {
        {"v_fma_f32", k_fma, 16, "fp32"}, {"v_mul_f32", k_mul, 16, "fp32"}, {"v_add_f32", k_add, 16, "fp32"},
        {"v_fmamk_f32", k_fmamk, 16, "fp32"}, {"v_mov_b32", k_mov, 16, "move"}, {"v_cndmask_b32", k_cndmask, 16, "move"},
        {"v_cndmask_b32 (sources not the destination)", k_cndmask_nodep, 16, "move"},
        {"v_cndmask_b32_e64 (mask in s[44:45])", k_cndmask_e64, 16, "move"}, {"v_cndmask_b32_e64 (mask all ones)", k_cndmask_ones, 16, "move"},
        {"v_max_f32", k_max, 16, "fp32"}, {"ds_bpermute_b32", k_bpermute, 16, "lds"}, {"v_readlane_b32", k_readlane, 16, "move"},
        {"v_xor_b32", k_xor, 16, "int"}, {"v_bitop3_b32", k_bitop3, 16, "int"}, {"v_cvt_f32_u32", k_cvt_f32_u32, 16, "cvt"},
        {"v_log_f32", k_log, 16, "trans"}, {"v_sqrt_f32", k_sqrt, 16, "trans"}, {"v_sin_f32", k_sin, 16, "trans"},
        {"v_exp_f32", k_exp, 16, "trans"}, {"v_rcp_f32", k_rcp, 16, "trans"},
        {"v_mul_lo_u32", k_mul_lo_u32, 16, "int-mul"}, {"v_mul_hi_u32", k_mul_hi_u32, 16, "int-mul"},
        {"v_mad_u64_u32 (+0)", k_mad_u64_u32, 16, "int-mul"}, {"v_mad_u64_u32 (+v[..])", k_mad_u64_u32_c, 16, "int-mul"},
        {"v_add_f32_dpp quad_perm", k_add_f32_dpp, 16, "dpp"}, {"v_mov_b32_dpp row_mirror", k_mov_dpp, 16, "dpp"},
        {"v_pk_mul_f32", k_pk_mul, 16, "pk-fp32"}, {"v_pk_add_f32", k_pk_add, 16, "pk-fp32"}, {"v_pk_fma_f32", k_pk_fma, 16, "pk-fp32"},
        {"v_add_f64", k_add_f64, 16, "fp64"},
        {"ds_write_b32", k_ds_write_b32, 16, "lds"}, {"ds_read_b32", k_ds_read_b32, 16, "lds"},
        {"s_nop 0", k_s_nop, 16, "scalar"}, {"s_nop 7", k_s_nop7, 16, "scalar"}, {"s_nop 15", k_s_nop15, 16, "scalar"},
        {"philox-like: 6 chains of (v_mad_u64_u32 -> v_bitop3_b32 -> v_mov), dependent", k_philox_round, 16, "mix"},
    };
    const int ITERS = 1024;
    const int Ws[] = {1, 2, 4, 8};
    const char *only = argc > 1 ? argv[1] : nullptr;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    fprintf(stderr, "# device %s, %d CUs; %d x 32 instructions per wave; cyc = s_memtime ticks\n", prop.gcnArchName, ncu, ITERS);
    unsigned long long *d;
    const size_t nmax = (size_t)ncu * 8 * 4 * 5;
    CK(hipMalloc(&d, nmax * 8));
    std::vector<unsigned long long> hbuf(nmax);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("[\n");
    bool first = true;
    for (const Op &op : ops) {
        if (only && !strstr(op.name, only)) continue;
        for (int W : Ws) {
            // ask for W workgroups per CU: LDS so that W fit with 1 KiB to spare each (more than W cannot fit for W <= 4;
            // the grid is ncu*W, which the dispatcher spreads evenly). The per-SIMD grouping below reports what really happened.
            const size_t lds = W == 8 ? 0 : ((size_t)(160 * 1024 / W) - 1024) & ~(size_t)255;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(op.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds ? lds : 1024)));
            const int grid = ncu * W;
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) { // the last repetition is the reported one (clocks ramped)
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(op.fn, dim3(grid), dim3(256), lds, 0, d, ITERS);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
            }
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            const size_t nw = (size_t)grid * 4;
            CK(hipMemcpy(hbuf.data(), d, nw * 40, hipMemcpyDeviceToHost));
            const double ninst = (double)ITERS * 2.0 * op.insts_per_body;
            // group the waves by the SIMD they ran on
            struct Simd { unsigned long long tmin = ~0ull, tmax = 0; int n = 0; };
            std::vector<std::pair<unsigned, size_t>> key(nw);
            for (size_t i = 0; i < nw; ++i) key[i] = {(unsigned)(hbuf[5 * i + 4] & 0xFFF30u) /* xcc | se sh cu | simd */, i};
            std::sort(key.begin(), key.end());
            std::vector<double> csimd, wsimd, cwave(nw), mhz(nw);
            for (size_t a = 0; a < nw;) {
                size_t b = a;
                Simd sd;
                while (b < nw && key[b].first == key[a].first) {
                    const size_t i = key[b].second;
                    sd.tmin = std::min(sd.tmin, hbuf[5 * i]); sd.tmax = std::max(sd.tmax, hbuf[5 * i + 1]); sd.n++;
                    ++b;
                }
                csimd.push_back((double)(sd.tmax - sd.tmin) / (ninst * sd.n));
                wsimd.push_back(sd.n);
                a = b;
            }
            for (size_t i = 0; i < nw; ++i) {
                const double dt = (double)(hbuf[5 * i + 1] - hbuf[5 * i]), dr = (double)(hbuf[5 * i + 3] - hbuf[5 * i + 2]);
                cwave[i] = dt / ninst;
                mhz[i] = dr > 0 ? dt / dr * 100.0 : 0.0;
            }
            std::sort(csimd.begin(), csimd.end());
            std::sort(wsimd.begin(), wsimd.end());
            std::sort(cwave.begin(), cwave.end());
            std::sort(mhz.begin(), mhz.end());
            const size_t ns = csimd.size();
            printf("%s {\"op\": \"%s\", \"class\": \"%s\", \"workgroups_per_cu_asked\": %d, \"simds_seen\": %zu, \"waves_per_simd_median\": %.0f, "
                   "\"waves_per_simd_max\": %.0f, \"cyc_per_inst_per_simd\": %.3f, \"p10\": %.3f, \"p90\": %.3f, "
                   "\"cyc_per_inst_one_wave\": %.3f, \"shader_mhz\": %.0f, \"wall_us\": %.1f}\n",
                   first ? " " : ",", op.name, op.cls, W, ns, wsimd[ns / 2], wsimd[ns - 1], csimd[ns / 2], csimd[ns / 10], csimd[ns * 9 / 10],
                   cwave[nw / 2], mhz[nw / 2], ms * 1e3);
            fflush(stdout);
            first = false;
        }
    }
    printf("]\n");
    return 0;
}
