// mfma_src_war.hip — can a write to an MFMA's SOURCE registers, issued shortly after the MFMA, corrupt the MFMA?
//
// Question behind it: k_rollout_mlp_bx3p produced wrong rollouts at random (lanes 16-31 of a
// 32-rollout column block, ~1e-5 relative), and its ISA shows, right after the last v_mfma_f32_32x32x16_bf16 of a
// k-block, (a) `ds_read_b128` into that MFMA's B registers (the next k-block's fragment, every wave), and (b) in the
// chain wave's conditional pieces `v_pk_add_f32` into them. Neither is padded by hipcc. Is either a real
// write-after-read hazard on gfx950, where the MFMAs of a k-block form a dependent accumulator chain and the matrix
// pipe is shared with a partner wave, so that an MFMA may sit issued-but-not-started for a while?
//
// Method: one inline-asm statement on fixed registers. A chain of L dependent MFMAs on one accumulator
//   D = A·B1 (+ A·B2 + ... ), B_i = v[44:47] (the LAST one) or copies in v[52:55].. for the earlier ones,
// then N x `s_nop 0`, then an overwrite of the last MFMA's B operand v[44:47] with different data —
//   KIND 0: four v_mov_b32;   KIND 1: one ds_read_b128 from LDS (asynchronous register write on data return);
//   KIND 2: no overwrite, instead D itself is READ (16 x v_mov_b32) N wait states after the last MFMA: the distance
//           hipcc pads to 12 wait states for this MFMA (s_nop 10 + one instruction, seen in the kernel's ISA)
// — then a long drain. The result is compared with the same statement at N = 64. `hammer`: the partner wave of every
// SIMD (waves 4-7 of the 512-thread workgroup) issues back-to-back MFMAs meanwhile.
// Output: per (kind, chain length, hammer, N) the number of D values that differ from the reference.
//
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/mfma_src_war.hip -o build/mfma_src_war
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int N, int KIND, int L>
__device__ __forceinline__ void chain_then_overwrite(const unsigned (&a)[4], const unsigned (&b)[4], unsigned lds_addr, float (&d)[16])
{
    asm volatile(
        "v_mov_b32 v40, %16\n\tv_mov_b32 v41, %17\n\tv_mov_b32 v42, %18\n\tv_mov_b32 v43, %19\n\t"
        "v_mov_b32 v44, %20\n\tv_mov_b32 v45, %21\n\tv_mov_b32 v46, %22\n\tv_mov_b32 v47, %23\n\t"
        "v_mov_b32 v52, %20\n\tv_mov_b32 v53, %21\n\tv_mov_b32 v54, %22\n\tv_mov_b32 v55, %23\n\t"
        "v_mov_b32 v48, %24\n\t"
        "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\t"
        "v_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\tv_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\t"
        "v_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
        "s_nop 7\n\t"
        ".rept %c27 - 1\n\t"
        "v_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[52:55], v[0:15]\n\t"
        ".endr\n\t"
        "v_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]\n\t"
        ".rept %c25\n\ts_nop 0\n\t.endr\n\t"
        ".if %c26 == 0\n\t"
        "v_mov_b32 v44, 0x3f803f80\n\tv_mov_b32 v45, 0x3f803f80\n\tv_mov_b32 v46, 0x3f803f80\n\tv_mov_b32 v47, 0x3f803f80\n\t"
        ".endif\n\t"
        ".if %c26 == 1\n\t"
        "ds_read_b128 v[44:47], v48\n\t"
        ".endif\n\t"
        ".if %c26 == 2\n\t" // early READ of D: copy it out right here (the copies at the end then return these)
        "v_mov_b32 v56, v0\n\tv_mov_b32 v57, v1\n\tv_mov_b32 v58, v2\n\tv_mov_b32 v59, v3\n\tv_mov_b32 v60, v4\n\tv_mov_b32 v61, v5\n\t"
        "v_mov_b32 v62, v6\n\tv_mov_b32 v63, v7\n\tv_mov_b32 v64, v8\n\tv_mov_b32 v65, v9\n\tv_mov_b32 v66, v10\n\tv_mov_b32 v67, v11\n\t"
        "v_mov_b32 v68, v12\n\tv_mov_b32 v69, v13\n\tv_mov_b32 v70, v14\n\tv_mov_b32 v71, v15\n\t"
        ".endif\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        ".if %c26 == 2\n\t"
        "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\tv_mov_b32 %4, v60\n\tv_mov_b32 %5, v61\n\t"
        "v_mov_b32 %6, v62\n\tv_mov_b32 %7, v63\n\tv_mov_b32 %8, v64\n\tv_mov_b32 %9, v65\n\tv_mov_b32 %10, v66\n\tv_mov_b32 %11, v67\n\t"
        "v_mov_b32 %12, v68\n\tv_mov_b32 %13, v69\n\tv_mov_b32 %14, v70\n\tv_mov_b32 %15, v71\n\t"
        ".else\n\t"
        "v_mov_b32 %0, v0\n\tv_mov_b32 %1, v1\n\tv_mov_b32 %2, v2\n\tv_mov_b32 %3, v3\n\tv_mov_b32 %4, v4\n\tv_mov_b32 %5, v5\n\t"
        "v_mov_b32 %6, v6\n\tv_mov_b32 %7, v7\n\tv_mov_b32 %8, v8\n\tv_mov_b32 %9, v9\n\tv_mov_b32 %10, v10\n\tv_mov_b32 %11, v11\n\t"
        "v_mov_b32 %12, v12\n\tv_mov_b32 %13, v13\n\tv_mov_b32 %14, v14\n\tv_mov_b32 %15, v15\n\t"
        ".endif\n\t"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]),
          "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]), "=&v"(d[14]), "=&v"(d[15])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(lds_addr), "i"(N), "i"(KIND), "i"(L)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v40", "v41", "v42",
          "v43", "v44", "v45", "v46", "v47", "v48", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "memory");
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int KIND, int L>
__global__ __launch_bounds__(512) void k_war(float *out, int hammer, int reps)
{
    __shared__ __attribute__((aligned(16))) unsigned other[256 * 4]; // what the ds_read overwrites B with
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 256) for (int i = 0; i < 4; ++i) other[threadIdx.x * 4 + i] = 0x3f803f80u; // bf16 (1.0, 1.0)
    __syncthreads();
    if (wave >= 4) { // partner waves: keep the matrix pipe of their SIMD busy (or idle, hammer = 0)
        if (!hammer) return;
        f32x16 acc = {0};
        bf16x8 x, y;
        for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (lane + i)); y[i] = (__bf16)(0.002f * (lane - i)); }
        for (int r = 0; r < reps * 12 * L; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0);
        if (acc[0] == 12345.678f) out[0] = acc[3]; // keep it alive
        return;
    }
    unsigned a[4], b[4];
    for (int i = 0; i < 4; ++i) { // two bf16 per register, values that make every product distinct
        const __bf16 a0 = (__bf16)(1.0f + 0.0078125f * ((lane * 7 + i * 3) & 63)), a1 = (__bf16)(1.0f + 0.0078125f * ((lane * 5 + i) & 63));
        const __bf16 b0 = (__bf16)(0.5f + 0.0078125f * ((lane * 3 + i * 5) & 63)), b1 = (__bf16)(0.5f + 0.0078125f * ((lane + i * 11) & 63));
        a[i] = (unsigned)__builtin_bit_cast(unsigned short, a0) | ((unsigned)__builtin_bit_cast(unsigned short, a1) << 16);
        b[i] = (unsigned)__builtin_bit_cast(unsigned short, b0) | ((unsigned)__builtin_bit_cast(unsigned short, b1) << 16);
    }
    const unsigned lds_addr = threadIdx.x * 16; // LDS byte address: `other` is this kernel's only LDS object, at offset 0
    float ref[16], d[16];
    chain_then_overwrite<64, KIND, L>(a, b, lds_addr, ref);
    int bad = 0;
    for (int r = 0; r < reps; ++r) {
        chain_then_overwrite<N, KIND, L>(a, b, lds_addr, d);
        for (int i = 0; i < 16; ++i) bad += d[i] != ref[i];
    }
    atomicAdd(out + 1, (float)bad);
}

template <int N, int KIND, int L>
static int run(float *d_out, int hammer)
{
    CK(hipMemset(d_out, 0, 8));
    const int reps = 64;
    hipLaunchKernelGGL((k_war<N, KIND, L>), dim3(512), dim3(512), 0, 0, d_out, hammer, reps);
    CK(hipDeviceSynchronize());
    float h[2];
    CK(hipMemcpy(h, d_out, 8, hipMemcpyDeviceToHost));
    printf("{\"overwrite\": \"%s\", \"dependent_mfma_chain\": %d, \"partner_wave_hammers_mfma\": %d, \"nops_between\": %d, \"wrong_values\": %.0f, \"values_checked\": %.0f}\n",
           KIND == 2 ? "none; D of the last MFMA is READ by 16 x v_mov_b32 (hipcc leaves 12 wait states here)"
                     : KIND ? "ds_read_b128 into B of the last MFMA" : "4 x v_mov_b32 into B of the last MFMA", L, hammer, N, h[1], 512.0 * 4 * 64 * 16 * reps);
    fflush(stdout);
    return 0;
}

template <int KIND, int L>
static int sweep(float *d_out)
{
    for (int hammer = 0; hammer < 2; ++hammer) {
        if (run<0, KIND, L>(d_out, hammer) || run<1, KIND, L>(d_out, hammer) || run<2, KIND, L>(d_out, hammer) ||
            run<4, KIND, L>(d_out, hammer) || run<8, KIND, L>(d_out, hammer) || run<16, KIND, L>(d_out, hammer) ||
            run<32, KIND, L>(d_out, hammer))
            return 1;
    }
    return 0;
}

int main()
{
    float *d_out;
    CK(hipMalloc(&d_out, 8));
    if (sweep<0, 1>(d_out) || sweep<0, 2>(d_out) || sweep<0, 6>(d_out) || sweep<1, 1>(d_out) || sweep<1, 2>(d_out) || sweep<1, 6>(d_out)) return 1;
    for (int hammer = 0; hammer < 2; ++hammer) // early read of D: finer sweep around hipcc's 12 wait states
        if (run<0, 2, 3>(d_out, hammer) || run<4, 2, 3>(d_out, hammer) || run<8, 2, 3>(d_out, hammer) || run<10, 2, 3>(d_out, hammer) ||
            run<11, 2, 3>(d_out, hammer) || run<12, 2, 3>(d_out, hammer) || run<13, 2, 3>(d_out, hammer) || run<14, 2, 3>(d_out, hammer) ||
            run<16, 2, 3>(d_out, hammer) || run<20, 2, 3>(d_out, hammer) || run<24, 2, 3>(d_out, hammer) || run<32, 2, 3>(d_out, hammer))
            return 1;
    return 0;
}
