// ds_write_war.hip — does a VALU write to the DATA registers of a ds_write_b128, issued right after it, corrupt the store?
//
// hipcc pads the corresponding hazard for buffer/flat stores of more than 64 bits (the store reads its data registers
// late); its hazard recogniser has no such rule for DS instructions. k_rollout_mlp_bx3p stores 16-byte bf16 fragments
// with ds_write_b128 and rewrites the data registers in the next instructions (relu/split of the next fragment).
//
// Method: one asm statement per trial: v[44:47] <- pattern A (lane dependent); ds_write_b128 to this lane's 16-byte slot;
// N x s_nop 0; four v_mov_b32 overwrite v[44:47] with pattern B; wait; ds_read_b128 the slot back; compare with A.
// All 8 waves of a 512-thread workgroup do it at once (the LDS queue is contended, as in the kernel), one workgroup
// per CU. Output: per N, wrong dwords in all, and split by 16-lane group of the storing wave.
// Build: hipcc -O2 --offload-arch=gfx950 tools/micro/ds_write_war.hip -o build/ds_write_war
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int N>
__global__ __launch_bounds__(512) void k_dsw(unsigned *out, int reps)
{
    __shared__ __attribute__((aligned(16))) unsigned slots[512 * 4];
    const unsigned addr = threadIdx.x * 16, lane = threadIdx.x & 63;
    const unsigned a0 = 0x10000000u + threadIdx.x * 4;
    unsigned bad[4] = {0, 0, 0, 0};
    if (threadIdx.x == 0) slots[0] = 0; // keep the array
    __syncthreads();
    for (int r = 0; r < reps; ++r) {
        unsigned g0, g1, g2, g3;
        const unsigned a = a0 + (r << 16);
        asm volatile(
            "v_mov_b32 v44, %4\n\tv_add_u32 v45, 1, %4\n\tv_add_u32 v46, 2, %4\n\tv_add_u32 v47, 3, %4\n\tv_mov_b32 v48, %5\n\t"
            "s_nop 4\n\t"
            "ds_write_b128 v48, v[44:47]\n\t"
            ".rept %c6\n\ts_nop 0\n\t.endr\n\t"
            "v_mov_b32 v44, 0x7fffffff\n\tv_mov_b32 v45, 0x7fffffff\n\tv_mov_b32 v46, 0x7fffffff\n\tv_mov_b32 v47, 0x7fffffff\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_read_b128 v[56:59], v48\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mov_b32 %0, v56\n\tv_mov_b32 %1, v57\n\tv_mov_b32 %2, v58\n\tv_mov_b32 %3, v59\n\t"
            : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3)
            : "v"(a), "v"(addr), "i"(N)
            : "v44", "v45", "v46", "v47", "v48", "v56", "v57", "v58", "v59", "memory");
        bad[lane >> 4] += (g0 != a) + (g1 != a + 1) + (g2 != a + 2) + (g3 != a + 3);
    }
    for (int g = 0; g < 4; ++g) if (bad[g]) atomicAdd(out + g, bad[g]);
}

template <int N>
static int run(unsigned *d)
{
    CK(hipMemset(d, 0, 16));
    const int reps = 256;
    hipLaunchKernelGGL(k_dsw<N>, dim3(256), dim3(512), 0, 0, d, reps);
    CK(hipDeviceSynchronize());
    unsigned h[4];
    CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("{\"nops_between_ds_write_b128_and_overwrite\": %d, \"wrong_dwords\": %u, \"by_lane_group_0_15__16_31__32_47__48_63\": [%u, %u, %u, %u], \"dwords_checked\": %.0f}\n",
           N, h[0] + h[1] + h[2] + h[3], h[0], h[1], h[2], h[3], 256.0 * 512 * 4 * reps);
    return 0;
}

int main()
{
    unsigned *d;
    CK(hipMalloc(&d, 16));
    return run<0>(d) || run<1>(d) || run<2>(d) || run<3>(d) || run<4>(d) || run<6>(d) || run<8>(d) || run<16>(d);
}
