/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See mppi_oracle.h. */
#include "mppi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#if defined(_OPENMP)
#include <omp.h>
#endif

#define SUF(x) x##_f32
#define REAL float
#define REAL_EXP(x) expf(x)
#include "mppi_oracle_impl.inc"
#include "mppi_oracle_auv.inc"
#undef SUF
#undef REAL
#undef REAL_EXP

#define SUF(x) x##_f64
#define REAL double
#define REAL_EXP(x) exp(x)
#include "mppi_oracle_impl.inc"
#include "mppi_oracle_auv.inc"
#undef SUF
#undef REAL
#undef REAL_EXP

/* Philox4x32-10, Random123 (Salmon, Moraes, Dror, Shaw, SC'11): multipliers 0xD2511F53 /
 * 0xCD9E8D57, Weyl keys 0x9E3779B9 / 0xBB67AE85, 10 rounds, key bumped between rounds. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0;
        uint32_t hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        if (r != 9) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_box_muller(uint32_t x, uint32_t y, float *n0, float *n1)
{
    const float two_pow32_inv = 2.3283064e-10f;
    const float two_pow32_inv_2pi = 1.46291807e-09f;
    float u = two_pow32_inv + ((float)x * two_pow32_inv);
    float v = two_pow32_inv_2pi + ((float)y * two_pow32_inv_2pi);
    float s = sqrtf(-2.0f * logf(u));
    *n0 = sinf(v) * s;
    *n1 = cosf(v) * s;
}

void orc_normals(uint64_t seed, uint64_t step, uint64_t k_offset, int k, int tau, int a, float *z_out)
{
    const int ng = (tau + 3) / 4; /* groups of 4 horizon steps */
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#if defined(_OPENMP)
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < k; ++i) {
        const uint64_t gk = k_offset + (uint64_t)i;
        for (int g = 0; g < ng; ++g) {
            for (int q = 0; q < a; ++q) {
                const uint64_t o = (step * (uint64_t)ng + (uint64_t)g) * (uint64_t)a + (uint64_t)q;
                const uint32_t ctr[4] = {(uint32_t)o, (uint32_t)(o >> 32), (uint32_t)gk, (uint32_t)(gk >> 32)};
                uint32_t r[4];
                float n[4];
                orc_philox4x32_10(ctr, key, r);
                orc_box_muller(r[0], r[1], &n[0], &n[1]);
                orc_box_muller(r[2], r[3], &n[2], &n[3]);
                for (int w = 0; w < 4; ++w) {
                    const int m = 4 * q + w;          /* normal m of the group */
                    const int t = 4 * g + m / a, j = m % a;
                    if (t < tau) z_out[((size_t)i * tau + t) * a + j] = n[w];
                }
            }
        }
    }
}

void orc_noise(uint64_t seed, uint64_t step, uint64_t k_offset, int k, int tau, int a,
               const float *sigma, float *eps_out)
{
    orc_normals(seed, step, k_offset, k, tau, a, eps_out);
    for (size_t r = 0; r < (size_t)k * tau; ++r) {
        float z[ORC_MAX_A];
        float *e = eps_out + r * a;
        for (int j = 0; j < a; ++j) z[j] = e[j];
        for (int i = 0; i < a; ++i) {
            float acc = 0.0f;
            for (int j = 0; j < a; ++j) acc = acc + sigma[i * a + j] * z[j];
            e[i] = acc;
        }
    }
}

/* thread count of the OpenMP regions that follow (the single-thread rows of the CPU baseline table) */
void orc_set_num_threads(int n)
{
#if defined(_OPENMP)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
