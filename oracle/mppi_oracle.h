/*
 * mppi_oracle.h — CPU ORACLE for the MPPI control-step hot path.
 *
 * >>> TEST INFRASTRUCTURE, NOT PRODUCT CODE. <<<
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import, call,
 * link or execute anything under oracle/ — and there only as the checker / the timed CPU
 * baseline, never as the thing measured as the product or shipped.  Nothing under
 * mppi-tf_amd/ or include/ may reference this directory.
 *
 * What it is: a plain-C restatement, written from the reference source read as text, of
 * NicolayP/mppi-tf's per-control-step path (SURVEY.md §8a rows A1–A10 + the build-defined
 * M2 learned step).  Each function cites the reference file:line it follows.
 *
 * How it is pinned: the reference can be neither compiled nor imported in the build
 * container (TensorFlow C++/Python, MuJoCo, gtest absent — SURVEY.md §8c), so the oracle is
 * pinned by the reference's own known-answer vectors, transcribed as data into
 * tests/golden/ (JSON) (test/test_controller.cpp, test_cost.cpp, test_model.cpp,
 * test_utile.cpp, scripts/test.py) and checked by tests/test_oracle_golden.py.
 * NOT pinned by the reference (no reference test or fixture exists): the full H-step
 * recurrence A7 end-to-end, next() A1, the RNG stream A2 ("RNG parity unpinned"), and the
 * learned MLP step M2 ("parity unpinned": the reference has no point-mass MLP).
 *
 * Third-party arithmetic restated here: Philox4x32-10 (Random123, Salmon et al. SC'11;
 * constants and round function as published, the same generator rocRAND 4.2.0 ships in
 * rocrand_philox4x32_10.h) and rocRAND's Box-Muller (rocrand_normal.h: u = 2^-32 + x·2^-32,
 * v = 2π·2^-32 + y·2π·2^-32, (sin v, cos v)·sqrt(-2 ln u)).  TensorFlow's own RandomNormal
 * stream (controller_base.cpp:196-199, seed 1) is NOT reproduced — unverifiable offline.
 */
#ifndef MPPI_ORACLE_H_
#define MPPI_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_S 32
#define ORC_MAX_A 16
#define ORC_MAX_HID 1024
#define ORC_MAX_LAYERS 8

#define ORC_ACTION_COST_CPP 0 /* λ uᵀΣ⁻¹ε                      cost_base.cpp:63-68 */
#define ORC_ACTION_COST_PY 1  /* ½[γ(uᵀΣ⁻¹u+2uᵀΣ⁻¹ε)+λ(1-1/υ)εᵀΣ⁻¹ε]  cost_base.py:114-170 */
#define ORC_STATE_COST_QUADRATIC 0 /* (x-g)ᵀQ(x-g)                  cost_base.cpp:56-61, static_cost.py:40-63 */
#define ORC_STATE_COST_ELLIPSE 1   /* 2D elliptic track               costs/elipse_cost.py:48-85 */
#define ORC_STATE_COST_QUAT 2      /* StaticQuatCost                  costs/static_cost.py:73-159 */
#define ORC_STATE_COST_ELLIPSE3D 3 /* ElipseCost3D                    costs/elipse_cost.py:101-246 */
#define ORC_MODEL_POINT_MASS 0
#define ORC_MODEL_MLP 1
#define ORC_MODEL_AUV 2            /* Fossen AUVModel                 models/auv_model.py:282-562 */
#define ORC_MODEL_NNAUV 3          /* NNAUVModel                      models/nn_model.py:215-304 */
#define ORC_MODEL_NNAUV_SPEED 4    /* NNAUVModelSpeed                 models/nn_model.py:307-588 */

#define SUF(x) x##_f32
#define REAL float
#include "mppi_oracle_decl.inc"
#undef SUF
#undef REAL

#define SUF(x) x##_f64
#define REAL double
#include "mppi_oracle_decl.inc"
#undef SUF
#undef REAL

/* Philox4x32-10 block function: out = Philox(counter, key), 10 rounds. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* rocRAND box_muller(x, y) -> (n0, n1). */
void orc_box_muller(uint32_t x, uint32_t y, float *n0, float *n1);
/* Standard normals z[k, tau, a] of the product's counter layout (groups of 4 horizon steps):
 *   key = seed ; counter = {lo(o), hi(o), lo(gk), hi(gk)} with gk = k_offset + k (the GLOBAL
 *   sample index: results do not depend on how K is sharded) and, for group g = t/4,
 *   o = (step*ceil(tau/4) + g)*a + q, q = 0..a-1 ; the 4 words of block q -> box_muller(x,y),
 *   box_muller(z,w) = normals m = 4q..4q+3 of the group ; normal m is z[t = 4g + m/a][j = m%a].
 *   Identical to rocrand_init(seed, gk, 4*o) + rocrand_normal4(). */
void orc_normals(uint64_t seed, uint64_t step, uint64_t k_offset, int k, int tau, int a, float *z_out);
/* ε = Σ · z per (k,t)  (controller_base.cpp:196-201: BatchMatMulV2(sigma, rng)). */
void orc_noise(uint64_t seed, uint64_t step, uint64_t k_offset, int k, int tau, int a,
               const float *sigma, float *eps_out);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
