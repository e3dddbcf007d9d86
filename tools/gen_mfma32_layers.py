#!/usr/bin/env python3
"""Writes mppi-tf_amd/csrc/mppi_mfma32.hip.h: the 32-wide Dense layers of k_rollout_mlp32 / k_rollout_nnauv32 as ONE inline-asm
statement per layer (all v_mfma_f32_32x32x2_f32 of the layer, the wait states their result needs, the 16 relus), on FIXED
physical accumulator registers. Why generated, and why one statement: hipcc does not see MFMAs inside an asm string, so its
hazard recognizer cannot protect a register copy it decides to place between a layer's last MFMA and the relu that reads the
accumulators (it did exactly that after an unrelated header change: v_mov of the accumulators one wait state after the MFMA,
costs off by 1e-4). Inside one statement nothing can be inserted; the sub-registers of the accumulator tuple can only be named
in the string if they are physical registers, hence v[64:79] / v[80:95]."""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mppi-tf_amd", "csrc", "mppi_mfma32.hip.h")


def relu(base):
    return "s_nop 15\\n\\ts_nop 1\\n\\t" + "\\n\\t".join("v_max_f32 v%d, 0, v%d" % (base + i, base + i) for i in range(16))


def layer1(k1h):
    ops = []
    for s in range(k1h):
        c = "%[bias]" if s == 0 else "v[64:79]"
        ops.append("s_nop 1\\n\\tv_mfma_f32_32x32x2_f32 v[64:79], %%[a%d], %%[b%d], %s" % (s, s, c))
    body = "\\n\\t".join(ops) + "\\n\\t" + relu(64)
    ins = ", ".join('[a%d] "v"(a[%d]), [b%d] "v"(b[%d])' % (s, s, s, s) for s in range(k1h))
    return ('    if constexpr (K1H == %d) {\n        asm volatile("%s"\n                     : "=&{v[64:79]}"(acc)\n                     : %s, [bias] "v"(bias));\n    }'
            % (k1h, body, ins))


def hidden(src, dst, pairs=16):
    """pairs = 16: a 32-wide layer; 8: a 16-wide one (units 0..15 are accumulator registers 0..7 of both lane halves; the tile's other
    16 rows carry zero weights and a zero bias)"""
    ops = []
    for s in range(pairs):
        c = "%[bias]" if s == 0 else "v[%d:%d]" % (dst, dst + 15)
        ops.append("s_nop 1\\n\\tv_mfma_f32_32x32x2_f32 v[%d:%d], %%[a%d], v%d, %s" % (dst, dst + 15, s, src + s, c))
    body = "\\n\\t".join(ops) + "\\n\\t" + relu(dst)
    ins = ", ".join('[a%d] "v"(a[%d])' % (s, s) for s in range(pairs))
    return ('    asm volatile("%s"\n                 : "=&{v[%d:%d]}"(out)\n                 : %s, [bias] "v"(bias), "{v[%d:%d]}"(in));'
            % (body, dst, dst + 15, ins, src, src + 15))


def relu2(ba, bb, n):
    """relu of the first n accumulator registers of two column blocks (a 16-wide layer's units live in registers 0..7)"""
    return "s_nop 15\\n\\ts_nop 1\\n\\t" + "\\n\\t".join("v_max_f32 v%d, 0, v%d" % (b + i, b + i) for b in (ba, bb) for i in range(n))


def layer1_x2(k1h, n):
    """layer 1 of TWO column blocks (rollouts 0..31 and 32..63 of the wave) with one set of weights: accA = v[64:79], accB = v[96:111]"""
    ops = []
    for s in range(k1h):
        ca = "%[bias]" if s == 0 else "v[64:79]"
        cb = "%[bias]" if s == 0 else "v[96:111]"
        ops.append("s_nop 1\\n\\tv_mfma_f32_32x32x2_f32 v[64:79], %%[a%d], %%[p%d], %s\\n\\tv_mfma_f32_32x32x2_f32 v[96:111], %%[a%d], %%[q%d], %s" % (s, s, ca, s, s, cb))
    body = "\\n\\t".join(ops) + "\\n\\t" + relu2(64, 96, n)
    ins = ", ".join('[a%d] "v"(a[%d]), [p%d] "v"(ba[%d]), [q%d] "v"(bb[%d])' % (s, s, s, s, s, s) for s in range(k1h))
    return ('    if constexpr (NR == %d) {\n        asm volatile("%s"\n                     : "=&{v[64:79]}"(accA), "=&{v[96:111]}"(accB)\n                     : %s, [bias] "v"(bias));\n    }'
            % (n, body, ins))


def layer1_x2_k(k1h):
    """layer 1 of two column blocks, K1H k pairs, a 32-wide layer (k_rollout_mlp32_pc: the point-mass shapes)"""
    ops = []
    for s in range(k1h):
        ca = "%[bias]" if s == 0 else "v[64:79]"
        cb = "%[bias]" if s == 0 else "v[96:111]"
        ops.append("s_nop 1\\n\\tv_mfma_f32_32x32x2_f32 v[64:79], %%[a%d], %%[p%d], %s\\n\\tv_mfma_f32_32x32x2_f32 v[96:111], %%[a%d], %%[q%d], %s" % (s, s, ca, s, s, cb))
    body = "\\n\\t".join(ops) + "\\n\\t" + relu2(64, 96, 16)
    ins = ", ".join('[a%d] "v"(a[%d]), [p%d] "v"(ba[%d]), [q%d] "v"(bb[%d])' % (s, s, s, s, s, s) for s in range(k1h))
    return ('    if constexpr (K1H == %d) {\n        asm volatile("%s"\n                     : "=&{v[64:79]}"(accA), "=&{v[96:111]}"(accB)\n                     : %s, [bias] "v"(bias));\n    }'
            % (k1h, body, ins))


def hidden_x2(sa, da, sb, db, pairs):
    n = pairs  # a 16-wide layer (8 pairs): 8 live registers; a 32-wide one: 16
    ops = []
    for s in range(pairs):
        ca = "%[bias]" if s == 0 else "v[%d:%d]" % (da, da + 15)
        cb = "%[bias]" if s == 0 else "v[%d:%d]" % (db, db + 15)
        ops.append("s_nop 1\\n\\tv_mfma_f32_32x32x2_f32 v[%d:%d], %%[a%d], v%d, %s\\n\\tv_mfma_f32_32x32x2_f32 v[%d:%d], %%[a%d], v%d, %s"
                   % (da, da + 15, s, sa + s, ca, db, db + 15, s, sb + s, cb))
    body = "\\n\\t".join(ops) + "\\n\\t" + relu2(da, db, n)
    ins = ", ".join('[a%d] "v"(a[%d])' % (s, s) for s in range(pairs))
    return ('    asm volatile("%s"\n                 : "=&{v[%d:%d]}"(outA), "=&{v[%d:%d]}"(outB)\n                 : %s, [bias] "v"(bias), "{v[%d:%d]}"(inA), "{v[%d:%d]}"(inB));'
            % (body, da, da + 15, db, db + 15, ins, sa, sa + 15, sb, sb + 15))


def hidden_x2_fn(name, sa, da, sb, db, pairs):
    return ('__device__ __forceinline__ void %s(f32x16_l &outA, f32x16_l &outB, const f32x16_l &inA, const f32x16_l &inB, const float (&a)[%d], const f32x16_l &bias)\n{\n%s\n}\n'
            % (name, pairs, hidden_x2(sa, da, sb, db, pairs)))


X2 = '''
// ---- two column blocks per wave (k_rollout_nnspeed_pc: lane l = rollout l, 64 rollouts per wave): every weight register feeds two
// MFMAs, block A (rollouts 0..31) on v[64:79] / v[80:95], block B (rollouts 32..63) on v[96:111] / v[112:127]; the B operands of a
// k pair come from ONE v_permlane32_swap of the lane's (even, odd) inputs. NR = the accumulator registers that carry live units
// (16: a 32-wide layer; 8: a 16-wide one, whose units 0..15 are registers 0..7 of the two lane halves).
template <int NR>
__device__ __forceinline__ void mfma32x2_layer1_8(f32x16_l &accA, f32x16_l &accB, const float (&a)[8], const float (&ba)[8], const float (&bb)[8], const f32x16_l &bias)
{
    static_assert(NR == 8 || NR == 16, "live accumulator registers");
''' + layer1_x2(8, 8) + "\n" + layer1_x2(8, 16) + '''
}

// the same with K1H k pairs and a 32-wide layer: the point-mass shapes of k_rollout_mlp32_pc ((s + a + 1) / 2 pairs)
template <int K1H>
__device__ __forceinline__ void mfma32x2_layer1(f32x16_l &accA, f32x16_l &accB, const float (&a)[K1H], const float (&ba)[K1H], const float (&bb)[K1H], const f32x16_l &bias)
{
    static_assert(K1H == 2 || K1H == 3 || K1H == 5 || K1H == 6, "k pairs of layer 1");
''' + "\n".join(layer1_x2_k(k) for k in (2, 3, 5, 6)) + '''
}

''' + hidden_x2_fn("mfma32x2_hidden_lo_hi", 64, 80, 96, 112, 16) + "\n" + hidden_x2_fn("mfma32x2_hidden_hi_lo", 80, 64, 112, 96, 16) + "\n" \
    + hidden_x2_fn("mfma32x2_hidden8_lo_hi", 64, 80, 96, 112, 8) + "\n" + hidden_x2_fn("mfma32x2_hidden8_hi_lo", 80, 64, 112, 96, 8)

text = '''// mppi_mfma32.hip.h — GENERATED by tools/gen_mfma32_layers.py (edit the generator, not this file).
// A 32-wide Dense(relu) layer on the matrix cores as ONE inline-asm statement on fixed physical registers: every
// v_mfma_f32_32x32x2_f32 of the layer (s_nop 1 in front: its B operand may just have been written by a vector instruction), the
// 18 wait states a 16-pass MFMA's result needs before a vector instruction may read it, and the 16 v_max of the relu. hipcc's
// hazard recognizer does not look inside asm strings: with the MFMAs and the relu in separate statements it is free to put a
// register copy of the accumulators between them (it did, after an unrelated change: results off by 1e-4). Nothing can be
// inserted into one statement. Accumulators: layer 1 and every second hidden layer write v[64:79], the others v[80:95]; the
// B operands of a hidden layer are the 16 registers of the previous layer's accumulator, named physically in the string.
// tools/check_mfma_hazards.py scans the built code objects for this hazard class (tests/test_capi_symbols.py runs it).
#pragma once

namespace mppi {

typedef float f32x16_l __attribute__((ext_vector_type(16)));

// layer 1: acc (v[64:79]) = relu(bias + sum_s a[s] (x) b[s]), K1H k pairs
template <int K1H>
__device__ __forceinline__ void mfma32_layer1(f32x16_l &acc, const float (&a)[K1H], const float (&b)[K1H], const f32x16_l &bias)
{
    static_assert(K1H == 2 || K1H == 3 || K1H == 5 || K1H == 6 || K1H == 8, "k pairs of layer 1: (s + a + 1) / 2 of the instantiated shapes");
''' + "\n".join(layer1(k) for k in (2, 3, 5, 6, 8)) + '''
}

// hidden layer reading v[64:79] (the previous layer's relu'd accumulators = its B operands), writing v[80:95]
__device__ __forceinline__ void mfma32_hidden_64_80(f32x16_l &out, const f32x16_l &in, const float (&a)[16], const f32x16_l &bias)
{
''' + hidden(64, 80) + '''
}

// hidden layer reading v[80:95], writing v[64:79]
__device__ __forceinline__ void mfma32_hidden_80_64(f32x16_l &out, const f32x16_l &in, const float (&a)[16], const f32x16_l &bias)
{
''' + hidden(80, 64) + '''
}

// the same for a 16-wide layer (k_rollout_nnspeed32<16>: NNAUVModelSpeed's Dense(16) network): 8 k pairs — the 16 input units are
// accumulator registers 0..7 of the two lane halves
__device__ __forceinline__ void mfma32_hidden8_64_80(f32x16_l &out, const f32x16_l &in, const float (&a)[8], const f32x16_l &bias)
{
''' + hidden(64, 80, 8) + '''
}

__device__ __forceinline__ void mfma32_hidden8_80_64(f32x16_l &out, const f32x16_l &in, const float (&a)[8], const f32x16_l &bias)
{
''' + hidden(80, 64, 8) + '''
}
''' + X2 + '''
} // namespace mppi
'''
open(OUT, "w").write(text)
print("wrote", OUT)
