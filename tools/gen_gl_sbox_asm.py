#!/usr/bin/env python3
"""Generates blobstreamx_amd/csrc/goldilocks_sbox_asm.h: the Poseidon full-round S-box of THREE state words as ONE hand-scheduled
gfx950 asm block — s_k <- (s_k + c_k)^7 mod p, k = 0..2 (hand-written body for gl_mul x 12 + the S-box chain).

Why one block and why generated: inline asm cannot name the halves of a 64-bit operand, so a body written as several asm statements
has to end wherever a half of a multiply-add result is read on its own, and the compiler pads every such boundary with s_nop and
copies halves it cannot prove adjacent (round-5 first pass: 16 boundaries and ~90 s_nop + ~90 v_mov per full round).  Here the
64-bit temporaries live in FIXED registers v[BASE .. BASE + 29] (declared as clobbers), so every half has a name and the three
independent chains are interleaved instruction by instruction through the whole x -> x + c -> x^2 -> x^3, x^4 -> x^7 sequence:

  * a carry (SGPR pair) written by one VALU instruction is read no sooner than the third instruction behind it (the gfx940
    VALU-writes-SGPR -> VALU-reads hazard, 2 wait states): no s_nop;
  * 64 x 64 -> 128: P = a0 b0, R = a0 b1, H = a1 b1, R += a1 b0 (carry K kept), then ONE carry chain over the halves
    (hi(P) += lo(R); lo(H) += hi(R) + c; hi(H) += c; hi(H) += K) — no zero-extended addends, no moves;
  * reduction: lo - hi_hi with the borrow's EPS given back (5 instructions), hi_lo * EPS + t as one v_mad_u64_u32 with its own
    carry-out, the carry's EPS added (3): 22 issue slots per multiplication (5 half-rate multiply-adds = 10, 12 others);
  * the round constant is added in the same block (6 slots; gfx9 allows one scalar source per VALU instruction, so the constant's
    upper half goes through a v_mov).

usage: python tools/gen_gl_sbox_asm.py   (writes the header; tests/test_oracle_poseidon.py checks that the committed file is current)
"""
import os
import sys

BASE = int(os.environ.get("BSX_SBOX_BASE", "98"))          # fixed VGPRs (BSX_SBOX_BASE: occupancy experiments, tools/exp_poseidon_occ.sh) v[BASE .. BASE + 29]; kernels that use the block stay at <= 128 VGPRs (4 waves per SIMD)


def chain_regs(k):
    b = BASE + 10 * k
    return {"P": b, "R": b + 2, "H": b + 4, "X2": b + 6, "X3": b + 8}


def pair(r):
    return "v[%d:%d]" % (r, r + 1)


def v(r):
    return "v%d" % r


def mul(rg, a, b, out, K, C):
    """a, b, out: (lo, hi) register / operand names; rg: the chain's fixed registers; K, C: SGPR-pair operand names"""
    P, R, H = rg["P"], rg["R"], rg["H"]
    return [
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (pair(P), a[0], b[0]),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (pair(R), a[0], b[1]),
        "v_mad_u64_u32 %s, vcc, %s, %s, 0" % (pair(H), a[1], b[1]),
        "v_mad_u64_u32 %s, %s, %s, %s, %s" % (pair(R), K, a[1], b[0], pair(R)),
        "v_add_co_u32 %s, %s, %s, %s" % (v(P + 1), C, v(P + 1), v(R)),
        "v_addc_co_u32 %s, %s, %s, %s, %s" % (v(H), C, v(H), v(R + 1), C),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (v(H + 1), C, v(H + 1), C),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (v(H + 1), C, v(H + 1), K),
        "v_sub_co_u32 %s, %s, %s, %s" % (v(P), C, v(P), v(H + 1)),
        "v_subbrev_co_u32 %s, %s, 0, %s, %s" % (v(P + 1), C, v(P + 1), C),
        "v_cndmask_b32 %s, 0, -1, %s" % (v(R), C),
        "v_sub_co_u32 %s, %s, %s, %s" % (v(P), C, v(P), v(R)),
        "v_subbrev_co_u32 %s, %s, 0, %s, %s" % (v(P + 1), C, v(P + 1), C),
        "v_mad_u64_u32 %s, %s, %s, -1, %s" % (pair(P), C, v(H), pair(P)),
        "v_cndmask_b32 %s, 0, -1, %s" % (v(R), C),
        "v_add_co_u32 %s, %s, %s, %s" % (out[0], C, v(P), v(R)),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (out[1], C, v(P + 1), C),
    ]


def add_const(rg, x, c, out, C):
    H = rg["H"]
    return [
        "v_add_co_u32 %s, %s, %s, %s" % (out[0], C, c[0], x[0]),
        "v_mov_b32 %s, %s" % (v(H), c[1]),
        "v_addc_co_u32 %s, %s, %s, %s, %s" % (out[1], C, v(H), x[1], C),
        "v_cndmask_b32 %s, 0, -1, %s" % (v(H), C),
        "v_add_co_u32 %s, %s, %s, %s" % (out[0], C, out[0], v(H)),
        "v_addc_co_u32 %s, %s, 0, %s, %s" % (out[1], C, out[1], C),
    ]


def chain(k):
    """operand numbering: outputs 0..5 = (lo, hi) of chain 0, 1, 2; 6..8 = K of chain k; 9..11 = C; inputs 12..17 = x (lo, hi);
    18..23 = constant (lo, hi) in SGPRs"""
    rg = chain_regs(k)
    out = ("%%%d" % (2 * k), "%%%d" % (2 * k + 1))
    K, C = "%%%d" % (6 + k), "%%%d" % (9 + k)
    x = ("%%%d" % (12 + 2 * k), "%%%d" % (13 + 2 * k))
    c = ("%%%d" % (18 + 2 * k), "%%%d" % (19 + 2 * k))
    x2 = (v(rg["X2"]), v(rg["X2"] + 1))
    x3 = (v(rg["X3"]), v(rg["X3"] + 1))
    seq = add_const(rg, x, c, out, C)              # xc = x + c lives in the output registers until the last multiplication
    seq += mul(rg, out, out, x2, K, C)             # x^2
    seq += mul(rg, x2, out, x3, K, C)              # x^3
    seq += mul(rg, x2, x2, x2, K, C)               # x^4 over x^2 (its operands are read by the first four instructions only)
    seq += mul(rg, x3, x2, out, K, C)              # x^7
    return seq


def main():
    chains = [chain(k) for k in range(3)]
    n = len(chains[0])
    lines = []
    for i in range(n):
        for k in range(3):
            lines.append(chains[k][i])
    clob = ", ".join('"v%d"' % r for r in range(BASE, BASE + 30))
    out = os.environ.get("BSX_SBOX_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "blobstreamx_amd", "csrc", "goldilocks_sbox_asm.h")
    with open(out, "w") as f:
        f.write("// goldilocks_sbox_asm.h — GENERATED by tools/gen_gl_sbox_asm.py; do not edit.\n"
                "// s_k <- (s_k + c_k)^7 mod p for three state words as one hand-scheduled gfx950 asm block: three independent chains\n"
                "// interleaved instruction by instruction (%d instructions, %d issue slots per word: 6 for the constant, 4 x 22 for the\n"
                "// multiplications; no s_nop, no v_mov besides the constant's upper half).  64-bit temporaries in the fixed registers\n"
                "// v[%d:%d] (clobbers).  See the generator for the derivation; device only.\n"
                "#pragma once\n#ifndef BSX_GL_SBOX_ASM_H   // (a variant header given with -include wins: tools/exp_poseidon_occ.sh)\n#define BSX_GL_SBOX_ASM_H\n#include <stdint.h>\n\n#if defined(__HIP_DEVICE_COMPILE__)\n#define BSX_GL_SBOX3_ASM 1\nnamespace bsx {\n\n"
                % (3 * n, 6 + 4 * 22, BASE, BASE + 29))
        f.write("__device__ __forceinline__ void gl_sbox3(uint64_t& s0, uint64_t& s1, uint64_t& s2, uint64_t c0, uint64_t c1, uint64_t c2) {\n"
                "    uint32_t o00, o01, o10, o11, o20, o21;\n"
                "    uint64_t K0, K1, K2, C0, C1, C2;\n"
                "    asm volatile(\n")
        for l in lines:
            f.write('        "%s\\n\\t"\n' % l)
        f.write('        : "=&v"(o00), "=&v"(o01), "=&v"(o10), "=&v"(o11), "=&v"(o20), "=&v"(o21),\n'
                '          "=&s"(K0), "=&s"(K1), "=&s"(K2), "=&s"(C0), "=&s"(C1), "=&s"(C2)\n'
                '        : "v"((uint32_t)s0), "v"((uint32_t)(s0 >> 32)), "v"((uint32_t)s1), "v"((uint32_t)(s1 >> 32)), "v"((uint32_t)s2), "v"((uint32_t)(s2 >> 32)),\n'
                '          "s"((uint32_t)c0), "s"((uint32_t)(c0 >> 32)), "s"((uint32_t)c1), "s"((uint32_t)(c1 >> 32)), "s"((uint32_t)c2), "s"((uint32_t)(c2 >> 32))\n'
                '        : "vcc", %s);\n' % clob)
        f.write("    s0 = (uint64_t)o00 | ((uint64_t)o01 << 32);\n    s1 = (uint64_t)o10 | ((uint64_t)o11 << 32);\n    s2 = (uint64_t)o20 | ((uint64_t)o21 << 32);\n}\n\n"
                "}  // namespace bsx\n#endif\n#endif  // BSX_GL_SBOX_ASM_H\n")
    print("wrote", os.path.normpath(out), "-", 3 * n, "instructions")


if __name__ == "__main__":
    sys.exit(main())
