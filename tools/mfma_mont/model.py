#!/usr/bin/env python3
"""Exact integer model of the MFMA-assisted Montgomery product (tools/mfma_mont): a*b on the VALU (radix 2^29 columns),
m = T_lo * N' mod R and m * N as two int8 GEMMs against CONSTANT digit matrices of N' and N (v_mfma_i32_32x32x32_i8).
Checks the arithmetic identities, the value bounds (i32 column sums, 64-bit accumulators, digit ranges) and emits the
constant tables the kernel reads (mont_tables.h).  Test / research infrastructure, not part of the product."""
import random
import sys

W, L = 29, 72
M29 = (1 << W) - 1
R = 1 << (W * L)
Q = int(
    "ffffffffffffffffc90fdaa22168c234c4c6628b80dc1cd129024e088a67cc74020bbea63b139b22514a08798e3404ddef9519b3cd3a43"
    "1b302b0a6df25f14374fe1356d6d51c245e485b576625e7ec6f44c42e9a637ed6b0bff5cb6f406b7edee386bfb5a899fa5ae9f24117c4b"
    "1fe649286651ece45b3dc2007cb8a163bf0598da48361c55d39a69163fa8fd24cf5f83655d23dca3ad961c62f356208552bb9ed5290770"
    "96966d670c354e4abc9804f1746c08ca18217c32905e462e36ce3be39e772c180e86039b2783a2ec07a28fb5c55df06f4c52c9de2bcbf6"
    "955817183995497cea956ae515d2261898fa051015728e5a8aacaa68ffffffffffffffff", 16)
N = Q
NPRIME = (-pow(N, -1, R)) % R
KB = 1 << 18                      # bias unit of the high columns (see DESIGN / kernel comments)


def limbs(z, n):
    return [(z >> (W * k)) & M29 for k in range(n)]


def signed_digits(z, nl):
    """limb-aligned signed digits: z = sum_k 2^(29k) sum_e 2^(8e) d[k][e], d[k][0..2] in [-128,127], d[k][3] in [0,32]"""
    out = []
    for zk in limbs(z, nl):
        d = [(zk >> (8 * e)) & 0xFF for e in range(3)] + [zk >> 24]
        for e in range(3):
            if d[e] >= 128:
                d[e] -= 256
                d[e + 1] += 1
        assert all(-128 <= x <= 127 for x in d) and sum(x << (8 * e) for e, x in enumerate(d)) == zk
        out.append(d)
    return out


# g1[f][dk][e]: digit (limb dk, byte e) of N' << 8f ; g2 the same for N
G1 = [signed_digits(NPRIME << (8 * f), L + 1) for f in range(4)]
G2 = [signed_digits(N << (8 * f), L + 1) for f in range(4)]
assert all(all(x == 0 for x in G2[f][L]) for f in range(4))     # N << 24 still fits 72 limbs


def g1(k, e, i, f):
    return G1[f][k - i][e] if 0 <= k - i and k < L else 0     # (N' << (29 i + 8 f)) mod R: limbs k >= 72 are cut off


def g2(k, e, i, f):
    return G2[f][k - i][e] if 0 <= k - i <= L else 0


def data_digits(word):
    """operand word (already XORed with 0x00808080) -> the four int8 digits the MFMA reads: for bytes 0..2 that is
    (original byte) - 128, for byte 3 the original byte as a signed value"""
    w = word & 0xFFFFFFFF
    b = [(w >> (8 * e)) & 0xFF for e in range(4)]
    return [x - 256 if x >= 128 else x for x in b]


def s32(x):
    assert -(1 << 31) <= x < (1 << 31), x
    return x


def mont_model(a, b, check=True):
    """a, b < 2N -> (a*b*R^-1 mod N as almost-normalised value < 2N-ish, stats)"""
    al, bl = limbs(a, L), limbs(b, L)
    # ---- phase A: T = a*b in radix 2^29 columns (the VALU part; modelled as plain integers + the bias)
    T = a * b
    x = limbs(T % R, L)                                   # T_lo: 72 exact limbs (emitted by the retire steps)
    Thi = T >> (W * L)
    # high columns as the kernel leaves them: column sums are not unique, only their total matters for the model
    bias = [(KB << W) if p == 0 else ((KB << W) - KB if p <= 70 else 0) for p in range(L)]
    # ---- GEMM 1: column sums S1[k][e] = sum_{i,f} g1 * X[i][f]  (X unsigned bytes; MFMA runs on X-128 with C-init)
    X = [[(xi >> (8 * f)) & 0xFF for f in range(3)] + [xi >> 24] for xi in x]
    V1 = []
    for k in range(L):
        S = []
        for e in range(4):
            mf = sum(g1(k, e, i, f) * (X[i][f] - (128 if f < 3 else 0)) for i in range(k + 1) for f in range(4))
            cinit = 128 * sum(g1(k, e, i, f) for i in range(k + 1) for f in range(3))
            S.append(s32(mf) + s32(cinit))
            s32(S[-1])
        p01, p23 = s32(S[0] + (S[1] << 8)), s32(S[2] + (S[3] << 8))
        V1.append((p23 << 16) + p01)
    assert sum(v << (W * k) for k, v in enumerate(V1)) % R == (T * NPRIME) % R
    v = []
    for k in range(L):
        vk = (V1[k] & M29) + ((V1[k - 1] >> W) if k else 0)      # arithmetic shift (Python's)
        assert -(1 << 31) <= vk < (1 << 31)
        v.append(vk)
    m2 = sum(vk << (W * k) for k, vk in enumerate(v))
    assert m2 % R == (T * NPRIME) % R and 0 <= m2 < R + (R >> 8)
    D = [data_digits(vk ^ 0x00808080) for vk in v]
    for k in range(L):
        assert sum((D[k][f] + (128 if f < 3 else 0)) << (8 * f) for f in range(4)) == v[k]
    # ---- GEMM 2: rows for the absolute limbs 71 (guard) and 72..142
    def V2(k):
        S = []
        for e in range(4):
            mf = sum(g2(k, e, i, f) * D[i][f] for i in range(L) for f in range(4))
            cinit = 128 * sum(g2(k, e, i, f) for i in range(L) for f in range(3))
            S.append(s32(s32(mf) + s32(cinit)))
        p01, p23 = s32(S[0] + (S[1] << 8)), s32(S[2] + (S[3] << 8))
        return (p23 << 16) + p01
    full = m2 * N
    guard = V2(L - 1)
    c = (x[L - 1] + guard + (1 << 28)) >> W
    if check:       # the exact carry of the low COLUMNS (not of the low limbs of the product: the columns are unnormalised)
        low = (T % R) + sum(V2(k) << (W * k) for k in range(L))
        assert low % R == 0 and c == low >> (W * L), (c, low >> (W * L))
    acc = Thi + c
    tot = 0
    for p in range(71):
        vp = V2(L + p)
        assert abs(vp) < (1 << 47)
        tot += vp << (W * p)
    assert V2(L + 71) == 0
    res = acc + tot
    assert res == (T + full) >> (W * L) and (T + full) % R == 0, "reduction identity"
    assert res < 2 * N and (res * R - a * b) % N == 0
    return res


if __name__ == "__main__":
    rng = random.Random(1)
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        a, b = rng.randrange(2 * N), rng.randrange(2 * N)
        if it == 0:
            a = b = 2 * N - 1
        r = mont_model(a, b)
        print("ok", it, r.bit_length())


# ---------------------------------------------------------------------------------------------------------------
# Tables for the kernel (mpvss_rs_amd/csrc/modp_mfma_tables.h).  MFMA v_mfma_i32_32x32x32_i8, D = A*B + C:
#   A (constant digit matrix): lane l holds A[row = l & 31][k = 16 (l >> 5) + jj], jj = 0..15   (16 bytes)
#   B (data):                  lane l holds B[k = 16 (l >> 5) + jj][col = l & 31]                (col = the number)
#   C/D:                       lane l, register reg: row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), col = l & 31
# Row / k conventions of both GEMMs (h = lane half of the RESULT, g = reg & 3, e = reg >> 2; h' = half of the k index):
#   k index 16 h' + 4 j' + f   <->  data limb i = 8 C + 4 h' + j', byte f          (K-block C = 0..8)
#   row r = g + 8 e + 4 h of tile R:
#     GEMM 1: limb k = 8 R + 4 h + g of m, byte e          (so that the results ARE the next operand)
#     GEMM 2: result limb rho = 36 h + 4 R + g (absolute limb 72 + rho), byte e;
#             (R, h, g) = (8, 1, 3), i.e. rho = 71, is the guard limb: absolute limb 71.
# Both matrices are Toeplitz in the limb index: entry (row, k) = digit e of limb (k_row - i) of (N' or N) << 8 f.  So a lane's
# 16 A-operand bytes of ANY tile are one 16-byte record GT[e][x], x = k_row - (8 C + 4 h'):  bytes [4 j' + f] =
# digit(f, x - j', e) -- 5 KB (GEMM 1) + 9.5 KB (GEMM 2, zero-padded so that no lane ever needs a clamp) instead of
# 36 one-kilobyte tiles.  Lanes 0..7 of a wave (g = 0..3, h = 0..1, e = 0) read 8 consecutive records: conflict-free.
GT1_X = 81          # records per e of GEMM 1: x = -4 .. 76
GT2_X = 148         # records per e of GEMM 2: x = -4 .. 143


def gv(G, f, d, e):
    return G[f][d][e] if 0 <= d <= L else 0


def gt_bytes(G, nx):
    out = bytearray()
    for e in range(4):
        for xi in range(nx):
            x = xi - 4
            for jp in range(4):
                for f in range(4):
                    out.append(gv(G, f, x - jp, e) & 0xFF)
    return bytes(out)


def row_geh(row):
    g = row & 3
    h = (row >> 2) & 1
    e = row >> 3
    return g, e, h


def gemm2_row_limb(R, h, g):
    rho = 36 * h + 4 * R + g
    return L - 1 if rho == 71 else L + rho


def emit(path):
    gt1, gt2 = gt_bytes(G1, GT1_X), gt_bytes(G2, GT2_X)
    # the record addressing of the kernel against the matrix definitions g1 / g2, every tile, every lane
    skip2 = [[1] * 9 for _ in range(9)]
    for R in range(9):
        for C in range(9):
            for lane in range(64):
                row, hp = lane & 31, lane >> 5
                g, e, h = row_geh(row)
                if C <= R:
                    xi = 4 * h + g - 4 * hp + 4 + 8 * (R - C)
                    assert 0 <= xi < GT1_X
                    rec = gt1[(e * GT1_X + xi) * 16:(e * GT1_X + xi) * 16 + 16]
                    for jp in range(4):
                        for f in range(4):
                            want = g1(8 * R + 4 * h + g, e, 8 * C + 4 * hp + jp, f)
                            assert rec[4 * jp + f] == want & 0xFF
                k = gemm2_row_limb(R, h, g)
                xi = k - 8 * C - 4 * hp + 4
                assert 0 <= xi < GT2_X
                rec = gt2[(e * GT2_X + xi) * 16:(e * GT2_X + xi) * 16 + 16]
                for jp in range(4):
                    for f in range(4):
                        want = g2(k, e, 8 * C + 4 * hp + jp, f)
                        assert rec[4 * jp + f] == want & 0xFF
                        if want:
                            skip2[R][C] = 0
    def cinit(rowlimb, gfun):
        out = []
        for R in range(9):
            for h in range(2):
                for reg in range(16):
                    g, e = reg & 3, reg >> 2
                    k = rowlimb(R, h, g)
                    out.append(128 * sum(gfun(k, e, i, f) for i in range(L) for f in range(3)))
        return out
    c1 = cinit(lambda R, h, g: 8 * R + 4 * h + g, g1)
    c2 = cinit(gemm2_row_limb, g2)
    assert all(-(1 << 31) <= v < (1 << 31) for v in c1 + c2)
    n_mfma1 = sum(R + 1 for R in range(9))
    n_mfma2 = sum(1 for R in range(9) for C in range(9) if not skip2[R][C])
    with open(path, "w") as f:
        f.write("// GENERATED by tools/mfma_mont/model.py -- constant digit records of N' and N for the int8 MFMA reduction (bn_pair.h).\n")
        f.write("#pragma once\n#include <stdint.h>\n")
        f.write(f"#define MM_GT1_X {GT1_X}\n#define MM_GT2_X {GT2_X}\n#define MM_MFMA1 {n_mfma1}\n#define MM_MFMA2 {n_mfma2}\n")
        def arr(name, data, ctype="uint8_t"):
            f.write(f"static const {ctype} {name}[{len(data)}] = {{\n")
            for o in range(0, len(data), 32):
                f.write("  " + ",".join(str(v) for v in data[o:o + 32]) + ",\n")
            f.write("};\n")
        arr("MM_GT1", gt1)
        arr("MM_GT2", gt2)
        arr("MM_C1", c1, "int32_t")
        arr("MM_C2", c2, "int32_t")
        f.write("static constexpr int8_t MM_SKIP2[81] = {" + ",".join(str(v) for row in skip2 for v in row) + "};\n")
    print(f"wrote {path}: {len(gt1)} + {len(gt2)} bytes of records, {n_mfma1} + {n_mfma2} MFMAs per product")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "emit":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    emit(os.path.join(here, "..", "..", "mpvss_rs_amd", "csrc", "modp_mfma_tables.h"))
