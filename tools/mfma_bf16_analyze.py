#!/usr/bin/env python3
"""Offline analysis of tools/mfma_probe_bf16.hip's dump: which fp32 operation sequence reproduces
D = mfma_f32_4x4x4_bf16(A, B, C) on gfx950 bit for bit.  Exact rational arithmetic on the host.

  python tools/mfma_bf16_analyze.py gpurun_out/mfma_bf16_cases.bin
"""
import struct
import sys
from fractions import Fraction

import numpy as np


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def rn32(x):
    """Round a Fraction to the nearest fp32 (ties to even); inputs here stay in the normal range."""
    if x == 0:
        return np.float32(0.0)
    return np.float32(float(x)) if abs(x) < 2 ** 52 and Fraction(float(x)) == x else _rn32_slow(x)


def _rn32_slow(x, mode="rne"):
    s = -1 if x < 0 else 1
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    q = a / Fraction(2) ** (e - 23)          # in [2^23, 2^24)
    n = q.numerator // q.denominator
    rem = q - n
    if mode == "rne":
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
            n += 1
    elif mode == "rz":
        pass
    return np.float32(s * float(Fraction(n) * Fraction(2) ** (e - 23)))


def rnd(x, mode="rne"):
    if x == 0:
        return np.float32(0.0)
    return _rn32_slow(x, mode)


def F(v):
    return Fraction(float(v))


def main():
    path = sys.argv[1]
    with open(path, "rb") as f:
        T, W = struct.unpack("ii", f.read(8))
        n = T * W * 4
        a = np.frombuffer(f.read(n * 2), np.uint16).reshape(T, W, 4)
        b = np.frombuffer(f.read(n * 2), np.uint16).reshape(T, W, 4)
        c = np.frombuffer(f.read(n * 4), np.float32).reshape(T, W, 4)
        d0 = np.frombuffer(f.read(n * 4), np.float32).reshape(T, W, 4)
        d4 = np.frombuffer(f.read(n * 4), np.float32).reshape(T, W, 4)
    af, bf = bf16_to_f32(a), bf16_to_f32(b)

    # layout hypothesis: block blk = lane // 4; A[i][k] = a[lane 4 blk + i][k]; B[k][j] = b[lane 4 blk + j][k];
    # D[i][j] = d[lane 4 blk + j][reg i]; C alike.  BLGP = 4: B comes from block 0's lanes for every block
    def case(t, blk, i, j, blgp):
        la, lb = 4 * blk + i, 4 * (0 if blgp == 4 else blk) + j
        ld = 4 * blk + j
        return ([F(af[t, la, k]) for k in range(4)], [F(bf[t, lb, k]) for k in range(4)], F(c[t, ld, i]),
                (d4 if blgp == 4 else d0)[t, ld, i])

    hyps = {
        "chain k=0..3 from c (fma)": lambda A, B, C: chain(A, B, C, range(4)),
        "chain k=3..0 from c (fma)": lambda A, B, C: chain(A, B, C, range(3, -1, -1)),
        "exact sum, one RNE": lambda A, B, C: rnd(C + sum(x * y for x, y in zip(A, B))),
        "exact sum, one RZ": lambda A, B, C: rnd(C + sum(x * y for x, y in zip(A, B)), "rz"),
        "RNE(c + RNE(sum products))": lambda A, B, C: rnd(C + F(rnd(sum(x * y for x, y in zip(A, B))))),
        "pairs: ((p0+p1)+(p2+p3))+c": lambda A, B, C: rnd(
            F(rnd(F(rnd(A[0] * B[0] + A[1] * B[1])) + F(rnd(A[2] * B[2] + A[3] * B[3])))) + C),
        "pairs chained: (c+(p0+p1))+(p2+p3), exact pairs": lambda A, B, C: rnd(
            F(rnd(C + A[0] * B[0] + A[1] * B[1])) + A[2] * B[2] + A[3] * B[3]),
        "pairs chained reversed": lambda A, B, C: rnd(
            F(rnd(C + A[2] * B[2] + A[3] * B[3])) + A[0] * B[0] + A[1] * B[1]),
    }

    def chain(A, B, C, order):
        acc = C
        for k in order:
            acc = F(rnd(acc + A[k] * B[k]))
        return np.float32(float(acc))

    for blgp in (0, 4):
        score = {h: 0 for h in hyps}
        total = 0
        first_bad = {}
        for t in range(0, T):
            for blk in (0, 5, 15):
                for i in range(4):
                    for j in (0, 3):
                        A, B, C, D = case(t, blk, i, j, blgp)
                        total += 1
                        for h, fn in hyps.items():
                            got = fn(A, B, C)
                            if np.float32(got).view(np.uint32) == np.float32(D).view(np.uint32):
                                score[h] += 1
                            elif h not in first_bad:
                                first_bad[h] = (t, blk, i, j, float(got), float(D))
        print(f"BLGP={blgp}: {total} dot products")
        for h in hyps:
            print(f"  {score[h]:6d} / {total}  {h}   first mismatch {first_bad.get(h)}")


if __name__ == "__main__":
    main()
