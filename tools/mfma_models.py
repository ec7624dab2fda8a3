#!/usr/bin/env python
"""Which arithmetic does v_mfma_f32_32x32x16_bf16 perform?  Compares the dump of tools/ubench/mfma_probe
(inputs + hardware outputs) with candidate models, bit for bit.

    python tools/mfma_models.py gpurun_out/mfma_probe.bin
"""
import sys
from fractions import Fraction

import numpy as np


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def round_f32(fr):
    """Fraction -> nearest float32 (ties to even), as numpy float32 (via exact float64 when representable)."""
    if fr == 0:
        return np.float32(0.0)
    # float64 conversion of a Fraction is correctly rounded; double rounding can differ from direct rounding only when
    # the float64 lies exactly on a float32 tie - handle that case exactly
    d = float(fr)
    f = np.float32(d)
    if Fraction(float(f)) == fr:
        return f
    lo = np.nextafter(f, np.float32(-np.inf)) if Fraction(float(f)) > fr else f
    hi = np.nextafter(lo, np.float32(np.inf))
    dl, dh = fr - Fraction(float(lo)), Fraction(float(hi)) - fr
    if dl < dh:
        return lo
    if dh < dl:
        return hi
    return lo if (lo.view(np.uint32) & 1) == 0 else hi


def trunc_f32(fr):
    f = round_f32(fr)
    if abs(Fraction(float(f))) > abs(fr):
        f = np.nextafter(f, np.float32(0.0))
    return f


def models(a, b, c):
    """a, b: 16 float32 (bf16 values); c: float32 -> dict of candidate results"""
    p = [Fraction(float(x)) * Fraction(float(y)) for x, y in zip(a, b)]
    out = {}
    # exact sum, one rounding
    out["exact_all"] = round_f32(sum(p) + Fraction(float(c)))
    out["exact_all_trunc"] = trunc_f32(sum(p) + Fraction(float(c)))
    # sequential fma chain in k order, starting from c
    acc = Fraction(float(c))
    for x in p:
        acc = Fraction(float(round_f32(acc + x)))
    out["seq_fma"] = np.float32(float(acc))
    # groups of g products exact, added to the accumulator sequentially
    for g in (2, 4, 8):
        acc = Fraction(float(c))
        for k0 in range(0, 16, g):
            acc = Fraction(float(round_f32(acc + sum(p[k0:k0 + g]))))
        out["groups%d" % g] = np.float32(float(acc))
        acc = Fraction(float(c))
        for k0 in range(0, 16, g):                       # interleaved halves: k = j and 8 + j together
            pass
    # the two lane halves (k 0..7 and 8..15) summed exactly each, then combined with c
    out["halves_then_c"] = round_f32(Fraction(float(round_f32(sum(p[:8]) + sum(p[8:])))) + Fraction(float(c)))
    # dot of all products rounded, then added to c
    out["dot_then_c"] = round_f32(Fraction(float(round_f32(sum(p)))) + Fraction(float(c)))
    # interleaved groups: {k, k+8} pairs, quads {k,k+1,k+8,k+9}
    for name, groups in (("pairs_k_k8", [[k, k + 8] for k in range(8)]),
                         ("quads_k_k8", [[k, k + 1, k + 8, k + 9] for k in range(0, 8, 2)]),
                         ("oct_k_k8", [[k, k + 1, k + 2, k + 3, k + 8, k + 9, k + 10, k + 11] for k in range(0, 8, 4)])):
        acc = Fraction(float(c))
        for gidx in groups:
            acc = Fraction(float(round_f32(acc + sum(p[i] for i in gidx))))
        out[name] = np.float32(float(acc))
    return out


def main():
    raw = open(sys.argv[1], "rb").read()
    W = int(np.frombuffer(raw[:4], dtype=np.int32)[0])
    off = 4
    A = np.frombuffer(raw[off:off + W * 512 * 2], dtype=np.uint16).reshape(W, 32, 16); off += W * 512 * 2
    B = np.frombuffer(raw[off:off + W * 512 * 2], dtype=np.uint16).reshape(W, 16, 32); off += W * 512 * 2
    C = np.frombuffer(raw[off:off + W * 1024 * 4], dtype=np.float32).reshape(W, 32, 32); off += W * 1024 * 4
    D = np.frombuffer(raw[off:off + W * 1024 * 4], dtype=np.float32).reshape(W, 32, 32)
    Af, Bf = bf16_to_f32(A), bf16_to_f32(B)
    rng = np.random.default_rng(0)
    score, n = {}, 0
    per_spread = {}
    for w in range(W):
        for _ in range(24):
            r, cidx = int(rng.integers(0, 32)), int(rng.integers(0, 32))
            m = models(Af[w, r, :], Bf[w, :, cidx], C[w, r, cidx])
            hw = D[w, r, cidx]
            n += 1
            for k, v in m.items():
                ok = np.float32(v).view(np.uint32) == hw.view(np.uint32)
                score[k] = score.get(k, 0) + int(ok)
                per_spread.setdefault((w % 4, k), [0, 0])
                per_spread[(w % 4, k)][0] += int(ok); per_spread[(w % 4, k)][1] += 1
    print("samples", n)
    for k, v in sorted(score.items(), key=lambda kv: -kv[1]):
        print("%-18s %5d / %d   by exponent spread 0/3/12/30: %s" % (
            k, v, n, " ".join("%d/%d" % tuple(per_spread[(s, k)]) for s in range(4))))


if __name__ == "__main__":
    main()
