#!/usr/bin/env python3
"""Planner for the wave-GEMM conv tilings: MFMA-bound time model per (TM, TN, KSPLIT) on 256 CUs."""
import itertools, math, sys

# (name, M, K(=taps*cin_pad), N) forward shapes of cfg #3 at B=8 (from profiles/r01_conv_per_shape_train_v9.txt)
LAYERS = [
    ("s5 3x1x1 2048->512", 3136, 3 * 2048, 512), ("s5 1x3x3 512->512", 3136, 9 * 512, 512),
    ("s5 1x1 512->2048", 3136, 512, 2048), ("s5 1x1 1152->2048 s2", 3136, 1152, 2048),
    ("s4 3x1x1 1024->256", 12544, 3 * 1024, 256), ("s4 1x3x3 256->256", 12544, 9 * 256, 256),
    ("s4 1x1 256->1024", 12544, 256, 1024), ("s4 3x1x1 1152->512", 12544, 3 * 1152, 512),
    ("s3 3x3 128->128", 50176, 9 * 128, 128), ("s3 1x1 128->512", 50176, 128, 512),
    ("s3 1x1 512->128", 50176, 512, 128), ("s3 3x1x1 576->256", 50176, 3 * 576, 256),
    ("s2 3x3 64->64", 200704, 9 * 64, 64), ("s2 1x1 64->256", 200704, 64, 256),
    ("s2 1x1 256->64", 200704, 256, 64), ("s2 1x1 288->128", 200704, 288, 128),
    ("f2 3x3 8->8", 802816, 9 * 16, 8), ("f2 1x1 8->32", 802816, 16, 32), ("f2 3x1x1 32->8", 802816, 96, 8),
    ("f3 3x3 16->16", 200704, 144, 16), ("f3 1x1 16->64", 200704, 16, 64), ("f3 3x1x1 64->16", 200704, 192, 16),
    ("f4 3x3 32->32", 50176, 288, 32), ("f4 1x1 32->128", 50176, 32, 128), ("f4 3x1x1 128->32", 50176, 384, 32),
    ("f5 3x3 64->64", 12544, 576, 64), ("f5 1x1 64->256", 12544, 64, 256), ("f5 3x1x1 256->64", 12544, 768, 64),
]
CUS = 256
OVERHEAD_STEPS = 0.0  # fixed per-WG cost in units of MFMA-steps (tuned later)


def plan(M, K, N, cfgs, fixed_us=4.0):
    nk = K // 16
    best = None
    for TM, TN, KS in cfgs:
        nbn = math.ceil(N / (TN * 16))
        nbm0 = math.ceil(M / (TM * 16))
        # let the number of M tiles grow a little so that tiles become a multiple of the chip (balanced rows)
        for nbm in range(nbm0, nbm0 + 64):
            rows = math.ceil(M / nbm)
            tiles = nbm * nbn
            tpw = 4 // KS
            wgs = math.ceil(tiles / tpw)
            rounds = math.ceil(wgs / CUS)
            steps = math.ceil(nk / KS)
            mfma_cycles = rounds * steps * TM * TN * 4 * 32  # per SIMD
            t_us = mfma_cycles / 2.1e3 + fixed_us  # 2.1 GHz under load
            if best is None or t_us < best[0]:
                best = (t_us, TM, TN, KS, nbm, rows, wgs)
    return best


if __name__ == "__main__":
    sets = {
        "all": [(tm, tn, ks) for tm in range(2, 14) for tn in (1, 2, 4) for ks in (1, 2, 4) if tm * tn <= 28],
        "few": [(13, 2, 4), (13, 2, 1), (7, 4, 4), (7, 4, 1), (13, 1, 1), (13, 1, 4), (7, 2, 4), (7,2,1), (4, 4, 4), (4,4,1)],
    }
    for name, cfgs in sets.items():
        print("==", name)
        tot = 0
        totf = 0
        for lname, M, K, N in LAYERS:
            t, TM, TN, KS, nbm, rows, wgs = plan(M, K, N, cfgs)
            fl = 2.0 * M * K * N
            tot += t
            totf += fl
            print("%-24s M=%7d K=%5d N=%5d -> TM=%2d TN=%d KS=%d rows=%3d wgs=%5d  %7.1f us  %6.1f TF" % (
                lname, M, K, N, TM, TN, KS, rows, wgs, t, fl / t / 1e6))
        print("sum %.1f us, %.1f TF/s" % (tot, totf / tot / 1e6))
