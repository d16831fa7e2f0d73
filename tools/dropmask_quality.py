#!/usr/bin/env python3
"""Offline quality check of the attention keep-mask generator (csrc/attention_common.h: drop_keep32), CPU only (numpy).

The forward derives 32 keep bits per (query row, key tile, lane group) counter: one strong mix of the counter, seven cheaper words
chained from it, folded digit by digit along the binary expansion of thr8 / 256 (AND for a 0 digit, OR for a 1 digit).  Every keep bit
must be Bernoulli(1 - thr8/256) and independent of its neighbours -- within the word (the 32 scores of a lane), across key groups,
across rows.  This script evaluates a generator on the counter lattice the kernel uses and prints

    rate          observed drop rate vs thr8 / 256
    bit-rate      largest deviation of a single bit position's rate (in standard errors)
    in-word       largest |correlation| between two bit positions of one word (32 x 32 pairs)
    key / row / diagonal   largest |correlation| of a bit with the same bit (and any bit) of the neighbouring counter

for `--chain old` (round 4: x + C; x ^= x >> 11; mul24 + (x >> 8): 5 VALU per word), `--chain new` (round 5, shipped:
(w >> 8) * C + rot(w, 13): 3 VALU per word) and `--chain two` (the rejected (w >> 8) * C + w).  Sampling noise for N words is 1 / sqrt(N) per correlation; with 32 x 32 pairs the largest of them is ~4.2 / sqrt(N).
"""
import argparse

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def u32(x):
    return (x & M32).astype(np.uint64)


def mul24(a, b):
    return u32((a & np.uint64(0xFFFFFF)) * np.uint64(b & 0xFFFFFF))


def drop_hash(x):
    x = x ^ (x >> np.uint64(11)); x = u32(mul24(x, 0xD35A2D) + (x >> np.uint64(8)))
    x = x ^ (x >> np.uint64(13)); x = u32(mul24(x, 0x9E3B35) + (x >> np.uint64(9)))
    return x ^ (x >> np.uint64(15))


def light_old(w):
    x = u32(w + np.uint64(0x9E3779B1))
    x = x ^ (x >> np.uint64(11))
    return u32(mul24(x, 0xD35A2D) + (x >> np.uint64(8)))


def rot(w, r):
    return u32((w >> np.uint64(r)) | (w << np.uint64(32 - r)))


def light_new(w):
    # csrc/attention_common.h drop_light (round 5): shift + rotate + multiply-add = three VALU slots
    return u32(mul24(w >> np.uint64(8), 0xD35A2D) + rot(w, 13))


def light_two_slot(w):
    # the rejected two-slot form: the un-rotated word as the addend leaves bit pairs of the folded mask correlated at 0.1 - 0.3
    return u32(mul24(w >> np.uint64(8), 0xD35A2D) + w)


def keep32(counter, thr8, light):
    w = drop_hash(counter)
    acc = np.zeros_like(w)
    for k in range(8):
        acc = (acc | w) if (thr8 >> k) & 1 else (acc & w)
        if k < 7:
            w = light(w)
    return u32(~acc)


def bits_of(words):
    return ((words[..., None] >> np.arange(32, dtype=np.uint64)) & np.uint64(1)).astype(np.float32)


def corr(a, b):
    a = a - a.mean(0); b = b - b.mean(0)
    return (a.T @ b) / np.sqrt((a * a).sum(0)[:, None] * (b * b).sum(0)[None, :])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chain", choices=("old", "new", "two"), default="new")
    ap.add_argument("--thr8", type=int, nargs="*", default=[26, 25, 64, 1, 128, 255])
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--groups", type=int, default=512, help="key groups (16 keys each: key tile * 4 + lane group)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--bh", type=int, default=37)
    args = ap.parse_args()
    light = {"old": light_old, "new": light_new, "two": light_two_slot}[args.chain]
    rows = np.arange(args.rows, dtype=np.uint64)
    rowc = u32((np.uint64(args.bh * 2048) + rows) * np.uint64(0x9E3779B1) + np.uint64(args.seed))     # drop_row_const
    grp = np.arange(args.groups, dtype=np.uint64)
    counter = u32(rowc[:, None] + mul24(grp, 0xEBCA77)[None, :])                                       # [rows, groups]
    n = counter.size
    print(f"# chain {args.chain}: {args.rows} rows x {args.groups} key groups = {n} words; noise 1/sqrt(N) = {n ** -0.5:.1e}, "
          f"largest of 1024 pairs ~ {4.2 * n ** -0.5:.1e}")
    worst = 0.0
    for thr8 in args.thr8:
        kw = keep32(counter, thr8, light)
        b = bits_of(kw)                                   # [rows, groups, 32] keep bits
        p = thr8 / 256.0
        rate = 1.0 - b.mean()
        se_bit = (p * (1 - p) / n) ** 0.5
        bitdev = np.abs((1.0 - b.reshape(-1, 32).mean(0)) - p).max() / max(se_bit, 1e-12)
        flat = b.reshape(-1, 32)
        cw = corr(flat, flat)
        np.fill_diagonal(cw, 0.0)
        key = np.abs(corr(b[:, :-1].reshape(-1, 32), b[:, 1:].reshape(-1, 32))).max()
        row = np.abs(corr(b[:-1].reshape(-1, 32), b[1:].reshape(-1, 32))).max()
        dia = np.abs(corr(b[:-1, :-1].reshape(-1, 32), b[1:, 1:].reshape(-1, 32))).max()
        row2 = np.abs(corr(b[:-2:2].reshape(-1, 32), b[1:-1:2].reshape(-1, 32))).max()
        print(f"thr8 {thr8:3d}  rate {rate:.5f} (want {p:.5f}, {abs(rate - p) / max((p * (1 - p) / (32 * n)) ** 0.5, 1e-12):.1f} se)  "
              f"bit-rate {bitdev:.1f} se  in-word {np.abs(cw).max():.1e}  key {key:.1e}  row {row:.1e}  row-pair {row2:.1e}  diagonal {dia:.1e}")
        worst = max(worst, np.abs(cw).max(), key, row, dia)
    print(f"# largest correlation seen: {worst:.1e}")


if __name__ == "__main__":
    main()
