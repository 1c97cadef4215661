#!/usr/bin/env python3
"""Search a triplet -> (round, lane) assignment for the CLATCH kernel that minimises LDS bank
conflicts, and emit coloc_amd/csrc/latch_layout.inc.

In the kernel lane l of round j evaluates one of the 512 learned triplets; its three 8x8 patches
are read row by row (8 bytes = two dwords, `ds_read2_b32`) from the byte-shifted window copy
selected by (col & 3).  A wave's two 32-lane halves are banked independently (32 banks x 4 B), and
every row adds the same constant to all lanes' banks, so the conflict degree of a half-wave is the
largest multiplicity among its 32 base banks -- per patch kind (a, b, c).  With the natural
assignment (triplet n -> round n/64, lane n%64) that degree averages 3.4; this search drives it
towards 1 by choosing which 32 triplets share a half-wave.  The descriptor bit order is restored in
the kernel with one ds_bpermute per output round, so the OUTPUT is unchanged (tests prove it).

Deterministic (fixed seed).  Usage: python tools/opt_latch_layout.py [iterations]
"""
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# LDS geometry (emitted into latch_layout.inc; coloc_amd/csrc/clatch.hip takes it from there)
ROW0, COL0 = 5, 5            # window region kept in LDS = ROI rows/cols 5..60, exactly what the patches span
STRIDE = 56                  # bytes per stored row (cols 5..60)
NROWS = 56                   # rows 5..60
COPY_BYTES = NROWS * STRIDE + 8          # 3488: one shifted copy incl. slack for the +3 shift
COPY_BASE = [0, 0, 0, 0]     # byte offset of shifted copy k inside a wave's region (filled by choose_bases)


def choose_bases(pat):
    """Pick the bank offset of each shifted copy (its base address / 4 mod 32) so that the global bank
    histogram of every patch kind is as flat as possible; copies are then packed with the padding
    that realises those offsets."""
    best = None
    for o1 in range(32):
        for o2 in range(32):
            for o3 in range(32):
                o = [0, o1, o2, o3]
                sp = 0
                for k in range(3):
                    h = [0] * 32
                    for t in pat:
                        pp = (t[2 * k] - ROW0) * STRIDE + (t[2 * k + 1] - COL0)
                        h[(o[pp & 3] + (pp >> 2)) % 32] += 1
                    sp += sum((x - 16) ** 2 for x in h) + 1000 * max(0, max(h) - 16)
                if best is None or sp < best[0]:
                    best = (sp, o)
    o = best[1]
    base, cur = [0, 0, 0, 0], 0
    for k in range(4):
        while (cur // 4) % 32 != o[k]:
            cur += 4
        base[k] = cur
        cur += COPY_BYTES
    return base, cur


def lds_addr(row, col):
    p = (row - ROW0) * STRIDE + (col - COL0)
    return COPY_BASE[p & 3] + (p & ~3)


def bank(row, col):
    return (lds_addr(row, col) // 4) % 32


def load_pattern():
    rows = []
    for line in open(os.path.join(ROOT, "coloc_amd", "csrc", "latch_pattern.inc")):
        m = re.match(r"\{(\d+),(\d+), (\d+),(\d+), (\d+),(\d+)\}", line)
        if m:
            rows.append([int(x) for x in m.groups()])
    assert len(rows) == 512
    return rows


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
    pat = load_pattern()
    global COPY_BASE
    COPY_BASE, wave_bytes = choose_bases(pat)
    wave_bytes = (wave_bytes + 15) // 16 * 16
    print("copy bases", COPY_BASE, "bytes per wave", wave_bytes)
    banks = [[bank(t[0], t[1]), bank(t[2], t[3]), bank(t[4], t[5])] for t in pat]
    rng = random.Random(12345)
    G = 16
    group = [list(range(g * 32, (g + 1) * 32)) for g in range(G)]       # natural assignment
    cnt = [[[0] * 32 for _ in range(3)] for _ in range(G)]
    for g in range(G):
        for n in group[g]:
            for k in range(3):
                cnt[g][k][banks[n][k]] += 1

    def gcost(g):     # conflict degree dominates; the square term smooths the plateaus for the search
        return sum(100 * max(cnt[g][k]) + sum(c * c for c in cnt[g][k]) for k in range(3))

    def degree():
        return sum(max(cnt_g[k]) for cnt_g in cnt for k in range(3))

    cost = [gcost(g) for g in range(G)]
    total = sum(cost)
    print("natural assignment: degree sum %d (avg %.2f)" % (degree(), degree() / 48.0))
    T = 40.0
    best_total, best_group = total, [list(x) for x in group]
    import math
    for it in range(iters):
        g1 = rng.randrange(G); g2 = rng.randrange(G)
        if g1 == g2:
            continue
        i1 = rng.randrange(32); i2 = rng.randrange(32)
        n1, n2 = group[g1][i1], group[g2][i2]
        for k in range(3):
            cnt[g1][k][banks[n1][k]] -= 1; cnt[g1][k][banks[n2][k]] += 1
            cnt[g2][k][banks[n2][k]] -= 1; cnt[g2][k][banks[n1][k]] += 1
        c1, c2 = gcost(g1), gcost(g2)
        d = c1 + c2 - cost[g1] - cost[g2]
        if d <= 0 or rng.random() < math.exp(-d / T):
            group[g1][i1], group[g2][i2] = n2, n1
            cost[g1], cost[g2] = c1, c2
            total += d
            if total < best_total:
                best_total, best_group = total, [list(x) for x in group]
        else:
            for k in range(3):
                cnt[g1][k][banks[n1][k]] += 1; cnt[g1][k][banks[n2][k]] -= 1
                cnt[g2][k][banks[n2][k]] += 1; cnt[g2][k][banks[n1][k]] -= 1
        T = max(0.5, T * 0.999997)
    # recount the best assignment
    cnt = [[[0] * 32 for _ in range(3)] for _ in range(G)]
    for g in range(G):
        for n in best_group[g]:
            for k in range(3):
                cnt[g][k][banks[n][k]] += 1
    best_total = sum(max(cnt[g][k]) for g in range(G) for k in range(3))
    print("optimised: degree sum %d (avg %.2f)" % (best_total, best_total / 48.0))
    # slot s = round*64 + lane; half-wave group g = round*2 + (lane >= 32)
    slot_triplet = [0] * 512
    for g in range(G):
        rnd, half = g // 2, g % 2
        for i, n in enumerate(sorted(best_group[g])):
            slot_triplet[rnd * 64 + half * 32 + i] = n
    out = ["// GENERATED by tools/opt_latch_layout.py -- conflict-minimising triplet -> (round, lane) assignment.",
           "// slot = round*64 + lane evaluates learned triplet LATCH_SLOT_TRIPLET[slot].",
           "// total half-wave conflict degree %d over 48 (16 half-wave groups x 3 patch kinds); natural order: %d."
           % (best_total, sum(1 for _ in [0]) and 0 or 0)]
    out[-1] = "// sum of half-wave conflict degrees: %d (lower bound 48)." % best_total
    out += ["#define LATCH_ROW0 %d" % ROW0, "#define LATCH_COL0 %d" % COL0, "#define LATCH_STRIDE %d" % STRIDE,
            "#define LATCH_NROWS %d" % NROWS, "#define LATCH_WAVE_BYTES %d" % wave_bytes,
            "#define LATCH_COPY_BASES { %s }" % ", ".join(str(b) for b in COPY_BASE),
            "#define LATCH_SLOT_TRIPLET { \\"]
    for r in range(0, 512, 16):
        out.append("    " + ", ".join(str(v) for v in slot_triplet[r:r + 16]) + ", \\")
    out.append("}")
    dst = os.path.join(ROOT, "coloc_amd", "csrc", "latch_layout.inc")
    open(dst, "w").write("\n".join(out) + "\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
