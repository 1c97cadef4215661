// latch_anneal8.c -- slot assignment for clatch8_kernel (round 5): TWO waves per keypoint share one window kept in EIGHT byte-shifted
// copies, so that every 8-pixel patch row is one 8-byte ALIGNED ds_read_b64 (256 B/clk; the four-copy kernel reads rows as
// ds_read2_b32 at 128 B/clk).  Which learned triplet a (wave, round, lane) slot evaluates is free (the kernel routes the result bits
// to the descriptor's order through LDS); this tool searches the assignment -- and a qword offset per copy -- for few bank conflicts.
//
//   gcc -O2 -o tools/bin/latch_anneal8 tools/latch_anneal8.c -lm
//   tools/bin/latch_anneal8 <iterations> <seed> [out.inc]
//
// Model (MI355X guide, LDS table): ds_read_b64 is serviced in two groups of 32 lanes; bank of byte address a = (a / 4) mod 64; a lane's
// aligned 8 bytes take banks 2 q, 2 q + 1 with q = (a / 8) mod 32, so a group is conflict-free when its 32 lanes have 32 distinct q.
// Every row step adds the same 56 bytes to all lanes, so the degree of a group's read of one patch kind is the largest multiplicity
// among the lanes' q.  Slot s = wave * 256 + round * 64 + lane; group = s / 32 (16 groups); three kinds (a, b, c) per slot; a slot may
// exchange the roles of a and c (S changes sign).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ROW0 5
#define COL0 5
#define STRIDE 56
#define COPY_STRIDE 3264          /* bytes between copy bases before the per-copy offset: 3136 window + 8 slack + up to 15 qwords of offset */

static int pat[512][6];
static int offs[8];               /* per copy: base = k * COPY_STRIDE + 8 * offs[k], offs in 0..15 */
static int q[512][3];             /* qword bank of triplet n's patch kind k under the current offsets */
static int slot_t[512], swp[512];

static uint64_t rs = 88172645463325252ull;
static inline uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 24); }
static inline double rndf(void) { return (rnd() & 0xFFFFFF) / 16777216.0; }

static int patch_addr(int row, int col)
{
    const int p = (row - ROW0) * STRIDE + (col - COL0);
    return (p & 7) * COPY_STRIDE + 8 * offs[p & 7] + (p & ~7);
}
static void banks(void)
{
    for (int i = 0; i < 512; ++i) for (int k = 0; k < 3; ++k) q[i][k] = (patch_addr(pat[i][2 * k], pat[i][2 * k + 1]) / 8) % 32;
}
static int group_cost(int g, int* deg)
{
    int cnt[3][32];
    memset(cnt, 0, sizeof cnt);
    for (int s = 32 * g; s < 32 * g + 32; ++s) {
        const int n = slot_t[s];
        for (int k = 0; k < 3; ++k) cnt[k][q[n][(swp[n] && k != 1) ? 2 - k : k]]++;
    }
    int c = 0, d = 0;
    for (int k = 0; k < 3; ++k) {
        int mx = 0, sq = 0;
        for (int b = 0; b < 32; ++b) { const int v = cnt[k][b]; if (v > mx) mx = v; sq += v * v; }
        c += 100 * mx + sq; d += mx;
    }
    if (deg) *deg = d;
    return c;
}
static int gc[16];
static int total_deg(void) { int t = 0; for (int g = 0; g < 16; ++g) { int d; group_cost(g, &d); t += d; } return t; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s iterations seed [out.inc]\n", argv[0]); return 2; }
    const long iters = atol(argv[1]); rs ^= (uint64_t)atol(argv[2]) * 0x9E3779B97F4A7C15ull;
    FILE* f = fopen("coloc_amd/csrc/latch_pattern.inc", "r");
    if (!f) { perror("latch_pattern.inc"); return 1; }
    char line[256]; int n = 0;
    while (fgets(line, sizeof line, f))
        if (line[0] == '{' && n < 512 && sscanf(line, "{%d,%d, %d,%d, %d,%d}", &pat[n][0], &pat[n][1], &pat[n][2], &pat[n][3], &pat[n][4], &pat[n][5]) == 6) ++n;
    fclose(f);
    if (n != 512) { fprintf(stderr, "pattern: %d rows\n", n); return 1; }
    for (int i = 0; i < 512; ++i) slot_t[i] = i;
    for (int k = 0; k < 8; ++k) offs[k] = rnd() & 15;
    banks();
    for (int g = 0; g < 16; ++g) gc[g] = group_cost(g, NULL);
    printf("start: degree sum %d (lower bound 48)\n", total_deg());
    const double T0 = 60.0;
    double T = T0; const double cool = pow(0.4 / T0, 1.0 / (double)iters);
    long cur = 0; for (int g = 0; g < 16; ++g) cur += gc[g];
    long best = cur;
    static int b_slot[512], b_swp[512], b_offs[8];
    memcpy(b_slot, slot_t, sizeof slot_t); memcpy(b_swp, swp, sizeof swp); memcpy(b_offs, offs, sizeof offs);
    for (long it = 0; it < iters; ++it, T *= cool) {
        const uint32_t kind = rnd() % 64;
        if (kind == 0 && it < iters / 2) {                 /* move a copy's offset: everything changes */
            const int k = rnd() & 7, old = offs[k];
            offs[k] = rnd() & 15;
            banks();
            int nc[16]; long d = 0;
            for (int g = 0; g < 16; ++g) { nc[g] = group_cost(g, NULL); d += nc[g] - gc[g]; }
            if (d <= 0 || rndf() < exp(-(double)d / T)) { memcpy(gc, nc, sizeof gc); cur += d; }
            else { offs[k] = old; banks(); }
        } else if (kind < 48) {                             /* exchange two slots of different groups */
            const int s1 = rnd() & 511, s2 = rnd() & 511;
            if ((s1 >> 5) == (s2 >> 5)) continue;
            int t = slot_t[s1]; slot_t[s1] = slot_t[s2]; slot_t[s2] = t;
            const int g1 = s1 >> 5, g2 = s2 >> 5;
            const int n1 = group_cost(g1, NULL), n2 = group_cost(g2, NULL);
            const long d = (long)n1 + n2 - gc[g1] - gc[g2];
            if (d <= 0 || rndf() < exp(-(double)d / T)) { gc[g1] = n1; gc[g2] = n2; cur += d; }
            else { t = slot_t[s1]; slot_t[s1] = slot_t[s2]; slot_t[s2] = t; }
        } else {                                            /* exchange the roles of a slot's a and c */
            const int s = rnd() & 511, tn = slot_t[s], g = s >> 5;
            swp[tn] ^= 1;
            const int n1 = group_cost(g, NULL);
            const long d = (long)n1 - gc[g];
            if (d <= 0 || rndf() < exp(-(double)d / T)) { gc[g] = n1; cur += d; }
            else swp[tn] ^= 1;
        }
        if (cur < best) { best = cur; memcpy(b_slot, slot_t, sizeof slot_t); memcpy(b_swp, swp, sizeof swp); memcpy(b_offs, offs, sizeof offs); }
    }
    memcpy(slot_t, b_slot, sizeof slot_t); memcpy(swp, b_swp, sizeof swp); memcpy(offs, b_offs, sizeof offs);
    banks();
    const int d = total_deg();
    int nswp = 0; for (int i = 0; i < 512; ++i) nswp += swp[i];
    int hist[8] = { 0 };
    for (int g = 0; g < 16; ++g) {
        int cnt[3][32]; memset(cnt, 0, sizeof cnt);
        for (int s = 32 * g; s < 32 * g + 32; ++s) { const int t = slot_t[s]; for (int k = 0; k < 3; ++k) cnt[k][q[t][(swp[t] && k != 1) ? 2 - k : k]]++; }
        for (int k = 0; k < 3; ++k) { int mx = 0; for (int b = 0; b < 32; ++b) if (cnt[k][b] > mx) mx = cnt[k][b]; hist[mx < 7 ? mx : 7]++; }
    }
    printf("degree sum %d of 48 (group reads by degree: 1:%d 2:%d 3:%d 4+:%d), %d slots with a / c exchanged, offsets %d %d %d %d %d %d %d %d\n",
           d, hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7], nswp, offs[0], offs[1], offs[2], offs[3], offs[4], offs[5], offs[6], offs[7]);
    if (argc > 3) {
        FILE* o = fopen(argv[3], "w");
        if (!o) { perror(argv[3]); return 1; }
        fprintf(o, "// GENERATED by tools/latch_anneal8.c (%ld iterations, seed %s): sum of the 48 group-read degrees %d (conflict-free: 48).\n", iters, argv[2], d);
        fprintf(o, "// clatch8_kernel: two waves per keypoint, eight byte-shifted window copies; slot = wave * 256 + round * 64 + lane evaluates\n");
        fprintf(o, "// learned triplet (v & 511), bit 10: a and c exchanged.  Copy k holds window byte i + k at byte LATCH8_COPY_BASES[k] + i.\n");
        fprintf(o, "#define LATCH8_COPY_BASES { ");
        for (int k = 0; k < 8; ++k) fprintf(o, "%d%s", k * COPY_STRIDE + 8 * offs[k], k < 7 ? ", " : " }\n");
        fprintf(o, "#define LATCH8_WINDOW_BYTES %d\n", 8 * COPY_STRIDE);
        fprintf(o, "#define LATCH8_SLOT_TRIPLET { \\\n");
        for (int s = 0; s < 512; s += 16) {
            fprintf(o, "   ");
            for (int i = 0; i < 16; ++i) { const int t = slot_t[s + i]; fprintf(o, " %d,", t | (swp[t] << 10)); }
            fprintf(o, " \\\n");
        }
        fprintf(o, "}\n");
        fclose(o);
    }
    return 0;
}
