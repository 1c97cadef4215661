// latch_anneal.c -- search a (round, lane) -> learned-triplet assignment for clatch_kernel that minimises LDS bank conflicts.
// C port and extension of tools/opt_latch_layout.py (which stays the generator of the geometry constants).
//
//   gcc -O2 -o tools/bin/latch_anneal tools/latch_anneal.c -lm
//   tools/bin/latch_anneal <mode> <rot> <swap> <iterations> <seed> [out.inc]
//     mode 0: any triplet in any slot (the kernel restores the descriptor bit order with ds_bpermute)
//     mode 1: lane l evaluates the eight triplets of ONE descriptor byte (its eight sign bits ARE that byte: no un-permute)
//     rot  1: a triplet may read its patch rows in the order 4..7, 0..3 (two base addresses per patch, same immediates)
//     swap 1: a triplet may exchange the roles of its a and c patches (S changes sign: the bit is then S > 0)
//
// Model (same as the Python tool): a wave's two 32-lane halves are banked independently, 32 banks x 4 B; every row step adds
// the same constant to all lanes' banks, so the conflict degree of a half-wave read is the largest multiplicity among the 32
// base banks of its patch kind; with rot the rotated lanes are 24 banks ahead during row steps 0..3 and 8 ahead during 4..7.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ROW0 5
#define COL0 5
#define STRIDE 56
static const int COPY_BASE[4] = { 0, 3192, 6376, 9544 };   /* = latch_layout.inc (tools/opt_latch_layout.py choose_bases) */

static int pat[512][6];
static int bank[512][3];
static int pair64[512][3];        /* b64 model: (address / 8) mod 32 if the patch rows are 8-byte aligned, else -1 */
static int use_b64;                /* cost of a (round, kind) cell: every lane aligned -> one 64-bank access, else two 32-bank accesses */
static int copy_shift4[4];         /* per shifted copy: base moved by 4 bytes (flips which of its patches are 8-byte aligned) */
static int slot_t[8][64];          /* triplet in (round, lane) */
static int rot[512], swp[512];
static int mode, use_rot, use_swap;
static int have_plan, plan_flag[8][3];      /* planned (round, kind) cells that must stay all-aligned (hard constraint of the search) */
static int compatible(int n, int r)
{
    for (int k = 0; k < 3; ++k) if (plan_flag[r][k]) { const int kk = (swp[n] && k != 1) ? 2 - k : k; if (pair64[n][kk] < 0) return 0; }
    return 1;
}

static uint64_t rs = 88172645463325252ull;
static inline uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 24); }
static inline double rndf(void) { return (rnd() & 0xFFFFFF) / 16777216.0; }

static int patch_addr(int row, int col)
{
    const int p = (row - ROW0) * STRIDE + (col - COL0);
    return COPY_BASE[p & 3] + 4 * copy_shift4[p & 3] + (p & ~3);
}
static int patch_bank(int row, int col) { return (patch_addr(row, col) / 4) % 32; }

/* cost of half-wave group (round r, half h): sum over kind and row phase of 100 * max + sum of squares; *deg = sum of max */
static int round_aligned(int r, int k)      /* all 64 lanes of round r read kind k from 8-byte aligned rows */
{
    for (int l = 0; l < 64; ++l) { const int n = slot_t[r][l]; const int kk = (swp[n] && k != 1) ? 2 - k : k; if (pair64[n][kk] < 0) return 0; }
    return 1;
}
static int group_cost(int r, int h, int* deg)
{
    int cnt[3][2][32];
    memset(cnt, 0, sizeof cnt);
    int al[3] = { 0, 0, 0 }, nal[3] = { 0, 0, 0 };
    if (use_b64) for (int k = 0; k < 3; ++k) {
        al[k] = round_aligned(r, k);
        for (int l = 32 * h; l < 32 * h + 32; ++l) { const int n = slot_t[r][l]; const int kk = (swp[n] && k != 1) ? 2 - k : k; nal[k] += pair64[n][kk] >= 0; }
    }
    for (int l = 32 * h; l < 32 * h + 32; ++l) {
        const int n = slot_t[r][l];
        for (int k = 0; k < 3; ++k) {
            const int kk = (swp[n] && k != 1) ? 2 - k : k;
            const int b = al[k] ? pair64[n][kk] : bank[n][kk];
            cnt[k][0][(b + (rot[n] ? 24 : 0)) & 31]++;
            cnt[k][1][(b + (rot[n] ? 8 : 0)) & 31]++;
        }
    }
    int c = 0, d = 0;
    for (int k = 0; k < 3; ++k)
        for (int ph = 0; ph < 2; ++ph) {
            int mx = 0, sq = 0;
            for (int b = 0; b < 32; ++b) { const int v = cnt[k][ph][b]; if (v > mx) mx = v; sq += v * v; }
            /* LDS cycles of this half-wave read, in half-steps: two 32-bank accesses, or one 64-bank access when the whole round is aligned */
            const int w = (use_b64 && al[k]) ? 1 : 2;
            c += w * (100 * mx + sq); d += w * mx;
            /* search gradient towards homogeneous rounds: reward rounds that are nearly all-aligned (or nearly all-unaligned costs nothing) */
            if (use_b64 && !al[k]) c += 6 * (nal[k] > 16 ? 32 - nal[k] : 0);
        }
    if (deg) *deg = d;
    return c;
}

double g_T0 = 60.0;
static int gcost[8][2];
static long total_cost(void) { long t = 0; for (int r = 0; r < 8; ++r) for (int h = 0; h < 2; ++h) t += gcost[r][h]; return t; }
static int total_deg2(void) { int t = 0; for (int r = 0; r < 8; ++r) for (int h = 0; h < 2; ++h) { int d; group_cost(r, h, &d); t += d; } return t; }

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s mode rot swap iterations seed [out] [b64] [copy shift mask]\n", argv[0]); return 2; }
    mode = atoi(argv[1]); use_rot = atoi(argv[2]); use_swap = atoi(argv[3]);
    use_b64 = argc > 7 ? atoi(argv[7]) : 0;
    if (argc > 8) for (int k = 0; k < 4; ++k) copy_shift4[k] = (atoi(argv[8]) >> k) & 1;
    const long iters = atol(argv[4]); rs ^= (uint64_t)atol(argv[5]) * 0x9E3779B97F4A7C15ull;
    const char* pattern_path = "coloc_amd/csrc/latch_pattern.inc";
    FILE* f = fopen(pattern_path, "r");
    if (!f) { perror(pattern_path); return 1; }
    char line[256]; int n = 0;
    while (fgets(line, sizeof line, f))
        if (line[0] == '{' && n < 512 && sscanf(line, "{%d,%d, %d,%d, %d,%d}", &pat[n][0], &pat[n][1], &pat[n][2], &pat[n][3], &pat[n][4], &pat[n][5]) == 6) ++n;
    fclose(f);
    if (n != 512) { fprintf(stderr, "pattern: %d rows\n", n); return 1; }
    for (int i = 0; i < 512; ++i) for (int k = 0; k < 3; ++k) {
        bank[i][k] = patch_bank(pat[i][2 * k], pat[i][2 * k + 1]);
        const int a = patch_addr(pat[i][2 * k], pat[i][2 * k + 1]);
        pair64[i][k] = (a & 7) ? -1 : (a / 8) % 32;
    }
    /* start: mode 0 natural order (triplet 64 r + l); mode 1 lane l <- byte l, round j <- bit j of the byte */
    for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) slot_t[r][l] = mode ? 8 * l + r : 64 * r + l;
    if (argc > 9) {       /* a planned start (tools/latch_plan_b64.py): copy shifts, the cells to keep aligned, a feasible assignment */
        FILE* pf = fopen(argv[9], "r");
        if (!pf) { perror(argv[9]); return 1; }
        int shm = 0;
        if (fscanf(pf, "%d", &shm) != 1) return 1;
        for (int k = 0; k < 4; ++k) copy_shift4[k] = (shm >> k) & 1;
        for (int i = 0; i < 512; ++i) for (int k = 0; k < 3; ++k) {
            bank[i][k] = patch_bank(pat[i][2 * k], pat[i][2 * k + 1]);
            const int a = patch_addr(pat[i][2 * k], pat[i][2 * k + 1]);
            pair64[i][k] = (a & 7) ? -1 : (a / 8) % 32;
        }
        for (int r = 0; r < 8; ++r) if (fscanf(pf, "%d %d %d", &plan_flag[r][0], &plan_flag[r][1], &plan_flag[r][2]) != 3) return 1;
        for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) { int v; if (fscanf(pf, "%d", &v) != 1) return 1; slot_t[r][l] = v & 511; swp[v & 511] = (v >> 10) & 1; }
        fclose(pf);
        have_plan = argc > 10 ? atoi(argv[10]) : 1; use_b64 = 1;      /* argv[10] = 0: the plan is only the starting point */
        for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) if (!compatible(slot_t[r][l], r)) { fprintf(stderr, "plan: slot (%d, %d) incompatible\n", r, l); return 1; }
        if (argc > 11) { const double t0 = atof(argv[11]); if (t0 > 0) { extern double g_T0; g_T0 = t0; } }
    }
    for (int r = 0; r < 8; ++r) for (int h = 0; h < 2; ++h) gcost[r][h] = group_cost(r, h, NULL);
    printf("start: degree sum %.1f (x2: %d)\n", total_deg2() / 2.0, total_deg2());
    double T = g_T0; const double cool = pow(0.4 / g_T0, 1.0 / (double)iters);
    long cur = total_cost(), best = cur;
    static int best_slot[8][64], best_rot[512], best_swp[512];
    memcpy(best_slot, slot_t, sizeof slot_t); memcpy(best_rot, rot, sizeof rot); memcpy(best_swp, swp, sizeof swp);
    for (long it = 0; it < iters; ++it, T *= cool) {
        const uint32_t kind = rnd() % 16;
        int touched[4][2], nt = 0;
        int r1 = 0, l1 = 0, r2 = 0, l2 = 0, tn = -1, what = 0;
        if (kind < 10 || (!use_rot && !use_swap)) {
            if (mode == 0) {                     /* swap two slots */
                r1 = rnd() & 7; l1 = rnd() & 63; r2 = rnd() & 7; l2 = rnd() & 63;
                if (r1 == r2 && (l1 >> 5) == (l2 >> 5)) continue;
                what = 1;
            } else if (rnd() & 1) {              /* two rounds of one lane */
                l1 = l2 = rnd() & 63; r1 = rnd() & 7; r2 = rnd() & 7;
                if (r1 == r2) continue;
                what = 1;
            } else {                             /* the bytes of two lanes (all eight rounds) */
                l1 = rnd() & 63; l2 = rnd() & 63;
                if ((l1 >> 5) == (l2 >> 5)) continue;      /* same half-wave: no effect on the cost */
                what = 2;
            }
        } else {
            tn = rnd() & 511;
            what = (use_rot && (!use_swap || (rnd() & 1))) ? 3 : 4;
        }
        long d = 0;
        if (have_plan) {          /* keep every planned cell aligned */
            if (what == 1 && (!compatible(slot_t[r1][l1], r2) || !compatible(slot_t[r2][l2], r1))) continue;
            if (what == 4) {
                int rr = -1; for (int r = 0; r < 8 && rr < 0; ++r) for (int l = 0; l < 64; ++l) if (slot_t[r][l] == tn) { rr = r; break; }
                swp[tn] ^= 1; const int ok = compatible(tn, rr); swp[tn] ^= 1;
                if (!ok) continue;
            }
        }
        if (what == 1) {
            int t = slot_t[r1][l1]; slot_t[r1][l1] = slot_t[r2][l2]; slot_t[r2][l2] = t;
            touched[nt][0] = r1; touched[nt++][1] = l1 >> 5;
            if (r2 != r1 || (l2 >> 5) != (l1 >> 5)) { touched[nt][0] = r2; touched[nt++][1] = l2 >> 5; }
        } else if (what == 2) {
            for (int r = 0; r < 8; ++r) { int t = slot_t[r][l1]; slot_t[r][l1] = slot_t[r][l2]; slot_t[r][l2] = t; }
        } else if (what == 3) rot[tn] ^= 1;
        else swp[tn] ^= 1;
        int newc[8][2]; int chg[8][2]; memset(chg, 0, sizeof chg);
        if (what == 1) { for (int i = 0; i < nt; ++i) { chg[touched[i][0]][touched[i][1]] = 1; if (use_b64) chg[touched[i][0]][touched[i][1] ^ 1] = 1; } }
        else if (what == 2) { for (int r = 0; r < 8; ++r) chg[r][0] = chg[r][1] = 1; }
        else { for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; ++l) if (slot_t[r][l] == tn) { chg[r][l >> 5] = 1; if (use_b64) chg[r][(l >> 5) ^ 1] = 1; } }
        for (int r = 0; r < 8; ++r) for (int h = 0; h < 2; ++h) if (chg[r][h]) { newc[r][h] = group_cost(r, h, NULL); d += newc[r][h] - gcost[r][h]; }
        if (d <= 0 || rndf() < exp(-(double)d / T)) {
            for (int r = 0; r < 8; ++r) for (int h = 0; h < 2; ++h) if (chg[r][h]) gcost[r][h] = newc[r][h];
            cur += d;
            if (cur < best) { best = cur; memcpy(best_slot, slot_t, sizeof slot_t); memcpy(best_rot, rot, sizeof rot); memcpy(best_swp, swp, sizeof swp); }
        } else {   /* undo */
            if (what == 1) { int t = slot_t[r1][l1]; slot_t[r1][l1] = slot_t[r2][l2]; slot_t[r2][l2] = t; }
            else if (what == 2) { for (int r = 0; r < 8; ++r) { int t = slot_t[r][l1]; slot_t[r][l1] = slot_t[r][l2]; slot_t[r][l2] = t; } }
            else if (what == 3) rot[tn] ^= 1;
            else swp[tn] ^= 1;
        }
    }
    memcpy(slot_t, best_slot, sizeof slot_t); memcpy(rot, best_rot, sizeof rot); memcpy(swp, best_swp, sizeof swp);
    const int d2 = total_deg2();
    int nrot = 0, nswp = 0; for (int i = 0; i < 512; ++i) { nrot += rot[i]; nswp += swp[i]; }
    int nb64 = 0; if (use_b64) for (int r = 0; r < 8; ++r) for (int k = 0; k < 3; ++k) nb64 += round_aligned(r, k);
    printf("mode %d rot %d swap %d b64 %d shift %d%d%d%d: %s %.1f (today's layout: 162 half-steps = degree sum 81; bound 48), %d rotated, %d swapped, %d of 24 (round, kind) reads as ds_read_b64\n",
           mode, use_rot, use_swap, use_b64, copy_shift4[0], copy_shift4[1], copy_shift4[2], copy_shift4[3], use_b64 ? "LDS half-steps / 2" : "degree sum", d2 / (use_b64 ? 4.0 : 2.0), nrot, nswp, nb64);
    if (argc > 6) {
        FILE* o = fopen(argv[6], "w");
        if (!o) { perror(argv[6]); return 1; }
        fprintf(o, "// GENERATED by tools/latch_anneal.c (mode %d rot %d swap %d, %ld iterations, seed %s): degree sum %.1f, lower bound 48.\n",
                mode, use_rot, use_swap, iters, argv[5], d2 / 2.0);
        fprintf(o, "// slot = round*64 + lane evaluates learned triplet (v & 511); bit 9: rows in the order 4..7,0..3; bit 10: a and c exchanged.\n");
        fprintf(o, "#define LATCH_COPY_SHIFT4 { %d, %d, %d, %d }\n", copy_shift4[0], copy_shift4[1], copy_shift4[2], copy_shift4[3]);
        fprintf(o, "#define LATCH_ROUND_B64 { \\\n");
        for (int r = 0; r < 8; ++r) fprintf(o, "    { %d, %d, %d }, \\\n", use_b64 ? round_aligned(r, 0) : 0, use_b64 ? round_aligned(r, 1) : 0, use_b64 ? round_aligned(r, 2) : 0);
        fprintf(o, "}\n");
        fprintf(o, "#define LATCH_SLOT_TRIPLET_EX { \\\n");
        for (int r = 0; r < 8; ++r) for (int l = 0; l < 64; l += 16) {
            fprintf(o, "   ");
            for (int i = 0; i < 16; ++i) { const int t = slot_t[r][l + i]; fprintf(o, " %d,", t | (rot[t] << 9) | (swp[t] << 10)); }
            fprintf(o, " \\\n");
        }
        fprintf(o, "}\n");
        fclose(o);
    }
    return 0;
}
