"""Lane-level emulations of the reference's two 32-lane butterflies, written from a reading of
src/CUDAK2NN.cu:47-75 and src/CLATCH.cu:169-188.  They restate WHAT EACH LANE HOLDS after every
shuffle step (numpy arrays indexed by lane), independently of the order-free restatement in
oracle/clc_oracle.c, and are used to cross-check it (tests/test_oracle_k2nn.py,
tests/test_oracle_clatch.py).  Test infrastructure only."""
import numpy as np

LANES = np.arange(32)


def _shfl_xor(v, m):
    return v[LANES ^ m]


def _popc64(x):
    return np.array([bin(int(v)).count("1") for v in x], dtype=np.uint32)


def k2nn_warp(Qw, T, threshold):
    """One warp = 32 queries.  Qw: (32, 8) uint64 query words (zero rows past num_q, like the border
    texture), T: (nt, 8) uint64.  Returns (match[32], best_v[32], second_v[32])."""
    tx = LANES
    # CUDAK2NN.cu:49-52: lane holds word (tx & 7) of the 8 queries (tx >> 3)*8 + i, i = 0..7
    q = np.stack([Qw[(tx >> 3) * 8 + i, tx & 7] for i in range(8)], 0)      # (8, 32)
    best_i = np.full(32, -1, dtype=np.int64)
    best_v = np.full(32, 100000, dtype=np.int64)
    second_v = np.full(32, 200000, dtype=np.int64)
    for t in range(T.shape[0]):
        train = T[t, tx & 7]                                                  # :47,:60
        # :58 __byte_perm(popc(q[i]^train), popc(q[i+4]^train), 0x5410): two 16-bit halves
        dist = [(_popc64(q[i] ^ train) | (_popc64(q[i + 4] ^ train) << np.uint32(16))).astype(np.uint32) for i in range(4)]
        dist = [d + _shfl_xor(d, 1) for d in dist]                            # :59
        odd = (tx & 1) != 0
        dist[0] = np.where(odd, dist[1], dist[0])                             # :61
        dist[2] = np.where(odd, dist[3], dist[2])                             # :62
        dist[0] = dist[0] + _shfl_xor(dist[0], 2)                             # :63
        dist[2] = dist[2] + _shfl_xor(dist[2], 2)                             # :64
        dist[0] = np.where((tx & 2) != 0, dist[2], dist[0])                   # :65
        s = dist[0] + _shfl_xor(dist[0], 4)                                   # :66
        d = np.where((tx & 4) != 0, s >> np.uint32(16), s & np.uint32(0xFFFF)).astype(np.int64)
        second_v = np.minimum(d, second_v)                                    # :67
        better = d < best_v                                                   # :68-72
        second_v = np.where(better, best_v, second_v)
        best_i = np.where(better, t, best_i)
        best_v = np.where(better, d, best_v)
    thr = threshold & 0xFF                                                    # uint8_t parameter, :46
    match = np.where(second_v - best_v > thr, best_i, -1)                     # :75
    return match.astype(np.int32), best_v, second_v


def k2nn_emulated(Q, T, threshold):
    Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
    T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 64)
    nq = Q.shape[0]
    Qw = Q.view("<u8").reshape(nq, 8)
    Tw = T.view("<u8").reshape(-1, 8)
    pad = (-nq) % 32
    Qp = np.concatenate([Qw, np.zeros((pad, 8), dtype=np.uint64)], 0)
    out = np.empty(nq + pad, dtype=np.int32)
    for w in range(0, nq + pad, 32):
        out[w:w + 32] = k2nn_warp(Qp[w:w + 32], Tw, threshold)[0]
    return out[:nq]


def clatch_bits_emulated(roi72, triplets):
    """roi72: (64, 72) uint8 window as stored in s_ROI (CLATCH.cu:158,167); triplets: (512, 3) int
    byte offsets (a, b, c) = row*72 + col.  Returns the 16 uint32 descriptor words, following the
    per-thread pixel split (:169, :174-178), the 8-lane reduce-transpose (:180-183), the 8/16
    shuffles and the bit placement (:184-187)."""
    flat = roi72.reshape(-1).astype(np.int64)
    tx = LANES
    roi_base = 144 * (tx & 3) + (tx >> 2)                                     # :169
    words = np.zeros(16, dtype=np.uint32)
    for ty in range(16):
        desc = np.zeros(32, dtype=np.uint32)
        for i in range(4):
            tb = ty * 32 + i * 8
            accum = []
            for j in range(8):
                a, b, c = (int(v) for v in triplets[tb + j])
                b1 = flat[roi_base + b]; b2 = flat[roi_base + b + 72]
                a1 = flat[roi_base + a] - b1; a2 = flat[roi_base + a + 72] - b2
                c1 = flat[roi_base + c] - b1; c2 = flat[roi_base + c + 72] - b2
                accum.append(a1 * a1 - c1 * c1 + a2 * a2 - c2 * c2)           # :178
            k = 1
            while k <= 4:                                                     # :180-183
                for s in range(0, 8, k):
                    accum[s] = accum[s] + _shfl_xor(accum[s], k)
                sel = (tx & k) != 0
                for s in range(0, 8, 2 * k):
                    accum[s] = np.where(sel, accum[s + k], accum[s])
                k <<= 1
            accum[0] = accum[0] + _shfl_xor(accum[0], 8)                      # :184
            tot = accum[0] + _shfl_xor(accum[0], 16)
            desc |= ((tot < 0).astype(np.uint32) << ((i << 3) + (tx & 7)).astype(np.uint32))   # :185
        for s in (1, 2, 4):                                                   # :187
            desc |= _shfl_xor(desc, s)
        words[ty] = desc[0]                                                   # :188 (threadIdx.x == 0)
    return words
