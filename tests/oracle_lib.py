"""ctypes access to the CPU oracle (oracle/liboracle.so) and the compiled reference feeders
(oracle/_ref/libref_feeder.so).  Test infrastructure only -- imported from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke(); never from the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

try:
    from . import synth
except ImportError:  # imported as a top-level module (bench.py / smoke)
    import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_u8p = C.POINTER(C.c_uint8)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all"], stdout=subprocess.DEVNULL)


def _ptr(a, ty=C.c_void_p):
    return a.ctypes.data_as(ty)


class Oracle:
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        self.lib.orc_k2nn_omp.restype = C.c_int
        self.lib.orc_k2nn_omp_kernel.restype = C.c_char_p
        self.lib.orc_fast9.restype = C.c_int
        self.lib.orc_feature_angle.restype = C.c_float
        self.lib.orc_latch_pattern.restype = C.POINTER(C.c_uint8)

    # -- K2NN
    def k2nn(self, Q, T, threshold, want_dist=False):
        Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
        T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 64)
        nq, nt = Q.shape[0], T.shape[0]
        m = np.empty(nq, dtype=np.int32)
        b = np.empty(nq, dtype=np.uint16); s = np.empty(nq, dtype=np.uint16)
        self.lib.orc_k2nn(_ptr(Q), C.c_int(nq), _ptr(T), C.c_int(nt), C.c_int(int(threshold)),
                          _ptr(m), _ptr(b), _ptr(s))
        return (m, b, s) if want_dist else m

    def k2nn_split(self, Q, T, threshold, nsplit):
        Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
        T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 64)
        m = np.empty(Q.shape[0], dtype=np.int32)
        self.lib.orc_k2nn_split(_ptr(Q), C.c_int(Q.shape[0]), _ptr(T), C.c_int(T.shape[0]),
                                C.c_int(int(threshold)), C.c_int(int(nsplit)), _ptr(m))
        return m

    def k2nn_omp_kernel(self):
        return self.lib.orc_k2nn_omp_kernel().decode()

    def k2nn_omp(self, Q, T, rule=0, threshold=40, ratio=0.8, kernel=-1):
        """kernel: 0 = 8 x popcount64 per pair, 1 = AVX-512 VPOPCNTDQ, -1 = auto."""
        Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
        T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 64)
        m = np.empty(Q.shape[0], dtype=np.int32)
        self.lib.orc_k2nn_omp_ex.restype = C.c_int
        nthr = self.lib.orc_k2nn_omp_ex(_ptr(Q), C.c_int(Q.shape[0]), _ptr(T), C.c_int(T.shape[0]), C.c_int(rule),
                                        C.c_int(int(threshold)), C.c_float(ratio), C.c_int(kernel), _ptr(m))
        return m, nthr

    def k2nn_omp_timed(self, Q, T, rule=0, threshold=40, ratio=0.8, kernel=-1, reps=7):
        """The same sweep `reps` times inside one parallel region; returns (matches, threads, best seconds of one sweep measured
        between two team barriers -- no thread wake-up in the figure)."""
        Q = np.ascontiguousarray(Q, dtype=np.uint8).reshape(-1, 64)
        T = np.ascontiguousarray(T, dtype=np.uint8).reshape(-1, 64)
        m = np.empty(Q.shape[0], dtype=np.int32)
        best = C.c_double(0.0)
        self.lib.orc_k2nn_omp_timed.restype = C.c_int
        nthr = self.lib.orc_k2nn_omp_timed(_ptr(Q), C.c_int(Q.shape[0]), _ptr(T), C.c_int(T.shape[0]), C.c_int(rule), C.c_int(int(threshold)),
                                           C.c_float(ratio), C.c_int(kernel), C.c_int(reps), _ptr(m), C.byref(best))
        return m, nthr, best.value

    def cpumatcher_pair(self, desc_i, xy_i, desc_j, xy_j, ratio=0.8, kernel=-1):
        """CPUMatcher::computeMatchesPair (CPUMatcher.hpp:67-76): regions_I = database, regions_J = queries.
        Returns (pairs[k,2] of (i_, j_), threads)."""
        di = np.ascontiguousarray(desc_i, dtype=np.uint8).reshape(-1, 64)
        dj = np.ascontiguousarray(desc_j, dtype=np.uint8).reshape(-1, 64)
        pi = np.ascontiguousarray(xy_i, dtype=np.float32).reshape(-1, 2)
        pj = np.ascontiguousarray(xy_j, dtype=np.float32).reshape(-1, 2)
        assert pi.shape[0] == di.shape[0] and pj.shape[0] == dj.shape[0]
        pairs = np.empty((max(dj.shape[0], 1), 2), dtype=np.int32)
        thr = C.c_int(0)
        self.lib.orc_cpumatcher_pair.restype = C.c_int
        k = self.lib.orc_cpumatcher_pair(_ptr(di), _ptr(pi), C.c_int(di.shape[0]), _ptr(dj), _ptr(pj), C.c_int(dj.shape[0]),
                                         C.c_float(ratio), C.c_int(kernel), _ptr(pairs), C.byref(thr))
        assert k >= 0
        return pairs[:k].copy(), thr.value

    def avx512_available(self):
        return bool(self.lib.orc_k2nn_avx512_available())

    # -- pyramid
    def pyramid_dims(self, W, H, scale_factor=1.2, levels=8):
        w = (C.c_uint32 * levels)(); h = (C.c_uint32 * levels)(); f = (C.c_float * levels)()
        self.lib.orc_pyramid_dims(C.c_uint32(W), C.c_uint32(H), C.c_float(scale_factor), C.c_int(levels), w, h, f)
        return list(w), list(h), [np.float32(v) for v in f]

    def lerp(self, img, f, neww, newh):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        H, W = img.shape
        out = np.zeros((newh, neww), dtype=np.uint8)
        self.lib.orc_lerp(_ptr(img), C.c_uint32(W), C.c_uint32(H), C.c_size_t(W), C.c_float(f), C.c_float(f),
                          _ptr(out), C.c_uint32(neww), C.c_uint32(newh), C.c_size_t(neww))
        return out

    def pyramid(self, img, scale_factor=1.2, levels=8):
        H, W = img.shape
        ws, hs, fs = self.pyramid_dims(W, H, scale_factor, levels)
        out = [np.ascontiguousarray(img, dtype=np.uint8)]
        for i in range(1, levels):
            out.append(self.lerp(img, float(fs[i]), ws[i], hs[i]))
        return out

    # -- CLATCH
    def clatch(self, levels, kps):
        n = len(kps)
        L = len(levels)
        levels = [np.ascontiguousarray(l, dtype=np.uint8) for l in levels]
        ptrs = (C.c_void_p * L)(*[l.ctypes.data for l in levels])
        w = (C.c_uint32 * L)(*[l.shape[1] for l in levels])
        h = (C.c_uint32 * L)(*[l.shape[0] for l in levels])
        p = (C.c_size_t * L)(*[l.shape[1] for l in levels])
        kps = np.ascontiguousarray(kps, dtype=synth.KP_DTYPE)
        desc = np.zeros((n, 64), dtype=np.uint8)
        self.lib.orc_clatch(ptrs, w, h, p, _ptr(kps), C.c_int(n), _ptr(desc))
        return desc

    def clatch_roi(self, level, kp):
        level = np.ascontiguousarray(level, dtype=np.uint8)
        kp = np.ascontiguousarray(kp, dtype=synth.KP_DTYPE).reshape(1)
        roi = np.zeros((64, 64), dtype=np.uint8)
        self.lib.orc_clatch_roi(_ptr(level), C.c_uint32(level.shape[1]), C.c_uint32(level.shape[0]),
                                C.c_size_t(level.shape[1]), _ptr(kp), _ptr(roi))
        return roi

    def latch_pattern(self):
        p = self.lib.orc_latch_pattern()
        return np.ctypeslib.as_array(p, shape=(512, 6)).copy()

    # -- feeders
    def fast9(self, img, threshold, cap=200000):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols = img.shape
        out = np.zeros(cap, dtype=synth.KP_DTYPE)
        n = self.lib.orc_fast9(_ptr(img), C.c_int(cols), C.c_int(rows), C.c_int(cols), C.c_uint8(threshold),
                               _ptr(out), C.c_int(cap))
        return out[:n]

    def feature_angle(self, img, px, py):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        return np.float32(self.lib.orc_feature_angle(_ptr(img), C.c_int(px), C.c_int(py), C.c_int(img.shape[1])))

    def features_from_kps(self, kps):
        kps = np.ascontiguousarray(kps, dtype=synth.KP_DTYPE)
        out = np.zeros((len(kps), 4), dtype=np.float32)
        self.lib.orc_features_from_kps(_ptr(kps), C.c_int(len(kps)), _ptr(out))
        return out

    # -- PnP
    def pnp_residuals(self, Rt, X, x, K):
        Rt = np.ascontiguousarray(Rt, dtype=np.float64).reshape(-1, 12)
        X = np.ascontiguousarray(X, dtype=np.float64).reshape(-1, 3)
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(9)
        H, N = Rt.shape[0], X.shape[0]
        err = np.zeros((H, N), dtype=np.float64)
        self.lib.orc_pnp_residuals(_ptr(Rt), C.c_int(H), _ptr(X), _ptr(x), C.c_int(N), _ptr(K), _ptr(err))
        return err

    def epipolar_residuals(self, F, x1, x2):
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, 9)
        x1 = np.ascontiguousarray(x1, dtype=np.float64).reshape(-1, 2)
        x2 = np.ascontiguousarray(x2, dtype=np.float64).reshape(-1, 2)
        err = np.zeros((F.shape[0], x1.shape[0]), dtype=np.float64)
        self.lib.orc_epipolar_residuals(_ptr(F), C.c_int(F.shape[0]), _ptr(x1), _ptr(x2), C.c_int(x1.shape[0]), _ptr(err))
        return err

    def pnp_score(self, err, thr2):
        err = np.ascontiguousarray(err, dtype=np.float64)
        H, N = err.shape
        cnt = np.zeros(H, dtype=np.int32); cost = np.zeros(H, dtype=np.float64)
        self.lib.orc_pnp_score(_ptr(err), C.c_int(H), C.c_int(N), C.c_double(thr2), _ptr(cnt), _ptr(cost))
        return cnt, cost


    # -- a-contrario RANSAC
    def acr_sample(self, seed, it, n_index, m):
        pos = (C.c_uint32 * 8)()
        self.lib.orc_acr_sample(C.c_uint64(int(seed)), C.c_uint32(int(it)), C.c_uint32(int(n_index)), C.c_int(m), pos)
        return [int(pos[j]) for j in range(m)]

    def acr_tables(self, n, m):
        a = np.zeros(n + 1, dtype=np.float32); b = np.zeros(n + 1, dtype=np.float32)
        self.lib.orc_acr_tables(C.c_int(n), C.c_int(m), _ptr(a), _ptr(b))
        return a, b

    def acr_best_nfa(self, err, m, max_models, logalpha0, mult):
        err = np.ascontiguousarray(err, dtype=np.float64)
        k = C.c_int()
        self.lib.orc_acr_best_nfa.restype = C.c_double
        v = self.lib.orc_acr_best_nfa(_ptr(err), C.c_int(err.shape[0]), C.c_int(m), C.c_int(max_models), C.c_double(logalpha0),
                                      C.c_double(mult), C.byref(k))
        return float(v), int(k.value)

    def acransac(self, kind, a, b, K1, fit, max_iteration=256, seed=1, precision=float("inf"), img_wh=(0, 0)):
        """Sequential AC-RANSAC.  kind 0: a = X (N,3), b = x (N,2); kind 1 (essential), 2 (fundamental), 3 (homography): a = x1,
        b = x2 in pixels.  `fit(sample) -> array (n_models, 12 | 18 | 9 | 9)` is the minimal solver (valid models only, solver order;
        kinds 2 / 3: models in the coordinates tv_normalize gives).  Returns a dict."""
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        K1 = np.ascontiguousarray(K1, dtype=np.float64).reshape(9)
        n = a.shape[0]
        md, m = {0: (12, 3), 1: (18, 5), 2: (9, 7), 3: (9, 4)}[kind]
        FIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double))
        calls = []

        def _fit(user, sample, out):
            smp = [int(sample[j]) for j in range(m)]
            calls.append(smp)
            models = np.ascontiguousarray(fit(smp), dtype=np.float64).reshape(-1, md)
            for k in range(models.shape[0]):
                for e in range(md):
                    out[k * md + e] = models[k, e]
            return int(models.shape[0])

        cb = FIT(_fit)
        model = np.zeros(md); inl = np.zeros(max(n, 1), dtype=np.uint32)
        n_inl, best_it, its = C.c_int(), C.c_int32(), C.c_int32()
        emax, nfa = C.c_double(), C.c_double()
        self.lib.orc_acransac.restype = C.c_int
        found = self.lib.orc_acransac(C.c_int(kind), _ptr(a), _ptr(b), C.c_int(n), _ptr(K1), C.c_int(int(img_wh[0])), C.c_int(int(img_wh[1])),
                                      C.c_int(int(max_iteration)), C.c_uint64(int(seed)), C.c_double(float(precision)),
                                      cb, None, _ptr(model), _ptr(inl), C.byref(n_inl), C.byref(emax),
                                      C.byref(nfa), C.byref(best_it), C.byref(its))
        return dict(found=bool(found), model=model, inliers=inl[:n_inl.value].copy(), error_max=emax.value, min_nfa=nfa.value,
                    best_iter=best_it.value, iterations=its.value, samples=calls)


    # ---- seven-point / four-point models (oracle/clc_oracle_twoview.c) ----
    def tv_normalize(self, wh, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 2)
        out = np.zeros_like(x)
        self.lib.orc_tv_normalize(C.c_int(int(wh[0])), C.c_int(int(wh[1])), _ptr(x), C.c_int(x.shape[0]), _ptr(out))
        return out

    def tv_unnormalize(self, homography, wh, Mn):
        Mn = np.ascontiguousarray(Mn, dtype=np.float64).reshape(9)
        out = np.zeros(9)
        self.lib.orc_tv_unnormalize(C.c_int(1 if homography else 0), C.c_int(int(wh[0])), C.c_int(int(wh[1])), _ptr(Mn), _ptr(out))
        return out.reshape(3, 3)

    def seven_point(self, q1, q2):
        q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(7, 2); q2 = np.ascontiguousarray(q2, dtype=np.float64).reshape(7, 2)
        F = np.zeros(27)
        self.lib.orc_seven_point.restype = C.c_int
        n = self.lib.orc_seven_point(_ptr(q1), _ptr(q2), _ptr(F))
        return [F[9 * k:9 * k + 9].copy() for k in range(n)]

    def four_point(self, q1, q2):
        q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(4, 2); q2 = np.ascontiguousarray(q2, dtype=np.float64).reshape(4, 2)
        H = np.zeros(9)
        self.lib.orc_four_point.restype = C.c_int
        self.lib.orc_four_point(_ptr(q1), _ptr(q2), _ptr(H))
        return H

    def tv_residuals(self, kind, M, q1, q2):
        q1 = np.ascontiguousarray(q1, dtype=np.float64).reshape(-1, 2); q2 = np.ascontiguousarray(q2, dtype=np.float64).reshape(-1, 2)
        M = np.ascontiguousarray(M, dtype=np.float64).reshape(9)
        e = np.zeros(q1.shape[0])
        self.lib.orc_tv_residuals(C.c_int(kind), _ptr(M), _ptr(q1), _ptr(q2), C.c_int(q1.shape[0]), _ptr(e))
        return e


class RefFeeder:
    """The reference's own KFAST.h / FeatureAngle.h, compiled into oracle/_ref/ (build container)."""

    def __init__(self):
        path = os.path.join(ORACLE_DIR, "_ref", "libref_feeder.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.lib.ref_kfast.restype = C.c_int
        self.lib.ref_feature_angle.restype = C.c_float
        assert self.lib.ref_sizeof_keypoint() == 20

    def kfast(self, img, threshold, multithreading=True, cap=200000):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols = img.shape
        # KFAST reads 32-byte vectors past row ends and 8 B per row in featureAngle: pad the buffer
        buf = np.zeros(rows * cols + 64, dtype=np.uint8)
        buf[:rows * cols] = img.reshape(-1)
        out = np.zeros(cap, dtype=synth.KP_DTYPE)
        n = self.lib.ref_kfast(_ptr(buf), C.c_int(cols), C.c_int(rows), C.c_int(cols), C.c_uint8(threshold),
                               C.c_int(1 if multithreading else 0), _ptr(out), C.c_int(cap))
        return out[:min(n, cap)]

    def feature_angle(self, img, px, py):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rows, cols = img.shape
        buf = np.zeros(rows * cols + 64, dtype=np.uint8)
        buf[:rows * cols] = img.reshape(-1)
        return np.float32(self.lib.ref_feature_angle(_ptr(buf), C.c_int(px), C.c_int(py), C.c_int(cols)))
