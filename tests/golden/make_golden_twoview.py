"""Golden vectors of the oracle's a-contrario filter under the seven-point ('F') and four-point ('H') models
(oracle/clc_oracle_acr.c kinds 2 / 3 with the solvers of oracle/clc_oracle_twoview.c): tests/golden/twoview_models.npz.
OpenMVG is absent from the reference tree and the reference holds no fixture for this step, so these freeze the RESTATEMENT
(regression vectors, not reference outputs).  usage: python tests/golden/make_golden_twoview.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib
import twoview_host as tvh


def main():
    orc = oracle_lib.Oracle()
    out = {}
    for model, kind, planar, seed in (("F", 2, False, 501), ("H", 3, True, 502)):
        sc = tvh.scene(240, seed, planar=planar)
        q1, q2 = orc.tv_normalize(sc["wh"], sc["x1"]), orc.tv_normalize(sc["wh"], sc["x2"])
        fit = (lambda s: orc.seven_point(q1[s], q2[s])) if model == "F" else (lambda s: [orc.four_point(q1[s], q2[s])])
        r = orc.acransac(kind, sc["x1"], sc["x2"], np.eye(3), fit, max_iteration=128, seed=seed, img_wh=sc["wh"])
        assert r["found"]
        smp = np.array(r["samples"][0])
        mods = orc.seven_point(q1[smp], q2[smp]) if model == "F" else [orc.four_point(q1[smp], q2[smp])]
        out.update({model + "_x1": sc["x1"], model + "_x2": sc["x2"], model + "_wh": np.array(sc["wh"]), model + "_seed": np.array(seed),
                    model + "_inliers": r["inliers"], model + "_model": r["model"], model + "_error_max": np.array(r["error_max"]),
                    model + "_min_nfa": np.array(r["min_nfa"]), model + "_iterations": np.array(r["iterations"]),
                    model + "_best_iter": np.array(r["best_iter"]), model + "_first_sample": smp,
                    model + "_first_models": np.array(mods), model + "_q1": q1, model + "_q2": q2})
    np.savez_compressed(os.path.join(HERE, "twoview_models.npz"), **out)
    print("written", os.path.join(HERE, "twoview_models.npz"))


if __name__ == "__main__":
    main()
