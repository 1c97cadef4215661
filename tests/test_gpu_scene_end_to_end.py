"""The whole hot path on geometrically consistent frames (BASELINE config[4] shape, SURVEY.md 8 rows a-1..a-8 + f-1, f-3):
a textured plane rendered from two camera poses -> pyramid + FAST + orientation + CLATCH on the GPU for both -> the first
view's features become the map (3-D points by back-projection, the role of the SfM scene in Localizer.hpp:59-75) ->
K2NN map match of the second view (GPUMatcher.hpp:174-178) -> a-contrario P3P + refinement (Localizer.hpp:77-108) ->
the recovered pose is the one the frame was rendered from.  No stage is mocked: the descriptors matched are the CLATCH
output of the rendered pixels."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

W, H = 640, 480
K = np.array([[520.0, 0, 320.0], [0, 520.0, 240.0], [0, 0, 1.0]])
PPU = 100.0                                    # texture pixels per world unit


def _features(ctx, img):
    kps, desc, found = ctx.detect_and_describe(img, capacity=6000)
    s = np.power(np.float32(1.2), kps["scale"].astype(np.float32))          # HIPDetector.hpp: Features()[i] = {s x, s y, ...}
    xy = np.stack([s * kps["x"], s * kps["y"]], axis=1).astype(np.float64)
    return kps, desc, xy


def _rot_angle_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))


@pytest.mark.parametrize("yaw,tilt,shift", [(0.05, (0.03, -0.02), (0.25, -0.15)), (0.30, (-0.06, 0.05), (-0.4, 0.3)),
                                            (-0.12, (0.0, 0.08), (0.1, 0.5))])
def test_localize_second_view_against_map_from_first(yaw, tilt, shift):
    from coloc_amd import Context
    tex = synth.plane_texture()
    ctx = Context(device=0, width=W, height=H, maxkp=6000, match_thresh=60)
    try:
        Ra, ta = synth.look_at_plane_pose((7.0, 7.0), 5.2)
        Rb, tb = synth.look_at_plane_pose((7.0 + shift[0], 7.0 + shift[1]), 5.0, yaw=yaw, tilt=tilt)
        img_a = synth.render_plane(tex, PPU, K, Ra, ta, W, H)
        img_b = synth.render_plane(tex, PPU, K, Rb, tb, W, H)
        kps_a, desc_a, xy_a = _features(ctx, img_a)
        kps_b, desc_b, xy_b = _features(ctx, img_b)
        assert len(kps_a) > 800 and len(kps_b) > 800
        # the map: view a's descriptors + their 3-D points on the plane
        Xmap = synth.backproject_to_plane(xy_a, K, Ra, ta)
        ctx.set_map(desc_a)
        m = ctx.match_map(desc_b, threshold=60)                     # m[i] = map index of query feature i, or -1
        sel = np.nonzero(m >= 0)[0]
        assert len(sel) > 150, len(sel)
        X, x = Xmap[m[sel]], xy_b[sel]
        # most accepted matches are geometrically right (reprojection under the TRUE pose within a few pixels: keypoints are
        # integer positions on coarser pyramid levels)
        proj = (X @ Rb.T + tb) @ K.T
        err = np.linalg.norm(proj[:, :2] / proj[:, 2:3] - x, axis=1)
        assert (err < 4.0).mean() > 0.7, (err < 4.0).mean()
        r = ctx.pnp_acransac(X, x, K, seed=3, refine=True)
        assert r["Rt"] is not None and len(r["inliers"]) > 0.6 * len(sel)
        Rt = np.asarray(r["Rt"]).reshape(3, 4)
        R_est, t_est = Rt[:, :3], Rt[:, 3]
        C_true, C_est = -Rb.T @ tb, -R_est.T @ t_est
        assert _rot_angle_deg(R_est, Rb) < 0.5
        assert np.linalg.norm(C_est - C_true) < 0.02 * 5.0           # 2 % of the camera height
    finally:
        ctx.close()
