"""Pins oracle/fit_oracle.py (the CPU restatement of `inference/fit_3D_gaussian.py`) to the reference's OWN outputs: for both demo
clips the reference ships the inputs of step 3 and what it wrote for them (tests/golden/demo_fit/README.md).  CPU only."""
import numpy as np
import pytest

from oracle import fit_oracle as fo

from _fit_fixtures import CLIPS, load_clip, tab20


def test_ellipse_element_5_is_opencvs():
    assert fo.ellipse_element(5).astype(int).tolist() == [[0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]


def test_threshold_closed_form_is_scipys():
    from scipy.stats import chi2
    assert fo.mahalanobis_threshold(0.97) == pytest.approx(chi2.ppf(0.97, df=2), rel=1e-12)


@pytest.mark.parametrize("clip", CLIPS)
def test_fit_matches_reference_json(clip):
    depth, K, masks, gold, _ = load_clip(clip)
    assert np.array_equal(np.array(gold["camera_info"]["intrinsic"], np.float32), K)
    E = np.eye(4, dtype=np.float32)
    assert gold["num_objects"] == len(masks)
    for oid, raw in masks.items():
        g = gold["gaussian_params"][str(oid)]
        m = fo.load_mask(raw)
        assert int(m.sum()) == g["num_mask_pixels"]                 # pins the threshold + the cv2 erosion, pixel-exact
        pts = fo.get_point_cloud_from_depth(depth, K, E, m)
        assert len(pts) == g["num_points"]
        mean, cov = fo.fit_3d_gaussian(pts)
        np.testing.assert_allclose(mean, np.array(g["mean"]), rtol=0, atol=2e-6)
        np.testing.assert_allclose(cov, np.array(g["cov"]), rtol=0, atol=2e-6 * np.abs(np.array(g["cov"])).max())
        assert np.trace(cov) == pytest.approx(g["trace"], rel=1e-5)
        np.testing.assert_allclose(np.linalg.eigvalsh(cov.astype(np.float64)), np.array(g["eigvals"]), rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize("clip", CLIPS)
def test_projection_picture_matches_reference_png(clip):
    depth, K, masks, gold, png = load_clip(clip)
    h, w = depth.shape
    params = {int(k): v for k, v in gold["gaussian_params"].items()}
    img, mask, idx = fo.visualize_gaussian_projections(params, K, np.eye(4, dtype=np.float32), (w, h), tab20())
    assert {str(k): v for k, v in idx.items()} == gold["obj_id_to_color_idx"]
    diff = np.abs(img.astype(int) - png.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-4            # the truncation to uint8 flips a handful of values by one
    assert mask.max() == 1.0 and 0.005 < mask.mean() < 0.5


def test_degenerate_inputs():
    assert fo.fit_3d_gaussian(np.zeros((2, 3), np.float32)) == (None, None)
    K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1]], np.float32)
    E = np.eye(4, dtype=np.float32)
    cov = np.eye(3, dtype=np.float32) * 0.01
    for mean in ([0, 0, 0.1], [100, 0, 1.0]):                       # behind the near plane; far off screen
        d, m, z = fo.project_gaussian_to_2d(np.array(mean, np.float32), cov, K, E, (640, 480))
        assert d.max() == 0 and np.isinf(m).all()
    depth = np.zeros((4, 5), np.float32); depth[1, 2] = 2.0
    pts = fo.get_point_cloud_from_depth(depth, np.array([[2, 0, 2], [0, 2, 2], [0, 0, 1]], np.float32), E)
    np.testing.assert_allclose(pts, [[0.0, -1.0, 2.0]])
