"""CPU: the renderer's oracle (oracle/render_oracle.py) and the product's HOST logic (versecrafter_amd/rendering/control_maps.py: file
readers, colours, camera arithmetic, the 3x3 Gaussian projection records, the icosphere) against fixtures recorded from the reference's
own inference/rendering_4D_control_maps.py (tests/golden/make_golden_render.py -> render_small.safetensors)."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from oracle import render_oracle as RO
from versecrafter_amd.rendering import control_maps as CM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    return load_file(os.path.join(ROOT, "tests", "golden", "render_small.safetensors"))


def test_oracle_depth_compositing_bit_exact(fx):
    rgb, depth = RO.composite_by_depth(fx["comp.bg_rgb"], fx["comp.bg_depth"], fx["comp.fg_rgb"], fx["comp.fg_depth"], fx["comp.fg_mask"].bool())
    assert torch.equal(rgb, fx["comp.out_rgb"]) and torch.equal(depth, fx["comp.out_depth"])
    mm = RO.merge_mask(fx["comp.bg_depth"], fx["comp.fg_depth"], fx["comp.bg_mask"].bool(), fx["comp.fg_mask"].bool())
    assert torch.equal(mm, fx["comp.merged_mask"])


def test_oracle_depth_visualisation_bit_exact(fx):
    frames = list(fx["comp.bg_depth"])
    lo, hi = RO.global_depth_range([frames, list(fx["comp.fg_depth"]), list(fx["comp.out_depth"])])
    assert [lo, hi] == fx["depth.range"].tolist()
    assert torch.equal(torch.stack(RO.depth_to_gray(frames, lo, hi)), fx["depth.gray_global"])
    assert torch.equal(torch.stack(RO.depth_to_gray(frames)), fx["depth.gray_auto"])
    assert torch.equal(torch.stack(RO.depth_to_gray([torch.zeros(20, 28)])), fx["depth.gray_empty"])
    assert list(RO.global_depth_range([[torch.zeros(4, 4)]])) == fx["depth.range_empty"].tolist()


def _frames(fx):
    out = []
    for f in range(3):
        ids = [str(int(i)) for i in fx[f"gauss.f{f}.ids"]]
        out.append({i: (fx[f"gauss.f{f}.means"][k], fx[f"gauss.f{f}.covs"][k]) for k, i in enumerate(ids)})
    col = {str(int(k)): int(v) for k, v in fx["gauss.color_idx"]}
    return out, col


def test_oracle_gaussian_projection_bit_exact(fx):
    K, E = fx["gauss.K"], fx["gauss.ext"]
    d = RO.density_map(fx["gauss.means0"], fx["gauss.covs0"], K, E[0][:3, :3], E[0][:3, 3:4], (48, 36))
    assert torch.equal(d, fx["gauss.density_sum"])
    d1 = RO.density_map(fx["gauss.means0"][1:2], fx["gauss.covs0"][1:2], K, E[0][:3, :3], E[0][:3, 3:4], (48, 36))
    assert torch.equal(d1, fx["gauss.density_1"])
    params, col = _frames(fx)
    for thr in (0.05, 0.003):
        rgbs, alphas = RO.project_gaussians(params, col, [K.numpy()] * 3, list(E.numpy()), (48, 36), threshold=thr)
        assert torch.equal(torch.stack(rgbs), fx[f"gauss.rgb_t{thr}"]) and torch.equal(torch.stack(alphas), fx[f"gauss.alpha_t{thr}"])
    blend = torch.stack([RO.blend_with_bg(r, a, b) for r, a, b in zip(rgbs, alphas, fx["gauss.bg"])])
    assert torch.equal(blend, fx["gauss.blend"])


def test_palette_and_colour_helpers_match_the_reference(fx):
    for i in range(22):
        assert torch.equal(RO.object_color(i, {i: i}, True), fx["color.float"][i])
        assert torch.equal(RO.object_color(i, {i: i}), fx["color.u8"][i])
        assert torch.equal(CM.get_object_color(i, {i: i}, "cpu", return_float=True), fx["color.float"][i])
        assert torch.equal(CM.get_object_color(i, {i: i}, "cpu"), fx["color.u8"][i])
    assert torch.equal(CM.get_object_color("x", {}, "cpu"), fx["color.u8"][0])          # unknown id -> colour 0
    assert np.array_equal(CM.COORD_TRANSFORM_CV2BLENDER, fx["coord.cv2blender"].numpy())


def test_camera_trajectory_reader_and_camera_arithmetic(fx, tmp_path):
    """custom_camera_trajectory.npz (`extrinsics` float [F,4,4], Blender camera-to-world) -> OpenCV world-to-camera."""
    p = tmp_path / "custom_camera_trajectory.npz"
    np.savez(p, extrinsics=fx["cam.c2w_blender"].numpy().astype(np.float64))
    w2c = CM.load_camera_trajectory(str(p), device="cpu")
    assert torch.equal(w2c, fx["cam.w2c_opencv"]) and torch.equal(RO.camera_trajectory(fx["cam.c2w_blender"].numpy()), fx["cam.w2c_opencv"])
    Ks = fx["gauss.K"].repeat(5, 1, 1)
    for fn in (CM.build_pytorch3d_camera_parameters, RO.p3d_cameras):
        R, T, focal, pp = fn(Ks, w2c.clone())
        assert torch.equal(R, fx["cam.p3d_R"]) and torch.equal(T, fx["cam.p3d_T"])
        assert torch.equal(focal, fx["cam.p3d_focal"]) and torch.equal(pp, fx["cam.p3d_pp"])


def test_ellipsoid_parameter_reader(fx, tmp_path):
    doc = json.loads(bytes(fx["ell.json"].tolist()).decode())
    p = tmp_path / "ell.json"
    p.write_text(json.dumps(doc))
    params, cidx, centers = CM.load_ellipsoid_parameters(str(p), device="cpu")
    assert cidx == {"3": 1, "5": 4} and sorted(centers) == [0, 1] and len(params) == 2
    for f in range(2):
        for j, oid in enumerate((3, 5)):
            assert torch.equal(params[f][oid][0], fx["ell.means"][f, j]) and torch.equal(params[f][oid][1], fx["ell.covs"][f, j])
            assert torch.equal(centers[f][oid], fx["ell.means"][f, j])
    op, oc = RO.ellipsoid_parameters(doc)
    assert oc == cidx and torch.equal(op[1][5][1], fx["ell.covs"][1, 1])


def test_host_gaussian_records_equal_the_oracle(fx):
    """The 12-float record the HIP kernels consume is the reference's own 3x3 arithmetic (:828-873), statement by statement."""
    K, E = fx["gauss.K"], fx["gauss.ext"]
    for f in range(3):
        R, t = E[f][:3, :3], E[f][:3, 3:4]
        for mean, cov in zip(fx[f"gauss.f{f}.means"], fx[f"gauss.f{f}.covs"]):
            rec, z = CM._gaussian_record(mean, cov, K, R, t)
            ok, m2, inv, coeff = RO.gaussian_record(mean, cov, K, R, t)
            assert z == float((R @ mean + t.squeeze())[2])
            assert bool(rec[7]) == ok
            if ok:
                got = torch.tensor(rec[:7])
                want = torch.cat([m2, inv.flatten(), coeff.reshape(1)])
                assert torch.equal(got, want)


def test_icosphere_is_a_closed_unit_sphere_mesh():
    for level, (nv, nf) in {0: (12, 20), 1: (42, 80), 3: (642, 1280)}.items():
        v, f = CM.ico_sphere(level)
        assert v.shape == (nv, 3) and f.shape == (nf, 3) and f.dtype == torch.int32
        assert torch.allclose(v.norm(dim=1), torch.ones(nv), atol=1e-6)
        edges = set()
        for a, b, c in f.tolist():
            edges |= {(min(a, b), max(a, b)), (min(b, c), max(b, c)), (min(a, c), max(a, c))}
        assert nv - len(edges) + nf == 2                            # Euler characteristic of a sphere
        n = torch.linalg.cross(v[f[:, 1].long()] - v[f[:, 0].long()], v[f[:, 2].long()] - v[f[:, 0].long()])
        assert ((n * v[f[:, 0].long()]).sum(1) > 0).all()           # consistently outward-facing
        vo, fo = RO.ico_sphere(level)
        assert torch.equal(v, vo) and torch.equal(f, fo)
    m = CM.make_ellipsoid_mesh(torch.tensor([1.0, 2.0, 3.0]), torch.diag(torch.tensor([0.04, 0.09, 0.25])), scale_factor=2.5, subdivisions=2,
                               color_rgb255=torch.tensor([10, 20, 30], dtype=torch.uint8), device="cpu")
    ext = (m.verts - torch.tensor([1.0, 2.0, 3.0])).abs().max(dim=0).values
    assert torch.allclose(ext, 2.5 * torch.tensor([0.2, 0.3, 0.5]), atol=2e-2)     # semi-axes = scale x sqrt(eigenvalues)
    assert torch.allclose(m.colors[0], torch.tensor([10, 20, 30]) / 255.0)
    both = CM.combine_meshes_for_scene([m, m])
    assert both.verts.shape[0] == 2 * m.verts.shape[0] and int(both.faces.max()) == 2 * m.verts.shape[0] - 1


def test_mask_dilation_matches_an_elliptical_structuring_element():
    m = torch.zeros(31, 31, dtype=torch.bool)
    m[15, 15] = True
    d = CM._dilate_ellipse(m, 10)
    ys, xs = torch.nonzero(d, as_tuple=True)
    assert d[15, 15] and int(d.sum()) > 60 and (ys - 15).abs().max() <= 5 and (xs - 15).abs().max() <= 5
    assert not d[10, 10] and not d[20, 20]                          # the corners of the 10 x 10 box are outside the ellipse
    pts = CM.depth_to_points(torch.full((4, 6), 2.0), torch.tensor([[3.0, 0, 2.5], [0, 4.0, 1.5], [0, 0, 1]]))
    assert torch.allclose(pts[1, 2], torch.tensor([(2 - 2.5) * 2 / 3, (1 - 1.5) * 2 / 4, 2.0]))


@pytest.mark.parametrize("clip", ["street", "indoor"])
def test_readers_on_the_reference_demo_files_and_the_step3_to_step5_convention(clip):
    """The reference's REAL files (tests/golden/demo_fit/README.md): its Blender step exported `custom_camera_trajectory.npz` and
    `custom_3D_gaussian_trajectory.json` from the `gaussian_params.json` its step 3 wrote.  The readers take them as they are, and in
    frame 0 (objects at rest) the exported Gaussians ARE the fitted ones under COORD_TRANSFORM_CV2BLENDER - which
    pins the world convention between the fit, the renderer and the trajectory files on reference data."""
    d = os.path.join(os.path.dirname(__file__), "golden", "demo_fit", clip)
    w2c = CM.load_camera_trajectory(os.path.join(d, "custom_camera_trajectory.npz"), device="cpu")
    assert w2c.shape == (81, 4, 4) and w2c.dtype == torch.float32
    # frame 0 is (all but a first step of) the camera of the input image: Blender world -> OpenCV camera = the inverse of CV2BLENDER
    np.testing.assert_allclose(w2c[0, :3, :3].numpy(), CM.COORD_TRANSFORM_CV2BLENDER.T, atol=1e-5)
    assert float(w2c[0, :3, 3].abs().max()) < 0.05
    assert torch.allclose(w2c[:, :3, :3] @ w2c[:, :3, :3].transpose(1, 2), torch.eye(3).expand(81, 3, 3), atol=1e-5)
    params, cidx, centers = CM.load_ellipsoid_parameters(os.path.join(d, "custom_3D_gaussian_trajectory_frames_0_1_40_80.json"), device="cpu")
    fit = json.load(open(os.path.join(d, "gaussian_params.json")))
    assert fit["obj_id_to_color_idx"].items() <= cidx.items() and len(params) == 4     # the indoor clip got a fourth object in Blender
    T = CM.COORD_TRANSFORM_CV2BLENDER.astype(np.float64)
    p0 = params[0]
    for oid, g in fit["gaussian_params"].items():
        mean, cov = p0[oid] if oid in p0 else p0[int(oid)]
        np.testing.assert_allclose(mean.numpy(), T @ np.array(g["mean"]), atol=1e-5)
        want = T @ np.array(g["cov"]) @ T.T
        if clip == "indoor":                 # this clip's ellipsoids were re-shaped by hand in Blender (axis-aligned): the size survives
            assert float(cov.trace()) == pytest.approx(np.trace(want), rel=1e-4)
            assert float((cov - torch.diag(torch.diag(cov))).abs().max()) == 0.0
        else:
            np.testing.assert_allclose(cov.numpy(), want, atol=2e-5)


def test_scripted_trajectory_files_have_the_reference_format(tmp_path):
    """tools/make_trajectory.py (step 4 of inference.sh without Blender): at rest it reproduces frame 0 of the reference's own Blender
    export for the street clip - same keys, same ids, means / covariances to float32 rounding - and the renderer's readers take its
    files; a scripted motion moves camera and objects linearly."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "tests", "golden", "demo_fit", "street")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "make_trajectory.py"), "--gaussian_json", os.path.join(d, "gaussian_params.json"),
                        "--output_dir", str(tmp_path), "--num_frames", "5", "--dolly", "2.0", "--yaw_deg", "90", "--object_shift", "2:1,0,0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    mine = json.load(open(tmp_path / "custom_3D_gaussian_trajectory.json"))
    demo = json.load(open(os.path.join(d, "custom_3D_gaussian_trajectory_frames_0_1_40_80.json")))
    assert mine["metadata"].keys() == demo["metadata"].keys() and mine["metadata"]["obj_id_to_color_idx"] == demo["metadata"]["obj_id_to_color_idx"]
    assert mine["frames"][0].keys() == demo["frames"][0].keys()
    for a, b in zip(mine["frames"][0]["objects"], demo["frames"][0]["objects"]):
        assert a.keys() == b.keys() and a["object_id"] == b["object_id"] and a["gaussian_3d"].keys() == b["gaussian_3d"].keys()
        np.testing.assert_allclose(a["gaussian_3d"]["mean"], b["gaussian_3d"]["mean"], atol=1e-6)
        np.testing.assert_allclose(a["gaussian_3d"]["covariance"], b["gaussian_3d"]["covariance"], atol=1e-6)
    params, cidx, centers = CM.load_ellipsoid_parameters(str(tmp_path / "custom_3D_gaussian_trajectory.json"), device="cpu")
    assert len(params) == 5 and cidx == demo["metadata"]["obj_id_to_color_idx"]
    key = "2" if "2" in params[0] else 2
    np.testing.assert_allclose((params[4][key][0] - params[0][key][0]).numpy(), [1, 0, 0], atol=1e-6)       # object 2 drifted, the others rest
    other = "1" if "1" in params[0] else 1
    assert torch.equal(params[4][other][0], params[0][other][0])
    w2c = CM.load_camera_trajectory(str(tmp_path / "custom_camera_trajectory.npz"), device="cpu")
    demo_w2c = CM.load_camera_trajectory(os.path.join(d, "custom_camera_trajectory.npz"), device="cpu")
    np.testing.assert_allclose(w2c[0].numpy(), np.vstack([np.hstack([CM.COORD_TRANSFORM_CV2BLENDER.T, np.zeros((3, 1))]), [[0, 0, 0, 1]]]), atol=1e-6)
    np.testing.assert_allclose(w2c[0, :3, :3].numpy(), demo_w2c[0, :3, :3].numpy(), atol=1e-5)               # the demo's frame-0 orientation
    # last frame: 2 units forward, turned 90 degrees to the left -> a world point 1 unit further ahead of the START pose (Blender +Y)
    # sits ... at the camera's right-hand side?  No: the camera turned left, so what was ahead is now on its right: x_cam > 0.
    p = w2c[4] @ torch.tensor([0.0, 3.0, 0.0, 1.0])
    np.testing.assert_allclose(p[:3].numpy(), [1.0, 0.0, 0.0], atol=1e-5)
