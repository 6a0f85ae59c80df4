"""4D control-map renderer kernels (csrc/render.hip behind vc_op_render_*, Python mirror versecrafter_amd/rendering/control_maps.py).

Pinned stages: against fixtures recorded from the reference's own inference/rendering_4D_control_maps.py (make_golden_render.py).
  byte / comparison work (depth compositing, merged mask): bit-exact;
  float32 maps (densities, alphas): |err| <= 2e-6 + 1e-5 |x|  (device expf vs the CPU's; everything else is the same fp32 arithmetic);
  uint8 images derived from them by truncation: at most 1 level on at most 0.5 % of the values.
Unpinned stages (PyTorch3D restatements): against oracle/render_oracle.py, which restates the same published algorithms on the CPU."""
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from oracle import render_oracle as RO

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    return load_file(os.path.join(ROOT, "tests", "golden", "render_small.safetensors"))


@pytest.fixture(scope="module")
def CM():
    from versecrafter_amd.rendering import control_maps
    return control_maps


def dev(t):
    return t.cuda()


def close_u8(got, want, frac=0.005):
    d = (got.cpu().int() - want.int()).abs()
    assert int(d.max()) <= 1, int(d.max())
    assert float((d > 0).float().mean()) <= frac, float((d > 0).float().mean())


def close_f32(got, want):
    got, want = got.cpu(), want
    assert torch.all((got - want).abs() <= 2e-6 + 1e-5 * want.abs()), float((got - want).abs().max())


def test_depth_compositing_and_merged_mask_bit_exact(fx, CM):
    rgb, depth = CM.composite_by_depth_batch(dev(fx["comp.bg_rgb"]), dev(fx["comp.bg_depth"]), dev(fx["comp.fg_rgb"]), dev(fx["comp.fg_depth"]),
                                             dev(fx["comp.fg_mask"]).bool())
    assert torch.equal(rgb.cpu(), fx["comp.out_rgb"]) and torch.equal(depth.cpu(), fx["comp.out_depth"])
    one_rgb, one_d = CM.composite_by_depth(dev(fx["comp.bg_rgb"][1]), dev(fx["comp.bg_depth"][1]), dev(fx["comp.fg_rgb"][1]),
                                           dev(fx["comp.fg_depth"][1]), dev(fx["comp.fg_mask"][1]).bool())
    assert torch.equal(one_rgb.cpu(), fx["comp.out_rgb"][1]) and torch.equal(one_d.cpu(), fx["comp.out_depth"][1])
    mm = CM.merge_bg_and_fg_mask(list(dev(fx["comp.bg_depth"])), list(dev(fx["comp.fg_depth"])), list(dev(fx["comp.bg_mask"]).bool()),
                                 list(dev(fx["comp.fg_mask"]).bool()))
    assert torch.equal(torch.stack(mm).cpu(), fx["comp.merged_mask"])
    seq = CM.merge_bg_and_fg_sequences(list(dev(fx["comp.bg_rgb"])), list(dev(fx["comp.bg_depth"])), [None] * 3, list(dev(fx["comp.fg_rgb"])),
                                       list(dev(fx["comp.fg_depth"])), list(dev(fx["comp.fg_mask"]).bool()))
    assert torch.equal(torch.stack(seq[0]).cpu(), fx["comp.out_rgb"]) and torch.equal(torch.stack(seq[1]).cpu(), fx["comp.out_depth"])
    with pytest.raises(RuntimeError):
        CM.composite_by_depth_batch(fx["comp.bg_rgb"], fx["comp.bg_depth"], fx["comp.fg_rgb"], fx["comp.fg_depth"], fx["comp.fg_mask"].bool())


def test_depth_visualisation(fx, CM):
    frames = list(dev(fx["comp.bg_depth"]))
    lo, hi = CM.compute_global_depth_range([frames, list(dev(fx["comp.fg_depth"])), list(dev(fx["comp.out_depth"]))])
    want_lo, want_hi = fx["depth.range"].tolist()
    assert abs(lo - want_lo) <= 1e-6 * want_lo and abs(hi - want_hi) <= 1e-6 * want_hi          # torch.quantile on the device
    close_u8(torch.stack(CM.visualize_depth_as_grayscale(frames, want_lo, want_hi)), fx["depth.gray_global"], frac=0.002)
    close_u8(torch.stack(CM.visualize_depth_as_grayscale(frames)), fx["depth.gray_auto"], frac=0.01)
    assert torch.equal(torch.stack(CM.visualize_depth_as_grayscale([torch.zeros(20, 28, device="cuda")])).cpu(), fx["depth.gray_empty"])
    assert list(CM.compute_global_depth_range([[torch.zeros(4, 4, device="cuda")]])) == fx["depth.range_empty"].tolist()


def _frames(fx):
    out = []
    for f in range(3):
        ids = [str(int(i)) for i in fx[f"gauss.f{f}.ids"]]
        out.append({i: (fx[f"gauss.f{f}.means"][k], fx[f"gauss.f{f}.covs"][k]) for k, i in enumerate(ids)})
    return out, {str(int(k)): int(v) for k, v in fx["gauss.color_idx"]}


def test_gaussian_density_projection_and_blending(fx, CM):
    K, E = fx["gauss.K"], fx["gauss.ext"]
    d = CM.compute_probability_density_map_gpu(fx["gauss.means0"], fx["gauss.covs0"], K, E[0][:3, :3], E[0][:3, 3:4], (48, 36))
    close_f32(d, fx["gauss.density_sum"])
    d1, z1 = CM.project_gaussian_to_2d_gpu(fx["gauss.means0"][1], fx["gauss.covs0"][1], K, E[0][:3, :3], E[0][:3, 3:4], (48, 36))
    close_f32(d1, fx["gauss.density_1"])
    assert z1 == fx["gauss.z_1"].item()
    params, col = _frames(fx)
    for thr in (0.05, 0.003):
        rgbs, alphas = CM.project_3d_gaussians_to_2d(params, col, [K.numpy()] * 3, list(E.numpy()), (48, 36), threshold=thr)
        close_f32(torch.stack(alphas), fx[f"gauss.alpha_t{thr}"])
        close_u8(torch.stack(rgbs), fx[f"gauss.rgb_t{thr}"])
    # blending of the reference's own projection (so that only this stage is under test): bit-exact byte arithmetic
    ref_rgb, ref_alpha = list(dev(fx["gauss.rgb_t0.003"])), list(dev(fx["gauss.alpha_t0.003"]))
    blend = CM.blend_gaussian_projection_with_bg(ref_rgb, ref_alpha, list(dev(fx["gauss.bg"])))
    close_u8(torch.stack(blend), fx["gauss.blend"], frac=0.002)
    masked = CM.mask_gaussian_projection(ref_rgb, ref_alpha)
    want = torch.stack([((r.float() / 255.0) * a.unsqueeze(-1) * 255).to(torch.uint8) for r, a in zip(fx["gauss.rgb_t0.003"], fx["gauss.alpha_t0.003"])])
    close_u8(torch.stack(masked), want, frac=0.002)
    empty_rgb, empty_a = CM.project_3d_gaussians_to_2d([{}], {}, [K.numpy()], [E[0].numpy()], (48, 36))
    assert int(empty_rgb[0].max()) == 0 and float(empty_a[0].max()) == 0.0


def _cams():
    K = torch.tensor([[60.0, 0, 32.0], [0, 60.0, 24.0], [0, 0, 1]])
    w2c = torch.eye(4)
    a = 0.15
    w2c[:3, :3] = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
    w2c[:3, 3] = torch.tensor([0.1, -0.05, 0.3])
    return K, w2c


def test_point_cloud_rasteriser_vs_oracle_specification(CM):
    """UNPINNED stage (PyTorch3D PointsRasterizer + AlphaCompositor restated): a few hundred points incl. coincident projections,
    points behind the camera and off-screen, radii below and above one pixel -- HIP vs the CPU restatement of the same algorithm."""
    g = torch.Generator().manual_seed(3)
    K, w2c = _cams()
    H, W = 48, 64
    pts = torch.cat([torch.randn(300, 3, generator=g) * torch.tensor([0.8, 0.6, 0.5]) + torch.tensor([0.0, 0.0, 2.5]),
                     torch.tensor([[0.0, 0.0, -1.0], [50.0, 0.0, 2.0]])])
    pts[10] = pts[11] + torch.tensor([0.0, 0.0, 0.3])                      # same pixel, different depth
    col = torch.randint(0, 256, (len(pts), 3), generator=g, dtype=torch.uint8)
    for radius in (0.02, 0.08):
        rgb, depth, mask = CM.render_point_cloud_pytorch3d_batch(pts.cuda(), col.cuda(), K[None].cuda(), w2c[None].cuda(), (H, W), point_size=radius)
        o_rgb, o_depth, o_mask = RO.render_points(pts, col, w2c, K, H, W, radius)
        assert torch.equal(mask[0].cpu(), o_mask)
        assert torch.allclose(depth[0].cpu(), o_depth, rtol=1e-5, atol=1e-6)
        close_u8(rgb[0], o_rgb, frac=0.01)
        assert int(mask.sum()) > 50 and (rgb[0][~mask[0]] == 127).all() and (depth[0][~mask[0]] == 0).all()
    rgb, depth, mask = CM.render_point_cloud_pytorch3d_batch(torch.zeros(0, 3).cuda(), torch.zeros(0, 3, dtype=torch.uint8).cuda(), K[None].cuda(),
                                                             w2c[None].cuda(), (H, W))
    assert not mask.any() and (rgb == 127).all()


def test_mesh_rasteriser_vs_oracle_specification(CM):
    """UNPINNED stage (MeshRasterizer + HardPhongShader restated): two overlapping ellipsoids; HIP vs the CPU restatement; the mask
    is the analytic ellipse silhouette up to the tessellation."""
    K, w2c = _cams()
    H, W = 48, 64
    m1 = CM.make_ellipsoid_mesh(torch.tensor([0.0, 0.0, 2.5]).cuda(), torch.diag(torch.tensor([0.05, 0.02, 0.03])), 2.5, 2,
                                torch.tensor([31, 119, 180], dtype=torch.uint8))
    m2 = CM.make_ellipsoid_mesh(torch.tensor([0.25, 0.1, 2.0]).cuda(), torch.diag(torch.tensor([0.01, 0.03, 0.01])), 2.5, 2,
                                torch.tensor([255, 127, 14], dtype=torch.uint8))
    host_mesh = CM.make_ellipsoid_mesh(torch.tensor([0.0, 0.0, 2.5]), torch.eye(3) * 0.01)       # as the reference: the mean's device decides
    with pytest.raises(RuntimeError):
        CM.render_meshes_pytorch3d_batch([host_mesh], K[None].cuda(), w2c[None].cuda(), (H, W))
    scene = CM.combine_meshes_for_scene([m1, m2])
    rgb, depth, mask = CM.render_meshes_pytorch3d_batch([scene, None], K.repeat(2, 1, 1).cuda(), w2c.repeat(2, 1, 1).cuda(), (H, W))
    o_rgb, o_depth, o_mask = RO.render_mesh(scene.verts.cpu(), scene.colors.cpu(), scene.faces.cpu(), w2c, K, H, W)
    diff = mask[0].cpu() != o_mask
    assert int(diff.sum()) <= 2                                            # a pixel centre within rounding of an edge
    same = ~diff & o_mask
    assert torch.allclose(depth[0].cpu()[same], o_depth[same], rtol=1e-4, atol=1e-5)
    d = (rgb[0].cpu().int() - o_rgb.int()).abs()[same]
    assert int(d.max()) <= 2 and float((d > 0).float().mean()) < 0.05
    assert 150 < int(mask[0].sum()) < 1500 and not mask[1].any() and int(rgb[1].max()) == 0
    assert float(depth[0][mask[0]].min()) > 1.5 and float(depth[0][mask[0]].max()) < 3.5
    assert (rgb[0][~mask[0]] == 0).all()


def test_full_sequence_flow_runs_at_the_clip_size(CM):
    """render_video_with_bg_and_fg + the depth / mask / Gaussian stages on an 81-frame 480 x 832 sequence (the bench clip): shapes,
    dtypes, and the properties main() relies on -- the composite is the background where no ellipsoid is in front."""
    g = torch.Generator().manual_seed(0)
    F_, H, W = 81, 480, 832
    K = torch.tensor([[700.0, 0, W / 2], [0, 700.0, H / 2], [0, 0, 1]])
    Ks = K.repeat(F_, 1, 1).cuda()
    Ts = torch.eye(4).repeat(F_, 1, 1)
    Ts[:, 0, 3] = torch.linspace(0, 0.3, F_)
    Ts = Ts.cuda()
    depth0 = 2.0 + torch.rand(H, W, generator=g)
    pts = CM.depth_to_points(depth0, K).reshape(-1, 3).cuda()
    cols = torch.randint(0, 256, (H * W, 3), generator=g, dtype=torch.uint8).cuda()
    mesh = CM.make_ellipsoid_mesh(torch.tensor([0.0, 0.0, 1.5]).cuda(), torch.diag(torch.tensor([0.02, 0.02, 0.02])), 2.5, 3,
                                  torch.tensor([255, 0, 0], dtype=torch.uint8))
    rgb, dep, bgm, fgm = CM.render_video_with_bg_and_fg(pts, cols, [mesh] * F_, Ks, Ts, (H, W), mode="full", point_size=0.005)
    assert len(rgb) == F_ and rgb[0].shape == (H, W, 3) and rgb[0].dtype == torch.uint8 and dep[0].dtype == torch.float32
    assert float(bgm[0].float().mean()) > 0.9 and 0.05 < float(fgm[40].float().mean()) < 0.4
    assert torch.all(dep[40][fgm[40]] < 1.6) and torch.all(dep[40][fgm[40]] > 1.0)          # the ellipsoid is in front of the cloud
    lo, hi = CM.compute_global_depth_range([dep])
    gray = CM.visualize_depth_as_grayscale(dep, lo, hi)
    assert gray[0].shape == (H, W, 3) and int(gray[40][fgm[40]].float().mean()) > int(gray[40][~fgm[40] & bgm[40]].float().mean())
    params = [{"1": (torch.tensor([0.0, 0.0, 1.5]), 0.02 * torch.eye(3))}] * F_
    g_rgb, g_a = CM.project_3d_gaussians_to_2d(params, {"1": 6}, [K.numpy()] * F_, list(Ts.cpu().numpy()), (W, H), threshold=0.003)
    assert len(g_rgb) == F_ and float(g_a[0].max()) > 0.99 and float(g_a[0].min()) == 0.0
    out = CM.blend_gaussian_projection_with_bg(g_rgb, g_a, rgb)
    assert out[0].shape == (H, W, 3)


def test_renderer_cli_writes_the_five_control_maps_the_inference_cli_reads(tmp_path):
    """inference/rendering_4D_control_maps.py end to end on a synthetic scene (PNG + depth npz + object mask + camera trajectory +
    ellipsoid json, the file set of the reference's demo_data folders): the five control videos come out under the names
    versecrafter_inference.py looks for (CLI.py:351-403), as frame dumps when the image has no codec."""
    import json
    import subprocess
    import sys
    from PIL import Image
    rs = np.random.RandomState(0)
    H, W, F_ = 64, 96, 5
    Image.fromarray(rs.randint(0, 255, (H, W, 3), dtype=np.uint8)).save(tmp_path / "0001.png")
    np.savez(tmp_path / "0001.npz", depth=(2.0 + rs.rand(H, W)).astype(np.float32),
             intrinsic=np.array([[0.8, 0, 0.5], [0, 1.2, 0.5], [0, 0, 1]], dtype=np.float32))
    (tmp_path / "masks").mkdir()
    m = np.zeros((H, W), dtype=np.uint8)
    m[20:40, 30:60] = 255
    Image.fromarray(m).save(tmp_path / "masks" / "obj1.png")
    c2w = np.tile(np.eye(4), (F_, 1, 1))
    c2w[:, :3, :3] = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)     # Blender camera at the origin looking along world +Y, up +Z
    c2w[:, 0, 3] = np.linspace(0, 0.1, F_)
    np.savez(tmp_path / "custom_camera_trajectory.npz", extrinsics=c2w)
    doc = {"metadata": {"num_frames": F_, "num_objects": 1, "obj_id_to_color_idx": {"1": 2}},
           "frames": [{"frame_index": f, "objects": [{"object_id": 1, "gaussian_3d": {"mean": [0.05 * f, 2.0, 0.0],
                                                                                     "covariance": [[0.02, 0, 0], [0, 0.02, 0], [0, 0, 0.03]]}}]}
                      for f in range(F_)]}
    (tmp_path / "ell.json").write_text(json.dumps(doc))
    out = tmp_path / "rendering_4D_maps"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference", "rendering_4D_control_maps.py"), "--png_path", str(tmp_path / "0001.png"),
                        "--npz_path", str(tmp_path / "0001.npz"), "--mask_dir", str(tmp_path / "masks"), "--trajectory_npz",
                        str(tmp_path / "custom_camera_trajectory.npz"), "--ellipsoid_json", str(tmp_path / "ell.json"), "--output_dir", str(out),
                        "--ellipsoid_subdiv", "2", "--point_size", "0.05"], capture_output=True, text=True, timeout=300)     # 1.6 px at 64 rows
    assert r.returncode == 0, r.stderr[-3000:]
    from versecrafter_amd.utils import video_io
    for name in ("background_RGB", "background_depth", "3D_gaussian_RGB", "3D_gaussian_depth", "merged_mask", "background_and_3D_gaussian"):
        v = video_io.read_video(str(out / f"{name}.mp4"), F_, (H, W))
        assert v.shape == (1, 3, F_, H, W) and torch.isfinite(v).all(), name
    fg = video_io.read_video(str(out / "3D_gaussian_depth.mp4"), F_, (H, W))
    mask = video_io.read_video(str(out / "merged_mask.mp4"), F_, (H, W))
    assert float(fg.max()) > 0.2 and 0.0 < float(mask.mean()) < 1.0           # the ellipsoid is in view; the mask has both values
