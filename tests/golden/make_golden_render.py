#!/usr/bin/env python3
"""Golden vectors for the 4D control-map renderer's PURE-TORCH functions, recorded by running the reference's own
inference/rendering_4D_control_maps.py in the build container (CPU, float32).

Run here only (the reference does not exist on the GPU box):   python tests/golden/make_golden_render.py
Writes tests/golden/render_small.safetensors.  Nothing of the reference is stored: the fixture holds inputs and outputs.

Third-party imports of that file that are absent from the image are replaced by EMPTY stand-ins (cv2, kornia, pytorch3d,
torchvision): none of the functions recorded here calls into them, except
  * _build_cam_from_extrinsics, which ends in PerspectiveCameras(...): the stand-in records the keyword arguments, so the
    reference's own camera arithmetic (two matrix inversions, the sign flip, the transposition) is what is pinned;
  * get_object_color, which reads matplotlib's tab20 palette (matplotlib is importable here and is the real one).
What stays UNPINNED: everything behind pytorch3d (ico_sphere, MeshRasterizer + HardPhongShader, PointsRasterizer +
AlphaCompositor), cv2 (imread, resize, dilate) and kornia.depth_to_3d_v2 -- see versecrafter_amd/rendering/control_maps.py."""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/inference"


class _Recorder:
    def __init__(self, *a, **kw):
        self.args, self.kwargs = a, kw


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    _stub("cv2")
    _stub("kornia"); _stub("kornia.geometry"); _stub("kornia.geometry.depth", depth_to_3d_v2=None)
    names = ["Pointclouds", "Meshes", "join_meshes_as_batch"]
    _stub("pytorch3d"); _stub("pytorch3d.structures", **{n: _Recorder for n in names})
    rn = ["PerspectiveCameras", "PointsRasterizationSettings", "PointsRenderer", "PointsRasterizer", "AlphaCompositor", "MeshRenderer",
          "MeshRasterizer", "RasterizationSettings", "HardPhongShader", "TexturesVertex", "PointLights"]
    _stub("pytorch3d.renderer", **{n: type(n, (_Recorder,), {}) for n in rn})
    _stub("pytorch3d.utils", ico_sphere=None)
    _stub("torchvision"); _stub("torchvision.io", write_video=None, read_video=None)
    _stub("torchvision.transforms"); _stub("torchvision.transforms.functional")
    sys.path.insert(0, REF)
    import rendering_4D_control_maps as R
    return R


def main():
    R = import_reference()
    g = torch.Generator().manual_seed(11)
    out = {}
    B, H, W = 3, 20, 28

    # ---- depth compositing (composite_by_depth_batch :398, merge_bg_and_fg_mask :736) ----
    bg_rgb = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    fg_rgb = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    bg_depth = torch.rand(B, H, W, generator=g) * 5 * (torch.rand(B, H, W, generator=g) > 0.2)
    fg_depth = torch.rand(B, H, W, generator=g) * 5 * (torch.rand(B, H, W, generator=g) > 0.3)
    fg_depth[0, :4] = bg_depth[0, :4] - 5e-7                     # inside the 1e-6 guard band
    fg_mask = torch.rand(B, H, W, generator=g) > 0.4
    bg_mask = torch.rand(B, H, W, generator=g) > 0.5
    o_rgb, o_depth = R.composite_by_depth_batch(bg_rgb, bg_depth, fg_rgb, fg_depth, fg_mask)
    out.update({"comp.bg_rgb": bg_rgb, "comp.fg_rgb": fg_rgb, "comp.bg_depth": bg_depth, "comp.fg_depth": fg_depth,
                "comp.fg_mask": fg_mask.to(torch.uint8), "comp.bg_mask": bg_mask.to(torch.uint8), "comp.out_rgb": o_rgb,
                "comp.out_depth": o_depth})
    mm = R.merge_bg_and_fg_mask(list(bg_depth), list(fg_depth), list(bg_mask), list(fg_mask), device="cpu")
    out["comp.merged_mask"] = torch.stack(mm)

    # ---- depth visualisation (:487) and the global range (:541) ----
    frames = [bg_depth[i] for i in range(B)]
    lo, hi = R.compute_global_depth_range([frames, [fg_depth[i] for i in range(B)], [o_depth[i] for i in range(B)]])
    out["depth.range"] = torch.tensor([lo, hi], dtype=torch.float64)
    out["depth.gray_global"] = torch.stack(R.visualize_depth_as_grayscale(frames, lo, hi))
    out["depth.gray_auto"] = torch.stack(R.visualize_depth_as_grayscale(frames))             # quantile path (< 1M samples: no sampling)
    out["depth.gray_empty"] = torch.stack(R.visualize_depth_as_grayscale([torch.zeros(H, W)]))
    e_lo, e_hi = R.compute_global_depth_range([[torch.zeros(H, W)]])
    out["depth.range_empty"] = torch.tensor([e_lo, e_hi], dtype=torch.float64)

    # ---- Gaussian projection (:765, :801) and compositing (:573), blending (:697), colours (:885) ----
    Wi, Hi = 48, 36
    K = torch.tensor([[40.0, 0, 24.0], [0, 42.0, 18.0], [0, 0, 1]])
    nF = 3
    ext, params = [], []
    for f in range(nF):
        a = 0.1 * f
        Rm = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=torch.float32)
        E = torch.eye(4)
        E[:3, :3] = Rm
        E[:3, 3] = torch.tensor([0.05 * f, -0.02, 0.1])
        ext.append(E.numpy())
        fr = {}
        for j, oid in enumerate(["1", "2", "7"]):
            A = torch.randn(3, 3, generator=g) * 0.25
            cov = A @ A.T + 0.02 * torch.eye(3)
            mean = torch.tensor([-0.4 + 0.4 * j + 0.03 * f, 0.1 * j - 0.1, 2.0 + 0.8 * j])
            fr[oid] = (mean, cov)
        if f == 2:
            fr["9"] = (torch.tensor([0.0, 0.0, 0.3]), 0.05 * torch.eye(3))      # z <= 0.5: skipped by the density, kept in the list
            fr["11"] = (torch.tensor([0.0, 0.0, -1.0]), 0.05 * torch.eye(3))    # behind the camera: dropped
        params.append(fr)
    col_idx = {"1": 0, "2": 3, "7": 12, "9": 5, "11": 6}
    means = torch.stack([params[0][k][0] for k in ("1", "2", "7")])
    covs = torch.stack([params[0][k][1] for k in ("1", "2", "7")])
    E0 = torch.from_numpy(ext[0])
    dens = R.compute_probability_density_map_gpu(means, covs, K, E0[:3, :3], E0[:3, 3:4], (Wi, Hi), device="cpu")
    d1, z1 = R.project_gaussian_to_2d_gpu(means[1], covs[1], K, E0[:3, :3], E0[:3, 3:4], (Wi, Hi), device="cpu")
    out.update({"gauss.K": K, "gauss.ext": torch.from_numpy(np.stack(ext)), "gauss.means0": means, "gauss.covs0": covs,
                "gauss.density_sum": dens, "gauss.density_1": d1, "gauss.z_1": torch.tensor([z1], dtype=torch.float64)})
    for f in range(nF):
        ids = sorted(params[f], key=int)
        out[f"gauss.f{f}.ids"] = torch.tensor([int(i) for i in ids])
        out[f"gauss.f{f}.means"] = torch.stack([params[f][i][0] for i in ids])
        out[f"gauss.f{f}.covs"] = torch.stack([params[f][i][1] for i in ids])
    out["gauss.color_idx"] = torch.tensor([[int(k), v] for k, v in col_idx.items()])
    for thr in (0.05, 0.003):                                      # function default / CLI default (--gaussian_mask_threshold)
        rgbs, alphas = R.project_3d_gaussians_to_2d(params, col_idx, [K.numpy()] * nF, ext, (Wi, Hi), threshold=thr, device="cpu")
        out[f"gauss.rgb_t{thr}"] = torch.stack(rgbs)
        out[f"gauss.alpha_t{thr}"] = torch.stack(alphas)
    bgs = [torch.randint(0, 256, (Hi, Wi, 3), generator=g, dtype=torch.uint8) for _ in range(nF)]
    out["gauss.bg"] = torch.stack(bgs)
    out["gauss.blend"] = torch.stack(R.blend_gaussian_projection_with_bg(rgbs, alphas, bgs))
    out["color.float"] = torch.stack([R.get_object_color(i, {i: i}, "cpu", return_float=True) for i in range(22)])
    out["color.u8"] = torch.stack([R.get_object_color(i, {i: i}, "cpu") for i in range(22)])

    # ---- cameras (:340, :1001) and the ellipsoid parameter file (:1012) ----
    c2w = torch.eye(4).repeat(5, 1, 1)
    for i in range(5):
        a = 0.2 * i
        c2w[i, :3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32) @ \
            torch.tensor([[1, 0, 0], [0, np.cos(0.3), -np.sin(0.3)], [0, np.sin(0.3), np.cos(0.3)]], dtype=torch.float32)
        c2w[i, :3, 3] = torch.tensor([0.1 * i, -0.2, 0.05 * i * i])
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "custom_camera_trajectory.npz")
        np.savez(pth, extrinsics=c2w.numpy().astype(np.float64))
        w2c = R.load_camera_trajectory(pth, device="cpu")
        out["cam.c2w_blender"] = c2w
        out["cam.w2c_opencv"] = w2c
        doc = {"metadata": {"num_frames": 2, "num_objects": 2, "obj_id_to_color_idx": {"3": 1, "5": 4}},
               "frames": [{"frame_index": f, "objects": [
                   {"object_id": oid, "gaussian_3d": {"mean": [0.1 * f + j, 0.2, 1.5 + j],
                                                      "covariance": [[0.3 + j, 0.01, 0], [0.01, 0.2, 0.02 * f], [0, 0.02 * f, 0.1]]}}
                   for j, oid in enumerate([3, 5])]} for f in range(2)]}
        jp = os.path.join(td, "ell.json")
        json.dump(doc, open(jp, "w"))
        gp, cidx, centers = R.load_ellipsoid_parameters(jp, device="cpu")
        out["ell.json"] = torch.tensor(list(json.dumps(doc).encode()), dtype=torch.uint8)
        out["ell.means"] = torch.stack([torch.stack([gp[f][o][0] for o in (3, 5)]) for f in range(2)])
        out["ell.covs"] = torch.stack([torch.stack([gp[f][o][1] for o in (3, 5)]) for f in range(2)])
        assert cidx == {"3": 1, "5": 4} and sorted(centers) == [0, 1]
    Ks = K.repeat(5, 1, 1)
    cams = R._build_cam_from_extrinsics(Ks, w2c, (Hi, Wi))
    out["cam.p3d_R"] = cams.kwargs["R"]
    out["cam.p3d_T"] = cams.kwargs["T"]
    out["cam.p3d_focal"] = cams.kwargs["focal_length"]
    out["cam.p3d_pp"] = cams.kwargs["principal_point"]
    assert cams.kwargs["in_ndc"] is False and cams.kwargs["image_size"] == [(Hi, Wi)] * 5
    out["coord.cv2blender"] = torch.from_numpy(R.COORD_TRANSFORM_CV2BLENDER)

    out = {k: v.contiguous() for k, v in out.items()}
    save_file(out, os.path.join(HERE, "render_small.safetensors"))
    print("wrote", len(out), "tensors,", sum(v.numel() * v.element_size() for v in out.values()), "bytes")


if __name__ == "__main__":
    main()
