"""4D control-map renderer on the HIP engine -- drop-in for the functions of the reference's
inference/rendering_4D_control_maps.py (same names, arguments and return values), SURVEY 8f row 4, second half.

What runs where
  * per-pixel stages (depth compositing, merged mask, depth visualisation, Gaussian density / projection / blending, the two
    rasterisers): HIP kernels behind the C ABI (csrc/render.hip, include/vcengine.h: vc_op_render_*).  No CPU fallback: a CPU tensor
    raises, a missing library raises.
  * per-video scalars and file formats (depth-range quantiles, the 3x3 camera / covariance arithmetic of a handful of Gaussians, npz /
    json readers, the icosphere): host code, written with the same torch calls as the reference so that its numbers are reproduced.

Parity
  * PINNED by fixtures recorded from the reference's own functions (tests/golden/make_golden_render.py): composite_by_depth[_batch],
    merge_bg_and_fg_mask, visualize_depth_as_grayscale, compute_global_depth_range, compute_probability_density_map_gpu,
    project_gaussian_to_2d_gpu, project_3d_gaussians_to_2d, blend_gaussian_projection_with_bg, get_object_color,
    load_camera_trajectory, load_ellipsoid_parameters, the camera arithmetic of _build_cam_from_extrinsics.
  * UNPINNED (PyTorch3D / cv2 / kornia are neither in the reference tree nor in the image): make_ellipsoid_mesh's icosphere,
    render_meshes_pytorch3d_batch, render_point_cloud_pytorch3d_batch, build_background's image decoding and mask dilation.  They
    restate the published algorithms; the specification they are tested against is oracle/render_oracle.py."""
import ctypes as C
import json
import logging
import math
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib

logger = logging.getLogger(__name__)

# OpenCV (x right, y down, z forward) -> Blender world (x right, y forward, z up)            reference :59-63
COORD_TRANSFORM_CV2BLENDER = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float32)

# matplotlib.colormaps['tab20'] rows 0..19 (RGB): the palette get_object_color reads (:898-900), kept here so that the GPU box needs no
# matplotlib; pinned by the fixture recorded through the reference's own get_object_color
TAB20 = (
    (0.12156862745098039, 0.4666666666666667, 0.7058823529411765), (0.6823529411764706, 0.7803921568627451, 0.9098039215686274),
    (1.0, 0.4980392156862745, 0.054901960784313725), (1.0, 0.7333333333333333, 0.47058823529411764),
    (0.17254901960784313, 0.6274509803921569, 0.17254901960784313), (0.596078431372549, 0.8745098039215686, 0.5411764705882353),
    (0.8392156862745098, 0.15294117647058825, 0.1568627450980392), (1.0, 0.596078431372549, 0.5882352941176471),
    (0.5803921568627451, 0.403921568627451, 0.7411764705882353), (0.7725490196078432, 0.6901960784313725, 0.8352941176470589),
    (0.5490196078431373, 0.33725490196078434, 0.29411764705882354), (0.7686274509803922, 0.611764705882353, 0.5803921568627451),
    (0.8901960784313725, 0.4666666666666667, 0.7607843137254902), (0.9686274509803922, 0.7137254901960784, 0.8235294117647058),
    (0.4980392156862745, 0.4980392156862745, 0.4980392156862745), (0.7803921568627451, 0.7803921568627451, 0.7803921568627451),
    (0.7372549019607844, 0.7411764705882353, 0.13333333333333333), (0.8588235294117647, 0.8588235294117647, 0.5529411764705883),
    (0.09019607843137255, 0.7450980392156863, 0.8117647058823529), (0.6196078431372549, 0.8549019607843137, 0.8980392156862745),
)


# ----------------------------------------------------------------------------------------------------------- plumbing
def _lib_():
    return _lib.load()


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: the renderer runs on the GPU only (versecrafter_amd has no CPU path)")
    return t.contiguous()


def _check(rc):
    if rc != 0:
        msg = _lib_().vc_render_last_error()
        msg = msg.decode() if msg else ""
        if rc == _lib.VC_E_INVALID:
            raise ValueError(f"libvcengine: {msg}")
        raise _lib.VcError(rc, msg)


def _f32x(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


# ----------------------------------------------------------------------------------------------------------- colours, files
def get_object_color(obj_id, obj_id_to_color_idx: Dict, device="cuda", return_float: bool = False) -> torch.Tensor:
    """:885-906."""
    rgb = TAB20[obj_id_to_color_idx.get(obj_id, 0) % 20]
    if return_float:
        return torch.tensor(rgb, dtype=torch.float32, device=device)
    return torch.tensor([c * 255 for c in rgb], dtype=torch.uint8, device=device)


def load_camera_trajectory(trajectory_npz: str, device="cuda") -> torch.Tensor:
    """:1001-1009: `extrinsics` of custom_camera_trajectory.npz (Blender camera-to-world, [F,4,4]) -> OpenCV world-to-camera."""
    data = np.load(trajectory_npz)
    c2w_blender = torch.from_numpy(data["extrinsics"].astype(np.float32)).to(device)
    c2w_blender[:, :3, 1:3] *= -1
    return torch.linalg.inv(c2w_blender)


def load_ellipsoid_parameters(json_path: str, device="cuda"):
    """:1012-1051 -> (per-frame {object id: (mean [3], covariance [3,3])}, obj_id_to_color_idx, per-frame centre points)."""
    with open(json_path, "r") as f:
        data = json.load(f)
    obj_id_to_color_idx = {k: v for k, v in data["metadata"]["obj_id_to_color_idx"].items()}
    params, centers = [], {}
    for frame_data in data["frames"]:
        fi = frame_data["frame_index"]
        centers.setdefault(fi, {})
        fp = {}
        for o in frame_data["objects"]:
            oid = o["object_id"]
            mean = torch.tensor(o["gaussian_3d"]["mean"], dtype=torch.float32, device=device)
            cov = torch.tensor(o["gaussian_3d"]["covariance"], dtype=torch.float32, device=device)
            fp[oid] = (mean, cov)
            if o["gaussian_3d"]["mean"] is not None:
                centers[fi][oid] = torch.tensor(o["gaussian_3d"]["mean"], dtype=torch.float32, device=device)
        params.append(fp)
    return params, obj_id_to_color_idx, centers


def build_pytorch3d_camera_parameters(Ks: torch.Tensor, Ts: torch.Tensor):
    """The arithmetic of _build_cam_from_extrinsics (:351-385) in front of PerspectiveCameras: (R, T, focal_length,
    principal_point).  The engine's rasterisers take the OpenCV matrices directly (the change of convention cancels in the image);
    this is kept for callers that want PyTorch3D's parameters."""
    orig = Ts.dtype
    if orig == torch.float16:
        Ks, Ts = Ks.float(), Ts.float()
    c2ws = torch.linalg.inv(Ts)
    c2ws[:, :3, :2] *= -1
    w2cs = torch.linalg.inv(c2ws)
    focal = torch.stack([Ks[:, 0, 0], Ks[:, 1, 1]], dim=1)
    pp = torch.stack([Ks[:, 0, 2], Ks[:, 1, 2]], dim=1)
    R, T = w2cs[:, :3, :3].permute(0, 2, 1), w2cs[:, :3, 3]
    if orig == torch.float16:
        focal, pp, R, T = focal.to(orig), pp.to(orig), R.to(orig), T.to(orig)
    return R, T, focal, pp


# ----------------------------------------------------------------------------------------------------------- depth compositing
def composite_by_depth_batch(bg_rgb, bg_depth, fg_rgb, fg_depth, fg_mask) -> Tuple[torch.Tensor, torch.Tensor]:
    """:398-411.  rgb uint8 [..., H, W, 3], depth float32 [..., H, W], mask bool [..., H, W]."""
    bg_rgb, fg_rgb = _need_cuda(bg_rgb, "bg_rgb"), _need_cuda(fg_rgb, "fg_rgb")
    bg_depth, fg_depth = _need_cuda(bg_depth, "bg_depth").float(), _need_cuda(fg_depth, "fg_depth").float()
    m = _need_cuda(fg_mask, "fg_mask").to(torch.uint8)
    if bg_rgb.dtype != torch.uint8 or fg_rgb.dtype != torch.uint8 or bg_rgb.shape != fg_rgb.shape or bg_rgb.shape[:-1] != bg_depth.shape:
        raise ValueError("composite_by_depth: rgb must be uint8 [..., H, W, 3] matching depth [..., H, W]")
    out_rgb, out_depth = torch.empty_like(bg_rgb), torch.empty_like(bg_depth)
    _check(_lib_().vc_op_render_composite(_ptr(bg_rgb), _ptr(bg_depth), _ptr(fg_rgb), _ptr(fg_depth), _ptr(m), None, _ptr(out_rgb),
                                          _ptr(out_depth), None, bg_depth.numel(), _stream(bg_rgb.device)))
    return out_rgb, out_depth


def composite_by_depth(bg_rgb, bg_depth, fg_rgb, fg_depth, fg_mask):
    """:437-453 (one frame)."""
    assert bg_rgb.shape[:2] == bg_depth.shape and fg_rgb.shape[:2] == fg_depth.shape
    return composite_by_depth_batch(bg_rgb, bg_depth, fg_rgb, fg_depth, fg_mask)


def merge_bg_and_fg_sequences(bg_rgb_frames, bg_depth_frames, bg_masks, fg_rgb_frames, fg_depth_frames, fg_masks):
    """:414-434."""
    assert len(bg_rgb_frames) == len(fg_rgb_frames), "Background and foreground frame counts must match"
    if len(bg_rgb_frames) == 0:
        return [], [], bg_masks, fg_masks
    rgb, depth = composite_by_depth_batch(torch.stack(list(bg_rgb_frames)), torch.stack(list(bg_depth_frames)), torch.stack(list(fg_rgb_frames)),
                                          torch.stack(list(fg_depth_frames)), torch.stack(list(fg_masks)))
    return list(rgb), list(depth), bg_masks, fg_masks


def merge_bg_and_fg_mask(background_depth_frames, foreground_depth_frames, background_masks, foreground_masks, device="cuda"):
    """:736-763 -> list of uint8 [H, W, 3] (255 where the background is NOT rendered or the foreground is in front)."""
    if len(background_depth_frames) == 0:
        return []
    bd = _need_cuda(torch.stack(list(background_depth_frames)), "bg_depth").float()
    fd = _need_cuda(torch.stack(list(foreground_depth_frames)), "fg_depth").float()
    bm = _need_cuda(torch.stack(list(background_masks)), "bg_mask").to(torch.uint8)
    fm = _need_cuda(torch.stack(list(foreground_masks)), "fg_mask").to(torch.uint8)
    out = torch.empty(tuple(bd.shape) + (3,), dtype=torch.uint8, device=bd.device)
    _check(_lib_().vc_op_render_composite(None, _ptr(bd), None, _ptr(fd), _ptr(fm), _ptr(bm), None, None, _ptr(out), bd.numel(),
                                          _stream(bd.device)))
    return list(out)


# ----------------------------------------------------------------------------------------------------------- depth visualisation
def _depth_percentiles(valid: List[torch.Tensor]):
    """The reference's range estimate (:496-515, :547-569): 0.1 % / 99 % quantiles of the valid depths (a random million of them when
    there are more -- unseeded in the reference too), min / max when quantile raises."""
    if not valid:
        return None
    d = torch.cat(valid)
    if len(d) > 1000000:
        d = d[torch.randperm(len(d), device=d.device)[:1000000]]
    try:
        return torch.quantile(d, 0.001), torch.quantile(d, 0.99)
    except RuntimeError:
        return torch.min(d), torch.max(d)


def compute_global_depth_range(depth_frames_list: List[List[torch.Tensor]]) -> Tuple[float, float]:
    """:541-571."""
    valid = [d[d > 0].flatten() for frames in depth_frames_list for d in frames if torch.any(d > 0)]
    r = _depth_percentiles(valid)
    return (0.0, 1.0) if r is None else (r[0].item(), r[1].item())


def visualize_depth_as_grayscale(depth_frames: List[torch.Tensor], global_min_depth: Optional[float] = None,
                                 global_max_depth: Optional[float] = None) -> List[torch.Tensor]:
    """:487-539 -> list of uint8 [H, W, 3]: disparity normalised to the range, closer = lighter."""
    if len(depth_frames) == 0:
        return []
    if global_min_depth is None or global_max_depth is None:
        r = _depth_percentiles([d[d > 0].flatten() for d in depth_frames if torch.any(d > 0)])
        min_depth, max_depth = (0.0, 1.0) if r is None else r
    else:
        min_depth, max_depth = global_min_depth, global_max_depth
    d = _need_cuda(torch.stack(list(depth_frames)), "depth").float()
    normalize = bool(max_depth > 0 and min_depth > 0)
    min_disp = denom = 0.0
    if normalize:                       # the reference's scalar arithmetic, in the types it happens in (python floats or 0-dim tensors)
        lo, hi = 1.0 / max_depth, 1.0 / min_depth
        min_disp, denom = float(lo), float(hi - lo + 1e-8)
    out = torch.empty(tuple(d.shape) + (3,), dtype=torch.uint8, device=d.device)
    _check(_lib_().vc_op_render_depth_gray(_ptr(d), _ptr(out), d.numel(), int(normalize), C.c_float(min_disp), C.c_float(denom),
                                           _stream(d.device)))
    return list(out)


# ----------------------------------------------------------------------------------------------------------- projected Gaussians
def _gaussian_record(mean, cov, K, R, t):
    """:828-873 for one Gaussian on the host (3x3 float32 torch arithmetic, the reference's statements): 12 floats
    {mean_u, mean_v, inv00, inv01, inv10, inv11, coeff, valid, 0, 0, 0, 0} and the camera-space depth of the mean."""
    mean, cov, K, R, t = (torch.as_tensor(x).detach().float().cpu() for x in (mean, cov, K, R, t))
    t_vec = t.squeeze() if t.dim() == 2 else t
    m = R @ mean + t_vec
    c = R @ cov @ R.T
    rec = [0.0] * 12
    z = m[2]
    if z <= 0.5:
        return rec, float(z)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y = m[0], m[1]
    J = torch.tensor([[fx / z, 0, -fx * x / (z * z)], [0, fy / z, -fy * y / (z * z)]], dtype=torch.float32)
    mean_2d = torch.tensor([fx * x / z + cx, fy * y / z + cy], dtype=torch.float32)
    cov_2d = J @ c @ J.T
    cov_2d += torch.eye(2) * 1e-6
    if torch.det(cov_2d) > 1e11:
        return rec, float(z)
    try:
        inv = torch.linalg.inv(cov_2d)
    except Exception as e:                         # the reference logs and skips (:879-881)
        logger.warning(f"Skipping Gaussian due to error: {e}")
        return rec, float(z)
    coeff = 1.0 / (2 * torch.pi * torch.sqrt(torch.det(cov_2d)))
    rec[:8] = [float(mean_2d[0]), float(mean_2d[1]), float(inv[0, 0]), float(inv[0, 1]), float(inv[1, 0]), float(inv[1, 1]), float(coeff), 1.0]
    return rec, float(z)


def compute_probability_density_map_gpu(means_3d, covs_3d, K, R, t, image_size: Tuple[int, int], device="cuda") -> torch.Tensor:
    """:801-883: image_size = (width, height); sum of the projected Gaussians' pdf at pixel (u, v) = (column, row)."""
    width, height = image_size
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("compute_probability_density_map_gpu: the renderer runs on the GPU only")
    recs = [_gaussian_record(means_3d[i], covs_3d[i], K, R, t)[0] for i in range(len(means_3d))]
    rt = torch.tensor(recs, dtype=torch.float32, device=dev).reshape(-1, 12)
    out = torch.empty(height, width, dtype=torch.float32, device=dev)
    _check(_lib_().vc_op_render_gauss_density(_ptr(rt) if len(recs) else None, len(recs), _ptr(out), width, height, _stream(dev)))
    return out


def project_gaussian_to_2d_gpu(mean, cov, K, R, t, image_size: Tuple[int, int], device="cuda") -> Tuple[torch.Tensor, float]:
    """:765-799 -> (density map with non-finite values zeroed, camera-space depth of the mean)."""
    density = compute_probability_density_map_gpu([mean], [cov], K, R, t, image_size, device)      # the kernel zeroes nan / inf
    return density, _gaussian_record(mean, cov, K, R, t)[1]


def project_3d_gaussians_to_2d(gaussian_params_per_frame, obj_id_to_color_idx, intrinsics, extrinsics, image_size: Tuple[int, int],
                               threshold: float = 0.05, device="cuda") -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """:573-695: per frame, every Gaussian in front of the camera is projected, its density map divided by its maximum, turned
    into an opacity above `threshold` and composited far to near in its object colour -> (rgb uint8 [H,W,3], alpha float32 [H,W])."""
    width, height = image_size
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("project_3d_gaussians_to_2d: the renderer runs on the GPU only")
    lib = _lib_()
    rgb_frames, alpha_frames = [], []
    span = float(1.0 - threshold + 1e-8)
    for fi, params in enumerate(gaussian_params_per_frame):
        if fi >= len(intrinsics) or fi >= len(extrinsics):
            break
        K = torch.as_tensor(intrinsics[fi]).float().cpu()
        E = torch.as_tensor(extrinsics[fi]).float().cpu()
        R, t = E[:3, :3], E[:3, 3:4]
        lst = []
        for oid, (mean, cov) in params.items():
            rec, z = _gaussian_record(mean, cov, K, R, t)
            if z > 0:
                col = TAB20[obj_id_to_color_idx.get(oid, 0) % 20]
                rec[8:11] = [np.float32(c) for c in col]
                lst.append((z, rec))
        lst.sort(key=lambda e: e[0], reverse=True)                 # far to near; stable, as list.sort in the reference
        n = len(lst)
        rt = torch.tensor([r for _, r in lst], dtype=torch.float32, device=dev).reshape(-1, 12)
        scratch = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        rgb = torch.empty(height, width, 3, dtype=torch.uint8, device=dev)
        alpha = torch.empty(height, width, dtype=torch.float32, device=dev)
        _check(lib.vc_op_render_gauss_frame(_ptr(rt) if n else None, n, _ptr(scratch), C.c_float(threshold), C.c_float(span), _ptr(rgb),
                                            _ptr(alpha), width, height, _stream(dev)))
        rgb_frames.append(rgb)
        alpha_frames.append(alpha)
    return rgb_frames, alpha_frames


def blend_gaussian_projection_with_bg(gaussian_rgb_frames, gaussian_alpha_frames, background_frames) -> List[torch.Tensor]:
    """:697-734: C_out = C_fg alpha + C_bg (1 - alpha) per frame -> uint8."""
    assert len(gaussian_rgb_frames) == len(gaussian_alpha_frames) == len(background_frames), "All input lists must have the same length"
    if len(gaussian_rgb_frames) == 0:
        return []
    g = _need_cuda(torch.stack(list(gaussian_rgb_frames)), "gaussian_rgb")
    a = _need_cuda(torch.stack(list(gaussian_alpha_frames)), "alpha").float()
    b = torch.stack([x.to(g.device) for x in background_frames]).contiguous()
    out = torch.empty_like(g)
    _check(_lib_().vc_op_render_blend(_ptr(g), _ptr(a), _ptr(b), _ptr(out), a.numel(), 0, _stream(g.device)))
    return list(out)


def mask_gaussian_projection(gaussian_rgb_frames, gaussian_alpha_frames) -> List[torch.Tensor]:
    """main() :1321-1326: (rgb / 255 * alpha * 255) -> uint8, the frames of 3D_gaussian_RGB.mp4."""
    if len(gaussian_rgb_frames) == 0:
        return []
    g = _need_cuda(torch.stack(list(gaussian_rgb_frames)), "gaussian_rgb")
    a = _need_cuda(torch.stack(list(gaussian_alpha_frames)), "alpha").float()
    out = torch.empty_like(g)
    _check(_lib_().vc_op_render_blend(_ptr(g), _ptr(a), None, _ptr(out), a.numel(), 1, _stream(g.device)))
    return list(out)


# ----------------------------------------------------------------------------------------------------------- meshes and point clouds
def ico_sphere(level: int, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """pytorch3d.utils.ico_sphere restated (UNPINNED): icosahedron, `level` subdivisions, new vertices on the unit sphere."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    faces = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
             [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    verts = [tuple(x) for x in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.array(verts[a]) + np.array(verts[b])) / 2.0
                verts.append(tuple(m / np.linalg.norm(m)))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        faces = nf
    return torch.tensor(np.array(verts), dtype=torch.float32, device=device), torch.tensor(faces, dtype=torch.int32, device=device)


class Mesh:
    """Vertices [V,3] float32 (world), faces [F,3] int32, per-vertex colours [V,3] float32 in [0,1] -- what the reference keeps in a
    pytorch3d Meshes + TexturesVertex."""

    def __init__(self, verts, faces, colors):
        self.verts, self.faces, self.colors = verts, faces, colors

    def num_verts(self):
        return int(self.verts.shape[0])


_ICO_CACHE = {}


def make_ellipsoid_mesh(mean, cov, scale_factor: float = 2.0, subdivisions: int = 3, color_rgb255=None, device="cuda") -> Mesh:
    """:66-112: the icosphere mapped by x = mean + U diag(scale sqrt(max(eval, 1e-8))) u, one colour per mesh."""
    device = mean.device if isinstance(mean, torch.Tensor) else torch.device(device)
    if subdivisions not in _ICO_CACHE:
        _ICO_CACHE[subdivisions] = ico_sphere(subdivisions)
    verts, faces = (x.to(device) for x in _ICO_CACHE[subdivisions])
    mean_t = torch.as_tensor(mean).to(device).float()
    cov_t = torch.as_tensor(cov).to(device).float()
    evals, evecs = torch.linalg.eigh(cov_t)
    axes = scale_factor * torch.sqrt(torch.clamp(evals, min=1e-8))
    M = evecs @ torch.diag(axes)
    verts_world = verts @ M.T + mean_t
    if color_rgb255 is None:
        color_rgb255 = torch.tensor([200, 60, 60], dtype=torch.uint8, device=device)
    colors = (color_rgb255.to(device).float() / 255.0).expand_as(verts_world).contiguous()
    return Mesh(verts_world.contiguous(), faces.contiguous(), colors)


def combine_meshes_for_scene(mesh_list: List[Mesh]) -> Optional[Mesh]:
    """:114-148: one mesh with the face indices offset."""
    if len(mesh_list) == 0:
        return None
    ofs, vs, fs, cs = 0, [], [], []
    for m in mesh_list:
        vs.append(m.verts)
        fs.append(m.faces + ofs)
        cs.append(m.colors)
        ofs += m.verts.shape[0]
    return Mesh(torch.cat(vs).contiguous(), torch.cat(fs).contiguous(), torch.cat(cs).contiguous())


def _host_mats(Ks, Ts):
    return Ks.detach().float().cpu().contiguous().numpy(), Ts.detach().float().cpu().contiguous().numpy()


def render_meshes_pytorch3d_batch(meshes_list: List[Optional[Mesh]], Ks, Ts, image_size: Tuple[int, int],
                                  background_color=(0.0, 0.0, 0.0), use_fp16: bool = False):
    """:150-241 (UNPINNED): per frame the nearest face per pixel (perspective-correct depth), flat Phong shading with a point light at
    the world origin -> (rgb uint8 [B,H,W,3], depth float32 [B,H,W] (0 = nothing), mask bool [B,H,W])."""
    H, W = image_size
    dev = Ks.device
    if dev.type != "cuda":
        raise RuntimeError("render_meshes: the renderer runs on the GPU only")
    B = len(meshes_list)
    bg = int(background_color[0] * 255)
    rgb = torch.full((B, H, W, 3), bg, dtype=torch.uint8, device=dev)
    depth = torch.zeros((B, H, W), dtype=torch.float32, device=dev)
    mask = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    Kh, Th = _host_mats(Ks, Ts)
    lib = _lib_()
    for i, m in enumerate(meshes_list):
        if m is None or m.num_verts() == 0:
            continue
        nv, nf = int(m.verts.shape[0]), int(m.faces.shape[0])
        verts, vcol = _need_cuda(m.verts, "mesh vertices").float(), _need_cuda(m.colors, "mesh colours").float()
        faces = _need_cuda(m.faces, "mesh faces").to(torch.int32)
        if verts.shape != (nv, 3) or vcol.shape != (nv, 3) or faces.shape != (nf, 3):
            raise ValueError("render_meshes: verts / colours must be [V, 3], faces [F, 3]")
        if nf and (int(faces.min()) < 0 or int(faces.max()) >= nv):
            raise ValueError("render_meshes: face index out of range")
        scratch = torch.empty(int(lib.vc_op_render_mesh_scratch_bytes(nv, W, H)), dtype=torch.uint8, device=dev)
        w2c = Th[i].astype(np.float32)
        eye = -(w2c[:3, :3].T @ w2c[:3, 3])
        _check(lib.vc_op_render_mesh(_ptr(verts), _ptr(vcol), nv, _ptr(faces), nf, _f32x(w2c.reshape(-1)), _f32x(Kh[i].reshape(-1)),
                                     _f32x([0.0, 0.0, 0.0]), _f32x(eye), W, H, bg, _ptr(scratch), _ptr(rgb[i]), _ptr(depth[i]), _ptr(mask[i]),
                                     _stream(dev)))
    return rgb, depth, mask.bool()


def render_point_cloud_pytorch3d_batch(points_3d, colors, Ks, Ts, image_size: Tuple[int, int], point_size: float = 0.01,
                                       background_color=(0.5, 0.5, 0.5), use_fp16: bool = False):
    """:243-338 (UNPINNED): every camera renders the same cloud; per pixel the 8 nearest points within `point_size` (NDC radius) of the
    pixel centre, alpha-composited front to back with weight 1 - d^2 / r^2; depth and mask from the nearest one."""
    H, W = image_size
    dev = points_3d.device
    if dev.type != "cuda":
        raise RuntimeError("render_point_cloud: the renderer runs on the GPU only")
    B = Ks.shape[0]
    rgb = torch.full((B, H, W, 3), int(background_color[0] * 255), dtype=torch.uint8, device=dev)
    depth = torch.zeros((B, H, W), dtype=torch.float32, device=dev)
    mask = torch.zeros((B, H, W), dtype=torch.uint8, device=dev)
    if len(points_3d) == 0:
        return rgb, depth, mask.bool()
    ok = torch.isfinite(points_3d).all(dim=1) & torch.isfinite(colors.float()).all(dim=1)
    if not bool(ok.all()):
        logger.warning(f"Filtering {int((~ok).sum())} invalid points before rendering")
        points_3d, colors = points_3d[ok], colors[ok]
        if len(points_3d) == 0:
            return rgb, depth, mask.bool()
    pts = _need_cuda(points_3d, "points").float()
    col = _need_cuda(colors, "colors").to(torch.uint8)
    if pts.dim() != 2 or pts.shape[1] != 3 or col.shape != pts.shape:
        raise ValueError("render_point_cloud: points and colours must both be [N, 3]")
    n = int(pts.shape[0])
    lib = _lib_()
    scratch = torch.empty(int(lib.vc_op_render_points_scratch_bytes(n, W, H, 8)), dtype=torch.uint8, device=dev)
    Kh, Th = _host_mats(Ks, Ts)
    for i in range(B):
        _check(lib.vc_op_render_points(_ptr(pts), _ptr(col), n, _f32x(Th[i].reshape(-1)), _f32x(Kh[i].reshape(-1)), W, H, C.c_float(point_size), 8,
                                       C.c_float(background_color[0]), _ptr(scratch), _ptr(rgb[i]), _ptr(depth[i]), _ptr(mask[i]), _stream(dev)))
    return rgb, depth, mask.bool()


def render_video_with_bg_and_fg(background_points, background_colors, foreground_meshes_per_frame, intrinsics, extrinsics,
                                image_size: Tuple[int, int], mode: str = "full", point_size: float = 0.01, device="cuda", batch_size: int = 1,
                                use_fp16: bool = False, pin_memory: bool = False):
    """:1054-1142 -> per-frame lists (rgb, depth, background mask, foreground mask)."""
    num_frames = len(foreground_meshes_per_frame)
    H, W = image_size
    dev = torch.device(device)
    Ks, Ts = intrinsics[:num_frames], extrinsics[:num_frames]
    if mode in ("full", "background") and background_points is not None:
        rgb_bg, depth_bg, mask_bg = render_point_cloud_pytorch3d_batch(background_points, background_colors, Ks, Ts, image_size, point_size)
    else:
        rgb_bg = torch.zeros((num_frames, H, W, 3), dtype=torch.uint8, device=dev)
        depth_bg = torch.zeros((num_frames, H, W), dtype=torch.float32, device=dev)
        mask_bg = torch.zeros((num_frames, H, W), dtype=torch.bool, device=dev)
    if mode in ("full", "foreground"):
        rgb_fg, depth_fg, mask_fg = render_meshes_pytorch3d_batch(list(foreground_meshes_per_frame), Ks, Ts, image_size)
    else:
        rgb_fg = torch.zeros((num_frames, H, W, 3), dtype=torch.uint8, device=dev)
        depth_fg = torch.zeros((num_frames, H, W), dtype=torch.float32, device=dev)
        mask_fg = torch.zeros((num_frames, H, W), dtype=torch.bool, device=dev)
    if mode == "foreground":
        rgb_out, depth_out = rgb_fg, depth_fg
    elif mode == "background":
        rgb_out, depth_out = rgb_bg, depth_bg
    else:
        rgb_out, depth_out = composite_by_depth_batch(rgb_bg, depth_bg, rgb_fg, depth_fg, mask_fg)
    return list(rgb_out), list(depth_out), list(mask_bg), list(mask_fg)


# ----------------------------------------------------------------------------------------------------------- background cloud
def depth_to_points(depth: torch.Tensor, intrinsic: torch.Tensor) -> torch.Tensor:
    """kornia.geometry.depth.depth_to_3d_v2(depth, K, normalize_points=False) restated (UNPINNED): pixel (u, v) = (column, row) at
    depth z unprojects to ((u - cx) z / fx, (v - cy) z / fy, z) -> [H, W, 3]."""
    H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, device=depth.device, dtype=depth.dtype), torch.arange(W, device=depth.device, dtype=depth.dtype),
                          indexing="ij")
    fx, fy, cx, cy = intrinsic[0, 0], intrinsic[1, 1], intrinsic[0, 2], intrinsic[1, 2]
    return torch.stack([(u - cx) * depth / fx, (v - cy) * depth / fy, depth], dim=-1)


def _dilate_ellipse(mask: torch.Tensor, k: int = 10) -> torch.Tensor:
    """cv2.dilate with cv2.getStructuringElement(MORPH_ELLIPSE, (k, k)) restated (UNPINNED): OpenCV's elliptical element of an even
    size k has its anchor at (k // 2, k // 2); row i spans the columns |j - c| <= round(c sqrt(1 - ((i - c) / c)^2)), c = k // 2
    (the first and last rows of the k x k box may be empty)."""
    c = k // 2
    H, W = mask.shape
    out = torch.zeros_like(mask)
    padded = torch.nn.functional.pad(mask[None, None].float(), (c, c, c, c))[0, 0] > 0
    for i in range(k):
        dy = i - c
        if abs(dy) > c:
            continue
        dx = int(round(c * math.sqrt(max(0.0, 1.0 - (dy / c) ** 2)))) if c else 0
        for j in range(max(0, c - dx), min(k, c + dx + 1)):
            # output(y, x) |= input(y + i - c, x + j - c)
            out |= padded[i:i + H, j:j + W]
    return out


def build_background(png_path: str, npz_path: str, mask_dir: str, device="cuda"):
    """:908-998 -> (points [N,3] in the Blender world, colours uint8 [N,3], intrinsic [3,3] in pixels, identity extrinsic, H, W).
    Image files are decoded with PIL (no cv2 in the image); the masks' nearest-neighbour resize and the elliptical dilation restate
    OpenCV's (UNPINNED)."""
    from PIL import Image
    image = np.asarray(Image.open(png_path).convert("RGB"))
    H, W = image.shape[:2]
    image_tensor = torch.from_numpy(image.copy()).to(device)
    data = np.load(npz_path)
    depth = torch.from_numpy(data["depth"].astype(np.float32)).to(device)
    intrinsic = torch.from_numpy(data["intrinsic"].astype(np.float32)).to(device).clone()
    intrinsic[0, 0] *= W
    intrinsic[1, 1] *= H
    intrinsic[0, 2] *= W
    intrinsic[1, 2] *= H
    extrinsic = torch.eye(4, dtype=torch.float32, device=device)
    combined = torch.zeros((H, W), dtype=torch.bool, device=device)
    for mf in sorted(Path(mask_dir).glob("*.png")) if mask_dir else []:
        m = Image.open(mf).convert("L").resize((W, H), Image.NEAREST)
        combined |= torch.from_numpy(np.asarray(m) > 127).to(device)
    combined = _dilate_ellipse(combined, 10)
    pts_cam = depth_to_points(depth, intrinsic).reshape(-1, 3)
    c2w = torch.linalg.inv(extrinsic)
    pts_h = torch.cat([pts_cam, torch.ones(len(pts_cam), 1, device=pts_cam.device)], dim=1)
    pts_world_cv = (c2w @ pts_h.T).T[:, :3]
    pts_world = (torch.from_numpy(COORD_TRANSFORM_CV2BLENDER).to(device) @ pts_world_cv.T).T
    keep = ~combined
    bg_points, bg_colors = pts_world[keep.reshape(-1)], image_tensor[keep]
    valid = torch.isfinite(bg_points).all(dim=1) & (bg_points.abs() < 1e6).all(dim=1)
    if int(valid.sum()) < len(bg_points):
        logger.warning(f"Filtered out {len(bg_points) - int(valid.sum())} invalid points from background point cloud")
        bg_points, bg_colors = bg_points[valid], bg_colors[valid]
    return bg_points, bg_colors, intrinsic, extrinsic, H, W
