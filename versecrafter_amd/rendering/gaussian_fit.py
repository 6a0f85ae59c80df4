"""Per-object 3D Gaussian fit on the HIP engine -- drop-in for the functions of the reference's inference/fit_3D_gaussian.py (same
names, arguments and return values): step 3 of the pre-processing chain, whose `gaussian_params.json` the Blender step and the
control-map renderer build on.

What runs where
  * per-pixel / per-point stages (mask threshold + erosion, unprojection with ordered compaction, the two moment passes, the
    projection's density / Mahalanobis maps, the alpha blend, the uint8 picture): HIP kernels behind the C ABI (csrc/gaussfit.hip,
    include/vcengine.h: vc_op_fit_*).  No CPU fallback: a CPU tensor raises, a missing library raises.
  * per-object scalars (the 2x2 projected covariance of one Gaussian, the 3x3 inverses), file formats (npz, png via PIL, json): host.

Parity: PINNED by the reference's own outputs -- it ships, for two demo clips, the inputs of this step and the files it wrote for
them; tests/test_gpu_fit.py replays both through this module (tests/golden/demo_fit/): mask pixel counts exact, means / covariances
to 2e-6, the projection picture to one uint8 step on < 1e-4 of its values.  cv2 is not in the image: the elliptical erosion is the
restatement that reproduces `num_mask_pixels` exactly."""
import argparse
import ctypes as C
import json
import logging
import math
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .control_maps import TAB20

logger = logging.getLogger(__name__)


# ----------------------------------------------------------------------------------------------------------- plumbing
def _lib_():
    return _lib.load()


def _ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: the Gaussian fit runs on the GPU only (versecrafter_amd has no CPU path)")
    return t.contiguous()


def _check(rc):
    if rc != 0:
        msg = _lib_().vc_fit_last_error()
        msg = msg.decode() if msg else ""
        if rc == _lib.VC_E_INVALID:
            raise ValueError(f"libvcengine: {msg}")
        raise _lib.VcError(rc, msg)


def _host_f32(vals):
    a = np.ascontiguousarray(np.asarray(vals, dtype=np.float32).reshape(-1))
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _np32(t):
    return t.detach().cpu().numpy().astype(np.float32) if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float32)


# ----------------------------------------------------------------------------------------------------------- :35-92
def get_point_cloud_from_depth(depth: torch.Tensor, intrinsic: torch.Tensor, extrinsic: torch.Tensor,
                               mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(H, W) depth + 3x3 intrinsic + 4x4 world-to-camera extrinsic -> (N, 3) world points of the pixels whose mask is set (mask None:
    depth > 0), in row-major pixel order like the reference's boolean indexing."""
    depth = _need_cuda(depth, "get_point_cloud_from_depth").float()
    H, W = depth.shape
    dev = depth.device
    m8 = None
    if mask is not None:
        if tuple(mask.shape) != (H, W):
            raise ValueError(f"get_point_cloud_from_depth: mask {tuple(mask.shape)} does not match depth {(H, W)}")
        m8 = _need_cuda(mask, "get_point_cloud_from_depth").reshape(H, W).ne(0).to(torch.uint8).contiguous()
    kinv = np.linalg.inv(_np32(intrinsic))
    c2w = np.linalg.inv(_np32(extrinsic))[:3]
    _, kp = _host_f32(kinv)
    _, cp = _host_f32(c2w)
    L = _lib_()
    scratch = torch.empty(int(L.vc_op_fit_points_scratch_bytes(W, H)), dtype=torch.uint8, device=dev)
    pts = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    _check(L.vc_op_fit_points(_ptr(depth), _ptr(m8), kp, cp, W, H, _ptr(scratch), _ptr(pts), _ptr(count), _stream(dev)))
    return pts[:int(count.item())]


# ----------------------------------------------------------------------------------------------------------- :95-136
def fit_3d_gaussian(points: torch.Tensor, device: str = "cuda") -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """(N, 3) points -> (mean (3,), covariance (3, 3) = unbiased sample covariance + 1e-6 I), or (None, None) below 3 points."""
    if len(points) == 0:
        logger.warning("fit_3d_gaussian: empty point cloud")
        return None, None
    if len(points) < 3:
        logger.warning(f"fit_3d_gaussian: {len(points)} points are too few")
        return None, None
    points = _need_cuda(points, "fit_3d_gaussian").float()
    dev = points.device
    L = _lib_()
    scratch = torch.empty(int(L.vc_op_fit_moments_scratch_bytes()), dtype=torch.uint8, device=dev)
    out = torch.empty(12, dtype=torch.float32, device=dev)
    _check(L.vc_op_fit_moments(_ptr(points), len(points), _ptr(scratch), _ptr(out), _stream(dev)))
    return out[:3].clone(), out[3:].reshape(3, 3).clone()


# ----------------------------------------------------------------------------------------------------------- :139-159
def load_mask(mask_path: str, device: str = "cuda", erode_kernel_size: int = 5) -> Optional[torch.Tensor]:
    """mask png -> bool (H, W) on the device: grey > 127, eroded once by the elliptical k x k element (boundary noise)."""
    try:
        from PIL import Image
        raw = np.array(Image.open(mask_path), dtype=np.uint8)
        if raw.ndim != 2:
            raise ValueError(f"expected a single-channel mask, got shape {raw.shape}")
        return erode_mask(torch.from_numpy(raw).to(device), erode_kernel_size)
    except Exception as e:  # the reference logs and skips an unreadable mask (:157-159)
        logger.error(f"{mask_path}: {e}")
        return None


def erode_mask(raw_u8: torch.Tensor, erode_kernel_size: int = 5) -> torch.Tensor:
    """The device half of load_mask: uint8 grey (H, W) -> bool (H, W)."""
    raw_u8 = _need_cuda(raw_u8, "erode_mask")
    if raw_u8.dtype != torch.uint8 or raw_u8.dim() != 2:
        raise ValueError("erode_mask: expected a uint8 (H, W) tensor")
    H, W = raw_u8.shape
    out = torch.empty_like(raw_u8)
    _check(_lib_().vc_op_fit_erode_mask(_ptr(raw_u8), _ptr(out), W, H, int(erode_kernel_size), _stream(raw_u8.device)))
    return out.bool()


# ----------------------------------------------------------------------------------------------------------- :162-169
def get_object_color(obj_id: int, obj_id_to_color_idx: Dict[int, int], device: str = "cuda") -> torch.Tensor:
    """tab20 colour of an object by its order of appearance."""
    return torch.tensor(TAB20[obj_id_to_color_idx.get(obj_id, 0) % 20], dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------------------------------------- :171-287
def _projection_record(mean, cov, intrinsic, extrinsic, image_size):
    """The per-Gaussian scalars of project_gaussian_to_2d (:198-257) in float32 on the host -> (record of 11 floats or None, z)."""
    width, height = image_size
    mean, cov, K, E = _np32(mean), _np32(cov), _np32(intrinsic), _np32(extrinsic)
    R, t = E[:3, :3], E[:3, 3]
    mean_cam = R @ mean + t
    z_depth = float(mean_cam[2])
    if z_depth <= 0.2:                                               # near-plane culling
        return None, z_depth
    m2h = K @ mean_cam
    mean_2d = m2h[:2] / m2h[2]
    u, v = float(mean_2d[0]), float(mean_2d[1])
    margin = 50
    if u < -margin or u > width + margin or v < -margin or v > height + margin:
        return None, z_depth
    cov_cam = R @ cov @ R.T
    fx, fy = K[0, 0], K[1, 1]
    x, y, z = mean_cam
    J = np.array([[fx / z, 0, -(fx * x) / (z * z)], [0, fy / z, -(fy * y) / (z * z)]], dtype=np.float32)
    cov_2d = (J @ cov_cam @ J.T + np.float32(1e-4) * np.eye(2, dtype=np.float32)).astype(np.float32)
    det = float(np.linalg.det(cov_2d.astype(np.float64)))
    if not det > 0:
        return None, z_depth
    inv = np.linalg.inv(cov_2d.astype(np.float64)).astype(np.float32)
    radius_int = int(math.ceil(3.0 * math.sqrt(float(max(cov_2d[0, 0], cov_2d[1, 1])))))
    mx, my = int(u), int(v)
    x0, x1 = max(0, mx - radius_int), min(width, mx + radius_int + 1)
    y0, y1 = max(0, my - radius_int), min(height, my + radius_int + 1)
    if x0 >= x1 or y0 >= y1:
        return None, z_depth
    coeff = 1.0 / (2 * math.pi * math.sqrt(det))
    return [u, v, inv[0, 0], inv[0, 1], inv[1, 0], inv[1, 1], coeff, x0, x1, y0, y1], z_depth


def _project(mean, cov, intrinsic, extrinsic, image_size, device):
    width, height = image_size
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("project_gaussian_to_2d: the Gaussian fit runs on the GPU only (versecrafter_amd has no CPU path)")
    rec, z_depth = _projection_record(mean, cov, intrinsic, extrinsic, image_size)
    density = torch.empty(height, width, dtype=torch.float32, device=dev)
    mahal = torch.empty(height, width, dtype=torch.float32, device=dev)
    dmax = torch.zeros(1, dtype=torch.float32, device=dev)
    _, rp = _host_f32(rec if rec is not None else [0] * 11)         # an empty box: density 0, distance inf everywhere
    with torch.cuda.device(dev):
        _check(_lib_().vc_op_fit_project(rp, _ptr(density), _ptr(mahal), _ptr(dmax), width, height, _stream(dev)))
    return density, mahal, z_depth, dmax


def project_gaussian_to_2d(mean: torch.Tensor, cov: torch.Tensor, intrinsic: torch.Tensor, extrinsic: torch.Tensor,
                           image_size: Tuple[int, int], device: str = "cuda") -> Tuple[torch.Tensor, torch.Tensor, float]:
    """One 3D Gaussian -> (density (H, W), squared Mahalanobis distance (H, W) = inf outside its 3-sigma box, z of its centre in the
    camera); a culled Gaussian (behind 0.2, > 50 px off screen, degenerate covariance) gives the empty maps.  image_size = (W, H)."""
    density, mahal, z_depth, _ = _project(mean, cov, intrinsic, extrinsic, image_size, device)
    return density, mahal, z_depth


# ----------------------------------------------------------------------------------------------------------- :290-437
def render_gaussian_projections(gaussian_params: Dict[int, Dict], intrinsic, extrinsic, image_size: Tuple[int, int],
                                probability_threshold: float = 0.97, device: str = "cuda"):
    """The device half of visualize_gaussian_projections (:337-400) -> (picture uint8 (H, W, 3), mask float32 (H, W), {obj_id:
    colour index}): objects in id order, those in front of the camera drawn far to near."""
    width, height = image_size
    dev = torch.device(device)
    threshold = -2.0 * math.log(1.0 - probability_threshold)        # scipy.stats.chi2.ppf(p, df=2) in closed form (:329)
    logger.info(f"Probability threshold: {probability_threshold*100:.1f}% -> Mahalanobis threshold: {threshold:.4f}")
    projections, obj_id_to_color_idx = [], {}
    for obj_id, params in sorted(gaussian_params.items()):
        density, mahal, z_depth, dmax = _project(np.array(params["mean"]), np.array(params["cov"]), intrinsic, extrinsic, image_size, dev)
        if z_depth > 0:
            obj_id_to_color_idx.setdefault(obj_id, len(obj_id_to_color_idx))
            projections.append((z_depth, density, mahal, dmax, TAB20[obj_id_to_color_idx[obj_id] % 20]))
    projections.sort(key=lambda p: p[0], reverse=True)
    picture = torch.zeros(height, width, 3, dtype=torch.float32, device=dev)
    mask = torch.zeros(height, width, dtype=torch.float32, device=dev)
    L = _lib_()
    with torch.cuda.device(dev):
        for _, density, mahal, dmax, color in projections:
            _, cp = _host_f32(color)
            _check(L.vc_op_fit_blend(_ptr(density), _ptr(mahal), _ptr(dmax), threshold, cp, _ptr(picture), _ptr(mask), height * width,
                                     _stream(dev)))
        out = torch.empty(height, width, 3, dtype=torch.uint8, device=dev)
        _check(L.vc_op_fit_picture_u8(_ptr(picture), _ptr(out), picture.numel(), _stream(dev)))
    return out, mask, obj_id_to_color_idx


def visualize_gaussian_projections(gaussian_params: Dict[int, Dict], intrinsic, extrinsic, image_size: Tuple[int, int], output_path: Path,
                                   probability_threshold: float = 0.97, device: str = "cuda",
                                   input_image_path: Optional[str] = None) -> Dict[int, int]:
    """Writes gaussian_projection.png (and gaussian_overlay_on_image.png over the input image) into output_path; returns the colour
    index of every drawn object."""
    from PIL import Image
    width, height = image_size
    picture, mask, obj_id_to_color_idx = render_gaussian_projections(gaussian_params, intrinsic, extrinsic, image_size,
                                                                     probability_threshold, device)
    proj_img = picture.cpu().numpy()
    output_path = Path(output_path)
    Image.fromarray(proj_img, mode="RGB").save(output_path / "gaussian_projection.png")
    logger.info(f"Saved Gaussian projection to {output_path / 'gaussian_projection.png'}")
    if input_image_path is not None:
        try:
            img = Image.open(input_image_path).convert("RGB") if isinstance(input_image_path, (str, Path)) else input_image_path
            if img.size != (width, height):
                logger.warning(f"Input image size {img.size} doesn't match expected {(width, height)}, resizing")
                img = img.resize((width, height), Image.Resampling.LANCZOS)
            base = np.array(img, dtype=np.uint8)
            weight = mask.cpu().numpy().astype(np.float32)[..., None] * 0.7          # blend factor of the overlay (:424-427)
            overlay = (proj_img.astype(np.float32) * weight + base * (1 - weight)).astype(np.uint8)
            Image.fromarray(overlay, mode="RGB").save(output_path / "gaussian_overlay_on_image.png")
            logger.info(f"Saved Gaussian overlay to {output_path / 'gaussian_overlay_on_image.png'}")
        except Exception as e:
            logger.warning(f"overlay picture not written: {e}")
    return obj_id_to_color_idx


def tensor_to_json_serializable(t):
    """Tensors / arrays -> nested lists for json (anything else passes through)."""
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return t.tolist() if isinstance(t, np.ndarray) else t


# ----------------------------------------------------------------------------------------------------------- :450-630
def _read_depth_npz(npz_path, device):
    """depth_intrinsics.npz (MoGe's output) -> (depth float32 (H, W) on the device, intrinsic 3x3 in PIXELS, depth array shape).
    A leading batch axis is dropped; normalised intrinsics (focal < 10) are scaled by the image size (:479-512)."""
    with np.load(npz_path) as z:
        depth_np = np.asarray(z["depth"], dtype=np.float32)
        k_np = np.asarray(z["intrinsic"], dtype=np.float32)
    depth_np = depth_np[0] if depth_np.ndim == 3 else depth_np
    k_np = k_np[0] if k_np.ndim == 3 else k_np
    depth = torch.from_numpy(np.ascontiguousarray(depth_np)).to(device)
    K = torch.from_numpy(np.ascontiguousarray(k_np)).to(device).clone()
    rows, cols = depth.shape
    if min(abs(float(K[0, 0])), abs(float(K[1, 1]))) < 10:
        K[0, 0] *= cols
        K[0, 2] *= cols
        K[1, 1] *= rows
        K[1, 2] *= rows
    return depth, K, depth_np.shape[:2]


def _fit_one_object(mask_file, depth, K, E, device):
    """One mask_NN_<label>.png -> (object id, its gaussian_params entry) or (id, None) when the object is skipped (:530-583)."""
    fields = mask_file.stem.split("_")
    obj_id = int(fields[1])
    label = "_".join(fields[2:]) or f"object_{obj_id}"
    mask = load_mask(str(mask_file), device=device)
    if mask is None:
        logger.warning(f"{mask_file.name}: unreadable mask, object skipped")
        return obj_id, None
    pts = get_point_cloud_from_depth(depth, K, E, mask)
    if len(pts) < 10:
        logger.warning(f"{label} (ID {obj_id}): only {len(pts)} points under the mask, object skipped")
        return obj_id, None
    mean, cov = fit_3d_gaussian(pts, device)
    if mean is None:
        logger.warning(f"{label} (ID {obj_id}): no Gaussian could be fitted, object skipped")
        return obj_id, None
    spectrum = torch.linalg.eigvalsh(cov.double().cpu()).float()                 # 3 x 3, reporting only
    logger.info(f"{label} (ID {obj_id}): {len(pts)} points, mean {mean.cpu().numpy()}, covariance trace {float(cov.trace()):.6f}")
    return obj_id, {
        "label": label,
        "mean": tensor_to_json_serializable(mean),
        "cov": tensor_to_json_serializable(cov),
        "num_points": len(pts),
        "num_mask_pixels": int(mask.sum().item()),
        "eigvals": tensor_to_json_serializable(spectrum),
        "trace": float(cov.trace()),
    }


def process_single_image(npz_path: str, masks_dir: str, output_dir: str, device: str = "cuda", input_image_path: Optional[str] = None,
                         enable_visualization: bool = True):
    """depth_intrinsics.npz + masks/mask_NN_<label>.png -> <output_dir>/gaussian_params.json (+ the two pictures); returns the dict
    that was written (None when the inputs are unusable, like the reference)."""
    out_dir = Path(output_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    try:
        depth, K, depth_shape = _read_depth_npz(npz_path, device)
    except Exception as e:
        logger.error(f"{npz_path}: cannot read depth / intrinsic ({e})")
        return None
    E = torch.eye(4, device=device, dtype=torch.float32)             # the camera of the first frame is the world origin (:494)
    rows, cols = depth.shape
    mask_files = sorted(Path(masks_dir).glob("mask_*.png")) if Path(masks_dir).is_dir() else []
    if not mask_files:
        logger.error(f"{masks_dir}: no mask_*.png files")
        return None

    fitted = {}
    for mf in mask_files:
        try:
            obj_id, entry = _fit_one_object(mf, depth, K, E, device)
        except Exception as e:                                       # the reference logs and goes on with the next mask (:585-587)
            logger.error(f"{mf.name}: {e}")
            continue
        if entry is not None:
            fitted[obj_id] = entry

    colour_of = {obj_id: i for i, obj_id in enumerate(sorted(fitted))}          # without pictures: by id (:606-611)
    if enable_visualization:
        colour_of = {}
        if fitted:
            try:
                colour_of = visualize_gaussian_projections(gaussian_params=fitted, intrinsic=K, extrinsic=E, image_size=(cols, rows),
                                                           output_path=out_dir, probability_threshold=0.97, device=device,
                                                           input_image_path=input_image_path)
            except Exception as e:
                logger.warning(f"pictures not written: {e}")
        else:
            logger.warning("nothing was fitted: no pictures")

    result = {
        "image_info": {"resolution": [int(cols), int(rows)], "depth_shape": list(depth_shape)},
        "camera_info": {"intrinsic": tensor_to_json_serializable(K), "extrinsic": tensor_to_json_serializable(E)},
        "gaussian_params": fitted,
        "num_objects": len(fitted),
        "obj_id_to_color_idx": colour_of,
    }
    with open(out_dir / "gaussian_params.json", "w") as fh:
        json.dump(result, fh, indent=2)
    logger.info(f"{out_dir / 'gaussian_params.json'}: {len(fitted)} objects")
    return result


def parse_args(argv=None):
    """The reference's flags (:633-677), same names and defaults."""
    p = argparse.ArgumentParser(description="Fit 3D Gaussians from single-image NPZ and segmentation masks")
    p.add_argument("--npz_path", type=str, required=True, help="Path to NPZ file (containing depth and intrinsic)")
    p.add_argument("--masks_dir", type=str, required=True, help="Path to segmentation masks directory")
    p.add_argument("--output_dir", type=str, default="./gaussian_results", help="Output directory")
    p.add_argument("--device", type=str, default="cuda", help="Computation device (the HIP engine: cuda only)")
    p.add_argument("--image_path", type=str, default=None, help="Input RGB image path (optional) for overlay visualization")
    p.add_argument("--no_visualization", action="store_true", help="Disable visualization (only save JSON parameters)")
    p.add_argument("--verbose", action="store_true", help="Enable debug-level logging")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    for what, path in (("NPZ file", args.npz_path), ("Masks directory", args.masks_dir), ("Image file", args.image_path)):
        if path is not None and not Path(path).exists():
            logger.error(f"{what} does not exist: {path}")
            return 1
    result = process_single_image(npz_path=args.npz_path, masks_dir=args.masks_dir, output_dir=args.output_dir, device=args.device,
                                  input_image_path=args.image_path, enable_visualization=not args.no_visualization)
    if result is None:
        return 1
    logger.info("Fitting 3D Gaussians complete")
    return 0
