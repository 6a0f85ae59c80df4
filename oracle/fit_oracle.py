"""TEST INFRASTRUCTURE - CPU restatement (numpy, float32 like the reference's tensors) of step 3 of the reference's pre-processing
chain, `inference/fit_3D_gaussian.py`: masked depth -> world points -> one 3D Gaussian per object -> its 2D projection picture.
Only tests/ may import this file; the product (versecrafter_amd/rendering/gaussian_fit.py) runs HIP kernels and never falls back here.

PINNED by the reference's own outputs: `demo_data/<clip>/fitted_3D_gaussian/{gaussian_params.json, gaussian_projection.png}` are what
the reference wrote for `demo_data/<clip>/{estimated_depth/depth_intrinsics.npz, object_mask/masks/mask_*.png}`; the test
tests/test_fit_oracle.py replays both clips (copies of those data files: tests/golden/demo_fit/).  `num_mask_pixels` in the json pins
the mask threshold + cv2 erosion restated here (cv2 itself is not in the image)."""
import math

import numpy as np

F32 = np.float32


def ellipse_element(k: int) -> np.ndarray:
    """cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)) (fit_3D_gaussian.py:152): row i covers the columns
    |j - c| <= round(c sqrt((r^2 - dy^2) / r^2)), r = c = k // 2, dy = i - r (OpenCV's published construction)."""
    r = c = k // 2
    el = np.zeros((k, k), bool)
    for i in range(k):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * math.sqrt((r * r - dy * dy) / float(r * r)))) if r else 0
            el[i, max(c - dx, 0):min(c + dx + 1, k)] = True
    return el


def load_mask(mask_u8: np.ndarray, erode_kernel_size: int = 5) -> np.ndarray:
    """fit_3D_gaussian.py:139-159 on the decoded grey image: (m > 127), eroded once by the k x k ellipse, anchor at the centre; pixels
    outside the image never remove anything (cv2.erode's default border)."""
    m = mask_u8 > 127
    k = erode_kernel_size
    a = k // 2
    H, W = m.shape
    padded = np.ones((H + k, W + k), bool)
    padded[a:a + H, a:a + W] = m
    out = np.ones((H, W), bool)
    for i, j in zip(*np.nonzero(ellipse_element(k))):
        out &= padded[i:i + H, j:j + W]
    return out


def get_point_cloud_from_depth(depth, intrinsic, extrinsic, mask=None) -> np.ndarray:
    """:35-92: K^-1 [x y 1] depth, then camera-to-world; rows kept where the mask is set (or depth > 0), in row-major pixel order."""
    h, w = depth.shape
    y, x = np.meshgrid(np.arange(h, dtype=F32), np.arange(w, dtype=F32), indexing="ij")
    xy1 = np.stack([x, y, np.ones_like(x)], 0).reshape(3, -1)
    cam = (np.linalg.inv(intrinsic.astype(F32)) @ xy1) * depth.reshape(-1).astype(F32)
    cam = np.concatenate([cam, np.ones((1, cam.shape[1]), F32)], 0)
    world = (np.linalg.inv(extrinsic.astype(F32)) @ cam)[:3].T
    keep = mask.reshape(-1).astype(bool) if mask is not None else depth.reshape(-1) > 0
    return world[keep].astype(F32)


def fit_3d_gaussian(points):
    """:95-136: sample mean, unbiased covariance + 1e-6 I; None below 3 points."""
    if len(points) < 3:
        return None, None
    mean = points.mean(0, dtype=np.float64)
    c = points.astype(np.float64) - mean
    cov = c.T @ c / (len(points) - 1) + 1e-6 * np.eye(3)
    return mean.astype(F32), cov.astype(F32)


def projection_record(mean, cov, intrinsic, extrinsic, image_size):
    """The scalar part of :171-267: None when the Gaussian is culled, else (mean_2d, inv_cov_2d, coeff, roi = (min_x, max_x, min_y,
    max_y), z_depth)."""
    width, height = image_size
    R, t = extrinsic[:3, :3].astype(F32), extrinsic[:3, 3].astype(F32)
    mean_cam = R @ mean.astype(F32) + t
    z = float(mean_cam[2])
    if z <= 0.2:
        return None, z
    m2h = intrinsic.astype(F32) @ mean_cam
    mean_2d = (m2h[:2] / m2h[2]).astype(F32)
    u, v = float(mean_2d[0]), float(mean_2d[1])
    margin = 50
    if u < -margin or u > width + margin or v < -margin or v > height + margin:
        return None, z
    cov_cam = R @ cov.astype(F32) @ R.T
    fx, fy = intrinsic[0, 0], intrinsic[1, 1]
    x, y, zz = mean_cam
    J = np.array([[fx / zz, 0, -(fx * x) / (zz * zz)], [0, fy / zz, -(fy * y) / (zz * zz)]], F32)
    cov_2d = (J @ cov_cam @ J.T + F32(1e-4) * np.eye(2, dtype=F32)).astype(F32)
    det = float(np.linalg.det(cov_2d.astype(np.float64)))
    if det <= 0:
        return None, z
    inv = np.linalg.inv(cov_2d.astype(np.float64)).astype(F32)
    radius_int = int(math.ceil(3.0 * math.sqrt(max(cov_2d[0, 0], cov_2d[1, 1]))))
    mx, my = int(u), int(v)
    roi = (max(0, mx - radius_int), min(width, mx + radius_int + 1), max(0, my - radius_int), min(height, my + radius_int + 1))
    if roi[0] >= roi[1] or roi[2] >= roi[3]:
        return None, z
    return (mean_2d, inv, F32(1.0 / (2 * math.pi * math.sqrt(det))), roi), z


def project_gaussian_to_2d(mean, cov, intrinsic, extrinsic, image_size):
    """:171-287 -> (density [H,W], squared Mahalanobis distance [H,W] (inf outside the 3-sigma box), z of the centre)."""
    width, height = image_size
    density = np.zeros((height, width), F32)
    mahal = np.full((height, width), np.inf, F32)
    rec, z = projection_record(mean, cov, intrinsic, extrinsic, image_size)
    if rec is None:
        return density, mahal, z
    mean_2d, inv, coeff, (x0, x1, y0, y1) = rec
    gx, gy = np.meshgrid(np.arange(x0, x1, dtype=F32), np.arange(y0, y1, dtype=F32), indexing="xy")
    dx, dy = gx - mean_2d[0], gy - mean_2d[1]
    m = (dx * dx * inv[0, 0] + dx * dy * (inv[0, 1] + inv[1, 0]) + dy * dy * inv[1, 1]).astype(F32)
    density[y0:y1, x0:x1] = coeff * np.exp(F32(-0.5) * m)
    mahal[y0:y1, x0:x1] = m
    return density, mahal, z


def mahalanobis_threshold(probability: float) -> float:
    """scipy.stats.chi2.ppf(p, df=2) (:329) in closed form."""
    return -2.0 * math.log(1.0 - probability)


def visualize_gaussian_projections(gaussian_params, intrinsic, extrinsic, image_size, colors, probability_threshold=0.97):
    """:337-397: projections of the objects in id order, those with z > 0 drawn far to near: picture = colour alpha + picture
    (1 - alpha), alpha = density / max density; mask = union of the confidence ellipses.  colors: {obj_id: rgb in [0, 1]} (tab20 by
    order of appearance).  -> (uint8 [H,W,3], float32 mask [H,W], {obj_id: colour index})."""
    width, height = image_size
    thr = mahalanobis_threshold(probability_threshold)
    projs, idx = [], {}
    for obj_id in sorted(gaussian_params):
        p = gaussian_params[obj_id]
        d, m, z = project_gaussian_to_2d(np.asarray(p["mean"], F32), np.asarray(p["cov"], F32), intrinsic, extrinsic, image_size)
        if z > 0:
            idx.setdefault(obj_id, len(idx))
            projs.append((z, d, m, np.asarray(colors[idx[obj_id]], F32)))
    projs.sort(key=lambda e: e[0], reverse=True)
    rgb = np.zeros((height, width, 3), F32)
    mask = np.zeros((height, width), F32)
    for z, d, m, col in projs:
        mask = np.maximum(mask, (m <= thr).astype(F32))
        dmax = d.max()
        a = np.clip(d / dmax, 0, 1)[..., None] if dmax > 0 else np.zeros_like(d)[..., None]
        rgb = col.reshape(1, 1, 3) * a + rgb * (1 - a)
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8), mask, idx
