"""CPU restatement (TEST INFRASTRUCTURE: imported by tests/ only) of the 4D control-map renderer,
inference/rendering_4D_control_maps.py of the reference -- plain torch / numpy on the CPU, float32.

Two classes of functions:

PINNED against fixtures recorded from the reference's own code (tests/golden/make_golden_render.py ->
tests/golden/render_small.safetensors; tests/test_render_oracle.py checks this file against them):
    composite_by_depth        :398-411, 437-453          merge_mask             :736-763
    depth_to_gray             :487-539                   global_depth_range     :541-571
    gaussian_records          :828-873                   density_map            :801-883
    project_gaussians         :573-695                   blend_with_bg          :697-734
    camera_trajectory         :1001-1009                 p3d_cameras            :340-396 (the arithmetic in front of PerspectiveCameras)
    TAB20 / object_color      :885-906 (matplotlib's tab20 palette, its first three channels)

PARITY UNPINNED -- PyTorch3D is neither in the reference tree nor importable here, so these restate its published
algorithms as the reference configures them; they are the SPECIFICATION the HIP kernels are tested against:
    ico_sphere / ellipsoid_mesh   :66-112   (pytorch3d.utils.ico_sphere: icosahedron, `level` 4-way subdivisions, every new vertex
                                             pushed to the unit sphere; vertex ORDER is ours -- it does not affect the image)
    render_points                 :243-338  (PointsRasterizer radius / points_per_pixel = 8, AlphaCompositor, background 0.5)
    render_mesh                   :150-241  (MeshRasterizer blur 0 / one face per pixel / perspective-correct barycentrics,
                                             HardPhongShader with PointLights at the world origin and default materials)
Camera convention for both: the OpenCV world-to-camera matrices of :1001-1009 and pixel intrinsics.  The reference converts them to
PyTorch3D's convention (:363-378: flip x and y of the camera-to-world rotation, in_ndc = False), which renders the same picture as the
plain pinhole u = fx x / z + cx, v = fy y / z + cy with pixel CENTRES at (column + 0.5, row + 0.5); `radius` is in PyTorch3D's NDC
units, in which the SHORTER image side spans [-1, 1]: radius * min(H, W) / 2 pixels."""
import json
import math

import numpy as np
import torch

# matplotlib.colormaps['tab20'] (rows 0..19, RGB), as floats; pinned by the fixture color.float
TAB20 = [
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
]


def object_color(obj_id, obj_id_to_color_idx, return_float=False):
    """:885-906."""
    rgb = TAB20[obj_id_to_color_idx.get(obj_id, 0) % 20]
    if return_float:
        return torch.tensor(rgb, dtype=torch.float32)
    return torch.tensor([c * 255 for c in rgb], dtype=torch.uint8)


# --------------------------------------------------------------------------------------------- depth compositing
def take_fg(bg_depth, fg_depth, fg_mask):
    return fg_mask & ((bg_depth <= 0) | ((fg_depth > 0) & (fg_depth < bg_depth - 1e-6)))


def composite_by_depth(bg_rgb, bg_depth, fg_rgb, fg_depth, fg_mask):
    """:398-411 (any leading batch dims)."""
    t = take_fg(bg_depth, fg_depth, fg_mask)
    return torch.where(t[..., None], fg_rgb, bg_rgb), torch.where(t, fg_depth, bg_depth)


def merge_mask(bg_depth, fg_depth, bg_mask, fg_mask):
    """:746-761 -> uint8 [..., 3]."""
    t = take_fg(bg_depth, fg_depth, fg_mask)
    out = torch.where(t, fg_mask, ~bg_mask)
    return (torch.stack([out, out, out], dim=-1) * 255).to(torch.uint8)


def _range_of(valid):
    if not valid:
        return None
    d = torch.cat(valid)
    if len(d) > 1000000:
        d = d[torch.randperm(len(d))[:1000000]]
    try:
        return torch.quantile(d, 0.001), torch.quantile(d, 0.99)
    except RuntimeError:
        return torch.min(d), torch.max(d)


def global_depth_range(depth_frames_list):
    """:541-571."""
    valid = [d[d > 0].flatten() for frames in depth_frames_list for d in frames if torch.any(d > 0)]
    r = _range_of(valid)
    return (0.0, 1.0) if r is None else (r[0].item(), r[1].item())


def depth_to_gray(depth_frames, gmin=None, gmax=None):
    """:487-539."""
    if gmin is None or gmax is None:
        r = _range_of([d[d > 0].flatten() for d in depth_frames if torch.any(d > 0)])
        lo, hi = (0.0, 1.0) if r is None else r
    else:
        lo, hi = gmin, gmax
    out = []
    for d in depth_frames:
        disp = torch.where(d > 0, 1.0 / d, torch.tensor(0.0))
        if hi > 0 and lo > 0:
            min_disp, max_disp = 1.0 / hi, 1.0 / lo
            disp = (disp - min_disp) / (max_disp - min_disp + 1e-8)
        g = (torch.clamp(disp, 0, 1) * 255).to(torch.uint8)
        out.append(g.unsqueeze(-1).repeat(1, 1, 3))
    return out


# --------------------------------------------------------------------------------------------- projected Gaussians
def gaussian_record(mean, cov, K, R, t):
    """:828-873 for one Gaussian: (valid, mean_2d [2], cov_inv [2,2], coeff) in float32 -- what the per-pixel pass needs."""
    t_vec = t.squeeze() if t.dim() == 2 else t
    m = R @ mean + t_vec
    c = R @ cov @ R.T
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y, z = m[0], m[1], m[2]
    if z <= 0.5:
        return False, None, None, None
    J = torch.tensor([[fx / z, 0, -fx * x / (z * z)], [0, fy / z, -fy * y / (z * z)]], dtype=torch.float32)
    mean_2d = torch.tensor([fx * x / z + cx, fy * y / z + cy], dtype=torch.float32)
    cov_2d = J @ c @ J.T
    cov_2d = cov_2d + torch.eye(2) * 1e-6
    if torch.det(cov_2d) > 1e11:
        return False, None, None, None
    inv = torch.linalg.inv(cov_2d)
    coeff = 1.0 / (2 * torch.pi * torch.sqrt(torch.det(cov_2d)))
    return True, mean_2d, inv, coeff


def density_map(means, covs, K, R, t, image_size):
    """:801-883: image_size = (width, height); pixel (u, v) = (column, row)."""
    W, H = image_size
    u, v = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    pix = torch.stack([u, v], dim=-1).reshape(-1, 2)
    out = torch.zeros(H, W)
    for mean, cov in zip(means, covs):
        ok, m2, inv, coeff = gaussian_record(mean, cov, K, R, t)
        if not ok:
            continue
        diff = pix - m2
        mahal = torch.sum((diff @ inv) * diff, dim=1)
        out += (coeff * torch.exp(-0.5 * mahal)).reshape(H, W)
    return out


def project_gaussians(params_per_frame, color_idx, intrinsics, extrinsics, image_size, threshold=0.05):
    """:573-695 -> (rgb uint8 [H,W,3], alpha float32 [H,W]) per frame."""
    W, H = image_size
    rgbs, alphas = [], []
    for f, params in enumerate(params_per_frame):
        if f >= len(intrinsics) or f >= len(extrinsics):
            break
        K = torch.as_tensor(intrinsics[f]).float()
        E = torch.as_tensor(extrinsics[f]).float()
        R, t = E[:3, :3], E[:3, 3:4]
        lst = []
        for oid, (mean, cov) in params.items():
            mean, cov = torch.as_tensor(mean).float(), torch.as_tensor(cov).float()
            z = (R @ mean + t.squeeze())[2].item()
            d = torch.nan_to_num(density_map(mean[None], cov[None], K, R, t, image_size), nan=0.0, posinf=0.0, neginf=0.0)
            if z > 0:
                mx = d.max()
                nd = d / (mx + 1e-8) if mx > 0 else d
                lst.append((nd, object_color(oid, color_idx, True), z))
        lst.sort(key=lambda e: e[2], reverse=True)
        rgb, alpha = torch.zeros(H, W, 3), torch.zeros(H, W)
        for nd, col, _ in lst:
            a = torch.where(nd > threshold, (nd - threshold) / (1.0 - threshold + 1e-8), torch.zeros_like(nd)).clamp(0.0, 1.0)
            rgb = col.view(1, 1, 3) * a[..., None] + rgb * (1 - a[..., None])
            alpha = a + alpha * (1 - a)
        rgbs.append((rgb.clamp(0, 1) * 255).to(torch.uint8))
        alphas.append(alpha.clamp(0, 1))
    return rgbs, alphas


def blend_with_bg(g_rgb, alpha, bg_rgb):
    """:719-732 (one frame)."""
    m = (g_rgb.float() / 255.0) * alpha[..., None] + (bg_rgb.float() / 255.0) * (1 - alpha[..., None])
    return (m.clamp(0, 1) * 255).to(torch.uint8)


# --------------------------------------------------------------------------------------------- cameras and parameter files
def camera_trajectory(c2w_blender):
    """:1001-1009 on the array of the npz: Blender camera-to-world -> OpenCV world-to-camera."""
    c = torch.as_tensor(np.asarray(c2w_blender).astype(np.float32)).clone()
    c[:, :3, 1:3] *= -1
    return torch.linalg.inv(c)


def p3d_cameras(Ks, Ts):
    """:363-378: what the reference hands to PerspectiveCameras (R, T, focal_length, principal_point)."""
    c2w = torch.linalg.inv(Ts)
    c2w[:, :3, :2] *= -1
    w2c = torch.linalg.inv(c2w)
    return (w2c[:, :3, :3].permute(0, 2, 1), w2c[:, :3, 3], torch.stack([Ks[:, 0, 0], Ks[:, 1, 1]], 1), torch.stack([Ks[:, 0, 2], Ks[:, 1, 2]], 1))


def ellipsoid_parameters(doc):
    """:1012-1051 on the parsed JSON document."""
    frames = []
    for fr in doc["frames"]:
        frames.append({o["object_id"]: (torch.tensor(o["gaussian_3d"]["mean"], dtype=torch.float32),
                                        torch.tensor(o["gaussian_3d"]["covariance"], dtype=torch.float32)) for o in fr["objects"]})
    return frames, dict(doc["metadata"]["obj_id_to_color_idx"])


# --------------------------------------------------------------------------------------------- PyTorch3D restatements (UNPINNED)
def ico_sphere(level):
    """Unit icosphere: 12 vertices / 20 faces, each level splits every face in four and normalises the new vertices."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
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
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return torch.tensor(np.array(verts), dtype=torch.float32), torch.tensor(f, dtype=torch.int32)


def ellipsoid_mesh(mean, cov, scale_factor=2.0, subdivisions=3):
    """:66-112: x = mean + U diag(scale sqrt(max(eval, 1e-8))) u for u on the icosphere."""
    verts, faces = ico_sphere(subdivisions)
    evals, evecs = torch.linalg.eigh(cov.float())
    axes = scale_factor * torch.sqrt(torch.clamp(evals, min=1e-8))
    M = evecs @ torch.diag(axes)
    return verts @ M.T + mean.float(), faces


def _project(p, w2c, K):
    cam = p @ w2c[:3, :3].T + w2c[:3, 3]
    z = cam[:, 2]
    zz = torch.where(z > 1e-8, z, torch.full_like(z, 1e-8))
    return K[0, 0] * cam[:, 0] / zz + K[0, 2], K[1, 1] * cam[:, 1] / zz + K[1, 2], z


def render_points(points, colors_u8, w2c, K, H, W, radius, k_nearest=8, background=0.5):
    """One camera.  Per pixel: the k nearest (in z) points whose projection is closer than the radius to the pixel centre,
    front to back: out = sum_k cum_k w_k c_k, w = 1 - d^2 / r^2, cum_k = prod_{j<k}(1 - w_j); no point: the background."""
    u, v, z = _project(points.float(), w2c, K)
    rp = radius * 0.5 * min(H, W)
    rgb = torch.full((H, W, 3), background)
    depth, mask = torch.zeros(H, W), torch.zeros(H, W, dtype=torch.bool)
    hits = {}
    for i in range(len(points)):
        if not (z[i] > 1e-8):
            continue
        ui, vi = u[i].item(), v[i].item()
        for py in range(max(0, math.floor(vi - rp - 0.5)), min(H - 1, math.ceil(vi + rp - 0.5)) + 1):
            for px in range(max(0, math.floor(ui - rp - 0.5)), min(W - 1, math.ceil(ui + rp - 0.5)) + 1):
                d2 = np.float32(px + 0.5 - np.float32(ui)) ** 2 + np.float32(py + 0.5 - np.float32(vi)) ** 2
                if d2 < np.float32(rp) ** 2:
                    hits.setdefault((py, px), []).append((z[i].item(), i, float(d2)))
    for (py, px), lst in hits.items():
        lst.sort()
        acc, cum = torch.zeros(3), 1.0
        for zz, i, d2 in lst[:k_nearest]:
            w = 1.0 - d2 / (rp * rp)
            acc += cum * w * (colors_u8[i].float() / 255.0)
            cum *= 1.0 - w
        rgb[py, px] = acc
        depth[py, px] = lst[0][0]
        mask[py, px] = True
    return torch.clamp(rgb * 255, 0, 255).to(torch.uint8), depth, mask


def render_mesh(verts, vert_colors, faces, w2c, K, H, W, light=(0.0, 0.0, 0.0), background_u8=0):
    """One camera.  Nearest face whose projection strictly contains the pixel centre (perspective-correct z); flat Phong:
    (0.5 + 0.3 max(n.l, 0)) texel + 0.2 max(v.r, 0)^64 [n.l > 0], light at `light` (world), eye at the camera centre."""
    u, v, z = _project(verts.float(), w2c, K)
    eye = -(w2c[:3, :3].T @ w2c[:3, 3])
    rgb = torch.full((H, W, 3), background_u8, dtype=torch.uint8)
    depth, mask = torch.zeros(H, W), torch.zeros(H, W, dtype=torch.bool)
    zb = torch.full((H, W), float("inf"))
    L = torch.tensor(light)

    def edge(ax, ay, bx, by, px, py):
        return (px - ax) * (by - ay) - (py - ay) * (bx - ax)
    for f in range(len(faces)):
        i0, i1, i2 = (int(i) for i in faces[f])
        if not (z[i0] > 1e-8 and z[i1] > 1e-8 and z[i2] > 1e-8):
            continue
        x0, y0, x1, y1, x2, y2 = (t.item() for t in (u[i0], v[i0], u[i1], v[i1], u[i2], v[i2]))
        area = edge(x0, y0, x1, y1, x2, y2)
        if abs(area) < 1e-12:
            continue
        for py in range(max(0, math.floor(min(y0, y1, y2) - 0.5)), min(H - 1, math.ceil(max(y0, y1, y2) - 0.5)) + 1):
            for px in range(max(0, math.floor(min(x0, x1, x2) - 0.5)), min(W - 1, math.ceil(max(x0, x1, x2) - 0.5)) + 1):
                cx, cy = px + 0.5, py + 0.5
                w0, w1, w2 = edge(x1, y1, x2, y2, cx, cy) / area, edge(x2, y2, x0, y0, cx, cy) / area, edge(x0, y0, x1, y1, cx, cy) / area
                if not (w0 > 0 and w1 > 0 and w2 > 0):
                    continue
                iz = w0 / z[i0].item() + w1 / z[i1].item() + w2 / z[i2].item()
                zz = 1.0 / iz
                if zz < zb[py, px]:
                    zb[py, px] = zz
                    b0, b1, b2 = w0 / z[i0].item() / iz, w1 / z[i1].item() / iz, w2 / z[i2].item() / iz
                    a, b, c = verts[i0], verts[i1], verts[i2]
                    p = b0 * a + b1 * b + b2 * c
                    n = torch.linalg.cross(b - a, c - a)
                    n = n / n.norm().clamp(min=1e-6)
                    l = L - p
                    l = l / l.norm().clamp(min=1e-6)
                    cosang = float(n @ l)
                    vdir = eye - p
                    vdir = vdir / vdir.norm().clamp(min=1e-6)
                    r = -l + 2 * cosang * n
                    spec = 0.2 * (max(float(vdir @ r), 0.0) if cosang > 0 else 0.0) ** 64
                    tex = b0 * vert_colors[i0] + b1 * vert_colors[i1] + b2 * vert_colors[i2]
                    col = (0.5 + 0.3 * max(cosang, 0.0)) * tex + spec
                    rgb[py, px] = (col.clamp(0, 1) * 255).to(torch.uint8)
                    depth[py, px] = zz
                    mask[py, px] = True
    return rgb, depth, mask


def parse_ellipsoid_json(path):
    with open(path) as f:
        return ellipsoid_parameters(json.load(f))
