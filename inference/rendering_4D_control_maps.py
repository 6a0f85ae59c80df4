#!/usr/bin/env python3
"""Rendering 4D control maps on the MI355X engine: same command line and the same five output videos as the reference's
inference/rendering_4D_control_maps.py (:1146-1378) -- background_RGB, background_depth, 3D_gaussian_depth, merged_mask,
3D_gaussian_RGB (+ background_and_3D_gaussian) -- which are exactly what inference/versecrafter_inference.py reads as
--rendering_maps_path.

Every per-pixel stage runs on the HIP engine (versecrafter_amd/rendering/control_maps.py).  The image has no video codec library: videos are
written by the package's own .mp4 writer (H.264 I_PCM, lossless in YCbCr 4:2:0; versecrafter_amd/utils/mp4_pcm.py), which the inference
CLI reads back (versecrafter_amd/utils/video_io.py)."""
import argparse
import logging
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from versecrafter_amd.rendering import control_maps as R

logger = logging.getLogger("rendering_4D_control_maps")
logging.basicConfig(level=logging.INFO)


def parse_args(argv=None):
    """The reference's flags (:1146-1168), same names and defaults."""
    p = argparse.ArgumentParser(description="Inference mode: Render video from pre-computed parameters")
    p.add_argument("--png_path", type=str, required=False, help="Path to first frame PNG image (optional)")
    p.add_argument("--video_path", type=str, required=False, help="Path to input MP4 video (optional)")
    p.add_argument("--npz_path", type=str, required=True, help="Path to NPZ file with depth and camera pose")
    p.add_argument("--mask_dir", type=str, required=False, help="Directory containing mask images")
    p.add_argument("--mask_video", type=str, required=False, help="Path to mask video (mp4)")
    p.add_argument("--trajectory_npz", type=str, required=True, help="Path to camera trajectory NPZ file")
    p.add_argument("--ellipsoid_json", type=str, required=True, help="Path to ellipsoid parameters JSON file")
    p.add_argument("--output_dir", type=str, default="outputs/inference", help="Output directory")
    p.add_argument("--device", type=str, default="cuda", help="Device to use")
    p.add_argument("--point_size", type=float, default=0.005, help="Point size for rendering")
    p.add_argument("--fps", type=int, default=10, help="Output video FPS")
    p.add_argument("--render_batch_size", type=int, default=27, help="Batch size for rendering")
    p.add_argument("--use_fp16", action="store_true", help="Use FP16 for rendering")
    p.add_argument("--pin_memory", action="store_true", help="Use pinned memory")
    p.add_argument("--ellipsoid_subdiv", type=int, default=3, help="Icosphere subdivisions for ellipsoid mesh")
    p.add_argument("--trajectory_radius", type=float, default=0.03, help="Trajectory line radius")
    p.add_argument("--gaussian_mask_threshold", type=float, default=0.003, help="Gaussian projection threshold")
    p.add_argument("--sample_frames", type=int, default=10, help="Number of frames to sample")
    return p.parse_args(argv)


def save_video_from_frames(frames, output_path: Path, fps: int = 10):
    """:455-485: a list of uint8 [H,W,3] (or [H,W]) frames -> video.  No codec library in the image: the package's own .mp4 writer
    (H.264 I_PCM, packed on the GPU; versecrafter_amd/utils/video_io.py), which the inference CLI of this repo reads back."""
    if len(frames) == 0:
        logger.warning(f"No frames to save for {output_path}")
        return None
    if frames[0].ndim == 2:
        frames = [f.unsqueeze(-1).repeat(1, 1, 3) for f in frames]
    from versecrafter_amd.utils.video_io import save_frames
    return save_frames(torch.stack(list(frames)), str(output_path), fps)


def main(argv=None):
    """The reference's main() (:1171-1378), step for step."""
    args = parse_args(argv)
    device = args.device
    if not str(device).startswith("cuda") or not torch.cuda.is_available():
        raise SystemExit("rendering_4D_control_maps: a HIP device is required (versecrafter_amd has no CPU path)")
    out = Path(args.output_dir)
    out.mkdir(parents=True, exist_ok=True)

    logger.info("Step 1: Loading background point cloud")
    bg_points, bg_colors, K0, _, H, W = R.build_background(args.png_path, args.npz_path, args.mask_dir, device)
    logger.info("Step 2: Loading camera trajectory")
    extrinsics = R.load_camera_trajectory(args.trajectory_npz, device)
    nF = len(extrinsics)
    intrinsics = K0 if K0.ndim == 3 else K0.unsqueeze(0).repeat(nF, 1, 1)
    logger.info("Step 3: Loading 3D Gaussian trajectory")
    params, color_idx, _ = R.load_ellipsoid_parameters(args.ellipsoid_json, device)
    logger.info("Step 4: Building ellipsoid meshes")
    meshes = []
    for f in range(nF):
        fp = params[f] if f < len(params) else {}
        ms = [R.make_ellipsoid_mesh(mean, cov, scale_factor=2.5, subdivisions=args.ellipsoid_subdiv,
                                    color_rgb255=R.get_object_color(oid, color_idx, device), device=device) for oid, (mean, cov) in fp.items()]
        meshes.append(R.combine_meshes_for_scene(ms) if ms else None)

    logger.info("Step 5: Rendering background")
    common = dict(point_size=args.point_size, device=device, batch_size=args.render_batch_size, use_fp16=args.use_fp16,
                  pin_memory=args.pin_memory)
    bg_rgb, bg_depth, bg_masks, _ = R.render_video_with_bg_and_fg(bg_points, bg_colors, meshes, intrinsics, extrinsics, (H, W),
                                                                   mode="background", **common)
    written = [save_video_from_frames(bg_rgb, out / "background_RGB.mp4", args.fps)]
    logger.info("Rendering ellipsoid foreground")
    fg_rgb, fg_depth, _, fg_masks = R.render_video_with_bg_and_fg(bg_points, bg_colors, meshes, intrinsics, extrinsics, (H, W),
                                                                   mode="foreground", **common)
    logger.info("Computing background and foreground depth")
    _, comb_depth, _, _ = R.merge_bg_and_fg_sequences(bg_rgb, bg_depth, bg_masks, fg_rgb, fg_depth, fg_masks)
    gmin, gmax = R.compute_global_depth_range([bg_depth, fg_depth, comb_depth])
    logger.info(f"Global depth range: min={gmin:.4f}, max={gmax:.4f}")
    written.append(save_video_from_frames(R.visualize_depth_as_grayscale(bg_depth, gmin, gmax), out / "background_depth.mp4", args.fps))
    written.append(save_video_from_frames(R.visualize_depth_as_grayscale(fg_depth, gmin, gmax), out / "3D_gaussian_depth.mp4", args.fps))
    written.append(save_video_from_frames(R.merge_bg_and_fg_mask(bg_depth, fg_depth, bg_masks, fg_masks, device=device),
                                          out / "merged_mask.mp4", args.fps))

    logger.info("Generating 3D Gaussian RGB projections")
    if len(params) > 0 and any(len(fp) > 0 for fp in params):
        while len(params) < nF:
            params.append({})
        g_rgb, g_alpha = R.project_3d_gaussians_to_2d(params, color_idx, intrinsics.cpu().numpy(), extrinsics.cpu().numpy(), (W, H),
                                                      threshold=args.gaussian_mask_threshold, device=device)
        proj = R.mask_gaussian_projection(g_rgb, g_alpha)
    else:
        g_rgb = [torch.zeros((H, W, 3), dtype=torch.uint8, device=device) for _ in range(nF)]
        g_alpha = [torch.zeros((H, W), dtype=torch.float32, device=device) for _ in range(nF)]
        proj = [torch.zeros((H, W, 3), dtype=torch.uint8, device=device) for _ in range(nF)]
        logger.info("No objects detected; generated empty Gaussian projection video")
    written.append(save_video_from_frames(proj, out / "3D_gaussian_RGB.mp4", args.fps))

    logger.info("Generating background + 3D Gaussian composite video")
    with_bg = R.blend_gaussian_projection_with_bg(g_rgb, g_alpha, bg_rgb[:len(g_rgb)])
    vis = [a > 0.001 for a in g_alpha]
    with_bg, _, _, _ = R.merge_bg_and_fg_sequences(bg_rgb[:len(g_rgb)], bg_depth[:len(g_rgb)], bg_masks[:len(g_rgb)], with_bg,
                                                   fg_depth[:len(g_rgb)], vis)
    written.append(save_video_from_frames(with_bg, out / "background_and_3D_gaussian.mp4", args.fps))
    logger.info("Rendering complete: " + ", ".join(str(w) for w in written))
    return written


if __name__ == "__main__":
    main()
