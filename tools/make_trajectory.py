#!/usr/bin/env python3
"""Step 4 of the reference's inference.sh WITHOUT Blender, for the simple cases: writes the two files its Blender export script
(`inference/blender_script/export_blender_custom_trajectories.py`) leaves for the renderer --

  custom_camera_trajectory.npz      `extrinsics` float32 [F, 4, 4]: Blender camera-to-world per frame (camera looks along its -Z, +Y up;
                                    frame 0 of an untouched scene = the input image's camera = looking along world +Y, world +Z up)
  custom_3D_gaussian_trajectory.json  per frame and object the 3D Gaussian in the Blender world (x right, y forward, z up)

-- from step 3's `gaussian_params.json` and a scripted motion: the camera dollies / trucks / yaws linearly over the clip, objects stay at
rest or drift by a constant offset per clip.  Formats and the world convention are pinned by the reference's demo files
(tests/test_render_oracle.py).  Anything beyond a straight line still wants Blender; this makes the chain fit -> render -> sample
runnable headless.

  python tools/make_trajectory.py --gaussian_json <clip>/fitted_3D_gaussian/gaussian_params.json --output_dir <clip>/camera_object_0 \\
         [--num_frames 81] [--dolly 0.5] [--truck 0.0] [--pedestal 0.0] [--yaw_deg 0.0] [--object_shift ID:dx,dy,dz ...]"""
import argparse
import json
import math
import os

import numpy as np

# OpenCV camera / world (x right, y down, z forward) -> Blender world (x right, y forward, z up); the reference's
# COORD_TRANSFORM_CV2BLENDER (inference/rendering_4D_control_maps.py:59-63)
CV2BLENDER = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0]], dtype=np.float64)
# Blender camera axes (x right, y up, -z forward) in the Blender world for the camera of the input image
CAM0 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64)


def camera_trajectory(num_frames, dolly=0.0, truck=0.0, pedestal=0.0, yaw_deg=0.0):
    """-> float32 [F, 4, 4] camera-to-world: position moves linearly to (truck, dolly, pedestal) [right, forward, up, Blender world units =
    the depth map's], the view direction turns linearly by yaw_deg about the world's up axis (positive = to the left)."""
    out = np.tile(np.eye(4), (num_frames, 1, 1))
    for f in range(num_frames):
        a = f / max(num_frames - 1, 1)
        y = math.radians(yaw_deg) * a
        Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
        out[f, :3, :3] = Rz @ CAM0
        out[f, :3, 3] = a * np.array([truck, dolly, pedestal])
    return out.astype(np.float32)


def gaussian_trajectory(fit, num_frames, shifts=None):
    """fit: the dict of gaussian_params.json.  shifts: {object id (str): (dx, dy, dz) in the Blender world, reached at the last frame}."""
    shifts = shifts or {}
    cidx = {str(k): int(v) for k, v in fit.get("obj_id_to_color_idx", {}).items()}
    objs = []
    for oid in sorted(fit["gaussian_params"], key=lambda k: int(k)):
        g = fit["gaussian_params"][oid]
        mean = CV2BLENDER @ np.asarray(g["mean"], dtype=np.float64)
        cov = CV2BLENDER @ np.asarray(g["cov"], dtype=np.float64) @ CV2BLENDER.T
        objs.append((str(oid), cidx.get(str(oid), len(objs)), mean, cov))
    frames = []
    for f in range(num_frames):
        a = f / max(num_frames - 1, 1)
        frames.append({"frame_index": f, "objects": [
            {"object_id": oid, "color_index": ci,
             "gaussian_3d": {"mean": (mean + a * np.asarray(shifts.get(oid, (0, 0, 0)), dtype=np.float64)).tolist(), "covariance": cov.tolist()}}
            for oid, ci, mean, cov in objs]})
    return {"metadata": {"num_objects": len(objs), "num_frames": num_frames, "frame_step": 1,
                         "description": "Ellipsoid Gaussian parameters scripted by tools/make_trajectory.py (no Blender)",
                         "obj_id_to_color_idx": {oid: ci for oid, ci, _, _ in objs}},
            "frames": frames}


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--gaussian_json", required=True)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--num_frames", type=int, default=81)
    p.add_argument("--dolly", type=float, default=0.0, help="camera moves forward by this much over the clip")
    p.add_argument("--truck", type=float, default=0.0, help="... to the right")
    p.add_argument("--pedestal", type=float, default=0.0, help="... up")
    p.add_argument("--yaw_deg", type=float, default=0.0, help="camera turns left by this angle over the clip")
    p.add_argument("--object_shift", action="append", default=[], metavar="ID:dx,dy,dz", help="object ID drifts by (dx, dy, dz) over the clip")
    a = p.parse_args(argv)
    shifts = {}
    for s in a.object_shift:
        oid, vec = s.split(":")
        shifts[oid] = tuple(float(v) for v in vec.split(","))
    fit = json.load(open(a.gaussian_json))
    os.makedirs(a.output_dir, exist_ok=True)
    np.savez(os.path.join(a.output_dir, "custom_camera_trajectory.npz"), extrinsics=camera_trajectory(a.num_frames, a.dolly, a.truck, a.pedestal, a.yaw_deg))
    with open(os.path.join(a.output_dir, "custom_3D_gaussian_trajectory.json"), "w") as fh:
        json.dump(gaussian_trajectory(fit, a.num_frames, shifts), fh, indent=2)
    print(f"wrote custom_camera_trajectory.npz and custom_3D_gaussian_trajectory.json ({a.num_frames} frames) to {a.output_dir}")


if __name__ == "__main__":
    main()
