#!/bin/bash
# GPU box: the renderer CLI on the reference's REAL step-3/4 files of the street demo clip (tests/golden/demo_fit/street) with a random
# 1280x720 first frame (the clip's 0001.png is not a fixture); trajectories cut to the 4 exported frames.
set -e
cd "$(dirname "$0")/.."
d=tests/golden/demo_fit/street
w=/tmp/real_clip; rm -rf $w; mkdir -p $w
python - <<PY
import numpy as np, json
from PIL import Image
rs=np.random.RandomState(0)
Image.fromarray(rs.randint(0,255,(720,1280,3),dtype=np.uint8)).save("$w/0001.png")
z=np.load("$d/custom_camera_trajectory.npz")["extrinsics"]
np.savez("$w/cam.npz", extrinsics=z[[0,1,40,80]])
j=json.load(open("$d/custom_3D_gaussian_trajectory_frames_0_1_40_80.json"))
for i,f in enumerate(j["frames"]): f["frame_index"]=i
j["metadata"]["num_frames"]=4
json.dump(j,open("$w/ell.json","w"))
PY
python inference/fit_3D_gaussian.py --npz_path $d/depth_intrinsics.npz --masks_dir $d/masks --output_dir $w/fit --image_path $w/0001.png 2>&1 | tail -2
python inference/rendering_4D_control_maps.py --png_path $w/0001.png --npz_path $d/depth_intrinsics.npz --mask_dir $d/masks --trajectory_npz $w/cam.npz --ellipsoid_json $w/ell.json --output_dir $w/maps 2>&1 | tail -3
python - <<PY
import sys; sys.path.insert(0,".")
from versecrafter_amd.utils import mp4_pcm
import os
for n in sorted(os.listdir("$w/maps")):
    a=mp4_pcm.read_mp4("$w/maps/"+n).cpu().numpy()
    print(n, a.shape, round(float(a.mean()),2), int((a>0).mean()*100), "% nonzero")
PY
ls $w/fit
