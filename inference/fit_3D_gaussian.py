#!/usr/bin/env python3
"""Fit one 3D Gaussian per segmented object on the MI355X engine: same command line and the same outputs
(`gaussian_params.json`, `gaussian_projection.png`, `gaussian_overlay_on_image.png`) as the reference's inference/fit_3D_gaussian.py
(:633-712; step 3 of inference.sh).  Every per-pixel stage runs on the HIP engine (versecrafter_amd/rendering/gaussian_fit.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from versecrafter_amd.rendering import gaussian_fit

if __name__ == "__main__":
    sys.exit(gaussian_fit.main())
