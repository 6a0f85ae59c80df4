"""Frame I/O of the CLI: stands in for videox_fun.utils.utils.{get_video_to_video_latent, get_image_latent, save_videos_grid}
(third-party, un-vendored; call sites inference/versecrafter_inference.py:370-403, 456 -- PARITY UNPINNED: decoder, resize
filter and codec settings of those helpers are not in the reference tree).

The build image has no video codec (no ffmpeg / cv2 / imageio / decord / av).  A control map is therefore looked up as
  1. a frame dump next to the .mp4 -- `<name>.npy` (uint8 [F,H,W,3] or [F,H,W]), `<name>.npz` (key "frames") or
     `<name>.safetensors` (key "frames") -- which any machine with a decoder can produce once, then
  2. the .mp4 itself through whichever decoder is importable (imageio, cv2, decord, av), else
  3. an error that says so.
Videos are returned as the reference's helpers return them: float32 [1, 3, F, H, W] in [0, 1], resized to `sample_size` (H, W)
(bilinear), cut to `video_length` frames.  save_video writes .mp4 when imageio / cv2 is importable, else a uint8 frame dump
(.npy) plus the first frame as .png (PIL is in the image)."""
import os

import numpy as np
import torch
import torch.nn.functional as F


def _frames_from_dump(stem):
    if os.path.isfile(stem + ".npy"):
        return np.load(stem + ".npy")
    if os.path.isfile(stem + ".npz"):
        return np.load(stem + ".npz")["frames"]
    if os.path.isfile(stem + ".safetensors"):
        from safetensors.numpy import load_file
        return load_file(stem + ".safetensors")["frames"]
    return None


def _frames_from_codec(path, max_frames):
    try:
        import imageio.v3 as iio
        return np.stack([f for _, f in zip(range(max_frames), iio.imiter(path))])
    except ImportError:
        pass
    try:
        import cv2
        cap, out = cv2.VideoCapture(path), []
        while len(out) < max_frames:
            ok, f = cap.read()
            if not ok:
                break
            out.append(cv2.cvtColor(f, cv2.COLOR_BGR2RGB))
        cap.release()
        return np.stack(out)
    except ImportError:
        pass
    try:
        import decord
        vr = decord.VideoReader(path)
        return vr.get_batch(list(range(min(max_frames, len(vr))))).asnumpy()
    except ImportError:
        pass
    try:
        import av
        with av.open(path) as c:
            return np.stack([f.to_ndarray(format="rgb24") for _, f in zip(range(max_frames), c.decode(video=0))])
    except ImportError:
        pass
    return None


def read_video(path, video_length, sample_size):
    """`path`: the .mp4 the reference names (a frame dump with the same stem takes precedence).  -> [1, 3, F, H, W] in [0, 1]."""
    stem = os.path.splitext(path)[0]
    frames = _frames_from_dump(stem)
    if frames is None and os.path.isfile(path):
        frames = _frames_from_codec(path, video_length)
    if frames is None:
        if not os.path.isfile(path):
            raise FileNotFoundError(path)
        raise RuntimeError(f"no video decoder is importable (imageio / cv2 / decord / av) and there is no frame dump "
                           f"{stem}.npy|.npz|.safetensors for {path}: decode it once elsewhere (uint8 [F,H,W,3])")
    frames = np.asarray(frames)[:video_length]
    if frames.ndim == 3:
        frames = np.repeat(frames[..., None], 3, axis=-1)
    v = torch.from_numpy(np.ascontiguousarray(frames)).float().permute(0, 3, 1, 2) / 255.0          # [F, 3, h, w]
    H, W = sample_size
    if tuple(v.shape[2:]) != (H, W):
        v = F.interpolate(v, size=(H, W), mode="bilinear", align_corners=False)
    return v.permute(1, 0, 2, 3).unsqueeze(0).contiguous()


def read_image(path, sample_size):
    """-> [1, 3, 1, H, W] in [0, 1] (get_image_latent)."""
    from PIL import Image
    H, W = sample_size
    img = Image.open(path).convert("RGB").resize((W, H))
    a = torch.from_numpy(np.asarray(img).copy()).float().permute(2, 0, 1) / 255.0
    return a[None, :, None]


def save_video(sample, path, fps=16):
    """sample [B, 3, F, H, W] in [0, 1] (first item is written).  Returns the path actually written."""
    frames = (sample[0].clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 3, 0).cpu().numpy()          # [F, H, W, 3]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if frames.shape[0] == 1:
        from PIL import Image
        out = os.path.splitext(path)[0] + ".png"
        Image.fromarray(frames[0]).save(out)
        return out
    try:
        import imageio
        imageio.mimsave(path, list(frames), fps=fps)
        return path
    except ImportError:
        pass
    try:
        import cv2
        w = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (frames.shape[2], frames.shape[1]))
        for f in frames:
            w.write(cv2.cvtColor(f, cv2.COLOR_RGB2BGR))
        w.release()
        return path
    except ImportError:
        pass
    out = os.path.splitext(path)[0] + ".npy"
    np.save(out, frames)
    from PIL import Image
    Image.fromarray(frames[0]).save(os.path.splitext(path)[0] + "_frame0.png")
    return out
