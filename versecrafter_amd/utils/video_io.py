"""Frame I/O of the CLI: stands in for videox_fun.utils.utils.{get_video_to_video_latent, get_image_latent, save_videos_grid}
(third-party, un-vendored; call sites inference/versecrafter_inference.py:370-403, 456 -- PARITY UNPINNED: decoder, resize
filter and codec settings of those helpers are not in the reference tree).

The build image has no video codec library (no ffmpeg / cv2 / imageio / decord / av).  Writing: a codec library when one is
importable, else the package's own .mp4 writer -- H.264 with every macroblock I_PCM, sample planes packed on the GPU
(utils/mp4_pcm.py, csrc/h264pcm.hip): real, playable .mp4 files, lossless in YCbCr 4:2:0 -- and only without a GPU a uint8 frame dump
(.npy).  Reading a control map:
  1. a frame dump next to the .mp4 -- `<name>.npy` (uint8 [F,H,W,3] or [F,H,W]), `<name>.npz` (key "frames") or
     `<name>.safetensors` (key "frames") -- which any machine with a decoder can produce once, then
  2. the .mp4 itself through whichever decoder is importable (imageio, cv2, decord, av), then
  3. the .mp4 through the package's own reader when it is one of its own I_PCM files (what the renderer CLI of this repo writes), else
  4. an error that says what the file is (the reference's demo clips are x264 High profile: they need a real decoder).
Videos are returned as the reference's helpers return them: float32 [1, 3, F, H, W] in [0, 1], resized to `sample_size` (H, W)
(bilinear), cut to `video_length` frames."""
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
        why = "no GPU for the package's own I_PCM reader"
        if torch.cuda.is_available():
            from . import mp4_pcm
            try:
                frames = mp4_pcm.read_mp4(path, video_length).cpu().numpy()
            except mp4_pcm.UnsupportedVideo as e:
                why = str(e)
        if frames is None:
            raise RuntimeError(f"{path}: no video decoder is importable (imageio / cv2 / decord / av), there is no frame dump "
                               f"{stem}.npy|.npz|.safetensors, and: {why}.  Decode it once elsewhere (uint8 [F,H,W,3])")
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


def save_frames(frames, path, fps=16):
    """frames: uint8 [F, H, W, 3] (torch, any device) -> `path` (.mp4).  Returns the path actually written: a codec library's .mp4
    when one is importable, else the package's own I_PCM .mp4 (GPU), else a frame dump `<stem>.npy` + the first frame as .png."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    try:
        import imageio
        imageio.mimsave(path, list(frames.cpu().numpy()), fps=fps)
        return path
    except ImportError:
        pass
    try:
        import cv2
        host = frames.cpu().numpy()
        w = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (host.shape[2], host.shape[1]))
        for f in host:
            w.write(cv2.cvtColor(f, cv2.COLOR_RGB2BGR))
        w.release()
        return path
    except ImportError:
        pass
    F_, H, W = frames.shape[:3]
    if torch.cuda.is_available() and H % 2 == 0 and W % 2 == 0:
        from . import mp4_pcm
        return mp4_pcm.write_mp4(path, frames if frames.is_cuda else frames.cuda(), fps=fps)
    out = os.path.splitext(path)[0] + ".npy"
    host = frames.cpu().numpy()
    np.save(out, host)
    from PIL import Image
    Image.fromarray(host[0]).save(os.path.splitext(path)[0] + "_frame0.png")
    return out


def save_video(sample, path, fps=16):
    """sample [B, 3, F, H, W] in [0, 1] (first item is written).  Returns the path actually written."""
    frames = (sample[0].clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 3, 0).contiguous()          # [F, H, W, 3]
    if frames.shape[0] == 1:
        from PIL import Image
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        out = os.path.splitext(path)[0] + ".png"
        Image.fromarray(frames[0].cpu().numpy()).save(out)
        return out
    return save_frames(frames, path, fps)
