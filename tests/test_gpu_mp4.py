"""The package's own .mp4 writer / reader on the GPU: csrc/h264pcm.hip (through the C ABI) bit-exact against oracle/h264_pcm_oracle.py,
files written by the product decoded by the oracle's independent generic-syntax H.264 / MP4 decoder, and the CLI-facing chain
save_video -> read_video."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import h264_pcm_oracle as PO  # noqa: E402
from versecrafter_amd.utils import mp4_pcm as M  # noqa: E402
from versecrafter_amd.utils import video_io  # noqa: E402


def frames_rgb(F, H, W, seed=0):
    rng = np.random.default_rng(seed)
    fr = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    fr[0] = 0
    if F > 1:
        fr[1] = 255
    if F > 2:
        fr[2] = (rng.random((H, W, 1)) < 0.5) * np.uint8(255)
    return fr


@pytest.mark.parametrize("F,H,W", [(3, 16, 16), (4, 32, 48), (3, 30, 50), (2, 2, 2), (3, 120, 208), (2, 480, 832)])
def test_pack_and_unpack_are_bit_exact(F, H, W):
    fr = frames_rgb(F, H, W, seed=H + W)
    want = PO.pack(fr)
    got = M.pack_frames(torch.from_numpy(fr).cuda())
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)          # incl. macroblock headers and edge replication
    assert int(got[..., 2:].min()) >= 1                                                  # pcm samples shall not be 0
    back = M.unpack_frames(got, H, W)
    assert np.array_equal(back.cpu().numpy(), PO.unpack(want, H, W))
    with pytest.raises(ValueError):
        M.unpack_frames(got, H + 16, W)


@pytest.mark.parametrize("F,H,W,fps", [(5, 48, 64, 16), (4, 30, 50, 10)])
def test_written_file_is_decodable_and_reads_back(tmp_path, F, H, W, fps):
    fr = frames_rgb(F, H, W, seed=1)
    p = M.write_mp4(str(tmp_path / "v.mp4"), torch.from_numpy(fr).cuda(), fps=fps)
    d = PO.decode_file(p)                                                                # independent decoder, generic syntax
    want = PO.unpack(PO.pack(fr), H, W)
    assert d["fps"] == fps and np.array_equal(d["frames"], want)
    mine = M.read_mp4(p)
    assert mine.is_cuda and np.array_equal(mine.cpu().numpy(), want)
    assert np.array_equal(M.read_mp4(p, max_frames=2).cpu().numpy(), want[:2])
    assert np.array_equal(want[0], fr[0]) and np.array_equal(want[1], fr[1]) and np.array_equal(want[2], fr[2])   # black / white / mask: lossless
    def luma(a):
        a = a.astype(int)
        return (66 * a[..., 0] + 129 * a[..., 1] + 25 * a[..., 2] + 128) >> 8
    d = np.abs(luma(want[3:]) - luma(fr[3:]))                                            # white noise: luma survives (but for out-of-gamut clipping),
    assert d.mean() < 0.5 and (d > 2).mean() < 0.05                                      # chroma is 2x2-averaged (4:2:0)


def test_grayscale_and_odd_sizes(tmp_path):
    g = torch.randint(0, 256, (3, 32, 32), dtype=torch.uint8, device="cuda")
    p = M.write_mp4(str(tmp_path / "g.mp4"), g)
    back = M.read_mp4(p)
    assert torch.equal(back[..., 0], back[..., 1]) and (back[..., 0].int() - g.int()).abs().max() <= 1
    with pytest.raises(ValueError):
        M.write_mp4(str(tmp_path / "odd.mp4"), torch.zeros(2, 31, 32, 3, dtype=torch.uint8, device="cuda"))


def test_cli_chain_save_video_then_read_video(tmp_path):
    """What the two CLIs do: the renderer / sampler writes `<name>.mp4`, the inference CLI reads it as [1, 3, F, H, W] in [0, 1]."""
    F, H, W = 9, 64, 96
    x = torch.linspace(0, 1, W, device="cuda").view(1, 1, 1, 1, W).expand(1, 3, F, H, W).clone()
    x[:, :, :, :, : W // 2] *= torch.linspace(0, 1, F, device="cuda").view(1, 1, F, 1, 1)
    mask = (torch.rand(1, 1, F, H, W, device="cuda") < 0.3).float().expand(1, 3, F, H, W)
    out = video_io.save_video(x, str(tmp_path / "maps" / "background_depth.mp4"), fps=10)
    assert out.endswith(".mp4") and os.path.getsize(out) > F * H * W * 3 // 2
    v = video_io.read_video(out, F, (H, W))
    assert v.shape == (1, 3, F, H, W) and (v - x.cpu()).abs().max() < 1.5 / 255         # greys: one level
    m = video_io.read_video(video_io.save_video(mask, str(tmp_path / "maps" / "merged_mask.mp4"), fps=10), F, (H, W))
    assert torch.equal(m, mask.cpu())                                                    # a 0 / 1 mask survives exactly
    assert video_io.read_video(out, 4, (32, 48)).shape == (1, 3, 4, 32, 48)
    foreign = bytearray(open(out, "rb").read())
    at = foreign.find(b"avcC") + 4 + 8
    foreign[at + 1] = 100
    open(tmp_path / "maps" / "x264.mp4", "wb").write(foreign)
    with pytest.raises(RuntimeError, match="High profile"):
        video_io.read_video(str(tmp_path / "maps" / "x264.mp4"), F, (H, W))


def test_full_size_video_timing(tmp_path):
    """81 x 720 x 1280 (cfg-4's frame size): one pack launch, file write, read back; sizes and times printed."""
    F, H, W = 81, 720, 1280
    fr = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    payload = M.pack_frames(fr)
    torch.cuda.synchronize()
    t1 = time.time()
    p = M.write_mp4(str(tmp_path / "full.mp4"), fr, fps=16)
    t2 = time.time()
    back = M.read_mp4(p)
    torch.cuda.synchronize()
    t3 = time.time()
    assert os.path.getsize(p) > payload.numel() and back.shape == fr.shape
    assert torch.equal(back, M.unpack_frames(payload, H, W))
    print(f"\n81x720x1280: pack {1e3 * (t1 - t0):.1f} ms, write_mp4 {t2 - t1:.2f} s ({os.path.getsize(p) >> 20} MiB), read_mp4 {t3 - t2:.2f} s")
