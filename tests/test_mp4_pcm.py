"""The package's own .mp4 writer / reader (versecrafter_amd/utils/mp4_pcm.py): host half on the CPU - parameter sets, slice headers and
MP4 boxes around a payload packed by the numpy oracle - decoded by the INDEPENDENT generic-syntax decoder of oracle/h264_pcm_oracle.py."""
import numpy as np
import pytest

from oracle import h264_pcm_oracle as PO
from versecrafter_amd.utils import mp4_pcm as M


def frames_rgb(F, H, W, seed=0):
    rng = np.random.default_rng(seed)
    fr = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    fr[0] = 0                                             # black, white and a 0 / 255 mask frame: the cases the CLI thresholds
    if F > 1:
        fr[1] = 255
    if F > 2:
        fr[2] = (rng.random((H, W, 1)) < 0.5) * np.uint8(255)
    return fr


@pytest.mark.parametrize("F,H,W,fps", [(3, 32, 48, 16), (4, 30, 50, 10), (2, 16, 16, 24), (5, 480 // 4, 832 // 4, 16)])
def test_written_file_decodes_by_the_generic_syntax(tmp_path, F, H, W, fps):
    fr = frames_rgb(F, H, W, seed=H)
    payload = PO.pack(fr)
    p = M.mux(str(tmp_path / "v.mp4"), payload, H, W, fps)
    d = PO.decode_file(p)
    assert d["frames"].shape == (F, H, W, 3) and d["fps"] == fps
    assert np.array_equal(d["frames"], PO.unpack(payload, H, W))
    s = d["sps"]
    assert (s["profile_idc"], s["constraints"] & 0xC0, s["level_idc"]) == (66, 0xC0, 51)
    assert s["colour"] == (6, 6, 6) and s["full_range"] == 0 and s["fixed_frame_rate"] == 1 and s["restriction"][4:] == [0, 1]
    assert all(h["disable_deblocking"] == 1 and h["slice_type"] == 7 and h["qp"] == 26 for h in d["headers"])
    # the package's own reader takes the file back to the same payload, frame size and rate
    back, H2, W2, fps2 = M.demux(p)
    assert (H2, W2, fps2) == (H, W, fps) and np.array_equal(back, payload)
    assert M.demux(p, max_frames=1)[0].shape[0] == 1


def test_round_trip_loss_is_the_colour_conversion_only():
    fr = frames_rgb(4, 64, 96, seed=7)
    back = PO.unpack(PO.pack(fr), 64, 96)
    assert np.array_equal(back[0], fr[0]) and np.array_equal(back[1], fr[1]) and np.array_equal(back[2], fr[2])   # black / white / mask: exact
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)[None].repeat(16, 0).transpose(1, 0, 2, 3).reshape(1, 16, 256, 3)
    g2 = PO.unpack(PO.pack(grey), 16, 256)
    assert np.abs(g2.astype(int) - grey.astype(int)).max() <= 1                                                   # greys: one level (219 luma codes)
    smooth = np.stack(list(np.meshgrid(np.linspace(0, 255, 96), np.linspace(0, 255, 64))) + [np.full((64, 96), 90.0)], -1).astype(np.uint8)[None]
    s2 = PO.unpack(PO.pack(smooth), 64, 96)
    assert np.abs(s2.astype(int) - smooth.astype(int)).max() <= 6                                                 # smooth colour: conversion rounding + 2x2 chroma


def test_bit_syntax_helpers():
    w = M.BitWriter().ue(0).ue(25).se(-3).se(2).u(5, 19).trailing()
    r = M.BitReader(w.bytes())
    assert (r.ue(), r.ue(), r.se(), r.se(), r.u(5)) == (0, 25, -3, 2, 19)
    raw = bytes([0, 0, 1, 0, 0, 0, 0, 3, 7, 0, 0])
    assert M.escape(raw) == bytes([0, 0, 3, 1, 0, 0, 3, 0, 0, 3, 3, 7, 0, 0]) and M.unescape(M.escape(raw)) == raw
    for f in range(4):                                    # slice prefixes end in a non-zero byte: no start-code emulation across the sample boundary
        assert M.slice_prefix(f)[-1] != 0 and M.slice_prefix(f)[0] == 0x65
    assert PO.parse_sps(M.sps_nal(720, 1280, 16))["crop"] == [0, 0, 0, 0]
    s = PO.parse_sps(M.sps_nal(480, 832, 10))
    assert (s["W"], s["H"], s["time_scale"] / (2 * s["num_units_in_tick"])) == (832, 480, 10)
    assert PO.parse_sps(M.sps_nal(30, 50, 16))["crop"] == [0, 7, 0, 1]
    with pytest.raises(ValueError):
        M.sps_nal(31, 50, 16)


def test_foreign_streams_are_refused_with_a_reason(tmp_path):
    """The reference's demo clips are x264 High profile / CABAC: the reader says so instead of returning garbage."""
    fr = frames_rgb(2, 32, 32)
    p = M.mux(str(tmp_path / "v.mp4"), PO.pack(fr), 32, 32, 16)
    raw = bytearray(open(p, "rb").read())
    at = raw.find(b"avcC")
    sps_at = at + 4 + 8
    assert raw[sps_at] == 0x67 and raw[sps_at + 1] == 66
    hi = bytearray(raw); hi[sps_at + 1] = 100
    open(tmp_path / "high.mp4", "wb").write(hi)
    with pytest.raises(M.UnsupportedVideo, match="High profile"):
        M.demux(str(tmp_path / "high.mp4"))
    cut = bytearray(raw); cut[raw.find(b"mdat") + 4 + 4] = 0x61          # first sample: a non-IDR slice NAL
    open(tmp_path / "p.mp4", "wb").write(cut)
    with pytest.raises(M.UnsupportedVideo, match="non-IDR"):
        M.demux(str(tmp_path / "p.mp4"))
    open(tmp_path / "nomoov.mp4", "wb").write(raw[:raw.find(b"moov") - 4])
    with pytest.raises(M.UnsupportedVideo, match="moov"):
        M.demux(str(tmp_path / "nomoov.mp4"))
    with pytest.raises(RuntimeError, match="GPU"):
        M.pack_frames(__import__("torch").zeros(1, 16, 16, 3, dtype=__import__("torch").uint8))
