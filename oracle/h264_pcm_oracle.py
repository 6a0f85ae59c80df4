"""TEST INFRASTRUCTURE - CPU restatement of the package's own .mp4 writer (versecrafter_amd/utils/mp4_pcm.py + csrc/h264pcm.hip).
Only tests/ may import this file.

The reference writes its videos through imageio / ffmpeg (third-party, absent: inference/versecrafter_inference.py:456,
inference/rendering_4D_control_maps.py:455-485), so there is no reference vector for the bytes of a file - PARITY UNPINNED at that
level, by nature.  What this oracle pins instead is conformance and losslessness:
  * `pack` / `unpack`: the integer BT.601 conversions, macroblock order, edge replication (numpy) - the HIP kernels must match
    bit for bit;
  * `decode_stream`: an independent decoder of the written stream that follows the GENERIC syntax of ITU-T H.264 (7.3.2.1.1 sequence
    parameter set incl. VUI, 7.3.2.2 picture parameter set, 7.3.3 slice header, 7.3.4 slice data, 7.3.5 macroblock layer) bit by
    bit - it does not know where the writer put things, so a misplaced or missing syntax element desynchronises it - and rejects every
    feature outside the I_PCM subset.  MP4 boxes are walked generically as well."""
import struct

import numpy as np


# ----------------------------------------------------------------------------------------------------------- samples
def pack(frames: np.ndarray) -> np.ndarray:
    """uint8 RGB [F, H, W, 3] -> uint8 [F, mbh * mbw, 386]."""
    F, H, W, _ = frames.shape
    mbw, mbh = (W + 15) // 16, (H + 15) // 16
    ys, xs = np.minimum(np.arange(mbh * 16), H - 1), np.minimum(np.arange(mbw * 16), W - 1)
    p = frames[:, ys][:, :, xs].astype(np.int32)                                            # edge replication
    r, g, b = p[..., 0], p[..., 1], p[..., 2]
    Y = np.clip(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16, 1, 255)
    q = (p[:, 0::2, 0::2] + p[:, 0::2, 1::2] + p[:, 1::2, 0::2] + p[:, 1::2, 1::2] + 2) >> 2
    ar, ag, ab = q[..., 0], q[..., 1], q[..., 2]
    Cb = np.clip(((-38 * ar - 74 * ag + 112 * ab + 128) >> 8) + 128, 1, 255)
    Cr = np.clip(((112 * ar - 94 * ag - 18 * ab + 128) >> 8) + 128, 1, 255)
    out = np.empty((F, mbh * mbw, 386), np.uint8)
    out[..., 0], out[..., 1] = 0x0D, 0x00
    out[..., 2:258] = Y.reshape(F, mbh, 16, mbw, 16).transpose(0, 1, 3, 2, 4).reshape(F, mbh * mbw, 256)
    out[..., 258:322] = Cb.reshape(F, mbh, 8, mbw, 8).transpose(0, 1, 3, 2, 4).reshape(F, mbh * mbw, 64)
    out[..., 322:386] = Cr.reshape(F, mbh, 8, mbw, 8).transpose(0, 1, 3, 2, 4).reshape(F, mbh * mbw, 64)
    return out


def planes(payload: np.ndarray, mbh: int, mbw: int):
    F = payload.shape[0]
    Y = payload[..., 2:258].reshape(F, mbh, mbw, 16, 16).transpose(0, 1, 3, 2, 4).reshape(F, mbh * 16, mbw * 16)
    Cb = payload[..., 258:322].reshape(F, mbh, mbw, 8, 8).transpose(0, 1, 3, 2, 4).reshape(F, mbh * 8, mbw * 8)
    Cr = payload[..., 322:386].reshape(F, mbh, mbw, 8, 8).transpose(0, 1, 3, 2, 4).reshape(F, mbh * 8, mbw * 8)
    return Y, Cb, Cr


def yuv_to_rgb(Y, Cb, Cr, H, W) -> np.ndarray:
    C = 298 * (Y.astype(np.int32) - 16)
    D = np.repeat(np.repeat(Cb.astype(np.int32) - 128, 2, axis=1), 2, axis=2)
    E = np.repeat(np.repeat(Cr.astype(np.int32) - 128, 2, axis=1), 2, axis=2)
    rgb = np.stack([(C + 409 * E + 128) >> 8, (C - 100 * D - 208 * E + 128) >> 8, (C + 516 * D + 128) >> 8], -1)
    return np.clip(rgb, 0, 255).astype(np.uint8)[:, :H, :W]


def unpack(payload: np.ndarray, H: int, W: int) -> np.ndarray:
    return yuv_to_rgb(*planes(payload, (H + 15) // 16, (W + 15) // 16), H, W)


# ----------------------------------------------------------------------------------------------------------- generic decoder
class Bits:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
            assert z < 32
        return (1 << z) - 1 + self.u(z)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)

    def aligned(self):
        return self.p & 7 == 0

    def more_rbsp_data(self):
        """7.2: data remains unless only the rbsp_trailing_bits (a 1 then zeros to the end) follow."""
        last = len(self.d) - 1
        while self.d[last] == 0:
            last -= 1
        stop = last * 8 + (7 - (self.d[last] & -self.d[last]).bit_length() + 1)       # position of the final 1 bit
        return self.p < stop


def rbsp(nal: bytes) -> bytes:
    out, zeros = bytearray(), 0
    for b in nal:
        if zeros >= 2 and b == 3:
            zeros = 0
            continue
        assert not (zeros >= 2 and b < 3), "start-code emulation inside a NAL unit"
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def parse_sps(nal: bytes) -> dict:
    assert nal[0] == 0x67
    r = Bits(rbsp(nal[1:]))
    s = dict(profile_idc=r.u(8), constraints=r.u(8), level_idc=r.u(8), sps_id=r.ue())
    assert s["profile_idc"] == 66, "only the Baseline syntax branch (no chroma_format_idc etc.) is restated"
    s["log2_max_frame_num"] = r.ue() + 4
    s["poc_type"] = r.ue()
    assert s["poc_type"] == 2
    s["max_num_ref_frames"], s["gaps"] = r.ue(), r.u(1)
    s["mbw"], s["mbh"] = r.ue() + 1, r.ue() + 1
    s["frame_mbs_only"] = r.u(1)
    assert s["frame_mbs_only"] == 1
    s["direct_8x8"] = r.u(1)
    s["crop"] = [r.ue(), r.ue(), r.ue(), r.ue()] if r.u(1) else [0, 0, 0, 0]
    if r.u(1):                                                          # vui_parameters()
        if r.u(1):                                                      # aspect_ratio_info
            if r.u(8) == 255:
                r.u(32)
        if r.u(1):
            r.u(1)                                                      # overscan
        if r.u(1):                                                      # video_signal_type
            s["video_format"], s["full_range"] = r.u(3), r.u(1)
            if r.u(1):
                s["colour"] = (r.u(8), r.u(8), r.u(8))
        if r.u(1):
            r.ue(); r.ue()                                              # chroma_loc
        if r.u(1):                                                      # timing_info
            s["num_units_in_tick"], s["time_scale"], s["fixed_frame_rate"] = r.u(32), r.u(32), r.u(1)
        assert r.u(1) == 0 and r.u(1) == 0, "HRD parameters are not restated"
        r.u(1)                                                          # pic_struct_present_flag
        if r.u(1):                                                      # bitstream_restriction
            r.u(1)
            s["restriction"] = [r.ue() for _ in range(6)]
    assert r.u(1) == 1 and not r.more_rbsp_data() and all(r.u(1) == 0 for _ in range(len(r.d) * 8 - r.p)), "rbsp_trailing_bits"
    s["W"], s["H"] = s["mbw"] * 16 - 2 * (s["crop"][0] + s["crop"][1]), s["mbh"] * 16 - 2 * (s["crop"][2] + s["crop"][3])
    return s


def parse_pps(nal: bytes) -> dict:
    assert nal[0] == 0x68
    r = Bits(rbsp(nal[1:]))
    p = dict(pps_id=r.ue(), sps_id=r.ue(), cabac=r.u(1), field_poc=r.u(1), slice_groups=r.ue() + 1)
    assert p["cabac"] == 0 and p["slice_groups"] == 1
    p["ref_l0"], p["ref_l1"], p["weighted_pred"], p["weighted_bipred"] = r.ue() + 1, r.ue() + 1, r.u(1), r.u(2)
    p["qp"], p["qs"], p["chroma_qp_offset"] = 26 + r.se(), 26 + r.se(), r.se()
    p["deblocking_control"], p["constrained_intra"], p["redundant_pic_cnt"] = r.u(1), r.u(1), r.u(1)
    assert r.u(1) == 1 and not r.more_rbsp_data()
    return p


def decode_slice(nal: bytes, sps: dict, pps: dict):
    """One IDR I slice covering the picture -> (header dict, uint8 [n_mb, 384] samples)."""
    assert nal[0] >> 7 == 0
    ref_idc, typ = (nal[0] >> 5) & 3, nal[0] & 31
    assert typ == 5 and ref_idc != 0, "every picture is an IDR picture"
    r = Bits(rbsp(nal[1:]))
    h = dict(first_mb=r.ue(), slice_type=r.ue(), pps_id=r.ue(), frame_num=r.u(sps["log2_max_frame_num"]), idr_pic_id=r.ue())
    assert h["first_mb"] == 0 and h["slice_type"] % 5 == 2 and h["frame_num"] == 0
    assert pps["redundant_pic_cnt"] == 0
    h["no_output_of_prior_pics"], h["long_term_reference"] = r.u(1), r.u(1)           # dec_ref_pic_marking of an IDR picture
    h["qp"] = pps["qp"] + r.se()
    if pps["deblocking_control"]:
        h["disable_deblocking"] = r.ue()
        if h["disable_deblocking"] != 1:
            r.se(); r.se()
    n_mb = sps["mbw"] * sps["mbh"]
    out = np.empty((n_mb, 384), np.uint8)
    for mb in range(n_mb):                                                           # slice_data(): no skip runs in an I slice
        assert r.more_rbsp_data()
        assert r.ue() == 25, "macroblock is not I_PCM"
        while not r.aligned():
            assert r.u(1) == 0, "pcm_alignment_zero_bit"
        at = r.p >> 3
        out[mb] = np.frombuffer(r.d[at:at + 384], np.uint8)
        assert out[mb].min() > 0, "pcm samples shall not be 0"
        r.p += 384 * 8
    assert not r.more_rbsp_data() and r.u(1) == 1
    return h, out


def walk(buf, lo, hi):
    while lo < hi:
        size, kind = struct.unpack(">I4s", buf[lo:lo + 8])
        head = 8
        if size == 1:
            size, head = struct.unpack(">Q", buf[lo + 8:lo + 16])[0], 16
        assert size >= head and lo + size <= hi, (kind, size)
        yield kind, lo + head, lo + size
        lo += size
    assert lo == hi


def tree(buf, lo, hi, path=()):
    """{(b'moov', b'trak', ...): (lo, hi)} of the container boxes and their children."""
    out = {}
    for kind, a, b in walk(buf, lo, hi):
        out[path + (kind,)] = (a, b)
        if kind in (b"moov", b"trak", b"mdia", b"minf", b"dinf", b"stbl"):
            out.update(tree(buf, a, b, path + (kind,)))
    return out


def decode_file(path):
    """-> dict(frames uint8 RGB [F, H, W, 3], fps, sps, pps, headers, boxes)."""
    buf = open(path, "rb").read()
    t = tree(buf, 0, len(buf))
    assert list(k for k in t if len(k) == 1)[0] == (b"ftyp",)
    stbl = (b"moov", b"trak", b"mdia", b"minf", b"stbl")
    for need in [(b"moov", b"mvhd"), (b"moov", b"trak", b"tkhd"), (b"moov", b"trak", b"mdia", b"mdhd"), (b"moov", b"trak", b"mdia", b"hdlr"),
                 (b"moov", b"trak", b"mdia", b"minf", b"vmhd"), (b"moov", b"trak", b"mdia", b"minf", b"dinf", b"dref"),
                 stbl + (b"stsd",), stbl + (b"stts",), stbl + (b"stsc",), stbl + (b"stsz",), (b"mdat",)]:
        assert need in t, need
    a, b = t[(b"moov", b"trak", b"mdia", b"hdlr")]
    assert buf[a + 8:a + 12] == b"vide"
    a, b = t[(b"moov", b"trak", b"mdia", b"mdhd")]
    timescale, duration = struct.unpack(">II", buf[a + 12:a + 20])
    a, b = t[stbl + (b"stsd",)]
    assert struct.unpack(">I", buf[a + 4:a + 8])[0] == 1
    (esize, ekind), e = struct.unpack(">I4s", buf[a + 8:a + 16]), a + 16
    assert ekind == b"avc1" and esize == b - a - 8
    width, height = struct.unpack(">HH", buf[e + 24:e + 28])
    (csize, ckind), c = struct.unpack(">I4s", buf[e + 78:e + 86]), e + 86
    assert ckind == b"avcC" and e + 78 + csize == b
    assert buf[c] == 1 and buf[c + 4] & 3 == 3 and buf[c + 5] & 31 == 1
    n = struct.unpack(">H", buf[c + 6:c + 8])[0]
    sps_nal = buf[c + 8:c + 8 + n]
    assert buf[c + 8 + n] == 1
    m = struct.unpack(">H", buf[c + 9 + n:c + 11 + n])[0]
    pps_nal = buf[c + 11 + n:c + 11 + n + m]
    assert c + 11 + n + m == b and tuple(buf[c + 1:c + 4]) == tuple(sps_nal[1:4])
    sps, pps = parse_sps(sps_nal), parse_pps(pps_nal)
    assert (sps["W"], sps["H"]) == (width, height)
    a, b = t[stbl + (b"stts",)]
    entries, count, delta = struct.unpack(">III", buf[a + 4:a + 16])
    assert entries == 1 and count * delta == duration
    a, b = t[stbl + (b"stsz",)]
    fixed, count2 = struct.unpack(">II", buf[a + 4:a + 12])
    assert fixed == 0 and count2 == count and b - a == 12 + 4 * count
    sizes = struct.unpack(">%dI" % count, buf[a + 12:b])
    a, b = t[stbl + (b"stsc",)]
    assert struct.unpack(">IIII", buf[a + 4:a + 20]) == (1, 1, count, 1)
    key = stbl + (b"stco",) if stbl + (b"stco",) in t else stbl + (b"co64",)
    a, b = t[key]
    assert struct.unpack(">I", buf[a + 4:a + 8])[0] == 1
    off = struct.unpack(">I" if key[-1] == b"stco" else ">Q", buf[a + 8:b])[0]
    lo, hi = t[(b"mdat",)]
    assert off == lo and sum(sizes) == hi - lo
    headers, samples = [], []
    for size in sizes:
        n = struct.unpack(">I", buf[off:off + 4])[0]
        assert n + 4 == size, "one NAL unit per sample"
        h, s = decode_slice(buf[off + 4:off + size], sps, pps)
        headers.append(h)
        samples.append(s)
        off += size
    for h0, h1 in zip(headers, headers[1:]):
        assert h0["idr_pic_id"] != h1["idr_pic_id"]                                   # 7.4.3: consecutive IDR pictures differ
    S = np.stack(samples)
    payload = np.concatenate([np.zeros(S.shape[:2] + (2,), np.uint8), S], -1)
    frames = yuv_to_rgb(*planes(payload, sps["mbh"], sps["mbw"]), sps["H"], sps["W"])
    fps = timescale / delta
    if "time_scale" in sps:
        assert abs(sps["time_scale"] / (2 * sps["num_units_in_tick"]) - fps) < 1e-6
    return dict(frames=frames, fps=fps, sps=sps, pps=pps, headers=headers, boxes=t)
