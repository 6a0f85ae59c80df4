"""A self-contained .mp4 writer / reader for the CLIs (stands in for the imageio / ffmpeg calls behind the reference's
`save_videos_grid` and `save_video_from_frames`: inference/versecrafter_inference.py:456, inference/rendering_4D_control_maps.py:455-485).

The image has no codec library.  An H.264 stream needs none when every macroblock is coded I_PCM (ITU-T H.264 7.3.5 / 7.4.5: raw
8-bit samples, no prediction, no transform, no entropy coding): Constrained Baseline, one IDR picture of one slice per frame, CAVLC
syntax (of which only `mb_type = ue(25)` remains), deblocking off.  The result is lossless in YCbCr 4:2:0 -- a mask of 0 / 255 comes
back exactly, colours within the rounding of the BT.601 integer conversion -- and any H.264 decoder plays it; the price is size
(1.5 bytes per pixel and frame).

Split: the per-pixel half (RGB <-> YCbCr 4:2:0 in macroblock order, i.e. the slice data itself) runs on the GPU behind the C ABI
(csrc/h264pcm.hip: vc_op_h264_pcm_pack / _unpack); parameter sets, slice headers and the ISO base-media boxes are a few hundred bytes
of host code below.  `read_mp4` reads what `write_mp4` wrote; for any other H.264 stream (the reference's demo clips are x264 High
profile, CABAC, B-frames) it raises `UnsupportedVideo` naming what it found -- decoding those needs a real decoder."""
import ctypes as C
import struct

import numpy as np
import torch

from .. import _lib

MB_BYTES = 386                     # 0x0D 0x00 (mb_type = ue(25) and its alignment bits) + 256 + 64 + 64 samples
PROFILE_BASELINE, LEVEL = 66, 51
WRITER_TAG = b"versecrafter_amd I_PCM"


class UnsupportedVideo(RuntimeError):
    pass


# ----------------------------------------------------------------------------------------------------------- bit syntax
class BitWriter:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))
        return self

    def ue(self, v):
        v += 1
        n = v.bit_length()
        return self.u(n - 1, 0).u(n, v)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def align_zero(self):
        while len(self.bits) % 8:
            self.bits.append(0)
        return self

    def trailing(self):                                  # rbsp_trailing_bits
        self.bits.append(1)
        return self.align_zero()

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


class BitReader:
    def __init__(self, data):
        self.bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))
        self.pos = 0

    def u(self, n):
        v = 0
        for b in self.bits[self.pos:self.pos + n]:
            v = (v << 1) | int(b)
        if self.pos + n > len(self.bits):
            raise UnsupportedVideo("truncated H.264 header")
        self.pos += n
        return v

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
            if z > 32:
                raise UnsupportedVideo("malformed exp-Golomb code")
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


def escape(rbsp: bytes) -> bytes:
    """emulation prevention (7.4.1): 00 00 0x with x <= 3 becomes 00 00 03 0x."""
    out, zeros = bytearray(), 0
    for b in rbsp:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def unescape(nal: bytes) -> bytes:
    out, zeros = bytearray(), 0
    for b in nal:
        if zeros >= 2 and b == 3:
            zeros = 0
            continue
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def sps_nal(H, W, fps):
    mbw, mbh = (W + 15) // 16, (H + 15) // 16
    w = BitWriter()
    w.u(8, PROFILE_BASELINE).u(8, 0xC0).u(8, LEVEL)      # constraint_set0 + set1: Constrained Baseline
    w.ue(0)                                              # seq_parameter_set_id
    w.ue(0)                                              # log2_max_frame_num_minus4
    w.ue(2)                                              # pic_order_cnt_type 2: output order = decoding order
    w.ue(1).u(1, 0)                                      # max_num_ref_frames, gaps_in_frame_num_value_allowed_flag
    w.ue(mbw - 1).ue(mbh - 1)
    w.u(1, 1).u(1, 1)                                    # frame_mbs_only_flag, direct_8x8_inference_flag
    cr, cb = mbw * 16 - W, mbh * 16 - H
    if cr or cb:
        if (cr | cb) & 1:
            raise ValueError("write_mp4: 4:2:0 video needs even width and height")
        w.u(1, 1).ue(0).ue(cr // 2).ue(0).ue(cb // 2)    # frame cropping in chroma sample units
    else:
        w.u(1, 0)
    w.u(1, 1)                                            # vui_parameters_present_flag
    w.u(1, 0).u(1, 0)                                    # aspect_ratio_info_present_flag, overscan_info_present_flag
    w.u(1, 1).u(3, 5).u(1, 0).u(1, 1).u(8, 6).u(8, 6).u(8, 6)   # video_signal_type: limited range, SMPTE 170M (BT.601) primaries / transfer / matrix
    w.u(1, 0)                                            # chroma_loc_info_present_flag
    w.u(1, 1).u(32, 1000).u(32, int(round(fps * 2000))).u(1, 1)  # timing_info: a field lasts num_units_in_tick / time_scale
    w.u(1, 0).u(1, 0).u(1, 0)                            # nal_hrd, vcl_hrd, pic_struct_present_flag
    w.u(1, 1).u(1, 1).ue(0).ue(0).ue(16).ue(16).ue(0).ue(1)      # bitstream_restriction: no reordering, one frame buffered
    return b"\x67" + escape(w.trailing().bytes())


def pps_nal():
    w = BitWriter()
    w.ue(0).ue(0).u(1, 0).u(1, 0).ue(0)                  # ids, entropy_coding_mode_flag = 0 (CAVLC), no field POC, one slice group
    w.ue(0).ue(0).u(1, 0).u(2, 0)                        # default ref idx counts, no weighted prediction
    w.se(0).se(0).se(0)                                  # pic_init_qp / qs, chroma_qp_index_offset
    w.u(1, 1).u(1, 0).u(1, 0)                            # deblocking_filter_control_present_flag, constrained_intra_pred, redundant_pic_cnt
    return b"\x68" + escape(w.trailing().bytes())


def slice_prefix(frame_index):
    """NAL header + slice header of an IDR I slice + the first macroblock's mb_type, padded to the byte boundary where its samples start."""
    w = BitWriter()
    w.ue(0).ue(7).ue(0)                                  # first_mb_in_slice, slice_type 7 (I, all slices of the picture), pic_parameter_set_id
    w.u(4, 0)                                            # frame_num
    w.ue(frame_index & 1)                                # idr_pic_id: differs between consecutive IDR pictures
    w.u(1, 0).u(1, 0)                                    # dec_ref_pic_marking: no_output_of_prior_pics_flag, long_term_reference_flag
    w.se(0)                                              # slice_qp_delta
    w.ue(1)                                              # disable_deblocking_filter_idc = 1
    w.ue(25).align_zero()                                # mb_type I_PCM, pcm_alignment_zero_bit
    return b"\x65" + escape(w.bytes())


# ----------------------------------------------------------------------------------------------------------- boxes
def box(kind, *payload):
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), kind) + body


def full(kind, version, flags, *payload):
    return box(kind, struct.pack(">I", (version << 24) | flags), *payload)


MATRIX = struct.pack(">9i", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def moov_box(H, W, fps, sizes, chunk_offset, sps, pps):
    n = len(sizes)
    scale, delta = int(round(fps * 1000)), 1000
    dur = n * delta
    avcc = box(b"avcC", bytes([1, sps[1], sps[2], sps[3], 0xFF, 0xE1]), struct.pack(">H", len(sps)), sps, b"\x01", struct.pack(">H", len(pps)), pps)
    avc1 = box(b"avc1", b"\0" * 6, struct.pack(">H", 1), b"\0" * 16, struct.pack(">HHIIIH", W, H, 0x480000, 0x480000, 0, 1),
               bytes([len(WRITER_TAG)]) + WRITER_TAG.ljust(31, b"\0"), struct.pack(">Hh", 0x18, -1), avcc)
    big = chunk_offset >= (1 << 32)
    stbl = box(b"stbl",
               full(b"stsd", 0, 0, struct.pack(">I", 1), avc1),
               full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
               full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
               full(b"stsz", 0, 0, struct.pack(">II", 0, n), np.asarray(sizes, dtype=">u4").tobytes()),
               full(b"co64", 0, 0, struct.pack(">IQ", 1, chunk_offset)) if big else full(b"stco", 0, 0, struct.pack(">II", 1, chunk_offset)))
    minf = box(b"minf", full(b"vmhd", 0, 1, b"\0" * 8), box(b"dinf", full(b"dref", 0, 0, struct.pack(">I", 1), full(b"url ", 0, 1))), stbl)
    mdia = box(b"mdia", full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, scale, dur, 0x55C4, 0)),
               full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide"), b"\0" * 12, b"VideoHandler\0"), minf)
    tkhd = full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur), b"\0" * 8, struct.pack(">hhhH", 0, 0, 0, 0), MATRIX,
                struct.pack(">II", W << 16, H << 16))
    mvhd = full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, scale, dur, 0x10000, 0x100), b"\0" * 10, MATRIX, b"\0" * 24, struct.pack(">I", 2))
    return box(b"moov", mvhd, box(b"trak", tkhd, mdia))


# ----------------------------------------------------------------------------------------------------------- device half
def _check(rc):
    if rc != 0:
        msg = _lib.load().vc_h264_pcm_last_error()
        raise _lib.VcError(rc, msg.decode() if msg else "")


def _need_cuda_u8(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.uint8:
        raise RuntimeError(f"{name}: expected a uint8 tensor on the GPU (the sample planes are packed by the HIP engine; no CPU path)")
    return t.contiguous()


def pack_frames(frames: torch.Tensor) -> torch.Tensor:
    """uint8 RGB [F, H, W, 3] on the GPU -> the macroblock layer of every frame, uint8 [F, mbh * mbw, 386]."""
    frames = _need_cuda_u8(frames, "pack_frames")
    F, H, W, ch = frames.shape
    if ch != 3:
        raise ValueError("pack_frames: expected [F, H, W, 3]")
    L = _lib.load()
    out = torch.empty(F, ((H + 15) // 16) * ((W + 15) // 16), MB_BYTES, dtype=torch.uint8, device=frames.device)
    assert out.numel() == L.vc_op_h264_pcm_bytes(F, H, W)
    _check(L.vc_op_h264_pcm_pack(C.c_void_p(frames.data_ptr()), C.c_void_p(out.data_ptr()), F, H, W,
                                 C.c_void_p(torch.cuda.current_stream(frames.device).cuda_stream)))
    return out


def unpack_frames(payload: torch.Tensor, H: int, W: int) -> torch.Tensor:
    payload = _need_cuda_u8(payload, "unpack_frames")
    F = payload.shape[0]
    if tuple(payload.shape[1:]) != (((H + 15) // 16) * ((W + 15) // 16), MB_BYTES):
        raise ValueError("unpack_frames: payload does not match the picture size")
    out = torch.empty(F, H, W, 3, dtype=torch.uint8, device=payload.device)
    _check(_lib.load().vc_op_h264_pcm_unpack(C.c_void_p(payload.data_ptr()), C.c_void_p(out.data_ptr()), F, H, W,
                                             C.c_void_p(torch.cuda.current_stream(payload.device).cuda_stream)))
    return out


# ----------------------------------------------------------------------------------------------------------- files
def mux(path, payload: np.ndarray, H, W, fps):
    """payload: uint8 [F, n_mb, 386] (host) -> the .mp4 file.  One sample per frame = 4-byte length + one IDR slice NAL."""
    F, n_mb, _ = payload.shape
    sps, pps = sps_nal(H, W, fps), pps_nal()
    ftyp = box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2avc1mp41")
    body = n_mb * MB_BYTES - 2 + 1                       # the first macroblock's two header bytes live in the prefix; + rbsp trailing byte
    prefixes = [slice_prefix(f) for f in range(F)]
    sizes = [4 + len(p) + body for p in prefixes]
    total = sum(sizes)
    big = 8 + total >= (1 << 32)
    with open(path, "wb") as fh:
        fh.write(ftyp)
        fh.write(struct.pack(">I4sQ", 1, b"mdat", 16 + total) if big else struct.pack(">I4s", 8 + total, b"mdat"))
        chunk_offset = fh.tell()
        for f in range(F):
            fh.write(struct.pack(">I", sizes[f] - 4))
            fh.write(prefixes[f])
            fh.write(payload[f].reshape(-1)[2:].tobytes())
            fh.write(b"\x80")                            # rbsp_slice_trailing_bits
        fh.write(moov_box(H, W, fps, sizes, chunk_offset, sps, pps))
    return path


def write_mp4(path, frames: torch.Tensor, fps=16):
    """frames: uint8 RGB [F, H, W, 3] (or [F, H, W]) on the GPU."""
    if frames.dim() == 3:
        frames = frames[..., None].expand(-1, -1, -1, 3)
    F, H, W, _ = frames.shape
    if (H | W) & 1:
        raise ValueError("write_mp4: 4:2:0 video needs even width and height")
    return mux(path, pack_frames(frames.contiguous()).cpu().numpy(), H, W, fps)


def _walk(buf, lo, hi):
    while lo + 8 <= hi:
        size, kind = struct.unpack(">I4s", buf[lo:lo + 8])
        head = 8
        if size == 1:
            size, head = struct.unpack(">Q", buf[lo + 8:lo + 16])[0], 16
        elif size == 0:
            size = hi - lo
        if size < head or lo + size > hi:
            raise UnsupportedVideo("malformed MP4 box structure")
        yield kind, lo + head, lo + size
        lo += size


def _find(buf, lo, hi, *kinds):
    for kind in kinds:
        for k, a, b in _walk(buf, lo, hi):
            if k == kind:
                lo, hi = a, b
                break
        else:
            return None
    return lo, hi


def parse_sps(nal: bytes):
    """-> dict(profile, constraint, level, mbw, mbh, H, W, poc_type, vui flag); raises for what the PCM reader cannot take."""
    r = BitReader(unescape(nal[1:]))
    profile, constraint, level = r.u(8), r.u(8), r.u(8)
    if profile != PROFILE_BASELINE:
        names = {66: "Baseline", 77: "Main", 88: "Extended", 100: "High", 110: "High 10", 122: "High 4:2:2", 244: "High 4:4:4"}
        raise UnsupportedVideo(f"H.264 {names.get(profile, 'profile %d' % profile)} profile stream: only the I_PCM streams this package "
                               f"writes are readable here (the image has no video decoder); provide a frame dump instead")
    r.ue(); r.ue()
    poc = r.ue()
    if poc == 0:
        r.ue()
    elif poc == 1:
        raise UnsupportedVideo("H.264 stream with pic_order_cnt_type 1")
    r.ue(); r.u(1)
    mbw, mbh = r.ue() + 1, r.ue() + 1
    if r.u(1) != 1:
        raise UnsupportedVideo("interlaced H.264 stream")
    r.u(1)
    crop = [0, 0, 0, 0]
    if r.u(1):
        crop = [r.ue(), r.ue(), r.ue(), r.ue()]
    return dict(profile=profile, constraint=constraint, level=level, mbw=mbw, mbh=mbh, W=mbw * 16 - 2 * (crop[0] + crop[1]),
                H=mbh * 16 - 2 * (crop[2] + crop[3]), poc_type=poc, crop=crop)


def demux(path, max_frames=None):
    """-> (payload uint8 [F, n_mb, 386] on the host, H, W, fps).  Checks every macroblock header of every slice."""
    buf = np.fromfile(path, dtype=np.uint8)
    view = memoryview(buf)

    def b(lo, hi):
        return bytes(view[lo:hi])
    top = dict((k, (a, e)) for k, a, e in _walk(view, 0, len(buf)))
    if b"moov" not in top:
        raise UnsupportedVideo(f"{path}: no moov box")
    stbl = _find(view, *top[b"moov"], b"trak", b"mdia", b"minf", b"stbl")
    mdhd = _find(view, *top[b"moov"], b"trak", b"mdia", b"mdhd")
    if stbl is None or mdhd is None:
        raise UnsupportedVideo(f"{path}: no video sample table")
    tables = dict((k, (a, e)) for k, a, e in _walk(view, *stbl))
    stsd = b(*tables[b"stsd"])
    at = stsd.find(b"avcC")
    if at < 0:
        raise UnsupportedVideo(f"{path}: not an H.264 (avc1) track")
    cfg = stsd[at + 4:]
    nlen = (cfg[4] & 3) + 1
    sps_len = struct.unpack(">H", cfg[6:8])[0]
    sps = cfg[8:8 + sps_len]
    info = parse_sps(sps)
    pps_at = 8 + sps_len
    pps_len = struct.unpack(">H", cfg[pps_at + 1:pps_at + 3])[0]
    pps = BitReader(unescape(cfg[pps_at + 3:pps_at + 3 + pps_len][1:]))
    pps.ue(); pps.ue()
    if pps.u(1) != 0:
        raise UnsupportedVideo("CABAC-coded H.264 stream: only the I_PCM streams this package writes are readable here")
    scale = struct.unpack(">I", b(mdhd[0] + 12, mdhd[0] + 16))[0]
    stts = b(*tables[b"stts"])
    delta = struct.unpack(">I", stts[12:16])[0] if len(stts) >= 16 else 0
    fps = scale / delta if delta else 0.0
    stsz = b(*tables[b"stsz"])
    fixed, n = struct.unpack(">II", stsz[4:12])
    sizes = np.full(n, fixed, dtype=np.int64) if fixed else np.frombuffer(stsz[12:12 + 4 * n], dtype=">u4").astype(np.int64)
    if b"co64" in tables:
        co = b(*tables[b"co64"])
        chunks = np.frombuffer(co[8:8 + 8 * struct.unpack(">I", co[4:8])[0]], dtype=">u8").astype(np.int64)
    else:
        co = b(*tables[b"stco"])
        chunks = np.frombuffer(co[8:8 + 4 * struct.unpack(">I", co[4:8])[0]], dtype=">u4").astype(np.int64)
    stsc = b(*tables[b"stsc"])
    runs = np.frombuffer(stsc[8:8 + 12 * struct.unpack(">I", stsc[4:8])[0]], dtype=">u4").reshape(-1, 3).astype(np.int64)
    offsets, s = [], 0
    for ci, base in enumerate(chunks, start=1):          # samples of a chunk are contiguous
        per = int(runs[np.searchsorted(runs[:, 0], ci, side="right") - 1, 1])
        off = int(base)
        for _ in range(per):
            if s >= n:
                break
            offsets.append(off)
            off += int(sizes[s])
            s += 1
    n = len(offsets) if max_frames is None else min(len(offsets), max_frames)
    n_mb = info["mbw"] * info["mbh"]
    body = n_mb * MB_BYTES - 2
    payload = np.empty((n, n_mb, MB_BYTES), dtype=np.uint8)
    for f in range(n):
        lo, hi = offsets[f], offsets[f] + int(sizes[f])
        nal_len = int.from_bytes(b(lo, lo + nlen), "big")
        if nal_len + nlen != hi - lo:
            raise UnsupportedVideo("more than one NAL unit per sample: not a stream written by this package")
        if buf[lo + nlen] & 0x1F != 5:
            raise UnsupportedVideo("non-IDR picture: not a stream written by this package")
        start = hi - 1 - body                            # samples of macroblock 0 .. the trailing byte
        if start <= lo + nlen or buf[hi - 1] != 0x80:
            raise UnsupportedVideo("slice is not I_PCM-coded: not a stream written by this package")
        r = BitReader(unescape(b(lo + nlen + 1, start)))
        head = (r.ue(), r.ue(), r.ue(), r.u(4), r.ue(), r.u(2), r.se(), r.ue(), r.ue())
        if head[0] != 0 or head[1] not in (2, 7) or head[8] != 25:
            raise UnsupportedVideo("slice is not I_PCM-coded: not a stream written by this package")
        flat = payload[f].reshape(-1)
        flat[0], flat[1] = 0x0D, 0x00
        flat[2:] = buf[start:hi - 1]
        if not ((payload[f, :, 0] == 0x0D) & (payload[f, :, 1] == 0)).all():
            raise UnsupportedVideo("a macroblock is not I_PCM: not a stream written by this package")
    return payload, info["H"], info["W"], fps


def read_mp4(path, max_frames=None, device="cuda") -> torch.Tensor:
    """-> uint8 RGB [F, H, W, 3] on the GPU."""
    payload, H, W, _ = demux(path, max_frames)
    return unpack_frames(torch.from_numpy(payload).to(device), H, W)
