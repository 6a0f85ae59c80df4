// Sample planes of the CLI's .mp4 files (the mp4 writer / reader next to the sampler).  The image has no codec
// library, so the package writes H.264 itself in the one form that needs no entropy coder: every macroblock I_PCM (ITU-T H.264
// 7.3.5: mb_type 25 in an I slice, then 256 luma + 2 x 64 chroma samples as raw bytes) -- lossless in YCbCr 4:2:0, playable by any
// H.264 decoder.  These two kernels are the per-pixel half: RGB <-> BT.601 limited-range YCbCr 4:2:0 (the integer forms below, the
// conversion ffmpeg applies for rgb24 <-> yuv420p), laid out directly as the macroblock layer of the slice data; the headers and the
// MP4 container are host code (versecrafter_amd/utils/mp4_pcm.py).  HBM-bound byte work, one pass per video.
//
//   vc_op_h264_pcm_pack     uint8 RGB [F][H][W][3] -> [F][mbh*mbw][386] = {0x0D, 0x00 (= ue(25) + alignment, byte-aligned), Y[16][16],
//                           Cb[8][8], Cr[8][8]}; pixels past H / W replicate the edge (the SPS crops them)
//   vc_op_h264_pcm_unpack   the inverse (chroma by nearest sample), cropped to H x W
#include <stdint.h>

#include "../../include/vcengine.h"
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

constexpr int MB_BYTES = 386;

VC_DEVICE uint8_t clip_sample(int v) { return (uint8_t)(v < 1 ? 1 : (v > 255 ? 255 : v)); }      // pcm samples shall not be 0 (7.4.5)
VC_DEVICE uint8_t clip_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// one workgroup of 256 threads = one macroblock, thread = one luma sample
__global__ __launch_bounds__(256) void h264_pcm_pack_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out, int H, int W, int mbw,
                                                            int mbh) {
    const int mb = blockIdx.x, f = blockIdx.y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = min((mb % mbw) * 16 + tx, W - 1), y = min((mb / mbw) * 16 + ty, H - 1);
    const uint8_t* px = rgb + (((int64_t)f * H + y) * W + x) * 3;
    uint8_t* o = out + ((int64_t)f * mbw * mbh + mb) * MB_BYTES;
    const int r = px[0], g = px[1], b = px[2];
    if (threadIdx.x == 0) { o[0] = 0x0D; o[1] = 0x00; }
    o[2 + threadIdx.x] = clip_sample(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16);
    // chroma: the 2 x 2 block mean of RGB (lanes tx, tx+1, and the row below = +16 lanes), converted by the lane at the even corner
    int sr = r + __shfl_down(r, 1), sg = g + __shfl_down(g, 1), sb = b + __shfl_down(b, 1);
    sr += __shfl_down(sr, 16); sg += __shfl_down(sg, 16); sb += __shfl_down(sb, 16);
    if (((tx | ty) & 1) == 0) {
        const int ar = (sr + 2) >> 2, ag = (sg + 2) >> 2, ab = (sb + 2) >> 2;
        const int c = (ty >> 1) * 8 + (tx >> 1);
        o[2 + 256 + c] = clip_sample(((-38 * ar - 74 * ag + 112 * ab + 128) >> 8) + 128);
        o[2 + 320 + c] = clip_sample(((112 * ar - 94 * ag - 18 * ab + 128) >> 8) + 128);
    }
}

__global__ __launch_bounds__(256) void h264_pcm_unpack_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ rgb, int H, int W, int mbw,
                                                              int mbh) {
    const int mb = blockIdx.x, f = blockIdx.y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = (mb % mbw) * 16 + tx, y = (mb / mbw) * 16 + ty;
    if (x >= W || y >= H) return;
    const uint8_t* o = in + ((int64_t)f * mbw * mbh + mb) * MB_BYTES;
    const int c = (ty >> 1) * 8 + (tx >> 1);
    const int C = 298 * ((int)o[2 + threadIdx.x] - 16), D = (int)o[2 + 256 + c] - 128, E = (int)o[2 + 320 + c] - 128;
    uint8_t* px = rgb + (((int64_t)f * H + y) * W + x) * 3;
    px[0] = clip_u8((C + 409 * E + 128) >> 8);
    px[1] = clip_u8((C - 100 * D - 208 * E + 128) >> 8);
    px[2] = clip_u8((C + 516 * D + 128) >> 8);
}

thread_local char g_pcm_err[160] = "";
int pfail(int code, const char* msg) {
    snprintf(g_pcm_err, sizeof g_pcm_err, "%s", msg);
    return code;
}
int pdone(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VC_OK;
    snprintf(g_pcm_err, sizeof g_pcm_err, "%s: %s", what, hipGetErrorString(e));
    return VC_E_HIP;
}
bool pcm_shape_ok(int frames, int H, int W) { return frames > 0 && H > 0 && W > 0 && frames <= 65535 && (int64_t)((H + 15) / 16) * ((W + 15) / 16) < (1ll << 31); }

}  // namespace

extern "C" {

const char* vc_h264_pcm_last_error(void) { return g_pcm_err; }

int64_t vc_op_h264_pcm_bytes(int frames, int H, int W) { return (int64_t)frames * ((H + 15) / 16) * ((W + 15) / 16) * MB_BYTES; }

int vc_op_h264_pcm_pack(const void* rgb, void* out, int frames, int H, int W, void* stream) {
    if (!rgb || !out || !pcm_shape_ok(frames, H, W)) return pfail(VC_E_INVALID, "vc_op_h264_pcm_pack: bad argument");
    const int mbw = (W + 15) / 16, mbh = (H + 15) / 16;
    hipLaunchKernelGGL(h264_pcm_pack_kernel, dim3(mbw * mbh, frames), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)rgb, (uint8_t*)out, H, W,
                       mbw, mbh);
    return pdone("h264_pcm_pack_kernel");
}

int vc_op_h264_pcm_unpack(const void* in, void* rgb, int frames, int H, int W, void* stream) {
    if (!in || !rgb || !pcm_shape_ok(frames, H, W)) return pfail(VC_E_INVALID, "vc_op_h264_pcm_unpack: bad argument");
    const int mbw = (W + 15) / 16, mbh = (H + 15) / 16;
    hipLaunchKernelGGL(h264_pcm_unpack_kernel, dim3(mbw * mbh, frames), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)in, (uint8_t*)rgb, H, W,
                       mbw, mbh);
    return pdone("h264_pcm_unpack_kernel");
}

}  // extern "C"
