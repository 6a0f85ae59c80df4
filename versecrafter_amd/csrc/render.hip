// 4D control-map renderer kernels (second half): the per-pixel stages of
// /root/reference inference/rendering_4D_control_maps.py.  Once per video (81 frames), HBM-bound byte / float work.
//
//   vc_op_render_composite      composite_by_depth_batch :398-411, composite_by_depth :437-453, merge_bg_and_fg_mask :736-763
//   vc_op_render_depth_gray     visualize_depth_as_grayscale :520-537 (the per-pixel map; the range statistics stay on the host)
//   vc_op_render_gauss_density  compute_probability_density_map_gpu :801-883 (per-pixel pdf of projected Gaussians, summed)
//   vc_op_render_gauss_frame    project_3d_gaussians_to_2d :634-693 (max-normalise, threshold, far-to-near "over" compositing)
//   vc_op_render_blend          blend_gaussian_projection_with_bg :719-732
// These five are pinned by fixtures recorded from the reference's own functions (tests/golden/make_golden_render.py).
//
//   vc_op_render_points         render_point_cloud_pytorch3d_batch :243-338  (PyTorch3D PointsRasterizer + AlphaCompositor)
//   vc_op_render_mesh           render_meshes_pytorch3d_batch :150-241        (PyTorch3D MeshRasterizer + HardPhongShader)
// PyTorch3D is not in the reference tree or the image: these two restate its published algorithms (oracle/render_oracle.py is
// the specification they are tested against) -- PARITY UNPINNED.
#include <math.h>
#include <stdint.h>

#include "../../include/vcengine.h"
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

inline int rblocks(int64_t n) { return (int)((n + 255) / 256); }

// ---- depth compositing ---------------------------------------------------------------------------------------------------------
// take_fg = fg_mask & ((bg_depth <= 0) | ((fg_depth > 0) & (fg_depth < bg_depth - 1e-6)))
__global__ __launch_bounds__(256) void render_composite_kernel(const uint8_t* __restrict__ bg_rgb, const float* __restrict__ bg_d,
                                                               const uint8_t* __restrict__ fg_rgb, const float* __restrict__ fg_d,
                                                               const uint8_t* __restrict__ fg_mask, const uint8_t* __restrict__ bg_mask,
                                                               uint8_t* __restrict__ out_rgb, float* __restrict__ out_d,
                                                               uint8_t* __restrict__ out_mask, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float b = bg_d[i], f = fg_d[i];
    const bool take = fg_mask[i] != 0 && (b <= 0.f || (f > 0.f && f < b - 1e-6f));
    if (out_rgb) {
        const uint8_t* s = take ? fg_rgb + 3 * i : bg_rgb + 3 * i;
        out_rgb[3 * i] = s[0]; out_rgb[3 * i + 1] = s[1]; out_rgb[3 * i + 2] = s[2];
    }
    if (out_d) out_d[i] = take ? f : b;
    if (out_mask) {        // merge_bg_and_fg_mask: the inverted background mask, the foreground mask where it is in front
        const uint8_t m = take ? (uint8_t)255 : (bg_mask[i] ? (uint8_t)0 : (uint8_t)255);
        out_mask[3 * i] = m; out_mask[3 * i + 1] = m; out_mask[3 * i + 2] = m;
    }
}

// disparity, normalised to the clip's range when normalize != 0, closer = lighter; uint8 by truncation, three equal channels
__global__ __launch_bounds__(256) void render_depth_gray_kernel(const float* __restrict__ d, uint8_t* __restrict__ out, int64_t n,
                                                                int normalize, float min_disp, float denom) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = d[i];
    float disp = x > 0.f ? 1.0f / x : 0.f;
    if (normalize) disp = (disp - min_disp) / denom;
    disp = fminf(fmaxf(disp, 0.f), 1.f);
    const uint8_t g = (uint8_t)(disp * 255.f);
    out[3 * i] = g; out[3 * i + 1] = g; out[3 * i + 2] = g;
}

// ---- projected Gaussians ----------------------------------------------------------------------------------------------------
// one record per Gaussian (host: the reference's own 3x3 arithmetic in torch, :828-873): valid = 0 -> contributes nothing
struct GaussRec { float mx, my, i00, i01, i10, i11, coeff, valid, r, g, b, pad; };

VC_DEVICE float gauss_pdf(const GaussRec& q, float u, float v) {
    const float dx = u - q.mx, dy = v - q.my;
    const float a0 = dx * q.i00 + dy * q.i10, a1 = dx * q.i01 + dy * q.i11;      // diff @ cov_inv
    const float mahal = a0 * dx + a1 * dy;
    float p = q.coeff * expf(-0.5f * mahal);
    if (!(p == p) || fabsf(p) == INFINITY) p = 0.f;                              // nan_to_num(nan = posinf = neginf = 0)
    return p;
}

__global__ __launch_bounds__(256) void render_gauss_density_kernel(const GaussRec* __restrict__ rec, int n, float* __restrict__ out, int W,
                                                                   int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)W * H) return;
    const float u = (float)(i % W), v = (float)(i / W);
    float acc = 0.f;
    for (int k = 0; k < n; ++k)
        if (rec[k].valid != 0.f) acc += gauss_pdf(rec[k], u, v);
    out[i] = acc;
}

// maxima of the n density maps over the image (densities are >= 0: the float order is the order of their bit patterns)
__global__ __launch_bounds__(256) void render_gauss_max_kernel(const GaussRec* __restrict__ rec, int n, unsigned* __restrict__ mx, int W, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < (int64_t)W * H;
    const float u = in ? (float)(i % W) : 0.f, v = in ? (float)(i / W) : 0.f;
    for (int k = 0; k < n; ++k) {
        float p = (in && rec[k].valid != 0.f) ? gauss_pdf(rec[k], u, v) : 0.f;
        p = wave_max(p);
        if ((threadIdx.x & 63) == 0 && p > 0.f) atomicMax(mx + k, __float_as_uint(p));
    }
}

// far-to-near "over" compositing of the n normalised density maps (records are in compositing order)
__global__ __launch_bounds__(256) void render_gauss_frame_kernel(const GaussRec* __restrict__ rec, int n, const unsigned* __restrict__ mx,
                                                                 float threshold, float inv_span, uint8_t* __restrict__ rgb,
                                                                 float* __restrict__ alpha, int W, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)W * H) return;
    const float u = (float)(i % W), v = (float)(i / W);
    float r = 0.f, g = 0.f, b = 0.f, a = 0.f;
    for (int k = 0; k < n; ++k) {
        float d = rec[k].valid != 0.f ? gauss_pdf(rec[k], u, v) : 0.f;
        const float m = __uint_as_float(mx[k]);
        if (m > 0.f) d = d / (m + 1e-8f);
        float an = d > threshold ? (d - threshold) / inv_span : 0.f;               // inv_span = 1 - threshold + 1e-8 (host, as the reference)
        an = fminf(fmaxf(an, 0.f), 1.f);
        r = rec[k].r * an + r * (1.f - an);
        g = rec[k].g * an + g * (1.f - an);
        b = rec[k].b * an + b * (1.f - an);
        a = an + a * (1.f - an);
    }
    alpha[i] = fminf(fmaxf(a, 0.f), 1.f);
    rgb[3 * i] = (uint8_t)(fminf(fmaxf(r, 0.f), 1.f) * 255.f);
    rgb[3 * i + 1] = (uint8_t)(fminf(fmaxf(g, 0.f), 1.f) * 255.f);
    rgb[3 * i + 2] = (uint8_t)(fminf(fmaxf(b, 0.f), 1.f) * 255.f);
}

// C = C_fg alpha + C_bg (1 - alpha) on [0, 1] floats of uint8 inputs; scale != 0: the masked projection rgb/255 * alpha * 255 (:1322-1326)
__global__ __launch_bounds__(256) void render_blend_kernel(const uint8_t* __restrict__ fg, const float* __restrict__ alpha,
                                                           const uint8_t* __restrict__ bg, uint8_t* __restrict__ out, int64_t n, int masked) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = alpha[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float f = (float)fg[3 * i + c] / 255.0f;
        if (masked) {
            out[3 * i + c] = (uint8_t)(f * a * 255.f);
        } else {
            const float m = f * a + ((float)bg[3 * i + c] / 255.0f) * (1.f - a);
            out[3 * i + c] = (uint8_t)(fminf(fmaxf(m, 0.f), 1.f) * 255.f);
        }
    }
}

// ---- point cloud (PyTorch3D PointsRasterizer, points_per_pixel K, + AlphaCompositor) -------------------------------------------
// Cameras are the OpenCV world-to-camera [R | t] of the reference's trajectory (:1001-1009) with pixel intrinsics: the detour through
// PyTorch3D's NDC convention (:340-396) is a change of variables that cancels -- a point lands at u = fx x / z + cx, v = fy y / z + cy
// and `radius` (NDC units, the shorter image side spans [-1, 1]) is radius * min(H, W) / 2 pixels.  A point covers the pixels whose
// CENTRE (px + 0.5, py + 0.5) is closer than that; per pixel the K nearest in z are kept, composited front to back with weight
// 1 - d^2 / r^2; depth / mask come from the nearest one.
struct Cam { float R[9], t[3], fx, fy, cx, cy; };

__global__ __launch_bounds__(256) void points_project_kernel(const float* __restrict__ pts, int64_t n, Cam cam, float* __restrict__ uvz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    const float cx = cam.R[0] * x + cam.R[1] * y + cam.R[2] * z + cam.t[0];
    const float cy = cam.R[3] * x + cam.R[4] * y + cam.R[5] * z + cam.t[1];
    const float cz = cam.R[6] * x + cam.R[7] * y + cam.R[8] * z + cam.t[2];
    float u = -1e30f, v = -1e30f;
    if (cz > 1e-8f && cx == cx && cy == cy && cz == cz) { u = cam.fx * cx / cz + cam.cx; v = cam.fy * cy / cz + cam.cy; }
    uvz[3 * i] = u; uvz[3 * i + 1] = v; uvz[3 * i + 2] = cz;
}

// round k of the K-nearest selection: per pixel the smallest key (z bits, point index) that is larger than the pixel's previous key
__global__ __launch_bounds__(256) void points_select_kernel(const float* __restrict__ uvz, int64_t n, float rp, int W, int H,
                                                            const unsigned long long* __restrict__ prev, unsigned long long* __restrict__ cur) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = uvz[3 * i], v = uvz[3 * i + 1], z = uvz[3 * i + 2];
    if (!(z > 0.f) || u < -rp || v < -rp || u > W + rp || v > H + rp) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)i;
    const int x0 = max(0, (int)floorf(u - rp - 0.5f)), x1 = min(W - 1, (int)ceilf(u + rp - 0.5f));
    const int y0 = max(0, (int)floorf(v - rp - 0.5f)), y1 = min(H - 1, (int)ceilf(v + rp - 0.5f));
    const float r2 = rp * rp;
    for (int py = y0; py <= y1; ++py)
        for (int px = x0; px <= x1; ++px) {
            const float dx = px + 0.5f - u, dy = py + 0.5f - v;
            if (dx * dx + dy * dy < r2) {
                const int64_t pix = (int64_t)py * W + px;
                if (!prev || key > prev[pix]) atomicMin(cur + pix, key);
            }
        }
}

__global__ __launch_bounds__(256) void points_composite_kernel(const float* __restrict__ uvz, const uint8_t* __restrict__ colors,
                                                               const unsigned long long* __restrict__ keys /*[K][H*W]*/, int K, float rp, int W,
                                                               int H, float bg, uint8_t* __restrict__ rgb, float* __restrict__ depth,
                                                               uint8_t* __restrict__ mask) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hw = (int64_t)W * H;
    if (pix >= hw) return;
    const float pxc = (float)(pix % W) + 0.5f, pyc = (float)(pix / W) + 0.5f;
    float acc[3] = {0.f, 0.f, 0.f}, cum = 1.f;
    const float r2 = rp * rp;
    bool any = false;
    float z0 = 0.f;
    for (int k = 0; k < K; ++k) {
        const unsigned long long key = keys[(int64_t)k * hw + pix];
        if (key == ~0ull) break;
        const unsigned idx = (unsigned)key;
        if (!any) { any = true; z0 = __uint_as_float((unsigned)(key >> 32)); }
        const float dx = pxc - uvz[3 * (int64_t)idx], dy = pyc - uvz[3 * (int64_t)idx + 1];
        const float w = 1.f - (dx * dx + dy * dy) / r2;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += cum * w * ((float)colors[3 * (int64_t)idx + c] / 255.0f);
        cum *= 1.f - w;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = any ? acc[c] : bg;
        rgb[3 * pix + c] = (uint8_t)fminf(fmaxf(v * 255.f, 0.f), 255.f);
    }
    depth[pix] = any ? z0 : 0.f;
    mask[pix] = any ? 1 : 0;
}

// ---- meshes (PyTorch3D MeshRasterizer: blur 0, one face per pixel, perspective-correct; HardPhongShader, point light) --------
__global__ __launch_bounds__(256) void mesh_project_kernel(const float* __restrict__ verts, int nv, Cam cam, float* __restrict__ uvz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float x = verts[3 * i], y = verts[3 * i + 1], z = verts[3 * i + 2];
    const float cx = cam.R[0] * x + cam.R[1] * y + cam.R[2] * z + cam.t[0];
    const float cy = cam.R[3] * x + cam.R[4] * y + cam.R[5] * z + cam.t[1];
    const float cz = cam.R[6] * x + cam.R[7] * y + cam.R[8] * z + cam.t[2];
    const float zz = cz > 1e-8f ? cz : 1e-8f;
    uvz[3 * i] = cam.fx * cx / zz + cam.cx; uvz[3 * i + 1] = cam.fy * cy / zz + cam.cy; uvz[3 * i + 2] = cz;
}

VC_DEVICE float edge_fn(float ax, float ay, float bx, float by, float px, float py) { return (px - ax) * (by - ay) - (py - ay) * (bx - ax); }

// one thread per face: pixels of its bounding box whose centre is strictly inside; nearest z wins (key = z bits, face index)
__global__ __launch_bounds__(256) void mesh_raster_kernel(const float* __restrict__ uvz, const int* __restrict__ faces, int nf, int W, int H,
                                                          unsigned long long* __restrict__ zbuf) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float x0 = uvz[3 * i0], y0 = uvz[3 * i0 + 1], z0 = uvz[3 * i0 + 2];
    const float x1 = uvz[3 * i1], y1 = uvz[3 * i1 + 1], z1 = uvz[3 * i1 + 2];
    const float x2 = uvz[3 * i2], y2 = uvz[3 * i2 + 1], z2 = uvz[3 * i2 + 2];
    if (!(z0 > 1e-8f && z1 > 1e-8f && z2 > 1e-8f)) return;          // a face with a vertex behind the camera is dropped (PyTorch3D z-clip)
    const float area = edge_fn(x0, y0, x1, y1, x2, y2);
    if (fabsf(area) < 1e-12f) return;
    const int bx0 = max(0, (int)floorf(fminf(fminf(x0, x1), x2) - 0.5f)), bx1 = min(W - 1, (int)ceilf(fmaxf(fmaxf(x0, x1), x2) - 0.5f));
    const int by0 = max(0, (int)floorf(fminf(fminf(y0, y1), y2) - 0.5f)), by1 = min(H - 1, (int)ceilf(fmaxf(fmaxf(y0, y1), y2) - 0.5f));
    for (int py = by0; py <= by1; ++py)
        for (int px = bx0; px <= bx1; ++px) {
            const float cx = px + 0.5f, cy = py + 0.5f;
            const float w0 = edge_fn(x1, y1, x2, y2, cx, cy) / area, w1 = edge_fn(x2, y2, x0, y0, cx, cy) / area,
                        w2 = edge_fn(x0, y0, x1, y1, cx, cy) / area;
            if (!(w0 > 0.f && w1 > 0.f && w2 > 0.f)) continue;
            const float iz = w0 / z0 + w1 / z1 + w2 / z2;          // perspective-correct depth
            const float z = 1.f / iz;
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)f;
            atomicMin(zbuf + (int64_t)py * W + px, key);
        }
}

// flat Phong: colour = (ambient + diffuse) * texel + specular with the face normal; light at `light` (world), camera centre `eye`
__global__ __launch_bounds__(256) void mesh_shade_kernel(const float* __restrict__ verts, const float* __restrict__ vcol, const float* __restrict__ uvz,
                                                         const int* __restrict__ faces, const unsigned long long* __restrict__ zbuf, int W, int H,
                                                         float lx, float ly, float lz, float ex, float ey, float ez, uint8_t bgc,
                                                         uint8_t* __restrict__ rgb, float* __restrict__ depth, uint8_t* __restrict__ mask) {
    const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (int64_t)W * H) return;
    const unsigned long long key = zbuf[pix];
    if (key == ~0ull) {
        rgb[3 * pix] = bgc; rgb[3 * pix + 1] = bgc; rgb[3 * pix + 2] = bgc;
        depth[pix] = 0.f; mask[pix] = 0;
        return;
    }
    const int f = (int)(unsigned)key;
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    const float cx = (float)(pix % W) + 0.5f, cy = (float)(pix / W) + 0.5f;
    const float x0 = uvz[3 * i0], y0 = uvz[3 * i0 + 1], z0 = uvz[3 * i0 + 2];
    const float x1 = uvz[3 * i1], y1 = uvz[3 * i1 + 1], z1 = uvz[3 * i1 + 2];
    const float x2 = uvz[3 * i2], y2 = uvz[3 * i2 + 1], z2 = uvz[3 * i2 + 2];
    const float area = edge_fn(x0, y0, x1, y1, x2, y2);
    float w0 = edge_fn(x1, y1, x2, y2, cx, cy) / area, w1 = edge_fn(x2, y2, x0, y0, cx, cy) / area, w2 = edge_fn(x0, y0, x1, y1, cx, cy) / area;
    const float iz = w0 / z0 + w1 / z1 + w2 / z2;
    w0 = w0 / z0 / iz; w1 = w1 / z1 / iz; w2 = w2 / z2 / iz;         // perspective-correct barycentrics
    const float* a = verts + 3 * i0; const float* b = verts + 3 * i1; const float* c = verts + 3 * i2;
    const float p[3] = {w0 * a[0] + w1 * b[0] + w2 * c[0], w0 * a[1] + w1 * b[1] + w2 * c[1], w0 * a[2] + w1 * b[2] + w2 * c[2]};
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
    const float nn = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-6f);
    nx /= nn; ny /= nn; nz /= nn;
    float dlx = lx - p[0], dly = ly - p[1], dlz = lz - p[2];
    const float dl = fmaxf(sqrtf(dlx * dlx + dly * dly + dlz * dlz), 1e-6f);
    dlx /= dl; dly /= dl; dlz /= dl;
    const float cosang = nx * dlx + ny * dly + nz * dlz;
    const float diff = 0.3f * fmaxf(cosang, 0.f);
    float vx = ex - p[0], vy = ey - p[1], vz = ez - p[2];
    const float vn = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-6f);
    vx /= vn; vy /= vn; vz /= vn;
    const float rx = -dlx + 2.f * cosang * nx, ry = -dly + 2.f * cosang * ny, rz = -dlz + 2.f * cosang * nz;
    const float sa = cosang > 0.f ? fmaxf(vx * rx + vy * ry + vz * rz, 0.f) : 0.f;
    const float spec = 0.2f * powf(sa, 64.f);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const float tex = w0 * vcol[3 * i0 + ch] + w1 * vcol[3 * i1 + ch] + w2 * vcol[3 * i2 + ch];
        const float v = (0.5f + diff) * tex + spec;
        rgb[3 * pix + ch] = (uint8_t)(fminf(fmaxf(v, 0.f), 1.f) * 255.f);
    }
    depth[pix] = __uint_as_float((unsigned)(key >> 32));
    mask[pix] = 1;
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = ~0ull;
}

thread_local char g_render_err[256] = "";
int rfail(int code, const char* msg) {
    snprintf(g_render_err, sizeof g_render_err, "%s", msg);
    return code;
}
int rdone(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VC_OK;
    snprintf(g_render_err, sizeof g_render_err, "%s: %s", what, hipGetErrorString(e));
    return VC_E_HIP;
}

Cam make_cam(const float* w2c /*[4][4] row-major*/, const float* K /*[3][3]*/) {
    Cam c;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) c.R[3 * i + j] = w2c[4 * i + j];
        c.t[i] = w2c[4 * i + 3];
    }
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    return c;
}

}  // namespace

extern "C" {

const char* vc_render_last_error(void) { return g_render_err; }

int vc_op_render_composite(const void* bg_rgb, const void* bg_depth, const void* fg_rgb, const void* fg_depth, const void* fg_mask,
                           const void* bg_mask, void* out_rgb, void* out_depth, void* out_mask, int64_t npix, void* stream) {
    if (!bg_depth || !fg_depth || !fg_mask || npix < 0) return rfail(VC_E_INVALID, "vc_op_render_composite: null argument");
    if (out_rgb && (!bg_rgb || !fg_rgb)) return rfail(VC_E_INVALID, "vc_op_render_composite: out_rgb needs both colour inputs");
    if (out_mask && !bg_mask) return rfail(VC_E_INVALID, "vc_op_render_composite: out_mask needs bg_mask");
    if (npix == 0) return VC_OK;
    hipLaunchKernelGGL(render_composite_kernel, dim3(rblocks(npix)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)bg_rgb,
                       (const float*)bg_depth, (const uint8_t*)fg_rgb, (const float*)fg_depth, (const uint8_t*)fg_mask, (const uint8_t*)bg_mask,
                       (uint8_t*)out_rgb, (float*)out_depth, (uint8_t*)out_mask, npix);
    return rdone("render_composite_kernel");
}

int vc_op_render_depth_gray(const void* depth, void* out_rgb, int64_t npix, int normalize, float min_disp, float denom, void* stream) {
    if (!depth || !out_rgb || npix < 0) return rfail(VC_E_INVALID, "vc_op_render_depth_gray: null argument");
    if (npix == 0) return VC_OK;
    hipLaunchKernelGGL(render_depth_gray_kernel, dim3(rblocks(npix)), dim3(256), 0, (hipStream_t)stream, (const float*)depth, (uint8_t*)out_rgb,
                       npix, normalize, min_disp, denom);
    return rdone("render_depth_gray_kernel");
}

int vc_op_render_gauss_density(const void* records, int n, void* out, int W, int H, void* stream) {
    if (!out || W <= 0 || H <= 0 || n < 0 || (n > 0 && !records)) return rfail(VC_E_INVALID, "vc_op_render_gauss_density: bad argument");
    hipLaunchKernelGGL(render_gauss_density_kernel, dim3(rblocks((int64_t)W * H)), dim3(256), 0, (hipStream_t)stream, (const GaussRec*)records, n,
                       (float*)out, W, H);
    return rdone("render_gauss_density_kernel");
}

int vc_op_render_gauss_frame(const void* records, int n, void* scratch_max, float threshold, float span, void* out_rgb, void* out_alpha, int W,
                             int H, void* stream) {
    if (!out_rgb || !out_alpha || W <= 0 || H <= 0 || n < 0 || (n > 0 && (!records || !scratch_max)))
        return rfail(VC_E_INVALID, "vc_op_render_gauss_frame: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (n > 0) {
        if (hipMemsetAsync(scratch_max, 0, (size_t)n * 4, s) != hipSuccess) return rdone("hipMemsetAsync");
        hipLaunchKernelGGL(render_gauss_max_kernel, dim3(rblocks((int64_t)W * H)), dim3(256), 0, s, (const GaussRec*)records, n,
                           (unsigned*)scratch_max, W, H);
    }
    hipLaunchKernelGGL(render_gauss_frame_kernel, dim3(rblocks((int64_t)W * H)), dim3(256), 0, s, (const GaussRec*)records, n,
                       (const unsigned*)scratch_max, threshold, span, (uint8_t*)out_rgb, (float*)out_alpha, W, H);
    return rdone("render_gauss_frame_kernel");
}

int vc_op_render_blend(const void* fg_rgb, const void* alpha, const void* bg_rgb, void* out_rgb, int64_t npix, int masked, void* stream) {
    if (!fg_rgb || !alpha || !out_rgb || (!masked && !bg_rgb) || npix < 0) return rfail(VC_E_INVALID, "vc_op_render_blend: null argument");
    if (npix == 0) return VC_OK;
    hipLaunchKernelGGL(render_blend_kernel, dim3(rblocks(npix)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)fg_rgb, (const float*)alpha,
                       (const uint8_t*)bg_rgb, (uint8_t*)out_rgb, npix, masked);
    return rdone("render_blend_kernel");
}

int64_t vc_op_render_points_scratch_bytes(int64_t npoints, int W, int H, int K) {
    return (npoints * 12 + 7) / 8 * 8 + (int64_t)K * W * H * 8;
}

int vc_op_render_points(const void* points, const void* colors, int64_t npoints, const float* w2c, const float* K3, int W, int H, float radius,
                        int K, float background, void* scratch, void* out_rgb, void* out_depth, void* out_mask, void* stream) {
    if (!w2c || !K3 || !out_rgb || !out_depth || !out_mask || W <= 0 || H <= 0 || K < 1 || K > 64 || npoints < 0 || !(radius > 0.f) ||
        (npoints > 0 && (!points || !colors)) || !scratch)
        return rfail(VC_E_INVALID, "vc_op_render_points: bad argument");
    if (npoints >= (1ll << 32)) return rfail(VC_E_UNSUPPORTED, "vc_op_render_points: more than 2^32 points");
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)W * H;
    float* uvz = (float*)scratch;
    unsigned long long* keys = (unsigned long long*)((char*)scratch + (npoints * 12 + 7) / 8 * 8);
    const float rp = radius * 0.5f * (float)(W < H ? W : H);
    const Cam cam = make_cam(w2c, K3);
    hipLaunchKernelGGL(fill_u64_kernel, dim3(rblocks(hw * K)), dim3(256), 0, s, keys, hw * K);
    if (npoints > 0) {
        hipLaunchKernelGGL(points_project_kernel, dim3(rblocks(npoints)), dim3(256), 0, s, (const float*)points, npoints, cam, uvz);
        for (int k = 0; k < K; ++k)
            hipLaunchKernelGGL(points_select_kernel, dim3(rblocks(npoints)), dim3(256), 0, s, uvz, npoints, rp, W, H,
                               k ? keys + (int64_t)(k - 1) * hw : (const unsigned long long*)nullptr, keys + (int64_t)k * hw);
    }
    hipLaunchKernelGGL(points_composite_kernel, dim3(rblocks(hw)), dim3(256), 0, s, uvz, (const uint8_t*)colors, keys, K, rp, W, H, background,
                       (uint8_t*)out_rgb, (float*)out_depth, (uint8_t*)out_mask);
    return rdone("vc_op_render_points");
}

int64_t vc_op_render_mesh_scratch_bytes(int nverts, int W, int H) { return ((int64_t)nverts * 12 + 7) / 8 * 8 + (int64_t)W * H * 8; }

int vc_op_render_mesh(const void* verts, const void* vert_colors, int nverts, const void* faces, int nfaces, const float* w2c, const float* K3,
                      const float* light_xyz, const float* eye_xyz, int W, int H, int background_u8, void* scratch, void* out_rgb,
                      void* out_depth, void* out_mask, void* stream) {
    if (!w2c || !K3 || !light_xyz || !eye_xyz || !out_rgb || !out_depth || !out_mask || !scratch || W <= 0 || H <= 0 || nverts < 0 ||
        nfaces < 0 || (nfaces > 0 && (!verts || !vert_colors || !faces)))
        return rfail(VC_E_INVALID, "vc_op_render_mesh: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t hw = (int64_t)W * H;
    float* uvz = (float*)scratch;
    unsigned long long* zbuf = (unsigned long long*)((char*)scratch + ((int64_t)nverts * 12 + 7) / 8 * 8);
    const Cam cam = make_cam(w2c, K3);
    hipLaunchKernelGGL(fill_u64_kernel, dim3(rblocks(hw)), dim3(256), 0, s, zbuf, hw);
    if (nfaces > 0) {
        hipLaunchKernelGGL(mesh_project_kernel, dim3(rblocks(nverts)), dim3(256), 0, s, (const float*)verts, nverts, cam, uvz);
        hipLaunchKernelGGL(mesh_raster_kernel, dim3(rblocks(nfaces)), dim3(256), 0, s, uvz, (const int*)faces, nfaces, W, H, zbuf);
    }
    hipLaunchKernelGGL(mesh_shade_kernel, dim3(rblocks(hw)), dim3(256), 0, s, (const float*)verts, (const float*)vert_colors, uvz, (const int*)faces,
                       zbuf, W, H, light_xyz[0], light_xyz[1], light_xyz[2], eye_xyz[0], eye_xyz[1], eye_xyz[2], (uint8_t)background_u8,
                       (uint8_t*)out_rgb, (float*)out_depth, (uint8_t*)out_mask);
    return rdone("vc_op_render_mesh");
}

}  // extern "C"
