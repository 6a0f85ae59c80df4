// Per-object 3D Gaussian fit (step 3 of the reference's pre-processing chain, upstream of the control-map renderer):
// the per-pixel / per-point stages of /root/reference inference/fit_3D_gaussian.py.  Once per clip on one 720p frame - HBM-bound
// byte / float work, a few launches of < 1 M threads.
//
//   vc_op_fit_erode_mask   load_mask :139-159                   (m > 127, one erosion by cv2's k x k MORPH_ELLIPSE element)
//   vc_op_fit_points       get_point_cloud_from_depth :35-92    (unproject, camera-to-world, ORDERED compaction of the kept pixels)
//   vc_op_fit_moments      fit_3d_gaussian :95-136              (mean, unbiased covariance + 1e-6 I; two passes, fp64 accumulation)
//   vc_op_fit_project      project_gaussian_to_2d :171-287      (pdf and squared Mahalanobis distance inside the 3-sigma box)
//   vc_op_fit_blend        visualize_gaussian_projections :374-397 (confidence-ellipse mask, density-normalised alpha blend)
//   vc_op_fit_picture_u8   :400 (clamp(0, 1) x 255 truncated)
// Pinned by the reference's own outputs for its two demo clips (tests/golden/demo_fit/, oracle/fit_oracle.py).
#include <math.h>
#include <stdint.h>

#include "../../include/vcengine.h"
#include "vc_common.h"
#include "vc_kernels.h"

namespace {

constexpr int FIT_MAX_K = 31;         // largest structuring element
constexpr int FIT_CHUNK = 2048;       // pixels per block of the compaction (8 rounds of 256)
constexpr int FIT_PARTS = 1024;       // partial sums of the moment passes

thread_local char g_fit_err[256] = "";
int ffail(int code, const char* msg) {
    snprintf(g_fit_err, sizeof g_fit_err, "%s", msg);
    return code;
}
int fdone(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return VC_OK;
    snprintf(g_fit_err, sizeof g_fit_err, "%s: %s", what, hipGetErrorString(e));
    return VC_E_HIP;
}
inline int fblocks(int64_t n) { return (int)((n + 255) / 256); }

// ---- mask ----------------------------------------------------------------------------------------------------------------------
struct Element { int k; int half[FIT_MAX_K]; };      // row i of the k x k element covers the columns |j - k/2| <= half[i] (-1: empty)

__global__ __launch_bounds__(256) void fit_erode_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ out, int W, int H, Element el) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)W * H) return;
    const int x = (int)(i % W), y = (int)(i / W), a = el.k / 2;
    bool keep = true;
    for (int r = 0; r < el.k && keep; ++r) {
        const int yy = y + r - a, h = el.half[r];
        if (h < 0 || yy < 0 || yy >= H) continue;                        // outside the image nothing is removed (cv2.erode's border)
        for (int xx = max(x - h, 0); xx <= min(x + min(h, el.k - 1 - a), W - 1); ++xx)      // an even-sized element is clipped on the right
            if (raw[(int64_t)yy * W + xx] <= 127) { keep = false; break; }
    }
    out[i] = keep ? 1 : 0;
}

// ---- depth -> world points, ordered compaction ------------------------------------------------------------------------------
struct Unproject { float ki[9]; float c2w[12]; };    // K^-1 row-major; the top three rows of the camera-to-world matrix

VC_DEVICE bool fit_keep(const float* depth, const uint8_t* mask, int64_t i) { return mask ? mask[i] != 0 : depth[i] > 0.f; }

__global__ __launch_bounds__(256) void fit_count_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ mask, int64_t n,
                                                        int* __restrict__ counts) {
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * FIT_CHUNK;
    int c = 0;
    for (int r = 0; r < FIT_CHUNK / 256; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        c += (i < n && fit_keep(depth, mask, i)) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts by ONE workgroup (a 720p frame has 450 of them); the total goes to *total
__global__ __launch_bounds__(256) void fit_scan_kernel(const int* __restrict__ counts, int nblk, int64_t* __restrict__ offsets,
                                                       int64_t* __restrict__ total) {
    __shared__ int64_t part[256];
    const int per = (nblk + 255) / 256, lo = threadIdx.x * per, hi = min(lo + per, nblk);
    int64_t s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int i = 0; i < 256; ++i) { const int64_t v = part[i]; part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
}

__global__ __launch_bounds__(256) void fit_scatter_kernel(const float* __restrict__ depth, const uint8_t* __restrict__ mask, int W, int64_t n,
                                                          Unproject u, const int64_t* __restrict__ offsets, float* __restrict__ points) {
    __shared__ int wcnt[4];
    const int64_t base = (int64_t)blockIdx.x * FIT_CHUNK;
    int64_t run = offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < FIT_CHUNK / 256; ++r) {
        const int64_t i = base + r * 256 + threadIdx.x;
        const bool keep = i < n && fit_keep(depth, mask, i);
        const unsigned long long b = __ballot(keep);
        if (lane == 0) wcnt[wave] = __popcll(b);
        __syncthreads();
        int before = __popcll(b & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) before += wcnt[w];
        const int round_total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (keep) {
            const float x = (float)(i % W), y = (float)(i / W), d = depth[i];
            const float cx = fmaf(u.ki[0], x, fmaf(u.ki[1], y, u.ki[2])) * d;
            const float cy = fmaf(u.ki[3], x, fmaf(u.ki[4], y, u.ki[5])) * d;
            const float cz = fmaf(u.ki[6], x, fmaf(u.ki[7], y, u.ki[8])) * d;
            float* p = points + 3 * (run + before);
            p[0] = fmaf(u.c2w[0], cx, fmaf(u.c2w[1], cy, fmaf(u.c2w[2], cz, u.c2w[3])));
            p[1] = fmaf(u.c2w[4], cx, fmaf(u.c2w[5], cy, fmaf(u.c2w[6], cz, u.c2w[7])));
            p[2] = fmaf(u.c2w[8], cx, fmaf(u.c2w[9], cy, fmaf(u.c2w[10], cz, u.c2w[11])));
        }
        run += round_total;
        __syncthreads();
    }
}

// ---- moments -----------------------------------------------------------------------------------------------------------------
// fixed-order tree reduction of NV doubles per thread over a 256-thread block: the result does not depend on timing
template <int NV>
VC_DEVICE void block_sum(double (&v)[NV], double* out /*[NV]*/) {
    __shared__ double sh[4][NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) out[k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
}

__global__ __launch_bounds__(256) void fit_sum_kernel(const float* __restrict__ pts, int64_t n, double* __restrict__ partial /*[PARTS][3]*/) {
    double v[3] = {0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        v[0] += pts[3 * i]; v[1] += pts[3 * i + 1]; v[2] += pts[3 * i + 2];
    }
    block_sum<3>(v, partial + 3 * blockIdx.x);
}

__global__ __launch_bounds__(256) void fit_centred_kernel(const float* __restrict__ pts, int64_t n, const double* __restrict__ mean,
                                                          double* __restrict__ partial /*[PARTS][6]*/) {
    const double m0 = mean[0], m1 = mean[1], m2 = mean[2];
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double a = pts[3 * i] - m0, b = pts[3 * i + 1] - m1, c = pts[3 * i + 2] - m2;
        v[0] += a * a; v[1] += a * b; v[2] += a * c; v[3] += b * b; v[4] += b * c; v[5] += c * c;
    }
    block_sum<6>(v, partial + 6 * blockIdx.x);
}

// one thread: partial sums in index order -> mean (kept in fp64 for the second pass, fp32 for the caller)
__global__ void fit_mean_final_kernel(const double* __restrict__ partial, int parts, int64_t n, double* __restrict__ mean, float* __restrict__ out) {
    double s[3] = {0, 0, 0};
    for (int p = 0; p < parts; ++p)
        for (int k = 0; k < 3; ++k) s[k] += partial[3 * p + k];
    for (int k = 0; k < 3; ++k) { mean[k] = s[k] / (double)n; out[k] = (float)mean[k]; }
}

__global__ void fit_cov_final_kernel(const double* __restrict__ partial, int parts, int64_t n, float* __restrict__ out /* mean[3] cov[9] */) {
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int p = 0; p < parts; ++p)
        for (int k = 0; k < 6; ++k) s[k] += partial[6 * p + k];
    const double d = (double)(n - 1);
    const double c00 = s[0] / d + 1e-6, c01 = s[1] / d, c02 = s[2] / d, c11 = s[3] / d + 1e-6, c12 = s[4] / d, c22 = s[5] / d + 1e-6;
    out[3] = (float)c00; out[4] = (float)c01; out[5] = (float)c02;
    out[6] = (float)c01; out[7] = (float)c11; out[8] = (float)c12;
    out[9] = (float)c02; out[10] = (float)c12; out[11] = (float)c22;
}

// ---- projection picture ----------------------------------------------------------------------------------------------------------
struct ProjRec { float mx, my, i00, i01, i10, i11, coeff; int x0, x1, y0, y1; };      // roi = [x0, x1) x [y0, y1); empty = culled

__global__ __launch_bounds__(256) void fit_project_kernel(ProjRec q, float* __restrict__ density, float* __restrict__ mahal, int W, int H,
                                                          unsigned* __restrict__ dmax_bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float d = 0.f, m = INFINITY;
    if (i < (int64_t)W * H) {
        const int x = (int)(i % W), y = (int)(i / W);
        if (x >= q.x0 && x < q.x1 && y >= q.y0 && y < q.y1) {
            const float dx = (float)x - q.mx, dy = (float)y - q.my;
            m = dx * q.i00 * dx + dx * q.i01 * dy + dy * q.i10 * dx + dy * q.i11 * dy;      // einsum 'ijk,kl,ijl->ij' term by term
            d = q.coeff * expf(-0.5f * m);
        }
        density[i] = d;
        mahal[i] = m;
    }
    if (dmax_bits) {                                                                       // densities are >= 0: their bit patterns order like the values
        float w = d;
        for (int o = 32; o > 0; o >>= 1) w = fmaxf(w, __shfl_down(w, o));
        if ((threadIdx.x & 63) == 0 && w > 0.f) atomicMax(dmax_bits, __float_as_uint(w));
    }
}

__global__ __launch_bounds__(256) void fit_blend_kernel(const float* __restrict__ density, const float* __restrict__ mahal,
                                                        const float* __restrict__ dmax, float thr, float r, float g, float b,
                                                        float* __restrict__ rgb, float* __restrict__ mask, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mask[i] = fmaxf(mask[i], mahal[i] <= thr ? 1.f : 0.f);
    const float top = *dmax;
    const float a = top > 0.f ? fminf(fmaxf(density[i] / top, 0.f), 1.f) : 0.f;
    rgb[3 * i] = r * a + rgb[3 * i] * (1.f - a);
    rgb[3 * i + 1] = g * a + rgb[3 * i + 1] * (1.f - a);
    rgb[3 * i + 2] = b * a + rgb[3 * i + 2] * (1.f - a);
}

__global__ __launch_bounds__(256) void fit_u8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (uint8_t)(fminf(fmaxf(src[i], 0.f), 1.f) * 255.f);
}

}  // namespace

extern "C" {

const char* vc_fit_last_error(void) { return g_fit_err; }

int vc_op_fit_erode_mask(const void* mask_u8, void* out_u8, int W, int H, int ksize, void* stream) {
    if (!mask_u8 || !out_u8 || W <= 0 || H <= 0) return ffail(VC_E_INVALID, "vc_op_fit_erode_mask: bad argument");
    if (ksize < 1 || ksize > FIT_MAX_K) return ffail(VC_E_UNSUPPORTED, "vc_op_fit_erode_mask: element size outside [1, 31]");
    Element el;
    el.k = ksize;
    const int r = ksize / 2, c = ksize / 2;
    for (int i = 0; i < FIT_MAX_K; ++i) el.half[i] = -1;
    for (int i = 0; i < ksize; ++i) {                 // cv2.getStructuringElement(MORPH_ELLIPSE): dx = round(c sqrt((r^2 - dy^2) / r^2))
        const int dy = i - r;
        if (abs(dy) > r) continue;
        el.half[i] = r ? (int)nearbyint(c * sqrt((double)(r * r - dy * dy) / (double)(r * r))) : 0;
    }
    hipLaunchKernelGGL(fit_erode_kernel, dim3(fblocks((int64_t)W * H)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)mask_u8,
                       (uint8_t*)out_u8, W, H, el);
    return fdone("fit_erode_kernel");
}

int64_t vc_op_fit_points_scratch_bytes(int W, int H) {
    const int64_t nblk = ((int64_t)W * H + FIT_CHUNK - 1) / FIT_CHUNK;
    return nblk * 4 + 8 + nblk * 8;
}

int vc_op_fit_points(const void* depth, const void* mask, const float* kinv, const float* c2w, int W, int H, void* scratch, void* out_points,
                     void* out_count, void* stream) {
    if (!depth || !kinv || !c2w || !scratch || !out_points || !out_count || W <= 0 || H <= 0)
        return ffail(VC_E_INVALID, "vc_op_fit_points: bad argument");
    const int64_t n = (int64_t)W * H;
    const int64_t nblk = (n + FIT_CHUNK - 1) / FIT_CHUNK;
    if (nblk > (1ll << 30)) return ffail(VC_E_UNSUPPORTED, "vc_op_fit_points: image too large");
    hipStream_t s = (hipStream_t)stream;
    int64_t* offsets = (int64_t*)scratch;
    int* counts = (int*)(offsets + nblk);
    Unproject u;
    for (int i = 0; i < 9; ++i) u.ki[i] = kinv[i];
    for (int i = 0; i < 12; ++i) u.c2w[i] = c2w[i];
    hipLaunchKernelGGL(fit_count_kernel, dim3((int)nblk), dim3(256), 0, s, (const float*)depth, (const uint8_t*)mask, n, counts);
    hipLaunchKernelGGL(fit_scan_kernel, dim3(1), dim3(256), 0, s, (const int*)counts, (int)nblk, offsets, (int64_t*)out_count);
    hipLaunchKernelGGL(fit_scatter_kernel, dim3((int)nblk), dim3(256), 0, s, (const float*)depth, (const uint8_t*)mask, W, n, u,
                       (const int64_t*)offsets, (float*)out_points);
    return fdone("fit_points kernels");
}

int64_t vc_op_fit_moments_scratch_bytes(void) { return (int64_t)FIT_PARTS * 6 * 8 + 3 * 8; }

int vc_op_fit_moments(const void* points, int64_t n, void* scratch, void* out12, void* stream) {
    if (!points || !scratch || !out12) return ffail(VC_E_INVALID, "vc_op_fit_moments: null argument");
    if (n < 3) return ffail(VC_E_INVALID, "vc_op_fit_moments: fewer than 3 points");
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)scratch;
    double* mean = partial + (int64_t)FIT_PARTS * 6;
    const int parts = (int)((n + 255) / 256 < FIT_PARTS ? (n + 255) / 256 : FIT_PARTS);
    hipLaunchKernelGGL(fit_sum_kernel, dim3(parts), dim3(256), 0, s, (const float*)points, n, partial);
    hipLaunchKernelGGL(fit_mean_final_kernel, dim3(1), dim3(1), 0, s, (const double*)partial, parts, n, mean, (float*)out12);
    hipLaunchKernelGGL(fit_centred_kernel, dim3(parts), dim3(256), 0, s, (const float*)points, n, (const double*)mean, partial);
    hipLaunchKernelGGL(fit_cov_final_kernel, dim3(1), dim3(1), 0, s, (const double*)partial, parts, n, (float*)out12);
    return fdone("fit_moments kernels");
}

int vc_op_fit_project(const float* rec11, void* density, void* mahal, void* dmax, int W, int H, void* stream) {
    if (!rec11 || !density || !mahal || W <= 0 || H <= 0) return ffail(VC_E_INVALID, "vc_op_fit_project: bad argument");
    ProjRec q;
    q.mx = rec11[0]; q.my = rec11[1]; q.i00 = rec11[2]; q.i01 = rec11[3]; q.i10 = rec11[4]; q.i11 = rec11[5]; q.coeff = rec11[6];
    q.x0 = (int)rec11[7]; q.x1 = (int)rec11[8]; q.y0 = (int)rec11[9]; q.y1 = (int)rec11[10];
    hipStream_t s = (hipStream_t)stream;
    if (dmax && hipMemsetAsync(dmax, 0, 4, s) != hipSuccess) return fdone("vc_op_fit_project: memset");
    hipLaunchKernelGGL(fit_project_kernel, dim3(fblocks((int64_t)W * H)), dim3(256), 0, s, q, (float*)density, (float*)mahal, W, H, (unsigned*)dmax);
    return fdone("fit_project_kernel");
}

int vc_op_fit_blend(const void* density, const void* mahal, const void* dmax, float threshold, const float* rgb3, void* picture, void* mask,
                    int64_t npix, void* stream) {
    if (!density || !mahal || !dmax || !rgb3 || !picture || !mask || npix < 0) return ffail(VC_E_INVALID, "vc_op_fit_blend: null argument");
    if (npix == 0) return VC_OK;
    hipLaunchKernelGGL(fit_blend_kernel, dim3(fblocks(npix)), dim3(256), 0, (hipStream_t)stream, (const float*)density, (const float*)mahal,
                       (const float*)dmax, threshold, rgb3[0], rgb3[1], rgb3[2], (float*)picture, (float*)mask, npix);
    return fdone("fit_blend_kernel");
}

int vc_op_fit_picture_u8(const void* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0) return ffail(VC_E_INVALID, "vc_op_fit_picture_u8: null argument");
    if (n == 0) return VC_OK;
    hipLaunchKernelGGL(fit_u8_kernel, dim3(fblocks(n)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (uint8_t*)dst, n);
    return fdone("fit_u8_kernel");
}

}  // extern "C"
