// Interpolation tail of the key-frame pipeline (all HBM/latency bound):
//   warp = F.grid_sample(bilinear, border, align_corners=False)       flow/model.py:244-249
//   up   = F.interpolate(bilinear, align_corners=True)                flow/model.py:193,206,218,228
//   fuse = (n-p)/n * fwd[p-1] + p/n * bwd[n-p-1]                      flow/model.py:232-237
//   post = F.interpolate(.., (1072,1920)) -> max(1)[1] -> uint8       flow/base.py:275-277
// plus the metric histogram of util/util.py:52-63.
#include "interp.h"
#include "kernels.h"

namespace fs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The flat element index of these kernels is decomposed with divisions by run-time sizes: as a 64-bit value that is ~100 instructions
// per division, several times the interpolation itself.  Every kernel is a template on the index type and runs on unsigned 32-bit
// indices whenever the element count fits (always, for the frame sizes of this path); int64_t remains for anything larger.
// ------------------------------------------------------------------ grid_sample, NCHW
template <typename I>
__global__ __launch_bounds__(256) void grid_sample_nchw_kernel(const float* __restrict__ in, int B, int C, int Hi, int Wi,
                                                               const float* __restrict__ grid, int Hg, int Wg,
                                                               float* __restrict__ out, int ac) {
    const I total = (I)B * Hg * Wg;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const int b = (int)(i / ((I)Hg * Wg));
        const I g = i - (I)b * Hg * Wg;
        const float gx = grid[i * 2 + 0], gy = grid[i * 2 + 1];
        const GsTaps t = gs_taps(gx, gy, Wi, Hi, ac);
        const int x1 = t.x1ok ? t.x0 + 1 : t.x0, y1 = t.y1ok ? t.y0 + 1 : t.y0;
        for (int c = 0; c < C; ++c) {
            const float* pl = in + ((size_t)b * C + c) * Hi * Wi;
            const float vnw = pl[(size_t)t.y0 * Wi + t.x0];
            const float vne = t.x1ok ? pl[(size_t)t.y0 * Wi + x1] : 0.f;
            const float vsw = t.y1ok ? pl[(size_t)y1 * Wi + t.x0] : 0.f;
            const float vse = (t.x1ok && t.y1ok) ? pl[(size_t)y1 * Wi + x1] : 0.f;
            out[((size_t)b * C + c) * Hg * Wg + g] = gs_combine(vnw, vne, vsw, vse, t);
        }
    }
}

int launch_grid_sample_nchw(const float* in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg, float* out,
                            int align_corners, hipStream_t s) {
    const int64_t total = (int64_t)B * Hg * Wg;
    if (total < ((int64_t)1 << 31)) hipLaunchKernelGGL((grid_sample_nchw_kernel<unsigned>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 8192)), dim3(256), 0, s, in,
                       B, C, Hi, Wi, grid, Hg, Wg, out, align_corners);
    else hipLaunchKernelGGL((grid_sample_nchw_kernel<int64_t>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 8192)), dim3(256), 0, s, in,
                       B, C, Hi, Wi, grid, Hg, Wg, out, align_corners);
    FS_HIP(hipGetLastError());
    return 0;
}

// The four taps of one grid_sample output for four channels of an NHWC map (`base` = the map + the channel offset).  All four loads are
// issued unconditionally from clamped, always valid addresses and the out-of-image taps are zeroed afterwards: written as
// `ok ? load : 0` the compiler has to branch around each load and waits for one before it issues the next (four serial round trips).
__device__ __forceinline__ f32x4 gs_gather_nhwc(const float* __restrict__ base, int Wi, int ld, const GsTaps& t) {
    const int x1 = t.x1ok ? t.x0 + 1 : t.x0, y1 = t.y1ok ? t.y0 + 1 : t.y0;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 vnw = *reinterpret_cast<const f32x4*>(base + ((size_t)t.y0 * Wi + t.x0) * ld);
    f32x4 vne = *reinterpret_cast<const f32x4*>(base + ((size_t)t.y0 * Wi + x1) * ld);
    f32x4 vsw = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * Wi + t.x0) * ld);
    f32x4 vse = *reinterpret_cast<const f32x4*>(base + ((size_t)y1 * Wi + x1) * ld);
    vne = t.x1ok ? vne : z;
    vsw = t.y1ok ? vsw : z;
    vse = (t.x1ok && t.y1ok) ? vse : z;
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = gs_combine(vnw[q], vne[q], vsw[q], vse[q], t);
    return r;
}

// ------------------------------------------------------------------ grid_sample, NHWC (C % 4 == 0)
// One wave-sized group of float4 lanes sweeps the channels of one output pixel: every tap is a
// contiguous C*4-byte run, so the gather is fully coalesced.
template <typename I>
__global__ __launch_bounds__(256) void grid_sample_nhwc_kernel(const float* __restrict__ in, int ld_in, int B, int C4, int Hi,
                                                               int Wi, const float* __restrict__ grid, int Hg, int Wg,
                                                               float* __restrict__ out, int ld_out, int ac) {
    const I total = (I)B * Hg * Wg * C4;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const I m = i / C4;
        const int b = (int)(m / ((I)Hg * Wg));
        const float gx = grid[m * 2 + 0], gy = grid[m * 2 + 1];
        const GsTaps t = gs_taps(gx, gy, Wi, Hi, ac);
        const float* base = in + (size_t)b * Hi * Wi * ld_in + c4 * 4;
        *reinterpret_cast<f32x4*>(out + (size_t)m * ld_out + c4 * 4) = gs_gather_nhwc(base, Wi, ld_in, t);
    }
}

int launch_grid_sample_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg,
                            float* out, int ld_out, int align_corners, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0, "grid_sample_nhwc: C/ld must be multiples of 4");
    const int64_t total = (int64_t)B * Hg * Wg * (C / 4);
    if (total < ((int64_t)1 << 31)) hipLaunchKernelGGL((grid_sample_nhwc_kernel<unsigned>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 8192)), dim3(256), 0, s, in,
                       ld_in, B, C / 4, Hi, Wi, grid, Hg, Wg, out, ld_out, align_corners);
    else hipLaunchKernelGGL((grid_sample_nhwc_kernel<int64_t>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 8192)), dim3(256), 0, s, in,
                       ld_in, B, C / 4, Hi, Wi, grid, Hg, Wg, out, ld_out, align_corners);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ bilinear resize
template <typename I>
__global__ __launch_bounds__(256) void resize_bilinear_nchw_kernel(const float* __restrict__ in, int BC, int Hi, int Wi,
                                                                   float* __restrict__ out, int Ho, int Wo, int ac, float sy,
                                                                   float sx) {
    const I total = (I)BC * Ho * Wo;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const int oy = (int)((i / Wo) % Ho);
        const I pc = i / ((I)Wo * Ho);
        const LinCoord cy = lin_coord(oy, Hi, sy, ac), cx = lin_coord(ox, Wi, sx, ac);
        const float* pl = in + (size_t)pc * Hi * Wi;
        out[i] = bilerp(pl[cy.i0 * Wi + cx.i0], pl[cy.i0 * Wi + cx.i1], pl[cy.i1 * Wi + cx.i0], pl[cy.i1 * Wi + cx.i1], cy, cx);
    }
}

int launch_resize_bilinear_nchw(const float* in, int BC, int Hi, int Wi, float* out, int Ho, int Wo, int align_corners,
                                hipStream_t s) {
    const int64_t total = (int64_t)BC * Ho * Wo;
    if (total < ((int64_t)1 << 31)) hipLaunchKernelGGL((resize_bilinear_nchw_kernel<unsigned>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s,
                       in, BC, Hi, Wi, out, Ho, Wo, align_corners, resize_scale(Hi, Ho, align_corners),
                       resize_scale(Wi, Wo, align_corners));
    else hipLaunchKernelGGL((resize_bilinear_nchw_kernel<int64_t>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s,
                       in, BC, Hi, Wi, out, Ho, Wo, align_corners, resize_scale(Hi, Ho, align_corners),
                       resize_scale(Wi, Wo, align_corners));
    FS_HIP(hipGetLastError());
    return 0;
}

template <typename I>
__global__ __launch_bounds__(256) void resize_bilinear_nhwc_kernel(const float* __restrict__ in, int ld_in, int B, int C4, int Hi,
                                                                   int Wi, float* __restrict__ out, int ld_out, int Ho, int Wo,
                                                                   int ac, float sy, float sx) {
    const I total = (I)B * Ho * Wo * C4;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const I m = i / C4;
        const int ox = (int)(m % Wo);
        const int oy = (int)((m / Wo) % Ho);
        const int b = (int)(m / ((I)Wo * Ho));
        const LinCoord cy = lin_coord(oy, Hi, sy, ac), cx = lin_coord(ox, Wi, sx, ac);
        const float* base = in + (size_t)b * Hi * Wi * ld_in + c4 * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)cy.i0 * Wi + cx.i0) * ld_in);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)cy.i0 * Wi + cx.i1) * ld_in);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)cy.i1 * Wi + cx.i0) * ld_in);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)cy.i1 * Wi + cx.i1) * ld_in);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = bilerp(v00[e], v01[e], v10[e], v11[e], cy, cx);
        *reinterpret_cast<f32x4*>(out + (size_t)m * ld_out + c4 * 4) = r;
    }
}

int launch_resize_bilinear_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, float* out, int ld_out, int Ho, int Wo,
                                int align_corners, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0, "resize_nhwc: C/ld must be multiples of 4");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    if (total < ((int64_t)1 << 31)) hipLaunchKernelGGL((resize_bilinear_nhwc_kernel<unsigned>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s,
                       in, ld_in, B, C / 4, Hi, Wi, out, ld_out, Ho, Wo, align_corners, resize_scale(Hi, Ho, align_corners),
                       resize_scale(Wi, Wo, align_corners));
    else hipLaunchKernelGGL((resize_bilinear_nhwc_kernel<int64_t>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s,
                       in, ld_in, B, C / 4, Hi, Wi, out, ld_out, Ho, Wo, align_corners, resize_scale(Hi, Ho, align_corners),
                       resize_scale(Wi, Wo, align_corners));
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ blend: out = wa*a + wb*b
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ a, float wa, const float* __restrict__ b, float wb,
                                                    float* __restrict__ out, int64_t n4, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
        f32x4 r;
        if (b) {
            const f32x4 y = reinterpret_cast<const f32x4*>(b)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __fadd_rn(__fmul_rn(wa, x[e]), __fmul_rn(wb, y[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __fmul_rn(wa, x[e]);
        }
        reinterpret_cast<f32x4*>(out)[i] = r;
    }
    // scalar tail
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < numel; i += 256)
            out[i] = b ? __fadd_rn(__fmul_rn(wa, a[i]), __fmul_rn(wb, b[i])) : __fmul_rn(wa, a[i]);
    }
}

__global__ __launch_bounds__(256) void blend_scalar_kernel(const float* __restrict__ a, float wa, const float* __restrict__ b, float wb,
                                                           float* __restrict__ out, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256)
        out[i] = b ? __fadd_rn(__fmul_rn(wa, a[i]), __fmul_rn(wb, b[i])) : __fmul_rn(wa, a[i]);
}

int launch_blend(const float* a, float wa, const float* b, float wb, float* out, int64_t numel, hipStream_t s) {
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) != 0) {
        // batch slices of odd-sized maps are only 4-B aligned: scalar path
        hipLaunchKernelGGL(blend_scalar_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(numel, 256), 16384))),
                           dim3(256), 0, s, a, wa, b, wb, out, numel);
        FS_HIP(hipGetLastError());
        return 0;
    }
    const int64_t n4 = numel / 4;
    hipLaunchKernelGGL(blend_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n4, 256), 16384))), dim3(256), 0, s,
                       a, wa, b, wb, out, n4, numel);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ fused predict_segmentation tail
// Virtual key-frame map: value of up_ac(lo)[k] at integer pixel (y, x) of the H x W frame.
__device__ __forceinline__ float up_at(const float* __restrict__ pl, int h, int w, int y, int x, float sy, float sx) {
    const LinCoord cy = lin_coord(y, h, sy, 1), cx = lin_coord(x, w, sx, 1);
    return bilerp(pl[cy.i0 * w + cx.i0], pl[cy.i0 * w + cx.i1], pl[cy.i1 * w + cx.i0], pl[cy.i1 * w + cx.i1], cy, cx);
}

// One warp step for both directions (blockIdx.y = direction).  Step 0 samples the virtual
// H x W map up_ac(lo) (flow/model.py:193 replaces o by its upsampled version before warping);
// later steps sample the previous Hg x Wg result (the chain stays at grid resolution).
__global__ __launch_bounds__(256) void seg_warp_step_kernel(const float* __restrict__ lo_prev, const float* __restrict__ lo_next,
                                                            const float* __restrict__ src_f, const float* __restrict__ src_b,
                                                            const float* __restrict__ grid_f, const float* __restrict__ grid_b,
                                                            float* __restrict__ dst_f, float* __restrict__ dst_b, int K, int h,
                                                            int w, int H, int W, int Hg, int Wg, int first, float sy, float sx) {
    const int dir = blockIdx.y;
    const float* lo = dir ? lo_next : lo_prev;
    const float* src = dir ? src_b : src_f;
    const float* grid = dir ? grid_b : grid_f;
    float* dst = dir ? dst_b : dst_f;
    const int G = Hg * Wg;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
        const float gx = grid[g * 2 + 0], gy = grid[g * 2 + 1];
        const int Hs = first ? H : Hg, Ws = first ? W : Wg;
        const GsTaps t = gs_taps(gx, gy, Ws, Hs, 0);
        const int x1 = t.x1ok ? t.x0 + 1 : t.x0, y1 = t.y1ok ? t.y0 + 1 : t.y0;
        for (int k = 0; k < K; ++k) {
            float vnw, vne, vsw, vse;
            if (first) {
                const float* pl = lo + (size_t)k * h * w;
                vnw = up_at(pl, h, w, t.y0, t.x0, sy, sx);
                vne = t.x1ok ? up_at(pl, h, w, t.y0, x1, sy, sx) : 0.f;
                vsw = t.y1ok ? up_at(pl, h, w, y1, t.x0, sy, sx) : 0.f;
                vse = (t.x1ok && t.y1ok) ? up_at(pl, h, w, y1, x1, sy, sx) : 0.f;
            } else {
                const float* pl = src + (size_t)k * G;
                vnw = pl[t.y0 * Ws + t.x0];
                vne = t.x1ok ? pl[t.y0 * Ws + x1] : 0.f;
                vsw = t.y1ok ? pl[y1 * Ws + t.x0] : 0.f;
                vse = (t.x1ok && t.y1ok) ? pl[y1 * Ws + x1] : 0.f;
            }
            dst[(size_t)k * G + g] = gs_combine(vnw, vne, vsw, vse, t);
        }
    }
}

// Fusion + upsample (+ argmax | + softmax accumulated into the sliding-crop canvas).  One thread per output pixel; KMAX
// classes kept in registers.  Canvas mode (p.canvas != nullptr) is compute_predict_crop + the accumulation of compute_output
// (flow/base.py:204-205, 226-234) without the [n,K,h,w] logits ever reaching HBM: softmax over K in fp32 exactly as
// softmax_accumulate_kernel does it on materialised logits (max, exp(x - max), sum, divide), added to the float64 canvas at
// the crop's offset; successive crops are successive launches on one stream, so overlapping pixels never race.
// WARP / CANVAS are compile-time: the headline route (linear interpolation, logits + masks) carries neither the warp path's eight
// gathers per class and frame nor the softmax / float64 canvas code (one run-time kernel for all modes needed 167 registers: three
// waves per SIMD for a pass that only waits for its stores).  Same operations in the same order in every instantiation.
template <int KMAX, bool WARP, bool CANVAS>
__global__ __launch_bounds__(256) void seg_fuse_kernel(SegTailParams p, float sy_lo, float sx_lo, float sy_g, float sx_g) {
    const int64_t HW = (int64_t)p.H * p.W;
    const int K = p.K, n = p.n;
    const size_t cHW = (size_t)p.cH * p.cW;
    // grid = (x blocks of 256 pixels, rows): no division per pixel (a flat 64-bit index cost ~200 instructions of i / W, i % W per
    // thread -- as much as the 10 bilinear taps)
    for (int y = blockIdx.y; y < p.H; y += gridDim.y) {
        const int x = blockIdx.x * 256 + threadIdx.x;
        if (x >= p.W) break;
        const int64_t i = (int64_t)y * p.W + x;
        const unsigned off4 = (unsigned)i * 4u;  // byte offset inside one H x W plane (H * W < 2^30, launcher): with a uniform plane base
                                                 // the stores take the scalar-base + 32-bit-offset form, no 64-bit address per plane
        const size_t cpix = (size_t)(p.y0 + y) * p.cW + (p.x0 + x);
        float a[KMAX], b[KMAX], v[KMAX];
        // one output frame: logits v[0..K) of this pixel -> the requested outputs
        auto emit = [&](int f) {
            if (p.out_logits)
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) *reinterpret_cast<float*>(reinterpret_cast<char*>(p.out_logits + ((size_t)f * K + k) * HW) + off4) = v[k];
            if (p.out_mask) {
                float best = -INFINITY;
                int arg = 0;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K && v[k] > best) { best = v[k]; arg = k; }
                p.out_mask[(size_t)f * HW + i] = (uint8_t)arg;
            }
            if (CANVAS) {
                float mx = v[0];
#pragma unroll
                for (int k = 1; k < KMAX; ++k)
                    if (k < K) mx = fmaxf(mx, v[k]);
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) sum += expf(v[k] - mx);
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) p.canvas[((size_t)f * K + k) * cHW + cpix] += (double)(expf(v[k] - mx) / sum);
                if (f == 0) p.count[cpix] += 1.0;
            }
        };
        // frame 0: the key frame itself
        {
            const LinCoord cy = lin_coord(y, p.h, sy_lo, 1), cx = lin_coord(x, p.w, sx_lo, 1);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    const float* pl = p.lo_prev + (size_t)k * p.h * p.w;
                    a[k] = bilerp(pl[cy.i0 * p.w + cx.i0], pl[cy.i0 * p.w + cx.i1], pl[cy.i1 * p.w + cx.i0],
                                  pl[cy.i1 * p.w + cx.i1], cy, cx);
                    if (p.lo_next && !WARP) {
                        const float* pn = p.lo_next + (size_t)k * p.h * p.w;
                        b[k] = bilerp(pn[cy.i0 * p.w + cx.i0], pn[cy.i0 * p.w + cx.i1], pn[cy.i1 * p.w + cx.i0],
                                      pn[cy.i1 * p.w + cx.i1], cy, cx);
                    }
                    v[k] = a[k];
                }
            }
            emit(0);
        }
        if (!p.lo_next) continue;
        LinCoord gy_c, gx_c;
        if (WARP) {
            gy_c = lin_coord(y, p.Hg, sy_g, 1);
            gx_c = lin_coord(x, p.Wg, sx_g, 1);
        }
        const int G = p.Hg * p.Wg;
        for (int f = 1; f < n; ++f) {
            const float wa = (float)((double)(n - f) / (double)n);
            const float wb = (float)((double)f / (double)n);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    float va, vb;
                    if (!WARP) {
                        va = a[k];
                        vb = b[k];
                    } else {
                        // scratch layout: [dir][step][K][Hg][Wg]; forward map f-1, backward map n-f-1
                        const float* pf = p.scratch + ((size_t)(f - 1) * K + k) * G;
                        const float* pb = p.scratch + ((size_t)(n - 1) * K + (size_t)(n - f - 1) * K + k) * G;
                        va = bilerp(pf[gy_c.i0 * p.Wg + gx_c.i0], pf[gy_c.i0 * p.Wg + gx_c.i1], pf[gy_c.i1 * p.Wg + gx_c.i0],
                                    pf[gy_c.i1 * p.Wg + gx_c.i1], gy_c, gx_c);
                        vb = bilerp(pb[gy_c.i0 * p.Wg + gx_c.i0], pb[gy_c.i0 * p.Wg + gx_c.i1], pb[gy_c.i1 * p.Wg + gx_c.i0],
                                    pb[gy_c.i1 * p.Wg + gx_c.i1], gy_c, gx_c);
                    }
                    v[k] = __fadd_rn(__fmul_rn(wa, va), __fmul_rn(wb, vb));
                }
            }
            emit(f);
        }
    }
}

int launch_seg_tail(const SegTailParams& p, hipStream_t s) {
    FS_REQUIRE(p.K >= 1 && p.K <= 32, "seg_tail: K=%d out of range (1..32)", p.K);
    FS_REQUIRE(p.n >= 1, "seg_tail: n must be >= 1");
    FS_REQUIRE(p.out_logits || p.out_mask || p.canvas, "seg_tail: no output requested");
    FS_REQUIRE((int64_t)p.H * p.W < ((int64_t)1 << 30), "seg_tail: frames of at most 2^30 pixels");
    FS_REQUIRE(!p.canvas || (p.count && p.y0 >= 0 && p.x0 >= 0 && p.y0 + p.H <= p.cH && p.x0 + p.W <= p.cW), "seg_tail: crop outside the canvas");
    const float sy_lo = resize_scale(p.h, p.H, 1), sx_lo = resize_scale(p.w, p.W, 1);
    float sy_g = 0.f, sx_g = 0.f;
    const bool warp = p.lo_next && !p.no_warp && p.n > 1;
    if (warp) {
        FS_REQUIRE(p.scratch && p.grids_left && p.grids_right, "seg_tail: warp mode needs grids and scratch");
        sy_g = resize_scale(p.Hg, p.H, 1);
        sx_g = resize_scale(p.Wg, p.W, 1);
        const int G = p.Hg * p.Wg;
        const size_t map = (size_t)p.K * G;
        float* fwd = p.scratch;
        float* bwd = p.scratch + (size_t)(p.n - 1) * map;
        for (int j = 0; j < p.n - 1; ++j) {
            hipLaunchKernelGGL(seg_warp_step_kernel, dim3(cdiv(G, 256), 2), dim3(256), 0, s, p.lo_prev, p.lo_next,
                               j ? fwd + (size_t)(j - 1) * map : nullptr, j ? bwd + (size_t)(j - 1) * map : nullptr,
                               p.grids_left[j], p.grids_right[j], fwd + (size_t)j * map, bwd + (size_t)j * map, p.K, p.h, p.w,
                               p.H, p.W, p.Hg, p.Wg, j == 0 ? 1 : 0, sy_lo, sx_lo);
        }
    }
    const dim3 grid((unsigned)cdiv(p.W, 256), (unsigned)std::min(p.H, 65535)), block(256);
    const bool canvas = p.canvas != nullptr;
#define FS_SEG_FUSE(KM_, W_, C_) hipLaunchKernelGGL((seg_fuse_kernel<KM_, W_, C_>), grid, block, 0, s, p, sy_lo, sx_lo, sy_g, sx_g)
    if (p.K <= 8) {
        if (warp) { if (canvas) FS_SEG_FUSE(8, true, true); else FS_SEG_FUSE(8, true, false); }
        else { if (canvas) FS_SEG_FUSE(8, false, true); else FS_SEG_FUSE(8, false, false); }
    } else {
        if (warp) { if (canvas) FS_SEG_FUSE(32, true, true); else FS_SEG_FUSE(32, true, false); }
        else { if (canvas) FS_SEG_FUSE(32, false, true); else FS_SEG_FUSE(32, false, false); }
    }
#undef FS_SEG_FUSE
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ fused predict_feature tail (flow/model.py:131-171)
// Everything FlowModel.predict_feature does between the encoder and the ONE batched decoder call, on NHWC feature maps:
//   warp chains   w_1 = warp(f, g_0), w_j = warp(w_{j-1}, g_{j-1})  kept at GRID resolution [Hg][Wg][C]         (:135-151)
//   stack[0]    = up(grid_sample(f, default grid [H0][W0], align_corners=True))                                 (:154-159)
//   stack[p]    = (n-p)/n * up(fwd[p-1]) + p/n * up(bwd[n-p-1]),  up = bilinear align_corners=True to fh x fw   (:166-171)
// The reference (and the op-by-op route) materialises the eight upsampled maps and the H0 x W0 resample; here a step of the two
// chains is one launch and two launches (key map; maps 1..n-1) write the decoder's batch straight from the low-resolution chains.
// Same operations in the same order per element (gs_combine, bilerp, separate mul / add roundings): bit-identical to
// fs_grid_sample_nhwc -> fs_resize_bilinear_nhwc -> fs_blend.
__global__ __launch_bounds__(256) void feat_warp_step_kernel(const float* __restrict__ src_f, const float* __restrict__ src_b, int Hs, int Ws,
                                                             const float* __restrict__ grid_f, const float* __restrict__ grid_b,
                                                             float* __restrict__ dst_f, float* __restrict__ dst_b, int C4, unsigned total, int S) {
    const int dir = blockIdx.y;
    const float* __restrict__ src = dir ? src_b : src_f;
    const float* __restrict__ grid = dir ? grid_b : grid_f;
    float* __restrict__ dst = dir ? dst_b : dst_f;
    const int ld = C4 * 4;
    // channel slab per XCD, cells in raster order (see the map kernels below): the four taps of neighbouring cells meet in one L2
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned e = (blockIdx.x >> 3) * 256u + threadIdx.x;
    const unsigned m = e / (unsigned)S;
    const int c4 = (int)(xcd * (unsigned)S + (e - m * (unsigned)S));
    if (m >= total || c4 >= C4) return;   // total = cells of the grid
    const float gx = grid[m * 2 + 0], gy = grid[m * 2 + 1];
    const GsTaps t = gs_taps(gx, gy, Ws, Hs, 0);
    *reinterpret_cast<f32x4*>(dst + (size_t)m * ld + c4 * 4) = gs_gather_nhwc(src + c4 * 4, Ws, ld, t);
}

struct FeatFuseArgs {
    const float* f_prev;   // [fh][fw][C]
    const float* f_next;   // [fh][fw][C] (no_warp) or nullptr
    const float* chains;   // warp: [2][n-1][Hg][Wg][C] forward maps then backward maps
    const float* grid0;    // warp: [H0][W0][2]
    float* stack;          // [nmaps][fh][fw][C]
    int C4, fh, fw, Hg, Wg, H0, W0, n;
    int same_g, same_0;    // the chain / identity maps already have the feature size: the reference skips the resize (:138, :158)
    float sy_g, sx_g, sy_0, sx_0;
};

// Two launches write every map of the decoder's batch.  What bounds them is not HBM but the CUs' vector-memory path: with one float4 of
// one pixel per thread a warped map costs 8 tap loads + 1 store per output (6 GB through the L1s for 0.66 GB written: 430-440 us
// measured, with or without L2 locality).  So:
//   * maps 1..n-1 (feat_fuse_warp_kernel): a thread owns ONE float4 of channels and a RUN of FEAT_RUN consecutive output pixels of a
//     row; the two low-resolution columns it interpolates between stay in registers and shift as the run advances (an upsample by
//     90 / 44 moves on by one column every other pixel): ~2.4 loads per output instead of 8;
//   * map 0 (feat_fuse_key_kernel): the key-frame map through the H0 x W0 default grid (4 cells x 4 taps per output), the cells'
//     coordinate arithmetic shared by four float4s of channels per thread;
//   * XCD-aware split in both parts: XCD x = blockIdx % 8 owns the channel slab [x * S, (x + 1) * S) float4s of every pixel and walks
//     the pixels in raster order, so all readers of a tap share one L2 and follow each other closely.
// Same loads, same operations in the same order per element as the op-by-op route: bit-identical.
constexpr int FEAT_RUN = 10;   // (3 .. 30 measured within 4 % of each other: profiles/r06_experiments.txt section 1)

// One cell of the default grid as the key-map kernel uses it: the four taps as element offsets into the key frame's map, the four
// weights, and which taps lie inside the image -- computed ONCE per thread and applied to FEAT_KEY_CH float4s of channels (the
// coordinate arithmetic of four cells is ~180 vector instructions; per float4 of channels it was three quarters of the kernel's work).
struct GsCell {
    unsigned nw, ne, sw, se;
    float wnw, wne, wsw, wse;
    bool all_in;
    bool x1ok, y1ok;
};

__device__ __forceinline__ GsCell feat_cell(const FeatFuseArgs& a, int gy, int gx) {
    const size_t m = (size_t)gy * a.W0 + gx;
    const float2 g = *reinterpret_cast<const float2*>(a.grid0 + m * 2);
    const GsTaps t = gs_taps(g.x, g.y, a.fw, a.fh, 1);
    const int x1 = t.x1ok ? t.x0 + 1 : t.x0, y1 = t.y1ok ? t.y0 + 1 : t.y0;  // clamped: every address is valid, outside taps are zeroed
    const unsigned ld = (unsigned)a.C4 * 4u;
    GsCell c;
    c.nw = (unsigned)(t.y0 * a.fw + t.x0) * ld;
    c.ne = (unsigned)(t.y0 * a.fw + x1) * ld;
    c.sw = (unsigned)(y1 * a.fw + t.x0) * ld;
    c.se = (unsigned)(y1 * a.fw + x1) * ld;
    c.wnw = t.nw; c.wne = t.ne; c.wsw = t.sw; c.wse = t.se;
    c.x1ok = t.x1ok; c.y1ok = t.y1ok;
    c.all_in = t.x1ok && t.y1ok;
    return c;
}

// gs_combine over four channels as channel pairs on the packed fp32 instructions (same products, same order of additions)
__device__ __forceinline__ f32x4 feat_cell_value(const float* __restrict__ cbase, const GsCell& c) {
    const f32x4 vnw = *reinterpret_cast<const f32x4*>(cbase + c.nw);
    f32x4 vne = *reinterpret_cast<const f32x4*>(cbase + c.ne);
    f32x4 vsw = *reinterpret_cast<const f32x4*>(cbase + c.sw);
    f32x4 vse = *reinterpret_cast<const f32x4*>(cbase + c.se);
    if (__builtin_amdgcn_ballot_w64(!c.all_in) != 0) {  // a tap beyond the last row / column (whole waves skip this: border cells only)
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        vne = c.x1ok ? vne : z;
        vsw = c.y1ok ? vsw : z;
        vse = c.all_in ? vse : z;
    }
    const f32x2 wnw = {c.wnw, c.wnw}, wne = {c.wne, c.wne}, wsw = {c.wsw, c.wsw}, wse = {c.wse, c.wse};
    f32x4 r;
    r.lo = ((vnw.lo * wnw + vne.lo * wne) + vsw.lo * wsw) + vse.lo * wse;
    r.hi = ((vnw.hi * wnw + vne.hi * wne) + vsw.hi * wsw) + vse.hi * wse;
    return r;
}

constexpr int FEAT_KEY_CH = 4;   // float4s of channels per thread of the key-map kernel

__global__ __launch_bounds__(256) void feat_fuse_key_kernel(FeatFuseArgs a, unsigned npix, int S, int LP) {
    const int ld = a.C4 * 4;
    // ---- map 0: LP = ceil(S / FEAT_KEY_CH) lanes per pixel inside the XCD's slab; lane l owns float4s l, l + LP, l + 2 LP, ...
    const unsigned xcd = blockIdx.x & 7u;
    const unsigned e = (blockIdx.x >> 3) * 256u + threadIdx.x;
    const unsigned m = e / (unsigned)LP;
    const int l = (int)(e - m * (unsigned)LP);
    if (m >= npix) return;
    const int oy = (int)(m / (unsigned)a.fw), ox = (int)(m - (unsigned)oy * (unsigned)a.fw);
    LinCoord cy, cx;
    if (a.same_0) {
        cy.i0 = cy.i1 = oy; cx.i0 = cx.i1 = ox;
        cy.w0 = cx.w0 = 1.f; cy.w1 = cx.w1 = 0.f;
    } else {
        cy = lin_coord(oy, a.H0, a.sy_0, 1);
        cx = lin_coord(ox, a.W0, a.sx_0, 1);
    }
    const GsCell c00 = feat_cell(a, cy.i0, cx.i0);
    GsCell c01 = c00, c10 = c00, c11 = c00;
    if (!a.same_0) {
        c01 = feat_cell(a, cy.i0, cx.i1);
        c10 = feat_cell(a, cy.i1, cx.i0);
        c11 = feat_cell(a, cy.i1, cx.i1);
    }
    const f32x2 wx0 = {cx.w0, cx.w0}, wx1 = {cx.w1, cx.w1}, wy0 = {cy.w0, cy.w0}, wy1 = {cy.w1, cy.w1};
#pragma unroll 1
    for (int k = 0; k < FEAT_KEY_CH; ++k) {
        const int cc = l + k * LP;
        const int c4 = (int)xcd * S + cc;
        if (cc >= S || c4 >= a.C4) break;
        const float* cbase = a.f_prev + c4 * 4;
        f32x4 r;
        if (a.same_0) {
            r = feat_cell_value(cbase, c00);   // the default grid has the feature size: no resize (flow/model.py:158)
        } else {
            const f32x4 v00 = feat_cell_value(cbase, c00), v01 = feat_cell_value(cbase, c01);
            const f32x4 v10 = feat_cell_value(cbase, c10), v11 = feat_cell_value(cbase, c11);
            r.lo = wy0 * (wx0 * v00.lo + wx1 * v01.lo) + wy1 * (wx0 * v10.lo + wx1 * v11.lo);   // bilerp, channel pairs
            r.hi = wy0 * (wx0 * v00.hi + wx1 * v01.hi) + wy1 * (wx0 * v10.hi + wx1 * v11.hi);
        }
        *reinterpret_cast<f32x4*>(a.stack + (size_t)m * ld + c4 * 4) = r;
    }
}

// ---- maps 1..n-1: (map, row, run) per thread, S float4 lanes of the XCD's slab side by side.  Its own kernel: 6 waves per SIMD (the
// key-map kernel holds 16 taps in registers), since what this loop waits for is the latency of its 4-load groups.
__global__ __launch_bounds__(256, 6) void feat_fuse_warp_kernel(FeatFuseArgs a, int S, int nruns) {
    const int ld = a.C4 * 4;
    const size_t map = (size_t)a.fh * a.fw * ld;
    const unsigned bid = blockIdx.x;
    const unsigned xcd = bid & 7u;
    const unsigned e = (bid >> 3) * 256u + threadIdx.x;
    const unsigned item = e / (unsigned)S;
    const int c4 = (int)(xcd * (unsigned)S + (e - item * (unsigned)S));
    const unsigned rowid = item / (unsigned)nruns;
    const int run = (int)(item - rowid * (unsigned)nruns);
    const int p = 1 + (int)(rowid / (unsigned)a.fh);
    const int oy = (int)(rowid - (unsigned)(p - 1) * (unsigned)a.fh);
    if (p >= a.n || c4 >= a.C4) return;
    const float wa = (float)((double)(a.n - p) / (double)a.n);
    const float wb = (float)((double)p / (double)a.n);
    const size_t cmap = (size_t)a.Hg * a.Wg * ld;
    const float* pf = a.chains + (size_t)(p - 1) * cmap + c4 * 4;                  // forward map p-1
    const float* pb = a.chains + (size_t)(a.n - 1 + a.n - p - 1) * cmap + c4 * 4;  // backward map n-p-1
    float* dst = a.stack + (size_t)p * map + ((size_t)oy * a.fw) * ld + c4 * 4;
    const int x0 = run * FEAT_RUN, x1 = min(a.fw, x0 + FEAT_RUN);
    if (a.same_g) {
        for (int ox = x0; ox < x1; ++ox) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(pf + ((size_t)oy * a.Wg + ox) * ld);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(pb + ((size_t)oy * a.Wg + ox) * ld);
            f32x4 r;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = __fadd_rn(__fmul_rn(wa, va[q]), __fmul_rn(wb, vb[q]));
            *reinterpret_cast<f32x4*>(dst + (size_t)ox * ld) = r;
        }
        return;
    }
    const LinCoord cy = lin_coord(oy, a.Hg, a.sy_g, 1);
    const float* f0 = pf + (size_t)cy.i0 * a.Wg * ld;
    const float* f1 = pf + (size_t)cy.i1 * a.Wg * ld;
    const float* b0 = pb + (size_t)cy.i0 * a.Wg * ld;
    const float* b1 = pb + (size_t)cy.i1 * a.Wg * ld;
    // Two register sets A / B hold the two low-resolution columns an output interpolates between (rows i0 / i1 of the forward and of
    // the backward map).  When the run moves on by a column only the set that fell behind is reloaded and the two sets swap roles --
    // no register moves: a + b == b + a bit for bit, so wx_left * left + wx_right * right is evaluated as wxA * A + wxB * B with the
    // weights following the roles.  Channel pairs go through the packed fp32 instructions (v_pk_mul_f32 / v_pk_add_f32: separate
    // roundings, like the scalar form).
    int colA = -1, colB = -1;
    f32x4 Af0, Af1, Ab0, Ab1, Bf0, Bf1, Bb0, Bb1;
    Af0 = Af1 = Ab0 = Ab1 = Bf0 = Bf1 = Bb0 = Bb1 = f32x4{0.f, 0.f, 0.f, 0.f};
#define FS_FEAT_LOAD(S_, col_)                                        \
    do {                                                              \
        const size_t o_ = (size_t)(col_) * ld;                        \
        S_##f0 = *reinterpret_cast<const f32x4*>(f0 + o_);            \
        S_##f1 = *reinterpret_cast<const f32x4*>(f1 + o_);            \
        S_##b0 = *reinterpret_cast<const f32x4*>(b0 + o_);            \
        S_##b1 = *reinterpret_cast<const f32x4*>(b1 + o_);            \
    } while (0)
    const f32x2 wy0 = {cy.w0, cy.w0}, wy1 = {cy.w1, cy.w1}, wa2 = {wa, wa}, wb2 = {wb, wb};
    for (int ox = x0; ox < x1; ++ox) {
        const LinCoord cx = lin_coord(ox, a.Wg, a.sx_g, 1);
        if (!((colA == cx.i0 && colB == cx.i1) || (colA == cx.i1 && colB == cx.i0))) {
            if (colA == cx.i0) { FS_FEAT_LOAD(B, cx.i1); colB = cx.i1; }
            else if (colB == cx.i0) { FS_FEAT_LOAD(A, cx.i1); colA = cx.i1; }
            else if (colA == cx.i1) { FS_FEAT_LOAD(B, cx.i0); colB = cx.i0; }
            else if (colB == cx.i1) { FS_FEAT_LOAD(A, cx.i0); colA = cx.i0; }
            else { FS_FEAT_LOAD(A, cx.i0); FS_FEAT_LOAD(B, cx.i1); colA = cx.i0; colB = cx.i1; }
        }
        const bool a_left = (colA == cx.i0 && colB == cx.i1);
        const float wA = a_left ? cx.w0 : cx.w1, wB = a_left ? cx.w1 : cx.w0;
        const f32x2 wxA = {wA, wA}, wxB = {wB, wB};
        f32x4 r;
#define FS_FEAT_HALF(h_)                                                                      \
    do {                                                                                      \
        const f32x2 tf = wxA * Af0.h_ + wxB * Bf0.h_, bf = wxA * Af1.h_ + wxB * Bf1.h_;       \
        const f32x2 tb = wxA * Ab0.h_ + wxB * Bb0.h_, bb = wxA * Ab1.h_ + wxB * Bb1.h_;       \
        const f32x2 va = wy0 * tf + wy1 * bf, vb = wy0 * tb + wy1 * bb;                       \
        r.h_ = wa2 * va + wb2 * vb;                                                           \
    } while (0)
        FS_FEAT_HALF(lo);
        FS_FEAT_HALF(hi);
#undef FS_FEAT_HALF
        *reinterpret_cast<f32x4*>(dst + (size_t)ox * ld) = r;
    }
#undef FS_FEAT_LOAD
}

// no_warp: every map is a blend of the two key-frame maps -- each float4 of f_prev / f_next is read ONCE and all n maps written from it
// (a launch per map, or a map per blockIdx.y, re-reads 2 x 133 MB per map: 1.9 GB moved for 0.66 GB written).
__global__ __launch_bounds__(256) void feat_fuse_nowarp_kernel(const float* __restrict__ f_prev, const float* __restrict__ f_next,
                                                               float* __restrict__ stack, unsigned total4, int n, int nmaps) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total4) return;
    const f32x4 x = reinterpret_cast<const f32x4*>(f_prev)[i];
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = __fmul_rn(1.f, x[q]);  // the op-by-op route's copy is fs_blend(f, 1.0)
    reinterpret_cast<f32x4*>(stack)[i] = r;
    if (nmaps == 1) return;
    const f32x4 y = reinterpret_cast<const f32x4*>(f_next)[i];
    for (int p = 1; p < n; ++p) {
        const float wa = (float)((double)(n - p) / (double)n);
        const float wb = (float)((double)p / (double)n);
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __fadd_rn(__fmul_rn(wa, x[q]), __fmul_rn(wb, y[q]));
        reinterpret_cast<f32x4*>(stack)[(size_t)p * total4 + i] = r;
    }
}

int launch_feat_tail(const FeatTailParams& p, hipStream_t s) {
    FS_REQUIRE(p.f_prev && p.stack && p.C >= 4 && p.C % 4 == 0 && p.fh >= 1 && p.fw >= 1, "feat_tail: bad arguments (C %% 4 == 0, NHWC maps)");
    FS_REQUIRE(p.n >= 1, "feat_tail: n must be >= 1");
    FS_REQUIRE((((uintptr_t)p.f_prev | (uintptr_t)p.f_next | (uintptr_t)p.stack | (uintptr_t)p.scratch) & 15) == 0, "feat_tail: maps must be 16-byte aligned");
    const int C4 = p.C / 4;
    const int64_t total = (int64_t)p.fh * p.fw * C4;
    FS_REQUIRE(total < ((int64_t)1 << 31), "feat_tail: feature map of at most 2^31 float4 elements");
    const bool warp = !p.no_warp;
    const int nmaps = p.f_next ? p.n : 1;
    FeatFuseArgs a{};
    a.f_prev = p.f_prev;
    a.f_next = p.f_next;
    a.stack = p.stack;
    a.C4 = C4;
    a.fh = p.fh;
    a.fw = p.fw;
    a.n = p.n;
    a.Hg = a.Wg = a.H0 = a.W0 = 1;
    if (warp) {
        FS_REQUIRE(p.grid0 && p.H0 >= 1 && p.W0 >= 1, "feat_tail: warp mode needs the default grid");
        a.grid0 = p.grid0;
        a.H0 = p.H0;
        a.W0 = p.W0;
        a.same_0 = (p.H0 == p.fh && p.W0 == p.fw);
        a.sy_0 = resize_scale(p.H0, p.fh, 1);
        a.sx_0 = resize_scale(p.W0, p.fw, 1);
        if (nmaps > 1) {
            FS_REQUIRE(p.scratch && p.grids_left && p.grids_right && p.Hg >= 1 && p.Wg >= 1, "feat_tail: warp mode needs grids and scratch");
            const int64_t gtotal = (int64_t)p.Hg * p.Wg * C4;
            FS_REQUIRE(gtotal < ((int64_t)1 << 31), "feat_tail: grid too large");
            const size_t cmap = (size_t)p.Hg * p.Wg * p.C;
            float* fwd = p.scratch;
            float* bwd = p.scratch + (size_t)(p.n - 1) * cmap;
            const int Sg = cdiv(C4, 8);
            const unsigned blocks = 8u * (unsigned)cdiv64((int64_t)p.Hg * p.Wg * Sg, 256);
            for (int j = 0; j < p.n - 1; ++j) {
                hipLaunchKernelGGL(feat_warp_step_kernel, dim3(blocks, 2), dim3(256), 0, s, j ? fwd + (size_t)(j - 1) * cmap : p.f_prev,
                                   j ? bwd + (size_t)(j - 1) * cmap : p.f_next, j ? p.Hg : p.fh, j ? p.Wg : p.fw, p.grids_left[j], p.grids_right[j],
                                   fwd + (size_t)j * cmap, bwd + (size_t)j * cmap, C4, (unsigned)(p.Hg * p.Wg), Sg);
            }
            a.chains = p.scratch;
            a.Hg = p.Hg;
            a.Wg = p.Wg;
            a.same_g = (p.Hg == p.fh && p.Wg == p.fw);
            a.sy_g = resize_scale(p.Hg, p.fh, 1);
            a.sx_g = resize_scale(p.Wg, p.fw, 1);
        }
    }
    if (!warp) {
        hipLaunchKernelGGL(feat_fuse_nowarp_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, p.f_prev, p.f_next, p.stack, (unsigned)total, p.n,
                           nmaps);
        FS_HIP(hipGetLastError());
        return 0;
    }
    const int S = cdiv(C4, 8);  // float4s of a pixel per XCD
    const int64_t npix = (int64_t)p.fh * p.fw;
    const int nruns = cdiv(p.fw, FEAT_RUN);
    const int64_t items = (int64_t)(nmaps - 1) * p.fh * nruns;  // (map, row, run) triples of maps 1..n-1
    FS_REQUIRE(npix * S < ((int64_t)1 << 28) && items * S < ((int64_t)1 << 28), "feat_tail: feature map too large");
    const unsigned blocks1 = 8u * (unsigned)cdiv64(items * S, 256);
    const int LP = cdiv(S, FEAT_KEY_CH);
    const unsigned blocks0 = 8u * (unsigned)cdiv64(npix * LP, 256);
    hipLaunchKernelGGL(feat_fuse_key_kernel, dim3(blocks0), dim3(256), 0, s, a, (unsigned)npix, S, LP);
    if (blocks1) hipLaunchKernelGGL(feat_fuse_warp_kernel, dim3(blocks1), dim3(256), 0, s, a, S, nruns);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ sliding crops, all crops of a window in one pass
// compute_output (flow/base.py:182-209) after the network: per crop the warp chains at grid resolution (one launch per step for
// ALL crops: blockIdx.z = crop), then ONE launch over the pixels of the full frame: every canvas pixel is produced once from the
// <= 4 crops that cover it, in the reference's crop order -- tail (upsample / warp / blend) -> fp32 softmax over K -> float64
// sum -> / count -> (optional) argmax.  The same operations in the same order as fs_seg_tail_accumulate per crop followed by
// fs_canvas_finish, hence bit-identical to them, without the 8 x 203 MB float64 read-modify-writes of the canvas (and without
// the canvas at all when only the masks are wanted).
__global__ __launch_bounds__(256) void seg_warp_step_crops_kernel(const float* __restrict__ lo_prev, const float* __restrict__ lo_next,
                                                                  const float* __restrict__ grids, float* __restrict__ scratch, int step, int n,
                                                                  int K, int h, int w, int H, int W, int Hg, int Wg, float sy, float sx) {
    const int dir = blockIdx.y, c = blockIdx.z;
    const int G = Hg * Wg;
    const size_t map = (size_t)K * G;
    const float* lo = (dir ? lo_next : lo_prev) + (size_t)c * K * h * w;
    float* base = scratch + (size_t)c * 2 * (n - 1) * map + (size_t)dir * (n - 1) * map;  // [crop][dir][step][K][Hg][Wg]
    const float* src = step ? base + (size_t)(step - 1) * map : nullptr;
    float* dst = base + (size_t)step * map;
    const float* grid = grids + ((size_t)c * 2 * (n - 1) + (size_t)dir * (n - 1) + step) * G * 2;
    const int first = step == 0;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < G; g += gridDim.x * 256) {
        const float gx = grid[g * 2 + 0], gy = grid[g * 2 + 1];
        const int Hs = first ? H : Hg, Ws = first ? W : Wg;
        const GsTaps t = gs_taps(gx, gy, Ws, Hs, 0);
        const int x1 = t.x1ok ? t.x0 + 1 : t.x0, y1 = t.y1ok ? t.y0 + 1 : t.y0;
        for (int k = 0; k < K; ++k) {
            float vnw, vne, vsw, vse;
            if (first) {
                const float* pl = lo + (size_t)k * h * w;
                vnw = up_at(pl, h, w, t.y0, t.x0, sy, sx);
                vne = t.x1ok ? up_at(pl, h, w, t.y0, x1, sy, sx) : 0.f;
                vsw = t.y1ok ? up_at(pl, h, w, y1, t.x0, sy, sx) : 0.f;
                vse = (t.x1ok && t.y1ok) ? up_at(pl, h, w, y1, x1, sy, sx) : 0.f;
            } else {
                const float* pl = src + (size_t)k * G;
                vnw = pl[t.y0 * Ws + t.x0];
                vne = t.x1ok ? pl[t.y0 * Ws + x1] : 0.f;
                vsw = t.y1ok ? pl[y1 * Ws + t.x0] : 0.f;
                vse = (t.x1ok && t.y1ok) ? pl[y1 * Ws + x1] : 0.f;
            }
            dst[(size_t)k * G + g] = gs_combine(vnw, vne, vsw, vse, t);
        }
    }
}

template <int KMAX, int FN>
__global__ __launch_bounds__(256) void crops_fuse_kernel(CropsFuseParams p) {
    const int64_t HW = (int64_t)p.H * p.W;
    const int K = p.K, n = p.n, G = p.Hg * p.Wg;
    const size_t lo_stride = (size_t)K * p.h * p.w, map = (size_t)K * G;
    for (int Y = blockIdx.y; Y < p.H; Y += gridDim.y) {  // grid = (x blocks of 256 pixels, rows): no 64-bit division per pixel
        const int X = blockIdx.x * 256 + threadIdx.x;
        if (X >= p.W) break;
        const int64_t i = (int64_t)Y * p.W + X;
        for (int f0 = 0; f0 < n; f0 += FN) {  // FN frames at a time: FN * K float64 sums per pixel stay in registers
            double acc[FN][KMAX];
#pragma unroll
            for (int f = 0; f < FN; ++f)
#pragma unroll
                for (int k = 0; k < KMAX; ++k) acc[f][k] = 0.0;
            int cnt = 0;
            for (int c = 0; c < p.nc; ++c) {  // the reference's crop order (flow/base.py:192-205): the float64 sums depend on it
                const int y = Y - p.cy[c], x = X - p.cx[c];
                if ((unsigned)y >= (unsigned)p.ch || (unsigned)x >= (unsigned)p.cw) continue;
                ++cnt;
                const float* lp = p.lo_prev + (size_t)c * lo_stride;
                const float* ln = p.lo_next ? p.lo_next + (size_t)c * lo_stride : nullptr;
                const float* sc = p.scratch + (size_t)c * 2 * (n - 1) * map;
                float a[KMAX], b[KMAX];
                const LinCoord cy = lin_coord(y, p.h, p.sy_lo, 1), cx = lin_coord(x, p.w, p.sx_lo, 1);
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    if (k < K) {
                        const float* pl = lp + (size_t)k * p.h * p.w;
                        a[k] = bilerp(pl[cy.i0 * p.w + cx.i0], pl[cy.i0 * p.w + cx.i1], pl[cy.i1 * p.w + cx.i0], pl[cy.i1 * p.w + cx.i1], cy, cx);
                        if (ln && p.no_warp) {
                            const float* pn = ln + (size_t)k * p.h * p.w;
                            b[k] = bilerp(pn[cy.i0 * p.w + cx.i0], pn[cy.i0 * p.w + cx.i1], pn[cy.i1 * p.w + cx.i0], pn[cy.i1 * p.w + cx.i1], cy, cx);
                        }
                    }
                }
                LinCoord gy_c, gx_c;
                if (!p.no_warp) {
                    gy_c = lin_coord(y, p.Hg, p.sy_g, 1);
                    gx_c = lin_coord(x, p.Wg, p.sx_g, 1);
                }
#pragma unroll
                for (int ff = 0; ff < FN; ++ff) {
                    const int f = f0 + ff;
                    if (f >= n) break;
                    float v[KMAX];
                    const float wa = (float)((double)(n - f) / (double)n);
                    const float wb = (float)((double)f / (double)n);
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        if (k < K) {
                            if (f == 0) {
                                v[k] = a[k];
                            } else {
                                float va, vb;
                                if (p.no_warp) {
                                    va = a[k];
                                    vb = b[k];
                                } else {
                                    const float* pf = sc + ((size_t)(f - 1) * K + k) * G;
                                    const float* pb = sc + ((size_t)(n - 1) * K + (size_t)(n - f - 1) * K + k) * G;
                                    va = bilerp(pf[gy_c.i0 * p.Wg + gx_c.i0], pf[gy_c.i0 * p.Wg + gx_c.i1], pf[gy_c.i1 * p.Wg + gx_c.i0],
                                                pf[gy_c.i1 * p.Wg + gx_c.i1], gy_c, gx_c);
                                    vb = bilerp(pb[gy_c.i0 * p.Wg + gx_c.i0], pb[gy_c.i0 * p.Wg + gx_c.i1], pb[gy_c.i1 * p.Wg + gx_c.i0],
                                                pb[gy_c.i1 * p.Wg + gx_c.i1], gy_c, gx_c);
                                }
                                v[k] = __fadd_rn(__fmul_rn(wa, va), __fmul_rn(wb, vb));
                            }
                        }
                    }
                    float mx = v[0];
#pragma unroll
                    for (int k = 1; k < KMAX; ++k)
                        if (k < K) mx = fmaxf(mx, v[k]);
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < K) sum += expf(v[k] - mx);
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < K) acc[ff][k] += (double)(expf(v[k] - mx) / sum);
                }
            }
            const double count = (double)cnt;
#pragma unroll
            for (int ff = 0; ff < FN; ++ff) {
                const int f = f0 + ff;
                if (f >= n) break;
                double best = -1.0;
                int arg = 0;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    if (k < K) {
                        const double val = acc[ff][k] / count;  // flow/base.py:208
                        if (p.canvas) p.canvas[((size_t)f * K + k) * HW + i] = val;
                        if (val > best) { best = val; arg = k; }
                    }
                }
                if (p.mask) p.mask[(size_t)f * HW + i] = (uint8_t)arg;
            }
        }
    }
}

int launch_crops_fuse(CropsFuseParams p, const float* grids, float* scratch, hipStream_t s) {
    FS_REQUIRE(p.K >= 1 && p.K <= 8, "crops_fuse: K=%d out of range (1..8)", p.K);
    FS_REQUIRE(p.nc >= 1 && p.nc <= 64 && p.n >= 1, "crops_fuse: 1..64 crops, n >= 1");
    FS_REQUIRE(p.canvas || p.mask, "crops_fuse: no output requested");
    for (int c = 0; c < p.nc; ++c)
        FS_REQUIRE(p.cy[c] >= 0 && p.cx[c] >= 0 && p.cy[c] + p.ch <= p.H && p.cx[c] + p.cw <= p.W, "crops_fuse: crop %d outside the canvas", c);
    p.sy_lo = resize_scale(p.h, p.ch, 1);
    p.sx_lo = resize_scale(p.w, p.cw, 1);
    p.sy_g = p.sx_g = 0.f;
    p.scratch = scratch;
    const bool warp = p.lo_next && !p.no_warp && p.n > 1;
    if (!p.lo_next) p.n = 1;
    if (warp) {
        FS_REQUIRE(scratch && grids, "crops_fuse: warp mode needs the crop grids and scratch");
        p.sy_g = resize_scale(p.Hg, p.ch, 1);
        p.sx_g = resize_scale(p.Wg, p.cw, 1);
        const int G = p.Hg * p.Wg;
        for (int j = 0; j < p.n - 1; ++j)
            hipLaunchKernelGGL(seg_warp_step_crops_kernel, dim3(cdiv(G, 256), 2, p.nc), dim3(256), 0, s, p.lo_prev, p.lo_next, grids, scratch, j, p.n,
                               p.K, p.h, p.w, p.ch, p.cw, p.Hg, p.Wg, p.sy_lo, p.sx_lo);
    } else {
        p.no_warp = 1;
    }
    hipLaunchKernelGGL((crops_fuse_kernel<8, 5>), dim3((unsigned)cdiv(p.W, 256), (unsigned)std::min(p.H, 65535)), dim3(256), 0, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ argmax / fused resize+argmax
__global__ __launch_bounds__(256) void argmax_u8_kernel(const float* __restrict__ in, int B, int K, int64_t HW,
                                                        uint8_t* __restrict__ out) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HW, px = i - b * HW;
        const float* p = in + (size_t)b * K * HW + px;
        float best = p[0];
        int arg = 0;
        for (int k = 1; k < K; ++k) {
            const float v = p[(size_t)k * HW];
            if (v > best) { best = v; arg = k; }
        }
        out[i] = (uint8_t)arg;
    }
}

int launch_argmax_u8(const float* in, int B, int K, int64_t HW, uint8_t* out, hipStream_t s) {
    FS_REQUIRE(K >= 1 && K <= 255, "argmax_u8: K out of range");
    const int64_t total = (int64_t)B * HW;
    hipLaunchKernelGGL(argmax_u8_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, in, B, K, HW,
                       out);
    FS_HIP(hipGetLastError());
    return 0;
}

template <typename I>
__global__ __launch_bounds__(256) void resize_argmax_u8_kernel(const float* __restrict__ in, int B, int K, int Hi, int Wi,
                                                               uint8_t* __restrict__ out, int Ho, int Wo, float sy, float sx) {
    const I total = (I)B * Ho * Wo;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const int ox = (int)(i % Wo);
        const int oy = (int)((i / Wo) % Ho);
        const I b = i / ((I)Wo * Ho);
        const LinCoord cy = lin_coord(oy, Hi, sy, 1), cx = lin_coord(ox, Wi, sx, 1);
        float best = -INFINITY;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const float* pl = in + ((size_t)b * K + k) * Hi * Wi;
            const float v = bilerp(pl[(size_t)cy.i0 * Wi + cx.i0], pl[(size_t)cy.i0 * Wi + cx.i1], pl[(size_t)cy.i1 * Wi + cx.i0],
                                   pl[(size_t)cy.i1 * Wi + cx.i1], cy, cx);
            if (v > best) { best = v; arg = k; }
        }
        out[i] = (uint8_t)arg;
    }
}

int launch_resize_argmax_u8(const float* in, int B, int K, int Hi, int Wi, uint8_t* out, int Ho, int Wo, hipStream_t s) {
    FS_REQUIRE(K >= 1 && K <= 255, "resize_argmax_u8: K out of range");
    const int64_t total = (int64_t)B * Ho * Wo;
    if (total < ((int64_t)1 << 31)) hipLaunchKernelGGL((resize_argmax_u8_kernel<unsigned>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, in, B,
                       K, Hi, Wi, out, Ho, Wo, resize_scale(Hi, Ho, 1), resize_scale(Wi, Wo, 1));
    else hipLaunchKernelGGL((resize_argmax_u8_kernel<int64_t>), dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, in, B,
                       K, Hi, Wi, out, Ho, Wo, resize_scale(Hi, Ho, 1), resize_scale(Wi, Wo, 1));
    FS_HIP(hipGetLastError());
    return 0;
}

// F.interpolate(in, (Hfull, Wfull), bilinear, ac)[:, :, :Ho, :Wo] -- the Segmenter's decoder tail (segm/model/segmenter.py:45-46:
// upsample the class masks to the padded frame, then crop the padding) -- as dense logits and / or their channel argmax, in one
// launch: the scale is the padded size's, only the kept pixels are computed, and the crop is never a strided view that something
// downstream has to copy.  Values and tie-breaking are resize_bilinear_nchw's and argmax_u8's.
template <bool LOGITS, bool MASK>
__global__ __launch_bounds__(256) void resize_crop_kernel(const float* __restrict__ in, int B, int K, int Hi, int Wi, int ac, float sy,
                                                          float sx, float* __restrict__ logits, uint8_t* __restrict__ mask, int Ho,
                                                          int Wo) {
    const int64_t HWo = (int64_t)Ho * Wo, total = (int64_t)B * HWo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / HWo, px = i - b * HWo;
        const int oy = (int)(px / Wo), ox = (int)(px - (int64_t)oy * Wo);
        const LinCoord cy = lin_coord(oy, Hi, sy, ac), cx = lin_coord(ox, Wi, sx, ac);
        const int o00 = cy.i0 * Wi + cx.i0, o01 = cy.i0 * Wi + cx.i1, o10 = cy.i1 * Wi + cx.i0, o11 = cy.i1 * Wi + cx.i1;
        float best = 0.f;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const float* pl = in + ((size_t)b * K + k) * Hi * Wi;
            const float v = bilerp(pl[o00], pl[o01], pl[o10], pl[o11], cy, cx);
            if (LOGITS) logits[((size_t)b * K + k) * HWo + px] = v;
            if (MASK && (k == 0 || v > best)) { best = v; arg = k; }
        }
        if (MASK) mask[i] = (uint8_t)arg;
    }
}

int launch_resize_crop(const float* in, int B, int K, int Hi, int Wi, int Hfull, int Wfull, int align_corners, float* logits,
                       uint8_t* mask, int Ho, int Wo, hipStream_t s) {
    FS_REQUIRE(Ho <= Hfull && Wo <= Wfull, "resize_crop: the kept region must lie inside the resized frame");
    FS_REQUIRE((int64_t)Hi * Wi < ((int64_t)1 << 31), "resize_crop: input plane too large");
    FS_REQUIRE(!mask || (K >= 1 && K <= 255), "resize_crop: K out of range for a uint8 mask");
    FS_REQUIRE(logits || mask, "resize_crop: no output requested");
    const int64_t total = (int64_t)B * Ho * Wo;
    const dim3 grid((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384));
    const float sy = resize_scale(Hi, Hfull, align_corners), sx = resize_scale(Wi, Wfull, align_corners);
    if (logits && mask) hipLaunchKernelGGL((resize_crop_kernel<true, true>), grid, dim3(256), 0, s, in, B, K, Hi, Wi, align_corners, sy, sx, logits, mask, Ho, Wo);
    else if (logits) hipLaunchKernelGGL((resize_crop_kernel<true, false>), grid, dim3(256), 0, s, in, B, K, Hi, Wi, align_corners, sy, sx, logits, mask, Ho, Wo);
    else hipLaunchKernelGGL((resize_crop_kernel<false, true>), grid, dim3(256), 0, s, in, B, K, Hi, Wi, align_corners, sy, sx, logits, mask, Ho, Wo);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ sliding-crop accumulation (flow/base.py:182-234)
// canvas[f][k][y0+y][x0+x] += softmax_k(logits[f][:, y, x]);  count[y0+y][x0+x] += 1   (float64 canvas, :190-191)
__global__ __launch_bounds__(256) void softmax_accumulate_kernel(const float* __restrict__ logits, int n, int K, int h, int w,
                                                                 double* __restrict__ canvas, double* __restrict__ count, int H, int W,
                                                                 int y0, int x0) {
    const int64_t total = (int64_t)n * h * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), f = (int)(i / ((int64_t)w * h));
        const float* p = logits + ((size_t)f * K) * h * w + (size_t)y * w + x;
        float mx = p[0];
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, p[(size_t)k * h * w]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(p[(size_t)k * h * w] - mx);
        const size_t pix = (size_t)(y0 + y) * W + (x0 + x);
        for (int k = 0; k < K; ++k) canvas[((size_t)f * K + k) * H * W + pix] += (double)(expf(p[(size_t)k * h * w] - mx) / sum);
        if (f == 0) count[pix] += 1.0;
    }
}

int launch_softmax_accumulate(const float* logits, int n, int K, int h, int w, double* canvas, double* count, int H, int W, int y0,
                              int x0, hipStream_t s) {
    FS_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W, "softmax_accumulate: crop outside the canvas");
    const int64_t total = (int64_t)n * h * w;
    hipLaunchKernelGGL(softmax_accumulate_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, logits, n,
                       K, h, w, canvas, count, H, W, y0, x0);
    FS_HIP(hipGetLastError());
    return 0;
}

// canvas /= count (flow/base.py:208), optionally the per-frame argmax of the result
__global__ __launch_bounds__(256) void canvas_finish_kernel(double* __restrict__ canvas, const double* __restrict__ count, int n, int K,
                                                            int64_t HW, uint8_t* __restrict__ mask) {
    const int64_t total = (int64_t)n * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i % HW, f = i / HW;
        const double c = count[pix];
        double best = -1.0;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            double* q = canvas + ((size_t)f * K + k) * HW + pix;
            const double v = *q / c;
            *q = v;
            if (v > best) { best = v; arg = k; }
        }
        if (mask) mask[i] = (uint8_t)arg;
    }
}

int launch_canvas_finish(double* canvas, const double* count, int n, int K, int64_t HW, uint8_t* mask, hipStream_t s) {
    const int64_t total = (int64_t)n * HW;
    hipLaunchKernelGGL(canvas_finish_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, canvas, count, n,
                       K, HW, mask);
    FS_HIP(hipGetLastError());
    return 0;
}

// F.interpolate(canvas, (Ho, Wo), bilinear, align_corners=True).max(1)[1] on the float64 crop-averaged probabilities
// (flow/base.py:275-276 after compute_output; ATen's upsample_bilinear2d with accscalar_t = double), without the
// [n,K,Ho,Wo] float64 intermediate.
__global__ __launch_bounds__(256) void canvas_resize_argmax_kernel(const double* __restrict__ canvas, int n, int K, int Hi, int Wi,
                                                                   uint8_t* __restrict__ mask, int Ho, int Wo, double sy, double sx) {
    const int64_t total = (int64_t)n * Ho * Wo;
    const size_t HWi = (size_t)Hi * Wi;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho);
        const int64_t f = i / ((int64_t)Wo * Ho);
        const double h1r = sy * oy, w1r = sx * ox;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < Hi - 1 ? 1 : 0, w1p = w1 < Wi - 1 ? 1 : 0;
        const double h1l = h1r - h1, h0l = 1.0 - h1l, w1l = w1r - w1, w0l = 1.0 - w1l;
        double best = -INFINITY;
        int arg = 0;
        for (int k = 0; k < K; ++k) {
            const double* pl = canvas + ((size_t)f * K + k) * HWi + (size_t)h1 * Wi + w1;
            const double v = h0l * (w0l * pl[0] + w1l * pl[w1p]) + h1l * (w0l * pl[(size_t)h1p * Wi] + w1l * pl[(size_t)h1p * Wi + w1p]);
            if (v > best) { best = v; arg = k; }
        }
        mask[i] = (uint8_t)arg;
    }
}

int launch_canvas_resize_argmax(const double* canvas, int n, int K, int Hi, int Wi, uint8_t* mask, int Ho, int Wo, hipStream_t s) {
    FS_REQUIRE(K >= 1 && K <= 255, "canvas_resize_argmax: K out of range");
    const int64_t total = (int64_t)n * Ho * Wo;
    const double sy = Ho > 1 ? (double)(Hi - 1) / (double)(Ho - 1) : 0.0, sx = Wo > 1 ? (double)(Wi - 1) / (double)(Wo - 1) : 0.0;
    hipLaunchKernelGGL(canvas_resize_argmax_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 16384)), dim3(256), 0, s, canvas,
                       n, K, Hi, Wi, mask, Ho, Wo, sy, sx);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ crop_motion_vector (flow/transform.py:215-261) for all
// crops and all grids of a window in one launch: cut the block range [bho, bho+bh) x [bwo, bwo+bw) of a full-frame grid,
// renormalise the (x, y) coordinates to the crop (fp32, numpy's op order: ((((g + 1) / 2) * size - offset) / den) * 2 - 1) and
// resize to fh x fw with half-pixel-centre bilinear interpolation (cv2.resize INTER_LINEAR on float data).
__global__ __launch_bounds__(256) void crop_grids_kernel(CropGridParams p) {
    const int per = p.fh * p.fw;
    const int total = p.ncrops * p.ngrids * per;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int px = i % per, j = (i / per) % p.ngrids, c = i / (per * p.ngrids);
        const int oy = px / p.fw, ox = px - oy * p.fw;
        const float* g = p.grids[j];
        const int bh = p.bh[c], bw = p.bw[c];
        const float* base = g + ((size_t)p.bho[c] * p.Wg + p.bwo[c]) * 2;
        auto at = [&](int yy, int xx, int comp) -> float {
            const float raw = base[((size_t)yy * p.Wg + xx) * 2 + comp];
            const float size = comp ? (float)p.H : (float)p.W, off = comp ? p.off_h[c] : p.off_w[c], den = comp ? p.den_h[c] : p.den_w[c];
            return __fadd_rn(__fmul_rn(__fdiv_rn(__fadd_rn(__fmul_rn(__fdiv_rn(__fadd_rn(raw, 1.f), 2.f), size), -off), den), 2.f), -1.f);
        };
        float rx, ry;
        if (bh == p.fh && bw == p.fw) {
            rx = at(oy, ox, 0);
            ry = at(oy, ox, 1);
        } else {
            const LinCoord cy = lin_coord(oy, bh, resize_scale(bh, p.fh, 0), 0), cx = lin_coord(ox, bw, resize_scale(bw, p.fw, 0), 0);
            rx = bilerp(at(cy.i0, cx.i0, 0), at(cy.i0, cx.i1, 0), at(cy.i1, cx.i0, 0), at(cy.i1, cx.i1, 0), cy, cx);
            ry = bilerp(at(cy.i0, cx.i0, 1), at(cy.i0, cx.i1, 1), at(cy.i1, cx.i0, 1), at(cy.i1, cx.i1, 1), cy, cx);
        }
        float* o = p.out + ((size_t)(c * p.ng_total + p.g0 + j) * per + px) * 2;  // this call's slice of [crops][all grids][fh][fw][2]
        o[0] = rx;
        o[1] = ry;
    }
}

int launch_crop_grids(const CropGridParams& p, hipStream_t s) {
    FS_REQUIRE(p.ncrops >= 1 && p.ncrops <= 32 && p.ngrids >= 1 && p.ngrids <= 32, "crop_grids: at most 32 crops x 32 grids per call");
    FS_REQUIRE(p.fh >= 1 && p.fw >= 1 && p.out && p.g0 >= 0 && p.g0 + p.ngrids <= p.ng_total, "crop_grids: bad output geometry");
    for (int c = 0; c < p.ncrops; ++c)
        FS_REQUIRE(p.bho[c] >= 0 && p.bwo[c] >= 0 && p.bh[c] >= 1 && p.bw[c] >= 1 && p.bho[c] + p.bh[c] <= p.Hg && p.bwo[c] + p.bw[c] <= p.Wg,
                   "crop_grids: block range of crop %d outside the %dx%d grid", c, p.Hg, p.Wg);
    const int total = p.ncrops * p.ngrids * p.fh * p.fw;
    hipLaunchKernelGGL(crop_grids_kernel, dim3(std::min(cdiv(total, 256), 4096)), dim3(256), 0, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ palette colourise (flow/base.py:308-312: colors[output])
__global__ __launch_bounds__(256) void colorize_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ palette, int K,
                                                       uint8_t* __restrict__ rgb, int64_t numel) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const int c = mask[i] < K ? mask[i] : 0;
        rgb[i * 3 + 0] = palette[c * 3 + 0];
        rgb[i * 3 + 1] = palette[c * 3 + 1];
        rgb[i * 3 + 2] = palette[c * 3 + 2];
    }
}

int launch_colorize(const uint8_t* mask, const uint8_t* palette, int K, uint8_t* rgb, int64_t numel, hipStream_t s) {
    hipLaunchKernelGGL(colorize_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(numel, 256), 16384))), dim3(256), 0, s,
                       mask, palette, K, rgb, numel);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ block motion vectors -> sampling grids
// dataset/flow/extract_motion_vectors.py:21-43.  mv rows: (source, w, h, src_x, src_y, dst_x, dst_y, ...) int32.
// The reference loops over the vectors in order, so for a block hit twice the LAST vector wins: pass 1 records the
// highest vector index per cell (atomicMax), pass 2 fills the cells from their owners; cells nobody hits keep the
// identity grid of flow/model.py:10-21.
__global__ __launch_bounds__(256) void mv_owner_kernel(const int* __restrict__ mv, int n, int stride, int hb, int wb, int bs,
                                                       int* __restrict__ own_fwd, int* __restrict__ own_inv) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int* m = mv + (size_t)i * stride;
        // Python floor division for possibly negative coordinates
        const int sx = (m[3] >= 0 ? m[3] / bs : -((-m[3] + bs - 1) / bs)), sy = (m[4] >= 0 ? m[4] / bs : -((-m[4] + bs - 1) / bs));
        const int dx = (m[5] >= 0 ? m[5] / bs : -((-m[5] + bs - 1) / bs)), dy = (m[6] >= 0 ? m[6] / bs : -((-m[6] + bs - 1) / bs));
        if (dx >= 0 && dx < wb && dy >= 0 && dy < hb) atomicMax(&own_fwd[dy * wb + dx], i);
        if (sx >= 0 && sx < wb && sy >= 0 && sy < hb) atomicMax(&own_inv[sy * wb + sx], i);
    }
}

__global__ __launch_bounds__(256) void mv_fill_kernel(const int* __restrict__ mv, int stride, int hb, int wb, int bs, int H, int W,
                                                      const int* __restrict__ own_fwd, const int* __restrict__ own_inv,
                                                      double* __restrict__ grid, double* __restrict__ inv_grid) {
    const int cells = hb * wb;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cells; c += gridDim.x * 256) {
        const int by = c / wb, bx = c - by * wb;
        // identity (flow/model.py:19-20 with the grid's own 1920x1072 geometry = wb*bs x hb*bs)
        double gx = (double)(bx * bs + bs / 2) / (double)(wb * bs) * 2 - 1, gy = (double)(by * bs + bs / 2) / (double)(hb * bs) * 2 - 1;
        double ix = gx, iy = gy;
        const int of = own_fwd[c], oi = own_inv[c];
        if (of >= 0) {
            const int* m = mv + (size_t)of * stride;
            const int sx = (m[3] >= 0 ? m[3] / bs : -((-m[3] + bs - 1) / bs)), sy = (m[4] >= 0 ? m[4] / bs : -((-m[4] + bs - 1) / bs));
            gx = (double)(sx * bs + bs / 2) / (double)W * 2 - 1;
            gy = (double)(sy * bs + bs / 2) / (double)H * 2 - 1;
        }
        if (oi >= 0) {
            const int* m = mv + (size_t)oi * stride;
            const int dx = (m[5] >= 0 ? m[5] / bs : -((-m[5] + bs - 1) / bs)), dy = (m[6] >= 0 ? m[6] / bs : -((-m[6] + bs - 1) / bs));
            ix = (double)(dx * bs + bs / 2) / (double)W * 2 - 1;
            iy = (double)(dy * bs + bs / 2) / (double)H * 2 - 1;
        }
        grid[c * 2 + 0] = gx;
        grid[c * 2 + 1] = gy;
        inv_grid[c * 2 + 0] = ix;
        inv_grid[c * 2 + 1] = iy;
    }
}

int launch_mv_to_grids(const int* mv, int n, int stride, int hb, int wb, int bs, int H, int W, int* owners /*2*hb*wb*/, double* grid,
                       double* inv_grid, hipStream_t s) {
    FS_REQUIRE(stride >= 7 && hb > 0 && wb > 0 && bs > 0, "mv_to_grids: bad geometry");
    FS_HIP(hipMemsetAsync(owners, 0xFF, (size_t)2 * hb * wb * sizeof(int), s));  // -1
    if (n > 0) {
        hipLaunchKernelGGL(mv_owner_kernel, dim3(std::min(cdiv(n, 256), 1024)), dim3(256), 0, s, mv, n, stride, hb, wb, bs, owners,
                           owners + hb * wb);
        FS_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(mv_fill_kernel, dim3(cdiv(hb * wb, 256)), dim3(256), 0, s, mv, stride, hb, wb, bs, H, W, owners, owners + hb * wb,
                       grid, inv_grid);
    FS_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ IoU histograms (util/util.py:52-63)
// hist[0][k] = |pred==target==k|, hist[1][k] = |pred==k| (after ignore masking), hist[2][k] = |target==k|;
// union = hist[1] + hist[2] - hist[0] is formed by the caller.
__global__ __launch_bounds__(256) void iou_hist_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ target,
                                                       int64_t numel, int K, int ignore, unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned int h[];  // [3][K]
    for (int i = threadIdx.x; i < 3 * K; i += 256) h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const int t = target[i];
        int o = pred[i];
        if (t == ignore) o = ignore;
        if (o < K) {
            atomicAdd(&h[K + o], 1u);
            if (o == t) atomicAdd(&h[o], 1u);
        }
        if (t < K) atomicAdd(&h[2 * K + t], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * K; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

int launch_iou_hist(const uint8_t* pred, const uint8_t* target, int64_t numel, int K, int ignore_index, long long* hist3K,
                    hipStream_t s) {
    FS_REQUIRE(K >= 1 && K <= 255, "iou_hist: K out of range");
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(numel, 256 * 16), 1024));
    hipLaunchKernelGGL(iou_hist_kernel, dim3(grid), dim3(256), 3 * K * sizeof(unsigned int), s, pred, target, numel, K,
                       ignore_index, reinterpret_cast<unsigned long long*>(hist3K));
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
