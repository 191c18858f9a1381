// Device helpers restating the sampling arithmetic of the torch ops the reference calls:
//   F.interpolate(mode='bilinear')            (flow/model.py:42..228, model/pspnet.py:33)
//   F.grid_sample(mode='bilinear', padding_mode='border')   (flow/model.py:157, 248)
// Index/weight formulas follow ATen (area_pixel_compute_scale / compute_source_index /
// grid_sampler_unnormalize + clip_coordinates); multiplications and additions are kept
// un-contracted (__fmul_rn/__fadd_rn) so that the op order of the CPU kernels is reproduced.
#pragma once
#include <hip/hip_runtime.h>

namespace fs {

struct LinCoord {
    int i0, i1;
    float w0, w1;
};

// scale as ATen computes it in float
__host__ __device__ inline float resize_scale(int in, int out, int align_corners) {
    if (align_corners) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    return (float)in / (float)out;
}

__device__ __forceinline__ LinCoord lin_coord(int dst, int in_size, float scale, int align_corners) {
    float src;
    if (align_corners) {
        src = __fmul_rn(scale, (float)dst);
    } else {
        src = __fadd_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), -0.5f);
        if (src < 0.f) src = 0.f;
    }
    int i0 = (int)src;  // src >= 0: truncation == floor
    if (i0 > in_size - 1) i0 = in_size - 1;
    LinCoord c;
    c.i0 = i0;
    c.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    float l1 = __fadd_rn(src, -(float)i0);
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    c.w1 = l1;
    c.w0 = __fadd_rn(1.f, -l1);
    return c;
}

// value = wy0*(wx0*v00 + wx1*v01) + wy1*(wx0*v10 + wx1*v11)
__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, const LinCoord& cy,
                                        const LinCoord& cx) {
    const float top = __fadd_rn(__fmul_rn(cx.w0, v00), __fmul_rn(cx.w1, v01));
    const float bot = __fadd_rn(__fmul_rn(cx.w0, v10), __fmul_rn(cx.w1, v11));
    return __fadd_rn(__fmul_rn(cy.w0, top), __fmul_rn(cy.w1, bot));
}

// grid_sample source coordinate: unnormalise [-1,1] -> pixel space, then clamp to the border.
// Follows ATen's vectorised CPU kernel (GridSamplerKernel.cpp ComputeLocation):
//   align_corners=False: x = (g + 1) * (size / 2) - 0.5 ; True: x = (g + 1) * ((size - 1) / 2)
__device__ __forceinline__ float gs_coord(float g, int size, int align_corners) {
    float x;
    if (align_corners)
        x = __fmul_rn(__fadd_rn(g, 1.f), (float)(size - 1) * 0.5f);
    else
        x = __fadd_rn(__fmul_rn(__fadd_rn(g, 1.f), (float)size * 0.5f), -0.5f);
    x = fminf((float)(size - 1), fmaxf(x, 0.f));  // clip_coordinates (border padding)
    return x;
}

struct GsTaps {
    int x0, y0;        // north-west integer tap
    float nw, ne, sw, se;
    bool x1ok, y1ok;   // south/east taps inside the image
};

// w = x - floor(x), e = 1 - w, n = y - floor(y), s = 1 - n; nw = s*e, ne = s*w, sw = n*e, se = n*w
__device__ __forceinline__ GsTaps gs_taps(float gx, float gy, int W, int H, int align_corners) {
    const float ix = gs_coord(gx, W, align_corners);
    const float iy = gs_coord(gy, H, align_corners);
    const float fx = floorf(ix), fy = floorf(iy);
    GsTaps t;
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const float w = __fadd_rn(ix, -fx), e = __fadd_rn(1.f, -w);
    const float n = __fadd_rn(iy, -fy), s = __fadd_rn(1.f, -n);
    t.nw = __fmul_rn(s, e);
    t.ne = __fmul_rn(s, w);
    t.sw = __fmul_rn(n, e);
    t.se = __fmul_rn(n, w);
    t.x1ok = t.x0 + 1 <= W - 1;
    t.y1ok = t.y0 + 1 <= H - 1;
    return t;
}

// out = nw_val*nw + ne_val*ne + sw_val*sw + se_val*se (out-of-image taps contribute 0)
__device__ __forceinline__ float gs_combine(float vnw, float vne, float vsw, float vse, const GsTaps& t) {
    float r = __fmul_rn(vnw, t.nw);
    r = __fadd_rn(r, __fmul_rn(vne, t.ne));
    r = __fadd_rn(r, __fmul_rn(vsw, t.sw));
    r = __fadd_rn(r, __fmul_rn(vse, t.se));
    return r;
}

}  // namespace fs
