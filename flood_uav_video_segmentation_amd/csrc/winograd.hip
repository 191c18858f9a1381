// Winograd F(4x4, 3x3) transforms (Lavin & Gray matrices) around the grouped fp32-MFMA GEMM.
// Replaces the direct 3x3 stride-1 pad-1 convolution + eval BatchNorm + ReLU of the PSPNet classifier head
// (reference model/pspnet.py:70-73: Conv2d(4096, 512, 3, padding=1, bias=False), BatchNorm2d, ReLU).
// All three kernels are HBM-bound elementwise-style passes over float4 channel groups (NHWC).
#include "kernels.h"

namespace fs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
// The 2-D transforms keep 36 values live per thread: float2 per thread (not float4) keeps that under 128 VGPRs.
typedef f32x2 wv_t;
constexpr int WV = 2;

// B^T d (1-D, 6 -> 6) for F(4,3)
__device__ __forceinline__ void wino_bt(const wv_t d[6], wv_t t[6]) {
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
    t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

// A^T m (1-D, 6 -> 4)
__device__ __forceinline__ void wino_at(const wv_t m[6], wv_t y[4]) {
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
    y[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
    y[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
}

// ---- filter transform U = G g G^T, once at load: thread per (o, c)
__global__ __launch_bounds__(256) void winograd_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int O, int I) {
    const int64_t total = (int64_t)O * I;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float* g = w + i * 9;  // OIHW: [o][c][3][3]
        float t[6][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float g0 = g[0 * 3 + s], g1 = g[1 * 3 + s], g2 = g[2 * 3 + s];
            t[0][s] = g0 * 0.25f;
            t[1][s] = (g0 + g1 + g2) * (-1.f / 6.f);
            t[2][s] = (g0 - g1 + g2) * (-1.f / 6.f);
            t[3][s] = g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
            t[4][s] = g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
            t[5][s] = g2;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const float a0 = t[r][0], a1 = t[r][1], a2 = t[r][2];
            float u[6];
            u[0] = a0 * 0.25f;
            u[1] = (a0 + a1 + a2) * (-1.f / 6.f);
            u[2] = (a0 - a1 + a2) * (-1.f / 6.f);
            u[3] = a0 * (1.f / 24.f) + a1 * (1.f / 12.f) + a2 * (1.f / 6.f);
            u[4] = a0 * (1.f / 24.f) - a1 * (1.f / 12.f) + a2 * (1.f / 6.f);
            u[5] = a2;
#pragma unroll
            for (int q = 0; q < 6; ++q) U[(size_t)(r * 6 + q) * total + i] = u[q];
        }
    }
}

int launch_winograd_filter(const float* w_oihw, float* U, int O, int I, hipStream_t s) {
    const int64_t total = (int64_t)O * I;
    hipLaunchKernelGGL(winograd_filter_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, w_oihw, U, O, I);
    FS_HIP(hipGetLastError());
    return 0;
}

// ---- input transform: thread per (tile, float2 channel pair); 36 coalesced loads (zero outside the image)
__global__ __launch_bounds__(256) void winograd_input_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ V, int B, int H,
                                                             int W, int CV, int th, int tw, int dil) {
    // dilation d: the conv splits into d*d independent undilated convs on the pixel lattices (py + d*i, px + d*j);
    // tile t = (b, py, px, ty, tx) covers lattice rows 4*ty-1 .. 4*ty+4 of phase (py, px)
    const int64_t T = (int64_t)B * dil * dil * th * tw;
    const int64_t total = T * CV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const int64_t t = i / CV;
        const int tx = (int)(t % tw), ty = (int)((t / tw) % th);
        const int ph = (int)((t / ((int64_t)tw * th)) % (dil * dil)), b = (int)(t / ((int64_t)tw * th * dil * dil));
        const int py = ph / dil, px = ph - py * dil;
        const int y0 = py + dil * (ty * 4 - 1), x0 = px + dil * (tx * 4 - 1);  // pad = dil <=> lattice pad 1
        const float* base = in + (size_t)b * H * W * ld_in + cv * WV;
        wv_t tmp[6][6];  // rows transformed: tmp[r][x] = (B^T d)[r][x]
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            wv_t col[6];
            const int ix = x0 + dil * x;
#pragma unroll
            for (int y = 0; y < 6; ++y) {
                const int iy = y0 + dil * y;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const size_t off = ok ? ((size_t)iy * W + ix) * ld_in : 0;
                const wv_t v = *reinterpret_cast<const wv_t*>(base + off);
                col[y] = ok ? v : wv_t{0.f, 0.f};
            }
            wv_t tc[6];
            wino_bt(col, tc);
#pragma unroll
            for (int r = 0; r < 6; ++r) tmp[r][x] = tc[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            wv_t o[6];
            wino_bt(tmp[r], o);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                *reinterpret_cast<wv_t*>(V + ((size_t)(r * 6 + q) * T + t) * CV * WV + cv * WV) = o[q];
        }
    }
}

int launch_winograd_input(const float* in, int ld_in, float* V, int B, int H, int W, int C, int dil, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_in % 4 == 0 && dil >= 1, "winograd_input: C must be a multiple of 4");
    const int th = (cdiv(H, dil) + 3) / 4, tw = (cdiv(W, dil) + 3) / 4;
    const int64_t total = (int64_t)B * dil * dil * th * tw * (C / WV);
    hipLaunchKernelGGL(winograd_input_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 20)), dim3(256), 0, s, in, ld_in, V,
                       B, H, W, C / WV, th, tw, dil);
    FS_HIP(hipGetLastError());
    return 0;
}

// ---- output transform + scale/shift + activation: thread per (tile, float2 output-channel pair)
__global__ __launch_bounds__(256) void winograd_output_kernel(const float* __restrict__ M, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, float* __restrict__ out, int ld_out, int B,
                                                              int H, int W, int NV, int th, int tw, int relu, int dil) {
    const int64_t T = (int64_t)B * dil * dil * th * tw;
    const int64_t total = T * NV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int nv = (int)(i % NV);
        const int64_t t = i / NV;
        const int tx = (int)(t % tw), ty = (int)((t / tw) % th);
        const int ph = (int)((t / ((int64_t)tw * th)) % (dil * dil)), b = (int)(t / ((int64_t)tw * th * dil * dil));
        const int py = ph / dil, px = ph - py * dil;
        wv_t tmp[4][6];  // tmp[a][q] = (A^T m)[a][q]
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            wv_t col[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) col[r] = *reinterpret_cast<const wv_t*>(M + ((size_t)(r * 6 + q) * T + t) * NV * WV + nv * WV);
            wv_t y[4];
            wino_at(col, y);
#pragma unroll
            for (int a = 0; a < 4; ++a) tmp[a][q] = y[a];
        }
        const wv_t sc = scale ? *reinterpret_cast<const wv_t*>(scale + nv * WV) : wv_t{1.f, 1.f};
        const wv_t sh = shift ? *reinterpret_cast<const wv_t*>(shift + nv * WV) : wv_t{0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            wv_t y[4];
            wino_at(tmp[a], y);
            const int oy = py + dil * (ty * 4 + a);
            if (oy >= H) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ox = px + dil * (tx * 4 + c);
                if (ox >= W) continue;
                wv_t v = y[c] * sc + sh;
                if (relu) {
#pragma unroll
                    for (int e = 0; e < WV; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                *reinterpret_cast<wv_t*>(out + ((size_t)(b * H + oy) * W + ox) * ld_out + nv * WV) = v;
            }
        }
    }
}

int launch_winograd_output(const float* M, const float* scale, const float* shift, float* out, int ld_out, int B, int H, int W, int N,
                           int relu, int dil, hipStream_t s) {
    FS_REQUIRE(N % 4 == 0 && ld_out % 4 == 0 && dil >= 1, "winograd_output: N must be a multiple of 4");
    const int th = (cdiv(H, dil) + 3) / 4, tw = (cdiv(W, dil) + 3) / 4;
    const int64_t total = (int64_t)B * dil * dil * th * tw * (N / WV);
    hipLaunchKernelGGL(winograd_output_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 1 << 20)), dim3(256), 0, s, M, scale,
                       shift, out, ld_out, B, H, W, N / WV, th, tw, relu, dil);
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
