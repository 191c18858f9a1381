// Winograd F(m x m, 3x3) transform matrices, m = 3 (points 0, +-1, 2, inf), m = 4 (Lavin & Gray) or m = 6 (points 0, +-1, +-2, +-1/2, inf, the NNPACK set):
// B^T d (input), A^T m (output), G g (filter).  Shared by the transform kernels (winograd.hip) and the fused kernel (wino_fused.hip).
#pragma once

namespace fs {

template <int MT> struct Wino;

// ---------------------------------------------------------------- F(3,3): points 0, 1, -1, 2, inf (Cook-Toom; round 6)
// 25 products per 9 outputs (2.78 per output, against 2.25 / 1.78) -- it pays where the tiles of the larger forms are mostly empty: the
// dilation-36 branch of DeepLab's ASPP on a 90 x 90 map is 36 x 36 lattices of 3 x 3 (or 2 x 2) pixels, ONE 3 x 3 tile each.
template <> struct Wino<3> {
    static constexpr int A = 5;
    template <typename T> __device__ static __forceinline__ void bt(const T d[5], T t[5]) {  // B^T d
        t[0] = 2.f * (d[0] - d[2]) - d[1] + d[3];
        t[1] = -2.f * d[1] - d[2] + d[3];
        t[2] = 2.f * d[1] - 3.f * d[2] + d[3];
        t[3] = d[3] - d[1];
        t[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    }
    template <typename T> __device__ static __forceinline__ void at(const T m[5], T y[3]) {  // A^T m
        y[0] = m[0] + m[1] + m[2] + m[3];
        y[1] = m[1] - m[2] + 2.f * m[3];
        y[2] = m[1] + m[2] + 4.f * m[3] + m[4];
    }
    __device__ static __forceinline__ void g(double g0, double g1, double g2, double u[5]) {  // G g
        u[0] = g0 / 2;
        u[1] = -(g0 + g1 + g2) / 2;
        u[2] = (-g0 + g1 - g2) / 6;
        u[3] = g0 / 6 + g1 / 3 + 2 * g2 / 3;
        u[4] = g2;
    }
};

// ---------------------------------------------------------------- F(4,3)
template <> struct Wino<4> {
    static constexpr int A = 6;
    template <typename T> __device__ static __forceinline__ void bt(const T d[6], T t[6]) {  // B^T d
        t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        t[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
        t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
        t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
        t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
        t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
    template <typename T> __device__ static __forceinline__ void at(const T m[6], T y[4]) {  // A^T m
        y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
        y[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
        y[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
        y[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
    }
    __device__ static __forceinline__ void g(double g0, double g1, double g2, double u[6]) {  // G g
        u[0] = g0 / 4;
        u[1] = -(g0 + g1 + g2) / 6;
        u[2] = -(g0 - g1 + g2) / 6;
        u[3] = g0 / 24 + g1 / 12 + g2 / 6;
        u[4] = g0 / 24 - g1 / 12 + g2 / 6;
        u[5] = g2;
    }
};

// ---------------------------------------------------------------- F(6,3)
template <> struct Wino<6> {
    static constexpr int A = 8;
    template <typename T> __device__ static __forceinline__ void bt(const T d[8], T t[8]) {
        const T a = d[2] - 4.25f * d[4] + d[6], b = d[1] - 4.25f * d[3] + d[5];
        const T c = 0.25f * d[2] - 1.25f * d[4] + d[6], e = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
        const T f = 4.f * d[2] - 5.f * d[4] + d[6], h = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
        t[0] = d[0] - d[6] + 5.25f * (d[4] - d[2]);
        t[1] = a + b;
        t[2] = a - b;
        t[3] = c + e;
        t[4] = c - e;
        t[5] = f + h;
        t[6] = f - h;
        t[7] = d[7] - d[1] + 5.25f * (d[3] - d[5]);
    }
    template <typename T> __device__ static __forceinline__ void at(const T m[8], T y[6]) {
        const T s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4], s3 = m[5] + m[6], d3 = m[5] - m[6];
        y[0] = m[0] + s1 + s2 + s3;
        y[1] = d1 + 2.f * d2 + 0.5f * d3;
        y[2] = s1 + 4.f * s2 + 0.25f * s3;
        y[3] = d1 + 8.f * d2 + 0.125f * d3;
        y[4] = s1 + 16.f * s2 + 0.0625f * s3;
        y[5] = d1 + 32.f * d2 + 0.03125f * d3 + m[7];
    }
    __device__ static __forceinline__ void g(double g0, double g1, double g2, double u[8]) {
        u[0] = g0;
        u[1] = -2.0 / 9 * (g0 + g1 + g2);
        u[2] = -2.0 / 9 * (g0 - g1 + g2);
        u[3] = g0 / 90 + g1 / 45 + 2 * g2 / 45;
        u[4] = g0 / 90 - g1 / 45 + 2 * g2 / 45;
        u[5] = 32 * g0 / 45 + 16 * g1 / 45 + 8 * g2 / 45;
        u[6] = 32 * g0 / 45 - 16 * g1 / 45 + 8 * g2 / 45;
        u[7] = g2;
    }
};

}  // namespace fs
