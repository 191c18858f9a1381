// Non-GEMM network kernels (HBM-bound or tiny): stem conv, pooling, PPM pieces, classifier,
// weight packing and layout changes.  All NHWC unless the name says otherwise.
#include "interp.h"
#include "kernels.h"

#include <type_traits>

namespace fs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------
// Stem conv (Cin = 3) from NCHW, fused scale/shift + ReLU, NHWC out.
// Block = 64 output pixels x 4 channel groups; filter bank lives in LDS ([tap*3+ci][Cout]).
// Cout must be a multiple of 16 and <= 128.
// -------------------------------------------------------------------------------------------
template <int CPT /*channels per thread*/>
__global__ __launch_bounds__(256) void stem_conv_kernel(StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    const int taps = p.KH * p.KW * 3;
    for (int i = threadIdx.x; i < taps * p.Cout; i += 256) wlds[i] = p.wgt[i];
    __syncthreads();

    const int groups = p.Cout / CPT;           // threads per pixel
    const int ppb = 256 / groups;              // pixels per block
    const int cg = threadIdx.x % groups;
    const int M = p.B * p.Ho * p.Wo;
    const int m = blockIdx.x * ppb + threadIdx.x / groups;
    if (m >= M) return;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw;
    const int rem = m - b * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;

    float acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
    size_t plane = (size_t)p.H * p.W;
    int rowW = p.W;
    const float* inb;
    if (p.src.ncrops) {  // crop window of a full frame: same taps, the frame's strides, zero padding at the CROP's border
        const int c = b < p.src.ncrops ? b : b - p.src.ncrops;
        plane = (size_t)p.src.FH * p.src.FW;
        rowW = p.src.FW;
        inb = (b < p.src.ncrops ? p.src.in : p.src.in2) + (size_t)p.src.cy[c] * rowW + p.src.cx[c];
    } else {
        inb = b < p.src.B1 ? p.src.in + (size_t)b * 3 * plane : p.src.in2 + (size_t)(b - p.src.B1) * 3 * plane;
    }
    for (int ky = 0; ky < p.KH; ++ky) {
        const int iy = iy0 + ky;
        const bool yok = (unsigned)iy < (unsigned)p.H;
        for (int kx = 0; kx < p.KW; ++kx) {
            const int ix = ix0 + kx;
            const bool ok = yok && (unsigned)ix < (unsigned)p.W;
            const size_t off = ok ? (size_t)iy * rowW + ix : 0;
            const float x0 = ok ? inb[off] : 0.f;
            const float x1 = ok ? inb[plane + off] : 0.f;
            const float x2 = ok ? inb[2 * plane + off] : 0.f;
            const float* w = wlds + (size_t)((ky * p.KW + kx) * 3) * p.Cout + cg * CPT;
#pragma unroll
            for (int c = 0; c < CPT; c += 4) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(w + p.Cout + c);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(w + 2 * p.Cout + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c + e] += x0 * w0[e] + x1 * w1[e] + x2 * w2[e];
            }
        }
    }
    float* o = p.out + (size_t)m * p.ld_out + cg * CPT;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = cg * CPT + c + e;
            float y = acc[c + e] * p.scale[ch] + p.shift[ch];
            v[e] = fmaxf(y, 0.f);
        }
        *reinterpret_cast<f32x4*>(o + c) = v;
    }
}

// -------------------------------------------------------------------------------------------
// The same stem conv on the matrix cores (Cout a multiple of 32, <= 128): an implicit GEMM with M = output pixels,
// N = Cout, K = KH*KW*3 taps (27 or 147, padded to an even count with zero weights).  One wave = 32 pixels x all channels;
// v_mfma_f32_32x32x2_f32 takes A[i][k] from lane (i = l&31, h = l>>5) with k = 2*step + h: each lane gathers ITS tap of ITS
// pixel straight from the NCHW frame (neighbouring pixels share lines: L1/L2 hits), B[k][n] comes from the filter bank in LDS.
// The VALU version above spends 27 (147) x Cout fmas per pixel on ds_read-fed VALU: 47 us (139 us for the 7x7 stem) at
// 713x713, B = 2, against an 11 us HBM write; this one needs 28 (148) MFMAs per 32 pixels.
// -------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT /* 32-channel sub-tiles */>
__global__ __launch_bounds__(256) void stem_conv_mfma_kernel(StemParams p, int ksteps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int taps = p.KH * p.KW * 3, kpad = 2 * ksteps;
    float* wl = smem;                                            // [kpad][Cout], rows >= taps are zero
    int* koff = reinterpret_cast<int*>(smem + kpad * p.Cout);    // [kpad]: element offset of tap k from the patch origin
    int* kyx = koff + kpad;                                      // [kpad]: ky << 16 | kx
    const int rowW = p.src.ncrops ? p.src.FW : p.W;
    const long long plane = p.src.ncrops ? (long long)p.src.FH * p.src.FW : (long long)p.H * p.W;
    for (int i = threadIdx.x; i < kpad * p.Cout; i += 256) wl[i] = i < taps * p.Cout ? p.wgt[i] : 0.f;
    for (int k = threadIdx.x; k < kpad; k += 256) {
        const int kk = k < taps ? k : 0;  // k = (ky*KW + kx)*3 + ci
        const int ci = kk % 3, kx = (kk / 3) % p.KW, ky = kk / (3 * p.KW);
        koff[k] = (int)(ci * plane + (long long)ky * rowW + kx);
        kyx[k] = k < taps ? (ky << 16 | kx) : (0x7fff << 16);  // padding taps: a row index no image has
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int M = p.B * p.Ho * p.Wo, hw = p.Ho * p.Wo;
    // rows >= M fall outside the descriptor's range: the stores are dropped (launcher: M * ld_out * 4 < 2 GiB)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (unsigned)((long long)M * p.ld_out * 4), 0x00020000);
    const int tiles = (M + 31) / 32;
    for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
        const int m = tile * 32 + i;
        const bool mok = m < M;
        const int mm = mok ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const float* inb;
        if (p.src.ncrops) {
            const int c = b < p.src.ncrops ? b : b - p.src.ncrops;
            inb = (b < p.src.ncrops ? p.src.in : p.src.in2) + (size_t)p.src.cy[c] * rowW + p.src.cx[c];
        } else {
            inb = b < p.src.B1 ? p.src.in + (size_t)b * 3 * plane : p.src.in2 + (size_t)(b - p.src.B1) * 3 * plane;
        }
        const float* origin = inb + (long long)iy0 * rowW + ix0;  // may point before the frame: only dereferenced for in-range taps
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        // taps in batches of 8 steps: 8 gathers in flight per lane before their MFMAs (kpad is padded to the batch, see launcher)
        for (int s0 = 0; s0 < ksteps; s0 += 8) {
            float av[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 2 * (s0 + u) + h;
                const int yx = kyx[k];
                const int iy = iy0 + (yx >> 16), ix = ix0 + (yx & 0xffff);
                const bool ok = mok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                av[u] = ok ? origin[koff[k]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = 2 * (s0 + u) + h;
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], wl[k * p.Cout + j * 32 + i], acc[j], 0, 0, 0);
            }
        }
        // D layout: col n = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * h: every store instruction writes two 128-B channel runs
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + i;
            const float sc = p.scale[n], sh = p.shift[n];
            // (straight-line stores through a buffer descriptor whose range check is the row guard: written as `if (mr < M) store` every
            //  store sat in a basic block of its own behind an s_waitcnt vmcnt(0), which also waits for the previous store: round 6)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int mr = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float v = fmaxf(acc[j][e] * sc + sh, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, (unsigned)(mr * p.ld_out + n) * 4u, 0, 0);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// (round 5) The same implicit GEMM on the bf16 matrix cores with SPLIT operands, the arithmetic of conv_igemm_dma_f32<SPLIT>: every
// fp32 pixel and filter value as the exact sum of three bf16 terms, six of the nine cross products on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation (the dropped ones are <= 2^-23 of a product).  Why here: five waves per SIMD shared the fp32 matrix pipe for
// 32 x 64 cycles each and a wave lived 29.7 k cycles for its ONE tile (r04_experiments.txt section 4); the split form needs 12 x NT MFMAs
// of 32 cycles per 16 taps instead of 8 x NT of 64 -- 2.7x less matrix time per tile -- and requests the NEXT step's taps before it
// multiplies the current ones.  A lane (i = l & 31, h = l >> 5) owns taps 16 s + 8 h .. + 7 of pixel i in step s; the filter bank is
// split once per workgroup into LDS as ready-made B fragments [plane][step][h][Cout][8 bf16].
// -------------------------------------------------------------------------------------------
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef int i32x4s __attribute__((ext_vector_type(4)));

template <int NT /* 32-channel sub-tiles */, int PF /* steps whose taps are in flight together */>
__global__ __launch_bounds__(256) void stem_conv_split_kernel(StemParams p, int nsteps) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int taps = p.KH * p.KW * 3, kpad = 16 * nsteps;
    u32x4s* wl = reinterpret_cast<u32x4s*>(smem);                              // [(pl * nsteps + s) * 2 + h][Cout] x 8 bf16
    int* koff = reinterpret_cast<int*>(smem) + 3 * (kpad / 2) * p.Cout;        // [kpad]: element offset of tap k from the patch origin
    int* kyx = koff + kpad;                                                    // [kpad]: ky << 16 | kx
    const int rowW = p.src.ncrops ? p.src.FW : p.W;
    const long long plane = p.src.ncrops ? (long long)p.src.FH * p.src.FW : (long long)p.H * p.W;
    for (int idx = threadIdx.x; idx < nsteps * 2 * p.Cout; idx += 256) {
        const int n = idx % p.Cout, sh_ = idx / p.Cout, hh = sh_ & 1, st = sh_ >> 1;
        u32x4s q[3];
        float wv[8];  // all eight loads in flight at once, from clamped addresses; padding taps carry zero weights
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = p.wgt[min(16 * st + 8 * hh + e, taps - 1) * p.Cout + n];
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[e] = 16 * st + 8 * hh + e < taps ? wv[e] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float w[2] = {wv[e], wv[e + 1]};
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2s){w[0], w[1]}, bf16x2s));
                q[pl][e >> 1] = pk;
                w[0] -= __builtin_bit_cast(float, pk << 16);
                w[1] -= __builtin_bit_cast(float, pk & 0xffff0000u);
            }
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wl[((pl * nsteps + st) * 2 + hh) * p.Cout + n] = q[pl];
    }
    for (int k = threadIdx.x; k < kpad; k += 256) {
        const int kk = k < taps ? k : 0;  // k = (ky*KW + kx)*3 + ci
        const int ci = kk % 3, kx = (kk / 3) % p.KW, ky = kk / (3 * p.KW);
        koff[k] = (int)(ci * plane + (long long)ky * rowW + kx);
        kyx[k] = k < taps ? (ky << 16 | kx) : (0x7fff << 16);  // padding taps: a row index no image has
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    const int M = p.B * p.Ho * p.Wo, hw = p.Ho * p.Wo;
    // rows >= M fall outside the descriptor's range: the stores are dropped (launcher: M * ld_out * 4 < 2 GiB)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (unsigned)((long long)M * p.ld_out * 4), 0x00020000);
    const int tiles = (M + 31) / 32;
    for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
        const int m = tile * 32 + i;
        const bool mok = m < M;
        const int mm = mok ? m : 0;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const float* inb;
        if (p.src.ncrops) {
            const int c = b < p.src.ncrops ? b : b - p.src.ncrops;
            inb = (b < p.src.ncrops ? p.src.in : p.src.in2) + (size_t)p.src.cy[c] * rowW + p.src.cx[c];
        } else {
            inb = b < p.src.B1 ? p.src.in + (size_t)b * 3 * plane : p.src.in2 + (size_t)(b - p.src.B1) * 3 * plane;
        }
        const float* origin = inb + (long long)iy0 * rowW + ix0;  // may point before the frame: only dereferenced for in-range taps
        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        // Branch-free, in chunks of PF steps (round 6).  Written as `ok ? origin[koff[k]] : 0`, one step ahead, every tap was a basic
        // block of its own -- exec mask, LDS read of its offset, wait, load -- and a 7 x 7 tile paid ten exposed round trips.  Now the
        // offsets and row / column pairs of a step are two 16-B LDS reads each, every tap loads unconditionally (a tap outside the
        // frame reads the patch's centre instead, which every output pixel has inside the frame), ALL loads of PF steps are issued
        // before the first is used, and the out-of-frame taps are zeroed when their step is split (a select next to the load would
        // make the wave wait there).  No `mok` in the test: rows >= M compute pixel 0's values and their stores are dropped; with it
        // the compiler wraps the taps in an `if (mok)` region and waits for every load inside it.
        const int centre = p.pad * rowW + p.pad;
        auto gather = [&](int st, float (&av)[8]) -> unsigned {
            const i32x4s* ko = reinterpret_cast<const i32x4s*>(koff + 16 * st + 8 * h);
            const i32x4s* ky = reinterpret_cast<const i32x4s*>(kyx + 16 * st + 8 * h);
            const i32x4s o4[2] = {ko[0], ko[1]}, y4[2] = {ky[0], ky[1]};
            unsigned okbits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int yx = y4[e >> 2][e & 3];
                const int iy = iy0 + (yx >> 16), ix = ix0 + (yx & 0xffff);
                const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                av[e] = origin[ok ? o4[e >> 2][e & 3] : centre];
                okbits |= (unsigned)ok << e;
            }
            return okbits;
        };
        for (int c0 = 0; c0 < nsteps; c0 += PF) {
            float raw[PF][8];
            unsigned okb[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) okb[j] = gather(min(c0 + j, nsteps - 1), raw[j]);
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int st = c0 + j;
                if (st < nsteps) {  // uniform
                    u32x4s a3[3];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        float x0 = (okb[j] >> e) & 1 ? raw[j][e] : 0.f, x1 = (okb[j] >> (e + 1)) & 1 ? raw[j][e + 1] : 0.f;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2s){x0, x1}, bf16x2s));
                            a3[pl][e >> 1] = pk;
                            x0 -= __builtin_bit_cast(float, pk << 16);
                            x1 -= __builtin_bit_cast(float, pk & 0xffff0000u);
                        }
                    }
                    // the six products of order <= 2^-16, smallest first: l h', h l', m m', m h', h m', h h'
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int jn = 0; jn < NT; ++jn)
                            acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8s, a3[PA[term]]),
                                                                              __builtin_bit_cast(bf16x8s, wl[((PB[term] * nsteps + st) * 2 + h) * p.Cout + jn * 32 + i]), acc[jn], 0, 0, 0);
                }
            }
        }
        // D layout: col n = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * h: every store instruction writes two 128-B channel runs
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + i;
            const float sc = p.scale[n], sh = p.shift[n];
            // (straight-line stores through a buffer descriptor whose range check is the row guard: written as `if (mr < M) store` every
            //  store sat in a basic block of its own behind an s_waitcnt vmcnt(0), which also waits for the previous store: round 6)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int mr = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float v = fmaxf(acc[j][e] * sc + sh, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, (unsigned)(mr * p.ld_out + n) * 4u, 0, 0);
            }
        }
    }
#endif
}

int launch_stem_conv(const StemParams& p, hipStream_t s) {
    FS_REQUIRE(p.Cout >= 16 && p.Cout <= 256 && p.Cout % 16 == 0, "stem_conv: unsupported Cout=%d", p.Cout);
    FS_REQUIRE(p.ld_out % 4 == 0, "stem_conv: ld_out must be a multiple of 4");
    FS_REQUIRE((long long)p.B * p.Ho * p.Wo * p.ld_out * 4 < (1ll << 31), "stem_conv: output tensor must be smaller than 2 GiB");
    const FrameSrc& f = p.src;
    if (f.ncrops) {
        FS_REQUIRE(f.ncrops <= 32 && f.in && (p.B == f.ncrops || (p.B == 2 * f.ncrops && f.in2)), "stem_conv: bad crop batch (%d crops, B=%d)", f.ncrops, p.B);
        for (int c = 0; c < f.ncrops; ++c)
            FS_REQUIRE(f.cy[c] >= 0 && f.cx[c] >= 0 && f.cy[c] + p.H <= f.FH && f.cx[c] + p.W <= f.FW, "stem_conv: crop %d (%d,%d)+%dx%d outside the %dx%d frame",
                       c, f.cy[c], f.cx[c], p.H, p.W, f.FH, f.FW);
    } else {
        FS_REQUIRE(f.B1 >= 0 && f.B1 <= p.B && (f.B1 == p.B || f.in2) && (f.B1 == 0 || f.in), "stem_conv: bad frame split B1=%d of B=%d", f.B1, p.B);
    }
    const int M = p.B * p.Ho * p.Wo;
    const int taps = p.KH * p.KW * 3;
    if (p.split && p.Cout % 32 == 0 && p.Cout <= 128 && p.KH < 0x7fff) {  // split-operand route: bf16 matrix cores, fp32 accuracy
        const int nsteps = cdiv(taps, 16);
        const size_t lds = (size_t)3 * 16 * nsteps * p.Cout * 2 + (size_t)2 * 16 * nsteps * sizeof(int);
        FS_REQUIRE(lds <= 150 * 1024, "stem_conv: filter planes %zu B exceed the LDS budget", lds);
        FS_REQUIRE((long long)3 * (f.ncrops ? (long long)f.FH * f.FW : (long long)p.H * p.W) < (1ll << 31), "stem_conv: frame too large for 32-bit tap offsets");
        const int tiles = cdiv(M, 32);
        // every workgroup splits the whole filter bank into LDS before its first tile (a 7 x 7 x 64 bank: 1280 rows of eight weights):
        // no more workgroups than a CU can hold at once (LDS), so that a wave walks several tiles behind one prologue (round 6: the 7 x 7
        // stem ran 1992 workgroups of ONE tile per wave, four rounds of prologue)
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (size_t)160 * 1024 / lds));
        const dim3 grid((unsigned)std::min(cdiv(tiles, 4), 256 * per_cu));
#define FS_STEM_SPLIT_PF(NT_, PF_)                                                                                               \
    {                                                                                                                            \
        if (lds > 64 * 1024)                                                                                                     \
            FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_split_kernel<NT_, PF_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((stem_conv_split_kernel<NT_, PF_>), grid, dim3(256), lds, s, p, nsteps);                              \
    }
#define FS_STEM_SPLIT(NT_)                                                                                                       \
    {                                                                                                                            \
        if (nsteps <= 2) FS_STEM_SPLIT_PF(NT_, 2) else FS_STEM_SPLIT_PF(NT_, 5)                                                  \
    }
        switch (p.Cout / 32) {
            case 1: FS_STEM_SPLIT(1) break;
            case 2: FS_STEM_SPLIT(2) break;
            case 3: FS_STEM_SPLIT(3) break;
            default: FS_STEM_SPLIT(4) break;
        }
#undef FS_STEM_SPLIT
#undef FS_STEM_SPLIT_PF
        FS_HIP(hipGetLastError());
        return 0;
    }
    if (p.Cout % 32 == 0 && p.Cout <= 128 && p.KH < 0x7fff) {  // matrix-core route
        const int ksteps = cdiv(cdiv(taps, 2), 8) * 8;  // whole batches of 8 MFMA k-steps; the padding taps carry zero weights
        const size_t lds = (size_t)2 * ksteps * (p.Cout + 2) * sizeof(float);
        FS_REQUIRE(lds <= 128 * 1024, "stem_conv: filter bank %zu B exceeds LDS budget", lds);
        FS_REQUIRE((long long)3 * (f.ncrops ? (long long)f.FH * f.FW : (long long)p.H * p.W) < (1ll << 31), "stem_conv: frame too large for 32-bit tap offsets");
        const int tiles = cdiv(M, 32);
        const dim3 grid((unsigned)std::min(cdiv(tiles, 4), 256 * 8));
#define FS_STEM_LAUNCH(NT_)                                                                                                      \
    {                                                                                                                            \
        if (lds > 64 * 1024)  /* 7x7 with > 64 channels: more dynamic LDS than the default cap */                                \
            FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_mfma_kernel<NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((stem_conv_mfma_kernel<NT_>), grid, dim3(256), lds, s, p, ksteps);                                    \
    }
        switch (p.Cout / 32) {
            case 1: FS_STEM_LAUNCH(1) break;
            case 2: FS_STEM_LAUNCH(2) break;
            case 3: FS_STEM_LAUNCH(3) break;
            default: FS_STEM_LAUNCH(4) break;
        }
#undef FS_STEM_LAUNCH
        FS_HIP(hipGetLastError());
        return 0;
    }
    FS_REQUIRE(256 % (p.Cout / 16) == 0, "stem_conv: the VALU kernel needs Cout / 16 to divide 256 (Cout=%d)", p.Cout);
    const int groups = p.Cout / 16, ppb = 256 / groups;
    const size_t lds = (size_t)taps * p.Cout * sizeof(float);
    FS_REQUIRE(lds <= 64 * 1024, "stem_conv: filter bank %zu B exceeds LDS budget", lds);
    hipLaunchKernelGGL((stem_conv_kernel<16>), dim3(cdiv(M, ppb)), dim3(256), lds, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// MaxPool 3x3 stride 2 pad 1 (padding never wins: -inf)
// -------------------------------------------------------------------------------------------
// grid = (blocks of 256 (pixel, channel quad) items of one output row, output rows, images): one 32-bit division per thread (the
// flat 64-bit index this kernel started with cost four 64-bit divisions per thread, ~400 instructions for 9 loads and a store)
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                           int ld_out, int B, int H, int W, int C4, int Ho, int Wo) {
    const unsigned r = blockIdx.x * 256 + threadIdx.x;
    if (r >= (unsigned)(Wo * C4)) return;
    const int ox = (int)(r / (unsigned)C4), c4 = (int)(r - (unsigned)ox * (unsigned)C4);
    for (int b = blockIdx.z; b < B; b += gridDim.z) {
        for (int oy = blockIdx.y; oy < Ho; oy += gridDim.y) {
            // all nine taps are loaded from clamped (always valid) addresses and the ones outside the map replaced by -inf afterwards:
            // `if (outside) continue; load` is a branch and an s_waitcnt vmcnt(0) per tap -- nine serial round trips (round 6)
            f32x4 tap[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = min(max(oy * 2 - 1 + ky, 0), H - 1);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = min(max(ox * 2 - 1 + kx, 0), W - 1);
                    tap[ky * 3 + kx] = *reinterpret_cast<const f32x4*>(in + ((size_t)(b * H + iy) * W + ix) * ld_in + c4 * 4);
                }
            }
            f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bool ok = (unsigned)(oy * 2 - 1 + ky) < (unsigned)H && (unsigned)(ox * 2 - 1 + kx) < (unsigned)W;
#pragma unroll
                    for (int e = 0; e < 4; ++e) best[e] = fmaxf(best[e], ok ? tap[ky * 3 + kx][e] : -INFINITY);
                }
            *reinterpret_cast<f32x4*>(out + ((size_t)(b * Ho + oy) * Wo + ox) * ld_out + c4 * 4) = best;
        }
    }
}

int launch_maxpool3x3s2(const float* in, int ld_in, float* out, int ld_out, int B, int H, int W, int C, int Ho, int Wo,
                        hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_in % 4 == 0 && ld_out % 4 == 0, "maxpool: C/ld must be multiples of 4");
    FS_REQUIRE((int64_t)Wo * (C / 4) < (int64_t)1 << 31, "maxpool: output row too large");
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)cdiv(Wo * (C / 4), 256), (unsigned)std::min(Ho, 65535), (unsigned)std::min(B, 65535)), dim3(256), 0, s,
                       in, ld_in, out, ld_out, B, H, W, C / 4, Ho, Wo);
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// AdaptiveAvgPool2d(bin): windows [floor(i*H/bin), ceil((i+1)*H/bin)).
// Block = 32 pixel lanes x 8 float4 channel lanes (32 channels = one 128-B line per pixel); grid (bin*bin, C/32, B):
// even the bin = 1 case (one 90x90 window) spreads over C/32 * B blocks with 254 pixels per thread.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                               int H, int W, int C, int bin) {
    __shared__ f32x4 part[32][8];
    const int cell = blockIdx.x, b = blockIdx.z;
    const int by = cell / bin, bx = cell - by * bin;
    const int ys = (by * H) / bin, ye = ((by + 1) * H + bin - 1) / bin;
    const int xs = (bx * W) / bin, xe = ((bx + 1) * W + bin - 1) / bin;
    const int wh = ye - ys, ww = xe - xs;
    const int cl = threadIdx.x & 7, pg = threadIdx.x >> 3;
    const int c = blockIdx.y * 32 + cl * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = pg; i < wh * ww; i += 32) {
        const int y = ys + i / ww, x = xs + i % ww;
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((size_t)(b * H + y) * W + x) * ld_in + c);
        acc += v;
    }
    part[pg][cl] = acc;
    __syncthreads();
    if (pg == 0) {
        f32x4 sum = part[0][cl];
        for (int g = 1; g < 32; ++g) sum += part[g][cl];
        const float inv = 1.f / (float)(wh * ww);
        sum *= inv;
        *reinterpret_cast<f32x4*>(out + ((size_t)b * bin * bin + cell) * C + c) = sum;
    }
}

int launch_adaptive_avgpool(const float* in, int ld_in, float* out, int B, int H, int W, int C, int bin, hipStream_t s) {
    FS_REQUIRE(C % 32 == 0 && ld_in % 4 == 0, "adaptive_avgpool: C=%d must be a multiple of 32", C);
    hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(bin * bin, C / 32, B), dim3(256), 0, s, in, ld_in, out, H, W, C, bin);
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// PPM fast path: all four AdaptiveAvgPool2d bins (1,2,3,6) of model/pspnet.py:42 in ONE pass over the map when
// H and W are multiples of 6 (90x90 at 713^2): stage 1 sums the 6x6 cells, stage 2 combines cells into the 3/2/1 bins.
// (Window sums are combined hierarchically, so the last bits differ from a flat sum; covered by the 5e-6 test tolerance.)
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppm_pool_combine_kernel(const float* __restrict__ cell_mean /*[B][36][C]*/, float* __restrict__ out1,
                                                               float* __restrict__ out2, float* __restrict__ out3, int B, int C) {
    const int64_t total = (int64_t)B * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C), b = (int)(i / C);
        const float* m = cell_mean + (size_t)b * 36 * C + c;
        float s3[9], s2[4] = {0.f, 0.f, 0.f, 0.f}, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s3[k] = 0.f;
#pragma unroll
        for (int y = 0; y < 6; ++y)
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                const float v = m[(size_t)(y * 6 + x) * C];
                s3[(y / 2) * 3 + x / 2] += v;
                s2[(y / 3) * 2 + x / 3] += v;
                s1 += v;
            }
#pragma unroll
        for (int k = 0; k < 9; ++k) out3[((size_t)b * 9 + k) * C + c] = s3[k] * 0.25f;
#pragma unroll
        for (int k = 0; k < 4; ++k) out2[((size_t)b * 4 + k) * C + c] = s2[k] * (1.f / 9.f);
        out1[(size_t)b * C + c] = s1 * (1.f / 36.f);
    }
}

int launch_ppm_pool_combine(const float* cell_mean, float* out1, float* out2, float* out3, int B, int C, hipStream_t s) {
    const int64_t total = (int64_t)B * C;
    hipLaunchKernelGGL(ppm_pool_combine_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, cell_mean, out1, out2, out3, B, C);
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// Small-M 1x1 conv: one wave per output channel n, weight row held in registers, loop over the rows of this
// block's row chunk (blockIdx.y): the rows are independent, so chunking them shortens the serial chain per wave.
// -------------------------------------------------------------------------------------------
template <int KV /* float4 per lane = K/256 */>
__global__ __launch_bounds__(256) void rowdot_1x1_kernel(RowdotBatch pb, int ld_in, int ld_out, int K, int N, int relu) {
    const RowdotProblem& q = pb.p[blockIdx.z];  // several small problems of one (K, N) in one launch: the four pyramid levels
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int M = q.M;
    const int per = (M + gridDim.y - 1) / gridDim.y;
    const int m_begin = blockIdx.y * per, m_end = min(M, (int)(blockIdx.y + 1) * per);
    if (m_begin >= m_end) return;
    f32x4 w[KV];
#pragma unroll
    for (int i = 0; i < KV; ++i) w[i] = *reinterpret_cast<const f32x4*>(q.wgt + (size_t)n * K + (i * 64 + lane) * 4);
    const float sc = q.scale ? q.scale[n] : 1.f, sh = q.shift ? q.shift[n] : 0.f;
    for (int m = m_begin; m < m_end; ++m) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < KV; ++i) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(q.in + (size_t)m * ld_in + (i * 64 + lane) * 4);
            acc += x[0] * w[i][0] + x[1] * w[i][1] + x[2] * w[i][2] + x[3] * w[i][3];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) {
            float y = acc * sc + sh;
            if (relu) y = fmaxf(y, 0.f);
            q.out[(size_t)m * ld_out + n] = y;
        }
    }
}

int launch_rowdot_1x1_batch(const RowdotBatch& pb, int nprob, int ld_in, int ld_out, int K, int N, int relu, hipStream_t s) {
    FS_REQUIRE(K % 256 == 0 && K <= 4096, "rowdot_1x1: K=%d must be a multiple of 256 and <= 4096", K);
    FS_REQUIRE(nprob >= 1 && nprob <= 4, "rowdot_1x1: 1..4 problems per launch");
    int mmax = 1;
    for (int i = 0; i < nprob; ++i) mmax = std::max(mmax, pb.p[i].M);
    const dim3 grid(cdiv(N, 4), std::max(1, std::min(12, mmax / 6)), nprob), block(256);  // ~6+ rows per wave of the largest problem
#define FS_ROWDOT(KV)                                                                                             \
    case KV:                                                                                                      \
        hipLaunchKernelGGL((rowdot_1x1_kernel<KV>), grid, block, 0, s, pb, ld_in, ld_out, K, N, relu);            \
        break;
    switch (K / 256) {
        FS_ROWDOT(1) FS_ROWDOT(2) FS_ROWDOT(3) FS_ROWDOT(4) FS_ROWDOT(5) FS_ROWDOT(6) FS_ROWDOT(7) FS_ROWDOT(8)
        FS_ROWDOT(9) FS_ROWDOT(10) FS_ROWDOT(11) FS_ROWDOT(12) FS_ROWDOT(13) FS_ROWDOT(14) FS_ROWDOT(15) FS_ROWDOT(16)
        default: return fail("rowdot_1x1: unsupported K");
    }
#undef FS_ROWDOT
    FS_HIP(hipGetLastError());
    return 0;
}

int launch_rowdot_1x1(const float* in, int ld_in, const float* wgt, const float* scale, const float* shift, float* out,
                      int ld_out, int M, int K, int N, int relu, hipStream_t s) {
    RowdotBatch pb{};
    pb.p[0] = RowdotProblem{in, wgt, scale, shift, out, M};
    return launch_rowdot_1x1_batch(pb, 1, ld_in, ld_out, K, N, relu, s);
}

// -------------------------------------------------------------------------------------------
// Bilinear upsample of a tiny pooled map [B][hi*wi][C] into a channel slice of an NHWC buffer.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_into_kernel(const float* __restrict__ in, int hi, int wi, float* __restrict__ out,
                                                            int ld_out, int B, int Ho, int Wo, int C4, int ac, float sy, float sx) {
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const int64_t m = i / C4;
        const int ox = (int)(m % Wo);
        const int oy = (int)((m / Wo) % Ho);
        const int b = (int)(m / ((int64_t)Wo * Ho));
        const LinCoord cy = lin_coord(oy, hi, sy, ac), cx = lin_coord(ox, wi, sx, ac);
        const float* base = in + (size_t)b * hi * wi * C4 * 4 + c4 * 4;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + (size_t)(cy.i0 * wi + cx.i0) * C4 * 4);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + (size_t)(cy.i0 * wi + cx.i1) * C4 * 4);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + (size_t)(cy.i1 * wi + cx.i0) * C4 * 4);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + (size_t)(cy.i1 * wi + cx.i1) * C4 * 4);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = bilerp(v00[e], v01[e], v10[e], v11[e], cy, cx);
        *reinterpret_cast<f32x4*>(out + (size_t)m * ld_out + c4 * 4) = r;
    }
}

int launch_upsample_into(const float* in, int hi, int wi, float* out, int ld_out, int B, int Ho, int Wo, int C,
                         int align_corners, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out & 15) == 0, "upsample_into: C/ld_out must be multiples of 4");
    const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
    const int grid = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 32);
    hipLaunchKernelGGL(upsample_into_kernel, dim3(grid), dim3(256), 0, s, in, hi, wi, out, ld_out, B, Ho, Wo, C / 4,
                       align_corners, resize_scale(hi, Ho, align_corners), resize_scale(wi, Wo, align_corners));
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// Classifier 1x1 conv + bias, NHWC -> NCHW.  16 lanes per pixel, 16 pixels per block.
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void classifier_nchw_kernel(const float* __restrict__ in, int ld_in, const float* __restrict__ wgt,
                                                              const float* __restrict__ bias, float* __restrict__ out, int B,
                                                              int HW, int C, int K) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [K][C]
    for (int i = threadIdx.x; i < K * C; i += 256) wl[i] = wgt[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    const int64_t M = (int64_t)B * HW;
    const int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t mc = m < M ? m : M - 1;  // keep all lanes alive for the shuffles
    const float* x = in + (size_t)mc * ld_in;
    for (int k0 = 0; k0 < K; k0 += 8) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int c = l16 * 4; c < C; c += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k0 + k < K) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wl + (size_t)(k0 + k) * C + c);
                    acc[k] += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
        }
        if (l16 == 0 && m < M) {
            const int64_t b = m / HW, pix = m - b * HW;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k < K) out[((size_t)b * K + k0 + k) * HW + pix] = acc[k] + (bias ? bias[k0 + k] : 0.f);
        }
    }
}

int launch_classifier_nchw(const float* in, int ld_in, const float* wgt, const float* bias, float* out, int B, int HW,
                           int C, int K, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && ld_in % 4 == 0, "classifier: C must be a multiple of 4");
    const size_t lds = (size_t)K * C * sizeof(float);
    FS_REQUIRE(lds <= 64 * 1024, "classifier: K*C too large for LDS");
    const int64_t M = (int64_t)B * HW;
    hipLaunchKernelGGL(classifier_nchw_kernel, dim3((unsigned)cdiv64(M, 16)), dim3(256), lds, s, in, ld_in, wgt, bias, out, B,
                       HW, C, K);
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// Weight packing (one-off at load time) and layout changes
// -------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I, int KH, int KW, int hwio) {
    const int64_t total = (int64_t)O * I * KH * KW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the source OIHW tensor
        const int s = (int)(i % KW);
        const int r = (int)((i / KW) % KH);
        const int ci = (int)((i / ((int64_t)KW * KH)) % I);
        const int o = (int)(i / ((int64_t)KW * KH * I));
        size_t dst;
        if (hwio == 1)
            dst = ((size_t)((r * KW + s) * I + ci)) * O + o;
        else if (hwio == 2)  // chunk-major k: [O][I/32][KH][KW][32] (ConvParams::korder == 1)
            dst = (((size_t)o * (I / 32) + ci / 32) * KH * KW + r * KW + s) * 32 + (ci & 31);
        else
            dst = ((size_t)((o * KH + r) * KW + s)) * I + ci;
        out[dst] = w[i];
    }
}

int launch_pack_oihw_to_ohwi(const float* w, float* out, int O, int I, int KH, int KW, hipStream_t s) {
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, w, out,
                       O, I, KH, KW, 0);
    FS_HIP(hipGetLastError());
    return 0;
}
int launch_pack_oihw_chunk_major(const float* w, float* out, int O, int I, int KH, int KW, hipStream_t s) {
    FS_REQUIRE(I % 32 == 0, "pack(chunk-major): Cin must be a multiple of 32");
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, w, out,
                       O, I, KH, KW, 2);
    FS_HIP(hipGetLastError());
    return 0;
}
int launch_pack_oihw_to_hwio(const float* w, float* out, int O, int I, int KH, int KW, hipStream_t s) {
    const int64_t total = (int64_t)O * I * KH * KW;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, w, out,
                       O, I, KH, KW, 1);
    FS_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void concat_scaled_filters_kernel(const float* __restrict__ wa, const float* __restrict__ sa,
                                                                    const float* __restrict__ ha, int Ka, const float* __restrict__ wb,
                                                                    const float* __restrict__ sb, const float* __restrict__ hb, int Kb,
                                                                    float* __restrict__ out, float* __restrict__ shift_out, int O) {
    const int K = Ka + Kb;
    const int64_t total = (int64_t)O * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k = (int)(i % K), o = (int)(i / K);
        out[i] = k < Ka ? sa[o] * wa[(size_t)o * Ka + k] : sb[o] * wb[(size_t)o * Kb + (k - Ka)];
        if (k == 0) shift_out[o] = ha[o] + hb[o];
    }
}
int launch_concat_scaled_filters(const float* wa, const float* sa, const float* ha, int Ka, const float* wb, const float* sb, const float* hb,
                                 int Kb, float* out, float* shift_out, int O, hipStream_t s) {
    FS_REQUIRE(wa && sa && ha && wb && sb && hb && out && shift_out && O >= 1 && Ka >= 1 && Kb >= 1, "concat_scaled_filters: bad arguments");
    const int64_t total = (int64_t)O * (Ka + Kb);
    hipLaunchKernelGGL(concat_scaled_filters_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, wa, sa, ha, Ka,
                       wb, sb, hb, Kb, out, shift_out, O);
    FS_HIP(hipGetLastError());
    return 0;
}

// Slice of the input channels of an OIHW filter bank, tap-major: out[(tap*O + o)*nc + c] = w[o][c0 + c][tap].
__global__ __launch_bounds__(256) void pack_slice_tap_major_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int I,
                                                                   int c0, int nc, int taps) {
    const int64_t total = (int64_t)taps * O * nc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % nc);
        const int o = (int)((i / nc) % O);
        const int tap = (int)(i / ((int64_t)nc * O));
        out[i] = w[((size_t)o * I + c0 + c) * taps + tap];
    }
}
int launch_pack_slice_tap_major(const float* w, float* out, int O, int I, int c0, int nc, int taps, hipStream_t s) {
    FS_REQUIRE(c0 >= 0 && nc >= 1 && c0 + nc <= I, "pack(slice): channel range outside the filter bank");
    const int64_t total = (int64_t)taps * O * nc;
    hipLaunchKernelGGL(pack_slice_tap_major_kernel, dim3((unsigned)std::min<int64_t>(cdiv64(total, 256), 65535)), dim3(256), 0, s, w, out, O,
                       I, c0, nc, taps);
    FS_HIP(hipGetLastError());
    return 0;
}

// -------------------------------------------------------------------------------------------
// PSPNet head without the upsampled pyramid channels (model/pspnet.py:28-34, 70-73).
// The head's 3x3 conv is linear and so is the bilinear upsample of the four pooled maps p_b (b x b cells):
//   conv3x3(cat(f, up(p_1), up(p_2), up(p_3), up(p_6)))[y,x,o]
//     = conv3x3_f(f)[y,x,o] + sum_b sum_tap [tap inside the map] sum_cell a_b(y + r - 1, x + s - 1; cell) * Z_b[cell][tap][o]
//   with Z_b[cell][tap][o] = sum_c W[o][2048 + 512 i_b + c][tap] * p_b[c][cell]   (a [B b^2] x [9*512] x 512 GEMM)
// and a_b the bilinear (align_corners=True) weights of F.interpolate.  This kernel adds the second term to the raw main
// conv output and applies BatchNorm scale/shift + ReLU in place.  The 2048 pyramid channels of the reference's 4096-
// channel concat never exist: half of the head's GEMM and of its Winograd input transform disappear.
// Two passes, using that the bilinear weights are separable:
//   R_b[y][j][s][o] = sum_r [row y+r-1 inside] sum_i wy_b(y+r-1; i) Z_b[(i, j)][(r, s)][o]      ppm_rows_kernel (tiny)
//   term[y][x][o]   = sum_b sum_s [column x+s-1 inside] sum_j wx_b(x+s-1; j) R_b[y][j][s][o]    ppm_term_classify_kernel
// i.e. 6 products per row entry and then <= 24 per output, instead of <= 144 per output in one pass (0.12 -> see profiles).
// Finish pass: one block = PPM_P consecutive pixels of a row x all C channels (a float4 per thread).
// -------------------------------------------------------------------------------------------
constexpr int PPM_P = 6;    // pixels of one row per block in the finish pass
constexpr int PPM_J = 12;   // 1 + 2 + 3 + 6 source columns over the four levels
struct PpmTermParams {
    float* T; int ld;        // [B*H*W][ld]: raw conv output in, finished activations out
    const float* Z[4];       // [B*bin*bin][9*C]
    float* R;                // [B*H][PPM_J][3][C] scratch
    int bin[4], joff[4];     // joff: first column slot of the level in R
    float sy[4], sx[4];      // resize_scale(bin, H, 1), resize_scale(bin, W, 1)
    const float* scale; const float* shift;
    int B, H, W, C, relu;
    // the finished activations go straight into the classifier 1x1 conv + bias (model/pspnet.py:75; Dropout2d at :74 is the identity
    // in eval) and never back to T: logits [B][K][H][W]
    const float* cls_w; const float* cls_b; float* logits; int K;
};

__global__ __launch_bounds__(256) void ppm_rows_kernel(PpmTermParams p) {
    const int js = blockIdx.x, y = blockIdx.y, b = blockIdx.z;  // js = column slot * 3 + tap column
    const int slot = js / 3, s = js - slot * 3;
    const int c = threadIdx.x * 4;
    if (c >= p.C) return;
    int bi = 3;
    while (slot < p.joff[bi]) --bi;
    const int bin = p.bin[bi], j = slot - p.joff[bi];
    const float* Zb = p.Z[bi] + (size_t)b * bin * bin * 9 * p.C + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 3; ++r) {
        const int Y = y + r - 1;
        if (Y < 0 || Y >= p.H) continue;  // zero padding of the conv
        const LinCoord cy = lin_coord(Y, bin, p.sy[bi], 1);
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(Zb + ((size_t)(cy.i0 * bin + j) * 9 + r * 3 + s) * p.C);
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(Zb + ((size_t)(cy.i1 * bin + j) * 9 + r * 3 + s) * p.C);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(cy.w1, z1[e], fmaf(cy.w0, z0[e], acc[e]));
    }
    *reinterpret_cast<f32x4*>(p.R + (((size_t)(b * p.H + y) * PPM_J + slot) * 3 + s) * p.C + c) = acc;
}

__global__ __launch_bounds__(256) void ppm_term_classify_kernel(PpmTermParams p) {
    const int xt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int x0 = xt * PPM_P;
    const bool dup = (int)threadIdx.x * 4 >= p.C;       // a block is whole waves: lanes beyond C stay alive for the barriers / shuffles,
    const int c = dup ? p.C - 4 : (int)threadIdx.x * 4;  // recompute the last float4 and neither store nor contribute
    // The interpolation weights depend on (level, tap column, pixel) only, not on the channel: the block works the 4 x 3 x 6 x 6 table
    // wt[level][tap][jj][pixel] = weight of source column j0 + jj out once, in parallel (each of the block's 128+ channel threads
    // used to redo all 72 coordinate computations: the pass was VALU-bound at 40 us for 66 MB), and every thread reads it back from
    // LDS.  The source column grows with the pixel, so j0 / the column count of a (level, tap) come from its first / last valid pixel.
    __shared__ int s_j0[4][3], s_nj[4][3];
    __shared__ float wt[4][3][6][PPM_P + 2];
    for (int e = threadIdx.x; e < 4 * 3 * 6 * PPM_P; e += blockDim.x) {
        const int px = e % PPM_P, jj = (e / PPM_P) % 6, s = (e / (6 * PPM_P)) % 3, bi = e / (18 * PPM_P);
        // valid pixels of this block for tap column s: 0 <= x0 + px + s - 1 < W and x0 + px < W
        const int pf = max(0, 1 - s - x0), pl = min(PPM_P - 1, min(p.W - 1 - x0, p.W - x0 - s));
        float w = 0.f;
        int j0 = 0, nj = 0;
        if (pf <= pl) {
            const LinCoord cf = lin_coord(x0 + pf + s - 1, p.bin[bi], p.sx[bi], 1), cl = lin_coord(x0 + pl + s - 1, p.bin[bi], p.sx[bi], 1);
            j0 = cf.i0;
            nj = cl.i1 - cf.i0 + 1;  // <= bin <= 6
            if (px >= pf && px <= pl) {
                const LinCoord cx = lin_coord(x0 + px + s - 1, p.bin[bi], p.sx[bi], 1);
                const int j = j0 + jj;
                w = (j == cx.i0 ? cx.w0 : 0.f) + (j == cx.i1 ? cx.w1 : 0.f);
            }
        }
        wt[bi][s][jj][px] = w;
        if (px == 0 && jj == 0) {
            s_j0[bi][s] = j0;
            s_nj[bi][s] = nj;
        }
    }
    __syncthreads();
    f32x4 acc[PPM_P];
#pragma unroll
    for (int px = 0; px < PPM_P; ++px) acc[px] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* Rrow = p.R + (size_t)(b * p.H + y) * PPM_J * 3 * p.C + c;
    for (int bs = 0; bs < 12; ++bs) {
        const int bi = bs / 3, s = bs - bi * 3;
        const int j0 = s_j0[bi][s], nj = s_nj[bi][s];
        for (int jj = 0; jj < nj; ++jj) {
            const f32x4 z = *reinterpret_cast<const f32x4*>(Rrow + ((size_t)(p.joff[bi] + j0 + jj) * 3 + s) * p.C);
#pragma unroll
            for (int px = 0; px < PPM_P; ++px) {
                const float w = wt[bi][s][jj][px];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[px][e] = fmaf(w, z[e], acc[px][e]);
            }
        }
    }
    const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    // The finished values stay in registers and meet the classifier rows there (fixed summation order that depends on nothing but C:
    // a frame's logits do not depend on its batch).  T is read once and never written back.
#pragma unroll
    for (int px = 0; px < PPM_P; ++px) {
        const bool in = x0 + px < p.W && !dup;
        const f32x4 v = in ? *reinterpret_cast<const f32x4*>(p.T + ((size_t)(b * p.H + y) * p.W + x0 + px) * p.ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[px][e] = (v[e] + acc[px][e]) * sc[e] + sh[e];
            if (p.relu) acc[px][e] = fmaxf(acc[px][e], 0.f);
            if (dup) acc[px][e] = 0.f;
        }
    }
    // Per class k and pixel a thread has the dot product of its 4 channels.  Sixteen lanes are added in registers (four DPP steps:
    // quad swaps, then row shifts by 4 and 8 -- a shuffle tree through ds_bpermute cost three instructions per step and value), lane
    // 15 of every row parks the row's sum in LDS and one thread per value adds the block's rows in order.
    constexpr int KC = 8;  // classes per pass
    __shared__ float part[PPM_P * KC][16];
    const int nrows = blockDim.x >> 4;
    auto dpp_add = [](float v, auto ctrl) {
        return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        const int kc = min(KC, p.K - k0);
        for (int k = 0; k < kc; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(p.cls_w + (size_t)(k0 + k) * p.C + c);
#pragma unroll
            for (int px = 0; px < PPM_P; ++px) {
                float d = acc[px][0] * w[0] + acc[px][1] * w[1] + acc[px][2] * w[2] + acc[px][3] * w[3];
                d = dpp_add(d, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
                d = dpp_add(d, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]: every lane of a quad has the quad's sum
                d = dpp_add(d, std::integral_constant<int, 0x114>{});  // row_shr:4
                d = dpp_add(d, std::integral_constant<int, 0x118>{});  // row_shr:8: lanes 12-15 of a row have the row's sum
                if ((threadIdx.x & 15) == 15) part[px * KC + k][threadIdx.x >> 4] = d;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < PPM_P * kc) {
            const int px = threadIdx.x / kc, k = threadIdx.x % kc;
            float sum = part[px * KC + k][0];
            for (int r = 1; r < nrows; ++r) sum += part[px * KC + k][r];
            if (x0 + px < p.W) p.logits[(((size_t)b * p.K + k0 + k) * p.H + y) * p.W + x0 + px] = sum + (p.cls_b ? p.cls_b[k0 + k] : 0.f);
        }
        __syncthreads();
    }
}

size_t ppm_term_scratch_floats(int B, int H, int C) { return (size_t)B * H * PPM_J * 3 * C; }

namespace {
int ppm_term_any(float* T, int ld, const float* const Z[4], const int bins[4], float* scratch, const float* scale, const float* shift, int B, int H,
                 int W, int C, int relu, const float* cls_w, const float* cls_b, float* logits, int K, hipStream_t s) {
    FS_REQUIRE(C % 4 == 0 && C <= 1024 && ld % 4 == 0 && ((uintptr_t)T & 15) == 0, "ppm_term_classify: C must be a multiple of 4, <= 1024");
    FS_REQUIRE(logits && cls_w && K >= 1 && ((uintptr_t)cls_w & 15) == 0, "ppm_term_classify: bad classifier");
    FS_REQUIRE(H <= 65535 && B <= 65535, "ppm_term_classify: map too large for the launch grid");
    FS_REQUIRE(bins[0] + bins[1] + bins[2] + bins[3] == PPM_J && scratch, "ppm_term_classify: pyramid levels must add up to %d columns", PPM_J);
    PpmTermParams p{};
    p.T = T; p.ld = ld; p.R = scratch; p.scale = scale; p.shift = shift;
    p.B = B; p.H = H; p.W = W; p.C = C; p.relu = relu;
    int jo = 0;
    for (int i = 0; i < 4; ++i) {
        p.Z[i] = Z[i];
        p.bin[i] = bins[i];
        p.joff[i] = jo;
        jo += bins[i];
        p.sy[i] = resize_scale(bins[i], H, 1);
        p.sx[i] = resize_scale(bins[i], W, 1);
    }
    p.cls_w = cls_w; p.cls_b = cls_b; p.logits = logits; p.K = K;
    const dim3 block(((C / 4 + 63) / 64) * 64);
    hipLaunchKernelGGL(ppm_rows_kernel, dim3(PPM_J * 3, H, B), block, 0, s, p);
    FS_HIP(hipGetLastError());
    hipLaunchKernelGGL(ppm_term_classify_kernel, dim3(cdiv(W, PPM_P), H, B), block, 0, s, p);
    FS_HIP(hipGetLastError());
    return 0;
}
}  // namespace

int launch_ppm_term_classify(const float* T, int ld, const float* const Z[4], const int bins[4], float* scratch, const float* scale,
                             const float* shift, int B, int H, int W, int C, int relu, const float* cls_w, const float* cls_b, float* logits_nchw,
                             int K, hipStream_t s) {
    FS_REQUIRE(logits_nchw, "ppm_term_classify: no output");
    return ppm_term_any(const_cast<float*>(T), ld, Z, bins, scratch, scale, shift, B, H, W, C, relu, cls_w, cls_b, logits_nchw, K, s);
}

// 32x32 LDS-tiled transposes between [B][C][HW] and [B][HW][ld]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int ld_out, int C,
                                                           int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, px = p0 + tx;
        tile[r][tx] = (c < C && px < HW) ? in[((size_t)b * C + c) * HW + px] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int px = p0 + r, c = c0 + tx;
        if (px < HW && c < C) out[((size_t)b * HW + px) * ld_out + c] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int C,
                                                           int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int px = p0 + r, c = c0 + tx;
        tile[r][tx] = (c < C && px < HW) ? in[((size_t)b * HW + px) * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, px = p0 + tx;
        if (px < HW && c < C) out[((size_t)b * C + c) * HW + px] = tile[tx][r];
    }
}

int launch_nchw_to_nhwc(const float* in, float* out, int ld_out, int B, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, s, in, out, ld_out, C, HW);
    FS_HIP(hipGetLastError());
    return 0;
}
int launch_nhwc_to_nchw(const float* in, int ld_in, float* out, int B, int C, int HW, hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), B), dim3(256), 0, s, in, ld_in, out, C, HW);
    FS_HIP(hipGetLastError());
    return 0;
}

}  // namespace fs
