// Op-level test hooks (include/floodseg_test.h): the building blocks behind the networks, reachable for the parity tests and the
// measurement tools through ONE exported symbol, fs_test_hooks(), that returns a table of function pointers -- they are not part of the
// product's symbol surface (include/floodseg.h).
#include "../../include/floodseg_test.h"
#include "kernels.h"
#include "net.h"

#include <algorithm>
#include <cmath>

#define FS_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t S(fs_stream s) { return reinterpret_cast<hipStream_t>(s); }

static int fs_pack_conv_weight(const float* oihw, float* ohwi, int O, int I, int KH, int KW, fs_stream stream) {
    if (!oihw || !ohwi || O < 1 || I < 1 || KH < 1 || KW < 1) return fs::fail("fs_pack_conv_weight: bad arguments");
    return fs::launch_pack_oihw_to_ohwi(oihw, ohwi, O, I, KH, KW, S(stream));
}
static int conv2d_entry(const float* in, int ld_in, const float* wgt_ohwi, const void* wgt3, const float* scale, const float* shift,
                        const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream) {
    if (!in || (!wgt_ohwi && !wgt3) || !out || B < 1 || H < 1 || W < 1 || stride < 1 || dil < 1 || pad < 0)
        return fs::fail("fs_conv2d_nhwc: bad arguments");
    fs::ConvParams p{};
    p.in = in;
    p.ld_in = ld_in;
    p.wgt = wgt_ohwi;
    p.wgt3 = wgt3;
    p.plane_bytes = (unsigned)((size_t)Cout * KH * KW * Cin * 2);
    p.scale = scale;
    p.shift = shift;
    p.res = res;
    p.ld_res = ld_res;
    p.out = out;
    p.ld_out = ld_out;
    p.B = B;
    p.H = H;
    p.W = W;
    p.Cin = Cin;
    p.Cout = Cout;
    p.KH = KH;
    p.KW = KW;
    p.stride = stride;
    p.pad = pad;
    p.dil = dil;
    p.relu = relu;
    p.Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    p.Wo = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    if (p.Ho < 1 || p.Wo < 1) return fs::fail("fs_conv2d_nhwc: empty output");
    // tile = workgroup tile id 0..4 or 6 (6: split route only), optionally | FS_CONV_CHUNK_MAJOR; anything else is refused (no hidden experiment bits)
    const int tid = tile & ~FS_CONV_CHUNK_MAJOR;
    if (tid < 0 || tid > 6 || tid == 5) return fs::fail("fs_conv2d_nhwc: tile must be 0..4 or 6, optionally | FS_CONV_CHUNK_MAJOR (got 0x%x)", tile);
    p.korder = (tile & FS_CONV_CHUNK_MAJOR) ? 1 : 0;
    return fs::launch_conv_igemm(p, S(stream), tile & ~FS_CONV_CHUNK_MAJOR);
}
static int fs_conv2d_nhwc(const float* in, int ld_in, const float* wgt_ohwi, const float* scale, const float* shift,
                          const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream) {
    return conv2d_entry(in, ld_in, wgt_ohwi, nullptr, scale, shift, res, ld_res, out, ld_out, B, H, W, Cin, Cout, KH, KW, stride, pad, dil, relu,
                        tile, stream);
}
static int fs_split_bf16x3(const float* w, int64_t n, void* planes, fs_stream stream) { return fs::launch_split_bf16x3(w, n, planes, S(stream)); }
static int fs_conv2d_nhwc_split(const float* in, int ld_in, const void* wgt_planes, const float* scale, const float* shift,
                                const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                                int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream) {
    if (!wgt_planes) return fs::fail("fs_conv2d_nhwc_split: bad arguments");
    return conv2d_entry(in, ld_in, nullptr, wgt_planes, scale, shift, res, ld_res, out, ld_out, B, H, W, Cin, Cout, KH, KW, stride, pad, dil,
                        relu, tile, stream);
}
static size_t fs_attention_workspace_floats(int B, int N, int heads, int split_operands) {
    if (B < 1 || N < 1 || heads < 1) return 0;
    return fs::attention_scratch_floats(B, N, heads) + (split_operands ? fs::attention_split_floats(B, N, heads) + 64 : 0) + 64;
}
static int fs_attention(const float* qkv, float* out, int B, int N, int heads, float scale, int split_operands, float* workspace, fs_stream stream) {
    if (!qkv || !out || !workspace || B < 1 || N < 1 || heads < 1 || split_operands < 0 || split_operands > 1) return fs::fail("fs_attention: bad arguments");
    const size_t sc = fs::attention_scratch_floats(B, N, heads);
    float* scratch = sc ? workspace : nullptr;
    if (!split_operands) return fs::launch_attention_f32(qkv, out, B, N, heads, scale, scratch, S(stream));
    float* planes = workspace + sc;
    planes += (64 - ((uintptr_t)planes / 4) % 64) % 64;  // 256-B aligned
    return fs::launch_attention_split(qkv, out, B, N, heads, scale, scratch, planes, S(stream));
}
static size_t fs_winograd_workspace_floats(int B, int H, int W, int Cin, int Cout, int dil, int tile_m) {
    if (B < 1 || H < 1 || W < 1 || dil < 1 || !(tile_m == 0 || tile_m == 3 || tile_m == 4 || tile_m == 6)) return 0;
    const int mt = tile_m ? tile_m : fs::winograd_pick_m(B, H, W, dil);
    const size_t G = (size_t)(mt + 2) * (mt + 2), T = (size_t)fs::winograd_tiles(B, H, W, dil, mt);
    return G * T * ((size_t)Cin + (size_t)Cout) + G * (size_t)Cout * Cin;
}
static int fs_conv3x3_winograd_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift,
                                    float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int dil, int relu, int tile_m,
                                    float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !out || !workspace || B < 1 || H < 1 || W < 1 || dil < 1 || Cin % 32 != 0 || Cout % 4 != 0 ||
        !(tile_m == 0 || tile_m == 3 || tile_m == 4 || tile_m == 6))
        return fs::fail("fs_conv3x3_winograd_nhwc: bad arguments (Cin %% 32, Cout %% 4, tile_m in {0, 3, 4, 6} required)");
    const int mt = tile_m ? tile_m : fs::winograd_pick_m(B, H, W, dil);
    const size_t G = (size_t)(mt + 2) * (mt + 2), T = (size_t)fs::winograd_tiles(B, H, W, dil, mt);
    float* U = workspace;
    float* V = U + G * (size_t)Cout * Cin;
    float* Mb = V + G * T * Cin;
    if (int rc = fs::launch_winograd_filter(wgt_oihw, U, Cout, Cin, mt, S(stream))) return rc;
    if (int rc = fs::launch_winograd_input(in, ld_in, V, B, H, W, Cin, dil, mt, S(stream))) return rc;
    fs::ConvParams p{};
    p.in = V;
    p.ld_in = Cin;
    p.wgt = U;
    p.out = Mb;
    p.ld_out = Cout;
    p.B = 1;
    p.H = (int)T;
    p.W = 1;
    p.Cin = Cin;
    p.Ho = (int)T;
    p.Wo = 1;
    p.Cout = Cout;
    p.KH = p.KW = 1;
    p.stride = 1;
    p.dil = 1;
    p.groups = (int)G;
    p.g_wgt = (long long)Cout * Cin;
    fs::winograd_gemm_params(p, mt, (int)T, Cin, Cout);
    if (int rc = fs::launch_conv_igemm(p, S(stream))) return rc;
    return fs::launch_winograd_output(Mb, scale, shift, out, ld_out, B, H, W, Cout, relu, dil, mt, S(stream));
}
static size_t fs_winograd_fused_workspace_floats(int Cin, int Cout) { return fs::wino_fused_bank_floats(Cin, Cout); }
static int fs_conv3x3_winograd_fused_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift,
                                          float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int relu, int variant,
                                          float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !out || !workspace || B < 1 || H < 1 || W < 1 || variant < 0 || variant > 3 ||
        !fs::wino_fused_supported(Cin, Cout, 3, 3, 1, 1, 1))
        return fs::fail("fs_conv3x3_winograd_fused_nhwc: bad arguments (Cin %% 32 == 0, 32 <= Cin <= 256, Cout %% 64 == 0, variant 0..3)");
    if (int rc = fs::launch_wino4_filter_packed(wgt_oihw, workspace, Cout, Cin, S(stream))) return rc;
    return fs::launch_wino4_fused(in, ld_in, workspace, scale, shift, out, ld_out, B, H, W, Cin, Cout, relu, S(stream), variant);
}
static int fs_conv3x3_winograd_fused_pool_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* pool,
                                               int B, int H, int W, int Cin, int Cout, float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !pool || !workspace || B < 1 || H < 1 || W < 1 || !fs::wino_fused_supported(Cin, Cout, 3, 3, 1, 1, 1))
        return fs::fail("fs_conv3x3_winograd_fused_pool_nhwc: bad arguments (Cin %% 32 == 0, 32 <= Cin <= 256, Cout %% 64 == 0)");
    if (int rc = fs::launch_wino4_filter_packed(wgt_oihw, workspace, Cout, Cin, S(stream))) return rc;
    return fs::launch_wino4_fused_pool(in, ld_in, workspace, scale, shift, pool, Cout, B, H, W, Cin, Cout, S(stream));
}
static int stem_entry(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc, int B, int H, int W, int Cout,
                      int KH, int KW, int stride, int pad, int split, fs_stream stream);
static int fs_stem_conv_nchw(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift,
                             float* out_nhwc, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                             fs_stream stream) {
    return stem_entry(in_nchw, wgt_hwio, scale, shift, out_nhwc, B, H, W, Cout, KH, KW, stride, pad, 0, stream);
}
static int fs_stem_conv_nchw_split(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift,
                                   float* out_nhwc, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                                   fs_stream stream) {
    return stem_entry(in_nchw, wgt_hwio, scale, shift, out_nhwc, B, H, W, Cout, KH, KW, stride, pad, 1, stream);
}
static int stem_entry(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc, int B, int H, int W, int Cout,
                      int KH, int KW, int stride, int pad, int split, fs_stream stream) {
    if (!in_nchw || !wgt_hwio || !scale || !shift || !out_nhwc || B < 1) return fs::fail("fs_stem_conv_nchw: bad arguments");
    fs::StemParams p{};
    p.src = fs::frames_plain(in_nchw, nullptr, B);
    p.wgt = wgt_hwio;
    p.scale = scale;
    p.shift = shift;
    p.out = out_nhwc;
    p.ld_out = Cout;
    p.B = B;
    p.H = H;
    p.W = W;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.Cout = Cout;
    p.KH = KH;
    p.KW = KW;
    p.stride = stride;
    p.pad = pad;
    p.split = split;
    return fs::launch_stem_conv(p, S(stream));
}
static int fs_maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, fs_stream stream) {
    if (!in || !out || B < 1 || H < 1 || W < 1) return fs::fail("fs_maxpool3x3s2_nhwc: bad arguments");
    return fs::launch_maxpool3x3s2(in, C, out, C, B, H, W, C, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, S(stream));
}
static int fs_adaptive_avgpool_nhwc(const float* in, int ld_in, float* out, int B, int H, int W, int C, int bin, fs_stream stream) {
    if (!in || !out || B < 1 || H < 1 || W < 1 || bin < 1) return fs::fail("fs_adaptive_avgpool_nhwc: bad arguments");
    return fs::launch_adaptive_avgpool(in, ld_in, out, B, H, W, C, bin, S(stream));
}
static int fs_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, fs_stream stream) {
    if (!in || !out || B < 1 || C < 1 || HW < 1) return fs::fail("fs_nchw_to_nhwc: bad arguments");
    return fs::launch_nchw_to_nhwc(in, out, C, B, C, HW, S(stream));
}
static int fs_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, fs_stream stream) {
    if (!in || !out || B < 1 || C < 1 || HW < 1) return fs::fail("fs_nhwc_to_nchw: bad arguments");
    return fs::launch_nhwc_to_nchw(in, C, out, B, C, HW, S(stream));
}

FS_API const fs_test_api* fs_test_hooks(void) {
    static const fs_test_api api = {
        sizeof(fs_test_api),
        fs_pack_conv_weight,
        fs_conv2d_nhwc,
        fs_split_bf16x3,
        fs_conv2d_nhwc_split,
        fs_attention_workspace_floats,
        fs_attention,
        fs_winograd_workspace_floats,
        fs_conv3x3_winograd_nhwc,
        fs_winograd_fused_workspace_floats,
        fs_conv3x3_winograd_fused_nhwc,
        fs_conv3x3_winograd_fused_pool_nhwc,
        fs_stem_conv_nchw,
        fs_stem_conv_nchw_split,
        fs_maxpool3x3s2_nhwc,
        fs_adaptive_avgpool_nhwc,
        fs_nchw_to_nhwc,
        fs_nhwc_to_nchw,
    };
    return &api;
}
