// extern "C" surface declared in include/floodseg.h.  Plain pointers and sizes only.
#include "../../include/floodseg.h"
#include "kernels.h"
#include "net.h"

#include <algorithm>
#include <cmath>

#define FS_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t S(fs_stream s) { return reinterpret_cast<hipStream_t>(s); }

FS_API int fs_version(void) { return 500; }
FS_API const char* fs_last_error(void) { return fs::last_error().c_str(); }

FS_API int fs_create(const fs_config* cfg, fs_handle* out) { return fs::net_create(cfg, out); }
FS_API int fs_destroy(fs_handle h) { return fs::net_destroy(h); }
FS_API int fs_load_weight(fs_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                          fs_stream stream) {
    return fs::net_load_weight(h, name, data, shape, ndim, on_device, S(stream));
}
FS_API int fs_finalize(fs_handle h, fs_stream stream) { return fs::net_finalize(h, S(stream)); }
FS_API int fs_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw) { return fs::net_feature_shape(h, H, W, C, fh, fw); }
FS_API size_t fs_workspace_bytes(fs_handle h, int B, int H, int W) { return fs::net_workspace_bytes(h, B, H, W); }
FS_API int fs_reserve(fs_handle h, int B, int H, int W, fs_stream stream) { return fs::net_reserve(h, B, H, W, S(stream)); }
FS_API size_t fs_reserved_bytes(fs_handle h) { return fs::net_reserved_bytes(h); }
FS_API int fs_encoder_forward(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nhwc, fs_stream stream) {
    return fs::net_encoder(h, fs::frames_plain(in_nchw, nullptr, B), B, H, W, out_nhwc, S(stream));
}
FS_API int fs_encoder_forward2(fs_handle h, const float* in_a, int Ba, const float* in_b, int Bb, int H, int W, float* out_nhwc,
                               fs_stream stream) {
    if (Ba < 0 || Bb < 0) return fs::fail("fs_encoder_forward2: negative batch");
    return fs::net_encoder(h, fs::frames_plain(in_a, Bb ? in_b : nullptr, Ba), Ba + Bb, H, W, out_nhwc, S(stream));
}
FS_API int fs_decoder_forward(fs_handle h, const float* feat_nhwc, int B, int fh, int fw, float* out_nchw, fs_stream stream) {
    return fs::net_decoder(h, feat_nhwc, B, fh, fw, out_nchw, S(stream));
}
FS_API int fs_segment_forward(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nchw, fs_stream stream) {
    return fs::net_segment(h, fs::frames_plain(in_nchw, nullptr, B), B, H, W, out_nchw, S(stream));
}
FS_API int fs_segment_forward2(fs_handle h, const float* in_a, int Ba, const float* in_b, int Bb, int H, int W, float* out_nchw,
                               fs_stream stream) {
    if (Ba < 0 || Bb < 0) return fs::fail("fs_segment_forward2: negative batch");
    return fs::net_segment(h, fs::frames_plain(in_a, Bb ? in_b : nullptr, Ba), Ba + Bb, H, W, out_nchw, S(stream));
}
FS_API int fs_segment_crops(fs_handle h, const float* frame_a, const float* frame_b, int FH, int FW, int ncrops, const int* crop_y,
                            const int* crop_x, int ch, int cw, float* out_nchw, fs_stream stream) {
    if (!frame_a || !crop_y || !crop_x || ncrops < 1 || ncrops > 32 || FH < 1 || FW < 1 || FH > 32767 || FW > 32767)
        return fs::fail("fs_segment_crops: bad arguments (1..32 crops, frame at most 32767 px)");
    fs::FrameSrc src{};
    src.in = frame_a;
    src.in2 = frame_b;
    src.ncrops = ncrops;
    src.FH = FH;
    src.FW = FW;
    for (int c = 0; c < ncrops; ++c) {
        if (crop_y[c] < 0 || crop_x[c] < 0 || crop_y[c] + ch > FH || crop_x[c] + cw > FW)
            return fs::fail("fs_segment_crops: crop %d at (%d,%d) size %dx%d leaves the %dx%d frame", c, crop_y[c], crop_x[c], ch, cw, FH, FW);
        src.cy[c] = (short)crop_y[c];
        src.cx[c] = (short)crop_x[c];
    }
    return fs::net_segment(h, src, frame_b ? 2 * ncrops : ncrops, ch, cw, out_nchw, S(stream));
}
FS_API int fs_profile_enable(fs_handle h, int on) {
    if (!h) return fs::fail("fs_profile_enable: null handle");
    h->profiling = on != 0;
    return 0;
}
FS_API int fs_profile_dump(fs_handle h, char* buf, size_t buflen) { return fs::net_profile_dump(h, buf, buflen); }

FS_API int fs_grid_sample_nchw(const float* in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg, float* out,
                               int align_corners, fs_stream stream) {
    if (!in || !grid || !out || B < 1 || C < 1 || Hi < 1 || Wi < 1 || Hg < 1 || Wg < 1) return fs::fail("fs_grid_sample_nchw: bad arguments");
    return fs::launch_grid_sample_nchw(in, B, C, Hi, Wi, grid, Hg, Wg, out, align_corners, S(stream));
}
FS_API int fs_grid_sample_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg,
                               float* out, int ld_out, int align_corners, fs_stream stream) {
    if (!in || !grid || !out || B < 1 || C < 1 || Hi < 1 || Wi < 1 || Hg < 1 || Wg < 1) return fs::fail("fs_grid_sample_nhwc: bad arguments");
    return fs::launch_grid_sample_nhwc(in, ld_in, B, C, Hi, Wi, grid, Hg, Wg, out, ld_out, align_corners, S(stream));
}
FS_API int fs_resize_bilinear_nchw(const float* in, int BC, int Hi, int Wi, float* out, int Ho, int Wo, int align_corners,
                                   fs_stream stream) {
    if (!in || !out || BC < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return fs::fail("fs_resize_bilinear_nchw: bad arguments");
    return fs::launch_resize_bilinear_nchw(in, BC, Hi, Wi, out, Ho, Wo, align_corners, S(stream));
}
FS_API int fs_resize_bilinear_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, float* out, int ld_out, int Ho,
                                   int Wo, int align_corners, fs_stream stream) {
    if (!in || !out || B < 1 || C < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return fs::fail("fs_resize_bilinear_nhwc: bad arguments");
    return fs::launch_resize_bilinear_nhwc(in, ld_in, B, C, Hi, Wi, out, ld_out, Ho, Wo, align_corners, S(stream));
}
FS_API int fs_blend(const float* a, float wa, const float* b, float wb, float* out, int64_t numel, fs_stream stream) {
    if (!a || !out || numel < 0) return fs::fail("fs_blend: bad arguments");
    if (numel == 0) return 0;
    return fs::launch_blend(a, wa, b, wb, out, numel, S(stream));
}

FS_API int fs_seg_tail(const float* lo_prev, const float* lo_next, const float* const* grids_left,
                       const float* const* grids_right, int K, int h, int w, int Hg, int Wg, int H, int W, int n, int no_warp,
                       float* out_logits, uint8_t* out_mask, float* scratch, fs_stream stream) {
    if (!lo_prev || h < 1 || w < 1 || H < 1 || W < 1) return fs::fail("fs_seg_tail: bad arguments");
    fs::SegTailParams p{};
    p.lo_prev = lo_prev;
    p.lo_next = lo_next;
    p.grids_left = grids_left;
    p.grids_right = grids_right;
    p.K = K;
    p.h = h;
    p.w = w;
    p.Hg = Hg;
    p.Wg = Wg;
    p.H = H;
    p.W = W;
    p.n = n;
    p.no_warp = no_warp;
    p.out_logits = out_logits;
    p.out_mask = out_mask;
    p.scratch = scratch;
    return fs::launch_seg_tail(p, S(stream));
}

FS_API int fs_feat_tail(const float* f_prev, const float* f_next, int C, int fh, int fw, const float* const* grids_left,
                        const float* const* grids_right, int Hg, int Wg, const float* grid0, int H0, int W0, int n, int no_warp, float* stack,
                        float* scratch, fs_stream stream) {
    if (!f_prev || !stack || C < 1 || fh < 1 || fw < 1) return fs::fail("fs_feat_tail: bad arguments");
    fs::FeatTailParams p{};
    p.f_prev = f_prev;
    p.f_next = f_next;
    p.grids_left = grids_left;
    p.grids_right = grids_right;
    p.grid0 = grid0;
    p.C = C;
    p.fh = fh;
    p.fw = fw;
    p.Hg = Hg;
    p.Wg = Wg;
    p.H0 = H0;
    p.W0 = W0;
    p.n = n;
    p.no_warp = no_warp;
    p.stack = stack;
    p.scratch = scratch;
    return fs::launch_feat_tail(p, S(stream));
}

FS_API int fs_seg_tail_accumulate(const float* lo_prev, const float* lo_next, const float* const* grids_left,
                                  const float* const* grids_right, int K, int h, int w, int Hg, int Wg, int H, int W, int n, int no_warp,
                                  double* canvas, double* count, int cH, int cW, int y0, int x0, float* scratch, fs_stream stream) {
    if (!lo_prev || !canvas || !count || h < 1 || w < 1 || H < 1 || W < 1 || cH < 1 || cW < 1) return fs::fail("fs_seg_tail_accumulate: bad arguments");
    fs::SegTailParams p{};
    p.lo_prev = lo_prev;
    p.lo_next = lo_next;
    p.grids_left = grids_left;
    p.grids_right = grids_right;
    p.K = K;
    p.h = h;
    p.w = w;
    p.Hg = Hg;
    p.Wg = Wg;
    p.H = H;
    p.W = W;
    p.n = n;
    p.no_warp = no_warp;
    p.scratch = scratch;
    p.canvas = canvas;
    p.count = count;
    p.cH = cH;
    p.cW = cW;
    p.y0 = y0;
    p.x0 = x0;
    return fs::launch_seg_tail(p, S(stream));
}

FS_API int fs_argmax_u8(const float* in, int B, int K, int64_t HW, uint8_t* out, fs_stream stream) {
    if (!in || !out || B < 1 || HW < 1) return fs::fail("fs_argmax_u8: bad arguments");
    return fs::launch_argmax_u8(in, B, K, HW, out, S(stream));
}
FS_API int fs_resize_argmax_u8(const float* in, int B, int K, int Hi, int Wi, uint8_t* out, int Ho, int Wo, fs_stream stream) {
    if (!in || !out || B < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return fs::fail("fs_resize_argmax_u8: bad arguments");
    return fs::launch_resize_argmax_u8(in, B, K, Hi, Wi, out, Ho, Wo, S(stream));
}
FS_API int fs_iou_hist(const uint8_t* pred, const uint8_t* target, int64_t numel, int K, int ignore_index, long long* hist3K,
                       fs_stream stream) {
    if (!pred || !target || !hist3K || numel < 0) return fs::fail("fs_iou_hist: bad arguments");
    if (numel == 0) return 0;
    return fs::launch_iou_hist(pred, target, numel, K, ignore_index, hist3K, S(stream));
}

FS_API int fs_colorize(const uint8_t* mask, const uint8_t* palette, int K, uint8_t* rgb, int64_t numel, fs_stream stream) {
    if (!mask || !palette || !rgb || K < 1 || numel < 0) return fs::fail("fs_colorize: bad arguments");
    if (numel == 0) return 0;
    return fs::launch_colorize(mask, palette, K, rgb, numel, S(stream));
}
FS_API int fs_mv_to_grids(const int* mv, int n, int stride, int hb, int wb, int block, int H, int W, int* owners, double* grid,
                          double* inv_grid, fs_stream stream) {
    if ((n > 0 && !mv) || !owners || !grid || !inv_grid || n < 0) return fs::fail("fs_mv_to_grids: bad arguments");
    return fs::launch_mv_to_grids(mv, n, stride, hb, wb, block, H, W, owners, grid, inv_grid, S(stream));
}
FS_API int fs_softmax_accumulate(const float* logits, int n, int K, int h, int w, double* canvas, double* count, int H, int W,
                                 int y0, int x0, fs_stream stream) {
    if (!logits || !canvas || !count || n < 1 || K < 1 || h < 1 || w < 1) return fs::fail("fs_softmax_accumulate: bad arguments");
    return fs::launch_softmax_accumulate(logits, n, K, h, w, canvas, count, H, W, y0, x0, S(stream));
}
FS_API int fs_canvas_finish(double* canvas, const double* count, int n, int K, int64_t HW, uint8_t* mask, fs_stream stream) {
    if (!canvas || !count || n < 1 || K < 1 || HW < 1) return fs::fail("fs_canvas_finish: bad arguments");
    return fs::launch_canvas_finish(canvas, count, n, K, HW, mask, S(stream));
}

FS_API int fs_canvas_resize_argmax(const double* canvas, int n, int K, int Hi, int Wi, uint8_t* mask, int Ho, int Wo, fs_stream stream) {
    if (!canvas || !mask || n < 1 || K < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return fs::fail("fs_canvas_resize_argmax: bad arguments");
    return fs::launch_canvas_resize_argmax(canvas, n, K, Hi, Wi, mask, Ho, Wo, S(stream));
}
FS_API int fs_crop_grids(const float* const* grids, int ngrids, int Hg, int Wg, int H, int W, int ncrops, const int* crop_y,
                         const int* crop_x, int ch, int cw, float* out, fs_stream stream) {
    if (!grids || !crop_y || !crop_x || !out || ngrids < 1 || ncrops < 1 || Hg < 1 || Wg < 1 || H < 1 || W < 1 || ch < 16 || cw < 16)
        return fs::fail("fs_crop_grids: bad arguments (>= 1 grid, >= 1 crop, crop >= 16 px)");
    for (int j = 0; j < ngrids; ++j)
        if (!grids[j]) return fs::fail("fs_crop_grids: null grid %d", j);
    // flow/transform.py:223-233 in double, Python's round() = round-half-to-even = nearbyint in the default rounding mode
    const double ppb_h = (double)H / Hg, ppb_w = (double)W / Wg;
    const int fh = ch / 16, fw = cw / 16;  // flow/transform.py:226-227
    // one launch takes at most 32 crops x 32 grids (fixed-size kernel arguments): frame_delta = 25 (the reference's default,
    // flow/base.py:350) has 48 grids per window, a 2160 x 3840 frame 40 crops -- cut into slices, each written in place
    for (int c0 = 0; c0 < ncrops; c0 += 32) {
        for (int g0 = 0; g0 < ngrids; g0 += 32) {
            fs::CropGridParams p{};
            p.ngrids = std::min(32, ngrids - g0);
            p.ncrops = std::min(32, ncrops - c0);
            p.ng_total = ngrids;
            p.g0 = g0;
            for (int j = 0; j < p.ngrids; ++j) p.grids[j] = grids[g0 + j];
            p.Hg = Hg;
            p.Wg = Wg;
            p.H = H;
            p.W = W;
            p.fh = fh;
            p.fw = fw;
            p.out = out + (size_t)c0 * ngrids * fh * fw * 2;
            for (int c = 0; c < p.ncrops; ++c) {
                const int y = crop_y[c0 + c], x = crop_x[c0 + c];
                const int bho = (int)std::nearbyint(y / ppb_h), bwo = (int)std::nearbyint(x / ppb_w);
                const int bh = (int)std::nearbyint((y + ch) / ppb_h) - bho, bw = (int)std::nearbyint((x + cw) / ppb_w) - bwo;
                p.bho[c] = (short)bho;
                p.bwo[c] = (short)bwo;
                p.bh[c] = (short)bh;
                p.bw[c] = (short)bw;
                p.off_h[c] = (float)y;
                p.off_w[c] = (float)x;
                p.den_h[c] = (float)(bh * ppb_h);
                p.den_w[c] = (float)(bw * ppb_w);
            }
            if (int rc = fs::launch_crop_grids(p, S(stream))) return rc;
        }
    }
    return 0;
}

FS_API int fs_crops_fuse(const float* lo_prev, const float* lo_next, const float* crop_grids, int ncrops, const int* crop_y, const int* crop_x,
                         int K, int h, int w, int Hg, int Wg, int ch, int cw, int n, int no_warp, double* canvas, uint8_t* mask, int H, int W,
                         float* scratch, fs_stream stream) {
    if (!lo_prev || !crop_y || !crop_x || ncrops < 1 || ncrops > 64 || h < 1 || w < 1 || ch < 1 || cw < 1 || H < 1 || W < 1 || H > 32767 || W > 32767)
        return fs::fail("fs_crops_fuse: bad arguments (1..64 crops, frame at most 32767 px)");
    fs::CropsFuseParams p{};
    p.lo_prev = lo_prev;
    p.lo_next = lo_next;
    p.nc = ncrops;
    for (int c = 0; c < ncrops; ++c) {
        p.cy[c] = (short)crop_y[c];
        p.cx[c] = (short)crop_x[c];
    }
    p.K = K; p.h = h; p.w = w; p.Hg = Hg; p.Wg = Wg; p.ch = ch; p.cw = cw; p.n = n; p.no_warp = no_warp;
    p.canvas = canvas;
    p.mask = mask;
    p.H = H;
    p.W = W;
    return fs::launch_crops_fuse(p, crop_grids, scratch, S(stream));
}

FS_API int fs_pack_conv_weight(const float* oihw, float* ohwi, int O, int I, int KH, int KW, fs_stream stream) {
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
    // tile = workgroup tile id 0..7 (6, 7: split route only), optionally | FS_CONV_CHUNK_MAJOR; anything else is refused (no hidden experiment bits)
    constexpr int kMaxTile = 7;
    if ((tile & ~FS_CONV_CHUNK_MAJOR) < 0 || (tile & ~FS_CONV_CHUNK_MAJOR) > kMaxTile)
        return fs::fail("fs_conv2d_nhwc: tile must be 0..7, optionally | FS_CONV_CHUNK_MAJOR (got 0x%x)", tile);
    p.korder = (tile & FS_CONV_CHUNK_MAJOR) ? 1 : 0;
    return fs::launch_conv_igemm(p, S(stream), tile & ~FS_CONV_CHUNK_MAJOR);
}
FS_API int fs_conv2d_nhwc(const float* in, int ld_in, const float* wgt_ohwi, const float* scale, const float* shift,
                          const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream) {
    return conv2d_entry(in, ld_in, wgt_ohwi, nullptr, scale, shift, res, ld_res, out, ld_out, B, H, W, Cin, Cout, KH, KW, stride, pad, dil, relu,
                        tile, stream);
}
FS_API int fs_split_bf16x3(const float* w, int64_t n, void* planes, fs_stream stream) { return fs::launch_split_bf16x3(w, n, planes, S(stream)); }
FS_API int fs_conv2d_nhwc_split(const float* in, int ld_in, const void* wgt_planes, const float* scale, const float* shift,
                                const float* res, int ld_res, float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int KH,
                                int KW, int stride, int pad, int dil, int relu, int tile, fs_stream stream) {
    if (!wgt_planes) return fs::fail("fs_conv2d_nhwc_split: bad arguments");
    return conv2d_entry(in, ld_in, nullptr, wgt_planes, scale, shift, res, ld_res, out, ld_out, B, H, W, Cin, Cout, KH, KW, stride, pad, dil,
                        relu, tile, stream);
}
FS_API int fs_conv_chain_nhwc(const float* in, int K1, const float* in2, int K1b, const void* w1_planes, const float* scale1, const float* shift1,
                              const float* res, float* mid, int C1, int relu1, const void* w2_planes, const float* scale2, const float* shift2,
                              float* out, int C2, int relu2, int M, int tile, fs_stream stream) {
    if (!in || !w1_planes || !mid || !w2_planes || !out || M < 1 || K1 < 32 || C1 < 1 || C2 < 1 || (in2 != nullptr) != (K1b > 0) || (in2 && res))
        return fs::fail("fs_conv_chain_nhwc: bad arguments");
    if (!(tile == 0 || tile == 1 || tile == 2 || tile == 3 || tile == 6)) return fs::fail("fs_conv_chain_nhwc: tile must be 0, 1, 2, 3 or 6 (got %d)", tile);
    fs::ConvParams a{}, b{};
    a.in = in; a.ld_in = K1; a.wgt = (const float*)w1_planes; a.wgt3 = w1_planes; a.plane_bytes = (unsigned)((size_t)C1 * (K1 + K1b) * 2);
    a.scale = scale1; a.shift = shift1; a.res = res; a.ld_res = C1; a.out = mid; a.ld_out = C1;
    a.B = 1; a.H = M; a.W = 1; a.Cin = K1; a.Ho = M; a.Wo = 1; a.Cout = C1; a.KH = a.KW = 1; a.stride = 1; a.dil = 1; a.relu = relu1;
    if (in2) { a.in2 = in2; a.ld_in2 = K1b; a.Cin2 = K1b; a.stride2 = 1; a.H2 = M; a.W2 = 1; }
    b.in = mid; b.ld_in = C1; b.wgt = (const float*)w2_planes; b.wgt3 = w2_planes; b.plane_bytes = (unsigned)((size_t)C2 * C1 * 2);
    b.scale = scale2; b.shift = shift2; b.out = out; b.ld_out = C2;
    b.B = 1; b.H = M; b.W = 1; b.Cin = C1; b.Ho = M; b.Wo = 1; b.Cout = C2; b.KH = b.KW = 1; b.stride = 1; b.dil = 1; b.relu = relu2;
    return fs::launch_conv_chain(a, b, S(stream), tile);
}
FS_API size_t fs_attention_workspace_floats(int B, int N, int heads, int split_operands) {
    if (B < 1 || N < 1 || heads < 1) return 0;
    return fs::attention_scratch_floats(B, N, heads) + (split_operands ? fs::attention_split_floats(B, N, heads) + 64 : 0) + 64;
}
FS_API int fs_attention(const float* qkv, float* out, int B, int N, int heads, float scale, int split_operands, float* workspace, fs_stream stream) {
    if (!qkv || !out || !workspace || B < 1 || N < 1 || heads < 1) return fs::fail("fs_attention: bad arguments");
    const size_t sc = fs::attention_scratch_floats(B, N, heads);
    float* scratch = sc ? workspace : nullptr;
    if (!split_operands) return fs::launch_attention_f32(qkv, out, B, N, heads, scale, scratch, S(stream));
    float* planes = workspace + sc;
    planes += (64 - ((uintptr_t)planes / 4) % 64) % 64;  // 256-B aligned
    return fs::launch_attention_split(qkv, out, B, N, heads, scale, scratch, planes, S(stream), split_operands == 2);
}
FS_API size_t fs_winograd_workspace_floats(int B, int H, int W, int Cin, int Cout, int dil, int tile_m) {
    if (B < 1 || H < 1 || W < 1 || dil < 1 || !(tile_m == 0 || tile_m == 4 || tile_m == 6)) return 0;
    const int mt = tile_m ? tile_m : fs::winograd_pick_m(B, H, W, dil);
    const size_t G = (size_t)(mt + 2) * (mt + 2), T = (size_t)fs::winograd_tiles(B, H, W, dil, mt);
    return G * T * ((size_t)Cin + (size_t)Cout) + G * (size_t)Cout * Cin;
}
FS_API int fs_conv3x3_winograd_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift,
                                    float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int dil, int relu, int tile_m,
                                    float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !out || !workspace || B < 1 || H < 1 || W < 1 || dil < 1 || Cin % 32 != 0 || Cout % 4 != 0 ||
        !(tile_m == 0 || tile_m == 4 || tile_m == 6))
        return fs::fail("fs_conv3x3_winograd_nhwc: bad arguments (Cin %% 32, Cout %% 4, tile_m in {0, 4, 6} required)");
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
FS_API int fs_gemm_bf16x3_planes(const void* a_planes, int64_t a_plane_elems, int ld_a, const void* b_planes, int64_t b_plane_elems, int ld_b,
                                 const float* scale, const float* shift, float* out, int ld_out, int M, int N, int K, int relu, int groups,
                                 int64_t g_a, int64_t g_b, int64_t g_out, int bn, fs_stream stream) {
    if (!a_planes || !b_planes || !out || a_plane_elems < 1 || b_plane_elems < 1 || a_plane_elems * 2 >= ((int64_t)1 << 31) ||
        b_plane_elems * 2 >= ((int64_t)1 << 31) || !((bn & 0xff) == 0 || (bn & 0xff) == 64 || (bn & 0xff) == 128) || bn < 0 || relu < 0 || relu > 2)
        return fs::fail("fs_gemm_bf16x3_planes: bad arguments (planes below 2 GiB, bn in {0, 64, 128}, relu in {0, 1, 2})");
    fs::PlaneGemmParams p{};
    p.a3 = a_planes; p.a_plane_bytes = (unsigned)(a_plane_elems * 2); p.ld_a = ld_a;
    p.b3 = b_planes; p.b_plane_bytes = (unsigned)(b_plane_elems * 2); p.ld_b = ld_b;
    p.scale = scale; p.shift = shift; p.out = out; p.ld_out = ld_out;
    p.M = M; p.N = N; p.K = K; p.relu = relu;
    p.groups = groups; p.g_a = g_a; p.g_b = g_b; p.g_out = g_out;
    return fs::launch_gemm_planes(p, S(stream), bn);
}
FS_API size_t fs_winograd_planes_workspace_floats(int B, int H, int W, int Cin, int Cout, int dil, int tile_m) {
    if (B < 1 || H < 1 || W < 1 || dil < 1 || Cin < 1 || Cout < 1 || !(tile_m == 0 || tile_m == 4 || tile_m == 6)) return 0;
    const int mt = tile_m ? tile_m : fs::winograd_pick_m(B, H, W, dil);
    const size_t G = (size_t)(mt + 2) * (mt + 2), T = (size_t)fs::winograd_tiles(B, H, W, dil, mt);
    // U (fp32) + its planes + the planes of V + M (fp32); every block a multiple of 8 floats
    return G * (size_t)Cout * Cin * 5 / 2 + G * T * (size_t)Cin * 3 / 2 + G * T * (size_t)Cout + 64;
}
FS_API int fs_conv3x3_winograd_planes_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift,
                                           float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int dil, int relu, int tile_m,
                                           float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !out || !workspace || B < 1 || H < 1 || W < 1 || dil < 1 || Cin % 32 != 0 || Cout % 4 != 0 ||
        !(tile_m == 0 || tile_m == 4 || tile_m == 6))
        return fs::fail("fs_conv3x3_winograd_planes_nhwc: bad arguments (Cin %% 32, Cout %% 4, tile_m in {0, 4, 6} required)");
    const int mt = tile_m ? tile_m : fs::winograd_pick_m(B, H, W, dil);
    const size_t G = (size_t)(mt + 2) * (mt + 2), T = (size_t)fs::winograd_tiles(B, H, W, dil, mt);
    const size_t u_elems = G * (size_t)Cout * Cin, v_elems = G * T * (size_t)Cin;
    float* U = workspace;
    float* U3 = U + u_elems;
    float* V3 = U3 + (3 * u_elems + 1) / 2 + 8 - ((3 * u_elems + 1) / 2) % 8;
    float* Mb = V3 + (3 * v_elems + 1) / 2 + 8 - ((3 * v_elems + 1) / 2) % 8;
    if (int rc = fs::launch_winograd_filter(wgt_oihw, U, Cout, Cin, mt, S(stream))) return rc;
    if (int rc = fs::launch_split_bf16x3(U, (long long)u_elems, U3, S(stream))) return rc;
    if (int rc = fs::launch_winograd_input_planes(in, ld_in, V3, (long long)v_elems, B, H, W, Cin, dil, mt, S(stream))) return rc;
    fs::PlaneGemmParams p{};
    if (int rc = fs::winograd_plane_gemm_params(p, mt, (int)T, Cin, Cout, V3, U3, Mb)) return rc;
    if (int rc = fs::launch_gemm_planes(p, S(stream))) return rc;
    return fs::launch_winograd_output(Mb, scale, shift, out, ld_out, B, H, W, Cout, relu, dil, mt, S(stream));
}
FS_API size_t fs_winograd_fused_workspace_floats(int Cin, int Cout) { return fs::wino_fused_bank_floats(Cin, Cout); }
FS_API int fs_conv3x3_winograd_fused_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift,
                                          float* out, int ld_out, int B, int H, int W, int Cin, int Cout, int relu, int variant,
                                          float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !out || !workspace || B < 1 || H < 1 || W < 1 || variant < 0 || variant > 3 ||
        !fs::wino_fused_supported(Cin, Cout, 3, 3, 1, 1, 1))
        return fs::fail("fs_conv3x3_winograd_fused_nhwc: bad arguments (Cin %% 32 == 0, 32 <= Cin <= 256, Cout %% 64 == 0, variant 0..3)");
    if (int rc = fs::launch_wino4_filter_packed(wgt_oihw, workspace, Cout, Cin, S(stream))) return rc;
    return fs::launch_wino4_fused(in, ld_in, workspace, scale, shift, out, ld_out, B, H, W, Cin, Cout, relu, S(stream), variant);
}
FS_API int fs_conv3x3_winograd_fused_pool_nhwc(const float* in, int ld_in, const float* wgt_oihw, const float* scale, const float* shift, float* pool,
                                               int B, int H, int W, int Cin, int Cout, float* workspace, fs_stream stream) {
    if (!in || !wgt_oihw || !pool || !workspace || B < 1 || H < 1 || W < 1 || !fs::wino_fused_supported(Cin, Cout, 3, 3, 1, 1, 1))
        return fs::fail("fs_conv3x3_winograd_fused_pool_nhwc: bad arguments (Cin %% 32 == 0, 32 <= Cin <= 256, Cout %% 64 == 0)");
    if (int rc = fs::launch_wino4_filter_packed(wgt_oihw, workspace, Cout, Cin, S(stream))) return rc;
    return fs::launch_wino4_fused_pool(in, ld_in, workspace, scale, shift, pool, Cout, B, H, W, Cin, Cout, S(stream));
}
static int stem_entry(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift, float* out_nhwc, int B, int H, int W, int Cout,
                      int KH, int KW, int stride, int pad, int split, fs_stream stream);
FS_API int fs_stem_conv_nchw(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift,
                             float* out_nhwc, int B, int H, int W, int Cout, int KH, int KW, int stride, int pad,
                             fs_stream stream) {
    return stem_entry(in_nchw, wgt_hwio, scale, shift, out_nhwc, B, H, W, Cout, KH, KW, stride, pad, 0, stream);
}
FS_API int fs_stem_conv_nchw_split(const float* in_nchw, const float* wgt_hwio, const float* scale, const float* shift,
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
FS_API int fs_maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, fs_stream stream) {
    if (!in || !out || B < 1 || H < 1 || W < 1) return fs::fail("fs_maxpool3x3s2_nhwc: bad arguments");
    return fs::launch_maxpool3x3s2(in, C, out, C, B, H, W, C, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1, S(stream));
}
FS_API int fs_adaptive_avgpool_nhwc(const float* in, int ld_in, float* out, int B, int H, int W, int C, int bin, fs_stream stream) {
    if (!in || !out || B < 1 || H < 1 || W < 1 || bin < 1) return fs::fail("fs_adaptive_avgpool_nhwc: bad arguments");
    return fs::launch_adaptive_avgpool(in, ld_in, out, B, H, W, C, bin, S(stream));
}
FS_API int fs_nchw_to_nhwc(const float* in, float* out, int B, int C, int HW, fs_stream stream) {
    if (!in || !out || B < 1 || C < 1 || HW < 1) return fs::fail("fs_nchw_to_nhwc: bad arguments");
    return fs::launch_nchw_to_nhwc(in, out, C, B, C, HW, S(stream));
}
FS_API int fs_nhwc_to_nchw(const float* in, float* out, int B, int C, int HW, fs_stream stream) {
    if (!in || !out || B < 1 || C < 1 || HW < 1) return fs::fail("fs_nhwc_to_nchw: bad arguments");
    return fs::launch_nhwc_to_nchw(in, C, out, B, C, HW, S(stream));
}
