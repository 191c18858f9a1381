// extern "C" surface declared in include/floodseg.h.  Plain pointers and sizes only.
#include "../../include/floodseg.h"
#include "kernels.h"
#include "net.h"

#include <algorithm>
#include <cmath>

#define FS_API extern "C" __attribute__((visibility("default")))

static inline hipStream_t S(fs_stream s) { return reinterpret_cast<hipStream_t>(s); }

FS_API int fs_version(void) { return 600; }
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

FS_API int fs_resize_crop(const float* in, int B, int K, int Hi, int Wi, int Hfull, int Wfull, int align_corners, float* logits,
                          uint8_t* mask, int Ho, int Wo, fs_stream stream) {
    if (!in || B < 1 || K < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1 || Hfull < 1 || Wfull < 1) return fs::fail("fs_resize_crop: bad arguments");
    return fs::launch_resize_crop(in, B, K, Hi, Wi, Hfull, Wfull, align_corners, logits, mask, Ho, Wo, S(stream));
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

