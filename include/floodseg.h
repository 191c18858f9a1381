/*
 * floodseg.h -- C ABI of the MI355X-native key-frame segmentation + flow-interpolation hot path.
 *
 * Drop-in boundary for lenke182/flood-uav-video-segmentation (citations are reference file:line):
 *   - the two callables the reference's FlowModel invokes on its network,
 *       self.model.encoder(x) / self.model.decoder(f)            flow/model.py:39-40,57-58,76-79,120,129,177,189-204
 *     for FlowPSPNet (model/pspnet.py:113-141) and FlowDeepLabv3 (model/deeplabv3.py:47-54);
 *   - the torch ops FlowModel applies around them,
 *       F.grid_sample(bilinear, border)                          flow/model.py:157,248
 *       F.interpolate(bilinear, align_corners=True)              flow/model.py:42,68,86,103,139,150,159,179,193,206,218,228
 *       (n-p)/n * a + p/n * b                                    flow/model.py:104,168,170,234-236
 *   - the post-processing inside the reference's timed region,
 *       F.interpolate(.., (1072,1920)); max(1)[1]; uint8         flow/base.py:275-277
 *   - the metric histogram intersectionAndUnionGPU               util/util.py:52-63
 *
 * Conventions
 *   - every pointer named *_dev / in / out is a DEVICE pointer (tensor.data_ptr()); nothing is freed or
 *     retained by the library except its own packed weights and workspace;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued on it.  fs_finalize, fs_reserve and fs_profile_dump synchronise the device.  A forward synchronises (and
 *     allocates) only when it has to GROW the library-owned workspace or build a Winograd filter bank for a geometry it has
 *     not seen: call fs_reserve(handle, B, H, W) once for the largest batch / frame size you will use and no later forward
 *     allocates, frees or synchronises (required before HIP-graph capture and before using one handle from several streams);
 *   - all functions return 0 on success; on failure a non-zero code, and fs_last_error() holds the text
 *     (the Python shim raises RuntimeError with it -- the reference itself only raises/asserts);
 *   - NCHW tensors are the reference's layout; "NHWC" tensors are pixel-major with an explicit pixel
 *     stride `ld` (in floats) so that channel slices of wider buffers can be addressed.
 */
#ifndef FLOODSEG_H_
#define FLOODSEG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fs_net* fs_handle;
typedef void* fs_stream;

enum { FS_ARCH_PSPNET = 0, FS_ARCH_DEEPLABV3 = 1, FS_ARCH_SEGMENTER = 2 };

typedef struct fs_config {
    int arch;        /* FS_ARCH_*                               flow/base.py:94-103            */
    int layers;      /* ResNet depth 50 | 101 | 152             model/pspnet.py:45-50          */
    int classes;     /* K                                       dataset/flow/config.yaml:2     */
    /* FS_ARCH_SEGMENTER only (ignored otherwise)               model/vit.py:13-56             */
    int patch;       /* patch size P (32 as shipped, 16 for ViT-S/16)                          */
    int d_model;     /* 768 | 384 ...; head_dim is fixed to 64 => heads = d_model / 64         */
    int n_layers;    /* encoder blocks (12)                                                    */
    int dec_layers;  /* mask-transformer blocks (2)                                            */
    int image_size;  /* construction size: pos_embed holds (image_size/P)^2 + 1 rows           */
    /* Explicit A/B options of the convolutional networks (0 = the shipped default).  They select between arithmetically
     * different but parity-tested evaluation routes; nothing is read from the process environment.                     */
    int flags;          /* FS_OPT_* bits                                                                                */
    int winograd_tile;  /* 0 = per-map choice of F(3x3,3x3) / F(4x4,3x3) / F(6x6,3x3); 3, 4 or 6 forces one tile size       */
} fs_config;

enum {
    FS_OPT_NO_WINOGRAD = 1,   /* every 3x3 conv on the direct implicit-GEMM kernel                                      */
    FS_OPT_NO_FUSED_HEAD = 2, /* fs_segment_forward = fs_decoder_forward(fs_encoder_forward(x)) over the 4096-ch concat */
    FS_OPT_NO_FUSED_SHORTCUT = 4, /* projection blocks: downsample and conv3 as two launches instead of one concatenated-K GEMM */
    FS_OPT_NO_FUSED_WINOGRAD = 8, /* the 3x3 convs with Cin <= 128 (deep stem, conv2 of layer1) on the direct kernel instead of the
                                    one-kernel Winograd F(4x4,3x3) (implied by FS_OPT_NO_WINOGRAD)                             */
    FS_OPT_NO_SPLIT_BF16 = 16, /* implicit-GEMM launches (1x1 / 3x3 convs, Winograd GEMMs, nn.Linear) on the fp32 matrix-core kernel
                                 (v_mfma_f32_32x32x2_f32) instead of the split-operand one: each fp32 operand as the exact sum of three
                                 bf16 terms, six cross products on the bf16 matrix cores, fp32 accumulation (fs_conv2d_nhwc_split) */
    FS_OPT_NO_FUSED_POOL = 256, /* round 5 A/B switch: the deep stem's last conv (layer0.6) and the max-pool behind it (model/resnet.py:114-117) as two
                                 launches instead of one (the one-kernel Winograd with MaxPool2d(3, 2, 1) in its epilogue: the 357 x 357 x 128 map
                                 is never written).  Bit-identical results either way */
    FS_OPT_NO_RES_TOUCH = 128, /* round 5 A/B switch: without it the split-operand conv kernels touch the lines of a bottleneck's shortcut tile
                                 (dead loads) before the last K chunk of their main loop, so that the epilogue's residual loads hit in L2.
                                 Same results bit for bit either way */
    FS_OPT_NO_FUSED_QKV = 1024 /* round 6 A/B switch (Segmenter): without it the qkv Linear of every block writes the attention's K planes and
                                 transposed V planes (three bf16 terms per value) from its epilogue; with it a separate pre-pass reads the fp32
                                 qkv rows back and writes them.  Same results bit for bit either way.
                                 (Bits 32, 64 and 512 were the opt-in experiment routes of rounds 4-5 -- pre-split Winograd operands, chained
                                 bottleneck launches, the wave-pipelined attention kernel; all measured null or slower, removed in round 6:
                                 fs_create refuses them.) */
};

/* 600: the handle API and the ops below; additions keep the number, a changed or removed signature raises it. */
int fs_version(void);
const char* fs_last_error(void);

/* ---- network lifecycle ------------------------------------------------------------------------- */
int fs_create(const fs_config* cfg, fs_handle* out);
int fs_destroy(fs_handle h);
/* Copy one state_dict tensor (canonical name) into the library.
 * on_device != 0: `data` is a device pointer. */
int fs_load_weight(fs_handle h, const char* name, const float* data, const int64_t* shape, int ndim, int on_device,
                   fs_stream stream);
/* Fold eval-mode BatchNorm into per-channel scale/shift, repack filters OIHW -> OHWI. Synchronises. */
int fs_finalize(fs_handle h, fs_stream stream);
/* Feature-map geometry produced by fs_encoder_forward for an H x W frame. */
int fs_feature_shape(fs_handle h, int H, int W, int* C, int* fh, int* fw);
/* Bytes of library-owned activation workspace a forward at this geometry needs (allocated lazily, or by fs_reserve). */
size_t fs_workspace_bytes(fs_handle h, int B, int H, int W);
/* Grow every library-owned block that fs_encoder_forward / fs_decoder_forward / fs_segment_forward(2) / fs_segment_crops over
 * at most B frames (crops) of H x W will touch -- activation buffers, pooled maps, the Winograd V / M workspace, the resized
 * position table of the Segmenter -- to its final size, and build (on `stream`) the Winograd filter banks this geometry
 * selects.  Synchronises the device.  After it, a forward at this geometry performs no hipMalloc / hipFree / device sync;
 * fs_reserved_bytes() is how a caller checks: it does not change across such forwards.  Blocks only ever grow: reserving
 * several geometries in turn leaves the handle ready for all of them. */
int fs_reserve(fs_handle h, int B, int H, int W, fs_stream stream);
/* Bytes of workspace + lazily built filter banks the handle holds right now (packed weights not included). */
size_t fs_reserved_bytes(fs_handle h);

/* model.encoder(x):  in NCHW [B,3,H,W]  ->  out NHWC [B,fh,fw,C] (ld = C; a channels_last torch tensor)
 * FS_ARCH_SEGMENTER: out = the ViT tokens after the final LayerNorm, cls token dropped, [B, gh*gw, D]
 * (segm/model/segmenter.py:37-42); decoder = MaskTransformer -> [B,K,gh,gw] (before the bilinear resize). */
int fs_encoder_forward(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nhwc, fs_stream stream);
/* model.decoder(f):  in NHWC [B,fh,fw,C]  ->  out NCHW [B,K,fh,fw] */
int fs_decoder_forward(fs_handle h, const float* feat_nhwc, int B, int fh, int fw, float* out_nchw, fs_stream stream);
/* model.decoder(model.encoder(x)) in one call -- the composition the segmentation-mode paths evaluate
 * (flow/model.py:39-40 forward_segmentation, :189-191 and :202-204 predict_segmentation; single-frame inference):
 * in NCHW [B,3,H,W] -> out NCHW [B,K,fh,fw] with (fh, fw) from fs_feature_shape.  Same result as the two calls above; the
 * PSPNet head skips the 4096-channel concat: model/pspnet.py:28-34 (PPM upsample + cat) and :70-73 (3x3 conv) are linear,
 * so the pyramid's share of the conv is evaluated on the pooled b x b maps and added before BatchNorm + ReLU. */
int fs_segment_forward(fs_handle h, const float* in_nchw, int B, int H, int W, float* out_nchw, fs_stream stream);
/* The same two calls on a batch held in TWO tensors: images 0..Ba-1 = in_a [Ba,3,H,W], images Ba..Ba+Bb-1 = in_b [Bb,3,H,W].
 * FlowModel hands its two key frames over as separate tensors (frame_prev, frame_next: flow/model.py:189-191 and :202-204 run
 * the network once on each); both go through the network as one batch here, read in place -- no concatenation copy.
 * Results are identical to the one-tensor calls on cat(in_a, in_b).  Bb may be 0 (in_b ignored). */
int fs_encoder_forward2(fs_handle h, const float* in_a, int Ba, const float* in_b, int Bb, int H, int W, float* out_nhwc,
                        fs_stream stream);
int fs_segment_forward2(fs_handle h, const float* in_a, int Ba, const float* in_b, int Bb, int H, int W, float* out_nchw,
                        fs_stream stream);

/* Sliding-crop route (the reference's default `no_cropping=False`, flow/base.py:182-209): decoder(encoder(.)) of `ncrops`
 * windows [ch x cw] of the full frame frame_a [1,3,FH,FW] -- and, when frame_b != NULL, of the same windows of frame_b -- as ONE
 * batch, read in place from the frames (flow/base.py:199-200 clones every crop; here none is copied).  crop_y / crop_x: host
 * arrays of the windows' top-left corners.  out: NCHW [ncrops * (frame_b ? 2 : 1), K, fh, fw], the crops of frame_a first;
 * image i of the result is bit-identical to fs_segment_forward on the cloned crop.  Convolutional networks only. */
int fs_segment_crops(fs_handle h, const float* frame_a, const float* frame_b, int FH, int FW, int ncrops, const int* crop_y,
                     const int* crop_x, int ch, int cw, float* out_nchw, fs_stream stream);

/* ---- per-op profiling (HIP events on `stream` around every launch of the next forward calls) -- */
int fs_profile_enable(fs_handle h, int on);
/* Synchronises the recorded events; writes one line per op: "name kernel flops bytes ms\n". */
int fs_profile_dump(fs_handle h, char* buf, size_t buflen);

/* ---- flow / interpolation ops -------------------------------------------------------------------- */
int fs_grid_sample_nchw(const float* in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg, float* out,
                        int align_corners, fs_stream stream);
int fs_grid_sample_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, const float* grid, int Hg, int Wg,
                        float* out, int ld_out, int align_corners, fs_stream stream);
int fs_resize_bilinear_nchw(const float* in, int BC, int Hi, int Wi, float* out, int Ho, int Wo, int align_corners,
                            fs_stream stream);
int fs_resize_bilinear_nhwc(const float* in, int ld_in, int B, int C, int Hi, int Wi, float* out, int ld_out, int Ho,
                            int Wo, int align_corners, fs_stream stream);
/* out = wa*a + wb*b (b may be NULL) */
int fs_blend(const float* a, float wa, const float* b, float wb, float* out, int64_t numel, fs_stream stream);

/* Fused tail of FlowModel.predict_segmentation (flow/model.py:184-241) from the two low-resolution
 * decoder outputs: upsample, (optional) warp chains at grid resolution, linear fusion, and optionally
 * the per-frame argmax.  grids_left/right: host arrays of n-1 device pointers [Hg,Wg,2] (ignored when
 * no_warp).  scratch: >= 2*(n-1)*K*Hg*Wg floats (warp mode).  lo_next may be NULL (single frame). */
int fs_seg_tail(const float* lo_prev, const float* lo_next, const float* const* grids_left,
                const float* const* grids_right, int K, int h, int w, int Hg, int Wg, int H, int W, int n, int no_warp,
                float* out_logits, uint8_t* out_mask, float* scratch, fs_stream stream);

/* Fused tail of FlowModel.predict_feature between the encoder and its one batched decoder call (flow/model.py:131-171), on the
 * NHWC feature maps fs_encoder_forward returns (C % 4 == 0, one image per key frame):
 *   warp chains at grid resolution (:135-151), stack[0] = up(grid_sample(f_prev, grid0 [H0,W0,2], align_corners=True)) (:154-159),
 *   stack[p] = (n-p)/n * up(fwd[p-1]) + p/n * up(bwd[n-p-1]) (:166-171), up = bilinear align_corners=True to fh x fw, skipped when the
 *   map already has that size (:138,:149,:158);  no_warp: stack[0] = f_prev, stack[p] = (n-p)/n * f_prev + p/n * f_next.
 * stack: NHWC [n][fh][fw][C] ([1][fh][fw][C] when f_next is NULL) = the decoder's batch; the upsampled chain maps and the H0 x W0
 * resample are never materialised.  Bit-identical to fs_grid_sample_nhwc -> fs_resize_bilinear_nhwc -> fs_blend per map.
 * grids_left / grids_right: host arrays of n-1 device pointers [Hg,Wg,2]; scratch: >= 2*(n-1)*Hg*Wg*C floats (warp mode, f_next given). */
int fs_feat_tail(const float* f_prev, const float* f_next, int C, int fh, int fw, const float* const* grids_left,
                 const float* const* grids_right, int Hg, int Wg, const float* grid0, int H0, int W0, int n, int no_warp, float* stack,
                 float* scratch, fs_stream stream);

/* The same tail feeding the sliding-crop canvas instead of returning logits: compute_predict_crop's softmax over K
 * (flow/base.py:226-234) of every output frame is added to canvas[n,K,cH,cW] (float64) at the crop's offset (y0, x0) and
 * count[cH,cW] += 1 over the crop (flow/base.py:204-205) -- fs_seg_tail + fs_softmax_accumulate without the [n,K,H,W] logits
 * in HBM, bit-identical to that pair.  Crops that overlap must be accumulated by successive calls on one stream. */
int fs_seg_tail_accumulate(const float* lo_prev, const float* lo_next, const float* const* grids_left,
                           const float* const* grids_right, int K, int h, int w, int Hg, int Wg, int H, int W, int n, int no_warp,
                           double* canvas, double* count, int cH, int cW, int y0, int x0, float* scratch, fs_stream stream);

int fs_argmax_u8(const float* in, int B, int K, int64_t HW, uint8_t* out, fs_stream stream);
int fs_resize_argmax_u8(const float* in, int B, int K, int Hi, int Wi, uint8_t* out, int Ho, int Wo, fs_stream stream);
/* F.interpolate(in, (Hfull, Wfull), bilinear, align_corners)[:, :, :Ho, :Wo] -- the Segmenter's upsample-then-unpad decoder tail
 * (segm/model/segmenter.py:45-46) -- in one launch: dense logits [B][K][Ho][Wo] (may be NULL) and / or their channel argmax
 * uint8 [B][Ho][Wo] (may be NULL).  Bit-identical to fs_resize_bilinear_nchw at the full size, cropped, then fs_argmax_u8. */
int fs_resize_crop(const float* in, int B, int K, int Hi, int Wi, int Hfull, int Wfull, int align_corners, float* logits,
                   uint8_t* mask, int Ho, int Wo, fs_stream stream);
/* hist3K: int64[3][K] = {intersection, |pred|, |target|}, accumulated (caller zeroes). */
int fs_iou_hist(const uint8_t* pred, const uint8_t* target, int64_t numel, int K, int ignore_index, long long* hist3K,
                fs_stream stream);

/* rgb[numel,3] = palette[K,3][mask[numel]]  (flow/base.py:308-312) */
int fs_colorize(const uint8_t* mask, const uint8_t* palette, int K, uint8_t* rgb, int64_t numel, fs_stream stream);
/* H.264 block motion vectors -> forward / inverse sampling grids (dataset/flow/extract_motion_vectors.py:21-43).
 * mv: int32 [n, stride>=7] rows (source, w, h, src_x, src_y, dst_x, dst_y, ...) on the device; grids: float64 [hb, wb, 2];
 * owners: 2*hb*wb ints of scratch.  A block hit by several vectors takes the LAST one, as the reference's loop does. */
int fs_mv_to_grids(const int* mv, int n, int stride, int hb, int wb, int block, int H, int W, int* owners, double* grid,
                   double* inv_grid, fs_stream stream);

/* Sliding-crop inference (flow/base.py:182-234): canvas[n,K,H,W] (float64, zeroed by the caller) += softmax_K(logits[n,K,h,w])
 * at (y0, x0); count[H,W] += 1 over the crop.  fs_canvas_finish divides by the count (flow/base.py:208) and, if mask != NULL,
 * writes the per-frame argmax. */
int fs_softmax_accumulate(const float* logits, int n, int K, int h, int w, double* canvas, double* count, int H, int W, int y0,
                          int x0, fs_stream stream);
int fs_canvas_finish(double* canvas, const double* count, int n, int K, int64_t HW, uint8_t* mask, fs_stream stream);

/* argmax over K of F.interpolate(canvas, (Ho, Wo), bilinear, align_corners=True) evaluated in float64 (flow/base.py:275-276
 * on compute_output's result), without the [n,K,Ho,Wo] intermediate.  canvas must already be divided by the count. */
int fs_canvas_resize_argmax(const double* canvas, int n, int K, int Hi, int Wi, uint8_t* mask, int Ho, int Wo, fs_stream stream);
/* crop_motion_vector (flow/transform.py:215-261) for every crop x every grid of a window in one launch.  grids: host array of
 * `ngrids` device pointers, each fp32 [Hg, Wg, 2] normalised to the H x W frame (any number: more than 32 grids or crops are cut into several launches); crop c is the ch x cw window at
 * (crop_y[c], crop_x[c]).  out: fp32 [ncrops][ngrids][ch/16][cw/16][2]: block range by Python's round(), coordinates
 * renormalised to the crop, half-pixel bilinear resize (cv2.INTER_LINEAR) to (ch/16) x (cw/16). */
int fs_crop_grids(const float* const* grids, int ngrids, int Hg, int Wg, int H, int W, int ncrops, const int* crop_y,
                  const int* crop_x, int ch, int cw, float* out, fs_stream stream);

/* compute_output after the network (flow/base.py:182-209, 226-234) for ALL crops of a window in one pass: lo_prev / lo_next =
 * per-crop decoder logits [ncrops,K,h,w] (lo_next NULL: single frame), crop_grids = fs_crop_grids' output
 * [ncrops][2(n-1)][Hg][Wg][2] (ignored when no_warp), crop (y, x) offsets in the reference's crop order.  Every pixel of the
 * H x W frame is written ONCE: per covering crop the predict_segmentation tail and the fp32 softmax over K, summed in float64 in
 * crop order, divided by the number of covering crops -> canvas[n,K,H,W] (float64, may be NULL) and / or its argmax mask[n,H,W]
 * (may be NULL).  Bit-identical to fs_seg_tail_accumulate per crop + fs_canvas_finish.  K <= 8, at most 64 crops;
 * scratch: ncrops * 2(n-1) * K * Hg * Wg floats (warp mode). */
int fs_crops_fuse(const float* lo_prev, const float* lo_next, const float* crop_grids, int ncrops, const int* crop_y, const int* crop_x,
                  int K, int h, int w, int Hg, int Wg, int ch, int cw, int n, int no_warp, double* canvas, uint8_t* mask, int H, int W,
                  float* scratch, fs_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOODSEG_H_ */
