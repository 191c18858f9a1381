"""Tensor-level wrappers over the C ABI (torch is only the allocator / stream provider here).

Every function enqueues on torch's current HIP stream and returns freshly allocated tensors;
inputs are never modified.  Reference ops restated: see include/floodseg.h.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, one_device, ptr, stream_ptr


def _f32c(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"floodseg: {name} must live on the GPU (no CPU fallback exists)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def is_channels_last_dense(t):
    """True when a logical NCHW tensor is stored pixel-major with pixel stride == C."""
    if t.dim() != 4:
        return False
    b, c, h, w = t.shape
    want = (h * w * c, 1, w * c, c)
    # (the stride of a size-1 dimension is arbitrary: torch keeps whatever the tensor was made with)
    return all(n == 1 or s == ws for n, s, ws in zip(t.shape, t.stride(), want)) or (c == 1 and t.is_contiguous())


def as_nhwc(t):
    """Logical NCHW tensor -> dense channels_last storage (copy only if needed)."""
    t = t if t.dtype == torch.float32 else t.float()
    if is_channels_last_dense(t):
        return t
    return t.contiguous(memory_format=torch.channels_last)


def empty_nhwc(b, c, h, w, device):
    return torch.empty((b, c, h, w), dtype=torch.float32, device=device, memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------ flow ops
def grid_sample(inp, grid, align_corners=False):
    """F.grid_sample(inp, grid, mode='bilinear', padding_mode='border') (flow/model.py:157,248)."""
    lib = _lib.load()
    b, c, hi, wi = inp.shape
    gb, hg, wg, two = grid.shape
    if two != 2 or gb != b:
        raise RuntimeError(f"floodseg.grid_sample: grid shape {tuple(grid.shape)} does not match input batch {b}")
    dev = one_device(inp, grid, what="floodseg.grid_sample")
    with torch.cuda.device(dev):
        grid = _f32c(grid, "grid")
        if b == 0:  # empty batch: nothing to launch (torch returns an empty tensor too)
            return torch.empty((0, c, hg, wg), dtype=torch.float32, device=dev)
        if inp.dim() == 4 and c % 4 == 0 and c >= 64 and is_channels_last_dense(inp):
            src = inp if inp.dtype == torch.float32 else inp.float()
            out = empty_nhwc(b, c, hg, wg, dev)
            check(lib.fs_grid_sample_nhwc(ptr(src), c, b, c, hi, wi, ptr(grid), hg, wg, ptr(out), c, int(align_corners), stream_ptr()))
            return out
        src = _f32c(inp, "input")
        out = torch.empty((b, c, hg, wg), dtype=torch.float32, device=dev)
        check(lib.fs_grid_sample_nchw(ptr(src), b, c, hi, wi, ptr(grid), hg, wg, ptr(out), int(align_corners), stream_ptr()))
        return out


def _check_out(out, shape, nhwc, what):
    """`out=`: a caller-owned destination (e.g. one image slot of a batch tensor) -- must already have the layout the op writes."""
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32:
        raise RuntimeError(f"{what}: out must be float32 {tuple(shape)}, got {out.dtype} {tuple(out.shape)}")
    if not (is_channels_last_dense(out) if nhwc else out.is_contiguous()):
        raise RuntimeError(f"{what}: out must be dense {'channels_last' if nhwc else 'contiguous'}")
    return out


def resize_bilinear(inp, size, align_corners=True, out=None):
    """F.interpolate(inp, size, mode='bilinear', align_corners=...) (flow/model.py:42..228)."""
    lib = _lib.load()
    b, c, hi, wi = inp.shape
    ho, wo = int(size[0]), int(size[1])
    dev = one_device(inp, out, what="floodseg.resize_bilinear")
    with torch.cuda.device(dev):
        if b == 0:
            return torch.empty((0, c, ho, wo), dtype=torch.float32, device=dev)
        if c % 4 == 0 and c >= 64 and is_channels_last_dense(inp):
            src = inp if inp.dtype == torch.float32 else inp.float()
            out = empty_nhwc(b, c, ho, wo, dev) if out is None else _check_out(out, (b, c, ho, wo), True, "floodseg.resize_bilinear")
            check(lib.fs_resize_bilinear_nhwc(ptr(src), c, b, c, hi, wi, ptr(out), c, ho, wo, int(align_corners), stream_ptr()))
            return out
        src = _f32c(inp, "input")
        out = torch.empty((b, c, ho, wo), dtype=torch.float32, device=dev) if out is None else _check_out(out, (b, c, ho, wo), False, "floodseg.resize_bilinear")
        check(lib.fs_resize_bilinear_nchw(ptr(src), b * c, hi, wi, ptr(out), ho, wo, int(align_corners), stream_ptr()))
        return out


def blend(a, wa, b=None, wb=0.0, out=None):
    """wa*a + wb*b with the reference's rounding order (flow/model.py:104,168,234-236).  out: optional destination of the
    layout the result has (channels_last when `a` is, else contiguous), e.g. one image slot of a preallocated batch."""
    lib = _lib.load()
    if b is not None and a.shape != b.shape:
        raise RuntimeError(f"floodseg.blend: shapes differ ({tuple(a.shape)} vs {tuple(b.shape)})")
    dev = one_device(a, b, out, what="floodseg.blend")
    with torch.cuda.device(dev):
        # the kernel walks both operands as flat arrays: bring BOTH to one dense layout (channels_last kept when `a` has it,
        # so the feature-mode maps are not transposed; any other view -- equal strides or not -- becomes plain contiguous)
        if is_channels_last_dense(a):
            if b is not None:
                b = as_nhwc(b)
        else:
            a = a.contiguous()
            if b is not None:
                b = b.contiguous()
        a = a if a.dtype == torch.float32 else a.float()
        if b is not None and b.dtype != torch.float32:
            b = b.float()
        out = torch.empty_like(a) if out is None else _check_out(out, a.shape, a.dim() == 4 and is_channels_last_dense(a) and not a.is_contiguous(),
                                                                 "floodseg.blend")
        if a.numel() == 0:
            return out
        check(lib.fs_blend(ptr(a), float(wa), ptr(b), float(wb), ptr(out), a.numel(), stream_ptr()))
        return out


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def seg_tail(lo_prev, lo_next, grids_left, grids_right, n, out_hw, no_warp, want_logits=True, want_mask=False):
    """Fused predict_segmentation tail (flow/model.py:184-241 after the two decoder calls).

    lo_prev/lo_next: [1,K,h,w] decoder logits; grids: lists of n-1 [1,Hg,Wg,2] tensors.
    Returns (logits [n,K,H,W] or None, mask uint8 [n,H,W] or None).
    """
    lib = _lib.load()
    grids = [] if (lo_next is None or no_warp) else list(grids_left) + list(grids_right)
    dev = one_device(lo_prev, lo_next, *grids, what="floodseg.seg_tail")
    with torch.cuda.device(dev):
        lo_prev = _f32c(lo_prev, "lo_prev")
        _, k, h, w = lo_prev.shape
        hh, ww = out_hw
        frames = n if lo_next is not None else 1
        logits = torch.empty((frames, k, hh, ww), dtype=torch.float32, device=dev) if want_logits else None
        mask = torch.empty((frames, hh, ww), dtype=torch.uint8, device=dev) if want_mask else None
        gl = gr = None
        hg = wg = 1
        scratch = None
        keep = []
        if lo_next is not None:
            lo_next = _f32c(lo_next, "lo_next")
            if not no_warp:
                if len(grids_left) != n - 1 or len(grids_right) != n - 1:
                    raise RuntimeError("floodseg.seg_tail: need n-1 grids per direction")
                keep = [_f32c(g, "grid") for g in grids]
                hg, wg = keep[0].shape[1], keep[0].shape[2]
                for g in keep:
                    if tuple(g.shape) != (1, hg, wg, 2):
                        raise RuntimeError("floodseg.seg_tail: all grids must be [1,Hg,Wg,2] of one size")
                gl = _ptr_array(keep[: n - 1])
                gr = _ptr_array(keep[n - 1:])
                scratch = torch.empty(2 * (n - 1) * k * hg * wg, dtype=torch.float32, device=dev)
        check(lib.fs_seg_tail(ptr(lo_prev), ptr(lo_next), gl, gr, k, h, w, hg, wg, hh, ww, int(n), int(bool(no_warp)),
                              ptr(logits), ptr(mask), ptr(scratch), stream_ptr()))
    return logits, mask


def feat_tail(f_prev, f_next, grids_left, grids_right, n, no_warp, default_grid=None):
    """Fused predict_feature tail (flow/model.py:131-171 between the encoder and the batched decoder call): f_prev / f_next
    [1,C,fh,fw] stored channels_last; grids: lists of n-1 [1,Hg,Wg,2]; default_grid [1,H0,W0,2] (warp mode).  Returns the decoder's
    batch [n,C,fh,fw] ([1,C,fh,fw] when f_next is None), channels_last -- bit-identical to the op-by-op route."""
    lib = _lib.load()
    warp = not no_warp
    grids = list(grids_left) + list(grids_right) if (warp and f_next is not None) else []
    dev = one_device(f_prev, f_next, default_grid if warp else None, *grids, what="floodseg.feat_tail")
    for t in (f_prev, f_next):
        if t is not None and not (t.dim() == 4 and t.shape[0] == 1 and t.dtype == torch.float32 and is_channels_last_dense(t)):
            raise RuntimeError("floodseg.feat_tail: feature maps must be float32 [1,C,fh,fw] stored channels_last")
    _, c, fh, fw = f_prev.shape
    if c % 4 != 0:
        raise RuntimeError("floodseg.feat_tail: C must be a multiple of 4")
    if f_next is not None and f_next.shape != f_prev.shape:
        raise RuntimeError("floodseg.feat_tail: f_prev / f_next shapes differ")
    with torch.cuda.device(dev):
        nmaps = n if f_next is not None else 1
        stack = empty_nhwc(nmaps, c, fh, fw, dev)
        gl = gr = None
        hg = wg = h0 = w0 = 1
        scratch = g0 = None
        keep = []
        if warp:
            if default_grid is None:
                raise RuntimeError("floodseg.feat_tail: warp mode needs the default grid")
            g0 = _f32c(default_grid, "default_grid")
            if g0.dim() != 4 or g0.shape[0] != 1 or g0.shape[3] != 2:
                raise RuntimeError("floodseg.feat_tail: default grid must be [1,H0,W0,2]")
            h0, w0 = g0.shape[1], g0.shape[2]
            if f_next is not None and n > 1:
                if len(grids_left) != n - 1 or len(grids_right) != n - 1:
                    raise RuntimeError("floodseg.feat_tail: need n-1 grids per direction")
                keep = [_f32c(g, "grid") for g in grids]
                hg, wg = keep[0].shape[1], keep[0].shape[2]
                for g in keep:
                    if tuple(g.shape) != (1, hg, wg, 2):
                        raise RuntimeError("floodseg.feat_tail: all grids must be [1,Hg,Wg,2] of one size")
                gl = _ptr_array(keep[: n - 1])
                gr = _ptr_array(keep[n - 1:])
                scratch = torch.empty(2 * (n - 1) * hg * wg * c, dtype=torch.float32, device=dev)
        check(lib.fs_feat_tail(ptr(f_prev), ptr(f_next), c, fh, fw, gl, gr, hg, wg, ptr(g0), h0, w0, int(n), int(bool(no_warp)), ptr(stack),
                               ptr(scratch), stream_ptr()))
    return stack


def seg_tail_accumulate(lo_prev, lo_next, grids_left, grids_right, n, crop_hw, no_warp, canvas, count, y0, x0):
    """The same tail feeding the sliding-crop canvas (flow/base.py:204-205, 226-234): softmax over K of every output frame
    of this crop is ADDED to canvas [n,K,H,W] (float64) at (y0, x0), count[H,W] += 1 over the crop -- in place."""
    lib = _lib.load()
    grids = [] if (lo_next is None or no_warp) else list(grids_left) + list(grids_right)
    dev = one_device(lo_prev, lo_next, canvas, count, *grids, what="floodseg.seg_tail_accumulate")
    if canvas.dtype != torch.float64 or count.dtype != torch.float64 or not canvas.is_contiguous() or not count.is_contiguous():
        raise RuntimeError("floodseg.seg_tail_accumulate: canvas / count must be contiguous float64 tensors")
    with torch.cuda.device(dev):
        lo_prev = _f32c(lo_prev, "lo_prev")
        _, k, h, w = lo_prev.shape
        frames = n if lo_next is not None else 1
        if tuple(canvas.shape[:2]) != (frames, k) or tuple(canvas.shape[2:]) != tuple(count.shape):
            raise RuntimeError(f"floodseg.seg_tail_accumulate: canvas {tuple(canvas.shape)} / count {tuple(count.shape)} do not match [{frames},{k},H,W]")
        gl = gr = None
        hg = wg = 1
        scratch = None
        keep = []
        if lo_next is not None:
            lo_next = _f32c(lo_next, "lo_next")
            if not no_warp:
                if len(grids_left) != n - 1 or len(grids_right) != n - 1:
                    raise RuntimeError("floodseg.seg_tail_accumulate: need n-1 grids per direction")
                keep = [_f32c(g, "grid") for g in grids]
                hg, wg = keep[0].shape[-3], keep[0].shape[-2]
                for g in keep:
                    if tuple(g.shape[-3:]) != (hg, wg, 2) or g.numel() != hg * wg * 2:
                        raise RuntimeError("floodseg.seg_tail_accumulate: all grids must be [1,Hg,Wg,2] of one size")
                gl = _ptr_array(keep[: n - 1])
                gr = _ptr_array(keep[n - 1:])
                scratch = torch.empty(2 * (n - 1) * k * hg * wg, dtype=torch.float32, device=dev)
        check(lib.fs_seg_tail_accumulate(ptr(lo_prev), ptr(lo_next), gl, gr, k, h, w, hg, wg, int(crop_hw[0]), int(crop_hw[1]), int(n),
                                         int(bool(no_warp)), ptr(canvas), ptr(count), canvas.shape[2], canvas.shape[3], int(y0), int(x0),
                                         ptr(scratch), stream_ptr()))


def crop_grids(grids, frame_hw, crop_yx, crop_hw):
    """crop_motion_vector (flow/transform.py:215-261) for all crops x all grids of a window in ONE launch.
    grids: list of [1,Hg,Wg,2] tensors normalised to the frame (H, W); crop_yx: [(y0, x0), ...]; returns fp32
    [ncrops, len(grids), ch//16, cw//16, 2] -- out[c, j][None] is the grid the reference hands to the network for crop c."""
    lib = _lib.load()
    dev = one_device(*grids, what="floodseg.crop_grids")
    with torch.cuda.device(dev):
        keep = [_f32c(g, "grid") for g in grids]
        hg, wg = keep[0].shape[-3], keep[0].shape[-2]
        for g in keep:
            if tuple(g.shape[-3:]) != (hg, wg, 2) or g.numel() != hg * wg * 2:
                raise RuntimeError("floodseg.crop_grids: all grids must be [1,Hg,Wg,2] of one size")
        nc = len(crop_yx)
        ys = (ctypes.c_int * nc)(*[int(y) for y, _ in crop_yx])
        xs = (ctypes.c_int * nc)(*[int(x) for _, x in crop_yx])
        out = torch.empty((nc, len(keep), int(crop_hw[0]) // 16, int(crop_hw[1]) // 16, 2), dtype=torch.float32, device=dev)
        check(lib.fs_crop_grids(_ptr_array(keep), len(keep), hg, wg, int(frame_hw[0]), int(frame_hw[1]), nc, ys, xs, int(crop_hw[0]),
                                int(crop_hw[1]), ptr(out), stream_ptr()))
    return out


def crops_fuse(lo_prev, lo_next, grids, crop_yx, crop_hw, n, no_warp, frame_hw, want_canvas=True, want_mask=False):
    """compute_output after the network for ALL crops of a window in one pass (fs_crops_fuse): lo_prev / lo_next = per-crop
    decoder logits [nc,K,h,w]; grids = crop_grids' output [nc, 2(n-1), fh, fw, 2] or None (no_warp).  Returns (float64 canvas
    [n,K,H,W] already divided by the crop count, or None; uint8 argmax [n,H,W] or None) -- each pixel written once."""
    lib = _lib.load()
    dev = one_device(lo_prev, lo_next, grids, what="floodseg.crops_fuse")
    with torch.cuda.device(dev):
        lo_prev = _f32c(lo_prev, "lo_prev")
        nc, k, h, w = lo_prev.shape
        if nc != len(crop_yx):
            raise RuntimeError(f"floodseg.crops_fuse: {nc} crops of logits but {len(crop_yx)} crop windows")
        frames = n if lo_next is not None else 1
        hh, ww = int(frame_hw[0]), int(frame_hw[1])
        warp = lo_next is not None and not no_warp and n > 1
        hg = wg = 1
        scratch = None
        if lo_next is not None:
            lo_next = _f32c(lo_next, "lo_next")
            if lo_next.shape != lo_prev.shape:
                raise RuntimeError("floodseg.crops_fuse: lo_prev / lo_next shapes differ")
        if warp:
            grids = _f32c(grids, "grids")
            if grids.dim() != 5 or grids.shape[0] != nc or grids.shape[1] != 2 * (n - 1) or grids.shape[4] != 2:
                raise RuntimeError(f"floodseg.crops_fuse: grids must be [nc, 2(n-1), Hg, Wg, 2], got {tuple(grids.shape)}")
            hg, wg = grids.shape[2], grids.shape[3]
            scratch = torch.empty(nc * 2 * (n - 1) * k * hg * wg, dtype=torch.float32, device=dev)
        canvas = torch.empty((frames, k, hh, ww), dtype=torch.float64, device=dev) if want_canvas else None
        mask = torch.empty((frames, hh, ww), dtype=torch.uint8, device=dev) if want_mask else None
        ys = (ctypes.c_int * nc)(*[int(y) for y, _ in crop_yx])
        xs = (ctypes.c_int * nc)(*[int(x) for _, x in crop_yx])
        check(lib.fs_crops_fuse(ptr(lo_prev), ptr(lo_next), ptr(grids) if warp else None, nc, ys, xs, k, h, w, hg, wg, int(crop_hw[0]),
                                int(crop_hw[1]), int(n), int(not warp), ptr(canvas), ptr(mask), hh, ww, ptr(scratch), stream_ptr()))
    return canvas, mask


def canvas_finish(canvas, count, out_size=None, want_mask=False):
    """canvas /= count in place (flow/base.py:208); optionally the uint8 argmax of its align_corners=True bilinear resize to
    `out_size` evaluated in float64 (flow/base.py:275-276) -- the identity resize when out_size is the canvas size."""
    lib = _lib.load()
    dev = one_device(canvas, count, what="floodseg.canvas_finish")
    n, k, h, w = canvas.shape
    with torch.cuda.device(dev):
        same = out_size is None or (int(out_size[0]), int(out_size[1])) == (h, w)
        mask = torch.empty((n, h, w), dtype=torch.uint8, device=dev) if (want_mask and same) else None
        check(lib.fs_canvas_finish(ptr(canvas), ptr(count), n, k, h * w, ptr(mask), stream_ptr()))
        if want_mask and not same:
            mask = torch.empty((n, int(out_size[0]), int(out_size[1])), dtype=torch.uint8, device=dev)
            check(lib.fs_canvas_resize_argmax(ptr(canvas), n, k, h, w, ptr(mask), int(out_size[0]), int(out_size[1]), stream_ptr()))
    return mask


def canvas_resize_argmax(canvas, out_size):
    """uint8 argmax of the align_corners=True bilinear resize (in float64) of a crop-averaged canvas (flow/base.py:275-276)."""
    lib = _lib.load()
    dev = one_device(canvas, what="floodseg.canvas_resize_argmax")
    n, k, h, w = canvas.shape
    with torch.cuda.device(dev):
        mask = torch.empty((n, int(out_size[0]), int(out_size[1])), dtype=torch.uint8, device=dev)
        check(lib.fs_canvas_resize_argmax(ptr(canvas), n, k, h, w, ptr(mask), int(out_size[0]), int(out_size[1]), stream_ptr()))
    return mask


def argmax_u8(logits):
    """logits.max(1)[1] as uint8 (flow/base.py:276-277)."""
    lib = _lib.load()
    with torch.cuda.device(one_device(logits, what="floodseg.argmax_u8")):
        x = _f32c(logits)
        b, k, h, w = x.shape
        out = torch.empty((b, h, w), dtype=torch.uint8, device=x.device)
        if b == 0:
            return out
        check(lib.fs_argmax_u8(ptr(x), b, k, h * w, ptr(out), stream_ptr()))
    return out


def resize_argmax_u8(logits, size):
    """F.interpolate(logits, size, bilinear, align_corners=True).max(1)[1] without the big intermediate."""
    lib = _lib.load()
    with torch.cuda.device(one_device(logits, what="floodseg.resize_argmax_u8")):
        x = _f32c(logits)
        b, k, h, w = x.shape
        out = torch.empty((b, int(size[0]), int(size[1])), dtype=torch.uint8, device=x.device)
        if b == 0:
            return out
        check(lib.fs_resize_argmax_u8(ptr(x), b, k, h, w, ptr(out), int(size[0]), int(size[1]), stream_ptr()))
    return out


def resize_crop(logits, full_size, size, align_corners=False, want_logits=True, want_mask=False):
    """F.interpolate(logits, full_size, bilinear, align_corners)[:, :, :size[0], :size[1]] as a DENSE tensor, and / or its
    .max(1)[1] as uint8 -- the Segmenter's upsample + unpadding (segm/model/segmenter.py:45-46, segm/model/utils.py:79-89) in
    one launch.  Returns (logits or None, mask or None)."""
    lib = _lib.load()
    if not (want_logits or want_mask):
        raise RuntimeError("floodseg.resize_crop: no output requested")
    with torch.cuda.device(one_device(logits, what="floodseg.resize_crop")):
        x = _f32c(logits)
        b, k, hi, wi = x.shape
        hf, wf, ho, wo = int(full_size[0]), int(full_size[1]), int(size[0]), int(size[1])
        if ho > hf or wo > wf or min(ho, wo) < 1:
            raise RuntimeError(f"floodseg.resize_crop: kept region {ho}x{wo} is not inside the resized frame {hf}x{wf}")
        out = torch.empty((b, k, ho, wo), dtype=torch.float32, device=x.device) if want_logits else None
        mask = torch.empty((b, ho, wo), dtype=torch.uint8, device=x.device) if want_mask else None
        if b:
            check(lib.fs_resize_crop(ptr(x), b, k, hi, wi, hf, wf, int(align_corners), ptr(out) if want_logits else None,
                                     ptr(mask) if want_mask else None, ho, wo, stream_ptr()))
    return out, mask


def iou_hist(pred_u8, target_u8, classes, ignore_index=255, hist=None):
    """Accumulate int64[3,K] = (intersection, |pred|, |target|) (util/util.py:52-63)."""
    lib = _lib.load()
    p = pred_u8.contiguous()
    t = target_u8.contiguous()
    if p.dtype != torch.uint8 or t.dtype != torch.uint8 or p.shape != t.shape:
        raise RuntimeError("floodseg.iou_hist: uint8 tensors of equal shape required")
    dev = one_device(p, t, hist, what="floodseg.iou_hist")
    with torch.cuda.device(dev):
        if hist is None:
            hist = torch.zeros((3, classes), dtype=torch.int64, device=dev)
        if p.numel() == 0:
            return hist
        check(lib.fs_iou_hist(ptr(p), ptr(t), p.numel(), classes, ignore_index, ptr(hist), stream_ptr()))
    return hist


# ------------------------------------------------------------------------------------------ building blocks
# (test / bring-up helpers over the op-level hooks of include/floodseg_test.h; nothing on the product path calls them)
def conv2d_nhwc(x, weight, scale=None, shift=None, residual=None, stride=1, pad=0, dil=1, relu=False, tile=0, out=None, split=False):
    """Conv2d on the matrix cores; x logical NCHW (stored NHWC), weight OIHW. Test/bring-up helper.  split=True: the split-operand
    kernel (three bf16 terms per fp32 value, bf16 MFMA, fp32 accumulate: fs_conv2d_nhwc_split) instead of the fp32-MFMA one."""
    lib = _lib.load()
    with torch.cuda.device(one_device(x, weight, scale, shift, residual, out, what="floodseg.conv2d_nhwc")):
        x = as_nhwc(x)
        b, cin, h, w = x.shape
        o, i, kh, kw = weight.shape
        wp = torch.empty((o, kh, kw, i), dtype=torch.float32, device=x.device)
        check(lib.fs_pack_conv_weight(ptr(_f32c(weight)), ptr(wp), o, i, kh, kw, stream_ptr()))
        ho = (h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        if out is None:
            out = empty_nhwc(b, o, ho, wo, x.device)
        res = as_nhwc(residual) if residual is not None else None
        if split:
            planes = torch.empty(3 * wp.numel(), dtype=torch.bfloat16, device=x.device)
            check(lib.fs_split_bf16x3(ptr(wp), wp.numel(), ptr(planes), stream_ptr()))
            check(lib.fs_conv2d_nhwc_split(ptr(x), cin, ptr(planes), ptr(scale), ptr(shift), ptr(res), o, ptr(out), o, b, h, w, cin, o, kh, kw,
                                           stride, pad, dil, int(relu), tile, stream_ptr()))
        else:
            check(lib.fs_conv2d_nhwc(ptr(x), cin, ptr(wp), ptr(scale), ptr(shift), ptr(res), o, ptr(out), o, b, h, w, cin, o, kh, kw,
                                     stride, pad, dil, int(relu), tile, stream_ptr()))
    return out


def attention(qkv, heads, split_operands=True):
    """softmax(q k^T / 8) v per head (segm/model/blocks.py:39-66) for qkv [B, N, 3 * heads * 64] -> [B, N, heads * 64].
    split_operands: True the bf16-matrix-core route with three bf16 terms per fp32 value (the networks' default), False the fp32-MFMA one."""
    lib = _lib.load()
    with torch.cuda.device(one_device(qkv, what="floodseg.attention")):
        qkv = _f32c(qkv)
        b, n, c = qkv.shape
        if c != 3 * heads * 64:
            raise ValueError(f"floodseg.attention: last dim {c} != 3 * {heads} * 64")
        out = torch.empty((b, n, heads * 64), dtype=torch.float32, device=qkv.device)
        ws = torch.empty(max(1, lib.fs_attention_workspace_floats(b, n, heads, int(split_operands))), dtype=torch.float32, device=qkv.device)
        check(lib.fs_attention(ptr(qkv), ptr(out), b, n, heads, 0.125, int(split_operands), ptr(ws), stream_ptr()))
    return out
