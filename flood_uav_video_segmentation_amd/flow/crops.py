"""Full-frame sliding-crop inference on the HIP path -- the reference's default real-video route
(`no_cropping=False`): flow/base.py:182-234 (`compute_output`, `compute_predict_crop`) and
flow/transform.py:215-261 (`crop_motion_vector`).

A 1072x1920 frame is covered by ceil-strided 713x713 crops (stride = ceil(713 * 2/3) = 476 -> 2 x 4 crops);
for every crop the motion grids are cropped to the nearest block edges, renormalised to the crop and
resized to (crop // 16)^2, the FlowModel predicts the n frames, their per-pixel softmax is accumulated in
a float64 canvas and finally divided by the number of crops covering each pixel.
"""
import math

import torch

from .. import _lib, ops
from .._lib import check, ptr, stream_ptr


def _is_grid_list(lst):
    return lst is not None and isinstance(lst, list) and len(lst) > 0 and lst[0].dim() >= 3


def crop_motion_vector(mvs_left, mvs_right, height, width, crop_height, crop_width, height_offset, width_offset):
    """Crop + renormalise + resize the block-motion grids for one crop window (flow/transform.py:215-261): block range by
    Python's round(), coordinates renormalised to the crop in fp32, half-pixel bilinear resize (cv2.INTER_LINEAR) to
    (crop // 16)^2 -- one HIP launch for all grids (fs_crop_grids).  Grids are [1,Hg,Wg,2] tensors; new tensors are returned
    (the reference mutates CPU inputs in place, a quirk of its numpy view; on GPU inputs it works on a copy -- kept)."""
    if not (_is_grid_list(mvs_left) or _is_grid_list(mvs_right)):
        return mvs_left, mvs_right  # no_warp placeholders pass through (flow/transform.py:216-221)
    nl = len(mvs_left) if mvs_left is not None else 0
    both = list(mvs_left or []) + list(mvs_right or [])
    out = ops.crop_grids(both, (height, width), [(height_offset, width_offset)], (crop_height, crop_width))[0]
    left = [out[j][None] for j in range(nl)] if mvs_left is not None else None
    right = [out[j][None] for j in range(nl, len(both))] if mvs_right is not None else None
    return left, right


def crop_windows(new_h, new_w, crop_h, crop_w, stride_rate=2 / 3):
    """(s_h, e_h, s_w, e_w) of every crop, in the reference's order (flow/base.py:183-203)."""
    stride_h = int(math.ceil(crop_h * stride_rate))
    stride_w = int(math.ceil(crop_w * stride_rate))
    grid_h = int(math.ceil(float(new_h - crop_h) / stride_h) + 1)
    grid_w = int(math.ceil(float(new_w - crop_w) / stride_w) + 1)
    out = []
    for ih in range(grid_h):
        for iw in range(grid_w):
            e_h = min(ih * stride_h + crop_h, new_h)
            e_w = min(iw * stride_w + crop_w, new_w)
            out.append((e_h - crop_h, e_h, e_w - crop_w, e_w))
    return out


def segment_crop_windows(flow_model, frame_a, frame_b, crop_h, crop_w, crop_batch=8):
    """Per-crop decoder logits of one or two full frames, all crop windows batched through the network in place:
    ([ncrops,K,fh,fw] for frame_a, the same for frame_b or None)."""
    net = flow_model.model
    _, _, new_h, new_w = frame_a.shape
    yx = [(s_h, s_w) for (s_h, _, s_w, _) in crop_windows(new_h, new_w, crop_h, crop_w)]
    parts_a, parts_b = [], []
    for c0 in range(0, len(yx), crop_batch):
        sub = yx[c0:c0 + crop_batch]
        lows = net.segment_crops(frame_a, frame_b, sub, (crop_h, crop_w))
        parts_a.append(lows[:len(sub)])
        if frame_b is not None:
            parts_b.append(lows[len(sub):])
    cat = lambda ps: ps[0] if len(ps) == 1 else torch.cat(ps, 0)  # noqa: E731
    return cat(parts_a), (cat(parts_b) if frame_b is not None else None)


def _blank_canvas(n, classes, h, w, dev):
    with torch.cuda.device(dev):
        return (torch.zeros((n, classes, h, w), dtype=torch.float64, device=dev), torch.zeros((h, w), dtype=torch.float64, device=dev))


def compute_output(flow_model, n, frame_prev, frame_next, mvs_left, mvs_right, crop_h, crop_w, classes, profiler=None,
                   want_mask=False, function=None, key_cache=None, out_size=None, crop_batch=8, lows=None, want_canvas=True):
    """flow/base.py:182-209: returns the float64 [n,K,H,W] crop-averaged softmax (and, with want_mask, the uint8 argmax of its
    align_corners=True resize to `out_size` -- flow/base.py:275-276; out_size None = the frame size).

    `function(prev_crop, next_crop, mvs_left_crop, mvs_right_crop) -> logits [n,K,h,w]` is the per-crop network call of the
    generic route: default `compute_predict_crop` (:226-234, FlowModel.predict); test_step passes `compute_test_crop`
    (:212-222, FlowModel.forward with left/right indices).

    Batched route (default function, segmentation mode, a HIP network mirror): the crops of BOTH key frames go through the
    network as batches of `crop_batch` windows per frame read in place from the full frames (fs_segment_crops), all crop
    grids come from one launch (fs_crop_grids) and every crop's tail runs fused with softmax + accumulation
    (fs_seg_tail_accumulate): per-crop logits never reach HBM.  key_cache (KeyframeCache.window): the previous key frame's
    per-crop logits are reused from the last window; lows = (lo_prev, lo_next): per-crop logits computed elsewhere
    (FlowPredictor.predict_clip batches the new key frames of two windows).  Same arithmetic per crop as the generic route.
    With K <= 8 and at most 64 crops the whole post-network part is ONE pass over the frame (fs_crops_fuse): every canvas
    pixel is written once from the crops covering it, in crop order -- bit-identical to the per-crop accumulation, without its
    float64 read-modify-writes; want_canvas=False (predict_step only needs the masks) then skips the canvas altogether and
    returns (None, mask)."""
    _, _, new_h, new_w = frame_prev.shape
    dev = frame_prev.device
    windows = crop_windows(new_h, new_w, crop_h, crop_w)
    net = getattr(flow_model, "model", None)
    batched = lows is not None or (function is None and not getattr(flow_model, "feature_based", True) and hasattr(net, "segment_crops")
                                   and frame_prev.shape[0] == 1 and frame_next is not None)
    if batched:
        yx = [(s_h, s_w) for (s_h, _, s_w, _) in windows]
        tag = ("crops", new_h, new_w, crop_h, crop_w, getattr(getattr(net, "_hip_net", None), "generation", 0))
        if lows is not None:
            lo_prev, lo_next = lows
        else:
            lo_prev = key_cache.prev(tag) if key_cache is not None else None
            if lo_prev is None:
                lo_prev, lo_next = segment_crop_windows(flow_model, frame_prev, frame_next, crop_h, crop_w, crop_batch)
            else:
                lo_next, _ = segment_crop_windows(flow_model, frame_next, None, crop_h, crop_w, crop_batch)
            if key_cache is not None:
                key_cache.store_next(tag, lo_next)
        no_warp = flow_model.no_warp or not _is_grid_list(mvs_left)
        grids = None
        if not no_warp:
            grids = ops.crop_grids(list(mvs_left) + list(mvs_right), (new_h, new_w), yx, (crop_h, crop_w))  # [nc, 2(n-1), fh, fw, 2]
        if classes <= 8 and len(yx) <= 64:
            same = out_size is None or (int(out_size[0]), int(out_size[1])) == (new_h, new_w)
            canvas, mask = ops.crops_fuse(lo_prev, lo_next, grids, yx, (crop_h, crop_w), n, no_warp, (new_h, new_w),
                                          want_canvas=want_canvas or (want_mask and not same), want_mask=want_mask and same)
            if want_mask and not same:
                mask = ops.canvas_resize_argmax(canvas, out_size)
            return (canvas if want_canvas else None, mask) if want_mask else canvas
        canvas, count = _blank_canvas(n, classes, new_h, new_w, dev)
        for c, (y0, x0) in enumerate(yx):
            gl = [grids[c, j][None] for j in range(n - 1)] if grids is not None else mvs_left
            gr = [grids[c, n - 1 + j][None] for j in range(n - 1)] if grids is not None else mvs_right
            ops.seg_tail_accumulate(lo_prev[c:c + 1], lo_next[c:c + 1], gl, gr, n, (crop_h, crop_w), no_warp, canvas, count, y0, x0)
    else:
        lib = _lib.load()
        canvas, count = _blank_canvas(n, classes, new_h, new_w, dev)
        for (s_h, e_h, s_w, e_w) in windows:
            prev_c = frame_prev[:, :, s_h:e_h, s_w:e_w].contiguous()
            next_c = frame_next[:, :, s_h:e_h, s_w:e_w].contiguous()
            ml, mr = crop_motion_vector(mvs_left, mvs_right, new_h, new_w, e_h - s_h, e_w - s_w, s_h, s_w)
            if function is None:
                logits = flow_model.predict(prev_c, next_c, ml, mr, n, profiler)["pred"]
            else:
                logits = function(prev_c, next_c, ml, mr)
            if logits.shape[2] != crop_h or logits.shape[3] != crop_w:
                logits = ops.resize_bilinear(logits, (crop_h, crop_w), align_corners=True)
            logits = logits.contiguous()
            with torch.cuda.device(dev):
                check(lib.fs_softmax_accumulate(ptr(logits), n, classes, crop_h, crop_w, ptr(canvas), ptr(count), new_h, new_w, s_h, s_w,
                                                stream_ptr()))
    mask = ops.canvas_finish(canvas, count, out_size, want_mask)
    return (canvas, mask) if want_mask else canvas
