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


def _py_round(x):
    return int(round(x))  # Python's round = banker's rounding, as the reference uses (flow/transform.py:230-233)


def crop_motion_vector(mvs_left, mvs_right, height, width, crop_height, crop_width, height_offset, width_offset):
    """Crop + renormalise + resize the block-motion grids for one crop window (flow/transform.py:215-261).
    Grids are [1,Hg,Wg,2] tensors; new tensors are returned (the reference mutates CPU inputs in place, a quirk of
    its numpy view; on GPU inputs it works on a copy -- the GPU behaviour is the one kept)."""
    first = None
    for lst in (mvs_left, mvs_right):
        if lst is not None and isinstance(lst, list) and len(lst) > 0 and lst[0].dim() >= 3:
            first = lst[0]
            break
    if first is None:
        return mvs_left, mvs_right
    mv_h, mv_w = first.shape[-3], first.shape[-2]
    ppb_h, ppb_w = height / mv_h, width / mv_w
    final_h, final_w = crop_height // 16, crop_width // 16
    bho = _py_round(height_offset / ppb_h)
    bwo = _py_round(width_offset / ppb_w)
    bh = _py_round((height_offset + crop_height) / ppb_h) - bho
    bw = _py_round((width_offset + crop_width) / ppb_w) - bwo

    def one(m):
        m = m.float()
        c = m[:, bho:bho + bh, bwo:bwo + bw, :]
        x = ((((c[..., 0] + 1) / 2) * width - width_offset) / (bw * ppb_w)) * 2 - 1
        y = ((((c[..., 1] + 1) / 2) * height - height_offset) / (bh * ppb_h)) * 2 - 1
        g = torch.stack((x, y), 1).contiguous()  # [1,2,bh,bw]
        if (bh, bw) != (final_h, final_w):
            # cv2.resize(INTER_LINEAR) on float data = half-pixel-centre bilinear = align_corners=False
            g = ops.resize_bilinear(g, (final_h, final_w), align_corners=False)
        return g.permute(0, 2, 3, 1).contiguous()

    left = [one(m) for m in mvs_left] if mvs_left is not None else None
    right = [one(m) for m in mvs_right] if mvs_right is not None else None
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


def compute_output(flow_model, n, frame_prev, frame_next, mvs_left, mvs_right, crop_h, crop_w, classes, profiler=None,
                   want_mask=False, function=None):
    """flow/base.py:182-209: returns the float64 [n,K,H,W] crop-averaged softmax (and, optionally, its per-frame
    argmax as uint8 [n,H,W]).  `function(prev_crop, next_crop, mvs_left_crop, mvs_right_crop) -> logits [n,K,h,w]`
    is the per-crop network call: default `compute_predict_crop` (:226-234, FlowModel.predict); test_step passes
    `compute_test_crop` (:212-222, FlowModel.forward with left/right indices)."""
    lib = _lib.load()
    _, _, new_h, new_w = frame_prev.shape
    dev = frame_prev.device
    canvas = torch.zeros((n, classes, new_h, new_w), dtype=torch.float64, device=dev)
    count = torch.zeros((new_h, new_w), dtype=torch.float64, device=dev)
    for (s_h, e_h, s_w, e_w) in crop_windows(new_h, new_w, crop_h, crop_w):
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
        check(lib.fs_softmax_accumulate(ptr(logits), n, classes, crop_h, crop_w, ptr(canvas), ptr(count), new_h, new_w, s_h, s_w,
                                        stream_ptr()))
    mask = torch.empty((n, new_h, new_w), dtype=torch.uint8, device=dev) if want_mask else None
    check(lib.fs_canvas_finish(ptr(canvas), ptr(count), n, classes, new_h * new_w, ptr(mask), stream_ptr()))
    return (canvas, mask) if want_mask else canvas
