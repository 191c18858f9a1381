"""Oracle for the sliding-crop route: flow/transform.py:215-261 (`crop_motion_vector`) and
flow/base.py:182-234 (`compute_output` + `compute_predict_crop`).

PINNED (round 4) wherever cv2 does not interpolate: tests/golden/transforms.npz holds what the reference's OWN
crop_motion_vector returns for every 704-crop window of a 1072x1920 frame, offsets off the block edges and two
round-half-to-even block ranges (the generator's cv2 stand-in knows only the same-size case of cv2.resize, a copy), and
tests/golden/mv_grids.npz what its own extract_motion_vectors.py writes on a synthetic frame source: both bit-exact here
(tests/test_oracle_golden.py); compute_output inside the reference's own predict_step / test_step on 704 crops likewise
(tests/golden/lightning_steps.npz).  PARITY UNPINNED for the interpolating grid resize (713 crops: 45 -> 44 blocks): cv2 is
absent offline; the restatement uses F.interpolate(bilinear, align_corners=False), the same half-pixel-centre formula cv2
documents for float data.  Everything else (block rounding with Python's banker's round, renormalisation, float64 accumulation, crop
order) is restated literally.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def crop_motion_vector(mvs_left, mvs_right, height, width, crop_height, crop_width, height_offset, width_offset):
    if not (mvs_left is not None and isinstance(mvs_left, list) and len(mvs_left) > 0 and len(mvs_left[0].shape) >= 3):
        return mvs_left, mvs_right  # no_warp placeholders pass through untouched (flow/transform.py:216-221)
    mv_h, mv_w = mvs_left[0].shape[-3], mvs_left[0].shape[-2]
    ppb_h, ppb_w = height / mv_h, width / mv_w
    final_h, final_w = crop_height // 16, crop_width // 16
    bho = round(height_offset / ppb_h)
    bwo = round(width_offset / ppb_w)
    bh = round((height_offset + crop_height) / ppb_h) - bho
    bw = round((width_offset + crop_width) / ppb_w) - bwo

    def one(m):
        m = m.float().numpy()[0].copy()  # copy: the GPU branch of the reference never mutates its input
        m = m[bho:bho + bh, bwo:bwo + bw]
        m[:, :, 0] = ((((m[:, :, 0] + 1) / 2) * width - width_offset) / (bw * ppb_w)) * 2 - 1
        m[:, :, 1] = ((((m[:, :, 1] + 1) / 2) * height - height_offset) / (bh * ppb_h)) * 2 - 1
        t = torch.from_numpy(np.ascontiguousarray(m)).permute(2, 0, 1)[None]
        if (bh, bw) != (final_h, final_w):
            t = F.interpolate(t, size=(final_h, final_w), mode="bilinear", align_corners=False)
        return t.permute(0, 2, 3, 1).contiguous()

    return [one(m) for m in mvs_left], [one(m) for m in mvs_right]


def compute_output(predict, n, frame_prev, frame_next, mvs_left, mvs_right, crop_h, crop_w, classes):
    """predict(prev_crop, next_crop, mvs_left_crop, mvs_right_crop) -> [n,K,h,w] logits."""
    stride_rate = 2 / 3
    _, _, new_h, new_w = frame_prev.shape
    stride_h = int(np.ceil(crop_h * stride_rate))
    stride_w = int(np.ceil(crop_w * stride_rate))
    grid_h = int(np.ceil(float(new_h - crop_h) / stride_h) + 1)
    grid_w = int(np.ceil(float(new_w - crop_w) / stride_w) + 1)
    pred = torch.zeros((n, classes, new_h, new_w), dtype=torch.float64)
    count = torch.zeros((new_h, new_w), dtype=torch.float64)
    for ih in range(grid_h):
        for iw in range(grid_w):
            s_h = ih * stride_h
            e_h = min(s_h + crop_h, new_h)
            s_h = e_h - crop_h
            s_w = iw * stride_w
            e_w = min(s_w + crop_w, new_w)
            s_w = e_w - crop_w
            ml, mr = crop_motion_vector(mvs_left, mvs_right, new_h, new_w, e_h - s_h, e_w - s_w, s_h, s_w)
            out = predict(frame_prev[:, :, s_h:e_h, s_w:e_w].clone(), frame_next[:, :, s_h:e_h, s_w:e_w].clone(), ml, mr)
            if out.shape[2:] != (crop_h, crop_w):
                out = F.interpolate(out, (crop_h, crop_w), mode="bilinear", align_corners=True)
            count[s_h:e_h, s_w:e_w] += 1
            pred[:, :, s_h:e_h, s_w:e_w] += F.softmax(out, dim=1)
    return pred / count[None, None]


def motion_vectors_to_grids(motion_vectors, H, W, default_grid):
    """dataset/flow/extract_motion_vectors.py:21-43, literally (a block hit twice keeps the LAST vector)."""
    bs, hb, wb = 16, 1072 // 16, 1920 // 16
    grid, inv_grid = np.copy(default_grid), np.copy(default_grid)
    for m in motion_vectors:
        assert m[0] == -1
        sx, sy, dx, dy = m[3] // bs, m[4] // bs, m[5] // bs, m[6] // bs
        if 0 <= dx < wb and 0 <= dy < hb:
            grid[dy][dx][0] = (sx * bs + bs // 2) / W * 2 - 1
            grid[dy][dx][1] = (sy * bs + bs // 2) / H * 2 - 1
        if 0 <= sx < wb and 0 <= sy < hb:
            inv_grid[sy][sx][0] = (dx * bs + bs // 2) / W * 2 - 1
            inv_grid[sy][sx][1] = (dy * bs + bs // 2) / H * 2 - 1
    return grid, inv_grid
