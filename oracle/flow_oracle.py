"""Oracle for the key-frame interpolation wrapper (reference flow/model.py, whole file).

Functional restatement: `enc` / `dec` are callables standing in for self.model.encoder /
self.model.decoder.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def get_default_grid():
    """Identity grid of 16x16 block centres for a 1920x1072 frame (flow/model.py:10-21)."""
    width, height, bs = 1920, 1072, 16
    bh, bw = height // bs, width // bs
    g = np.zeros((bh, bw, 2))
    xv, yv = np.meshgrid(np.linspace(0, bw - 1, bw), np.linspace(0, bh - 1, bh))
    g[:, :, 0] = (xv * bs + bs // 2) / width * 2 - 1
    g[:, :, 1] = (yv * bs + bs // 2) / height * 2 - 1
    return g


def _up(t, h, w):
    # flow/model.py:41-42 and every other "if shape != (h, w): interpolate(..., align_corners=True)"
    if t.shape[2] != h or t.shape[3] != w:
        t = F.interpolate(t, size=(h, w), mode="bilinear", align_corners=True)
    return t


def warp(frame, mv, no_warp):
    """flow/model.py:244-249."""
    if no_warp:
        return frame
    return F.grid_sample(frame, mv.float(), mode="bilinear", padding_mode="border", align_corners=False)


def predict_segmentation(enc, dec, frame_prev, frame_next, mvs_left, mvs_right, n, no_warp):
    """flow/model.py:184-241."""
    h, w = frame_prev.shape[2], frame_prev.shape[3]
    o = _up(dec(enc(frame_prev)), h, w)
    maps = [o]
    if frame_next is not None:
        o_next = _up(dec(enc(frame_next)), h, w)
        fwd, bwd = [], []
        cur = o
        for m in mvs_left:
            cur = warp(cur, m, no_warp)  # NB: later steps sample the grid-resolution map
            fwd.append(_up(cur, h, w))
        cur = o_next
        for m in mvs_right:
            cur = warp(cur, m, no_warp)
            bwd.append(_up(cur, h, w))
        for p in range(1, n):
            maps.append((n - p) / n * fwd[p - 1] + p / n * bwd[n - p - 1])
    return {"pred": torch.cat(maps, 0)}


def predict_feature(enc, dec, frame_prev, frame_next, mvs_left, mvs_right, n, no_warp):
    """flow/model.py:116-181."""
    h, w = frame_prev.shape[2], frame_prev.shape[3]
    f = enc(frame_prev)
    fh, fw = f.shape[2], f.shape[3]
    fmaps, fwd, bwd = [], [], []
    f_next = None
    if frame_next is not None:
        f_next = enc(frame_next)
        if not no_warp:
            cur = f
            for m in mvs_left:
                cur = warp(cur, m, no_warp)
                fwd.append(_up(cur, fh, fw))
            cur = f_next
            for m in mvs_right:
                cur = warp(cur, m, no_warp)
                bwd.append(_up(cur, fh, fw))
    if not no_warp:
        # key-frame feature passes through the 67x120 identity grid, align_corners=True (:154-159)
        dg = torch.from_numpy(get_default_grid()).float().unsqueeze(0)
        f = F.grid_sample(f, dg, padding_mode="border", align_corners=True)
        f = _up(f, fh, fw)
    fmaps.append(f)
    if frame_next is not None:
        for p in range(1, n):
            if not no_warp:
                fmaps.append((n - p) / n * fwd[p - 1] + p / n * bwd[n - p - 1])
            else:
                fmaps.append((n - p) / n * f + p / n * f_next)
    out = dec(torch.cat(fmaps, 0))
    return {"pred": _up(out, h, w)}


def predict(enc, dec, frame_prev, frame_next, mvs_left, mvs_right, n, feature_based, no_warp):
    """flow/model.py:109-113."""
    fn = predict_feature if feature_based else predict_segmentation
    return fn(enc, dec, frame_prev, frame_next, mvs_left, mvs_right, n, no_warp)


def warp_batch(inp, mvs, index_list, n_list, no_warp):
    """flow/model.py:92-106 (including the :102 quirk: shape[1]/shape[2] vs (i_h, i_w))."""
    i_h, i_w = inp.shape[2], inp.shape[3]
    outs = []
    for i, index in enumerate(index_list):
        cur = inp[i].unsqueeze(0)
        if not no_warp:
            for j in range(index):
                cur = warp(cur, mvs[j][i].unsqueeze(0), no_warp)
            if cur.shape[1] != i_h or cur.shape[2] != i_w:
                cur = F.interpolate(cur, size=(i_h, i_w), mode="bilinear", align_corners=True)
        outs.append(cur * ((n_list[i] - index) / n_list[i]))
    return torch.cat(outs)


def forward(enc, dec, frame_prev, frame_next, mvs_left, mvs_right, left_index, right_index, feature_based, no_warp):
    """Eval path of flow/model.py:35-88 (the training-only branch :37-43 is out of scope)."""
    left_index = [int(i) for i in left_index]
    right_index = [int(i) for i in right_index]
    n_list = [a + b for a, b in zip(left_index, right_index)]
    h, w = frame_prev.shape[2], frame_prev.shape[3]
    f_prev, f_next = enc(frame_prev), enc(frame_next)
    if feature_based:
        f = warp_batch(f_prev, mvs_left, left_index, n_list, no_warp) + warp_batch(f_next, mvs_right, right_index, n_list, no_warp)
        return {"pred": _up(dec(f), h, w)}
    o = warp_batch(dec(f_prev), mvs_left, left_index, n_list, no_warp) + warp_batch(dec(f_next), mvs_right, right_index, n_list, no_warp)
    return {"pred": _up(o, h, w)}


def postprocess(output, size=(1072, 1920)):
    """flow/base.py:275-277: upsample, argmax (first max wins), uint8."""
    output = F.interpolate(output, size, mode="bilinear", align_corners=True)
    return output.max(1)[1].to(torch.uint8)


def intersection_and_union(output, target, K, ignore_index=255):
    """util/util.py:36-47 (numpy)."""
    output = np.asarray(output).reshape(-1).copy()
    target = np.asarray(target).reshape(-1)
    output[np.where(target == ignore_index)[0]] = ignore_index
    inter = output[np.where(output == target)[0]]
    ai, _ = np.histogram(inter, bins=np.arange(K + 1))
    ao, _ = np.histogram(output, bins=np.arange(K + 1))
    at, _ = np.histogram(target, bins=np.arange(K + 1))
    return ai, ao + at - ai, at


def miou(inter_sum, union_sum):
    """flow/base.py:332-336."""
    return float(np.mean(inter_sum / (union_sum + 1e-10)))
