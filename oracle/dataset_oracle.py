"""Oracle for the index arithmetic of FlowData (flow/dataset.py:16-43 `make_dataset`, :80-181 `__getitem__`) and the
label side of the test transforms (flow/transform.py:91-106 Resize -> INTER_NEAREST, :361-371 IgnoreClasses).

PINNED (round 4) for the index arithmetic: tests/golden/dataset_index.npz holds what the reference's OWN FlowData class returns
on a synthetic file set with missing images / grids (predict, val and test splits, frame_delta 5 / 8 / 25); the generator
registers a stand-in for skimage.io.imread -- the only skimage call of flow/dataset.py -- that returns the number in the file name,
and runs with transform=None so flow/transform.py (cv2) is never touched (tests/test_oracle_golden.py::
test_window_indexing_matches_the_references_flowdata).  The restatement follows the source line by line (the `exists` predicate is
passed in, so it runs on an in-memory file set).  IgnoreClasses, Crop('center'), ToTensor and Normalize are pinned to the reference's own
transform_val / transform_test chains at the native size (tests/golden/transforms.npz).  PARITY UNPINNED for an interpolating
Resize: flow/transform.py calls cv2, absent offline; cv2's INTER_NEAREST is restated from its documented rule:
src = min(floor(dst * src_size / dst_size), src_size - 1).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import random

import numpy as np


def make_dataset(lines, frame_delta):
    """flow/dataset.py:16-43: (label, video, frame) of every list line whose frame id is >= frame_delta // 2."""
    out = []
    for line in lines:
        f = line.strip().split(" ")
        if int(f[2]) < frame_delta // 2:
            continue
        out.append((f[0], f[1], int(f[2])))
    return out


def eval_item(exists, index, f_index, frame_delta, split):
    """flow/dataset.py:89-181 for split in {"val", "test", "predict"}; `exists(f_id)` says whether frame f_id has its
    image AND both grid files.  Returns (l, r, prev_real, next_real, left_ids, right_ids): grid ids are frame numbers,
    None marks the identity `default_grid`."""
    if split == "predict":
        l = r = None
        prev, nxt = f_index, f_index + frame_delta
    else:
        l = random.Random(index).randrange(1, frame_delta)
        r = frame_delta - l
        prev, nxt = f_index - l, f_index + r
    prev_real = prev
    while not exists(prev_real):
        prev_real += 1
    next_real = nxt
    while not exists(next_real):
        next_real -= 1
    left, right = [], []
    if split == "predict":
        left = [f_index + i + 1 for i in range(frame_delta - 1)]
        right = [f_index + i + 1 for i in range(frame_delta - 1)]
        right.reverse()
    else:
        for i in range(l):
            g = f_index - l + i + 1
            left.append(g if g > prev_real else None)
        while len(left) < frame_delta - 1:
            left.append(None)
        for i in range(r):
            g = f_index + i + 1
            right.append(g if g <= next_real else None)
        right.reverse()
        while len(right) < frame_delta - 1:
            right.append(None)
    return l, r, prev_real, next_real, left, right


def resize_label_nearest(label, size):
    """cv2.resize(label, (w, h), INTER_NEAREST) on a 2-D uint8 array."""
    h, w = size
    H, W = label.shape
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return label[ys][:, xs]


def ignore_classes(label, classes_to_ignore):
    out = label.copy()
    for c in classes_to_ignore or []:
        out[out == c] = 0
    return out
