"""Predict-split input side of the hot path: which frames / grids make up window i and how they become the
batch dict FlowBaseModel.predict_step consumes (reference flow/dataset.py:61-64, 80-146, 198-216, 218-240;
transforms flow/base.py:426-431 -> Resize, ToTensor, Normalize of flow/transform.py:26-106).

`PredictWindows` mirrors `split == "predict"`, `EvalWindows` the labelled `val` / `test` splits that feed
validation_step / test_step (flow/base.py:143-176); the random training sampling is out of scope.  Decoding is done with
PIL (the reference uses skimage.io.imread), resize + normalisation run on the GPU.
"""
import os
import random

import numpy as np
import torch

from .. import ops
from .grids import load_grid
from .model import get_default_grid

# base/foundation.py:27-31
MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]


class PredictWindows:
    """Window i of a video = key frames (i*delta, (i+1)*delta) + the delta-1 grids in between (flow/dataset.py:112-146)."""

    def __init__(self, data_root, predict_v_id, frame_delta=5, no_warp=False, size=None, device="cuda"):
        self.data_root, self.video_id = data_root, predict_v_id
        self.frame_delta, self.no_warp = frame_delta, no_warp
        self.size = size  # (h, w) of transform_predict's Resize, None = native
        self.device = device
        # flow/dataset.py:64 -- windows = frames // delta
        self.length = len(os.listdir(os.path.join(data_root, "frames", predict_v_id, "images"))) // frame_delta

    def __len__(self):
        return self.length

    # -- paths (flow/dataset.py:222-236)
    def frame_path(self, f_id):
        return os.path.join(self.data_root, "frames", self.video_id, "images", f"{f_id}.jpg")

    def grid_path(self, i, name):
        return os.path.join(self.data_root, "frames", self.video_id, name, f"{i}.npy")

    def _complete(self, f_id):
        return all(os.path.exists(p) for p in (self.frame_path(f_id), self.grid_path(f_id, "grids"), self.grid_path(f_id, "inv_grids")))

    def indices(self, index, max_search=100000):
        """(f_index, prev_real, next_real): the key frames actually used.  A key frame whose image or grids are missing is
        replaced by the next complete one going FORWARD (previous key) / BACKWARD (next key) -- flow/dataset.py:119-131."""
        f_index = index * self.frame_delta
        prev_real, next_real = f_index, f_index + self.frame_delta
        steps = 0
        while not self._complete(prev_real):
            prev_real += 1
            steps += 1
            if steps > max_search:
                raise FileNotFoundError(f"no complete frame at or after {f_index}")
        steps = 0
        while not self._complete(next_real):
            next_real -= 1
            steps += 1
            if steps > max_search or next_real < 0:
                raise FileNotFoundError(f"no complete frame at or before {f_index + self.frame_delta}")
        return f_index, prev_real, next_real

    def grid_ids(self, index):
        """Forward grids for frames f+1..f+delta-1, inverse grids for the same frames REVERSED (flow/dataset.py:138-146)."""
        f_index = index * self.frame_delta
        ids = [f_index + i + 1 for i in range(self.frame_delta - 1)]
        return ids, ids[::-1]

    def _frame(self, f_id):
        from PIL import Image

        img = np.array(Image.open(self.frame_path(f_id)).convert("RGB"))  # writable copy
        x = torch.from_numpy(img).to(self.device).permute(2, 0, 1)[None].float()       # ToTensor (flow/transform.py:26-51)
        if self.size is not None and tuple(x.shape[2:]) != tuple(self.size):
            # Resize: cv2.INTER_LINEAR on the uint8 image (:91-106) = half-pixel bilinear, result stored back as uint8
            x = ops.resize_bilinear(x, self.size, align_corners=False).round_().clamp_(0, 255)
        mean = torch.tensor(MEAN, device=self.device).view(1, 3, 1, 1)
        std = torch.tensor(STD, device=self.device).view(1, 3, 1, 1)
        return (x - mean) / std                                                          # Normalize (:56-86)

    def __getitem__(self, index):
        if not 0 <= index < self.length:
            raise IndexError(index)
        f_index, prev_real, next_real = self.indices(index)
        # key_ids: the frames actually used as keys.  Window i's next key is window i+1's previous key (same index, same
        # deterministic transform: :113-114), which is what FlowPredictor's key-frame cache is keyed on.
        item = {"frame_prev": self._frame(prev_real), "frame_next": self._frame(next_real), "frame_id": f_index,
                "key_ids": (prev_real, next_real)}
        if self.no_warp:
            # placeholders whose COUNT still encodes n (flow/dataset.py:198-205, flow/base.py:266)
            item["mvs_left"] = [torch.zeros(1, 1, device=self.device) for _ in range(self.frame_delta - 1)]
            item["mvs_right"] = [torch.zeros(1, 1, device=self.device) for _ in range(self.frame_delta - 1)]
        else:
            fwd, inv = self.grid_ids(index)
            item["mvs_left"] = [load_grid(self.grid_path(i, "grids"))[None].to(self.device) for i in fwd]
            item["mvs_right"] = [load_grid(self.grid_path(i, "inv_grids"))[None].to(self.device) for i in inv]
        return item


def read_label_list(data_list, frame_delta):
    """make_dataset (flow/dataset.py:16-43): [(label path relative to data_root, video id, frame id)], dropping labelled
    frames closer than frame_delta // 2 to the start of the video.  The list files the reference ships and writes
    (dataset/flow/make_flow.py:107,137) hold THREE fields per line although make_dataset's check asks for four (it would
    reject its own lists); three or four fields are accepted here, anything else raises like the reference."""
    out = []
    with open(data_list) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            f = line.split(" ")
            if len(f) not in (3, 4):
                raise RuntimeError("Image list file read line error : " + line + "\n")
            if int(f[2]) < frame_delta // 2:
                continue
            out.append((f[0], f[1], int(f[2])))
    return out


def resize_label_nearest(label, size):
    """Resize's label branch, cv2.INTER_NEAREST (flow/transform.py:104-105): src = min(floor(dst * src/dst), src - 1)."""
    h, w = size
    H, W = label.shape
    if (H, W) == (h, w):
        return label
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return label[ys][:, xs]


class EvalWindows(PredictWindows):
    """Item i of the `val` / `test` split = labelled frame f with key frames f-l and f+r, l drawn from Random(i) and
    l + r = frame_delta (flow/dataset.py:89-92, 115-117); items carry the batch dimension the DataLoader would add
    (batch_size_test = 1, flow/base.py:164)."""

    def __init__(self, data_root, data_list, split="test", frame_delta=5, no_warp=False, size=None, center_crop=None,
                 classes_ignore=(), device="cuda"):
        if split not in ("val", "test"):
            raise ValueError("EvalWindows mirrors the val / test splits; use PredictWindows for predict")
        self.data_root, self.split = data_root, split
        self.frame_delta, self.no_warp = frame_delta, no_warp
        self.size, self.center_crop = size, center_crop      # Resize target (h, w); Crop('center') size of transform_val
        self.classes_ignore = tuple(classes_ignore or ())    # data_classes_ignore (dataset/flow/config.yaml:6)
        self.device = device
        self.label_list = read_label_list(data_list, frame_delta)
        self.length = len(self.label_list)
        self.video_id = None
        self.default_grid = torch.from_numpy(get_default_grid()).float()   # flow/dataset.py:68 + ToTensor

    def plan(self, index):
        """Pure index arithmetic of item `index`: dict(l, r, prev_real, next_real, left_ids, right_ids); a None grid id
        stands for the identity default grid (flow/dataset.py:147-171)."""
        _, v_id, f_index = self.label_list[index]
        self.video_id = v_id
        delta = self.frame_delta
        l = random.Random(index).randrange(1, delta)
        r = delta - l
        prev_real, next_real = f_index - l, f_index + r
        steps = 0
        while not self._complete(prev_real):
            prev_real, steps = prev_real + 1, steps + 1
            if steps > 100000:
                raise FileNotFoundError(f"no complete frame at or after {f_index - l}")
        while not self._complete(next_real):
            next_real -= 1
            if next_real < 0:
                raise FileNotFoundError(f"no complete frame at or before {f_index + r}")
        left = [(g if g > prev_real else None) for g in range(f_index - l + 1, f_index + 1)]
        left += [None] * (delta - 1 - len(left))
        right = [(g if g <= next_real else None) for g in range(f_index + 1, f_index + r + 1)]
        right.reverse()
        right += [None] * (delta - 1 - len(right))
        return {"video": v_id, "frame": f_index, "l": l, "r": r, "prev_real": prev_real, "next_real": next_real,
                "left_ids": left, "right_ids": right}

    def _label(self, rel_path):
        from PIL import Image

        lab = np.array(Image.open(os.path.join(self.data_root, rel_path)))
        if lab.ndim != 2:
            raise RuntimeError(f"label {rel_path} is not single-channel")
        if self.size is not None:
            lab = resize_label_nearest(lab, self.size)
        lab = lab.copy()
        for c in self.classes_ignore:                         # IgnoreClasses (flow/transform.py:361-371)
            lab[lab == c] = 0
        return torch.from_numpy(lab).long()

    def __getitem__(self, index):
        if not 0 <= index < self.length:
            raise IndexError(index)
        p = self.plan(index)
        frame_prev, frame_next = self._frame(p["prev_real"]), self._frame(p["next_real"])
        label = self._label(self.label_list[index][0])[None].to(self.device)
        n1 = self.frame_delta - 1
        if self.no_warp:
            left = [torch.zeros(1, 1, device=self.device) for _ in range(n1)]
            right = [torch.zeros(1, 1, device=self.device) for _ in range(n1)]
        else:
            def grid(g, name):
                t = self.default_grid if g is None else load_grid(self.grid_path(g, name))
                return t[None].to(self.device)
            left = [grid(g, "grids") for g in p["left_ids"]]
            right = [grid(g, "inv_grids") for g in p["right_ids"]]
        if self.center_crop is not None:                      # Crop(..., 'center') of transform_val (flow/transform.py:183-211)
            from .crops import crop_motion_vector

            ch, cw = self.center_crop
            h, w = label.shape[-2:]
            assert h > ch and w > cw
            ho, wo = int((h - ch) / 2), int((w - cw) / 2)
            frame_prev = frame_prev[:, :, ho:ho + ch, wo:wo + cw].contiguous()
            frame_next = frame_next[:, :, ho:ho + ch, wo:wo + cw].contiguous()
            label = label[:, ho:ho + ch, wo:wo + cw].contiguous()
            if not self.no_warp:
                left, right = crop_motion_vector(left, right, h, w, ch, cw, ho, wo)
        return {"frame_prev": frame_prev, "frame_next": frame_next, "mvs_left": left, "mvs_right": right, "label": label,
                "left_index": torch.tensor([p["l"]]), "right_index": torch.tensor([p["r"]])}
