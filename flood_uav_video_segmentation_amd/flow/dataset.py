"""Predict-split input side of the hot path: which frames / grids make up window i and how they become the
batch dict FlowBaseModel.predict_step consumes (reference flow/dataset.py:61-64, 80-146, 198-216, 218-240;
transforms flow/base.py:426-431 -> Resize, ToTensor, Normalize of flow/transform.py:26-106).

Only the `split == "predict"` path is mirrored (training / validation sampling is out of scope).  Decoding is done
with PIL (the reference uses skimage.io.imread), resize + normalisation run on the GPU.
"""
import os

import numpy as np
import torch

from .. import ops
from .grids import load_grid

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
            x = ops.resize_bilinear(x, self.size, align_corners=False)                  # Resize: cv2.INTER_LINEAR (:91-106)
        mean = torch.tensor(MEAN, device=self.device).view(1, 3, 1, 1)
        std = torch.tensor(STD, device=self.device).view(1, 3, 1, 1)
        return (x - mean) / std                                                          # Normalize (:56-86)

    def __getitem__(self, index):
        if not 0 <= index < self.length:
            raise IndexError(index)
        f_index, prev_real, next_real = self.indices(index)
        item = {"frame_prev": self._frame(prev_real), "frame_next": self._frame(next_real), "frame_id": f_index}
        if self.no_warp:
            # placeholders whose COUNT still encodes n (flow/dataset.py:198-205, flow/base.py:266)
            item["mvs_left"] = [torch.zeros(1, 1, device=self.device) for _ in range(self.frame_delta - 1)]
            item["mvs_right"] = [torch.zeros(1, 1, device=self.device) for _ in range(self.frame_delta - 1)]
        else:
            fwd, inv = self.grid_ids(index)
            item["mvs_left"] = [load_grid(self.grid_path(i, "grids"))[None].to(self.device) for i in fwd]
            item["mvs_right"] = [load_grid(self.grid_path(i, "inv_grids"))[None].to(self.device) for i in inv]
        return item
