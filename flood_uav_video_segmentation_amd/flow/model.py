"""FlowModel on the HIP path -- drop-in for the reference's flow/model.py (same constructor,
attributes, forward / predict / warp / warp_batch signatures and return dicts).

What differs is only HOW the work is done:
  * both key frames go through the encoder / decoder as ONE batch of two;
  * the tail of predict_segmentation (upsample, warp chains at grid resolution, linear fusion)
    is one fused HIP launch sequence (fs_seg_tail) instead of ~25 torch ops;
  * every warp / resize / blend is a HIP kernel (ops.py); nothing falls back to torch arithmetic.
`self.model` may be any object exposing `.encoder` / `.decoder` callables (as in the reference);
the HIP network mirrors (model/pspnet.py, model/deeplabv3.py) are the intended ones.
"""
import contextlib

import numpy as np
import torch
from torch import nn

from .. import ops


def get_default_grid():
    """Identity block-motion grid of a 1920x1072 frame cut in 16x16 blocks: float64 [67,120,2],
    last dim (x, y) in [-1,1] = block centres (reference flow/model.py:10-21)."""
    width, height, block = 1920, 1072, 16
    nby, nbx = height // block, width // block
    cx = (np.arange(nbx, dtype=np.float64) * block + block // 2) / width * 2 - 1
    cy = (np.arange(nby, dtype=np.float64) * block + block // 2) / height * 2 - 1
    grid = np.empty((nby, nbx, 2), dtype=np.float64)
    grid[..., 0] = cx[None, :]
    grid[..., 1] = cy[:, None]
    return grid


def _region(profiler, name):
    """The reference passes Lightning's profiler; any object with .profile(name) works, None is allowed."""
    if profiler is None:
        return contextlib.nullcontext()
    return profiler.profile(name)


class KeyframeCache:
    """One-slot cache of a key frame's network output.  Consecutive predict windows share a key frame -- window i's
    `frame_next` is window i+1's `frame_prev`: same frame index, deterministic transform (flow/dataset.py:112-114) -- so a
    video needs ONE new key-frame inference per window, not two.  The caller names the frames:

        cache = KeyframeCache()
        fm.predict(prev, nxt, mvl, mvr, n, profiler, key_cache=cache.window(prev_id, next_id))

    The ids only have to be hashable and to name a frame uniquely for as long as the cache lives: when frame numbers restart
    per video, use (video, frame) pairs -- or call clear() between videos (FlowPredictor.reset() does).  What is kept is
    whatever the mode propagates: decoder logits (segmentation mode), encoder features (feature mode) or the per-crop logits of
    the sliding-crop route; a tag keeps different modes / geometries / weight generations apart (reloading the network's
    state_dict invalidates the slot).  Results are bit-identical to the uncached call: a frame's network output does not depend
    on the batch it was computed in."""

    def __init__(self):
        self.frame_id, self.tag, self.value = None, None, None
        self.hits = self.misses = 0

    def clear(self):
        self.frame_id, self.tag, self.value = None, None, None

    def window(self, prev_id, next_id):
        return _KeyWindow(self, prev_id, next_id)


class _KeyWindow:
    def __init__(self, cache, prev_id, next_id):
        self.cache, self.prev_id, self.next_id = cache, prev_id, next_id

    def prev(self, tag):
        c = self.cache
        if self.prev_id is not None and c.frame_id == self.prev_id and c.tag == tag:
            c.hits += 1
            return c.value
        c.misses += 1
        return None

    def store_next(self, tag, value):
        self.cache.frame_id, self.cache.tag, self.cache.value = self.next_id, tag, value


class FlowModel(nn.Module):
    def __init__(self, model, feature_based=True, no_warp=False, no_interpolation_percentage=0.0):
        super().__init__()
        self.model = model
        self.feature_based = feature_based
        self.no_warp = no_warp
        self.no_interpolation_percentage = no_interpolation_percentage
        # plain attribute (not a buffer), moved to the device lazily -- as the reference does (:32, :155-156)
        self.default_motion_vector = torch.from_numpy(get_default_grid()).float().unsqueeze(0)
        # extension (A/B switch, not a constructor argument of the reference): predict_feature's tail as fs_feat_tail (True) or
        # op by op as the reference issues it (False); the two give the same bits
        self.fused_feature_tail = True

    # ------------------------------------------------------------------------------------ helpers
    def _encode(self, *frames):
        """Encoder over the key frames as ONE batch -> [sum(B_i), C, fh, fw] (slice it, do not re-cat).  The HIP mirrors read the
        separate frame tensors in place (`encode_frames`); any other module gets the concatenation."""
        many = getattr(self.model, "encode_frames", None)
        if many is not None:
            return many(*frames)
        x = frames[0] if len(frames) == 1 else torch.cat(frames, 0)
        return self.model.encoder(x)

    def _segment(self, *frames):
        """decoder(encoder(frames)) as ONE batch -> low-resolution logits [sum(B_i), K, fh, fw].  Networks that offer a
        fused `segment` (the HIP mirrors) are called once; any other module goes through .encoder / .decoder."""
        seg = getattr(self.model, "segment", None)
        if seg is not None and (len(frames) == 1 or hasattr(self.model, "encode_frames")):
            # the HIP mirrors (they advertise the multi-tensor call with `encode_frames`) read separate tensors in place
            # (fs_segment_forward2): no torch.cat on the window path.  A user network's segment(x) only ever sees ONE tensor.
            return seg(*frames)
        x = frames[0] if len(frames) == 1 else torch.cat(frames, 0)
        return seg(x) if seg is not None else self.model.decoder(self.model.encoder(x))

    def _tag(self, kind, h, w):
        """Cache tag of a key frame's output: mode, frame geometry and the generation of the network's weights (the HIP mirrors
        bump it in load_state_dict), so that a cached output never survives a change of any of them."""
        return (kind, h, w, getattr(getattr(self.model, "_hip_net", None), "generation", 0))

    def _key_outputs(self, fn, tag, frame_prev, frame_next, key_cache):
        """(out_prev, out_next) of `fn` = _segment / _encode for the two key frames of a window.  With a key_cache the
        previous key frame's output is taken from the last window when it is the same frame, and only frame_next runs."""
        if frame_next is None:
            return fn(frame_prev)[0:1], None
        if key_cache is not None:
            cached = key_cache.prev(tag)
            if cached is not None:
                nxt = fn(frame_next)[0:1]
                key_cache.store_next(tag, nxt)
                return cached, nxt
        outs = fn(frame_prev, frame_next)
        if key_cache is not None:
            key_cache.store_next(tag, outs[1:2])
        return outs[0:1], outs[1:2]

    @staticmethod
    def _fit(t, h, w):
        if t.shape[2] != h or t.shape[3] != w:
            t = ops.resize_bilinear(t, (h, w), align_corners=True)
        return t

    def _fit_out(self, out, h, w):
        """Decoder output -> frame size.  The reference resizes (flow/model.py:68,179); a network that pads its input to a
        patch multiple (the Segmenter mirror) offers `fit_output` and gets its padding cropped instead."""
        fit = getattr(self.model, "fit_output", None)
        return fit(out, h, w) if fit is not None else self._fit(out, h, w)

    def _decode_fit(self, stack, h, w, with_mask=False):
        """decoder -> frame size (-> argmax): (logits, mask or None).  A network whose decoder ends in upsample + unpadding (the
        Segmenter mirror's `decode_fit`) does the three in one launch; bit-identical to the steps taken one by one."""
        fused = getattr(self.model, "decode_fit", None)
        if fused is not None:
            return fused(stack, h, w, with_mask)
        out = self._fit_out(self.model.decoder(stack), h, w)
        return out, (ops.argmax_u8(out) if with_mask else None)

    def warp(self, frame, motion_vectors):
        """grid_sample(bilinear, border, align_corners=False); identity when no_warp (reference :244-249)."""
        if self.no_warp:
            return frame
        return ops.grid_sample(frame, motion_vectors, align_corners=False)

    # ------------------------------------------------------------------------------------ eval forward
    def forward(self, frame_current, frame_prev, frame_next, mvs_left, mvs_right, left_index, right_index):
        """One interpolated frame per sample from two key frames (reference :35-53, eval path)."""
        if self.training:
            raise NotImplementedError("FlowModel(HIP) is an inference path; call .eval() (training branch: flow/model.py:37-43)")
        left = [int(i) for i in left_index]
        right = [int(i) for i in right_index]
        total = [a + b for a, b in zip(left, right)]
        if self.feature_based:
            return self.forward_feature(frame_prev, frame_next, mvs_left, mvs_right, left, right, total)
        return self.forward_segmentation(frame_prev, frame_next, mvs_left, mvs_right, left, right, total)

    def forward_feature(self, frame_prev, frame_next, mvs_left, mvs_right, left_index, right_index, n_list):
        """Warp + weight the encoder FEATURES of the two key frames, decode their sum (reference :55-70)."""
        h, w = frame_prev.shape[2], frame_prev.shape[3]
        nb = frame_prev.shape[0]
        feats = self._encode(frame_prev, frame_next)
        mixed = ops.blend(self.warp_batch(feats[:nb], mvs_left, left_index, n_list), 1.0,
                          self.warp_batch(feats[nb:], mvs_right, right_index, n_list), 1.0)
        return {"pred": self._decode_fit(mixed, h, w)[0]}

    def forward_segmentation(self, frame_prev, frame_next, mvs_left, mvs_right, left_index, right_index, n_list):
        """Segment the two key frames, warp + weight the LOGITS and add them (reference :73-88)."""
        h, w = frame_prev.shape[2], frame_prev.shape[3]
        nb = frame_prev.shape[0]
        lows = self._segment(frame_prev, frame_next)
        out = ops.blend(self.warp_batch(lows[:nb], mvs_left, left_index, n_list), 1.0,
                        self.warp_batch(lows[nb:], mvs_right, right_index, n_list), 1.0)
        return {"pred": self._fit(out, h, w)}

    def warp_batch(self, input, mvs, index_list, n_list):
        """Per-sample chains of `index` warps, resized back and weighted by (n-index)/n (reference :92-106).
        The reference's size test compares shape[1]/shape[2] with (i_h, i_w) (:102), i.e. channels vs
        height: it is almost always true, so the resize runs whenever warping is enabled -- kept."""
        i_h, i_w = input.shape[2], input.shape[3]
        nhwc = ops.is_channels_last_dense(input) and input.shape[1] > 1 and not input.is_contiguous()
        out = (ops.empty_nhwc(len(index_list), input.shape[1], i_h, i_w, input.device) if nhwc else
               torch.empty((len(index_list), input.shape[1], i_h, i_w), dtype=torch.float32, device=input.device))
        for i, index in enumerate(index_list):
            cur = input[i:i + 1]
            if not self.no_warp:
                for j in range(index):
                    cur = self.warp(cur, mvs[j][i:i + 1])
                if cur.shape[1] != i_h or cur.shape[2] != i_w:
                    cur = ops.resize_bilinear(cur, (i_h, i_w), align_corners=True)
            ops.blend(cur, (n_list[i] - index) / n_list[i], out=out[i:i + 1])  # written in place: no per-sample copy
        return out

    # ------------------------------------------------------------------------------------ inference
    def predict(self, *args, **kwargs):
        if self.feature_based:
            return self.predict_feature(*args, **kwargs)
        return self.predict_segmentation(*args, **kwargs)

    def predict_segmentation(self, frame_prev, frame_next, mvs_left, mvs_right, n, profiler=None, key_cache=None, with_mask=False):
        """Segment the key frames, propagate the LOGITS (reference :184-241).
        Returns {"pred": [n,K,h,w]} ([1,K,h,w] when frame_next is None).  key_cache: see KeyframeCache (extension).
        with_mask=True (extension): the fused tail also emits the per-frame argmax it has in registers anyway -- an extra
        "mask" entry, uint8 [n,h,w] = pred.max(1)[1] (flow/base.py:276) -- so that a caller who wants both does not re-read
        the logits for it."""
        h, w = frame_prev.shape[2], frame_prev.shape[3]
        with _region(profiler, "predict_encoder"), _region(profiler, "predict_decoder"):
            lo_prev, lo_next = self._key_outputs(self._segment, self._tag("seg", h, w), frame_prev, frame_next, key_cache)
        with _region(profiler, "predict_warp"), _region(profiler, "predict_fusion"):
            logits, mask = ops.seg_tail(lo_prev, lo_next, mvs_left, mvs_right, n, (h, w), self.no_warp, want_logits=True, want_mask=with_mask)
        return {"pred": logits, "mask": mask} if with_mask else {"pred": logits}

    def predict_masks(self, frame_prev, frame_next, mvs_left, mvs_right, n, profiler=None, key_cache=None):
        """Same pipeline, but the fused tail emits the per-frame argmax directly: uint8 [n,h,w].
        (Extension for the native-resolution timed region of bench.py; not a reference method.)"""
        h, w = frame_prev.shape[2], frame_prev.shape[3]
        with _region(profiler, "predict_encoder"), _region(profiler, "predict_decoder"):
            lo_prev, lo_next = self._key_outputs(self._segment, self._tag("seg", h, w), frame_prev, frame_next, key_cache)
        with _region(profiler, "predict_fusion"):
            _, mask = ops.seg_tail(lo_prev, lo_next, mvs_left, mvs_right, n, (h, w), self.no_warp, want_logits=False, want_mask=True)
        return mask

    def predict_feature(self, frame_prev, frame_next, mvs_left, mvs_right, n, profiler=None, key_cache=None, with_mask=False):
        """Propagate encoder FEATURES, decode all n maps in one batch (reference :116-181).  with_mask: as in predict_segmentation."""
        h, w = frame_prev.shape[2], frame_prev.shape[3]
        with _region(profiler, "predict_encoder"):
            f, f_next = self._key_outputs(self._encode, self._tag("feat", h, w), frame_prev, frame_next, key_cache)
        f_h, f_w = f.shape[2], f.shape[3]
        # the n maps the decoder sees are produced straight into ONE batch tensor (the reference stacks them with torch.cat, :173-176)
        nmaps = n if f_next is not None else 1
        nhwc = ops.is_channels_last_dense(f) and f.shape[1] > 1 and not f.is_contiguous()
        if self.fused_feature_tail and nhwc and f.shape[0] == 1 and f.shape[1] % 4 == 0 and f.dtype == torch.float32:
            # the HIP mirrors' NHWC features: warp chains at grid resolution + ONE launch that writes every map of the decoder's batch
            # (fs_feat_tail) -- bit-identical to the op-by-op route below, without its eight upsampled maps and 67x120 resample
            if not self.no_warp and self.default_motion_vector.device != f.device:
                self.default_motion_vector = self.default_motion_vector.to(device=f.device)
            with _region(profiler, "predict_warp"), _region(profiler, "predict_fusion"):
                stack = ops.feat_tail(f, f_next, mvs_left, mvs_right, n, self.no_warp, None if self.no_warp else self.default_motion_vector)
            with _region(profiler, "predict_decoder"):
                out, mask = self._decode_fit(stack, h, w, with_mask)
            return {"pred": out, "mask": mask} if with_mask else {"pred": out}
        stack = (ops.empty_nhwc(nmaps, f.shape[1], f_h, f_w, f.device) if nhwc else
                 torch.empty((nmaps, f.shape[1], f_h, f_w), dtype=torch.float32, device=f.device))
        fwd, bwd = [], []
        if f_next is not None and not self.no_warp:
            with _region(profiler, "predict_warp"):
                cur = f
                for m in mvs_left:
                    cur = self.warp(cur, m)
                    fwd.append(self._fit(cur, f_h, f_w))
                cur = f_next
                for m in mvs_right:
                    cur = self.warp(cur, m)
                    bwd.append(self._fit(cur, f_h, f_w))
        if not self.no_warp:
            # the key-frame feature goes through the 67x120 identity grid, align_corners=True (:154-159)
            if self.default_motion_vector.device != f.device:
                self.default_motion_vector = self.default_motion_vector.to(device=f.device)
            g0 = ops.grid_sample(f, self.default_motion_vector, align_corners=True)
            if g0.shape[2] != f_h or g0.shape[3] != f_w:
                f = ops.resize_bilinear(g0, (f_h, f_w), align_corners=True, out=stack[0:1])
            else:
                f = ops.blend(g0, 1.0, out=stack[0:1])
        else:
            f = ops.blend(f, 1.0, out=stack[0:1])  # 1.0 * x is x: a HIP copy into the batch slot
        if f_next is not None:
            with _region(profiler, "predict_fusion"):
                for p in range(1, n):
                    if self.no_warp:
                        ops.blend(f, (n - p) / n, f_next, p / n, out=stack[p:p + 1])
                    else:
                        ops.blend(fwd[p - 1], (n - p) / n, bwd[n - p - 1], p / n, out=stack[p:p + 1])
        with _region(profiler, "predict_decoder"):
            out, mask = self._decode_fit(stack, h, w, with_mask)
        return {"pred": out, "mask": mask} if with_mask else {"pred": out}
