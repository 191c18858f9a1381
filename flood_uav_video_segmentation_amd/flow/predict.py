"""predict_step on the HIP path -- mirrors the timed unit and the post-processing of the reference's
FlowBaseModel.predict_step / on_predict_end (flow/base.py:236-343) without Lightning:

    p = FlowPredictor(flow_model, classes=5, out_size=(1072, 1920), crop=None)
    masks = p.predict_window(frame_prev, frame_next, mvs_left, mvs_right)   # uint8 numpy [n, 1072, 1920]
    p.temporal_consistency()                                               # mIoU / mAcc / accuracy between consecutive frames
"""
import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, ptr, stream_ptr
from . import crops

PALETTE = np.array([[0, 0, 0], [30, 95, 170], [65, 117, 5], [212, 98, 1], [255, 244, 116]], dtype=np.uint8)  # dataset/flow/list/colors.txt


class FlowPredictor:
    """cache_keyframes=True + `key_ids=(prev_frame_id, next_frame_id)` in predict_window: the network output of the previous
    window's `frame_next` is reused when it is this window's `frame_prev` (flow/dataset.py:112-114 builds consecutive windows
    that way), so a video costs one new key-frame inference per window; masks are bit-identical to the uncached run."""

    def __init__(self, flow_model, classes=5, out_size=(1072, 1920), crop=None, compute_metrics=True, ignore_index=255,
                 cache_keyframes=False):
        from .model import KeyframeCache

        self.model = flow_model
        self.key_cache = KeyframeCache() if cache_keyframes else None
        self.classes = classes
        self.out_size = tuple(out_size)
        self.crop = crop  # (crop_h, crop_w) -> sliding crops (no_cropping=False); None -> whole frame (no_cropping=True)
        self.compute_metrics = compute_metrics
        self.ignore_index = ignore_index
        self.last_output = None  # flow/base.py:247, :295
        self.hist = None         # int64[3,K]: intersection, |pred|, |target| accumulated over the run

    def reset(self):
        """Start a new video: forget the cached key frame and the last mask (the temporal-consistency metric pairs each frame with
        its predecessor, flow/base.py:247,295 -- which must not be another video's last frame).  The histogram keeps running."""
        if self.key_cache is not None:
            self.key_cache.clear()
        self.last_output = None

    def predict_window(self, frame_prev, frame_next, mvs_left, mvs_right, profiler=None, to_host=True, key_ids=None, key_cache=None):
        """key_cache: a KeyframeCache to use for this call instead of the predictor's own (predict_clip's fallback passes a
        cache that lives for the clip only)."""
        assert frame_prev.shape[0] == 1                      # flow/base.py:263
        assert len(mvs_left) == len(mvs_right)               # :264
        n = len(mvs_left) + 1                                # :266 -- the list length encodes n, also for no_warp dummies
        cache = key_cache if key_cache is not None else self.key_cache
        kc = cache.window(*key_ids) if (cache is not None and key_ids is not None) else None
        if self.crop is None:
            extra = {} if kc is None else {"key_cache": kc}
            if self._native(frame_prev) and not getattr(self.model, "feature_based", True) and hasattr(self.model, "predict_masks"):
                # out_size IS the frame size: the align_corners=True resize of :275 is the identity (source index = destination
                # index, weight 0), so :275-276 is the argmax of the logits themselves -- which the fused tail emits without
                # writing the fp32 logits out and reading them back
                masks = self.model.predict_masks(frame_prev, frame_next, mvs_left, mvs_right, n, profiler, **extra)
            else:
                logits = self.model.predict(frame_prev, frame_next, mvs_left, mvs_right, n, profiler, **extra)["pred"]
                masks = ops.resize_argmax_u8(logits, self.out_size)       # :275-276 without the fp32 intermediate
        else:
            # :273 compute_output, then :275-276 (float64 resize + argmax) fused into the canvas's last pass
            _, masks = crops.compute_output(self.model, n, frame_prev, frame_next, mvs_left, mvs_right, self.crop[0], self.crop[1],
                                            self.classes, profiler, want_mask=True, key_cache=kc, out_size=self.out_size, want_canvas=False)
        self._score(masks, n)
        return masks.cpu().numpy() if to_host else masks                # :277

    def _native(self, frame):
        return self.out_size == (frame.shape[2], frame.shape[3])

    def _score(self, masks, n):
        if self.compute_metrics:                                      # :280-295 temporal consistency between consecutive frames
            for p in range(n):
                prev = masks[p - 1] if p > 0 else self.last_output
                if prev is not None:
                    self.hist = ops.iou_hist(masks[p], prev, self.classes, self.ignore_index, self.hist)
            self.last_output = masks[n - 1].clone()

    def predict_clip(self, items, profiler=None, to_host=True, keys_per_pass=2):
        """A clip's consecutive windows (dicts as PredictWindows yields them: frame_prev, frame_next, mvs_left, mvs_right,
        key_ids) with the key-frame cache AND look-ahead: key frames go through the network two at a time (each exactly once),
        and a window is emitted as soon as both of its key frames are there -- every key frame at the efficiency of a full batch
        (a lone frame runs ~10 % slower per frame: half-empty tile rounds).  Yields the masks of every window, in order,
        bit-identical to predict_window on the same windows (a frame's network output does not depend on its batch).

        `items` is consumed LAZILY: windows are pulled only until two not-yet-segmented key frames are at hand, so at most two
        windows' frames (and three key frames' low-resolution logits) are alive at once, however long the clip -- a lazily
        loading iterable such as PredictWindows streams.  Segmentation mode (whole frame or sliding crops); feature mode, a
        network without a fused `segment`, or a window without key_ids take predict_window with a cache that lives for this
        clip only (the predictor's own cache, when it has one).

        keys_per_pass (round 5; whole-frame route): how many NEW key frames go through the network per pass.  2 = one window of
        look-ahead.  4 or 6 trade latency (that many windows are pulled ahead) for throughput: a pass over four frames costs 8 % less per
        frame than two passes over two (11 % at six; profiles/r05_experiments.txt section 9) -- every launch of layers 1-3 is short enough
        to be bound by its fixed cost.  Masks stay bit-identical (a frame's network output does not depend on its batch)."""
        from collections import deque

        from .model import KeyframeCache, _region

        fm = self.model
        lookahead = not getattr(fm, "feature_based", True) and hasattr(fm.model, "segment") and hasattr(fm.model, "encode_frames")
        group = max(2, int(keys_per_pass)) if self.crop is None else 2  # the sliding-crop route batches the crops of TWO frames
        local_cache = self.key_cache if self.key_cache is not None else KeyframeCache()
        store = {}         # frame id -> decoder logits of that key frame ([1,K,fh,fw]; [ncrops,K,fh,fw] on the sliding-crop route)
        queue = []         # key frames not segmented yet, in order of first use: (frame id, tensor)
        pending = deque()  # windows pulled from `items` and not emitted yet
        last_next = None   # the newest emitted window's next key: the window still to come names it as its previous key
        it = iter(items)
        exhausted = False

        def run(frames):  # the queued key frames (up to `group`) through the network as ONE batch
            with _region(profiler, "predict_encoder"), _region(profiler, "predict_decoder"):
                if self.crop is None:
                    lows = fm._segment(*frames)
                    return [lows[j:j + 1] for j in range(len(frames))]
                a, b = crops.segment_crop_windows(fm, frames[0], frames[1] if len(frames) > 1 else None, self.crop[0], self.crop[1])
                return [a] if b is None else [a, b]

        def emit(w):
            assert w["frame_prev"].shape[0] == 1 and len(w["mvs_left"]) == len(w["mvs_right"])   # flow/base.py:263-264
            n = len(w["mvs_left"]) + 1
            lo_prev, lo_next = store[w["key_ids"][0]], store[w["key_ids"][1]]
            h, wd = w["frame_prev"].shape[2], w["frame_prev"].shape[3]
            if self.crop is None and self._native(w["frame_prev"]):
                with _region(profiler, "predict_warp"), _region(profiler, "predict_fusion"):  # identity resize: see predict_window
                    _, masks = ops.seg_tail(lo_prev, lo_next, w["mvs_left"], w["mvs_right"], n, (h, wd), fm.no_warp, want_logits=False, want_mask=True)
            elif self.crop is None:
                with _region(profiler, "predict_warp"), _region(profiler, "predict_fusion"):
                    logits, _ = ops.seg_tail(lo_prev, lo_next, w["mvs_left"], w["mvs_right"], n, (h, wd), fm.no_warp, want_logits=True)
                masks = ops.resize_argmax_u8(logits, self.out_size)
            else:
                _, masks = crops.compute_output(fm, n, w["frame_prev"], w["frame_next"], w["mvs_left"], w["mvs_right"], self.crop[0],
                                                self.crop[1], self.classes, profiler, want_mask=True, out_size=self.out_size,
                                                lows=(lo_prev, lo_next), want_canvas=False)
            self._score(masks, n)
            return masks.cpu().numpy() if to_host else masks

        while True:
            plain = None  # a window that cannot take the look-ahead route (no key_ids / feature mode / foreign network)
            while len(queue) < group and not exhausted and plain is None:
                if pending and not queue and all(k in store for k in pending[0]["key_ids"]):
                    break  # nothing to wait for: emit before pulling more
                try:
                    w = next(it)
                except StopIteration:
                    exhausted = True
                    break
                if not lookahead or w.get("key_ids") is None:
                    plain = w
                    break
                pending.append(w)
                for fid, t in zip(w["key_ids"], (w["frame_prev"], w["frame_next"])):
                    if fid not in store and all(fid != q for q, _ in queue):
                        queue.append((fid, t))
            # segment what is queued: full groups while there are full groups; a smaller one only when nothing can join it any more
            while len(queue) >= group or (queue and (exhausted or plain is not None)):
                pair, queue = queue[:group], queue[group:]
                for (fid, _), lo in zip(pair, run([t for _, t in pair])):
                    store[fid] = lo
            while pending and all(k in store for k in pending[0]["key_ids"]):
                w = pending.popleft()
                last_next = w["key_ids"][1]
                yield emit(w)
            live = {k for w in pending for k in w["key_ids"]} | {last_next}
            store = {k: v for k, v in store.items() if k in live}  # only what a window still to come can need
            if plain is not None:
                yield self.predict_window(plain["frame_prev"], plain["frame_next"], plain["mvs_left"], plain["mvs_right"], profiler, to_host,
                                          plain.get("key_ids"), key_cache=local_cache)
            elif exhausted and not pending and not queue:
                return

    def temporal_consistency(self):
        """on_predict_end's summary (flow/base.py:330-343): (mIoU, mAcc, accuracy) with the reference's 1e-10 epsilon."""
        if self.hist is None:
            return None
        h = self.hist.cpu().numpy().astype(np.float64)
        inter, union, target = h[0], h[1] + h[2] - h[0], h[2]
        return float(np.mean(inter / (union + 1e-10))), float(np.mean(inter / (target + 1e-10))), float(inter.sum() / (target.sum() + 1e-10))


class FlowEvaluator:
    """validation_step / test_step on the HIP path (flow/base.py:143-176, summary of on_test_epoch_end in
    base/foundation.py): one interpolated frame per item from FlowModel.forward, argmax, intersection / union / target
    histograms against the label, one meter set per test list (Florida = 0, Texas = 1, flow/base.py:170-175).

        ev = FlowEvaluator(flow_model, classes=5, crop=(713, 713))       # crop=None <=> no_cropping=True
        for item in EvalWindows(...): ev.test_step(item, test_idx=0)
        miou, macc, acc, iou_class, acc_class = ev.summary(0)
    """

    def __init__(self, flow_model, classes=5, crop=None, ignore_index=255):
        self.model = flow_model
        self.classes, self.crop, self.ignore_index = classes, crop, ignore_index
        self.hist = {}  # meter id -> int64[3,K] (intersection, |pred|, |target|)

    def forward(self, frame_prev, frame_next, mvs_left, mvs_right, left_index, right_index):
        return self.model(None, frame_prev, frame_next, mvs_left, mvs_right, left_index, right_index)   # flow/base.py:134-135

    def _update(self, meter, pred, label):
        pred, label = pred.to(torch.uint8).contiguous(), label.to(torch.uint8).contiguous()   # ids 0..K-1 and 255 fit
        self.hist[meter] = ops.iou_hist(pred, label, self.classes, self.ignore_index, self.hist.get(meter))

    def validation_step(self, batch):
        out = self.forward(batch["frame_prev"], batch["frame_next"], batch["mvs_left"], batch["mvs_right"], batch["left_index"],
                           batch["right_index"])["pred"]
        pred = ops.argmax_u8(out)
        self._update("val", pred, batch["label"])
        return pred

    def test_step(self, batch, test_idx=0):
        frame_prev, frame_next, label = batch["frame_prev"], batch["frame_next"], batch["label"]
        assert frame_prev.shape[0] == 1 and label.shape[0] == 1                                    # flow/base.py:160
        li, ri = batch["left_index"], batch["right_index"]
        if self.crop is None:
            out = self.forward(frame_prev, frame_next, batch["mvs_left"], batch["mvs_right"], li, ri)["pred"]
            pred = ops.argmax_u8(out)
        else:
            fn = lambda p, q, ml, mr: self.forward(p, q, ml, mr, li, ri)["pred"]                # compute_test_crop (:212-222)
            _, pred = crops.compute_output(self.model, frame_prev.shape[0], frame_prev, frame_next, batch["mvs_left"],
                                           batch["mvs_right"], self.crop[0], self.crop[1], self.classes, want_mask=True,
                                           function=fn)
        self._update(1 if test_idx > 0 else 0, pred, label)
        return pred

    def summary(self, meter=0):
        """(mIoU, mAcc, accuracy, iou_class, accuracy_class) with the reference's 1e-10 epsilon (flow/base.py:332-336)."""
        if meter not in self.hist:
            return None
        h = self.hist[meter].cpu().numpy().astype(np.float64)
        inter, union, target = h[0], h[1] + h[2] - h[0], h[2]
        iou_class, acc_class = inter / (union + 1e-10), inter / (target + 1e-10)
        return float(np.mean(iou_class)), float(np.mean(acc_class)), float(inter.sum() / (target.sum() + 1e-10)), iou_class, acc_class


def colorize(masks_u8, palette=PALETTE):
    """colors[output] (flow/base.py:308-312): uint8 [..., 3] RGB frames for the video writer / PNG dump."""
    lib = _lib.load()
    m = masks_u8.contiguous()
    pal = torch.as_tensor(palette, dtype=torch.uint8, device=m.device).contiguous()
    out = torch.empty(tuple(m.shape) + (3,), dtype=torch.uint8, device=m.device)
    if m.numel() == 0:
        return out
    check(lib.fs_colorize(ptr(m), ptr(pal), pal.shape[0], ptr(out), m.numel(), stream_ptr()))
    return out
