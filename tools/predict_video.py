#!/usr/bin/env python3
"""Stand-alone counterpart of the reference's `predict_flow.sh` run (Lightning `predict` of FlowBaseModel,
flow/base.py:236-343) on the HIP path: walks the key-frame windows of one video, segments the key frames, interpolates the
frames in between and writes the colourised masks.

    python tools/predict_video.py --data-root dataset/flow --video-id florida-01 --frame-delta 5 \\
        --arch pspnet --layers 50 --ckpt logs/<run>/last.ckpt --out out/florida-01

Directory layout read (flow/dataset.py:222-240): <data-root>/frames/<video-id>/{images/<i>.jpg, grids/<i>.npy, inv_grids/<i>.npy}.
Checkpoints are loaded with `torch.load(..., weights_only=True)` (a Lightning `state_dict` with the `model_G.model.` prefix, or
a bare state_dict); `--synthetic-weights` uses the seeded random weights of the test-suite instead (no checkpoint ships with
the reference).  Multi-GPU: launch with torchrun; each rank takes a contiguous block of windows, the one temporal-consistency
pair across each block boundary is scored from the neighbour's last mask (one all_gather at the end), metrics are reduced.
Consecutive windows share a key frame: it is segmented once (`--no-keyframe-cache` recomputes it, as the reference does).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import ops, shard, synth  # noqa: E402
from flood_uav_video_segmentation_amd.flow.dataset import PredictWindows  # noqa: E402
from flood_uav_video_segmentation_amd.flow.model import FlowModel  # noqa: E402
from flood_uav_video_segmentation_amd.flow.predict import PALETTE, FlowPredictor, colorize  # noqa: E402
from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402


def load_weights(net, args):
    if args.synthetic_weights:
        make = synth.make_pspnet_state if args.arch == "pspnet" else synth.make_deeplab_state
        net.load_state_dict(make(args.layers, args.classes, seed=0))
        return
    ckpt = torch.load(args.ckpt, map_location="cpu", weights_only=True)
    state = ckpt.get("state_dict", ckpt)
    for prefix in ("model_G.model.", "model.model.", "model."):
        sub = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
        if sub:
            state = sub
            break
    net.load_state_dict(state)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data-root", required=True)
    ap.add_argument("--video-id", default="florida-01")                      # data.predict_v_id
    ap.add_argument("--frame-delta", type=int, default=5)                    # data.frame_delta
    ap.add_argument("--arch", choices=("pspnet", "deeplabv3"), default="pspnet")
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--ckpt")
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--feature-based", action="store_true")                  # model.feature_based
    ap.add_argument("--no-warp", action="store_true")                        # model.no_warp
    ap.add_argument("--no-cropping", action="store_true")                    # model.no_cropping: whole frame instead of sliding crops
    ap.add_argument("--crop", type=int, nargs=2, default=(713, 713), metavar=("H", "W"))   # model.test_h / test_w
    ap.add_argument("--size", type=int, nargs=2, default=(1072, 1920), metavar=("H", "W"))  # transform_predict Resize, and the output size
    ap.add_argument("--palette", help="colors.txt (dataset/flow/list/colors.txt); default: the 5-class flood palette")
    ap.add_argument("--out", help="directory for <frame>.png (model.save_images); omit to only time and score")
    ap.add_argument("--no-metrics", action="store_true")                     # model.compute_metrics False
    ap.add_argument("--no-keyframe-cache", action="store_true", help="segment both key frames of every window (reference behaviour)")
    args = ap.parse_args()
    if not args.synthetic_weights and not args.ckpt:
        ap.error("give --ckpt or --synthetic-weights")

    rank, local_rank, world = shard.init()
    torch.cuda.set_device(local_rank)
    torch.set_grad_enabled(False)

    class HP:
        layers, classes, pretrained = args.layers, args.classes, False

    net = (FlowPSPNet if args.arch == "pspnet" else FlowDeepLabv3)(HP()).eval()
    load_weights(net, args)
    fm = FlowModel(net, feature_based=args.feature_based, no_warp=args.no_warp).eval()
    pred = FlowPredictor(fm, classes=args.classes, out_size=tuple(args.size), crop=None if args.no_cropping else tuple(args.crop),
                         compute_metrics=not args.no_metrics, cache_keyframes=not args.no_keyframe_cache)
    ds = PredictWindows(args.data_root, args.video_id, frame_delta=args.frame_delta, no_warp=args.no_warp, size=tuple(args.size))
    palette = np.loadtxt(args.palette).astype("uint8") if args.palette else PALETTE
    if args.out and rank == 0:
        os.makedirs(args.out, exist_ok=True)
    shard.barrier()

    # windows are independent units given their two key frames: each rank takes a contiguous block (SURVEY 8e "frame-window
    # sharding"), so that the temporal-consistency pairs inside a block are the reference's; the pair across each block
    # boundary (this rank's first frame vs the previous block's last) is scored after the loop from the neighbour's mask
    mine = shard.window_block(len(ds), rank, world)
    frames = 0
    first_mask = last_mask = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in mine:
        item = ds[w]
        masks = pred.predict_window(item["frame_prev"], item["frame_next"], item["mvs_left"], item["mvs_right"], to_host=False,
                                    key_ids=item["key_ids"])
        if first_mask is None:
            first_mask = masks[0].clone()
        last_mask = masks[-1]
        frames += masks.shape[0]
        if args.out:
            from PIL import Image

            rgb = colorize(masks, palette).cpu().numpy()
            for p in range(rgb.shape[0]):
                Image.fromarray(rgb[p]).save(os.path.join(args.out, f"{item['frame_id'] + p}.png"))
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0  # this rank's own work: the end-of-run exchange below waits for the slowest rank and is not part of it
    if world > 1 and not args.no_metrics:
        dev = torch.device("cuda", local_rank)
        blank = torch.zeros(tuple(args.size), dtype=torch.uint8, device=dev)
        neighbour = shard.exchange_boundary(last_mask if last_mask is not None else blank, len(mine) > 0, dev)
        if neighbour is not None:  # flow/base.py:284-291 for p == 0 with last_output = the previous block's final frame
            pred.hist = ops.iou_hist(first_mask, neighbour, args.classes, 255, pred.hist)
    torch.cuda.synchronize()
    hist = pred.hist if pred.hist is not None else torch.zeros(3, args.classes, dtype=torch.int64)
    hist, frames, seconds = shard.reduce_run(hist.cpu() if world == 1 else hist, frames, seconds, "cpu" if world == 1 else torch.device("cuda", local_rank))
    if rank == 0:
        h = hist.double()
        line = f"{frames} frames of {args.video_id} in {seconds:.2f} s = {frames / seconds:.1f} FPS on {world} GPU(s)"
        if not args.no_metrics and float(h[2].sum()) > 0:
            inter, union, target = h[0], h[1] + h[2] - h[0], h[2]
            line += (f"; temporal consistency mIoU {float((inter / (union + 1e-10)).mean()):.4f}"
                     f" mAcc {float((inter / (target + 1e-10)).mean()):.4f} acc {float(inter.sum() / (target.sum() + 1e-10)):.4f}")
        print(line)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
