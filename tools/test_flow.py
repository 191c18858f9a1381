#!/usr/bin/env python3
"""Stand-alone counterpart of the reference's `test` run of FlowBaseModel (flow/base.py:156-176 `test_step`, summary of
`test_epoch_end` in base/foundation.py:224-259) on the HIP path: labelled frames of the `test` lists -> mIoU / mAcc / accuracy.

    python tools/test_flow.py --data-root dataset/flow --list dataset/flow/list/all/test.txt [--list2 .../test2.txt] \\
        --frame-delta 5 --arch pspnet --ckpt logs/<run>/last.ckpt

Each list line is `<label png> <video id> <frame id>`; frames and grids are read from <data-root>/frames/<video id>/...
(flow/dataset.py:16-43, 89-181).  Checkpoints are loaded with `torch.load(..., weights_only=True)`.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flood_uav_video_segmentation_amd import synth  # noqa: E402
from flood_uav_video_segmentation_amd.flow.dataset import EvalWindows  # noqa: E402
from flood_uav_video_segmentation_amd.flow.model import FlowModel  # noqa: E402
from flood_uav_video_segmentation_amd.flow.predict import FlowEvaluator  # noqa: E402
from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3  # noqa: E402
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data-root", required=True)
    ap.add_argument("--list", required=True, help="test list (Florida video in the reference: meter set 1)")
    ap.add_argument("--list2", help="second test list (Texas video: meter set 2)")
    ap.add_argument("--frame-delta", type=int, default=5)
    ap.add_argument("--arch", choices=("pspnet", "deeplabv3"), default="pspnet")
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--classes-ignore", type=int, nargs="*", default=[5])     # data_classes_ignore (dataset/flow/config.yaml)
    ap.add_argument("--ckpt")
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--feature-based", action="store_true")
    ap.add_argument("--no-warp", action="store_true")
    ap.add_argument("--no-cropping", action="store_true")
    ap.add_argument("--crop", type=int, nargs=2, default=(713, 713), metavar=("H", "W"))
    ap.add_argument("--size", type=int, nargs=2, default=(1072, 1920), metavar=("H", "W"))
    args = ap.parse_args()
    if not args.synthetic_weights and not args.ckpt:
        ap.error("give --ckpt or --synthetic-weights")
    torch.set_grad_enabled(False)

    class HP:
        layers, classes, pretrained = args.layers, args.classes, False

    net = (FlowPSPNet if args.arch == "pspnet" else FlowDeepLabv3)(HP()).eval()
    if args.synthetic_weights:
        net.load_state_dict((synth.make_pspnet_state if args.arch == "pspnet" else synth.make_deeplab_state)(args.layers, args.classes, seed=0))
    else:
        ckpt = torch.load(args.ckpt, map_location="cpu", weights_only=True)
        state = ckpt.get("state_dict", ckpt)
        for prefix in ("model_G.model.", "model.model.", "model."):
            sub = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
            if sub:
                state = sub
                break
        net.load_state_dict(state)
    fm = FlowModel(net, feature_based=args.feature_based, no_warp=args.no_warp).eval()
    ev = FlowEvaluator(fm, classes=args.classes, crop=None if args.no_cropping else tuple(args.crop))
    for idx, lst in enumerate([args.list, args.list2]):
        if not lst:
            continue
        ds = EvalWindows(args.data_root, lst, split="test", frame_delta=args.frame_delta, no_warp=args.no_warp, size=tuple(args.size),
                         classes_ignore=args.classes_ignore)
        for i in range(len(ds)):
            ev.test_step(ds[i], test_idx=idx)
        miou, macc, acc, iou_c, _ = ev.summary(idx)
        print(f"test{idx + 1}: {len(ds)} labelled frames  mIoU {miou:.4f}  mAcc {macc:.4f}  accuracy {acc:.4f}  IoU/class {[round(float(v), 4) for v in iou_c]}")


if __name__ == "__main__":
    main()
