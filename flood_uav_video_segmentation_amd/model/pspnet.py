"""FlowPSPNet on the HIP path -- mirrors the reference's model/pspnet.py:113-141 interface.

    net = FlowPSPNet(hparams)            # hparams.layers / .pretrained / .classes
    net.load_state_dict(ckpt)            # reference keys, any of their aliases
    f = net.encoder(x); o = net.decoder(f)

`encoder` = dilated ResNet (model/resnet.py:99-147 with the dilation patch of model/pspnet.py:55-64)
+ pyramid pooling (model/pspnet.py:16-34); `decoder` = cls head (model/pspnet.py:70-76).  Inference only:
the aux head, PSPNetSemi and the training-mode branches are out of scope.
"""
import re

from .. import _lib, ops
from .hipnet import HipSegNet, HipStage


class FlowPSPNet(HipSegNet):
    ARCH = _lib.ARCH_PSPNET

    def __init__(self, hparams, *args, **kwargs):
        super().__init__(hparams)
        if getattr(hparams, "pretrained", False):
            # the reference reads ./initmodel/resnet*_v2.pth here (model/resnet.py:201); weights for the
            # HIP path always arrive through load_state_dict
            raise RuntimeError("FlowPSPNet(HIP): pretrained=True is not supported, load a state_dict instead")
        # attribute names the reference exposes (flow/base.py:96-98 builds optimiser groups from them)
        self.layers = HipStage(None, "layers")
        self.ppm = HipStage(None, "ppm")

    _ALIAS = (
        (re.compile(r"^layers\.(\d)\.(.*)$"), r"layer\1.\2"),
        (re.compile(r"^encoder\.0\.(\d)\.(.*)$"), r"layer\1.\2"),
        (re.compile(r"^encoder\.1\.(.*)$"), r"ppm.\1"),
    )

    def segment(self, *frames):
        """decoder(encoder(x)) in one library call (the fused head never builds the 4096-channel concat); FlowModel's segmentation-mode paths
        (flow/model.py:39-40, 189-191, 202-204) use it when the wrapped network offers it."""
        return self._hip_net.segment(*frames)

    def segment_crops(self, frame_a, frame_b, crop_yx, crop_hw):
        """The same composition on crop windows of full frames, read in place: the sliding-crop route (flow/base.py:182-209)
        batches its crops through the network with this."""
        return self._hip_net.segment_crops(frame_a, frame_b, crop_yx, crop_hw)

    @staticmethod
    def canonical_name(key):
        """Map any alias FlowPSPNet's state_dict holds (1046 keys, 362 tensors) to
        its canonical name; None for keys the inference path does not need."""
        if key.endswith("num_batches_tracked"):
            return None
        for rx, rep in FlowPSPNet._ALIAS:
            if rx.match(key):
                key = rx.sub(rep, key)
                break
        if re.match(r"^(layer[0-4]|ppm|decoder)\.", key):
            return key
        return None


class PSPNet(FlowPSPNet):
    """Single-frame PSPNet (reference model/pspnet.py:38-110; BASELINE configs[0]): `forward(x) -> {"pred": logits}` with
    the logits brought back to the input size (zoom_factor 8: h = (H-1)/8*8+1 = H, :88-90, :99-100).  Its state_dict names
    the head `cls.*` (the FlowPSPNet alias is `decoder.*`); the training-only `aux.*` head is ignored."""

    zoom_factor = 8

    @staticmethod
    def canonical_name(key):
        if key.startswith("aux."):
            return None
        if key.startswith("cls."):
            key = "decoder." + key[len("cls."):]
        return FlowPSPNet.canonical_name(key)

    def forward(self, x):
        if self.training:
            raise NotImplementedError("PSPNet(HIP) is an inference path; call .eval() (training adds the aux head: model/pspnet.py:102-106)")
        hh, ww = x.shape[2], x.shape[3]
        assert (hh - 1) % 8 == 0 and (ww - 1) % 8 == 0  # model/pspnet.py:88
        h = int((hh - 1) / 8 * self.zoom_factor + 1)
        w = int((ww - 1) / 8 * self.zoom_factor + 1)
        out = self.segment(x)
        if self.zoom_factor != 1:
            out = ops.resize_bilinear(out, (h, w), align_corners=True)
        return {"pred": out}
