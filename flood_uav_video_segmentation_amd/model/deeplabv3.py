"""FlowDeepLabv3 on the HIP path -- mirrors the reference's model/deeplabv3.py:47-54 interface.

encoder = torchvision ResNet backbone ["out"] (output stride 8), decoder = DeepLabHead(2048, K).
PARITY UNPINNED: torchvision is not vendored by the reference and absent offline, so the architecture
is restated from its public definition and checked only against oracle/deeplab_oracle.py.
"""
from .. import _lib, ops
from .hipnet import HipSegNet


class FlowDeepLabv3(HipSegNet):
    ARCH = _lib.ARCH_DEEPLABV3

    def __init__(self, hparams, *args, **kwargs):
        super().__init__(hparams)
        if getattr(hparams, "pretrained", False):
            # the reference fetches pytorch/vision:v0.10.0 through torch.hub here (model/deeplabv3.py:15)
            raise RuntimeError("FlowDeepLabv3(HIP): pretrained=True is not supported, load a state_dict instead")

    def segment(self, *frames):
        """decoder(encoder(x)) in one library call (encoder + decoder over a library-owned feature map); FlowModel's segmentation-mode paths
        (flow/model.py:39-40, 189-191, 202-204) use it when the wrapped network offers it."""
        return self._hip_net.segment(*frames)

    def segment_crops(self, frame_a, frame_b, crop_yx, crop_hw):
        """The same composition on crop windows of full frames, read in place: the sliding-crop route (flow/base.py:182-209)
        batches its crops through the network with this."""
        return self._hip_net.segment_crops(frame_a, frame_b, crop_yx, crop_hw)

    @staticmethod
    def canonical_name(key):
        if key.endswith("num_batches_tracked"):
            return None
        if key.startswith("encoder.model."):
            return "backbone." + key[len("encoder.model."):]
        if key.startswith("decoder."):
            return "classifier." + key[len("decoder."):]
        if key.startswith(("backbone.", "classifier.")):
            return key
        return None


class DeepLabv3(FlowDeepLabv3):
    """Single-frame DeepLabv3 (reference model/deeplabv3.py:11-33): `forward(x) -> {"pred": out}`, where torchvision's
    segmentation wrapper brings the classifier output back to the input size with bilinear interpolation,
    align_corners=False.  Checkpoint keys `model.backbone.*` / `model.classifier.*`; `model.aux_classifier.*` is
    training-only and ignored.  PARITY UNPINNED like FlowDeepLabv3 (torchvision absent)."""

    @staticmethod
    def canonical_name(key):
        if key.startswith("model.aux_classifier."):
            return None
        if key.startswith("model."):
            key = key[len("model."):]
        return FlowDeepLabv3.canonical_name(key)

    def forward(self, x):
        if self.training:
            raise NotImplementedError("DeepLabv3(HIP) is an inference path; call .eval() (training adds the aux output: model/deeplabv3.py:27-31)")
        out = self.segment(x)
        return {"pred": ops.resize_bilinear(out, (x.shape[2], x.shape[3]), align_corners=False)}
