"""VITSegmentModel (Segmenter) on the HIP path -- mirrors the reference's model/vit.py:13-56.

    net = VITSegmentModel(num_classes, image_size)      # ViT-B/32 as the reference hard-codes it
    net.load_state_dict(ckpt); net(x) -> {"pred": [B,K,H,W]}

forward = segm/model/segmenter.py:32-48: zero-pad to a multiple of the patch size, ViT encoder
(patch-embed GEMM, pre-LN blocks with fp32-MFMA attention, final LayerNorm), drop the cls token,
MaskTransformer decoder, bilinear (align_corners=False) to the padded size, crop.

`patch_size`, `d_model`, `n_layers` are parameters here (the reference fixes 32 / 768 / 12, model/vit.py:17-38;
BASELINE's "ViT-S/16" is patch 16, d_model 384).  `.encoder` / `.decoder` expose the token map as a
[B,D,gh,gw] feature map so that FlowModel's feature propagation can run on it -- an EXTENSION: the reference
never wires ViT into the flow path (flow/base.py:94-103), so that combination is parity-unpinned.
"""
from torch import nn

from .. import _lib, ops
from .hipnet import HipNet, HipSegNet, HipStage, check_hip_options


class VITSegmentModel(HipSegNet):
    ARCH = _lib.ARCH_SEGMENTER

    def __init__(self, num_classes, image_size, dropout=0.1, patch_size=32, d_model=768, n_layers=12, dec_layers=2, *args, **kwargs):
        nn.Module.__init__(self)
        self.patch_size = patch_size
        self.image_size = image_size
        self.num_classes = num_classes
        self.dropout = dropout  # identity in eval
        self.d_model = d_model
        # library A/B options as keywords (include/floodseg.h): hip_no_split_bf16=True -- Linears and attention on the fp32-MFMA kernels;
        # hip_no_fused_qkv=True -- the attention's K / V^T planes by a separate pre-pass instead of the qkv Linear's epilogue
        check_hip_options(kwargs)
        self._hip_net = HipNet(self.ARCH, 0, num_classes, patch_size, d_model, n_layers, dec_layers, image_size,
                               no_split_bf16=bool(kwargs.get("hip_no_split_bf16", False)), no_fused_qkv=bool(kwargs.get("hip_no_fused_qkv", False)))
        self.encoder = HipStage(self._hip_net.encode, "encoder")  # [B,3,H,W] -> [B, D, gh, gw], stored as [B, gh*gw, D]
        self.decoder = HipStage(self._decode_map, "decoder")

    @staticmethod
    def canonical_name(key):
        if key.startswith("model."):
            key = key[len("model."):]
        if key.startswith("encoder.head.") or key.startswith("encoder.pre_logits"):
            return None  # classification head, unused by Segmenter.forward
        if key.startswith(("encoder.", "decoder.")):
            return key
        return None

    # -- FlowModel-facing callables (feature map view of the tokens).  Stateless: the decoder returns the masks at the PADDED
    # frame size (gh*P x gw*P, segm/model/segmenter.py:45) and `fit_output` removes the padding for the frame size the
    # caller knows (segmenter.py:46 unpadding) -- no per-object record of "the last frame seen", so two sizes / two streams
    # can interleave.
    def _decode_map(self, f):
        masks = self._hip_net.decode(f)  # [B, K, gh, gw]
        gh, gw = masks.shape[2], masks.shape[3]
        return ops.resize_bilinear(masks, (gh * self.patch_size, gw * self.patch_size), align_corners=False)

    def fit_output(self, out, h, w):
        """Decoder output -> frame size: crop the right/bottom zero padding (segm/model/utils.py:79-89)."""
        return out[:, :, :h, :w]

    def decode_fit(self, f, h, w, with_mask=False):
        """fit_output(decoder(f), h, w) as ONE launch after the mask transformer: the upsample computes only the pixels the
        unpadding keeps and writes them dense (bit-identical to the two steps; no strided view for the caller to copy).
        Returns (logits [B,K,h,w], uint8 argmax [B,h,w] or None)."""
        masks = self._hip_net.decode(f)  # [B, K, gh, gw]
        gh, gw = masks.shape[2], masks.shape[3]
        return ops.resize_crop(masks, (gh * self.patch_size, gw * self.patch_size), (h, w), align_corners=False,
                               want_logits=True, want_mask=with_mask)

    def forward(self, x):
        if self.training:
            raise NotImplementedError("VITSegmentModel(HIP) is an inference path; call .eval()")
        h, w = x.shape[2], x.shape[3]
        return {"pred": self.decode_fit(self._hip_net.encode(x), h, w)[0]}
