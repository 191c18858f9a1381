"""Oracle for VITSegmentModel.forward (reference model/vit.py:13-56 -> segm/model/segmenter.py:32-48,
segm/model/vit.py:108-137, blocks.py:16-95, decoder.py:80-102, utils.py:22-40,65-89).

Functional torch-CPU restatement over a flat state dict with the reference's keys minus the leading
"model." (encoder.*, decoder.*).  Pinned by tests/golden/vit_b32.npz (the reference itself, run with a
stub for the three timm symbols it imports -- timm supplies only initialisers and an identity DropPath).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F


def _ln(x, s, p):
    return F.layer_norm(x, (x.shape[-1],), s[p + ".weight"], s[p + ".bias"], 1e-5)


def _lin(x, s, p):
    return F.linear(x, s[p + ".weight"], s[p + ".bias"])


def _block(x, s, p, heads):
    """blocks.py:89-95 (pre-LN), attention blocks.py:56-77, MLP :29-36 (GELU erf)."""
    B, N, C = x.shape
    y = _ln(x, s, p + "norm1")
    qkv = _lin(y, s, p + "attn.qkv").reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + _lin(y, s, p + "attn.proj")
    y = _ln(x, s, p + "norm2")
    return x + _lin(F.gelu(_lin(y, s, p + "mlp.fc1")), s, p + "mlp.fc2")


def resize_pos_embed(posemb, grid_old, grid_new):
    """utils.py:22-40 with one extra (cls) token."""
    tok, grid = posemb[:, :1], posemb[0, 1:]
    grid = grid.reshape(1, grid_old[0], grid_old[1], -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=grid_new, mode="bilinear")
    grid = grid.permute(0, 2, 3, 1).reshape(1, grid_new[0] * grid_new[1], -1)
    return torch.cat([tok, grid], dim=1)


def encoder_tokens(im, s, patch, n_layers, image_size):
    """VisionTransformer.forward(return_features=True) (vit.py:108-137) on an already padded image."""
    B, _, H, W = im.shape
    d = s["encoder.cls_token"].shape[-1]
    x = F.conv2d(im, s["encoder.patch_embed.proj.weight"], s["encoder.patch_embed.proj.bias"], stride=patch).flatten(2).transpose(1, 2)
    x = torch.cat((s["encoder.cls_token"].expand(B, -1, -1), x), dim=1)
    pos = s["encoder.pos_embed"]
    if x.shape[1] != pos.shape[1]:
        g0 = image_size // patch
        pos = resize_pos_embed(pos, (g0, g0), (H // patch, W // patch))
    x = x + pos
    for i in range(n_layers):
        x = _block(x, s, f"encoder.blocks.{i}.", d // 64)
    return _ln(x, s, "encoder.norm")


def mask_decoder(x, s, gs_h, dec_layers, n_cls):
    """MaskTransformer.forward (decoder.py:80-102); x = patch tokens [B, N, D]."""
    d = x.shape[-1]
    x = _lin(x, s, "decoder.proj_dec")
    x = torch.cat((x, s["decoder.cls_emb"].expand(x.size(0), -1, -1)), 1)
    for i in range(dec_layers):
        x = _block(x, s, f"decoder.blocks.{i}.", d // 64)
    x = _ln(x, s, "decoder.decoder_norm")
    patches, cls_feat = x[:, :-n_cls] @ s["decoder.proj_patch"], x[:, -n_cls:] @ s["decoder.proj_classes"]
    patches = patches / patches.norm(dim=-1, keepdim=True)
    cls_feat = cls_feat / cls_feat.norm(dim=-1, keepdim=True)
    masks = _ln(patches @ cls_feat.transpose(1, 2), s, "decoder.mask_norm")
    B, N, K = masks.shape
    return masks.reshape(B, gs_h, N // gs_h, K).permute(0, 3, 1, 2)


def forward(im, s, patch=32, n_layers=12, dec_layers=2, image_size=704, n_cls=5):
    """Segmenter.forward (segmenter.py:32-48) -> {"pred": [B,K,H,W]} as VITSegmentModel.forward wraps it."""
    H0, W0 = im.shape[2], im.shape[3]
    ph, pw = (-H0) % patch, (-W0) % patch
    if ph or pw:
        im = F.pad(im, (0, pw, 0, ph), value=0)  # utils.py:65-76
    H, W = im.shape[2], im.shape[3]
    tokens = encoder_tokens(im, s, patch, n_layers, image_size)[:, 1:]
    masks = mask_decoder(tokens, s, H // patch, dec_layers, n_cls)
    masks = F.interpolate(masks, size=(H, W), mode="bilinear")
    return {"pred": masks[:, :, :H0, :W0]}  # utils.py:79-89
