"""Seed-driven synthetic weights, frames and motion grids (numpy PCG64, so the GPU box regenerates
bit-identical tensors).  Follows SURVEY.md 8(d): no real checkpoints, frames or motion vectors ship
with the reference (dataset/flow/README.md:3), so every benchmark and parity input is synthetic.

Weight names are the reference's state_dict keys (canonical, un-aliased):
  PSPNet     model/pspnet.py:52-76, 113-141 + model/resnet.py:60-147
  DeepLabv3  torchvision deeplabv3_resnet101 as wrapped by model/deeplabv3.py:47-54
"""
import math

import numpy as np
import torch

RESNET_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _conv(rng, cout, cin, k, gain=1.0):
    std = gain * math.sqrt(2.0 / (cout * k * k))  # kaiming_normal_(fan_out), model/resnet.py:127
    return torch.from_numpy((rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32))


def _bn(rng, state, prefix, c):
    state[prefix + ".weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))
    state[prefix + ".bias"] = torch.from_numpy((rng.standard_normal(c) * 0.1).astype(np.float32))
    state[prefix + ".running_mean"] = torch.from_numpy((rng.standard_normal(c) * 0.1).astype(np.float32))
    state[prefix + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32))


def _resnet_stages(rng, state, prefix, layers, inplanes):
    for li, nblk in enumerate(RESNET_BLOCKS[layers]):
        planes = 64 * 2 ** li
        for b in range(nblk):
            p = f"{prefix}layer{li + 1}.{b}."
            cin = inplanes if b == 0 else planes * 4
            state[p + "conv1.weight"] = _conv(rng, planes, cin, 1)
            _bn(rng, state, p + "bn1", planes)
            state[p + "conv2.weight"] = _conv(rng, planes, planes, 3)
            _bn(rng, state, p + "bn2", planes)
            # damp the residual branch so that depth does not blow activations up
            state[p + "conv3.weight"] = _conv(rng, planes * 4, planes, 1, gain=0.5)
            _bn(rng, state, p + "bn3", planes * 4)
            if b == 0:
                state[p + "downsample.0.weight"] = _conv(rng, planes * 4, cin, 1)
                _bn(rng, state, p + "downsample.1", planes * 4)
        inplanes = planes * 4


def make_pspnet_state(layers=50, classes=5, seed=0):
    """State dict of FlowPSPNet (canonical keys: layer0..4, ppm, decoder)."""
    rng = _rng(seed)
    s = {}
    s["layer0.0.weight"] = _conv(rng, 64, 3, 3)
    _bn(rng, s, "layer0.1", 64)
    s["layer0.3.weight"] = _conv(rng, 64, 64, 3)
    _bn(rng, s, "layer0.4", 64)
    s["layer0.6.weight"] = _conv(rng, 128, 64, 3)
    _bn(rng, s, "layer0.7", 128)
    _resnet_stages(rng, s, "", layers, 128)
    for i in range(4):
        s[f"ppm.features.{i}.1.weight"] = _conv(rng, 512, 2048, 1)
        _bn(rng, s, f"ppm.features.{i}.2", 512)
    s["decoder.0.weight"] = _conv(rng, 512, 4096, 3)
    _bn(rng, s, "decoder.1", 512)
    s["decoder.4.weight"] = torch.from_numpy((rng.standard_normal((classes, 512, 1, 1)) * 0.05).astype(np.float32))
    s["decoder.4.bias"] = torch.from_numpy((rng.standard_normal(classes) * 0.01).astype(np.float32))
    if (layers, classes, seed) in _LOGIT_CENTRES:
        s["decoder.4.bias"] = -torch.tensor(_LOGIT_CENTRES[(layers, classes, seed)], dtype=torch.float32)
    return s


# Random deep nets emit logits with a large spatially-constant per-class offset, which makes every
# synthetic mask a single class.  These constants (per-class mean logit of the un-biased net on clip
# seed 1000, frame 0) centre the logits so that all classes appear in the synthetic masks.
_LOGIT_CENTRES = {
    (50, 5, 0): [5.79, 14.89, -5.38, 23.80, -22.28],
}


def make_deeplab_state(layers=101, classes=5, seed=0):
    """State dict of FlowDeepLabv3 (canonical keys: backbone.*, classifier.*)."""
    rng = _rng(seed + 7919)
    s = {}
    s["backbone.conv1.weight"] = _conv(rng, 64, 3, 7)
    _bn(rng, s, "backbone.bn1", 64)
    _resnet_stages(rng, s, "backbone.", layers, 64)
    s["classifier.0.convs.0.0.weight"] = _conv(rng, 256, 2048, 1)
    _bn(rng, s, "classifier.0.convs.0.1", 256)
    for i in (1, 2, 3):
        s[f"classifier.0.convs.{i}.0.weight"] = _conv(rng, 256, 2048, 3)
        _bn(rng, s, f"classifier.0.convs.{i}.1", 256)
    s["classifier.0.convs.4.1.weight"] = _conv(rng, 256, 2048, 1)
    _bn(rng, s, "classifier.0.convs.4.2", 256)
    s["classifier.0.project.0.weight"] = _conv(rng, 256, 1280, 1)
    _bn(rng, s, "classifier.0.project.1", 256)
    s["classifier.1.weight"] = _conv(rng, 256, 256, 3)
    _bn(rng, s, "classifier.2", 256)
    s["classifier.4.weight"] = torch.from_numpy((rng.standard_normal((classes, 256, 1, 1)) * 0.05).astype(np.float32))
    s["classifier.4.bias"] = torch.from_numpy((rng.standard_normal(classes) * 0.01).astype(np.float32))
    return s


def make_vit_state(classes=5, image_size=704, patch=32, d_model=768, n_layers=12, dec_layers=2, seed=0):
    """State dict of VITSegmentModel's Segmenter (keys encoder.* / decoder.*; segm/model/{vit,decoder}.py).
    Larger-than-init scales (0.02 trunc-normal in the reference) so that attention is not near-uniform."""
    rng = _rng(seed + 104729)
    s = {}
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    d, g0 = d_model, image_size // patch

    def lin(prefix, out_f, in_f, std):
        s[prefix + ".weight"] = f32(rng.standard_normal((out_f, in_f)) * std)
        s[prefix + ".bias"] = f32(rng.standard_normal(out_f) * 0.02)

    def norm(prefix, n):
        s[prefix + ".weight"] = f32(rng.uniform(0.8, 1.2, n))
        s[prefix + ".bias"] = f32(rng.standard_normal(n) * 0.05)

    def block(prefix):
        norm(prefix + "norm1", d)
        norm(prefix + "norm2", d)
        lin(prefix + "attn.qkv", 3 * d, d, 2.0 / math.sqrt(d))
        lin(prefix + "attn.proj", d, d, 0.5 / math.sqrt(d))
        lin(prefix + "mlp.fc1", 4 * d, d, 1.0 / math.sqrt(d))
        lin(prefix + "mlp.fc2", d, 4 * d, 0.5 / math.sqrt(4 * d))

    s["encoder.patch_embed.proj.weight"] = f32(rng.standard_normal((d, 3, patch, patch)) / math.sqrt(3 * patch * patch))
    s["encoder.patch_embed.proj.bias"] = f32(rng.standard_normal(d) * 0.02)
    s["encoder.cls_token"] = f32(rng.standard_normal((1, 1, d)) * 0.5)
    s["encoder.pos_embed"] = f32(rng.standard_normal((1, g0 * g0 + 1, d)) * 0.5)
    for i in range(n_layers):
        block(f"encoder.blocks.{i}.")
    norm("encoder.norm", d)
    for i in range(dec_layers):
        block(f"decoder.blocks.{i}.")
    s["decoder.cls_emb"] = f32(rng.standard_normal((1, classes, d)) * 0.5)
    lin("decoder.proj_dec", d, d, 1.0 / math.sqrt(d))
    s["decoder.proj_patch"] = f32(rng.standard_normal((d, d)) / math.sqrt(d))
    s["decoder.proj_classes"] = f32(rng.standard_normal((d, d)) / math.sqrt(d))
    norm("decoder.decoder_norm", d)
    norm("decoder.mask_norm", classes)
    centres = _VIT_LOGIT_CENTRES.get((classes, image_size, patch, d_model, n_layers, dec_layers, seed))
    if centres is not None:
        s["decoder.mask_norm.bias"] = s["decoder.mask_norm.bias"] - torch.tensor(centres, dtype=torch.float32)
    return s


# As for the conv nets (_LOGIT_CENTRES): a random Segmenter emits class scores with a spatially constant per-class offset larger
# than their spatial variation, so its argmax is one class almost everywhere (ViT-S/16 seed 3: 99.97 % class 4) and a mask
# comparison would pass for a constant output.  These are the per-class mean scores of the un-centred nets on the frames the
# goldens use (a blend of the 704 and the 713 frame's, whose offsets differ: the 713 input is padded and its position table
# resized); subtracting them from the mask LayerNorm's bias (decoder.py:100) gives masks in which every class appears and none
# covers more than 70 % of either frame.
_VIT_LOGIT_CENTRES = {
    (5, 704, 32, 768, 12, 2, 0): [-0.65, 0.35, -0.20, 0.52, 0.09],    # ViT-B/32 as shipped (vit_b32.npz)
    (5, 704, 16, 384, 12, 2, 3): [-0.43, -0.94, 0.00, -0.25, 1.64],   # ViT-S/16 of the parity tests (vit_s16.npz)
    (5, 704, 16, 384, 12, 2, 0): [0.64, -0.13, -0.01, 0.61, -1.39],   # ViT-S/16 of bench.py's configs[3] variant
}


def make_clip(frames, size, seed, shift=(2, 1), only=None):
    """[T,3,S,S] float32 normalised frames: a smooth random field translated by `shift` px/frame + noise.
    only: optional list of frame indices to materialise (same values as in the full clip)."""
    h = w = size
    if isinstance(size, (tuple, list)):
        h, w = size
    rng = _rng(seed)
    comps = []
    for c in range(3):
        for _ in range(6):
            fx, fy = rng.uniform(0.004, 0.05, 2)
            ph = rng.uniform(0, 2 * np.pi, 2)
            comps.append((c, fx, fy, ph[0], ph[1], rng.uniform(0.3, 1.0)))
    keep = list(range(frames)) if only is None else list(only)
    out = np.empty((len(keep), 3, h, w), dtype=np.float32)
    for t in range(frames):
        noise = rng.standard_normal((3, h, w)) * 0.1  # drawn for every frame: frame t depends on (seed, t) only
        if t not in keep:
            continue
        # the field lives in absolute scene coordinates; frame t looks at the window shifted by t*shift
        yy = (np.arange(h, dtype=np.float64) + shift[1] * t)[:, None]
        xx = (np.arange(w, dtype=np.float64) + shift[0] * t)[None, :]
        img = np.zeros((3, h, w))
        for (c, fx, fy, p0, p1, amp) in comps:
            img[c] += amp * np.sin(2 * np.pi * fx * xx + p0) * np.cos(2 * np.pi * fy * yy + p1)
        out[keep.index(t)] = img + noise
    return torch.from_numpy(out)


def identity_grid(hg, wg):
    """Identity sampling grid of block centres for a crop of hg x wg 16-px blocks (flow/model.py:10-21 semantics)."""
    x = ((np.arange(wg) * 16 + 8) / (wg * 16)) * 2 - 1
    y = ((np.arange(hg) * 16 + 8) / (hg * 16)) * 2 - 1
    g = np.zeros((hg, wg, 2))
    g[:, :, 0] = x[None, :]
    g[:, :, 1] = y[:, None]
    return g


def make_grids(n, hg, wg, seed, shift=(2, 1), frame=(713, 713), jitter=0.02):
    """(mvs_left, mvs_right): n-1 forward grids and n-1 backward grids ALREADY in the order FlowData emits
    them (flow/dataset.py:138-146: inverse grids reversed).  Each [1,hg,wg,2] float32."""
    rng = _rng(seed)
    ident = identity_grid(hg, wg)
    d = np.array([shift[0] / frame[1] * 2, shift[1] / frame[0] * 2])
    left = [ident - d + rng.uniform(-jitter, jitter, ident.shape) for _ in range(n - 1)]
    right = [ident + d + rng.uniform(-jitter, jitter, ident.shape) for _ in range(n - 1)]
    right = right[::-1]
    to_t = lambda a: torch.from_numpy(a.astype(np.float32))[None]  # noqa: E731
    return [to_t(a) for a in left], [to_t(a) for a in right]


def dummy_grids(n):
    """no_warp placeholders: n-1 zeros(1) tensors whose count encodes n (flow/dataset.py:198-205)."""
    return [torch.zeros(1, 1) for _ in range(n - 1)], [torch.zeros(1, 1) for _ in range(n - 1)]


def motion_vectors(h, w, n, seed):
    """Seeded H.264 block motion vectors of one h x w frame in mvextractor's row layout (source, w, h, src_x, src_y, dst_x, dst_y,
    motion_x, motion_y, motion_scale; dataset/flow/extract_motion_vectors.py:25-28): sources and destinations reach past every
    frame edge (negative and too large), and with more vectors than blocks many blocks are hit several times."""
    rng = np.random.default_rng(seed)
    src = np.stack([rng.integers(-40, w + 40, n), rng.integers(-40, h + 40, n)], 1)
    dst = src + rng.integers(-48, 49, (n, 2))
    return np.concatenate([np.full((n, 1), -1), np.full((n, 2), 16), src, dst, dst - src, np.full((n, 1), 4)], 1).astype(np.int64)


def transform_frames(h, w, gh, gw, ids, ignore, seed):
    """Raw files of a tiny labelled video as FlowData reads them (flow/dataset.py:172-187): per frame id a uint8 image [h,w,3], a
    forward and an inverse float64 grid [gh,gw,2], and a uint8 label [h,w] holding every class, the ignored ones and 255."""
    out = {}
    ident = identity_grid(gh, gw)
    for f in ids:
        rng = _rng(seed * 1000 + f)
        label = rng.integers(0, max(ignore) + 1, (h, w)).astype(np.uint8)
        label[rng.random((h, w)) < 0.05] = 255
        out[f] = dict(image=rng.integers(0, 256, (h, w, 3)).astype(np.uint8), label=label,
                      grid=ident + rng.uniform(-0.03, 0.03, ident.shape), inv_grid=ident + rng.uniform(-0.03, 0.03, ident.shape))
    return out
