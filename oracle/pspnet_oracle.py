"""Oracle for FlowPSPNet.encoder / .decoder (reference model/pspnet.py:16-141, model/resnet.py:60-165).

Functional torch-CPU restatement driven by a flat state dict with the reference's canonical keys;
BatchNorm is evaluated un-folded (F.batch_norm, eval), exactly the op sequence the reference runs.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
BINS = (1, 2, 3, 6)


def _bn(x, s, p):
    return F.batch_norm(x, s[p + ".running_mean"], s[p + ".running_var"], s[p + ".weight"], s[p + ".bias"], False, 0.0, 1e-5)


def _bottleneck(x, s, p, stride, dil, has_ds, ds_stride):
    """model/resnet.py:76-96 with the PSPNet dilation patch (model/pspnet.py:55-64)."""
    out = F.relu(_bn(F.conv2d(x, s[p + "conv1.weight"]), s, p + "bn1"))
    out = F.relu(_bn(F.conv2d(out, s[p + "conv2.weight"], None, stride, dil, dil), s, p + "bn2"))
    out = _bn(F.conv2d(out, s[p + "conv3.weight"]), s, p + "bn3")
    res = x
    if has_ds:
        res = _bn(F.conv2d(x, s[p + "downsample.0.weight"], None, ds_stride), s, p + "downsample.1")
    return F.relu(out + res)


def backbone(x, s, layers=50, taps=None):
    """layer0..layer4 (model/pspnet.py:52-64, model/resnet.py:109-121)."""
    x = F.relu(_bn(F.conv2d(x, s["layer0.0.weight"], None, 2, 1), s, "layer0.1"))
    x = F.relu(_bn(F.conv2d(x, s["layer0.3.weight"], None, 1, 1), s, "layer0.4"))
    x = F.relu(_bn(F.conv2d(x, s["layer0.6.weight"], None, 1, 1), s, "layer0.7"))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["layer0"] = x
    for li, nblk in enumerate(BLOCKS[layers]):
        L = li + 1
        for b in range(nblk):
            stride, dil = 1, 1
            if L == 2 and b == 0:
                stride = 2
            if L == 3:
                dil = 2
            if L == 4:
                dil = 4
            x = _bottleneck(x, s, f"layer{L}.{b}.", stride, dil, b == 0, stride)
        if taps is not None:
            taps[f"layer{L}"] = x
    return x


def ppm(x, s):
    """model/pspnet.py:29-34."""
    out = [x]
    for i, b in enumerate(BINS):
        y = F.adaptive_avg_pool2d(x, b)
        y = F.relu(_bn(F.conv2d(y, s[f"ppm.features.{i}.1.weight"]), s, f"ppm.features.{i}.2"))
        out.append(F.interpolate(y, x.shape[2:], mode="bilinear", align_corners=True))
    return torch.cat(out, 1)


def encoder(x, s, layers=50, taps=None):
    """FlowPSPNet.encoder = Sequential(layers, ppm) (model/pspnet.py:136-139)."""
    return ppm(backbone(x, s, layers, taps), s)


def decoder(f, s):
    """FlowPSPNet.decoder = cls (model/pspnet.py:70-76, 141); Dropout2d is identity in eval."""
    y = F.relu(_bn(F.conv2d(f, s["decoder.0.weight"], None, 1, 1), s, "decoder.1"))
    return F.conv2d(y, s["decoder.4.weight"], s["decoder.4.bias"])


def single_frame(x, s, layers=50):
    """PSPNet.forward in eval (model/pspnet.py:87-109): config-1 supervised single-frame inference."""
    h, w = x.shape[2], x.shape[3]
    assert (h - 1) % 8 == 0 and (w - 1) % 8 == 0
    y = decoder(encoder(x, s, layers), s)
    return {"pred": F.interpolate(y, size=(h, w), mode="bilinear", align_corners=True)}
