"""Oracle for FlowDeepLabv3.encoder / .decoder (reference model/deeplabv3.py:11-54).

PARITY UNPINNED: the arithmetic lives in torchvision (deeplabv3_resnet101 fetched by torch.hub at tag
v0.10.0; DeepLabHead from the installed torchvision, pinned 0.12.0 in Pipfile.lock) and torchvision is
not installed offline, so this file restates the published architecture (ResNet v1.5 bottlenecks with
replace_stride_with_dilation=[False, True, True]; ASPP rates 12/24/36) and can only be self-checked
(HIP path vs this restatement).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from .pspnet_oracle import BLOCKS, _bn


def _bottleneck(x, s, p, stride, dil, has_ds):
    out = F.relu(_bn(F.conv2d(x, s[p + "conv1.weight"]), s, p + "bn1"))
    out = F.relu(_bn(F.conv2d(out, s[p + "conv2.weight"], None, stride, dil, dil), s, p + "bn2"))
    out = _bn(F.conv2d(out, s[p + "conv3.weight"]), s, p + "bn3")
    res = x
    if has_ds:
        res = _bn(F.conv2d(x, s[p + "downsample.0.weight"], None, stride), s, p + "downsample.1")
    return F.relu(out + res)


def encoder(x, s, layers=101):
    """backbone(x)["out"] (model/deeplabv3.py:36-43, 53)."""
    x = F.relu(_bn(F.conv2d(x, s["backbone.conv1.weight"], None, 2, 3), s, "backbone.bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nblk in enumerate(BLOCKS[layers]):
        L = li + 1
        for b in range(nblk):
            stride, dil = 1, 1
            if L == 2 and b == 0:
                stride = 2
            if L == 3:
                dil = 1 if b == 0 else 2
            if L == 4:
                dil = 2 if b == 0 else 4
            x = _bottleneck(x, s, f"backbone.layer{L}.{b}.", stride, dil, b == 0)
    return x


def decoder(f, s):
    """DeepLabHead(2048, K) (model/deeplabv3.py:18, 54)."""
    size = f.shape[2:]
    br = [F.relu(_bn(F.conv2d(f, s["classifier.0.convs.0.0.weight"]), s, "classifier.0.convs.0.1"))]
    for i, r in ((1, 12), (2, 24), (3, 36)):
        br.append(F.relu(_bn(F.conv2d(f, s[f"classifier.0.convs.{i}.0.weight"], None, 1, r, r), s, f"classifier.0.convs.{i}.1")))
    g = F.adaptive_avg_pool2d(f, 1)
    g = F.relu(_bn(F.conv2d(g, s["classifier.0.convs.4.1.weight"]), s, "classifier.0.convs.4.2"))
    br.append(F.interpolate(g, size=size, mode="bilinear", align_corners=False))
    y = torch.cat(br, 1)
    y = F.relu(_bn(F.conv2d(y, s["classifier.0.project.0.weight"]), s, "classifier.0.project.1"))  # Dropout(0.5): eval identity
    y = F.relu(_bn(F.conv2d(y, s["classifier.1.weight"], None, 1, 1), s, "classifier.2"))
    return F.conv2d(y, s["classifier.4.weight"], s["classifier.4.bias"])
