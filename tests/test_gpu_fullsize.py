"""Full-size GPU parity: every BASELINE.json config at its own geometry (713x713 / 704x704 / 1072x1920), against the CPU oracle
run on the same seeded inputs and, where the reference itself could be run (PSPNet), against its committed outputs.

  configs[2]  DeepLabv3-ResNet101, key frames + optical-flow warp of the logits, n = 5, 44x44 grids   (PARITY UNPINNED: torchvision)
  configs[3]  Segmenter ViT-S/16 (12 layers, d = 384, 6 heads) per frame @704 / @713 and key frames + feature propagation
  A2          FlowPSPNet feature-based propagation at 713x713 (C = 4096 NHWC warps 90x90 -> 44x44)
  8(f)-1      the reference's default real-video route: 8 overlapping 713x713 crops of a 1072x1920 frame, batched
  key-frame cache (one new key frame per window), A/B options, ModelRepresentation on the GPU

Tolerances are 3-5x the errors measured on MI355X (gpurun_out/parity_measured.txt, quoted in DESIGN.md section 5)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import memo_by_content, load_golden, note, rel_err
from flood_uav_video_segmentation_amd import ops, synth
from flood_uav_video_segmentation_amd.flow import crops
from flood_uav_video_segmentation_amd.flow.model import FlowModel, KeyframeCache
from flood_uav_video_segmentation_amd.flow.predict import FlowPredictor
from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet, PSPNet
from flood_uav_video_segmentation_amd.model.vit import VITSegmentModel
from flood_uav_video_segmentation_amd.model.wrapper import ModelRepresentation
from oracle import crops_oracle, deeplab_oracle, flow_oracle, pspnet_oracle, vit_oracle

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

LOGIT_TOL = 3e-5   # conv networks, fp32 with Winograd F(6,3): measured 4-8e-6 of max|logit| (worst: 8.0e-6, head over the concat)
MASK_MIN = 0.9999  # per-pixel argmax agreement: measured >= 0.999986
VIT_TOL = 1.5e-4   # Segmenter logits = LayerNorm over K = 5 cosine similarities: amplifies the fp32 noise of 14 blocks (measured 3.4e-5)
N = 5


class HP:
    def __init__(self, layers=50, classes=5, **opts):
        self.layers, self.classes, self.pretrained = layers, classes, False
        for k, v in opts.items():
            setattr(self, k, v)


def cu(ts):
    return [t.cuda() for t in ts]


def memo(fn):
    """The oracle's per-frame network call, memoised on the input's storage: a 713x713 forward costs seconds on the CPU."""
    cache = {}

    def wrapped(x):
        key = (x.data_ptr(), tuple(x.shape))
        if key not in cache:
            cache[key] = fn(x)
        return cache[key]
    return wrapped


_CROP_SEG = {}


def _crop_oracle_seg(state):
    """pspnet_oracle's decoder(encoder(.)) for `state`, memoised by input content across the tests of this module."""
    key = id(state)
    if key not in _CROP_SEG:
        _CROP_SEG[key] = (state, memo_by_content(lambda x: pspnet_oracle.decoder(pspnet_oracle.encoder(x, state, 50), state)))
    return _CROP_SEG[key][1]


@pytest.fixture(scope="module")
def psp():
    state = synth.make_pspnet_state(50, 5, seed=0)
    net = FlowPSPNet(HP()).eval()
    net.load_state_dict(state)
    return net, state


# ------------------------------------------------------------------------------------------------ configs[2]
def test_config2_deeplabv3_r101_warp_713_against_oracle_parity_unpinned():
    """BASELINE configs[2] at its own size: FlowModel(FlowDeepLabv3(R101), feature_based=False, no_warp=False).predict, n = 5,
    44x44 grids.  At 713x713 the 90x90 feature map sends ALL THREE dilated ASPP convs (12/24/36: lattices 8/4/3 pixels wide)
    to the Winograd lattice path.  Reference: model/deeplabv3.py:47-54 + flow/model.py:184-241.  PARITY UNPINNED (torchvision
    absent): the oracle restates the public architecture."""
    state = synth.make_deeplab_state(101, 5, seed=0)
    net = FlowDeepLabv3(HP(101)).eval()
    net.load_state_dict(state)
    clip = synth.make_clip(6, 713, seed=1000)
    prev, nxt = clip[0:1], clip[5:6]
    mvl, mvr = synth.make_grids(N, 44, 44, seed=2000)
    fm = FlowModel(net, feature_based=False, no_warp=False).eval()
    got = fm.predict(prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), N, None)["pred"]
    assert got.shape == (N, 5, 713, 713)
    enc = lambda x: x  # noqa: E731  (encoder and decoder are composed in `dec`, memoised per key frame)
    dec = memo(lambda x: deeplab_oracle.decoder(deeplab_oracle.encoder(x, state, 101), state))
    ref = flow_oracle.predict_segmentation(enc, dec, prev, nxt, mvl, mvr, N, False)["pred"]
    lo = net.segment(prev.cuda(), nxt.cuda()).cpu()
    assert lo.shape == (2, 5, 90, 90)
    assert note("cfg2_deeplab_r101_713_lowres_logits", rel_err(lo, torch.cat([dec(prev), dec(nxt)], 0))) < LOGIT_TOL
    assert note("cfg2_deeplab_r101_713_pred_logits", rel_err(got.cpu(), ref)) < LOGIT_TOL
    agree = (ops.argmax_u8(got).cpu() == ref.max(1)[1].to(torch.uint8)).float().mean().item()
    assert note("cfg2_deeplab_r101_713_mask_disagreement", 1 - agree) < 1 - MASK_MIN
    hist = ops.iou_hist(ops.argmax_u8(got), ref.max(1)[1].to(torch.uint8).cuda(), 5).cpu().numpy().astype(np.float64)
    miou = np.mean(hist[0] / (hist[1] + hist[2] - hist[0] + 1e-10))
    assert note("cfg2_deeplab_r101_713_miou_delta_pp", (1 - miou) * 100) < 0.1  # north star: mIoU within 0.1 pp


@pytest.mark.parametrize("size", [97, 713])
def test_config2_deeplabv3_r101_encoder_against_the_references_own_resnet(size):
    """A7, the part that IS pinned: FlowDeepLabv3(R101).encoder on the HIP path against the reference's own model/resnet.py
    classes assembled as the torchvision backbone (7x7 stem, [3, 4, 23, 3] v1.5 bottlenecks, dilation schedule [False, True, True];
    tests/golden/deeplab_backbone.npz, gen_goldens.py::gen_deeplab_backbone) -- about 70 % of configs[2]'s FLOPs.  The ASPP head
    (torchvision DeepLabHead, model/deeplabv3.py:18) stays HIP <-> oracle only: parity unpinned for that part."""
    z = load_golden("deeplab_backbone.npz")
    net = FlowDeepLabv3(HP(101)).eval()
    net.load_state_dict(synth.make_deeplab_state(101, 5, seed=0))
    if size == 97:
        f = net.encoder(synth.make_clip(2, 97, seed=150).cuda()).cpu()
        assert f.shape == (2, 2048, 13, 13)
        assert note("cfg2_deeplab_r101_encoder_97_vs_reference_resnet", rel_err(f[:, ::4], z["feat97_sub"])) < LOGIT_TOL
    else:
        f = net.encoder(synth.make_clip(6, 713, seed=1000)[0:1].cuda()).cpu()
        assert f.shape == (1, 2048, 90, 90)
        assert note("cfg2_deeplab_r101_encoder_713_vs_reference_resnet", rel_err(f[:, ::32, ::3, ::3], z["feat713_sub"])) < LOGIT_TOL
    st = z[f"stat{size}_layer4"]
    assert abs(f.double().abs().mean().item() - st[1]) < 1e-4 * st[1]


# ------------------------------------------------------------------------------------------------ configs[3]
S16 = dict(patch=16, d_model=384, n_layers=12, dec_layers=2, image_size=704)


@pytest.fixture(scope="module")
def vit_s16():
    state = synth.make_vit_state(5, 704, 16, 384, 12, 2, seed=3)
    net = VITSegmentModel(5, 704, patch_size=16, d_model=384, n_layers=12, dec_layers=2).eval()
    net.load_state_dict(state)
    return net, state


@pytest.mark.parametrize("size", [704, 713])
def test_config3_vit_s16_per_frame_against_oracle(vit_s16, size):
    """12-layer ViT-S/16 (d = 384, 6 heads) at 704 (44x44 = 1936 patches + cls: the 4-way key split + attention_combine) and
    at 713 (padded to 720: 45x45 = 2025 patches, resized position embedding, unpadding).  model/vit.py:13-56 with S/16 dims,
    segm/model/blocks.py:56-77."""
    net, state = vit_s16
    x = synth.make_clip(2, size, seed=310 + size)
    got = net(x.cuda())["pred"]
    ref = vit_oracle.forward(x, state, 16, 12, 2, 704, 5)["pred"]
    assert got.shape == ref.shape == (2, 5, size, size)
    pad = (-size) % 16
    tok = net.encoder(x.cuda())
    g = (size + pad) // 16
    ref_tok = vit_oracle.encoder_tokens(F.pad(x, (0, pad, 0, pad)), state, 16, 12, 704)[:, 1:]
    assert note(f"cfg3_vit_s16_{size}_tokens", rel_err(tok.permute(0, 2, 3, 1).reshape(2, g * g, 384).cpu(), ref_tok)) < 4e-5
    assert note(f"cfg3_vit_s16_{size}_logits", rel_err(got.cpu(), ref)) < VIT_TOL
    agree = (got.max(1)[1].cpu() == ref.max(1)[1]).float().mean().item()
    assert note(f"cfg3_vit_s16_{size}_mask_disagreement", 1 - agree) < 1e-3
    # ... and against the reference's own classes assembled with the S/16 numbers (tests/golden/gen_goldens.py::gen_vit_s16)
    z = load_golden("vit_s16.npz")
    assert note(f"cfg3_vit_s16_{size}_logits_vs_reference", rel_err(got[0:1, :, ::8, ::8].cpu(), z[f"pred{size}_sub"])) < VIT_TOL
    assert (got[0:1].max(1)[1].to(torch.uint8)[:, ::2, ::2].cpu().numpy() == z[f"mask{size}"]).mean() > 0.999


@pytest.mark.parametrize("size", [704, 713])
def test_config3_vit_s16_feature_flow_against_oracle_parity_unpinned(vit_s16, size):
    """BASELINE configs[3]: key-frame ViT-S/16 + feature-based propagation at full size.  The reference has no such path
    (flow/base.py:94-103): OUR definition -- the token map [B,D,gh,gw] through FlowModel.predict_feature (flow/model.py:116-181),
    the Segmenter's own un-padding at the end -- checked against the oracle's predict_feature on the same callables."""
    net, state = vit_s16
    clip = synth.make_clip(6, size, seed=1000)
    prev, nxt = clip[0:1], clip[5:6]
    pad = (-size) % 16
    g = (size + pad) // 16
    mvl, mvr = synth.make_grids(N, 44, 44, seed=2000, frame=(size, size))
    fm = FlowModel(net, feature_based=True, no_warp=False).eval()
    got = fm.predict(prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), N, None)["pred"]

    def enc(x):
        t = vit_oracle.encoder_tokens(F.pad(x, (0, pad, 0, pad)), state, 16, 12, 704)[:, 1:]
        return t.transpose(1, 2).reshape(x.shape[0], 384, g, g)

    def dec(f):
        b, d, gh, gw = f.shape
        m = vit_oracle.mask_decoder(f.reshape(b, d, gh * gw).transpose(1, 2), state, gh, 2, 5)
        return F.interpolate(m, size=(gh * 16, gw * 16), mode="bilinear")[:, :, :size, :size]

    ref = flow_oracle.predict_feature(memo(enc), dec, prev, nxt, mvl, mvr, N, False)["pred"]
    assert got.shape == ref.shape == (N, 5, size, size)
    assert note(f"cfg3_vit_s16_{size}_feature_flow_logits", rel_err(got.cpu(), ref)) < VIT_TOL
    agree = (got.max(1)[1].cpu() == ref.max(1)[1]).float().mean().item()
    assert note(f"cfg3_vit_s16_{size}_feature_flow_mask_disagreement", 1 - agree) < 1e-3


def test_vit_two_frame_sizes_interleaved_do_not_share_state(vit_s16):
    """The frame size travels with the call (fit_output), not with the object: a 704 decode between the encode and the decode
    of a 713 frame must not change the 713 result (round 1 kept `_frame_hw` on the module)."""
    net, _ = vit_s16
    a, b = synth.make_clip(1, 713, seed=5).cuda(), synth.make_clip(1, 704, seed=6).cuda()
    want = net(a)["pred"]
    fa = net.encoder(a)
    net(b)
    got = net.fit_output(net.decoder(fa), 713, 713)
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------ A2 at the BASELINE size
def test_pspnet_feature_based_713_against_oracle(psp):
    """A2 (flow/model.py:116-181) at 713x713: C = 4096 NHWC warps 90x90 -> 44x44, resizes back to 90x90, the key-frame feature
    through the 67x120 identity grid, ONE batched decoder call on [5,4096,90,90].  Against the oracle (all pixels) and against
    the reference's own output of the same call (tests/golden/predict_feature_713.npz, warp and no_warp)."""
    net, state = psp
    clip = synth.make_clip(6, 713, seed=1000)
    prev, nxt = clip[0:1], clip[5:6]
    mvl, mvr = synth.make_grids(N, 44, 44, seed=2000)
    z = load_golden("predict_feature_713.npz")
    dl, dr = synth.dummy_grids(N)
    got_nw = FlowModel(net, feature_based=True, no_warp=True).eval().predict(prev.cuda(), nxt.cuda(), cu(dl), cu(dr), N, None)["pred"]
    assert note("a2_pspnet_feature_713_nowarp_logits_vs_reference", rel_err(got_nw[:, :, ::16, ::16].cpu(), z["nowarp_logits_sub"])) < LOGIT_TOL
    assert (ops.argmax_u8(got_nw)[:, ::2, ::2].cpu().numpy() == z["nowarp_mask_sub"]).mean() > MASK_MIN
    del got_nw
    fm = FlowModel(net, feature_based=True, no_warp=False).eval()
    got = fm.predict(prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), N, None)["pred"]
    assert note("a2_pspnet_feature_713_warp_logits_vs_reference", rel_err(got[:, :, ::16, ::16].cpu(), z["warp_logits_sub"])) < LOGIT_TOL
    fm.fused_feature_tail = False  # fs_feat_tail against the op-by-op route it replaces: bit-identical at the BASELINE size
    assert torch.equal(fm.predict(prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), N, None)["pred"], got)
    fm.fused_feature_tail = True
    assert note("a2_pspnet_feature_713_warp_mask_disagreement_vs_reference",
                1 - (ops.argmax_u8(got)[:, ::2, ::2].cpu().numpy() == z["warp_mask_sub"]).mean()) < 1 - MASK_MIN
    enc = memo(lambda x: pspnet_oracle.encoder(x, state, 50))
    dec = lambda f: pspnet_oracle.decoder(f, state)  # noqa: E731
    ref = flow_oracle.predict_feature(enc, dec, prev, nxt, mvl, mvr, N, False)["pred"]
    assert got.shape == ref.shape == (N, 5, 713, 713)
    assert note("a2_pspnet_feature_713_logits", rel_err(got.cpu(), ref)) < LOGIT_TOL
    agree = (ops.argmax_u8(got).cpu() == ref.max(1)[1].to(torch.uint8)).float().mean().item()
    assert note("a2_pspnet_feature_713_mask_disagreement", 1 - agree) < 1 - MASK_MIN


# ------------------------------------------------------------------------------------------------ A10
def test_model_representation_wraps_a_hip_network(psp):
    """A10 (model/wrapper.py:50-51): in eval the wrapper returns the wrapped network's dict unchanged -- here around the HIP
    single-frame PSPNet, against the reference golden at 713x713."""
    _, state = psp
    renamed = {("cls." + k[len("decoder."):] if k.startswith("decoder.") else k): v for k, v in state.items()}
    inner = PSPNet(HP()).eval()
    inner.load_state_dict(renamed)
    wrapped = ModelRepresentation(inner, rep=None, rep_forward=None).eval()
    x = synth.make_clip(6, 713, seed=1000)[0:1].cuda()
    out = wrapped(x)
    assert set(out) == {"pred"} and torch.equal(out["pred"], inner(x)["pred"])
    z = load_golden("pspnet_713.npz")
    ref = F.interpolate(torch.from_numpy(z["logits_lo"]), (713, 713), mode="bilinear", align_corners=True)
    assert rel_err(out["pred"].cpu(), ref) < LOGIT_TOL
    with pytest.raises(NotImplementedError):
        wrapped.train()(x)


def test_pspnet_r101_713_against_the_reference():
    """A6 on the deeper backbone at the BASELINE frame size: FlowPSPNet(layers=101) -- 23 layer3 blocks on the 90x90 map -- against
    the reference's own output on one 713x713 frame (tests/golden/pspnet101_713.npz): the fused segment route and the
    encoder -> decoder route."""
    state = synth.make_pspnet_state(101, 5, seed=1)
    net = FlowPSPNet(HP(101)).eval()
    net.load_state_dict(state)
    z = load_golden("pspnet101_713.npz")
    x = synth.make_clip(6, 713, seed=1000)[0:1].cuda()
    assert note("pspnet101_713_segment_logits_vs_reference", rel_err(net.segment(x).cpu(), z["logits_lo"])) < LOGIT_TOL
    feat = net.encoder(x)
    assert note("pspnet101_713_encoder_vs_reference", rel_err(feat[:, ::256, ::6, ::6].cpu(), z["feat_slice"])) < LOGIT_TOL
    assert note("pspnet101_713_decoder_logits_vs_reference", rel_err(net.decoder(feat).cpu(), z["logits_lo"])) < LOGIT_TOL


# ------------------------------------------------------------------------------------------------ A/B options at full size
@pytest.mark.parametrize("opts", [dict(hip_no_winograd=True), dict(hip_winograd_tile=3), dict(hip_winograd_tile=4), dict(hip_winograd_tile=6),
                                  dict(hip_no_fused_head=True), dict(hip_no_fused_shortcut=True), dict(hip_no_fused_winograd=True),
                                  dict(hip_no_split_bf16=True),  # the fp32-MFMA kernels (round 2's arithmetic)
                                  dict(hip_no_fused_pool=True),   # round 5 A/B switch: layer0.6 and the max-pool as two launches again
                                  dict(hip_no_res_touch=True),    # round 5 A/B switch: no L2 touch of the shortcut tile before the epilogue
                                  dict(hip_no_split_bf16=True, hip_no_winograd=True),
                                  dict(hip_no_winograd=True, hip_no_fused_head=True, hip_no_fused_shortcut=True)])
def test_every_shipped_option_matches_the_reference_golden_at_713(psp, opts):
    """Each arithmetic-changing route the library ships (direct conv instead of Winograd, F(3,3) / F(4,3) / F(6,3) forced, head over
    the 4096-channel concat instead of the fused pyramid term, projection shortcut + conv3 as two launches instead of one
    concatenated-K GEMM, the small-Cin 3x3 convs on the direct kernel instead of the one-kernel Winograd) against the
    reference's own 713x713 outputs."""
    _, state = psp
    net = FlowPSPNet(HP(**opts)).eval()
    net.load_state_dict(state)
    z = load_golden("pspnet_713.npz")
    zp = load_golden("predict_713.npz")
    clip = synth.make_clip(6, 713, seed=1000)
    prev, nxt = clip[0:1].cuda(), clip[5:6].cuda()
    lo = net.segment(prev)
    tag = "+".join(f"{k}={v}" for k, v in opts.items())
    assert note(f"option[{tag}]_713_logits_lo", rel_err(lo.cpu(), z["logits_lo"])) < LOGIT_TOL
    dl, dr = synth.dummy_grids(N)
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    mask = fm.predict_masks(prev, nxt, cu(dl), cu(dr), N)
    assert note(f"option[{tag}]_713_cfg1_mask_disagreement", 1 - (mask.cpu().numpy() == zp["cfg2_mask"]).mean()) < 1 - MASK_MIN


@pytest.mark.parametrize("size,b", [((713, 713), 2), ((713, 713), 1), ((257, 323), 3), ((129, 161), 2), ((65, 65), 1)])
def test_fused_stem_maxpool_and_residual_touch_are_bit_identical_to_the_plain_routes(psp, size, b):
    """Round 5: layer0.6 + max-pool as one launch (default), and the residual-touch experiment: whole-network outputs BIT-identical
    to the two-launch stem / the untouched epilogue.  (Below 500 tiles per image -- 129 x 161 and 65 x 65 frames -- the stem conv does
    not take the one-kernel Winograd: the pooled form must step aside with it.)"""
    _, state = psp
    nets = [FlowPSPNet(HP(**o)).eval() for o in (dict(), dict(hip_no_fused_pool=True), dict(hip_no_res_touch=True))]
    for n in nets:
        n.load_state_dict(state)
    x = synth.make_clip(b, size, seed=57).cuda()
    outs = [n.segment(x) for n in nets]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    kernels = [r[1] for r in _profile_rows(nets[0], x)]
    h1, w1 = (size[0] - 1) // 2 + 1, (size[1] - 1) // 2 + 1
    fused = -(-h1 // 4) * -(-w1 // 4) >= 500
    assert ("wino_fused_pool" in kernels) == fused and ("maxpool3x3s2" in kernels) == (not fused)
    assert "wino_fused_pool" not in [r[1] for r in _profile_rows(nets[1], x)]
    assert torch.equal(nets[0].segment(x), outs[0])  # run to run


def _profile_rows(net, x):
    net._hip_net.profile(True)
    net.segment(x)
    rows = net._hip_net.profile_dump()
    net._hip_net.profile(False)
    return rows


@pytest.mark.parametrize("size,b", [((713, 713), 2), ((129, 161), 3)])
def test_fused_projection_shortcut_equals_the_two_launch_route(psp, size, b):
    """relu(bn3(conv3(f)) + bn_ds(downsample(x))) (model/resnet.py:86-94) as ONE GEMM over the concatenated K, BatchNorm scales
    folded into the filters, against downsample-then-conv3-with-residual: every projection block (stride 1 and the stride-2
    one of layer2), whole-network outputs within fp32 reassociation noise."""
    _, state = psp
    fused = FlowPSPNet(HP()).eval()
    plain = FlowPSPNet(HP(hip_no_fused_shortcut=True)).eval()
    fused.load_state_dict(state)
    plain.load_state_dict(state)
    x = synth.make_clip(b, size, seed=55).cuda()
    a, c = fused.segment(x), plain.segment(x)
    assert note(f"fused_shortcut_vs_two_launches_{size[0]}", rel_err(a.cpu(), c.cpu())) < 1e-5
    assert (a.max(1)[1] == c.max(1)[1]).float().mean().item() > 0.9998
    assert torch.equal(fused.segment(x), a)  # bit-repeatable
    fa, fc = fused.encoder(x), plain.encoder(x)
    assert rel_err(fa.cpu(), fc.cpu()) < 1e-5


# ------------------------------------------------------------------------------------------------ key-frame cache
@pytest.mark.parametrize("mode", ["segmentation_linear", "segmentation_warp", "feature"])
def test_keyframe_cache_is_bit_identical_over_consecutive_windows(psp, mode):
    """Window i's next key is window i+1's previous key (flow/dataset.py:112-114): with the cache each window segments ONE new
    key frame; masks / logits must be bit-identical to the uncached path over >= 3 consecutive windows at 713x713."""
    net, _ = psp
    keys = synth.make_clip(16, 713, seed=1000, only=[0, 5, 10, 15]).cuda()
    fb, nw = mode == "feature", mode == "segmentation_linear"
    size = 713 if not fb else 321  # feature mode: 4096-channel maps, same code path at a smaller frame
    if fb:
        keys = synth.make_clip(16, size, seed=1000, only=[0, 5, 10, 15]).cuda()
    grids = [synth.dummy_grids(N) if nw else synth.make_grids(N, 44 if not fb else 20, 44 if not fb else 20, seed=2000 + i, frame=(size, size))
             for i in range(3)]
    fm = FlowModel(net, feature_based=fb, no_warp=nw).eval()
    cache = KeyframeCache()
    for i in range(3):
        prev, nxt = keys[i:i + 1], keys[i + 1:i + 2]
        mvl, mvr = cu(grids[i][0]), cu(grids[i][1])
        plain = fm.predict(prev, nxt, mvl, mvr, N, None)["pred"]
        cached = fm.predict(prev, nxt, mvl, mvr, N, None, key_cache=cache.window(5 * i, 5 * i + 5))["pred"]
        assert torch.equal(plain, cached), f"window {i}"
    assert (cache.hits, cache.misses) == (2, 1)
    # a window that does not continue the sequence must miss, not reuse a stale frame
    again = fm.predict(keys[0:1], keys[1:2], cu(grids[0][0]), cu(grids[0][1]), N, None, key_cache=cache.window(0, 5))["pred"]
    assert cache.misses == 2 and torch.equal(again, fm.predict(keys[0:1], keys[1:2], cu(grids[0][0]), cu(grids[0][1]), N, None)["pred"])


def test_keyframe_cache_is_bit_identical_for_the_segmenter(vit_s16):
    """configs[3] with the cache: the ViT tokens of a key frame computed alone (B = 1) are the ones it gets as half of a pair --
    attention key splits and split-K Linears are decided per image -- so the cached windows equal the uncached ones bit for bit."""
    net, _ = vit_s16
    keys = synth.make_clip(11, 704, seed=1000, only=[0, 5, 10]).cuda()
    fm = FlowModel(net, feature_based=True, no_warp=False).eval()
    cache = KeyframeCache()
    for i in range(2):
        mvl, mvr = (cu(g) for g in synth.make_grids(N, 44, 44, seed=2000 + i, frame=(704, 704)))
        plain = fm.predict(keys[i:i + 1], keys[i + 1:i + 2], mvl, mvr, N, None)["pred"]
        cached = fm.predict(keys[i:i + 1], keys[i + 1:i + 2], mvl, mvr, N, None, key_cache=cache.window(5 * i, 5 * i + 5))["pred"]
        assert torch.equal(plain, cached), f"window {i}"
    assert (cache.hits, cache.misses) == (1, 1)


def test_predictor_with_keyframe_cache_matches_uncached_run(psp):
    """FlowPredictor(cache_keyframes=True) over 3 windows: the same 1072x1920 masks and the same temporal-consistency metric."""
    net, _ = psp
    keys = synth.make_clip(16, 713, seed=1000, only=[0, 5, 10, 15]).cuda()
    dl, dr = (cu(g) for g in synth.dummy_grids(N))
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    a, b = FlowPredictor(fm, 5, (1072, 1920)), FlowPredictor(fm, 5, (1072, 1920), cache_keyframes=True)
    for i in range(3):
        ma = a.predict_window(keys[i:i + 1], keys[i + 1:i + 2], dl, dr, to_host=False)
        mb = b.predict_window(keys[i:i + 1], keys[i + 1:i + 2], dl, dr, to_host=False, key_ids=(5 * i, 5 * i + 5))
        assert torch.equal(ma, mb)
    assert b.key_cache.hits == 2 and torch.equal(a.hist, b.hist) and a.temporal_consistency() == b.temporal_consistency()


# ------------------------------------------------------------------------------------------------ 8(f)-1 at full size
@pytest.fixture(scope="module")
def full_hd(psp):
    net, state = psp
    clip = synth.make_clip(6, (1072, 1920), seed=1200, only=[0, 5])
    mvl, mvr = synth.make_grids(N, 67, 120, seed=2100, frame=(1072, 1920), jitter=0.01)
    return net, state, clip[0:1], clip[1:2], mvl, mvr


def test_crop_grids_kernel_matches_the_oracle_for_all_eight_crops(full_hd):
    """fs_crop_grids = crop_motion_vector (flow/transform.py:215-261) for the 8 windows of a 1072x1920 frame in one launch:
    Python-rounded block ranges (45 or 44 rows, 45 or 44 / 45 columns), renormalisation, resize to 44x44."""
    _, _, _, _, mvl, mvr = full_hd
    wins = crops.crop_windows(1072, 1920, 713, 713)
    yx = [(w[0], w[2]) for w in wins]
    got = ops.crop_grids(cu(mvl) + cu(mvr), (1072, 1920), yx, (713, 713))
    assert got.shape == (8, 2 * (N - 1), 44, 44, 2)
    worst = 0.0
    for c, (y0, x0) in enumerate(yx):
        ol, orr = crops_oracle.crop_motion_vector([m.clone() for m in mvl], [m.clone() for m in mvr], 1072, 1920, 713, 713, y0, x0)
        for j, ref in enumerate(ol + orr):
            worst = max(worst, (got[c, j].cpu() - ref[0]).abs().max().item())
        one_l, one_r = crops.crop_motion_vector(cu(mvl), cu(mvr), 1072, 1920, 713, 713, y0, x0)
        assert all(torch.equal(one_l[j][0], got[c, j]) for j in range(N - 1))          # the per-crop entry = the batched launch
        assert all(torch.equal(one_r[j][0], got[c, N - 1 + j]) for j in range(N - 1))
    assert note("crop_grids_1072x1920_max_abs", worst) < 2e-6  # coordinates in [-1.1, 1.1]


@pytest.mark.parametrize("no_warp", [False, True])
def test_sliding_crops_1072x1920_pspnet_batched_route_against_oracle(full_hd, no_warp):
    """The reference's default real-video route (no_cropping=False, flow/base.py:182-209, 269-277) at its real size with the
    real network: 8 overlapping 713x713 crops of both key frames through ONE batched network call each way, fused tail +
    softmax + float64 accumulation, count normalisation, argmax.  Against crops_oracle (cv2 resize restated: unpinned) and
    against the generic one-crop-at-a-time route on the GPU."""
    net, state, prev, nxt, mvl, mvr = full_hd
    if no_warp:
        mvl, mvr = synth.dummy_grids(N)
    fm = FlowModel(net, feature_based=False, no_warp=no_warp).eval()
    canvas, mask = crops.compute_output(fm, N, prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), 713, 713, 5, want_mask=True)
    assert canvas.dtype == torch.float64 and canvas.shape == (N, 5, 1072, 1920) and mask.shape == (N, 1072, 1920)
    # generic route (what any non-HIP flow model takes): per-crop FlowModel.predict + fs_softmax_accumulate
    fn = lambda p, q, ml, mr: fm.predict(p, q, ml, mr, N, None)["pred"]  # noqa: E731
    canvas_g, mask_g = crops.compute_output(fm, N, prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), 713, 713, 5, want_mask=True, function=fn)
    assert torch.equal(canvas, canvas_g) and torch.equal(mask, mask_g)  # same arithmetic per crop, bit for bit
    # memoised on the crops' CONTENT (the oracle clones every crop: addresses repeat, contents do not lie): the 16 oracle forwards are
    # the same for the warp and the no_warp run of this test -- only the tail differs
    seg = _crop_oracle_seg(state)
    pred = lambda p, q, ml, mr: flow_oracle.predict_segmentation(lambda x: x, seg, p, q, ml, mr, N, no_warp)["pred"]  # noqa: E731
    ref = crops_oracle.compute_output(pred, N, prev, nxt, mvl, mvr, 713, 713, 5)
    tag = "nowarp" if no_warp else "warp"
    assert note(f"crops_1072x1920_{tag}_prob_max_abs", (canvas.cpu() - ref).abs().max().item()) < 2e-4  # probabilities in [0,1]
    agree = (mask.cpu() == ref.max(1)[1].to(torch.uint8)).float().mean().item()
    assert note(f"crops_1072x1920_{tag}_mask_disagreement", 1 - agree) < 1 - MASK_MIN
    # predict_step on this route: masks at (1072,1920) come straight from the canvas pass; and with the key-frame cache
    p0 = FlowPredictor(fm, 5, (1072, 1920), crop=(713, 713))
    m0 = p0.predict_window(prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), to_host=False)
    assert torch.equal(m0, mask)
    p1 = FlowPredictor(fm, 5, (1072, 1920), crop=(713, 713), cache_keyframes=True)
    p1.predict_window(nxt.cuda(), prev.cuda(), cu(mvl), cu(mvr), to_host=False, key_ids=(-5, 0))  # leaves frame 0 = `prev` cached
    m1 = p1.predict_window(prev.cuda(), nxt.cuda(), cu(mvl), cu(mvr), to_host=False, key_ids=(0, 5))
    assert p1.key_cache.hits == 1 and torch.equal(m1, mask)


def test_canvas_resize_argmax_f64_matches_torch():
    """flow/base.py:275-276 on the float64 canvas when the frame is not 1072x1920: bilinear (align_corners=True) in double +
    argmax, fused; and the identity case through fs_canvas_finish."""
    g = torch.Generator().manual_seed(3)
    canvas = torch.rand(3, 5, 97, 131, generator=g, dtype=torch.float64)
    count = torch.randint(1, 4, (97, 131), generator=g).double()
    ref_c = canvas / count
    ref = F.interpolate(ref_c, (211, 307), mode="bilinear", align_corners=True).max(1)[1].to(torch.uint8)
    cd = canvas.cuda()
    got = ops.canvas_finish(cd, count.cuda(), (211, 307), want_mask=True)
    assert torch.equal(cd.cpu(), ref_c)
    assert (got.cpu() == ref).float().mean().item() > 0.9999  # ties between interpolated doubles may break either way
    cd2 = canvas.cuda()
    same = ops.canvas_finish(cd2, count.cuda(), (97, 131), want_mask=True)
    assert torch.equal(same.cpu(), ref_c.max(1)[1].to(torch.uint8))


def test_segment_crops_equals_segment_of_cloned_crops(psp):
    """fs_segment_crops reads the windows in place from the full frames: bit-identical to cloning each crop (flow/base.py:199-200)
    and segmenting it, for both frames, at ragged offsets."""
    net, _ = psp
    fr = synth.make_clip(2, (300, 420), seed=77).cuda()
    yx = [(0, 0), (3, 17), (300 - 161, 420 - 161), (64, 259)]
    got = net.segment_crops(fr[0:1], fr[1:2], yx, (161, 161))
    assert got.shape[0] == 8
    for f in range(2):
        for c, (y, x) in enumerate(yx):
            one = net.segment(fr[f:f + 1, :, y:y + 161, x:x + 161].contiguous())
            assert torch.equal(got[f * 4 + c:f * 4 + c + 1], one), (f, c)
    only_a = net.segment_crops(fr[0:1], None, yx, (161, 161))
    assert torch.equal(only_a, got[:4])
    with pytest.raises(RuntimeError, match="leaves the"):
        net.segment_crops(fr[0:1], None, [(200, 0)], (161, 161))


def test_two_tensor_batch_equals_concatenation(psp):
    """fs_segment_forward2 / fs_encoder_forward2 read the two key frames in place: bit-identical to the torch.cat they replace."""
    net, _ = psp
    a, b = synth.make_clip(1, 161, seed=1).cuda(), synth.make_clip(2, 161, seed=2).cuda()
    assert torch.equal(net.segment(a, b), net.segment(torch.cat([a, b], 0)))
    assert torch.equal(net.encode_frames(a, b), net.encoder(torch.cat([a, b], 0)))
    assert torch.equal(net.segment(a, b, a), net.segment(torch.cat([a, b, a], 0)))


def test_seg_tail_accumulate_equals_tail_then_softmax_accumulate():
    """fs_seg_tail_accumulate (tail + softmax + float64 canvas fused) against the two-call route it replaces
    (fs_seg_tail -> fs_softmax_accumulate), bit for bit: warp and no_warp, two overlapping crops, K = 5 and K = 11."""
    from flood_uav_video_segmentation_amd import _lib
    from flood_uav_video_segmentation_amd._lib import check, ptr, stream_ptr

    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for k, no_warp in ((5, False), (5, True), (11, False)):
        n, h, w, ch, cw, H, W = 4, 12, 13, 90, 97, 150, 140
        lo_p, lo_n = torch.randn(1, k, h, w, generator=g).cuda(), torch.randn(1, k, h, w, generator=g).cuda()
        mvl, mvr = synth.dummy_grids(n) if no_warp else synth.make_grids(n, 6, 7, seed=3, frame=(ch, cw), jitter=0.05)
        mvl, mvr = cu(mvl), cu(mvr)
        fused_c = torch.zeros(n, k, H, W, dtype=torch.float64, device="cuda")
        fused_n = torch.zeros(H, W, dtype=torch.float64, device="cuda")
        two_c, two_n = torch.zeros_like(fused_c), torch.zeros_like(fused_n)
        for (y0, x0) in ((0, 0), (60, 43)):
            ops.seg_tail_accumulate(lo_p, lo_n, mvl, mvr, n, (ch, cw), no_warp, fused_c, fused_n, y0, x0)
            logits, _ = ops.seg_tail(lo_p, lo_n, mvl, mvr, n, (ch, cw), no_warp, want_logits=True)
            check(lib.fs_softmax_accumulate(ptr(logits), n, k, ch, cw, ptr(two_c), ptr(two_n), H, W, y0, x0, stream_ptr()))
        assert torch.equal(fused_c, two_c) and torch.equal(fused_n, two_n)
        assert float(fused_n.max()) == 2.0 and float(fused_n.min()) == 0.0
    with pytest.raises(RuntimeError, match="outside the canvas"):
        ops.seg_tail_accumulate(lo_p, lo_n, mvl, mvr, n, (ch, cw), False, fused_c, fused_n, 100, 100)


def test_crops_fuse_single_pass_equals_per_crop_accumulation():
    """fs_crops_fuse (every canvas pixel written once from the crops covering it, in crop order) against the route it replaces
    -- fs_seg_tail_accumulate per crop, then fs_canvas_finish -- bit for bit: warp and no_warp, n = 7 (two register chunks of
    frames), four overlapping crops incl. one pixel covered by all four, K = 5; the masks-only mode gives the same masks."""
    g = torch.Generator().manual_seed(11)
    k, n, h, w, ch, cw, H, W = 5, 7, 12, 13, 90, 97, 150, 160
    yx = [(0, 0), (0, 63), (60, 0), (60, 63)]
    lo_p, lo_n = torch.randn(4, k, h, w, generator=g).cuda(), torch.randn(4, k, h, w, generator=g).cuda()
    for no_warp in (False, True):
        grids = None
        if not no_warp:
            mvl, mvr = synth.make_grids(n, 10, 10, seed=3, frame=(H, W), jitter=0.03)
            grids = ops.crop_grids(cu(mvl) + cu(mvr), (H, W), yx, (ch, cw))
        ref_c = torch.zeros(n, k, H, W, dtype=torch.float64, device="cuda")
        ref_n = torch.zeros(H, W, dtype=torch.float64, device="cuda")
        for c, (y0, x0) in enumerate(yx):
            gl = [grids[c, j][None] for j in range(n - 1)] if grids is not None else cu(synth.dummy_grids(n)[0])
            gr = [grids[c, n - 1 + j][None] for j in range(n - 1)] if grids is not None else cu(synth.dummy_grids(n)[1])
            ops.seg_tail_accumulate(lo_p[c:c + 1], lo_n[c:c + 1], gl, gr, n, (ch, cw), no_warp, ref_c, ref_n, y0, x0)
        assert float(ref_n.max()) == 4.0 and float(ref_n.min()) == 1.0
        ref_m = ops.canvas_finish(ref_c, ref_n, None, want_mask=True)
        canvas, mask = ops.crops_fuse(lo_p, lo_n, grids, yx, (ch, cw), n, no_warp, (H, W), want_canvas=True, want_mask=True)
        assert torch.equal(canvas, ref_c) and torch.equal(mask, ref_m)
        none, only = ops.crops_fuse(lo_p, lo_n, grids, yx, (ch, cw), n, no_warp, (H, W), want_canvas=False, want_mask=True)
        assert none is None and torch.equal(only, ref_m)
    with pytest.raises(RuntimeError, match="outside the canvas"):
        ops.crops_fuse(lo_p, lo_n, None, [(0, 0), (0, 63), (60, 0), (100, 63)], (ch, cw), n, True, (H, W))
    # single frame (lo_next None: flow/base.py's compute_output on one key frame), one crop covering the whole canvas, K = 8
    lo1 = torch.randn(1, 8, h, w, generator=g).cuda()
    c1, m1 = ops.crops_fuse(lo1, None, None, [(0, 0)], (ch, cw), 1, True, (ch, cw), want_canvas=True, want_mask=True)
    ref1 = torch.softmax(F.interpolate(lo1.cpu(), (ch, cw), mode="bilinear", align_corners=True), 1).double()
    assert c1.shape == (1, 8, ch, cw) and (c1.cpu() - ref1).abs().max().item() < 1e-6
    assert (m1.cpu() == ref1.max(1)[1].to(torch.uint8)).float().mean().item() > 0.9999


def test_crop_grids_identity_block_range_and_edge_cases():
    """A crop whose block range already is (crop // 16)^2 skips the resize (64x64 crop on 16-px blocks at a block-aligned
    offset): the output is the renormalised cut, exactly; f64 grids are accepted; a crop outside the grid is refused."""
    mvl, mvr = synth.make_grids(3, 20, 30, seed=8, frame=(320, 480), jitter=0.02)
    got = ops.crop_grids(cu(mvl) + cu(mvr), (320, 480), [(32, 48), (0, 0)], (64, 64))
    assert got.shape == (2, 4, 4, 4, 2)
    for c, (y0, x0) in enumerate(((32, 48), (0, 0))):
        ref_l, ref_r = crops_oracle.crop_motion_vector([m.clone() for m in mvl], [m.clone() for m in mvr], 320, 480, 64, 64, y0, x0)
        for j, ref in enumerate(ref_l + ref_r):
            assert torch.equal(got[c, j].cpu(), ref[0]), (c, j)  # no interpolation involved: the same fp32 op chain as numpy
    g64 = ops.crop_grids([m.double().cuda() for m in mvl], (320, 480), [(32, 48)], (64, 64))
    assert torch.equal(g64[0], got[0, :2])
    with pytest.raises(RuntimeError, match="outside"):
        ops.crop_grids(cu(mvl), (320, 480), [(300, 0)], (64, 64))


def test_crop_grids_more_than_32_grids_and_more_than_32_crops():
    """frame_delta = 25 is the reference's default (flow/base.py:350): 48 grids per window; a 2160x3840 frame with 713 crops
    has 40 windows.  fs_crop_grids cuts both into slices of at most 32 per launch, written in place (the single
    launch used to refuse them)."""
    n = 25
    mvl, mvr = synth.make_grids(n, 135, 240, seed=9, frame=(2160, 3840), jitter=0.004)
    wins = crops.crop_windows(2160, 3840, 713, 713)
    yx = [(w[0], w[2]) for w in wins]
    assert len(yx) == 40 and len(mvl) + len(mvr) == 48
    got = ops.crop_grids(cu(mvl) + cu(mvr), (2160, 3840), yx, (713, 713))
    assert got.shape == (40, 48, 44, 44, 2)
    worst = 0.0
    for c in (0, 7, 31, 32, 39):  # both sides of the crop slice boundary
        ol, orr = crops_oracle.crop_motion_vector([m.clone() for m in mvl], [m.clone() for m in mvr], 2160, 3840, 713, 713, *yx[c])
        for j, ref in enumerate(ol + orr):  # all 48 grids: both sides of the grid slice boundary
            worst = max(worst, (got[c, j].cpu() - ref[0]).abs().max().item())
    assert note("crop_grids_48grids_40crops_max_abs", worst) < 2e-6
    # a slice of the problem through its own call = the same values (the slicing changes nothing per element)
    sub = ops.crop_grids(cu(mvl[:3]), (2160, 3840), yx[30:35], (713, 713))
    assert torch.equal(sub, got[30:35, :3])


def test_segment_crops_deeplab_equals_cloned_crops():
    """The crop-window read of the stem is shared by every conv network: DeepLabv3 (7x7 stem on the matrix cores) too."""
    state = synth.make_deeplab_state(50, 5, seed=4)
    net = FlowDeepLabv3(HP(50)).eval()
    net.load_state_dict(state)
    fr = synth.make_clip(2, (260, 330), seed=78).cuda()
    yx = [(0, 0), (7, 33), (260 - 193, 330 - 193)]
    got = net.segment_crops(fr[0:1], fr[1:2], yx, (193, 193))
    for f in range(2):
        for c, (y, x) in enumerate(yx):
            assert torch.equal(got[f * 3 + c:f * 3 + c + 1], net.segment(fr[f:f + 1, :, y:y + 193, x:x + 193].contiguous())), (f, c)


def test_full_size_networks_are_bit_repeatable_over_many_launches(psp, vit_s16):
    """A race between a wave's direct-to-LDS DMA and its neighbours' fragment reads shows as run-to-run differences at 713x713
    (one was found and fixed in round 1).  Every network at its BASELINE geometry, 40 back-to-back forwards each, all equal to
    the first bit for bit -- covers the concatenated-K instantiation, the LDS two-pass Winograd transforms, the MFMA stem,
    the key-split attention and the split-K Linears."""
    net, _ = psp
    x = synth.make_clip(2, 713, seed=99).cuda()
    first = net.segment(x)
    enc0 = net.encoder(x)
    for _ in range(40):
        assert torch.equal(net.segment(x), first)
    assert torch.equal(net.encoder(x), enc0)
    vit, _ = vit_s16
    v0 = vit.encoder(x)
    for _ in range(20):
        assert torch.equal(vit.encoder(x), v0)
    dl = FlowDeepLabv3(HP(101)).eval()
    dl.load_state_dict(synth.make_deeplab_state(101, 5, seed=0))
    d0 = dl.segment(x)
    for _ in range(20):
        assert torch.equal(dl.segment(x), d0)


def test_predict_clip_streams_its_input_and_leaves_the_predictor_alone(psp):
    """predict_clip must consume `items` lazily (at most two windows ahead of what it has emitted: a whole video's
    key frames must never be resident at once), must not install a permanent key cache on the predictor, and reset() starts a
    new video (no cached key frame, no last mask carried over)."""
    net, _ = psp
    size, nwin = 161, 7
    keys = synth.make_clip(5 * nwin + 1, size, seed=1001, only=[5 * i for i in range(nwin + 1)]).cuda()
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    dl, dr = synth.dummy_grids(N)
    pulled = []

    def stream():
        for i in range(nwin):
            pulled.append(i)
            yield {"frame_prev": keys[i:i + 1], "frame_next": keys[i + 1:i + 2], "mvs_left": cu(dl), "mvs_right": cu(dr), "key_ids": (5 * i, 5 * i + 5)}

    look = FlowPredictor(fm, 5, (size, size))
    ahead = []
    got = []
    for k, m in enumerate(look.predict_clip(stream(), to_host=False)):
        got.append(m)
        ahead.append(len(pulled) - (k + 1))
    assert len(got) == nwin and max(ahead) <= 1, ahead  # never more than one window beyond the one being emitted
    assert look.key_cache is None
    plain = FlowPredictor(fm, 5, (size, size))
    want = [plain.predict_window(keys[i:i + 1], keys[i + 1:i + 2], cu(dl), cu(dr), to_host=False) for i in range(nwin)]
    assert all(torch.equal(g, w) for g, w in zip(got, want)) and torch.equal(look.hist, plain.hist)
    # feature mode takes the per-window fallback with a clip-local cache: still no permanent cache on the predictor
    ff = FlowPredictor(FlowModel(net, feature_based=True, no_warp=True).eval(), 5, (size, size))
    list(ff.predict_clip(stream(), to_host=False))
    assert ff.key_cache is None
    # reset(): a second video whose frame ids restart must not hit the first video's cached key frame
    cached = FlowPredictor(fm, 5, (size, size), cache_keyframes=True)
    cached.predict_window(keys[0:1], keys[1:2], cu(dl), cu(dr), to_host=False, key_ids=(0, 5))
    cached.reset()
    assert cached.last_output is None
    other = cached.predict_window(keys[3:4], keys[4:5], cu(dl), cu(dr), to_host=False, key_ids=(5, 10))  # id 5 again, another frame
    assert cached.key_cache.hits == 0 and torch.equal(other, want[3])
    # reloading the weights invalidates the slot as well (the tag carries the weight generation)
    cached.predict_window(keys[4:5], keys[5:6], cu(dl), cu(dr), to_host=False, key_ids=(10, 15))
    assert cached.key_cache.hits == 1
    net.load_state_dict(psp[1])
    cached.predict_window(keys[5:6], keys[6:7], cu(dl), cu(dr), to_host=False, key_ids=(15, 20))
    assert cached.key_cache.hits == 1


@pytest.mark.parametrize("no_warp", [True, False])
def test_predictor_at_the_frames_own_size_takes_the_fused_mask_and_equals_the_resize_route(psp, no_warp):
    """r5: when out_size is the frame size the align_corners=True resize of flow/base.py:275 is the identity, so predict_window /
    predict_clip emit the fused tail's argmax instead of writing the fp32 logits and re-reading them.  Same masks, bit for bit."""
    net, _ = psp
    size = 161
    keys = synth.make_clip(11, size, seed=1003, only=[0, 5, 10]).cuda()
    fm = FlowModel(net, feature_based=False, no_warp=no_warp).eval()
    items = []
    for i in range(2):
        mvl, mvr = synth.dummy_grids(N) if no_warp else synth.make_grids(N, 10, 10, seed=2100 + i, frame=(size, size), jitter=0.02)
        items.append({"frame_prev": keys[i:i + 1], "frame_next": keys[i + 1:i + 2], "mvs_left": cu(mvl), "mvs_right": cu(mvr), "key_ids": (5 * i, 5 * i + 5)})
    want = [ops.resize_argmax_u8(fm.predict(it["frame_prev"], it["frame_next"], it["mvs_left"], it["mvs_right"], N, None)["pred"], (size, size))
            for it in items]
    p = FlowPredictor(fm, 5, (size, size), compute_metrics=False)
    got_w = [p.predict_window(it["frame_prev"], it["frame_next"], it["mvs_left"], it["mvs_right"], to_host=False) for it in items]
    got_c = list(p.predict_clip(items, to_host=False))
    assert all(torch.equal(a, b) for a, b in zip(got_w, want)) and all(torch.equal(a, b) for a, b in zip(got_c, want))


def test_a_user_network_with_single_tensor_segment_is_called_with_one_tensor():
    """FlowModel hands two tensors to `segment` only when the network advertises the multi-tensor call
    (`encode_frames`, as the HIP mirrors do); any other module's segment(x) sees the concatenated batch."""
    import torch.nn as nn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.seen = []

        def segment(self, x):
            self.seen.append(tuple(x.shape))
            return x[:, :, ::8, ::8].repeat(1, 2, 1, 1)[:, :5].contiguous()

    net = Net()
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    a, b = torch.rand(1, 3, 64, 64, device="cuda"), torch.rand(1, 3, 64, 64, device="cuda")
    dl, dr = synth.dummy_grids(3)
    out = fm.predict(a, b, cu(dl), cu(dr), 3, None)["pred"]
    assert out.shape == (3, 5, 64, 64) and net.seen == [(2, 3, 64, 64)]


@pytest.mark.parametrize("route", ["whole_frame_linear", "whole_frame_warp", "crops"])
def test_predict_clip_lookahead_batches_new_key_frames_and_is_bit_identical(psp, route):
    """FlowPredictor.predict_clip: key-frame cache + one window of look-ahead (the new key frames of two consecutive windows as
    one batch of two).  5 windows (an odd count: the last group has one window), masks and the temporal-consistency
    histogram equal to the plain per-window run bit for bit; every key frame is segmented exactly once."""
    net, _ = psp
    nwin = 5
    if route == "crops":
        size, crop, nwin = (300, 420), (161, 161), 3
        gsz = (size[0] // 16, size[1] // 16)
    else:
        size, crop, gsz = (713, 713), None, (44, 44)
    keys = synth.make_clip(5 * nwin + 1, size, seed=1000, only=[5 * i for i in range(nwin + 1)]).cuda()
    nw = route == "whole_frame_linear"
    fm = FlowModel(net, feature_based=False, no_warp=nw).eval()
    items = []
    for i in range(nwin):
        mvl, mvr = synth.dummy_grids(N) if nw else synth.make_grids(N, gsz[0], gsz[1], seed=2000 + i, frame=size, jitter=0.02)
        items.append({"frame_prev": keys[i:i + 1], "frame_next": keys[i + 1:i + 2], "mvs_left": cu(mvl), "mvs_right": cu(mvr),
                      "key_ids": (5 * i, 5 * i + 5)})
    plain = FlowPredictor(fm, 5, (size[0] + 7, size[1] + 9) if crop else (1072, 1920), crop=crop)
    want = [plain.predict_window(it["frame_prev"], it["frame_next"], it["mvs_left"], it["mvs_right"], to_host=False) for it in items]
    calls = []
    real = net._hip_net.segment if crop is None else net._hip_net.segment_crops
    if crop is None:
        net._hip_net.segment = lambda *fr: (calls.append(len(fr)), real(*fr))[1]
    else:
        net._hip_net.segment_crops = lambda a, b, yx, hw: (calls.append(1 if b is None else 2), real(a, b, yx, hw))[1]
    try:
        look = FlowPredictor(fm, 5, plain.out_size, crop=crop)
        got = list(look.predict_clip(items, to_host=False))
    finally:
        if crop is None:
            net._hip_net.segment = real
        else:
            net._hip_net.segment_crops = real
    assert len(got) == nwin and all(torch.equal(g, w) for g, w in zip(got, want))
    assert torch.equal(look.hist, plain.hist)
    if crop is None:  # r5: four new key frames per pass (three windows of look-ahead): the same masks, bit for bit
        look4 = FlowPredictor(fm, 5, plain.out_size, crop=crop)
        got4 = list(look4.predict_clip(items, to_host=False, keys_per_pass=4))
        assert len(got4) == nwin and all(torch.equal(g, w) for g, w in zip(got4, want)) and torch.equal(look4.hist, plain.hist)
    chunks = 1 if crop is None else -(-len(crops.crop_windows(size[0], size[1], crop[0], crop[1])) // 8)  # crop batches per frame pair
    assert sum(calls) == (nwin + 1) * chunks  # every key frame once ...
    assert calls.count(2) == (nwin + 1) // 2 * chunks and calls.count(1) == (nwin + 1) % 2 * chunks  # ... two at a time
