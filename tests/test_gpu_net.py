"""GPU parity of the network path (FlowPSPNet / FlowDeepLabv3 mirrors over the C ABI) against the CPU
oracle and the reference-generated golden fixtures.  fp32 end to end.

Tolerance stated by SURVEY.md 8d: decoder logits max-abs error <= 1e-3 x max|logit|; masks >= 99.9 % equal.  The asserts
are far tighter: 3e-5 (3-5x the 3-8e-6 measured on MI355X with Winograd F(6,3), gpurun_out/parity_measured.txt) and 99.99 %
mask agreement, so that a numerical regression of the Winograd / fused-head paths cannot hide under the stated tolerance."""
import pytest
import torch

from conftest import load_golden, note, rel_err
from flood_uav_video_segmentation_amd import synth
from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet
from oracle import deeplab_oracle, pspnet_oracle

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LOGIT_TOL = 3e-5
MASK_MIN = 0.9999


class HP:
    def __init__(self, layers=50, classes=5):
        self.layers, self.classes, self.pretrained = layers, classes, False


@pytest.fixture(scope="module")
def psp():
    state = synth.make_pspnet_state(50, 5, seed=0)
    net = FlowPSPNet(HP(50, 5)).eval()
    net.load_state_dict(state)
    return net, state


def test_weights_required_before_forward():
    net = FlowPSPNet(HP())
    with pytest.raises(RuntimeError, match="weights not loaded"):
        net.encoder(torch.zeros(1, 3, 65, 65, device="cuda"))


def test_state_dict_aliases_are_accepted(psp):
    """A Lightning checkpoint as the reference's own modules name it (tests/golden/checkpoint_keys.json: FlowBaseModel.model_G =
    FlowModel(FlowPSPNet), 1046 keys with every alias and the num_batches_tracked counters) loads strictly into
    FlowModel(FlowPSPNet) on the HIP path and gives the same logits as the canonical names."""
    import json
    import os

    from conftest import GOLDEN
    from flood_uav_video_segmentation_amd.flow.model import FlowModel

    ref, state = psp
    with open(os.path.join(GOLDEN, "checkpoint_keys.json")) as fh:
        keys = json.load(fh)["pspnet50"]
    ckpt = {}
    for k, shape in keys.items():
        c = FlowPSPNet.canonical_name(k[len("model_G.model."):])
        ckpt[k[len("model_G."):]] = torch.zeros(shape, dtype=torch.int64) if c is None else state[c]
        assert c is None or list(state[c].shape) == shape
    net = FlowPSPNet(HP()).eval()
    fm = FlowModel(net, feature_based=False, no_warp=True)
    res = fm.load_state_dict(ckpt, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    x = synth.make_clip(1, 65, seed=100).cuda()
    assert torch.equal(net.decoder(net.encoder(x)), ref.decoder(ref.encoder(x)))
    # the sub-module form used by the round-1 test: prefix handed to _load_from_state_dict
    net2 = FlowPSPNet(HP()).eval()
    missing, unexpected, errors = [], [], []
    net2._load_from_state_dict({"model_G." + k: v for k, v in ckpt.items()}, "model_G.model.", {}, True, missing, unexpected, errors)
    assert not missing and not unexpected and not errors
    assert torch.equal(net2.decoder(net2.encoder(x)), ref.decoder(ref.encoder(x)))


def test_pspnet_small_against_oracle_and_reference_golden(psp):
    net, state = psp
    z = load_golden("pspnet_small.npz")
    clip = synth.make_clip(2, 65, seed=100)
    feat = net.encoder(clip.cuda())
    assert feat.shape == (2, 4096, 9, 9)
    assert feat.stride() == (9 * 9 * 4096, 1, 9 * 4096, 4096)  # channels_last: zero-copy hand-off to the decoder
    logits = net.decoder(feat)
    assert logits.shape == (2, 5, 9, 9) and logits.is_contiguous()
    ofeat = pspnet_oracle.encoder(clip, state, 50)
    assert note("pspnet_65_feat_vs_oracle", rel_err(feat.cpu(), ofeat)) < LOGIT_TOL
    assert note("pspnet_65_logits_vs_oracle", rel_err(logits.cpu(), pspnet_oracle.decoder(ofeat, state))) < LOGIT_TOL
    assert rel_err(feat.cpu()[:, ::128], z["feat_slice"]) < LOGIT_TOL       # reference itself
    assert rel_err(logits.cpu(), z["logits"]) < LOGIT_TOL                   # reference itself


def test_pspnet_713_single_frame_against_reference_golden(psp):
    net, _ = psp
    z = load_golden("pspnet_713.npz")
    prev = synth.make_clip(6, 713, seed=1000)[0:1].cuda()
    feat = net.encoder(prev)
    assert feat.shape == (1, 4096, 90, 90)
    assert note("pspnet_713_feat_vs_reference", rel_err(feat.cpu()[:, ::256, ::6, ::6], z["feat_slice"])) < LOGIT_TOL
    assert abs(feat.double().abs().mean().item() / float(z["feat_absmean"]) - 1) < 1e-5
    lo = net.decoder(feat)
    assert note("pspnet_713_logits_lo_vs_reference", rel_err(lo.cpu(), z["logits_lo"])) < LOGIT_TOL
    from flood_uav_video_segmentation_amd import ops
    _, mask = ops.seg_tail(lo, None, [], [], 1, (713, 713), True, want_logits=False, want_mask=True)
    assert note("pspnet_713_single_mask_disagreement", 1 - (mask[0].cpu().numpy() == z["mask"]).mean()) < 1 - MASK_MIN


def test_batch_of_two_equals_two_single_frames(psp):
    net, _ = psp
    clip = synth.make_clip(2, 129, seed=7).cuda()
    both = net.decoder(net.encoder(clip))
    one = torch.cat([net.decoder(net.encoder(clip[i:i + 1])) for i in range(2)], 0)
    assert torch.equal(both, one)  # same kernels, same per-image reduction order


def test_decoder_accepts_nchw_contiguous_features(psp):
    net, _ = psp
    x = synth.make_clip(1, 65, seed=3).cuda()
    f = net.encoder(x)
    assert torch.equal(net.decoder(f.contiguous()), net.decoder(f))


@pytest.mark.parametrize("layers,seed", [(101, 1), (152, 2)])
def test_pspnet_deep_small_against_oracle_and_reference(layers, seed):
    """ResNet-101 / ResNet-152 backbones (model/pspnet.py:45-50): the deeper plans of the same executor, against the oracle and
    against the reference's own output (tests/golden/pspnet_deep_small.npz)."""
    state = synth.make_pspnet_state(layers, 5, seed=seed)
    net = FlowPSPNet(HP(layers, 5)).eval()
    net.load_state_dict(state)
    x = synth.make_clip(1, 65, seed=8)
    got = net.decoder(net.encoder(x.cuda())).cpu()
    assert note(f"pspnet{layers}_65_logits_vs_oracle", rel_err(got, pspnet_oracle.decoder(pspnet_oracle.encoder(x, state, layers), state))) < LOGIT_TOL
    assert note(f"pspnet{layers}_65_logits_vs_reference", rel_err(got, load_golden("pspnet_deep_small.npz")[f"logits{layers}"])) < LOGIT_TOL
    assert torch.equal(net.segment(x.cuda()).cpu().max(1)[1], got.max(1)[1]) or rel_err(net.segment(x.cuda()).cpu(), got) < 2e-5


def test_deeplabv3_r101_against_oracle_parity_unpinned():
    """DeepLabv3's arithmetic lives in un-vendored torchvision: no reference output exists offline, so this
    only checks HIP path == our own restatement of the public architecture (PARITY UNPINNED)."""
    state = synth.make_deeplab_state(101, 5, seed=0)
    net = FlowDeepLabv3(HP(101, 5)).eval()
    net.load_state_dict(state)
    x = synth.make_clip(2, 97, seed=9)
    feat = net.encoder(x.cuda())
    assert feat.shape == (2, 2048, 13, 13)
    ofeat = deeplab_oracle.encoder(x, state, 101)
    assert note("deeplab101_97_feat_vs_oracle", rel_err(feat.cpu(), ofeat)) < LOGIT_TOL
    assert note("deeplab101_97_logits_vs_oracle", rel_err(net.decoder(feat).cpu(), deeplab_oracle.decoder(ofeat, state))) < LOGIT_TOL
    # keys as FlowDeepLabv3's state_dict spells them (encoder.model.* / decoder.*)
    ref_keys = {("encoder.model." + k[9:] if k.startswith("backbone.") else "decoder." + k[11:]): v for k, v in state.items()}
    net2 = FlowDeepLabv3(HP(101, 5)).eval()
    net2.load_state_dict(ref_keys)
    assert torch.equal(net2.decoder(net2.encoder(x.cuda())), net.decoder(feat))


@pytest.mark.parametrize("size,b", [((713, 713), 2), ((161, 225), 1), ((97, 130), 3), ((33, 41), 2)])
def test_fused_segment_route_equals_decoder_of_encoder(psp, size, b):
    """fs_segment_forward (PSPNet: pyramid share of the head conv evaluated on the pooled maps, no 4096-channel concat) must
    give what fs_decoder_forward(fs_encoder_forward(x)) gives -- divisible (90x90) and ragged (21x29, 13x17) feature maps, and the
    smallest input the encoder accepts (33 px: a 5x6 map, smaller than the 6x6 pyramid level)."""
    net, _ = psp
    x = synth.make_clip(b, size, seed=21).cuda()
    two_step = net.decoder(net.encoder(x))
    fused = net.segment(x)
    assert fused.shape == two_step.shape and fused.dtype == torch.float32
    assert torch.equal(net.segment(x), fused)  # bit-repeatable from call to call
    assert rel_err(fused.cpu(), two_step.cpu()) < 2e-5
    assert (fused.max(1)[1] == two_step.max(1)[1]).float().mean().item() > 0.9998


def test_full_hd_frame_no_cropping_route(psp):
    """no_cropping=True on the 1072x1920 frames of the dataset (flow/base.py:271-272): the largest map the path sees
    (stem output 536x960x128, feature map 134x240, ragged pyramid windows).  Both routes run and agree."""
    net, _ = psp
    x = synth.make_clip(1, (1072, 1920), seed=23).cuda()
    fused = net.segment(x)
    two_step = net.decoder(net.encoder(x))
    assert fused.shape == (1, 5, 134, 240)
    assert torch.isfinite(fused).all()
    assert rel_err(fused.cpu(), two_step.cpu()) < 2e-5


def test_single_frame_pspnet_class_configs0(psp):
    """BASELINE configs[0]: the reference's single-frame `PSPNet.forward` (model/pspnet.py:86-110): head named `cls.*`, an
    `aux.*` head in the checkpoint that inference ignores, logits returned at the input size.  Against the reference golden."""
    from flood_uav_video_segmentation_amd.model.pspnet import PSPNet

    _, state = psp
    renamed = {("cls." + k[len("decoder."):] if k.startswith("decoder.") else k): v for k, v in state.items()}
    renamed["aux.0.weight"] = torch.zeros(256, 1024, 3, 3)  # present in reference checkpoints, unused in eval
    net = PSPNet(HP(50, 5)).eval()
    net.load_state_dict(renamed)
    z = load_golden("pspnet_713.npz")
    x = synth.make_clip(6, 713, seed=1000)[0:1].cuda()
    out = net(x)["pred"]
    assert out.shape == (1, 5, 713, 713)
    assert (out.max(1)[1].cpu().numpy() == z["mask"]).mean() > MASK_MIN
    lo = torch.from_numpy(z["logits_lo"])
    ref = torch.nn.functional.interpolate(lo, (713, 713), mode="bilinear", align_corners=True)
    assert rel_err(out.cpu(), ref) < LOGIT_TOL
    with pytest.raises(AssertionError):
        net(torch.zeros(1, 3, 100, 100, device="cuda"))  # (H - 1) % 8 != 0, as the reference asserts


def test_single_frame_deeplabv3_class_parity_unpinned():
    """Reference model/deeplabv3.py:11-33 (`DeepLabv3.forward`): checkpoint keys `model.backbone.*` / `model.classifier.*`
    (+ an ignored `model.aux_classifier.*`), output interpolated to the input size with align_corners=False as torchvision's
    wrapper does.  PARITY UNPINNED (torchvision absent): HIP path vs oracle/deeplab_oracle.py."""
    from flood_uav_video_segmentation_amd.model.deeplabv3 import DeepLabv3

    state = synth.make_deeplab_state(50, 5, seed=1)
    keys = {"model." + k: v for k, v in state.items()}
    keys["model.aux_classifier.0.weight"] = torch.zeros(256, 1024, 3, 3)
    net = DeepLabv3(HP(50, 5)).eval()
    net.load_state_dict(keys)
    x = synth.make_clip(1, (97, 113), seed=10)
    out = net(x.cuda())["pred"]
    lo = deeplab_oracle.decoder(deeplab_oracle.encoder(x, state, 50), state)
    ref = torch.nn.functional.interpolate(lo, (97, 113), mode="bilinear", align_corners=False)
    assert out.shape == (1, 5, 97, 113) and rel_err(out.cpu(), ref) < LOGIT_TOL


def test_two_handles_on_two_streams_do_not_interfere(psp):
    """The two-windows-in-flight mode of bench.py: two library handles (own workspaces) driven from two HIP streams at the
    same time must each give what a handle gives alone."""
    net, state = psp
    net2 = FlowPSPNet(HP(50, 5)).eval()
    net2.load_state_dict(state)
    xa = synth.make_clip(2, (321, 321), seed=31).cuda()
    xb = synth.make_clip(2, (321, 321), seed=32).cuda()
    ra, rb = net.segment(xa), net2.segment(xb)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(4):
        side.wait_stream(torch.cuda.current_stream())
        outs_a.append(net.segment(xa))
        with torch.cuda.stream(side):
            outs_b.append(net2.segment(xb))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ra) for o in outs_a) and all(torch.equal(o, rb) for o in outs_b)


def test_deeplabv3_aspp_on_the_winograd_lattice_path_parity_unpinned():
    """At 257x257 the feature map is 33x33: the dilation-12 ASPP conv takes the Winograd lattice decomposition (144 phases of
    3x3 pixels) while dilation 24 / 36 stay on the direct kernel (their lattices would be 2x2 / 1x1 pixels: the row rule of
    net.hip run_conv rejects them).  At 713x713 all three go to the lattice path: tests/test_gpu_fullsize.py.  HIP vs oracle."""
    state = synth.make_deeplab_state(50, 5, seed=2)
    net = FlowDeepLabv3(HP(50, 5)).eval()
    net.load_state_dict(state)
    x = synth.make_clip(1, (257, 257), seed=11)
    got = net.segment(x.cuda()).cpu()
    ref = deeplab_oracle.decoder(deeplab_oracle.encoder(x, state, 50), state)
    assert got.shape == ref.shape == (1, 5, 33, 33)
    assert note("deeplab50_257_logits_vs_oracle", rel_err(got, ref)) < LOGIT_TOL


def test_reserve_then_no_forward_grows_the_workspace():
    """fs_reserve (include/floodseg.h): after it, forwards at that geometry -- encoder, decoder, fused segment, crops, fewer
    frames -- leave fs_reserved_bytes unchanged: every allocation site of a forward is the grow arm of one helper (ws_grow,
    Winograd banks included), so an unchanged byte count means no hipMalloc / hipFree / device sync happened.  A larger geometry
    grows it; reserving is idempotent."""
    state = synth.make_pspnet_state(50, 5, seed=0)
    net = FlowPSPNet(HP(50, 5)).eval()
    net.load_state_dict(state)
    hn = net._hip_net
    assert hn.reserved_bytes() == 0
    net.reserve(2, 161, 161)
    r0 = hn.reserved_bytes()
    assert r0 >= hn._lib.fs_workspace_bytes(hn._h, 2, 161, 161) > 0
    x = synth.make_clip(2, 161, seed=5).cuda()
    lo = net.segment(x[0:1], x[1:2])
    f = net.encoder(x)
    net.decoder(f)
    net.segment(x[0:1])
    net.segment_crops(x[0:1], x[1:2], [(0, 0)], (161, 161))
    torch.cuda.synchronize()
    assert hn.reserved_bytes() == r0
    net.reserve(2, 161, 161)
    assert hn.reserved_bytes() == r0
    # same results as a handle that grew lazily
    ref, _ = (lambda n: (n, n.load_state_dict(state)))(FlowPSPNet(HP(50, 5)).eval())
    assert torch.equal(ref.segment(x[0:1], x[1:2]), lo)
    net.segment(synth.make_clip(2, 321, seed=6).cuda())  # unreserved, larger: grows lazily
    assert hn.reserved_bytes() > r0
    # DeepLabv3 and the Segmenter reserve as well
    dl = FlowDeepLabv3(HP(50, 5)).eval()
    dl.load_state_dict(synth.make_deeplab_state(50, 5, seed=1))
    dl.reserve(2, 193, 193)
    r1 = dl._hip_net.reserved_bytes()
    dl.decoder(dl.encoder(x[:, :, :, :].new_zeros(2, 3, 193, 193)))
    dl.segment(x.new_zeros(1, 3, 193, 193))
    assert dl._hip_net.reserved_bytes() == r1


def test_first_forward_after_reserve_is_hip_graph_capturable():
    """fs_reserve's contract from the caller's side: on a FRESH handle, reserve, then capture the very first forward (both key
    frames + the fused tail) into a HIP graph -- a capture fails on any hipMalloc / hipFree / device synchronisation, so this is
    the strongest check that nothing of the kind is left in a reserved forward (lazily built Winograd banks included) -- and the
    replay gives the eager result bit for bit."""
    from flood_uav_video_segmentation_amd import ops

    state = synth.make_pspnet_state(50, 5, seed=0)
    net = FlowPSPNet(HP(50, 5)).eval()
    net.load_state_dict(state)
    keys = synth.make_clip(2, 193, seed=12).cuda()
    dl, dr = [[g.cuda() for g in gs] for gs in synth.dummy_grids(5)]
    net.reserve(2, 193, 193)

    def window():
        lows = net.segment(keys[0:1], keys[1:2])
        return ops.seg_tail(lows[0:1], lows[1:2], dl, dr, 5, (193, 193), True, want_logits=True, want_mask=True)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):  # NO warm-up forward before the capture: the reserve alone must suffice
            logits, mask = window()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    got_l, got_m = logits.clone(), mask.clone()
    ref_l, ref_m = window()
    assert torch.equal(got_l, ref_l) and torch.equal(got_m, ref_m)


def test_split_operand_route_is_as_close_to_a_float64_network_as_the_fp32_mfma_route(psp):
    """The default arithmetic route (fp32 operands as three exact bf16 terms on the bf16 matrix cores) against the
    SAME network evaluated in float64 (the oracle's functions on double tensors): its error is that of the fp32-MFMA route
    (hip_no_split_bf16) -- both are fp32 evaluations of the network, neither is a reduced-precision one."""
    net, state = psp

    class HP32(HP):
        hip_no_split_bf16 = True
    net32 = FlowPSPNet(HP32(50, 5)).eval()
    net32.load_state_dict(state)
    s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in state.items()}
    for size, seed in ((257, 71), (129, 72)):
        x = synth.make_clip(2, size, seed=seed)
        ref = pspnet_oracle.decoder(pspnet_oracle.encoder(x.double(), s64, 50), s64)
        e_split = note(f"pspnet_{size}_split_route_vs_float64", rel_err(net.segment(x.cuda()).double().cpu(), ref))
        e_f32 = note(f"pspnet_{size}_fp32_mfma_route_vs_float64", rel_err(net32.segment(x.cuda()).double().cpu(), ref))
        e_cpu = note(f"pspnet_{size}_torch_cpu_fp32_vs_float64",
                     rel_err(pspnet_oracle.decoder(pspnet_oracle.encoder(x, state, 50), state).double(), ref))
        assert e_split < LOGIT_TOL and e_f32 < LOGIT_TOL
        assert e_split < 2.0 * max(e_f32, e_cpu) + 2e-7, (e_split, e_f32, e_cpu)  # measured 0.86x / 1.49x (129 / 257)


@pytest.mark.parametrize("classes", [19, 3, 9])
def test_other_class_counts_through_the_fused_head_and_the_tail(classes):
    """Every test above uses the dataset's K = 5.  The head finish applies the classifier in passes of 8 classes (19 = 8 + 8 + 3,
    9 = 8 + 1) and the fused tail has an instantiation for K > 8: a PSPNet with K classes at 97x97 against the oracle -- the fused
    segment route, the decoder(encoder) route, and a whole logit-warp window (logits and masks)."""
    from flood_uav_video_segmentation_amd.flow.model import FlowModel
    from oracle import flow_oracle

    state = synth.make_pspnet_state(50, classes, seed=5)
    net = FlowPSPNet(HP(50, classes)).eval()
    net.load_state_dict(state)
    clip = synth.make_clip(2, 97, seed=41)
    ref_lo = pspnet_oracle.decoder(pspnet_oracle.encoder(clip, state, 50), state)
    assert ref_lo.shape[1] == classes
    assert note(f"pspnet50_K{classes}_97_segment_vs_oracle", rel_err(net.segment(clip.cuda()).cpu(), ref_lo)) < LOGIT_TOL
    assert rel_err(net.decoder(net.encoder(clip.cuda())).cpu(), ref_lo) < LOGIT_TOL
    n = 4
    mvl, mvr = synth.make_grids(n, 6, 6, seed=42, frame=(97, 97), jitter=0.05)
    fm = FlowModel(net, feature_based=False, no_warp=False).eval()
    got = fm.predict(clip[0:1].cuda(), clip[1:2].cuda(), [m.cuda() for m in mvl], [m.cuda() for m in mvr], n, None, with_mask=True)
    enc = lambda x: pspnet_oracle.encoder(x, state, 50)  # noqa: E731
    dec = lambda f: pspnet_oracle.decoder(f, state)  # noqa: E731
    ref = flow_oracle.predict_segmentation(enc, dec, clip[0:1], clip[1:2], mvl, mvr, n, False)["pred"]
    assert got["pred"].shape == (n, classes, 97, 97)
    assert note(f"pspnet50_K{classes}_97_window_logits_vs_oracle", rel_err(got["pred"].cpu(), ref)) < LOGIT_TOL
    assert (got["mask"].cpu() == ref.max(1)[1].to(torch.uint8)).float().mean().item() > 0.999
    assert torch.equal(got["mask"], got["pred"].max(1)[1].to(torch.uint8))
