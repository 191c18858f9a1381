"""GPU parity of FlowModel (HIP path) against the reference-generated goldens and the CPU oracle."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden, memo_by_content, note, rel_err, toy_weights
from flood_uav_video_segmentation_amd import ops, synth
from flood_uav_video_segmentation_amd.flow.model import FlowModel, get_default_grid
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet
from oracle import flow_oracle, pspnet_oracle

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 2e-5       # toy model: tiny torch convs on the GPU + HIP interpolation kernels
NET_TOL = 3e-5   # full PSPNet in fp32: 3-5x the 3-8e-6 measured (see test_gpu_net.py)
MASK_MIN = 0.9999


def toy_model():
    w = toy_weights()
    m = nn.Module()
    enc, dec = nn.Conv2d(3, 8, 3, stride=4, padding=1), nn.Conv2d(8, 5, 1)
    enc.weight.data, enc.bias.data, dec.weight.data, dec.bias.data = w["enc_w"], w["enc_b"], w["dec_w"], w["dec_b"]
    m.encoder, m.decoder = nn.Sequential(enc, nn.ReLU()), dec
    return m.cuda().eval()


def cu(ts):
    return [t.cuda() for t in ts]


def test_default_grid_is_the_reference_array():
    g = get_default_grid()
    assert g.dtype == np.float64 and np.array_equal(g, load_golden("default_grid.npz")["grid"])


@pytest.mark.parametrize("n", [3, 5])
@pytest.mark.parametrize("fb", [False, True])
@pytest.mark.parametrize("nw", [False, True])
def test_predict_toy_model_matches_reference(n, fb, nw, profiler):
    z = load_golden("toy_predict.npz")
    prev, nxt = torch.from_numpy(z["prev"]).cuda(), torch.from_numpy(z["next"]).cuda()
    h, w = prev.shape[2:]
    mvl, mvr = synth.dummy_grids(n) if nw else synth.make_grids(n, 4, 5, seed=40 + n, frame=(h, w), jitter=0.05)
    fm = FlowModel(toy_model(), feature_based=fb, no_warp=nw).eval()
    keep = prev.clone()
    out = fm.predict(prev, nxt, cu(mvl), cu(mvr), n, profiler)["pred"]
    assert out.shape == (n, 5, h, w) and out.dtype == torch.float32
    assert rel_err(out.cpu(), z[f"predict_n{n}_fb{int(fb)}_nw{int(nw)}"]) < TOL
    assert torch.equal(prev, keep)  # inputs are never mutated
    assert {"predict_encoder", "predict_decoder"} <= set(profiler.names)
    both = fm.predict(prev, nxt, cu(mvl), cu(mvr), n, None, with_mask=True)  # extension: the tail's argmax next to the logits
    assert torch.equal(both["pred"], out) and torch.equal(both["mask"], ops.argmax_u8(out))


def test_predict_single_frame_and_f64_grids(profiler):
    z = load_golden("toy_predict.npz")
    prev, nxt = torch.from_numpy(z["prev"]).cuda(), torch.from_numpy(z["next"]).cuda()
    fm = FlowModel(toy_model(), feature_based=False, no_warp=False).eval()
    out = fm.predict(prev, None, [], [], 5, profiler)["pred"]
    assert out.shape[0] == 1 and rel_err(out.cpu(), z["single_n5"]) < TOL
    mvl, mvr = synth.make_grids(5, 4, 5, seed=45, frame=tuple(prev.shape[2:]), jitter=0.05)
    out64 = fm.predict(prev, nxt, [m.double().cuda() for m in mvl], [m.double().cuda() for m in mvr], 5, profiler)["pred"]
    assert rel_err(out64.cpu(), z["predict_n5_fb0_nw0"]) < TOL


@pytest.mark.parametrize("fb", [False, True])
@pytest.mark.parametrize("nw", [False, True])
def test_eval_forward_mixed_distances_matches_reference(fb, nw):
    z = load_golden("toy_predict.npz")
    fp, fn_ = torch.from_numpy(z["fwd_prev"]).cuda(), torch.from_numpy(z["fwd_next"]).cuda()
    h, w = fp.shape[2:]
    n = 5
    per = [synth.make_grids(n, 4, 5, seed=60 + b, frame=(h, w), jitter=0.05) for b in range(3)]
    mvl = [torch.cat([per[b][0][j] for b in range(3)], 0).cuda() for j in range(n - 1)]
    mvr = [torch.cat([per[b][1][j] for b in range(3)], 0).cuda() for j in range(n - 1)]
    fm = FlowModel(toy_model(), feature_based=fb, no_warp=nw).eval()
    out = fm(None, fp, fn_, mvl, mvr, torch.tensor([1, 2, 4]), torch.tensor([4, 3, 1]))["pred"]
    assert rel_err(out.cpu(), z[f"forward_fb{int(fb)}_nw{int(nw)}"]) < TOL


def test_training_mode_is_refused():
    fm = FlowModel(toy_model()).train()
    with pytest.raises(NotImplementedError):
        fm(None, None, None, [], [], [1], [1])


class HP:
    layers, classes, pretrained = 50, 5, False


@pytest.fixture(scope="module")
def psp_flow():
    net = FlowPSPNet(HP()).eval()
    state = synth.make_pspnet_state(50, 5, seed=0)
    net.load_state_dict(state)
    return net, state


def test_pspnet_713_config2_and_config3_match_reference_masks(psp_flow, profiler):
    """BASELINE configs[1] (key-frame + linear interp) and configs[2]-style (logit warp) at full size."""
    net, _ = psp_flow
    z = load_golden("predict_713.npz")
    clip = synth.make_clip(6, 713, seed=1000)
    prev, nxt = clip[0:1].cuda(), clip[5:6].cuda()
    n = 5
    dl, dr = synth.dummy_grids(n)
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    out2 = fm.predict(prev, nxt, cu(dl), cu(dr), n, profiler)["pred"]
    assert out2.shape == (5, 5, 713, 713)
    assert note("cfg1_pspnet_713_logits_vs_reference", rel_err(out2[:, :, ::16, ::16].cpu(), z["cfg2_logits_sub"])) < NET_TOL
    assert note("cfg1_pspnet_713_mask_disagreement", 1 - (ops.argmax_u8(out2).cpu().numpy() == z["cfg2_mask"]).mean()) < 1 - MASK_MIN
    assert torch.equal(fm.predict_masks(prev, nxt, cu(dl), cu(dr), n), ops.argmax_u8(out2))
    post = ops.resize_argmax_u8(out2, (1072, 1920))  # flow/base.py:275-277 without the 206 MB intermediate
    assert (post[:, ::4, ::4].cpu().numpy() == z["cfg2_post_mask_sub"]).mean() > MASK_MIN
    mvl, mvr = synth.make_grids(n, 44, 44, seed=2000)
    fm = FlowModel(net, feature_based=False, no_warp=False).eval()
    out3 = fm.predict(prev, nxt, cu(mvl), cu(mvr), n, profiler)["pred"]
    assert note("warp_pspnet_713_logits_vs_reference", rel_err(out3[:, :, ::16, ::16].cpu(), z["cfg3_logits_sub"])) < NET_TOL
    assert note("warp_pspnet_713_mask_disagreement", 1 - (ops.argmax_u8(out3).cpu().numpy() == z["cfg3_mask"]).mean()) < 1 - MASK_MIN
    # mIoU of the HIP masks against the reference masks (util/util.py semantics): within 0.1 pp of 100 %
    hist = ops.iou_hist(ops.argmax_u8(out3), torch.from_numpy(z["cfg3_mask"]).cuda(), 5).cpu().numpy().astype(np.float64)
    miou = np.mean(hist[0] / (hist[1] + hist[2] - hist[0] + 1e-10))
    assert miou > 0.9995


@pytest.mark.parametrize("nw", [False, True])
def test_pspnet_feature_based_against_oracle(psp_flow, nw, profiler):
    """Feature propagation (predict_feature) on the real network at a small frame: C=4096 NHWC warps."""
    net, state = psp_flow
    n = 3
    clip = synth.make_clip(2, 129, seed=21)
    mvl, mvr = synth.dummy_grids(n) if nw else synth.make_grids(n, 8, 8, seed=77, frame=(129, 129), jitter=0.05)
    fm = FlowModel(net, feature_based=True, no_warp=nw).eval()
    out = fm.predict(clip[0:1].cuda(), clip[1:2].cuda(), cu(mvl), cu(mvr), n, profiler)["pred"]
    enc = lambda x: pspnet_oracle.encoder(x, state, 50)  # noqa: E731
    dec = lambda f: pspnet_oracle.decoder(f, state)  # noqa: E731
    ref = flow_oracle.predict_feature(enc, dec, clip[0:1], clip[1:2], mvl, mvr, n, nw)["pred"]
    assert out.shape == ref.shape == (3, 5, 129, 129)
    assert note(f"pspnet_feature_129_nw{int(nw)}_vs_oracle", rel_err(out.cpu(), ref)) < NET_TOL
    assert {"predict_warp", "predict_fusion"} <= set(profiler.names)
    fm.fused_feature_tail = False  # the op-by-op route (one HIP launch per torch op of flow/model.py:131-171): same bits
    assert torch.equal(fm.predict(clip[0:1].cuda(), clip[1:2].cuda(), cu(mvl), cu(mvr), n, None)["pred"], out)
    one = FlowModel(net, feature_based=True, no_warp=nw).eval()
    single = one.predict(clip[0:1].cuda(), None, cu(mvl), cu(mvr), n, None)["pred"]  # frame_next=None (:127): one map
    one.fused_feature_tail = False
    assert single.shape == (1, 5, 129, 129) and torch.equal(one.predict(clip[0:1].cuda(), None, cu(mvl), cu(mvr), n, None)["pred"], single)


def test_sliding_crop_inference_against_oracle_parity_unpinned():
    """compute_output + crop_motion_vector on a frame larger than the crop (toy network so the
    CPU side stays cheap).  cv2 is absent offline -> the grid resize is restated (half-pixel bilinear): unpinned."""
    from flood_uav_video_segmentation_amd.flow import crops
    from oracle import crops_oracle

    n, H, W, ch, cw, K = 3, 160, 272, 97, 97, 5
    clip = synth.make_clip(2, (H, W), seed=41)
    # full-frame block grids (H/16 x W/16), identity + motion + jitter
    mvl, mvr = synth.make_grids(n, H // 16, W // 16, seed=42, frame=(H, W), jitter=0.03)
    toy = toy_model()
    fm = FlowModel(toy, feature_based=False, no_warp=False).eval()
    got, mask = crops.compute_output(fm, n, clip[0:1].cuda(), clip[1:2].cuda(), cu(mvl), cu(mvr), ch, cw, K, want_mask=True)
    w = toy_weights()
    enc = lambda x: torch.relu(torch.nn.functional.conv2d(x, w["enc_w"], w["enc_b"], 4, 1))  # noqa: E731
    dec = lambda f: torch.nn.functional.conv2d(f, w["dec_w"], w["dec_b"])  # noqa: E731
    pred = lambda p, q, ml, mr: flow_oracle.predict_segmentation(enc, dec, p, q, ml, mr, n, False)["pred"]  # noqa: E731
    ref = crops_oracle.compute_output(pred, n, clip[0:1], clip[1:2], mvl, mvr, ch, cw, K)
    assert got.dtype == torch.float64 and got.shape == ref.shape == (n, K, H, W)
    assert (got.cpu() - ref).abs().max().item() < 1e-5  # probabilities in [0,1]
    assert (mask.cpu() == ref.max(1)[1].to(torch.uint8)).float().mean().item() > 0.999
    assert crops.crop_windows(1072, 1920, 713, 713) == [(0, 713, 0, 713), (0, 713, 476, 1189), (0, 713, 952, 1665), (0, 713, 1207, 1920),
                                                         (359, 1072, 0, 713), (359, 1072, 476, 1189), (359, 1072, 952, 1665), (359, 1072, 1207, 1920)]


def test_sliding_crop_inference_with_the_reference_default_frame_delta_25():
    """frame_delta = 25 (FlowDataModule's default, flow/base.py:350): 48 grids per window through crop_motion_vector and the
    warp chains of every crop -- ADVICE r2: the one-launch crop_grids used to refuse more than 32 grids."""
    from flood_uav_video_segmentation_amd.flow import crops
    from oracle import crops_oracle

    n, H, W, ch, cw, K = 25, 160, 224, 97, 97, 5
    clip = synth.make_clip(2, (H, W), seed=51)
    mvl, mvr = synth.make_grids(n, H // 16, W // 16, seed=52, frame=(H, W), jitter=0.01)
    fm = FlowModel(toy_model(), feature_based=False, no_warp=False).eval()
    got = crops.compute_output(fm, n, clip[0:1].cuda(), clip[1:2].cuda(), cu(mvl), cu(mvr), ch, cw, K)
    w = toy_weights()
    enc = lambda x: torch.relu(torch.nn.functional.conv2d(x, w["enc_w"], w["enc_b"], 4, 1))  # noqa: E731
    dec = lambda f: torch.nn.functional.conv2d(f, w["dec_w"], w["dec_b"])  # noqa: E731
    pred = lambda p, q, ml, mr: flow_oracle.predict_segmentation(enc, dec, p, q, ml, mr, n, False)["pred"]  # noqa: E731
    ref = crops_oracle.compute_output(pred, n, clip[0:1], clip[1:2], mvl, mvr, ch, cw, K)
    assert got.shape == ref.shape == (n, K, H, W)
    assert (got.cpu() - ref).abs().max().item() < 2e-5  # 24-deep warp chains; probabilities in [0,1]


def test_motion_vectors_to_grids_last_writer_wins_and_npy_round_trip(tmp_path):
    """the grid producer against the oracle restatement (pinned to the reference's script by the next test)."""
    from flood_uav_video_segmentation_amd.flow import grids
    from oracle import crops_oracle

    rng = np.random.default_rng(7)
    n = 9000  # more vectors than blocks: many cells are hit several times
    src = np.stack([rng.integers(-40, 1960, n), rng.integers(-40, 1100, n)], 1)
    dst = src + rng.integers(-48, 49, (n, 2))
    mv = np.concatenate([np.full((n, 1), -1), np.full((n, 2), 16), src, dst, np.zeros((n, 3), int)], 1).astype(np.int64)
    grid, inv = grids.motion_vectors_to_grids(mv, 1080, 1920)
    rgrid, rinv = crops_oracle.motion_vectors_to_grids(mv, 1080, 1920, get_default_grid())
    assert grid.dtype == torch.float64 and grid.shape == (67, 120, 2)
    assert np.array_equal(grid.cpu().numpy(), rgrid) and np.array_equal(inv.cpu().numpy(), rinv)  # float64, bit-exact
    g0, i0 = grids.motion_vectors_to_grids(np.zeros((0, 10), np.int64), 1072, 1920)
    assert np.array_equal(g0.cpu().numpy(), get_default_grid()) and np.array_equal(i0.cpu().numpy(), get_default_grid())
    grids.save_grid(tmp_path / "0.npy", grid)
    back = grids.load_grid(tmp_path / "0.npy")
    assert back.dtype == torch.float32 and torch.equal(back, grid.float().cpu())
    with pytest.raises(AssertionError):
        bad = mv.copy()
        bad[3, 0] = 1
        grids.motion_vectors_to_grids(bad, 1080, 1920)


def test_motion_vectors_to_grids_matches_the_references_script():
    """fs_mv_to_grids against the reference's own script run top to bottom (tests/golden/mv_grids.npz, see
    test_oracle_golden.py::test_grid_producer_oracle_matches_the_references_script): float64, bit-exact, four frames."""
    from flood_uav_video_segmentation_amd.flow import grids

    z = load_golden("mv_grids.npz")
    for i, (h, w, n, seed) in enumerate(z["frames"]):
        grid, inv = grids.motion_vectors_to_grids(synth.motion_vectors(int(h), int(w), int(n), int(seed)), int(h), int(w))
        assert np.array_equal(grid.cpu().numpy(), z[f"grids_{i}"]) and np.array_equal(inv.cpu().numpy(), z[f"inv_grids_{i}"]), i


def test_crop_motion_vector_matches_the_references_function():
    """fs_crop_grids (flow/crops.py::crop_motion_vector) against the reference's own flow/transform.py:215-261 on the windows
    whose block range already has the final size (tests/golden/transforms.npz; see test_oracle_golden.py::
    test_crop_motion_vector_oracle_matches_the_references_function for what is and is not covered): all eight 704-crop windows
    of a 1072 x 1920 frame, offsets off the block edges, two round-half-to-even cases, a small centre crop."""
    from flood_uav_video_segmentation_amd.flow import crops

    z = load_golden("transforms.npz")
    worst = 0.0
    for k, (h, w, gh, gw, ch, cw, ho, wo) in enumerate(z["geometries"].tolist()):
        ml, mr = synth.make_grids(3, gh, gw, seed=300 + k, frame=(h, w), jitter=0.03)
        cl, cr = crops.crop_motion_vector(cu(ml), cu(mr), h, w, ch, cw, ho, wo)
        assert cl[0].shape == (1, ch // 16, cw // 16, 2)
        worst = max(worst, (torch.cat(cl).cpu() - torch.from_numpy(z[f"crop_left_{k}"])).abs().max().item(),
                    (torch.cat(cr).cpu() - torch.from_numpy(z[f"crop_right_{k}"])).abs().max().item())
    note("crop_motion_vector_vs_reference_same_size_abs", worst)
    assert worst < 1e-6   # coordinates in [-1.1, 1.1], fp32 chain of 6 operations (the compiler may contract mul + add)


@pytest.mark.parametrize("split", ["val", "test"])
def test_eval_windows_match_the_references_transform_chain(tmp_path, split):
    """EvalWindows.__getitem__ (decode, Resize at the native size, IgnoreClasses, Crop('center') incl. the grids, ToTensor,
    Normalize on the GPU) against the reference's own transform_val / transform_test chains (flow/base.py:396-431,
    flow/transform.py) on the two items of a tiny labelled video -- tests/golden/transforms.npz.  Frames are stored as PNG
    (lossless: the decoded bytes are the seeded ones), so the .jpg suffix of frame_path is overridden; the 10 x 16 raster's
    identity grid stands in for get_default_grid() where FlowData pads, as it does in the generator."""
    import os

    from PIL import Image

    from flood_uav_video_segmentation_amd.flow.dataset import EvalWindows

    class PngWindows(EvalWindows):
        def frame_path(self, f_id):
            return os.path.join(self.data_root, "frames", self.video_id, "images", f"{f_id}.png")

    z = load_golden("transforms.npz")
    h, w, gh, gw, ch, cw, delta, seed, frames = z["item_shape"].tolist()
    ignore = tuple(z["item_ignore"].tolist())
    files = synth.transform_frames(h, w, gh, gw, range(frames), ignore, seed)
    base = os.path.join(tmp_path, "frames", "vid")
    for d in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(base, d))
    os.makedirs(os.path.join(tmp_path, "masks"))
    for f, rec in files.items():
        Image.fromarray(rec["image"]).save(os.path.join(base, "images", f"{f}.png"))
        np.save(os.path.join(base, "grids", f"{f}.npy"), rec["grid"])
        np.save(os.path.join(base, "inv_grids", f"{f}.npy"), rec["inv_grid"])
    with open(os.path.join(tmp_path, "list.txt"), "w") as fh:
        for f, _ in z["items"].tolist():
            Image.fromarray(files[f]["label"]).save(os.path.join(tmp_path, "masks", f"{f}.png"))
            fh.write(f"masks/{f}.png vid {f}\n")
    ds = PngWindows(str(tmp_path), os.path.join(tmp_path, "list.txt"), split=split, frame_delta=delta, size=(h, w),
                    center_crop=(ch, cw) if split == "val" else None, classes_ignore=ignore)
    ds.default_grid = torch.from_numpy(synth.identity_grid(gh, gw)).float()
    assert len(ds) == 2
    for k, (f, l) in enumerate(z["items"].tolist()):
        item = ds[k]
        assert int(item["left_index"][0]) == l and int(item["right_index"][0]) == delta - l
        assert torch.equal(item["label"][0].cpu(), torch.from_numpy(z[f"{split}{k}_label"]))                    # int64, bit-exact
        for name in ("frame_prev", "frame_next"):
            ref = torch.from_numpy(z[f"{split}{k}_{name}"])
            assert item[name].shape[1:] == ref.shape
            assert (item[name][0].cpu() - ref).abs().max().item() < 1e-6                                         # (x - mean) / std, |x| < 2.7
        for name in ("mvs_left", "mvs_right"):
            ref = torch.from_numpy(z[f"{split}{k}_{name}"])
            got = torch.cat(item[name]).cpu()
            assert got.shape == ref.shape
            # test: float64 -> float32 only (bit-exact).  val: the reference crops in float64 and then rounds, the HIP path rounds first
            assert torch.equal(got, ref) if split == "test" else (got - ref).abs().max().item() < 1e-6


@pytest.mark.parametrize("route", ["whole", "crops"])
def test_predictor_and_evaluator_match_the_references_lightning_steps(route):
    """FlowPredictor.predict_window / temporal_consistency and FlowEvaluator.test_step / summary on the HIP path against THE
    REFERENCE'S OWN predict_step / on_predict_end and test_step / test_epoch_end (tests/golden/lightning_steps.npz; how the
    reference's FlowBaseModel is run without its training stack is described in the generator and in
    test_oracle_golden.py::test_oracle_chain_matches_the_references_predict_step_and_test_step).  Three consecutive windows of a
    1072 x 1920 clip, whole frame and 704 x 704 sliding crops; the toy encoder / decoder are torch convs on the GPU, everything
    around them (crop grids, warp chain, softmax canvas, resize + argmax, histograms, palette) is the HIP path."""
    from flood_uav_video_segmentation_amd.flow.predict import PALETTE, FlowEvaluator, FlowPredictor, colorize

    z = load_golden("lightning_steps.npz")
    H, W = 1072, 1920
    crop = None if route == "whole" else tuple(z["crop"].tolist())
    clip = synth.make_clip(16, (H, W), seed=1300, only=z["keys"].tolist()).cuda()
    fm = FlowModel(toy_model(), feature_based=False, no_warp=False).eval()
    pred = FlowPredictor(fm, classes=5, out_size=(H, W), crop=crop, compute_metrics=True)
    masks = []
    for k in range(3):
        mvl, mvr = synth.make_grids(5, 67, 120, seed=1310 + k, frame=(H, W), jitter=0.01)
        masks.append(pred.predict_window(clip[k:k + 1], clip[k + 1:k + 2], cu(mvl), cu(mvr)))
    masks = np.concatenate(masks)
    assert masks.shape == (15, H, W) and masks.dtype == np.uint8
    agree = (masks[:, ::8, ::8] == z[f"predict_{route}_masks_sub"]).mean()
    note(f"lightning_predict_{route}_mask_disagreement", 1 - agree)
    assert agree > 0.9999                                                          # measured: identical on the sub-grid
    counts = np.stack([np.bincount(m.ravel(), minlength=5) for m in masks])
    assert np.abs(counts - z[f"predict_{route}_class_pixels"]).max() <= 400        # of 2.06 M pixels per frame
    h = pred.hist.cpu().numpy()
    meters = np.stack([h[0], h[1] + h[2] - h[0], h[2]])
    ref = z[f"predict_{route}_meters"]
    assert np.abs(meters - ref).max() <= 1e-3 * ref.max()
    got = np.array(pred.temporal_consistency())
    note(f"lightning_predict_{route}_miou_delta", float(np.abs(got - z[f"predict_{route}_summary"]).max()))
    assert np.abs(got - z[f"predict_{route}_summary"]).max() < 1e-4               # mIoU, mAcc, accuracy as wandb.summary holds them (measured 1.4e-7)
    rgb = colorize(torch.from_numpy(masks[7]).cuda()).cpu().numpy()               # the frame the reference hands to its video writer
    assert np.array_equal(rgb[::8, ::8], PALETTE[masks[7][::8, ::8]])
    # ---- test_step on three labelled items (real label maps), Florida / Texas meters, test_epoch_end's logs
    ev = FlowEvaluator(fm, classes=5, crop=crop)
    for k in range(3):
        lab = load_golden("label_pairs.npz")[("florida-05_49", "florida-07_29", "florida-04_27")[k]][:H].astype(np.int64)
        lab[:4] = 255
        mvl, mvr = synth.make_grids(5, 67, 120, seed=1320 + k, frame=(H, W), jitter=0.01)
        l, r = z["index"][k].tolist()
        ev.test_step({"frame_prev": clip[k:k + 1], "frame_next": clip[k + 1:k + 2], "mvs_left": cu(mvl), "mvs_right": cu(mvr),
                      "left_index": torch.tensor([l]), "right_index": torch.tensor([r]), "label": torch.from_numpy(lab)[None].cuda()},
                     test_idx=1 if k == 2 else 0)
    logged = []
    for meter, name in ((0, "meters1"), (1, "meters2")):
        h = ev.hist[meter].cpu().numpy()
        got_m, ref = np.stack([h[0], h[1] + h[2] - h[0], h[2]]), z[f"test_{route}_{name}"]
        assert np.array_equal(got_m[2], ref[2]) and np.abs(got_m - ref).max() <= 1e-3 * ref.max()
        logged += list(ev.summary(meter)[:3])
    logged.append((logged[0] + logged[3]) / 2)                                     # test_miou_epoch (base/foundation.py:254)
    assert np.abs(np.array(logged) - z[f"test_{route}_logged"]).max() < 1e-3
    assert np.abs(ev.summary(0)[3] - z[f"test_{route}_iou_classes1"]).max() < 2e-3


def test_hip_pspnet_inside_the_predict_step_and_test_step_chain_on_small_crops(psp_flow):
    """the Lightning-step pin above runs a toy torch-conv network (the reference's golden was made with it), so
    the HIP PSPNet on the crops route was only checked two hops away.  Here the REAL HIP PSPNet sits inside the same chain --
    predict_step x2 (65 x 65 sliding crops of a 160 x 272 frame: 4 x 6 crops, per-crop crop_motion_vector, batched network, fused
    tail + softmax canvas, resize + argmax, temporal-consistency meters across the windows) and test_step x2 -- against the oracle
    chain that test_oracle_golden.py pins to the reference's own predict_step / test_step, with pspnet_oracle as its network."""
    from flood_uav_video_segmentation_amd.flow.predict import FlowEvaluator, FlowPredictor
    from oracle import crops_oracle

    H, W, ch, cw, n = 160, 272, 65, 65, 5
    clip = synth.make_clip(11, (H, W), seed=1400, only=[0, 5, 10])
    # the synthetic classifier puts 99.99 % of such small crops into one class: re-centre its bias on the oracle's mean logits of a
    # few crops, so that the masks under comparison are mixed (checked below) and the comparison is not vacuous
    state = dict(psp_flow[1])
    probe = torch.cat([pspnet_oracle.decoder(pspnet_oracle.encoder(clip[i:i + 1, :, y:y + ch, x:x + cw].contiguous(), state, 50), state)
                       for i in range(3) for y, x in ((0, 0), (48, 100), (95, 207))])
    state["decoder.4.bias"] = state["decoder.4.bias"] - probe.mean((0, 2, 3))
    net = FlowPSPNet(HP()).eval()
    net.load_state_dict(state)
    fm = FlowModel(net, feature_based=False, no_warp=False).eval()
    pred = FlowPredictor(fm, classes=5, out_size=(H, W), crop=(ch, cw), compute_metrics=True)
    # (memoised on the crops' content: window 1's previous key frame is window 0's next one, and test_step sees the same frames again)
    seg = memo_by_content(lambda x: pspnet_oracle.decoder(pspnet_oracle.encoder(x, state, 50), state))
    oseg = lambda p, q, a, b: flow_oracle.predict_segmentation(lambda x: x, seg, p, q, a, b, n, False)["pred"]  # noqa: E731
    meters, last, agree = np.zeros((3, 5), np.int64), None, []
    for k in range(2):
        mvl, mvr = synth.make_grids(n, H // 16, W // 16, seed=1410 + k, frame=(H, W), jitter=0.02)
        got = pred.predict_window(clip[k:k + 1].cuda(), clip[k + 1:k + 2].cuda(), cu(mvl), cu(mvr))
        ref = np.asarray(flow_oracle.postprocess(crops_oracle.compute_output(oseg, n, clip[k:k + 1], clip[k + 1:k + 2], mvl, mvr, ch, cw, 5), (H, W)))
        assert got.shape == ref.shape == (n, H, W) and got.dtype == np.uint8
        assert np.bincount(ref.ravel(), minlength=5).max() < 0.7 * ref.size  # mixed masks
        agree.append((got == ref).mean())
        for p_ in range(n):
            prev = ref[p_ - 1] if p_ > 0 else last
            if prev is not None:
                meters += np.stack(flow_oracle.intersection_and_union(ref[p_][None], prev[None], 5, 255)).astype(np.int64)
        last = ref[n - 1]
    note("hip_pspnet_in_predict_step_chain_mask_disagreement", 1 - float(np.mean(agree)))
    assert min(agree) > 0.999
    h = pred.hist.cpu().numpy()
    got_m = np.stack([h[0], h[1] + h[2] - h[0], h[2]])
    assert np.abs(got_m - meters).max() <= 2e-3 * meters.max()
    want = np.array([np.mean(meters[0] / (meters[1] + 1e-10)), np.mean(meters[0] / (meters[2] + 1e-10)), meters[0].sum() / (meters[2].sum() + 1e-10)])
    assert note("hip_pspnet_in_predict_step_chain_miou_delta", float(np.abs(np.array(pred.temporal_consistency()) - want).max())) < 2e-3
    # test_step: one interpolated frame per labelled item from FlowModel.forward inside compute_output
    ev = FlowEvaluator(fm, classes=5, crop=(ch, cw))
    lab = np.random.default_rng(1420).integers(0, 5, (2, H, W)).astype(np.int64)
    lab[:, :3] = 255
    m_ref = np.zeros((3, 5), np.int64)
    for k, (l, r) in enumerate(((2, 3), (4, 1))):
        mvl, mvr = synth.make_grids(n, H // 16, W // 16, seed=1430 + k, frame=(H, W), jitter=0.02)
        ev.test_step({"frame_prev": clip[k:k + 1].cuda(), "frame_next": clip[k + 1:k + 2].cuda(), "mvs_left": cu(mvl), "mvs_right": cu(mvr),
                      "left_index": torch.tensor([l]), "right_index": torch.tensor([r]), "label": torch.from_numpy(lab[k:k + 1]).cuda()})
        fwd = lambda p, q, a, b: flow_oracle.forward(lambda x: x, seg, p, q, a, b, [l], [r], False, False)["pred"]  # noqa: E731
        out = crops_oracle.compute_output(fwd, 1, clip[k:k + 1], clip[k + 1:k + 2], mvl, mvr, ch, cw, 5)
        m_ref += np.stack(flow_oracle.intersection_and_union(out.max(1)[1].numpy(), lab[k:k + 1], 5, 255)).astype(np.int64)
    h = ev.hist[0].cpu().numpy()
    got_m = np.stack([h[0], h[1] + h[2] - h[0], h[2]])
    assert np.array_equal(got_m[2], m_ref[2]) and np.abs(got_m - m_ref).max() <= 2e-3 * m_ref.max()


def test_evaluator_validation_step_matches_the_references_with_a_batch_of_three():
    """FlowEvaluator.validation_step (FlowModel.forward at B = 3 with per-sample warp counts, fs_argmax_u8, fs_iou_hist) against the
    reference's own validation_step / validation_epoch_end over two batches (tests/golden/lightning_steps.npz)."""
    from flood_uav_video_segmentation_amd.flow.predict import FlowEvaluator

    z = load_golden("lightning_steps.npz")
    hv, wv = 160, 272
    vclip = synth.make_clip(6, (hv, wv), seed=1340).cuda()
    per = [synth.make_grids(5, hv // 16, wv // 16, seed=1341 + b, frame=(hv, wv), jitter=0.02) for b in range(3)]
    mvl = [torch.cat([per[b][0][j] for b in range(3)], 0).cuda() for j in range(4)]
    mvr = [torch.cat([per[b][1][j] for b in range(3)], 0).cuda() for j in range(4)]
    lab = np.random.default_rng(1345).integers(0, 5, (3, hv, wv)).astype(np.int64)
    lab[:, :3] = 255
    ev = FlowEvaluator(FlowModel(toy_model(), feature_based=False, no_warp=False).eval(), classes=5)
    for li, ri in (([1, 2, 4], [4, 3, 1]), ([3, 3, 2], [2, 2, 3])):
        ev.validation_step({"frame_prev": vclip[0:3], "frame_next": vclip[3:6], "mvs_left": mvl, "mvs_right": mvr,
                            "left_index": torch.tensor(li), "right_index": torch.tensor(ri), "label": torch.from_numpy(lab).cuda()})
    h = ev.hist["val"].cpu().numpy()
    meters, ref = np.stack([h[0], h[1] + h[2] - h[0], h[2]]), z["val_meters"]
    assert np.array_equal(meters[2], ref[2]) and np.abs(meters - ref).max() <= 1e-3 * ref.max()
    assert np.abs(np.array(ev.summary("val")[:3]) - z["val_logged"]).max() < 1e-3


def test_predict_step_mirror_masks_metric_and_palette(psp_flow):
    """flow/base.py:259-343 on the HIP path: 1072x1920 masks, temporal-consistency mIoU over two windows, palette."""
    from flood_uav_video_segmentation_amd.flow.predict import PALETTE, FlowPredictor, colorize

    net, _ = psp_flow
    n = 5
    keys = synth.make_clip(11, 713, seed=1000, only=[0, 5, 10]).cuda()
    dl, dr = synth.dummy_grids(n)
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    p = FlowPredictor(fm, classes=5, out_size=(1072, 1920))
    m0 = p.predict_window(keys[0:1], keys[1:2], cu(dl), cu(dr))
    m1 = p.predict_window(keys[1:2], keys[2:3], cu(dl), cu(dr))
    assert m0.shape == m1.shape == (5, 1072, 1920) and m0.dtype == np.uint8
    z = load_golden("predict_713.npz")
    assert (m0[:, ::4, ::4] == z["cfg2_post_mask_sub"]).mean() > 0.999  # the reference's own post-processed masks
    # temporal consistency against the oracle's metric on the same 10 masks (9 consecutive pairs)
    allm = np.concatenate([m0, m1])
    inter = union = target = 0
    for i in range(1, 10):
        a, u, t = flow_oracle.intersection_and_union(allm[i], allm[i - 1], 5, 255)
        inter, union, target = inter + a, union + u, target + t
    miou, macc, acc = p.temporal_consistency()
    assert abs(miou - np.mean(inter / (union + 1e-10))) < 1e-12
    assert abs(macc - np.mean(inter / (target + 1e-10))) < 1e-12 and abs(acc - inter.sum() / (target.sum() + 1e-10)) < 1e-12
    rgb = colorize(torch.from_numpy(m0[0]).cuda()).cpu().numpy()
    assert rgb.shape == (1072, 1920, 3) and np.array_equal(rgb, PALETTE[m0[0]])


def test_end_to_end_fake_video_through_dataset_predictor_and_grid_producer(tmp_path, psp_flow):
    """A tiny 'video' on disk -> PredictWindows (flow/dataset.py mirror) -> FlowPredictor (predict_step mirror):
    the host pieces either side of the hot path hand tensors of the contract shapes to each other."""
    import os

    from PIL import Image

    from flood_uav_video_segmentation_amd.flow import grids
    from flood_uav_video_segmentation_amd.flow.dataset import MEAN, STD, PredictWindows
    from flood_uav_video_segmentation_amd.flow.predict import FlowPredictor

    base = os.path.join(tmp_path, "frames", "vid")
    for d in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(base, d))
    rng = np.random.default_rng(3)
    mv = np.concatenate([np.full((400, 1), -1), np.full((400, 2), 16), rng.integers(0, 1900, (400, 4)), np.zeros((400, 3), int)], 1)
    for i in range(11):
        img = (rng.random((161, 225, 3)) * 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(base, "images", f"{i}.jpg"), quality=95)
        g, gi = grids.motion_vectors_to_grids(mv, 1072, 1920)
        grids.save_grid(os.path.join(base, "grids", f"{i}.npy"), g)
        grids.save_grid(os.path.join(base, "inv_grids", f"{i}.npy"), gi)
    ds = PredictWindows(str(tmp_path), "vid", frame_delta=5, no_warp=False)
    assert len(ds) == 2
    item = ds[0]
    assert item["frame_prev"].shape == (1, 3, 161, 225) and item["frame_prev"].dtype == torch.float32 and item["frame_id"] == 0
    assert len(item["mvs_left"]) == len(item["mvs_right"]) == 4 and item["mvs_left"][0].shape == (1, 67, 120, 2)
    ref = (torch.from_numpy(np.asarray(Image.open(os.path.join(base, "images", "0.jpg")))).permute(2, 0, 1).float()
           - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1)
    assert torch.allclose(item["frame_prev"][0].cpu(), ref, atol=1e-5)   # Normalize with mean/std * 255 (base/foundation.py:27-31)
    net, _ = psp_flow
    fm = FlowModel(net, feature_based=False, no_warp=False).eval()
    pred = FlowPredictor(fm, classes=5, out_size=(161, 225))
    for i in range(len(ds)):
        it = ds[i]
        masks = pred.predict_window(it["frame_prev"], it["frame_next"], it["mvs_left"], it["mvs_right"])
        assert masks.shape == (5, 161, 225) and masks.dtype == np.uint8 and masks.max() < 5
    miou, macc, acc = pred.temporal_consistency()
    assert 0.0 <= miou <= 1.0 and 0.0 <= acc <= 1.0


def test_pspnet_non_square_frame_with_full_frame_grids_against_oracle(psp_flow, profiler):
    """no_cropping=True route: non-square frame, non-divisible PPM windows (feature map 13x21), 67x120-style full-frame
    grids of a different size than the logits."""
    net, state = psp_flow
    n, H, W = 3, 97, 161
    clip = synth.make_clip(2, (H, W), seed=51)
    mvl, mvr = synth.make_grids(n, H // 16, W // 16, seed=52, frame=(H, W), jitter=0.04)
    fm = FlowModel(net, feature_based=False, no_warp=False).eval()
    out = fm.predict(clip[0:1].cuda(), clip[1:2].cuda(), cu(mvl), cu(mvr), n, profiler)["pred"]
    enc = lambda x: pspnet_oracle.encoder(x, state, 50)  # noqa: E731
    dec = lambda f: pspnet_oracle.decoder(f, state)  # noqa: E731
    ref = flow_oracle.predict_segmentation(enc, dec, clip[0:1], clip[1:2], mvl, mvr, n, False)["pred"]
    assert out.shape == ref.shape == (n, 5, H, W)
    assert rel_err(out.cpu(), ref) < NET_TOL


@pytest.mark.parametrize("crop", [None, (97, 97)])
@pytest.mark.parametrize("nw", [False, True])
def test_test_step_mirror_against_oracle_parity_unpinned(tmp_path, crop, nw):
    """labelled `test` split -> EvalWindows (flow/dataset.py:89-181) -> FlowEvaluator.test_step
    (flow/base.py:156-176; whole frame or compute_output + compute_test_crop) -> intersection / union / target meters.
    The CPU side restates the same chain with oracle/ (dataset indexing and cv2 resizes are unpinned: cv2 / skimage absent)."""
    import os

    from PIL import Image

    from flood_uav_video_segmentation_amd.flow.dataset import MEAN, STD, EvalWindows
    from flood_uav_video_segmentation_amd.flow.predict import FlowEvaluator
    from oracle import crops_oracle, dataset_oracle

    H, W, delta, K = 160, 272, 5, 5
    base = os.path.join(tmp_path, "frames", "florida")
    for d in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(base, d))
    os.makedirs(os.path.join(tmp_path, "masks"))
    missing = (7, 16)   # key frames of items 1, 2 and 3 (a missing IN-BETWEEN grid raises in the reference as well)
    clip = synth.make_clip(24, (H, W), seed=51)
    for i in range(24):
        img = ((clip[i].permute(1, 2, 0) * 40 + 128).clamp(0, 255)).to(torch.uint8).numpy()
        Image.fromarray(img).save(os.path.join(base, "images", f"{i}.jpg"), quality=95)
        if i in missing:
            continue
        ml, mr = synth.make_grids(2, 67, 120, seed=60 + i, frame=(H, W), jitter=0.01)
        np.save(os.path.join(base, "grids", f"{i}.npy"), ml[0][0].double().numpy())
        np.save(os.path.join(base, "inv_grids", f"{i}.npy"), mr[0][0].double().numpy())
    rng = np.random.default_rng(7)
    labelled = [4, 9, 12, 18]
    with open(os.path.join(tmp_path, "test.txt"), "w") as fh:
        for j, f in enumerate(labelled):
            lab = rng.integers(0, 6, (H + 8, W)).astype(np.uint8)     # 5 -> IgnoreClasses; taller than the frame -> nearest resize
            lab[:3] = 255                                             # ignore_index rows
            Image.fromarray(lab).save(os.path.join(tmp_path, "masks", f"{j}.png"))
            fh.write(f"masks/{j}.png florida {f}\n")
    ds = EvalWindows(str(tmp_path), os.path.join(tmp_path, "test.txt"), split="test", frame_delta=delta, no_warp=nw, size=(H, W),
                     classes_ignore=(5,))
    fm = FlowModel(toy_model(), feature_based=False, no_warp=nw).eval()
    ev = FlowEvaluator(fm, classes=K, crop=crop)

    w = toy_weights()
    enc = lambda x: torch.relu(torch.nn.functional.conv2d(x, w["enc_w"], w["enc_b"], 4, 1))  # noqa: E731
    dec = lambda f: torch.nn.functional.conv2d(f, w["dec_w"], w["dec_b"])  # noqa: E731
    have = lambda f: f not in missing and 0 <= f < 24  # noqa: E731
    default = torch.from_numpy(flow_oracle.get_default_grid()).float()[None]
    hist = np.zeros((3, K), np.int64)
    agree = []
    for i in range(len(ds)):
        item = ds[i]
        assert item["label"].shape == (1, H, W) and item["label"].dtype == torch.int64
        pred = ev.test_step(item, test_idx=0)
        # ---- CPU restatement of the same item
        l, r, prev_real, next_real, left, right = dataset_oracle.eval_item(have, i, labelled[i], delta, "test")
        assert int(item["left_index"][0]) == l and int(item["right_index"][0]) == r

        def frame(f):
            x = torch.from_numpy(np.asarray(Image.open(os.path.join(base, "images", f"{f}.jpg")))).permute(2, 0, 1).float()
            return ((x - torch.tensor(MEAN).view(3, 1, 1)) / torch.tensor(STD).view(3, 1, 1))[None]

        def grid(g, name):
            return default if g is None else torch.from_numpy(np.load(os.path.join(base, name, f"{g}.npy"))).float()[None]

        mvl = [torch.zeros(1, 1)] * (delta - 1) if nw else [grid(g, "grids") for g in left]
        mvr = [torch.zeros(1, 1)] * (delta - 1) if nw else [grid(g, "inv_grids") for g in right]
        fwd = lambda p, q, a, b: flow_oracle.forward(enc, dec, p, q, a, b, [l], [r], False, nw)["pred"]  # noqa: E731
        if crop is None:
            ref = fwd(frame(prev_real), frame(next_real), mvl, mvr).max(1)[1]
        else:
            ref = crops_oracle.compute_output(fwd, 1, frame(prev_real), frame(next_real), mvl, mvr, crop[0], crop[1], K).max(1)[1]
        raw = np.array(Image.open(os.path.join(tmp_path, "masks", f"{i}.png")))
        lab = dataset_oracle.ignore_classes(dataset_oracle.resize_label_nearest(raw, (H, W)), (5,))
        assert np.array_equal(item["label"][0].cpu().numpy(), lab)
        agree.append((pred.cpu() == ref.to(torch.uint8)).float().mean().item())
        # meters from the HIP masks with the oracle's histogram: isolates the metric path from fp-level mask flips
        a, u, t = flow_oracle.intersection_and_union(pred.cpu().long().numpy(), lab[None].astype(np.int64), K, 255)
        hist += np.stack([a, u, t]).astype(np.int64)
    assert min(agree) > 0.999
    h = ev.hist[0].cpu().numpy()
    inter, union, target = h[0], h[1] + h[2] - h[0], h[2]
    assert np.array_equal(inter, hist[0]) and np.array_equal(union, hist[1]) and np.array_equal(target, hist[2])
    miou, macc, acc, iou_c, acc_c = ev.summary(0)
    assert abs(miou - float(np.mean(hist[0] / (hist[1] + 1e-10)))) < 1e-12 and ev.summary(1) is None


def test_full_window_is_bit_repeatable_under_back_to_back_load(psp_flow):
    """configs[1] window at 713x713, 12 times back to back without host synchronisation in between: every repetition must
    give the same logits bit for bit (no race between launches, workspaces or direct-to-LDS stages)."""
    net, _ = psp_flow
    fm = FlowModel(net, feature_based=False, no_warp=True).eval()
    keys = synth.make_clip(6, 713, seed=1000, only=[0, 5]).cuda()
    dl, dr = [[g.cuda() for g in gs] for gs in synth.dummy_grids(5)]
    outs = [fm.predict(keys[0:1], keys[1:2], dl, dr, 5, None)["pred"] for _ in range(12)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_predict_video_entry_point_on_a_fake_video(tmp_path):
    """tools/predict_video.py = the role of the reference's predict_flow.sh run on the HIP path: frames + grids on disk ->
    colourised masks on disk + the temporal-consistency summary (flow/base.py:236-343)."""
    import os
    import subprocess
    import sys

    from PIL import Image

    from flood_uav_video_segmentation_amd.flow import grids

    base = os.path.join(tmp_path, "data", "frames", "vid")
    for d in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(base, d))
    rng = np.random.default_rng(4)
    mv = np.concatenate([np.full((300, 1), -1), np.full((300, 2), 16), rng.integers(0, 1900, (300, 4)), np.zeros((300, 3), int)], 1)
    clip = synth.make_clip(11, (161, 225), seed=71)
    for i in range(11):
        img = ((clip[i].permute(1, 2, 0) * 40 + 128).clamp(0, 255)).to(torch.uint8).numpy()
        Image.fromarray(img).save(os.path.join(base, "images", f"{i}.jpg"), quality=95)
        g, gi = grids.motion_vectors_to_grids(mv, 1072, 1920)
        grids.save_grid(os.path.join(base, "grids", f"{i}.npy"), g)
        grids.save_grid(os.path.join(base, "inv_grids", f"{i}.npy"), gi)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(tmp_path, "out")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "predict_video.py"), "--data-root", os.path.join(tmp_path, "data"),
                        "--video-id", "vid", "--frame-delta", "5", "--synthetic-weights", "--no-cropping", "--size", "161", "225",
                        "--out", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "10 frames of vid" in r.stdout and "temporal consistency mIoU" in r.stdout
    pngs = sorted(os.listdir(out), key=lambda n: int(n.split(".")[0]))
    assert pngs == [f"{i}.png" for i in range(10)]
    im = np.array(Image.open(os.path.join(out, "3.png")))
    assert im.shape == (161, 225, 3) and set(map(tuple, im.reshape(-1, 3))) <= set(map(tuple, __import__("flood_uav_video_segmentation_amd.flow.predict", fromlist=["PALETTE"]).PALETTE))


def test_test_flow_entry_point_on_a_fake_labelled_video(tmp_path):
    """tools/test_flow.py = the reference's `test` run (flow/base.py:156-176) without Lightning: list file + labels + frames +
    grids on disk -> mIoU line."""
    import os
    import subprocess
    import sys

    from PIL import Image

    H, W = 160, 272
    base = os.path.join(tmp_path, "frames", "florida")
    for d in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(base, d))
    os.makedirs(os.path.join(tmp_path, "masks"))
    clip = synth.make_clip(16, (H, W), seed=81)
    for i in range(16):
        img = ((clip[i].permute(1, 2, 0) * 40 + 128).clamp(0, 255)).to(torch.uint8).numpy()
        Image.fromarray(img).save(os.path.join(base, "images", f"{i}.jpg"), quality=95)
        ml, mr = synth.make_grids(2, 67, 120, seed=90 + i, frame=(H, W), jitter=0.01)
        np.save(os.path.join(base, "grids", f"{i}.npy"), ml[0][0].double().numpy())
        np.save(os.path.join(base, "inv_grids", f"{i}.npy"), mr[0][0].double().numpy())
    rng = np.random.default_rng(9)
    with open(os.path.join(tmp_path, "test.txt"), "w") as fh:
        for j, f in enumerate([5, 9]):
            Image.fromarray(rng.integers(0, 5, (H, W)).astype(np.uint8)).save(os.path.join(tmp_path, "masks", f"{j}.png"))
            fh.write(f"masks/{j}.png florida {f}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "test_flow.py"), "--data-root", str(tmp_path), "--list",
                        os.path.join(tmp_path, "test.txt"), "--frame-delta", "5", "--synthetic-weights", "--no-cropping", "--size", str(H),
                        str(W)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "test1: 2 labelled frames" in r.stdout and "mIoU" in r.stdout
