"""Host-side logic that needs no GPU: key canonicalisation, grids, synthetic generators, loud failures."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from flood_uav_video_segmentation_amd import ops, synth
from flood_uav_video_segmentation_amd.flow.model import FlowModel, get_default_grid
from flood_uav_video_segmentation_amd.model.deeplabv3 import FlowDeepLabv3
from flood_uav_video_segmentation_amd.model.pspnet import FlowPSPNet
from flood_uav_video_segmentation_amd.model.wrapper import ModelRepresentation


class HP:
    layers, classes, pretrained = 50, 5, False


def test_default_grid_is_bitwise_the_reference_grid():
    g = get_default_grid()
    assert g.dtype == np.float64 and g.shape == (67, 120, 2)
    assert np.array_equal(g, load_golden("default_grid.npz")["grid"])


def test_pspnet_alias_canonicalisation():
    c = FlowPSPNet.canonical_name
    assert c("layer3.4.conv2.weight") == "layer3.4.conv2.weight"
    assert c("layers.3.4.conv2.weight") == "layer3.4.conv2.weight"
    assert c("encoder.0.3.4.conv2.weight") == "layer3.4.conv2.weight"
    assert c("encoder.1.features.2.1.weight") == "ppm.features.2.1.weight"
    assert c("ppm.features.2.2.running_var") == "ppm.features.2.2.running_var"
    assert c("decoder.4.bias") == "decoder.4.bias"
    assert c("layer0.1.num_batches_tracked") is None
    assert c("aux.0.weight") is None  # training-only head
    names = set(synth.make_pspnet_state(50, 5, 0))
    assert all(c(k) == k for k in names)
    assert len(names) == 362 - 54 + 1 or len(names) > 300  # every conv/BN tensor of the inference path


def test_the_references_own_checkpoint_keys_are_all_consumed():
    """tests/golden/checkpoint_keys.json: names -> shapes of the state_dict the reference's OWN modules produce
    (FlowBaseModel.get_new_model_arch_G -> FlowModel(FlowPSPNet) under `model_G`, 1046 keys with every alias; VITSegmentModel under
    `model`, 185 keys).  Every key maps to a tensor the HIP network loads (same shape as the synthetic state) or is one of the
    known unused ones (num_batches_tracked; the ViT's classification head); every tensor the HIP network needs is covered."""
    import json
    import os

    from conftest import GOLDEN
    from flood_uav_video_segmentation_amd.model.vit import VITSegmentModel

    with open(os.path.join(GOLDEN, "checkpoint_keys.json")) as fh:
        ref = json.load(fh)
    assert len(ref["pspnet50"]) == 1046 and ref["pspnet50_optimizer_groups"] == {"head_modules": 2, "backbone_modules": 1}
    for keys, prefix, canon, state, unused in (
            (ref["pspnet50"], "model_G.model.", FlowPSPNet.canonical_name, synth.make_pspnet_state(50, 5, 0), ("num_batches_tracked",)),
            (ref["vit_b32"], "model.", VITSegmentModel.canonical_name, synth.make_vit_state(5, 704, seed=0), ("encoder.head.weight", "encoder.head.bias"))):
        covered = set()
        for k, shape in keys.items():
            assert k.startswith(prefix), k
            c = canon(k[len(prefix):])
            if c is None:
                assert k.endswith(unused), k
                continue
            assert c in state and list(state[c].shape) == shape, (k, c)
            covered.add(c)
        assert covered == set(state)


def test_deeplab_alias_canonicalisation():
    c = FlowDeepLabv3.canonical_name
    assert c("encoder.model.layer4.2.bn3.running_mean") == "backbone.layer4.2.bn3.running_mean"
    assert c("decoder.0.convs.4.1.weight") == "classifier.0.convs.4.1.weight"
    assert c("backbone.conv1.weight") == "backbone.conv1.weight"
    assert c("encoder.model.bn1.num_batches_tracked") is None


def test_flowmodel_attributes_match_the_reference_contract():
    net = FlowPSPNet(HP())
    fm = FlowModel(net, feature_based=False, no_warp=True, no_interpolation_percentage=0.25)
    assert fm.model is net and fm.feature_based is False and fm.no_warp is True and fm.no_interpolation_percentage == 0.25
    assert fm.default_motion_vector.shape == (1, 67, 120, 2) and fm.default_motion_vector.dtype == torch.float32
    assert "default_motion_vector" not in dict(fm.named_buffers())  # plain attribute, as in the reference
    for attr in ("encoder", "decoder", "layers", "ppm"):  # flow/base.py:96-101 builds optimiser groups from these
        assert list(getattr(net, attr).parameters()) == []
    x = torch.zeros(1, 5, 4, 4)
    assert fm.warp(x, None) is x  # no_warp: identity
    # the reference's public methods, same positional signatures (flow/model.py:24-249)
    import inspect

    want = {"forward": ["frame_current", "frame_prev", "frame_next", "mvs_left", "mvs_right", "left_index", "right_index"],
            "forward_feature": ["frame_prev", "frame_next", "mvs_left", "mvs_right", "left_index", "right_index", "n_list"],
            "forward_segmentation": ["frame_prev", "frame_next", "mvs_left", "mvs_right", "left_index", "right_index", "n_list"],
            "warp_batch": ["input", "mvs", "index_list", "n_list"],
            "predict_feature": ["frame_prev", "frame_next", "mvs_left", "mvs_right", "n", "profiler"],
            "predict_segmentation": ["frame_prev", "frame_next", "mvs_left", "mvs_right", "n", "profiler"],
            "warp": ["frame", "motion_vectors"]}
    for name, params in want.items():
        sig = inspect.signature(getattr(fm, name)).parameters
        assert list(sig)[:len(params)] == params, name
        # anything beyond the reference's positional signature is an optional keyword (key_cache)
        assert all(sig[k].default is not inspect.Parameter.empty for k in list(sig)[len(params):]), name


def test_pretrained_flag_is_refused_loudly():
    class P(HP):
        pretrained = True
    with pytest.raises(RuntimeError, match="pretrained"):
        FlowPSPNet(P())
    with pytest.raises(RuntimeError, match="pretrained"):
        FlowDeepLabv3(P())


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback():
    net = FlowPSPNet(HP())
    with pytest.raises(RuntimeError, match="HIP device is required"):
        net.load_state_dict(synth.make_pspnet_state(50, 5, 0))
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        ops.grid_sample(torch.zeros(1, 5, 4, 4), torch.zeros(1, 2, 2, 2))


def test_tensors_on_another_device_than_the_network_are_refused():
    """the library launches on the current device, so a tensor on cuda:1 with a network on cuda:0 must raise
    before any launch.  Device logic is plain Python: checked here with stand-in tensors (no GPU in this container)."""
    from flood_uav_video_segmentation_amd._lib import one_device

    class T:
        def __init__(self, dev, cuda=True):
            self.device, self.is_cuda = torch.device(dev), cuda

    assert one_device(T("cuda:1"), None, T("cuda:1")) == torch.device("cuda:1")
    with pytest.raises(RuntimeError, match="different devices"):
        one_device(T("cuda:0"), T("cuda:1"))
    with pytest.raises(RuntimeError, match="weights live on cuda:0"):
        one_device(T("cuda:1"), handle_device=torch.device("cuda:0"))
    with pytest.raises(RuntimeError, match="must live on the GPU"):
        one_device(T("cpu", cuda=False))


def test_hip_options_come_from_hparams_not_from_the_environment(monkeypatch):
    from flood_uav_video_segmentation_amd import _lib
    from flood_uav_video_segmentation_amd.model.hipnet import hip_options

    class O(HP):
        hip_no_winograd, hip_winograd_tile = True, 6
    monkeypatch.setenv("FS_NO_FUSED_HEAD", "1")  # a round-1 knob: must be ignored now
    net = FlowPSPNet(O())
    assert hip_options(O()) == dict(no_winograd=True, no_fused_winograd=False, no_split_bf16=False, no_res_touch=False, no_fused_pool=False, no_fused_qkv=False,
                                    no_fused_head=False, no_fused_shortcut=False, winograd_tile=6)
    assert net._hip_net.flags == _lib.OPT_NO_WINOGRAD and net._hip_net.winograd_tile == 6
    assert FlowPSPNet(HP())._hip_net.flags == 0

    class P32(HP):
        hip_no_split_bf16 = True  # round 3: the implicit GEMMs on the fp32-MFMA kernel instead of the split-operand one
    assert FlowPSPNet(P32())._hip_net.flags == _lib.OPT_NO_SPLIT_BF16

    class Retired(HP):
        hip_chain = True  # a route removed in round 6 (or any mistyped name): refused, never silently the defaults
    with pytest.raises(ValueError, match="unknown hip option"):
        FlowPSPNet(Retired())

    class DictLike(dict):  # Lightning hands hparams over as an AttributeDict: options are keys
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k) from None
    with pytest.raises(ValueError, match="unknown hip option"):
        hip_options(DictLike(layers=50, classes=5, hip_res_touch=True))
    assert hip_options(DictLike(layers=50, classes=5, hip_no_res_touch=True))["no_res_touch"] is True


def test_model_representation_eval_is_pass_through():
    inner = torch.nn.Identity()
    m = ModelRepresentation(inner, rep=None, rep_forward=None).eval()
    x = torch.ones(2)
    assert m(x) is x
    with pytest.raises(NotImplementedError):
        m.train()(x)


def test_synthetic_generators_are_deterministic_and_shaped():
    a, b = synth.make_clip(3, 65, seed=5), synth.make_clip(3, 65, seed=5)
    assert torch.equal(a, b) and a.shape == (3, 3, 65, 65) and a.dtype == torch.float32
    l1, r1 = synth.make_grids(5, 44, 44, seed=2000)
    l2, r2 = synth.make_grids(5, 44, 44, seed=2000)
    assert len(l1) == len(r1) == 4 and all(torch.equal(x, y) for x, y in zip(l1 + r1, l2 + r2))
    assert l1[0].shape == (1, 44, 44, 2) and l1[0].dtype == torch.float32
    ident = synth.identity_grid(44, 44)
    assert abs(ident[0, 0, 0] - ((8 / 704) * 2 - 1)) < 1e-12  # block centre of a 713-crop (flow/transform.py:226-227)
    dl, dr = synth.dummy_grids(5)
    assert len(dl) == len(dr) == 4  # the list length still encodes n (flow/base.py:266)
    s1, s2 = synth.make_pspnet_state(50, 5, 0), synth.make_pspnet_state(50, 5, 0)
    assert all(torch.equal(s1[k], s2[k]) for k in s1)
    assert s1["decoder.0.weight"].shape == (512, 4096, 3, 3) and s1["layer3.5.conv2.weight"].shape == (256, 256, 3, 3)


def _fake_video(root, n_frames, missing=()):
    import os

    from PIL import Image

    base = os.path.join(root, "frames", "vid")
    for d in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(base, d))
    for i in range(n_frames):
        Image.fromarray(np.full((8, 12, 3), i, np.uint8)).save(os.path.join(base, "images", f"{i}.jpg"))
        if i not in missing:
            np.save(os.path.join(base, "grids", f"{i}.npy"), np.full((67, 120, 2), i, np.float64))
            np.save(os.path.join(base, "inv_grids", f"{i}.npy"), np.full((67, 120, 2), -i, np.float64))


def test_predict_window_indexing_and_missing_frame_fallback(tmp_path):
    """flow/dataset.py:64, 112-146: windows = frames // delta; a key frame without grids slides forward (prev) / back (next);
    inverse grids come reversed."""
    from flood_uav_video_segmentation_amd.flow.dataset import PredictWindows

    _fake_video(str(tmp_path), 23, missing=(5, 10))
    w = PredictWindows(str(tmp_path), "vid", frame_delta=5, no_warp=False, device="cpu")
    assert len(w) == 23 // 5
    assert w.indices(0) == (0, 0, 4)       # next key 5 has no grids -> 4
    assert w.indices(1) == (5, 6, 9)       # prev key 5 -> 6 ; next key 10 -> 9
    assert w.indices(2) == (10, 11, 15)
    assert w.grid_ids(1) == ([6, 7, 8, 9], [9, 8, 7, 6])
    nw = PredictWindows(str(tmp_path), "vid", frame_delta=5, no_warp=True, device="cpu")
    assert nw.grid_ids(3) == ([16, 17, 18, 19], [19, 18, 17, 16])


def _fake_labelled_video(root, n_frames, labelled, missing=()):
    import os

    from PIL import Image

    _fake_video(root, n_frames, missing)
    os.rename(os.path.join(root, "frames", "vid"), os.path.join(root, "frames", "florida"))
    os.makedirs(os.path.join(root, "masks", "florida-01"))
    os.makedirs(os.path.join(root, "list"))
    rng = np.random.default_rng(5)
    lines = []
    for j, f in enumerate(labelled):
        lab = rng.integers(0, 7, (9, 14)).astype(np.uint8)   # ids 5, 6 exercise IgnoreClasses
        Image.fromarray(lab).save(os.path.join(root, "masks", "florida-01", f"{j}.png"))
        lines.append(f"masks/florida-01/{j}.png florida {f}")
    with open(os.path.join(root, "list", "test.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return os.path.join(root, "list", "test.txt")


def test_eval_window_plan_matches_the_restated_flowdata_indexing(tmp_path):
    """flow/dataset.py:16-43, 89-92, 115-171 (val / test split): seeded l/r split, missing key frames slide inward,
    grids at or before the real previous key / after the real next key become the identity grid, inverse grids reversed,
    both lists padded to frame_delta - 1.  Checked against oracle/dataset_oracle, which tests/test_oracle_golden.py pins to the
    reference's own FlowData (dataset_index.npz, round 4)."""
    import os

    from flood_uav_video_segmentation_amd.flow.dataset import EvalWindows, read_label_list
    from oracle import dataset_oracle

    labelled = [1, 12, 20, 33, 41, 57]
    missing = (9, 10, 22, 23, 24, 35, 43, 44)
    lst = _fake_labelled_video(str(tmp_path), 70, labelled, missing)
    have = lambda f: f not in missing and 0 <= f < 70  # noqa: E731
    for delta in (5, 8):
        ds = EvalWindows(str(tmp_path), lst, split="test", frame_delta=delta, device="cpu")
        ref_list = dataset_oracle.make_dataset(open(lst).read().splitlines(), delta)
        assert ds.label_list == ref_list == read_label_list(lst, delta)
        assert len(ds) == len(labelled) - 1   # frame 1 < delta // 2 is dropped
        for i, (_, _, f) in enumerate(ref_list):
            l, r, prev_real, next_real, left, right = dataset_oracle.eval_item(have, i, f, delta, "test")
            p = ds.plan(i)
            assert (p["l"], p["r"], p["prev_real"], p["next_real"]) == (l, r, prev_real, next_real)
            assert p["left_ids"] == left and p["right_ids"] == right
            assert len(left) == len(right) == delta - 1 and l + r == delta
    with open(os.path.join(tmp_path, "bad.txt"), "w") as fh:
        fh.write("masks/x.png florida\n")
    with pytest.raises(RuntimeError, match="Image list file read line error"):
        read_label_list(os.path.join(tmp_path, "bad.txt"), 5)
    with pytest.raises(ValueError):
        EvalWindows(str(tmp_path), lst, split="train")


def test_eval_window_label_transform_and_item_layout(tmp_path):
    """Resize(INTER_NEAREST) + IgnoreClasses on the label (flow/transform.py:104-105, 361-371); default-grid slots hold
    get_default_grid(); left_index / right_index arrive as 1-element tensors like the collated batch."""
    from flood_uav_video_segmentation_amd.flow.dataset import EvalWindows, resize_label_nearest
    from flood_uav_video_segmentation_amd.flow.model import get_default_grid
    from oracle import dataset_oracle

    rng = np.random.default_rng(6)
    lab = rng.integers(0, 5, (1080, 1920)).astype(np.uint8)
    for size in ((1072, 1920), (536, 960), (1080, 1920), (713, 713)):
        assert np.array_equal(resize_label_nearest(lab, size), dataset_oracle.resize_label_nearest(lab, size))
    lst = _fake_labelled_video(str(tmp_path), 30, [12], missing=(13,))
    ds = EvalWindows(str(tmp_path), lst, split="test", frame_delta=5, size=(8, 12), classes_ignore=(5, 6), device="cpu")
    p = ds.plan(0)
    # pure host path: labels and grids only (frames need the HIP resize -> covered by the GPU test)
    label = ds._label(ds.label_list[0][0])
    from PIL import Image
    raw = np.array(Image.open(tmp_path / "masks" / "florida-01" / "0.png"))
    want = dataset_oracle.ignore_classes(dataset_oracle.resize_label_nearest(raw, (8, 12)), (5, 6))
    assert label.dtype == torch.int64 and np.array_equal(label.numpy(), want) and label.max() <= 4
    assert None in p["left_ids"] + p["right_ids"]
    assert ds.default_grid.dtype == torch.float32 and np.array_equal(ds.default_grid.numpy(), get_default_grid().astype(np.float32))


def test_bench_quotes_pmc_traffic_only_for_the_running_build(tmp_path, monkeypatch):
    """roofline.traffic comes from the newest profiles/r*_pmc_traffic.json and only while that file was
    measured on THIS build (sha256 of the kernel sources); otherwise null + the reason."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    assert bench.kernel_symbol("igemm128x128") == "conv_igemm_dma_f32<128, 128, 2, 2, false, false>"
    assert bench.kernel_symbol("igemm128x64cat") == "conv_igemm_dma_f32<128, 64, 2, 2, true, false>"
    assert bench.kernel_symbol("split128x128") == "conv_igemm_dma_f32<128, 128, 4, 1, false, true>"      # the split-operand instantiations
    assert bench.kernel_symbol("split128x64cat") == "conv_igemm_dma_f32<128, 64, 4, 1, true, true>"
    assert bench.kernel_symbol("split128x96") == "conv_igemm_dma_f32<128, 96, 4, 1, false, true>"
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "build_id", lambda: "abc")
    assert bench.pmc_traffic("igemm128x128")[0] is None                      # nothing committed
    key = "void fs::conv_igemm_dma_f32<128, 128, 2, 2, false, false>(fs::ConvParams, int, int)"
    (prof / "r01_pmc_traffic.json").write_text(json.dumps({key: {"hbm_bytes_per_launch": 1.0}}))
    t, why = bench.pmc_traffic("igemm128x128")
    assert t is None and "stale" in why                                      # a file without build meta (round 1's format)
    (prof / "r03_pmc_traffic.json").write_text(json.dumps({"meta": {"build_id": "zzz", "git_head": "g"}, "kernels": {key: {"hbm_bytes_per_launch": 2.0}}}))
    t, why = bench.pmc_traffic("igemm128x128")
    assert t is None and "zzz" in why and "abc" in why                       # newest file, other build
    (prof / "r04_pmc_traffic.json").write_text(json.dumps({"meta": {"build_id": "abc", "git_head": "g"}, "kernels": {key: {"hbm_bytes_per_launch": 3.0}}}))
    t, why = bench.pmc_traffic("igemm128x128")
    assert t == 3 and "r04_pmc_traffic.json" in why
