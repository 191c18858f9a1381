#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python3 /root/repo/tests/golden/gen_goldens.py

Inputs/weights come from flood_uav_video_segmentation_amd.synth (seeded numpy PCG64), so tests regenerate
them bit-identically and only the reference's OUTPUTS are stored.  Nothing from the reference's source
is copied: the fixtures are arrays.
"""
import contextlib
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import flow.model as ref_flow  # reference
import model.pspnet as ref_psp  # reference
import util.util as ref_util  # reference

from flood_uav_video_segmentation_amd import synth

OUT = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.path.dirname(os.path.dirname(os.path.abspath(ref_flow.__file__)))
torch.manual_seed(0)
torch.set_grad_enabled(False)


class Prof:
    @contextlib.contextmanager
    def profile(self, name):
        yield


class HP:
    layers = 50
    pretrained = False
    classes = 5


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def toy_model(seed=3):
    g = torch.Generator().manual_seed(seed)
    m = nn.Module()
    enc = nn.Conv2d(3, 8, 3, stride=4, padding=1)
    dec = nn.Conv2d(8, 5, 1)
    for p in list(enc.parameters()) + list(dec.parameters()):
        p.data = torch.randn(p.shape, generator=g) * 0.5
    m.encoder = nn.Sequential(enc, nn.ReLU())
    m.decoder = dec
    return m.eval(), {"enc_w": enc.weight.data, "enc_b": enc.bias.data, "dec_w": dec.weight.data, "dec_b": dec.bias.data}


def gen_default_grid():
    g = ref_flow.get_default_grid()
    save("default_grid.npz", grid=g, sha256=np.frombuffer(hashlib.sha256(g.tobytes()).digest(), dtype=np.uint8))


def gen_ops_small():
    g = torch.Generator().manual_seed(21)
    fm = ref_flow.FlowModel(nn.Module(), feature_based=False, no_warp=False)
    x = torch.randn(1, 3, 7, 9, generator=g)
    grid = torch.rand(1, 4, 5, 2, generator=g) * 2.8 - 1.4  # out-of-range values exercise the border clamp
    grid64 = grid.double()  # the reference casts non-float grids (flow/model.py:246-247)
    save("ops_small.npz", x=x, grid=grid, warp=fm.warp(x, grid), warp_from_f64=fm.warp(x, grid64),
         up_ac=F.interpolate(x, size=(15, 20), mode="bilinear", align_corners=True),
         down_ac=F.interpolate(x, size=(4, 5), mode="bilinear", align_corners=True),
         ident_ac=F.grid_sample(x, torch.from_numpy(ref_flow.get_default_grid()).float()[None], padding_mode="border",
                                align_corners=True)[:, :, ::8, ::12])


def gen_toy_predict():
    toy, w = toy_model()
    arrays = dict(w)
    h, wd = 33, 41
    clip = synth.make_clip(2, (h, wd), seed=5)
    prev, nxt = clip[0:1], clip[1:2]
    arrays["prev"], arrays["next"] = prev, nxt
    for n in (3, 5):
        mvl, mvr = synth.make_grids(n, 4, 5, seed=40 + n, frame=(h, wd), jitter=0.05)
        for fb in (False, True):
            for nw in (False, True):
                fm = ref_flow.FlowModel(toy, feature_based=fb, no_warp=nw).eval()
                ml, mr = (synth.dummy_grids(n) if nw else (mvl, mvr))
                out = fm.predict(prev, nxt, ml, mr, n, Prof())["pred"]
                arrays[f"predict_n{n}_fb{int(fb)}_nw{int(nw)}"] = out
        fm = ref_flow.FlowModel(toy, feature_based=False, no_warp=False).eval()
        arrays[f"single_n{n}"] = fm.predict(prev, None, mvl, mvr, n, Prof())["pred"]
    # eval forward(): batch 3, mixed distances, grids [n-1][B,Hg,Wg,2]
    clip3 = synth.make_clip(6, (h, wd), seed=6)
    fp, fn_ = clip3[0:3], clip3[3:6]
    n = 5
    per = [synth.make_grids(n, 4, 5, seed=60 + b, frame=(h, wd), jitter=0.05) for b in range(3)]
    mvl = [torch.cat([per[b][0][j] for b in range(3)], 0) for j in range(n - 1)]
    mvr = [torch.cat([per[b][1][j] for b in range(3)], 0) for j in range(n - 1)]
    left, right = [1, 2, 4], [4, 3, 1]
    arrays["fwd_prev"], arrays["fwd_next"] = fp, fn_
    for fb in (False, True):
        for nw in (False, True):
            fm = ref_flow.FlowModel(toy, feature_based=fb, no_warp=nw).eval()
            out = fm(None, fp, fn_, mvl, mvr, torch.tensor(left), torch.tensor(right))["pred"]
            arrays[f"forward_fb{int(fb)}_nw{int(nw)}"] = out
    save("toy_predict.npz", **arrays)


def build_ref_pspnet(state, layers=50):
    hp = HP()
    hp.layers = layers
    net = ref_psp.FlowPSPNet(hparams=hp).eval()
    res = net.load_state_dict(state, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    bad = [k for k in res.missing_keys if not (k.startswith(("layers.", "encoder.")) or k.endswith("num_batches_tracked"))]
    assert not bad, bad
    return net


def stage_stats(net, x):
    stats = {}
    y = x
    for i, layer in enumerate([net.layer0, net.layer1, net.layer2, net.layer3, net.layer4]):
        y = layer(y)
        stats[f"layer{i}"] = np.array([y.double().mean().item(), y.double().abs().mean().item(), y.abs().max().item()])
    return stats, y


def gen_pspnet():
    state = synth.make_pspnet_state(50, 5, seed=0)
    net = build_ref_pspnet(state)
    # small: 65x65, batch 2 -> full tensors
    clip = synth.make_clip(2, 65, seed=100)
    feat = net.encoder(clip)
    logits = net.decoder(feat)
    stats, _ = stage_stats(net, clip)
    save("pspnet_small.npz", logits=logits, feat_slice=feat[:, ::128], feat_absmean=feat.double().abs().mean().item(),
         **{"stat_" + k: v for k, v in stats.items()})
    # 713: config-1 single frame + config-2 / config-3-style windows
    clip = synth.make_clip(6, 713, seed=1000)
    prev, nxt = clip[0:1], clip[5:6]
    feat = net.encoder(prev)
    lo = net.decoder(feat)
    stats, _ = stage_stats(net, prev)
    full = F.interpolate(lo, size=(713, 713), mode="bilinear", align_corners=True)
    mask = full.max(1)[1].to(torch.uint8)
    print("713 single-frame class histogram:", np.bincount(mask.numpy().ravel(), minlength=5),
          "logit range", lo.min().item(), lo.max().item())
    save("pspnet_713.npz", logits_lo=lo, mask=mask, feat_absmean=feat.double().abs().mean().item(),
         feat_slice=feat[:, ::256, ::6, ::6], **{"stat_" + k: v for k, v in stats.items()})
    n = 5
    fm = ref_flow.FlowModel(net, feature_based=False, no_warp=True).eval()
    dl, dr = synth.dummy_grids(n)
    out2 = fm.predict(prev, nxt, dl, dr, n, Prof())["pred"]
    mvl, mvr = synth.make_grids(n, 44, 44, seed=2000)
    fm = ref_flow.FlowModel(net, feature_based=False, no_warp=False).eval()
    out3 = fm.predict(prev, nxt, mvl, mvr, n, Prof())["pred"]
    m2 = out2.max(1)[1].to(torch.uint8)
    m3 = out3.max(1)[1].to(torch.uint8)
    post2 = F.interpolate(out2, (1072, 1920), mode="bilinear", align_corners=True).max(1)[1].to(torch.uint8)
    for f in range(n):
        print(f"frame {f}: cfg2 hist {np.bincount(m2[f].numpy().ravel(), minlength=5)} cfg3 hist {np.bincount(m3[f].numpy().ravel(), minlength=5)}")
    save("predict_713.npz", cfg2_mask=m2, cfg3_mask=m3, cfg2_logits_sub=out2[:, :, ::16, ::16], cfg3_logits_sub=out3[:, :, ::16, ::16],
         cfg2_post_mask_sub=post2[:, ::4, ::4])
    # metric fixture on real masks (util/util.py:36-47)
    tgt = m3[1].numpy().copy()
    tgt[:40] = 255
    ai, au, at = ref_util.intersectionAndUnion(m2[1].numpy(), tgt, 5, 255)
    save("metrics.npz", inter=ai, union=au, target=at)


def _timm_stand_ins():
    """timm is absent offline; the reference imports three timm symbols that do not take part in the eval forward
    (DropPath(0) == Identity, trunc_normal_ and _load_weights are initialisers/loaders): stand-ins for exactly those names."""
    import types

    if "timm" in sys.modules:
        return
    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    timm.models.layers = types.ModuleType("timm.models.layers")
    timm.models.vision_transformer = types.ModuleType("timm.models.vision_transformer")
    timm.models.layers.DropPath = lambda p=0.0: nn.Identity()
    timm.models.layers.trunc_normal_ = nn.init.trunc_normal_
    timm.models.vision_transformer._load_weights = lambda *a, **k: None
    for name, mod in (("timm", timm), ("timm.models", timm.models), ("timm.models.layers", timm.models.layers),
                      ("timm.models.vision_transformer", timm.models.vision_transformer)):
        sys.modules[name] = mod


def gen_vit():
    """VITSegmentModel (ViT-B/32 as model/vit.py hard-codes it)."""
    _timm_stand_ins()
    import model.vit as ref_vit  # reference

    state = synth.make_vit_state(5, 704, seed=0)
    net = ref_vit.VITSegmentModel(5, 704).eval()
    res = net.load_state_dict({"model." + k: v for k, v in state.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith("model.encoder.head") for k in res.missing_keys), res.missing_keys
    x704 = synth.make_clip(2, 704, seed=300)
    x713 = synth.make_clip(1, 713, seed=301)
    o704 = net(x704)["pred"]
    o713 = net(x713)["pred"]
    print("vit 704 hist", np.bincount(o704.max(1)[1].numpy().ravel(), minlength=5), "713 hist",
          np.bincount(o713.max(1)[1].numpy().ravel(), minlength=5))
    save("vit_b32.npz", pred704_sub=o704[:, :, ::8, ::8], mask704=o704.max(1)[1].to(torch.uint8)[:, ::2, ::2],
         pred713_sub=o713[:, :, ::8, ::8], mask713=o713.max(1)[1].to(torch.uint8)[:, ::2, ::2])


def gen_pspnet_feature():
    """A2 at the BASELINE size: FlowModel(FlowPSPNet, feature_based=True).predict at 713x713, warp and no_warp
    (flow/model.py:116-181: C = 4096 feature warps 90x90 -> 44x44, ONE decoder call on [5,4096,90,90]).  Same weights, key frames
    and grids as the segmentation-mode goldens of gen_pspnet()."""
    state = synth.make_pspnet_state(50, 5, seed=0)
    net = build_ref_pspnet(state)
    clip = synth.make_clip(6, 713, seed=1000)
    prev, nxt = clip[0:1], clip[5:6]
    n = 5
    out = {}
    for tag, no_warp in (("warp", False), ("nowarp", True)):
        mvl, mvr = synth.dummy_grids(n) if no_warp else synth.make_grids(n, 44, 44, seed=2000)
        fm = ref_flow.FlowModel(net, feature_based=True, no_warp=no_warp).eval()
        o = fm.predict(prev, nxt, mvl, mvr, n, Prof())["pred"]
        assert o.shape == (n, 5, 713, 713)
        m = o.max(1)[1].to(torch.uint8)
        print(f"feature {tag}: hist", np.bincount(m.numpy().ravel(), minlength=5), "range", o.min().item(), o.max().item())
        out[f"{tag}_logits_sub"] = o[:, :, ::16, ::16]
        out[f"{tag}_mask_sub"] = m[:, ::2, ::2]
    save("predict_feature_713.npz", **out)


def gen_pspnet_deep():
    """FlowPSPNet on ResNet-101 and ResNet-152 (model/pspnet.py:45-50) at 65x65: the deeper plans of the same executor."""
    out = {}
    for layers, seed in ((101, 1), (152, 2)):
        state = synth.make_pspnet_state(layers, 5, seed=seed)
        net = build_ref_pspnet(state, layers)
        x = synth.make_clip(1, 65, seed=8)
        out[f"logits{layers}"] = net.decoder(net.encoder(x))
        print(f"pspnet{layers} 65: range", out[f"logits{layers}"].min().item(), out[f"logits{layers}"].max().item())
    save("pspnet_deep_small.npz", **out)


def gen_pspnet_deep_713():
    """FlowPSPNet on ResNet-101 at the BASELINE frame size (713x713, one frame): the deeper plan at the geometry the configs use
    (23 layer3 blocks on the 90x90 map: every Winograd / fused-head route of the R50 network, more often)."""
    state = synth.make_pspnet_state(101, 5, seed=1)
    net = build_ref_pspnet(state, 101)
    x = synth.make_clip(6, 713, seed=1000)[0:1]
    feat = net.encoder(x)
    lo = net.decoder(feat)
    print("pspnet101 713: logits range", lo.min().item(), lo.max().item(), "|feat| mean", feat.double().abs().mean().item())
    save("pspnet101_713.npz", logits_lo=lo, feat_absmean=feat.double().abs().mean().item(), feat_slice=feat[:, ::256, ::6, ::6])


# The synthetic "dataset" the FlowData index fixtures are generated on; tests/test_oracle_golden.py rebuilds the same file set.
DATASET_FRAMES = 47                           # frames 0..46 of video "vid"
DATASET_MISSING_IMAGES = (10, 20, 21, 35)     # no <id>.jpg
DATASET_MISSING_GRIDS = (5, 30)               # no grids/<id>.npy
DATASET_MISSING_INV = (15, 16)                # no inv_grids/<id>.npy
DATASET_LABELLED = (1, 2, 4, 7, 11, 12, 19, 22, 23, 26, 31, 33, 36, 40, 44)  # labelled frame ids, in list order


def make_index_dataset(root):
    """Empty image files + tiny grid files whose VALUES are their frame ids, label PNG stand-ins, a 4-field list file."""
    for sub in ("images", "grids", "inv_grids"):
        os.makedirs(os.path.join(root, "frames", "vid", sub), exist_ok=True)
    os.makedirs(os.path.join(root, "labels"), exist_ok=True)
    for i in range(DATASET_FRAMES):
        if i not in DATASET_MISSING_IMAGES:
            open(os.path.join(root, "frames", "vid", "images", f"{i}.jpg"), "wb").close()
        if i not in DATASET_MISSING_GRIDS:
            np.save(os.path.join(root, "frames", "vid", "grids", f"{i}.npy"), np.full((2, 2, 2), i + 0.25))
        if i not in DATASET_MISSING_INV:
            np.save(os.path.join(root, "frames", "vid", "inv_grids", f"{i}.npy"), np.full((2, 2, 2), i + 0.5))
    with open(os.path.join(root, "list.txt"), "w") as fh:
        for f in DATASET_LABELLED:
            open(os.path.join(root, "labels", f"{f}.png"), "wb").close()
            fh.write(f"labels/{f}.png vid {f} x\n")   # four fields: make_dataset's check (flow/dataset.py:28)


def gen_dataset_index():
    """Index arithmetic of the reference's FlowData (flow/dataset.py:45-216) -- which key frames and grids make up item i of the
    predict / val / test splits, the forward / backward search for a complete key frame, the default-grid padding, make_dataset's
    filter -- run in THE REFERENCE'S OWN CLASS.  flow/dataset.py imports skimage.io for one call, io.imread (:217-218, :174);
    skimage is absent offline, so a stand-in module is registered whose imread returns an array filled with the NUMBER IN THE FILE
    NAME (the recipe of the timm stand-ins above: the stand-in decodes nothing and takes no part in the arithmetic under test; the
    frames' content is irrelevant to which frames are chosen).  transform=None, so flow/transform.py (cv2) is never touched."""
    import tempfile
    import types

    if "skimage" not in sys.modules:
        sk, sk_io = types.ModuleType("skimage"), types.ModuleType("skimage.io")
        sk_io.imread = lambda path: np.full((2, 2, 3), int(os.path.splitext(os.path.basename(path))[0]), dtype=np.int64)
        sk.io = sk_io
        sys.modules["skimage"], sys.modules["skimage.io"] = sk, sk_io
    import flow.dataset as ref_ds  # reference

    default = ref_flow.get_default_grid()
    out = {}
    with tempfile.TemporaryDirectory() as root, contextlib.redirect_stdout(open(os.devnull, "w")):
        make_index_dataset(root)

        def gid(g):  # grid array -> frame id, -1 = the identity default grid
            return -1 if g.shape == default.shape and np.array_equal(g, default) else int(g.flat[0])

        for delta in (5, 8, 25):
            ds = ref_ds.FlowData(split="predict", data_root=root, transform=None, frame_delta=delta, no_warp=False, predict_v_id="vid")
            rows = []
            for i in range(len(ds)):
                try:
                    it = ds[i]
                except FileNotFoundError:  # the window's grids are missing: np.load raises (the reference has no fallback in predict)
                    rows.append([i, -9, -9, -9] + [-9] * (2 * (delta - 1)))
                    continue
                rows.append([i, int(it["frame_prev"].flat[0]), int(it["frame_next"].flat[0]), it["frame_id"]] +
                            [gid(g) for g in it["mvs_left"]] + [gid(g) for g in it["mvs_right"]])
            out[f"predict_d{delta}"] = np.array(rows, dtype=np.int64)
            out[f"predict_len_d{delta}"] = np.array(len(ds))
            for split in ("val", "test"):
                ds = ref_ds.FlowData(split=split, type="l", data_root=root, data_list=os.path.join(root, "list.txt"), transform=None,
                                     frame_delta=delta, no_warp=False)
                rows = []
                for i in range(len(ds)):
                    try:
                        it = ds[i]
                    except FileNotFoundError:  # a grid BETWEEN the key frames is missing: only the key frames are searched for
                        rows.append([i, -9, -9, -9, -9, -9] + [-9] * (2 * (delta - 1)))
                        continue
                    rows.append([i, int(it["label"].flat[0]), int(it["frame_prev"].flat[0]), int(it["frame_next"].flat[0]),
                                 it["left_index"], it["right_index"]] + [gid(g) for g in it["mvs_left"]] + [gid(g) for g in it["mvs_right"]])
                out[f"{split}_d{delta}"] = np.array(rows, dtype=np.int64)
        # no_warp: placeholders whose COUNT encodes n (flow/dataset.py:198-205)
        ds = ref_ds.FlowData(split="predict", data_root=root, transform=None, frame_delta=5, no_warp=True, predict_v_id="vid")
        it = ds[1]
        out["nowarp_counts"] = np.array([len(it["mvs_left"]), len(it["mvs_right"]), int(it["mvs_left"][0].numel())])
    # the file set itself, so that a test can rebuild it without this module
    out.update(frames=np.array(DATASET_FRAMES), missing_images=np.array(DATASET_MISSING_IMAGES), missing_grids=np.array(DATASET_MISSING_GRIDS),
               missing_inv=np.array(DATASET_MISSING_INV), labelled=np.array(DATASET_LABELLED))
    for k, v in out.items():
        print(k, v.shape)
    save("dataset_index.npz", **out)


MV_FRAMES = ((1080, 1920, 9000, 11), (1072, 1920, 4000, 12), (720, 1280, 6000, 13), (1080, 1920, 0, 14))  # (H, W, vectors, seed)


def gen_mv_grids():
    """The grid producer (dataset/flow/extract_motion_vectors.py) run AS THE REFERENCE'S OWN SCRIPT, top to bottom: it is a script,
    so importing it opens the videos and writes frames/<video>/{grids,inv_grids}/<i>.npy.  Its two absent imports are the H.264
    decoder (mvextractor.videocap.VideoCap) and cv2 (one call, cv2.imwrite of the decoded frame); decoding is out of this
    repository's scope, so the stand-ins are a frame SOURCE -- VideoCap.read() hands out the seeded motion-vector rows of
    synth.motion_vectors with an empty frame of the stated size, then ret=False -- and an imwrite that writes nothing.  Neither takes
    part in the arithmetic under test (lines 21-43 and the np.save at :101-104), which the script performs itself."""
    import runpy
    import tempfile
    import types

    class SyntheticCap:
        def open(self, name):
            self.i = 0
            return True

        def read(self):
            if self.i >= len(MV_FRAMES):
                return False, None, None, None, None
            h, w, n, seed = MV_FRAMES[self.i]
            self.i += 1
            return True, np.zeros((h, w, 3), np.uint8), synth.motion_vectors(h, w, n, seed), "P", float(self.i)

        def release(self):
            pass

    cv2 = types.ModuleType("cv2")
    cv2.imwrite = lambda path, frame: open(path, "wb").close()   # the script re-extracts a frame whose .jpg is missing
    mvx, mvx_cap = types.ModuleType("mvextractor"), types.ModuleType("mvextractor.videocap")
    mvx_cap.VideoCap = SyntheticCap
    mvx.videocap = mvx_cap
    saved = {k: sys.modules.get(k) for k in ("cv2", "mvextractor", "mvextractor.videocap")}
    sys.modules.update({"cv2": cv2, "mvextractor": mvx, "mvextractor.videocap": mvx_cap})
    out = {}
    cwd, argv = os.getcwd(), sys.argv
    try:
        with tempfile.TemporaryDirectory() as root, contextlib.redirect_stdout(open(os.devnull, "w")):
            os.chdir(root)
            sys.argv = ["extract_motion_vectors.py", "synthetic.mp4"]
            runpy.run_path(os.path.join(REFERENCE, "dataset", "flow", "extract_motion_vectors.py"), run_name="__main__")
            for i in range(len(MV_FRAMES)):
                for kind in ("grids", "inv_grids"):
                    a = np.load(os.path.join(root, "frames", "synthetic", kind, f"{i}.npy"), allow_pickle=False)
                    assert a.dtype == np.float64 and a.shape == (67, 120, 2)
                    out[f"{kind}_{i}"] = a
            assert not os.path.exists(os.path.join(root, "frames", "synthetic", "grids", f"{len(MV_FRAMES)}.npy"))
    finally:
        os.chdir(cwd)
        sys.argv = argv
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    out["frames"] = np.array(MV_FRAMES, dtype=np.int64)
    save("mv_grids.npz", **out)


# (height, width, grid_h, grid_w, crop_h, crop_w, h_off, w_off): every window the ViT sliding-crop route cuts out of a 1072 x 1920
# frame (crop 704, stride ceil(704 * 2/3): flow/base.py:183-203), then offsets off the block edges, then two whose block
# quotients end in .5 (Python's round-half-to-even decides the range), then the small centre crop of transform_val below.
CROP_GEOMETRIES = [(1072, 1920, 67, 120, 704, 704, ho, wo) for ho in (0, 368) for wo in (0, 470, 940, 1216)] + [
    (1072, 1920, 67, 120, 704, 704, 100, 333), (1072, 1920, 67, 120, 704, 704, 8, 24), (1072, 1920, 67, 120, 704, 704, 24, 8),
    (160, 256, 10, 16, 96, 128, 32, 64)]
# the default route's 713 crops cut 45 blocks and resize them to 44: cv2 interpolates, the stand-in refuses -> NOT covered
CROP_GEOMETRIES_INTERPOLATED = [(1072, 1920, 67, 120, 713, 713, 0, 0), (1072, 1920, 67, 120, 713, 713, 359, 476)]
TRANSFORM_ITEM = dict(h=160, w=256, gh=10, gw=16, crop=(96, 128), delta=5, ignore=(5,), seed=77, frames=24, items=((10, 4), (20, 2)))


def _cv2_same_size_stand_in():
    """flow/transform.py imports cv2 (absent offline).  The stand-in knows ONE case of cv2.resize: the requested size is the
    size the array already has, for which cv2::resize copies its source (its early-out for dsize == ssize) -- so the stand-in
    returns src.copy() and takes no part in the arithmetic; any other call raises, and the generator records that geometry as
    not covered.  `collections.Iterable` (removed in Python 3.10, used by the reference's Resize / Crop constructors) is aliased to
    collections.abc.Iterable, which is what it was."""
    import collections
    import collections.abc
    import types

    class Interpolating(Exception):
        pass

    def resize(src, dsize, fx=None, fy=None, interpolation=None):
        if dsize is None or (dsize[1], dsize[0]) != tuple(src.shape[:2]):
            raise Interpolating(f"{src.shape[:2]} -> {dsize}")
        return src.copy()

    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.INTER_NEAREST, cv2.resize, cv2.Interpolating = 1, 0, resize, Interpolating
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable
    return cv2


def gen_transforms():
    """The reference's OWN flow/transform.py on the cases where its cv2.resize calls are same-size copies (see
    _cv2_same_size_stand_in): (a) crop_motion_vector (:215-261) -- block range by round(), renormalisation to the crop -- for every
    704-crop window of a 1072 x 1920 frame and offsets off the block edges, on float32 CPU tensors [1,67,120,2] (the branch
    compute_output feeds, flow/base.py:208); (b) transform_val = Resize, IgnoreClasses, Crop('center'), ToTensor, Normalize
    (flow/base.py:396-406) on one seeded item whose frames already have the Resize size."""
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2 = _cv2_same_size_stand_in()
    try:
        import flow.transform as ref_t  # reference
    finally:
        if saved is None:
            sys.modules.pop("cv2", None)
        else:
            sys.modules["cv2"] = saved
    out = {"geometries": np.array(CROP_GEOMETRIES, dtype=np.int64), "interpolated": np.array(CROP_GEOMETRIES_INTERPOLATED, dtype=np.int64)}
    for k, (h, w, gh, gw, ch, cw, ho, wo) in enumerate(CROP_GEOMETRIES):
        ml, mr = synth.make_grids(3, gh, gw, seed=300 + k, frame=(h, w), jitter=0.03)
        cl, cr = ref_t.crop_motion_vector([m.clone() for m in ml], [m.clone() for m in mr], h, w, ch, cw, ho, wo)
        assert all(tuple(m.shape) == (1, ch // 16, cw // 16, 2) and m.dtype == torch.float32 for m in cl + cr)
        out[f"crop_left_{k}"] = torch.cat(cl).numpy()
        out[f"crop_right_{k}"] = torch.cat(cr).numpy()
    for h, w, gh, gw, ch, cw, ho, wo in CROP_GEOMETRIES_INTERPOLATED:
        ml, mr = synth.make_grids(2, gh, gw, seed=1, frame=(h, w))
        try:
            ref_t.crop_motion_vector(ml, mr, h, w, ch, cw, ho, wo)
            raise SystemExit("expected an interpolating resize")
        except cv2.Interpolating:
            pass
    # (b) the validation / test transform chains on the two items of a tiny labelled video
    t = TRANSFORM_ITEM
    files = synth.transform_frames(t["h"], t["w"], t["gh"], t["gw"], range(t["frames"]), t["ignore"], t["seed"])
    ident = synth.identity_grid(t["gh"], t["gw"])                                                    # stands where FlowData pads (:147-171)
    mean, std = [0.485 * 255, 0.456 * 255, 0.406 * 255], [0.229 * 255, 0.224 * 255, 0.225 * 255]   # base/foundation.py:27-31 in [0, 255]
    val = ref_t.Compose([ref_t.Resize((t["h"], t["w"])), ref_t.IgnoreClasses(list(t["ignore"])),
                         ref_t.Crop(list(t["crop"]), crop_type="center", ignore_label=255), ref_t.ToTensor(), ref_t.Normalize(mean=mean, std=std)])
    test = ref_t.Compose([ref_t.Resize((t["h"], t["w"])), ref_t.IgnoreClasses(list(t["ignore"])), ref_t.ToTensor(),
                          ref_t.Normalize(mean=mean, std=std)])                                      # flow/base.py:410-431: no Crop
    n1 = t["delta"] - 1
    for k, (f, l) in enumerate(t["items"]):   # l = Random(index).randrange(1, delta), pinned by dataset_index.npz
        r = t["delta"] - l
        left = [files[g]["grid"] for g in range(f - l + 1, f + 1)]
        right = [files[g]["inv_grid"] for g in range(f + 1, f + r + 1)][::-1]
        left, right = left + [ident] * (n1 - len(left)), right + [ident] * (n1 - len(right))
        for name, chain in (("val", val), ("test", test)):
            _, fp, fn, mvl, mvr, lab = chain(None, files[f - l]["image"].copy(), files[f + r]["image"].copy(), [g.copy() for g in left],
                                             [g.copy() for g in right], files[f]["label"].copy())
            assert fp.dtype == torch.float32 and lab.dtype == torch.int64 and mvl[0].dtype == torch.float32
            out.update({f"{name}{k}_frame_prev": fp.numpy(), f"{name}{k}_frame_next": fn.numpy(), f"{name}{k}_label": lab.numpy(),
                        f"{name}{k}_mvs_left": torch.stack(mvl).numpy(), f"{name}{k}_mvs_right": torch.stack(mvr).numpy()})
    out.update(item_shape=np.array([t["h"], t["w"], t["gh"], t["gw"], t["crop"][0], t["crop"][1], t["delta"], t["seed"], t["frames"]]),
               item_ignore=np.array(t["ignore"]), items=np.array(t["items"]))
    save("transforms.npz", **out)


STEP_FRAME = (1072, 1920)          # the frame geometry predict_step hard-codes for its output (flow/base.py:275)
STEP_CROP = (704, 704)             # test_h / test_w of the sliding-crop runs: every window's cv2.resize is a same-size copy
STEP_KEYS = (0, 5, 10, 15)         # three consecutive predict windows of one clip (seed 1300)
STEP_LABELS = ("florida-05_49", "florida-07_29", "florida-04_27")   # tests/golden/label_pairs.npz: the reference's own label maps, rows [:1072]
STEP_INDEX = ((2, 3), (1, 4), (4, 1))  # (left_index, right_index) of the three test items


def gen_lightning_steps():
    """predict_step / on_predict_end and test_step / test_epoch_end run in THE REFERENCE'S OWN FlowBaseModel (flow/base.py:143-343,
    base/foundation.py:224-262, 333-344): compute_output's crop order, softmax, float64 canvas and count normalisation,
    crop_motion_vector per window, the hard-coded 1072 x 1920 upsample + argmax, the temporal-consistency meters across windows
    (last_output), the palette frames handed to the video writer, the Florida / Texas meter split and the epoch summaries.
    flow/base.py imports the training stack (pytorch_lightning, wandb, imageio, dataclasses_json, torchvision, skimage, cv2), all
    absent offline and none of them part of this arithmetic: import-only stand-ins are registered (LightningModule = an empty
    class, so the object is made with object.__new__ and given the attributes the methods read: hparams, model_G, trainer.profiler,
    logger; wandb.summary = a dict the reference writes its results into; imageio.get_writer = a collector of the frames the
    reference appends; cv2.resize = the same-size copy of _cv2_same_size_stand_in).  The network is the toy encoder / decoder of
    toy_predict.npz inside the reference's own FlowModel (logit warp), frames are 1072 x 1920 with 67 x 120 grids."""
    import tempfile
    import time
    import types

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Collector:
        def __init__(self):
            self.frames = []

        def append_data(self, frame):
            self.frames.append(np.array(frame))

        def close(self):
            pass

    class LightningModule:   # what FlowBaseModel's methods call on their base class
        def log(self, name, value, **kw):
            self.logged[name] = float(value)

        def test_epoch_end(self, outputs):
            return None

        def validation_epoch_end(self, outputs):
            return None

    class StepProfiler:      # trainer.profiler: context manager + recorded_durations (flow/base.py:321-323)
        def __init__(self):
            self.recorded_durations = {}

        @contextlib.contextmanager
        def profile(self, name):
            t0 = time.perf_counter()
            yield
            self.recorded_durations.setdefault(name, []).append(time.perf_counter() - t0)

    class OnDevice(torch.Tensor):
        """crop_motion_vector takes `m.cpu().numpy()[0]` and writes into it (flow/transform.py:245-248): for a CUDA tensor -- the
        reference's real runs -- .cpu() is a copy and the caller's grid is untouched, for a CPU tensor it is the tensor itself and
        every crop window would see the previous windows' renormalisation (SURVEY 8c quirk v).  The grids are handed over as this
        subclass, whose .cpu() returns a copy like a device tensor's: no arithmetic, the GPU semantics on a CPU-only machine."""

        def cpu(self, *a, **k):
            return torch.Tensor.cpu(self, *a, **k).clone().as_subclass(torch.Tensor)

    def on_device(grids):
        return [g.as_subclass(OnDevice) for g in grids]

    collector = Collector()
    saved = {k: sys.modules.get(k) for k in ("cv2", "wandb", "imageio", "pytorch_lightning", "pytorch_lightning.profilers", "dataclasses_json",
                                             "torchvision", "skimage", "skimage.io")}
    sys.modules["cv2"] = _cv2_same_size_stand_in()
    wandb = mod("wandb", run=object(), summary={})
    mod("imageio", get_writer=lambda *a, **k: collector)
    prof_mod = mod("pytorch_lightning.profilers", PyTorchProfiler=type("PyTorchProfiler", (), {}))
    mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=type("LightningDataModule", (), {}), profilers=prof_mod)
    mod("dataclasses_json", dataclass_json=lambda cls: cls)
    mod("torchvision")
    if "skimage" not in sys.modules or not hasattr(sys.modules["skimage"], "io"):
        mod("skimage", io=mod("skimage.io", imread=None))
    try:
        import flow.base as ref_base  # reference
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    from PIL import Image

    toy, _ = toy_model()
    H, W = STEP_FRAME
    clip = synth.make_clip(16, (H, W), seed=1300, only=list(STEP_KEYS))
    colors = np.loadtxt(os.path.join(REFERENCE, "dataset", "flow", "list", "colors.txt")).astype(np.uint8)
    out = {"keys": np.array(STEP_KEYS), "crop": np.array(STEP_CROP), "index": np.array(STEP_INDEX)}

    def make(no_cropping, log_dir):
        obj = object.__new__(ref_base.FlowBaseModel)
        obj.hparams = types.SimpleNamespace(no_cropping=no_cropping, test_h=STEP_CROP[0], test_w=STEP_CROP[1], classes=5, ignore_index=255,
                                            compute_metrics=True, save_images=False, save_video=True, predict_v_id="vid",
                                            data_root=os.path.join(REFERENCE, "dataset", "flow"))
        obj.logger = types.SimpleNamespace(log_dir=log_dir)
        obj.trainer = types.SimpleNamespace(profiler=StepProfiler())
        obj.model_G = ref_flow.FlowModel(toy, feature_based=False, no_warp=False).eval()
        obj.logged = {}
        return obj

    with tempfile.TemporaryDirectory() as tmp:
        for route, no_cropping in (("whole", True), ("crops", False)):
            # ---- predict_step over three consecutive windows
            obj = make(no_cropping, tmp)
            collector.frames.clear()
            wandb.summary.clear()
            obj.on_predict_start()
            assert obj.video_writer is collector
            for k in range(3):
                mvl, mvr = synth.make_grids(5, 67, 120, seed=1310 + k, frame=(H, W), jitter=0.01)
                obj.predict_step({"frame_prev": clip[k:k + 1], "frame_next": clip[k + 1:k + 2], "mvs_left": on_device(mvl), "mvs_right": on_device(mvr),
                                  "frame_id": torch.tensor([5 * k])}, k)
            obj.on_predict_end()
            frames = np.stack(collector.frames)                                   # [15,1072,1920,3] palette frames
            assert frames.shape == (15, H, W, 3)
            ids = np.zeros(frames.shape[:3], np.uint8)
            for c in range(5):
                ids[(frames == colors[c]).all(-1)] = c
            assert np.array_equal(colors[ids], frames)
            out[f"predict_{route}_masks_sub"] = ids[:, ::8, ::8].copy()
            out[f"predict_{route}_class_pixels"] = np.stack([np.bincount(m.ravel(), minlength=5) for m in ids]).astype(np.int64)
            out[f"predict_{route}_meters"] = np.stack([obj.intersection_meter_predict.sum, obj.union_meter_predict.sum, obj.target_meter_predict.sum]).astype(np.int64)
            out[f"predict_{route}_summary"] = np.array([wandb.summary["predict_miou1_epoch"], wandb.summary["predict_macc1_epoch"],
                                                        wandb.summary["predict_accuracy1_epoch"]], dtype=np.float64)
            out[f"predict_{route}_iou_classes"] = np.asarray(wandb.summary["predict_miou1_epoch_classes"], dtype=np.float64)
            assert len(obj.trainer.profiler.recorded_durations["predict_interference"]) == 3
            # ---- test_step on three labelled items (two in the Florida meters, one in the Texas meters) + test_epoch_end
            obj = make(no_cropping, tmp)
            obj.init_metrics_test()
            wandb.summary.clear()
            for k in range(3):
                with np.load(os.path.join(OUT, "label_pairs.npz")) as lz:
                    lab = lz[STEP_LABELS[k]][:H].astype(np.int64)
                lab[:4] = 255
                mvl, mvr = synth.make_grids(5, 67, 120, seed=1320 + k, frame=(H, W), jitter=0.01)
                l, r = STEP_INDEX[k]
                batch = {"frame_prev": clip[k:k + 1], "frame_next": clip[k + 1:k + 2], "mvs_left": on_device(mvl), "mvs_right": on_device(mvr),
                         "left_index": torch.tensor([l]), "right_index": torch.tensor([r]), "label": torch.from_numpy(lab)[None]}
                obj.test_step((batch, 1 if k == 2 else 0), k)
            out[f"test_{route}_meters1"] = np.stack([obj.intersection_meter_test1.sum, obj.union_meter_test1.sum, obj.target_meter_test1.sum]).astype(np.int64)
            out[f"test_{route}_meters2"] = np.stack([obj.intersection_meter_test2.sum, obj.union_meter_test2.sum, obj.target_meter_test2.sum]).astype(np.int64)
            obj.test_epoch_end([])
            out[f"test_{route}_logged"] = np.array([obj.logged[k] for k in ("test_miou1_epoch", "test_macc1_epoch", "test_accuracy1_epoch",
                                                                             "test_miou2_epoch", "test_macc2_epoch", "test_accuracy2_epoch",
                                                                             "test_miou_epoch")], dtype=np.float64)
            out[f"test_{route}_iou_classes1"] = np.asarray(wandb.summary["test_miou1_epoch_classes"], dtype=np.float64)
            print(route, out[f"predict_{route}_summary"], out[f"test_{route}_logged"])
        # ---- validation_step with a BATCH of three items (per-sample warp counts, flow/model.py:92-106) + validation_epoch_end
        hv, wv = 160, 272
        obj = make(True, tmp)
        obj.init_metrics_val()
        vclip = synth.make_clip(6, (hv, wv), seed=1340)
        per = [synth.make_grids(5, hv // 16, wv // 16, seed=1341 + b, frame=(hv, wv), jitter=0.02) for b in range(3)]
        mvl = [torch.cat([per[b][0][j] for b in range(3)], 0) for j in range(4)]
        mvr = [torch.cat([per[b][1][j] for b in range(3)], 0) for j in range(4)]
        lab = torch.from_numpy(np.random.default_rng(1345).integers(0, 5, (3, hv, wv))).long()
        lab[:, :3] = 255
        for rep in range(2):   # two batches: the meters accumulate over the epoch
            li, ri = (torch.tensor([1, 2, 4]), torch.tensor([4, 3, 1])) if rep == 0 else (torch.tensor([3, 3, 2]), torch.tensor([2, 2, 3]))
            obj.validation_step({"frame_prev": vclip[0:3], "frame_next": vclip[3:6], "mvs_left": mvl, "mvs_right": mvr, "left_index": li,
                                 "right_index": ri, "label": lab}, rep)
        out["val_meters"] = np.stack([obj.intersection_meter_val.sum, obj.union_meter_val.sum, obj.target_meter_val.sum]).astype(np.int64)
        obj.validation_epoch_end([])
        out["val_logged"] = np.array([obj.logged[k] for k in ("val_miou_epoch", "val_macc_epoch", "val_accuracy_epoch")], dtype=np.float64)
        print("val", out["val_logged"])
    save("lightning_steps.npz", **out)


def gen_checkpoint_keys():
    """Names and shapes of a Lightning checkpoint's state_dict as the reference's own modules produce them: FlowBaseModel.
    get_new_model_arch_G (flow/base.py:87-108) for arch = pspnet (layers 50), i.e. FlowModel(FlowPSPNet) under the attribute
    model_G, and VITSegmentModel (model/vit.py) under `model` -- every alias the modules register (layer0..4, layers.*, encoder.*,
    ppm, decoder, aux heads, num_batches_tracked).  A JSON of names -> shapes: the drop-in loaders are tested against it."""
    import json
    import types

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    names = ("cv2", "wandb", "imageio", "pytorch_lightning", "pytorch_lightning.profilers", "dataclasses_json", "torchvision", "skimage", "skimage.io")
    saved = {k: sys.modules.get(k) for k in names}
    sys.modules["cv2"] = _cv2_same_size_stand_in()
    mod("wandb", run=None, summary={})
    mod("imageio")
    prof_mod = mod("pytorch_lightning.profilers", PyTorchProfiler=type("PyTorchProfiler", (), {}))
    mod("pytorch_lightning", LightningModule=type("LightningModule", (), {}), LightningDataModule=type("LightningDataModule", (), {}), profilers=prof_mod)
    mod("dataclasses_json", dataclass_json=lambda cls: cls)
    mod("torchvision")
    if "skimage" not in sys.modules or not hasattr(sys.modules["skimage"], "io"):
        mod("skimage", io=mod("skimage.io", imread=None))
    try:
        import flow.base as ref_base  # reference
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    obj = object.__new__(ref_base.FlowBaseModel)
    obj.hparams = types.SimpleNamespace(arch="pspnet", layers=50, classes=5, pretrained=False, test_h=713, test_w=713, feature_based=False,
                                        no_warp=True, no_interpolation_percentage=0.0)
    model_G, heads, backs = obj.get_new_model_arch_G()
    out = {"pspnet50": {"model_G." + k: list(v.shape) for k, v in model_G.state_dict().items()},
           "pspnet50_optimizer_groups": {"head_modules": len(heads), "backbone_modules": len(backs)}}
    _timm_stand_ins()
    import model.vit as ref_vit  # reference

    vit = ref_vit.VITSegmentModel(5, 704)
    out["vit_b32"] = {"model." + k: list(v.shape) for k, v in vit.state_dict().items()}
    path = os.path.join(OUT, "checkpoint_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("checkpoint_keys.json", {k: len(v) for k, v in out.items()}, f"{os.path.getsize(path) / 1024:.1f} KiB")


LABEL_PAIRS = ("florida-05/49", "florida-07/29", "florida-04/27")   # the three smallest pairs that hold every class


def gen_label_pairs():
    """Known-answer vectors the reference ships as DATA: dataset/flow/masks/<video>/<i>.png (uint8 classes 0..4 at 1080 x 1920) and
    dataset/flow/masks_color/<video>/<i>.png, the same frame through the palette dataset/flow/list/colors.txt -- input and expected
    output of the palette lookup at flow/base.py:308-312 (`colors[output]`), and realistic label maps for the metric / label-transform
    tests.  All 314 pairs of the reference satisfy colors[mask] == masks_color exactly (checked here); three are decoded and stored
    as arrays (label_pairs.npz): labels as uint8 [1080,1920], colour images as uint8 [1080,1920,3]."""
    import glob

    from PIL import Image

    root = os.path.join(REFERENCE, "dataset", "flow")
    colors = np.loadtxt(os.path.join(root, "list", "colors.txt")).astype(np.uint8)
    n = 0
    for m in sorted(glob.glob(os.path.join(root, "masks", "*", "*.png"))):
        lab = np.array(Image.open(m))
        rgb = np.array(Image.open(m.replace(os.sep + "masks" + os.sep, os.sep + "masks_color" + os.sep)).convert("RGB"))
        assert np.array_equal(colors[lab], rgb), m
        n += 1
    out = {"pairs_checked": np.array(n), "colors": colors}
    for pair in LABEL_PAIRS:
        name = pair.replace("/", "_")
        out[name] = np.array(Image.open(os.path.join(root, "masks", pair + ".png")))
        out[name + "_color"] = np.array(Image.open(os.path.join(root, "masks_color", pair + ".png")).convert("RGB"))
        assert out[name].shape == (1080, 1920) and out[name].dtype == np.uint8 and sorted(np.unique(out[name])) == [0, 1, 2, 3, 4]
    print("label pairs checked:", n)
    save("label_pairs.npz", **out)


def gen_vit_s16():
    """BASELINE configs[3] names a ViT-S/16; model/vit.py hard-codes B/32 (patch 32, d_model 768), so the S/16 network is
    assembled from THE REFERENCE'S OWN CLASSES exactly as model/vit.py:24-52 assembles them, with S/16 numbers (patch 16,
    d_model 384 -> 6 heads, 12 + 2 layers; segm/config.yml:52-59).  Same weights / frames as tests/test_gpu_fullsize.py."""
    _timm_stand_ins()
    import segm.model.decoder as ref_dec  # reference
    import segm.model.segmenter as ref_seg  # reference
    import segm.model.vit as ref_enc  # reference

    d, patch, size = 384, 16, 704
    encoder = ref_enc.VisionTransformer(image_size=(size, size), patch_size=patch, n_layers=12, d_model=d, d_ff=4 * d, n_heads=d // 64,
                                        n_cls=5, dropout=0.1, drop_path_rate=0.0, distilled=False, channels=3)
    decoder = ref_dec.MaskTransformer(n_cls=5, patch_size=encoder.patch_size, d_encoder=d, n_layers=2, n_heads=d // 64, d_model=d,
                                      d_ff=4 * d, drop_path_rate=0.0, dropout=0.1)
    net = ref_seg.Segmenter(encoder, decoder, n_cls=5).eval()
    state = synth.make_vit_state(5, size, patch, d, 12, 2, seed=3)
    res = net.load_state_dict(state, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith("encoder.head") for k in res.missing_keys), res.missing_keys
    out = {}
    for sz in (704, 713):
        x = synth.make_clip(2, sz, seed=310 + sz)[0:1]
        o = net(x)
        assert o.shape == (1, 5, sz, sz)
        print(f"vit-s16 {sz} hist", np.bincount(o.max(1)[1].numpy().ravel(), minlength=5), "range", o.min().item(), o.max().item())
        out[f"pred{sz}_sub"] = o[:, :, ::8, ::8]
        out[f"mask{sz}"] = o.max(1)[1].to(torch.uint8)[:, ::2, ::2]
    save("vit_s16.npz", **out)


def gen_deeplab_backbone():
    """A7, the part the reference's own code can pin: torchvision is absent (and the reference fetches deeplabv3_resnet101 over
    torch.hub, model/deeplabv3.py:15), but the ENCODER of FlowDeepLabv3 is a torchvision-style ResNet-101 -- 7x7 stem, v1.5
    bottlenecks (stride on the 3x3) -- and the reference ships exactly that network as its own code: model/resnet.py:99-165 with
    deep_base=False, Bottleneck :60-96.  It is assembled here from those classes, avgpool / fc dropped (forward re-stated as the
    module sequence of :147-156), with torchvision's replace_stride_with_dilation=[False, True, True] schedule applied to the
    modules the same way model/pspnet.py:55-64 applies PSPNet's: layer3 block 0 keeps dilation 1 and loses its stride, the other
    layer3 blocks dilate by 2; layer4 block 0 dilates by 2, the rest by 4.  The ASPP head stays unpinned."""
    import model.resnet as ref_resnet  # reference

    net = ref_resnet.ResNet(ref_resnet.Bottleneck, [3, 4, 23, 3], deep_base=False).eval()
    for layer, first, rest in ((net.layer3, 1, 2), (net.layer4, 2, 4)):
        for bi, blk in enumerate(layer):
            d = first if bi == 0 else rest
            blk.conv2.stride, blk.conv2.dilation, blk.conv2.padding = (1, 1), (d, d), (d, d)
            if blk.downsample is not None:
                blk.downsample[0].stride = (1, 1)
    state = synth.make_deeplab_state(101, 5, 0)
    res = net.load_state_dict({k[len("backbone."):]: v for k, v in state.items() if k.startswith("backbone.")}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.startswith("fc.") or k.endswith("num_batches_tracked") for k in res.missing_keys), res.missing_keys

    def stages(x):
        out = {}
        y = net.maxpool(net.relu(net.bn1(net.conv1(x))))
        out["stem"] = y
        for i, layer in enumerate((net.layer1, net.layer2, net.layer3, net.layer4)):
            y = layer(y)
            out[f"layer{i + 1}"] = y
        return out

    arrays = {}
    small = stages(synth.make_clip(2, 97, seed=150))
    arrays["feat97_sub"] = small["layer4"][:, ::4]
    full = stages(synth.make_clip(6, 713, seed=1000)[0:1])
    f = full["layer4"]
    assert f.shape == (1, 2048, 90, 90), f.shape
    arrays["feat713_sub"] = f[:, ::32, ::3, ::3]
    for k, v in full.items():
        arrays["stat713_" + k] = np.array([v.double().mean().item(), v.double().abs().mean().item(), v.abs().max().item()])
    for k, v in small.items():
        arrays["stat97_" + k] = np.array([v.double().mean().item(), v.double().abs().mean().item(), v.abs().max().item()])
    print("deeplab backbone 713: |feat| mean", f.double().abs().mean().item(), "max", f.abs().max().item())
    save("deeplab_backbone.npz", **arrays)


if __name__ == "__main__":
    which = sys.argv[1:] or ["grid", "ops", "toy", "pspnet", "vit", "vit_s16", "pspnet_feature", "pspnet_deep", "deeplab_backbone", "pspnet_deep_713", "dataset_index", "mv_grids", "transforms", "lightning_steps", "checkpoint_keys", "label_pairs"]
    if "deeplab_backbone" in which:
        gen_deeplab_backbone()
    if "pspnet_deep_713" in which:
        gen_pspnet_deep_713()
    if "dataset_index" in which:
        gen_dataset_index()
    if "mv_grids" in which:
        gen_mv_grids()
    if "transforms" in which:
        gen_transforms()
    if "label_pairs" in which:
        gen_label_pairs()
    if "lightning_steps" in which:
        gen_lightning_steps()
    if "checkpoint_keys" in which:
        gen_checkpoint_keys()
    if "pspnet_deep" in which:
        gen_pspnet_deep()
    if "vit_s16" in which:
        gen_vit_s16()
    if "pspnet_feature" in which:
        gen_pspnet_feature()
    if "vit" in which:
        gen_vit()
    if "grid" in which:
        gen_default_grid()
    if "ops" in which:
        gen_ops_small()
    if "toy" in which:
        gen_toy_predict()
    if "pspnet" in which:
        gen_pspnet()
