"""Size-independent properties at the BASELINE sizes (713x713 logits, 90x90 features, 1072x1920 masks) and the
edge cases of the flow ops: empty tensors, ragged / unaligned shapes, out-of-range grids, ties, ignore labels.
The oracle (torch-CPU) cannot run the full sizes in seconds for every op, so these tests lean on identities the
arithmetic must satisfy exactly, plus a few sampled rows/pixels checked against torch on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from flood_uav_video_segmentation_amd import ops
from flood_uav_video_segmentation_amd.flow.predict import PALETTE, colorize

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


# ------------------------------------------------------------------------------------------------ full-size properties
def test_resize_to_same_size_is_the_identity_bitwise_at_713():
    """align_corners=True with in == out: src = dst exactly, lambda = 0 -> the op must return its input bit for bit."""
    x = _rand(5, 5, 713, 713, seed=1).cuda()
    assert torch.equal(ops.resize_bilinear(x, (713, 713), align_corners=True), x)
    f = _rand(1, 256, 90, 90, seed=2).cuda().contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.resize_bilinear(f, (90, 90), align_corners=True), f)


def test_resize_of_a_linear_ramp_is_the_same_ramp_at_713():
    """Bilinear interpolation reproduces affine functions: up(a*y + b*x + c) sampled at 713^2 == the ramp itself (fp tolerance)."""
    ys = torch.linspace(0, 1, 90).view(1, 1, 90, 1)
    xs = torch.linspace(0, 1, 90).view(1, 1, 1, 90)
    lo = (2.0 * ys - 3.0 * xs + 0.5).cuda()
    up = ops.resize_bilinear(lo, (713, 713), align_corners=True)
    want = (2.0 * torch.linspace(0, 1, 713).view(713, 1) - 3.0 * torch.linspace(0, 1, 713).view(1, 713) + 0.5).cuda()
    assert (up[0, 0] - want).abs().max().item() < 5e-6


def test_resize_is_linear_in_its_input_at_713():
    a, b = _rand(1, 5, 90, 90, seed=3).cuda(), _rand(1, 5, 90, 90, seed=4).cuda()
    lhs = ops.resize_bilinear(a + b, (713, 713))
    rhs = ops.resize_bilinear(a, (713, 713)) + ops.resize_bilinear(b, (713, 713))
    assert (lhs - rhs).abs().max().item() < 5e-6


def test_identity_grid_warp_equals_resize_with_matching_corner_convention():
    """grid_sample(x, identity grid of size (Hg,Wg), align_corners=True) == interpolate(x, (Hg,Wg), align_corners=True): the
    two formulas coincide when the grid holds linspace(-1, 1) (the key-frame feature path, flow/model.py:154-159)."""
    x = _rand(1, 64, 90, 90, seed=5).cuda().contiguous(memory_format=torch.channels_last)
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, 67), torch.linspace(-1, 1, 120), indexing="ij")
    grid = torch.stack((gx, gy), -1)[None].cuda()
    a = ops.grid_sample(x, grid, align_corners=True)
    b = ops.resize_bilinear(x, (67, 120), align_corners=True)
    # not bit-equal: grid_sample un-normalises ((g + 1) / 2) * (W - 1) in fp32 (ATen's formula, kept), a ~1e-5-pixel coordinate
    # error times a unit-variance gradient
    assert a.shape == b.shape == (1, 64, 67, 120) and (a - b).abs().max().item() < 2e-4


def test_grid_sample_border_clamp_for_far_out_of_range_and_huge_coordinates():
    """padding_mode='border': any coordinate beyond the frame samples the edge pixel; +-1e30 must not overflow the index math."""
    x = _rand(1, 5, 44, 44, seed=6).cuda()
    grid = torch.zeros(1, 4, 4, 2)
    grid[0, 0, :, 0], grid[0, 0, :, 1] = -5.0, -7.0          # top-left corner
    grid[0, 1, :, 0], grid[0, 1, :, 1] = 9.0, 11.0           # bottom-right corner
    grid[0, 2, :, 0], grid[0, 2, :, 1] = 1e30, -1e30         # top-right corner
    grid[0, 3, :, 0], grid[0, 3, :, 1] = -1e30, 1e30         # bottom-left corner
    out = ops.grid_sample(x, grid.cuda()).cpu()
    xc = x.cpu()
    for row, (iy, ix) in enumerate([(0, 0), (43, 43), (0, 43), (43, 0)]):
        assert torch.equal(out[0, :, row, :], xc[0, :, iy, ix][:, None].expand(5, 4))
    assert torch.equal(out, F.grid_sample(xc, grid, mode="bilinear", padding_mode="border", align_corners=False))


def test_blend_identities_at_full_logit_size():
    a = _rand(5, 5, 713, 713, seed=7).cuda()
    assert torch.equal(ops.blend(a, 1.0), a)                           # x * 1
    assert torch.equal(ops.blend(a, 0.5, a, 0.5), a)                   # 0.5a + 0.5a is exact in binary fp
    assert torch.equal(ops.blend(a, 0.25, a, 0.75), a * 0.25 + a * 0.75)
    z = ops.blend(a, 0.0)
    assert torch.equal(z, a * 0.0)                                     # signed zeros included


def test_argmax_of_one_hot_and_first_index_on_ties_at_1072x1920():
    lab = torch.randint(0, 5, (2, 1072, 1920), generator=torch.Generator().manual_seed(8))
    one_hot = F.one_hot(lab, 5).permute(0, 3, 1, 2).float().contiguous().cuda()
    assert torch.equal(ops.argmax_u8(one_hot).cpu(), lab.to(torch.uint8))
    ties = torch.zeros(1, 5, 64, 64).cuda()                           # all equal -> class 0 (torch.max returns the first index)
    assert int(ops.argmax_u8(ties).max()) == 0
    ties[:, 3:] = 1.0                                                  # classes 3 and 4 tie -> 3
    assert torch.equal(ops.argmax_u8(ties), torch.full((1, 64, 64), 3, dtype=torch.uint8, device="cuda"))


def test_resize_argmax_equals_resize_then_argmax_at_1072x1920():
    """The fused post-processing kernel (flow/base.py:275-276) must give what the two-step route gives, everywhere."""
    lo = _rand(2, 5, 713, 713, seed=9).cuda()
    fused = ops.resize_argmax_u8(lo, (1072, 1920))
    two_step = ops.argmax_u8(ops.resize_bilinear(lo, (1072, 1920), align_corners=True))
    assert torch.equal(fused, two_step)
    rows = [0, 1, 535, 1070, 1071]                                     # sampled rows against torch on the CPU
    ref = F.interpolate(lo.cpu(), (1072, 1920), mode="bilinear", align_corners=True)[:, :, rows].max(1)[1].to(torch.uint8)
    assert (fused.cpu()[:, rows] == ref).float().mean().item() > 0.9999


def test_iou_hist_bookkeeping_at_1072x1920():
    """intersection <= min(|pred|, |target|); sum(|target|) = #pixels not ignored; pred == target -> IoU 1 per present class;
    a checksum of checksums: accumulating two halves equals one pass over the whole."""
    g = torch.Generator().manual_seed(10)
    pred = torch.randint(0, 5, (5, 1072, 1920), generator=g, dtype=torch.uint8).cuda()
    tgt = torch.randint(0, 5, (5, 1072, 1920), generator=g, dtype=torch.uint8).cuda()
    tgt[:, :7] = 255
    h = ops.iou_hist(pred, tgt, 5).cpu().numpy()
    assert h.dtype == np.int64 and (h[0] <= np.minimum(h[1], h[2])).all()
    assert h[2].sum() == 5 * (1072 - 7) * 1920 == h[1].sum()           # ignored pixels leave both area histograms
    same = ops.iou_hist(tgt, tgt, 5).cpu().numpy()
    assert np.array_equal(same[0], same[1]) and np.array_equal(same[1], same[2])
    acc = ops.iou_hist(pred[:2], tgt[:2], 5)
    acc = ops.iou_hist(pred[2:], tgt[2:], 5, hist=acc).cpu().numpy()
    assert np.array_equal(acc, h)


def test_colorize_is_a_table_lookup():
    m = torch.randint(0, 5, (3, 1072, 1920), generator=torch.Generator().manual_seed(11), dtype=torch.uint8).cuda()
    rgb = colorize(m)
    assert rgb.shape == (3, 1072, 1920, 3) and rgb.dtype == torch.uint8
    assert torch.equal(rgb.cpu(), torch.from_numpy(PALETTE)[m.cpu().long()])


def test_colorize_and_metrics_on_the_label_pairs_the_reference_ships():
    """tests/golden/label_pairs.npz: three of the reference's own (label, colour label) pairs at 1080 x 1920.  fs_colorize on the label
    gives the reference's colour image byte for byte (flow/base.py:310 with dataset/flow/list/colors.txt); fs_iou_hist between real
    label maps equals the oracle's intersectionAndUnion (util/util.py:36-47 semantics, pinned by metrics.npz) -- realistic region
    shapes instead of uniform noise; the label-side transform chain of the test split (nearest resize 1080 -> 1072 rows,
    IgnoreClasses) equals the oracle's."""
    from conftest import load_golden
    from flood_uav_video_segmentation_amd.flow.dataset import resize_label_nearest
    from oracle import dataset_oracle, flow_oracle

    names = ("florida-05_49", "florida-07_29", "florida-04_27")
    z = load_golden("label_pairs.npz")
    labs = [z[n] for n in names]
    for n, lab in zip(names, labs):
        rgb = z[n + "_color"]
        assert torch.equal(colorize(torch.from_numpy(lab).cuda()).cpu(), torch.from_numpy(rgb))
        small = resize_label_nearest(lab, (1072, 1920))
        assert small.shape == (1072, 1920) and np.array_equal(small, dataset_oracle.resize_label_nearest(lab, (1072, 1920)))
    pred, tgt = np.stack([labs[0], labs[1], labs[2]]), np.stack([labs[1], labs[2], labs[0]])
    tgt = tgt.copy()
    tgt[:, :8] = 255
    h = ops.iou_hist(torch.from_numpy(pred).cuda(), torch.from_numpy(tgt).cuda(), 5).cpu().numpy()
    a, u, t = flow_oracle.intersection_and_union(pred.astype(np.int64), tgt.astype(np.int64), 5, 255)
    assert np.array_equal(h[0], a) and np.array_equal(h[1] + h[2] - h[0], u) and np.array_equal(h[2], t)
    assert (t > 0).all() and a.sum() > 0   # every class is present in the targets


def test_conv_is_linear_without_activation_at_layer_size():
    """1x1 conv 1024 -> 256 on the 90x90 map of a 713 frame: conv(x1 + x2) == conv(x1) + conv(x2) up to fp32 rounding, and a
    zero input gives exactly the shift."""
    x1 = _rand(1, 1024, 90, 90, seed=12).cuda().contiguous(memory_format=torch.channels_last)
    x2 = _rand(1, 1024, 90, 90, seed=13).cuda().contiguous(memory_format=torch.channels_last)
    w = (_rand(256, 1024, 1, 1, seed=14) * 0.03).cuda()
    shift = _rand(256, seed=15).cuda()
    y12 = ops.conv2d_nhwc(x1 + x2, w)
    y1, y2 = ops.conv2d_nhwc(x1, w), ops.conv2d_nhwc(x2, w)
    assert ((y12 - (y1 + y2)).abs().max() / y12.abs().max()).item() < 2e-5
    y0 = ops.conv2d_nhwc(torch.zeros_like(x1), w, shift=shift)
    assert torch.equal(y0, shift.view(1, 256, 1, 1).expand(1, 256, 90, 90))


# ------------------------------------------------------------------------------------------------ empty / ragged inputs
def test_empty_batches_are_no_ops_not_faults():
    e = torch.empty(0, 5, 13, 17, device="cuda")
    assert ops.resize_bilinear(e, (29, 31)).shape == (0, 5, 29, 31)
    assert ops.grid_sample(e, torch.empty(0, 4, 4, 2, device="cuda")).shape == (0, 5, 4, 4)
    assert ops.blend(e, 0.5, e, 0.5).shape == (0, 5, 13, 17)
    assert ops.argmax_u8(e).shape == (0, 13, 17)
    assert ops.resize_argmax_u8(e, (20, 20)).shape == (0, 20, 20)
    h = ops.iou_hist(torch.empty(0, dtype=torch.uint8, device="cuda"), torch.empty(0, dtype=torch.uint8, device="cuda"), 5)
    assert int(h.sum()) == 0
    assert colorize(torch.empty(0, 8, 8, dtype=torch.uint8, device="cuda")).shape == (0, 8, 8, 3)
    torch.cuda.synchronize()


@pytest.mark.parametrize("shape,size", [((1, 5, 1, 1), (7, 9)), ((2, 3, 1, 17), (5, 1)), ((1, 7, 13, 1), (1, 1)), ((3, 1, 2, 3), (713, 2))])
def test_degenerate_one_pixel_shapes_match_torch(shape, size):
    """1-pixel sources / targets: align_corners=True divides by (out - 1) -> the reference (ATen) uses scale 0 there."""
    x = _rand(*shape, seed=16)
    got = ops.resize_bilinear(x.cuda(), size, align_corners=True).cpu()
    assert torch.allclose(got, F.interpolate(x, size, mode="bilinear", align_corners=True), atol=1e-6, rtol=0)
    got = ops.resize_bilinear(x.cuda(), size, align_corners=False).cpu()
    assert torch.allclose(got, F.interpolate(x, size, mode="bilinear", align_corners=False), atol=1e-6, rtol=0)


def test_unaligned_views_and_non_contiguous_inputs_are_handled():
    """Slices that start at odd element offsets and permuted views: results equal the contiguous computation bit for bit."""
    base = _rand(5 * 5 * 37 * 41 + 3, seed=17).cuda()
    for off in (1, 2, 3):
        v = base[off:off + 5 * 5 * 37 * 41].view(5, 5, 37, 41)
        c = v.clone()
        assert torch.equal(ops.blend(v, 0.3, v, 0.7), ops.blend(c, 0.3, c, 0.7))
        assert torch.equal(ops.resize_bilinear(v, (50, 60)), ops.resize_bilinear(c, (50, 60)))
        assert torch.equal(ops.argmax_u8(v), ops.argmax_u8(c))
    t = _rand(5, 37, 41, 5, seed=18).cuda().permute(0, 3, 1, 2)          # NHWC storage, C = 5 (not the fast NHWC path)
    assert torch.equal(ops.resize_bilinear(t, (50, 60)), ops.resize_bilinear(t.contiguous(), (50, 60)))
    assert torch.equal(ops.argmax_u8(t), ops.argmax_u8(t.contiguous()))


def test_cpu_tensors_and_wrong_dtypes_are_refused_not_silently_computed():
    with pytest.raises(RuntimeError, match="GPU"):
        ops.resize_bilinear(torch.zeros(1, 5, 4, 4), (8, 8))
    with pytest.raises(RuntimeError):
        ops.iou_hist(torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros(4, dtype=torch.uint8, device="cuda"), 5)
    with pytest.raises(RuntimeError):
        ops.grid_sample(torch.zeros(2, 5, 4, 4, device="cuda"), torch.zeros(1, 4, 4, 2, device="cuda"))
