"""Host handle around one fs_net (libfloodseg.so): weight upload, encoder / decoder launches.

Shared by the FlowPSPNet / FlowDeepLabv3 mirrors.  All compute happens in the HIP library; this file
only validates shapes, allocates outputs through torch and passes raw pointers + the current stream.
"""
import ctypes

import torch
from torch import nn

from .. import _lib, ops
from .._lib import check, one_device, ptr, stream_ptr


HIP_OPTIONS = ("hip_no_winograd", "hip_no_fused_head", "hip_no_fused_shortcut", "hip_no_fused_winograd", "hip_no_split_bf16", "hip_no_res_touch",
               "hip_no_fused_pool", "hip_no_fused_qkv", "hip_winograd_tile")


def check_hip_options(names):
    """Every hip_* option a caller names must exist: a mistyped or retired one raises instead of silently running the defaults
    (an A/B script would otherwise compare the defaults with themselves)."""
    unknown = sorted(n for n in names if n.startswith("hip_") and n not in HIP_OPTIONS)
    if unknown:
        raise ValueError(f"floodseg: unknown hip option(s) {unknown}; known: {list(HIP_OPTIONS)}")


def hip_options(hparams):
    """Explicit A/B options of the library (include/floodseg.h: fs_config.flags / .winograd_tile), taken from optional
    attributes of the reference-style `hparams` object (HIP_OPTIONS); any other attribute named hip_* is refused."""
    names = [n for n in dir(hparams) if not n.startswith("__")]
    if hasattr(hparams, "keys"):  # dict-like hparams (Lightning's AttributeDict): the options are keys, not attributes
        names += [str(k) for k in hparams.keys()]
    check_hip_options(names)
    return dict(no_winograd=bool(getattr(hparams, "hip_no_winograd", False)),
                no_fused_winograd=bool(getattr(hparams, "hip_no_fused_winograd", False)),
                no_split_bf16=bool(getattr(hparams, "hip_no_split_bf16", False)),
                no_res_touch=bool(getattr(hparams, "hip_no_res_touch", False)),
                no_fused_pool=bool(getattr(hparams, "hip_no_fused_pool", False)),
                no_fused_qkv=bool(getattr(hparams, "hip_no_fused_qkv", False)),
                no_fused_head=bool(getattr(hparams, "hip_no_fused_head", False)),
                no_fused_shortcut=bool(getattr(hparams, "hip_no_fused_shortcut", False)),
                winograd_tile=int(getattr(hparams, "hip_winograd_tile", 0)))


class HipNet:
    def __init__(self, arch, layers, classes, patch=0, d_model=0, n_layers=0, dec_layers=0, image_size=0, no_winograd=False,
                 no_fused_head=False, no_fused_shortcut=False, winograd_tile=0, no_fused_winograd=False, no_split_bf16=False,
                 no_res_touch=False, no_fused_pool=False, no_fused_qkv=False):
        self.arch, self.layers, self.classes = arch, int(layers), int(classes)
        self.vit = (int(patch), int(d_model), int(n_layers), int(dec_layers), int(image_size))
        self.flags = ((_lib.OPT_NO_WINOGRAD if no_winograd else 0) | (_lib.OPT_NO_FUSED_HEAD if no_fused_head else 0)
                      | (_lib.OPT_NO_FUSED_SHORTCUT if no_fused_shortcut else 0) | (_lib.OPT_NO_FUSED_WINOGRAD if no_fused_winograd else 0)
                      | (_lib.OPT_NO_SPLIT_BF16 if no_split_bf16 else 0) | (_lib.OPT_NO_RES_TOUCH if no_res_touch else 0)
                      | (_lib.OPT_NO_FUSED_POOL if no_fused_pool else 0) | (_lib.OPT_NO_FUSED_QKV if no_fused_qkv else 0))
        self.winograd_tile = int(winograd_tile)
        self._h = None
        self.ready = False
        self._lib = None
        self.device = None  # torch.device the handle's weights and workspace live on (set by load())
        self.generation = 0  # bumped by every load(): key-frame caches tag their entries with it (flow/model.py)

    # -- lifecycle ---------------------------------------------------------------------------
    def _create(self):
        lib = self._lib = _lib.load()
        cfg = _lib.FsConfig(self.arch, self.layers, self.classes, *self.vit, self.flags, self.winograd_tile)
        h = ctypes.c_void_p()
        check(lib.fs_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h

    def load(self, canonical_state, device=None):
        """canonical_state: {canonical name -> tensor (any device)}; replaces any previous weights.  The handle is created
        on `device`: the device of the first GPU tensor in the state dict, else the current device -- and stays there."""
        if not torch.cuda.is_available():
            raise RuntimeError("floodseg: a HIP device is required (there is no CPU fallback for this path)")
        self.close()
        if device is None:
            on_gpu = {t.device for t in canonical_state.values() if t.is_cuda}
            if len(on_gpu) > 1:
                raise RuntimeError(f"floodseg: state_dict tensors live on several devices ({sorted(map(str, on_gpu))})")
            device = on_gpu.pop() if on_gpu else torch.device("cuda", torch.cuda.current_device())
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(f"floodseg: the network can only live on a HIP device, not {device}")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(device):
            self._create()
            lib = self._lib
            keep = []  # the copies are asynchronous: every source tensor must outlive the synchronize below
            for name, t in canonical_state.items():
                t = t.detach()
                if t.is_cuda and t.device != device:
                    raise RuntimeError(f"floodseg: weight '{name}' lives on {t.device}, the network on {device}")
                if t.dtype != torch.float32:
                    t = t.float()
                t = t.contiguous()
                keep.append(t)
                shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
                check(lib.fs_load_weight(self._h, name.encode(), ptr(t), shape, t.dim(), int(t.is_cuda), stream_ptr()))
            torch.cuda.current_stream().synchronize()
            del keep
            check(lib.fs_finalize(self._h, stream_ptr()))
        self.device = device
        self.ready = True
        self.generation += 1

    def close(self):
        if self._h is not None and self._lib is not None:
            if self.device is not None:
                with torch.cuda.device(self.device):
                    self._lib.fs_destroy(self._h)
            else:
                self._lib.fs_destroy(self._h)
        self._h = None
        self.ready = False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  interpreter shutdown
            pass

    # -- forward -----------------------------------------------------------------------------
    def _need_ready(self):
        if not self.ready:
            raise RuntimeError("floodseg: weights not loaded -- call load_state_dict() on the network first")

    def feature_shape(self, h, w):
        self._need_ready()
        c, fh, fw = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self._lib.fs_feature_shape(self._h, h, w, ctypes.byref(c), ctypes.byref(fh), ctypes.byref(fw)))
        return c.value, fh.value, fw.value

    def reserve(self, batch, h, w):
        """fs_reserve: size the library's workspace (and build the Winograd filter banks) for forwards over at most `batch`
        frames / crops of h x w, once, so that no later forward at that geometry allocates or synchronises."""
        self._need_ready()
        with torch.cuda.device(self.device):
            check(self._lib.fs_reserve(self._h, int(batch), int(h), int(w), stream_ptr()))

    def reserved_bytes(self):
        self._need_ready()
        return int(self._lib.fs_reserved_bytes(self._h))

    def _frames(self, frames, what):
        """Validate a batch given as one or two [B_i,3,H,W] tensors (a longer list is concatenated: rare, not the window path)."""
        if len(frames) > 2:
            frames = (torch.cat([f.float() for f in frames], 0),)
        for x in frames:
            if x.dim() != 4 or x.shape[1] != 3:
                raise RuntimeError(f"floodseg {what}: expected [B,3,H,W], got {tuple(x.shape)}")
            if not x.is_cuda:
                raise RuntimeError(f"floodseg {what}: input must be on the GPU")
            if x.shape[2:] != frames[0].shape[2:]:
                raise RuntimeError(f"floodseg {what}: frames of different sizes in one batch")
        return frames

    def encode(self, *frames):
        """[B,3,H,W] NCHW -> [B,C,fh,fw] logical NCHW stored channels_last (model.encoder(x)).  Two tensors (the two key
        frames of a window) go through as ONE batch, read in place (fs_encoder_forward2): no torch.cat."""
        self._need_ready()
        frames = self._frames(frames, "encoder")
        with torch.cuda.device(one_device(*frames, handle_device=self.device, what="floodseg encoder")):
            xs = [x.float().contiguous() for x in frames]
            h, w = xs[0].shape[2], xs[0].shape[3]
            b = sum(x.shape[0] for x in xs)
            c, fh, fw = self.feature_shape(h, w)
            out = ops.empty_nhwc(b, c, fh, fw, xs[0].device)
            if len(xs) == 1:
                check(self._lib.fs_encoder_forward(self._h, ptr(xs[0]), b, h, w, ptr(out), stream_ptr()))
            else:
                check(self._lib.fs_encoder_forward2(self._h, ptr(xs[0]), xs[0].shape[0], ptr(xs[1]), xs[1].shape[0], h, w, ptr(out),
                                                    stream_ptr()))
        return out

    def decode(self, f):
        """[B,C,fh,fw] -> [B,K,fh,fw] NCHW logits (model.decoder(f))."""
        self._need_ready()
        c_expected = {_lib.ARCH_PSPNET: 4096, _lib.ARCH_DEEPLABV3: 2048}.get(self.arch, self.vit[1])
        if f.dim() != 4 or f.shape[1] != c_expected:
            raise RuntimeError(f"floodseg decoder: expected [B,{c_expected},h,w], got {tuple(f.shape)}")
        with torch.cuda.device(one_device(f, handle_device=self.device, what="floodseg decoder")):
            f = ops.as_nhwc(f)
            b, _, fh, fw = f.shape
            out = torch.empty((b, self.classes, fh, fw), dtype=torch.float32, device=f.device)
            check(self._lib.fs_decoder_forward(self._h, ptr(f), b, fh, fw, ptr(out), stream_ptr()))
        return out

    def segment(self, *frames):
        """[B,3,H,W] -> [B,K,fh,fw] NCHW logits: model.decoder(model.encoder(x)) in one library call (the PSPNet head then
        never builds the 4096-channel concat, include/floodseg.h).  Two tensors = one batch read in place, as in encode()."""
        self._need_ready()
        frames = self._frames(frames, "segment")
        with torch.cuda.device(one_device(*frames, handle_device=self.device, what="floodseg segment")):
            xs = [x.float().contiguous() for x in frames]
            h, w = xs[0].shape[2], xs[0].shape[3]
            b = sum(x.shape[0] for x in xs)
            _, fh, fw = self.feature_shape(h, w)
            out = torch.empty((b, self.classes, fh, fw), dtype=torch.float32, device=xs[0].device)
            if len(xs) == 1:
                check(self._lib.fs_segment_forward(self._h, ptr(xs[0]), b, h, w, ptr(out), stream_ptr()))
            else:
                check(self._lib.fs_segment_forward2(self._h, ptr(xs[0]), xs[0].shape[0], ptr(xs[1]), xs[1].shape[0], h, w, ptr(out),
                                                    stream_ptr()))
        return out

    def segment_crops(self, frame_a, frame_b, crop_yx, crop_hw):
        """decoder(encoder(.)) of the crop windows of one or two FULL frames [1,3,H,W] as one batch, read in place
        (fs_segment_crops; flow/base.py:199-200 clones each crop).  -> [ncrops * (2 if frame_b is given else 1), K, fh, fw]."""
        self._need_ready()
        frames = self._frames((frame_a,) if frame_b is None else (frame_a, frame_b), "segment_crops")
        if any(f.shape[0] != 1 for f in frames):
            raise RuntimeError("floodseg segment_crops: full frames must be [1,3,H,W] (flow/base.py:263)")
        with torch.cuda.device(one_device(*frames, handle_device=self.device, what="floodseg segment_crops")):
            xs_ = [x.float().contiguous() for x in frames]
            fh_, fw_ = xs_[0].shape[2], xs_[0].shape[3]
            nc = len(crop_yx)
            ch, cw = int(crop_hw[0]), int(crop_hw[1])
            ys = (ctypes.c_int * nc)(*[int(y) for y, _ in crop_yx])
            xs = (ctypes.c_int * nc)(*[int(x) for _, x in crop_yx])
            _, fh, fw = self.feature_shape(ch, cw)
            out = torch.empty((nc * len(xs_), self.classes, fh, fw), dtype=torch.float32, device=xs_[0].device)
            check(self._lib.fs_segment_crops(self._h, ptr(xs_[0]), ptr(xs_[1]) if len(xs_) > 1 else None, fh_, fw_, nc, ys, xs, ch, cw,
                                             ptr(out), stream_ptr()))
        return out

    # -- profiling ---------------------------------------------------------------------------
    def profile(self, on):
        self._need_ready()
        check(self._lib.fs_profile_enable(self._h, int(on)))

    def profile_dump(self):
        """[(name, kernel, flops, bytes, ms)] for every launch recorded since profile(True)."""
        buf = ctypes.create_string_buffer(1 << 20)
        with torch.cuda.device(self.device):
            check(self._lib.fs_profile_dump(self._h, buf, len(buf)))
        rows = []
        for line in buf.value.decode().splitlines():
            name, kernel, flops, nbytes, ms = line.rsplit(" ", 4)
            rows.append((name, kernel, float(flops), float(nbytes), float(ms)))
        return rows


class HipStage(nn.Module):
    """Parameter-less nn.Module standing where the reference has a torch sub-network
    (`model.encoder`, `model.decoder`, `model.layers`, `model.ppm`: flow/base.py:96-101 only calls
    .parameters() on them)."""

    def __init__(self, fn=None, what=""):
        super().__init__()
        self._fn = fn
        self._what = what

    def _load_from_state_dict(self, *args, **kwargs):
        return  # the owning HipSegNet consumes every key under its prefix, aliases included

    def forward(self, x):
        if self._fn is None:
            raise RuntimeError(f"floodseg: '{self._what}' has no standalone forward on the HIP path")
        return self._fn(x)


class HipSegNet(nn.Module):
    """Common base of the network mirrors: swallows the reference's state_dict keys (with their
    aliases) in load_state_dict and forwards them to the HIP library."""

    ARCH = None

    def __init__(self, hparams):
        super().__init__()
        self._hip_net = HipNet(self.ARCH, hparams.layers, hparams.classes, **hip_options(hparams))
        self.encoder = HipStage(self._hip_net.encode, "encoder")
        self.decoder = HipStage(self._hip_net.decode, "decoder")

    def reserve(self, batch, h, w):
        """Pre-size the HIP library's workspace for forwards over at most `batch` frames (or crops) of h x w (fs_reserve)."""
        self._hip_net.reserve(batch, h, w)

    def encode_frames(self, *frames):
        """model.encoder over a batch given as separate tensors (FlowModel: frame_prev, frame_next) -- read in place."""
        return self._hip_net.encode(*frames)

    @staticmethod
    def canonical_name(key):
        raise NotImplementedError

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        canon = {}
        for key, t in state_dict.items():
            if not key.startswith(prefix):
                continue
            name = self.canonical_name(key[len(prefix):])
            if name is None:
                continue
            if name in canon:
                continue  # an alias of a tensor already taken (FlowPSPNet registers modules under several names)
            canon[name] = t
        if not canon:
            if strict:
                missing_keys.append(prefix + "<all weights>")
            return
        try:
            self._hip_net.load(canon)
        except RuntimeError as e:
            error_msgs.append(str(e))

    def load_state_dict(self, state_dict, strict=True, assign=False):  # noqa: ARG002
        missing, unexpected, errors = [], [], []
        self._load_from_state_dict(dict(state_dict), "", {}, strict, missing, unexpected, errors)
        if errors or (strict and missing):
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(type(self).__name__, "\n\t".join(errors + missing)))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)
