import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C-ABI library is built in-tree and is git-ignored: (re)build it when it is missing (hipcc cross-compiles on CPU)
    lib = os.path.join(ROOT, "flood_uav_video_segmentation_amd", "libfloodseg.so")
    if not os.path.exists(lib):
        import __graft_entry__

        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


class NullProfiler:
    """Stands in for Lightning's profiler: records the region names FlowModel.predict opens."""

    def __init__(self):
        self.names = []

    def profile(self, name):
        import contextlib

        self.names.append(name)
        return contextlib.nullcontext()


@pytest.fixture()
def profiler():
    return NullProfiler()


def toy_weights():
    z = load_golden("toy_predict.npz")
    return {k: torch.from_numpy(z[k]) for k in ("enc_w", "enc_b", "dec_w", "dec_b")}


def memo_by_content(fn):
    """A CPU-oracle network call memoised on the CONTENT of its input (blake2b of the bytes): the sliding-crop chains clone every crop,
    so an address says nothing, but the same crop of the same frame comes back in the next window / the other test parametrisation --
    and a 713 x 713 oracle forward costs seconds."""
    import hashlib

    cache = {}

    def wrapped(x):
        xc = x.detach().contiguous()
        key = (tuple(xc.shape), str(xc.dtype), hashlib.blake2b(xc.numpy().tobytes(), digest_size=16).hexdigest())
        if key not in cache:
            cache[key] = fn(x)
        return cache[key]
    wrapped.cache = cache
    return wrapped


class Err(float):
    """A max-relative error that also carries the RMS-relative error of the same comparison (note() records both)."""
    rms = None


def rms_rel_err(got, ref):
    """sqrt(mean((got - ref)^2) / mean(ref^2)): the error relative to the tensor's typical magnitude, not to its one largest
    element (with synthetic weights the deep networks' logits reach +-1e3..1e4)."""
    got = torch.as_tensor(got).double()
    ref = torch.as_tensor(ref).double()
    return ((got - ref).pow(2).mean() / ref.pow(2).mean().clamp_min(1e-300)).sqrt().item()


def rel_err(got, ref):
    """max |got - ref| / max |ref|  (the figure every tolerance in tests/ is stated in); `.rms` = rms_rel_err of the same pair."""
    got = torch.as_tensor(got).double()
    ref = torch.as_tensor(ref).double()
    e = Err(((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item())
    e.rms = rms_rel_err(got, ref)
    return e


def note(name, value):
    """Record a measured parity error (GPU runs): appended to gpurun_out/parity_measured.txt so that the asserted tolerances can
    be kept at a small multiple of what is actually measured (DESIGN.md section 5 quotes this file).  A value that came from
    rel_err() is written with its RMS-relative companion: `name max_rel rms_rel=...`."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        rms = getattr(value, "rms", None)
        with open(os.path.join(d, "parity_measured.txt"), "a") as f:
            f.write(f"{name} {value:.3e}" + (f" rms_rel={rms:.3e}" if rms is not None else "") + "\n")
    except OSError:
        pass
    return value
