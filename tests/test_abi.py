"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that include/floodseg.h declares -- and nothing
else but fs_test_hooks(), the one door to the op-level hooks of include/floodseg_test.h (no compute calls here: there is no GPU in the
build container)."""
import ctypes
import os
import re
import subprocess

from flood_uav_video_segmentation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "floodseg.h")
TEST_HEADER = os.path.join(ROOT, "include", "floodseg_test.h")
MAX_EXPORTS = 40  # the product's symbol surface stays small: op-level hooks go into the fs_test_api table, not into the export list


def _strip_comments(path):
    return re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)


def header_symbols():
    """Functions include/floodseg.h declares (the product ABI)."""
    return sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", _strip_comments(HEADER))))


def hook_members():
    """Function-pointer members of fs_test_api, in declaration order."""
    text = _strip_comments(TEST_HEADER)
    body = text[text.index("typedef struct fs_test_api {"):text.index("} fs_test_api;")]
    return re.findall(r"\(\*([a-z0-9_]+)\)\s*\(", body)


def test_header_declares_the_bound_symbols():
    assert sorted(header_symbols() + ["fs_test_hooks"]) == _lib.exported_symbols()
    assert len(_lib.exported_symbols()) <= MAX_EXPORTS


def test_the_two_headers_split_product_and_test_surface():
    """floodseg.h = handle API + the ops INTEGRATION.md maps to reference call sites; floodseg_test.h = ONE exported function and a
    table.  No hook name may appear in the product header, the table in the header and the ctypes binding list the same members in the
    same order, and the library's table is exactly as long as the header says."""
    members = hook_members()
    assert members == _lib.hook_names() and len(members) >= 10
    product = _strip_comments(HEADER)
    for m in members:
        assert ("fs_" + m) not in product, f"fs_{m} is an op-level hook: it belongs to floodseg_test.h"
    fns = sorted(set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", _strip_comments(TEST_HEADER))))
    assert fns == ["fs_test_hooks"]
    lib = _lib.load()
    table = ctypes.cast(lib.fs_test_hooks(), ctypes.POINTER(_lib.FsTestApi)).contents
    assert table.size == ctypes.sizeof(_lib.FsTestApi) == ctypes.sizeof(ctypes.c_size_t) + ctypes.sizeof(ctypes.c_void_p) * len(members)
    for m in members:
        assert ctypes.cast(getattr(table, m), ctypes.c_void_p).value, m  # every member is populated


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (fs_[a-z0-9_]+)", out))
    assert exported == set(header_symbols()) | {"fs_test_hooks"}  # exactly the declared surface: no stray op-level export
    # nothing but the C ABI leaks out of the shared object
    leaked = [l for l in out.splitlines() if " T " in l and " T fs_" not in l and " T _init" not in l and " T _fini" not in l]
    assert not leaked, leaked


def test_version_and_error_string_are_callable_without_a_gpu():
    lib = _lib.load()
    assert lib.fs_version() == 600  # frozen: additions keep it, a changed or removed signature of floodseg.h raises it
    assert isinstance(lib.fs_last_error(), bytes)


def test_bad_config_is_rejected_with_a_message():
    lib = _lib.load()
    cfg = _lib.FsConfig(7, 50, 5)
    h = ctypes.c_void_p()
    assert lib.fs_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"arch" in lib.fs_last_error()
    cfg = _lib.FsConfig(_lib.ARCH_PSPNET, 34, 5)
    assert lib.fs_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"layers" in lib.fs_last_error()


def test_unknown_option_bits_and_tile_sizes_are_rejected():
    """The A/B routes are explicit fs_config options (nothing is read from the environment); anything undefined is refused."""
    lib = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.FsConfig(_lib.ARCH_PSPNET, 50, 5, 0, 0, 0, 0, 0, 1 << 20, 0)   # flags = 2^20: not an FS_OPT_* bit
    assert lib.fs_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and b"option" in lib.fs_last_error()
    for retired in (32, 64, 512):  # the experiment routes of rounds 4-5 (plane operands, chained launches, pipelined attention): removed
        cfg = _lib.FsConfig(_lib.ARCH_PSPNET, 50, 5, 0, 0, 0, 0, 0, retired, 0)
        assert lib.fs_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and b"option" in lib.fs_last_error()
    cfg = _lib.FsConfig(_lib.ARCH_PSPNET, 50, 5, 0, 0, 0, 0, 0, 0, 5)   # winograd_tile = 5
    assert lib.fs_create(ctypes.byref(cfg), ctypes.byref(h)) != 0 and b"winograd_tile" in lib.fs_last_error()
    src = open(os.path.join(ROOT, "flood_uav_video_segmentation_amd", "csrc", "net.hip")).read()
    assert "getenv" not in src  # the library's behaviour must not depend on the ambient environment


def test_conv2d_refuses_every_bit_that_is_not_a_tile_id_or_the_chunk_major_flag():
    """fs_conv2d_nhwc's `tile` once carried bring-up experiment bits that broke the result; the release ABI rejects them
    before anything is launched (dummy non-null pointers: the call must fail in argument validation)."""
    lib = _lib.load()
    fake = ctypes.c_void_p(0x1000)
    args = lambda tile: (fake, 32, fake, None, None, None, 0, fake, 32, 1, 8, 8, 32, 32, 1, 1, 1, 0, 1, 0, tile, None)  # noqa: E731
    for bad in (5, 7, 1 << 11, 1 << 12, (1 << 15) | 1, _lib.CONV_CHUNK_MAJOR | 7, 1 << 8, -1):
        assert lib.fs_conv2d_nhwc(*args(bad)) != 0, bad
        assert b"tile" in lib.fs_last_error()
    sym = subprocess.run(["nm", "-DC", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "fs_trace_buf" not in sym  # the FS_TRACE instrumentation never ships


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libfloodseg.so"))
    try:
        _lib.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_header_is_plain_c():
    """include/floodseg.h is the drop-in boundary for a C / cgo / JNI / ctypes caller: it (and the test header) must compile as C99."""
    import subprocess
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        f.write('#include "floodseg.h"\n#include "floodseg_test.h"\n'
                'int main(void) { fs_config c; int (*v)(void) = fs_version; const fs_test_api* (*t)(void) = fs_test_hooks; (void)c; (void)v; (void)t; return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert r.returncode == 0, r.stderr
