"""Static guards on the compiled gfx950 code (no GPU needed): properties of the ISA that the source cannot express and
that a compiler upgrade or an innocent-looking edit could silently break."""
import glob
import os
import re
import shutil
import subprocess

import pytest

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flood_uav_video_segmentation_amd", "libfloodseg.so")


def _device_disassembly(tmp_path):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    so = os.path.join(tmp_path, "libfloodseg.so")
    shutil.copy(LIB, so)
    subprocess.run([OBJDUMP, "--offloading", so], check=True, capture_output=True, cwd=tmp_path)
    text = []
    for co in sorted(glob.glob(os.path.join(tmp_path, "libfloodseg.so.*amdgcn*"))):
        text.append(subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout)
    assert text, "no gfx950 code object found in libfloodseg.so"
    return "\n".join(text)


def test_conv_kernel_waits_for_its_lds_dma_before_every_barrier(tmp_path):
    """conv_igemm_dma_f32 stages tiles with `buffer_load ... lds`.  Another wave may read what this wave's DMA wrote only
    after this wave's vmcnt reached 0, and `__syncthreads()` does not imply that wait (the compiler once sank it below the
    barrier in the 128x64 instantiation: a race that showed as run-to-run differences at 713x713).  The source now issues
    an explicit s_waitcnt; this test checks the ISA: every s_barrier of every instantiation is preceded by vmcnt(0)."""
    dis = _device_disassembly(str(tmp_path))
    kernels = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis)
    checked = attention = 0
    for k in kernels:
        head = k.split("\n", 1)[0]
        # round 3: the attention kernels stage K / V the same way (attention_dma_kernel; attention_bf16x3_kernel: the K / V^T planes of
        # the split-operand route) and are held to the same rule, as are the split-operand conv instantiations
        if not any(n in head for n in ("conv_igemm_dma_f32", "attention_dma_kernel", "attention_bf16x3_kernel")):
            continue
        attention += "attention_dma_kernel" in head or "attention_bf16x3_kernel" in head
        lines = [l.split("\t", 1)[-1].strip() if "\t" in l else l.strip() for l in k.splitlines()[1:]]
        ops = [re.sub(r"\s*//.*", "", l) for l in lines if l]
        barriers = [i for i, o in enumerate(ops) if o.startswith("s_barrier")]
        assert barriers, f"no barrier found in {head}"
        for i in barriers:
            window = [o for o in ops[max(0, i - 4):i] if o.startswith("s_waitcnt")]
            assert any("vmcnt(0)" in o for o in window), f"{head}: s_barrier without a preceding s_waitcnt vmcnt(0): {ops[max(0, i - 4):i + 1]}"
            assert any("lgkmcnt(0)" in o for o in window), f"{head}: s_barrier without a preceding s_waitcnt lgkmcnt(0): {ops[max(0, i - 4):i + 1]}"
        assert any("buffer_load_dwordx4" in o and "lds" in o for o in ops), f"{head}: the direct-to-LDS loads are gone"
        checked += 1
    assert checked - attention >= 10, f"expected the fp32 and the split-operand tile instantiations, found {checked - attention}"
    assert attention == 4, f"expected attention_dma_kernel and attention_bf16x3_kernel, key split / no split each, found {attention}"


def test_no_float_atomics_anywhere_and_the_fused_maxpool_uses_integer_max(tmp_path):
    """Bit-repeatability rests on this: the library contains NO floating-point atomic (their result depends on the order of
    arrival).  The one atomic it has (round 5) is the signed-integer max of the stem's fused max-pool -- exact and order-independent
    on values that are >= +0 -- and it lives in wino4_fused_kernel<1, 4, true> only."""
    dis = _device_disassembly(str(tmp_path))
    kernels = re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis)
    with_atomics = {}
    for k in kernels:
        head = k.split("\n", 1)[0]
        ops = [l.split("\t", 1)[-1].strip() for l in k.splitlines()[1:] if "\t" in l]
        atomics = sorted({o.split()[0] for o in ops if "atomic" in o.split()[0]})
        if atomics:
            with_atomics[head] = atomics
    floaty = {h: a for h, a in with_atomics.items() if any(re.search(r"f32|f64|f16|bf16|fadd|fmin|fmax", x) for x in a)}
    assert not floaty, f"floating-point atomics found: {floaty}"
    pool = [h for h in with_atomics if "wino4_fused_kernel" in h and "Lb1E" in h]
    assert pool and all(any("smax" in x for x in with_atomics[h]) for h in pool), with_atomics
    # every other atomic in the library is an integer one as well (histograms: iou_hist / mv_to_grids)
    for h, a in with_atomics.items():
        assert all(re.search(r"atomic_(add|smax|umax|umin|smin|inc|or|and|cmpswap|swap)(_x2)?(_u32|_u64|_i32|_b32|_b64)?$", x) for x in a), (h, a)


def test_no_kernel_puts_each_of_its_stores_behind_a_wait_of_its_own(tmp_path):
    """Round 6: `if (row_ok) out[..] = v` inside an unrolled epilogue loop compiles to one basic block per store, each behind an
    `s_waitcnt vmcnt(0)` -- and gfx9 has ONE in-order counter for loads and stores, so that wait is also a wait for the previous store's
    acknowledgement: the stem kernels had 30 of 32 stores of a lane serialised that way, winograd_output all six.  They now store in a
    straight line through a buffer descriptor whose range check drops the out-of-range rows.  The guard: in no kernel may more than two
    stores sit behind a vmcnt(0) wait that has no load between it and the previous store (a wait that covers loads is doing its job)."""
    dis = _device_disassembly(str(tmp_path))
    bad = {}
    for k in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis):
        head = k.split("\n", 1)[0]
        ops = [re.sub(r"\s*//.*", "", l.split("\t", 1)[-1]).strip() for l in k.splitlines()[1:] if "\t" in l]
        serial = stores = 0
        have_prev = seen_wait = seen_load = False
        for o in ops:
            if re.match(r"(global|buffer|flat)_(store|atomic)", o):
                stores += 1
                serial += have_prev and seen_wait and not seen_load
                have_prev, seen_wait, seen_load = True, False, False
            elif re.match(r"(global|buffer|flat|scratch)_load", o):
                seen_load = True
            elif o.startswith("s_waitcnt") and "vmcnt(0)" in o:
                seen_wait = True
        if serial > 2:
            bad[head] = (serial, stores)
    assert not bad, f"stores serialised behind their own vmcnt(0) waits: {bad}"
