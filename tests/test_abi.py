"""CPU tests of the C-ABI boundary: the shared library loads without a GPU, exports every symbol
include/rt_hip.h declares (and the binding declares nothing the header lacks), and refuses to
run without a device instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "rt_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from raytracing_amd import _native

    names = _declared()
    assert len(names) >= 40
    lib = C.CDLL(_native.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding covers exactly the header
    assert sorted(_native.SIGNATURES) == names


def test_every_entry_point_cites_what_it_replaces():
    """The comment in front of each declaration names the reference file:line it stands in for, or
    says that the reference has no counterpart (consecutive declarations may share a comment)."""
    text = open(HEADER).read()
    body = text[text.index("backend */"):]
    for m in re.finditer(r"\n(?:int|const char\*|void\*)\s+(rt_[a-z0-9_]+)\s*\(", body):
        before = body[: m.start()]
        comment = before[before.rindex("/*"):]
        assert re.search(r"\.(cpp|hpp|comp|glsl):\d+|no\s+reference\s+counterpart", comment, re.I), m.group(1)


def test_no_device_means_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import raytracing_amd as rt

    with pytest.raises(rt.RtError) as e:
        rt.Renderer(64, 64)
    assert e.value.code == -2 and "no CPU path" in e.value.message


def test_product_does_not_touch_the_oracle():
    """Nothing under raytracing_amd/ may import, link or load the oracle."""
    pkg = os.path.join(ROOT, "raytracing_amd")
    for base, _, files in os.walk(pkg):
        if os.path.basename(base) == "build":
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)


def test_kernels_of_the_static_entry_point_need_no_scratch_memory():
    """rt_dispatch_static promises to allocate nothing.  The HIP runtime allocates scratch memory by
    itself when a kernel with a private segment is first launched, so the kernels that entry point
    launches -- the lite kernel, the stream kernel with its geometry in LDS (production
    instantiations), its own two helpers -- must have a private segment of zero bytes and no
    spilled vector registers.  Read off the code objects embedded in the built library."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import kernel_resources
    from raytracing_amd import _native

    res = kernel_resources(_native.LIB_PATH)
    assert len(res) >= 20
    wanted = [k for k in res
              if "rt_lite_kernel" in k or "rt_static_" in k or re.search(r"rt_stream_kernelILb[01]ELb0ELi0E", k)]
    assert len(wanted) == 8, wanted  # 2 stream + 4 lite instantiations + 2 helpers
    for k in wanted:
        assert res[k]["private_segment_fixed_size"] == 0 and res[k]["vgpr_spill_count"] == 0, (k, res[k])
    # and no instantiation of any kernel spills vector registers
    assert all(r["vgpr_spill_count"] == 0 for r in res.values())


def test_binding_and_library_agree_on_the_abi_version():
    from raytracing_amd import _native

    m = re.search(r"#define\s+RT_ABI_VERSION\s+(\d+)", open(HEADER).read())
    assert m and int(m.group(1)) == _native.ABI_VERSION == _native.lib().rt_abi_version()


def test_shipped_library_reads_no_development_knobs():
    """RT_DEBUG_* environment variables are read only by a build with -DRT_DEV_KNOBS; the library that
    ships must not look at the environment when it picks a kernel."""
    from raytracing_amd import _native

    blob = open(_native.LIB_PATH, "rb").read()
    assert b"RT_DEBUG_" not in blob
