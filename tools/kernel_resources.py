#!/usr/bin/env python3
"""Registers, scratch (private segment), LDS and spill counts of every kernel in the built
libraytracing_amd.so, read from the code objects' metadata notes (no GPU needed).

    python3 tools/kernel_resources.py [path/to/lib.so]      # prints a table
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
          "sgpr_spill_count", "vgpr_spill_count", "max_flat_workgroup_size")


def kernel_resources(lib_path):
    """{kernel symbol: {field: int}} over all gfx950 code objects embedded in `lib_path`."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, copy)
        # llvm-objdump writes one file per embedded bundle next to its input
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", copy], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", os.path.join(tmp, name)],
                                   check=True, capture_output=True, text=True).stdout
            # one YAML map per kernel under amdhsa.kernels, each starting with "  - .agpr_count:" or similar
            for entry in re.split(r"\n  - ", notes.split("amdhsa.kernels:")[-1]):
                m = re.search(r"\.name:\s+(\S+)", entry)
                if not m:
                    continue
                rec = {}
                for f in FIELDS:
                    v = re.search(r"\." + f + r":\s+(\d+)", entry)
                    rec[f] = int(v.group(1)) if v else 0
                out[m.group(1)] = rec
    return out


def demangled(names):
    try:
        text = subprocess.run([os.path.join(LLVM_BIN, "llvm-cxxfilt")], input="\n".join(names), check=True,
                              capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, text))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "raytracing_amd", "libraytracing_amd.so")
    res = kernel_resources(lib)
    pretty = demangled(list(res))
    print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B':>9s} {'LDS B':>6s} "
          f"{'SGPR spills':>11s} {'VGPR spills':>11s} {'waves/SIMD':>10s}")
    for k, r in res.items():
        regs = max(r["vgpr_count"] + r["agpr_count"], 1)
        unified = (regs + 7) // 8 * 8  # gfx950: 512 registers per SIMD lane, granule 8
        waves = min(8, 512 // unified)
        name = pretty[k].replace("(RtParams)", "").replace("void ", "")
        print(f"{name[:58]:58s} {r['vgpr_count']:5d} {r['agpr_count']:5d} {r['sgpr_count']:5d} "
              f"{r['private_segment_fixed_size']:9d} {r['group_segment_fixed_size']:6d} "
              f"{r['sgpr_spill_count']:11d} {r['vgpr_spill_count']:11d} {waves:10d}")
