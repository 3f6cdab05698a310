#!/usr/bin/env python3
"""One timed dispatch of the headline workload (development aid for counter collection):
   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -- python3 tools/one_dispatch.py [kernel=0]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracing_amd as rt  # noqa: E402

kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scene = rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
with rt.Renderer(3840, 2160) as r:
    r.set_kernel(kernel)
    r.set_scene(scene, frame_count=1)
    r.dispatch(8)
    r.sync()
    r.reset_counters()
    r.timer_begin()
    r.dispatch(8)
    ms = r.timer_end()
    c = r.counters()
    print(f"kernel {kernel}: {ms:.3f} ms, {c['rays'] / ms / 1e6:.3f} Grays/s, rays {c['rays']}")
