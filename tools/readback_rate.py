import sys, time
sys.path.insert(0, '.')
import raytracing_amd as rt
scene = rt.Scene.default()
for (w, h) in ((1920, 1080), (3840, 2160)):
    with rt.Renderer(w, h) as r:
        r.set_scene(scene, frame_count=1)
        r.dispatch(1); r.sync()
        for name, fn in (("read_color", r.read_color), ("read_accum", r.read_accum)):
            fn()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            dt = (time.perf_counter() - t0) / 10
            nbytes = w * h * (4 if name == "read_color" else 16)
            print(f"{w}x{h} {name}: {dt*1e3:.3f} ms  {nbytes/dt/1e9:.1f} GB/s")
        t0 = time.perf_counter()
        for i in range(50):
            r.dispatch(1); r.read_color()
        print(f"{w}x{h} dispatch+read_color: {(time.perf_counter()-t0)/50*1e3:.3f} ms/frame")
