import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import raytracing_amd as rt
import oracle as O
for name, scene in (("rtiow", rt.Scene.generate(rt.KIND_RTIOW, 0, 7)), ("default", rt.Scene.default())):
    ubo = scene.camera.copy(); ubo["frameCount"][0] = 1
    for profile in (0, 1):
        with rt.Renderer(256, 144) as r:
            r.set_kernel(rt.KERNEL_STREAM_MFMA); r.set_profile(profile)
            r.upload_spheres(scene.spheres); r.set_uniforms(ubo); r.dispatch(3)
            acc, col, cnt = r.read_accum(), r.read_color(), r.counters()
            info = r.info()
        cacc, ccol, crays = O.render(ubo, scene.spheres, 256, 144, profile=profile, n_frames=3, threads=8)
        print(name, profile, "accum equal", np.array_equal(acc.view(np.uint32), cacc.view(np.uint32)), "color", np.array_equal(col, ccol),
              "rays", cnt["rays"], crays, info)
    # audit on grazing rays
    rng = np.random.default_rng(1)
    sp = scene.spheres[:scene.sphere_count]
    n = 200000
    idx = rng.integers(0, len(sp), n)
    c = np.stack([sp["center"][idx, k] if "center" in sp.dtype.names else None for k in range(3)], 1) if "center" in sp.dtype.names else None
    if c is None:
        raw = np.ascontiguousarray(sp).view(np.float32).reshape(-1, 8)
        c = raw[idx, :3]; rad = raw[idx, 3]
    else:
        rad = sp["radius"][idx]
    o = scene.camera["camera_position"][0][None, :] + rng.normal(0, 3, (n, 3)).astype(np.float32)
    # aim at a point at distance r*(1+eps) from the centre, perpendicular offset
    d = c - o
    perp = np.cross(d, rng.normal(0, 1, (n, 3)))
    perp /= np.linalg.norm(perp, axis=1, keepdims=True) + 1e-30
    eps = rng.choice([-1e-3, -1e-5, -1e-6, 0, 1e-6, 1e-5, 1e-3], n)
    target = c + perp * (np.abs(rad) * (1 + eps))[:, None]
    d = target - o
    rays = np.concatenate([o, d * rng.uniform(0.2, 3, (n, 1))], 1).astype(np.float32)
    with rt.Renderer(64, 64) as r:
        r.upload_spheres(scene.spheres); r.set_uniforms(ubo)
        for k in (rt.KERNEL_STREAM, rt.KERNEL_STREAM_MFMA):
            r.set_kernel(k)
            print(name, "audit kernel", k, r.sieve_audit(rays))
