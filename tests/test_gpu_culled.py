"""GPU parity of RT_KERNEL_STREAM_CULLED, the stream kernel over a spatially sorted copy of the
sphere list that skips groups of 8 spheres no ray of a wave can touch, and of RT_KERNEL_BVH, which
walks the same list as a hierarchy (blocks of 64, groups of 8, spheres) lane by lane.  Neither is
brute force any more, so neither is what the headline benchmark runs; their results must still be
the oracle's bit for bit -- same hit indices (ties go to the lowest ORIGINAL index), same images,
same ray count."""
import numpy as np
import pytest

import oracle as O
from test_gpu_parity import _bits, _render_cpu, _render_gpu, _tangent_rays

pytestmark = pytest.mark.gpu

CULLED = 4
BVH = 8
BOTH = pytest.mark.parametrize("kernel", [CULLED, BVH], ids=["culled", "bvh"])


@pytest.fixture(scope="module")
def rt():
    import raytracing_amd

    return raytracing_amd


@pytest.mark.parametrize("scene_name", ["default", "rtiow", "cube1000", "cube4096", "cube9000"])
def test_group_test_never_culls_a_literal_candidate(rt, default_scene, rtiow_scene, scene_name):
    """Audit of the group test alone: over camera-like, random and limb-grazing rays, no sphere with
    a non-negative literal discriminant (either profile) may sit in a culled group; and the test
    must cull something."""
    scene = {"default": default_scene, "rtiow": rtiow_scene}.get(scene_name) or \
        rt.Scene.generate(rt.KIND_CUBE, int(scene_name[4:]), 5)
    sph = scene.spheres[: scene.sphere_count]
    rng = np.random.default_rng(7)
    cam = scene.camera["camera_position"][0].astype(np.float64)
    lo = np.maximum(sph["center"].min(0) - 5.0, -60.0)
    hi = np.minimum(sph["center"].max(0) + 5.0, 60.0)
    k = rng.integers(0, len(sph), 600)
    n = rng.standard_normal((600, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    surface = sph["center"][k] + n * (np.abs(sph["radius"][k])[:, None] * (1 + 1e-4))
    far = cam[None] + rng.standard_normal((100, 3)) * 500.0   # far origins stress the o.o margin
    origins = np.concatenate([np.repeat(cam[None], 200, 0), rng.uniform(lo, hi, (600, 3)), surface, far])
    rays = [_tangent_rays(sph, origins, rng, 64)]
    d = rng.standard_normal((60000, 3))
    o = origins[rng.integers(0, len(origins), 60000)]
    rays.append(np.concatenate([o, d], axis=1).astype(np.float32))
    rays = np.concatenate(rays)
    with rt.Renderer(64, 64) as r:
        r.upload_spheres(scene.spheres)
        r.set_uniforms(scene.camera)
        res = r.cull_audit(rays)
    assert res["missed"] == 0, res
    assert res["tested"] > 0 and res["culled"] > 0.1 * res["tested"], res
    assert res["hits"] > 1000 and res["hits_pruned"] == 0, res
    assert res["boxes_tested"] > 0 and res["boxes_culled"] > 0.1 * res["boxes_tested"], res


@BOTH
@pytest.mark.parametrize("profile", [0, 1])
def test_culled_images_match_oracle(rt, default_scene, rtiow_scene, profile, kernel):
    for scene, (w, h), frames, depth in ((default_scene, (320, 184), 3, 50), (rtiow_scene, (320, 184), 4, 50)):
        acc, col, cnt = _render_gpu(rt, scene, w, h, profile=profile, kernel=kernel, n_frames=frames, depth=depth)
        cacc, ccol, crays = _render_cpu(scene, w, h, profile=profile, n_frames=frames, depth=depth)
        assert np.array_equal(_bits(acc), _bits(cacc))
        assert np.array_equal(col, ccol)
        assert cnt["rays"] == crays


@BOTH
@pytest.mark.parametrize("count", [0, 1, 33, 96, 97, 100, 121, 128, 129, 255, 481, 513, 1025, 4096])
def test_culled_sphere_counts(rt, count, kernel):
    """Ragged groups and chunks, the 512-sphere supergroup boundary, the smallest (97) and largest
    (4096) scenes the culled kernel takes (smaller ones run the brute-force kernel);
    sphere_count smaller than the upload (the sorted copy holds the traced ones only)."""
    scene = rt.Scene.generate(rt.KIND_CUBE, max(count, 3) + 5, 11)
    acc, col, cnt = _render_gpu(rt, scene, 96, 64, profile=0, kernel=kernel, n_frames=2, depth=6, sphere_count=count)
    cacc, ccol, crays = _render_cpu(scene, 96, 64, profile=0, n_frames=2, depth=6, sphere_count=count)
    assert np.array_equal(_bits(acc), _bits(cacc)), count
    assert np.array_equal(col, ccol) and cnt["rays"] == crays


@BOTH
@pytest.mark.parametrize("count", [5000, 4097, 20000])
def test_above_4096_spheres(rt, kernel, count):
    """Beyond what LDS holds the culled kernel falls back to the brute-force kernel, and the per-lane
    traversal walks out of global memory with a level of 512-slot superblocks on top."""
    scene = rt.Scene.generate(rt.KIND_CUBE, count, 21)
    acc, col, cnt = _render_gpu(rt, scene, 96, 64, profile=1, kernel=kernel, n_frames=2, depth=4)
    cacc, ccol, crays = _render_cpu(scene, 96, 64, profile=1, n_frames=2, depth=4)
    assert np.array_equal(_bits(acc), _bits(cacc)) and np.array_equal(col, ccol) and cnt["rays"] == crays
    with rt.Renderer(96, 64) as r:
        r.set_kernel(kernel)
        r.set_scene(scene, frame_count=1)
        r.dispatch(1)
        r.sync()
        assert r.sieve_mode() == (6 if kernel == BVH else 1)


@BOTH
def test_culled_ties_degenerates_and_records(rt, kernel):
    """Duplicates must still resolve to the lowest original index although the sorted list visits
    them in another order; non-finite spheres keep their group always evaluated; the per-bounce
    records (original indices) equal the oracle's."""
    scene = rt.Scene.generate(rt.KIND_CUBE, 160, 17)
    sph = scene.spheres
    sph[40] = sph[1]
    sph[2] = sph[41]
    sph[77] = sph[41]
    sph[52]["radius"] = 0.0
    sph[53]["radius"] = -sph[53]["radius"]
    sph[55]["center"] = (0.0, -1.0e6, 0.0)
    sph[55]["radius"] = 1.0e6 - 20.0
    sph[57]["center"] = (np.inf, 0.0, 0.0)
    sph[58]["center"] = (np.nan, 0.0, 0.0)
    sph[59]["radius"] = np.inf
    sph[60]["radius"] = np.nan
    for profile in (0, 1):
        acc, col, cnt = _render_gpu(rt, scene, 160, 96, profile=profile, kernel=kernel, n_frames=3, depth=12)
        cacc, ccol, crays = _render_cpu(scene, 160, 96, profile=profile, n_frames=3, depth=12)
        assert np.array_equal(_bits(acc), _bits(cacc)) and np.array_equal(col, ccol) and cnt["rays"] == crays
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 2
    with rt.Renderer(160, 96) as r:
        r.set_kernel(kernel)
        r.set_max_depth(12)
        r.upload_spheres(sph)
        r.set_uniforms(ubo)
        rec, light = r.trace_crop(2, 40, 20, 64, 48)
    crec, clight = O.trace_crop(ubo, sph, 160, 96, 40, 20, 64, 48, max_depth=12, n_frames=2)
    assert np.array_equal(rec["hit"], crec["hit"]) and np.array_equal(rec["hash"], crec["hash"])
    assert np.array_equal(_bits(light), _bits(clight))


@BOTH
def test_culled_with_materials_and_reupload(rt, kernel):
    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 2)
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    with rt.Renderer(256, 144) as r:
        r.set_kernel(kernel)
        r.set_max_depth(16)
        r.set_scene(scene, frame_count=1)
        r.dispatch(2)
        acc = r.read_accum()
        cacc, _, _ = O.render(ubo, scene.spheres, 256, 144, n_frames=2, max_depth=16, materials=scene.materials,
                              threads=8)
        assert np.array_equal(_bits(acc), _bits(cacc))
        # another sphere list on the same context: the sorted copy must follow
        other = rt.Scene.generate(rt.KIND_CUBE, 300, 9)
        r.set_scene(other, frame_count=1)
        r.dispatch(2)
        acc2 = r.read_accum()
        ubo2 = other.camera.copy()
        ubo2["frameCount"][0] = 1
        cacc2, _, _ = O.render(ubo2, other.spheres, 256, 144, n_frames=2, max_depth=16, threads=8)
        assert np.array_equal(_bits(acc2), _bits(cacc2))
        # fewer traced spheres of the same upload: rebuilt for the new count
        ubo2["sphereCount"][0] = 100
        r.set_uniforms(ubo2)
        r.dispatch(2)
        acc3 = r.read_accum()
        cacc3, _, _ = O.render(ubo2, other.spheres, 256, 144, n_frames=2, max_depth=16, threads=8)
        assert np.array_equal(_bits(acc3), _bits(cacc3))


@BOTH
def test_culled_full_size_equals_brute_force(rt, rtiow_scene, kernel):
    """3840x2160: the culled kernel against the brute-force stream kernel over the whole image."""
    w, h = 3840, 2160
    with rt.Renderer(w, h) as r:
        r.set_scene(rtiow_scene, frame_count=1)
        r.dispatch(2)
        a0, c0, n0 = r.read_accum(), r.read_color(), r.counters()["rays"]
        r.reset_counters()
        r.set_kernel(kernel)
        r.dispatch(2)
        a1, c1, n1 = r.read_accum(), r.read_color(), r.counters()["rays"]
    assert np.array_equal(_bits(a0), _bits(a1)) and np.array_equal(c0, c1) and n0 == n1
