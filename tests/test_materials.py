"""Material extension (lambertian / metal / dielectric per sphere).

The reference has one surface response (res/shaders/shader.comp:97-118, kind 0 here); its
def.glsl:65-73 holds an unused random_unit_vector.  The extension is this build's own design on
top of that primitive, so there is nothing in the reference to pin it against: "parity
unpinned".  What the tests can and do fix: the CPU restatement is deterministic and reduces to
the reference path when every kind is 0, scene files without materials are unchanged, and (in
the gpu tests) the HIP kernels reproduce the restatement bit for bit.
"""
import numpy as np
import pytest

import oracle as O


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ---------------------------------------------------------------- CPU: oracle + host scene model

def test_sincos_2pi_accuracy_and_symmetry():
    """The shared {cos, sin}(2*pi*u) sequence: within 2.5e-7 of the true value on [0,1]; exact at
    the quadrant points."""
    u = np.linspace(0.0, 1.0, 20001, dtype=np.float32)
    got = np.array([O.sincos_2pi(float(x)) for x in u], np.float64)
    th = 2.0 * np.pi * u.astype(np.float64)
    assert np.abs(got[:, 0] - np.cos(th)).max() < 2.5e-7
    assert np.abs(got[:, 1] - np.sin(th)).max() < 2.5e-7
    assert O.sincos_2pi(0.0) == (1.0, 0.0)
    assert O.sincos_2pi(0.25) == (-0.0, 1.0) or O.sincos_2pi(0.25) == (0.0, 1.0)
    assert O.sincos_2pi(0.5)[0] == -1.0
    assert O.sincos_2pi(1.0)[0] == 1.0
    n = np.hypot(got[:, 0], got[:, 1])
    assert np.abs(n - 1.0).max() < 4e-7


def test_kind_zero_materials_reduce_to_the_reference_path(rtiow_scene):
    ubo = rtiow_scene.camera.copy()
    ubo["frameCount"][0] = 1
    zeros = np.zeros((len(rtiow_scene.spheres), 2), np.float32)
    for profile in (O.STRICT, O.FUSED):
        a0, c0, n0 = O.render(ubo, rtiow_scene.spheres, 96, 56, profile=profile, n_frames=2, max_depth=10)
        a1, c1, n1 = O.render(ubo, rtiow_scene.spheres, 96, 56, profile=profile, n_frames=2, max_depth=10,
                              materials=zeros)
        assert np.array_equal(_bits(a0), _bits(a1)) and np.array_equal(c0, c1) and n0 == n1


def test_material_oracle_is_deterministic_and_differs():
    import raytracing_amd as rt

    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 1)
    plain = rt.Scene.generate(rt.KIND_RTIOW, 0, 1)
    assert np.array_equal(scene.spheres.view(np.uint8), plain.spheres.view(np.uint8))
    mats = scene.materials
    assert mats.shape == (len(scene.spheres), 2)
    kinds = mats[:, 0]
    assert set(np.unique(kinds)) == {1.0, 2.0, 3.0}
    assert not plain.materials.any()
    assert np.all(mats[kinds == 3, 1] == 1.5) and np.all((mats[kinds == 2, 1] >= 0) & (mats[kinds == 2, 1] <= 0.5))
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    a0, c0, n0 = O.render(ubo, scene.spheres, 128, 72, n_frames=2, max_depth=12, materials=mats, threads=4)
    a1, c1, n1 = O.render(ubo, scene.spheres, 128, 72, n_frames=2, max_depth=12, materials=mats, threads=1)
    assert np.array_equal(_bits(a0), _bits(a1)) and np.array_equal(c0, c1) and n0 == n1
    a2, _, _ = O.render(ubo, scene.spheres, 128, 72, n_frames=2, max_depth=12, threads=4)
    assert not np.array_equal(_bits(a0), _bits(a2))
    assert np.isfinite(a0).all() and a0[..., :3].min() >= 0.0
    # all three kinds are exercised, including absorbed metal rays and refraction
    rec, light = O.trace_crop(ubo, scene.spheres, 128, 72, 0, 0, 128, 72, max_depth=12, materials=mats)
    hit = rec["hit"][0]
    seen = kinds[hit[hit >= 0]]
    assert {1.0, 2.0, 3.0} <= set(np.unique(seen))


def test_scene_yaml_carries_materials_only_when_present(tmp_path):
    import raytracing_amd as rt

    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 7)
    p = tmp_path / "m.yaml"
    scene.save(str(p))
    back = rt.Scene.load(str(p))
    assert np.array_equal(back.spheres.view(np.uint8), scene.spheres.view(np.uint8))
    assert np.array_equal(_bits(back.materials), _bits(scene.materials))
    p2 = tmp_path / "m2.yaml"
    back.save(str(p2))
    assert p.read_bytes() == p2.read_bytes()
    # a scene without materials is written exactly as the reference writes it (3 elements per sphere)
    plain = rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
    q = tmp_path / "p.yaml"
    plain.save(str(q))
    n = len(plain.spheres)
    # block style as the reference's emitter writes it: 10 lines per sphere, 13 with a material
    assert abs(len(p.read_text().splitlines()) - len(q.read_text().splitlines())) == 3 * n
    good = p.read_text()
    first = "    -\n      - 3\n      - 1.5\n"
    assert first in good
    for bad in ("    -\n      - 9\n      - 1.5\n", "    -\n      - 3\n      - 0\n", "    -\n      - 3\n"):
        with pytest.raises(rt.RtError):
            rt.Scene.parse(good.replace(first, bad, 1))


def test_glass_sphere_is_entered_and_left():
    """A dielectric sphere is also hit from inside (far root), so a ray refracts in, crosses it and
    refracts out: seen head-on, most paths are [glass, glass, sky]."""
    import raytracing_amd as rt

    scene = rt.Scene.generate(rt.KIND_CUBE, 4, 1)
    sph = scene.spheres
    cam = scene.camera["camera_position"][0]
    fwd = scene.camera["camera_forward"][0]
    for i in range(3):  # the three emitters, out of sight behind the camera
        sph[i]["center"] = cam - fwd * 1000.0 + np.float32(i * 50)
        sph[i]["radius"] = 1.0
    sph[3]["center"] = cam + fwd * 10.0
    sph[3]["radius"] = 2.0
    sph[3]["color"] = (1.0, 1.0, 1.0)
    mats = np.zeros((4, 2), np.float32)
    mats[3] = (3.0, 1.5)
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    for profile in (O.STRICT, O.FUSED):
        rec, light = O.trace_crop(ubo, sph, 256, 144, 112, 56, 32, 32, max_depth=8, profile=profile, materials=mats)
        hit = rec["hit"][0].reshape(-1, 8)
        through = np.all(hit[:, :3] == (3, 3, -1), axis=1)
        assert through.mean() > 0.85
        assert np.all(hit[:, 0] == 3)
        # without the material the near-mirror reflects every one of them straight back to the sky
        rec0, _ = O.trace_crop(ubo, sph, 256, 144, 112, 56, 32, 32, max_depth=8, profile=profile)
        assert np.all(rec0["hit"][0].reshape(-1, 8)[:, :2] == (3, -1))
        # white glass loses nothing: the sky seen through it keeps the full first-bounce weight
        sky = np.array([0.6, 0.7, 0.9], np.float32) * np.float32(0.15)
        np.testing.assert_allclose(light[0].reshape(-1, 3)[through], np.tile(sky, (through.sum(), 1)), rtol=1e-6)


# ---------------------------------------------------------------- GPU: kernels vs the restatement

@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1, 2, 3])
@pytest.mark.parametrize("profile", [0, 1])
def test_gpu_material_images_match_oracle(profile, kernel):
    import raytracing_amd as rt

    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 1)
    w, h = 320, 184
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    with rt.Renderer(w, h) as r:
        r.set_profile(profile)
        r.set_kernel(kernel)
        r.set_max_depth(16)
        r.set_scene(scene, frame_count=1)
        r.dispatch(3)
        acc, col, cnt = r.read_accum(), r.read_color(), r.counters()
    cacc, ccol, crays = O.render(ubo, scene.spheres, w, h, profile=profile, n_frames=3, max_depth=16,
                                 materials=scene.materials, threads=8)
    assert np.array_equal(_bits(acc), _bits(cacc))
    assert np.array_equal(col, ccol)
    assert cnt["rays"] == crays


@pytest.mark.gpu
@pytest.mark.parametrize("profile", [0, 1])
def test_gpu_material_records_match_oracle(profile):
    import raytracing_amd as rt

    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 3)
    w, h = 640, 360
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 4
    with rt.Renderer(w, h) as r:
        r.set_profile(profile)
        r.set_scene(scene, frame_count=4)
        rec, light = r.trace_crop(2, 280, 150, 48, 32)
    crec, clight = O.trace_crop(ubo, scene.spheres, w, h, 280, 150, 48, 32, profile=profile, n_frames=2,
                                materials=scene.materials)
    assert np.array_equal(rec["hit"], crec["hit"])
    assert np.array_equal(rec["hash"], crec["hash"])
    assert np.array_equal(_bits(light), _bits(clight))
    kinds = scene.materials[:, 0]
    assert {1.0, 2.0, 3.0} <= set(np.unique(kinds[crec["hit"][crec["hit"] >= 0]]))


@pytest.mark.gpu
def test_gpu_materials_switch_off_and_errors(rtiow_scene):
    import raytracing_amd as rt

    scene = rt.Scene.generate(rt.KIND_RTIOW_MATERIALS, 0, 1)
    ubo = scene.camera.copy()
    ubo["frameCount"][0] = 1
    with rt.Renderer(160, 96) as r:
        r.set_max_depth(8)
        r.set_scene(scene, frame_count=1)
        r.dispatch(1)
        with_m = r.read_accum()
        r.upload_materials(None)
        r.dispatch(1)
        without = r.read_accum()
        # a new sphere upload drops the old materials
        r.upload_materials(scene.materials)
        r.upload_spheres(scene.spheres)
        r.dispatch(1)
        after_reupload = r.read_accum()
        with pytest.raises(rt.RtError):
            r.upload_materials(scene.materials[:-1])  # count differs from the spheres uploaded
        bad = scene.materials.copy()
        bad[0, 0] = 4.0
        with pytest.raises(rt.RtError):
            r.upload_materials(bad)
        bad = scene.materials.copy()
        bad[0] = (3.0, 0.0)  # dielectric needs a positive index
        with pytest.raises(rt.RtError):
            r.upload_materials(bad)
    cplain, _, _ = O.render(ubo, scene.spheres, 160, 96, max_depth=8, threads=8)
    cmat, _, _ = O.render(ubo, scene.spheres, 160, 96, max_depth=8, threads=8, materials=scene.materials)
    assert np.array_equal(_bits(without), _bits(cplain)) and np.array_equal(_bits(after_reupload), _bits(cplain))
    assert np.array_equal(_bits(with_m), _bits(cmat))
