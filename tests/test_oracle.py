"""CPU tests of the oracle itself: it must be pinned before it is trusted as a checker.

Integer RNG: hand-evaluated known-answer vectors (tests/golden/rng_kat.json).
Float path : no reference output exists ("parity unpinned"); what is checked here is that the
oracle is deterministic, agrees with its committed golden vectors, and has the structural
properties the shader has (frame-1 clear, division by frameCount+1, unrendered remainder rows,
pixel/frame independence).
"""
import json
import os

import numpy as np
import pytest

import oracle as O
from conftest import GOLDEN


def _h(x):
    return int(x, 16)


def test_wang_hash_kat():
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for src, dst in kat["wang_hash"]:
        assert O.wang_hash(_h(src)) == _h(dst), src


def test_seed_kat_and_wraparound():
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for x, y, frame, bounce, s, h0, h1, h2 in kat["seed"]:
        sd = O.seed(x, y, frame, bounce)
        assert sd == _h(s)
        assert O.wang_hash(sd) == _h(h0)
        assert O.wang_hash((sd + 1) & 0xFFFFFFFF) == _h(h1)
        assert O.wang_hash((sd + 1920) & 0xFFFFFFFF) == _h(h2)


def test_seed_aliasing_of_the_hard_coded_stride():
    # the RNG row stride is 1920 whatever the image width, and 256 frames alias one bounce
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for a, b, s in kat["alias"]:
        assert O.seed(*a) == O.seed(*b) == _h(s)


def test_hash_to_float_reaches_one():
    f = O.lib().rto_hash_to_float
    assert f(0) == 0.0
    assert f(0xFFFFFFFF) == 1.0  # uint->float rounds to nearest even
    assert f(0x80000000) == 0.5
    assert f(0xFFFFFF00) == np.float32(0xFFFFFF00) / np.float32(4294967296.0)


def test_unorm8():
    u = O.lib().rto_unorm8
    assert u(0.0) == 0 and u(-1.0) == 0 and u(float("nan")) == 0
    assert u(1.0) == 255 and u(7.5) == 255 and u(float("inf")) == 255
    assert u(0.5) == 128  # floor(127.5 + 0.5)
    assert u(0.2) == 51


def test_layouts():
    import raytracing_amd as rt

    assert rt.SPHERE_DTYPE.itemsize == 32
    assert rt.UBO_DTYPE.itemsize == 80
    assert rt.UBO_DTYPE.fields["sphereCount"][1] == 60
    assert rt.UBO_DTYPE.fields["frameCount"][1] == 64
    assert rt.UBO_DTYPE.fields["camera_position"][1] == 48


@pytest.mark.parametrize("profile", ["strict", "fused"])
def test_golden_crop(default_scene, profile):
    gold = np.load(os.path.join(GOLDEN, "default_crop16.npz"))
    prof = O.STRICT if profile == "strict" else O.FUSED
    deepest = 0
    for frame in (1, 2, 64):
        ubo = default_scene.camera.copy()
        ubo["frameCount"][0] = frame
        rec, light = O.trace_crop(ubo, default_scene.spheres, 1280, 720, 604, 197, 16, 16, profile=prof)
        assert np.array_equal(rec["hit"][0], gold[f"{profile}_hit_f{frame}"].astype(np.int32))
        assert np.array_equal(rec["hash"][0], gold[f"{profile}_hash_f{frame}"])
        assert np.array_equal(light[0].view(np.uint32), gold[f"{profile}_light_f{frame}"].view(np.uint32))
        deepest = max(deepest, int((rec["hit"][0] >= 0).sum(axis=-1).max()))
    assert deepest >= 3, "the golden crop must contain multi-bounce paths"


@pytest.mark.parametrize("profile", ["strict", "fused"])
def test_golden_image(default_scene, profile):
    gold = np.load(os.path.join(GOLDEN, "default_64x36.npz"))
    prof = O.STRICT if profile == "strict" else O.FUSED
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    acc, col, rays = O.render(ubo, default_scene.spheres, 64, 36, profile=prof, n_frames=64, threads=4)
    assert np.array_equal(acc.view(np.uint32), gold[f"{profile}_accum_f64"].view(np.uint32))
    assert np.array_equal(col, gold[f"{profile}_color_f64"])
    assert rays == int(gold[f"{profile}_rays"][1])
    # 36 is not a multiple of 8: rows 32..35 are never touched (compute_pipeline.cpp:430-432)
    assert not acc[32:].any() and not col[32:].any()
    # alpha counts frames since the clear (shader.comp:127)
    assert np.all(acc[:32, :, 3] == 64.0)


def test_frames_are_independent_and_order_preserving(default_scene):
    """64 frames in one call == 64 calls of one frame, bit for bit."""
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    acc_all, col_all, rays_all = O.render(ubo, default_scene.spheres, 32, 16, n_frames=12)
    acc = None
    rays = 0
    for f in range(1, 13):
        ubo["frameCount"][0] = f
        acc, col, r = O.render(ubo, default_scene.spheres, 32, 16, n_frames=1, accum=acc)
        rays += r
    assert np.array_equal(acc.view(np.uint32), acc_all.view(np.uint32))
    assert np.array_equal(col, col_all)
    assert rays == rays_all


def test_clear_only_on_frame_one(default_scene):
    ubo = default_scene.camera.copy()
    junk = np.full((16, 32, 4), 7.0, np.float32)
    ubo["frameCount"][0] = 1
    acc1, _, _ = O.render(ubo, default_scene.spheres, 32, 16, accum=junk)
    assert np.all(acc1[..., 3] == 1.0)
    ubo["frameCount"][0] = 5
    acc5, col5, _ = O.render(ubo, default_scene.spheres, 32, 16, accum=junk)
    assert np.all(acc5[..., 3] == 8.0)
    # the displayed colour divides by frameCount + 1, not by the sample count (shader.comp:129)
    expect = np.clip(acc5[..., :3] / np.float32(6.0), 0, 1)
    assert np.array_equal(col5[..., :3], np.floor(expect * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))


def test_threads_and_tile_rows_do_not_change_bits(default_scene):
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    a1, c1, r1 = O.render(ubo, default_scene.spheres, 64, 48, n_frames=3, threads=1)
    a4, c4, r4 = O.render(ubo, default_scene.spheres, 64, 48, n_frames=3, threads=4)
    assert np.array_equal(a1.view(np.uint32), a4.view(np.uint32)) and np.array_equal(c1, c4) and r1 == r4
    top, _, rt_ = O.render(ubo, default_scene.spheres, 64, 48, n_frames=3, rows=(0, 2))
    bot, _, rb = O.render(ubo, default_scene.spheres, 64, 48, n_frames=3, rows=(2, 6))
    assert not top[16:].any() and not bot[:16].any()
    assert np.array_equal((top + bot).view(np.uint32), a1.view(np.uint32)) and rt_ + rb == r1


def test_strided_frames_partition_the_sample_set(default_scene):
    """Frames 1..8 split over two ranks by stride 2 hold the same per-frame light values."""
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    _, full = O.trace_crop(ubo, default_scene.spheres, 64, 48, 24, 16, 8, 8, n_frames=8)
    _, even = O.trace_crop(ubo, default_scene.spheres, 64, 48, 24, 16, 8, 8, n_frames=4, frame_stride=2)
    ubo["frameCount"][0] = 2
    _, odd = O.trace_crop(ubo, default_scene.spheres, 64, 48, 24, 16, 8, 8, n_frames=4, frame_stride=2)
    assert np.array_equal(full[0::2].view(np.uint32), even.view(np.uint32))
    assert np.array_equal(full[1::2].view(np.uint32), odd.view(np.uint32))


def test_profiles_agree_within_image_tolerance(default_scene):
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    _, cs, _ = O.render(ubo, default_scene.spheres, 128, 72, n_frames=32, threads=4)
    _, cf, _ = O.render(ubo, default_scene.spheres, 128, 72, n_frames=32, threads=4, profile=O.FUSED)
    mse = np.mean((cs[:72, :, :3].astype(np.float64) - cf[:72, :, :3].astype(np.float64)) ** 2)
    psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    assert psnr > 35.0, psnr


def test_empty_scene_and_zero_count(default_scene):
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    ubo["sphereCount"][0] = 0
    acc, col, rays = O.render(ubo, default_scene.spheres, 16, 8)
    # every primary ray misses: light = 0.15 * sky
    sky = np.float32(0.15) * np.array([0.6, 0.7, 0.9], np.float32)
    assert np.array_equal(acc[..., :3], np.broadcast_to(sky, (8, 16, 3)))
    assert rays == 16 * 8


def test_bad_arguments():
    import raytracing_amd as rt

    s = rt.Scene.default()
    ubo = s.camera.copy()
    ubo["sphereCount"][0] = 41  # more than the 40 in the buffer
    with pytest.raises(RuntimeError):
        O.render(ubo, s.spheres, 16, 8)


def test_converged_sky_pixel_equals_the_reference_screenshot(default_scene):
    """The one rendered output the reference ships is thumbnail.png, a screenshot at an unrecorded
    camera pose.  Its most frequent colour -- sky after many accumulated frames, the only value in
    it that does not depend on the pose -- is committed as tests/golden/thumbnail_sky.json
    (tools/make_thumbnail_fixture.py).  It pins sky * 0.15, the accumulate / (frameCount + 1)
    resolve and round-to-nearest of the rgba8 store: truncation would give (22, 26, 34)."""
    import json

    gold = json.load(open(os.path.join(GOLDEN, "thumbnail_sky.json")))
    ubo = default_scene.camera.copy()
    ubo["frameCount"][0] = 1
    ubo["sphereCount"][0] = 0  # every ray sees the sky
    for profile in (O.STRICT, O.FUSED):
        _, col, rays = O.render(ubo, default_scene.spheres, 16, 8, profile=profile, n_frames=256)
        assert rays == 16 * 8 * 256
        assert np.all(col[..., :3] == np.array(gold["sky_rgb"], np.uint8)), col[0, 0]
        assert np.all(col[..., 3] == 255)


def test_sky_mirrored_by_a_sphere_equals_the_reference_screenshot(default_scene):
    """The next two most frequent colours of thumbnail.png are large flat areas: spheres that mirror
    the sky.  Such a pixel is sky * (0.15 * albedo) whatever the pose, so each must be the one-bounce
    colour of exactly one of the 40 traced spheres of the reference's scene file -- which pins
    `contribution *= albedo`, `light += sky * contribution` and the resolve against the reference's
    own output."""
    import json

    gold = json.load(open(os.path.join(GOLDEN, "thumbnail_sky.json")))
    base = default_scene.spheres[:40].copy()
    found = []
    for entry in gold["surface_rgb"]:
        want = np.array(entry["rgb"], np.uint8)
        hits = []
        for i in range(3, 40):  # spheres 0-2 emit
            sph = np.zeros(4, base.dtype)
            sph[:3]["center"] = (3.0e4, 2.0e4, -1.0e4)   # the three emitters, far away and off every ray
            sph[:3]["radius"] = 1.0
            sph[3] = base[i]
            sph[3]["center"] = (0.0, 0.0, 10.0)
            sph[3]["radius"] = 2.0
            ubo = default_scene.camera.copy()
            ubo["camera_position"][0] = (0.0, 0.0, 0.0)
            ubo["sphereCount"][0] = 4
            ubo["frameCount"][0] = 1
            _, col, _ = O.render(ubo, sph, 16, 8, n_frames=256)
            if np.array_equal(col[4, 8, :3], want):  # centre pixel: hits the sphere head-on, bounces to the sky
                hits.append(i)
        assert len(hits) == 1, (entry, hits)
        found.append(hits[0])
    assert found == [11, 13]


def test_oracle_reproduces_the_reference_screenshot(default_scene):
    """thumbnail.png at the pose recovered for it (tools/recover_thumbnail_pose.py; five fitted
    numbers against 1.9 M pixels): primary visibility, mirror reflections of reflections, emitters
    and sky all land where the reference drew them.  Stated tolerance: PSNR >= 28.5 dB against the
    screenshot reduced 4x, >= 31.5 dB reduced 8x, sky/non-sky agreement >= 99 % of the reduced
    pixels.  What is left is misalignment at the edges of mirrored highlights and the blur of the
    close-up sphere, which the screenshot shows with a ring-shaped kernel (constant roughness, the
    commented-out line res/shaders/shader.comp:101) where the shipped shader draws it at random."""
    import thumbnail as T

    ubo = T.posed_ubo(default_scene)
    frames = 8
    acc, _, _ = O.render(ubo, default_scene.spheres, T.WIDTH, T.HEIGHT, n_frames=frames, threads=8)
    p4, p8, sky = T.compare(acc, frames)
    assert p4 >= 28.5 and p8 >= 31.5 and sky >= 0.99, (p4, p8, sky)


@pytest.mark.skipif(not os.path.exists("/root/reference/res/scenes/scene.yaml"), reason="needs the reference's 1400-sphere file")
def test_screenshot_says_forty_spheres_are_traced():
    """The reference's scene file holds 1400 spheres and camera[4] = 40; the shader loops to
    sphereCount.  Against the screenshot, tracing more than 40 is visibly wrong."""
    import raytracing_amd as rt
    import thumbnail as T

    scene = rt.Scene.load("/root/reference/res/scenes/scene.yaml")
    assert len(scene.spheres) == 1400 and scene.sphere_count == 40
    ubo = T.posed_ubo(scene)
    p8 = {}
    for count in (40, 48):
        ubo["sphereCount"][0] = count
        acc, _, _ = O.render(ubo, scene.spheres, T.WIDTH, T.HEIGHT, n_frames=4, threads=8)
        p8[count] = T.compare(acc, 4)[1]
    assert p8[40] >= 31.0 and p8[48] <= 25.0, p8


def test_screenshot_was_taken_with_the_constant_roughness_variant(default_scene):
    """What is left between the oracle and the screenshot sits on the close-up sphere, whose blur is
    ring-shaped in the screenshot.  The shader keeps a constant-roughness line commented out
    (res/shaders/shader.comp:101); with a constant 0.01 -- the mean of the shipped rand * 0.02 -- the
    oracle fits that sphere about 3 dB better, with 0.02 about 6 dB worse: the screenshot comes from
    that variant, and the residual is the roughness rule, not the path."""
    import thumbnail as T

    ubo = T.posed_ubo(default_scene)
    rec, _ = O.trace_crop(ubo, default_scene.spheres, T.WIDTH, T.HEIGHT, 0, 0, 1904, 1008, max_depth=1)
    close_up = T.box((rec["hit"][0, :, :, 0] == 11)[..., None].astype(np.float64), 4)[..., 0] == 1.0
    assert 0.3 < close_up.mean() < 0.45
    ref = T.reduced_screenshot()
    fit = {}
    for rough in (0.0, 0.01, 0.02):
        acc, _, _ = O.render(ubo, default_scene.spheres, T.WIDTH, T.HEIGHT, n_frames=6, threads=8, const_roughness=rough)
        img = T.box(np.clip(acc[:1008, :1904, :3] / np.float32(6), 0.0, 1.0).astype(np.float64) * 255.0, 4)
        fit[rough] = T.psnr(img[close_up], ref[close_up])
    assert fit[0.01] >= fit[0.0] + 1.5 and fit[0.02] <= fit[0.0] - 3.0, fit
    assert fit[0.01] >= 31.0, fit
