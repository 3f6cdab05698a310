// scene_api.cpp -- the rt_scene half of the C ABI (include/rt_hip.h): thin wrappers that turn
// rt::Scene's exceptions into status codes.
#include <new>
#include <string>

#include "rt_hip.h"
#include "scene.hpp"

void rt_set_error(const char* msg);  // rt_api.cpp

struct rt_scene {
    rt::Scene scene;
    explicit rt_scene(rt::Scene&& s) : scene(std::move(s)) {}
};

namespace {

template <class F>
int guarded(F&& f)
{
    try {
        return f();
    } catch (const rt::SceneError& e) {
        rt_set_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        rt_set_error("out of host memory");
        return RT_E_NOMEM;
    } catch (const std::exception& e) {
        rt_set_error(e.what());
        return RT_E_INVALID;
    }
}

int bad(const char* msg)
{
    rt_set_error(msg);
    return RT_E_INVALID;
}

}  // namespace

extern "C" {

int rt_scene_load(const char* path, rt_scene** out)
{
    if (!path || !out) return bad("null argument");
    *out = nullptr;
    return guarded([&] {
        *out = new rt_scene(rt::Scene(std::string(path)));
        return RT_OK;
    });
}

int rt_scene_parse(const char* text, size_t len, rt_scene** out)
{
    if (!text || !out) return bad("null argument");
    *out = nullptr;
    return guarded([&] {
        *out = new rt_scene(rt::Scene::fromText(std::string(text, len)));
        return RT_OK;
    });
}

int rt_scene_save(const rt_scene* s, const char* path)
{
    if (!s || !path) return bad("null argument");
    return guarded([&] {
        s->scene.saveTo(path);
        return RT_OK;
    });
}

int rt_scene_free(rt_scene* s)
{
    delete s;
    return RT_OK;
}

int rt_scene_sphere_count(const rt_scene* s) { return s ? (int)s->scene.m_spheres.size() : bad("null argument"); }

void* rt_scene_spheres(rt_scene* s) { return s ? (void*)s->scene.m_spheres.data() : nullptr; }

void* rt_scene_ubo(rt_scene* s) { return s ? (void*)&s->scene.m_camera : nullptr; }

float* rt_scene_materials(rt_scene* s)
{
    if (!s) return nullptr;
    s->scene.syncMaterials();
    return s->scene.m_materials.empty() ? nullptr : &s->scene.m_materials[0].kind;
}

int rt_scene_set_orientation(rt_scene* s, float yaw_deg, float pitch_deg)
{
    if (!s) return bad("null argument");
    s->scene.yaw = yaw_deg;
    s->scene.pitch = pitch_deg;
    s->scene.applyOrientation();
    return RT_OK;
}

int rt_scene_update(rt_scene* s, float dt)
{
    if (!s) return bad("null argument");
    s->scene.update(dt);
    return RT_OK;
}

int rt_scene_add_acceleration(rt_scene* s, float ax, float ay, float az)
{
    if (!s) return bad("null argument");
    s->scene.acceleration += rt::vec3(ax, ay, az);
    return RT_OK;
}

int rt_scene_get_motion(const rt_scene* s, float acceleration[3], float velocity[3])
{
    if (!s || !acceleration || !velocity) return bad("null argument");
    const rt::vec3 a = s->scene.acceleration, v = s->scene.velocity;
    acceleration[0] = a.x; acceleration[1] = a.y; acceleration[2] = a.z;
    velocity[0] = v.x; velocity[1] = v.y; velocity[2] = v.z;
    return RT_OK;
}

int rt_scene_reset_frame_count(rt_scene* s)
{
    if (!s) return bad("null argument");
    s->scene.resetFrameCount();
    return RT_OK;
}

int rt_scene_generate(int kind, int count, uint32_t seed, rt_scene** out)
{
    if (!out) return bad("null argument");
    *out = nullptr;
    if (kind < 0 || kind > 2) return bad("unknown scene kind");
    if (kind == 1 && (count < 0 || count > (1 << 24))) return bad("bad sphere count");
    return guarded([&] {
        *out = new rt_scene(kind == 1 ? rt::makeCubeScene(count, seed) : rt::makeRtiowScene(seed, kind == 2));
        return RT_OK;
    });
}

}  // extern "C"
