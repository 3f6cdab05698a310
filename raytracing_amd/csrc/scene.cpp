// scene.cpp -- see scene.hpp.  Behaviour follows the reference's src/scene.cpp and
// includes/scene.hpp; arithmetic follows the glm 1.0.0 expressions the reference evaluates
// (lib/glm/detail/func_geometric.inl: dot sums x+y+z, normalize = v * (1/sqrt(dot)),
// cross as written there; radians(d) = d * 0.0174532925199432957...).
#include "scene.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "rt_hip.h"
#include "scene_yaml.hpp"

namespace rt {

float dot(vec3 a, vec3 b)
{
    const float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z;
    return x + y + z;
}
float length(vec3 a) { return std::sqrt(dot(a, a)); }
vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
vec3 cross(vec3 x, vec3 y)
{
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }

namespace {

std::string defaultPath()
{
    const char* env = std::getenv("RT_SCENE");
    return env && *env ? std::string(env) : std::string("../res/scenes/scene.yaml");
}

std::string readFile(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError(RT_E_IO, "cannot open scene file '" + path + "'");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// includes/scene.hpp:48-54: a vec3 is a 3-sequence of floats.
vec3 decodeVec3(const yaml::Node& n, const char* what)
{
    vec3 v;
    if (!n.is_sequence() || n.size() != 3 || !yaml::to_float(n.items[0], &v.x) ||
        !yaml::to_float(n.items[1], &v.y) || !yaml::to_float(n.items[2], &v.z))
        throw SceneError(RT_E_PARSE, std::string("bad vec3 in ") + what);
    return v;
}

// Counter-based generator for the synthetic scenes: value n of stream `seed`.
uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct Rng {
    uint32_t seed, n = 0;
    explicit Rng(uint32_t s) : seed(s) {}
    float next() { return (float)(mix32(seed ^ mix32(++n)) >> 8) * (1.0f / 16777216.0f); }  // [0,1)
    float range(float lo, float hi) { return lo + (hi - lo) * next(); }
};

void emitVec3(std::string& out, const char* indent, vec3 v)
{
    for (float c : {v.x, v.y, v.z}) {
        out += indent;
        out += "- ";
        out += yaml::format_float(c);
        out += '\n';
    }
}

}  // namespace

Scene::Scene() : Scene(defaultPath()) {}

Scene::Scene(NoLoad) { std::memset(static_cast<void*>(&m_camera), 0, sizeof m_camera); }

Scene::Scene(const std::string& p, uint32_t grow_seed) : path(p)
{
    parseText(readFile(p));
    if (grow_seed != 0) {
        // src/scene.cpp:20-35 with a seeded generator
        Rng rng(grow_seed);
        for (int i = 0; i < m_camera.sphereCount; i++) {
            Sphere s;
            const float x = rng.range(-15.0f, 15.0f), y = rng.range(-15.0f, 15.0f),
                        z = rng.range(-15.0f, 15.0f);
            const float r = rng.next(), g = rng.next(), b = rng.next();
            s.center = vec3(x, y, z);
            s.radius = rng.range(0.5f, 3.0f);
            s.color = vec3(r, g, b);
            m_spheres.push_back(s);
        }
    }
    // src/scene.cpp:36-42 overwrites spheres 0 and 1; the values equal what the default file
    // already holds, so it is applied only when such entries exist.
    if (m_spheres.size() >= 2 && grow_seed != 0) {
        m_spheres[0].color = vec3(1.0f, 0.0f, 0.0f);
        m_spheres[0].center = vec3(0.0f, -20.0f, 0.0f);
        m_spheres[0].radius = 10.0f;
        m_spheres[1].color = vec3(0.0f, 1.0f, 0.0f);
        m_spheres[1].center = vec3(20.0f, 20.0f, 0.0f);
        m_spheres[1].radius = 10.0f;
    }
    syncMaterials();
}

Scene Scene::fromText(const std::string& text)
{
    Scene s{NoLoad{}};
    s.parseText(text);
    return s;
}

Scene Scene::empty() { return Scene{NoLoad{}}; }

Scene::~Scene() { m_spheres.clear(); }

void Scene::parseText(const std::string& text)
{
    yaml::Node root;
    std::string err;
    if (!yaml::parse(text, &root, &err)) throw SceneError(RT_E_PARSE, "scene YAML: " + err);
    const yaml::Node* spheres = root.find("spheres");
    const yaml::Node* camera = root.find("camera");
    if (!spheres || !spheres->is_sequence()) throw SceneError(RT_E_PARSE, "scene YAML: no 'spheres' sequence");
    if (!camera) throw SceneError(RT_E_PARSE, "scene YAML: no 'camera' entry");

    std::vector<Sphere> list;
    std::vector<Material> materials;
    list.reserve(spheres->size());
    materials.reserve(spheres->size());
    for (const yaml::Node& n : spheres->items) {
        // includes/scene.hpp:89-95: [center, radius, color]; the material extension allows a
        // 4th element [kind, param]
        Sphere s;
        Material m;
        if (!n.is_sequence() || (n.size() != 3 && n.size() != 4))
            throw SceneError(RT_E_PARSE, "sphere is not a 3-sequence");
        s.center = decodeVec3(n.items[0], "sphere center");
        if (!yaml::to_float(n.items[1], &s.radius)) throw SceneError(RT_E_PARSE, "bad sphere radius");
        s.color = decodeVec3(n.items[2], "sphere color");
        if (n.size() == 4) {
            const yaml::Node& mat = n.items[3];
            if (!mat.is_sequence() || mat.size() != 2 || !yaml::to_float(mat.items[0], &m.kind) ||
                !yaml::to_float(mat.items[1], &m.param) ||
                !(m.kind == 0.f || m.kind == 1.f || m.kind == 2.f || m.kind == 3.f) ||
                (m.kind == 3.f && !(m.param > 0.f)))
                throw SceneError(RT_E_PARSE, "bad sphere material: expected [kind 0..3, param] (index of refraction > 0 for kind 3)");
        }
        list.push_back(s);
        materials.push_back(m);
    }
    // includes/scene.hpp:68-78: [position, forward, right, up, sphereCount, ignored]
    if (!camera->is_sequence() || camera->size() != 6) throw SceneError(RT_E_PARSE, "camera is not a 6-sequence");
    const vec3 position = decodeVec3(camera->items[0], "camera position");
    const vec3 forward = decodeVec3(camera->items[1], "camera forward");
    const vec3 right = decodeVec3(camera->items[2], "camera right");
    const vec3 up = decodeVec3(camera->items[3], "camera up");
    int count = 0;
    if (!yaml::to_int(camera->items[4], &count)) throw SceneError(RT_E_PARSE, "bad camera sphereCount");
    m_spheres = std::move(list);
    m_materials = std::move(materials);
    std::memset(static_cast<void*>(&m_camera), 0, sizeof m_camera);  // padding travels to the device too
    m_camera.camera_position = position;
    m_camera.camera_forward = forward;
    m_camera.camera_right = right;
    m_camera.camera_up = up;
    m_camera.sphereCount = count;
    m_camera.frameCount = 0;
}

void Scene::update(float dt)
{
    m_camera.frameCount++;
    const vec3 old_position = m_camera.camera_position;
    velocity += acceleration * dt;
    m_camera.camera_position += velocity * dt;
    velocity = velocity * 0.98f;
    acceleration = acceleration * 0.98f;
    if (length(old_position - m_camera.camera_position) > 0.01f) m_camera.frameCount = 1;
}

void Scene::reloadScene()
{
    if (path.empty()) throw SceneError(RT_E_STATE, "scene has no file to reload from");
    parseText(readFile(path));
}

void Scene::applyOrientation()
{
    vec3 f;
    f.x = std::cos(radians(yaw)) * std::cos(radians(pitch));
    f.y = std::sin(radians(pitch));
    f.z = std::sin(radians(yaw)) * std::cos(radians(pitch));
    m_camera.camera_forward = normalize(f);
    m_camera.camera_right = normalize(cross(m_camera.camera_forward, vec3(0.0f, 1.0f, 0.0f)));
    m_camera.camera_up = normalize(cross(m_camera.camera_right, m_camera.camera_forward));
}

// Block style, one scalar per line, no trailing newline: the layout yaml-cpp's emitter gives
// the reference's save() (res/scenes/scene.yaml).
std::string Scene::toYaml() const
{
    std::string out;
    out.reserve(m_spheres.size() * 160 + 256);
    if (m_spheres.empty()) {
        out += "spheres:\n  []\n";
    } else {
        out += "spheres:\n";
        for (size_t i = 0; i < m_spheres.size(); i++) {
            const Sphere& s = m_spheres[i];
            out += "  -\n    -\n";
            emitVec3(out, "      ", s.center);
            out += "    - " + yaml::format_float(s.radius) + "\n    -\n";
            emitVec3(out, "      ", s.color);
            // material extension: a 4th element, only where it says something
            if (i < m_materials.size() && m_materials[i].kind != 0.f)
                out += "    -\n      - " + yaml::format_float(m_materials[i].kind) + "\n      - " +
                       yaml::format_float(m_materials[i].param) + "\n";
        }
    }
    out += "camera:\n";
    for (vec3 v : {m_camera.camera_position, m_camera.camera_forward, m_camera.camera_right, m_camera.camera_up}) {
        out += "  -\n";
        emitVec3(out, "    ", v);
    }
    out += "  - " + std::to_string(m_camera.sphereCount) + "\n  - 0";
    return out;
}

void Scene::saveTo(const std::string& p) const
{
    std::ofstream f(p, std::ios::binary);
    if (!f) throw SceneError(RT_E_IO, "cannot write scene file '" + p + "'");
    const std::string text = toYaml();
    f.write(text.data(), (std::streamsize)text.size());
    if (!f) throw SceneError(RT_E_IO, "write to '" + p + "' failed");
}

void Scene::save()
{
    if (path.empty()) throw SceneError(RT_E_STATE, "scene has no file to save to");
    saveTo(path);
}

// ------------------------------------------------------------ synthetic scenes

namespace {

void lookAt(Scene& s, vec3 from, vec3 at)
{
    s.m_camera.camera_position = from;
    s.m_camera.camera_forward = normalize(at - from);
    s.m_camera.camera_right = normalize(cross(s.m_camera.camera_forward, vec3(0.0f, 1.0f, 0.0f)));
    s.m_camera.camera_up = normalize(cross(s.m_camera.camera_right, s.m_camera.camera_forward));
}

Sphere sphere(vec3 c, float r, vec3 col)
{
    Sphere s;
    s.center = c;
    s.radius = r;
    s.color = col;
    return s;
}

}  // namespace

// The layout of the "Ray Tracing in One Weekend" cover scene: three unit spheres, a
// radius-1000 ground sphere and a 22x22 grid of jittered radius-0.2 spheres, about 488 in
// all.  The shader has one surface response and lights spheres 0-2, so the three large
// spheres take those indices.
//
// with_materials = false: every sphere keeps the reference's surface response (kind 0) and the
// palette only hints at the intended material.  with_materials = true: the material extension
// assigns glass / matte / metal to the three large spheres, matte to the ground, and
// matte 80 % / metal 15 % (fuzz U(0,0.5)) / glass 5 % to the small ones.
Scene makeRtiowScene(uint32_t seed, bool with_materials)
{
    Scene s = Scene::empty();
    Rng rng(seed);
    auto mat = [&](float kind, float param) {
        Material m;
        if (with_materials) { m.kind = kind; m.param = param; }
        s.m_materials.push_back(m);
    };
    s.m_spheres.push_back(sphere(vec3(0.0f, 1.0f, 0.0f), 1.0f, vec3(0.95f, 0.95f, 0.9f)));
    mat(3.f, 1.5f);
    s.m_spheres.push_back(sphere(vec3(-4.0f, 1.0f, 0.0f), 1.0f, vec3(0.4f, 0.2f, 0.1f)));
    mat(1.f, 0.f);
    s.m_spheres.push_back(sphere(vec3(4.0f, 1.0f, 0.0f), 1.0f, vec3(0.7f, 0.6f, 0.5f)));
    mat(2.f, 0.f);
    s.m_spheres.push_back(sphere(vec3(0.0f, -1000.0f, 0.0f), 1000.0f, vec3(0.5f, 0.5f, 0.5f)));
    mat(1.f, 0.f);
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            const float choose = rng.next();
            const vec3 c((float)a + 0.9f * rng.next(), 0.2f, (float)b + 0.9f * rng.next());
            const float r0 = rng.next(), r1 = rng.next(), r2 = rng.next(), r3 = rng.next(),
                        r4 = rng.next(), r5 = rng.next();
            if (length(c - vec3(4.0f, 0.2f, 0.0f)) <= 0.9f) continue;
            vec3 col;
            if (choose < 0.8f) {
                col = vec3(r0 * r1, r2 * r3, r4 * r5);  // matte palette
                mat(1.f, 0.f);
            } else if (choose < 0.95f) {
                col = vec3(0.5f + 0.5f * r0, 0.5f + 0.5f * r2, 0.5f + 0.5f * r4);  // metal palette
                mat(2.f, 0.5f * r1);
            } else {
                col = vec3(0.95f, 0.95f, 0.95f);  // glass
                mat(3.f, 1.5f);
            }
            s.m_spheres.push_back(sphere(c, 0.2f, col));
        }
    s.m_camera.sphereCount = (int)s.m_spheres.size();
    lookAt(s, vec3(8.0f, 1.8f, 2.6f), vec3(0.0f, 0.6f, 0.0f));
    return s;
}

// `count` spheres with the distributions the reference appends at load (src/scene.cpp:21-35):
// centres U(-15,15)^3, radius U(0.5,3), colour U(0,1)^3; spheres 0-2 are the three lights of
// the default scene; camera at the default scene's position with the launch orientation.
Scene makeCubeScene(int count, uint32_t seed)
{
    Scene s = Scene::empty();
    Rng rng(seed);
    for (int i = 0; i < count; i++) {
        const float x = rng.range(-15.0f, 15.0f), y = rng.range(-15.0f, 15.0f), z = rng.range(-15.0f, 15.0f);
        const float r = rng.next(), g = rng.next(), b = rng.next();
        const float radius = rng.range(0.5f, 3.0f);
        s.m_spheres.push_back(sphere(vec3(x, y, z), radius, vec3(r, g, b)));
    }
    const Sphere lights[3] = {sphere(vec3(0.0f, -20.0f, 0.0f), 10.0f, vec3(1.0f, 0.0f, 0.0f)),
                              sphere(vec3(20.0f, 20.0f, 0.0f), 10.0f, vec3(0.0f, 1.0f, 0.0f)),
                              sphere(vec3(-20.0f, 20.0f, 0.0f), 10.0f, vec3(0.0f, 0.0f, 1.0f))};
    for (int i = 0; i < 3 && i < count; i++) s.m_spheres[i] = lights[i];
    s.m_camera.sphereCount = count;
    s.m_camera.camera_position = vec3(2.1030166f, 5.258916f, -46.312107f);
    s.applyOrientation();
    s.syncMaterials();
    return s;
}

}  // namespace rt
