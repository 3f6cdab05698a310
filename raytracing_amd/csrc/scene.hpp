// scene.hpp -- host scene model with the interface of the reference's Scene class
// (includes/scene.hpp:100-129): the same public members and methods, the same GPU-layout
// structs, the same YAML schema, minus yaml-cpp and glm.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {

struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
inline vec3& operator-=(vec3& a, vec3 b) { a = a - b; return a; }
float dot(vec3 a, vec3 b);    // x, then y, then z
float length(vec3 a);         // sqrt(dot(a, a))
vec3 normalize(vec3 a);       // a * (1 / sqrt(dot(a, a)))
vec3 cross(vec3 a, vec3 b);
float radians(float degrees);

// std140 uniform block of the kernel (reference includes/scene.hpp:22-29): 80 bytes,
// sphereCount at 60, frameCount at 64.
// The padding the reference gets from alignas(16) is spelled out so that copies carry it.
struct alignas(16) UniformBufferObject {
    vec3 camera_forward;
    float _pad0 = 0.f;
    vec3 camera_right;
    float _pad1 = 0.f;
    vec3 camera_up;
    float _pad2 = 0.f;
    vec3 camera_position;
    int sphereCount = 0;
    uint32_t frameCount = 0;
    uint32_t _pad3[3] = {0, 0, 0};
};
static_assert(sizeof(UniformBufferObject) == 80, "UBO layout");

// std430 sphere (reference includes/scene.hpp:31-36): 32 bytes.
struct Sphere {
    alignas(16) vec3 center;
    alignas(4) float radius = 0.f;
    alignas(16) vec3 color;
    alignas(4) float padding = 0.f;
};
static_assert(sizeof(Sphere) == 32, "Sphere layout");

// Material extension: 0 reference response, 1 lambertian, 2 metal (param = fuzz),
// 3 dielectric (param = index of refraction).
struct Material {
    float kind = 0.f;
    float param = 0.f;
};
static_assert(sizeof(Material) == 8, "Material layout");

struct SceneError : std::runtime_error {
    int code;  // RT_E_IO or RT_E_PARSE
    SceneError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class Scene {
   public:
    // The reference constructor reads "<cwd>/../res/scenes/scene.yaml" (src/scene.cpp:10-11).
    // Here the path is $RT_SCENE if set, else that same relative path.
    Scene();
    // Load `path`.  grow_seed != 0 reproduces the reference's habit of appending sphereCount
    // random spheres at load (src/scene.cpp:20-35) with a seeded generator instead of
    // std::random_device; they land beyond sphereCount and are never traced.
    explicit Scene(const std::string& path, uint32_t grow_seed = 0);
    static Scene fromText(const std::string& yaml);
    static Scene empty();
    ~Scene();

    const std::vector<Sphere>& spheres() const { return m_spheres; }
    const UniformBufferObject& camera() const { return m_camera; }
    void update(float dt);     // includes/scene.hpp:106-116
    void reloadScene();        // src/scene.cpp:48-55
    void resetFrameCount() { m_camera.frameCount = 0; }
    void save();               // src/scene.cpp:57-72, to the path the scene was loaded from
    void saveTo(const std::string& path) const;
    std::string toYaml() const;

    // Camera basis from yaw/pitch in degrees (src/application.cpp:10-23).
    void applyOrientation();

    std::vector<Sphere> m_spheres;
    UniformBufferObject m_camera;
    // Material extension (the reference's Sphere has no material): kept parallel to m_spheres by
    // every loader and generator; kind 0 = the reference's surface response.
    std::vector<Material> m_materials;
    void syncMaterials() { m_materials.resize(m_spheres.size()); }
    float mouseSensitivity = 0.25f;
    float movementSpeed = 100.0f;
    vec3 acceleration = vec3(0.0f);
    vec3 velocity = vec3(0.0f);
    float yaw = 90.0f;
    float pitch = 0.0f;

    std::string path;  // where reloadScene()/save() read and write

   private:
    struct NoLoad {};
    explicit Scene(NoLoad);
    void parseText(const std::string& text);
};

// Seeded synthetic scenes in the same representation.
Scene makeRtiowScene(uint32_t seed, bool with_materials = false);
Scene makeCubeScene(int count, uint32_t seed);

}  // namespace rt
