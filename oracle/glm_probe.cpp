// glm_probe.cpp -- prints, as hex bit patterns, what the glm 1.0.0 headers vendored in
// the reference (lib/glm) compute for the host-side camera math.  TEST INFRASTRUCTURE
// ONLY.  It is compiled against the headers where they lie under the reference tree
// (oracle/Makefile, target _ref/glm_probe); nothing of the reference is copied.
//
// The expressions evaluated are the ones the reference host evaluates:
//   basis : src/application.cpp:10-23  (forward/right/up from yaw, pitch)
//   update: includes/scene.hpp:106-116 (camera integrator + "moved" test)
// Output: one line per case, fields separated by spaces, floats as 8 hex digits.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include <glm.hpp>

static uint32_t bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static void basis(float yaw, float pitch)
{
    glm::vec3 f;
    f.x = cos(glm::radians(yaw)) * cos(glm::radians(pitch));
    f.y = sin(glm::radians(pitch));
    f.z = sin(glm::radians(yaw)) * cos(glm::radians(pitch));
    f = glm::normalize(f);
    glm::vec3 r = glm::normalize(glm::cross(f, glm::vec3(0.0f, 1.0f, 0.0f)));
    glm::vec3 u = glm::normalize(glm::cross(r, f));
    std::printf("basis %08x %08x %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", bits(yaw),
                bits(pitch), bits(f.x), bits(f.y), bits(f.z), bits(r.x), bits(r.y), bits(r.z),
                bits(u.x), bits(u.y), bits(u.z));
}

int main()
{
    const float cases[][2] = { { 90.0f, 0.0f },   { 0.0f, 0.0f },    { 45.0f, 30.0f },
                               { -120.5f, 89.0f }, { 270.25f, -89.0f }, { 12.75f, -33.5f } };
    for (auto& c : cases) basis(c[0], c[1]);

    // Integrator: position from the default scene, an impulse on the first two steps.
    glm::vec3 pos(2.1030166f, 5.258916f, -46.312107f), vel(0.0f), acc(0.0f);
    uint32_t frame = 0;
    const float dt = 1.0f / 60.0f;
    for (int step = 0; step < 12; step++) {
        if (step < 2) acc += glm::vec3(-4.371139e-08f, 0.0f, 1.0f) * 100.0f;
        frame++;
        glm::vec3 old = pos;
        vel += acc * dt;
        pos += vel * dt;
        vel = vel * 0.98f;
        acc = acc * 0.98f;
        if (glm::length(old - pos) > 0.01f) frame = 1;
        std::printf("update %d %08x %08x %08x %08x %08x %08x %08x %08x %08x %u\n", step, bits(pos.x),
                    bits(pos.y), bits(pos.z), bits(vel.x), bits(vel.y), bits(vel.z), bits(acc.x),
                    bits(acc.y), bits(acc.z), frame);
    }
    return 0;
}
