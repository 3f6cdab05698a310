// engine.cpp -- see engine.hpp.
#include "engine.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace rt {

void check(int status)
{
    if (status != RT_OK) throw std::runtime_error(rt_last_error());
}

namespace {

int envInt(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}

}  // namespace

// ------------------------------------------------------------ ComputePipeline

ComputePipeline::ComputePipeline(int device, uint32_t width, uint32_t height, Scene& scene)
    : m_scene(scene), m_width(width), m_height(height)
{
    check(rt_create(device, (int)width, (int)height, &m_ctx));
    // The shader hard-codes 50 bounces (res/shaders/shader.comp:95); RT_MAX_DEPTH, RT_PROFILE and
    // RT_KERNEL override the defaults for experiments.
    check(rt_set_max_depth(m_ctx, envInt("RT_MAX_DEPTH", 50)));
    check(rt_set_profile(m_ctx, envInt("RT_PROFILE", RT_PROFILE_STRICT)));
    check(rt_set_kernel(m_ctx, envInt("RT_KERNEL", RT_KERNEL_STREAM)));
}

ComputePipeline::~ComputePipeline() { rt_destroy(m_ctx); }

void ComputePipeline::updateScene()
{
    const std::vector<Sphere>& s = m_scene.spheres();
    const bool same = m_haveUpload && s.size() == m_uploaded.size() &&
                      (s.empty() || std::memcmp(s.data(), m_uploaded.data(), s.size() * sizeof(Sphere)) == 0);
    if (!same) {
        check(rt_upload_spheres(m_ctx, s.data(), (int)s.size()));
        m_uploaded = s;
        m_haveUpload = true;
        m_uploadedMaterials.clear();  // a sphere upload drops the device's materials
    }
    // material extension: parallel to the spheres, uploaded when it says anything and has changed
    m_scene.syncMaterials();
    const std::vector<Material>& m = m_scene.m_materials;
    bool any = false;
    for (const Material& x : m) any = any || x.kind != 0.f;
    const bool sameMaterials = m.size() == m_uploadedMaterials.size() &&
                               (m.empty() || std::memcmp(m.data(), m_uploadedMaterials.data(), m.size() * sizeof(Material)) == 0);
    if (any && !sameMaterials) {
        check(rt_upload_materials(m_ctx, &m[0].kind, (int)m.size()));
        m_uploadedMaterials = m;
    } else if (!any && !m_uploadedMaterials.empty()) {
        check(rt_upload_materials(m_ctx, nullptr, 0));
        m_uploadedMaterials.clear();
    }
    check(rt_set_uniforms(m_ctx, &m_scene.camera()));
}

void ComputePipeline::render(uint32_t /*imageIndex*/, uint32_t /*currentFrame*/)
{
    updateScene();
    check(rt_dispatch(m_ctx));
}

void ComputePipeline::windowResized(uint32_t width, uint32_t height)
{
    check(rt_resize(m_ctx, (int)width, (int)height));
    m_width = width;
    m_height = height;
    m_scene.resetFrameCount();
}

// ----------------------------------------------------------- GraphicsPipeline

void GraphicsPipeline::pressReset() { m_pending.push_back({0, 0, {0.f, 0.f, 0.f}}); }

void GraphicsPipeline::dragCamera(float x, float y, float z) { m_pending.push_back({1, 0, {x, y, z}}); }

void GraphicsPipeline::dragSphere(uint32_t index, float x, float y, float z)
{
    m_pending.push_back({2, index, {x, y, z}});
}

void GraphicsPipeline::render(uint32_t /*imageIndex*/, uint32_t /*currentFrame*/)
{
    auto slide = [](float& value, float to, float lo, float hi) {
        if (to != to) return;  // NaN: this slider is not touched
        value = to < lo ? lo : to > hi ? hi : to;
    };
    for (const Action& a : m_pending) {
        if (a.kind == 0) {
            m_scene.m_camera.frameCount = 0;
        } else if (a.kind == 1) {
            const vec3 old = m_scene.m_camera.camera_position;
            vec3& p = m_scene.m_camera.camera_position;
            slide(p.x, a.v[0], -100.0f, 100.0f);
            slide(p.y, a.v[1], -100.0f, 100.0f);
            slide(p.z, a.v[2], -100.0f, 100.0f);
            if (old.x != p.x || old.y != p.y || old.z != p.z) m_scene.m_camera.frameCount = 0;
        } else if (a.index < m_scene.m_spheres.size()) {
            vec3& c = m_scene.m_spheres[a.index].center;
            slide(c.x, a.v[0], -10.0f, 10.0f);
            slide(c.y, a.v[1], -10.0f, 10.0f);
            slide(c.z, a.v[2], -10.0f, 10.0f);
        }
    }
    m_pending.clear();
}

// --------------------------------------------------------------------- Engine

Engine::Engine(uint32_t width, uint32_t height, void* window, Scene& scene)
    : m_pendingWidth(width), m_pendingHeight(height)
{
    if (window != nullptr)
        throw std::runtime_error("this build is headless: Engine accepts only a null window");
    m_computePipeline = new ComputePipeline(envInt("RT_DEVICE", 0), width, height, scene);
    m_graphicsPipeline = new GraphicsPipeline(scene);
}

Engine::~Engine()
{
    if (m_computePipeline) {
        rt_sync(m_computePipeline->context());
        delete m_computePipeline;
    }
    delete m_graphicsPipeline;
}

void Engine::render()
{
    // engine.cpp:48: wait for the fence of the frame slot about to be reused -- with two frames in
    // flight that is the dispatch before the last one; the last one may still be running
    check(rt_sync_dispatch(m_computePipeline->context(), 1));
    m_computePipeline->render(0, m_currentFrame);
    m_graphicsPipeline->render(0, m_currentFrame);  // engine.cpp:68: after the scene went to the device
    if (m_framebufferResized) {  // engine.cpp:123-127
        m_framebufferResized = false;
        m_computePipeline->windowResized(m_pendingWidth, m_pendingHeight);
    }
    m_currentFrame = (m_currentFrame + 1) % 2;  // MAX_FRAMES_IN_FLIGHT, includes/config.hpp:12
}

void Engine::setFramebufferSize(uint32_t width, uint32_t height)
{
    m_pendingWidth = width;
    m_pendingHeight = height;
}

void Engine::waitIdle() { check(rt_sync(m_computePipeline->context())); }

std::vector<uint8_t> Engine::readColor()
{
    std::vector<uint8_t> out((size_t)m_computePipeline->width() * m_computePipeline->height() * 4);
    check(rt_read_color(m_computePipeline->context(), out.data()));
    return out;
}

std::vector<float> Engine::readAccumulation()
{
    std::vector<float> out((size_t)m_computePipeline->width() * m_computePipeline->height() * 4);
    check(rt_read_accum(m_computePipeline->context(), out.data()));
    return out;
}

void Engine::writePPM(const std::string& path)
{
    const std::vector<uint8_t> rgba = readColor();
    const uint32_t w = m_computePipeline->width(), h = m_computePipeline->height();
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write '" + path + "'");
    f << "P6\n" << w << " " << h << "\n255\n";
    std::vector<uint8_t> row((size_t)w * 3);
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) std::memcpy(&row[(size_t)x * 3], &rgba[((size_t)y * w + x) * 4], 3);
        f.write(reinterpret_cast<const char*>(row.data()), (std::streamsize)row.size());
    }
    if (!f) throw std::runtime_error("write to '" + path + "' failed");
}

}  // namespace rt
