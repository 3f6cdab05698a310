// application.cpp -- see application.hpp.
#include "application.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <sstream>

namespace rt {

Application::Application(uint32_t width, uint32_t height, const char* /*name*/)
{
    m_scene = Scene();
    _engine = std::make_unique<Engine>(width, height, nullptr, m_scene);
    // src/application.cpp:10-23: the camera basis is rebuilt from yaw/pitch at launch; the
    // orientation rows of the scene file are discarded, its position row is kept.
    m_scene.applyOrientation();
}

Application::~Application()
{
    _engine.reset();
    if (saveOnExit) {
        try {
            m_scene.save();  // src/application.cpp:26
        } catch (const std::exception&) {
        }
    }
}

void Application::setScript(std::vector<InputEvent> events)
{
    std::stable_sort(events.begin(), events.end(),
                     [](const InputEvent& a, const InputEvent& b) { return a.frame < b.frame; });
    m_script = std::move(events);
    m_next = 0;
}

void Application::loadScript(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open input script '" + path + "'");
    std::vector<InputEvent> events;
    std::string line;
    int number = 0;
    while (std::getline(f, line)) {
        number++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream in(line);
        InputEvent e;
        std::string what;
        if (!(in >> e.frame)) continue;  // blank line
        auto bad = [&](const char* why) {
            throw std::runtime_error(path + ":" + std::to_string(number) + ": " + why);
        };
        if (!(in >> what)) bad("missing event kind");
        if (what == "key") {
            std::string key, state;
            if (!(in >> key >> state) || key.size() != 1) bad("expected: key <W|A|S|D|Q|E> <down|up>");
            e.key = (char)std::toupper((unsigned char)key[0]);
            if (std::string("WASDQE").find(e.key) == std::string::npos) bad("unknown key");
            if (state == "down") e.kind = InputEvent::KeyDown;
            else if (state == "up") e.kind = InputEvent::KeyUp;
            else bad("key state must be down or up");
        } else if (what == "mouse") {
            e.kind = InputEvent::MouseDrag;
            if (!(in >> e.dx >> e.dy)) bad("expected: mouse <dx> <dy>");
        } else if (what == "reload") {
            e.kind = InputEvent::Reload;
        } else if (what == "resize") {
            e.kind = InputEvent::Resize;
            if (!(in >> e.width >> e.height) || e.width == 0 || e.height == 0) bad("expected: resize <w> <h>");
        } else if (what == "ui") {
            std::string widget;
            if (!(in >> widget)) bad("expected: ui <reset|camera|sphere> ...");
            auto slider = [&](float& v) {
                std::string tok;
                if (!(in >> tok)) return false;
                if (tok == "=") {
                    v = std::numeric_limits<float>::quiet_NaN();
                    return true;
                }
                char* end = nullptr;
                v = std::strtof(tok.c_str(), &end);
                return end != tok.c_str() && *end == 0;
            };
            if (widget == "reset") {
                e.kind = InputEvent::UiReset;
            } else if (widget == "camera") {
                e.kind = InputEvent::UiCamera;
                if (!slider(e.value[0]) || !slider(e.value[1]) || !slider(e.value[2])) bad("expected: ui camera <x> <y> <z>");
            } else if (widget == "sphere") {
                e.kind = InputEvent::UiSphere;
                if (!(in >> e.index) || !slider(e.value[0]) || !slider(e.value[1]) || !slider(e.value[2]))
                    bad("expected: ui sphere <index> <x> <y> <z>");
            } else {
                bad("unknown ui widget");
            }
        } else if (what == "quit") {
            e.kind = InputEvent::Quit;
        } else {
            bad("unknown event kind");
        }
        events.push_back(e);
    }
    setScript(std::move(events));
}

void Application::Run()
{
    while (!m_quit && m_frame < maxFrames) {
        handleInput();
        dt = fixedDt;
        m_scene.update(dt);
        _engine->render();
        m_frame++;
    }
    _engine->waitIdle();
}

void Application::handleInput()
{
    // scripted state changes due this frame
    while (m_next < m_script.size() && m_script[m_next].frame <= m_frame) {
        const InputEvent& e = m_script[m_next++];
        switch (e.kind) {
            case InputEvent::KeyDown: m_keys[(unsigned char)e.key] = true; break;
            case InputEvent::KeyUp: m_keys[(unsigned char)e.key] = false; break;
            case InputEvent::Reload: m_scene.reloadScene(); break;  // src/application.cpp:39-41
            case InputEvent::Quit: m_quit = true; break;
            // the overlay's widgets: queued here, applied by GraphicsPipeline::render after this
            // frame's dispatch (src/engine/engine.cpp:67-68)
            case InputEvent::UiReset: _engine->graphicsPipeline().pressReset(); break;
            case InputEvent::UiCamera:
                _engine->graphicsPipeline().dragCamera(e.value[0], e.value[1], e.value[2]);
                break;
            case InputEvent::UiSphere:
                _engine->graphicsPipeline().dragSphere(e.index, e.value[0], e.value[1], e.value[2]);
                break;
            case InputEvent::Resize:
                _engine->setFramebufferSize(e.width, e.height);
                _engine->setFramebufferResized(true);
                break;
            case InputEvent::MouseDrag: {
                // src/application.cpp:85-112
                m_scene.yaw += e.dx * m_scene.mouseSensitivity;
                m_scene.pitch += e.dy * m_scene.mouseSensitivity;
                if (m_scene.pitch > 89.0f) m_scene.pitch = 89.0f;
                if (m_scene.pitch < -89.0f) m_scene.pitch = -89.0f;
                m_scene.applyOrientation();
                m_scene.m_camera.frameCount = 0;
                break;
            }
        }
    }
    // keys held this frame (src/application.cpp:44-67)
    UniformBufferObject& cam = m_scene.m_camera;
    if (m_keys['W']) m_scene.acceleration += cam.camera_forward * m_scene.movementSpeed;
    if (m_keys['S']) m_scene.acceleration -= cam.camera_forward * m_scene.movementSpeed;
    if (m_keys['A']) m_scene.acceleration -= m_scene.movementSpeed * cam.camera_right;
    if (m_keys['D']) m_scene.acceleration += m_scene.movementSpeed * cam.camera_right;
    if (m_keys['E']) m_scene.acceleration -= cam.camera_up * m_scene.movementSpeed;
    if (m_keys['Q']) m_scene.acceleration += cam.camera_up * m_scene.movementSpeed;
}

}  // namespace rt
