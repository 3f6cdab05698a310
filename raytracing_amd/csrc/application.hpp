// application.hpp -- the reference's Application (includes/application.hpp:12-43) without a
// window: same constructor, destructor and Run(), same per-frame order
// (handleInput -> showFPS -> Scene::update -> Engine::render, src/application.cpp:28-36), with
// the GLFW key/mouse polling replaced by a script of input events and the wall clock by a
// fixed time step, so a run is reproducible.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "engine.hpp"
#include "scene.hpp"

namespace rt {

// One scripted input change, applied at the start of frame `frame` (0-based).
struct InputEvent {
    enum Kind { KeyDown, KeyUp, MouseDrag, Reload, Resize, Quit, UiReset, UiCamera, UiSphere };
    uint32_t frame = 0;
    Kind kind = Quit;
    char key = 0;        // KeyDown/KeyUp: one of W A S D Q E (src/application.cpp:44-67)
    float dx = 0, dy = 0;  // MouseDrag: cursor delta in pixels (src/application.cpp:85-88)
    uint32_t width = 0, height = 0;  // Resize
    // UiCamera / UiSphere: the slider values of the settings window (NaN = slider not touched);
    // UiSphere: index = the "Object number" (src/engine/src/graphics_pipeline.cpp:92-121)
    uint32_t index = 0;
    float value[3] = {0, 0, 0};
};

class Application {
   public:
    Application(uint32_t width, uint32_t height, const char* name);
    ~Application();  // saves the scene when saveOnExit is set (the reference always saves)
    void Run();      // until a Quit event or maxFrames frames

    // Script: one event per line, '#' comments --
    //   <frame> key <W|A|S|D|Q|E> <down|up>
    //   <frame> mouse <dx> <dy>
    //   <frame> reload | quit
    //   <frame> resize <width> <height>
    //   <frame> ui reset                          the settings window's "Reset frame count" button
    //   <frame> ui camera <x> <y> <z>             its camera sliders; '=' leaves a slider alone
    //   <frame> ui sphere <i> <x> <y> <z>         the centre sliders of "Object number i"
    // (the ui events act after the frame's dispatch, as the overlay does, and show one frame later)
    void loadScript(const std::string& path);
    void setScript(std::vector<InputEvent> events);

    Scene& scene() { return m_scene; }
    uint32_t framesRun() const { return m_frame; }

    std::unique_ptr<Engine> _engine;
    uint32_t maxFrames = 64;
    float fixedDt = 1.0f / 60.0f;  // the reference measures dt with glfwGetTime (src/application.cpp:120-135)
    bool saveOnExit = false;

   private:
    void handleInput();

    Scene m_scene;
    std::vector<InputEvent> m_script;
    size_t m_next = 0;
    uint32_t m_frame = 0;
    bool m_keys[256] = {false};
    bool m_quit = false;
    float dt = 0.0f;
};

}  // namespace rt
