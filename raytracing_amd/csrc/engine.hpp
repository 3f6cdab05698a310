// engine.hpp -- per-frame orchestration with the interface of the reference's Engine and
// ComputePipeline classes (src/engine/engine.hpp:15-20, src/engine/includes/compute_pipeline.hpp:8-16),
// on top of the C ABI in include/rt_hip.h.  There is no window, swap chain or ImGui overlay:
// Engine is headless, and a non-null window pointer is rejected.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_hip.h"
#include "scene.hpp"

namespace rt {

// Non-zero status of a C-ABI call, thrown as the reference throws std::runtime_error
// (e.g. src/engine/src/compute_pipeline.cpp:88-104).
void check(int status);

class ComputePipeline {
   public:
    ComputePipeline(int device, uint32_t width, uint32_t height, Scene& scene);
    ~ComputePipeline();
    ComputePipeline(const ComputePipeline&) = delete;
    ComputePipeline& operator=(const ComputePipeline&) = delete;

    // updateScene + dispatch (compute_pipeline.cpp:657-663).  The two indices select the
    // reference's swap-chain image and in-flight frame; here they are accepted and ignored.
    void render(uint32_t imageIndex, uint32_t currentFrame);
    // New accumulation image and Scene::resetFrameCount() (compute_pipeline.cpp:52-65).
    void windowResized(uint32_t width, uint32_t height);

    rt_context* context() const { return m_ctx; }
    uint32_t width() const { return m_width; }
    uint32_t height() const { return m_height; }

   private:
    void updateScene();  // compute_pipeline.cpp:491-496; spheres are re-uploaded only when changed

    Scene& m_scene;
    rt_context* m_ctx = nullptr;
    uint32_t m_width, m_height;
    std::vector<Sphere> m_uploaded;  // what the device holds
    std::vector<Material> m_uploadedMaterials;  // empty: the device has no materials
    bool m_haveUpload = false;
};

// The part of the reference's GraphicsPipeline that reaches the kernel: its ImGui settings window
// (src/engine/src/graphics_pipeline.cpp:83-123) mutates the scene AFTER the compute pipeline has
// copied it for the current frame (src/engine/engine.cpp:67-68), so every change shows from the
// next frame on.  Here the widgets are operated by queued actions instead of a mouse; nothing is
// drawn.
class GraphicsPipeline {
   public:
    explicit GraphicsPipeline(Scene& scene) : m_scene(scene) {}
    // "Reset frame count" button (graphics_pipeline.cpp:92-94).
    void pressReset();
    // camera.x/y/z sliders (:95-105): range [-100, 100]; a NaN leaves that slider alone.  The frame
    // count is reset when a value actually changes.
    void dragCamera(float x, float y, float z);
    // sphere.x/y/z sliders of "Object number i" (:107-121): range [-10, 10]; NaN leaves a slider
    // alone.  The frame count is NOT reset: the image keeps accumulating over the moved sphere.
    void dragSphere(uint32_t index, float x, float y, float z);
    // Applies the queued actions in order (graphics_pipeline.cpp:76-134 without the drawing).
    void render(uint32_t imageIndex, uint32_t currentFrame);

   private:
    struct Action {
        int kind;  // 0 reset, 1 camera, 2 sphere
        uint32_t index;
        float v[3];
    };
    Scene& m_scene;
    std::vector<Action> m_pending;
};

class Engine {
   public:
    // `window` is the reference's GLFWwindow*; only nullptr (headless) is supported.
    Engine(uint32_t width, uint32_t height, void* window, Scene& scene);
    ~Engine();
    void render();  // src/engine/engine.cpp:47-133 without acquire/submit/present
    void setFramebufferResized(bool resized) { m_framebufferResized = resized; }

    // Headless stand-ins for what the window system provides.
    void setFramebufferSize(uint32_t width, uint32_t height);  // size used at the next resize
    void waitIdle();                                           // vkDeviceWaitIdle
    std::vector<uint8_t> readColor();                          // rgba8, width*height*4
    std::vector<float> readAccumulation();                     // rgba32f, width*height*4
    void writePPM(const std::string& path);                    // the colour image as binary PPM
    ComputePipeline& computePipeline() { return *m_computePipeline; }
    GraphicsPipeline& graphicsPipeline() { return *m_graphicsPipeline; }

   private:
    ComputePipeline* m_computePipeline = nullptr;
    GraphicsPipeline* m_graphicsPipeline = nullptr;
    bool m_framebufferResized = false;
    uint32_t m_pendingWidth, m_pendingHeight;
    uint32_t m_currentFrame = 0;
};

}  // namespace rt
