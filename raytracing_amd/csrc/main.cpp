// main.cpp -- headless driver with the shape of the reference's main.cpp:4-14: construct an
// Application at 1280x720, Run(), print the message of any exception and exit 1.
//
// usage: rt_headless [--size WxH] [--frames N] [--scene file.yaml] [--script events.txt]
//                    [--position X,Y,Z] [--yaw DEG] [--pitch DEG] [--out image.ppm] [--save-on-exit]
// --position/--yaw/--pitch place the camera where keys and mouse would have taken it (the state
// src/application.cpp:38-118 changes) before the first frame.
// The scene defaults to $RT_SCENE, then ../res/scenes/scene.yaml (the reference's path).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "application.hpp"

int main(int argc, char** argv)
{
    try {
        uint32_t width = 1280, height = 720, frames = 64;
        std::string script, out;
        bool save = false, havePosition = false, haveAngles = false;
        float position[3] = {0.f, 0.f, 0.f}, yaw = 90.0f, pitch = 0.0f;
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto value = [&]() -> const char* {
                if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
                return argv[++i];
            };
            if (a == "--size") {
                if (std::sscanf(value(), "%ux%u", &width, &height) != 2 || !width || !height)
                    throw std::runtime_error("--size expects WxH");
            } else if (a == "--frames") {
                frames = (uint32_t)std::strtoul(value(), nullptr, 10);
            } else if (a == "--scene") {
                setenv("RT_SCENE", value(), 1);
            } else if (a == "--script") {
                script = value();
            } else if (a == "--position") {
                if (std::sscanf(value(), "%f,%f,%f", &position[0], &position[1], &position[2]) != 3)
                    throw std::runtime_error("--position expects X,Y,Z");
                havePosition = true;
            } else if (a == "--yaw") {
                yaw = std::strtof(value(), nullptr);
                haveAngles = true;
            } else if (a == "--pitch") {
                pitch = std::strtof(value(), nullptr);
                haveAngles = true;
            } else if (a == "--out") {
                out = value();
            } else if (a == "--save-on-exit") {
                save = true;
            } else {
                throw std::runtime_error("unknown option " + a);
            }
        }
        rt::Application app(width, height, "Raytracing");
        app.maxFrames = frames;
        app.saveOnExit = save;
        if (!script.empty()) app.loadScript(script);
        if (haveAngles) {
            app.scene().yaw = yaw;
            app.scene().pitch = pitch;
            app.scene().applyOrientation();
        }
        if (havePosition) app.scene().m_camera.camera_position = rt::vec3(position[0], position[1], position[2]);
        app.Run();
        if (!out.empty()) app._engine->writePPM(out);
        std::printf("rendered %u frame(s), accumulated frame count %u\n", app.framesRun(),
                    app.scene().camera().frameCount);
    } catch (const std::exception& e) {
        std::printf("%s\n", e.what());
        return 1;
    }
    return 0;
}
