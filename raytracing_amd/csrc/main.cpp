// main.cpp -- headless driver with the shape of the reference's main.cpp:4-14: construct an
// Application at 1280x720, Run(), print the message of any exception and exit 1.
//
// usage: rt_headless [--size WxH] [--frames N] [--scene file.yaml] [--script events.txt]
//                    [--out image.ppm] [--save-on-exit]
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
        bool save = false;
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
