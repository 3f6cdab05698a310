#!/bin/bash
# Build the scene model with host sanitizers and run the parser fuzz check (no GPU, ~20 s).
set -e
cd "$(dirname "$0")/../raytracing_amd/csrc"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -I. \
    ../../tools/fuzz_scene_parser.cpp scene_yaml.cpp scene.cpp scene_api.cpp -o /tmp/rt_fuzz_scene
/tmp/rt_fuzz_scene ../scenes/default40.yaml
