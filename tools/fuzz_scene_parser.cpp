// fuzz_scene_parser.cpp -- host-only robustness check of the scene model (development aid):
// parses a good scene file and 3000 random mutations of it (byte flips, deletions, insertions,
// truncations, duplications), saves and reloads what parses, under AddressSanitizer and
// UndefinedBehaviorSanitizer.  Built and run by tools/fuzz_scene.sh; no GPU involved.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include "rt_hip.h"
static std::string g_err;
void rt_set_error(const char* m) { g_err = m ? m : ""; }
int main(int argc, char** argv)
{
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    const std::string good = ss.str();
    rt_scene* s = nullptr;
    int rc = rt_scene_parse(good.data(), good.size(), &s);
    printf("good: %d spheres %d\n", rc, s ? rt_scene_sphere_count(s) : -1);
    if (s) { rt_scene_save(s, "/tmp/rt_fuzz_out.yaml"); rt_scene_update(s, 0.016f); rt_scene_materials(s); rt_scene_free(s); }
    std::mt19937 rng(1);
    int ok = 0, bad = 0;
    for (int it = 0; it < 3000; it++) {
        std::string t = (it & 1) ? good : good.substr(0, 4000 + rng() % 2000);
        const int nmut = (it & 1) ? 1 + rng() % 2 : 1 + rng() % 6;
        for (int k = 0; k < nmut; k++) {
            const size_t pos = rng() % t.size();
            switch (rng() % 5) {
                case 0: t[pos] = (char)(rng() % 256); break;
                case 1: t.erase(pos, 1 + rng() % 8); break;
                case 2: t.insert(pos, std::string(1 + rng() % 4, "-[]:, \n#e.9"[rng() % 11])); break;
                case 3: t.resize(pos); break;
                default: t.insert(pos, t.substr(rng() % t.size(), rng() % 40)); break;
            }
            if (t.empty()) t = "x";
        }
        rt_scene* q = nullptr;
        rc = rt_scene_parse(t.data(), t.size(), &q);
        if (rc == 0 && q) { ok++; rt_scene_spheres(q); rt_scene_materials(q); rt_scene_save(q, "/tmp/rt_fuzz_out2.yaml"); rt_scene_free(q); } else bad++;
    }
    for (int kind = 0; kind < 3; kind++) { rt_scene* g = nullptr; rt_scene_generate(kind, 500, 3, &g); rt_scene_save(g, "/tmp/rt_fuzz_g.yaml"); rt_scene* h = nullptr; rt_scene_load("/tmp/rt_fuzz_g.yaml", &h); rt_scene_free(g); rt_scene_free(h); }
    printf("mutants parsed ok %d, rejected %d\n", ok, bad);
    return 0;
}
