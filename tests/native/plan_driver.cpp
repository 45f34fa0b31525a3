// Test driver (tests/test_sanitizers.py): feeds texts separated by 0x01 through the host-only half
// of the C ABI -- config parser, planner, launch/halo accessors -- in a build of rf_config.cpp,
// rf_plan.cpp, rf_user.cpp and rf_abi.cpp with AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "rfhip.h"
int main(int argc, char** argv) {
    std::ifstream f(argv[1], std::ios::binary);
    std::stringstream ss; ss << f.rdbuf();
    std::string all = ss.str();
    size_t pos = 0; int n = 0, okc = 0;
    while (pos < all.size()) {
        size_t e = all.find(std::string("\x01", 1), pos);
        if (e == std::string::npos) e = all.size();
        std::string text = all.substr(pos, e - pos);
        pos = e + 1; ++n;
        rf_config* c = nullptr;
        if (rf_config_parse(text.c_str(), 1, &c) != RF_OK) continue;
        for (unsigned flags : {0u, 2u}) {
            rf_plan* p = nullptr;
            if (rf_plan_create(c, flags, &p) == RF_OK) {
                ++okc;
                int nl = rf_plan_num_launches(p);
                for (int i = 0; i < nl; ++i) { (void)rf_plan_launch_label(p, i); (void)rf_plan_launch_serial(p, i); for (int k = 0; k < rf_plan_launch_num_inputs(p, i); ++k) (void)rf_plan_launch_input(p, i, k); }
                int ns[64], nd[64], ni, gh;
                if (nl <= 64) (void)rf_plan_halo_schedule(p, 1, ns, nd, 64, &ni, &gh);
                rf_plan_destroy(p);
            }
        }
        rf_config_destroy(c);
    }
    std::printf("texts %d plans %d\n", n, okc);
    // argv[2] / argv[3]: a directory of generated stage files f0000.stage.hip ... and how many -- the "reflection" of user types
    // (rf_user.cpp: parse_user_stage) on untrusted text, reached the way a config reaches it
    if (argc > 3) {
        rf_set_shader_path(argv[2]);
        int parsed = 0;
        const int count = std::atoi(argv[3]);
        for (int k = 0; k < count; ++k) {
            char name[32];
            std::snprintf(name, sizeof(name), "f%04d", k);
            const std::string text = std::string("input -> nn -> output\nnn: ") + name + " { amount: 1.0 }";
            rf_config* c = nullptr;
            if (rf_config_parse(text.c_str(), 1, &c) != RF_OK) continue;
            rf_plan* p = nullptr;
            if (rf_plan_create(c, 0, &p) == RF_OK) {
                ++parsed;
                int ns[8], nd[8], ni, gh;
                (void)rf_plan_halo_schedule(p, 0, ns, nd, 8, &ni, &gh);
                (void)rf_registry_binding(name, "input_image");
                (void)rf_registry_buffer_binding(name, "ToneCurve");
                rf_plan_destroy(p);
            }
            rf_config_destroy(c);
        }
        std::printf("stages %d parsed %d\n", count, parsed);
    }
    // argv[4]: how many generated GLSL files g0000.comp ... the same directory holds -- the translator and reflection of rf_glsl.cpp on
    // untrusted text, directly (rf_glsl_translate / rf_glsl_reflect) and the way a config reaches them
    if (argc > 4) {
        int translated = 0, planned = 0;
        const int count = std::atoi(argv[4]);
        std::vector<char> buf(1 << 20);
        for (int k = 0; k < count; ++k) {
            char name[32];
            std::snprintf(name, sizeof(name), "g%04d", k);
            std::ifstream gf(std::string(argv[2]) + "/" + name + ".comp", std::ios::binary);
            std::stringstream gs; gs << gf.rdbuf();
            const std::string text = gs.str();
            size_t len = 0;
            if (rf_glsl_translate(name, text.c_str(), buf.data(), buf.size(), &len) == RF_OK) ++translated;
            (void)rf_glsl_reflect(name, text.c_str(), buf.data(), buf.size(), &len);
            (void)rf_glsl_translate(name, text.c_str(), buf.data(), 16, &len);      // a buffer that is too small
            const std::string cfg = std::string("input -> nn -> output\nnn: ") + name + " { amount: 1.0 }";
            rf_config* c = nullptr;
            if (rf_config_parse(cfg.c_str(), 1, &c) != RF_OK) continue;
            rf_plan* p = nullptr;
            if (rf_plan_create(c, 0, &p) == RF_OK) {
                int ns[8], nd[8], ni, gh;
                if (rf_plan_halo_schedule(p, 0, ns, nd, 8, &ni, &gh) == RF_OK) ++planned;
                rf_plan_destroy(p);
            }
            rf_config_destroy(c);
        }
        std::printf("glsl %d translated %d planned %d\n", count, translated, planned);
    }
    return 0;
}
