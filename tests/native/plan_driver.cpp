// Test driver (tests/test_sanitizers.py): feeds texts separated by 0x01 through the host-only half
// of the C ABI -- config parser, planner, launch/halo accessors -- in a build of rf_config.cpp,
// rf_plan.cpp and rf_abi.cpp with AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer.
#include <cstdio>
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
    return 0;
}
