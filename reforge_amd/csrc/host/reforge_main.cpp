// reforge_main.cpp -- the C++ host over librfhip.so: counterpart of the reference's
// src/main.rs (CLI + one headless frame) and the frame half of src/render.rs.  It only
// speaks the C ABI of include/rfhip.h, exactly as a Rust host would through FFI
// (INTEGRATION.md).
//
// Kept from the reference: the flags of `Args` (main.rs:43-71), the config/shader
// exclusivity check (main.rs:80-83), get_dim (utils.rs:56-74), the default graph
// (render.rs:115), single-shader mode (config.rs:77-90), headless = one frame then
// encode (main.rs:220-224), the status line (main.rs:157), and -- with --frames N --watch --
// the windowed loop's live reload of the config (render.rs:121-165,:497-519; main.rs:139-143).
// Out of scope: the winit/swapchain window (no display on an MI355X box) and the
// ffmpeg codecs -- images are read as PNG (host/png_io.h), binary PPM (P6) or raw RGBA8 and
// written as PNG (stored deflate), PPM or raw RGBA8, chosen by file extension.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include "png_io.h"
#include "rfhip.h"

namespace {

// utils.rs:13-18
void warnln(const std::string& msg) { std::fprintf(stderr, "\r\x1b[2K\x1b[33m%s\x1b[0m\n", msg.c_str()); }

struct Args {   // main.rs:43-71
    std::string shader_file_path, input_file, output_file, config, shader_path = "shaders";
    int width = -1, height = -1;
    std::string shader_format = "rgba32f";
    int num_frames = 2;
    bool decode_only = false;   // -i file -o file without touching the GPU: decoder check
    // additions for headless benchmarking
    long synthetic_seed = -1;
    int frames = 1;
    bool no_fusion = false, hipgraph = false;
    bool files_first = false;   // {shader_path}/{type}.comp | .stage.hip before the built-in registry (config.rs:59-75: the file is the type)
    bool watch = false;         // poll the config's mtime every frame and rebuild the graph when it changes
    int frame_interval_ms = 0;
};

void usage()
{
    std::fputs(
        "Usage: reforge [OPTIONS] [shader]\n\n"
        "Arguments:\n  [shader]  A single filter to execute instead of a config: a type name, or the path of a .comp (GLSL) / .stage.hip file\n\n"
        "Options:\n"
        "  -i, --input-file <INPUT_FILE>      File to read from (.png, .ppm P6 or .rgba raw)\n"
        "  -o, --output-file <OUTPUT_FILE>    File to write to (.png, .ppm or .rgba)\n"
        "      --width <WIDTH>\n"
        "      --height <HEIGHT>\n"
        "      --shader-format <rgba8|rgba32f>  Shader image format [default: rgba32f]\n"
        "      --config <config>              Path to the pipeline configuration file\n"
        "      --shader-path <shader-path>    Where a node type that is not built in is looked for: {type}.stage.hip, else {type}.comp (GLSL) [default: shaders]\n"
        "      --shader-files-first           A file in the shader path wins over a built-in type of the same name (the reference's rule)\n"
        "      --num-frames <NUM_FRAMES>      Frames in flight in the --watch loop (one headless frame forces 1) [default: 2]\n"
        "      --synthetic <SEED>             Generate the input on the GPU instead of reading a file\n"
        "      --frames <N>                   Execute the graph N times and report the mean frame time\n"
        "      --no-fusion                    One kernel launch per node, as the reference dispatches\n"
        "      --decode-only                  Decode -i and re-encode to -o without touching the GPU\n"
        "      --hipgraph                     Replay the frame as one hipGraph\n"
        "      --watch                        With --frames: live-reload the config when its mtime changes, status line per frame\n"
        "      --frame-interval-ms <MS>       With --frames: sleep between frames\n"
        "  -h, --help                         Print help\n",
        stderr);
}

bool parse_args(int argc, char** argv, Args& a)
{
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        auto val = [&](std::string& dst) {
            if (i + 1 >= argc) { warnln("error: a value is required for '" + s + "'"); return false; }
            dst = argv[++i];
            return true;
        };
        std::string v;
        if (s == "-h" || s == "--help") { usage(); std::exit(0); }
        else if (s == "-i" || s == "--input-file") { if (!val(a.input_file)) return false; }
        else if (s == "-o" || s == "--output-file") { if (!val(a.output_file)) return false; }
        else if (s == "--width") { if (!val(v)) return false; a.width = std::atoi(v.c_str()); }
        else if (s == "--height") { if (!val(v)) return false; a.height = std::atoi(v.c_str()); }
        else if (s == "--shader-format") { if (!val(a.shader_format)) return false; }
        else if (s == "--config") { if (!val(a.config)) return false; }
        else if (s == "--shader-path") { if (!val(a.shader_path)) return false; }
        else if (s == "--num-frames") { if (!val(v)) return false; a.num_frames = std::atoi(v.c_str()); }
        else if (s == "--synthetic") { if (!val(v)) return false; a.synthetic_seed = std::strtol(v.c_str(), nullptr, 0); }
        else if (s == "--frames") { if (!val(v)) return false; a.frames = std::atoi(v.c_str()); }
        else if (s == "--decode-only") a.decode_only = true;
        else if (s == "--no-fusion") a.no_fusion = true;
        else if (s == "--shader-files-first") a.files_first = true;
        else if (s == "--hipgraph") a.hipgraph = true;
        else if (s == "--watch") a.watch = true;
        else if (s == "--frame-interval-ms") { if (!val(v)) return false; a.frame_interval_ms = std::atoi(v.c_str()); }
        else if (!s.empty() && s[0] == '-') { warnln("error: unexpected argument '" + s + "'"); return false; }
        else if (a.shader_file_path.empty()) a.shader_file_path = s;
        else { warnln("error: unexpected argument '" + s + "'"); return false; }
    }
    return true;
}

// utils::get_dim, utils.rs:56-74
void get_dim(int width, int height, int new_w, int new_h, int& w, int& h)
{
    w = width;
    h = height;
    if (new_w >= 0 && new_h >= 0) { w = new_w; h = new_h; return; }
    if (new_w >= 0) { w = new_w; h = (int)(((float)w / (float)width) * (float)height); }
    else if (new_h >= 0) { h = new_h; w = (int)(((float)h / (float)height) * (float)width); }
}

bool ends_with(const std::string& s, const char* suf)
{
    size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// ---- minimal image I/O (imagefileio.rs is ffmpeg; codecs are out of scope) ---------
bool read_ppm(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgba)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P6") return false;
    auto next_int = [&](int& v) {
        for (;;) {
            int c = f.peek();
            if (c == '#') { std::string line; std::getline(f, line); }
            else if (std::isspace(c)) f.get();
            else break;
        }
        f >> v;
        return (bool)f;
    };
    int maxv = 0;
    if (!next_int(w) || !next_int(h) || !next_int(maxv) || maxv != 255 || w < 1 || h < 1) return false;
    f.get();
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    f.read((char*)rgb.data(), (std::streamsize)rgb.size());
    if (!f) return false;
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        rgba[i * 4 + 0] = rgb[i * 3 + 0];
        rgba[i * 4 + 1] = rgb[i * 3 + 1];
        rgba[i * 4 + 2] = rgb[i * 3 + 2];
        rgba[i * 4 + 3] = 255;
    }
    return true;
}

bool write_image(const std::string& path, int w, int h, const uint8_t* rgba)
{
    if (ends_with(path, ".ppm")) {
        std::ofstream f(path, std::ios::binary);
        f << "P6\n" << w << " " << h << "\n255\n";
        for (size_t i = 0; i < (size_t)w * h; ++i) f.write((const char*)rgba + i * 4, 3);
        return (bool)f;
    }
    if (ends_with(path, ".rgba")) {
        std::ofstream f(path, std::ios::binary);
        f.write((const char*)rgba, (std::streamsize)((size_t)w * h * 4));
        return (bool)f;
    }
    return pngio::write_png(path, w, h, rgba);   // the reference always writes PNG (imagefileio.rs:221)
}

// utils::get_modified_time, utils.rs:33-42: whole seconds, 0 when the file cannot be reached
uint64_t modified_time(const std::string& path)
{
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return 0;
    return (uint64_t)st.st_mtime;
}

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

#define RF_CHECK(expr)                                                                          \
    do {                                                                                        \
        rf_status st_ = (expr);                                                                 \
        if (st_ != RF_OK) {                                                                     \
            warnln(std::string(#expr) + " -> " + std::to_string((int)st_) + ": " + rf_last_error()); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

}  // namespace

static int run(int argc, char** argv);

// Nothing may escape main as an exception: a malformed input file or an allocation failure ends the
// run with a warning and exit code 1, like the reference's Option::None + warnln! paths.
int main(int argc, char** argv)
{
    try {
        return run(argc, argv);
    } catch (const std::exception& e) {
        warnln(std::string("fatal: ") + e.what());
    } catch (...) {
        warnln("fatal: unknown exception");
    }
    return 1;
}

static int run(int argc, char** argv)
{
    Args args;
    if (!parse_args(argc, argv, args)) { usage(); return 2; }
    if (args.output_file.empty()) {
        warnln("No output file given: window/swapchain presentation is out of scope on an MI355X box; pass -o <file>");
        return 1;
    }
    // main.rs:77-78: headless uses one frame in flight; the --watch loop stands in for the windowed
    // loop and cycles --num-frames slots like it (main.rs:69-70, render.rs:328-337)
    const int num_frames = args.watch ? (args.num_frames < 1 ? 1 : args.num_frames) : 1;
    if (!args.config.empty() && !args.shader_file_path.empty()) {   // main.rs:80-83
        warnln("Cannot specify both a config and shader file");
        return 1;
    }
    rf_format format;
    if (args.shader_format == "rgba8") format = RF_FORMAT_RGBA8;
    else if (args.shader_format == "rgba32f") format = RF_FORMAT_RGBA32F;
    else { warnln("error: invalid value '" + args.shader_format + "' for '--shader-format' [possible values: rgba8, rgba32f]"); return 2; }

    // decode (main.rs:85-100,:126-132)
    double t0 = now_ms();
    int in_w = 800, in_h = 600;   // main.rs:99
    std::vector<uint8_t> staging;
    const bool has_file = !args.input_file.empty();
    const bool has_input = has_file || args.synthetic_seed >= 0;
    if (has_file) {
        if (ends_with(args.input_file, ".rgba")) {
            if (args.width < 1 || args.height < 1) { warnln("a raw .rgba input needs --width and --height"); return 1; }
            in_w = args.width; in_h = args.height;
            std::ifstream f(args.input_file, std::ios::binary);
            staging.resize((size_t)in_w * in_h * 4);
            f.read((char*)staging.data(), (std::streamsize)staging.size());
            if (!f) { warnln("Error reading file '" + args.input_file + "'"); return 1; }
        } else if (ends_with(args.input_file, ".png")) {
            std::string perr;
            if (!pngio::read_png(args.input_file, in_w, in_h, staging, perr)) {
                warnln("Error reading file '" + args.input_file + "': " + perr);
                return 1;
            }
        } else if (!read_ppm(args.input_file, in_w, in_h, staging)) {
            warnln("Error reading file '" + args.input_file + "': only PNG (8-bit, non-interlaced), binary PPM (P6) and raw .rgba are decoded");
            return 1;
        }
    }
    int width, height;
    get_dim(in_w, in_h, args.width, args.height, width, height);
    if (has_file && (width != in_w || height != in_h)) {
        warnln("resizing on load (swscale in the reference, imagefileio.rs:150-176) is out of scope: drop --width/--height");
        return 1;
    }
    if (has_file) std::printf("File Decode and resize: %.2fms\n", now_ms() - t0);
    if (args.decode_only) {   // no GPU involved: decode -i, encode -o
        if (!has_file) { warnln("--decode-only needs -i"); return 1; }
        return write_image(args.output_file, width, height, staging.data()) ? 0 : 1;
    }

    // create_config, render.rs:100-119 (nullptr + warning on a user error)
    (void)rf_set_shader_path(args.shader_path.c_str());       // Config::new(path, shader_path), config.rs:59-75
    (void)rf_set_type_lookup(args.files_first ? 1 : 0);
    {
        // `reforge [shader]` names a FILE (single_shader_parse, config.rs:77-90: the pipeline is that file, whatever the shader path):
        // an existing .comp / .stage.hip given by path is looked up where it lies, and it wins over a built-in type of its name
        struct stat st;
        const std::string& sp = args.shader_file_path;
        const bool is_file = sp.size() > 5 && (sp.compare(sp.size() - 5, 5, ".comp") == 0 || (sp.size() > 10 && sp.compare(sp.size() - 10, 10, ".stage.hip") == 0));
        if (is_file && ::stat(sp.c_str(), &st) == 0) {
            const size_t slash = sp.find_last_of('/');
            args.shader_path = slash == std::string::npos ? "." : sp.substr(0, slash);
            (void)rf_set_shader_path(args.shader_path.c_str());
            (void)rf_set_type_lookup(1);
        }
    }
    auto create_config = [&]() -> rf_config* {
        rf_config* c = nullptr;
        if (!args.config.empty()) {
            std::ifstream f(args.config);
            std::stringstream ss;
            ss << f.rdbuf();
            if (!f) { warnln("Error reading file '" + args.config + "'"); return nullptr; }
            if (ss.str().empty()) { warnln("File was empty: " + args.config); return nullptr; }
            if (rf_config_parse(ss.str().c_str(), has_input, &c) != RF_OK) { warnln(rf_last_error()); return nullptr; }
        } else if (!args.shader_file_path.empty()) {
            std::string stem = args.shader_file_path;   // config.rs:79: file stem
            size_t slash = stem.find_last_of('/');
            if (slash != std::string::npos) stem = stem.substr(slash + 1);
            size_t dot = stem.find('.');      // (the type of `x.stage.hip` is x)
            if (dot != std::string::npos) stem = stem.substr(0, dot);
            if (rf_config_single(stem.c_str(), has_input, &c) != RF_OK) { warnln(rf_last_error()); return nullptr; }
        } else if (rf_config_parse("input -> passthrough -> output", has_input, &c) != RF_OK) {
            warnln(rf_last_error());
            return nullptr;
        }
        return c;
    };
    rf_config* cfg = create_config();
    if (!cfg) { warnln("Unable to create config"); return 1; }   // render.rs:543

    rf_ctx* ctx = nullptr;
    RF_CHECK(rf_ctx_create(0, &ctx));
    rf_graph_options opt{};
    opt.width = width;
    opt.height = height;
    opt.format = format;
    opt.num_frames = num_frames;
    opt.flags = RF_GRAPH_TIMERS | (args.no_fusion ? RF_GRAPH_NO_FUSION : 0u);
    if (args.hipgraph) opt.flags = (opt.flags & ~RF_GRAPH_TIMERS) | RF_GRAPH_HIPGRAPH;
    rf_graph* graph = nullptr;
    RF_CHECK(rf_graph_create(ctx, cfg, &opt, &graph));

    // record_initial_image_load (render.rs:264-313), once per graph
    auto load_input = [&]() -> rf_status {
        if (has_file) return rf_graph_upload_srgb8(graph, staging.data(), (size_t)width * 4);
        if (args.synthetic_seed >= 0) return rf_graph_fill_synthetic(graph, (uint32_t)args.synthetic_seed);
        return RF_OK;
    };
    // config_changed + recreate_graph (render.rs:121-165): mtime in whole seconds as
    // utils.rs:33-42; a config that does not parse or plan keeps the old graph running
    uint64_t last_mtime = args.config.empty() ? 0 : modified_time(args.config);
    // the stage files of user types the running graph names (reload_changed_pipelines, render.rs:225-249): type -> mtime (ns)
    std::map<std::string, long long> stage_mtimes;
    auto stage_file_ns = [&](const std::string& type) -> long long {
        struct stat st;
        if (::stat((args.shader_path + "/" + type + ".stage.hip").c_str(), &st) != 0 && ::stat((args.shader_path + "/" + type + ".comp").c_str(), &st) != 0) return -1;
        return (long long)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
    };
    auto note_stage_files = [&]() {
        stage_mtimes.clear();
        for (int n = 0; n < rf_config_num_nodes(cfg); ++n) {
            const char* t = rf_config_node_type(cfg, n);
            if (t && rf_user_stage_mtime(t) >= 0) stage_mtimes[t] = stage_file_ns(t);
        }
    };
    note_stage_files();
    auto trigger_reloads = [&]() -> bool {
        bool changed = false;
        if (!args.config.empty()) {
            const uint64_t m = modified_time(args.config);
            if (m == 0) {
                if (last_mtime != 0) warnln("Unable to access config file: " + args.config);
                last_mtime = 0;
            } else if (m != last_mtime) {
                last_mtime = m;
                changed = true;
            }
        }
        for (auto& kv : stage_mtimes) {
            const long long now = stage_file_ns(kv.first);
            if (now != kv.second) { kv.second = now; changed = true; }   // a file that fails to build is not retried until it changes again
        }
        if (!changed) return false;
        rf_config* c2 = create_config();
        if (!c2) return false;
        for (int k = 0; k < num_frames; ++k) (void)rf_graph_wait(graph, k);   // device_wait_idle, render.rs:126
        rf_graph* g2 = nullptr;
        if (rf_graph_create(ctx, c2, &opt, &g2) != RF_OK) {
            warnln(rf_last_error());
            rf_config_destroy(c2);
            return false;
        }
        rf_graph_destroy(graph);
        rf_config_destroy(cfg);
        graph = g2;
        cfg = c2;
        note_stage_files();
        return true;
    };

    // render_fn, main.rs:134-182, headless: --frames iterations of the windowed loop's body
    RF_CHECK(load_input());
    RF_CHECK(rf_graph_wait(graph, 0));
    const int frames = args.frames < 1 ? 1 : args.frames;
    const double t_start = now_ms();
    double timer = t_start, avg_ms = 0.0, sum_ms = 0.0;
    char times[4096] = "";
    int slot = 0;
    for (int i = 0; i < frames; ++i) {
        slot = i % num_frames;
        RF_CHECK(rf_graph_wait(graph, slot));   // wait_for_frame_fence: the previous use of this slot (render.rs:328-337)
        if (args.frame_interval_ms > 0) {   // pace the loop like a display would, so the config can be edited while it runs
            std::this_thread::sleep_for(std::chrono::milliseconds(args.frame_interval_ms));
        }
        if (args.watch && trigger_reloads()) {
            std::fputs("\r\x1b[2K", stderr);   // main.rs:141
            RF_CHECK(load_input());
            slot = 0;                           // recreate_graph resets frame_index (render.rs:134)
        }
        if (args.watch || frames == 1) {
            rf_status ts = rf_graph_set_time(graph, (float)((now_ms() - t_start) / 1e3));   // update_ubos, render.rs:212-223
            if (ts != RF_OK) { warnln(rf_last_error()); return 1; }
        }
        if (args.watch) {   // the status line of every frame, main.rs:150-157
            const double elapsed = now_ms() - timer;
            timer = now_ms();
            avg_ms = avg_ms - avg_ms / 60.0 + elapsed / 60.0;   // utils::moving_avg, utils.rs:76-82
            if ((opt.flags & RF_GRAPH_TIMERS) && i >= num_frames) RF_CHECK(rf_graph_times_string(graph, slot, times, sizeof(times)));   // this slot's previous frame
            std::fprintf(stderr, "\rFrame: %5.2fms, Frame-Avg: %5.2fms, GPU: {%s}", elapsed, avg_ms, times);
        }
        RF_CHECK(rf_graph_execute(graph, slot));
    }
    for (int k = 0; k < num_frames; ++k) RF_CHECK(rf_graph_wait(graph, k));
    sum_ms = now_ms() - t_start;
    const double frame_ms = sum_ms / (double)frames;
    if (opt.flags & RF_GRAPH_TIMERS) RF_CHECK(rf_graph_times_string(graph, slot, times, sizeof(times)));
    std::fprintf(stderr, "\rFrame: %5.2fms, Frame-Avg: %5.2fms, GPU: {%s}\n", frame_ms, args.watch ? avg_ms : frame_ms, times);   // main.rs:157

    staging.resize((size_t)width * height * 4);
    RF_CHECK(rf_graph_download_srgb8(graph, slot, staging.data(), (size_t)width * 4));
    if (!write_image(args.output_file, width, height, staging.data())) { warnln("Encoding error: cannot write " + args.output_file); return 1; }

    rf_graph_destroy(graph);
    rf_ctx_destroy(ctx);
    rf_config_destroy(cfg);
    return 0;
}
