// rt_render: the command-line face of the library, standing in for the reference's `main` (main.rs:70-184), which
// loads a scene file, renders it and writes a PPM per chapter.
//
//   rt_render <scene.toml> [-o out.ppm] [--jpeg out.jpg] [--width W --height H] [--device D | --devices D0,D1,... [--band-rows R]] [--depth N] [--stats]
//   rt_render --all-chapters <scenes_dir> <outputs_dir> [--device D | --devices D0,D1,... [--band-rows R]]
//
// --devices renders one frame on several GPUs (rt_multi_render: bands of rows taken from a shared cursor; an ordinal
// may be listed more than once for several workers on one GPU); --band-rows sets the height of a band (a multiple of
// 8, the tile height; default: chosen by the library).
// The second form reproduces main()'s set of scene-file renders: chapter_5 .. chapter_14 into the reference's
// output names, a .ppm and a .jpg each as main() writes them (main.rs:77-169).  (The chapter 2 and 4 plots of main()
// draw on a canvas directly and never touch the renderer: not provided.)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/raytracer_rs_amd.h"

namespace {

int fail(const char *what) {
    std::fprintf(stderr, "rt_render: %s: %s\n", what, rt_last_error());
    return 1;
}

struct Options {
    uint32_t width = 0, height = 0;
    int device = 0, depth = -1;  // depth < 0: the reference's default budget
    std::vector<int> devices;    // non-empty: render on these GPUs together
    bool stats = false;
    uint32_t band_rows = 0;      // with --devices: rows per band, 0 = the library's choice
};

int render_file(const std::string &scene_path, const std::string &out_path, const std::string &jpeg_path, const Options &opt) {
    rt_scene_file *file = nullptr;
    if (rt_scene_file_load(scene_path.c_str(), &file) != RT_OK) return fail(scene_path.c_str());
    rt_camera_desc cam;
    rt_world_desc world;
    rt_scene *scene = nullptr;
    rt_multi *multi = nullptr;
    int rc = 1;
    std::vector<double> rgb;
    do {
        if (rt_scene_file_build_camera(file, &cam) != RT_OK) { fail("camera"); break; }
        if (opt.width && opt.height) {  // the scene's camera at another resolution (loader.rs:196-212 with new sizes)
            rt_camera_desc sized;
            if (rt_camera_new(opt.width, opt.height, cam.field_of_view, &sized) != RT_OK) { fail("camera"); break; }
            std::memcpy(sized.transform, cam.transform, sizeof cam.transform);
            cam = sized;
        }
        if (rt_scene_file_build_world(file, &world) != RT_OK) { fail("world"); break; }
        rgb.resize((size_t)cam.hsize * cam.vsize * 3);
        rt_render_opts ro;
        std::memset(&ro, 0, sizeof ro);
        if (opt.depth >= 0) {
            ro.max_depth = opt.depth;
            ro.flags = RT_RENDER_EXACT_DEPTH;
        }
        ro.collect_stats = opt.stats ? 1 : 0;
        rt_stats st;
        if (!opt.devices.empty()) {
            ro.band_rows = opt.band_rows;
            if (rt_multi_create(&world, opt.devices.data(), (uint32_t)opt.devices.size(), &multi) != RT_OK) { fail("scene"); break; }
            if (rt_multi_render(multi, &cam, &ro, rgb.data(), &st) != RT_OK) { fail("render"); break; }
        } else {
            if (rt_scene_create(&world, opt.device, &scene) != RT_OK) { fail("scene"); break; }
            if (rt_render(scene, &cam, &ro, rgb.data(), &st) != RT_OK) { fail("render"); break; }
        }
        if (rt_canvas_write_ppm(rgb.data(), cam.hsize, cam.vsize, out_path.c_str()) != RT_OK) { fail(out_path.c_str()); break; }
        if (!jpeg_path.empty() && rt_canvas_write_jpeg(rgb.data(), cam.hsize, cam.vsize, jpeg_path.c_str()) != RT_OK) { fail(jpeg_path.c_str()); break; }
        std::printf("%s -> %s  %ux%u  kernel %.3f ms", scene_path.c_str(), out_path.c_str(), cam.hsize, cam.vsize, st.kernel_ms);
        if (multi) std::printf("  gather %.3f ms", st.gather_ms);
        if (opt.stats)
            std::printf("  rays %llu (primary %llu, shadow %llu, reflected %llu, refracted %llu)", (unsigned long long)st.rays,
                        (unsigned long long)st.primary, (unsigned long long)st.shadow, (unsigned long long)st.reflect,
                        (unsigned long long)st.refract);
        std::printf("\n");
        rc = 0;
    } while (false);
    rt_scene_destroy(scene);
    rt_multi_destroy(multi);
    rt_scene_file_free(file);
    return rc;
}

}  // namespace

int main(int argc, char **argv) {
    Options opt;
    std::vector<std::string> pos;
    std::string out, jpeg;
    bool all = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "rt_render: %s needs a value\n", name);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-o" || a == "--out") out = value("-o");
        else if (a == "--jpeg") jpeg = value("--jpeg");
        else if (a == "--width") opt.width = (uint32_t)std::strtoul(value("--width"), nullptr, 10);
        else if (a == "--height") opt.height = (uint32_t)std::strtoul(value("--height"), nullptr, 10);
        else if (a == "--device") opt.device = std::atoi(value("--device"));
        else if (a == "--devices") {
            for (const char *p = value("--devices"); *p;) {
                char *end = nullptr;
                opt.devices.push_back((int)std::strtol(p, &end, 10));
                if (end == p) {
                    std::fprintf(stderr, "rt_render: --devices wants a comma-separated list of ordinals\n");
                    return 2;
                }
                p = *end == ',' ? end + 1 : end;
            }
        }
        else if (a == "--band-rows") opt.band_rows = (uint32_t)std::strtoul(value("--band-rows"), nullptr, 10);
        else if (a == "--depth") opt.depth = std::atoi(value("--depth"));
        else if (a == "--stats") opt.stats = true;
        else if (a == "--all-chapters") all = true;
        else if (a == "-h" || a == "--help") pos.clear(), argc = 0;
        else pos.push_back(a);
    }
    if ((all && pos.size() != 2) || (!all && pos.size() != 1) || (opt.width != 0) != (opt.height != 0) || (opt.band_rows && opt.devices.empty()) ||
        opt.band_rows % 8 != 0) {
        std::fprintf(stderr,
                     "usage: rt_render <scene.toml> [-o out.ppm] [--jpeg out.jpg] [--width W --height H] [--device D | --devices D0,D1,... [--band-rows R]] [--depth N] [--stats]\n"
                     "       rt_render --all-chapters <scenes_dir> <outputs_dir> [--device D | --devices D0,D1,... [--band-rows R]]\n"
                     "       (--band-rows: a multiple of 8, only with --devices)\n");
        return 2;
    }
    if (!all) {
        if (out.empty()) {
            out = pos[0];
            const size_t dot = out.rfind('.');
            if (dot != std::string::npos && out.find('/', dot) == std::string::npos) out.resize(dot);
            out += ".ppm";
        }
        return render_file(pos[0], out, jpeg, opt);
    }
    // main.rs:70-184: scene file -> output name
    static const char *const chapters[][2] = {
        {"chapter_5", "chapter_5_circle"},       {"chapter_6", "chapter_6_sphere"},     {"chapter_7_and_8", "chapter_7_and_8_world"},
        {"chapter_9", "chapter_9_plane_scene"},  {"chapter_10", "chapter_10_patterns"}, {"chapter_11", "chapter_11_reflections"},
        {"chapter_12", "chapter_12_cubes"},      {"chapter_13", "chapter_13_rocket"},   {"chapter_14", "chapter_14_groups"}};
    int bad = 0;
    for (const auto &c : chapters) bad += render_file(pos[0] + "/" + c[0] + ".toml", pos[1] + "/" + c[1] + ".ppm", pos[1] + "/" + c[1] + ".jpg", opt);
    return bad ? 1 : 0;
}
