// rt_cli.cpp — headless batch renderer on top of the C ABI, the counterpart of the
// reference's command-line mode (src/main.cpp:478-535: load scene, build BVH, render every
// configured camera, write one BMP each, print the timings). Options replace the TOML file:
//
//   rt_cli --data DIR --scene cornell|monkey|teapot|cube|cube_textured|single_triangle|spheres|custom
//          | --obj FILE [--normalize]
//          [--size W H] [--depth N] [--features shading,recursive,hard_shadow,soft_shadow,
//            normal_interp,texture,bvh,sah,bilinear] [--camera FOV DIST RX RY]... [--out PREFIX]
//
//   rt_cli CONFIG.toml            the reference's own configuration file (src/config.cpp:193-374:
//          window_size, data_path, scene, output_dir, [features], [features.extra], [[cameras]],
//          [[lights]]); images are named {scene}_{start time}_cam_{i}.bmp in output_dir like
//          src/main.cpp:518-521. Keys the reference does not read but its Features struct has
//          (enable_soft_shadow, extra.enable_bvh_sah_binning, extra.enable_transparency) are accepted too.
//
// Several --camera options render several images (the reference spawns one thread per
// camera; here the frames go to the GPU one after the other).
#include <chrono>
#include <ctime>
#include <fstream>
#include <sstream>
#include <sys/stat.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_whitted.h"
#include "mini_toml.hpp"

extern "C" {
typedef struct rt_loaded_scene rt_loaded_scene;
}

static int die(const char* what, const char* msg)
{
    std::fprintf(stderr, "rt_cli: %s: %s\n", what, msg);
    return 1;
}

int main(int argc, char** argv)
{
    std::string data = ".", scene = "", obj = "", out = "render";
    bool normalize = false;
    int W = 800, H = 800, depth = 5; // config.h: windowSize 800x800; render.cpp:318: depth 5
    rt_features f;
    memset(&f, 0, sizeof(f));
    f.enableShading = f.enableAccelStructure = 1;
    struct Cam {
        float fov, dist, rx, ry, rz = 0.0f;
        float look[3] = { 0.0f, 0.0f, 0.0f };
    };
    std::vector<Cam> cams;
    std::vector<rt_light_desc> cfg_lights; // [[lights]] of a config file (scene files only, like loadSceneFromFile)
    std::string config_path, scene_label;
    bool have_config = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int n) { return i + n < argc; };
        if (a == "--config" && need(1)) config_path = argv[++i];
        else if (i == 1 && a.size() > 5 && a.substr(a.size() - 5) == ".toml") config_path = a; // src/main.cpp:48-49
        else if (a == "--data" && need(1)) data = argv[++i];
        else if (a == "--scene" && need(1)) scene = argv[++i];
        else if (a == "--obj" && need(1)) obj = argv[++i];
        else if (a == "--normalize") normalize = true;
        else if (a == "--out" && need(1)) out = argv[++i];
        else if (a == "--depth" && need(1)) depth = std::atoi(argv[++i]);
        else if (a == "--size" && need(2)) W = std::atoi(argv[++i]), H = std::atoi(argv[++i]);
        else if (a == "--camera" && need(4)) {
            Cam c { (float)std::atof(argv[i + 1]), (float)std::atof(argv[i + 2]), (float)std::atof(argv[i + 3]), (float)std::atof(argv[i + 4]) };
            cams.push_back(c);
            i += 4;
        } else if (a == "--features" && need(1)) {
            memset(&f, 0, sizeof(f));
            std::string s = argv[++i];
            s += ',';
            for (size_t p = 0, q; (q = s.find(',', p)) != std::string::npos; p = q + 1) {
                const std::string k = s.substr(p, q - p);
                if (k == "shading") f.enableShading = 1;
                else if (k == "recursive") f.enableRecursive = 1;
                else if (k == "hard_shadow") f.enableHardShadow = 1;
                else if (k == "soft_shadow") f.enableSoftShadow = 1;
                else if (k == "normal_interp") f.enableNormalInterp = 1;
                else if (k == "texture") f.enableTextureMapping = 1;
                else if (k == "bvh") f.enableAccelStructure = 1;
                else if (k == "sah") f.enableBvhSahBinning = 1;
                else if (k == "bilinear") f.enableBilinearTextureFiltering = 1;
                else if (!k.empty()) return die("unknown feature", k.c_str());
            }
        } else {
            return die("unknown or incomplete option", a.c_str());
        }
    }
    if (!config_path.empty()) { // ≙ readConfigFile, src/config.cpp:193-374
        std::ifstream in(config_path);
        if (!in)
            return die("cannot open the config file", config_path.c_str());
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string text = ss.str();
        mini_toml::Value cfg;
        const std::string perr = mini_toml::Parser(text).parse(cfg);
        if (!perr.empty())
            return die("config file", perr.c_str());
        have_config = true;
        auto vec = [](const mini_toml::Value* v, size_t n, float* out) {
            if (!v || v->kind != mini_toml::Value::ARRAY || v->a.size() < n)
                return false;
            for (size_t k = 0; k < n; k++) {
                if (!v->a[k].is_number())
                    return false;
                out[k] = (float)v->a[k].number();
            }
            return true;
        };
        float ws[2];
        if (vec(cfg.get("window_size"), 2, ws))
            W = (int)ws[0], H = (int)ws[1];
        const std::string dp = cfg.string_or("data_path", "default");
        if (dp != "default")
            data = dp;
        if (const mini_toml::Value* sv = cfg.get("scene")) {
            static const char* serial[] = { "single_triangle", "cube", "cube_textured", "cornell_box",
                "cornell_box_parallelogram_light", "monkey", "teapot", "dragon", "spheres", "custom" };
            static const char* cli[] = { "single_triangle", "cube", "cube_textured", "cornell", "cornell_parallelogram",
                "monkey", "teapot", "dragon", "spheres", "custom" };
            if (sv->kind == mini_toml::Value::INT && sv->i >= 0 && sv->i < 10) {
                scene = cli[sv->i], scene_label = serial[sv->i];
            } else if (sv->kind == mini_toml::Value::STRING) {
                std::string low, flat;
                for (char ch : sv->s)
                    low += (char)std::tolower((unsigned char)ch);
                for (char ch : low)
                    if (ch != '_' && ch != '-')
                        flat += ch;
                for (int t = 0; t < 10 && scene.empty(); t++) { // deserialize(), src/config.cpp:403-430
                    std::string f2;
                    for (const char* q = serial[t]; *q; q++)
                        if (*q != '_')
                            f2 += *q;
                    if (flat == f2)
                        scene = cli[t], scene_label = serial[t];
                }
                if (scene.empty()) { // a scene file below data_path
                    obj = data + "/" + sv->s;
                    const size_t slash = sv->s.find_last_of('/'), dot = sv->s.find_last_of('.');
                    const size_t b0 = slash == std::string::npos ? 0 : slash + 1;
                    scene_label = sv->s.substr(b0, dot == std::string::npos || dot < b0 ? std::string::npos : dot - b0);
                }
            }
        }
        std::string od = cfg.string_or("output_dir", "");
        if (!od.empty() && od[0] == '~' && std::getenv("HOME"))
            od.replace(0, 1, std::getenv("HOME"));
        if (od.rfind("$HOME", 0) == 0 && std::getenv("HOME"))
            od.replace(0, 5, std::getenv("HOME"));
        if (!od.empty()) {
            std::string partial;
            for (size_t k = 0; k <= od.size(); k++) // create_directories
                if (k == od.size() || od[k] == '/') {
                    partial = od.substr(0, k);
                    if (!partial.empty())
                        mkdir(partial.c_str(), 0777);
                }
        }
        char stamp[64];
        const std::time_t now = std::time(nullptr);
        std::strftime(stamp, sizeof(stamp), "%Y-%m-%d-%H:%M:%S", std::localtime(&now));
        out = (od.empty() ? std::string(".") : od) + "/" + scene_label + "_" + stamp;
        memset(&f, 0, sizeof(f));
        if (const mini_toml::Value* ft = cfg.get("features")) {
            f.enableShading = ft->boolean_or("enable_shading", false);
            f.enableRecursive = ft->boolean_or("enable_recursive", false);
            f.enableHardShadow = ft->boolean_or("enable_hard_shadow", false);
            f.enableSoftShadow = ft->boolean_or("enable_soft_shadow", false);
            f.enableNormalInterp = ft->boolean_or("enable_normal_interp", false);
            f.enableTextureMapping = ft->boolean_or("enable_texture_mapping", false);
            f.enableAccelStructure = ft->boolean_or("enable_accel_structure", false);
            if (const mini_toml::Value* ex = ft->get("extra")) {
                f.enableBloomEffect = ex->boolean_or("enable_bloom_effect", false);
                f.enableMultipleRaysPerPixel = ex->boolean_or("enable_multiple_rays_per_pixel", false);
                f.enableMotionBlur = ex->boolean_or("enable_motion_blur", false);
                f.enableDepthOfField = ex->boolean_or("enable_depth_of_field", false);
                f.enableGlossyReflection = ex->boolean_or("enable_glossy_reflection", false);
                f.enableEnvironmentMapping = ex->boolean_or("enable_environment_mapping", false);
                f.enableBilinearTextureFiltering = ex->boolean_or("enable_bilinear_texture_filtering", false);
                f.enableMipmapTextureFiltering = ex->boolean_or("enable_mipmap_texture_filtering", false);
                f.enableBvhSahBinning = ex->boolean_or("enable_bvh_sah_binning", false);
                f.enableTransparency = ex->boolean_or("enable_transparency", false);
            }
        }
        if (const mini_toml::Value* cs = cfg.get("cameras"))
            for (const mini_toml::Value& c : cs->a) {
                Cam cam { (float)c.number_or("field_of_view", 50.0), (float)c.number_or("distance_from_look_at", 3.0), 20.0f, 20.0f };
                vec(c.get("look_at"), 3, cam.look);
                float rot[3];
                if (vec(c.get("rotation"), 3, rot))
                    cam.rx = rot[0], cam.ry = rot[1], cam.rz = rot[2];
                cams.push_back(cam);
            }
        if (const mini_toml::Value* ls = cfg.get("lights")) {
            for (const mini_toml::Value& l : ls->a) {
                rt_light_desc ld;
                memset(&ld, 0, sizeof(ld));
                const std::string type = l.string_or("type", "none");
                auto sub = [&](const char* key, size_t idx) -> const mini_toml::Value* {
                    const mini_toml::Value* arr = l.get(key);
                    return arr && arr->kind == mini_toml::Value::ARRAY && idx < arr->a.size() ? &arr->a[idx] : nullptr;
                };
                if (type == "point") { // members in declaration order, src/common.h:37-52
                    ld.kind = RT_LIGHT_POINT;
                    vec(l.get("position"), 3, ld.v[0]), vec(l.get("color"), 3, ld.v[1]);
                } else if (type == "segment") {
                    ld.kind = RT_LIGHT_SEGMENT;
                    vec(sub("endpoints", 0), 3, ld.v[0]), vec(sub("endpoints", 1), 3, ld.v[1]);
                    vec(sub("colors", 0), 3, ld.v[2]), vec(sub("colors", 1), 3, ld.v[3]);
                } else if (type == "parallelogram") {
                    ld.kind = RT_LIGHT_PARALLELOGRAM;
                    vec(l.get("corner"), 3, ld.v[0]), vec(sub("edges", 0), 3, ld.v[1]), vec(sub("edges", 1), 3, ld.v[2]);
                    for (size_t k = 0; k < 4; k++)
                        vec(sub("colors", k), 3, ld.v[3 + k]);
                } else {
                    std::fprintf(stderr, "Unknown light type: %s -- Skip\n", type.c_str());
                    continue;
                }
                cfg_lights.push_back(ld);
            }
        } else {
            std::fprintf(stderr, "WARN: No lights found in config file.\n");
        }
    }
    if (cams.empty() && !have_config)
        cams.push_back(Cam { 50.0f, 3.0f, 20.0f, 20.0f }); // CameraConfig defaults, src/config.h:16-21

    static const char* names[] = { "single_triangle", "cube", "cube_textured", "cornell", "cornell_parallelogram",
        "monkey", "teapot", "dragon", "spheres", "custom" };
    rt_loaded_scene* loaded = nullptr;
    int rc;
    if (!obj.empty()) {
        rt_light_desc light; // loadSceneFromFile gets the config's lights; default: the presets' usual point light
        memset(&light, 0, sizeof(light));
        light.kind = RT_LIGHT_POINT;
        light.v[0][0] = -1, light.v[0][1] = 1, light.v[0][2] = -1, light.v[1][0] = light.v[1][1] = light.v[1][2] = 1;
        if (have_config) // loadSceneFromFile(path, config.lights), src/scene.cpp:94-103 (no normalisation)
            rc = rt_load_obj(obj.c_str(), 0, cfg_lights.empty() ? nullptr : cfg_lights.data(), (uint32_t)cfg_lights.size(), &loaded);
        else
            rc = rt_load_obj(obj.c_str(), normalize ? 1 : 0, &light, 1, &loaded);
    } else {
        int type = -1;
        for (int t = 0; t < 10; t++)
            if (scene == names[t])
                type = t;
        if (type < 0)
            return die("unknown --scene", scene.c_str());
        rc = rt_load_preset(type, data.c_str(), &loaded);
    }
    if (rc != RT_OK)
        return die("loading the scene failed", rt_io_last_error());
    rt_scene_desc desc;
    rt_loaded_scene_desc(loaded, &desc);
    rt_scene* sc = nullptr;
    rt_bvh* bvh = nullptr;
    using clock = std::chrono::high_resolution_clock;
    const auto t0 = clock::now();
    if (rt_scene_create(&desc, &sc) != RT_OK || rt_bvh_build(sc, &f, &bvh) != RT_OK)
        return die("scene/BVH", rt_last_error());
    const auto t1 = clock::now();
    std::printf("BVH generation took %lld ms.\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
    std::printf("levels %d, leaves %d\n", rt_bvh_num_levels(bvh), rt_bvh_num_leaves(bvh));

    const float deg = 0.01745329251994329576923690768489f; // glm::radians
    const rt_render_params params { depth, 5, 25, 0, 3, 5, 3.0f, 0.005f, 0.3f, 0.4f, 0, 3, 40, 1.0f }; // the reference's defaults
    const auto r0 = clock::now();
    // The reference renders its cameras from one std::thread each (src/main.cpp:512-531); here they go to the
    // GPU together: rt_render_frames puts up to eight of them into one launch.
    std::vector<rt_camera> cameras(cams.size());
    for (size_t i = 0; i < cams.size(); i++) {
        const float rot[3] = { cams[i].rx * deg, cams[i].ry * deg, cams[i].rz * deg }; // glm::radians(rotation)
        rt_camera_from_trackball(cams[i].fov * deg, float(W) / float(H), cams[i].look, cams[i].dist, rot, &cameras[i]);
    }
    const size_t frame_floats = (size_t)W * H * 3;
    std::vector<float> frames(frame_floats * cams.size());
    if (rt_render_frames(bvh, cameras.data(), (int32_t)cameras.size(), &f, &params, W, H, frames.data()) != RT_OK)
        return die("rt_render_frames", rt_last_error());
    for (size_t i = 0; i < cams.size(); i++) {
        const std::string path = out + "_cam_" + std::to_string(i) + ".bmp";
        if (rt_write_bmp(path.c_str(), frames.data() + i * frame_floats, W, H) != RT_OK)
            return die("rt_write_bmp", rt_io_last_error());
        std::printf("Image %zu saved to %s\n", i, path.c_str());
    }
    const auto r1 = clock::now();
    std::printf("Rendering took %lld ms, %zu images rendered.\n",
        (long long)std::chrono::duration_cast<std::chrono::milliseconds>(r1 - r0).count(), cams.size());
    rt_bvh_destroy(bvh);
    rt_scene_destroy(sc);
    rt_loaded_scene_destroy(loaded);
    return 0;
}
