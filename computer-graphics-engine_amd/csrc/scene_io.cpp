// scene_io.cpp — scene ingestion: OBJ/MTL → meshes, PNG →
// fp32 texels, the reference's preset scenes. Host C++, no reference code; every
// step restates what the reference's loader chain computes so that the arrays are
// bit-identical to it (tests/test_scene_io.py compares against the fixtures the
// reference's own loader produced):
//
//   loadMesh / centerAndScaleToUnitMesh     framework/src/mesh.cpp:52-176
//   tinyobjloader 2.0 (vendored third party: float parser tryParseDouble, index
//   fixing, quad split along the shorter diagonal, per-face material ids, .mtl
//   keys Kd/Ks/Ns/d/Tr/map_Kd)               framework/third_party/tinyobjloader
//   Image::Image (stb_image → RGB8 → /255.0f)  framework/src/image.cpp:13-35
//   loadScenePrebuilt / loadSceneFromFile   src/scene.cpp:5-103
//
// Limits (fail loudly): polygons with more than 4 vertices (tinyobjloader's ear
// clipping is not restated), interlaced or 16-bit PNGs.
#include <zlib.h>

#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rt_whitted.h"

namespace {

thread_local std::string g_io_error;

struct Vec3 {
    float x, y, z;
};
inline Vec3 sub(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 normalize(Vec3 v)
{
    const float s = 1.0f / std::sqrt(dot(v, v));
    return { v.x * s, v.y * s, v.z * s };
}

// ---- tinyobjloader's number parser (tiny_obj_loader.h:866-996), restated -------
// The digit-by-digit double accumulation below IS the parity contract for every parsed float (a different
// but equally correct parser rounds differently), so this function follows tinyobjloader's tryParseDouble
// closely — same accumulation order, same names. tinyobjloader is
//   Copyright (c) 2012-Present, Syoyo Fujita and many contributors, The MIT License (MIT):
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without
//   limitation the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the
//   Software, and to permit persons to whom the Software is furnished to do so, subject to the following
//   conditions: The above copyright notice and this permission notice shall be included in all copies or
//   substantial portions of the Software. THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS
//   OR IMPLIED, INCLUDING BUT NOT LIMITED TO THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE
//   AND NONINFRINGEMENT. IN NO EVENT SHALL THE AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR
//   OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT, TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION
//   WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE SOFTWARE.
bool try_parse_double(const char* s, const char* s_end, double* result)
{
    if (s >= s_end)
        return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read = 0;
    bool end_not_reached = false, leading_decimal_dots = false;
    auto is_digit = [](char c) { return c >= '0' && c <= '9'; };
    if (*curr == '+' || *curr == '-') {
        sign = *curr;
        curr++;
        if (curr != s_end && *curr == '.')
            leading_decimal_dots = true;
    } else if (is_digit(*curr)) {
    } else if (*curr == '.') {
        leading_decimal_dots = true;
    } else {
        return false;
    }
    end_not_reached = curr != s_end;
    if (!leading_decimal_dots) {
        while (end_not_reached && is_digit(*curr)) {
            mantissa *= 10;
            mantissa += (int)(*curr - 0x30);
            curr++, read++;
            end_not_reached = curr != s_end;
        }
        if (read == 0)
            return false;
    }
    if (end_not_reached) {
        bool to_assemble = false;
        if (*curr == '.') {
            curr++;
            read = 1;
            end_not_reached = curr != s_end;
            static const double pow_lut[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
            while (end_not_reached && is_digit(*curr)) {
                mantissa += (int)(*curr - 0x30) * (read < 8 ? pow_lut[read] : std::pow(10.0, -read));
                read++, curr++;
                end_not_reached = curr != s_end;
            }
        } else if (*curr == 'e' || *curr == 'E') {
        } else {
            to_assemble = true;
        }
        if (!to_assemble && end_not_reached && (*curr == 'e' || *curr == 'E')) {
            curr++;
            end_not_reached = curr != s_end;
            if (end_not_reached && (*curr == '+' || *curr == '-')) {
                exp_sign = *curr;
                curr++;
            } else if (is_digit(*curr)) {
            } else {
                return false;
            }
            read = 0;
            end_not_reached = curr != s_end;
            while (end_not_reached && is_digit(*curr)) {
                if (exponent > 2147483647 / 10)
                    return false;
                exponent *= 10;
                exponent += (int)(*curr - 0x30);
                curr++, read++;
                end_not_reached = curr != s_end;
            }
            exponent *= (exp_sign == '+' ? 1 : -1);
            if (read == 0)
                return false;
        }
    }
    *result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

float parse_real(const char** token, double default_value = 0.0)
{
    (*token) += strspn(*token, " \t");
    const char* end = (*token) + strcspn(*token, " \t\r");
    double val = default_value;
    try_parse_double(*token, end, &val);
    *token = end;
    return (float)val;
}

std::string parse_string(const char** token)
{
    (*token) += strspn(*token, " \t");
    const size_t e = strcspn(*token, " \t\r");
    std::string s(*token, *token + e);
    (*token) += e;
    return s;
}

struct ObjMaterial {
    std::string name;
    float diffuse[3] = { 0, 0, 0 }, specular[3] = { 0, 0, 0 };
    float shininess = 1.0f, dissolve = 1.0f;
    std::string diffuse_texname;
};

// LoadMtl (tiny_obj_loader.h) — the keys the reference reads
void load_mtl(const std::string& path, std::vector<ObjMaterial>& materials, std::map<std::string, int>& material_map)
{
    std::ifstream in(path);
    if (!in)
        return; // tinyobjloader warns and carries on with the default material
    ObjMaterial material;
    bool first = true, has_d = false;
    std::string linebuf;
    while (std::getline(in, linebuf)) {
        while (!linebuf.empty() && (linebuf.back() == '\n' || linebuf.back() == '\r'))
            linebuf.pop_back();
        const char* token = linebuf.c_str();
        token += strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#')
            continue;
        if (0 == strncmp(token, "newmtl", 6) && (token[6] == ' ' || token[6] == '\t')) {
            if (!first || !material.name.empty()) {
                if (!material.name.empty()) {
                    material_map.insert({ material.name, (int)materials.size() });
                    materials.push_back(material);
                }
            }
            first = false;
            material = ObjMaterial();
            has_d = false;
            token += 7;
            std::stringstream sstr;
            sstr << token;
            material.name = sstr.str();
            continue;
        }
        auto is_sp = [](char c) { return c == ' ' || c == '\t'; };
        if (token[0] == 'K' && token[1] == 'd' && is_sp(token[2])) {
            token += 2;
            for (int k = 0; k < 3; k++)
                material.diffuse[k] = parse_real(&token);
        } else if (token[0] == 'K' && token[1] == 's' && is_sp(token[2])) {
            token += 2;
            for (int k = 0; k < 3; k++)
                material.specular[k] = parse_real(&token);
        } else if (token[0] == 'N' && token[1] == 's' && is_sp(token[2])) {
            token += 2;
            material.shininess = parse_real(&token);
        } else if (token[0] == 'd' && is_sp(token[1])) {
            token += 1;
            material.dissolve = parse_real(&token);
            has_d = true;
        } else if (token[0] == 'T' && token[1] == 'r' && is_sp(token[2])) {
            token += 2;
            if (!has_d)
                material.dissolve = 1.0f - parse_real(&token);
        } else if (0 == strncmp(token, "map_Kd", 6) && is_sp(token[6])) {
            token += 7;
            // texture options (-o, -s, ...) are not used by the reference's assets
            token += strspn(token, " \t");
            material.diffuse_texname = std::string(token);
        }
    }
    if (!material.name.empty()) {
        material_map.insert({ material.name, (int)materials.size() });
        materials.push_back(material);
    }
}

// ---- PNG → RGB8 (what stbi_load(..., STBI_rgb) returns) -------------------------
bool load_png_rgb8(const std::string& path, int& w, int& h, std::vector<uint8_t>& rgb, std::string& err)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) {
        err = "texture file " + path + " does not exist";
        return false;
    }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) {
        err = path + ": not a PNG (only PNG textures are restated)";
        return false;
    }
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | (file[o + 1] << 16) | (file[o + 2] << 8) | file[o + 3]; };
    size_t pos = 8;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    while (pos + 8 <= file.size()) {
        const uint32_t len = be32(pos);
        const std::string type((const char*)&file[pos + 4], 4);
        const size_t data = pos + 8;
        if (data + len + 4 > file.size())
            break;
        if (type == "IHDR") {
            w = (int)be32(data), h = (int)be32(data + 4);
            depth = file[data + 8], color = file[data + 9], interlace = file[data + 12];
        } else if (type == "PLTE") {
            palette.assign(file.begin() + data, file.begin() + data + len);
        } else if (type == "IDAT") {
            idat.insert(idat.end(), file.begin() + data, file.begin() + data + len);
        } else if (type == "IEND") {
            break;
        }
        pos = data + len + 4;
    }
    if (w <= 0 || h <= 0 || interlace != 0 || depth == 16) {
        err = path + ": interlaced / 16-bit PNGs are not restated";
        return false;
    }
    const int channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4;
    const size_t stride = ((size_t)w * channels * depth + 7) / 8;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) {
        err = path + ": zlib stream is damaged";
        return false;
    }
    // unfilter (PNG spec §9), bytes-per-pixel rounded up to 1
    const size_t bpp = std::max<size_t>(1, (size_t)channels * depth / 8);
    std::vector<uint8_t> img(stride * (size_t)h);
    for (int y = 0; y < h; y++) {
        const uint8_t ft = raw[(stride + 1) * (size_t)y];
        const uint8_t* src = &raw[(stride + 1) * (size_t)y + 1];
        uint8_t* dst = &img[stride * (size_t)y];
        const uint8_t* up = y ? &img[stride * (size_t)(y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int v = src[i];
            switch (ft) {
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: break;
            }
            dst[i] = (uint8_t)v;
        }
    }
    rgb.resize((size_t)w * h * 3);
    // stb_image scales 1/2/4-bit grey to 8 bits with these factors (stbi__depth_scale_table)
    static const uint8_t depth_scale[9] = { 0, 0xff, 0x55, 0, 0x11, 0, 0, 0, 0x01 };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t s[4] = { 0, 0, 0, 255 };
            for (int c = 0; c < channels; c++) {
                const size_t bit = ((size_t)x * channels + c) * depth;
                const uint8_t byte = img[stride * (size_t)y + bit / 8];
                s[c] = depth == 8 ? byte : (uint8_t)((byte >> (8 - depth - bit % 8)) & ((1 << depth) - 1));
            }
            uint8_t* o = &rgb[((size_t)y * w + x) * 3];
            if (color == 3) {
                const size_t pi = (size_t)s[0] * 3;
                for (int c = 0; c < 3; c++)
                    o[c] = pi + c < palette.size() ? palette[pi + c] : 0;
            } else if (color == 0 || color == 4) {
                const uint8_t g = depth < 8 ? (uint8_t)(s[0] * depth_scale[depth]) : s[0];
                o[0] = o[1] = o[2] = g;
            } else {
                o[0] = s[0], o[1] = s[1], o[2] = s[2];
            }
        }
    return true;
}

struct LoadedMesh {
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> triangles;
    rt_material material;
};
struct LoadedTexture {
    std::string path;
    int width = 0, height = 0;
    std::vector<float> pixels;
};

// key of the vertex cache (mesh.cpp:36-50, 88, 111-118): all eight floats, by value
struct VertexKey {
    float v[8];
    bool operator==(const VertexKey& o) const
    {
        for (int i = 0; i < 8; i++)
            if (!(v[i] == o.v[i]))
                return false;
        return true;
    }
};
struct VertexKeyHash {
    size_t operator()(const VertexKey& k) const
    {
        size_t seed = 0;
        for (int i = 0; i < 8; i++)
            seed ^= std::hash<float>()(k.v[i]) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
        return seed;
    }
};

} // namespace

struct rt_loaded_scene {
    std::vector<LoadedMesh> meshes;
    std::vector<LoadedTexture> textures;
    std::vector<rt_sphere_desc> spheres;
    std::vector<rt_light_desc> lights;
    // descriptor storage
    std::vector<rt_mesh_desc> mesh_descs;
    std::vector<rt_texture_desc> texture_descs;
};

namespace {

int io_fail(const std::string& msg)
{
    g_io_error = msg;
    return RT_ERR_INVALID;
}

int texture_index(rt_loaded_scene& sc, const std::string& path, std::string& err)
{
    for (size_t i = 0; i < sc.textures.size(); i++)
        if (sc.textures[i].path == path)
            return (int)i; // (each reference mesh owns its own Image; pixels are identical)
    LoadedTexture t;
    t.path = path;
    std::vector<uint8_t> rgb;
    if (!load_png_rgb8(path, t.width, t.height, rgb, err))
        return -2;
    t.pixels.resize(rgb.size());
    for (size_t i = 0; i < rgb.size(); i++)
        t.pixels[i] = rgb[i] / 255.0f; // image.cpp:31
    sc.textures.push_back(std::move(t));
    return (int)sc.textures.size() - 1;
}

// loadMesh (mesh.cpp:52-149) on top of a restated tinyobj::LoadObj
int load_obj(rt_loaded_scene& sc, const std::string& file, bool center_and_normalize)
{
    std::ifstream in(file);
    if (!in)
        return io_fail("File " + file + " does not exist."); // mesh.cpp:54-57
    const size_t slash = file.find_last_of('/');
    const std::string base_dir = slash == std::string::npos ? std::string() : file.substr(0, slash + 1);

    std::vector<float> v, vn, vt;
    std::vector<ObjMaterial> materials;
    std::map<std::string, int> material_map;
    struct Idx {
        int v, vt, vn;
    };
    struct Shape {
        std::vector<Idx> indices; // 3 per triangle
        std::vector<int> material_ids;
    };
    std::vector<Shape> shapes;
    Shape shape;
    std::vector<std::vector<Idx>> face_group;
    int material = -1;

    auto fix_index = [](int idx, int n, int* ret) {
        if (idx > 0) {
            *ret = idx - 1;
            return true;
        }
        if (idx == 0)
            return false;
        *ret = n + idx;
        return true;
    };
    std::string fail_msg;
    auto export_group = [&]() { // exportGroupsToShape, faces only
        for (const auto& face : face_group) {
            const size_t np = face.size();
            if (np < 3)
                continue;
            if (np == 4) {
                const Idx &i0 = face[0], &i1 = face[1], &i2 = face[2], &i3 = face[3];
                bool ok = true;
                for (const Idx* i : { &i0, &i1, &i2, &i3 })
                    if ((size_t)(3 * i->v + 2) >= v.size())
                        ok = false;
                if (!ok)
                    continue;
                auto P = [&](const Idx& i, int c) { return v[(size_t)i.v * 3 + c]; };
                const float e02x = P(i2, 0) - P(i0, 0), e02y = P(i2, 1) - P(i0, 1), e02z = P(i2, 2) - P(i0, 2);
                const float e13x = P(i3, 0) - P(i1, 0), e13y = P(i3, 1) - P(i1, 1), e13z = P(i3, 2) - P(i1, 2);
                const float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                const float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                if (sqr02 < sqr13) {
                    for (const Idx& i : { i0, i1, i2, i0, i2, i3 })
                        shape.indices.push_back(i);
                } else {
                    for (const Idx& i : { i0, i1, i3, i1, i2, i3 })
                        shape.indices.push_back(i);
                }
                shape.material_ids.push_back(material);
                shape.material_ids.push_back(material);
            } else if (np == 3) {
                for (const Idx& i : face)
                    shape.indices.push_back(i);
                shape.material_ids.push_back(material);
            } else {
                // tinyobjloader's built-in ear clipping (tiny_obj_loader.h:1540-1819), np > 4:
                // pick the two axes of the first non-degenerate corner, then clip ears.
                size_t axes[2] = { 1, 2 };
                for (size_t k = 0; k < np; ++k) {
                    const Idx &i0 = face[(k + 0) % np], &i1 = face[(k + 1) % np], &i2 = face[(k + 2) % np];
                    const size_t vi0 = (size_t)i0.v, vi1 = (size_t)i1.v, vi2 = (size_t)i2.v;
                    if (((3 * vi0 + 2) >= v.size()) || ((3 * vi1 + 2) >= v.size()) || ((3 * vi2 + 2) >= v.size()))
                        continue;
                    const float v0x = v[vi0 * 3 + 0], v0y = v[vi0 * 3 + 1], v0z = v[vi0 * 3 + 2];
                    const float v1x = v[vi1 * 3 + 0], v1y = v[vi1 * 3 + 1], v1z = v[vi1 * 3 + 2];
                    const float v2x = v[vi2 * 3 + 0], v2y = v[vi2 * 3 + 1], v2z = v[vi2 * 3 + 2];
                    const float e0x = v1x - v0x, e0y = v1y - v0y, e0z = v1z - v0z;
                    const float e1x = v2x - v1x, e1y = v2y - v1y, e1z = v2z - v1z;
                    const float cx = std::fabs(e0y * e1z - e0z * e1y);
                    const float cy = std::fabs(e0z * e1x - e0x * e1z);
                    const float cz = std::fabs(e0x * e1y - e0y * e1x);
                    const float epsilon = std::numeric_limits<float>::epsilon();
                    if (cx > epsilon || cy > epsilon || cz > epsilon) {
                        if (cx > cy && cx > cz) {
                        } else {
                            axes[0] = 0;
                            if (cz > cx && cz > cy)
                                axes[1] = 1;
                        }
                        break;
                    }
                }
                auto pnpoly = [](int nvert, const float* vertx, const float* verty, float testx, float testy) {
                    int i, j, c = 0;
                    for (i = 0, j = nvert - 1; i < nvert; j = i++) {
                        if (((verty[i] > testy) != (verty[j] > testy))
                            && (testx < (vertx[j] - vertx[i]) * (testy - verty[i]) / (verty[j] - verty[i]) + vertx[i]))
                            c = !c;
                    }
                    return c;
                };
                std::vector<Idx> remaining = face;
                size_t guess_vert = 0;
                Idx ind[3];
                float vx[3], vy[3];
                size_t remainingIterations = face.size();
                size_t previousRemainingVertices = remaining.size();
                while (remaining.size() > 3 && remainingIterations > 0) {
                    const size_t npolys = remaining.size();
                    if (guess_vert >= npolys)
                        guess_vert -= npolys;
                    if (previousRemainingVertices != npolys) {
                        previousRemainingVertices = npolys;
                        remainingIterations = npolys;
                    } else {
                        remainingIterations--;
                    }
                    for (size_t k = 0; k < 3; k++) {
                        ind[k] = remaining[(guess_vert + k) % npolys];
                        const size_t vi = (size_t)ind[k].v;
                        if (((vi * 3 + axes[0]) >= v.size()) || ((vi * 3 + axes[1]) >= v.size())) {
                            vx[k] = 0.0f;
                            vy[k] = 0.0f;
                        } else {
                            vx[k] = v[vi * 3 + axes[0]];
                            vy[k] = v[vi * 3 + axes[1]];
                        }
                    }
                    const float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0];
                    const float e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
                    const float cross = e0x * e1y - e0y * e1x;
                    const float area = (vx[0] * vy[1] - vy[0] * vx[1]) * 0.5f;
                    if (cross * area < 0.0f) { // an internal angle
                        guess_vert += 1;
                        continue;
                    }
                    bool overlap = false;
                    for (size_t otherVert = 3; otherVert < npolys; ++otherVert) {
                        const size_t idx = (guess_vert + otherVert) % npolys;
                        if (idx >= remaining.size())
                            continue;
                        const size_t ovi = (size_t)remaining[idx].v;
                        if (((ovi * 3 + axes[0]) >= v.size()) || ((ovi * 3 + axes[1]) >= v.size()))
                            continue;
                        const float tx = v[ovi * 3 + axes[0]], ty = v[ovi * 3 + axes[1]];
                        if (pnpoly(3, vx, vy, tx, ty)) {
                            overlap = true;
                            break;
                        }
                    }
                    if (overlap) {
                        guess_vert += 1;
                        continue;
                    }
                    for (int k = 0; k < 3; k++) // this triangle is an ear
                        shape.indices.push_back(ind[k]);
                    shape.material_ids.push_back(material);
                    size_t removed_vert_index = (guess_vert + 1) % npolys; // remove v1 from the list
                    while (removed_vert_index + 1 < npolys) {
                        remaining[removed_vert_index] = remaining[removed_vert_index + 1];
                        removed_vert_index += 1;
                    }
                    remaining.pop_back();
                }
                if (remaining.size() == 3) {
                    for (int k = 0; k < 3; k++)
                        shape.indices.push_back(remaining[k]);
                    shape.material_ids.push_back(material);
                }
            }
        }
    };

    std::string linebuf;
    while (std::getline(in, linebuf)) {
        while (!linebuf.empty() && (linebuf.back() == '\n' || linebuf.back() == '\r'))
            linebuf.pop_back();
        if (linebuf.empty())
            continue;
        const char* token = linebuf.c_str();
        token += strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#')
            continue;
        auto is_sp = [](char c) { return c == ' ' || c == '\t'; };
        if (token[0] == 'v' && is_sp(token[1])) {
            token += 2;
            for (int k = 0; k < 3; k++)
                v.push_back(parse_real(&token));
            continue;
        }
        if (token[0] == 'v' && token[1] == 'n' && is_sp(token[2])) {
            token += 3;
            for (int k = 0; k < 3; k++)
                vn.push_back(parse_real(&token));
            continue;
        }
        if (token[0] == 'v' && token[1] == 't' && is_sp(token[2])) {
            token += 3;
            for (int k = 0; k < 2; k++)
                vt.push_back(parse_real(&token));
            continue;
        }
        if (token[0] == 'f' && is_sp(token[1])) {
            token += 2;
            token += strspn(token, " \t");
            std::vector<Idx> face;
            while (token[0] != '\0' && token[0] != '\r' && token[0] != '\n') {
                Idx vi { -1, -1, -1 };
                // parseTriple (tiny_obj_loader.h:1136-1190)
                if (!fix_index(atoi(token), (int)(v.size() / 3), &vi.v))
                    return io_fail(file + ": failed to parse an `f' line");
                token += strcspn(token, "/ \t\r");
                if (token[0] == '/') {
                    token++;
                    if (token[0] == '/') {
                        token++;
                        if (!fix_index(atoi(token), (int)(vn.size() / 3), &vi.vn))
                            return io_fail(file + ": failed to parse an `f' line");
                        token += strcspn(token, "/ \t\r");
                    } else {
                        if (!fix_index(atoi(token), (int)(vt.size() / 2), &vi.vt))
                            return io_fail(file + ": failed to parse an `f' line");
                        token += strcspn(token, "/ \t\r");
                        if (token[0] == '/') {
                            token++;
                            if (!fix_index(atoi(token), (int)(vn.size() / 3), &vi.vn))
                                return io_fail(file + ": failed to parse an `f' line");
                            token += strcspn(token, "/ \t\r");
                        }
                    }
                }
                face.push_back(vi);
                token += strspn(token, " \t\r");
            }
            face_group.push_back(std::move(face));
            continue;
        }
        if (0 == strncmp(token, "usemtl", 6)) {
            token += 6;
            const std::string name = parse_string(&token);
            int new_id = -1;
            auto it = material_map.find(name);
            if (it != material_map.end())
                new_id = it->second;
            if (new_id != material) {
                export_group();
                face_group.clear();
                material = new_id;
            }
            continue;
        }
        if (0 == strncmp(token, "mtllib", 6) && is_sp(token[6])) {
            token += 7;
            std::stringstream ss(token);
            std::string fname;
            while (ss >> fname) {
                const size_t before = materials.size();
                load_mtl(base_dir + fname, materials, material_map);
                if (materials.size() != before || std::ifstream(base_dir + fname))
                    break; // first file that could be read
            }
            continue;
        }
        if ((token[0] == 'g' || token[0] == 'o') && is_sp(token[1])) {
            export_group();
            if (!shape.indices.empty())
                shapes.push_back(shape);
            shape = Shape();
            face_group.clear();
            continue;
        }
        // s, t, l, p, ...: nothing the reference reads
    }
    export_group();
    if (!shape.indices.empty())
        shapes.push_back(shape);
    if (!fail_msg.empty())
        return io_fail(fail_msg);

    const size_t first_mesh = sc.meshes.size();
    for (const Shape& sh : shapes) {
        const size_t n_tri = sh.indices.size() / 3;
        size_t start = 0;
        int prev_mat = sh.material_ids[0];
        for (size_t end = 0; end < n_tri; ++end) { // mesh.cpp:78-142, including its end-of-shape quirk
            if (end == n_tri - 1)
                ++end;
            else if (sh.material_ids[end] == prev_mat)
                continue;
            else
                prev_mat = sh.material_ids[end];
            LoadedMesh mesh;
            std::unordered_map<VertexKey, uint32_t, VertexKeyHash> cache;
            for (size_t i = start * 3; i != end * 3; i += 3) {
                auto P = [&](const Idx& ix) { return Vec3 { v[3 * (size_t)ix.v], v[3 * (size_t)ix.v + 1], v[3 * (size_t)ix.v + 2] }; };
                const Vec3 v0 = P(sh.indices[i]), v1 = P(sh.indices[i + 1]), v2 = P(sh.indices[i + 2]);
                const Vec3 gn = normalize(cross(sub(v1, v0), sub(v2, v0)));
                for (unsigned j = 0; j < 3; j++) {
                    const Idx& ix = sh.indices[i + j];
                    VertexKey key;
                    const Vec3 p = P(ix);
                    key.v[0] = p.x, key.v[1] = p.y, key.v[2] = p.z;
                    if (ix.vn != -1 && !vn.empty())
                        key.v[3] = vn[3 * (size_t)ix.vn], key.v[4] = vn[3 * (size_t)ix.vn + 1], key.v[5] = vn[3 * (size_t)ix.vn + 2];
                    else
                        key.v[3] = gn.x, key.v[4] = gn.y, key.v[5] = gn.z;
                    key.v[6] = key.v[7] = 0.0f;
                    if (ix.vt != -1 && !vt.empty())
                        key.v[6] = vt[2 * (size_t)ix.vt], key.v[7] = vt[2 * (size_t)ix.vt + 1];
                    auto it = cache.find(key);
                    uint32_t id;
                    if (it != cache.end()) {
                        id = it->second;
                    } else {
                        id = (uint32_t)(mesh.positions.size() / 3);
                        cache[key] = id;
                        mesh.positions.insert(mesh.positions.end(), { key.v[0], key.v[1], key.v[2] });
                        mesh.normals.insert(mesh.normals.end(), { key.v[3], key.v[4], key.v[5] });
                        mesh.texcoords.insert(mesh.texcoords.end(), { key.v[6], key.v[7] });
                    }
                    mesh.triangles.push_back(id);
                }
            }
            memset(&mesh.material, 0, sizeof(mesh.material));
            mesh.material.shininess = 1.0f, mesh.material.transparency = 1.0f, mesh.material.texture = -1;
            const int mid = sh.material_ids[start];
            if (mid == -1) {
                for (int k = 0; k < 3; k++)
                    mesh.material.kd[k] = 1.0f;
            } else {
                const ObjMaterial& om = materials[(size_t)mid];
                memcpy(mesh.material.kd, om.diffuse, 12);
                memcpy(mesh.material.ks, om.specular, 12);
                mesh.material.shininess = om.shininess;
                mesh.material.transparency = om.dissolve;
                if (!om.diffuse_texname.empty()) {
                    std::string err;
                    const int t = texture_index(sc, base_dir + om.diffuse_texname, err);
                    if (t < 0)
                        return io_fail(err);
                    mesh.material.texture = t;
                }
            }
            sc.meshes.push_back(std::move(mesh));
            start = end;
        }
    }
    if (center_and_normalize && sc.meshes.size() > first_mesh) { // mesh.cpp:151-176
        float cx = 0.0f, cy = 0.0f, cz = 0.0f; // std::accumulate, sequential binary32 sums
        size_t n = 0;
        for (size_t m = first_mesh; m < sc.meshes.size(); m++)
            for (size_t i = 0; i < sc.meshes[m].positions.size(); i += 3) {
                cx = cx + sc.meshes[m].positions[i], cy = cy + sc.meshes[m].positions[i + 1], cz = cz + sc.meshes[m].positions[i + 2];
                n++;
            }
        const float fn = (float)n;
        cx = cx / fn, cy = cy / fn, cz = cz / fn;
        float max_d = 0.0f;
        for (size_t m = first_mesh; m < sc.meshes.size(); m++)
            for (size_t i = 0; i < sc.meshes[m].positions.size(); i += 3) {
                const Vec3 d { sc.meshes[m].positions[i] - cx, sc.meshes[m].positions[i + 1] - cy, sc.meshes[m].positions[i + 2] - cz };
                max_d = std::max(std::sqrt(dot(d, d)), max_d);
            }
        for (size_t m = first_mesh; m < sc.meshes.size(); m++)
            for (size_t i = 0; i < sc.meshes[m].positions.size(); i += 3) {
                float* p = &sc.meshes[m].positions[i];
                p[0] = (p[0] - cx) / max_d, p[1] = (p[1] - cy) / max_d, p[2] = (p[2] - cz) / max_d;
            }
    }
    return RT_OK;
}

rt_light_desc make_light(int kind, std::initializer_list<std::initializer_list<float>> vs)
{
    rt_light_desc d;
    memset(&d, 0, sizeof(d));
    d.kind = kind;
    int i = 0;
    for (const auto& v : vs) {
        int c = 0;
        for (float f : v)
            d.v[i][c++] = f;
        i++;
    }
    return d;
}

rt_sphere_desc make_sphere(float cx, float cy, float cz, float r, float kr, float kg, float kb)
{
    rt_sphere_desc s;
    memset(&s, 0, sizeof(s));
    s.center[0] = cx, s.center[1] = cy, s.center[2] = cz, s.radius = r;
    s.material.kd[0] = kr, s.material.kd[1] = kg, s.material.kd[2] = kb;
    s.material.shininess = 1.0f, s.material.transparency = 1.0f, s.material.texture = -1; // Material{kd} defaults
    return s;
}

} // namespace

extern "C" {

const char* rt_io_last_error(void) { return g_io_error.c_str(); }

// ≙ loadSceneFromFile (src/scene.cpp:94-103): loadMesh(path) + the given lights
int rt_load_obj(const char* path, int center_and_normalize, const rt_light_desc* lights, uint32_t n_lights,
    rt_loaded_scene** out)
{
    if (!path || !out || (n_lights && !lights))
        return io_fail("rt_load_obj: null argument");
    std::unique_ptr<rt_loaded_scene> sc(new rt_loaded_scene);
    const int rc = load_obj(*sc, path, center_and_normalize != 0);
    if (rc != RT_OK)
        return rc;
    sc->lights.assign(lights, lights + n_lights);
    *out = sc.release();
    return RT_OK;
}

// ≙ loadScenePrebuilt (src/scene.cpp:5-92); scene_type = the SceneType enum value
int rt_load_preset(int scene_type, const char* data_dir, rt_loaded_scene** out)
{
    if (!data_dir || !out)
        return io_fail("rt_load_preset: null argument");
    std::string dir(data_dir);
    if (!dir.empty() && dir.back() != '/')
        dir += '/';
    std::unique_ptr<rt_loaded_scene> sc(new rt_loaded_scene);
    int rc = RT_OK;
    switch (scene_type) {
    case 0: // SingleTriangle
        rc = load_obj(*sc, dir + "triangle.obj", false);
        if (rc == RT_OK)
            for (int k = 0; k < 3; k++)
                sc->meshes[0].material.kd[k] = 1.0f;
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { -1, 1, -1 }, { 1, 1, 1 } }));
        break;
    case 1: // Cube
        rc = load_obj(*sc, dir + "cube.obj", false);
        sc->lights.push_back(make_light(RT_LIGHT_SEGMENT,
            { { 1.5f, 0.5f, -0.6f }, { -1, 0.5f, -0.5f }, { 0.9f, 0.2f, 0.1f }, { 0.2f, 1, 0.3f } }));
        break;
    case 2: // CubeTextured
        rc = load_obj(*sc, dir + "cube-textured.obj", false);
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { -1.0f, 1.5f, -1.0f }, { 1, 1, 1 } }));
        break;
    case 3: // CornellBox
        rc = load_obj(*sc, dir + "CornellBox-Mirror-Rotated.obj", true);
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { 0, 0.58f, 0 }, { 1, 1, 1 } }));
        break;
    case 4: // CornellBoxParallelogramLight
        rc = load_obj(*sc, dir + "CornellBox-Mirror-Rotated.obj", true);
        sc->lights.push_back(make_light(RT_LIGHT_PARALLELOGRAM,
            { { -0.2f, 0.5f, 0 }, { 0.4f, 0, 0 }, { 0.0f, 0.0f, 0.4f }, { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 }, { 0, 1, 1 } }));
        break;
    case 5: // Monkey
        rc = load_obj(*sc, dir + "monkey.obj", true);
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { -1, 1, -1 }, { 1, 1, 1 } }));
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { 1, -1, -1 }, { 1, 1, 1 } }));
        break;
    case 6: // Teapot
        rc = load_obj(*sc, dir + "teapot.obj", true);
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { -1, 1, -1 }, { 1, 1, 1 } }));
        break;
    case 7: // Dragon
        rc = load_obj(*sc, dir + "dragon.obj", true);
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { -1, 1, -1 }, { 1, 1, 1 } }));
        break;
    case 8: // Spheres
        sc->spheres.push_back(make_sphere(3.0f, -2.0f, 10.2f, 1.0f, 0.8f, 0.2f, 0.2f));
        sc->spheres.push_back(make_sphere(-2.0f, 2.0f, 4.0f, 2.0f, 0.6f, 0.8f, 0.2f));
        sc->spheres.push_back(make_sphere(0.0f, 0.0f, 6.0f, 0.75f, 0.2f, 0.2f, 0.8f));
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { 3, 0, 3 }, { 15, 15, 15 } }));
        break;
    case 9: // Custom
        rc = load_obj(*sc, dir + "custom.obj", false);
        sc->lights.push_back(make_light(RT_LIGHT_POINT, { { -1, 1, -1 }, { 1, 1, 1 } }));
        break;
    default:
        return io_fail("rt_load_preset: unknown scene type");
    }
    if (rc != RT_OK)
        return rc;
    *out = sc.release();
    return RT_OK;
}

// Fill an rt_scene_desc whose pointers stay valid until rt_loaded_scene_destroy.
int rt_loaded_scene_desc(rt_loaded_scene* sc, rt_scene_desc* out)
{
    if (!sc || !out)
        return io_fail("rt_loaded_scene_desc: null argument");
    sc->mesh_descs.resize(sc->meshes.size());
    for (size_t m = 0; m < sc->meshes.size(); m++) {
        const LoadedMesh& lm = sc->meshes[m];
        rt_mesh_desc& d = sc->mesh_descs[m];
        d.positions = lm.positions.data(), d.normals = lm.normals.data(), d.texcoords = lm.texcoords.data();
        d.triangles = lm.triangles.data();
        d.n_vertices = (uint32_t)(lm.positions.size() / 3), d.n_triangles = (uint32_t)(lm.triangles.size() / 3);
        d.material = lm.material;
    }
    sc->texture_descs.resize(sc->textures.size());
    for (size_t t = 0; t < sc->textures.size(); t++)
        sc->texture_descs[t] = rt_texture_desc { sc->textures[t].pixels.data(), sc->textures[t].width, sc->textures[t].height };
    out->meshes = sc->mesh_descs.data(), out->n_meshes = (uint32_t)sc->mesh_descs.size();
    out->spheres = sc->spheres.data(), out->n_spheres = (uint32_t)sc->spheres.size();
    out->lights = sc->lights.data(), out->n_lights = (uint32_t)sc->lights.size();
    out->textures = sc->texture_descs.data(), out->n_textures = (uint32_t)sc->texture_descs.size();
    return RT_OK;
}

int rt_loaded_scene_destroy(rt_loaded_scene* sc)
{
    delete sc;
    return RT_OK;
}

// ≙ Screen::writeBitmapToFile (src/screen.cpp:49-60): clamp to [0,1], u8vec4(vec4(c,1)*255)
// (truncation; NaN → 0 like cvttss2si's low byte), then stb_image_write's 4-channel BMP:
// 14-byte file header + 108-byte V4 header (BI_BITFIELDS, 32 bpp), rows bottom-up, B G R A.
// `rgb` is the row-flipped W*H*3 frame of rt_render (row 0 = top of the image).
int rt_write_bmp(const char* path, const float* rgb, int32_t width, int32_t height)
{
    if (!path || !rgb || width <= 0 || height <= 0)
        return io_fail("rt_write_bmp: bad argument");
    std::vector<uint8_t> out;
    auto u16 = [&](uint32_t v) { out.push_back(v & 255), out.push_back((v >> 8) & 255); };
    auto u32 = [&](uint32_t v) { u16(v & 0xffff), u16(v >> 16); };
    out.push_back('B'), out.push_back('M');
    u32(14 + 108 + (uint32_t)width * height * 4), u16(0), u16(0), u32(14 + 108);
    u32(108), u32((uint32_t)width), u32((uint32_t)height), u16(1), u16(32), u32(3);
    for (int i = 0; i < 5; i++)
        u32(0);
    u32(0xff0000), u32(0xff00), u32(0xff), u32(0xff000000u);
    for (int i = 0; i < 13; i++)
        u32(0);
    auto to_u8 = [](float c) -> uint8_t {
        const float lo = c < 0.0f ? 0.0f : c;   // glm::max(x, 0): (x < 0) ? 0 : x — NaN stays NaN
        const float cl = 1.0f < lo ? 1.0f : lo; // glm::min(., 1): (1 < x) ? 1 : x
        const float s = cl * 255.0f;
        if (!(s > -2147483904.0f && s < 2147483648.0f))
            return 0; // cvttss2si: 0x80000000 → low byte 0
        return (uint8_t)(int32_t)s;
    };
    for (int32_t j = height - 1; j >= 0; j--)
        for (int32_t i = 0; i < width; i++) {
            const float* p = rgb + ((size_t)j * width + i) * 3;
            out.push_back(to_u8(p[2])), out.push_back(to_u8(p[1])), out.push_back(to_u8(p[0])), out.push_back(255);
        }
    FILE* f = std::fopen(path, "wb");
    if (!f)
        return io_fail(std::string("rt_write_bmp: cannot open ") + path);
    const size_t n = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return n == out.size() ? RT_OK : io_fail("rt_write_bmp: short write");
}

} // extern "C"
