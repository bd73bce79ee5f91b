/*
 * ref_bmp.cpp — container-only: the BMP bytes of Screen::writeBitmapToFile.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md). src/screen.cpp itself cannot be
 * compiled here (it includes framework/opengl_includes.h → GL/glu.h, absent), so this
 * TU compiles the reference's vendored stb_image_write.h BY PATH (the writer whose byte
 * layout rt_write_bmp restates) and performs the five lines of
 * Screen::writeBitmapToFile (src/screen.cpp:49-60) around it with the vendored glm:
 * clamp to [0,1], vec4(c,1) * 255 → u8vec4 (truncation), stbi_write_bmp(…, 4 channels).
 * Built into oracle/_ref/libcge_ref.so by oracle/Makefile.
 */
#include <algorithm>
#include <vector>

#include <glm/common.hpp>
#include <glm/vec3.hpp>
#include <glm/vec4.hpp>
#define STB_IMAGE_WRITE_IMPLEMENTATION
#define STB_IMAGE_WRITE_STATIC
#include <stb/stb_image_write.h>

extern "C" int ref_write_bmp(const char* path, const float* rgb, int width, int height)
{
    const glm::vec3* src = reinterpret_cast<const glm::vec3*>(rgb);
    std::vector<glm::u8vec4> textureData8Bits((size_t)width * (size_t)height);
    std::transform(src, src + textureData8Bits.size(), std::begin(textureData8Bits), [](const glm::vec3& color) {
        const glm::vec3 clampedColor = glm::clamp(color, 0.0f, 1.0f);
        return glm::u8vec4(glm::vec4(clampedColor, 1.0f) * 255.0f);
    });
    return stbi_write_bmp(path, width, height, 4, textureData8Bits.data());
}
