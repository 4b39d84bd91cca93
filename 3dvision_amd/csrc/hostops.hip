// Host-side steps that sit directly before / after the GPU path: the ASCII-PLY
// reference-model loader and the duplicate-pose filter.  Plain C++; no device code.
#include "tdv_hip.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

extern "C" {

// Pipeline::filterDuplicates, /root/reference/src/pipeline.cpp:153-180.
int tdv_filter_duplicates(const float* poses, int n, float min_distance, float* out_poses, int* n_out) {
    if (!n_out || n < 0 || (n > 0 && (!poses || !out_poses))) return TDV_ERR_BAD_ARG;
    auto norm3 = [](float x, float y, float z) { return std::sqrt(x * x + (y * y + z * z)); };
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const float* wp = poses + (size_t)i * 16;
        const float px = wp[12], py = wp[13], pz = wp[14];  // block<3,1>(0,3) of a column-major 4x4
        bool is_dup = false;
        for (int k = 0; k < kept; ++k) {
            float* f = out_poses + (size_t)k * 16;
            float dist = norm3(px - f[12], py - f[13], pz - f[14]);
            if (dist < min_distance) {
                is_dup = true;
                float existing = norm3(f[12], f[13], f[14]);
                float current = norm3(px, py, pz);
                if (current < existing) std::memcpy(f, wp, 64);
                break;
            }
        }
        if (!is_dup) { std::memcpy(out_poses + (size_t)kept * 16, wp, 64); ++kept; }
    }
    *n_out = kept;
    return TDV_OK;
}

// Registration::loadReferenceModel, /root/reference/src/registration.cpp:416-461.
int tdv_load_ply_ascii(const char* path, float* out_xyz, float* out_rgb, int capacity, int* n_out, int* has_color_out) {
    if (!path || !n_out || capacity < 0) return TDV_ERR_BAD_ARG;
    *n_out = 0;
    if (has_color_out) *has_color_out = 0;
    std::ifstream file(path);
    if (!file.is_open()) return TDV_ERR_BAD_ARG;
    std::string line;
    int vertex_count = 0;
    bool has_color = false, in_header = true;
    // `while (std::getline(file, line) && in_header)`: getline runs BEFORE the in_header test, so the
    // line following end_header (the first vertex) is read and dropped
    while (std::getline(file, line) && in_header) {
        if (line.find("element vertex") != std::string::npos) std::sscanf(line.c_str(), "element vertex %d", &vertex_count);
        if (line.find("red") != std::string::npos || line.find("diffuse_red") != std::string::npos) has_color = true;
        if (line == "end_header") in_header = false;
    }
    if (has_color_out) *has_color_out = has_color ? 1 : 0;
    int n = 0;
    for (int i = 0; i < vertex_count; ++i) {
        float x = 0.f, y = 0.f, z = 0.f;  // the reference leaves y, z uninitialised when the read fails
        file >> x >> y >> z;
        float r = 0.f, g = 0.f, b = 0.f;
        if (has_color) {
            file >> r >> g >> b;
            if (r > 1.0f) { r /= 255.0f; g /= 255.0f; b /= 255.0f; }
        }
        if (n < capacity) {
            if (out_xyz) { out_xyz[3 * n] = x; out_xyz[3 * n + 1] = y; out_xyz[3 * n + 2] = z; }
            if (out_rgb && has_color) { out_rgb[3 * n] = r; out_rgb[3 * n + 1] = g; out_rgb[3 * n + 2] = b; }
        }
        ++n;
        std::getline(file, line);
    }
    *n_out = n;
    return (n > capacity && (out_xyz || out_rgb)) ? TDV_ERR_BAD_ARG : TDV_OK;
}

}  // extern "C"
