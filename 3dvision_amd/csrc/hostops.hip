// Host-side steps that sit directly before / after the GPU path: the ASCII-PLY
// reference-model loader, the duplicate-pose filter and the mask-directory loader.  Plain C++; no device code.
#include "tdv_hip.h"
#include <zlib.h>
#include <algorithm>
#include <cstdlib>
#include <filesystem>
#include <iterator>
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
        if (!file) {
            // Once a read has failed every later `file >> x` is a no-op: the reference keeps pushing vertex_count points
            // whose values are whatever its uninitialised locals hold.  Here they are zeros, written without looping over a
            // (possibly hostile, up to 2^31) header count.
            for (long long j = n; j < (long long)std::min(vertex_count, capacity); ++j) {
                if (out_xyz) { out_xyz[3 * j] = 0.f; out_xyz[3 * j + 1] = 0.f; out_xyz[3 * j + 2] = 0.f; }
                if (out_rgb && has_color) { out_rgb[3 * j] = 0.f; out_rgb[3 * j + 1] = 0.f; out_rgb[3 * j + 2] = 0.f; }
            }
            n = vertex_count;
            break;
        }
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

// ---- PNG mask loading (Segmentation::loadMasksFromDir, /root/reference/src/segmentation.cpp:12-42) -------------------
// The reference reads every .png/.jpg/.jpeg of a directory in sorted order as 8-bit grey (cv::imread, IMREAD_GRAYSCALE)
// and thresholds it (> 10 -> 255, else 0).  Here: non-interlaced greyscale PNGs (colour type 0 or 4, bit depth 1-16),
// decoded with zlib and the five PNG row filters.  Sub-byte depths expand as libpng does (x255, x85, x17), 16-bit
// keeps the high byte, an alpha channel is dropped.  Colour / palette PNGs and JPEGs need a colour-to-grey rule that
// depends on the image library's build, so they are reported as unsupported instead of guessed.
namespace {
inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
int decode_grey_png(const std::vector<unsigned char>& file, std::vector<uint8_t>& grey, int& w, int& h) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return TDV_ERR_BAD_ARG;
    size_t pos = 8;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<unsigned char> idat;
    bool have_ihdr = false, end = false;
    while (!end && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return TDV_ERR_BAD_ARG;
        const unsigned char* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return TDV_ERR_BAD_ARG;
            w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w <= 0 || h <= 0 || (long long)w * h > (1ll << 28)) return TDV_ERR_BAD_ARG;
    if (interlace != 0 || !(ctype == 0 || ctype == 4)) return TDV_ERR_BAD_ARG;   // unsupported: colour, palette, Adam7
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) || (ctype == 4 && depth < 8)) return TDV_ERR_BAD_ARG;
    const int channels = ctype == 4 ? 2 : 1;
    const size_t bpp_bits = (size_t)depth * channels;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    const size_t bpp = std::max<size_t>(1, bpp_bits / 8);   // filter distance in bytes
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return TDV_ERR_BAD_ARG;
    grey.assign((size_t)w * h, 0);
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    for (int y = 0; y < h; ++y) {
        const unsigned char* line = &raw[(stride + 1) * (size_t)y];
        const int ft = line[0];
        if (ft > 4) return TDV_ERR_BAD_ARG;
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
            int v = line[1 + x];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            cur[x] = (unsigned char)v;
        }
        uint8_t* o = &grey[(size_t)y * w];
        for (int x = 0; x < w; ++x) {
            unsigned g;
            if (depth == 8) g = cur[(size_t)x * channels];
            else if (depth == 16) g = cur[(size_t)x * channels * 2];   // high byte
            else {
                const size_t bit = (size_t)x * depth;
                const unsigned s = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                g = s * (depth == 1 ? 255u : depth == 2 ? 85u : 17u);
            }
            o[x] = (uint8_t)g;
        }
        prev.swap(cur);
    }
    return TDV_OK;
}
int read_file(const char* path, std::vector<unsigned char>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return TDV_ERR_BAD_ARG;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return TDV_OK;
}
}  // namespace

extern "C" int tdv_load_mask_png(const char* path, uint8_t* out, long long capacity, int* width, int* height) {
    if (!path || !width || !height || capacity < 0) return TDV_ERR_BAD_ARG;
    *width = 0; *height = 0;
    std::vector<unsigned char> file; std::vector<uint8_t> grey; int w = 0, h = 0;
    int st = read_file(path, file);
    if (st != TDV_OK) return st;
    st = decode_grey_png(file, grey, w, h);
    if (st != TDV_OK) return st;
    *width = w; *height = h;
    if ((long long)w * h > capacity || !out) return out ? TDV_ERR_BAD_ARG : TDV_OK;   // out == NULL: size query
    for (size_t i = 0; i < grey.size(); ++i) out[i] = grey[i] > 10 ? 255 : 0;          // cv::threshold(mask, binary, 10, 255, THRESH_BINARY)
    return TDV_OK;
}

extern "C" int tdv_load_masks_from_dir(const char* dir, int width, int height, uint8_t* out, int capacity_masks, int* n_out, int* n_skipped) {
    if (!dir || !n_out || width <= 0 || height <= 0 || capacity_masks < 0) return TDV_ERR_BAD_ARG;
    *n_out = 0;
    if (n_skipped) *n_skipped = 0;
    namespace fs = std::filesystem;
    std::error_code ec;
    if (!fs::is_directory(dir, ec)) return TDV_OK;   // the reference prints a message and returns no masks
    std::vector<fs::path> files;
    for (const auto& entry : fs::directory_iterator(dir, ec)) {
        std::string ext = entry.path().extension().string();
        std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
        if (ext == ".png" || ext == ".jpg" || ext == ".jpeg") files.push_back(entry.path());
    }
    std::sort(files.begin(), files.end());
    int n = 0, skipped = 0;
    for (const auto& f : files) {
        int w = 0, h = 0;
        std::vector<uint8_t> m((size_t)width * height);
        const int st = tdv_load_mask_png(f.string().c_str(), m.data(), (long long)m.size(), &w, &h);
        if (st != TDV_OK || w != width || h != height) { ++skipped; continue; }   // undecodable here (JPEG, colour PNG) or a different frame size
        if (n < capacity_masks && out) std::memcpy(out + (size_t)n * width * height, m.data(), m.size());
        ++n;
    }
    *n_out = n;
    if (n_skipped) *n_skipped = skipped;
    return (out && n > capacity_masks) ? TDV_ERR_BAD_ARG : TDV_OK;
}
