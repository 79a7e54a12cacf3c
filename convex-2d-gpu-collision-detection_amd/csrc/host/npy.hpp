// npy.hpp — minimal NumPy .npy reader/writer for C-order little-endian float32 arrays.
//
// Stands in for the un-vendored llohse/libnpy the reference uses (load_numpy_array,
// reference utils.cu:217-224; npy::SaveArrayAsNumpy call sites generate_dataset.cu:303,
// :332, :351-352, :500 and compute_collision_probability.cu:355).  Written against
// the NPY format specification (versions 1.0 and 2.0), float32 only — the only dtype
// the reference ever reads or writes.
#pragma once

#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace npy {

struct Array {
    std::vector<size_t> shape;
    std::vector<float> data;
    size_t rows() const { return shape.empty() ? 0 : shape[0]; }
    size_t cols() const { return shape.size() < 2 ? 1 : shape[1]; }
};

inline void save_f32(const std::string& path, const std::vector<size_t>& shape, const float* data)
{
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (";
    size_t count = 1;
    for (size_t i = 0; i < shape.size(); i++) {
        dict += std::to_string(shape[i]);
        if (shape.size() == 1 || i + 1 < shape.size()) dict += ",";
        if (i + 1 < shape.size()) dict += " ";
        count *= shape[i];
    }
    dict += "), }";
    // magic(6) + version(2) + header_len(2) + dict + padding + '\n' must be a multiple of 64
    size_t unpadded = 10 + dict.size() + 1;
    size_t pad = (64 - unpadded % 64) % 64;
    dict.append(pad, ' ');
    dict.push_back('\n');
    if (dict.size() > 65535) throw std::runtime_error("npy: header too long");
    // The file appears under its name only when it is complete (written beside it, then renamed): a run that dies in the
    // middle — the drivers _exit() on a multi-GPU time-out while a table is still being saved — leaves no truncated
    // poses.npy / variances.npy / batch file that a later run with --pose_dir / --variance_dir would load.
    const std::string tmp = path + ".tmp." + std::to_string(static_cast<long long>(getpid()));
    std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("npy: cannot open for writing: " + tmp);
    const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
    f.write(reinterpret_cast<const char*>(magic), 8);
    const uint16_t hl = static_cast<uint16_t>(dict.size());
    const unsigned char hlb[2] = {static_cast<unsigned char>(hl & 0xff), static_cast<unsigned char>(hl >> 8)};
    f.write(reinterpret_cast<const char*>(hlb), 2);
    f.write(dict.data(), static_cast<std::streamsize>(dict.size()));
    f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(count * sizeof(float)));
    f.close();
    if (!f) { std::remove(tmp.c_str()); throw std::runtime_error("npy: write failed: " + path); }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); throw std::runtime_error("npy: cannot rename into place: " + path); }
}

inline std::string dict_value(const std::string& dict, const std::string& key)
{
    size_t k = dict.find("'" + key + "'");
    if (k == std::string::npos) throw std::runtime_error("npy: header key missing: " + key);
    size_t c = dict.find(':', k);
    if (c == std::string::npos) throw std::runtime_error("npy: malformed header");
    size_t b = dict.find_first_not_of(" ", c + 1);
    size_t e;
    if (dict[b] == '(') e = dict.find(')', b) + 1;
    else if (dict[b] == '\'') e = dict.find('\'', b + 1) + 1;
    else e = dict.find_first_of(",}", b);
    return dict.substr(b, e - b);
}

inline Array load_f32(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("npy: cannot open: " + path);
    unsigned char pre[10];
    f.read(reinterpret_cast<char*>(pre), 10);
    if (!f || pre[0] != 0x93 || std::memcmp(pre + 1, "NUMPY", 5) != 0) throw std::runtime_error("npy: bad magic: " + path);
    size_t hl;
    if (pre[6] == 1) {
        hl = pre[8] | (static_cast<size_t>(pre[9]) << 8);
    } else if (pre[6] == 2 || pre[6] == 3) {
        unsigned char more[2];
        f.read(reinterpret_cast<char*>(more), 2);
        hl = pre[8] | (static_cast<size_t>(pre[9]) << 8) | (static_cast<size_t>(more[0]) << 16) | (static_cast<size_t>(more[1]) << 24);
    } else {
        throw std::runtime_error("npy: unsupported version: " + path);
    }
    std::string dict(hl, '\0');
    f.read(&dict[0], static_cast<std::streamsize>(hl));
    if (!f) throw std::runtime_error("npy: truncated header: " + path);
    const std::string descr = dict_value(dict, "descr");
    if (descr != "'<f4'" && descr != "'=f4'") throw std::runtime_error("npy: only little-endian float32 is supported, got " + descr + ": " + path);
    if (dict_value(dict, "fortran_order").rfind("False", 0) != 0) throw std::runtime_error("npy: fortran_order arrays are not supported: " + path);
    const std::string shp = dict_value(dict, "shape");
    Array a;
    size_t count = 1;
    for (size_t i = 1; i < shp.size();) {
        while (i < shp.size() && (shp[i] == ' ' || shp[i] == ',')) i++;
        if (i >= shp.size() || shp[i] == ')') break;
        size_t j = i;
        while (j < shp.size() && shp[j] >= '0' && shp[j] <= '9') j++;
        if (j == i) throw std::runtime_error("npy: malformed shape: " + path);
        a.shape.push_back(std::stoull(shp.substr(i, j - i)));
        count *= a.shape.back();
        i = j;
    }
    a.data.resize(count);
    f.read(reinterpret_cast<char*>(a.data.data()), static_cast<std::streamsize>(count * sizeof(float)));
    if (static_cast<size_t>(f.gcount()) != count * sizeof(float)) throw std::runtime_error("npy: truncated data: " + path);
    return a;
}

}  // namespace npy
