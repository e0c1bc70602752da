// q3_common.h -- shared host-side helpers for the MI355X Qwen3-TTS libraries.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define Q3_LOG(...)                              \
    do {                                         \
        fprintf(stderr, "[qwen3tts] " __VA_ARGS__); \
        fputc('\n', stderr);                     \
    } while (0)

// HIP call that makes the enclosing function return `ret` on failure.
#define Q3_HIP(call, ret)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            Q3_LOG("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return ret;                                                                \
        }                                                                              \
    } while (0)

namespace q3 {

typedef _Float16 half_t;

// ---------------------------------------------------------------------------
// Q3TTSW1 container (written by weights.py)
// ---------------------------------------------------------------------------
enum DType : uint32_t { F32 = 0, F16 = 1, I32 = 2, I64 = 3, BF16 = 4 };   // BF16: safetensors only (q3_formats.cpp)

struct PackTensor {
    std::string name;
    uint32_t dtype = 0, ndim = 0;
    uint64_t shape[4] = {0, 0, 0, 0};
    uint64_t offset = 0, nbytes = 0;
    const uint8_t* data = nullptr;  // into the mapping
    uint64_t numel() const {
        uint64_t n = 1;
        for (uint32_t i = 0; i < ndim; i++) n *= shape[i];
        return n;
    }
};

struct Pack {
    std::map<std::string, double> meta;
    std::map<std::string, PackTensor> tensors;
    uint8_t* map = nullptr;
    size_t map_size = 0;
    int fd = -1;

    bool open(const char* path);
    // the reference's deployed files instead of a container (q3_formats.cpp): .npy / .npz / .safetensors
    // parsed natively, names mapped to the container's; `aux_dir` = the servers' --embeddings_dir
    bool open_auto(const char* path, const char* aux_dir = nullptr);
    bool add_npy(const char* path, const std::string& name);
    bool add_npz(const char* path, std::string (*rename)(const std::string&));
    bool add_safetensors(const char* path, std::string (*rename)(const std::string&));
    bool add_npy_bytes(const uint8_t* p, size_t n, const std::string& name, const char* what);
    const uint8_t* map_file(const char* path, size_t* size);
    std::vector<std::pair<uint8_t*, size_t>> extra_maps;   // files mapped by the add_* readers
    std::vector<std::vector<uint8_t>> owned;                // inflated / converted arrays
    void close();
    ~Pack() { close(); }
    const PackTensor* find(const std::string& n) const {
        auto it = tensors.find(n);
        return it == tensors.end() ? nullptr : &it->second;
    }
    double get(const char* key, double dflt) const {
        auto it = meta.find(key);
        return it == meta.end() ? dflt : it->second;
    }
};

std::string map_cp_npz_key(const std::string& k);
std::string map_safetensors_key(const std::string& k);

inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// host fp16 <-> fp32 (round to nearest even, saturating to +-65504)
float h2f(uint16_t h);
uint16_t f2h_sat(float f);

struct ModelCfg {
    int hidden = 1024, head_dim = 128, n_heads = 16, n_kv = 8;
    int talker_layers = 28, talker_ffn = 3072, talker_vocab = 3072;
    int cp_layers = 5, cp_ffn = 3072, cp_vocab = 2048, cp_groups = 15;
    float eps = 1e-6f;
    double rope_theta = 1e6;
    int codec_eos = 2150;
    void from_pack(const Pack& p);
};

}  // namespace q3
