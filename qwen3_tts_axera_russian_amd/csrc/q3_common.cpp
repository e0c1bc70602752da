// q3_common.cpp -- container reader and host fp16 helpers.
#include "q3_common.h"
#include <cstdlib>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace q3 {

namespace {
#pragma pack(push, 1)
struct FileHeader {
    char magic[8];
    uint32_t version, n_tensors, n_meta, reserved;
    uint64_t data_offset;
};
struct MetaRec {
    char key[48];
    double value;
};
struct TensorRec {
    char name[96];
    uint32_t dtype, ndim;
    uint64_t shape[4];
    uint64_t offset, nbytes;
};
#pragma pack(pop)
static_assert(sizeof(FileHeader) == 32, "header");
static_assert(sizeof(MetaRec) == 56, "meta");
static_assert(sizeof(TensorRec) == 152, "tensor");
}  // namespace

bool Pack::open(const char* path) {
    close();
    fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        Q3_LOG("cannot open weight file %s", path);
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(FileHeader)) {
        Q3_LOG("%s: too small for a Q3TTSW1 container", path);
        close();
        return false;
    }
    map_size = (size_t)st.st_size;
    void* m = mmap(nullptr, map_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) {
        Q3_LOG("%s: mmap failed", path);
        map = nullptr;
        close();
        return false;
    }
    map = (uint8_t*)m;
    const FileHeader* h = (const FileHeader*)map;
    if (memcmp(h->magic, "Q3TTSW1\0", 8) != 0 || h->version != 1) {
        Q3_LOG("%s: bad magic/version (not a Q3TTSW1 v1 container)", path);
        close();
        return false;
    }
    size_t need = sizeof(FileHeader) + (size_t)h->n_meta * sizeof(MetaRec) +
                  (size_t)h->n_tensors * sizeof(TensorRec);
    if (need > map_size) {
        Q3_LOG("%s: truncated tables", path);
        close();
        return false;
    }
    const MetaRec* mr = (const MetaRec*)(map + sizeof(FileHeader));
    for (uint32_t i = 0; i < h->n_meta; i++) {
        char key[49];
        memcpy(key, mr[i].key, 48);
        key[48] = 0;
        meta[key] = mr[i].value;
    }
    const TensorRec* tr = (const TensorRec*)(map + sizeof(FileHeader) + (size_t)h->n_meta * sizeof(MetaRec));
    for (uint32_t i = 0; i < h->n_tensors; i++) {
        char name[97];
        memcpy(name, tr[i].name, 96);
        name[96] = 0;
        PackTensor t;
        t.name = name;
        t.dtype = tr[i].dtype;
        t.ndim = tr[i].ndim;
        if (t.ndim > 4 || t.dtype > 3) {
            Q3_LOG("%s: tensor %s has bad ndim/dtype", path, name);
            close();
            return false;
        }
        for (int d = 0; d < 4; d++) t.shape[d] = tr[i].shape[d];
        t.offset = tr[i].offset;
        t.nbytes = tr[i].nbytes;
        static const uint64_t esz[4] = {4, 2, 4, 8};
        if (t.offset + t.nbytes > map_size || t.nbytes != t.numel() * esz[t.dtype]) {
            Q3_LOG("%s: tensor %s out of bounds / size mismatch", path, name);
            close();
            return false;
        }
        t.data = map + t.offset;
        tensors[t.name] = t;
    }
    return true;
}

void Pack::close() {
    for (auto& m : extra_maps) munmap(m.first, m.second);
    extra_maps.clear();
    owned.clear();
    if (map) munmap(map, map_size);
    map = nullptr;
    map_size = 0;
    if (fd >= 0) ::close(fd);
    fd = -1;
    meta.clear();
    tensors.clear();
}

float h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
    if (e == 0) {
        if (m == 0) {
            u = s;
        } else {
            int sh = 0;
            while (!(m & 0x400)) {
                m <<= 1;
                sh++;
            }
            m &= 0x3ff;
            u = s | ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
    } else if (e == 31) {
        u = s | 0x7f800000u | (m << 13);
    } else {
        u = s | ((e + 112) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

uint16_t f2h_sat(float f) {
    if (f != f) return 0x7e00;
    if (f > 65504.f) f = 65504.f;
    if (f < -65504.f) f = -65504.f;
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000;
    int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15;
    uint32_t m = u & 0x7fffff;
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t hm = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(s | hm);
    }
    uint32_t hm = m >> 13, rem = m & 0x1fff;
    uint32_t r = (uint32_t)(s | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000 || (rem == 0x1000 && (hm & 1))) r++;
    return (uint16_t)r;
}

void ModelCfg::from_pack(const Pack& p) {
    hidden = (int)p.get("hidden", hidden);
    head_dim = (int)p.get("head_dim", head_dim);
    n_heads = (int)p.get("n_heads", n_heads);
    n_kv = (int)p.get("n_kv_heads", n_kv);
    talker_layers = (int)p.get("talker_layers", talker_layers);
    talker_ffn = (int)p.get("talker_ffn", talker_ffn);
    talker_vocab = (int)p.get("talker_vocab", talker_vocab);
    cp_layers = (int)p.get("cp_layers", cp_layers);
    cp_ffn = (int)p.get("cp_ffn", cp_ffn);
    cp_vocab = (int)p.get("cp_vocab", cp_vocab);
    cp_groups = (int)p.get("cp_groups", cp_groups);
    eps = (float)p.get("rms_eps", eps);
    rope_theta = p.get("rope_theta", rope_theta);
    codec_eos = (int)p.get("codec_eos", codec_eos);
}

}  // namespace q3

// ---- HIP runtime configuration, applied when the library is loaded (before the runtime initialises: it reads its
// flags at the first HIP call of the process) ----
// GPU_MAX_HW_QUEUES=1: every stream of the process shares ONE hardware queue.  Measured on MI355X / ROCm 7.2
// ("one hardware queue"): the replayed frame graph is 8-9 % faster (2.69 -> 2.46 ms per frame at 32 rows,
// 2.49 -> 2.23 at one) -- the command processor has one queue to service between dependent nodes -- and nothing is lost:
// kernels of the frame loop and of the vocoder do not overlap on this chip anyway (the step time is the serial sum with
// any number of queues).  This is a PROCESS-WIDE policy of the HIP runtime, so it is the host program's to decide: the
// Python entry points (hiplib.load, the servers, bench.py) export the variable themselves before anything initialises
// HIP.  For a host that dlopens this library directly -- the reference's llama_cpp_bindings.py loading it as
// llama_wrapper.so, a C++ server -- the constructor below does it, says so on stderr once, and stands back when the user
// has chosen a value or exported Q3_KEEP_HW_QUEUES=1 (co-hosting another HIP engine that wants its own queues).
__attribute__((constructor)) static void q3_runtime_defaults() {
    if (getenv("GPU_MAX_HW_QUEUES")) return;                                    // the user's (or the entry point's) value wins
    if (const char* k = getenv("Q3_KEEP_HW_QUEUES"))
        if (atoi(k) != 0) return;
    setenv("GPU_MAX_HW_QUEUES", "1", 0);
    fprintf(stderr, "[qwen3tts] GPU_MAX_HW_QUEUES=1 set for this process (one HIP hardware queue: -9 %% per frame step; export "
                    "GPU_MAX_HW_QUEUES or Q3_KEEP_HW_QUEUES=1 to keep the runtime's default)\n");
}

// ---- device selection (include/qwen3tts_engine.h): one process per GPU ----
extern "C" int q3_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
// Compute units of the current device (256 on an MI355X); 0 when there is none.
extern "C" int q3_device_compute_units(void) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return p.multiProcessorCount;
}
// Select the HIP device used by every handle created afterwards on this thread.
extern "C" int q3_set_device(int dev) { return hipSetDevice(dev) == hipSuccess ? 0 : -1; }
