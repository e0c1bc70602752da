// q3_formats.cpp -- native readers for the on-disk formats the reference deploys, so the libraries
// start from the reference's own files without Python:
//   .npy  v1/v2/v3, little-endian f4 / f8 (converted) / f2 / i4 / i8, C order
//         (the reference's C++ server parses f4/f8 v1/v2: dual_npu/code_predictor_cpp/npy_reader.h:22-108)
//   .npz  zip of .npy members, stored (np.savez: scripts/export_code_predictor_weights.py:76) or deflated
//         (np.savez_compressed: scripts/extract_embeddings.py:93), zip64 extras tolerated
//   .safetensors  8-byte header length + JSON table + raw tensors (F32 / F16 / BF16), zero-copy from the mapping
// and the key maps from the reference's names to the container names (weights.py):
//   code_predictor_weights.npz   layer_{i}_{part}, final_norm, codec_emb_{g}, lm_head_{g}
//                                (scripts/export_code_predictor_weights.py:51-74)
//   model.safetensors            HF keys talker.model.layers.*, talker.code_predictor.* (scripts/extract_embeddings.py:47-98)
//                                or the re-keyed Qwen3 talker model.layers.* (scripts/extract_talker_as_qwen3.py:53-71)
//   embeddings/*.npy             codec_embedding.npy, codec_head.npy (scripts/extract_embeddings.py:62-72)
#include "q3_common.h"

#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

namespace q3 {

namespace {

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// ---- .npy header -----------------------------------------------------------------------------
// Fills dtype / shape of `t` and returns the offset of the data, or 0 on error.  f8 is reported as
// dtype 100 (the caller converts).
size_t npy_header(const uint8_t* p, size_t n, PackTensor& t, std::string& err) {
    if (n < 10 || memcmp(p, "\x93NUMPY", 6) != 0) {
        err = "not an .npy file (bad magic)";
        return 0;
    }
    const int major = p[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = (size_t)p[8] | ((size_t)p[9] << 8);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        if (n < 12) {
            err = "truncated .npy header";
            return 0;
        }
        hlen = (size_t)p[8] | ((size_t)p[9] << 8) | ((size_t)p[10] << 16) | ((size_t)p[11] << 24);
        hoff = 12;
    } else {
        err = "unsupported .npy version " + std::to_string(major);
        return 0;
    }
    if (hoff + hlen > n) {
        err = "truncated .npy header";
        return 0;
    }
    const std::string h((const char*)p + hoff, hlen);
    auto value_after = [&](const char* key) -> size_t {
        size_t k = h.find(std::string("'") + key + "'");
        if (k == std::string::npos) k = h.find(std::string("\"") + key + "\"");
        if (k == std::string::npos) return std::string::npos;
        k = h.find(':', k);
        return k == std::string::npos ? k : k + 1;
    };
    size_t d = value_after("descr");
    if (d == std::string::npos) {
        err = ".npy header lacks descr";
        return 0;
    }
    size_t q0 = h.find_first_of("'\"", d);
    size_t q1 = q0 == std::string::npos ? q0 : h.find_first_of("'\"", q0 + 1);
    if (q1 == std::string::npos) {
        err = ".npy header: structured dtypes are not supported";
        return 0;
    }
    const std::string descr = h.substr(q0 + 1, q1 - q0 - 1);
    if (descr == "<f4" || descr == "=f4") t.dtype = F32;
    else if (descr == "<f2" || descr == "=f2") t.dtype = F16;
    else if (descr == "<i4" || descr == "=i4") t.dtype = I32;
    else if (descr == "<i8" || descr == "=i8") t.dtype = I64;
    else if (descr == "<f8" || descr == "=f8") t.dtype = 100;
    else {
        err = ".npy dtype " + descr + " is not supported (f4, f8, f2, i4, i8 little-endian are)";
        return 0;
    }
    size_t f = value_after("fortran_order");
    if (f != std::string::npos && h.compare(h.find_first_not_of(' ', f), 4, "True") == 0) {
        err = ".npy fortran_order=True is not supported";
        return 0;
    }
    size_t s = value_after("shape");
    size_t lp = s == std::string::npos ? s : h.find('(', s);
    size_t rp = lp == std::string::npos ? lp : h.find(')', lp);
    if (rp == std::string::npos) {
        err = ".npy header lacks shape";
        return 0;
    }
    t.ndim = 0;
    for (size_t i = lp + 1; i < rp;) {
        while (i < rp && (h[i] == ' ' || h[i] == ',')) i++;
        if (i >= rp) break;
        uint64_t v = 0;
        bool any = false;
        while (i < rp && h[i] >= '0' && h[i] <= '9') {
            v = v * 10 + (uint64_t)(h[i] - '0');
            i++;
            any = true;
        }
        if (!any || t.ndim >= 4) {
            err = ".npy shape: more than 4 dimensions or malformed";
            return 0;
        }
        t.shape[t.ndim++] = v;
    }
    return hoff + hlen;
}

// ---- minimal JSON (safetensors header) --------------------------------------------------------
struct JTensor {
    std::string dtype;
    std::vector<uint64_t> shape;
    uint64_t begin = 0, end = 0;
};
struct JParser {
    const char* p;
    const char* e;
    bool ok = true;
    void ws() {
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    bool eat(char c) {
        ws();
        if (p < e && *p == c) {
            p++;
            return true;
        }
        return false;
    }
    std::string str() {
        ws();
        std::string s;
        if (p >= e || *p != '"') {
            ok = false;
            return s;
        }
        p++;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) p++;   // names here carry no escapes worth decoding
            s.push_back(*p++);
        }
        if (p >= e) ok = false;
        else p++;
        return s;
    }
    uint64_t num() {
        ws();
        uint64_t v = 0;
        bool any = false;
        while (p < e && *p >= '0' && *p <= '9') {
            v = v * 10 + (uint64_t)(*p++ - '0');
            any = true;
        }
        if (!any) ok = false;
        return v;
    }
    void skip_value() {   // any JSON value (used for __metadata__)
        ws();
        if (p >= e) {
            ok = false;
            return;
        }
        if (*p == '"') {
            str();
        } else if (*p == '{' || *p == '[') {
            const char open = *p, close = open == '{' ? '}' : ']';
            p++;
            while (ok && !eat(close)) {
                if (open == '{') {
                    str();
                    if (!eat(':')) ok = false;
                }
                skip_value();
                eat(',');
                if (p >= e) ok = false;
            }
        } else {
            while (p < e && *p != ',' && *p != '}' && *p != ']') p++;
        }
    }
    bool tensor(JTensor& t) {
        if (!eat('{')) return false;
        while (ok && !eat('}')) {
            const std::string k = str();
            if (!eat(':')) return false;
            if (k == "dtype") {
                t.dtype = str();
            } else if (k == "shape") {
                if (!eat('[')) return false;
                while (ok && !eat(']')) {
                    t.shape.push_back(num());
                    eat(',');
                }
            } else if (k == "data_offsets") {
                if (!eat('[')) return false;
                t.begin = num();
                if (!eat(',')) return false;
                t.end = num();
                if (!eat(']')) return false;
            } else {
                skip_value();
            }
            eat(',');
        }
        return ok;
    }
};

// ---- name maps ---------------------------------------------------------------------------------
const char* const kLayerParts[] = {"input_ln", "q_proj", "k_proj", "v_proj", "o_proj", "q_norm",
                                   "k_norm",   "post_ln", "gate_proj", "up_proj", "down_proj"};
const char* const kHfParts[] = {"input_layernorm.weight",     "self_attn.q_proj.weight", "self_attn.k_proj.weight",
                                "self_attn.v_proj.weight",    "self_attn.o_proj.weight", "self_attn.q_norm.weight",
                                "self_attn.k_norm.weight",    "post_attention_layernorm.weight",
                                "mlp.gate_proj.weight",       "mlp.up_proj.weight",      "mlp.down_proj.weight"};

bool split_int(const std::string& s, size_t pos, int& v, size_t& next) {
    v = 0;
    size_t i = pos;
    while (i < s.size() && s[i] >= '0' && s[i] <= '9') v = v * 10 + (s[i++] - '0');
    next = i;
    return i > pos;
}

// "<prefix>{i}.<hf part>" -> "<stack>.layers.{i}.<part>"
std::string map_hf_layer(const std::string& k, const char* prefix, const char* stack) {
    const size_t n = strlen(prefix);
    if (k.compare(0, n, prefix) != 0) return "";
    int i;
    size_t nx;
    if (!split_int(k, n, i, nx) || nx >= k.size() || k[nx] != '.') return "";
    const std::string rest = k.substr(nx + 1);
    for (int p = 0; p < 11; p++)
        if (rest == kHfParts[p]) return std::string(stack) + ".layers." + std::to_string(i) + "." + kLayerParts[p];
    return "";
}

}  // namespace

// code_predictor_weights.npz member (without ".npy") -> container name
std::string map_cp_npz_key(const std::string& k) {
    int i;
    size_t nx;
    if (k.compare(0, 6, "layer_") == 0 && split_int(k, 6, i, nx) && nx < k.size() && k[nx] == '_') {
        const std::string part = k.substr(nx + 1);
        for (const char* p : kLayerParts)
            if (part == p) return "cp.layers." + std::to_string(i) + "." + part;
        return "";
    }
    if (k == "final_norm") return "cp.norm";
    if (k.compare(0, 10, "codec_emb_") == 0 && split_int(k, 10, i, nx) && nx == k.size()) return "cp.codec_emb." + std::to_string(i);
    if (k.compare(0, 8, "lm_head_") == 0 && split_int(k, 8, i, nx) && nx == k.size()) return "cp.lm_head." + std::to_string(i);
    return "";
}

// safetensors key (HF checkpoint, or the re-keyed Qwen3 talker) -> container name
std::string map_safetensors_key(const std::string& k) {
    std::string r = map_hf_layer(k, "talker.model.layers.", "talker");
    if (!r.empty()) return r;
    r = map_hf_layer(k, "model.layers.", "talker");
    if (!r.empty()) return r;
    r = map_hf_layer(k, "talker.code_predictor.model.layers.", "cp");
    if (!r.empty()) return r;
    if (k == "talker.model.norm.weight" || k == "model.norm.weight") return "talker.norm";
    if (k == "talker.model.codec_embedding.weight" || k == "model.embed_tokens.weight") return "talker.codec_embedding";
    if (k == "talker.codec_head.weight" || k == "lm_head.weight") return "talker.codec_head";
    if (k == "talker.code_predictor.model.norm.weight") return "cp.norm";
    int i;
    size_t nx;
    const char* e = "talker.code_predictor.model.codec_embedding.";
    if (k.compare(0, strlen(e), e) == 0 && split_int(k, strlen(e), i, nx) && k.substr(nx) == ".weight")
        return "cp.codec_emb." + std::to_string(i);
    const char* h = "talker.code_predictor.lm_head.";
    if (k.compare(0, strlen(h), h) == 0 && split_int(k, strlen(h), i, nx) && k.substr(nx) == ".weight")
        return "cp.lm_head." + std::to_string(i);
    if (k == "talker.model.text_embedding.weight") return "text.embedding";
    if (k == "talker.text_projection.linear_fc1.weight") return "text.fc1.weight";
    if (k == "talker.text_projection.linear_fc1.bias") return "text.fc1.bias";
    if (k == "talker.text_projection.linear_fc2.weight") return "text.fc2.weight";
    if (k == "talker.text_projection.linear_fc2.bias") return "text.fc2.bias";
    return "";
}

const uint8_t* Pack::map_file(const char* path, size_t* size) {
    int f = ::open(path, O_RDONLY);
    if (f < 0) return nullptr;
    struct stat st;
    if (fstat(f, &st) != 0 || st.st_size <= 0) {
        ::close(f);
        return nullptr;
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, f, 0);
    ::close(f);
    if (m == MAP_FAILED) return nullptr;
    extra_maps.push_back({(uint8_t*)m, (size_t)st.st_size});
    *size = (size_t)st.st_size;
    return (const uint8_t*)m;
}

bool Pack::add_npy_bytes(const uint8_t* p, size_t n, const std::string& name, const char* what) {
    PackTensor t;
    std::string err;
    const size_t off = npy_header(p, n, t, err);
    if (!off) {
        Q3_LOG("%s: %s", what, err.c_str());
        return false;
    }
    t.name = name;
    const uint64_t ne = t.numel();
    if (t.dtype == 100) {   // f8 -> f4, as the reference's reader does
        if (off + ne * 8 > n) {
            Q3_LOG("%s: truncated data", what);
            return false;
        }
        owned.emplace_back(ne * 4);
        float* dst = (float*)owned.back().data();
        for (uint64_t i = 0; i < ne; i++) {
            double v;
            memcpy(&v, p + off + i * 8, 8);
            dst[i] = (float)v;
        }
        t.dtype = F32;
        t.data = owned.back().data();
        t.nbytes = ne * 4;
    } else {
        static const uint64_t esz[4] = {4, 2, 4, 8};
        t.nbytes = ne * esz[t.dtype];
        if (off + t.nbytes > n) {
            Q3_LOG("%s: truncated data", what);
            return false;
        }
        t.data = p + off;
    }
    tensors[name] = t;
    return true;
}

bool Pack::add_npy(const char* path, const std::string& name) {
    size_t n = 0;
    const uint8_t* p = map_file(path, &n);
    if (!p) {
        Q3_LOG("cannot open %s", path);
        return false;
    }
    return add_npy_bytes(p, n, name, path);
}

// Zip: end-of-central-directory record -> central directory -> local headers.  Sizes come from the
// central directory (numpy writes zip64 extras into the local headers with force_zip64=True).
bool Pack::add_npz(const char* path, std::string (*rename)(const std::string&)) {
    size_t n = 0;
    const uint8_t* p = map_file(path, &n);
    if (!p || n < 22) {
        Q3_LOG("cannot open %s as a zip archive", path);
        return false;
    }
    auto u16 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    auto u64 = [&](size_t o) { return (uint64_t)u32(o) | ((uint64_t)u32(o + 4) << 32); };
    size_t eocd = std::string::npos;
    for (size_t i = n - 22;; i--) {
        if (u32(i) == 0x06054b50u) {
            eocd = i;
            break;
        }
        if (i == 0 || n - i > 22 + 65535) break;
    }
    if (eocd == std::string::npos) {
        Q3_LOG("%s: no zip end-of-central-directory record", path);
        return false;
    }
    uint64_t n_ent = u16(eocd + 10), cd_off = u32(eocd + 16);
    if ((n_ent == 0xffff || cd_off == 0xffffffffu) && eocd >= 20 && u32(eocd - 20) == 0x07064b50u) {
        const uint64_t z64 = u64(eocd - 20 + 8);   // zip64 EOCD locator -> zip64 EOCD record
        if (z64 + 56 <= n && u32(z64) == 0x06064b50u) {
            n_ent = u64(z64 + 32);
            cd_off = u64(z64 + 48);
        }
    }
    size_t c = cd_off;
    int added = 0;
    for (uint64_t e = 0; e < n_ent; e++) {
        if (c + 46 > n || u32(c) != 0x02014b50u) {
            Q3_LOG("%s: corrupt zip central directory", path);
            return false;
        }
        const uint32_t method = u16(c + 10), nlen = u16(c + 28), xlen = u16(c + 30), clen = u16(c + 32);
        uint64_t csize = u32(c + 20), usize = u32(c + 24), lho = u32(c + 42);
        const std::string member((const char*)p + c + 46, nlen);
        // zip64 extended information (header id 1): only the fields that overflowed, in this order
        for (size_t x = c + 46 + nlen; x + 4 <= c + 46 + nlen + xlen;) {
            const uint32_t id = u16(x), sz = u16(x + 2);
            if (id == 1) {
                size_t q = x + 4;
                if (usize == 0xffffffffu && q + 8 <= x + 4 + sz) { usize = u64(q); q += 8; }
                if (csize == 0xffffffffu && q + 8 <= x + 4 + sz) { csize = u64(q); q += 8; }
                if (lho == 0xffffffffu && q + 8 <= x + 4 + sz) { lho = u64(q); q += 8; }
            }
            x += 4 + sz;
        }
        c += 46 + nlen + xlen + clen;
        if (!ends_with(member, ".npy")) continue;
        const std::string name = rename(member.substr(0, member.size() - 4));
        if (name.empty()) continue;
        if (lho + 30 > n || u32(lho) != 0x04034b50u) {
            Q3_LOG("%s: corrupt local header of %s", path, member.c_str());
            return false;
        }
        const size_t data = lho + 30 + u16(lho + 26) + u16(lho + 28);
        if (data + csize > n) {
            Q3_LOG("%s: member %s runs past the end of the archive", path, member.c_str());
            return false;
        }
        const std::string what = std::string(path) + ":" + member;
        if (method == 0) {
            if (!add_npy_bytes(p + data, (size_t)csize, name, what.c_str())) return false;
        } else if (method == 8) {
            owned.emplace_back((size_t)usize);
            std::vector<uint8_t>& buf = owned.back();
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) return false;
            zs.next_in = (Bytef*)(p + data);
            zs.next_out = buf.data();
            uint64_t in_left = csize, out_left = usize;
            int rc = Z_OK;
            while (rc == Z_OK) {   // avail_* are 32-bit: feed in slices
                zs.avail_in = (uInt)(in_left > (1u << 30) ? (1u << 30) : in_left);
                zs.avail_out = (uInt)(out_left > (1u << 30) ? (1u << 30) : out_left);
                const uInt ai = zs.avail_in, ao = zs.avail_out;
                rc = inflate(&zs, Z_NO_FLUSH);
                in_left -= ai - zs.avail_in;
                out_left -= ao - zs.avail_out;
                if (rc == Z_OK && ai == zs.avail_in && ao == zs.avail_out) break;
            }
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || out_left != 0) {
                Q3_LOG("%s: inflate failed (%d)", what.c_str(), rc);
                return false;
            }
            if (!add_npy_bytes(buf.data(), buf.size(), name, what.c_str())) return false;
        } else {
            Q3_LOG("%s: zip method %u is not supported (stored and deflate are)", what.c_str(), method);
            return false;
        }
        added++;
    }
    return added > 0;
}

bool Pack::add_safetensors(const char* path, std::string (*rename)(const std::string&)) {
    size_t n = 0;
    const uint8_t* p = map_file(path, &n);
    if (!p || n < 8) {
        Q3_LOG("cannot open %s", path);
        return false;
    }
    uint64_t hlen;
    memcpy(&hlen, p, 8);
    if (hlen > n - 8 || hlen < 2) {
        Q3_LOG("%s: bad safetensors header length", path);
        return false;
    }
    JParser j{(const char*)p + 8, (const char*)p + 8 + hlen};
    const uint8_t* base = p + 8 + hlen;
    const uint64_t avail = n - 8 - hlen;
    if (!j.eat('{')) {
        Q3_LOG("%s: safetensors header is not a JSON object", path);
        return false;
    }
    int added = 0;
    while (j.ok && !j.eat('}')) {
        const std::string key = j.str();
        if (!j.eat(':')) {
            j.ok = false;
            break;
        }
        if (key == "__metadata__") {
            j.skip_value();
        } else {
            JTensor jt;
            if (!j.tensor(jt)) {
                j.ok = false;
                break;
            }
            // The code predictor's first op (scripts/export_code_predictor_onnx.py:38-41): talker hidden -> predictor
            // hidden.  Identity -- no tensors -- for the 0.6 B model whose two widths are both 1024
            // (export_code_predictor_weights.py:51-74 carries none); a checkpoint that DOES hold it is a model this
            // build has no op for, and dropping the tensor would compute something else silently.
            if (key.find("small_to_mtp_projection") != std::string::npos) {
                Q3_LOG("%s: tensor %s: this checkpoint has a talker -> code-predictor projection (small_to_mtp_projection); "
                       "only the identity form (Qwen3-TTS 0.6B: both widths 1024) is built -- refusing to ignore it", path, key.c_str());
                return false;
            }
            const std::string name = rename(key);
            if (!name.empty()) {
                PackTensor t;
                t.name = name;
                if (jt.dtype == "F32") t.dtype = F32;
                else if (jt.dtype == "F16") t.dtype = F16;
                else if (jt.dtype == "BF16") t.dtype = BF16;
                else {
                    Q3_LOG("%s: tensor %s has dtype %s (F32, F16, BF16 are supported)", path, key.c_str(), jt.dtype.c_str());
                    return false;
                }
                if (jt.shape.size() > 4 || jt.end < jt.begin || jt.end > avail) {
                    Q3_LOG("%s: tensor %s: bad shape / offsets", path, key.c_str());
                    return false;
                }
                t.ndim = (uint32_t)jt.shape.size();
                for (size_t d = 0; d < jt.shape.size(); d++) t.shape[d] = jt.shape[d];
                t.nbytes = jt.end - jt.begin;
                if (t.nbytes != t.numel() * (t.dtype == F32 ? 4 : 2)) {
                    Q3_LOG("%s: tensor %s: size does not match its shape", path, key.c_str());
                    return false;
                }
                t.data = base + jt.begin;
                tensors[name] = t;
                added++;
            }
        }
        j.eat(',');
    }
    if (!j.ok) {
        Q3_LOG("%s: malformed safetensors header", path);
        return false;
    }
    return added > 0;
}

// A Q3TTSW1 container, or the reference's deployed files: `path` may be a container file, a
// .safetensors / .npz / file, or a directory holding model.safetensors and/or
// code_predictor_weights.npz; `aux_dir` (the servers' --embeddings_dir) adds codec_embedding.npy /
// codec_head.npy and the text front-end's text_embedding / text_projection_* files.  Geometry (layer counts, vocabularies) is taken from what was found.
bool Pack::open_auto(const char* path, const char* aux_dir) {
    // the reference's embeddings/ directory (scripts/extract_embeddings.py:47-66; loaded by
    // llamacpp_talker_server.py:79-93): codec tables + the text front-end's table and projection MLP
    static const char* const kNpyTables[7][2] = {{"codec_embedding.npy", "talker.codec_embedding"},
                                                 {"codec_head.npy", "talker.codec_head"},
                                                 {"text_embedding.npy", "text.embedding"},
                                                 {"text_projection_linear_fc1_weight.npy", "text.fc1.weight"},
                                                 {"text_projection_linear_fc1_bias.npy", "text.fc1.bias"},
                                                 {"text_projection_linear_fc2_weight.npy", "text.fc2.weight"},
                                                 {"text_projection_linear_fc2_bias.npy", "text.fc2.bias"}};
    struct stat st;
    if (!path || stat(path, &st) != 0) {
        Q3_LOG("cannot open %s", path ? path : "(null)");
        return false;
    }
    auto try_file = [&](const std::string& f) -> int {   // 1 added, 0 absent, -1 error
        struct stat s2;
        if (stat(f.c_str(), &s2) != 0 || !S_ISREG(s2.st_mode)) return 0;
        if (ends_with(f, ".safetensors")) return add_safetensors(f.c_str(), map_safetensors_key) ? 1 : -1;
        if (ends_with(f, ".npz")) return add_npz(f.c_str(), map_cp_npz_key) ? 1 : -1;
        return 0;
    };
    auto try_npy = [&](const std::string& dir, const char* file, const char* name) -> int {
        const std::string f = dir + "/" + file;
        struct stat s2;
        if (tensors.count(name) || stat(f.c_str(), &s2) != 0) return 0;
        return add_npy(f.c_str(), name) ? 1 : -1;
    };
    int found = 0;
    if (S_ISREG(st.st_mode)) {
        const std::string f(path);
        if (!ends_with(f, ".safetensors") && !ends_with(f, ".npz")) return open(path);
        const int r = try_file(f);
        if (r < 0) return false;
        found += r;
    } else if (S_ISDIR(st.st_mode)) {
        const std::string d(path);
        for (const char* f : {"model.safetensors", "code_predictor_weights.npz"}) {
            const int r = try_file(d + "/" + f);
            if (r < 0) return false;
            found += r;
        }
        for (const auto& e : kNpyTables) {
            const int r = try_npy(d, e[0], e[1]);
            if (r < 0) return false;
            found += r;
        }
    }
    if (aux_dir && *aux_dir) {
        for (const auto& e : kNpyTables) {
            const int r = try_npy(aux_dir, e[0], e[1]);
            if (r < 0) return false;
            found += r;
        }
    }
    if (!found) {
        Q3_LOG("%s: no Q3TTSW1 container, model.safetensors or code_predictor_weights.npz found", path);
        return false;
    }
    // geometry from the tensors: layers present, vocabulary rows (the re-keyed talker pads both
    // tables to the text vocabulary: only the codec rows are used, scripts/extract_talker_as_qwen3.py:57-69)
    auto count_layers = [&](const char* stack) {
        int nl = 0;
        while (tensors.count(std::string(stack) + ".layers." + std::to_string(nl) + ".q_proj")) nl++;
        return nl;
    };
    const int tl = count_layers("talker"), cl = count_layers("cp");
    if (tl) meta["talker_layers"] = tl;
    if (cl) meta["cp_layers"] = cl;
    int groups = 0;
    while (tensors.count("cp.lm_head." + std::to_string(groups))) groups++;
    if (groups) meta["cp_groups"] = groups;
    const int talker_vocab = 3072;
    for (const char* nm : {"talker.codec_embedding", "talker.codec_head"}) {
        auto it = tensors.find(nm);
        if (it != tensors.end() && it->second.ndim == 2 && it->second.shape[0] > (uint64_t)talker_vocab) {
            PackTensor& t = it->second;
            t.nbytes = t.nbytes / t.shape[0] * talker_vocab;
            t.shape[0] = talker_vocab;
        }
    }
    if (const PackTensor* g = find("talker.layers.0.gate_proj")) meta["talker_ffn"] = (double)g->shape[0];
    if (const PackTensor* g = find("cp.layers.0.gate_proj")) meta["cp_ffn"] = (double)g->shape[0];
    return true;
}

}  // namespace q3
