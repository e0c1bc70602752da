// q3_text_api.hip -- text projection + dual-stream prefix on the GPU (include/qwen3tts_text.h): the device form of
// Qwen3TTSTalkerServer._embed_text / _build_prefix (dual_npu/llamacpp_talker_server.py:115-161).
#include "../../include/qwen3tts_text.h"
#include "q3_model.h"

using namespace q3;

namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ size_t fidx(int m, int k, int K) {   // frag_idx of q3_kernels.hip
    return ((((size_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + (m & 15) + 16 * ((k >> 3) & 3)) << 3) + (k & 7);
}

// row r of the GEMM input (fp16, fragment order) = text table row ids[r]; an id outside the table gives zeros
__global__ void __launch_bounds__(256) text_gather_kernel(const half_t* __restrict__ table, int V, int TD,
                                                          const int* __restrict__ ids, half_t* __restrict__ x16) {
    const int r = blockIdx.x, t = ids[r];
    for (int k8 = threadIdx.x; k8 < TD / 8; k8 += 256) {
        h8v v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (t >= 0 && t < V) v = *(const h8v*)(table + (size_t)t * TD + k8 * 8);
        *(h8v*)(x16 + fidx(r, k8 * 8, TD)) = v;
    }
}
// accumulator seed of a biased linear layer: h[r][n] = bias[n] (f32, fragment order)
__global__ void __launch_bounds__(256) bias_rows_kernel(const float* __restrict__ bias, int N, float* __restrict__ h) {
    const int r = blockIdx.x;
    for (int n4 = threadIdx.x; n4 < N / 4; n4 += 256) *(float4*)(h + fidx(r, n4 * 4, N)) = *(const float4*)(bias + n4 * 4);
}
// x16 = fp16(silu(h)), same (fragment) order on both sides
__global__ void __launch_bounds__(256) silu_f16_kernel(const float* __restrict__ h, half_t* __restrict__ x16, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = h[i];
        const float s = v * (1.0f / (1.0f + __expf(-v)));
        x16[i] = (half_t)fminf(fmaxf(s, -65504.f), 65504.f);
    }
}
// out[j][:] = proj[src[j]][:] (+ codec[cod[j]][:] when cod[j] >= 0): rows of _embed_text / _build_prefix, row-major f32
__global__ void __launch_bounds__(256) prefix_rows_kernel(const float* __restrict__ proj, int H, const int* __restrict__ src,
                                                          const int* __restrict__ cod, const float* __restrict__ codec,
                                                          int codec_vocab, float* __restrict__ out) {
    const int j = blockIdx.x, s = src[j], c = cod[j];
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += 256) {
        float4 v = *(const float4*)(proj + fidx(s, k4 * 4, H));
        if (c >= 0 && c < codec_vocab) {
            const float4 e = *(const float4*)(codec + (size_t)c * H + k4 * 4);
            v.x += e.x;
            v.y += e.y;
            v.z += e.z;
            v.w += e.w;
        }
        *(float4*)(out + (size_t)j * H + k4 * 4) = v;
    }
}

struct TextFE {
    int V = 0, TD = 0, H = 0, codec_vocab = 0, max_rows = 0;
    int special[12] = {151644, 77091, 198, 151671, 151672, 151673, 2148, 2149, 2155, 2156, 2157, 0};
    half_t* table = nullptr;          // [V][TD] fp16
    DevLinear fc1, fc2;
    float *b1 = nullptr, *b2 = nullptr, *codec = nullptr;
    half_t *x16 = nullptr, *a16 = nullptr;        // gathered rows; silu(fc1) rows
    float *h1 = nullptr, *h2 = nullptr, *ssq = nullptr, *out = nullptr;
    int *d_ids = nullptr, *d_src = nullptr, *d_cod = nullptr;
    hipStream_t s = nullptr;
    std::vector<void*> allocs;
};

void* dal(TextFE* t, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    hipMemset(p, 0, bytes ? bytes : 16);
    t->allocs.push_back(p);
    return p;
}

// any float tensor [N][K] -> fp16 on the device, converted in slices (the text table is 1.24 GB as f32)
half_t* upload_f16(TextFE* t, const PackTensor* pt) {
    const size_t ne = pt->numel();
    half_t* d = (half_t*)dal(t, ne * 2);
    if (!d) return nullptr;
    const size_t slice = (size_t)8 << 20;
    std::vector<uint16_t> buf;
    for (size_t o = 0; o < ne; o += slice) {
        const size_t n = ne - o < slice ? ne - o : slice;
        const void* src = nullptr;
        if (pt->dtype == F16) src = (const uint16_t*)pt->data + o;
        else {
            buf.resize(n);
            if (pt->dtype == F32) {
                const float* f = (const float*)pt->data + o;
                for (size_t i = 0; i < n; i++) buf[i] = f2h_sat(f[i]);
            } else if (pt->dtype == BF16) {
                const uint16_t* b = (const uint16_t*)pt->data + o;
                for (size_t i = 0; i < n; i++) buf[i] = f2h_sat(bf16_to_f32(b[i]));
            } else return nullptr;
            src = buf.data();
        }
        if (hipMemcpy(d + o, src, n * 2, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    }
    return d;
}
float* upload_f32(TextFE* t, const PackTensor* pt) {
    const size_t ne = pt->numel();
    float* d = (float*)dal(t, ne * 4);
    if (!d) return nullptr;
    std::vector<float> tmp;
    const void* src = pt->data;
    if (pt->dtype != F32) {
        tmp.resize(ne);
        const uint16_t* u = (const uint16_t*)pt->data;
        for (size_t i = 0; i < ne; i++) tmp[i] = pt->dtype == BF16 ? bf16_to_f32(u[i]) : h2f(u[i]);
        src = tmp.data();
    }
    return hipMemcpy(d, src, ne * 4, hipMemcpyHostToDevice) == hipSuccess ? d : nullptr;
}
bool pack_linear(TextFE* t, const PackTensor* pt, DevLinear& L) {
    L.N = (int)pt->shape[0];
    L.K = (int)pt->shape[1];
    half_t* rowmajor = upload_f16(t, pt);
    L.wp = (half_t*)dal(t, (size_t)L.N * L.K * 2);
    if (!rowmajor || !L.wp) return false;
    if (launch_pack_linear(t->s, rowmajor, L.N, L.K, L.wp, 0, 1)) return false;
    return hipStreamSynchronize(t->s) == hipSuccess;
}

// rows [0, R) of d_ids through the projection MLP -> h2 (f32, fragment order, [R][H])
int project(TextFE* t, int R) {
    hipLaunchKernelGGL(text_gather_kernel, dim3(R), dim3(256), 0, t->s, t->table, t->V, t->TD, t->d_ids, t->x16);
    hipLaunchKernelGGL(bias_rows_kernel, dim3(R), dim3(256), 0, t->s, t->b1, t->fc1.N, t->h1);
    LinArgs a;
    a.wp = t->fc1.wp;
    a.N = t->fc1.N;
    a.K = t->fc1.K;
    a.M = R;
    a.x16 = t->x16;
    a.h_out = t->h1;              // h1 = bias1 + x . W1^T
    a.ssq_out = t->ssq;
    if (launch_linear(t->s, a, PRO_F16, EPI_RESID)) return -1;
    const int Rp = (R + 15) / 16 * 16;
    const size_t n1 = (size_t)Rp * t->fc1.N;
    hipLaunchKernelGGL(silu_f16_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, t->s, t->h1, t->a16, n1);
    hipLaunchKernelGGL(bias_rows_kernel, dim3(R), dim3(256), 0, t->s, t->b2, t->fc2.N, t->h2);
    a = LinArgs();
    a.wp = t->fc2.wp;
    a.N = t->fc2.N;
    a.K = t->fc2.K;
    a.M = R;
    a.x16 = t->a16;
    a.h_out = t->h2;
    a.ssq_out = t->ssq;
    if (launch_linear(t->s, a, PRO_F16, EPI_RESID)) return -1;
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

int emit(TextFE* t, const std::vector<int>& ids, const std::vector<int>& src, const std::vector<int>& cod, float* out) {
    const int R = (int)ids.size(), J = (int)src.size();
    if (R > t->max_rows || J > t->max_rows) return -1;
    for (int id : ids)
        if (id < 0 || id >= t->V) {
            Q3_LOG("text front-end: token id %d outside the table of %d rows", id, t->V);
            return -1;
        }
    Q3_HIP(hipMemcpyAsync(t->d_ids, ids.data(), sizeof(int) * R, hipMemcpyHostToDevice, t->s), -1);
    Q3_HIP(hipMemcpyAsync(t->d_src, src.data(), sizeof(int) * J, hipMemcpyHostToDevice, t->s), -1);
    Q3_HIP(hipMemcpyAsync(t->d_cod, cod.data(), sizeof(int) * J, hipMemcpyHostToDevice, t->s), -1);
    if (project(t, R)) return -1;
    hipLaunchKernelGGL(prefix_rows_kernel, dim3(J), dim3(256), 0, t->s, t->h2, t->H, t->d_src, t->d_cod, t->codec, t->codec_vocab,
                       t->out);
    Q3_HIP(hipGetLastError(), -1);
    Q3_HIP(hipMemcpyAsync(out, t->out, sizeof(float) * (size_t)J * t->H, hipMemcpyDeviceToHost, t->s), -1);
    Q3_HIP(hipStreamSynchronize(t->s), -1);
    return J;
}

}  // namespace

extern "C" {

void tfe_free(void* hh) {
    TextFE* t = (TextFE*)hh;
    if (!t) return;
    if (t->s) hipStreamSynchronize(t->s);
    for (void* p : t->allocs) hipFree(p);
    if (t->s) hipStreamDestroy(t->s);
    delete t;
}

void* tfe_load(const char* weights, const char* embeddings_dir, int max_tokens) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        Q3_LOG("no HIP device available -- this library has no CPU path");
        return nullptr;
    }
    if (!weights && !embeddings_dir) return nullptr;
    Pack p;
    if (!p.open_auto(weights ? weights : embeddings_dir, embeddings_dir)) return nullptr;
    const PackTensor *te = p.find("text.embedding"), *w1 = p.find("text.fc1.weight"), *c1 = p.find("text.fc1.bias"),
                     *w2 = p.find("text.fc2.weight"), *c2 = p.find("text.fc2.bias"), *ce = p.find("talker.codec_embedding");
    if (!te || !w1 || !c1 || !w2 || !c2 || !ce) {
        Q3_LOG("tfe_load: %s lacks text.embedding / text.fc{1,2}.{weight,bias} / talker.codec_embedding", weights ? weights : embeddings_dir);
        return nullptr;
    }
    TextFE* t = new TextFE();
    t->V = (int)te->shape[0];
    t->TD = (int)te->shape[1];
    t->H = (int)w2->shape[0];
    t->codec_vocab = (int)ce->shape[0];
    const int mid = (int)w1->shape[0];
    if (te->ndim != 2 || (int)w1->shape[1] != t->TD || (int)w2->shape[1] != mid || (int)ce->shape[1] != t->H ||
        (t->TD != 1024 && t->TD != 2048 && t->TD != 3072) || (mid != 1024 && mid != 2048 && mid != 3072) || t->H % 128 || mid % 128) {
        Q3_LOG("tfe_load: unsupported geometry table [%d][%d] -> %d -> %d (the GEMM kernels take K in {1024, 2048, 3072}, N %% 128 == 0)",
               t->V, t->TD, mid, t->H);
        delete t;
        return nullptr;
    }
    static const char* keys[12] = {"im_start", "assistant", "newline", "tts_pad", "tts_bos", "tts_eos", "codec_pad", "codec_bos",
                                   "codec_nothink", "codec_think_bos", "codec_think_eos", ""};
    for (int i = 0; i < 11; i++) t->special[i] = (int)p.get(keys[i], (double)t->special[i]);
    t->max_rows = (max_tokens > 0 ? max_tokens : 512) + 16;
    const size_t Rp = ((size_t)t->max_rows + 127) / 128 * 128;
    bool ok = hipStreamCreate(&t->s) == hipSuccess;
    ok = ok && (t->table = upload_f16(t, te)) && pack_linear(t, w1, t->fc1) && pack_linear(t, w2, t->fc2);
    ok = ok && (t->b1 = upload_f32(t, c1)) && (t->b2 = upload_f32(t, c2)) && (t->codec = upload_f32(t, ce));
    ok = ok && (t->x16 = (half_t*)dal(t, Rp * t->TD * 2)) && (t->a16 = (half_t*)dal(t, Rp * mid * 2));
    ok = ok && (t->h1 = (float*)dal(t, Rp * mid * 4)) && (t->h2 = (float*)dal(t, Rp * t->H * 4));
    ok = ok && (t->ssq = (float*)dal(t, Rp * (mid / 16) * 4)) && (t->out = (float*)dal(t, Rp * t->H * 4));
    ok = ok && (t->d_ids = (int*)dal(t, Rp * 4)) && (t->d_src = (int*)dal(t, Rp * 4)) && (t->d_cod = (int*)dal(t, Rp * 4));
    if (!ok) {
        Q3_LOG("tfe_load: upload failed");
        tfe_free(t);
        return nullptr;
    }
    return t;
}

int tfe_hidden_size(void* h) { return h ? ((TextFE*)h)->H : 0; }
int tfe_text_vocab(void* h) { return h ? ((TextFE*)h)->V : 0; }

int tfe_embed_text(void* hh, const int32_t* token_ids, int n, float* out) {
    TextFE* t = (TextFE*)hh;
    if (!t || !token_ids || !out || n <= 0) return -1;
    std::vector<int> ids(token_ids, token_ids + n), src(n), cod(n, -1);
    for (int i = 0; i < n; i++) src[i] = i;
    return emit(t, ids, src, cod, out) == n ? 0 : -1;
}

int tfe_build_prefix(void* hh, const int32_t* text_token_ids, int n, const int32_t* special, float* out) {
    TextFE* t = (TextFE*)hh;
    if (!t || !out || n < 0 || (n > 0 && !text_token_ids)) return -1;
    const int* sp = special ? special : t->special;
    // projected rows: 0-2 role tokens, 3 tts_pad, 4 tts_bos, 5 tts_eos, 6.. the text
    std::vector<int> ids = {sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]};
    ids.insert(ids.end(), text_token_ids, text_token_ids + n);
    std::vector<int> src, cod;
    auto row = [&](int s_, int c_) {
        src.push_back(s_);
        cod.push_back(c_);
    };
    for (int i = 0; i < 3; i++) row(i, -1);                         // role rows: text stream only (:132-138)
    row(3, sp[8]);                                                  // tts_pad + codec nothink / think_bos / think_eos
    row(3, sp[9]);
    row(3, sp[10]);
    row(4, sp[6]);                                                  // tts_bos + codec_pad
    for (int i = 0; i < n; i++) row(6 + i, sp[6]);                  // text + codec_pad
    row(5, sp[6]);                                                  // tts_eos + codec_pad
    row(3, sp[7]);                                                  // tts_pad + codec_bos
    return emit(t, ids, src, cod, out);
}

int tfe_tts_pad_embed(void* hh, float* out) {
    TextFE* t = (TextFE*)hh;
    if (!t || !out) return -1;
    const int32_t id = t->special[3];
    return tfe_embed_text(t, &id, 1, out);
}

}  // extern "C"
