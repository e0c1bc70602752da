// q3_model.hip -- weight upload / repack, KV cache, workspaces, layer-stack runner.
#include "q3_model.h"

#include <cmath>

namespace q3 {

namespace {

struct Loader {
    Model* m;
    const Pack& p;
    hipStream_t s = nullptr;
    half_t* stage = nullptr;  // device staging for one row-major matrix
    size_t stage_elems = 0;
    std::vector<uint16_t> host16;
    bool ok = true;

    void* dalloc(size_t bytes) {
        void* d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) {
            Q3_LOG("hipMalloc(%zu) failed", bytes);
            ok = false;
            return nullptr;
        }
        m->allocs.push_back(d);
        m->device_bytes += bytes;
        return d;
    }
    const PackTensor* need(const std::string& n, uint32_t ndim, uint64_t d0, uint64_t d1 = 0) {
        const PackTensor* t = p.find(n);
        if (!t) {
            Q3_LOG("weight file lacks tensor %s", n.c_str());
            ok = false;
            return nullptr;
        }
        if (t->ndim != ndim || t->shape[0] != d0 || (ndim > 1 && t->shape[1] != d1) ||
            (t->dtype != F32 && t->dtype != F16 && t->dtype != BF16)) {
            Q3_LOG("tensor %s: unexpected shape/dtype", n.c_str());
            ok = false;
            return nullptr;
        }
        return t;
    }
    // any float tensor -> device f32
    float* up_f32(const std::string& n, uint32_t ndim, uint64_t d0, uint64_t d1 = 0) {
        const PackTensor* t = need(n, ndim, d0, d1);
        if (!t) return nullptr;
        size_t ne = t->numel();
        float* d = (float*)dalloc(ne * 4);
        if (!d) return nullptr;
        if (t->dtype == F32) {
            if (hipMemcpy(d, t->data, ne * 4, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        } else {
            std::vector<float> tmp(ne);
            const uint16_t* src = (const uint16_t*)t->data;
            if (t->dtype == BF16)
                for (size_t i = 0; i < ne; i++) tmp[i] = bf16_to_f32(src[i]);
            else
                for (size_t i = 0; i < ne; i++) tmp[i] = h2f(src[i]);
            if (hipMemcpy(d, tmp.data(), ne * 4, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        }
        return d;
    }
    // stage a [N][K] matrix as fp16 row-major on the device, then scatter its 16-row tiles
    bool pack_into(const std::string& n, int N, int K, half_t* dst, int tile_off, int tile_stride) {
        const PackTensor* t = need(n, 2, (uint64_t)N, (uint64_t)K);
        if (!t) return false;
        size_t ne = (size_t)N * K;
        if (ne > stage_elems) {
            if (stage) hipFree(stage);
            stage = nullptr;
            if (hipMalloc((void**)&stage, ne * 2) != hipSuccess) {
                ok = false;
                return false;
            }
            stage_elems = ne;
        }
        const void* src = t->data;
        if (t->dtype == F32) {
            host16.resize(ne);
            const float* f = (const float*)t->data;
            for (size_t i = 0; i < ne; i++) host16[i] = f2h_sat(f[i]);
            src = host16.data();
        } else if (t->dtype == BF16) {
            host16.resize(ne);
            const uint16_t* b = (const uint16_t*)t->data;
            for (size_t i = 0; i < ne; i++) host16[i] = f2h_sat(bf16_to_f32(b[i]));
            src = host16.data();
        }
        if (hipMemcpy(stage, src, ne * 2, hipMemcpyHostToDevice) != hipSuccess) {
            ok = false;
            return false;
        }
        if (launch_pack_linear(s, stage, N, K, dst, tile_off, tile_stride) != 0) {
            ok = false;
            return false;
        }
        if (hipStreamSynchronize(s) != hipSuccess) {
            ok = false;
            return false;
        }
        return true;
    }
    DevLinear lin_alloc(int N, int K) {
        DevLinear l;
        l.N = N;
        l.K = K;
        l.wp = (half_t*)dalloc((size_t)N * K * 2);
        return l;
    }
    bool load_stack(DevStack& st, const char* prefix, int n_layers, int ffn) {
        const ModelCfg& c = m->cfg;
        const int H = c.hidden, D = c.head_dim, NQ = c.n_heads * D, NKV = c.n_kv * D;
        st.ffn = ffn;
        st.L.resize(n_layers);
        for (int i = 0; i < n_layers && ok; i++) {
            DevLayer& L = st.L[i];
            std::string b = std::string(prefix) + ".layers." + std::to_string(i) + ".";
            L.in_ln = up_f32(b + "input_ln", 1, H);
            L.post_ln = up_f32(b + "post_ln", 1, H);
            L.q_norm = up_f32(b + "q_norm", 1, D);
            L.k_norm = up_f32(b + "k_norm", 1, D);
            L.qkv = lin_alloc(NQ + 2 * NKV, H);
            L.o = lin_alloc(H, NQ);
            L.gu = lin_alloc(2 * ffn, H);
            L.down = lin_alloc(H, ffn);
            if (!ok) break;
            pack_into(b + "q_proj", NQ, H, L.qkv.wp, 0, 1);
            pack_into(b + "k_proj", NKV, H, L.qkv.wp, NQ / 16, 1);
            pack_into(b + "v_proj", NKV, H, L.qkv.wp, (NQ + NKV) / 16, 1);
            pack_into(b + "o_proj", H, NQ, L.o.wp, 0, 1);
            pack_into(b + "gate_proj", ffn, H, L.gu.wp, 0, 2);  // tile 2i   = gate rows 16i..
            pack_into(b + "up_proj", ffn, H, L.gu.wp, 1, 2);    // tile 2i+1 = up rows 16i..
            pack_into(b + "down_proj", H, ffn, L.down.wp, 0, 1);
            st.weight_bytes += ((size_t)(NQ + 2 * NKV) * H + (size_t)H * NQ + (size_t)3 * ffn * H) * 2;
        }
        st.final_norm = up_f32(std::string(prefix) + ".norm", 1, H);
        return ok;
    }
};

}  // namespace

Model* model_load(const char* path, bool want_talker, bool want_cp, const char* aux_dir) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        Q3_LOG("no HIP device available -- this library has no CPU path");
        return nullptr;
    }
    Pack p;
    if (!p.open_auto(path, aux_dir)) return nullptr;
    Model* m = new Model();
    m->cfg.from_pack(p);
    const ModelCfg& c = m->cfg;
    if (c.head_dim != 128 || c.hidden != 1024 || c.n_heads != 2 * c.n_kv) {
        Q3_LOG("unsupported geometry: hidden=%d head_dim=%d heads=%d/%d (kernels are built for 1024/128, GQA 2)",
               c.hidden, c.head_dim, c.n_heads, c.n_kv);
        delete m;
        return nullptr;
    }
    hipGetDevice(&m->device);
    Loader L{m, p};
    if (hipStreamCreate(&L.s) != hipSuccess) {
        delete m;
        return nullptr;
    }
    const int H = c.hidden;
    // the talker codec table is needed by both stages (code_0 embedding)
    if (p.find("talker.codec_embedding")) m->talker_emb = L.up_f32("talker.codec_embedding", 2, c.talker_vocab, H);
    if (want_talker) {
        m->talker.nt = 1;
        L.load_stack(m->talker, "talker", c.talker_layers, c.talker_ffn);
        m->talker_head = L.lin_alloc(c.talker_vocab, H);
        if (L.ok) L.pack_into("talker.codec_head", c.talker_vocab, H, m->talker_head.wp, 0, 1);
        if (!m->talker_emb) {
            Q3_LOG("weight file lacks talker.codec_embedding");
            L.ok = false;
        }
        m->has_talker = L.ok;
    }
    if (want_cp && L.ok) {
        m->cp.nt = 0;
        L.load_stack(m->cp, "cp", c.cp_layers, c.cp_ffn);
        m->cp.tail_gamma = m->cp.final_norm;   // the group heads are GEMMs over the final-normed output
        m->cp_emb.resize(c.cp_groups);
        m->cp_head.resize(c.cp_groups);
        for (int g = 0; g < c.cp_groups && L.ok; g++) {
            m->cp_emb[g] = L.up_f32("cp.codec_emb." + std::to_string(g), 2, c.cp_vocab, H);
            m->cp_head[g] = L.lin_alloc(c.cp_vocab, H);
            if (L.ok) L.pack_into("cp.lm_head." + std::to_string(g), c.cp_vocab, H, m->cp_head[g].wp, 0, 1);
        }
        if (L.ok) {
            m->d_cp_emb_ptrs = (const float**)L.dalloc(sizeof(float*) * c.cp_groups);
            if (m->d_cp_emb_ptrs &&
                hipMemcpy((void*)m->d_cp_emb_ptrs, m->cp_emb.data(), sizeof(float*) * c.cp_groups,
                          hipMemcpyHostToDevice) != hipSuccess)
                L.ok = false;
        }
        if (!m->talker_emb) {
            Q3_LOG("weight file lacks talker.codec_embedding (code_0 embedding of the code predictor)");
            L.ok = false;
        }
        m->has_cp = L.ok;
    }
    // RoPE tables, float32 arithmetic like the HF rotary embedding (inv_freq = 1/theta^(2i/d))
    if (L.ok) {
        m->max_pos = 8192;
        const int half = c.head_dim / 2;
        std::vector<float> cs((size_t)m->max_pos * half), sn((size_t)m->max_pos * half);
        for (int i = 0; i < half; i++) {
            const float inv_freq = 1.0f / powf((float)c.rope_theta, (float)(2 * i) / (float)c.head_dim);
            for (int pos = 0; pos < m->max_pos; pos++) {
                const float ang = (float)pos * inv_freq;
                cs[(size_t)pos * half + i] = cosf(ang);
                sn[(size_t)pos * half + i] = sinf(ang);
            }
        }
        m->rope_cos = (float*)L.dalloc(cs.size() * 4);
        m->rope_sin = (float*)L.dalloc(sn.size() * 4);
        if (L.ok) {
            if (hipMemcpy(m->rope_cos, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) L.ok = false;
            if (hipMemcpy(m->rope_sin, sn.data(), sn.size() * 4, hipMemcpyHostToDevice) != hipSuccess) L.ok = false;
        }
    }
    if (L.stage) hipFree(L.stage);
    hipStreamDestroy(L.s);
    if (!L.ok) {
        model_free(m);
        return nullptr;
    }
    return m;
}

void model_free(Model* m) {
    if (!m) return;
    for (void* d : m->allocs) hipFree(d);
    delete m;
}

int kv_alloc(KVCache& kv, int n_layers, int n_slots, int n_kv, int n_ctx) {
    kv.n_layers = n_layers;
    kv.n_slots = n_slots;
    kv.n_kv = n_kv;
    kv.n_ctx = n_ctx;
    const size_t bytes = kv.layer_stride() * n_layers * sizeof(half_t);
    Q3_HIP(hipMalloc((void**)&kv.k, bytes), -1);
    Q3_HIP(hipMalloc((void**)&kv.v, bytes), -1);
    Q3_HIP(hipMemset(kv.k, 0, bytes), -1);
    Q3_HIP(hipMemset(kv.v, 0, bytes), -1);
    return 0;
}
void kv_free(KVCache& kv) {
    if (kv.k) hipFree(kv.k);
    if (kv.v) hipFree(kv.v);
    kv.k = kv.v = nullptr;
}

int work_alloc(Work& w, const ModelCfg& c, int max_rows, int ffn, int max_vocab) {
    max_rows = (max_rows + 127) / 128 * 128;   // the largest row tile (gemm_kernel: 128)
    w.max_rows = max_rows;
    w.hidden = c.hidden;
    const size_t R = (size_t)max_rows;
    const int qkv_ld = (c.n_heads + 2 * c.n_kv) * c.head_dim;
    Q3_HIP(hipMalloc((void**)&w.rows_in, R * c.hidden * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.h, R * c.hidden * 4), -1);
    Q3_HIP(hipMemset(w.h, 0, R * c.hidden * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.ssq, R * (c.hidden / 16) * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.xh, R * c.hidden * 2), -1);
    Q3_HIP(hipMemset(w.xh, 0, R * c.hidden * 2), -1);
    Q3_HIP(hipMalloc((void**)&w.qkv, R * qkv_ld * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.attn, R * c.n_heads * c.head_dim * 2), -1);
    Q3_HIP(hipMemset(w.attn, 0, R * c.n_heads * c.head_dim * 2), -1);
    Q3_HIP(hipMalloc((void**)&w.act, R * ffn * 2), -1);
    Q3_HIP(hipMemset(w.act, 0, R * ffn * 2), -1);
    Q3_HIP(hipMalloc((void**)&w.hidden_f32, R * c.hidden * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.hidden_f16, R * c.hidden * 2), -1);
    Q3_HIP(hipMemset(w.hidden_f16, 0, R * c.hidden * 2), -1);
    Q3_HIP(hipMalloc((void**)&w.logits, R * max_vocab * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.map_slot, R * 4), -1);
    Q3_HIP(hipMalloc((void**)&w.map_pos, R * 4), -1);
    w.map_R16 = 0;
    return 0;
}
void work_free(Work& w) {
    void* ps[] = {w.rows_in, w.h, w.ssq, w.xh, w.qkv, w.attn, w.act, w.hidden_f32, w.hidden_f16, w.logits, w.map_slot, w.map_pos};
    for (void* p : ps)
        if (p) hipFree(p);
    w = Work();
}

int run_stack(hipStream_t s, const Model& m, const DevStack& st, Work& w, KVCache& kv, int R, const RowMap& rm,
              int attn_threads, int row0) {
    const ModelCfg& c = m.cfg;
    const int H = c.hidden, D = c.head_dim;
    const int qkv_ld = (c.n_heads + 2 * c.n_kv) * D;
    if (row0 % 16) {
        Q3_LOG("run_stack: row0=%d must be a multiple of 16 (fragment-ordered activations)", row0);
        return -1;
    }
    if (row0 + R > w.max_rows) {
        Q3_LOG("run_stack: %d rows > workspace %d", R, w.max_rows);
        return -1;
    }
    for (size_t li = 0; li < st.L.size(); li++) {
        const DevLayer& L = st.L[li];
        LinArgs a;
        // q/k/v projections on the RMS-normed residual
        a = LinArgs();
        a.wp = L.qkv.wp;
        a.N = L.qkv.N;
        a.K = H;
        a.M = row0 + R;
        a.m_begin = row0;
        a.nt = st.nt;
        a.x16 = w.xh;          // fp16((h * in_ln) / 16), written by the producer of h
        a.ssq = w.ssq;
        a.ssq_parts = H / 16;
        a.eps = c.eps;
        a.y = w.qkv;
        a.ldy = qkv_ld;
        if (launch_linear(s, a, PRO_NORM, EPI_STORE)) return -1;
        // attention
        AttnArgs t;
        t.qkv = w.qkv;
        t.ld = qkv_ld;
        t.R = R;
        t.row0 = row0;
        t.q_norm = L.q_norm;
        t.k_norm = L.k_norm;
        t.eps = c.eps;
        t.rope_cos = m.rope_cos;
        t.rope_sin = m.rope_sin;
        t.slot = rm.slot;
        t.pos = rm.pos;
        t.slot_base = rm.slot_base;
        t.slot_stride = rm.slot_stride;
        t.pos_base = rm.pos_base;
        t.pos_stride = rm.pos_stride;
        t.kc = kv.k + li * kv.layer_stride();
        t.vc = kv.v + li * kv.layer_stride();
        t.n_ctx = kv.n_ctx;
        t.n_kv = c.n_kv;
        t.n_heads = c.n_heads;
        t.out = w.attn;
        t.scale = 1.0f / sqrtf((float)D);
        t.threads = attn_threads;
        t.valid_mod = rm.valid_mod;
        t.valid_n = rm.valid_n;
        if (rm.same_slot_rows && R > 1) {
            if (launch_attn(s, t, ATTN_PREP)) return -1;
            if (rm.tiles && rm.n_tiles > 0) {
                t.tiles = rm.tiles;
                t.n_tiles = rm.n_tiles;
            } else if (!rm.slot && !rm.pos && rm.slot_stride == 0 && rm.pos_stride == 1 && rm.valid_mod == 0) {
                t.tiles = nullptr;                  // one run of consecutive positions: tiled implicitly
                t.n_tiles = (R + 15) / 16;
            }
            if (launch_attn(s, t, ATTN_ATTEND)) return -1;
        } else {
            if (launch_attn(s, t, ATTN_FUSED)) return -1;
        }
        // output projection + residual
        a = LinArgs();
        a.wp = L.o.wp;
        a.N = H;
        a.K = L.o.K;
        a.M = row0 + R;
        a.m_begin = row0;
        a.nt = st.nt;
        a.x16 = w.attn;
        a.h_out = w.h;
        a.ssq_out = w.ssq;
        a.xh_out = w.xh;
        a.gamma = L.post_ln;   // consumer: this layer's gate/up
        if (launch_linear(s, a, PRO_F16, EPI_RESID)) return -1;
        // gate/up + SwiGLU
        a = LinArgs();
        a.wp = L.gu.wp;
        a.N = L.gu.N;
        a.K = H;
        a.M = row0 + R;
        a.m_begin = row0;
        a.nt = st.nt;
        a.x16 = w.xh;
        a.ssq = w.ssq;
        a.ssq_parts = H / 16;
        a.eps = c.eps;
        a.act = w.act;
        if (launch_linear(s, a, PRO_NORM, EPI_SWIGLU)) return -1;
        // down projection + residual
        a = LinArgs();
        a.wp = L.down.wp;
        a.N = H;
        a.K = L.down.K;
        a.M = row0 + R;
        a.m_begin = row0;
        a.nt = st.nt;
        a.x16 = w.act;
        a.h_out = w.h;
        a.ssq_out = w.ssq;
        // consumer: the next layer's q/k/v, or whatever GEMM reads the stack's output (st.tail_gamma)
        a.gamma = li + 1 < st.L.size() ? st.L[li + 1].in_ln : st.tail_gamma;
        a.xh_out = a.gamma ? w.xh : nullptr;
        if (launch_linear(s, a, PRO_F16, EPI_RESID)) return -1;
    }
    return 0;
}

void GraphExec::reset() {
    if (e) hipGraphExecDestroy(e);
    if (g) hipGraphDestroy(g);
    e = nullptr;
    g = nullptr;
}

}  // namespace q3
