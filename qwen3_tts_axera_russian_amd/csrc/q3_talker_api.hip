// q3_talker_api.hip -- the wrapper_* C ABI (include/qwen3tts_talker.h), i.e. the
// MI355X stand-in for dual_npu/llama_wrapper.c.
#include "../../include/qwen3tts_talker.h"
#include "q3_model.h"

using namespace q3;

namespace {

struct TalkerModel {
    Model* m = nullptr;
};

struct TalkerCtx {
    TalkerModel* tm = nullptr;
    int n_ctx = 0, n_batch = 0, n_slots = 1;
    KVCache kv;
    Work w;
    hipStream_t s = nullptr;
    int *d_slot = nullptr, *d_pos = nullptr;  // [n_batch]
    std::vector<int> n_used;                  // per slot: highest cached position + 1
    GraphExec g1;                             // single-row decode of slot/pos arrays
    int g1_rows = 0;
    float* h_pinned = nullptr;                // [hidden]
    int* i_pinned = nullptr;                  // [2]
};

const int kStateMagic = 0x564b3351;  // "Q3KV"

int attn_threads_for(int n_ctx) { return n_ctx <= 64 ? 256 : 1024; }

// rows already in w.h (f32) -> hidden of `out_rows` rows (row_map on device or null = identity)
int forward_rows(TalkerCtx* c, int R, const RowMap& rm) {
    const Model& m = *c->tm->m;
    if (launch_ssq_rows(c->s, c->w.rows_in, c->w.h, c->w.ssq, R, m.cfg.hidden, c->w.xh, m.talker.L[0].in_ln)) return -1;
    if (run_stack(c->s, m, m.talker, c->w, c->kv, R, rm, attn_threads_for(c->n_ctx))) return -1;
    return 0;
}

int final_hidden(TalkerCtx* c, int first_row, int R) {
    const Model& m = *c->tm->m;
    const int H = m.cfg.hidden;
    FinalNormArgs f;
    f.h = c->w.h;
    f.ssq = c->w.ssq;
    f.src_off = first_row;
    f.ssq_parts = H / 16;
    f.gamma = m.talker.final_norm;
    f.eps = m.cfg.eps;
    f.R = R;
    f.H = H;
    f.out_f32 = c->w.hidden_f32;
    f.out_f16 = c->w.hidden_f16;
    return launch_final_norm(c->s, f);
}

}  // namespace

extern "C" {

void wrapper_backend_init(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        Q3_LOG("wrapper_backend_init: no HIP device visible; model loading will fail (no CPU path)");
}

void wrapper_backend_free(void) {}

void* wrapper_load_model(const char* path, int n_gpu_layers) {
    (void)n_gpu_layers;
    if (!path) return nullptr;
    Model* m = model_load(path, true, false);
    if (!m) return nullptr;
    TalkerModel* tm = new TalkerModel();
    tm->m = m;
    return tm;
}

void wrapper_free_model(void* model) {
    TalkerModel* tm = (TalkerModel*)model;
    if (!tm) return;
    model_free(tm->m);
    delete tm;
}

int wrapper_model_n_embd(const void* model) {
    const TalkerModel* tm = (const TalkerModel*)model;
    return tm && tm->m ? tm->m->cfg.hidden : 0;
}

void* wrapper_create_context_slots(void* model, int n_ctx, int n_batch, int n_slots) {
    TalkerModel* tm = (TalkerModel*)model;
    if (!tm || !tm->m || n_ctx <= 0 || n_slots <= 0) return nullptr;
    if (n_batch <= 0) n_batch = n_ctx;
    if (n_ctx > tm->m->max_pos) {
        Q3_LOG("wrapper_create_context: n_ctx=%d exceeds the RoPE table (%d)", n_ctx, tm->m->max_pos);
        return nullptr;
    }
    TalkerCtx* c = new TalkerCtx();
    c->tm = tm;
    c->n_ctx = n_ctx;
    c->n_batch = n_batch > n_slots ? n_batch : n_slots;
    c->n_slots = n_slots;
    c->n_used.assign(n_slots, 0);
    const ModelCfg& cfg = tm->m->cfg;
    bool ok = hipStreamCreate(&c->s) == hipSuccess;
    ok = ok && kv_alloc(c->kv, cfg.talker_layers, n_slots, cfg.n_kv, n_ctx) == 0;
    ok = ok && work_alloc(c->w, cfg, c->n_batch, cfg.talker_ffn, cfg.talker_vocab) == 0;
    ok = ok && hipMalloc((void**)&c->d_slot, sizeof(int) * c->n_batch) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_pos, sizeof(int) * c->n_batch) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_pinned, sizeof(float) * cfg.hidden * 2, 0) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->i_pinned, sizeof(int) * 2, 0) == hipSuccess;
    if (!ok) {
        Q3_LOG("wrapper_create_context: allocation failed");
        wrapper_free_context(c);
        return nullptr;
    }
    return c;
}

void* wrapper_create_context(void* model, int n_ctx, int n_batch, int n_threads, int embeddings) {
    (void)n_threads;
    if (!embeddings) {
        Q3_LOG("wrapper_create_context: only embeddings mode exists in this library");
        return nullptr;
    }
    return wrapper_create_context_slots(model, n_ctx, n_batch, 1);
}

void wrapper_free_context(void* ctx) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c) return;
    if (c->s) hipStreamSynchronize(c->s);
    c->g1.reset();
    kv_free(c->kv);
    work_free(c->w);
    if (c->d_slot) hipFree(c->d_slot);
    if (c->d_pos) hipFree(c->d_pos);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->i_pinned) hipHostFree(c->i_pinned);
    if (c->s) hipStreamDestroy(c->s);
    delete c;
}

int wrapper_ctx_n_slots(void* ctx) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    return c ? c->n_slots : 0;
}

void wrapper_kv_clear(void* ctx) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c) return;
    // positions are overwritten before they are read again (causal, caller-supplied pos);
    // clearing only resets the bookkeeping used by the state file.
    c->n_used.assign(c->n_slots, 0);
}

int wrapper_decode_embd_slot(void* ctx, int slot, const float* embd, int n_tokens, int n_embd, int pos_start,
                             float* out_hidden) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c || !embd || !out_hidden) return -1;
    const Model& m = *c->tm->m;
    const int H = m.cfg.hidden;
    if (n_embd != H || n_tokens <= 0 || n_tokens > c->n_batch || slot < 0 || slot >= c->n_slots || pos_start < 0 ||
        pos_start + n_tokens > c->n_ctx) {
        Q3_LOG("wrapper_decode_embd: bad arguments (n_tokens=%d n_embd=%d pos_start=%d slot=%d; n_ctx=%d n_batch=%d)",
               n_tokens, n_embd, pos_start, slot, c->n_ctx, c->n_batch);
        return -1;
    }
    Q3_HIP(hipMemcpyAsync(c->w.rows_in, embd, sizeof(float) * (size_t)n_tokens * H, hipMemcpyHostToDevice, c->s), -1);
    if (n_tokens == 1) {
        // decode step: one captured graph, (slot,pos) read from device arrays
        c->i_pinned[0] = slot;  // the previous call synchronised the stream: the buffer is free
        c->i_pinned[1] = pos_start;
        Q3_HIP(hipMemcpyAsync(c->d_slot, &c->i_pinned[0], sizeof(int), hipMemcpyHostToDevice, c->s), -1);
        Q3_HIP(hipMemcpyAsync(c->d_pos, &c->i_pinned[1], sizeof(int), hipMemcpyHostToDevice, c->s), -1);
        RowMap rm;
        rm.slot = c->d_slot;
        rm.pos = c->d_pos;
        if (!c->g1.e || c->g1_rows != 1) {
            // eager once (sets kernel attributes), then capture
            if (forward_rows(c, 1, rm) || final_hidden(c, 0, 1)) return -1;
            Q3_HIP(hipStreamSynchronize(c->s), -1);
            c->g1.reset();
            Q3_HIP(hipStreamBeginCapture(c->s, hipStreamCaptureModeThreadLocal), -1);
            int rc = forward_rows(c, 1, rm) || final_hidden(c, 0, 1);
            hipError_t e = hipStreamEndCapture(c->s, &c->g1.g);
            if (rc || e != hipSuccess) {
                Q3_LOG("wrapper_decode_embd: graph capture failed");
                return -1;
            }
            Q3_HIP(hipGraphInstantiate(&c->g1.e, c->g1.g, nullptr, nullptr, 0), -1);
            c->g1_rows = 1;
            // the eager pass above already produced this step's result
        } else {
            Q3_HIP(hipGraphLaunch(c->g1.e, c->s), -1);
        }
    } else {
        RowMap rm;
        rm.slot_base = slot;
        rm.slot_stride = 0;
        rm.pos_base = pos_start;
        rm.pos_stride = 1;
        rm.same_slot_rows = true;
        if (forward_rows(c, n_tokens, rm)) return -1;
        if (final_hidden(c, n_tokens - 1, 1)) return -1;
    }
    Q3_HIP(hipMemcpyAsync(c->h_pinned, c->w.hidden_f32, sizeof(float) * H, hipMemcpyDeviceToHost, c->s), -1);
    Q3_HIP(hipStreamSynchronize(c->s), -1);
    memcpy(out_hidden, c->h_pinned, sizeof(float) * H);
    if (pos_start + n_tokens > c->n_used[slot]) c->n_used[slot] = pos_start + n_tokens;
    return 0;
}

int wrapper_decode_embd(void* ctx, const float* embd, int n_tokens, int n_embd, int pos_start, float* out_hidden) {
    return wrapper_decode_embd_slot(ctx, 0, embd, n_tokens, n_embd, pos_start, out_hidden);
}

int wrapper_decode_embd_batch(void* ctx, const float* embd, int n_rows, int n_embd, const int32_t* slot_ids,
                              const int32_t* pos, float* out_hidden) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c || !embd || !out_hidden || !slot_ids || !pos) return -1;
    const Model& m = *c->tm->m;
    const int H = m.cfg.hidden;
    if (n_embd != H || n_rows <= 0 || n_rows > c->n_batch || n_rows > c->n_slots) return -1;
    for (int r = 0; r < n_rows; r++) {
        if (slot_ids[r] < 0 || slot_ids[r] >= c->n_slots || pos[r] < 0 || pos[r] >= c->n_ctx) return -1;
        for (int q = 0; q < r; q++)
            if (slot_ids[q] == slot_ids[r]) return -1;  // rows must hit distinct sequences
    }
    Q3_HIP(hipMemcpyAsync(c->w.rows_in, embd, sizeof(float) * (size_t)n_rows * H, hipMemcpyHostToDevice, c->s), -1);
    Q3_HIP(hipMemcpyAsync(c->d_slot, slot_ids, sizeof(int) * n_rows, hipMemcpyHostToDevice, c->s), -1);
    Q3_HIP(hipMemcpyAsync(c->d_pos, pos, sizeof(int) * n_rows, hipMemcpyHostToDevice, c->s), -1);
    RowMap rm;
    rm.slot = c->d_slot;
    rm.pos = c->d_pos;
    if (forward_rows(c, n_rows, rm) || final_hidden(c, 0, n_rows)) return -1;
    Q3_HIP(hipMemcpyAsync(out_hidden, c->w.hidden_f32, sizeof(float) * (size_t)n_rows * H, hipMemcpyDeviceToHost, c->s),
           -1);
    Q3_HIP(hipStreamSynchronize(c->s), -1);
    for (int r = 0; r < n_rows; r++)
        if (pos[r] + 1 > c->n_used[slot_ids[r]]) c->n_used[slot_ids[r]] = pos[r] + 1;
    return 0;
}

int wrapper_codec_head(void* ctx, const float* hidden, int n_rows, float* logits_out) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c || !hidden || !logits_out || n_rows <= 0 || n_rows > c->n_batch) return -1;
    const Model& m = *c->tm->m;
    const int H = m.cfg.hidden, V = m.cfg.talker_vocab;
    std::vector<uint16_t> h16((size_t)((n_rows + 15) / 16 * 16) * H, 0);
    for (int r = 0; r < n_rows; r++)
        for (int k = 0; k < H; k++) h16[frag_idx_host(r, k, H)] = f2h_sat(hidden[(size_t)r * H + k]);
    Q3_HIP(hipMemcpyAsync(c->w.hidden_f16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice, c->s), -1);
    LinArgs a;
    a.wp = m.talker_head.wp;
    a.N = V;
    a.K = H;
    a.M = n_rows;
    a.nt = 1;
    a.x16 = c->w.hidden_f16;
    a.y = c->w.logits;
    a.ldy = V;
    if (launch_linear(c->s, a, PRO_F16, EPI_STORE)) return -1;
    Q3_HIP(hipMemcpyAsync(logits_out, c->w.logits, sizeof(float) * (size_t)n_rows * V, hipMemcpyDeviceToHost, c->s), -1);
    Q3_HIP(hipStreamSynchronize(c->s), -1);
    return V;
}

// ---- KV prefix cache files (slot 0) ---------------------------------------
// header: int32 magic, version, n_layers, n_kv, head_dim, n_pos; then per layer K[n_kv][n_pos][128] and
// V[n_kv][n_pos][128] (fp16).
size_t wrapper_state_get_size(void* ctx) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c) return 0;
    const size_t per = (size_t)c->kv.n_kv * c->n_used[0] * c->kv.head_dim * sizeof(half_t);
    return 6 * sizeof(int32_t) + 2 * per * c->kv.n_layers;
}

int wrapper_state_save_file(void* ctx, const char* path) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c || !path) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) {
        Q3_LOG("wrapper_state_save_file: failed to save to %s", path);
        return -1;
    }
    const int n_pos = c->n_used[0], D = c->kv.head_dim;
    int32_t hdr[6] = {kStateMagic, 1, c->kv.n_layers, c->kv.n_kv, D, n_pos};
    bool ok = fwrite(hdr, sizeof(hdr), 1, f) == 1;
    std::vector<uint16_t> buf((size_t)n_pos * D);
    hipStreamSynchronize(c->s);
    for (int l = 0; l < c->kv.n_layers && ok; l++)
        for (int kvsel = 0; kvsel < 2 && ok; kvsel++)
            for (int g = 0; g < c->kv.n_kv && ok; g++) {
                const half_t* base = (kvsel ? c->kv.v : c->kv.k) + l * c->kv.layer_stride() +
                                     ((size_t)0 * c->kv.n_kv + g) * (size_t)c->kv.n_ctx * D;
                if (n_pos > 0) {
                    ok = hipMemcpy(buf.data(), base, buf.size() * 2, hipMemcpyDeviceToHost) == hipSuccess;
                    ok = ok && fwrite(buf.data(), 2, buf.size(), f) == buf.size();
                }
            }
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        Q3_LOG("wrapper_state_save_file: failed to save to %s", path);
        return -1;
    }
    return 0;
}

int wrapper_state_load_file(void* ctx, const char* path) {
    TalkerCtx* c = (TalkerCtx*)ctx;
    if (!c || !path) return -1;
    FILE* f = fopen(path, "rb");
    if (!f) {
        Q3_LOG("wrapper_state_load_file: failed to load from %s", path);
        return -1;
    }
    int32_t hdr[6];
    bool ok = fread(hdr, sizeof(hdr), 1, f) == 1;
    const int D = c->kv.head_dim;
    ok = ok && hdr[0] == kStateMagic && hdr[1] == 1 && hdr[2] == c->kv.n_layers && hdr[3] == c->kv.n_kv &&
         hdr[4] == D && hdr[5] >= 0 && hdr[5] <= c->n_ctx;
    if (!ok) {
        fclose(f);
        Q3_LOG("wrapper_state_load_file: %s does not match this context", path);
        return -1;
    }
    const int n_pos = hdr[5];
    std::vector<uint16_t> buf((size_t)n_pos * D);
    hipStreamSynchronize(c->s);
    for (int l = 0; l < c->kv.n_layers && ok; l++)
        for (int kvsel = 0; kvsel < 2 && ok; kvsel++)
            for (int g = 0; g < c->kv.n_kv && ok; g++) {
                half_t* base = (kvsel ? c->kv.v : c->kv.k) + l * c->kv.layer_stride() +
                               ((size_t)0 * c->kv.n_kv + g) * (size_t)c->kv.n_ctx * D;
                if (n_pos > 0) {
                    ok = fread(buf.data(), 2, buf.size(), f) == buf.size();
                    ok = ok && hipMemcpy(base, buf.data(), buf.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
                }
            }
    fclose(f);
    if (!ok) {
        Q3_LOG("wrapper_state_load_file: failed to load from %s", path);
        return -1;
    }
    c->n_used[0] = n_pos;
    fprintf(stderr, "wrapper_state_load_file: loaded from %s, n_tokens=%d\n", path, n_pos);
    return 0;
}

}  // extern "C"
