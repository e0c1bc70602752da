// q3_cp_api.hip -- code-predictor frame on the device + the cp_* C ABI
// (include/qwen3tts_cp.h).  Loop semantics follow dual_npu/code_predictor_server.py:94-140
// (sequential prefill: positions 0 and 1 are two passes, the numerically exact form).
#include "../../include/qwen3tts_cp.h"
#include "q3_cp.h"

namespace q3 {

static bool two_row_enabled() {
    static const bool on = !(getenv("Q3_CP_TWO_ROW") && atoi(getenv("Q3_CP_TWO_ROW")) == 0);
    return on;
}

int cp_seed_row0(const Work& w, int R, int row0, int R_total) {
    if (R_total <= 0) R_total = R;
    const int R16 = (R + 15) / 16 * 16;
    const bool two = two_row_enabled() && row0 == 0 && R_total == R && 2 * R16 <= w.max_rows;
    return two ? R16 : 0;
}

int cp_frame(hipStream_t s, const Model& m, Work& w, KVCache& kv, int R, const CpFrameIO& io, int row0, int R_total) {
    const ModelCfg& c = m.cfg;
    if (R_total <= 0) R_total = R;
    const int H = c.hidden, G = c.cp_groups;
    const int seed0 = cp_seed_row0(w, R, row0, R_total);
    RowMap rm;
    rm.slot_base = 0;
    rm.slot_stride = 1;  // row r owns KV slot r
    rm.pos_stride = 0;
    if (seed0 > 0) {
        // positions 0 and 1 in ONE pass: rows [0, R) = position 1 (the TALKER codec embedding of code_0,
        // code_predictor_server.py:97-98), rows [R16, R16 + R) = position 0 (the talker hidden); position 1 attends
        // to both through the cache (prep, then attend).  What follows continues in rows [0, R).
        const int R16 = seed0;
        if (w.map_R16 != R16) {   // (first, eager call for this batch size; the captured replays find it ready)
            std::vector<int> slot(2 * R16), pos(2 * R16);
            for (int i = 0; i < 2 * R16; i++) {
                slot[i] = i % R16;
                pos[i] = i < R16 ? 1 : 0;
            }
            Q3_HIP(hipMemcpyAsync(w.map_slot, slot.data(), sizeof(int) * 2 * R16, hipMemcpyHostToDevice, s), -1);
            Q3_HIP(hipMemcpyAsync(w.map_pos, pos.data(), sizeof(int) * 2 * R16, hipMemcpyHostToDevice, s), -1);
            Q3_HIP(hipStreamSynchronize(s), -1);
            w.map_R16 = R16;
        }
        if (launch_gather_embed(s, m.talker_emb, c.talker_vocab, H, io.codes, 0, io.n_frames, io.frame_cap, 0, w.h, w.ssq, R,
                                row0, R_total, io.forced, w.xh, m.cp.L[0].in_ln))
            return -1;
        RowMap r2;
        r2.slot = w.map_slot;
        r2.pos = w.map_pos;
        r2.same_slot_rows = true;
        r2.valid_mod = R16;
        r2.valid_n = R;
        if (run_stack(s, m, m.cp, w, kv, R16 + R, r2, 256, 0)) return -1;
    } else {
        // position 0: the talker hidden (code_predictor_server.py:121-122)
        rm.pos_base = 0;
        if (run_stack(s, m, m.cp, w, kv, R, rm, 256, row0)) return -1;
        // position 1: TALKER codec embedding of code_0 (:97-98,123-124)
        if (launch_gather_embed(s, m.talker_emb, c.talker_vocab, H, io.codes, 0, io.n_frames, io.frame_cap, 0, w.h, w.ssq, R,
                                row0, R_total, io.forced, w.xh, m.cp.L[0].in_ln))
            return -1;
        rm.pos_base = 1;
        if (run_stack(s, m, m.cp, w, kv, R, rm, 256, row0)) return -1;
    }
    for (int g = 0; g < G; g++) {
        if (g > 0) {
            rm.pos_base = g + 1;
            if (run_stack(s, m, m.cp, w, kv, R, rm, 256, row0)) return -1;
        }
        // final RMSNorm folded into the head GEMV's prologue (code_predictor_server.py:129,136)
        LinArgs a;
        a.wp = m.cp_head[g].wp;
        a.N = c.cp_vocab;
        a.K = H;
        a.M = row0 + R;
        a.m_begin = row0;
        a.nt = 0;
        a.x16 = w.xh;          // fp16((h * final_norm) / 16) from the last layer's down projection
        a.ssq = w.ssq;
        a.ssq_parts = H / 16;
        a.eps = c.eps;
        a.y = w.logits;
        a.ldy = c.cp_vocab;
        if (launch_linear(s, a, PRO_NORM, EPI_STORE)) return -1;
        CpArgmaxArgs x;
        x.logits = w.logits;
        x.V = c.cp_vocab;
        x.R = R;
        x.row0 = row0;
        x.R_total = R_total;
        x.H = H;
        x.group = g;
        x.codes = io.codes;
        x.n_frames = io.n_frames;
        x.frame_cap = io.frame_cap;
        x.temperature = io.temperature;
        x.top_k = io.top_k;
        x.seed = io.seed;
        x.seed_ptr = io.seed_ptr;
        x.forced = io.forced;
        if (g + 1 < G) {
            x.next_table = m.cp_emb[g];  // group g+1 embeds token g with CP table g (:134)
            x.h_out = w.h;
            x.ssq_out = w.ssq;
            x.xh_out = w.xh;
            x.gamma_next = m.cp.L[0].in_ln;
        } else if (io.fb_h) {
            x.talker_emb = m.talker_emb;
            x.talker_vocab = c.talker_vocab;
            x.cp_tables = m.d_cp_emb_ptrs;
            x.pad_embed = io.pad_embed;
            x.n_groups = G;
            x.h_out = io.fb_h;
            x.ssq_out = io.fb_ssq;
            x.xh_out = io.fb_xh;
            x.gamma_next = io.fb_gamma;
        }
        if (launch_cp_argmax(s, x)) return -1;
    }
    return 0;
}

}  // namespace q3

using namespace q3;

namespace {

struct CpHandle {
    Model* m = nullptr;
    int max_batch = 1;
    KVCache kv;
    Work w;
    hipStream_t s = nullptr;
    int *d_codes = nullptr, *d_nframes = nullptr;  // [max_batch][16], [max_batch] (all ones)
    std::vector<int> h_codes;
    GraphExec graph;
    int graph_rows = 0;
    int step_next_pos = 0;
};

}  // namespace

extern "C" {

void* cp_load(const char* weights, const char* embeddings_dir, int max_batch) {
    // `weights`: a Q3TTSW1 container, or the reference's --model_dir (code_predictor_weights.npz, parsed
    // natively) with `embeddings_dir` = its --embeddings_dir (codec_embedding.npy): code_predictor_server.py:43-51
    if (!weights) return nullptr;
    if (max_batch <= 0) max_batch = 1;
    Model* m = model_load(weights, false, true, embeddings_dir);
    if (!m) return nullptr;
    CpHandle* h = new CpHandle();
    h->m = m;
    h->max_batch = max_batch;
    const ModelCfg& c = m->cfg;
    bool ok = hipStreamCreate(&h->s) == hipSuccess;
    ok = ok && kv_alloc(h->kv, c.cp_layers, max_batch, c.n_kv, c.cp_groups + 1) == 0;
    ok = ok && work_alloc(h->w, c, 2 * ((max_batch + 15) / 16 * 16), c.cp_ffn, c.cp_vocab) == 0;   // two rows per utterance in the first pass
    ok = ok && hipMalloc((void**)&h->d_codes, sizeof(int) * 16 * max_batch) == hipSuccess;
    ok = ok && hipMalloc((void**)&h->d_nframes, sizeof(int) * max_batch) == hipSuccess;
    if (ok) {
        std::vector<int> ones(max_batch, 1);
        ok = hipMemcpy(h->d_nframes, ones.data(), sizeof(int) * max_batch, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        Q3_LOG("cp_load: allocation failed");
        cp_free(h);
        return nullptr;
    }
    h->h_codes.resize((size_t)16 * max_batch);
    return h;
}

void cp_free(void* hh) {
    CpHandle* h = (CpHandle*)hh;
    if (!h) return;
    if (h->s) hipStreamSynchronize(h->s);
    h->graph.reset();
    kv_free(h->kv);
    work_free(h->w);
    if (h->d_codes) hipFree(h->d_codes);
    if (h->d_nframes) hipFree(h->d_nframes);
    if (h->s) hipStreamDestroy(h->s);
    model_free(h->m);
    delete h;
}

int cp_hidden_size(void* hh) {
    CpHandle* h = (CpHandle*)hh;
    return h ? h->m->cfg.hidden : 0;
}

int cp_predict_batch(void* hh, const float* hidden, const int32_t* code_0, int n_rows, float temperature, int top_k,
                     uint64_t seed, int32_t* out_codes) {
    CpHandle* h = (CpHandle*)hh;
    if (!h || !hidden || !code_0 || !out_codes || n_rows <= 0 || n_rows > h->max_batch) return -1;
    const bool stochastic = temperature > 1e-6f;
    const Model& m = *h->m;
    const int H = m.cfg.hidden, G = m.cfg.cp_groups, R = n_rows;
    for (int r = 0; r < R; r++) {
        h->h_codes[(size_t)r * 16] = code_0[r];
        for (int j = 1; j < 16; j++) h->h_codes[(size_t)r * 16 + j] = -1;
    }
    Q3_HIP(hipMemcpyAsync(h->d_codes, h->h_codes.data(), sizeof(int) * 16 * R, hipMemcpyHostToDevice, h->s), -1);
    Q3_HIP(hipMemcpyAsync(h->w.rows_in, hidden, sizeof(float) * (size_t)R * H, hipMemcpyHostToDevice, h->s), -1);
    CpFrameIO io;
    io.codes = h->d_codes;
    io.n_frames = h->d_nframes;
    io.frame_cap = 1;
    io.temperature = temperature;
    io.top_k = top_k;
    io.seed = seed;
    auto body = [&]() -> int {
        if (launch_ssq_rows(h->s, h->w.rows_in, h->w.h, h->w.ssq, R, H, h->w.xh, m.cp.L[0].in_ln,
                            cp_seed_row0(h->w, R)))
            return -1;
        return cp_frame(h->s, m, h->w, h->kv, R, io);
    };
    if (stochastic) {
        // sampling parameters and seed change per call: launch eagerly (the greedy path keeps its graph)
        if (body()) return -1;
    } else if (!h->graph.e || h->graph_rows != R) {
        if (body()) return -1;  // eager once (kernel attributes), result is valid
        Q3_HIP(hipStreamSynchronize(h->s), -1);
        h->graph.reset();
        Q3_HIP(hipStreamBeginCapture(h->s, hipStreamCaptureModeThreadLocal), -1);
        int rc = body();
        hipError_t e = hipStreamEndCapture(h->s, &h->graph.g);
        if (rc || e != hipSuccess) {
            Q3_LOG("cp_predict: graph capture failed");
            return -1;
        }
        Q3_HIP(hipGraphInstantiate(&h->graph.e, h->graph.g, nullptr, nullptr, 0), -1);
        h->graph_rows = R;
    } else {
        Q3_HIP(hipGraphLaunch(h->graph.e, h->s), -1);
    }
    Q3_HIP(hipMemcpyAsync(h->h_codes.data(), h->d_codes, sizeof(int) * 16 * R, hipMemcpyDeviceToHost, h->s), -1);
    Q3_HIP(hipStreamSynchronize(h->s), -1);
    for (int r = 0; r < R; r++)
        for (int g = 0; g < G; g++) out_codes[(size_t)r * G + g] = h->h_codes[(size_t)r * 16 + 1 + g];
    return 0;
}

int cp_predict(void* hh, const float* hidden, int32_t code_0, float temperature, int top_k, uint64_t seed,
               int32_t* out_codes) {
    return cp_predict_batch(hh, hidden, &code_0, 1, temperature, top_k, seed, out_codes);
}

int cp_step(void* hh, const float* embed, int position, float* out_hidden) {
    CpHandle* h = (CpHandle*)hh;
    if (!h || !embed || !out_hidden) return -1;
    const Model& m = *h->m;
    const int H = m.cfg.hidden;
    if (position < 0 || position > m.cfg.cp_groups) return -1;
    Q3_HIP(hipMemcpyAsync(h->w.rows_in, embed, sizeof(float) * H, hipMemcpyHostToDevice, h->s), -1);
    if (launch_ssq_rows(h->s, h->w.rows_in, h->w.h, h->w.ssq, 1, H, h->w.xh, m.cp.L[0].in_ln)) return -1;
    RowMap rm;
    rm.pos_base = position;
    if (run_stack(h->s, m, m.cp, h->w, h->kv, 1, rm, 256)) return -1;
    FinalNormArgs f;
    f.h = h->w.h;
    f.ssq = h->w.ssq;
    f.ssq_parts = H / 16;
    f.gamma = m.cp.final_norm;
    f.eps = m.cfg.eps;
    f.R = 1;
    f.H = H;
    f.out_f32 = h->w.hidden_f32;
    if (launch_final_norm(h->s, f)) return -1;
    Q3_HIP(hipMemcpyAsync(out_hidden, h->w.hidden_f32, sizeof(float) * H, hipMemcpyDeviceToHost, h->s), -1);
    Q3_HIP(hipStreamSynchronize(h->s), -1);
    return 0;
}

int cp_lm_head(void* hh, int group, const float* hidden, float* logits_out) {
    CpHandle* h = (CpHandle*)hh;
    if (!h || !hidden || !logits_out) return -1;
    const Model& m = *h->m;
    const int H = m.cfg.hidden, V = m.cfg.cp_vocab;
    if (group < 0 || group >= m.cfg.cp_groups) return -1;
    std::vector<uint16_t> h16((size_t)16 * H, 0);
    for (int i = 0; i < H; i++) h16[frag_idx_host(0, i, H)] = f2h_sat(hidden[i]);
    Q3_HIP(hipMemcpyAsync(h->w.hidden_f16, h16.data(), h16.size() * 2, hipMemcpyHostToDevice, h->s), -1);
    LinArgs a;
    a.wp = m.cp_head[group].wp;
    a.N = V;
    a.K = H;
    a.M = 1;
    a.x16 = h->w.hidden_f16;
    a.y = h->w.logits;
    a.ldy = V;
    if (launch_linear(h->s, a, PRO_F16, EPI_STORE)) return -1;
    Q3_HIP(hipMemcpyAsync(logits_out, h->w.logits, sizeof(float) * V, hipMemcpyDeviceToHost, h->s), -1);
    Q3_HIP(hipStreamSynchronize(h->s), -1);
    return V;
}

}  // extern "C"
