// q3_engine.hip -- fused on-device frame loop for B utterances (include/qwen3tts_engine.h).
//
// One frame = [talker_sample -> 16 code-predictor positions (+15 heads/argmax) -> feedback sum ->
// talker step -> final norm -> codec head], captured once per batch size as a hipGraph.  All
// per-utterance state (positions, EOS bookkeeping, emitted codes) lives in device arrays so the
// same graph serves every frame.
#include "../../include/qwen3tts_engine.h"
#include "q3_cp.h"

#include <chrono>

using namespace q3;

namespace {

struct Engine {
    Model* m = nullptr;
    int max_batch = 0, n_ctx = 0, max_frames = 0;
    hipStream_t s = nullptr;
    KVCache kv_t, kv_c;
    Work wt, wc;  // talker / code-predictor activations
    int prefill_rows = 0;
    // device state
    int *d_slot = nullptr, *d_pos = nullptr;       // prefill row maps [prefill_rows]
    int* d_tiles = nullptr;                        // prefill attention tiles, int[4] each (<= prefill_rows of them)
    int *d_iota = nullptr;                         // [max_batch]
    int *d_past = nullptr, *d_npast = nullptr, *d_ntext = nullptr, *d_done = nullptr, *d_nframes = nullptr;
    int *d_pos0 = nullptr, *d_posdec = nullptr, *d_lastrow = nullptr;
    int* d_codes = nullptr;  // [max_frames][B][16]
    int* d_forced = nullptr; // [max_frames][B][16] teacher-forced ids (q3e_set_forced_codes), allocated on first use
    bool forced_on = false;
    unsigned long long* d_seed = nullptr;  // [max_batch] draw-stream seed of every slot (device array: graph-safe)
    unsigned long long n_requests = 0, req_seed = 0, n_refills = 0;
    float* d_pad = nullptr;
    // run state
    int B = 0, ignore_eos = 0, cap_frames = 0, frames_run = 0;
    int frames_hi = 0;    // frames any slot may have recorded since q3e_start (q3e_refill restarts frames_run, not this)
    GraphExec graph[8];   // one captured frame per chain, replayed on the chain's own stream (own HW queue)
    int graph_B = 0, graph_ignore = -1, graph_cap = -1, graph_chains = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_run_ms = 0.f, last_prefill_ms = 0.f, last_host_launch_ms = 0.f;
    int* h_done = nullptr;  // pinned [max_batch]
    // sampling (0 temperature = greedy, the reference's --temperature 0 limit)
    float t_temp = 0.f, t_top_p = 0.95f, c_temp = 0.f;
    int t_top_k = 50, c_top_k = 50;
    unsigned long long seed = 0;
    // independent row groups of one frame run as parallel branches of the graph (latency hiding)
    int n_chains = 1;
    hipStream_t cs[8] = {nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[8] = {nullptr};
};

int talker_tail(Engine* e, hipStream_t st, int row0, int R) {
    // final norm (+ CP seed copy) and codec head for rows row0..row0+R-1
    const Model& m = *e->m;
    const int H = m.cfg.hidden;
    FinalNormArgs f;
    f.h = e->wt.h;
    f.ssq = e->wt.ssq;
    f.ssq_parts = H / 16;
    f.gamma = m.talker.final_norm;
    f.eps = m.cfg.eps;
    f.R = R;
    f.row0 = row0;
    f.H = H;
    f.out_f32 = e->wt.hidden_f32;
    f.out_f16 = e->wt.hidden_f16;
    f.out_copy = e->wc.h;
    f.out_copy_ssq = e->wc.ssq;
    f.out_copy_xh = e->wc.xh;
    f.out_copy_gamma = m.cp.L[0].in_ln;
    f.out_copy_row_off = cp_seed_row0(e->wc, R, row0, e->B);   // where cp_frame expects its position-0 rows
    if (launch_final_norm(st, f)) return -1;
    LinArgs a;
    a.wp = m.talker_head.wp;
    a.N = m.cfg.talker_vocab;
    a.K = H;
    a.M = row0 + R;
    a.m_begin = row0;
    a.nt = 1;
    a.x16 = e->wt.hidden_f16;
    a.y = e->wt.logits;
    a.ldy = m.cfg.talker_vocab;
    return launch_linear(st, a, PRO_F16, EPI_STORE);
}

int frame_chain(Engine* e, hipStream_t st, int row0, int R) {
    const Model& m = *e->m;
    TalkerSampleArgs sa;
    sa.logits = e->wt.logits;
    sa.V = m.cfg.talker_vocab;
    sa.R = R;
    sa.row0 = row0;
    sa.R_total = e->B;
    sa.audio_vocab = m.cfg.cp_vocab;
    sa.eos = m.cfg.codec_eos;
    sa.past = e->d_past;
    sa.n_past = e->d_npast;
    sa.n_text = e->d_ntext;
    sa.done = e->d_done;
    sa.codes = e->d_codes;
    sa.n_frames = e->d_nframes;
    sa.frame_cap = e->max_frames;
    sa.pos0 = e->d_pos0;
    sa.pos = e->d_posdec;
    sa.ignore_eos = e->ignore_eos;
    sa.max_frames = e->cap_frames;
    sa.temperature = e->t_temp;
    sa.top_k = e->t_top_k;
    sa.top_p = e->t_top_p;
    sa.seed = e->seed;
    sa.seed_ptr = e->d_seed;
    sa.forced = e->forced_on ? e->d_forced : nullptr;
    if (launch_talker_sample(st, sa)) return -1;
    CpFrameIO io;
    io.codes = e->d_codes;
    io.n_frames = e->d_nframes;
    io.frame_cap = e->max_frames;
    io.fb_h = e->wt.h;
    io.fb_ssq = e->wt.ssq;
    io.fb_xh = e->wt.xh;
    io.fb_gamma = m.talker.L[0].in_ln;
    io.pad_embed = e->d_pad;
    io.temperature = e->c_temp;
    io.top_k = e->c_top_k;
    io.seed = e->seed ^ 0x5851F42D4C957F2Dull;
    io.seed_ptr = e->d_seed;   // (the group index separates the talker's and the code predictor's draws)
    io.forced = e->forced_on ? e->d_forced : nullptr;
    if (cp_frame(st, m, e->wc, e->kv_c, R, io, row0, e->B)) return -1;
    RowMap rm;
    rm.slot_base = 0;        // row r of the batch owns KV slot r (no table: one dependent load less in front of every attention)
    rm.slot_stride = 1;
    rm.pos = e->d_posdec;
    if (run_stack(st, m, m.talker, e->wt, e->kv_t, R, rm, 1024, row0)) return -1;
    return talker_tail(e, st, row0, R);
}

int n_chains_eff(const Engine* e) {
    // chains own contiguous row ranges that must start on a 16-row fragment block
    const int nc = e->n_chains < 1 ? 1 : e->n_chains;
    return (nc > 1 && e->B % (16 * nc) == 0) ? nc : 1;
}

void chain_rows(const Engine* e, int c, int& row0, int& R) {
    const int nc = n_chains_eff(e);
    row0 = 0;
    for (int i = 0; i < c; i++) row0 += e->B / nc + (i < e->B % nc ? 1 : 0);
    R = e->B / nc + (c < e->B % nc ? 1 : 0);
}

// one frame of every chain, eagerly, each on its own stream
int frame_eager(Engine* e) {
    const int nc = n_chains_eff(e);
    for (int c = 0; c < nc; c++) {
        int row0, R;
        chain_rows(e, c, row0, R);
        if (frame_chain(e, e->cs[c], row0, R)) return -1;
    }
    return 0;
}

int fork_chains(Engine* e) {   // chain streams start after everything queued on the main stream
    Q3_HIP(hipEventRecord(e->ev_fork, e->s), -1);
    for (int c = 0; c < n_chains_eff(e); c++) Q3_HIP(hipStreamWaitEvent(e->cs[c], e->ev_fork, 0), -1);
    return 0;
}
int join_chains(Engine* e) {   // the main stream continues after every chain stream
    for (int c = 0; c < n_chains_eff(e); c++) {
        Q3_HIP(hipEventRecord(e->ev_join[c], e->cs[c]), -1);
        Q3_HIP(hipStreamWaitEvent(e->s, e->ev_join[c], 0), -1);
    }
    return 0;
}

}  // namespace

extern "C" {

void q3e_free(void* ee) {
    Engine* e = (Engine*)ee;
    if (!e) return;
    if (e->s) hipStreamSynchronize(e->s);
    for (auto& g : e->graph) g.reset();
    kv_free(e->kv_t);
    kv_free(e->kv_c);
    work_free(e->wt);
    work_free(e->wc);
    void* ps[] = {e->d_tiles, e->d_slot, e->d_pos,  e->d_iota,   e->d_past,    e->d_npast, e->d_ntext, e->d_done,
                  e->d_nframes, e->d_pos0, e->d_posdec, e->d_lastrow, e->d_codes, e->d_pad, e->d_forced, e->d_seed};
    for (void* p : ps)
        if (p) hipFree(p);
    if (e->h_done) hipHostFree(e->h_done);
    for (int c = 0; c < 8; c++) {
        if (e->cs[c]) hipStreamDestroy(e->cs[c]);
        if (e->ev_join[c]) hipEventDestroy(e->ev_join[c]);
    }
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev0) hipEventDestroy(e->ev0);
    if (e->ev1) hipEventDestroy(e->ev1);
    if (e->s) hipStreamDestroy(e->s);
    if (e->m) model_free(e->m);
    delete e;
}

void* q3e_create(const char* weights, int max_batch, int n_ctx, int max_frames) {
    if (!weights || max_batch <= 0 || n_ctx <= 0 || max_frames <= 0) return nullptr;
    Model* m = model_load(weights, true, true);
    if (!m) return nullptr;
    if (n_ctx > m->max_pos) {
        Q3_LOG("q3e_create: n_ctx=%d exceeds the RoPE table (%d)", n_ctx, m->max_pos);
        model_free(m);
        return nullptr;
    }
    Engine* e = new Engine();
    e->m = m;
    e->max_batch = max_batch;
    e->n_ctx = n_ctx;
    e->max_frames = max_frames;
    const ModelCfg& c = m->cfg;
    e->prefill_rows = max_batch > 2048 ? max_batch : 2048;
    // the frame loop is a latency-bound dependent chain: its queues get the highest priority so that its
    // short kernels are placed ahead of throughput work (the vocoder stream asks for the lowest)
    int prio_lo = 0, prio_hi = 0;
    const bool use_prio = !(getenv("Q3_STREAM_PRIO") && atoi(getenv("Q3_STREAM_PRIO")) == 0) &&
                          hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_lo != prio_hi;
    auto mkstream = [&](hipStream_t* st) {
        return (use_prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_hi)
                         : hipStreamCreateWithFlags(st, hipStreamNonBlocking)) == hipSuccess;
    };
    bool ok = mkstream(&e->s);
    ok = ok && hipEventCreate(&e->ev0) == hipSuccess && hipEventCreate(&e->ev1) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int c = 0; c < 8 && ok; c++)
        ok = mkstream(&e->cs[c]) &&
             hipEventCreateWithFlags(&e->ev_join[c], hipEventDisableTiming) == hipSuccess;
    if (const char* nc = getenv("Q3_CHAINS")) e->n_chains = atoi(nc) < 1 ? 1 : atoi(nc) > 8 ? 8 : atoi(nc);
    else e->n_chains = 1;  // measured on MI355X/ROCm 7.2: graphs on separate streams do not overlap here
    ok = ok && kv_alloc(e->kv_t, c.talker_layers, max_batch, c.n_kv, n_ctx) == 0;
    ok = ok && kv_alloc(e->kv_c, c.cp_layers, max_batch, c.n_kv, c.cp_groups + 1) == 0;
    ok = ok && work_alloc(e->wt, c, e->prefill_rows, c.talker_ffn, c.talker_vocab) == 0;
    ok = ok && work_alloc(e->wc, c, 2 * ((max_batch + 15) / 16 * 16), c.cp_ffn, c.cp_vocab) == 0;   // two rows per utterance in the CP's first pass
    auto ialloc = [&](int** p, size_t n) { return hipMalloc((void**)p, sizeof(int) * n) == hipSuccess; };
    ok = ok && ialloc(&e->d_slot, e->prefill_rows) && ialloc(&e->d_pos, e->prefill_rows);
    ok = ok && ialloc(&e->d_tiles, (size_t)4 * e->prefill_rows);
    ok = ok && ialloc(&e->d_iota, max_batch) && ialloc(&e->d_past, (size_t)max_batch * 32);
    ok = ok && ialloc(&e->d_npast, max_batch) && ialloc(&e->d_ntext, max_batch) && ialloc(&e->d_done, max_batch);
    ok = ok && ialloc(&e->d_nframes, max_batch) && ialloc(&e->d_pos0, max_batch) && ialloc(&e->d_posdec, max_batch);
    ok = ok && ialloc(&e->d_lastrow, max_batch);
    ok = ok && ialloc(&e->d_codes, (size_t)max_frames * max_batch * 16);
    ok = ok && hipMalloc((void**)&e->d_pad, sizeof(float) * c.hidden) == hipSuccess;
    ok = ok && hipMalloc((void**)&e->d_seed, sizeof(unsigned long long) * max_batch) == hipSuccess;
    ok = ok && hipMemset(e->d_seed, 0, sizeof(unsigned long long) * max_batch) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&e->h_done, sizeof(int) * max_batch, 0) == hipSuccess;
    if (ok) {
        std::vector<int> iota(max_batch);
        for (int i = 0; i < max_batch; i++) iota[i] = i;
        ok = hipMemcpy(e->d_iota, iota.data(), sizeof(int) * max_batch, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemset(e->d_pad, 0, sizeof(float) * c.hidden) == hipSuccess;
    }
    if (!ok) {
        Q3_LOG("q3e_create: allocation failed");
        q3e_free(e);
        return nullptr;
    }
    return e;
}

int q3e_set_sampling(void* ee, float talker_temperature, int talker_top_k, float talker_top_p, float cp_temperature,
                     int cp_top_k, uint64_t seed) {
    Engine* e = (Engine*)ee;
    if (!e) return -1;
    e->t_temp = talker_temperature;
    e->t_top_k = talker_top_k;
    e->t_top_p = talker_top_p;
    e->c_temp = cp_temperature;
    e->c_top_k = cp_top_k;
    e->seed = seed;
    e->n_requests = 0;
    for (auto& g : e->graph) g.reset();  // the captured kernels carry the old parameters
    return 0;
}

int q3e_set_forced_codes(void* ee, const int32_t* forced, int n_frames) {
    Engine* e = (Engine*)ee;
    if (!e) return -1;
    const bool on = forced != nullptr && n_frames > 0;
    if (on) {
        if (e->B <= 0 || n_frames > e->max_frames) return -1;
        const size_t total = (size_t)e->max_frames * e->max_batch * 16;
        if (!e->d_forced) Q3_HIP(hipMalloc((void**)&e->d_forced, sizeof(int) * total), -1);
        Q3_HIP(hipMemset(e->d_forced, 0xff, sizeof(int) * total), -1);   // -1 = free-running
        Q3_HIP(hipMemcpy(e->d_forced, forced, sizeof(int) * 16 * (size_t)e->B * n_frames, hipMemcpyHostToDevice), -1);
    }
    if (on != e->forced_on) {
        e->forced_on = on;
        for (auto& g : e->graph) g.reset();  // the captured kernels carry the old pointer
    }
    return 0;
}

int q3e_set_chains(void* ee, int n) {
    Engine* e = (Engine*)ee;
    if (!e || n < 1 || n > 8) return -1;
    if (n != e->n_chains) {
        e->n_chains = n;
        for (auto& g : e->graph) g.reset();  // the captured frames cover the old row ranges
    }
    return 0;
}

int q3e_set_pad_embed(void* ee, const float* pad) {
    Engine* e = (Engine*)ee;
    if (!e || !pad) return -1;
    Q3_HIP(hipMemcpy(e->d_pad, pad, sizeof(float) * e->m->cfg.hidden, hipMemcpyHostToDevice), -1);
    return 0;
}

// Ragged prefill of n utterances into the KV slots / output rows ids[0..n) (prefix rows concatenated in that order):
// utterances are packed into passes of at most prefill_rows rows; every row carries its own (slot, position).  The
// hidden of each utterance's last row lands in row ids[u] of the post-norm buffers, one group at a time.
static int prefill_ids(Engine* e, int n, const int* ids, const float* prefix, const int32_t* n_rows, int B_total) {
    const Model& m = *e->m;
    const int H = m.cfg.hidden;
    size_t row_off = 0;
    int u0 = 0;
    std::vector<int> slot, pos, tiles, last(n);
    while (u0 < n) {
        int u1 = u0, rows = 0;
        while (u1 < n && rows + n_rows[u1] <= e->prefill_rows) rows += n_rows[u1++];
        slot.resize(rows);
        pos.resize(rows);
        int r = 0;
        tiles.clear();
        bool contiguous = true;
        for (int u = u0; u < u1; u++) {
            const int b = ids[u];
            if (u > u0 && b != ids[u - 1] + 1) contiguous = false;
            for (int i = 0; i < n_rows[u]; i += 16) {      // the utterance's rows as runs of <= 16 positions (attn_tile_kernel)
                const int nn = n_rows[u] - i < 16 ? n_rows[u] - i : 16;
                const int t4[4] = {r + i, nn, b, i};
                tiles.insert(tiles.end(), t4, t4 + 4);
            }
            for (int i = 0; i < n_rows[u]; i++, r++) {
                slot[r] = b;
                pos[r] = i;
            }
            last[u] = r - 1;
        }
        Q3_HIP(hipMemcpyAsync(e->wt.rows_in, prefix + row_off * H, sizeof(float) * (size_t)rows * H, hipMemcpyHostToDevice, e->s), -1);
        Q3_HIP(hipMemcpyAsync(e->d_slot, slot.data(), sizeof(int) * rows, hipMemcpyHostToDevice, e->s), -1);
        Q3_HIP(hipMemcpyAsync(e->d_pos, pos.data(), sizeof(int) * rows, hipMemcpyHostToDevice, e->s), -1);
        Q3_HIP(hipMemcpyAsync(e->d_tiles, tiles.data(), sizeof(int) * tiles.size(), hipMemcpyHostToDevice, e->s), -1);
        if (contiguous) {
            Q3_HIP(hipMemcpyAsync(e->d_lastrow + ids[u0], last.data() + u0, sizeof(int) * (u1 - u0), hipMemcpyHostToDevice, e->s), -1);
        } else {
            for (int u = u0; u < u1; u++)
                Q3_HIP(hipMemcpyAsync(e->d_lastrow + ids[u], last.data() + u, sizeof(int), hipMemcpyHostToDevice, e->s), -1);
        }
        if (launch_ssq_rows(e->s, e->wt.rows_in, e->wt.h, e->wt.ssq, rows, H, e->wt.xh, m.talker.L[0].in_ln)) return -1;
        RowMap rm;
        rm.slot = e->d_slot;
        rm.pos = e->d_pos;
        rm.same_slot_rows = true;
        rm.tiles = e->d_tiles;
        rm.n_tiles = (int)(tiles.size() / 4);
        if (run_stack(e->s, m, m.talker, e->wt, e->kv_t, rows, rm, 1024)) return -1;
        // final norm of the last rows of this group into rows ids[u0..u1) of the post-norm buffers
        for (int u = u0; u < u1; u += contiguous ? (u1 - u0) : 1) {
            FinalNormArgs f;
            f.h = e->wt.h;
            f.ssq = e->wt.ssq;
            f.ssq_parts = H / 16;
            f.gamma = m.talker.final_norm;
            f.eps = m.cfg.eps;
            f.R = contiguous ? u1 - u0 : 1;
            f.row0 = ids[u];   // output rows (fragment-ordered buffers are indexed, not offset)
            f.H = H;
            f.row_map = e->d_lastrow;
            f.out_f32 = e->wt.hidden_f32;
            f.out_f16 = e->wt.hidden_f16;
            f.out_copy = e->wc.h;
            f.out_copy_ssq = e->wc.ssq;
            f.out_copy_xh = e->wc.xh;
            f.out_copy_gamma = m.cp.L[0].in_ln;
            // the first frame's code predictor pass reads its position-0 rows where cp_frame expects them; with
            // parallel chains (rows split over several frame chains) that is the sequential layout, offset 0
            f.out_copy_row_off = n_chains_eff(e) == 1 ? cp_seed_row0(e->wc, B_total, 0, B_total) : 0;
            if (launch_final_norm(e->s, f)) return -1;
        }
        Q3_HIP(hipStreamSynchronize(e->s), -1);  // host staging vectors are reused by the next group
        row_off += rows;
        u0 = u1;
    }
    return 0;
}

// codec head over rows 0..B-1 of the post-norm hidden (rows of running utterances give the logits they already have)
static int head_all_rows(Engine* e) {
    const Model& m = *e->m;
    LinArgs a;
    a.wp = m.talker_head.wp;
    a.N = m.cfg.talker_vocab;
    a.K = m.cfg.hidden;
    a.M = e->B;
    a.nt = 1;
    a.x16 = e->wt.hidden_f16;
    a.y = e->wt.logits;
    a.ldy = m.cfg.talker_vocab;
    return launch_linear(e->s, a, PRO_F16, EPI_STORE);
}

int q3e_start(void* ee, int B, const float* prefix, const int32_t* n_rows, const int32_t* n_text, int ignore_eos,
              int max_frames) {
    Engine* e = (Engine*)ee;
    if (!e || !prefix || !n_rows || !n_text || B <= 0 || B > e->max_batch) return -1;
    const Model& m = *e->m;
    const int H = m.cfg.hidden;
    if (max_frames <= 0 || max_frames > e->max_frames) max_frames = e->max_frames;
    std::vector<int> pos0(B);
    for (int b = 0; b < B; b++) {
        if (n_rows[b] <= 0 || n_rows[b] > e->prefill_rows || n_rows[b] + max_frames > e->n_ctx) {
            Q3_LOG("q3e_start: utterance %d: %d prefix rows + %d frames do not fit n_ctx=%d", b, n_rows[b], max_frames,
                   e->n_ctx);
            return -1;
        }
        pos0[b] = n_rows[b];
    }
    e->B = B;
    e->ignore_eos = ignore_eos ? 1 : 0;
    e->cap_frames = max_frames;
    e->frames_run = 0;
    e->frames_hi = 0;
    Q3_HIP(hipMemsetAsync(e->d_npast, 0, sizeof(int) * B, e->s), -1);
    Q3_HIP(hipMemsetAsync(e->d_done, 0, sizeof(int) * B, e->s), -1);
    Q3_HIP(hipMemsetAsync(e->d_nframes, 0, sizeof(int) * B, e->s), -1);
    Q3_HIP(hipMemsetAsync(e->d_past, 0, sizeof(int) * 32 * B, e->s), -1);
    Q3_HIP(hipMemsetAsync(e->d_codes, 0xff, sizeof(int) * 16 * (size_t)B * e->max_frames, e->s), -1);
    {   // a fresh draw stream per request (the reference's generators advance from their seed across requests)
        unsigned long long z = e->seed + 0x9E3779B97F4A7C15ull * e->n_requests++;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        e->req_seed = e->n_requests == 1 ? e->seed : (z ^ (z >> 31));
        e->n_refills = 0;
        std::vector<unsigned long long> seeds(B, e->req_seed);   // (the row index separates the slots' draws)
        Q3_HIP(hipMemcpy(e->d_seed, seeds.data(), sizeof(unsigned long long) * B, hipMemcpyHostToDevice), -1);
    }
    if (e->forced_on) {   // forcing belongs to the batch it was set for
        e->forced_on = false;
        for (auto& g : e->graph) g.reset();
    }
    Q3_HIP(hipMemcpyAsync(e->d_ntext, n_text, sizeof(int) * B, hipMemcpyHostToDevice, e->s), -1);
    Q3_HIP(hipMemcpyAsync(e->d_pos0, pos0.data(), sizeof(int) * B, hipMemcpyHostToDevice, e->s), -1);
    Q3_HIP(hipMemcpyAsync(e->d_posdec, pos0.data(), sizeof(int) * B, hipMemcpyHostToDevice, e->s), -1);
    Q3_HIP(hipStreamSynchronize(e->s), -1);
    Q3_HIP(hipEventRecord(e->ev0, e->s), -1);
    {
        std::vector<int> ids(B);
        for (int b = 0; b < B; b++) ids[b] = b;
        if (prefill_ids(e, B, ids.data(), prefix, n_rows, B)) return -1;
    }
    if (head_all_rows(e)) return -1;
    Q3_HIP(hipEventRecord(e->ev1, e->s), -1);
    Q3_HIP(hipStreamSynchronize(e->s), -1);
    hipEventElapsedTime(&e->last_prefill_ms, e->ev0, e->ev1);
    return 0;
}

int q3e_run(void* ee, int n_frames) {
    Engine* e = (Engine*)ee;
    if (!e || e->B <= 0 || n_frames <= 0) return -1;
    // never step past the frames the batch was started for (nor past the codes array): a further step would
    // have no frame to record into
    const int room = (e->cap_frames < e->max_frames ? e->cap_frames : e->max_frames) - e->frames_run;
    if (room <= 0) return 0;
    if (n_frames > room) n_frames = room;
    int done_frames = 0;
    const int nc = n_chains_eff(e);
    Q3_HIP(hipEventRecord(e->ev0, e->s), -1);
    if (fork_chains(e)) return -1;
    // Q3_NO_GRAPH=1: eager launches (rocprofv3 --kernel-trace crashes on the graph replay)
    static const bool no_graph = getenv("Q3_NO_GRAPH") && atoi(getenv("Q3_NO_GRAPH")) != 0;
    const bool need_capture = !e->graph[0].e || e->graph_B != e->B || e->graph_ignore != e->ignore_eos ||
                              e->graph_cap != e->cap_frames || e->graph_chains != nc;
    if (no_graph) {
        for (; done_frames < n_frames; done_frames++)
            if (frame_eager(e)) return -1;
    } else if (need_capture) {
        // first frame eagerly (real work; also sets the kernels' LDS attributes), then capture per chain
        if (frame_eager(e)) return -1;
        done_frames++;
        for (int c = 0; c < nc; c++) Q3_HIP(hipStreamSynchronize(e->cs[c]), -1);
        for (int c = 0; c < nc; c++) {
            int row0, R;
            chain_rows(e, c, row0, R);
            e->graph[c].reset();
            Q3_HIP(hipStreamBeginCapture(e->cs[c], hipStreamCaptureModeRelaxed), -1);
            int rc = frame_chain(e, e->cs[c], row0, R);
            hipError_t er = hipStreamEndCapture(e->cs[c], &e->graph[c].g);
            if (rc || er != hipSuccess) {
                Q3_LOG("q3e_run: graph capture failed");
                return -1;
            }
            Q3_HIP(hipGraphInstantiate(&e->graph[c].e, e->graph[c].g, nullptr, nullptr, 0), -1);
        }
        e->graph_B = e->B;
        e->graph_ignore = e->ignore_eos;
        e->graph_cap = e->cap_frames;
        e->graph_chains = nc;
    }
    const int check_every = 16;
    const auto th0 = std::chrono::steady_clock::now();
    while (!no_graph && done_frames < n_frames) {
        int chunk = n_frames - done_frames;
        if (!e->ignore_eos && chunk > check_every) chunk = check_every;
        for (int i = 0; i < chunk; i++)
            for (int c = 0; c < nc; c++) Q3_HIP(hipGraphLaunch(e->graph[c].e, e->cs[c]), -1);
        done_frames += chunk;
        if (!e->ignore_eos && done_frames < n_frames) {
            if (join_chains(e)) return -1;
            Q3_HIP(hipMemcpyAsync(e->h_done, e->d_done, sizeof(int) * e->B, hipMemcpyDeviceToHost, e->s), -1);
            Q3_HIP(hipStreamSynchronize(e->s), -1);
            bool all = true;
            for (int b = 0; b < e->B; b++) all = all && e->h_done[b];
            if (all) break;
            if (fork_chains(e)) return -1;
        }
    }
    e->last_host_launch_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - th0).count();
    if (join_chains(e)) return -1;
    Q3_HIP(hipEventRecord(e->ev1, e->s), -1);
    Q3_HIP(hipStreamSynchronize(e->s), -1);
    hipEventElapsedTime(&e->last_run_ms, e->ev0, e->ev1);
    e->frames_run += done_frames;
    e->frames_hi += done_frames;
    return done_frames;
}

float q3e_last_run_ms(void* ee) { return ee ? ((Engine*)ee)->last_run_ms : -1.f; }
float q3e_last_host_launch_ms(void* ee) { return ee ? ((Engine*)ee)->last_host_launch_ms : -1.f; }
float q3e_last_prefill_ms(void* ee) { return ee ? ((Engine*)ee)->last_prefill_ms : -1.f; }

int q3e_get_codes(void* ee, int32_t* out, int max_out_frames, int32_t* n_frames_per_utt) {
    Engine* e = (Engine*)ee;
    if (!e || !out || e->B <= 0) return -1;
    int nf = e->frames_hi < e->max_frames ? e->frames_hi : e->max_frames;
    if (nf > max_out_frames) nf = max_out_frames;
    Q3_HIP(hipMemcpy(out, e->d_codes, sizeof(int) * 16 * (size_t)e->B * nf, hipMemcpyDeviceToHost), -1);
    if (n_frames_per_utt) {
        Q3_HIP(hipMemcpy(n_frames_per_utt, e->d_npast, sizeof(int) * e->B, hipMemcpyDeviceToHost), -1);
    }
    return nf;
}

int q3e_get_done(void* ee, int32_t* done, int32_t* frames) {
    Engine* e = (Engine*)ee;
    if (!e || !done || e->B <= 0) return -1;
    Q3_HIP(hipMemcpy(done, e->d_done, sizeof(int) * e->B, hipMemcpyDeviceToHost), -1);
    std::vector<int> np(e->B);
    Q3_HIP(hipMemcpy(np.data(), e->d_npast, sizeof(int) * e->B, hipMemcpyDeviceToHost), -1);
    // the device raises done[b] on the step AFTER the budget's last frame, a step q3e_run never takes: an utterance
    // that has emitted its whole budget has ended
    for (int b = 0; b < e->B; b++)
        if (np[b] >= e->cap_frames) done[b] = 1;
    if (frames) memcpy(frames, np.data(), sizeof(int) * e->B);
    return 0;
}

int q3e_refill(void* ee, int n, const int32_t* slots, const float* prefix, const int32_t* n_rows, const int32_t* n_text) {
    Engine* e = (Engine*)ee;
    if (!e || e->B <= 0 || n <= 0 || n > e->B || !slots || !prefix || !n_rows || !n_text) return -1;
    if (e->forced_on) {
        Q3_LOG("q3e_refill: a teacher-forced batch cannot be refilled");
        return -1;
    }
    std::vector<int> ids(slots, slots + n), seen(e->B, 0);
    for (int u = 0; u < n; u++) {
        const int b = ids[u];
        if (b < 0 || b >= e->B || seen[b]++) {
            Q3_LOG("q3e_refill: slot %d is out of range or listed twice (batch of %d)", b, e->B);
            return -1;
        }
        if (n_rows[u] <= 0 || n_rows[u] > e->prefill_rows || n_rows[u] + e->cap_frames > e->n_ctx) {
            Q3_LOG("q3e_refill: utterance %d: %d prefix rows + %d frames do not fit n_ctx=%d", u, n_rows[u], e->cap_frames, e->n_ctx);
            return -1;
        }
    }
    Q3_HIP(hipStreamSynchronize(e->s), -1);
    Q3_HIP(hipEventRecord(e->ev0, e->s), -1);
    // per-slot state back to "just started": counters, the emitted-token ring, the slot's column of the codes array
    for (int u = 0; u < n; u++) {
        const int b = ids[u];
        Q3_HIP(hipMemsetAsync(e->d_npast + b, 0, sizeof(int), e->s), -1);
        Q3_HIP(hipMemsetAsync(e->d_done + b, 0, sizeof(int), e->s), -1);
        Q3_HIP(hipMemsetAsync(e->d_nframes + b, 0, sizeof(int), e->s), -1);
        Q3_HIP(hipMemsetAsync(e->d_past + 32 * b, 0, sizeof(int) * 32, e->s), -1);
        Q3_HIP(hipMemset2DAsync(e->d_codes + 16 * (size_t)b, sizeof(int) * 16 * (size_t)e->B, 0xff, sizeof(int) * 16, e->max_frames, e->s), -1);
        Q3_HIP(hipMemcpyAsync(e->d_ntext + b, n_text + u, sizeof(int), hipMemcpyHostToDevice, e->s), -1);
        Q3_HIP(hipMemcpyAsync(e->d_pos0 + b, n_rows + u, sizeof(int), hipMemcpyHostToDevice, e->s), -1);
        Q3_HIP(hipMemcpyAsync(e->d_posdec + b, n_rows + u, sizeof(int), hipMemcpyHostToDevice, e->s), -1);
        // a fresh draw stream for the new occupant: the counters (frame, group) restart with the slot, so keeping the
        // request's seed would replay the previous occupant's uniforms
        unsigned long long z = e->req_seed + 0xD1B54A32D192ED03ull * ++e->n_refills;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        Q3_HIP(hipMemcpyAsync(e->d_seed + b, &z, sizeof(z), hipMemcpyHostToDevice, e->s), -1);   // (synchronised below: z is a local)
        Q3_HIP(hipStreamSynchronize(e->s), -1);
    }
    Q3_HIP(hipStreamSynchronize(e->s), -1);
    if (prefill_ids(e, n, ids.data(), prefix, n_rows, e->B)) return -1;
    if (head_all_rows(e)) return -1;
    Q3_HIP(hipEventRecord(e->ev1, e->s), -1);
    Q3_HIP(hipStreamSynchronize(e->s), -1);
    hipEventElapsedTime(&e->last_prefill_ms, e->ev0, e->ev1);
    e->frames_run = 0;   // the new utterances have their whole frame budget; running ones stop at theirs on the device
    return 0;
}

int q3e_get_hidden(void* ee, float* out) {
    Engine* e = (Engine*)ee;
    if (!e || !out || e->B <= 0) return -1;
    Q3_HIP(hipMemcpy(out, e->wt.hidden_f32, sizeof(float) * (size_t)e->B * e->m->cfg.hidden, hipMemcpyDeviceToHost), -1);
    return 0;
}

double q3e_step_weight_bytes(void* ee) {
    Engine* e = (Engine*)ee;
    if (!e) return 0.0;
    const ModelCfg& c = e->m->cfg;
    const double head = 2.0 * c.hidden;
    return (double)e->m->talker.weight_bytes + head * c.talker_vocab +
           (double)(c.cp_groups + 1) * (double)e->m->cp.weight_bytes + head * c.cp_vocab * c.cp_groups;
}

}  // extern "C"
