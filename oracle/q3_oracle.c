/*
 * q3_oracle.c -- CPU restatement of the Qwen3-TTS talker / code-predictor arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (qwen3_tts_axera_russian_amd/, include/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED at the third-party boundary: the reference keeps this arithmetic in
 * llama.cpp (talker, called at dual_npu/llama_wrapper.c:125-163) and onnxruntime (code
 * predictor, dual_npu/code_predictor_server.py:77-85); neither library, nor any weight file or
 * golden vector for them, exists in /root/reference.  What is restated here
 * is the published Qwen3 decoder layer with the dimensions the reference fixes
 * (scripts/extract_talker_as_qwen3.py:89-110), the weight inventory of
 * scripts/export_code_predictor_weights.py:51-74, the graph order of
 * scripts/export_code_predictor_onnx.py:40-46 and the loop of
 * dual_npu/code_predictor_server.py:94-140.  The front-end logic that CAN be pinned
 * (prefix, sampling, feedback, chunking) lives in oracle/frontend.py and is checked against
 * vectors produced by the reference's own Python (tests/golden/).
 *
 * Numerics contract (same as the HIP kernels, DESIGN.md "Numerics"): projection weights are
 * fp16 values (passed here already widened to f32), every GEMM input is rounded to fp16
 * (saturating), accumulation and everything else is f32; K/V are stored rounded to fp16; RMSNorm is
 * folded around the GEMM it feeds (input fp16((h*gamma)/16), results times 16*inv_rms).
 * What pins it: tests/test_oracle_vs_hf.py compares this file with transformers' Qwen3Model (fp32) on
 * seeded weights -- 1e-6 in exact mode, 1e-3 under the rounding contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const float *in_ln, *q, *k, *v, *o, *q_norm, *k_norm, *post_ln, *gate, *up, *down;
} orc_layer;

typedef struct {
    int hidden, head_dim, n_heads, n_kv, ffn, n_layers;
    float eps;
    const orc_layer* layers;
    const float* final_norm;
    const float* rope_cos; /* [max_pos][head_dim/2] */
    const float* rope_sin;
} orc_stack;

/* ---- fp16 rounding (round-to-nearest-even, saturating at +-65504) ---- */
static float h2f(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            int sh = 0;
            while (!(m & 0x400)) { m <<= 1; sh++; }
            m &= 0x3ff;
            u = s | ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f2h_sat(float f) {
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
        uint32_t shift = (uint32_t)(14 - e), hm = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(s | hm);
    }
    uint32_t hm = m >> 13, rem = m & 0x1fff;
    uint32_t r = (uint32_t)(s | ((uint32_t)e << 10) | hm);
    if (rem > 0x1000 || (rem == 0x1000 && (hm & 1))) r++;
    return (uint16_t)r;
}
float orc_round_f16(float x) { return h2f(f2h_sat(x)); }
void orc_round_f16_array(const float* x, float* y, long n) {
    for (long i = 0; i < n; i++) y[i] = orc_round_f16(x[i]);
}
/* Exact mode: activations and K/V are NOT rounded to fp16 (weights keep the values they were given).
 * This is the plain fp32 Qwen3 layer, the form that tests/golden/make_hf_golden.py compares with
 * transformers' Qwen3Model; the default (rounded) mode is the device's numerics contract. */
static int g_exact = 0;
void orc_set_exact(int on) { g_exact = on; }
static float act_round(float x) { return g_exact ? x : orc_round_f16(x); }

/* RoPE tables exactly as the HIP library builds them on the host (q3_model.hip): float32
 * arithmetic like the HF rotary embedding. */
void orc_rope_tables(double theta, int head_dim, int max_pos, float* cs, float* sn) {
    const int half = head_dim / 2;
    for (int i = 0; i < half; i++) {
        const float inv_freq = 1.0f / powf((float)theta, (float)(2 * i) / (float)head_dim);
        for (int pos = 0; pos < max_pos; pos++) {
            const float ang = (float)pos * inv_freq;
            cs[(size_t)pos * half + i] = cosf(ang);
            sn[(size_t)pos * half + i] = sinf(ang);
        }
    }
}

/* Dot product in f32 with eight interleaved partial sums (lane j takes k = j mod 8), combined pairwise at the
 * end: a fixed order the compiler can keep in one 8-wide register.  K is a multiple of 8 everywhere here. */
static inline float dot8(const float* w, const float* x, int K) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 8)
        for (int j = 0; j < 8; j++) a[j] += w[k + j] * x[k + j];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

/* Y[t][n] = sum_k W[n][k] X[t][k] for T rows (ldx / ldy = row strides), f32 accumulate.  Weight rows are walked
 * once and reused for every token of a block while they sit in cache. */
static void matmul_rows(const float* W, const float* X, int ldx, float* Y, int ldy, int T, int N, int K) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        const float* w = W + (size_t)n * K;
        for (int t = 0; t < T; t++) Y[(size_t)t * ldy + n] = dot8(w, X + (size_t)t * ldx, K);
    }
}
static void matvec(const float* W, const float* x, float* y, int N, int K) { matmul_rows(W, x, K, y, N, 1, N, K); }

/* RMSNorm folded around the GEMM (the device's numerics contract): the GEMM input is
 * x16 = fp16((h * gamma) / 16) -- the norm weight applied, a fixed power-of-two pre-scale -- and the GEMM's f32
 * results are multiplied by the returned inv_rms * 16.  In exact mode (no rounding) this IS RMSNorm followed by
 * the projection, up to f32 round-off. */
#define NORM_PRE 0.0625f
#define NORM_POST 16.0f
static float rmsnorm_fold(const float* h, const float* gamma, float eps, int H, float* x16) {
    double ss = 0.0;
    for (int k = 0; k < H; k++) ss += (double)h[k] * h[k];
    for (int k = 0; k < H; k++) x16[k] = act_round((h[k] * gamma[k]) * NORM_PRE);
    return (1.0f / sqrtf((float)ss / (float)H + eps)) * NORM_POST;
}
static void scale_rows(float* Y, int ld, int T, int N, const float* post) {
    for (int t = 0; t < T; t++)
        for (int n = 0; n < N; n++) Y[(size_t)t * ld + n] *= post[t];
}

static void head_norm_rope(float* x, const float* gamma, float eps, int D, const float* cs, const float* sn) {
    float ss = 0.f;
    for (int i = 0; i < D; i++) ss += x[i] * x[i];
    const float inv = 1.0f / sqrtf(ss / (float)D + eps);
    const int half = D / 2;
    for (int i = 0; i < D; i++) x[i] = (x[i] * inv) * gamma[i];
    for (int i = 0; i < half; i++) { /* rotate-half pairs (i, i+half) */
        const float x0 = x[i], x1 = x[i + half];
        x[i] = x0 * cs[i] - x1 * sn[i];
        x[i + half] = x1 * cs[i] + x0 * sn[i];
    }
}

/* One row of a layer-major pass: the cache it appends to / attends over, and its position there. */
typedef struct {
    float *kc, *vc; /* [n_layers][n_kv][n_ctx][D] holding fp16-rounded values */
    int n_ctx, pos;
} orc_row;

/* T rows through every layer, layer by layer (each row's arithmetic is the same as walking it alone; only the
 * loop order differs, so weights are read once per layer for all rows).  Rows may belong to one sequence
 * (prefill: consecutive positions of one cache; a row attends to the rows before it, whose K/V of this layer are
 * appended first) or to different sequences (a batch of decode steps, every row with its own cache).
 * h (in/out): residual streams [T][H]. */
static int rows_forward(const orc_stack* st, const orc_row* rows, int T, float* h) {
    const int H = st->hidden, D = st->head_dim, NH = st->n_heads, NKV = st->n_kv, F = st->ffn;
    const int rep = NH / NKV, QD = NH * D, KD = NKV * D;
    const size_t per_tok = (size_t)H + QD + 2 * KD + QD + 2 * (size_t)F + H;
    float* buf = (float*)malloc(sizeof(float) * (per_tok * T));
    if (!buf) return -1;
    float* x16 = buf;                        /* [T][H]  */
    float* q = x16 + (size_t)T * H;          /* [T][QD] */
    float* kn = q + (size_t)T * QD;          /* [T][KD] */
    float* vn = kn + (size_t)T * KD;         /* [T][KD] */
    float* att = vn + (size_t)T * KD;        /* [T][QD] */
    float* g = att + (size_t)T * QD;         /* [T][F]  */
    float* u = g + (size_t)T * F;            /* [T][F]  */
    float* y = u + (size_t)T * F;            /* [T][H]  */
    float* post = (float*)malloc(sizeof(float) * T);
    if (!post) return -1;
    const float scale = 1.0f / sqrtf((float)D);
    for (int l = 0; l < st->n_layers; l++) {
        const orc_layer* L = &st->layers[l];
        for (int t = 0; t < T; t++) post[t] = rmsnorm_fold(h + (size_t)t * H, L->in_ln, st->eps, H, x16 + (size_t)t * H);
        matmul_rows(L->q, x16, H, q, QD, T, QD, H);
        matmul_rows(L->k, x16, H, kn, KD, T, KD, H);
        matmul_rows(L->v, x16, H, vn, KD, T, KD, H);
        scale_rows(q, QD, T, QD, post);
        scale_rows(kn, KD, T, KD, post);
        scale_rows(vn, KD, T, KD, post);
        for (int t = 0; t < T; t++) {
            const int pos = rows[t].pos, n_ctx = rows[t].n_ctx;
            float* kl = rows[t].kc + (size_t)l * NKV * n_ctx * D;
            float* vl = rows[t].vc + (size_t)l * NKV * n_ctx * D;
            const float* cs = st->rope_cos + (size_t)pos * (D / 2);
            const float* sn = st->rope_sin + (size_t)pos * (D / 2);
            for (int hd = 0; hd < NH; hd++) head_norm_rope(q + (size_t)t * QD + hd * D, L->q_norm, st->eps, D, cs, sn);
            for (int gk = 0; gk < NKV; gk++) {
                float* kt = kn + (size_t)t * KD + gk * D;
                head_norm_rope(kt, L->k_norm, st->eps, D, cs, sn);
                for (int i = 0; i < D; i++) {
                    kl[((size_t)gk * n_ctx + pos) * D + i] = act_round(kt[i]);
                    vl[((size_t)gk * n_ctx + pos) * D + i] = act_round(vn[(size_t)t * KD + gk * D + i]);
                }
            }
        }
#pragma omp parallel for schedule(dynamic, 4)
        for (int th = 0; th < T * NH; th++) {
            const int t = th / NH, hd = th % NH, pos = rows[t].pos, n_ctx = rows[t].n_ctx;
            const int gk = hd / rep;
            const float* kl = rows[t].kc + (size_t)l * NKV * n_ctx * D;
            const float* vl = rows[t].vc + (size_t)l * NKV * n_ctx * D;
            const float* qh = q + (size_t)t * QD + hd * D;
            float* sc = (float*)malloc(sizeof(float) * (pos + 1));
            float mx = -INFINITY;
            for (int tt = 0; tt <= pos; tt++) {
                sc[tt] = dot8(qh, kl + ((size_t)gk * n_ctx + tt) * D, D) * scale;
                if (sc[tt] > mx) mx = sc[tt];
            }
            float sum = 0.f;
            for (int tt = 0; tt <= pos; tt++) {
                sc[tt] = expf(sc[tt] - mx);
                sum += sc[tt];
            }
            float* oh = att + (size_t)t * QD + hd * D;
            for (int i = 0; i < D; i++) oh[i] = 0.f;
            for (int tt = 0; tt <= pos; tt++) {
                const float* vr = vl + ((size_t)gk * n_ctx + tt) * D;
                const float p = sc[tt];
                for (int i = 0; i < D; i++) oh[i] += p * vr[i];
            }
            for (int i = 0; i < D; i++) oh[i] = act_round(oh[i] / sum);
            free(sc);
        }
        matmul_rows(L->o, att, QD, y, H, T, H, QD);
        for (size_t i = 0; i < (size_t)T * H; i++) h[i] += y[i];
        for (int t = 0; t < T; t++) post[t] = rmsnorm_fold(h + (size_t)t * H, L->post_ln, st->eps, H, x16 + (size_t)t * H);
        matmul_rows(L->gate, x16, H, g, F, T, F, H);
        matmul_rows(L->up, x16, H, u, F, T, F, H);
        scale_rows(g, F, T, F, post);
        scale_rows(u, F, T, F, post);
        for (size_t j = 0; j < (size_t)T * F; j++) {
            const float sg = g[j] / (1.0f + expf(-g[j]));
            g[j] = act_round(sg * u[j]);
        }
        matmul_rows(L->down, g, F, y, H, T, H, F);
        for (size_t i = 0; i < (size_t)T * H; i++) h[i] += y[i];
    }
    free(buf);
    free(post);
    return 0;
}

/* T consecutive tokens of one sequence (positions pos0 .. pos0+T-1) */
static int tokens_forward(const orc_stack* st, float* kc, float* vc, int n_ctx, int pos0, int T, float* h) {
    orc_row* rows = (orc_row*)malloc(sizeof(orc_row) * T);
    if (!rows) return -1;
    for (int t = 0; t < T; t++) rows[t] = (orc_row){kc, vc, n_ctx, pos0 + t};
    const int rc = rows_forward(st, rows, T, h);
    free(rows);
    return rc;
}

static void token_forward(const orc_stack* st, float* kc, float* vc, int n_ctx, int pos, float* h, float* scratch) {
    (void)scratch;
    tokens_forward(st, kc, vc, n_ctx, pos, 1, h);
}

static size_t scratch_floats(const orc_stack* st, int n_ctx) {
    return (size_t)st->hidden * 2 + (size_t)st->n_heads * st->head_dim * 2 + (size_t)st->n_kv * st->head_dim * 2 +
           (size_t)st->ffn * 2 + (size_t)n_ctx + 64;
}

static void final_norm(const orc_stack* st, const float* h, float* out) {
    const int H = st->hidden;
    double ss = 0.0;
    for (int k = 0; k < H; k++) ss += (double)h[k] * h[k];
    const float inv = 1.0f / sqrtf((float)ss / (float)H + st->eps);
    for (int k = 0; k < H; k++) out[k] = (h[k] * inv) * st->final_norm[k];
}

/* llama_wrapper.c:125-163 semantics: n_tokens embedding rows at pos_start.., causal; writes the
 * post-final-norm hidden of every row to out_all (if non-null) and of the last row to out_last. */
int orc_forward(const orc_stack* st, float* kc, float* vc, int n_ctx, const float* embd, int n_tokens, int pos_start,
                float* out_last, float* out_all) {
    if (pos_start < 0 || pos_start + n_tokens > n_ctx) return -1;
    const int H = st->hidden, BLK = 64;
    float* h = (float*)malloc(sizeof(float) * (size_t)H * (n_tokens < BLK ? n_tokens : BLK));
    if (!h) return -1;
    for (int t0 = 0; t0 < n_tokens; t0 += BLK) {
        const int T = n_tokens - t0 < BLK ? n_tokens - t0 : BLK;
        memcpy(h, embd + (size_t)t0 * H, sizeof(float) * (size_t)T * H);
        if (tokens_forward(st, kc, vc, n_ctx, pos_start + t0, T, h)) return -1;
        for (int t = 0; t < T; t++) {
            if (out_all) final_norm(st, h + (size_t)t * H, out_all + (size_t)(t0 + t) * H);
            if (t0 + t == n_tokens - 1 && out_last) final_norm(st, h + (size_t)t * H, out_last);
        }
    }
    free(h);
    return 0;
}

/* logits[V] = head[V][H] . fp16(hidden)  (llamacpp_talker_server.py:165; code_predictor_server.py:129) */
void orc_head(const float* head, int V, int H, const float* hidden, float* logits) {
    float* x = (float*)malloc(sizeof(float) * H);
    for (int k = 0; k < H; k++) x[k] = act_round(hidden[k]);
    matvec(head, x, logits, V, H);
    free(x);
}

static int argmax_lowest(const float* l, int n, float* margin) {
    int bi = 0;
    float best = l[0], second = -INFINITY;
    for (int i = 1; i < n; i++) {
        if (l[i] > best) {
            second = best;
            best = l[i];
            bi = i;
        } else if (l[i] > second) second = l[i];
    }
    if (margin) *margin = best - second;
    return bi;
}

/* dual_npu/code_predictor_server.py:94-140, greedy, sequential prefill.
 * cp_emb[g], cp_head[g]: f32 [cp_vocab][H].  forced (optional, [n_groups]): teacher-forced tokens
 * fed forward instead of the argmax (entries < 0 = free-running).  margins: top1-top2 per group. */
int orc_cp_predict(const orc_stack* cp, const float* talker_emb, int talker_vocab, const float* const* cp_emb,
                   const float* const* cp_head, int cp_vocab, int n_groups, const float* hidden, int code0,
                   const int* forced, int* out_codes, float* margins, float* out_hidden_all) {
    const int H = cp->hidden, D = cp->head_dim, n_ctx = n_groups + 1;
    const size_t kvn = (size_t)cp->n_layers * cp->n_kv * n_ctx * D;
    float* kc = (float*)calloc(kvn, sizeof(float));
    float* vc = (float*)calloc(kvn, sizeof(float));
    float* scratch = (float*)malloc(sizeof(float) * scratch_floats(cp, n_ctx));
    float* h = (float*)malloc(sizeof(float) * H);
    float* hid = (float*)malloc(sizeof(float) * H);
    float* logits = (float*)malloc(sizeof(float) * cp_vocab);
    if (!kc || !vc || !scratch || !h || !hid || !logits) return -1;
    memcpy(h, hidden, sizeof(float) * H);
    token_forward(cp, kc, vc, n_ctx, 0, h, scratch);
    if (code0 >= 0 && code0 < talker_vocab) memcpy(h, talker_emb + (size_t)code0 * H, sizeof(float) * H);
    else memset(h, 0, sizeof(float) * H);
    for (int g = 0; g < n_groups; g++) {
        token_forward(cp, kc, vc, n_ctx, g + 1, h, scratch);
        final_norm(cp, h, hid);
        if (out_hidden_all) memcpy(out_hidden_all + (size_t)g * H, hid, sizeof(float) * H);
        {   /* the group head is a GEMM over the final-normed state: folded like every normed GEMM */
            const float post = rmsnorm_fold(h, cp->final_norm, cp->eps, H, hid);
            matvec(cp_head[g], hid, logits, cp_vocab, H);
            for (int v = 0; v < cp_vocab; v++) logits[v] *= post;
        }
        int tok = argmax_lowest(logits, cp_vocab, margins ? &margins[g] : NULL);
        out_codes[g] = tok;
        if (forced && forced[g] >= 0) tok = forced[g];
        if (g + 1 < n_groups) {
            if (tok >= 0 && tok < cp_vocab) memcpy(h, cp_emb[g] + (size_t)tok * H, sizeof(float) * H);
            else memset(h, 0, sizeof(float) * H);
        }
    }
    free(kc); free(vc); free(scratch); free(h); free(hid); free(logits);
    return 0;
}

/* One decode step of B independent sequences (row b: cache kc[b]/vc[b] of n_ctx[b] positions, appended at pos[b]):
 * the batched form of orc_forward(n_tokens = 1); per row bit-identical to it.  out: post-final-norm hidden [B][H]. */
int orc_forward_batch(const orc_stack* st, float* const* kc, float* const* vc, const int* n_ctx, const int* pos, int B,
                      const float* embd, float* out) {
    const int H = st->hidden;
    orc_row* rows = (orc_row*)malloc(sizeof(orc_row) * B);
    float* h = (float*)malloc(sizeof(float) * (size_t)B * H);
    if (!rows || !h) return -1;
    for (int b = 0; b < B; b++) {
        if (pos[b] < 0 || pos[b] >= n_ctx[b]) return -1;
        rows[b] = (orc_row){kc[b], vc[b], n_ctx[b], pos[b]};
    }
    memcpy(h, embd, sizeof(float) * (size_t)B * H);
    if (rows_forward(st, rows, B, h)) return -1;
    for (int b = 0; b < B; b++) final_norm(st, h + (size_t)b * H, out + (size_t)b * H);
    free(rows);
    free(h);
    return 0;
}

/* orc_cp_predict for B rows at once (weights read once per pass for the whole batch); per row bit-identical to
 * it.  hidden [B][H], code0 [B], forced [B][n_groups] or NULL, out_codes / margins [B][n_groups]. */
int orc_cp_predict_batch(const orc_stack* cp, const float* talker_emb, int talker_vocab, const float* const* cp_emb,
                         const float* const* cp_head, int cp_vocab, int n_groups, int B, const float* hidden,
                         const int* code0, const int* forced, int* out_codes, float* margins) {
    const int H = cp->hidden, D = cp->head_dim, n_ctx = n_groups + 1;
    const size_t kvn = (size_t)cp->n_layers * cp->n_kv * n_ctx * D;
    float* kc = (float*)calloc(kvn * B, sizeof(float));
    float* vc = (float*)calloc(kvn * B, sizeof(float));
    orc_row* rows = (orc_row*)malloc(sizeof(orc_row) * B);
    float* h = (float*)malloc(sizeof(float) * (size_t)B * H);
    float* x = (float*)malloc(sizeof(float) * (size_t)B * H);
    float* logits = (float*)malloc(sizeof(float) * (size_t)B * cp_vocab);
    float* rows_post = (float*)malloc(sizeof(float) * B);
    if (!kc || !vc || !rows || !h || !x || !logits || !rows_post) return -1;
    for (int b = 0; b < B; b++) rows[b] = (orc_row){kc + kvn * b, vc + kvn * b, n_ctx, 0};
    memcpy(h, hidden, sizeof(float) * (size_t)B * H);
    if (rows_forward(cp, rows, B, h)) return -1;
    for (int b = 0; b < B; b++) {
        if (code0[b] >= 0 && code0[b] < talker_vocab) memcpy(h + (size_t)b * H, talker_emb + (size_t)code0[b] * H, sizeof(float) * H);
        else memset(h + (size_t)b * H, 0, sizeof(float) * H);
    }
    for (int g = 0; g < n_groups; g++) {
        for (int b = 0; b < B; b++) rows[b].pos = g + 1;
        if (rows_forward(cp, rows, B, h)) return -1;
        for (int b = 0; b < B; b++) rows_post[b] = rmsnorm_fold(h + (size_t)b * H, cp->final_norm, cp->eps, H, x + (size_t)b * H);
        matmul_rows(cp_head[g], x, H, logits, cp_vocab, B, cp_vocab, H);
        scale_rows(logits, cp_vocab, B, cp_vocab, rows_post);
        for (int b = 0; b < B; b++) {
            int tok = argmax_lowest(logits + (size_t)b * cp_vocab, cp_vocab, margins ? &margins[(size_t)b * n_groups + g] : NULL);
            out_codes[(size_t)b * n_groups + g] = tok;
            if (forced && forced[(size_t)b * n_groups + g] >= 0) tok = forced[(size_t)b * n_groups + g];
            if (g + 1 < n_groups) {
                if (tok >= 0 && tok < cp_vocab) memcpy(h + (size_t)b * H, cp_emb[g] + (size_t)tok * H, sizeof(float) * H);
                else memset(h + (size_t)b * H, 0, sizeof(float) * H);
            }
        }
    }
    free(kc); free(vc); free(rows); free(h); free(x); free(logits); free(rows_post);
    return 0;
}

/* logits[B][V] = head . fp16(hidden[b])  (batched orc_head) */
void orc_head_batch(const float* head, int V, int H, const float* hidden, int B, float* logits) {
    float* x = (float*)malloc(sizeof(float) * (size_t)B * H);
    for (size_t i = 0; i < (size_t)B * H; i++) x[i] = act_round(hidden[i]);
    matmul_rows(head, x, H, logits, V, B, V, H);
    free(x);
}

/* Plain linear pieces for kernel-level checks: y = W . fp16round(x) etc. */
void orc_matvec(const float* W, const float* x, float* y, int N, int K) { matvec(W, x, y, N, K); }
float orc_rmsnorm_fold(const float* h, const float* gamma, float eps, int H, float* x16) {
    return rmsnorm_fold(h, gamma, eps, H, x16);
}
