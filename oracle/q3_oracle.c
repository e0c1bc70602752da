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
 * (saturating), accumulation and everything else is f32; K/V are stored rounded to fp16.
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

/* y[n] = sum_k W[n][k] x[k], f32 accumulate */
static void matvec(const float* W, const float* x, float* y, int N, int K) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        const float* w = W + (size_t)n * K;
        float acc = 0.f;
        for (int k = 0; k < K; k++) acc += w[k] * x[k];
        y[n] = acc;
    }
}

static void rmsnorm_round(const float* h, const float* gamma, float eps, int H, float* x16) {
    double ss = 0.0;
    for (int k = 0; k < H; k++) ss += (double)h[k] * h[k];
    const float inv = 1.0f / sqrtf((float)ss / (float)H + eps);
    for (int k = 0; k < H; k++) x16[k] = act_round((h[k] * inv) * gamma[k]);
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

/* One token through every layer.  kc/vc: [n_layers][n_kv][n_ctx][D] holding fp16-rounded values.
 * h (in/out): residual stream [H]. */
static void token_forward(const orc_stack* st, float* kc, float* vc, int n_ctx, int pos, float* h, float* scratch) {
    const int H = st->hidden, D = st->head_dim, NH = st->n_heads, NKV = st->n_kv, F = st->ffn;
    const int rep = NH / NKV;
    float* x16 = scratch;                 /* H */
    float* q = x16 + H;                   /* NH*D */
    float* kn = q + NH * D;               /* NKV*D */
    float* vn = kn + NKV * D;             /* NKV*D */
    float* att = vn + NKV * D;            /* NH*D */
    float* g = att + NH * D;              /* F */
    float* u = g + F;                     /* F */
    float* y = u + F;                     /* H */
    float* sc = y + H;                    /* n_ctx */
    const float scale = 1.0f / sqrtf((float)D);
    const float* cs = st->rope_cos + (size_t)pos * (D / 2);
    const float* sn = st->rope_sin + (size_t)pos * (D / 2);
    for (int l = 0; l < st->n_layers; l++) {
        const orc_layer* L = &st->layers[l];
        rmsnorm_round(h, L->in_ln, st->eps, H, x16);
        matvec(L->q, x16, q, NH * D, H);
        matvec(L->k, x16, kn, NKV * D, H);
        matvec(L->v, x16, vn, NKV * D, H);
        for (int hd = 0; hd < NH; hd++) head_norm_rope(q + hd * D, L->q_norm, st->eps, D, cs, sn);
        float* kl = kc + (size_t)l * NKV * n_ctx * D;
        float* vl = vc + (size_t)l * NKV * n_ctx * D;
        for (int gk = 0; gk < NKV; gk++) {
            head_norm_rope(kn + gk * D, L->k_norm, st->eps, D, cs, sn);
            for (int i = 0; i < D; i++) {
                kl[((size_t)gk * n_ctx + pos) * D + i] = act_round(kn[gk * D + i]);
                vl[((size_t)gk * n_ctx + pos) * D + i] = act_round(vn[gk * D + i]);
            }
        }
        for (int hd = 0; hd < NH; hd++) {
            const int gk = hd / rep;
            const float* qh = q + hd * D;
            float mx = -INFINITY;
            for (int t = 0; t <= pos; t++) {
                const float* kr = kl + ((size_t)gk * n_ctx + t) * D;
                float d = 0.f;
                for (int i = 0; i < D; i++) d += qh[i] * kr[i];
                sc[t] = d * scale;
                if (sc[t] > mx) mx = sc[t];
            }
            float sum = 0.f;
            for (int t = 0; t <= pos; t++) {
                sc[t] = expf(sc[t] - mx);
                sum += sc[t];
            }
            float* oh = att + hd * D;
            for (int i = 0; i < D; i++) oh[i] = 0.f;
            for (int t = 0; t <= pos; t++) {
                const float* vr = vl + ((size_t)gk * n_ctx + t) * D;
                const float p = sc[t];
                for (int i = 0; i < D; i++) oh[i] += p * vr[i];
            }
            for (int i = 0; i < D; i++) oh[i] = act_round(oh[i] / sum);
        }
        matvec(L->o, att, y, H, NH * D);
        for (int k = 0; k < H; k++) h[k] += y[k];
        rmsnorm_round(h, L->post_ln, st->eps, H, x16);
        matvec(L->gate, x16, g, F, H);
        matvec(L->up, x16, u, F, H);
        for (int j = 0; j < F; j++) {
            const float sg = g[j] / (1.0f + expf(-g[j]));
            g[j] = act_round(sg * u[j]);
        }
        matvec(L->down, g, y, H, F);
        for (int k = 0; k < H; k++) h[k] += y[k];
    }
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
    const int H = st->hidden;
    float* scratch = (float*)malloc(sizeof(float) * scratch_floats(st, n_ctx));
    float* h = (float*)malloc(sizeof(float) * H);
    if (!scratch || !h) return -1;
    for (int t = 0; t < n_tokens; t++) {
        memcpy(h, embd + (size_t)t * H, sizeof(float) * H);
        token_forward(st, kc, vc, n_ctx, pos_start + t, h, scratch);
        if (out_all) final_norm(st, h, out_all + (size_t)t * H);
        if (t == n_tokens - 1 && out_last) final_norm(st, h, out_last);
    }
    free(scratch);
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
        orc_head(cp_head[g], cp_vocab, H, hid, logits);
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

/* Plain linear pieces for kernel-level checks: y = W . fp16round(x) etc. */
void orc_matvec(const float* W, const float* x, float* y, int N, int K) { matvec(W, x, y, N, K); }
void orc_rmsnorm_round(const float* h, const float* gamma, float eps, int H, float* x16) {
    rmsnorm_round(h, gamma, eps, H, x16);
}
