// q3_kernels.h -- launch interface of the gfx950 kernels (q3_kernels.hip).
//
// Numerics contract shared with oracle/q3_oracle.c (DESIGN.md "Numerics"):
//   * projection weights fp16, residual stream / norms / softmax / tables f32,
//     every GEMM input rounded to fp16 (saturating), f32 accumulation;
//   * RMSNorm folded around the GEMM: input xh = fp16((h*gamma)/16) written by the producer of h,
//     accumulators multiplied by inv_rms(row)*16;
//   * K/V cache fp16, q kept f32.
#pragma once
#include "q3_common.h"

namespace q3 {

enum { PRO_F16 = 0, PRO_NORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

// y[M][N] = A[M][K] . W[N][K]^T with W in MFMA-fragment order (see pack_linear).
struct LinArgs {
    int tl_node = -1;  // diagnostic timeline build only: graph node index
    const half_t* wp = nullptr;  // packed weights
    int N = 0, K = 0, M = 0;   // rows [m_begin, M) are computed
    int m_begin = 0;
    int swap_grid = 0;  // set by the launcher
    int nt = 0;  // 1: stream the weights with non-temporal loads (read once per step: talker)
    // A operands are stored in MFMA fragment order (frag_idx), buffers padded to 16 rows.
    // prologue PRO_F16: A = x16[M][K] (fp16).  PRO_NORM: A = x16 = the producer's xh = fp16((h*gamma)/16) and the
    // accumulators are multiplied by 16*inv[m], inv[m] = 1/sqrt(sum(ssq[m][0..ssq_parts))/K + eps).
    const half_t* x16 = nullptr;
    const float* ssq = nullptr;
    int ssq_parts = 0;
    float eps = 1e-6f;
    // EPI_STORE: y[m*ldy + n] = acc
    float* y = nullptr;
    int ldy = 0;
    // EPI_RESID: h_out[m*N+n] += acc; ssq_out[m*(N/16) + n/16] = sum of squares of the new 16 values;
    // xh_out[m][n] = fp16((h_out*gamma[n])/16) for the consumer whose norm weight is `gamma` (null: none)
    float* h_out = nullptr;
    float* ssq_out = nullptr;
    half_t* xh_out = nullptr;
    const float* gamma = nullptr;
    // EPI_SWIGLU (gate/up tile-interleaved weights): act[m*(N/2)+j] = fp16(silu(g)*u)
    half_t* act = nullptr;
};
int launch_linear(hipStream_t s, const LinArgs& a, int pro, int epi);

// Repack a row-major fp16 matrix src[N][K] into fragment order inside dst:
// source 16-row tile ts lands at destination tile (tile_off + ts*tile_stride).
int launch_pack_linear(hipStream_t s, const half_t* src, int N, int K, half_t* dst, int tile_off,
                       int tile_stride);

enum { ATTN_FUSED = 0, ATTN_PREP = 1, ATTN_ATTEND = 2 };
struct AttnArgs {
    int tl_node = -1;  // diagnostic timeline build only: graph node index
    float* qkv = nullptr;  // [R][ld]: q heads, then k heads, then v heads (raw projections)
    int ld = 0, R = 0, row0 = 0;   // rows row0 .. row0+R-1
    const float* q_norm = nullptr;
    const float* k_norm = nullptr;
    float eps = 1e-6f;
    const float* rope_cos = nullptr;  // [max_pos][64]
    const float* rope_sin = nullptr;
    const int* slot = nullptr;  // [R] or null -> slot_base + r*slot_stride
    const int* pos = nullptr;   // [R] or null -> pos_base + r*pos_stride
    int slot_base = 0, slot_stride = 0, pos_base = 0, pos_stride = 0;
    half_t* kc = nullptr;  // this layer: [slot][n_kv][n_ctx][128]
    half_t* vc = nullptr;
    int n_ctx = 0, n_kv = 0, n_heads = 0;
    half_t* out = nullptr;  // [R][n_heads*128]
    float scale = 0.f;
    int threads = 256;
    // rows r with (r % valid_mod) >= valid_n are padding (no slot of their own): skipped.  0 = every row is real.
    int valid_mod = 0, valid_n = 0;
    // ATTN_ATTEND over runs of consecutive positions of one slot (prefill): tiles[i] = {first row, rows (1..16), slot,
    // position of the first row}; null with n_tiles > 0 = the rows row0.. are ONE run (slot_base, pos_base + i)
    const int* tiles = nullptr;
    int n_tiles = 0;
};
int launch_attn(hipStream_t s, const AttnArgs& a, int mode);

// Uploaded row-major rows[R][H] -> residual stream h in fragment order (see frag_idx in q3_kernels.hip)
// + ssq[m][p] = sum_{k in 16-block p} rows[m][k]^2  (H/16 partials per row)
// + xh = fp16((rows*gamma)/16), the pre-scaled GEMM input of the layer whose norm weight is gamma (null: none)
int launch_ssq_rows(hipStream_t s, const float* rows, float* h, float* ssq, int R, int H, half_t* xh = nullptr,
                    const float* gamma = nullptr, int dst_row0 = 0);   // input row r lands in row dst_row0 + r

// hidden = (h*inv)*gamma per row; optional outputs: f32 hidden, fp16 hidden,
// a second f32 copy (+ its ssq partials) that seeds the code predictor.
struct FinalNormArgs {
    int tl_node = -1;  // diagnostic timeline build only: graph node index
    const float* h = nullptr;
    const float* ssq = nullptr;
    int ssq_parts = 0;
    const float* gamma = nullptr;
    float eps = 1e-6f;
    int R = 0, H = 0, row0 = 0;    // output rows row0 .. row0+R-1
    const int* row_map = nullptr;  // optional: source row of output row r (indexed by the output row)
    int src_off = 0;               // without a map: source row = output row + src_off
    float* out_f32 = nullptr;
    half_t* out_f16 = nullptr;
    float* out_copy = nullptr;
    float* out_copy_ssq = nullptr;
    int out_copy_row_off = 0;                // the copy of output row r lands in row r + out_copy_row_off
    half_t* out_copy_xh = nullptr;           // pre-scaled GEMM input of the copy's consumer ...
    const float* out_copy_gamma = nullptr;   // ... whose norm weight this is
};
int launch_final_norm(hipStream_t s, const FinalNormArgs& a);

// h[r] = table[tok_r] (zeros when tok_r is out of range), with ssq partials.  tok_r = tok[r*tok_stride]
// when n_frames is null, else column `col` of row r's current frame in a codes array laid out as in
// TalkerSampleArgs (frame = n_frames[r]-1, clamped to [0, frame_cap)).
int launch_gather_embed(hipStream_t s, const float* table, int V, int H, const int* tok, int tok_stride,
                        const int* n_frames, int frame_cap, int col, float* h, float* ssq, int R, int row0 = 0,
                        int R_total = 0, const int* forced = nullptr, half_t* xh = nullptr, const float* gamma = nullptr);

// Talker sampling (llamacpp_talker_server.py:163-206, greedy form).
struct TalkerSampleArgs {
    int tl_node = -1;  // diagnostic timeline build only: graph node index
    const float* logits = nullptr;  // [R][V]
    int V = 0, R = 0, row0 = 0, R_total = 0;   // rows row0..row0+R-1 of a batch of R_total (0 = R)
    int audio_vocab = 2048, eos = 2150;
    int* past = nullptr;    // [R][32] ring of emitted code_0
    int* n_past = nullptr;  // [R]
    const int* n_text = nullptr;  // [R]
    int* done = nullptr;          // [R]
    // codes of frame f of row r live at codes[(f*R + r)*16 .. +16); f = n_frames[r] (per-row counter,
    // incremented here; clamped to frame_cap-1).  Column 0 is written: code_0, or -1 once the row finished.
    int* codes = nullptr;
    int* n_frames = nullptr;      // [R]
    int frame_cap = 0;
    const int* pos0 = nullptr;    // [R] prefix length
    int* pos = nullptr;           // [R] position of the talker step that follows = pos0 + frames emitted before
    int ignore_eos = 0;
    int max_frames = 0;
    float rep_penalty = 1.2f;
    // temperature <= 1e-6: arg-max (the reference's limit); else top-k / temperature / top-p on the device
    float temperature = 0.f, top_p = 0.95f;
    int top_k = 50;              // <= 0 or >= V: every entry (the reference skips its argpartition then)
    unsigned long long seed = 0;
    const unsigned long long* seed_ptr = nullptr;  // device array [rows] overriding `seed`: one draw stream per slot (per request and per refill, graph-safe)
    // teacher forcing (tests): same layout as `codes`; entries >= 0 replace the decision that is FED BACK
    // (ring of past ids, CP input, feedback sum) while `codes` still records what the device decided
    const int* forced = nullptr;
};
int launch_talker_sample(hipStream_t s, const TalkerSampleArgs& a);

// Code-predictor group argmax (+ next embedding gather, or the feedback sum
// of tts_client.py:199-208 after the last group).
struct CpArgmaxArgs {
    int tl_node = -1;  // diagnostic timeline build only: graph node index
    const float* logits = nullptr;  // [R][V]
    int V = 0, R = 0, H = 0, row0 = 0, R_total = 0;
    int group = 0;                 // writes column group+1 of the row's current frame
    int* codes = nullptr;          // as in TalkerSampleArgs (frame = n_frames[r]-1)
    const int* n_frames = nullptr;
    int frame_cap = 0;
    const float* next_table = nullptr;  // f32 [V][H] of this group, or null
    float* h_out = nullptr;
    float* ssq_out = nullptr;
    half_t* xh_out = nullptr;           // pre-scaled GEMM input for the layer that consumes h_out ...
    const float* gamma_next = nullptr;  // ... whose input norm weight this is
    // feedback (when talker_emb != null): h_out = talker_emb[code0] + sum_g cp_tables[g][code_{g+1}] + pad
    const float* talker_emb = nullptr;
    int talker_vocab = 0;
    const float* const* cp_tables = nullptr;  // device array [n_groups]
    const float* pad_embed = nullptr;
    int n_groups = 15;
    float temperature = 0.f;   // <= 1e-6: arg-max
    int top_k = 50;            // <= 0 or >= V: every entry
    unsigned long long seed = 0;
    const unsigned long long* seed_ptr = nullptr;  // device array [rows] overriding `seed` (one stream per slot)
    const int* forced = nullptr;                   // teacher forcing (tests), see TalkerSampleArgs
};
int launch_cp_argmax(hipStream_t s, const CpArgmaxArgs& a);

// Stand-alone feedback sum (tts_client.py:199-208) for host-provided codes.
int launch_feedback(hipStream_t s, const int* codes16, int R, const float* talker_emb, int talker_vocab,
                    const float* const* cp_tables, int cp_vocab, int n_groups, const float* pad_embed,
                    float* h_out, float* ssq_out, int H, half_t* xh_out = nullptr, const float* gamma = nullptr);

// diagnostic timeline build: node numbering of the launches that follow (no-op otherwise)
int tl_next_node();

}  // namespace q3
