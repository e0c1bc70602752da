// q3_model.h -- device-resident model, KV cache, workspaces and the layer-stack runner.
#pragma once
#include "q3_kernels.h"

namespace q3 {

struct DevLinear {
    half_t* wp = nullptr;  // fragment-packed fp16
    int N = 0, K = 0;
};

struct DevLayer {
    DevLinear qkv, o, gu, down;
    float *in_ln = nullptr, *post_ln = nullptr, *q_norm = nullptr, *k_norm = nullptr;
};

struct DevStack {
    std::vector<DevLayer> L;
    float* final_norm = nullptr;
    // norm weight of whatever consumes the stack's output THROUGH A GEMM (code predictor: final_norm, the group
    // heads read the last layer's xh); null when the output is only normed element-wise (talker)
    const float* tail_gamma = nullptr;
    int ffn = 0;
    int nt = 0;  // non-temporal weight stream
    size_t weight_bytes = 0;
};

struct Model {
    ModelCfg cfg;
    int device = 0;
    bool has_talker = false, has_cp = false;
    DevStack talker, cp;
    float* talker_emb = nullptr;  // f32 [talker_vocab][hidden]
    DevLinear talker_head;
    std::vector<float*> cp_emb;   // f32 [cp_vocab][hidden] each
    std::vector<DevLinear> cp_head;
    const float** d_cp_emb_ptrs = nullptr;  // device array of the cp_emb pointers
    float *rope_cos = nullptr, *rope_sin = nullptr;  // [max_pos][head_dim/2]
    int max_pos = 0;
    std::vector<void*> allocs;
    size_t device_bytes = 0;
};

// `path`: a Q3TTSW1 container, or the reference's own files (Pack::open_auto); `aux_dir`: its --embeddings_dir
Model* model_load(const char* path, bool want_talker, bool want_cp, const char* aux_dir = nullptr);
void model_free(Model* m);

struct KVCache {
    half_t *k = nullptr, *v = nullptr;
    int n_layers = 0, n_slots = 0, n_kv = 0, n_ctx = 0, head_dim = 128;
    size_t layer_stride() const { return (size_t)n_slots * n_kv * n_ctx * head_dim; }
    size_t bytes() const { return layer_stride() * n_layers * sizeof(half_t) * 2; }
};
int kv_alloc(KVCache& kv, int n_layers, int n_slots, int n_kv, int n_ctx);
void kv_free(KVCache& kv);

// Activations of one stack pass over up to max_rows rows.
// host mirror of frag_idx (q3_kernels.hip): position of element (m, k) of a [rows][K] GEMM-input matrix
inline size_t frag_idx_host(int m, int k, int K) {
    return ((((size_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + (m & 15) + 16 * ((k >> 3) & 3)) << 3) + (k & 7);
}

struct Work {
    int max_rows = 0, hidden = 0;   // max_rows is padded to a multiple of 128 (largest row tile)
    float* rows_in = nullptr;       // row-major staging of uploaded embedding rows
    float *h = nullptr, *ssq = nullptr, *qkv = nullptr;   // h: fragment order
    half_t* xh = nullptr;          // fp16((h*gamma_consumer)/16), fragment order: the next normed GEMM's input
    half_t *attn = nullptr, *act = nullptr;
    float* hidden_f32 = nullptr;   // post-final-norm
    half_t* hidden_f16 = nullptr;
    float* logits = nullptr;       // [max_rows][max vocab]
    // (slot, position) of the rows of the code predictor's two-position first pass (cp_frame): rows [0, R16) are
    // position 1, rows [R16, 2*R16) position 0 of slots 0..R16-1; filled for map_R16
    int *map_slot = nullptr, *map_pos = nullptr;
    int map_R16 = 0;
};
int work_alloc(Work& w, const ModelCfg& c, int max_rows, int ffn, int max_vocab);
void work_free(Work& w);

struct RowMap {            // which (slot, position) each row feeds
    const int* slot = nullptr;  // device [R] or null -> slot_base + r*slot_stride
    const int* pos = nullptr;   // device [R] or null -> pos_base + r*pos_stride
    int slot_base = 0, slot_stride = 0, pos_base = 0, pos_stride = 0;
    bool same_slot_rows = false;  // rows depend on each other through the cache (prefill): split prep/attend
    int valid_mod = 0, valid_n = 0;  // rows r with (r % valid_mod) >= valid_n are padding (AttnArgs)
    // same_slot_rows: the rows as runs of <= 16 consecutive positions of one slot (device int[4] per tile: first row,
    // rows, slot, first position); without a table a pass with slot_stride 0 / pos_stride 1 is tiled implicitly
    const int* tiles = nullptr;
    int n_tiles = 0;
};

// Run every layer of `st` over R rows whose residual stream (+ssq partials) sits in w.h / w.ssq.
int run_stack(hipStream_t s, const Model& m, const DevStack& st, Work& w, KVCache& kv, int R,
              const RowMap& rm, int attn_threads, int row0 = 0);

struct GraphExec {
    hipGraph_t g = nullptr;
    hipGraphExec_t e = nullptr;
    void reset();
};

}  // namespace q3
