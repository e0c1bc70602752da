// q3_cp.h -- one code-predictor frame over R rows (shared by the cp_* ABI and the fused engine).
#pragma once
#include "q3_model.h"

namespace q3 {

struct CpFrameIO {
    int* codes = nullptr;           // device codes array (see TalkerSampleArgs); column 0 = code_0 is read
    const int* n_frames = nullptr;  // device [R]
    int frame_cap = 1;
    // feedback after the last group (tts_client.py:199-208); null = none
    float* fb_h = nullptr;
    float* fb_ssq = nullptr;
    half_t* fb_xh = nullptr;          // pre-scaled GEMM input of the talker's first layer ...
    const float* fb_gamma = nullptr;  // ... and that layer's input norm weight
    const float* pad_embed = nullptr;
    // sampling of the 15 groups (code_predictor_server.py:87-92): temperature <= 1e-6 = arg-max
    float temperature = 0.f;
    int top_k = 50;
    unsigned long long seed = 0;
    const unsigned long long* seed_ptr = nullptr;  // device array [rows] overriding `seed`: one draw stream per slot (engine: per request and per refill)
    const int* forced = nullptr;                   // teacher forcing (tests): see TalkerSampleArgs
};

// Runs positions 0..n_groups, writes columns 1..n_groups of each row's frame.  Rows row0..row0+R-1 of a batch of
// R_total rows.  The talker hidden of row r (position 0's input) must sit in row cp_seed_row0(R, row0, R_total) + r
// of w.h / w.ssq / w.xh: 0 in the sequential form (positions 0 and 1 as two passes), R16 = R rounded up to 16 in the
// two-position form (one pass over 2 rows per utterance = the reference's --batch_prefill,
// code_predictor_server.py:106-118, C++ default: code_predictor_server.cpp:257), which is used for whole batches.
int cp_seed_row0(const Work& w, int R, int row0 = 0, int R_total = 0);
int cp_frame(hipStream_t s, const Model& m, Work& w, KVCache& kv, int R, const CpFrameIO& io, int row0 = 0,
             int R_total = 0);

}  // namespace q3
