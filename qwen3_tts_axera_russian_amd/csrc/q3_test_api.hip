// q3_test_api.hip -- kernel-level entry points used only by tests/ and bench.py (host arrays in, host arrays
// out).  Built into lib/libqwen3tts_test.so, which links against the product library; NOT part of
// libqwen3tts.so / llama_wrapper.so.
#include "q3_model.h"
#include <chrono>

using namespace q3;

namespace q3 {
int set_linear_tuning(int K, int mt16, int kbw);
int set_linear_split_rows(int on);
int set_linear_narrow8(int on);
int set_linear_wide_tiles(int on);
int set_attn_short(int on);
int set_gemm_min_rows(int n);
int set_gemm_glds(int on);
}

namespace {
struct DBuf {
    void* p = nullptr;
    ~DBuf() {
        if (p) hipFree(p);
    }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess; }
    bool up(const void* src, size_t n) { return alloc(n) && (n == 0 || hipMemcpy(p, src, n, hipMemcpyHostToDevice) == hipSuccess); }
};
}  // namespace

extern "C" {

int q3t_set_linear_tuning(int K, int mt16, int kbw) { return set_linear_tuning(K, mt16, kbw); }
int q3t_set_linear_split_rows(int on) { return set_linear_split_rows(on); }
int q3t_set_linear_narrow8(int on) { return set_linear_narrow8(on); }   // 0: o / down through linear_kernel (round 2)
int q3t_set_linear_wide_tiles(int on) { return set_linear_wide_tiles(on); }
int q3t_set_attn_short(int on) { return set_attn_short(on); }
// rows >= n take the tiled GEMM (gemm_kernel) instead of the weight-streaming kernel; default 65
int q3t_set_gemm_min_rows(int n) { return set_gemm_min_rows(n); }
// 1: LDS-DMA ring GEMM (default), 0: the register-staged double-buffer GEMM
int q3t_set_gemm_glds(int on) { return set_gemm_glds(on); }

// One linear launch.  W is row-major fp16 [N][K]; gateup != 0 means rows [0,N/2) are gate and
// [N/2,N) up (tile-interleaved on the device like the model loader does).
// pro: 0 = x16[M][K] fp16, 1 = RMSNorm(h[M][K], gamma, eps).
// epi: 0 = y[M][N] store, 1 = h_io[M][N] += y with ssq_out[M][N/16], 2 = act_out[M][N/2] fp16.
int q3t_linear(int M, int N, int K, const uint16_t* W, int gateup, int pro, int epi, const uint16_t* x16,
               const float* h, const float* gamma, float eps, float* y_or_h_io, float* ssq_out, uint16_t* act_out,
               int nt) {
    hipStream_t s = nullptr;
    const int Mp = (M + 127) / 128 * 128;  // buffers padded to the largest row tile
    DBuf dW, dWp, dx, dhrows, dh, dssq, dg, dy, dso, dact, dxh;
    if (!dW.up(W, (size_t)N * K * 2) || !dWp.alloc((size_t)N * K * 2)) return -1;
    if (gateup) {
        if (launch_pack_linear(s, (const half_t*)dW.p, N / 2, K, (half_t*)dWp.p, 0, 2)) return -1;
        if (launch_pack_linear(s, (const half_t*)dW.p + (size_t)(N / 2) * K, N / 2, K, (half_t*)dWp.p, 1, 2)) return -1;
    } else {
        if (launch_pack_linear(s, (const half_t*)dW.p, N, K, (half_t*)dWp.p, 0, 1)) return -1;
    }
    LinArgs a;
    a.wp = (const half_t*)dWp.p;
    a.N = N;
    a.K = K;
    a.M = M;
    a.nt = nt;
    if (pro == PRO_F16) {
        std::vector<uint16_t> xp((size_t)Mp * K, 0);   // host rows -> fragment order
        for (int m = 0; m < M; m++)
            for (int k = 0; k < K; k++) xp[frag_idx_host(m, k, K)] = x16[(size_t)m * K + k];
        if (!dx.up(xp.data(), xp.size() * 2)) return -1;
        a.x16 = (const half_t*)dx.p;
    } else {
        if (!dhrows.up(h, (size_t)M * K * 4) || !dh.alloc((size_t)Mp * K * 4) || !dssq.alloc((size_t)Mp * (K / 16) * 4) ||
            !dg.up(gamma, (size_t)K * 4))
            return -1;
        hipMemset(dh.p, 0, (size_t)Mp * K * 4);
        if (!dxh.alloc((size_t)Mp * K * 2)) return -1;
        hipMemset(dxh.p, 0, (size_t)Mp * K * 2);
        // the producer's side of the folded RMSNorm: h, its sum-of-squares partials and xh = fp16((h*gamma)/16)
        if (launch_ssq_rows(s, (const float*)dhrows.p, (float*)dh.p, (float*)dssq.p, M, K, (half_t*)dxh.p, (const float*)dg.p)) return -1;
        a.x16 = (const half_t*)dxh.p;
        a.ssq = (const float*)dssq.p;
        a.ssq_parts = K / 16;
        a.eps = eps;
    }
    std::vector<float> hp;
    if (epi == EPI_STORE) {
        if (!dy.alloc((size_t)M * N * 4)) return -1;
        a.y = (float*)dy.p;
        a.ldy = N;
    } else if (epi == EPI_RESID) {
        hp.assign((size_t)Mp * N, 0.f);
        for (int m = 0; m < M; m++)
            for (int n = 0; n < N; n++) hp[frag_idx_host(m, n, N)] = y_or_h_io[(size_t)m * N + n];
        if (!dy.up(hp.data(), hp.size() * 4) || !dso.alloc((size_t)Mp * (N / 16) * 4)) return -1;
        a.h_out = (float*)dy.p;
        a.ssq_out = (float*)dso.p;
    } else {
        if (!dact.alloc((size_t)Mp * (N / 2) * 2)) return -1;
        a.act = (half_t*)dact.p;
    }
    if (launch_linear(s, a, pro, epi)) return -1;
    Q3_HIP(hipDeviceSynchronize(), -1);
    if (epi == EPI_STORE) Q3_HIP(hipMemcpy(y_or_h_io, dy.p, (size_t)M * N * 4, hipMemcpyDeviceToHost), -1);
    if (epi == EPI_RESID) {
        Q3_HIP(hipMemcpy(hp.data(), dy.p, hp.size() * 4, hipMemcpyDeviceToHost), -1);
        for (int m = 0; m < M; m++)
            for (int n = 0; n < N; n++) y_or_h_io[(size_t)m * N + n] = hp[frag_idx_host(m, n, N)];
        if (ssq_out) Q3_HIP(hipMemcpy(ssq_out, dso.p, (size_t)M * (N / 16) * 4, hipMemcpyDeviceToHost), -1);
    }
    if (epi == EPI_SWIGLU) {
        std::vector<uint16_t> ap((size_t)Mp * (N / 2));
        Q3_HIP(hipMemcpy(ap.data(), dact.p, ap.size() * 2, hipMemcpyDeviceToHost), -1);
        for (int m = 0; m < M; m++)
            for (int j = 0; j < N / 2; j++) act_out[(size_t)m * (N / 2) + j] = ap[frag_idx_host(m, j, N / 2)];
    }
    return 0;
}

// Time `iters` back-to-back launches of one linear shape over `n_copies` distinct weight copies
// (cold weights like the real layer walk).  Returns average microseconds per launch, <0 on error.
float q3t_bench_linear(int M, int N, int K, int pro, int epi, int nt, int n_copies, int iters) {
    hipStream_t s = nullptr;
    if (hipStreamCreate(&s) != hipSuccess) return -1.f;
    const size_t wbytes = (size_t)N * K * 2;
    const int Mreal = M;
    M = (M + 127) / 128 * 128;  // allocation padding; the launch uses Mreal rows
    DBuf dW, dx, dh, dssq, dg, dy, dso, dact;
    if (!dW.alloc(wbytes * n_copies)) return -1.f;
    hipMemset(dW.p, 0x11, wbytes * n_copies);
    dx.alloc((size_t)M * K * 2);
    hipMemset(dx.p, 0, (size_t)M * K * 2);
    dh.alloc((size_t)M * K * 4);
    hipMemset(dh.p, 0, (size_t)M * K * 4);
    dssq.alloc((size_t)M * (K / 16) * 4);
    hipMemset(dssq.p, 0, (size_t)M * (K / 16) * 4);
    dg.alloc((size_t)K * 4);
    hipMemset(dg.p, 0, (size_t)K * 4);
    dy.alloc((size_t)M * N * 4);
    dso.alloc((size_t)M * (N / 16) * 4);
    dact.alloc((size_t)M * (N / 2) * 2);
    LinArgs a;
    a.N = N;
    a.K = K;
    a.M = Mreal;
    a.nt = nt;
    a.x16 = (const half_t*)dx.p;
    a.ssq = (const float*)dssq.p;
    a.ssq_parts = K / 16;
    a.y = (float*)dy.p;
    a.ldy = N;
    a.h_out = (float*)dy.p;
    a.ssq_out = (float*)dso.p;
    a.act = (half_t*)dact.p;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = -n_copies; it < iters; it++) {
        if (it == 0) hipEventRecord(e0, s);
        a.wp = (const half_t*)((char*)dW.p + wbytes * (size_t)((it + n_copies) % n_copies));
        if (launch_linear(s, a, pro, epi)) return -1.f;
    }
    hipEventRecord(e1, s);
    if (hipStreamSynchronize(s) != hipSuccess) return -1.f;
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipStreamDestroy(s);
    return ms * 1000.f / iters;
}

}  // extern "C"

// Talker sampling kernel on host arrays: logits[V], the chronological list of already emitted
// code_0 (n_past entries), n_text.  Returns the emitted code, or -1 when the utterance ends
// (EOS / non-audio id / forced EOS); -1000 on error.
extern "C" int q3t_talker_sample(const float* logits, int V, const int* past, int n_past, int n_text, int ignore_eos) {
    DBuf dl, dpast, dn, dnt, ddone, dcodes, dnf, dpos0, dpos;
    int ring[32] = {0};
    for (int i = n_past > 32 ? n_past - 32 : 0; i < n_past; i++) ring[i & 31] = past[i];
    int zero = 0, pos0 = 7;
    if (!dl.up(logits, (size_t)V * 4) || !dpast.up(ring, sizeof(ring)) || !dn.up(&n_past, 4) || !dnt.up(&n_text, 4) ||
        !ddone.up(&zero, 4) || !dcodes.alloc(16 * 4) || !dnf.up(&zero, 4) || !dpos0.up(&pos0, 4) || !dpos.up(&zero, 4))
        return -1000;
    TalkerSampleArgs a;
    a.logits = (const float*)dl.p;
    a.V = V;
    a.R = 1;
    a.past = (int*)dpast.p;
    a.n_past = (int*)dn.p;
    a.n_text = (const int*)dnt.p;
    a.done = (int*)ddone.p;
    a.codes = (int*)dcodes.p;
    a.n_frames = (int*)dnf.p;
    a.frame_cap = 1;
    a.pos0 = (const int*)dpos0.p;
    a.pos = (int*)dpos.p;
    a.ignore_eos = ignore_eos;
    if (launch_talker_sample(nullptr, a)) return -1000;
    int code = -1000;
    if (hipDeviceSynchronize() != hipSuccess) return -1000;
    if (hipMemcpy(&code, dcodes.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1000;
    return code;
}

// ---- launch-boundary microbenchmark: a dependent chain of n small kernels, captured as a graph ----
namespace {
__global__ void chain_empty_kernel(float* buf) { (void)buf; }
// every thread reads what the previous kernel wrote (another workgroup's element) and writes its own
__global__ void chain_dep_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int hops) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = (i * 97 + 13) % n;
    float v = in[j];
    for (int h = 1; h < hops; h++) {  // extra dependent round trips
        j = ((int)(v * 0.f) + j * 31 + 7) % n;
        v += in[j];
    }
    out[i] = v + 1.0f;
}
}  // namespace

// kind 0: empty kernels; kind >=1: `kind` dependent global round trips per kernel.
// Returns average microseconds per kernel over `iters` graph replays of an n-kernel chain.
extern "C" float q3t_bench_chain(int kind, int blocks, int threads, int n_kernels, int iters, int use_graph) {
    hipStream_t s = nullptr;
    if (hipStreamCreate(&s) != hipSuccess) return -1.f;
    const int n = blocks * threads;
    DBuf a, b;
    if (!a.alloc((size_t)n * 4) || !b.alloc((size_t)n * 4)) return -1.f;
    hipMemset(a.p, 0, (size_t)n * 4);
    hipMemset(b.p, 0, (size_t)n * 4);
    auto chain = [&]() {
        for (int k = 0; k < n_kernels; k++) {
            float* in = (float*)((k & 1) ? b.p : a.p);
            float* out = (float*)((k & 1) ? a.p : b.p);
            if (kind == 0) hipLaunchKernelGGL(chain_empty_kernel, dim3(blocks), dim3(threads), 0, s, out);
            else hipLaunchKernelGGL(chain_dep_kernel, dim3(blocks), dim3(threads), 0, s, in, out, n, kind);
        }
    };
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    chain();
    hipStreamSynchronize(s);
    if (use_graph) {
        hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        chain();
        if (hipStreamEndCapture(s, &g) != hipSuccess) return -1.f;
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1.f;
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int it = 0; it < iters; it++) {
        if (use_graph) hipGraphLaunch(ge, s);
        else chain();
    }
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (ge) hipGraphExecDestroy(ge);
    if (g) hipGraphDestroy(g);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipStreamDestroy(s);
    return ms * 1000.f / ((float)iters * n_kernels);
}

// ---- CPU-side hook: parse weight files the way the loaders do (container, or the reference's .npz /
// .npy / .safetensors natively) and report what was found.  No GPU call.  Writes up to `cap` bytes of
// "name dtype ndim d0 d1 d2 d3 fnv1a64-of-bytes\n" lines; returns the tensor count or -1.
extern "C" int q3t_inspect_weights(const char* path, const char* aux_dir, char* out, int cap) {
    Pack p;
    if (!p.open_auto(path, aux_dir)) return -1;
    std::string s;
    for (const auto& kv : p.tensors) {
        const PackTensor& t = kv.second;
        unsigned long long h = 1469598103934665603ull;
        for (uint64_t i = 0; i < t.nbytes; i++) h = (h ^ t.data[i]) * 1099511628211ull;
        char line[256];
        snprintf(line, sizeof(line), "%s %u %u %llu %llu %llu %llu %llx\n", t.name.c_str(), t.dtype, t.ndim,
                 (unsigned long long)t.shape[0], (unsigned long long)t.shape[1], (unsigned long long)t.shape[2],
                 (unsigned long long)t.shape[3], h);
        s += line;
    }
    for (const auto& kv : p.meta) {
        char line[128];
        snprintf(line, sizeof(line), "meta %s %.17g\n", kv.first.c_str(), kv.second);
        s += line;
    }
    if (out && cap > 0) {
        const size_t n = s.size() < (size_t)cap - 1 ? s.size() : (size_t)cap - 1;
        memcpy(out, s.data(), n);
        out[n] = 0;
    }
    return (int)p.tensors.size();
}
