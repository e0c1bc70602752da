// q3_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the Qwen3-TTS talker /
// code-predictor decode path.  Wave = 64 lanes everywhere.
//
// Hot-path map (reference file:line each kernel stands in for):
//   linear_kernel   every projection llama_decode / the ONNX decode step runs
//                   (llama_wrapper.c:145, code_predictor_server.py:80), as a
//                   weight-streaming MFMA 16x16x32 f16 kernel with split-K over
//                   the waves of a workgroup; RMSNorm fused as prologue, residual
//                   add / SwiGLU fused as epilogue.
//   attn_kernel     per-head q/k RMSNorm + RoPE + KV-cache append + GQA decode
//                   attention (one workgroup per (row, kv head), K/V straight to
//                   VGPRs, scores staged in LDS).
//   talker_sample   llamacpp_talker_server.py:163-206 (greedy form) on device.
//   cp_argmax       code_predictor_server.py:87-92,128-137 (greedy) + next
//                   embedding gather, and tts_client.py:199-208 feedback sum.
#include "q3_kernels.h"

namespace q3 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- diagnostic timeline (only in the -DQ3_TIMELINE build; the product build compiles it away) ----
#ifdef Q3_TIMELINE
__device__ unsigned long long* g_tl = nullptr;   // [cap][2] start/end stamps (100 MHz wall clock)
__device__ unsigned int g_tl_idx = 0;
__device__ unsigned int g_tl_cap = 0;
__device__ int g_skip = 0;
__device__ unsigned long long* g_ph = nullptr;  // [cap][8] phase stamps of block 0 (shader clock)   // diagnostic: every kernel returns at once (measures the pure dispatch chain)
struct TlScope {
    unsigned int slot = 0xffffffffu;
    int id;
    __device__ __forceinline__ TlScope(int id_) : id(id_) {
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && g_tl) {
            slot = atomicAdd(&g_tl_idx, 1u);
            if (slot < g_tl_cap) g_tl[2 * slot] = (wall_clock64() << 8) | (unsigned)id;
        }
    }
    __device__ __forceinline__ ~TlScope() {
        if (slot < g_tl_cap) g_tl[2 * slot + 1] = wall_clock64();
    }
};
__device__ unsigned long long* g_tl2 = nullptr;  // [nodes][TL2_MAXB][2] per-block start/end stamps
__device__ int* g_tl2_kind = nullptr;             // [nodes] kernel kind id (the Q3_TL id) of each node
constexpr int TL2_MAXB = 1024;
struct TlScope2 {
    unsigned long long t0;
    int node, kind;
    __device__ __forceinline__ TlScope2(int n, int k) : node(n), kind(k) { t0 = wall_clock64(); }
    __device__ __forceinline__ ~TlScope2() {
        if (g_tl2 && node >= 0 && threadIdx.x == 0) {
            const unsigned b = blockIdx.x + gridDim.x * blockIdx.y;
            if (b < TL2_MAXB) {
                g_tl2[((size_t)node * TL2_MAXB + b) * 2] = t0;
                g_tl2[((size_t)node * TL2_MAXB + b) * 2 + 1] = wall_clock64();
            }
            if (b == 0 && g_tl2_kind) g_tl2_kind[node] = kind;
        }
    }
};
#define Q3_TL(id)      \
    if (g_skip) return; \
    TlScope2 tl_scope2_(a.tl_node, id); \
    TlScope tl_scope_(id)
#define Q3_PH(n)                                                                                       \
    do {                                                                                               \
        if (tl_scope_.slot < g_tl_cap && g_ph) g_ph[(size_t)tl_scope_.slot * 8 + (n)] = wall_clock64(); \
    } while (0)
#else
// Every kernel of the frame loop (and the prefill) raises its waves' issue priority at entry (s_setprio 3; -DQ3_WAVE_PRIO=0 builds
// without).  Alone on the chip it changes nothing (2.40 ms per frame step either way).  Beside the vocoder's one-workgroup-per-CU
// grid (voc_set_max_workgroups(-1)) a frame-loop wave shares its SIMD with one vocoder wave that issues MFMAs and LDS reads back to
// back; with priority the frame step beside the decode takes 3.0 instead of 3.2 ms and the benchmark's step 211 instead of 232 ms
// (round 2 tried the priority beside an UNCAPPED vocoder grid, where the frame loop's workgroups found no room at all: no effect).
#ifndef Q3_WAVE_PRIO
#define Q3_WAVE_PRIO 3
#endif
#if Q3_WAVE_PRIO > 0
#define Q3_TL(id) __builtin_amdgcn_s_setprio(Q3_WAVE_PRIO)
#else
#define Q3_TL(id)
#endif
#define Q3_PH(n)
#endif

// Kernel arguments passed as one struct live in the kernarg segment and are fetched with scalar loads where the code
// first needs them; behind control flow that becomes a CHAIN of s_load -> s_waitcnt round trips before the first
// global load of a kernel is even issued (six of them in attn_kernel, seen in the ISA).  Naming the scalar fields in
// one empty asm statement at the top makes the compiler fetch them all at once (adjacent fields merge into wide
// loads) and wait once.
#define Q3_FETCH_ARGS(...) asm volatile("" ::__VA_ARGS__)

// A per-row scalar (position, slot, frame counter ...) that a previous kernel wrote and every lane of the workgroup
// needs: read through the constant address space, i.e. with a SCALAR load (s_load_dword: straight into an SGPR, its
// own counter, ~half the latency of a vector load, and the vector loads that follow are issued without waiting for
// it).  Written by an earlier launch only, never by this one: the scalar cache is invalidated at every dispatch.
__device__ __forceinline__ int uniform_load(const int* p) {
    return *(const __attribute__((address_space(4))) int*)p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ half_t sat_half(float x) {
    return (half_t)fminf(fmaxf(x, -65504.f), 65504.f);
}
// RMSNorm folded around the GEMM (numerics contract): the PRODUCER of a residual row also writes
// xh = fp16((h * gamma_consumer) * NORM_PRE) -- the consumer's norm weight applied, a fixed power-of-two pre-scale
// keeping |h * gamma| up to 1e6 inside fp16 -- and the consumer multiplies its f32 accumulators by
// inv_rms(row) * NORM_POST.  The GEMM reads 2 bytes per activation instead of 4 and has no prologue to wait for.
constexpr float NORM_PRE = 0.0625f, NORM_POST = 16.0f;
// Stores of the decode kernels' outputs (a few hundred KB per launch, read by the NEXT launch on other XCDs): plain.
// -DQ3_NT_OUT makes them non-temporal (written through instead of left dirty for the end-of-kernel write-back) --
// measured in round 3 on MI355X: 2.41 -> 2.49 ms per frame at 32 rows, 2.14 -> 2.20 at one; rejected, kept as a probe.
#ifdef Q3_NT_OUT
#define Q3_OUT_STORE(v_, p_) __builtin_nontemporal_store((v_), (p_))
#else
#define Q3_OUT_STORE(v_, p_) (*(p_) = (v_))
#endif
__device__ __forceinline__ half_t pre_scaled(float h, float g) { return sat_half((h * g) * NORM_PRE); }

// Activations that feed a GEMM (residual stream h, attention output, SwiGLU output) live in MFMA
// A-fragment order, like the weights: 16-row x 32-k blocks (mt = m/16, kb = k/32), inside a block lane
// (m%16) + 16*((k%32)/8) owns 8 consecutive k.  A wave's fragment load is then 64 lanes x 16 B (fp16) or
// 2 x 64 x 16 B (f32) of CONTIGUOUS memory instead of 16 rows x 4 KB apart (measured: 3-4 us just to issue
// the uncoalesced loads of a 32-row tile).  Element (m, k) of a [rows][K] matrix sits at frag_idx(m, k, K);
// groups of 8 consecutive k (k % 8 == 0) stay contiguous, so float4 / 8-byte row accesses still work.
__device__ __forceinline__ size_t frag_idx(int m, int k, int K) {
    return ((((size_t)(m >> 4) * (K >> 5) + (k >> 5)) * 64 + (m & 15) + 16 * ((k >> 3) & 3)) << 3) + (k & 7);
}

// ---------------------------------------------------------------------------
// Weight repack: row-major [N][K] fp16 -> fragment order.  Element (n,k) goes
// to ((t*KB + kb)*64 + lane)*8 + j with t=n/16, kb=k/32, lane=(n%16)+16*((k%32)/8),
// j=k%8: one 16x32 weight block is one 1 KiB wave-wide 16-B-per-lane load whose
// lane contents are exactly the B operand of v_mfma_f32_16x16x32_f16.
// ---------------------------------------------------------------------------
__global__ void pack_linear_kernel(const half_t* __restrict__ src, int N, int K, half_t* __restrict__ dst,
                                   int tile_off, int tile_stride) {
    const int KB = K / 32;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-B group each
    size_t total = (size_t)(N / 16) * KB * 64;
    if (idx >= total) return;
    int lane = (int)(idx & 63);
    size_t blk = idx >> 6;
    int kb = (int)(blk % KB);
    int ts = (int)(blk / KB);
    int n = ts * 16 + (lane & 15);
    int k = kb * 32 + (lane >> 4) * 8;
    h8 v = *(const h8*)(src + (size_t)n * K + k);
    size_t td = (size_t)tile_off + (size_t)ts * tile_stride;
    *(h8*)(dst + ((td * KB + kb) * 64 + lane) * 8) = v;
}

int launch_pack_linear(hipStream_t s, const half_t* src, int N, int K, half_t* dst, int tile_off,
                       int tile_stride) {
    if (N % 16 || K % 32) {
        Q3_LOG("pack_linear: N=%d K=%d not multiples of 16/32", N, K);
        return -1;
    }
    size_t total = (size_t)(N / 16) * (K / 32) * 64;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(pack_linear_kernel, dim3(blocks), dim3(256), 0, s, src, N, K, dst, tile_off, tile_stride);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// linear_kernel<NB16, MT16, KBW, NW, PRO, EPI, NT>
//   workgroup = NW waves; it owns NB16 column tiles (16 weight rows each) for
//   MT16*16 batch rows; wave w owns k-blocks [w*KBW, (w+1)*KBW) (32 k each), so
//   K = NW*KBW*32.  All of a wave's weight fragments are requested up front
//   (KBW*NB16 16-B loads per lane in flight: the HBM stream), the A fragments
//   come from L2, partial tiles are summed across waves through LDS in a fixed
//   order (deterministic), then the epilogue runs on the summed tile.
// ---------------------------------------------------------------------------
// The first arguments are the ones the load addresses need: with -amdgpu-kernarg-preload-count=16 they
// arrive in SGPRs at wave launch instead of behind two dependent scalar-memory round trips.
template <int NB16, int MT16, int KBW, int NW, int PRO, int EPI, bool NT>
__global__ void __launch_bounds__(NW * 64)
    linear_kernel(const half_t* __restrict__ p_wp, const void* __restrict__ p_asrc, const float* __restrict__ p_ssq,
                  const float* __restrict__ p_gamma, void* __restrict__ p_out, int p_M, int p_N, int p_m_begin,
                  int p_swap, float p_eps, LinArgs a) {
    a.wp = p_wp;
    a.x16 = (const half_t*)p_asrc;
    a.ssq = p_ssq;
    a.gamma = p_gamma;
    a.y = (float*)p_out;
    a.h_out = (float*)p_out;
    a.act = (half_t*)p_out;
    a.M = p_M;
    a.N = p_N;
    a.m_begin = p_m_begin;
    a.swap_grid = p_swap;
    a.eps = p_eps;
    constexpr int MR = MT16 * 16, NB = NB16 * 16, NBP = NB + 4, KB = NW * KBW, K = KB * 32;
    constexpr int NTH = NW * 64;
    constexpr int NOUT = (EPI == EPI_SWIGLU) ? MR * NB / 2 : MR * NB;   // outputs of this workgroup
    constexpr int OPT = (NOUT + NTH - 1) / NTH;                         // outputs per thread
    constexpr int SQI = (MR * 16 + NTH - 1) / NTH;                      // ssq float4 groups per thread
    Q3_TL(10 + PRO * 4 + EPI);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    // many row blocks (prefill): the row block is the fast grid index, so the blocks that share a
    // weight tile run back to back and the tile is streamed from HBM once
    const int tile0 = (a.swap_grid ? blockIdx.y : blockIdx.x) * NB16;
    const int m0 = a.m_begin + (a.swap_grid ? blockIdx.x : blockIdx.y) * MR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;             // [NW][MR][NBP]
    float* inv_s = red + NW * MR * NBP;    // [MR]

    // ---- 1. every global load of the kernel is issued here: nothing later starts a second memory
    // round trip.  vmcnt retires in order, so the few operands that gate the prologue go first, the
    // HBM weight stream next, and the L2-resident fragments (needed only together with the weights) last.
    float4 sq[SQI];
    h8 af[MT16][KBW];
    float hold[OPT], gnext[OPT];
    // (a) tiny operands that gate the prologue / epilogue
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int i = 0; i < SQI; i++) {
            // float4 group: row = g4/16 (64 partials = 16 float4).  UNCONDITIONAL load (index clamped; rows beyond a.M are
            // allocated padding of the row block): a predicated load becomes a branch, and the compiler sinks the first
            // use of the loaded value -- with its s_waitcnt vmcnt(0) -- into that branch, ahead of the weight stream's
            // issue: a whole dependent memory round trip in front of every normed GEMM (seen in the ISA, round 2)
            const int g4 = tid + i * NTH, g4c = g4 < MR * 16 ? g4 : 0;
            sq[i] = *(const float4*)(a.ssq + (size_t)(m0 + g4c / 16) * 64 + (g4c & 15) * 4);
        }
    }
    if (EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < OPT; i++) {
            const int o = tid + i * NTH;
            const int m = m0 + o / NB;
            hold[i] = 0.f;
            gnext[i] = 0.f;
            if (o < NOUT && m < a.M) hold[i] = a.h_out[frag_idx(m, tile0 * 16 + (o % NB), a.N)];
            if (o < NOUT && a.gamma) gnext[i] = a.gamma[tile0 * 16 + (o % NB)];   // the consumer's norm weight
        }
    }
    // (b) / (c): the weight stream (HBM, the long pole: everything this wave will need, in flight at once) and the
    // activation fragments (L2; fp16 in either prologue -- PRO_NORM reads the producer's pre-scaled xh).
    h8 wf[NB16][KBW];
#pragma unroll
    for (int nb = 0; nb < NB16; nb++)
#pragma unroll
        for (int kbi = 0; kbi < KBW; kbi++) {
            const h8* p = (const h8*)(a.wp + (((size_t)(tile0 + nb) * KB + (size_t)w * KBW + kbi) * 64 + lane) * 8);
            wf[nb][kbi] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
    for (int kbi = 0; kbi < KBW; kbi++) {
        const int k0 = (w * KBW + kbi) * 32 + q * 8;
#pragma unroll
        for (int mt = 0; mt < MT16; mt++) {
            // rows beyond a.M are padding of the last 16-row block (allocated, never stored from)
            af[mt][kbi] = *(const h8*)(a.x16 + frag_idx(m0 + mt * 16 + c, k0, K));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // the epilogue's scalar arguments beyond the preloaded ones: fetched now, under the weight stream, not at the epilogue
    if (EPI == EPI_RESID) Q3_FETCH_ARGS("s"(a.ssq_out), "s"(a.xh_out));
    if (EPI == EPI_STORE) Q3_FETCH_ARGS("s"(a.ldy));
    if (PRO == PRO_NORM) Q3_FETCH_ARGS("s"(a.eps));
    Q3_PH(0);  // all loads issued

    // ---- 2. RMSNorm scale per row from the producer's 64 sum-of-squares partials (a.ssq_parts == 64): only the
    // epilogue needs it (the LDS reduction's barrier orders it) ----
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int i = 0; i < SQI; i++) {
            const int g4 = tid + i * NTH;
            float s = (sq[i].x + sq[i].y) + (sq[i].z + sq[i].w);
            s += __shfl_xor(s, 8, 16);
            s += __shfl_xor(s, 4, 16);
            s += __shfl_xor(s, 2, 16);
            s += __shfl_xor(s, 1, 16);
            if (g4 < MR * 16 && (g4 & 15) == 0) inv_s[g4 / 16] = (1.0f / sqrtf(s / (float)K + a.eps)) * NORM_POST;
        }
    }

    Q3_PH(1);  // prologue done (norm scale known, activations converted)
    // ---- 3. MFMA over this wave's K slice ----
    f4 acc[MT16][NB16];
#pragma unroll
    for (int mt = 0; mt < MT16; mt++)
#pragma unroll
        for (int nb = 0; nb < NB16; nb++) acc[mt][nb] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kbi = 0; kbi < KBW; kbi++)
#pragma unroll
        for (int mt = 0; mt < MT16; mt++)
#pragma unroll
            for (int nb = 0; nb < NB16; nb++)
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][kbi], wf[nb][kbi], acc[mt][nb], 0, 0, 0);

    asm volatile("" ::"v"(acc[0][0][0]));  // (diagnostic stamp below must not float above the MFMAs)
    Q3_PH(2);  // MFMAs done (weights landed)
    // ---- 4. partial tiles -> LDS.  D layout: col = lane&15, row = 4*(lane>>4) + reg.  (A layout with the four
    // accumulator registers of a lane adjacent -- one ds_write_b128 per tile instead of four scalar stores -- was
    // measured in round 3: 2.420 against 2.410 ms per frame at 32 rows, no gain.) ----
#define RED(w_, row_, col_) red[((w_) * MR + (row_)) * NBP + (col_)]
#pragma unroll
    for (int mt = 0; mt < MT16; mt++)
#pragma unroll
        for (int nb = 0; nb < NB16; nb++)
#pragma unroll
            for (int r = 0; r < 4; r++) RED(w, mt * 16 + 4 * q + r, nb * 16 + c) = acc[mt][nb][r];
    __syncthreads();
    Q3_PH(3);  // partials in LDS, barrier passed

    // ---- 5. fixed-order sum over waves + epilogue ----
    if (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < OPT; i++) {
            const int o = tid + i * NTH;
            if (o < NOUT) {   // wave-uniform (NOUT and NTH are multiples of 64)
                const int mr = o / NB, n = o % NB;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ww++) v += RED(ww, mr, n);
                const int m = m0 + mr;
                const int ng = tile0 * 16 + n;
                const bool ok = m < a.M;
                if (PRO == PRO_NORM) v *= inv_s[mr];
                if (EPI == EPI_STORE) {
                    if (ok) Q3_OUT_STORE(v, &a.y[(size_t)m * a.ldy + ng]);
                } else {
                    const float hn = ok ? hold[i] + v : 0.f;
                    if (ok) Q3_OUT_STORE(hn, &a.h_out[frag_idx(m, ng, a.N)]);
                    if (ok && a.xh_out) Q3_OUT_STORE(pre_scaled(hn, gnext[i]), &a.xh_out[frag_idx(m, ng, a.N)]);
                    float s = hn * hn;
                    s += __shfl_xor(s, 8, 16);
                    s += __shfl_xor(s, 4, 16);
                    s += __shfl_xor(s, 2, 16);
                    s += __shfl_xor(s, 1, 16);
                    if (ok && (n & 15) == 0) Q3_OUT_STORE(s, &a.ssq_out[(size_t)m * (a.N / 16) + (ng >> 4)]);
                }
            }
        }
    } else {  // EPI_SWIGLU: tile 2i = gate rows, tile 2i+1 = the matching up rows
        constexpr int NH = NB / 2;
#pragma unroll
        for (int i = 0; i < OPT; i++) {
            const int o = tid + i * NTH;
            if (o < NOUT) {
                const int mr = o / NH, j = o % NH;
                const int ii = j >> 4, cc = j & 15;
                float g = 0.f, u = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ww++) {
                    g += RED(ww, mr, (2 * ii) * 16 + cc);
                    u += RED(ww, mr, (2 * ii + 1) * 16 + cc);
                }
                const int m = m0 + mr;
                if (PRO == PRO_NORM) {
                    g *= inv_s[mr];
                    u *= inv_s[mr];
                }
                if (m < a.M) {
                    const float sg = __fdividef(g, 1.0f + __expf(-g));   // hardware exp/rcp: ~1e-6 relative, far below the fp16 rounding that follows
                    Q3_OUT_STORE(sat_half(sg * u), &a.act[frag_idx(m, (tile0 / NB16) * NH + j, a.N / 2)]);
                }
            }
        }
    }
#undef RED
}

template <int NB16, int MT16, int KBW, int NW, int PRO, int EPI, bool NT>
static int launch_linear_nt(hipStream_t s, const LinArgs& a) {
    constexpr int MR = MT16 * 16, NBP = NB16 * 16 + 4;
    constexpr size_t lds = (size_t)NW * MR * NBP * 4 + MR * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (lds > 48 * 1024)
            Q3_HIP(hipFuncSetAttribute((const void*)linear_kernel<NB16, MT16, KBW, NW, PRO, EPI, NT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), -1);
        attr_set = true;
    }
    LinArgs b = a;
    b.tl_node = tl_next_node();
    const unsigned nt_ = a.N / (16 * NB16), nr_ = (a.M - a.m_begin + MR - 1) / MR;
    b.swap_grid = nr_ > 2 ? 1 : 0;
    dim3 grid(b.swap_grid ? nr_ : nt_, b.swap_grid ? nt_ : nr_);
    const void* asrc = (const void*)b.x16;
    void* outp = EPI == EPI_STORE ? (void*)b.y : EPI == EPI_RESID ? (void*)b.h_out : (void*)b.act;
    hipLaunchKernelGGL((linear_kernel<NB16, MT16, KBW, NW, PRO, EPI, NT>), grid, dim3(NW * 64), lds, s, b.wp, asrc, b.ssq,
                       b.gamma, outp, b.M, b.N, b.m_begin, b.swap_grid, b.eps, b);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// Tile of a workgroup in the tiled GEMMs.  Workgroups are dealt to the 8 XCDs round-robin (linear id % 8), and every
// XCD has its own L2: the R row tiles that share one weight column tile must sit on ONE XCD, or that tile is pulled
// from HBM / Infinity Cache eight times (measured: the 971-row q/k/v GEMM moved ~80 MB for 8.4 MB of weights).
// XCD x therefore owns column tiles x, x + 8, ...; its j-th workgroup is (row tile j % R, column tile x + 8 * (j / R)).
__device__ __forceinline__ void gemm_tile_of_block(int R, int T, int& row_tile, int& col_tile) {
    const int L = blockIdx.x;
    if (T % 8 == 0) {
        const int x = L & 7, j = L >> 3;
        row_tile = j % R;
        col_tile = x + 8 * (j / R);
    } else {
        row_tile = L % R;
        col_tile = L / R;
    }
}

// epilogue shared by the tiled GEMM kernels, straight from the accumulators
template <int BM, int BN, int PRO, int EPI>
__device__ __forceinline__ void gemm_epilogue(const LinArgs& a, f4 (&acc)[BM / 32][BN / 32], const float* post, int m0,
                                              int tile0, int wm, int wn, int q, int c) {
    constexpr int WM = BM / 32, WN = BN / 32;
    // ---- epilogue straight from the accumulators.  D layout: column = lane & 15, row = 4 * (lane >> 4) + reg ----
#pragma unroll
    for (int i = 0; i < WM; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int mr = (wm * WM + i) * 16 + 4 * q + r;
            const int m = m0 + mr;
            const bool ok = m < a.M;
            const float ps = PRO == PRO_NORM ? post[mr] : 1.0f;
            if (EPI == EPI_SWIGLU) {
#pragma unroll
                for (int j = 0; j < WN; j += 2) {
                    const float g = acc[i][j][r] * ps, u = acc[i][j + 1][r] * ps;
                    const float sg = __fdividef(g, 1.0f + __expf(-g));
                    const int jn = ((tile0 + wn * WN + j) >> 1) * 16 + c;     // column of act[M][N/2]
                    if (ok) a.act[frag_idx(m, jn, a.N / 2)] = sat_half(sg * u);
                }
            } else {
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    const int ng = (tile0 + wn * WN + j) * 16 + c;
                    const float v = acc[i][j][r] * ps;
                    if (EPI == EPI_STORE) {
                        if (ok) a.y[(size_t)m * a.ldy + ng] = v;
                    } else {
                        const size_t hi = frag_idx(m, ng, a.N);
                        const float hn = ok ? a.h_out[hi] + v : 0.f;
                        if (ok) a.h_out[hi] = hn;
                        if (ok && a.xh_out) a.xh_out[hi] = pre_scaled(hn, a.gamma[ng]);
                        float s2 = hn * hn;
                        s2 += __shfl_xor(s2, 8, 16);
                        s2 += __shfl_xor(s2, 4, 16);
                        s2 += __shfl_xor(s2, 2, 16);
                        s2 += __shfl_xor(s2, 1, 16);
                        if (ok && c == 0) a.ssq_out[(size_t)m * (a.N / 16) + (ng >> 4)] = s2;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// gemm_kernel<BM, BN, PRO, EPI> -- the same linear layers for MANY rows (prefill: llama_wrapper.c:125-163 with
// n_tokens = n_text + 9, all utterances of a batch in one ragged pass).  A real tiled GEMM: workgroup = 4 waves
// (2 x 2), tile BM x BN, each wave (BM/2) x (BN/2) as 16x16x32 f16 MFMA fragments; per stage 64 k of the A tile
// (activations, already in fragment order) and of the B tile (weights, fragment order) go global -> registers ->
// LDS, double buffered, one barrier per stage; every fragment is one conflict-free ds_read_b128.  The operands,
// prologue/epilogue semantics and output layouts are exactly linear_kernel's, so the two are interchangeable
// per launch (launch_linear picks by row count).
// ---------------------------------------------------------------------------
template <int BM, int BN, int KS, int PRO, int EPI>
__global__ void __launch_bounds__(256) gemm_kernel(LinArgs a) {
    // KS = k-blocks (32 k each) per stage.  A stage is bounded by the latency of its own global loads (one stage is
    // prefetched while the previous one computes: ~2000 cycles against ~128 * KS cycles of MFMA work), so stages are
    // made as deep as LDS allows: 2 buffers x (BM + BN) / 16 x KS KiB = 128 KiB.
    constexpr int RA = BM / 16, RB = BN / 16;                   // row / column fragments of the tile
    constexpr int NF = (RA + RB) * KS, FPW = NF / 4;            // fragments per stage, per wave to fetch
    constexpr int WM = BM / 32, WN = BN / 32;                   // fragments per wave (rows, columns)
    static_assert(NF % 4 == 0 && WN % 2 == 0, "tile shape");
    Q3_TL(20 + PRO * 4 + EPI);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1, q = lane >> 4, c = lane & 15;
    int row_tile, col_tile;
    gemm_tile_of_block((a.M - a.m_begin + BM - 1) / BM, a.N / BN, row_tile, col_tile);
    const int m0 = a.m_begin + row_tile * BM;
    const int tile0 = col_tile * RB;
    const int KB = a.K >> 5, NST = KB / KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h8* lds = (h8*)smem;                                        // [2][NF][64] fragments
    __shared__ float post[BM];
    if (PRO == PRO_NORM) {
        for (int r = tid; r < BM; r += 256) {
            const int m = m0 + r;
            float sum = 0.f;
            if (m < a.M) {
                const float4* sp = (const float4*)(a.ssq + (size_t)m * 64);
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const float4 v = sp[p];
                    sum += (v.x + v.y) + (v.z + v.w);
                }
            }
            post[r] = (1.0f / sqrtf(sum / (float)a.K + a.eps)) * NORM_POST;
        }
    }
    const size_t arow = (size_t)(m0 >> 4) * KB;
    auto frag_ptr = [&](int f, int kb0) -> const h8* {
        if (f < RA * KS) {
            const int rb = f / KS, kk = f % KS;
            return (const h8*)(a.x16 + ((arow + (size_t)rb * KB + kb0 + kk) * 64 + lane) * 8);
        }
        const int g = f - RA * KS, t = g / KS, kk = g % KS;
        return (const h8*)(a.wp + (((size_t)(tile0 + t) * KB + kb0 + kk) * 64 + lane) * 8);
    };
    h8 st[FPW];
#pragma unroll
    for (int i = 0; i < FPW; i++) st[i] = *frag_ptr(w * FPW + i, 0);
#pragma unroll
    for (int i = 0; i < FPW; i++) lds[(w * FPW + i) * 64 + lane] = st[i];
    f4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int ks = 0; ks < NST; ks++) {
        const bool more = ks + 1 < NST;
        if (more) {
#pragma unroll
            for (int i = 0; i < FPW; i++) st[i] = *frag_ptr(w * FPW + i, (ks + 1) * KS);
        }
        const h8* cur = lds + (size_t)(ks & 1) * NF * 64;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            h8 af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; i++) af[i] = cur[((wm * WM + i) * KS + kk) * 64 + lane];
#pragma unroll
            for (int j = 0; j < WN; j++) bf[j] = cur[(RA * KS + (wn * WN + j) * KS + kk) * 64 + lane];
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int j = 0; j < WN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            h8* nxt = lds + (size_t)((ks + 1) & 1) * NF * 64;
#pragma unroll
            for (int i = 0; i < FPW; i++) nxt[(w * FPW + i) * 64 + lane] = st[i];
        }
        __syncthreads();
    }
    gemm_epilogue<BM, BN, PRO, EPI>(a, acc, post, m0, tile0, wm, wn, q, c);
}

// ---------------------------------------------------------------------------
// gemm_glds_kernel<BM, BN, PRO, EPI> -- the same tile GEMM with its operand stream as LDS-DMA
// (global_load_lds_dwordx4: global -> LDS with no VGPR hop) into a ring of NBUF stage buffers, PF stages in flight.
// A stage of gemm_kernel is bounded by the latency of its own loads (~2000 cycles against ~250 cycles of MFMA work
// per 64 k): with M ~ 1000 rows the grid is one workgroup per CU, nothing else hides it.  Here three stages
// (96 KiB per CU for the 128 x 128 tile) are always in flight; a wave waits only for ITS OWN pieces of the stage it
// is about to read (counted s_waitcnt vmcnt), then one raw s_barrier makes the other waves' pieces visible.  The
// fragment order of both operands in global memory is already the lane-linear 1 KiB image LDS-DMA writes.
// ---------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int NBUF, int PRO, int EPI, int KS = 2>
__global__ void __launch_bounds__(512) gemm_glds_kernel(LinArgs a) {
    // 8 waves: two groups of 4 (2 x 2 over the tile).  A stage holds KS = 2 k-blocks (64 k); group g multiplies
    // k-block g of every stage, so each SIMD carries two waves whose LDS reads and MFMAs interleave (with one wave
    // per SIMD they alternate: the tile ran at ~2800 cycles per stage against 512 of MFMA work).  The two groups'
    // accumulators meet once, through LDS, in a fixed order (even k-blocks + odd k-blocks).
    constexpr int RA = BM / 16, RB = BN / 16;                   // fragments of the tile; KS k-blocks (32 k) per stage, KS / 2 per group
    static_assert(KS % 2 == 0, "two wave groups share a stage's k-blocks");
    constexpr int NF = (RA + RB) * KS, FPW = NF / 8;            // fragments per stage; per wave to fetch
    constexpr int WM = BM / 32, WN = BN / 32;
    constexpr int PF = NBUF - 1;                                // ring size NBUF; stages in flight
    static_assert(NF % 8 == 0 && WN % 2 == 0, "tile shape");
    static_assert((size_t)NBUF * NF * 1024 >= (size_t)4 * WM * WN * 4 * 64 * 4, "the ring must hold one group's accumulators");
    Q3_TL(24 + PRO * 4 + EPI);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int grp = w >> 2, wq = w & 3;
    const int wm = wq >> 1, wn = wq & 1, q = lane >> 4, c = lane & 15;
    int row_tile, col_tile;
    gemm_tile_of_block((a.M - a.m_begin + BM - 1) / BM, a.N / BN, row_tile, col_tile);
    const int m0 = a.m_begin + row_tile * BM;
    const int tile0 = col_tile * RB;
    const int KB = a.K >> 5, NST = KB / KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ONE LDS object: [NBUF][NF][64] fragments, then post[BM]
    h8* lds = (h8*)smem;
    float* post = (float*)(smem + (size_t)NBUF * NF * 1024);
    const size_t arow = (size_t)(m0 >> 4) * KB;
    auto issue = [&](int st) {       // this wave's FPW pieces of stage st -> ring buffer st % NBUF
        h8* buf = lds + (size_t)(st % NBUF) * NF * 64;
        const int kb0 = st * KS;
#pragma unroll
        for (int i = 0; i < FPW; i++) {
            const int f = w * FPW + i;
            const h8* src;
            if (f < RA * KS) {
                const int rb = f / KS, kk = f % KS;
                src = (const h8*)(a.x16 + ((arow + (size_t)rb * KB + kb0 + kk) * 64 + lane) * 8);
            } else {
                const int g = f - RA * KS, t = g / KS, kk = g % KS;
                src = (const h8*)(a.wp + (((size_t)(tile0 + t) * KB + kb0 + kk) * 64 + lane) * 8);
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + f * 64 + lane), 16, 0, 0);
        }
    };
    // the norm scales first (ordinary loads: the compiler drains them with vmcnt(0) here, before any LDS-DMA exists)
    if (PRO == PRO_NORM) {
        for (int r = tid; r < BM; r += 512) {
            const int m = m0 + r;
            float sum = 0.f;
            if (m < a.M) {
                const float4* sp = (const float4*)(a.ssq + (size_t)m * 64);
#pragma unroll
                for (int p = 0; p < 16; p++) {
                    const float4 v = sp[p];
                    sum += (v.x + v.y) + (v.z + v.w);
                }
            }
            post[r] = (1.0f / sqrtf(sum / (float)a.K + a.eps)) * NORM_POST;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < PF; st++)
        if (st < NST) issue(st);
    f4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < NST; ks++) {
        // stages issued beyond ks at this point: min(PF - 1, NST - 1 - ks); wait until stage ks (mine) has landed
        const int ahead = NST - 1 - ks;
        if (ahead >= PF - 1) wait_vmcnt<(PF - 1) * FPW>();
        else if (PF > 2 && ahead == 1) wait_vmcnt<FPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();          // everyone's pieces of stage ks are in LDS; everyone is done with stage ks - 1
        asm volatile("" ::: "memory");
        if (ks + PF < NST) issue(ks + PF);     // into the buffer stage ks - 1 occupied
        const h8* cur = lds + (size_t)(ks % NBUF) * NF * 64;
#pragma unroll
        for (int kk = 0; kk < KS / 2; kk++) {
            const int kb = grp * (KS / 2) + kk;      // this group's k-blocks of the stage
            h8 af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; i++) af[i] = cur[((wm * WM + i) * KS + kb) * 64 + lane];
#pragma unroll
            for (int j = 0; j < WN; j++) bf[j] = cur[(RA * KS + (wn * WN + j) * KS + kb) * 64 + lane];
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int j = 0; j < WN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // ---- the odd k-blocks' accumulators (group 1) join the even ones (group 0) through the now idle ring ----
    __builtin_amdgcn_s_barrier();              // every wave is done reading the last stage
    asm volatile("" ::: "memory");
    f4* xch = (f4*)smem;                       // [4 waves][WM * WN][64 lanes]
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < WM; i++)
#pragma unroll
            for (int j = 0; j < WN; j++) xch[((size_t)wq * WM * WN + i * WN + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) {
            const f4 o = xch[((size_t)wq * WM * WN + i * WN + j) * 64 + lane];
            acc[i][j] += o;
        }
    gemm_epilogue<BM, BN, PRO, EPI>(a, acc, post, m0, tile0, wm, wn, q, c);
}

// 1 (default): operand stream by LDS-DMA into a 4-stage ring (gemm_glds_kernel); 0: register-staged double buffer
static int g_gemm_glds = getenv("Q3_GEMM_GLDS") ? atoi(getenv("Q3_GEMM_GLDS")) : 1;
int set_gemm_glds(int on) { g_gemm_glds = on; return 0; }

template <int BM, int BN, int PRO, int EPI>
static int launch_gemm_t(hipStream_t s, const LinArgs& a) {
    constexpr int KS = 1024 / (BM + BN);    // 128 x 128: 4 k-blocks (128 k) per stage; 64 x 64: 8
    constexpr size_t lds = (size_t)2 * ((BM + BN) / 16) * KS * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        if (lds > 48 * 1024)
            Q3_HIP(hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, KS, PRO, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), -1);
        attr_set = true;
    }
    LinArgs b = a;
    b.tl_node = tl_next_node();
    dim3 grid(((a.M - a.m_begin + BM - 1) / BM) * (a.N / BN));   // 1-D: the kernels map it to tiles XCD by XCD
    if (g_gemm_glds) {
#ifndef Q3_GEMM_KS_SMALL
#define Q3_GEMM_KS_SMALL 4
#endif
        // (round 3: a 9-deep ring for the 64 x 64 tiles -- 128 KB in flight instead of 48 -- changed nothing, 3.17 vs 3.10 ms
        // per 971-row prefill: those tiles are bound by the per-stage barrier, not by bytes in flight; hence KS below)
        constexpr int NBUF = (BM + BN) > 256 ? 3 : 4;
        constexpr int KS2 = (BM + BN) <= 128 ? Q3_GEMM_KS_SMALL : 2;                        // k-blocks (32 k) per stage
        constexpr size_t lds2 = (size_t)NBUF * ((BM + BN) / 16) * KS2 * 1024 + BM * 4;     // ring + post[BM]
        static bool attr2 = false;
        if (!attr2) {
            Q3_HIP(hipFuncSetAttribute((const void*)gemm_glds_kernel<BM, BN, NBUF, PRO, EPI, KS2>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2), -1);
            attr2 = true;
        }
        if ((a.K >> 5) % KS2) {
            Q3_LOG("launch_gemm: K=%d is no multiple of %d", a.K, 32 * KS2);
            return -1;
        }
        hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, NBUF, PRO, EPI, KS2>), grid, dim3(512), lds2, s, b);
    } else {
        hipLaunchKernelGGL((gemm_kernel<BM, BN, KS, PRO, EPI>), grid, dim3(256), lds, s, b);
    }
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// rows beyond this use the tiled GEMM (buffers are padded to GEMM_ROW_PAD rows)
static int g_gemm_min_rows = 65;
int set_gemm_min_rows(int n) { g_gemm_min_rows = n; return 0; }

static int launch_gemm(hipStream_t s, const LinArgs& a, int pro, int epi) {
    if (a.m_begin % 16 || a.K % 256 || a.N % 128 || (pro == PRO_NORM && a.ssq_parts != 64)) {
        Q3_LOG("launch_gemm: unsupported shape N=%d K=%d m_begin=%d", a.N, a.K, a.m_begin);
        return -1;
    }
    const bool narrow = a.N <= 2048;   // o / down (N = 1024): 64 x 64 tiles, else too few workgroups to fill 256 CUs
    if (pro == PRO_NORM && epi == EPI_STORE) return launch_gemm_t<128, 128, PRO_NORM, EPI_STORE>(s, a);
    if (pro == PRO_NORM && epi == EPI_SWIGLU) {
        // gate/up, N = 6144: 192-column tiles = 32 x 8 = 256 workgroups for ~1000 rows, one per CU, one round
        // (128-column tiles: 384 workgroups, two rounds)
        if (g_gemm_glds && a.N % 192 == 0 && (a.N / 192) % 8 == 0) return launch_gemm_t<128, 192, PRO_NORM, EPI_SWIGLU>(s, a);
        return launch_gemm_t<128, 128, PRO_NORM, EPI_SWIGLU>(s, a);
    }
    if (pro == PRO_F16 && epi == EPI_RESID)
        return narrow ? launch_gemm_t<64, 64, PRO_F16, EPI_RESID>(s, a) : launch_gemm_t<128, 128, PRO_F16, EPI_RESID>(s, a);
    if (pro == PRO_F16 && epi == EPI_STORE)
        return narrow ? launch_gemm_t<64, 64, PRO_F16, EPI_STORE>(s, a) : launch_gemm_t<128, 128, PRO_F16, EPI_STORE>(s, a);
    Q3_LOG("launch_gemm: no instantiation for pro=%d epi=%d", pro, epi);
    return -1;
}

// (round 3 probe: at <= 4 rows -- one utterance -- only the lanes of rows 0-3 fetching activation fragments, compiled in as a
// template variant of linear_kernel and linear_narrow_kernel: 2.145-2.163 against 2.145-2.153 ms per frame, nothing; the
// activation bytes are not what a one-row pass waits for)
template <int NB16, int MT16, int KBW, int NW, int PRO, int EPI>
static int launch_linear_t(hipStream_t s, const LinArgs& a) {
    return a.nt ? launch_linear_nt<NB16, MT16, KBW, NW, PRO, EPI, true>(s, a)
                : launch_linear_nt<NB16, MT16, KBW, NW, PRO, EPI, false>(s, a);
}

// (KBW, NW) per K; overridable for tuning through q3_set_linear_tuning().
static int g_wide_tiles = 0;  // measured no gain on MI355X (4.06 vs 4.12 ms/frame at 32 rows)
int set_linear_wide_tiles(int on) { g_wide_tiles = on; return 0; }
static int g_split_rows_narrow = 1;
int set_linear_split_rows(int on) { g_split_rows_narrow = on; return 0; }
// k-blocks per wave by [K = 1024, 2048, 3072][rows <= 16, <= 32, more]
static int g_tune_kbw[3][3] = {{4, 4, 4}, {16, 8, 8}, {12, 6, 6}};
int set_linear_tuning(int K, int mt16, int kbw) {
    int i = K == 1024 ? 0 : K == 2048 ? 1 : K == 3072 ? 2 : -1;
    int j = mt16 == 1 ? 0 : mt16 == 2 ? 1 : mt16 == 4 ? 2 : -1;
    if (i < 0 || j < 0) return -1;
    g_tune_kbw[i][j] = kbw;
    return 0;
}

// ---------------------------------------------------------------------------
// linear_narrow_kernel<KBW, NW, NT>: the N = 1024 projections (o: K = 2048, down: K = 3072; fp16 input, residual
// epilogue) at <= 64 rows, as their own kernel.  The body of a weight-streaming launch is bound by the bytes each CU
// pulls through its L1 (~77 GB/s per CU), and N = 1024 has only 64 column tiles: with 16-row groups
// a 32-row pass runs 128 workgroups of (16 rows + 16 weight rows) x K x 2 B = 128 / 192 KB each on half the chip.
// Here a workgroup owns 8 rows x 16 columns: 256 workgroups at 32 rows, (8 + 16) x K x 2 B = 96 / 144 KB each, one
// per CU -- and at <= 8 rows (one utterance) the 64 workgroups fetch 8 activation rows instead of 16.  The MFMA is
// still 16 x 16 x 32: lanes of the other 8 rows load nothing (exec-masked, zeros) and their accumulator rows are
// dropped.  Everything else is linear_kernel's EPI_RESID path: every load issued up front, split-K over the NW waves,
// fixed-order LDS reduction, h += acc, 16-column sum-of-squares partials and the consumer's pre-scaled xh.
// Workgroups of one column tile have equal blockIdx.x % 8 (the grid's x extent is 64): one XCD under round-robin
// placement, so the tile's weights come from HBM once and from that L2 for the other row groups.
// ---------------------------------------------------------------------------
template <int KBW, int NW, bool NT>
__global__ void __launch_bounds__(NW * 64)
    linear_narrow_kernel(const half_t* __restrict__ p_wp, const half_t* __restrict__ p_x16, float* __restrict__ p_h,
                         const float* __restrict__ p_gamma, float* __restrict__ p_ssq_out, half_t* __restrict__ p_xh_out,
                         int p_M, int p_m_begin, int tl_node) {
    constexpr int KB = NW * KBW, K = KB * 32, N = 1024, RG = 8;
#ifdef Q3_TIMELINE
    LinArgs a;
    a.tl_node = tl_node;
#endif
    Q3_TL(10 + PRO_F16 * 4 + EPI_RESID);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int tile0 = blockIdx.x;
    const int m0 = p_m_begin + blockIdx.y * RG;          // first row of this group (p_m_begin % 16 == 0)
    const int blk0 = m0 & ~15, half = (m0 >> 3) & 1;     // its 16-row fragment block, and which half of it
    // partial tiles: [wave][4-row group of the 8 rows][column][4 rows] -- a lane's four accumulator registers are
    // four consecutive floats, one ds_write_b128 (as four scalar stores at a row stride of 20 floats the ROCm 7.2
    // compiler merged two of them into a ds_write2_b32 with a wrong first offset -- 5 instead of 20 dwords -- when
    // the accumulators lived in AGPRs: seen in the ISA and in the results of the KBW = 16, NW = 4 instantiation)
    __shared__ __attribute__((aligned(16))) float red[NW * 2 * 16 * 4];
    // ---- every global load of the kernel, up front (epilogue operands first: vmcnt retires in order) ----
    const int mr = tid >> 4, n = tid & 15;               // epilogue mapping: threads 0..127 = 8 rows x 16 columns
    const int m = m0 + mr, ng = tile0 * 16 + n;
    const bool epi_thread = tid < RG * 16, ok = epi_thread && m < p_M;
    float hold = 0.f, gnext = 0.f;
    if (ok) hold = p_h[frag_idx(m, ng, N)];
    if (epi_thread && p_gamma) gnext = p_gamma[ng];
    h8 wf[KBW], af[KBW];
#pragma unroll
    for (int kbi = 0; kbi < KBW; kbi++) {
        const h8* p = (const h8*)(p_wp + (((size_t)tile0 * KB + (size_t)w * KBW + kbi) * 64 + lane) * 8);
        wf[kbi] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    // Lane c holds row blk0 + c of the 16-row block; the lanes of the block's other half fetch nothing.  Their A rows
    // may hold anything (a row of D depends on its own row of A only, and their accumulator rows are dropped), so
    // ONE exec-masked region covers all KBW loads and nothing is merged afterwards.
    const bool live = (c >> 3) == half;
    if (live) {
#pragma unroll
        for (int kbi = 0; kbi < KBW; kbi++)
            af[kbi] = *(const h8*)(p_x16 + frag_idx(blk0 + c, (w * KBW + kbi) * 32 + q * 8, K));
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3_PH(0);  // all loads issued
    f4 acc = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kbi = 0; kbi < KBW; kbi++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kbi], wf[kbi], acc, 0, 0, 0);
    Q3_PH(2);  // MFMAs done (weights landed)
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg: this group's rows 8 half .. 8 half + 7 sit in q = 2 half, 2 half + 1
    if ((q >> 1) == half) *(f4*)(red + ((w * 2 + (q & 1)) * 16 + c) * 4) = acc;
    __syncthreads();
    Q3_PH(3);
    if (epi_thread) {      // wave-uniform (128 threads = waves 0 and 1)
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ww++) v += red[((ww * 2 + (mr >> 2)) * 16 + n) * 4 + (mr & 3)];
        const float hn = ok ? hold + v : 0.f;
        if (ok) Q3_OUT_STORE(hn, &p_h[frag_idx(m, ng, N)]);
        if (ok && p_xh_out) Q3_OUT_STORE(pre_scaled(hn, gnext), &p_xh_out[frag_idx(m, ng, N)]);
        float s = hn * hn;
        s += __shfl_xor(s, 8, 16);
        s += __shfl_xor(s, 4, 16);
        s += __shfl_xor(s, 2, 16);
        s += __shfl_xor(s, 1, 16);
        if (ok && n == 0) Q3_OUT_STORE(s, &p_ssq_out[(size_t)m * (N / 16) + tile0]);
    }
}

static int g_narrow8 = getenv("Q3_LINEAR_NARROW8") ? atoi(getenv("Q3_LINEAR_NARROW8")) : 1;
int set_linear_narrow8(int on) { g_narrow8 = on; return 0; }

template <int KBW, int NW>
static int launch_linear_narrow_t(hipStream_t s, const LinArgs& a) {
    const int groups = (a.M - a.m_begin + 7) / 8;
    const int node = tl_next_node();
    if (a.nt)
        hipLaunchKernelGGL((linear_narrow_kernel<KBW, NW, true>), dim3(64, groups), dim3(NW * 64), 0, s, a.wp, a.x16, a.h_out, a.gamma,
                           a.ssq_out, a.xh_out, a.M, a.m_begin, node);
    else
        hipLaunchKernelGGL((linear_narrow_kernel<KBW, NW, false>), dim3(64, groups), dim3(NW * 64), 0, s, a.wp, a.x16, a.h_out, a.gamma,
                           a.ssq_out, a.xh_out, a.M, a.m_begin, node);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

#define Q3_LIN_CASE(NB16_, MT16_, KBW_, NW_, PRO_, EPI_)                                   \
    if (nb16 == NB16_ && mt16 == MT16_ && kbw == KBW_ && nw == NW_ && pro == PRO_ && epi == EPI_) \
        return launch_linear_t<NB16_, MT16_, KBW_, NW_, PRO_, EPI_>(s, a);

#define Q3_LIN_MT(NB16_, KBW_, NW_, PRO_, EPI_) \
    Q3_LIN_CASE(NB16_, 1, KBW_, NW_, PRO_, EPI_) \
    Q3_LIN_CASE(NB16_, 2, KBW_, NW_, PRO_, EPI_) \
    Q3_LIN_CASE(NB16_, 4, KBW_, NW_, PRO_, EPI_)

int launch_linear(hipStream_t s, const LinArgs& a, int pro, int epi) {
    const int rows = a.M - a.m_begin;
    if (rows <= 0) return 0;
    if (rows >= g_gemm_min_rows) return launch_gemm(s, a, pro, epi);
    const int K = a.K;
    int ki = K == 1024 ? 0 : K == 2048 ? 1 : K == 3072 ? 2 : -1;
    if (ki < 0 || a.N % 32) {
        Q3_LOG("launch_linear: unsupported shape N=%d K=%d", a.N, K);
        return -1;
    }
    if (g_narrow8 && pro == PRO_F16 && epi == EPI_RESID && a.N == 1024 && a.m_begin % 16 == 0 && a.ssq_out) {
        // o / down: 8-row groups, one workgroup per CU at 32 rows (linear_narrow_kernel)
        // (waves per workgroup probed on MI355X: 4 / 8 / 16 for K = 2048 and 8 / 16 / 4 for K = 3072 all within 0.3 %)
        if (K == 2048) return launch_linear_narrow_t<16, 4>(s, a);
        if (K == 3072) return launch_linear_narrow_t<12, 8>(s, a);
    }
    int mt16 = rows <= 16 ? 1 : rows <= 32 ? 2 : 4;
    // narrow outputs (o/down, N = 1024) have only N/16 = 64 column tiles: split the rows over two
    // workgroups per tile instead (same XCD under round-robin placement, the weights come once from HBM)
    if (rows > 16 && rows <= 32 && ((g_split_rows_narrow == 1 && a.N <= 1024) || g_split_rows_narrow == 2)) mt16 = 1;
    const int kbw = g_tune_kbw[ki][mt16 == 1 ? 0 : mt16 == 2 ? 1 : 2];
    const int nw = K / 32 / kbw;
    // two column tiles per workgroup halve the activation traffic through L2 (it is the larger stream
    // once rows > 16); SwiGLU needs the gate/up pair anyway
    const int nb16 = (epi == EPI_SWIGLU || (g_wide_tiles && rows > 16 && K == 1024 && epi == EPI_STORE)) ? 2 : 1;
    // K = 1024
    Q3_LIN_MT(1, 8, 4, PRO_NORM, EPI_STORE)
    Q3_LIN_MT(1, 4, 8, PRO_NORM, EPI_STORE)
    Q3_LIN_MT(2, 8, 4, PRO_NORM, EPI_SWIGLU)
    Q3_LIN_MT(2, 4, 8, PRO_NORM, EPI_SWIGLU)
    Q3_LIN_MT(1, 8, 4, PRO_F16, EPI_STORE)
    Q3_LIN_MT(2, 4, 8, PRO_NORM, EPI_STORE)
    Q3_LIN_MT(2, 4, 8, PRO_F16, EPI_STORE)
    Q3_LIN_MT(1, 4, 8, PRO_F16, EPI_STORE)
    // K = 2048
    Q3_LIN_MT(1, 8, 8, PRO_F16, EPI_RESID)
    Q3_LIN_MT(1, 4, 16, PRO_F16, EPI_RESID)
    Q3_LIN_MT(1, 16, 4, PRO_F16, EPI_RESID)
    // K = 3072
    Q3_LIN_MT(1, 6, 16, PRO_F16, EPI_RESID)
    Q3_LIN_MT(1, 12, 8, PRO_F16, EPI_RESID)
    Q3_LOG("launch_linear: no instantiation for K=%d kbw=%d nw=%d mt16=%d nb16=%d pro=%d epi=%d", K, kbw, nw,
           mt16, nb16, pro, epi);
    return -1;
}

// ---------------------------------------------------------------------------
// ssq partials of uploaded rows
// ---------------------------------------------------------------------------
// A residual row leaves its producer as: h (f32), 64 sum-of-squares partials, and xh = the consumer's pre-scaled
// fp16 GEMM input (see NORM_PRE); gamma = the consuming layer's norm weight (null: no xh).
__device__ __forceinline__ void store_row_ssq(float* h, float* ssq, int r, int H, int k4, float4 v,
                                              half_t* xh = nullptr, const float* gamma = nullptr) {
    *(float4*)(h + frag_idx(r, k4 * 4, H)) = v;
    if (xh) {
        const float4 g = *(const float4*)(gamma + k4 * 4);
        half_t* p = xh + frag_idx(r, k4 * 4, H);
        p[0] = pre_scaled(v.x, g.x);
        p[1] = pre_scaled(v.y, g.y);
        p[2] = pre_scaled(v.z, g.z);
        p[3] = pre_scaled(v.w, g.w);
    }
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    s += __shfl_xor(s, 1, 4);
    s += __shfl_xor(s, 2, 4);
    if ((k4 & 3) == 0) ssq[(size_t)r * (H / 16) + (k4 >> 2)] = s;
}

__global__ void ssq_rows_kernel(const float* __restrict__ rows, float* __restrict__ h, float* __restrict__ ssq, int H,
                                half_t* __restrict__ xh, const float* __restrict__ gamma, int dst_row0) {
    const int r = blockIdx.x;
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += blockDim.x)
        store_row_ssq(h, ssq, dst_row0 + r, H, k4, *(const float4*)(rows + (size_t)r * H + k4 * 4), xh, gamma);
}
int launch_ssq_rows(hipStream_t s, const float* rows, float* h, float* ssq, int R, int H, half_t* xh, const float* gamma,
                    int dst_row0) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(ssq_rows_kernel, dim3(R), dim3(256), 0, s, rows, h, ssq, H, xh, gamma, dst_row0);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// final RMSNorm of selected rows
// ---------------------------------------------------------------------------
__global__ void final_norm_kernel(FinalNormArgs a) {
    Q3_FETCH_ARGS("s"(a.h), "s"(a.ssq), "s"(a.ssq_parts), "s"(a.gamma), "s"(a.eps), "s"(a.H), "s"(a.row0), "s"(a.row_map), "s"(a.src_off),
                  "s"(a.out_f32), "s"(a.out_f16), "s"(a.out_copy), "s"(a.out_copy_ssq), "s"(a.out_copy_xh), "s"(a.out_copy_gamma),
                  "s"(a.out_copy_row_off));
    Q3_TL(40);
    __shared__ float inv_sh;
    const int r = a.row0 + blockIdx.x;
    const int src = a.row_map ? uniform_load(a.row_map + r) : r + a.src_off;
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int p = threadIdx.x; p < a.ssq_parts; p += 64) s += a.ssq[(size_t)src * a.ssq_parts + p];
        s = wave_sum(s);
        if (threadIdx.x == 0) inv_sh = 1.0f / sqrtf(s / (float)a.H + a.eps);
    }
    __syncthreads();
    const float iv = inv_sh;
    for (int k4 = threadIdx.x; k4 < a.H / 4; k4 += blockDim.x) {
        const float4 v = *(const float4*)(a.h + frag_idx(src, k4 * 4, a.H));
        const float4 g = *(const float4*)(a.gamma + k4 * 4);
        float4 o;
        o.x = (v.x * iv) * g.x;
        o.y = (v.y * iv) * g.y;
        o.z = (v.z * iv) * g.z;
        o.w = (v.w * iv) * g.w;
        if (a.out_f32) *(float4*)(a.out_f32 + (size_t)r * a.H + k4 * 4) = o;
        if (a.out_f16) {
            half_t* p = a.out_f16 + frag_idx(r, k4 * 4, a.H);
            p[0] = sat_half(o.x);
            p[1] = sat_half(o.y);
            p[2] = sat_half(o.z);
            p[3] = sat_half(o.w);
        }
        if (a.out_copy)
            store_row_ssq(a.out_copy, a.out_copy_ssq, r + a.out_copy_row_off, a.H, k4, o, a.out_copy_xh, a.out_copy_gamma);
    }
}
int launch_final_norm(hipStream_t s, const FinalNormArgs& a) {
    if (a.R <= 0) return 0;
    FinalNormArgs a2 = a;
    a2.tl_node = tl_next_node();
    hipLaunchKernelGGL(final_norm_kernel, dim3(a.R), dim3(256), 0, s, a2);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// embedding gather with ssq partials
// ---------------------------------------------------------------------------
__global__ void gather_embed_kernel(const float* __restrict__ table, int V, int H, const int* __restrict__ tok,
                                    int tok_stride, const int* __restrict__ n_frames, int frame_cap, int col,
                                    float* __restrict__ h, float* __restrict__ ssq, int row0, int R_total,
                                    const int* __restrict__ forced, half_t* __restrict__ xh,
                                    const float* __restrict__ gamma) {
    const int r = row0 + blockIdx.x;
    int t;
    if (n_frames) {
        int f = uniform_load(n_frames + r) - 1;
        if (f < 0) f = 0;
        if (f >= frame_cap) f = frame_cap - 1;
        t = uniform_load(tok + ((size_t)f * R_total + r) * 16 + col);
        if (forced) {   // teacher forcing (tests): continue with the forced id where one is given
            const int fz = uniform_load(forced + ((size_t)f * R_total + r) * 16 + col);
            if (fz >= 0 && t >= 0) t = fz;
        }
    } else {
        t = tok[(size_t)r * tok_stride];
    }
    const bool ok = t >= 0 && t < V;
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *(const float4*)(table + (size_t)t * H + k4 * 4);
        store_row_ssq(h, ssq, r, H, k4, v, xh, gamma);
    }
}
int launch_gather_embed(hipStream_t s, const float* table, int V, int H, const int* tok, int tok_stride,
                        const int* n_frames, int frame_cap, int col, float* h, float* ssq, int R, int row0,
                        int R_total, const int* forced, half_t* xh, const float* gamma) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(gather_embed_kernel, dim3(R), dim3(256), 0, s, table, V, H, tok, tok_stride, n_frames,
                       frame_cap, col, h, ssq, row0, R_total > 0 ? R_total : R, forced, xh, gamma);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// attention: one workgroup per (row, kv head).  Phase A (waves 0-3): per-head RMSNorm + RoPE of the
// two q heads and the k head, v pass-through, K/V appended to the cache.  Phase B: every 16-lane
// group walks cached rows t = grp, grp+ngrp, ... (16 B of K and of V per lane, straight to VGPRs)
// with an online softmax per group, groups are merged by shuffles inside a wave and through LDS
// across waves: two barriers in all, no n_ctx-sized LDS.
// ---------------------------------------------------------------------------
template <int MODE, int APRE>
__global__ void __launch_bounds__(1024) attn_kernel(AttnArgs a) {
    constexpr int D = 128;
    Q3_FETCH_ARGS("s"(a.qkv), "s"(a.ld), "s"(a.row0), "s"(a.q_norm), "s"(a.k_norm), "s"(a.eps), "s"(a.rope_cos), "s"(a.rope_sin),
                  "s"(a.slot), "s"(a.pos), "s"(a.slot_base), "s"(a.slot_stride), "s"(a.pos_base), "s"(a.pos_stride), "s"(a.kc),
                  "s"(a.vc), "s"(a.n_ctx), "s"(a.n_kv), "s"(a.n_heads), "s"(a.out), "s"(a.scale), "s"(a.valid_mod), "s"(a.valid_n));
    Q3_TL(30 + MODE);
    const int r = a.row0 + blockIdx.x, g = blockIdx.y;
    if (a.valid_mod > 0 && (r % a.valid_mod) >= a.valid_n) return;   // padding row of a multi-position pass (block-uniform)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nwv = blockDim.x >> 6;
    const int slot = a.slot ? uniform_load(a.slot + r) : a.slot_base + r * a.slot_stride;
    const int pos = a.pos ? uniform_load(a.pos + r) : a.pos_base + r * a.pos_stride;
    __shared__ float qs[2][D];
    __shared__ float knew[D], vnew[D];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* pm = dyn;                    // [nwv][2]  running max per wave/head
    float* pl = dyn + 2 * nwv;          // [nwv][2]  running sum
    float* pacc = dyn + 4 * nwv;        // [nwv][2][D]
    float* row = a.qkv + (size_t)r * a.ld;
    const size_t cbase = ((size_t)slot * a.n_kv + g) * (size_t)a.n_ctx * D;
    const int T = (MODE == ATTN_FUSED) ? pos : pos + 1;  // rows read from the cache
    const int l16 = tid & 15, grp = tid >> 4, ngrp = blockDim.x >> 4;
    const half_t* kbase = a.kc + cbase;
    const half_t* vbase = a.vc + cbase;

    // ---- all loads of the short-T case are issued here: phase A operands first (consumed first;
    // vmcnt retires in order), then the first APRE cached K/V rows of every group ----
    float x0 = 0.f, x1 = 0.f, gm0 = 1.f, gm1 = 1.f, cs = 1.f, sn = 0.f;
    if (MODE != ATTN_ATTEND && w < 4) {
        // wave 0,1: q heads 2g, 2g+1; wave 2: k head g; wave 3: v head g
        const float* src = w < 2 ? row + (size_t)(2 * g + w) * D
                         : w == 2 ? row + (size_t)(a.n_heads + g) * D
                                  : row + (size_t)(a.n_heads + a.n_kv + g) * D;
        x0 = src[lane];
        x1 = src[lane + 64];
        if (w < 3) {
            const float* gam = w < 2 ? a.q_norm : a.k_norm;
            gm0 = gam[lane];
            gm1 = gam[lane + 64];
            cs = a.rope_cos[(size_t)pos * 64 + lane];
            sn = a.rope_sin[(size_t)pos * 64 + lane];
        }
    }
    float qpre = 0.f;
    if (MODE == ATTN_ATTEND && tid < 2 * D) qpre = row[(size_t)(2 * g) * D + tid];
    // APRE cached rows per 16-lane group are requested up front: 4 x 64 groups (1024 threads, one utterance) or
    // 8 x 16 groups (256 threads, a batch of 32) = the first 256 / 128 positions without a second memory round trip
    // (a batch-32 step at positions 65-113 took the dependent in-loop loads with APRE = 4: -0.8 % per frame with 8)
    h8 kpre[APRE], vpre[APRE];
    if (MODE != ATTN_PREP) {
#pragma unroll
        for (int i = 0; i < APRE; i++) {
            const int t = grp + i * ngrp;
            if (t < T) {
                kpre[i] = *(const h8*)(kbase + (size_t)t * D + l16 * 8);
                vpre[i] = *(const h8*)(vbase + (size_t)t * D + l16 * 8);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3_PH(0);  // loads issued

    // ---- phase A: per-head RMSNorm + RoPE (rotate-half pairs i, i+64) ----
    if (MODE != ATTN_ATTEND) {
        if (w < 4) {
            if (w < 3) {
                float ss = wave_sum(x0 * x0 + x1 * x1);
                const float iv = 1.0f / sqrtf(ss / (float)D + a.eps);
                x0 = (x0 * iv) * gm0;
                x1 = (x1 * iv) * gm1;
                const float y0 = x0 * cs - x1 * sn;
                const float y1 = x1 * cs + x0 * sn;
                x0 = y0;
                x1 = y1;
            }
            if (w < 2) {
                if (MODE == ATTN_PREP) {
                    float* dst = row + (size_t)(2 * g + w) * D;
                    dst[lane] = x0;
                    dst[lane + 64] = x1;
                } else {
                    qs[w][lane] = x0;
                    qs[w][lane + 64] = x1;
                }
            } else {
                const half_t h0 = sat_half(x0), h1 = sat_half(x1);
                half_t* cd = (w == 2 ? a.kc : a.vc) + cbase + (size_t)pos * D;
                cd[lane] = h0;
                cd[lane + 64] = h1;
                float* nd = w == 2 ? knew : vnew;
                nd[lane] = (float)h0;
                nd[lane + 64] = (float)h1;
            }
        }
        if (MODE == ATTN_PREP) return;
    } else {
        if (tid < 2 * D) qs[tid / D][tid % D] = qpre;
    }
    __syncthreads();
    Q3_PH(1);  // phase A done (first global round trip + norm + rope), barrier passed

    // ---- phase B: online softmax per 16-lane group ----
    float q0[8], q1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        q0[j] = qs[0][l16 * 8 + j] * a.scale;   // fold the 1/sqrt(D) into q once
        q1[j] = qs[1][l16 * 8 + j] * a.scale;
    }
    float m0 = -INFINITY, m1 = -INFINITY, l0 = 0.f, l1 = 0.f;
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a0[j] = a1[j] = 0.f;

    auto step = [&](const h8& kk, const h8& vv) {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float kf = (float)kk[j];
            d0 += q0[j] * kf;
            d1 += q1[j] * kf;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            d0 += __shfl_xor(d0, o, 16);
            d1 += __shfl_xor(d1, o, 16);
        }
        const float n0 = fmaxf(m0, d0), n1 = fmaxf(m1, d1);
        const float c0 = __expf(m0 - n0), c1 = __expf(m1 - n1);   // exp(-inf) = 0 on the first row
        const float p0 = __expf(d0 - n0), p1 = __expf(d1 - n1);
        l0 = l0 * c0 + p0;
        l1 = l1 * c1 + p1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float vf = (float)vv[j];
            a0[j] = a0[j] * c0 + p0 * vf;
            a1[j] = a1[j] * c1 + p1 * vf;
        }
        m0 = n0;
        m1 = n1;
    };
#pragma unroll
    for (int i = 0; i < APRE; i++) {
        const int t = grp + i * ngrp;
        if (t < T) step(kpre[i], vpre[i]);   // uniform per 16-lane group
    }
#pragma unroll 2
    for (int t = grp + APRE * ngrp; t < T; t += ngrp) {
        const h8 kk = *(const h8*)(kbase + (size_t)t * D + l16 * 8);
        const h8 vv = *(const h8*)(vbase + (size_t)t * D + l16 * 8);
        step(kk, vv);
    }
    if (MODE == ATTN_FUSED && grp == 0) {   // the token being appended, from LDS (fp16-rounded values)
        h8 kk, vv;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            kk[j] = (half_t)knew[l16 * 8 + j];
            vv[j] = (half_t)vnew[l16 * 8 + j];
        }
        step(kk, vv);
    }
    asm volatile("" ::"v"(a0[0]), "v"(l0));
    Q3_PH(2);  // cached rows + the new token processed
    // merge the 4 groups of a wave (lanes xor 16, 32), then the waves through LDS
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float om0 = __shfl_xor(m0, o, 64), om1 = __shfl_xor(m1, o, 64);
        const float ol0 = __shfl_xor(l0, o, 64), ol1 = __shfl_xor(l1, o, 64);
        const float n0 = fmaxf(m0, om0), n1 = fmaxf(m1, om1);
        // a group that saw no row has m = -inf and l = 0: its scale is exp(-inf - n) = 0 unless n is -inf too
        const float c0 = (m0 == -INFINITY) ? 0.f : __expf(m0 - n0), d0 = (om0 == -INFINITY) ? 0.f : __expf(om0 - n0);
        const float c1 = (m1 == -INFINITY) ? 0.f : __expf(m1 - n1), d1 = (om1 == -INFINITY) ? 0.f : __expf(om1 - n1);
        l0 = l0 * c0 + ol0 * d0;
        l1 = l1 * c1 + ol1 * d1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float oa0 = __shfl_xor(a0[j], o, 64), oa1 = __shfl_xor(a1[j], o, 64);
            a0[j] = a0[j] * c0 + oa0 * d0;
            a1[j] = a1[j] * c1 + oa1 * d1;
        }
        m0 = n0;
        m1 = n1;
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            pacc[(w * 2 + 0) * D + lane * 8 + j] = a0[j];
            pacc[(w * 2 + 1) * D + lane * 8 + j] = a1[j];
        }
        if (lane == 0) {
            pm[w * 2 + 0] = m0;
            pm[w * 2 + 1] = m1;
            pl[w * 2 + 0] = l0;
            pl[w * 2 + 1] = l1;
        }
    }
    __syncthreads();
    Q3_PH(3);  // groups merged, partials in LDS, barrier passed
    if (tid < 2 * D) {
        const int hh = tid / D, d = tid % D;
        float M = -INFINITY;
        for (int i = 0; i < nwv; i++) M = fmaxf(M, pm[i * 2 + hh]);
        float L = 0.f, o = 0.f;
        for (int i = 0; i < nwv; i++) {
            const float mi = pm[i * 2 + hh];
            const float c = (mi == -INFINITY) ? 0.f : __expf(mi - M);
            L += pl[i * 2 + hh] * c;
            o += pacc[(i * 2 + hh) * D + d] * c;
        }
        a.out[frag_idx(r, (2 * g + hh) * D + d, a.n_heads * D)] = sat_half(o / L);
    }
}

// ---------------------------------------------------------------------------
// attention over a short cache (code predictor: at most 15 cached positions + the token being appended):
// same phase A; then every 16-lane group scores at most two entries and parks them in LDS, and every
// (head, dim) thread finishes its own output -- max, exponentials, weighted sum over <= 32 entries with the
// V column it prefetched at kernel start.  No merge of partial softmaxes across groups and waves: two
// barriers, one short serial tail (the general kernel's merge + combine cost 1.9 of its 3.7 us here).
// ---------------------------------------------------------------------------
constexpr int ATTN_SHORT_MAX_T = 15;
static int g_attn_short = 1;
int set_attn_short(int on) { g_attn_short = on; return 0; }
// The leading arguments are what the first loads (the row's q / k / v, the norm weights, the RoPE row) need: 16 dwords that
// arrive in SGPRs at wave launch (-amdgpu-kernarg-preload-count=16, like linear_kernel); the rest of the struct is fetched
// in one batch under those loads' latency instead of in front of them.
__global__ void __launch_bounds__(256)
    attn_short_kernel(float* __restrict__ p_qkv, int p_ld, int p_row0, int p_n_heads, int p_n_kv,
                      const float* __restrict__ p_q_norm, const float* __restrict__ p_k_norm,
                      const float* __restrict__ p_rope_cos, const float* __restrict__ p_rope_sin, int p_pos_base,
                      int p_slot_base, AttnArgs a) {
    constexpr int D = 128, NE = ATTN_SHORT_MAX_T + 1;   // entries: <= 15 cached rows + the appended token
    a.qkv = p_qkv;
    a.ld = p_ld;
    a.row0 = p_row0;
    a.n_heads = p_n_heads;
    a.n_kv = p_n_kv;
    a.q_norm = p_q_norm;
    a.k_norm = p_k_norm;
    a.rope_cos = p_rope_cos;
    a.rope_sin = p_rope_sin;
    a.pos_base = p_pos_base;
    a.slot_base = p_slot_base;
    Q3_FETCH_ARGS("s"(a.eps), "s"(a.slot), "s"(a.slot_stride), "s"(a.kc), "s"(a.vc), "s"(a.n_ctx), "s"(a.out), "s"(a.scale));
    Q3_TL(33);
    const int r = a.row0 + blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int slot = a.slot ? uniform_load(a.slot + r) : a.slot_base + r * a.slot_stride;
    const int T = a.pos_base;                       // cached rows = position of the appended token (row-uniform)
    __shared__ float qs[2][D];
    __shared__ float knew[D];
    __shared__ __attribute__((aligned(16))) half_t vs[NE][D];   // V rows of the entries (row T = the appended token)
    __shared__ __attribute__((aligned(16))) float sc[2][NE];
    float* row = a.qkv + (size_t)r * a.ld;
    const size_t cbase = ((size_t)slot * a.n_kv + g) * (size_t)a.n_ctx * D;
    const int l16 = tid & 15, grp = tid >> 4;       // 16 groups of 16 lanes: group e owns entry e
    const int hh = tid >> 7, d = tid & 127;         // the output this thread finishes
    // ---- every global load, issued up front: phase A operands, then this group's cached K and V row ----
    float x0 = 0.f, x1 = 0.f, gm0 = 1.f, gm1 = 1.f, cs = 1.f, sn = 0.f;
    {
        const float* src = w < 2 ? row + (size_t)(2 * g + w) * D
                         : w == 2 ? row + (size_t)(a.n_heads + g) * D
                                  : row + (size_t)(a.n_heads + a.n_kv + g) * D;
        x0 = src[lane];
        x1 = src[lane + 64];
        if (w < 3) {
            const float* gam = w < 2 ? a.q_norm : a.k_norm;
            gm0 = gam[lane];
            gm1 = gam[lane + 64];
            cs = a.rope_cos[(size_t)T * 64 + lane];
            sn = a.rope_sin[(size_t)T * 64 + lane];
        }
    }
    h8 kpre, vpre;
    if (grp < T) {
        kpre = *(const h8*)(a.kc + cbase + (size_t)grp * D + l16 * 8);
        vpre = *(const h8*)(a.vc + cbase + (size_t)grp * D + l16 * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3_PH(0);
    // ---- phase A: per-head RMSNorm + RoPE, K/V appended (identical arithmetic to attn_kernel) ----
    if (w < 3) {
        float ss = wave_sum(x0 * x0 + x1 * x1);
        const float iv = 1.0f / sqrtf(ss / (float)D + a.eps);
        x0 = (x0 * iv) * gm0;
        x1 = (x1 * iv) * gm1;
        const float y0 = x0 * cs - x1 * sn;
        const float y1 = x1 * cs + x0 * sn;
        x0 = y0;
        x1 = y1;
    }
    if (w < 2) {
        qs[w][lane] = x0;
        qs[w][lane + 64] = x1;
    } else {
        const half_t h0 = sat_half(x0), h1 = sat_half(x1);
        half_t* cd = (w == 2 ? a.kc : a.vc) + cbase + (size_t)T * D;
        cd[lane] = h0;
        cd[lane + 64] = h1;
        if (w == 2) {
            knew[lane] = (float)h0;
            knew[lane + 64] = (float)h1;
        } else {
            vs[T][lane] = h0;
            vs[T][lane + 64] = h1;
        }
    }
    if (grp < T) *(h8*)(&vs[grp][l16 * 8]) = vpre;   // cached V rows parked for the last phase
    __syncthreads();
    Q3_PH(1);
    // ---- scores: entry e < T from the cache, entry T = the appended token (fp16-rounded, from LDS) ----
    if (grp <= T) {   // uniform per 16-lane group
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float kf = grp < T ? (float)kpre[j] : knew[l16 * 8 + j];
            d0 += (qs[0][l16 * 8 + j] * a.scale) * kf;
            d1 += (qs[1][l16 * 8 + j] * a.scale) * kf;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            d0 += __shfl_xor(d0, o, 16);
            d1 += __shfl_xor(d1, o, 16);
        }
        if (l16 == 0) {
            sc[0][grp] = d0;
            sc[1][grp] = d1;
        }
    }
    __syncthreads();
    Q3_PH(2);
    // ---- every (head, dim) thread: softmax over the T+1 scores, weighted sum of its V column ----
    float sv[NE];
#pragma unroll
    for (int e4 = 0; e4 < NE; e4 += 4) {     // 4 broadcast reads of 16 B
        const float4 t4 = *(const float4*)(&sc[hh][e4]);
        sv[e4] = t4.x;
        sv[e4 + 1] = t4.y;
        sv[e4 + 2] = t4.z;
        sv[e4 + 3] = t4.w;
    }
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < NE; e++) m = fmaxf(m, e <= T ? sv[e] : -INFINITY);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const float p = e <= T ? __expf(sv[e] - m) : 0.f;       // (rows beyond T hold stale LDS: masked, never NaN-multiplied)
        const float vf = e <= T ? (float)vs[e][d] : 0.f;
        L += p;
        o += p * vf;
    }
    Q3_PH(3);
    a.out[frag_idx(r, (2 * g + hh) * D + d, a.n_heads * D)] = sat_half(o / L);
}

// ---------------------------------------------------------------------------
// attn_tile_mfma_kernel -- the ATTEND step of a prefill for a tile of up to 16 consecutive positions of one utterance
// and one kv head (2 q heads = 32 query rows); K/V of the tile's own positions are already in the cache (ATTN_PREP).
// The generic kernel spends one workgroup per (row, kv head): 7 768 workgroups of a microsecond each for the
// benchmark's 971-row prefill, 30 us per layer.  Here both products run on the MFMA (v_mfma_f32_16x16x32_f16):
// S = Q.K^T with K rows straight from the cache image (the B operand wants 8 consecutive head dims of one key: a K
// row), O = P.V with V rows staged as they are and read through the hardware transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads a 4 x 16 block and every lane receives one COLUMN of it -- the B operand
// wants 8 consecutive keys of one head dim).  The contract keeps q and the softmax in f32: q and P are carried as two fp16 terms (hi,
// lo = fp16((x - hi) * 2048)), two MFMAs per product, recombined in f32 -- 22 mantissa bits, the fp16 K / V values
// exact.  Wave w owns keys 16w..16w+15 of a 64-key chunk for S and head dims 32w..32w+31 for O; row maxima and sums
// meet through LDS; the O accumulators share S's row layout, so the online-softmax rescale needs no data movement.
// Every global load of a chunk (and, for the first, the queries) is issued before anything waits.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) attn_tile_mfma_kernel(AttnArgs a) {
    constexpr int D = 128, QP = 136, VP = 72, CH = 64;
    constexpr float LO = 2048.0f, ILO = 1.0f / 2048.0f;
    Q3_TL(35);
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, q4 = lane >> 4;
    int row0, nrows, slot, pos0;
    if (a.tiles) {
        const int4 t = *(const int4*)(a.tiles + 4 * blockIdx.x);
        row0 = t.x; nrows = t.y; slot = t.z; pos0 = t.w;
    } else {   // one run of consecutive positions of one slot
        row0 = a.row0 + blockIdx.x * 16;
        nrows = a.R - blockIdx.x * 16 < 16 ? a.R - blockIdx.x * 16 : 16;
        slot = a.slot_base;
        pos0 = a.pos_base + blockIdx.x * 16;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* qh = (half_t*)smem;                 // [32][QP]   query rows: head * 16 + position, pre-scaled, hi term
    half_t* ql = qh + 32 * QP;                  // [32][QP]   lo term
    half_t* ks = ql + 32 * QP;                  // [CH][QP]   K rows of the chunk
    half_t* vs = ks + CH * QP;                  // [CH][QP]   V rows of the chunk
    half_t* ph = vs + CH * QP;                  // [32][VP]   P hi
    half_t* pl = ph + 32 * VP;                  // [32][VP]   P lo
    float* red = (float*)(pl + 32 * VP);        // [4][32] per-wave row maxima, then [4][32] row sums
    float* red2 = red + 4 * 32;
    const size_t cbase = ((size_t)slot * a.n_kv + g) * (size_t)a.n_ctx * D;
    const int T = pos0 + nrows;
    // K and V chunks: thread -> 4 pieces of 16 B each, rows idx / 16 (coalesced 256-B rows), zeros beyond T
    h8 kreg[4], vreg[4];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int idx = tid + it * 256, kr = idx >> 4, d8 = (idx & 15) * 8;
            kreg[it] = vreg[it] = (h8){0, 0, 0, 0, 0, 0, 0, 0};
            if (c0 + kr < T) {
                kreg[it] = *(const h8*)(a.kc + cbase + (size_t)(c0 + kr) * D + d8);
                vreg[it] = *(const h8*)(a.vc + cbase + (size_t)(c0 + kr) * D + d8);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int idx = tid + it * 256, kr = idx >> 4, d8 = (idx & 15) * 8;
            *(h8*)(ks + kr * QP + d8) = kreg[it];
            *(h8*)(vs + kr * QP + d8) = vreg[it];
        }
    };
    load_chunk(0);          // in flight together with the query rows
    for (int idx = tid; idx < 32 * (D / 8); idx += 256) {
        const int qr = idx / (D / 8), d8 = (idx - qr * (D / 8)) * 8, i = qr & 15, hh = qr >> 4;
        h8 hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
        if (i < nrows) {
            const float* src = a.qkv + (size_t)(row0 + i) * a.ld + (size_t)(2 * g + hh) * D + d8;
            const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
            const float e[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = e[j] * a.scale;
                hi[j] = sat_half(x);
                lo[j] = (half_t)((x - (float)hi[j]) * LO);
            }
        }
        *(h8*)(qh + qr * QP + d8) = hi;
        *(h8*)(ql + qr * QP + d8) = lo;
    }
    // rows this lane sees in every accumulator fragment: head qb, position 4 * q4 + r
    float mrow[2][4], lrow[2][4];
    f4 oh[2][2], ol[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            mrow[qb][r] = -INFINITY;
            lrow[qb][r] = 0.f;
        }
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int j = 0; j < 2; j++) oh[qb][j] = ol[qb][j] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < T; c0 += CH) {
        __syncthreads();                                    // previous chunk consumed
        store_chunk();                                      // (keys beyond T are zeros; their P is exactly 0 anyway)
        if (c0 + CH < T) load_chunk(c0 + CH);               // the next chunk travels under this chunk's products
        __syncthreads();
        // ---- S = Q.K^T for keys 16w .. 16w+15 of the chunk ----
        f4 sh[2], sl[2];
        sh[0] = sh[1] = sl[0] = sl[1] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; kb++) {
            const h8 bk = *(const h8*)(ks + (16 * w + c) * QP + kb * 32 + q4 * 8);
#pragma unroll
            for (int qb = 0; qb < 2; qb++) {
                const h8 ahi = *(const h8*)(qh + (16 * qb + c) * QP + kb * 32 + q4 * 8);
                const h8 alo = *(const h8*)(ql + (16 * qb + c) * QP + kb * 32 + q4 * 8);
                sh[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bk, sh[qb], 0, 0, 0);
                sl[qb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bk, sl[qb], 0, 0, 0);
            }
        }
        const int key = c0 + 16 * w + c;
        float sc[2][4], mx[2][4];
#pragma unroll
        for (int qb = 0; qb < 2; qb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 4 * q4 + r;
                float v = sh[qb][r] + sl[qb][r] * ILO;
                if (!(i < nrows && key <= pos0 + i)) v = -INFINITY;          // causal mask, rows beyond the tile
                sc[qb][r] = v;
                float x = v;
                x = fmaxf(x, __shfl_xor(x, 1, 16));
                x = fmaxf(x, __shfl_xor(x, 2, 16));
                x = fmaxf(x, __shfl_xor(x, 4, 16));
                x = fmaxf(x, __shfl_xor(x, 8, 16));
                mx[qb][r] = x;
                if (c == 0) red[w * 32 + 16 * qb + i] = x;
            }
        __syncthreads();
        float corr[2][4];
#pragma unroll
        for (int qb = 0; qb < 2; qb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int qr = 16 * qb + 4 * q4 + r;
                const float cm = fmaxf(fmaxf(red[qr], red[32 + qr]), fmaxf(red[64 + qr], red[96 + qr]));
                const float mo = mrow[qb][r], mn = fmaxf(mo, cm);
                corr[qb][r] = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
                const float p = (sc[qb][r] == -INFINITY) ? 0.f : __expf(sc[qb][r] - mn);
                const half_t hi = (half_t)p;
                ph[qr * VP + 16 * w + c] = hi;
                pl[qr * VP + 16 * w + c] = (half_t)((p - (float)hi) * LO);
                float ssum = p;
                ssum += __shfl_xor(ssum, 1, 16);
                ssum += __shfl_xor(ssum, 2, 16);
                ssum += __shfl_xor(ssum, 4, 16);
                ssum += __shfl_xor(ssum, 8, 16);
                if (c == 0) red2[w * 32 + qr] = ssum;
                mrow[qb][r] = mn;
            }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 2; qb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int qr = 16 * qb + 4 * q4 + r;
                lrow[qb][r] = lrow[qb][r] * corr[qb][r] + ((red2[qr] + red2[32 + qr]) + (red2[64 + qr] + red2[96 + qr]));
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    oh[qb][j][r] *= corr[qb][r];
                    ol[qb][j][r] *= corr[qb][r];
                }
            }
        // ---- O += P.V for head dims 32w .. 32w+31 ----
#pragma unroll
        for (int kst = 0; kst < 2; kst++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // B[k = 8 * q4 + e][column c] = V[key kst * 32 + 8 * q4 + e][head dim 32w + 16j + c]: lane 4q + p of a
                // 16-lane group addresses row q, columns 4p .. 4p+3 of a 4 x 16 block and receives column (lane & 15)
                typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
                const half_t* vb = vs + (kst * 32 + 8 * q4 + (c >> 2)) * QP + 32 * w + 16 * j + 4 * (c & 3);
                const hv4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)vb);
                const hv4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)(vb + 4 * QP));
                h8 bv;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    bv[e] = (half_t)t0[e];
                    bv[4 + e] = (half_t)t1[e];
                }
#pragma unroll
                for (int qb = 0; qb < 2; qb++) {
                    const h8 ahi = *(const h8*)(ph + (16 * qb + c) * VP + kst * 32 + q4 * 8);
                    const h8 alo = *(const h8*)(pl + (16 * qb + c) * VP + kst * 32 + q4 * 8);
                    oh[qb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bv, oh[qb][j], 0, 0, 0);
                    ol[qb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bv, ol[qb][j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 4 * q4 + r;
            if (i < nrows) {
                const float inv = 1.0f / lrow[qb][r];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float o = (oh[qb][j][r] + ol[qb][j][r] * ILO) * inv;
                    a.out[frag_idx(row0 + i, (2 * g + qb) * D + 32 * w + 16 * j + c, a.n_heads * D)] = sat_half(o);
                }
            }
        }
}

int launch_attn(hipStream_t s, const AttnArgs& a, int mode) {
    if (a.R <= 0) return 0;
    if (a.n_heads != 2 * a.n_kv) {
        Q3_LOG("attn: only GQA group 2 (16q/8kv) is built, got %d/%d", a.n_heads, a.n_kv);
        return -1;
    }
    // every row at the same short position, known on the host (the code predictor's passes): the short-cache kernel
    if (g_attn_short && mode == ATTN_FUSED && !a.pos && a.pos_stride == 0 && a.pos_base >= 0 &&
        a.pos_base <= ATTN_SHORT_MAX_T && a.pos_base < a.n_ctx) {
        AttnArgs a3 = a;
        a3.tl_node = tl_next_node();
        hipLaunchKernelGGL(attn_short_kernel, dim3(a.R, a.n_kv), dim3(256), 0, s, a3.qkv, a3.ld, a3.row0, a3.n_heads, a3.n_kv,
                           a3.q_norm, a3.k_norm, a3.rope_cos, a3.rope_sin, a3.pos_base, a3.slot_base, a3);
        Q3_HIP(hipGetLastError(), -1);
        return 0;
    }
    // prefill: runs of consecutive positions of one utterance -> 16-position tiles
    // prefill: runs of consecutive positions of one utterance -> 16-position tiles on the MFMA
    if (mode == ATTN_ATTEND && a.n_tiles > 0) {
        constexpr size_t lds = (size_t)(2 * 32 * 136 + 2 * 64 * 136 + 2 * 32 * 72) * 2 + 8 * 32 * 4;
        static bool attr = false;
        if (!attr) {
            Q3_HIP(hipFuncSetAttribute((const void*)attn_tile_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), -1);
            attr = true;
        }
        AttnArgs a5 = a;
        a5.tl_node = tl_next_node();
        hipLaunchKernelGGL(attn_tile_mfma_kernel, dim3(a.n_tiles, a.n_kv), dim3(256), lds, s, a5);
        Q3_HIP(hipGetLastError(), -1);
        return 0;
    }
    // keep the launch small: ~64K threads fill in ~2 us, every further 64K cost ~1 us of ramp
    int threads = a.threads;
    const int fit = 65536 / (a.R * a.n_kv);     // (round 3: 512 threads per workgroup at batch 32 -- 131 072 in all -- changed nothing: 2.41 ms per frame either way)
    if (threads > fit) threads = fit / 64 * 64;
    if (threads < 256) threads = 256;
    if (threads > 1024) threads = 1024;
    const size_t lds = ((size_t)(threads / 64) * (4 + 2 * 128)) * sizeof(float);
    AttnArgs a2 = a;
    a2.tl_node = tl_next_node();
    dim3 grid(a.R, a.n_kv);
#define Q3_ATTN(MODE_)                                                                                   \
    {                                                                                                    \
        if (threads <= 256)                                                                              \
            hipLaunchKernelGGL((attn_kernel<MODE_, 8>), grid, dim3(256), mode == ATTN_PREP ? 0 : lds, s, a2); \
        else                                                                                             \
            hipLaunchKernelGGL((attn_kernel<MODE_, 4>), grid, dim3(mode == ATTN_PREP ? 256 : threads),   \
                               mode == ATTN_PREP ? 0 : lds, s, a2);                                      \
    }
    if (mode == ATTN_FUSED) Q3_ATTN(ATTN_FUSED)
    else if (mode == ATTN_PREP) Q3_ATTN(ATTN_PREP)
    else Q3_ATTN(ATTN_ATTEND)
#undef Q3_ATTN
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// talker sampling, greedy form of llamacpp_talker_server.py:163-206
// ---------------------------------------------------------------------------
__device__ __forceinline__ void block_argmax(float& v, int& idx, float* sv, int* si) {
    // lowest index wins ties
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) {
            v = ov;
            idx = oi;
        }
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sv[w] = v;
        si[w] = idx;
    }
    __syncthreads();
    v = sv[0];
    idx = si[0];
    for (int i = 1; i < nw; i++)
        if (sv[i] > v || (sv[i] == v && si[i] < idx)) {
            v = sv[i];
            idx = si[i];
        }
    __syncthreads();
}

// ---- stochastic sampling on the device (top-k / temperature / optional top-p) -----------------
// Counter-based generator: the draw depends only on (seed, row, frame, group), so a captured graph
// replays deterministically for a given seed.
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned row, unsigned frame, unsigned group) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (1ull + row + ((unsigned long long)frame << 20) +
                                                            ((unsigned long long)group << 44));
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0, 1)
}

// NaN logits: numpy's argmax (the reference's greedy limit) lets the first NaN win; the device does the same by
// ordering a NaN as +inf.  A row whose best key is not finite (NaN, +inf, or nothing but -inf) has no softmax:
// the sampler then returns that best index instead of drawing (the reference would raise inside np.random.choice
// and answer -2; a device kernel cannot raise, and must never index with an invalid winner).
__device__ __forceinline__ float nan_as_inf(float l) { return l != l ? INFINITY : l; }

constexpr int SAMPLE_SEL_CAP = 64;      // top_k up to this: k selection rounds; beyond (or "all"): a full LDS sort
constexpr int SAMPLE_SORT_CAP = 4096;   // largest vocabulary the full sort handles (talker 3072, code predictor 2048)
// top_k as the reference means it: <= 0 or >= n keeps every entry (np.argpartition is skipped there)
__host__ __device__ __forceinline__ int effective_top_k(int top_k, int n) { return (top_k <= 0 || top_k > n) ? n : top_k; }
__host__ __device__ __forceinline__ int sample_sort_len(int n) {
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}
// dynamic LDS bytes the samplers need for a vocabulary of n at this top_k (0 when greedy)
static size_t sample_lds_bytes(int n, int top_k, float temperature) {
    if (!(temperature > 1e-6f)) return 0;
    return effective_top_k(top_k, n) <= SAMPLE_SEL_CAP ? (size_t)n * 4 : (size_t)sample_sort_len(n) * 8;
}

// lg: n processed logits in LDS (destroyed; capacity n floats, or sample_sort_len(n) floats followed by as many
// ints when top_k needs the full sort).  Keeps the top_k largest (ties: lowest index first), applies
// softmax((l - max) / max(T, 1e-6)) like the reference samplers, optionally keeps the smallest prefix of the
// descending order whose mass reaches top_p (llamacpp_talker_server.py:199-205), and draws with u.
// All threads of the block must call it; every thread returns the chosen index (always in [0, n)).
__device__ int block_sample_topk(float* lg, int n, int top_k, float temperature, float top_p, float u, float* sv,
                                 int* si, float* selv, int* seli) {
    __shared__ int chosen;
    top_k = effective_top_k(top_k, n);
    const float inv_t = 1.0f / fmaxf(temperature, 1e-6f);
    if (top_k <= SAMPLE_SEL_CAP) {
        int found = 0;
        for (int k = 0; k < top_k; k++) {
            float best = -INFINITY;
            int bidx = 0x7fffffff;
            for (int v = threadIdx.x; v < n; v += blockDim.x) {
                const float l = lg[v];
                if (l > best) {
                    best = l;
                    bidx = v;
                }
            }
            block_argmax(best, bidx, sv, si);
            if (bidx >= n) break;            // nothing above -inf is left (block-uniform)
            if (threadIdx.x == 0) {
                selv[k] = best;
                seli[k] = bidx;
                lg[bidx] = -INFINITY;
            }
            found = k + 1;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            int pick = 0;
            if (found > 0 && !(selv[0] < INFINITY)) pick = seli[0];       // NaN / +inf on top: no softmax exists
            else if (found > 0) {
                // the selection is in descending order: selv[0] is the maximum
                float sum = 0.f;
                for (int k = 0; k < found; k++) {
                    selv[k] = expf((selv[k] - selv[0]) * inv_t);
                    sum += selv[k];
                }
                int keep = found;
                if (top_p > 0.f && top_p < 1.f) {
                    float c = 0.f;
                    for (int k = 0; k < found; k++) {
                        c += selv[k] / sum;
                        if (c >= top_p) {   // np.searchsorted(cumsum, top_p) + 1 entries
                            keep = k + 1;
                            break;
                        }
                    }
                    sum = 0.f;
                    for (int k = 0; k < keep; k++) sum += selv[k];
                }
                float c = 0.f;
                pick = seli[keep - 1];
                const float target = u * sum;
                for (int k = 0; k < keep; k++) {
                    c += selv[k];
                    if (target < c) {
                        pick = seli[k];
                        break;
                    }
                }
            }
            chosen = pick;
        }
        __syncthreads();
        return chosen;
    }
    // ---- top_k beyond the selection cap (or "all"): bitonic sort of (key, index), descending key, ascending index ----
    const int n2 = sample_sort_len(n);
    int* li = (int*)(lg + n2);
    for (int v = threadIdx.x; v < n2; v += blockDim.x) {
        if (v >= n) lg[v] = -INFINITY;
        li[v] = v;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= n2; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float a0 = lg[i], a1 = lg[ixj];
                    const int i0 = li[i], i1 = li[ixj];
                    const bool before = a0 > a1 || (a0 == a1 && i0 < i1);   // element i belongs ahead of element ixj
                    const bool desc = (i & k2) == 0;
                    if (desc ? !before : before) {
                        lg[i] = a1;
                        lg[ixj] = a0;
                        li[i] = i1;
                        li[ixj] = i0;
                    }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        int pick = li[0] < n ? li[0] : 0;
        const float top = lg[0];
        if (top < INFINITY && top > -INFINITY) {
            float sum = 0.f;
            int found = 0;
            for (int k = 0; k < top_k; k++) {
                if (!(lg[k] > -INFINITY)) break;
                lg[k] = expf((lg[k] - top) * inv_t);
                sum += lg[k];
                found = k + 1;
            }
            int keep = found;
            if (top_p > 0.f && top_p < 1.f) {
                float c = 0.f;
                for (int k = 0; k < found; k++) {
                    c += lg[k] / sum;
                    if (c >= top_p) {
                        keep = k + 1;
                        break;
                    }
                }
                sum = 0.f;
                for (int k = 0; k < keep; k++) sum += lg[k];
            }
            float c = 0.f;
            pick = li[keep - 1];
            const float target = u * sum;
            for (int k = 0; k < keep; k++) {
                c += lg[k];
                if (target < c) {
                    pick = li[k];
                    break;
                }
            }
        }
        chosen = pick;
    }
    __syncthreads();
    return chosen;
}

__global__ void talker_sample_kernel(TalkerSampleArgs a) {
    Q3_TL(42);
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int win[32];
    __shared__ int nwin;
    const int r = a.row0 + blockIdx.x;
    const int RT = a.R_total > 0 ? a.R_total : a.R;
    const int np = uniform_load(a.n_past + r);
    const int nt = uniform_load(a.n_text + r);
    const bool was_done = uniform_load(a.done + r) != 0;
    if (threadIdx.x == 0) nwin = np < 30 ? np : 30;
    if (threadIdx.x < 30 && threadIdx.x < np) {
        // last 30 emitted tokens (ring of 32)
        win[threadIdx.x] = a.past[r * 32 + ((np - 1 - threadIdx.x) & 31)];
    }
    __syncthreads();
    // adaptive EOS boost (python floats = doubles; numpy>=2 adds it as a float32)
    double progress = 0.0;
    float boost = 0.f;
    bool force_eos = false;
    if (nt > 0) {
        const double expected = (double)nt * 3.0;
        progress = (double)np / expected;
        if (progress > 0.8) {
            double b = (progress - 0.8) / 0.7;
            if (b > 1.0) b = 1.0;
            boost = (float)(b * 15.0);
        }
        if (progress > 2.0) force_eos = true;
    }
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const int nw_ = nwin;
    extern __shared__ float slg[];   // [V] processed logits, only when sampling stochastically
    __shared__ float selv[64];
    __shared__ int seli[64];
    const bool stochastic = a.temperature > 1e-6f;
    for (int v = threadIdx.x; v < a.V; v += blockDim.x) {
        float l = nan_as_inf(a.logits[(size_t)r * a.V + v]);
        if (v >= a.audio_vocab && v != a.eos) l = -1e10f;
        if (v == a.eos) {
            if (a.ignore_eos) l = -1e10f;
            else if (nt > 0 && progress > 0.8) l += boost;
        }
        bool rep = false;
        for (int i = 0; i < nw_; i++) rep |= (win[i] == v);
        if (rep) l = l > 0.f ? __fdiv_rn(l, a.rep_penalty) : l * a.rep_penalty;
        if (stochastic) slg[v] = l;
        if (l > best || (l == best && v < bidx)) {
            best = l;
            bidx = v;
        }
    }
    if (stochastic) {
        __syncthreads();
        const float u = uniform01(a.seed_ptr ? a.seed_ptr[r] : a.seed, (unsigned)r, (unsigned)a.n_frames[r], 0u);
        bidx = block_sample_topk(slg, a.V, a.top_k, a.temperature, a.top_p, u, sv, si, selv, seli);
    } else {
        block_argmax(best, bidx, sv, si);
        if (bidx >= a.V) bidx = a.eos;   // unreachable (the mask leaves finite entries); never index with an invalid winner
    }
    if (threadIdx.x == 0) {
        int code = bidx;
        if (force_eos && !a.ignore_eos) code = a.eos;
        bool fin = was_done || code == a.eos || code >= a.audio_vocab || (a.max_frames > 0 && np >= a.max_frames);
        const int f = a.n_frames[r];
        a.n_frames[r] = f + 1;
        const bool keep = f < a.frame_cap;     // a frame beyond the codes array is not recorded (q3e_run never asks for one)
        int* fc = a.codes + ((size_t)(keep ? f : 0) * RT + r) * 16;
        // teacher forcing (tests): the decision is recorded, the forced id is what the stream continues with
        int used = code;
        if (a.forced && keep && !fin) {
            const int fz = a.forced[((size_t)f * RT + r) * 16];
            if (fz >= 0) used = fz;
        }
        if (fin || !keep) {
            a.done[r] = 1;
            if (keep) fc[0] = -1;
        } else {
            fc[0] = code;
            a.past[r * 32 + (np & 31)] = used;
            a.n_past[r] = np + 1;
            a.pos[r] = a.pos0[r] + np;
        }
    }
}
int launch_talker_sample(hipStream_t s, const TalkerSampleArgs& a) {
    if (a.R <= 0) return 0;
    if (a.temperature > 1e-6f && effective_top_k(a.top_k, a.V) > SAMPLE_SEL_CAP && a.V > SAMPLE_SORT_CAP) {
        Q3_LOG("talker_sample: top_k=%d over a vocabulary of %d is beyond the device sampler (sort cap %d)", a.top_k, a.V,
               SAMPLE_SORT_CAP);
        return -1;
    }
    const size_t lds = sample_lds_bytes(a.V, a.top_k, a.temperature);
    TalkerSampleArgs a2 = a;
    a2.tl_node = tl_next_node();
    hipLaunchKernelGGL(talker_sample_kernel, dim3(a.R), dim3(256), lds, s, a2);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// code-predictor argmax + gather / feedback
// ---------------------------------------------------------------------------
// `codes` points at the row's 16 codes; (ov_g, ov_tok) overrides group ov_g with a value the caller
// just computed (no same-kernel global read-after-write).
__device__ __forceinline__ void feedback_row(const int* codes, int r, const float* talker_emb, int talker_vocab,
                                             const float* const* cp_tables, int cp_vocab, int n_groups,
                                             const float* pad, float* h_out, float* ssq_out, int H,
                                             int ov_g = -1, int ov_tok = 0, const int* fz = nullptr,
                                             half_t* xh_out = nullptr, const float* gamma = nullptr) {
    // tts_client.py:199-208: copy codec_embedding[code_0], += cp table g row, += tts_pad, in this order
    // (fz: teacher-forced ids of this frame, tests only -- entries >= 0 replace the recorded decisions)
    int c0 = codes[0];
    if (fz && fz[0] >= 0 && c0 >= 0) c0 = fz[0];
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 >= 0 && c0 < talker_vocab) v = *(const float4*)(talker_emb + (size_t)c0 * H + k4 * 4);
        for (int g = 0; g < n_groups; g++) {
            int t = g == ov_g ? ov_tok : codes[1 + g];
            if (fz && g != ov_g && fz[1 + g] >= 0) t = fz[1 + g];
            if (t >= 0 && t < cp_vocab) {
                const float4 e = *(const float4*)(cp_tables[g] + (size_t)t * H + k4 * 4);
                v.x += e.x;
                v.y += e.y;
                v.z += e.z;
                v.w += e.w;
            }
        }
        if (pad) {
            const float4 e = *(const float4*)(pad + k4 * 4);
            v.x += e.x;
            v.y += e.y;
            v.z += e.z;
            v.w += e.w;
        }
        store_row_ssq(h_out, ssq_out, r, H, k4, v, xh_out, gamma);
    }
}

// One workgroup per row: the 2048 logits arrive as two float4 per thread (one round trip), one barrier
// picks the winner, then the next embedding row is gathered (second round trip).
__global__ void __launch_bounds__(256) cp_argmax_kernel(CpArgmaxArgs a) {
    Q3_FETCH_ARGS("s"(a.logits), "s"(a.V), "s"(a.H), "s"(a.row0), "s"(a.R_total), "s"(a.R), "s"(a.group), "s"(a.codes), "s"(a.n_frames),
                  "s"(a.frame_cap), "s"(a.next_table), "s"(a.h_out), "s"(a.ssq_out), "s"(a.xh_out), "s"(a.gamma_next),
                  "s"(a.talker_emb), "s"(a.temperature), "s"(a.forced));
    Q3_TL(43);
    __shared__ float sv[4];
    __shared__ int si[4];
    const int r = a.row0 + blockIdx.x, tid = threadIdx.x;
    const int RT = a.R_total > 0 ? a.R_total : a.R;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const float4* lg = (const float4*)(a.logits + (size_t)r * a.V);
    const int n4 = a.V / 4;
    float4 l0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), l1 = l0;
    if (tid < n4) l0 = lg[tid];
    if (tid + 256 < n4) l1 = lg[tid + 256];
    const int nf_r = uniform_load(a.n_frames + r);   // independent of the arg-max: a scalar load beside the logits' vector loads
    Q3_PH(0);
    {
        const float e[8] = {nan_as_inf(l0.x), nan_as_inf(l0.y), nan_as_inf(l0.z), nan_as_inf(l0.w),
                            nan_as_inf(l1.x), nan_as_inf(l1.y), nan_as_inf(l1.z), nan_as_inf(l1.w)};
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = (j < 4 ? tid : tid + 256) * 4 + (j & 3);
            if (e[j] > best) {   // ascending index within a thread: strict > keeps the lowest index
                best = e[j];
                bidx = idx;
            }
        }
    }
    for (int v4 = tid + 512; v4 < n4; v4 += 256) {   // vocabularies beyond 2048
        const float4 l = lg[v4];
        const float e[4] = {nan_as_inf(l.x), nan_as_inf(l.y), nan_as_inf(l.z), nan_as_inf(l.w)};
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (e[j] > best) {
                best = e[j];
                bidx = v4 * 4 + j;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > best || (ov == best && oi < bidx)) {
            best = ov;
            bidx = oi;
        }
    }
    if ((tid & 63) == 0) {
        sv[tid >> 6] = best;
        si[tid >> 6] = bidx;
    }
    __syncthreads();
    Q3_PH(1);  // logits landed, wave arg-max done, barrier passed
    best = sv[0];
    bidx = si[0];
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (sv[i] > best || (sv[i] == best && si[i] < bidx)) {
            best = sv[i];
            bidx = si[i];
        }
    if (a.temperature > 1e-6f) {   // code_predictor_server.py:87-92: top-k, softmax((l-max)/T), categorical draw
        extern __shared__ float slg[];
        __shared__ float selv[64];
        __shared__ int seli[64];
        __shared__ float sv2[16];
        __shared__ int si2[16];
        __syncthreads();
        for (int v = tid; v < a.V; v += 256) slg[v] = nan_as_inf(a.logits[(size_t)r * a.V + v]);
        __syncthreads();
        const float u = uniform01(a.seed_ptr ? a.seed_ptr[r] : a.seed, (unsigned)r, (unsigned)nf_r, 1u + (unsigned)a.group);
        bidx = block_sample_topk(slg, a.V, a.top_k, a.temperature, 0.f, u, sv2, si2, selv, seli);
    }
    if (bidx < 0 || bidx >= a.V) bidx = 0;   // every logit -inf: numpy's argmax answers 0; never gather with an invalid winner
    int f = nf_r - 1;
    if (f < 0) f = 0;
    const bool keep = f < a.frame_cap;        // a frame beyond the codes array is not recorded
    if (!keep) f = a.frame_cap - 1;
    int* fc = a.codes + ((size_t)f * RT + r) * 16;
    if (tid == 0 && keep) fc[1 + a.group] = bidx;
    const int* fz = (a.forced && keep) ? a.forced + ((size_t)f * RT + r) * 16 : nullptr;
    if (fz && fz[1 + a.group] >= 0) bidx = fz[1 + a.group];   // teacher forcing (tests): continue with the forced id
    Q3_PH(2);
    if (a.talker_emb) {
        feedback_row(fc, r, a.talker_emb, a.talker_vocab, a.cp_tables, a.V, a.n_groups, a.pad_embed,
                     a.h_out, a.ssq_out, a.H, a.group, bidx, fz, a.xh_out, a.gamma_next);
    } else if (a.next_table) {
        for (int k4 = tid; k4 < a.H / 4; k4 += 256) {
            const float4 v = *(const float4*)(a.next_table + (size_t)bidx * a.H + k4 * 4);
            store_row_ssq(a.h_out, a.ssq_out, r, a.H, k4, v, a.xh_out, a.gamma_next);
        }
    }
}
int launch_cp_argmax(hipStream_t s, const CpArgmaxArgs& a) {
    if (a.R <= 0) return 0;
    if (a.temperature > 1e-6f && effective_top_k(a.top_k, a.V) > SAMPLE_SEL_CAP && a.V > SAMPLE_SORT_CAP) {
        Q3_LOG("cp_argmax: top_k=%d over a vocabulary of %d is beyond the device sampler (sort cap %d)", a.top_k, a.V,
               SAMPLE_SORT_CAP);
        return -1;
    }
    const size_t lds = sample_lds_bytes(a.V, a.top_k, a.temperature);
    CpArgmaxArgs a2 = a;
    a2.tl_node = tl_next_node();
    hipLaunchKernelGGL(cp_argmax_kernel, dim3(a.R), dim3(256), lds, s, a2);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

__global__ void feedback_kernel(const int* codes, const float* talker_emb, int talker_vocab,
                                const float* const* cp_tables, int cp_vocab, int n_groups, const float* pad,
                                float* h_out, float* ssq_out, int H, half_t* xh_out, const float* gamma) {
    feedback_row(codes + (size_t)blockIdx.x * 16, blockIdx.x, talker_emb, talker_vocab, cp_tables, cp_vocab, n_groups,
                 pad, h_out, ssq_out, H, -1, 0, nullptr, xh_out, gamma);
}
int launch_feedback(hipStream_t s, const int* codes16, int R, const float* talker_emb, int talker_vocab,
                    const float* const* cp_tables, int cp_vocab, int n_groups, const float* pad_embed,
                    float* h_out, float* ssq_out, int H, half_t* xh_out, const float* gamma) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(feedback_kernel, dim3(R), dim3(256), 0, s, codes16, talker_emb, talker_vocab, cp_tables,
                       cp_vocab, n_groups, pad_embed, h_out, ssq_out, H, xh_out, gamma);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

}  // namespace q3

#ifndef Q3_TIMELINE
namespace q3 {
int tl_next_node() { return -1; }
}
#endif
#ifdef Q3_TIMELINE
namespace q3 {
// host control of the diagnostic timeline
static unsigned long long* g_ph_host = nullptr;
static unsigned long long* g_tl2_host = nullptr;
static int* g_tl2_kind_host = nullptr;
static int g_tl_node_counter = -1;  // -1: numbering off
int tl_next_node() { return g_tl_node_counter < 0 ? -1 : g_tl_node_counter++; }
int tl2_begin(int max_nodes) {
    if (hipMalloc((void**)&g_tl2_host, (size_t)max_nodes * TL2_MAXB * 16) != hipSuccess) return -1;
    hipMemset(g_tl2_host, 0, (size_t)max_nodes * TL2_MAXB * 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl2), &g_tl2_host, sizeof(g_tl2_host));
    if (hipMalloc((void**)&g_tl2_kind_host, (size_t)max_nodes * 4) != hipSuccess) return -1;
    hipMemset(g_tl2_kind_host, 0xff, (size_t)max_nodes * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl2_kind), &g_tl2_kind_host, sizeof(g_tl2_kind_host));
    g_tl_node_counter = 0;
    return 0;
}
int tl2_kinds(int* out, int max_nodes) {
    return hipMemcpy(out, g_tl2_kind_host, (size_t)max_nodes * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
int tl2_end(unsigned long long* out, int max_nodes) {
    hipDeviceSynchronize();
    const int n = g_tl_node_counter;
    g_tl_node_counter = -1;
    hipMemcpy(out, g_tl2_host, (size_t)(n < max_nodes ? n : max_nodes) * TL2_MAXB * 16, hipMemcpyDeviceToHost);
    return n;
}
int tl_begin(unsigned cap) {
    unsigned long long* buf = nullptr;
    if (hipMalloc((void**)&buf, (size_t)cap * 16) != hipSuccess) return -1;
    hipMemset(buf, 0, (size_t)cap * 16);
    if (hipMalloc((void**)&g_ph_host, (size_t)cap * 64) != hipSuccess) return -1;
    hipMemset(g_ph_host, 0, (size_t)cap * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ph), &g_ph_host, sizeof(g_ph_host));
    unsigned zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl_idx), &zero, 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl_cap), &cap, 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &buf, sizeof(buf));
    return 0;
}
int tl_end(unsigned long long* out, unsigned cap) {
    hipDeviceSynchronize();
    unsigned long long* buf = nullptr;
    unsigned n = 0;
    hipMemcpyFromSymbol(&buf, HIP_SYMBOL(g_tl), sizeof(buf));
    hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tl_idx), 4);
    if (!buf) return -1;
    if (n > cap) n = cap;
    hipMemcpy(out, buf, (size_t)n * 16, hipMemcpyDeviceToHost);
    unsigned long long* nul = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &nul, sizeof(nul));
    hipFree(buf);
    return (int)n;
}
}  // namespace q3
extern "C" int q3t_set_skip(int on) { return hipMemcpyToSymbol(HIP_SYMBOL(q3::g_skip), &on, 4) == hipSuccess ? 0 : -1; }
extern "C" int q3t_tl_begin(unsigned cap) { return q3::tl_begin(cap); }
extern "C" int q3t_tl_end(unsigned long long* out, unsigned cap) { return q3::tl_end(out, cap); }
extern "C" int q3t_tl2_begin(int max_nodes) { return q3::tl2_begin(max_nodes); }
extern "C" int q3t_tl2_end(unsigned long long* out, int max_nodes) { return q3::tl2_end(out, max_nodes); }
extern "C" int q3t_tl2_kinds(int* out, int max_nodes) { return q3::tl2_kinds(out, max_nodes); }
extern "C" int q3t_tl_phases(unsigned long long* out, unsigned n) {
    return hipMemcpy(out, q3::g_ph_host, (size_t)n * 64, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif
