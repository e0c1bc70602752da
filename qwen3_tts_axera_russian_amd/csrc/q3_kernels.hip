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
template <int NB16, int MT16, int KBW, int NW, int PRO, int EPI, bool NT>
__global__ void __launch_bounds__(NW * 64) linear_kernel(LinArgs a) {
    constexpr int MR = MT16 * 16, NB = NB16 * 16, NBP = NB + 4, KB = NW * KBW, K = KB * 32;
    constexpr int NTH = NW * 64;
    constexpr int NOUT = (EPI == EPI_SWIGLU) ? MR * NB / 2 : MR * NB;   // outputs of this workgroup
    constexpr int OPT = (NOUT + NTH - 1) / NTH;                         // outputs per thread
    constexpr int SQI = (MR * 16 + NTH - 1) / NTH;                      // ssq float4 groups per thread
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int tile0 = blockIdx.x * NB16;
    const int m0 = blockIdx.y * MR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;             // [NW][MR][NBP]
    float* inv_s = red + NW * MR * NBP;    // [MR]

    // ---- 1. every global load of the kernel is issued here, activation-side first ----
    // vmcnt retires in order: the small L2-resident operands must not queue behind the HBM weight
    // stream, and nothing later in the kernel starts a second memory round trip.
    float4 sq[SQI];
    float4 hraw[MT16][KBW][2];
    float4 graw[KBW][2];
    h8 af[MT16][KBW];
    float hold[OPT];
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int i = 0; i < SQI; i++) {
            const int g4 = tid + i * NTH;          // float4 group: row = g4/16 (64 partials = 16 float4)
            const int m = m0 + g4 / 16;
            sq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g4 < MR * 16 && m < a.M) sq[i] = *(const float4*)(a.ssq + (size_t)m * 64 + (g4 & 15) * 4);
        }
#pragma unroll
        for (int kbi = 0; kbi < KBW; kbi++) {
            const int k0 = (w * KBW + kbi) * 32 + q * 8;
            graw[kbi][0] = *(const float4*)(a.gamma + k0);
            graw[kbi][1] = *(const float4*)(a.gamma + k0 + 4);
#pragma unroll
            for (int mt = 0; mt < MT16; mt++) {
                int m = m0 + mt * 16 + c;
                if (m >= a.M) m = a.M - 1;  // padded rows recompute a valid row; never stored
                hraw[mt][kbi][0] = *(const float4*)(a.h + (size_t)m * K + k0);
                hraw[mt][kbi][1] = *(const float4*)(a.h + (size_t)m * K + k0 + 4);
            }
        }
    } else {
#pragma unroll
        for (int kbi = 0; kbi < KBW; kbi++) {
            const int k0 = (w * KBW + kbi) * 32 + q * 8;
#pragma unroll
            for (int mt = 0; mt < MT16; mt++) {
                int m = m0 + mt * 16 + c;
                if (m >= a.M) m = a.M - 1;
                af[mt][kbi] = *(const h8*)(a.x16 + (size_t)m * K + k0);
            }
        }
    }
    if (EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < OPT; i++) {
            const int o = tid + i * NTH;
            const int m = m0 + o / NB;
            hold[i] = 0.f;
            if (o < NOUT && m < a.M) hold[i] = a.h_out[(size_t)m * a.N + tile0 * 16 + (o % NB)];
        }
    }
    // the weight stream: everything this wave will need, in flight at once
    h8 wf[NB16][KBW];
#pragma unroll
    for (int nb = 0; nb < NB16; nb++)
#pragma unroll
        for (int kbi = 0; kbi < KBW; kbi++) {
            const h8* p = (const h8*)(a.wp + (((size_t)(tile0 + nb) * KB + (size_t)w * KBW + kbi) * 64 + lane) * 8);
            wf[nb][kbi] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. RMSNorm scale per row from the producer's 64 sum-of-squares partials (a.ssq_parts == 64) ----
    if (PRO == PRO_NORM) {
#pragma unroll
        for (int i = 0; i < SQI; i++) {
            const int g4 = tid + i * NTH;
            float s = (sq[i].x + sq[i].y) + (sq[i].z + sq[i].w);
            s += __shfl_xor(s, 8, 16);
            s += __shfl_xor(s, 4, 16);
            s += __shfl_xor(s, 2, 16);
            s += __shfl_xor(s, 1, 16);
            if (g4 < MR * 16 && (g4 & 15) == 0) inv_s[g4 / 16] = 1.0f / sqrtf(s / (float)K + a.eps);
        }
        __syncthreads();
#pragma unroll
        for (int kbi = 0; kbi < KBW; kbi++)
#pragma unroll
            for (int mt = 0; mt < MT16; mt++) {
                const float iv = inv_s[mt * 16 + c];
                const float4 h0 = hraw[mt][kbi][0], h1 = hraw[mt][kbi][1];
                const float4 g0 = graw[kbi][0], g1 = graw[kbi][1];
                h8 t;
                t[0] = sat_half((h0.x * iv) * g0.x);
                t[1] = sat_half((h0.y * iv) * g0.y);
                t[2] = sat_half((h0.z * iv) * g0.z);
                t[3] = sat_half((h0.w * iv) * g0.w);
                t[4] = sat_half((h1.x * iv) * g1.x);
                t[5] = sat_half((h1.y * iv) * g1.y);
                t[6] = sat_half((h1.z * iv) * g1.z);
                t[7] = sat_half((h1.w * iv) * g1.w);
                af[mt][kbi] = t;
            }
    }

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

    // ---- 4. partial tiles -> LDS.  D layout: col = lane&15, row = 4*(lane>>4) + reg. ----
#pragma unroll
    for (int mt = 0; mt < MT16; mt++)
#pragma unroll
        for (int nb = 0; nb < NB16; nb++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                red[(w * MR + mt * 16 + 4 * q + r) * NBP + nb * 16 + c] = acc[mt][nb][r];
    __syncthreads();

    // ---- 5. fixed-order sum over waves + epilogue ----
    if (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
        for (int i = 0; i < OPT; i++) {
            const int o = tid + i * NTH;
            if (o < NOUT) {   // wave-uniform (NOUT and NTH are multiples of 64)
                const int mr = o / NB, n = o % NB;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ww++) v += red[(ww * MR + mr) * NBP + n];
                const int m = m0 + mr;
                const int ng = tile0 * 16 + n;
                const bool ok = m < a.M;
                if (EPI == EPI_STORE) {
                    if (ok) a.y[(size_t)m * a.ldy + ng] = v;
                } else {
                    const float hn = ok ? hold[i] + v : 0.f;
                    if (ok) a.h_out[(size_t)m * a.N + ng] = hn;
                    float s = hn * hn;
                    s += __shfl_xor(s, 8, 16);
                    s += __shfl_xor(s, 4, 16);
                    s += __shfl_xor(s, 2, 16);
                    s += __shfl_xor(s, 1, 16);
                    if (ok && (n & 15) == 0) a.ssq_out[(size_t)m * (a.N / 16) + (ng >> 4)] = s;
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
                    g += red[(ww * MR + mr) * NBP + (2 * ii) * 16 + cc];
                    u += red[(ww * MR + mr) * NBP + (2 * ii + 1) * 16 + cc];
                }
                const int m = m0 + mr;
                if (m < a.M) {
                    const float sg = g / (1.0f + expf(-g));
                    a.act[(size_t)m * (a.N / 2) + (size_t)blockIdx.x * NH + j] = sat_half(sg * u);
                }
            }
        }
    }
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
    dim3 grid(a.N / (16 * NB16), (a.M + MR - 1) / MR);
    hipLaunchKernelGGL((linear_kernel<NB16, MT16, KBW, NW, PRO, EPI, NT>), grid, dim3(NW * 64), lds, s, a);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

template <int NB16, int MT16, int KBW, int NW, int PRO, int EPI>
static int launch_linear_t(hipStream_t s, const LinArgs& a) {
    return a.nt ? launch_linear_nt<NB16, MT16, KBW, NW, PRO, EPI, true>(s, a)
                : launch_linear_nt<NB16, MT16, KBW, NW, PRO, EPI, false>(s, a);
}

// (KBW, NW) per K; overridable for tuning through q3_set_linear_tuning().
// k-blocks per wave by [K = 1024, 2048, 3072][rows <= 16, <= 32, more]
static int g_tune_kbw[3][3] = {{8, 4, 4}, {8, 8, 8}, {6, 6, 6}};
int set_linear_tuning(int K, int mt16, int kbw) {
    int i = K == 1024 ? 0 : K == 2048 ? 1 : K == 3072 ? 2 : -1;
    int j = mt16 == 1 ? 0 : mt16 == 2 ? 1 : mt16 == 4 ? 2 : -1;
    if (i < 0 || j < 0) return -1;
    g_tune_kbw[i][j] = kbw;
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
    if (a.M <= 0) return 0;
    const int K = a.K;
    int ki = K == 1024 ? 0 : K == 2048 ? 1 : K == 3072 ? 2 : -1;
    if (ki < 0 || a.N % 32) {
        Q3_LOG("launch_linear: unsupported shape N=%d K=%d", a.N, K);
        return -1;
    }
    const int mt16 = a.M <= 16 ? 1 : a.M <= 32 ? 2 : 4;
    const int kbw = g_tune_kbw[ki][mt16 == 1 ? 0 : mt16 == 2 ? 1 : 2];
    const int nw = K / 32 / kbw;
    const int nb16 = (epi == EPI_SWIGLU) ? 2 : 1;
    // K = 1024
    Q3_LIN_MT(1, 8, 4, PRO_NORM, EPI_STORE)
    Q3_LIN_MT(1, 4, 8, PRO_NORM, EPI_STORE)
    Q3_LIN_MT(2, 8, 4, PRO_NORM, EPI_SWIGLU)
    Q3_LIN_MT(2, 4, 8, PRO_NORM, EPI_SWIGLU)
    Q3_LIN_MT(1, 8, 4, PRO_F16, EPI_STORE)
    Q3_LIN_MT(1, 4, 8, PRO_F16, EPI_STORE)
    // K = 2048
    Q3_LIN_MT(1, 8, 8, PRO_F16, EPI_RESID)
    Q3_LIN_MT(1, 4, 16, PRO_F16, EPI_RESID)
    // K = 3072
    Q3_LIN_MT(1, 6, 16, PRO_F16, EPI_RESID)
    Q3_LOG("launch_linear: no instantiation for K=%d kbw=%d nw=%d mt16=%d nb16=%d pro=%d epi=%d", K, kbw, nw,
           mt16, nb16, pro, epi);
    return -1;
}

// ---------------------------------------------------------------------------
// ssq partials of uploaded rows
// ---------------------------------------------------------------------------
__global__ void ssq_rows_kernel(const float* __restrict__ h, float* __restrict__ ssq, int H) {
    const int r = blockIdx.x;
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += blockDim.x) {
        const float4 v = *(const float4*)(h + (size_t)r * H + k4 * 4);
        float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        s += __shfl_xor(s, 1, 4);
        s += __shfl_xor(s, 2, 4);
        if ((k4 & 3) == 0) ssq[(size_t)r * (H / 16) + (k4 >> 2)] = s;
    }
}
int launch_ssq_rows(hipStream_t s, const float* h, float* ssq, int R, int H) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(ssq_rows_kernel, dim3(R), dim3(256), 0, s, h, ssq, H);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// final RMSNorm of selected rows
// ---------------------------------------------------------------------------
__global__ void final_norm_kernel(FinalNormArgs a) {
    __shared__ float inv_sh;
    const int r = blockIdx.x;
    const int src = a.row_map ? a.row_map[r] : r;
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int p = threadIdx.x; p < a.ssq_parts; p += 64) s += a.ssq[(size_t)src * a.ssq_parts + p];
        s = wave_sum(s);
        if (threadIdx.x == 0) inv_sh = 1.0f / sqrtf(s / (float)a.H + a.eps);
    }
    __syncthreads();
    const float iv = inv_sh;
    for (int k4 = threadIdx.x; k4 < a.H / 4; k4 += blockDim.x) {
        const float4 v = *(const float4*)(a.h + (size_t)src * a.H + k4 * 4);
        const float4 g = *(const float4*)(a.gamma + k4 * 4);
        float4 o;
        o.x = (v.x * iv) * g.x;
        o.y = (v.y * iv) * g.y;
        o.z = (v.z * iv) * g.z;
        o.w = (v.w * iv) * g.w;
        if (a.out_f32) *(float4*)(a.out_f32 + (size_t)r * a.H + k4 * 4) = o;
        if (a.out_f16) {
            half_t* p = a.out_f16 + (size_t)r * a.H + k4 * 4;
            p[0] = sat_half(o.x);
            p[1] = sat_half(o.y);
            p[2] = sat_half(o.z);
            p[3] = sat_half(o.w);
        }
        if (a.out_copy) {
            *(float4*)(a.out_copy + (size_t)r * a.H + k4 * 4) = o;
            float s = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
            s += __shfl_xor(s, 1, 4);
            s += __shfl_xor(s, 2, 4);
            if ((k4 & 3) == 0) a.out_copy_ssq[(size_t)r * (a.H / 16) + (k4 >> 2)] = s;
        }
    }
}
int launch_final_norm(hipStream_t s, const FinalNormArgs& a) {
    if (a.R <= 0) return 0;
    hipLaunchKernelGGL(final_norm_kernel, dim3(a.R), dim3(256), 0, s, a);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// embedding gather with ssq partials
// ---------------------------------------------------------------------------
__device__ __forceinline__ void store_row_ssq(float* h, float* ssq, int r, int H, int k4, float4 v) {
    *(float4*)(h + (size_t)r * H + k4 * 4) = v;
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    s += __shfl_xor(s, 1, 4);
    s += __shfl_xor(s, 2, 4);
    if ((k4 & 3) == 0) ssq[(size_t)r * (H / 16) + (k4 >> 2)] = s;
}

__global__ void gather_embed_kernel(const float* __restrict__ table, int V, int H, const int* __restrict__ tok,
                                    int tok_stride, const int* __restrict__ n_frames, int frame_cap, int col,
                                    float* __restrict__ h, float* __restrict__ ssq) {
    const int r = blockIdx.x;
    int t;
    if (n_frames) {
        int f = n_frames[r] - 1;
        if (f < 0) f = 0;
        if (f >= frame_cap) f = frame_cap - 1;
        t = tok[((size_t)f * gridDim.x + r) * 16 + col];
    } else {
        t = tok[(size_t)r * tok_stride];
    }
    const bool ok = t >= 0 && t < V;
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *(const float4*)(table + (size_t)t * H + k4 * 4);
        store_row_ssq(h, ssq, r, H, k4, v);
    }
}
int launch_gather_embed(hipStream_t s, const float* table, int V, int H, const int* tok, int tok_stride,
                        const int* n_frames, int frame_cap, int col, float* h, float* ssq, int R) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(gather_embed_kernel, dim3(R), dim3(256), 0, s, table, V, H, tok, tok_stride, n_frames,
                       frame_cap, col, h, ssq);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

// ---------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(1024) attn_kernel(AttnArgs a) {
    constexpr int D = 128;
    const int r = blockIdx.x, g = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nwv = blockDim.x >> 6;
    const int slot = a.slot ? a.slot[r] : a.slot_base + r * a.slot_stride;
    const int pos = a.pos ? a.pos[r] : a.pos_base + r * a.pos_stride;
    __shared__ float qs[2][D];
    __shared__ float knew[D], vnew[D];
    __shared__ float redbuf[64];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* sc = dyn;                       // [2][n_ctx]
    float* pv = dyn + 2 * (size_t)a.n_ctx; // [nwv][2][D]
    float* row = a.qkv + (size_t)r * a.ld;
    const size_t cbase = ((size_t)slot * a.n_kv + g) * (size_t)a.n_ctx * D;
    const int T = (MODE == ATTN_FUSED) ? pos : pos + 1;  // rows read from the cache
    const int l16 = tid & 15, grp = tid >> 4, ngrp = blockDim.x >> 4;
    const half_t* kbase = a.kc + cbase;
    const half_t* vbase = a.vc + cbase;

    // ---- phase A loads first (they are consumed first; vmcnt retires in order), then the first
    // APRE cached K and V rows of every 16-lane group: one memory round trip covers T <= APRE*ngrp ----
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
    constexpr int APRE = 4;
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

    // ---- phase B: scores.  16 lanes per cached row (16 B each), 4 rows per wave step ----
    const int ntot = pos + 1;
    float q0[8], q1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        q0[j] = qs[0][l16 * 8 + j];
        q1[j] = qs[1][l16 * 8 + j];
    }
#pragma unroll
    for (int i = 0; i < APRE; i++) {
        const int t = grp + i * ngrp;
        if (t < T) {
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float kf = (float)kpre[i][j];
                d0 += q0[j] * kf;
                d1 += q1[j] * kf;
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                d0 += __shfl_xor(d0, o, 16);
                d1 += __shfl_xor(d1, o, 16);
            }
            if (l16 == 0) {
                sc[t] = d0 * a.scale;
                sc[a.n_ctx + t] = d1 * a.scale;
            }
        }
    }
#pragma unroll 4
    for (int t = grp + APRE * ngrp; t < T; t += ngrp) {
        const h8 kk = *(const h8*)(kbase + (size_t)t * D + l16 * 8);
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
        if (l16 == 0) {
            sc[t] = d0 * a.scale;
            sc[a.n_ctx + t] = d1 * a.scale;
        }
    }
    if (MODE == ATTN_FUSED && grp == 0) {
        float d0 = 0.f, d1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float kf = knew[l16 * 8 + j];
            d0 += q0[j] * kf;
            d1 += q1[j] * kf;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            d0 += __shfl_xor(d0, o, 16);
            d1 += __shfl_xor(d1, o, 16);
        }
        if (l16 == 0) {
            sc[pos] = d0 * a.scale;
            sc[a.n_ctx + pos] = d1 * a.scale;
        }
    }
    __syncthreads();

    // ---- softmax statistics (both heads) ----
    float mx0 = -INFINITY, mx1 = -INFINITY;
    for (int t = tid; t < ntot; t += blockDim.x) {
        mx0 = fmaxf(mx0, sc[t]);
        mx1 = fmaxf(mx1, sc[a.n_ctx + t]);
    }
    mx0 = wave_max(mx0);
    mx1 = wave_max(mx1);
    if (lane == 0) {
        redbuf[w] = mx0;
        redbuf[16 + w] = mx1;
    }
    __syncthreads();
    mx0 = redbuf[0];
    mx1 = redbuf[16];
    for (int i = 1; i < nwv; i++) {
        mx0 = fmaxf(mx0, redbuf[i]);
        mx1 = fmaxf(mx1, redbuf[16 + i]);
    }
    float s0 = 0.f, s1 = 0.f;
    for (int t = tid; t < ntot; t += blockDim.x) {
        const float e0 = expf(sc[t] - mx0), e1 = expf(sc[a.n_ctx + t] - mx1);
        sc[t] = e0;
        sc[a.n_ctx + t] = e1;
        s0 += e0;
        s1 += e1;
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    __syncthreads();  // everyone has read the maxima
    if (lane == 0) {
        redbuf[32 + w] = s0;
        redbuf[48 + w] = s1;
    }
    __syncthreads();
    s0 = 0.f;
    s1 = 0.f;
    for (int i = 0; i < nwv; i++) {
        s0 += redbuf[32 + i];
        s1 += redbuf[48 + i];
    }

    // ---- P.V: 16 lanes per cached row, 8 output dims per lane ----
    float a0[8], a1[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a0[j] = a1[j] = 0.f;
#pragma unroll
    for (int i = 0; i < APRE; i++) {
        const int t = grp + i * ngrp;
        if (t < T) {
            const float p0 = sc[t], p1 = sc[a.n_ctx + t];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float vf = (float)vpre[i][j];
                a0[j] += p0 * vf;
                a1[j] += p1 * vf;
            }
        }
    }
#pragma unroll 4
    for (int t = grp + APRE * ngrp; t < T; t += ngrp) {
        const h8 vv = *(const h8*)(vbase + (size_t)t * D + l16 * 8);
        const float p0 = sc[t], p1 = sc[a.n_ctx + t];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float vf = (float)vv[j];
            a0[j] += p0 * vf;
            a1[j] += p1 * vf;
        }
    }
    if (MODE == ATTN_FUSED && grp == 0) {
        const float p0 = sc[pos], p1 = sc[a.n_ctx + pos];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float vf = vnew[l16 * 8 + j];
            a0[j] += p0 * vf;
            a1[j] += p1 * vf;
        }
    }
    // the 4 row groups of a wave hold the same dims: fold them
#pragma unroll
    for (int j = 0; j < 8; j++) {
        a0[j] += __shfl_xor(a0[j], 16, 64);
        a0[j] += __shfl_xor(a0[j], 32, 64);
        a1[j] += __shfl_xor(a1[j], 16, 64);
        a1[j] += __shfl_xor(a1[j], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            pv[(w * 2 + 0) * D + lane * 8 + j] = a0[j];
            pv[(w * 2 + 1) * D + lane * 8 + j] = a1[j];
        }
    }
    __syncthreads();
    if (tid < 2 * D) {
        const int hh = tid / D, d = tid % D;
        float o = 0.f;
        for (int i = 0; i < nwv; i++) o += pv[(i * 2 + hh) * D + d];
        o = o / (hh ? s1 : s0);
        a.out[(size_t)r * (a.n_heads * D) + (size_t)(2 * g + hh) * D + d] = sat_half(o);
    }
}

int launch_attn(hipStream_t s, const AttnArgs& a, int mode) {
    if (a.R <= 0) return 0;
    if (a.n_heads != 2 * a.n_kv) {
        Q3_LOG("attn: only GQA group 2 (16q/8kv) is built, got %d/%d", a.n_heads, a.n_kv);
        return -1;
    }
    // keep the launch small: ~64K threads fill in ~2 us, every further 64K cost ~1 us of ramp
    int threads = a.threads;
    const int fit = 65536 / (a.R * a.n_kv);
    if (threads > fit) threads = fit / 64 * 64;
    if (threads < 256) threads = 256;
    if (threads > 1024) threads = 1024;
    const size_t lds = ((size_t)2 * a.n_ctx + (size_t)(threads / 64) * 2 * 128) * sizeof(float);
    if (lds > 150 * 1024) {
        Q3_LOG("attn: n_ctx=%d needs %zu B of LDS", a.n_ctx, lds);
        return -1;
    }
    dim3 grid(a.R, a.n_kv);
#define Q3_ATTN(MODE_)                                                                                   \
    {                                                                                                    \
        static bool set_ = false;                                                                        \
        if (!set_) {                                                                                     \
            Q3_HIP(hipFuncSetAttribute((const void*)attn_kernel<MODE_>,                                  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024), -1);     \
            set_ = true;                                                                                 \
        }                                                                                                \
        hipLaunchKernelGGL((attn_kernel<MODE_>), grid, dim3(mode == ATTN_PREP ? 256 : threads),          \
                           mode == ATTN_PREP ? 0 : lds, s, a);                                           \
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

__global__ void talker_sample_kernel(TalkerSampleArgs a) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int win[32];
    __shared__ int nwin;
    const int r = blockIdx.x;
    const int np = a.n_past[r];
    const int nt = a.n_text[r];
    const bool was_done = a.done[r] != 0;
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
    for (int v = threadIdx.x; v < a.V; v += blockDim.x) {
        float l = a.logits[(size_t)r * a.V + v];
        if (v >= a.audio_vocab && v != a.eos) l = -1e10f;
        if (v == a.eos) {
            if (a.ignore_eos) l = -1e10f;
            else if (nt > 0 && progress > 0.8) l += boost;
        }
        bool rep = false;
        for (int i = 0; i < nw_; i++) rep |= (win[i] == v);
        if (rep) l = l > 0.f ? __fdiv_rn(l, a.rep_penalty) : l * a.rep_penalty;
        if (l > best || (l == best && v < bidx)) {
            best = l;
            bidx = v;
        }
    }
    block_argmax(best, bidx, sv, si);
    if (threadIdx.x == 0) {
        int code = bidx;
        if (force_eos && !a.ignore_eos) code = a.eos;
        bool fin = was_done || code == a.eos || code >= a.audio_vocab || (a.max_frames > 0 && np >= a.max_frames);
        int f = a.n_frames[r];
        a.n_frames[r] = f + 1;
        if (f >= a.frame_cap) f = a.frame_cap - 1;
        int* fc = a.codes + ((size_t)f * a.R + r) * 16;
        if (fin) {
            a.done[r] = 1;
            fc[0] = -1;
        } else {
            fc[0] = code;
            a.past[r * 32 + (np & 31)] = code;
            a.n_past[r] = np + 1;
            a.pos[r] = a.pos0[r] + np;
        }
    }
}
int launch_talker_sample(hipStream_t s, const TalkerSampleArgs& a) {
    if (a.R <= 0) return 0;
    hipLaunchKernelGGL(talker_sample_kernel, dim3(a.R), dim3(256), 0, s, a);
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
                                             int ov_g = -1, int ov_tok = 0) {
    // tts_client.py:199-208: copy codec_embedding[code_0], += cp table g row, += tts_pad, in this order
    const int c0 = codes[0];
    for (int k4 = threadIdx.x; k4 < H / 4; k4 += blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 >= 0 && c0 < talker_vocab) v = *(const float4*)(talker_emb + (size_t)c0 * H + k4 * 4);
        for (int g = 0; g < n_groups; g++) {
            const int t = g == ov_g ? ov_tok : codes[1 + g];
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
        store_row_ssq(h_out, ssq_out, r, H, k4, v);
    }
}

__global__ void cp_argmax_kernel(CpArgmaxArgs a) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int tok_sh;
    const int r = blockIdx.x;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int v = threadIdx.x; v < a.V; v += blockDim.x) {
        const float l = a.logits[(size_t)r * a.V + v];
        if (l > best || (l == best && v < bidx)) {
            best = l;
            bidx = v;
        }
    }
    block_argmax(best, bidx, sv, si);
    int f = a.n_frames[r] - 1;
    if (f < 0) f = 0;
    if (f >= a.frame_cap) f = a.frame_cap - 1;
    int* fc = a.codes + ((size_t)f * a.R + r) * 16;
    if (threadIdx.x == 0) {
        fc[1 + a.group] = bidx;
        tok_sh = bidx;
    }
    __syncthreads();
    if (a.talker_emb) {
        feedback_row(fc, r, a.talker_emb, a.talker_vocab, a.cp_tables, a.V, a.n_groups, a.pad_embed,
                     a.h_out, a.ssq_out, a.H, a.group, tok_sh);
    } else if (a.next_table) {
        const int t = tok_sh;
        for (int k4 = threadIdx.x; k4 < a.H / 4; k4 += blockDim.x) {
            const float4 v = *(const float4*)(a.next_table + (size_t)t * a.H + k4 * 4);
            store_row_ssq(a.h_out, a.ssq_out, r, a.H, k4, v);
        }
    }
}
int launch_cp_argmax(hipStream_t s, const CpArgmaxArgs& a) {
    if (a.R <= 0) return 0;
    hipLaunchKernelGGL(cp_argmax_kernel, dim3(a.R), dim3(256), 0, s, a);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

__global__ void feedback_kernel(const int* codes, const float* talker_emb, int talker_vocab,
                                const float* const* cp_tables, int cp_vocab, int n_groups, const float* pad,
                                float* h_out, float* ssq_out, int H) {
    feedback_row(codes + (size_t)blockIdx.x * 16, blockIdx.x, talker_emb, talker_vocab, cp_tables, cp_vocab, n_groups,
                 pad, h_out, ssq_out, H);
}
int launch_feedback(hipStream_t s, const int* codes16, int R, const float* talker_emb, int talker_vocab,
                    const float* const* cp_tables, int cp_vocab, int n_groups, const float* pad_embed,
                    float* h_out, float* ssq_out, int H) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(feedback_kernel, dim3(R), dim3(256), 0, s, codes16, talker_emb, talker_vocab, cp_tables,
                       cp_vocab, n_groups, pad_embed, h_out, ssq_out, H);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

}  // namespace q3
