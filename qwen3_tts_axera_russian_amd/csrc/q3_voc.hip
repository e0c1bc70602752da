// q3_voc.hip -- fp32 vocoder (codec ids -> waveform) for gfx950, include/qwen3tts_voc.h.
//
// Stands where the reference calls onnxruntime on the traced Qwen3TTSTokenizerV2 decoder
// (dual_npu/vocoder_server.py:67-71; scripts/export_vocoder_traced.py:38-52).  The decoder's layer
// list is not in the reference, so the library interprets an op table from the weight container
// (tensor `voc.program`, int32 [n_ops][8]); DESIGN.md documents the table and the default one
// (split-RVQ de-quantiser -> causal conv -> x2 x2 transposed-conv upsamplers -> BigVGAN-style
// decoder: rates 8,5,4,3, residual units with dilations 1,3,9, Snake activations).
//
// Kernels:
//   rvq_kernel     16 codebook gathers per frame, summed per quantiser half, two 256->512 projections
//   conv_kernel    causal Conv1d / polyphase ConvTranspose1d as an implicit GEMM on the exact-fp32 MFMA
//                  (v_mfma_f32_32x32x2_f32): one input tile [8 ch][128+halo] is staged once in LDS (the
//                  line buffer) and read at every dilated tap offset; Snake is applied while staging,
//                  bias / residual add / clamp in the epilogue.
#include "../../include/qwen3tts_voc.h"
#include "q3_common.h"

#include <algorithm>
#include <cmath>
#include <utility>

namespace q3 {

typedef float f16v __attribute__((ext_vector_type(16)));

enum { VOP_RVQ = 1, VOP_CONV = 2, VOP_CONVT = 3, VOP_DWCONV = 4, VOP_NORM = 5, VOP_ATTN = 6, VOP_GLU = 7, VOP_EMBMEAN = 8 };
enum { VF_SNAKE = 1, VF_RES_ADD = 2, VF_RES_SAVE = 4, VF_CLAMP = 8, VF_GELU = 16 };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

struct ConvArgs {
    const float* x = nullptr;   // [B][Cin][Lin]
    float* y = nullptr;         // [B][Cout][Lin*stride]
    const float* wk = nullptr;  // [Cin/8][K][8][Mp]: rows contiguous (Mp = M rounded up to 4), M = Cout*stride virtual rows
    int Mp = 0;
    const float* bias = nullptr;
    const float* alpha = nullptr;     // [Cin] Snake: x + inv_beta * sin^2(alpha x), applied to the input
    const float* inv_beta = nullptr;
    const float* res = nullptr;       // [B][Cout][L] added in the epilogue
    int gelu = 0;                     // exact GELU applied to the input (ConvNeXt's second pointwise conv)
    int Cin = 0, M = 0, K = 0, dil = 1, Lin = 0, stride = 1, Cout = 0, clamp = 0;
    // Activations are [B][C][ld]: rows of L valid columns at a pitch ld = L rounded up to 32 floats (pitch4()), so that
    // every row starts on a 128-byte line whatever L is (the transposed convs of the decoder family trim k - s samples at both ends:
    // 64 frames -> 256 -> 2040 -> 10195 -> 40776 -> 122325 columns).  Pad columns hold junk that only ever feeds pad
    // columns: every op is causal per column (a GEMM column depends on its own B column only).
    int ldx = 0, ldy = 0;
    // transposed conv: virtual row m = co * stride + p of input column l lands at output column l * stride + p - lt
    // (lt samples trimmed on the left), kept when 0 <= that < Lout; Lc = columns of the polyphase GEMM that reach a
    // kept output (= Lin for the trims in use; inputs at l >= Lin read as zero)
    int lt = 0, Lout = 0, Lc = 0;
    // Lc of this op when the decode runs the full chunk length: the launcher's variant rule looks at it, so that a decode of
    // fewer frames (voc_run's T) sums every column in the same order as the full-length one (0: use Lc)
    int Lrule = 0;
    int n_tiles = 0, tiles_l = 0, tiles_m = 0;  // set by the launcher
    // one-tap, stride-1 convs (pointwise projections): the columns of all B chunks form ONE axis of B*Lin columns
    // (a column needs no neighbour), so 128-column tiles stay full when a chunk is only 64 columns long
    int flat_B = 0;                              // > 0: flattened, B chunks
};

static int g_voc_split = 1;    // 1 (default): split-precision fp16 MFMA path where Cin % 16 == 0; 0: exact-fp32 MFMA everywhere
static int g_voc_max_wgs = 0;  // 0 = one workgroup per tile; >0 caps the grid (persistent tile loop)
constexpr int VKC = 8;     // input channels per LDS stage
constexpr int VTN = 128;   // output columns per workgroup (4 waves x 32)
// LDS row pitches of the staged operands.  An MFMA operand read is 64 lanes x 4 B: lanes 0-31 walk 32 consecutive floats of
// row ci, lanes 32-63 of row ci + 1; the two halves hit disjoint banks when the row pitch is 32 mod 64 floats.
// (round 3, per-op profile at 32 chunks: the input tile's pitch padded that way removes every bank conflict of the 7-tap and fused
// kernels -- SQ_LDS_BANK_CONFLICT 0 -- and takes 1.5 % off the decode, 92.5 -> 91.1 ms; padding the weight rows too costs LDS
// and gains nothing)
#ifndef Q3_VOC_XPAD
#define Q3_VOC_XPAD 1
#endif
#ifndef Q3_VOC_WPAD
#define Q3_VOC_WPAD 0
#endif
__host__ __device__ constexpr int voc_wpitch(int TM) { return Q3_VOC_WPAD ? (TM % 64 == 0 ? TM + 32 : (TM % 64 == 32 ? TM : TM + 4)) : TM + 4; }
__host__ __device__ inline int voc_xpitch(int XW) { return Q3_VOC_XPAD ? ((XW + 31) / 64) * 64 + 32 : XW; }

// conv_kernel<MT, KT, KC>: MT 32-row MFMA tiles per wave, KT taps, KC input channels per LDS stage.
// Staging goes global -> LDS directly; ~4 workgroups per CU hide its latency (a register-staged software
// pipeline was tried: 199-256 VGPRs, one workgroup per SIMD, 1.6x slower at 32 chunks).
// ACT: what is applied to the input while it is staged -- 0 nothing, 1 Snake, 2 exact GELU, 3 decided at run time (a.alpha /
// a.gelu).  Compiled in for the one-tap convs (round 3, per-op profile at 32 chunks: the Snake 1 x 1 convs that close the 768- /
// 384-channel residual units 0.82 -> 0.75 and 1.27 -> 1.11 ms); for two and more taps the run-time form is the faster one
// (7-tap 126 vs 124 TFLOP/s, transposed convs 105 vs 99: the specialised kernels are scheduled worse), so those keep it.
template <int MT, int KT, int KC, bool CT = false, int ACT = 3>   // CT: transposed conv (stride > 1, no residual), stores go through an LDS slab
__global__ void __launch_bounds__(256, (KC >= 32 && MT >= 3) ? 2 : (MT >= 4 ? 3 : (MT == 3 ? 3 : 4))) conv_kernel(ConvArgs a) {
    constexpr int TM = 32 * MT, TMP = voc_wpitch(TM), Q = KC / 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int halo = (KT - 1) * a.dil;
    // staged columns: l0-HA .. l0+127, HA = the halo rounded up to 4 columns, so that the tile starts on a 16-byte boundary of
    // its row and is fetched as float4 groups (round 3: as 4-byte loads -- six per thread and stage, each with its own bounds
    // logic and LDS store -- the input tile cost as much as the five times larger weight tile; timing with either staging
    // compiled out)
    const int HA = (halo + 3) & ~3;
    const int XW = VTN + HA;
    const int XP = voc_xpitch(XW);     // their row pitch in LDS
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ws = lds;                   // [KT][KC][TMP]
    float* Xs = lds + KT * KC * TMP;   // [KC][XP]
    // persistent over output tiles (the grid may be capped, see voc_set_max_workgroups)
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int lx = tile % a.tiles_l, my = (tile / a.tiles_l) % a.tiles_m, b = tile / (a.tiles_l * a.tiles_m);
        const int l0 = lx * VTN, m0 = my * TM;
        const float* xb = a.x + (size_t)b * a.Cin * a.ldx;
        const int Lcols = a.flat_B > 0 ? a.flat_B * a.ldx : a.Lc;   // columns of the tiled axis (flattened: pads included)
        f16v acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mt][i] = 0.f;
        // Weights of all taps for KC channels.  Packed layout [ci/8][k][ci%8][Mp] (rows contiguous): a tile row is
        // TM contiguous floats, copied with 16-B loads / ds_write_b128, no transposition
        constexpr int WN = KT * KC * (TM / 4), WIT = (WN + 255) / 256;
        constexpr int LPC = 256 / KC;
        constexpr int XJ1 = (VTN / 4 + LPC - 1) / LPC;   // one-tap path: float4 groups of the input tile per thread
        const int xci = tid / LPC, xl = tid - xci * LPC;
        float4 wv[WIT];
        float4 xv1[KT == 1 ? XJ1 : 1];
        float al = 0.f, ib = 0.f;
        // Staging issues EVERY global load of a stage before the first use (fixed trip counts, clamped addresses,
        // predicated results): a load inside an `if` gets its own s_waitcnt in that branch, which made the stage a
        // chain of dependent round trips (one per 256 elements) instead of one.
        auto load_w = [&](int ci0) {
#pragma unroll
            for (int i = 0; i < WIT; i++) {
                const int idx = tid + i * 256, ic = idx < WN ? idx : 0;
                const int m4 = ic % (TM / 4), ci = (ic / (TM / 4)) % KC, k = ic / ((TM / 4) * KC);
                const int m = m0 + m4 * 4, mc = m < a.Mp ? m : 0, cg = ci0 + ci;
                wv[i] = *(const float4*)(a.wk + (unsigned)((((cg >> 3) * KT + k) * 8 + (cg & 7)) * a.Mp + mc));
                if (m >= a.Mp) wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // the input line buffer (causal: columns left of 0 are zero; Snake(0) = 0 so padding commutes).  A thread
            // stays on ONE channel of the stage (LPC lanes per channel), so its Snake parameters are two registers
            if (ACT == 1 || (ACT == 3 && a.alpha)) { al = a.alpha[ci0 + xci]; ib = a.inv_beta[ci0 + xci]; }
        };
        auto load_x1 = [&](int ci0) {   // one tap: no halo, the tile's 128 columns start 16-byte aligned
#pragma unroll
            for (int j = 0; j < (KT == 1 ? XJ1 : 1); j++) {
                const int c4 = (xl + j * LPC) * 4, c4c = c4 < VTN ? c4 : 0;
                const int gl = l0 + c4c, glc = gl < Lcols ? gl : 0;
                const int bb = a.flat_B > 0 ? glc / a.ldx : 0, l = glc - bb * a.ldx;   // (4 | ldx: a group stays in its chunk)
                xv1[j] = *(const float4*)(xb + (unsigned)((bb * a.Cin + ci0 + xci) * a.ldx + l));
            }
        };
        // One tap: the stage is short (KC / 2 MFMAs per row tile), so the NEXT stage's operands are requested into
        // registers before this stage's MFMAs and land under them (2 + 2 float4 per thread at 16 channels); with more
        // taps the prefetch registers cost a workgroup per CU (tried: 1.6x slower).
        // (32-channel stages keep the plain form: 16 + 16 more live registers put the 128-row tile 99 registers over)
        const bool al1 = KT == 1;                           // one tap: 16-byte staging (rows are 16-byte aligned: 4 | ldx)
        const bool pre1 = al1 && KC <= 16;                  // ... with the next stage prefetched
        if (pre1) {
            load_w(0);
            load_x1(0);
        }
        for (int ci0 = 0; ci0 < a.Cin; ci0 += KC) {
            __syncthreads();  // previous stage (or tile) fully consumed
            if (!pre1) {
                load_w(ci0);
                if (al1) load_x1(ci0);
            }
            auto store_w = [&]() {
#pragma unroll
                for (int i = 0; i < WIT; i++) {
                    const int idx = tid + i * 256;
                    if (idx < WN) {
                        const int m4 = idx % (TM / 4), ci = (idx / (TM / 4)) % KC, k = idx / ((TM / 4) * KC);
                        *(float4*)(Ws + (k * KC + ci) * TMP + m4 * 4) = wv[i];
                    }
                }
            };
            if (al1) {
                // operands of this stage are in registers (requested a stage ago when prefetching) -> ds_write_b128
                store_w();
#pragma unroll
                for (int j = 0; j < (KT == 1 ? XJ1 : 1); j++) {
                    const int c4 = (xl + j * LPC) * 4;
                    if (c4 < VTN) {
                        float4 v = xv1[j];
                        if (ACT == 1 || (ACT == 3 && a.alpha)) {
                            float sn;
                            sn = __sinf(al * v.x); v.x = v.x + ib * (sn * sn);
                            sn = __sinf(al * v.y); v.y = v.y + ib * (sn * sn);
                            sn = __sinf(al * v.z); v.z = v.z + ib * (sn * sn);
                            sn = __sinf(al * v.w); v.w = v.w + ib * (sn * sn);
                        }
                        if (ACT == 2 || (ACT == 3 && a.gelu)) {
                            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                        }
                        if (l0 + c4 >= Lcols) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        *(float4*)(Xs + xci * XP + c4) = v;
                    }
                }
            } else {
                // float4 groups of the tile: (VTN + HA) / 4 per channel, dealt to the LPC lanes of the channel (dilation <= 9: launcher)
                constexpr int XG = (VTN + (((KT - 1) * 9 + 3) & ~3)) / 4, XJ = (XG + LPC - 1) / LPC;
                float4 xv[XJ];
#pragma unroll
                for (int j = 0; j < XJ; j++) {
                    const int c4 = (xl + j * LPC) * 4;
                    // a group is wholly left of column 0 or not at all (l0 - HA is a multiple of 4); its row is 16-byte aligned
                    // (4 | ldx) and padded to the pitch, so a group that straddles Lin reads allocated columns
                    const int l = l0 - HA + c4, lc = (c4 < XW && l >= 0 && l < a.Lin) ? l : 0;
                    xv[j] = *(const float4*)(xb + (unsigned)((ci0 + xci) * a.ldx + lc));
                }
                store_w();
#pragma unroll
                for (int j = 0; j < XJ; j++) {
                    const int c4 = (xl + j * LPC) * 4;
                    if (c4 < XW) {
                        const int l = l0 - HA + c4;
                        float4 v = xv[j];
                        if (ACT == 1 || (ACT == 3 && a.alpha)) {
                            float sn;
                            sn = __sinf(al * v.x); v.x = v.x + ib * (sn * sn);
                            sn = __sinf(al * v.y); v.y = v.y + ib * (sn * sn);
                            sn = __sinf(al * v.z); v.z = v.z + ib * (sn * sn);
                            sn = __sinf(al * v.w); v.w = v.w + ib * (sn * sn);
                        }
                        if (ACT == 2 || (ACT == 3 && a.gelu)) {
                            v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
                        }
                        if (l < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (l >= a.Lin) v.x = 0.f;
                        if (l + 1 >= a.Lin) v.y = 0.f;
                        if (l + 2 >= a.Lin) v.z = 0.f;
                        if (l + 3 >= a.Lin) v.w = 0.f;
                        *(float4*)(Xs + xci * XP + c4) = v;
                    }
                }
            }
            __syncthreads();
            if (pre1 && ci0 + KC < a.Cin) {   // the next stage's operands: in flight under this stage's MFMAs
                load_w(ci0 + KC);
                load_x1(ci0 + KC);
            }
            // (round 3 probes of this loop, all measured per op at 32 chunks: the compiler's own schedule -- two A reads, wait, two
            // MFMAs, twice per step -- beats "all reads, one wait, four MFMAs" by 5 % (forced with sched_barrier: 91.1 -> 96.4 ms
            // per decode), and a hand-made software pipeline that issues half-step t + 1's LDS reads before half-step t's MFMAs
            // is worth 1 % at 17 spilled registers; 3 workgroups per CU run as fast as 4; co-resident workgroups started a quarter stage
            // apart: no change.  With the staging of all but the first stage compiled out (wrong results, timing only) the 7-tap
            // convs run at 139-140 TFLOP/s = 89 % and the transposed ones at 130-134 = 84 %, with or without the two barriers:
            // the loop itself holds 11-16 % of the peak back, the staging WORK (global loads, Snake, LDS writes -- not the barriers)
            // another 10 % of the 7-tap and 20 % of the transposed convs.)
#pragma unroll 1
            for (int k = 0; k < KT; k++) {
                const int off = HA - (KT - 1 - k) * a.dil + w * 32 + (lane & 31);
#pragma unroll
                for (int kk = 0; kk < KC; kk += 2) {
                    const int ci = kk + (lane >> 5);
                    const float bv = Xs[ci * XP + off];
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const float av = Ws[(k * KC + ci) * TMP + mt * 32 + (lane & 31)];
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue.  D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int gl = l0 + w * 32 + (lane & 31);
        const int be = a.flat_B > 0 ? gl / a.ldx : b;
        const int l = a.flat_B > 0 ? gl - be * a.ldx : gl;
        if constexpr (CT) {
            // transposed conv: row m = co * stride + p lands at y[co][l * stride + p] -- stored straight from the D
            // layout that is one 4-byte store per lane at a stride of `stride` floats (32-byte sectors filled a few
            // bytes at a time).  Each 32-row tile goes through LDS instead and leaves as runs of 128 * stride
            // consecutive floats per output channel.
            constexpr int TP = VTN + 1;
            float* T = lds;   // [32][TP] (the launcher sizes the LDS request for it)
            const int s = a.stride;
#pragma unroll   // (static register indices: a rolled loop would put the accumulators in scratch)
            for (int mt = 0; mt < MT; mt++) {
                __syncthreads();   // the last stage's operands (or the previous slab) are consumed
#pragma unroll
                for (int r = 0; r < 16; r++)
                    T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TP + w * 32 + (lane & 31)] = acc[mt][r];
                __syncthreads();
                const int m_lo = m0 + mt * 32, m_hi = (m_lo + 32 < a.M) ? m_lo + 32 : a.M;
                if (m_lo < m_hi) {
                    const int ncol = (a.Lc - l0 < VTN) ? a.Lc - l0 : VTN;   // live columns of this tile
                    for (int co = m_lo / s; co * s < m_hi; co++) {
                        const float bv = a.bias ? a.bias[co] : 0.f;
                        float* yrow = a.y + (size_t)b * a.Cout * a.ldy + (unsigned)(co * a.ldy);
                        for (int j = tid; j < ncol * s; j += 256) {
                            const int lc = j / s, ph = j - lc * s, m = co * s + ph;
                            const int jo = l0 * s + j - a.lt;                     // output column after the left trim
                            if (m >= m_lo && m < m_hi && jo >= 0 && jo < a.Lout) {
                                float v = T[(m - m_lo) * TP + lc] + bv;
                                if (a.clamp) v = fminf(fmaxf(v, -1.f), 1.f);
                                __builtin_nontemporal_store(v, &yrow[jo]);
                            }
                        }
                    }
                }
            }
        } else if (gl < Lcols) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < a.M) {
                        const int co = a.stride == 1 ? m : m / a.stride;
                        const int p = a.stride == 1 ? 0 : m % a.stride;
                        const int jo = l * a.stride + p - a.lt;                    // (stride 1: lt = 0, every column is kept)
                        if (a.stride != 1 && (jo < 0 || jo >= a.Lout)) continue;
                        const unsigned idx = (unsigned)((be * a.Cout + co) * a.ldy + jo);   // (launcher: < 2^31)
                        float v = acc[mt][r];
                        if (a.bias) v += a.bias[co];
                        if (a.res) v += a.res[idx];
                        if (a.clamp) v = fminf(fmaxf(v, -1.f), 1.f);
                        // streaming store: activations are far larger than L2 and are read next by another launch; a
                        // line left dirty in L2 is written back at the NEXT kernel boundary of any queue -- the frame
                        // loop's, 553 times per frame, when the vocoder runs beside it
                        __builtin_nontemporal_store(v, &a.y[idx]);
                    }
                }
        }
    }  // tile loop
}

template <int MT, int KT, int KC, bool CT = false, int ACT = -1>
static int launch_conv_t(hipStream_t s, const ConvArgs& a, int B) {
    if constexpr (!CT && KT <= 2) {
        if (a.stride > 1 && a.res == nullptr) return launch_conv_t<MT, KT, KC, true, ACT>(s, a, B);
    }
    if constexpr (ACT < 0) {      // the input activation becomes a template argument
        if (a.alpha && a.gelu) {
            Q3_LOG("voc conv: Snake and GELU on one input are not built");
            return -1;
        }
        // (only the variant the long Snake 1 x 1 convs run is specialised: every further one is another kernel to compile)
        if constexpr (!(KT == 1 && KC == 16 && MT == 4 && !CT)) return launch_conv_t<MT, KT, KC, CT, 3>(s, a, B);
        else return a.alpha ? launch_conv_t<MT, KT, KC, CT, 1>(s, a, B) : a.gelu ? launch_conv_t<MT, KT, KC, CT, 2>(s, a, B)
                                                                             : launch_conv_t<MT, KT, KC, CT, 0>(s, a, B);
    } else {
    constexpr int TM = 32 * MT, TMP = voc_wpitch(TM);
    const int halo = (KT - 1) * a.dil;
    if (a.dil > 9) {
        Q3_LOG("voc conv: dilation %d > 9 is not built", a.dil);
        return -1;
    }
    size_t lds = ((size_t)KT * KC * TMP + (size_t)KC * voc_xpitch(VTN + ((halo + 3) & ~3))) * sizeof(float);
    if (CT && lds < (size_t)32 * (VTN + 1) * sizeof(float)) lds = (size_t)32 * (VTN + 1) * sizeof(float);   // store slab
    // experiment knob: Q3_VOC_LDS_PAD=bytes raises every conv launch's LDS request, i.e. lowers the vocoder's
    // residency per CU evenly (room for the frame loop's workgroups when the two run side by side)
    static const size_t lds_pad = getenv("Q3_VOC_LDS_PAD") ? (size_t)atol(getenv("Q3_VOC_LDS_PAD")) : 0;
    if (lds_pad > lds) {
        lds = lds_pad;
        static bool attr = false;
        if (!attr) {
            Q3_HIP(hipFuncSetAttribute((const void*)conv_kernel<MT, KT, KC, CT, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), -1);
            attr = true;
        }
    }
    ConvArgs c = a;
    c.Mp = (a.M + 3) / 4 * 4;
    c.flat_B = (KT == 1 && a.stride == 1) ? B : 0;
    if ((a.ldx & 3) || (a.ldy & 3) || a.ldx < a.Lin || a.ldy < a.Lout || a.Lc < a.Lin || a.Lc > a.Lin + KT - 1) {
        Q3_LOG("voc conv: bad geometry (Lin %d pitch %d, Lout %d pitch %d, Lc %d)", a.Lin, a.ldx, a.Lout, a.ldy, a.Lc);
        return -1;
    }
    {   // the kernel indexes activations with 32-bit offsets from a.x / a.y (all chunks: the epilogue's `be` is per lane)
        if ((size_t)B * a.Cin * a.ldx >= ((size_t)1 << 31) || (size_t)B * a.Cout * a.ldy >= ((size_t)1 << 31)) {
            Q3_LOG("voc conv: activation of %d x %d x %d / %d x %d x %d floats is beyond the kernel's 32-bit indexing", B, a.Cin, a.ldx, B, a.Cout, a.ldy);
            return -1;
        }
    }
    c.tiles_l = ((c.flat_B > 0 ? a.ldx * B : a.Lc) + VTN - 1) / VTN;
    c.tiles_m = (a.M + TM - 1) / TM;
    c.n_tiles = c.tiles_l * c.tiles_m * (c.flat_B > 0 ? 1 : B);
    int grid = c.n_tiles;
    if (g_voc_max_wgs > 0 && grid > g_voc_max_wgs) grid = g_voc_max_wgs;
    hipLaunchKernelGGL((conv_kernel<MT, KT, KC, CT, ACT>), dim3(grid), dim3(256), lds, s, c);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
    }
}

template <int KT, int KC>
static int launch_conv_mt(hipStream_t s, const ConvArgs& a, int B) {
    const int t32 = (a.M + 31) / 32;  // 32-row MFMA tiles needed
    int mt = 4;
    if (t32 % 4 != 0) mt = (t32 % 3 == 0) ? 3 : (t32 % 2 == 0) ? 2 : (t32 < 4 ? t32 : 4);
    // short activations (the 12.5 Hz / 25 Hz stages: 64-256 columns per chunk): tall tiles leave most CUs without a
    // workgroup (1024 -> 512 over 2048 columns is 4 x 16 = 64 tiles of 128 rows) -- take shorter tiles until the grid
    // covers the chip
    // covers the chip twice (measured, 32 chunks: pre-transformer 5.8 -> 4.2 ms, the 4096 -> 1024 ConvNeXt conv 0.83 -> 0.62)
    static const int fill = getenv("Q3_VOC_FILL") ? atoi(getenv("Q3_VOC_FILL")) : 512;
    if (fill > 0) {
        const long cols = (KT == 1 && a.stride == 1) ? (long)a.ldx * B : (long)a.Lc;
        const long col_tiles = (cols + VTN - 1) / VTN * ((KT == 1 && a.stride == 1) ? 1 : B);
        while (mt > 1 && col_tiles * ((t32 + mt - 1) / mt) < fill) mt = (mt == 4 || mt == 2) ? mt / 2 : 1;
    }
    switch (mt) {
        case 1: return launch_conv_t<1, KT, KC>(s, a, B);
        case 2: return launch_conv_t<2, KT, KC>(s, a, B);
        case 3: return launch_conv_t<3, KT, KC>(s, a, B);
        default: return launch_conv_t<4, KT, KC>(s, a, B);
    }
}

// ---------------------------------------------------------------------------
// The decoder's last conv: C channels -> ONE output row (7 taps, Snake on the input, clamp).  On the MFMA it is a
// 32-row tile with one live row (1.6 ms per 32 chunks at 0.97 TB/s); it is a dot product per sample and HBM-bound:
// each thread owns 8 consecutive samples, walks the channels, reads the 14 inputs they need as four aligned float4
// (neighbouring threads' overlap comes from L1), applies Snake once per input and accumulates the 7 taps in f32.
// Weights are read from conv_kernel's packed layout ([C/8][7][8][Mp], row 0) with wave-uniform (scalar) loads.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv_out1_kernel(ConvArgs a) {
    const int b = blockIdx.y;
    const int l0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (l0 >= a.Lin) return;
    const float* xb = a.x + (size_t)b * a.Cin * a.ldx;
    float acc[8];
    const float b0 = a.bias ? a.bias[0] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = b0;
    for (int c = 0; c < a.Cin; c++) {
        const float* xr = xb + (size_t)c * a.ldx;      // (16-byte aligned: 4 | ldx)
        float v[16];   // columns l0 - 8 .. l0 + 7
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int l = l0 - 8 + 4 * q;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l >= 0 && l + 3 < a.Lin) t = *(const float4*)(xr + l);
            else {
                if (l >= 0 && l < a.Lin) t.x = xr[l];
                if (l + 1 >= 0 && l + 1 < a.Lin) t.y = xr[l + 1];
                if (l + 2 >= 0 && l + 2 < a.Lin) t.z = xr[l + 2];
                if (l + 3 >= 0 && l + 3 < a.Lin) t.w = xr[l + 3];
            }
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        if (a.alpha) {   // Snake(0) = 0: the causal zero padding commutes with it
            const float al = a.alpha[c], ib = a.inv_beta[c];
#pragma unroll
            for (int i = 2; i < 16; i++) {
                const float sn = __sinf(al * v[i]);
                v[i] = v[i] + ib * (sn * sn);
            }
        }
        const float* wc = a.wk + (size_t)((c >> 3) * 7 * 8 + (c & 7)) * a.Mp;   // tap k: + k * 8 * Mp
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const float wv = wc[(size_t)k * 8 * a.Mp];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = fmaf(wv, v[2 + j + k], acc[j]);   // tap k reads column l - (6 - k)
        }
    }
    float* yb = a.y + (size_t)b * a.ldy;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float o = acc[j];
        if (a.clamp) o = fminf(fmaxf(o, -1.f), 1.f);
        if (l0 + j < a.Lin) yb[l0 + j] = o;
    }
}

static int launch_conv(hipStream_t s, const ConvArgs& a, int B) {
    const int c = a.Cin;
    if (c % 8) {
        Q3_LOG("voc conv: Cin=%d is not a multiple of 8", c);
        return -1;
    }
    static const int out1 = getenv("Q3_VOC_OUT1") ? atoi(getenv("Q3_VOC_OUT1")) : 1;
    if (out1 && a.M == 1 && a.K == 7 && a.dil == 1 && a.stride == 1 && !a.res && !a.gelu && (a.ldx & 3) == 0 &&
        (size_t)B * c * a.ldx < ((size_t)1 << 31)) {
        ConvArgs k = a;
        k.Mp = 4;
        hipLaunchKernelGGL(conv_out1_kernel, dim3((a.Lin + 2047) / 2048, B), dim3(256), 0, s, k);
        Q3_HIP(hipGetLastError(), -1);
        return 0;
    }
    // one / two taps: 16-channel stages (the one-tap form prefetches the next stage's operands into registers; the
    // 32-channel variants of the 128-row tile spill 27-35 registers).  Measured at 32 chunks: 384 -> 384 k1 1.74 -> 1.28 ms,
    // 768 -> 768 k1 1.06 -> 0.84, the ConvNeXt 1024 -> 4096 convs 0.39 / 0.74 -> 0.35 / 0.64; Q3_VOC_KC_MAX=32 restores
    // the 32-channel stages (channel counts that are no multiple of 16 take them or the 8-channel ones anyway).
    // Short activations (the 12.5 / 25 / 50 Hz stages: <= 512 columns per chunk) are the other way round: their tiles are
    // 32-64 rows (launch_conv_mt shrinks them until the grid covers the chip), a stage is a handful of MFMAs, and the
    // barrier pair per stage is what they pay for -- 32-channel stages there (round 3, per-op profile at 32 chunks:
    // the pre-transformer's 1024 -> 512 projections 69-75 -> 58 us, ConvNeXt 4096 -> 1024 0.60 / 0.88 -> 0.56 / 0.83 ms).
    static const int kc_max = getenv("Q3_VOC_KC_MAX") ? atoi(getenv("Q3_VOC_KC_MAX")) : 16;
    // (the rule looks at ONE chunk's columns, never at the batch: a chunk must decode to the same bits alone and inside a
    // batch, and with two taps the stage width changes the order in which taps and channels are summed)
    const bool short_act = (a.Lrule > 0 ? a.Lrule : a.Lc) <= 512 && a.M <= 4096 && c % 32 == 0 && a.K <= 2;   // (not the 1536 -> 768 x 8 transposed conv: 2.92 -> 3.10 ms)
    if (kc_max < 32 && !short_act && c % 16 == 0 && (a.K == 1 || a.K == 2))
        return a.K == 1 ? launch_conv_mt<1, 16>(s, a, B) : launch_conv_mt<2, 16>(s, a, B);
    switch (a.K) {
        case 1: return c % 32 == 0 ? launch_conv_mt<1, 32>(s, a, B) : c % 16 == 0 ? launch_conv_mt<1, 16>(s, a, B) : launch_conv_mt<1, 8>(s, a, B);
        case 2: return c % 32 == 0 ? launch_conv_mt<2, 32>(s, a, B) : c % 16 == 0 ? launch_conv_mt<2, 16>(s, a, B) : launch_conv_mt<2, 8>(s, a, B);
        case 3: return c % 16 == 0 ? launch_conv_mt<3, 16>(s, a, B) : launch_conv_mt<3, 8>(s, a, B);
        case 7: return launch_conv_mt<7, 8>(s, a, B);
        default:
            Q3_LOG("voc conv: kernel with %d taps is not built (1, 2, 3, 7 are)", a.K);
            return -1;
    }
}

// ---------------------------------------------------------------------------
// Fused residual unit of the decoder blocks at 96 / 192 channels (the two HBM-bound stages):
//     y = x + conv1x1(Snake_b(conv7_dilated(Snake_a(x))))
// in ONE launch.  The 7-tap conv's accumulators never leave the registers: the MFMA's D layout holds, per lane,
// one column and 16 channels of each 32-row tile -- exactly a B operand of the 1x1 conv if its K axis is walked
// in the order (tile, register): step t = (mt, q) contracts channel 32 mt + (q & 3) + 8 (q >> 2) in lanes 0-31
// and that channel + 4 in lanes 32-63.  The 1x1 weights are stored at load time in that order as the matching A
// operands (w1p[row tile][t][lane]), so the second GEMM is `mfma(Ws[t*64 + lane], acc1[mt][q], acc2)`.
// HBM traffic per unit: x once (+ halo), y once -- against x, the copy kept for the residual (read + write), the
// 7-tap output (write + read), the residual read and y for the three launches it replaces.
struct ResUnitArgs {
    const float* x = nullptr;     // [B][C][Lin]
    float* y = nullptr;           // [B][C][Lin]
    const float* w7 = nullptr;    // [C/8][7][8][C]   (conv_kernel's stage-major layout)
    const float* w1p = nullptr;   // [C/32][C/2][64]  (A operands of the 1x1 conv in the order above)
    const float *b7 = nullptr, *b1 = nullptr;                    // biases (may be null)
    const float *al7 = nullptr, *ib7 = nullptr;                  // Snake of the unit's input
    const float *al1 = nullptr, *ib1 = nullptr;                  // Snake between the convs
    int Lin = 0, ld = 0, dil = 1, tiles_l = 0, n_tiles = 0;   // ld: row pitch of x and y (ConvArgs)
};

static int g_voc_fuse = 1;   // 1 (default): residual units at <= 192 channels run fused on the exact path

template <int MT>
__global__ void __launch_bounds__(256, MT <= 3 ? 3 : 2) resunit_kernel(ResUnitArgs a) {
    constexpr int C = 32 * MT, KT = 7, KC = 8, TMP = voc_wpitch(C);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int halo = (KT - 1) * a.dil, HA = (halo + 3) & ~3, XW = VTN + HA, XP = voc_xpitch(XW);   // (conv_kernel: float4 staging)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ws = lds;                   // [KT][KC][TMP]; later one row tile of the 1x1 weights [C/2][64]
    float* Xs = lds + KT * KC * TMP;   // [KC][XP]
    float* Ps = Xs + KC * XP;          // [4][C]: b7, al1, ib1, b1
    for (int i = tid; i < C; i += 256) {
        Ps[i] = a.b7 ? a.b7[i] : 0.f;
        Ps[C + i] = a.al1[i];
        Ps[2 * C + i] = a.ib1[i];
        Ps[3 * C + i] = a.b1 ? a.b1[i] : 0.f;
    }
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int lx = tile % a.tiles_l, b = tile / a.tiles_l;
        const int l0 = lx * VTN;
        const float* xb = a.x + (size_t)b * C * a.ld;
        f16v acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mt][i] = 0.f;
        // ---- the dilated 7-tap conv (same staging and MFMA loop as conv_kernel<MT, 7, 8> over all C rows) ----
        for (int ci0 = 0; ci0 < C; ci0 += KC) {
            __syncthreads();
            constexpr int WN = KT * KC * (C / 4), WIT = (WN + 255) / 256;
            float4 wv[WIT];
#pragma unroll
            for (int i = 0; i < WIT; i++) {
                const int idx = tid + i * 256, ic = idx < WN ? idx : 0;
                const int m4 = ic % (C / 4), ci = (ic / (C / 4)) % KC, k = ic / ((C / 4) * KC);
                const int cg = ci0 + ci;
                wv[i] = *(const float4*)(a.w7 + (unsigned)((((cg >> 3) * KT + k) * 8 + (cg & 7)) * C + m4 * 4));
            }
            constexpr int LPC = 256 / KC, XG = (VTN + (((KT - 1) * 9 + 3) & ~3)) / 4, XJ = (XG + LPC - 1) / LPC;
            const int xci = tid / LPC, xl = tid - xci * LPC;
            const float al = a.al7[ci0 + xci], ib = a.ib7[ci0 + xci];
            float4 xv[XJ];
#pragma unroll
            for (int j = 0; j < XJ; j++) {
                const int c4 = (xl + j * LPC) * 4;
                const int l = l0 - HA + c4, lc = (c4 < XW && l >= 0 && l < a.Lin) ? l : 0;
                xv[j] = *(const float4*)(xb + (unsigned)((ci0 + xci) * a.ld + lc));
            }
#pragma unroll
            for (int i = 0; i < WIT; i++) {
                const int idx = tid + i * 256;
                if (idx < WN) {
                    const int m4 = idx % (C / 4), ci = (idx / (C / 4)) % KC, k = idx / ((C / 4) * KC);
                    *(float4*)(Ws + (k * KC + ci) * TMP + m4 * 4) = wv[i];
                }
            }
#pragma unroll
            for (int j = 0; j < XJ; j++) {
                const int c4 = (xl + j * LPC) * 4;
                if (c4 < XW) {
                    const int l = l0 - HA + c4;
                    float4 v = xv[j];
                    float sn;
                    sn = __sinf(al * v.x); v.x = v.x + ib * (sn * sn);
                    sn = __sinf(al * v.y); v.y = v.y + ib * (sn * sn);
                    sn = __sinf(al * v.z); v.z = v.z + ib * (sn * sn);
                    sn = __sinf(al * v.w); v.w = v.w + ib * (sn * sn);
                    if (l < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (l >= a.Lin) v.x = 0.f;
                    if (l + 1 >= a.Lin) v.y = 0.f;
                    if (l + 2 >= a.Lin) v.z = 0.f;
                    if (l + 3 >= a.Lin) v.w = 0.f;
                    *(float4*)(Xs + xci * XP + c4) = v;
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int k = 0; k < KT; k++) {
                const int off = HA - (KT - 1 - k) * a.dil + w * 32 + (lane & 31);
#pragma unroll
                for (int kk = 0; kk < KC; kk += 2) {
                    const int ci = kk + (lane >> 5);
                    const float bv = Xs[ci * XP + off];
#pragma unroll
                    for (int mt = 0; mt < MT; mt++) {
                        const float av = Ws[(k * KC + ci) * TMP + mt * 32 + (lane & 31)];
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt], 0, 0, 0);
                    }
                }
            }
        }
        // ---- bias + Snake on the accumulators: they become the 1x1 conv's B operands in place ----
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int c = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[mt][r] + Ps[c];
                const float sn = __sinf(Ps[C + c] * v);
                acc[mt][r] = v + Ps[2 * C + c] * (sn * sn);
            }
        // ---- the 1x1 conv, one 32-row output tile at a time, + bias + residual ----
        const int gl = l0 + w * 32 + (lane & 31);
        const bool live = gl < a.Lin;
#pragma unroll 1
        for (int mt2 = 0; mt2 < MT; mt2++) {
            // element (row m, column gl): the row splits into a wave-uniform part (32 mt2 + the register's row: scalar
            // address arithmetic) and ONE per-lane offset; written as 16 per-lane offsets the compiler computed all of
            // them (and the 16 of the stores) at kernel entry and spilled them (36 registers, round 2)
            const unsigned lane_off = (unsigned)(4 * (lane >> 5) * a.ld + (live ? gl : 0));
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float* rowp = xb + (unsigned)((32 * mt2 + (r & 3) + 8 * (r >> 2)) * a.ld);
                res[r] = rowp[lane_off];
            }
            __syncthreads();   // the 7-tap stage (or the previous row tile) is consumed by every wave
#pragma unroll
            for (int i = 0; i < MT; i++) {   // 32 C floats = 8 C float4 = MT per thread
                const int idx = tid + i * 256;
                *(float4*)(Ws + idx * 4) = *(const float4*)(a.w1p + (unsigned)(mt2 * 32 * C + idx * 4));
            }
            __syncthreads();
            f16v o;
#pragma unroll
            for (int i = 0; i < 16; i++) o[i] = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int q = 0; q < 16; q++)
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[(mt * 16 + q) * 64 + lane], acc[mt][q], o, 0, 0, 0);
            if (live) {
                float* yb = a.y + (size_t)b * C * a.ld;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int mu = 32 * mt2 + (r & 3) + 8 * (r >> 2);
                    float* rowp = yb + (unsigned)(mu * a.ld);
                    __builtin_nontemporal_store(o[r] + Ps[3 * C + mu + 4 * (lane >> 5)] + res[r], &rowp[lane_off]);
                }
            }
        }
    }
}

template <int MT>
static int launch_resunit_t(hipStream_t s, ResUnitArgs a, int B) {
    constexpr int C = 32 * MT;
    const int halo = 6 * a.dil;
    if (a.dil > 9) return -1;
    const size_t lds = ((size_t)7 * 8 * voc_wpitch(C) + (size_t)8 * voc_xpitch(VTN + ((halo + 3) & ~3)) + 4 * C) * sizeof(float);
    a.tiles_l = (a.Lin + VTN - 1) / VTN;
    a.n_tiles = a.tiles_l * B;
    int grid = a.n_tiles;
    if (g_voc_max_wgs > 0 && grid > g_voc_max_wgs) grid = g_voc_max_wgs;
    hipLaunchKernelGGL((resunit_kernel<MT>), dim3(grid), dim3(256), lds, s, a);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

static bool resunit_channels(int c) { return c == 96 || c == 192; }

static int launch_resunit(hipStream_t s, const ResUnitArgs& a, int C, int B) {
    return C == 96 ? launch_resunit_t<3>(s, a, B) : C == 192 ? launch_resunit_t<6>(s, a, B) : -1;
}

// ---------------------------------------------------------------------------
// Split-precision path: fp32-grade products on the fp16 MFMA (16x the rate of the exact-fp32 MFMA).
// Every operand v is carried as two fp16 terms: hi = fp16(v), lo = fp16((v - hi) * 2048) (22 mantissa bits
// together; the scale keeps lo out of the subnormals).  a*b ~= hi_a*hi_b + (hi_a*lo_b + lo_a*hi_b)/2048:
// three v_mfma_f32_32x32x16_f16 with f32 accumulation (products of fp16 values are exact in f32); the
// dropped lo*lo term is 2^-22 relative.  Measured against a float64 evaluation of the same table the result
// is as close as the exact-fp32 MFMA path and torch's CPU fp32 (2e-7 of full scale; tests/test_gpu_vocoder.py).
//   snake_split_kernel  x f32 [B][C][L] -> Snake -> hi/lo planes fp16 [B][C/8][L][8] (8-channel groups,
//                       channel-minor: one 16-B record = one lane's MFMA B operand; a conv stage's input tile
//                       is two contiguous runs, a tap a row shift; the producing conv's epilogue writes whole
//                       records with lanes l, l+32 side by side)
//   conv_split_kernel   implicit GEMM, K dimension = 16 input channels per MFMA; weights split once at load
//                       into [Cin/16][tap][rows][16] planes.  Workgroup = 64 rows x 256 columns, 4 waves side
//                       by side (64 x 64 each: all share the weight fragments); staging is pure 16-B copies.
// ---------------------------------------------------------------------------
typedef _Float16 hv8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) snake_split_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                          const float* __restrict__ inv_beta, _Float16* __restrict__ xh,
                                                          _Float16* __restrict__ xl, int C, int L, int ld, int gelu,
                                                          int* __restrict__ ovf) {
    const int l = blockIdx.x * 256 + threadIdx.x, cg = blockIdx.y, b = blockIdx.z;   // cg: 8-channel group
    if (l >= L) return;
    const float* xp = x + ((size_t)b * C + cg * 8) * ld + l;     // f32 rows at pitch ld; the planes are dense in L
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = xp[(size_t)j * ld];
    if (alpha) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float sn = __sinf(alpha[cg * 8 + j] * v[j]);
            v[j] = v[j] + inv_beta[cg * 8 + j] * (sn * sn);
        }
    }
    if (gelu) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = gelu_erf(v[j]);
    }
    hv8 h, lo;
    bool big = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        big |= !(fabsf(v[j]) <= 65504.f);   // beyond the hi term's range (or NaN): the split form cannot carry it
        const _Float16 hi = (_Float16)fminf(fmaxf(v[j], -65504.f), 65504.f);
        h[j] = hi;
        lo[j] = (_Float16)((v[j] - (float)hi) * 2048.0f);
    }
    const size_t o = (((size_t)b * (C >> 3) + cg) * L + l) * 8;
    *(hv8*)(xh + o) = h;
    *(hv8*)(xl + o) = lo;
    if (big) *ovf = 1;
}

struct SplitArgs {
    const _Float16* xh = nullptr;        // [B][Cin/8][Lin][8]
    const _Float16* xl = nullptr;
    float* y = nullptr;                  // [B][Cout][Lin*stride] f32
    const _Float16* w_hi = nullptr;      // [Cin/16][K][Mp][16]  (Mp = rows padded to 128)
    const _Float16* w_lo = nullptr;
    const float* bias = nullptr;
    const float* res = nullptr;
    // optional second output (stride 1 only): the result already in the NEXT conv's input form -- its Snake
    // applied, split into hi/lo planes [B][Cout/8][Lin][8] -- so no separate pass re-reads it
    _Float16* oh = nullptr;
    _Float16* ol = nullptr;
    const float* oalpha = nullptr;
    const float* oinv_beta = nullptr;
    int* ovf = nullptr;                  // set to 1 when an output plane value leaves the fp16 range
    int Cin = 0, M = 0, Mp = 0, dil = 1, Lin = 0, stride = 1, Cout = 0, clamp = 0, B = 0;
    int ldy = 0, lt = 0, Lout = 0, Lc = 0;   // f32 output pitch, left trim / kept outputs / GEMM columns (ConvArgs)
    int n_tiles = 0, tiles_l = 0, tiles_m = 0;
    int my_fast = 0;   // tile order, see conv_split_kernel
};

constexpr int SKC = 16;  // input channels per k-step

// KT taps; KS 16-channel k-steps per LDS stage (few-tap convs stage several, so a barrier pair buys more MFMAs);
// MW 32-row MFMA tiles per workgroup (rows = 32*MW: 96 divides every channel count of the decoder, so the
// input tile is read by Cout/96 workgroups instead of Cout/64 and no row is padding)
// NJ 32-column tiles per wave (workgroup = 4 waves side by side = 128*NJ columns): 2 for the MFMA-bound layers,
// 1 for the few-tap HBM-bound ones, whose half-size accumulators let a third workgroup per CU overlap the phases
template <int KT, int KS, int MW, int NJ>
__global__ void __launch_bounds__(256, NJ == 1 ? 3 : 2) conv_split_kernel(SplitArgs a) {
    constexpr int STM = 32 * MW, STN = 128 * NJ;
    constexpr int UNR = (MW == 3 && KT == 7) ? 1 : KS * KT;   // unroll of the tap loop
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int halo = (KT - 1) * a.dil;
    const int XW = STN + halo;
    extern __shared__ __attribute__((aligned(16))) char slds[];
    _Float16* Wh = (_Float16*)slds;                    // [KS][KT][STM][16]
    _Float16* Wl = Wh + KS * KT * STM * SKC;
    _Float16* Xh = Wl + KS * KT * STM * SKC;           // [KS][2][XW][8]: per k-step its two 8-channel groups
    _Float16* Xl = Xh + (size_t)KS * XW * SKC;
    const int C16 = a.Cin >> 4;
    const hv8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // Tile order.  Big weights (my_fast = 0): columns fastest, then chunk, row tile slowest -- at any moment the
    // chip works on one or two row tiles, whose weights stay in every XCD's L2 while the input tiles stream.
    // Small weights (my_fast = 1, they fit every L2 whole): XCD x (= workgroup id mod 8 under round-robin
    // placement) owns the column tiles x, x+8, ... and walks each one's row tiles back to back, so the input
    // tile comes from HBM once and from that XCD's L2 for the other row tiles.
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        int lx, b, my;
        if (a.my_fast) {
            const int xcd = tile & 7, j = tile >> 3;
            my = j % a.tiles_m;
            const int cg = (j / a.tiles_m) * 8 + xcd;
            if (cg >= a.tiles_l * a.B) continue;
            lx = cg % a.tiles_l;
            b = cg / a.tiles_l;
        } else {
            lx = tile % a.tiles_l;
            b = (tile / a.tiles_l) % a.B;
            my = tile / (a.tiles_l * a.B);
        }
        const int l0 = lx * STN, m0 = my * STM;
        f16v acc[MW][NJ], accx[MW][NJ];
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    acc[i][j][r] = 0.f;
                    accx[i][j][r] = 0.f;
                }
        for (int cb = 0; cb < C16; cb += KS) {
            __syncthreads();
            // weights: per (k-step, tap) one contiguous 2 KiB run of each plane
            for (int idx = tid; idx < KS * KT * STM * 2; idx += 256) {      // 16-byte pieces
                const int sk = idx / (STM * 2), rem = idx - sk * (STM * 2);      // sk = ks*KT + k
                const size_t g = ((size_t)(cb * KT + sk) * a.Mp + m0) * SKC + rem * 8;
                *(hv8*)(Wh + idx * 8) = *(const hv8*)(a.w_hi + g);
                *(hv8*)(Wl + idx * 8) = *(const hv8*)(a.w_lo + g);
            }
            // input: columns l0-halo .. l0+255 of the stage's 2*KS 8-channel groups, one contiguous run each
            for (int idx = tid; idx < KS * 2 * XW; idx += 256) {
                const int grp = idx / XW, col = idx - grp * XW;
                const int l = l0 - halo + col;
                hv8 vh = zero8, vl = zero8;
                if (l >= 0 && l < a.Lin) {
                    const size_t g = ((((size_t)b * C16 + cb) * 2 + grp) * a.Lin + l) * 8;
                    vh = *(const hv8*)(a.xh + g);
                    vl = *(const hv8*)(a.xl + g);
                }
                *(hv8*)(Xh + idx * 8) = vh;
                *(hv8*)(Xl + idx * 8) = vl;
            }
            __syncthreads();
            // the 96-row 7-tap form sits at the 256-register limit: walking its taps one at a time keeps it from spilling
#pragma unroll UNR
            for (int sk = 0; sk < KS * KT; sk++) {
                const int ks = sk / KT, k = sk % KT;
                const int off = k * a.dil;   // tap k reads column l - (KT-1-k)*dil = staged column (l-l0) + k*dil
                hv8 ah[MW], al[MW], bh[NJ], bl[NJ];
#pragma unroll
                for (int i = 0; i < MW; i++) {
                    const int row = i * 32 + (lane & 31);
                    ah[i] = *(const hv8*)(Wh + (sk * STM + row) * SKC + (lane >> 5) * 8);
                    al[i] = *(const hv8*)(Wl + (sk * STM + row) * SKC + (lane >> 5) * 8);
                }
#pragma unroll
                for (int i = 0; i < NJ; i++) {
                    const int col = w * (32 * NJ) + i * 32 + (lane & 31) + off;
                    bh[i] = *(const hv8*)(Xh + ((size_t)(ks * 2 + (lane >> 5)) * XW + col) * 8);
                    bl[i] = *(const hv8*)(Xl + ((size_t)(ks * 2 + (lane >> 5)) * XW + col) * 8);
                }
#pragma unroll
                for (int i = 0; i < MW; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                        accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                    }
            }
        }
        typedef _Float16 hv4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int l = l0 + w * (32 * NJ) + j * 32 + (lane & 31);
                if (l < a.Lc) {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; r++) {     // the residual reads of the whole 32x32 tile go out together
                        const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        rv[r] = 0.f;
                        if (a.res && m < a.M) {
                            const int co = a.stride == 1 ? m : m / a.stride;
                            const int p = a.stride == 1 ? 0 : m % a.stride;
                            rv[r] = a.res[((size_t)b * a.Cout + co) * a.ldy + (size_t)l * a.stride + p];   // (stride 1 only)
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; g++) {      // accumulator registers 4g..4g+3 = 4 consecutive rows
                        const int mg = m0 + i * 32 + 8 * g + 4 * (lane >> 5);
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int m = mg + q, r = 4 * g + q;
                            v[q] = acc[i][j][r] + accx[i][j][r] * (1.0f / 2048.0f);
                            if (m < a.M) {
                                const int co = a.stride == 1 ? m : m / a.stride;
                                const int p = a.stride == 1 ? 0 : m % a.stride;
                                const int jo = l * a.stride + p - a.lt;          // output column after the left trim
                                const size_t idx = ((size_t)b * a.Cout + co) * a.ldy + (size_t)(jo > 0 ? jo : 0);
                                if (a.bias) v[q] += a.bias[co];
                                v[q] += rv[r];
                                if (a.clamp) v[q] = fminf(fmaxf(v[q], -1.f), 1.f);
                                if (a.y && jo >= 0 && jo < a.Lout) a.y[idx] = v[q];
                            }
                        }
                        if (a.oh && mg < a.M) {
                            hv4 vh, vl;
                            bool big = false;
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                float t = v[q];
                                if (a.oalpha) {
                                    const float sn = __sinf(a.oalpha[mg + q] * t);
                                    t = t + a.oinv_beta[mg + q] * (sn * sn);
                                }
                                big |= !(fabsf(t) <= 65504.f);
                                const _Float16 hi = (_Float16)fminf(fmaxf(t, -65504.f), 65504.f);
                                vh[q] = hi;
                                vl[q] = (_Float16)((t - (float)hi) * 2048.0f);
                            }
                            if (big) *a.ovf = 1;
                            const size_t o = (((size_t)b * (a.Cout >> 3) + (mg >> 3)) * a.Lin + l) * 8 + (mg & 7);
                            *(hv4*)(a.oh + o) = vh;
                            *(hv4*)(a.ol + o) = vl;
                        }
                    }
                }
            }
    }
}

template <int KT, int KS, int MW, int NJ>
static int launch_conv_split_t(hipStream_t s, const SplitArgs& a, int B) {
    constexpr int STM = 32 * MW, STN = 128 * NJ;
    if (a.dil > 9) return -1;
    const int halo = (KT - 1) * a.dil;
    const size_t lds = ((size_t)2 * KS * KT * STM * SKC + (size_t)2 * KS * (STN + halo) * SKC) * sizeof(_Float16);
    static bool set_ = false;
    if (!set_) {
        Q3_HIP(hipFuncSetAttribute((const void*)conv_split_kernel<KT, KS, MW, NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), -1);
        set_ = true;
    }
    if (lds > 80 * 1024 || (a.Cin / 16) % KS || a.Mp % STM) return -1;   // <= 80 KB: two workgroups per CU
    SplitArgs c = a;
    c.B = B;
    c.tiles_l = (a.Lc + STN - 1) / STN;
    c.tiles_m = (a.M + STM - 1) / STM;
    c.n_tiles = c.tiles_l * c.tiles_m * B;
    // both planes of all taps of the weights: small enough to live in every XCD's 4 MiB L2 beside the stream?
    c.my_fast = c.tiles_m > 1 && (size_t)a.Cin * a.Mp * KT * 4 <= (size_t)2 << 20;
    if (c.my_fast) c.n_tiles = (c.tiles_l * B + 7) / 8 * 8 * c.tiles_m;
    int grid = c.n_tiles;
    if (g_voc_max_wgs > 0 && grid > g_voc_max_wgs) grid = g_voc_max_wgs;
    hipLaunchKernelGGL((conv_split_kernel<KT, KS, MW, NJ>), dim3(grid), dim3(256), lds, s, c);
    Q3_HIP(hipGetLastError(), -1);
    return 0;
}

static int g_voc_narrow_k1 = 1;   // 128-column tiles (three workgroups per CU) for the 1-tap convs

template <int MW>
static int launch_conv_split_m(hipStream_t s, const SplitArgs& a, int K, int B) {
    const int c16 = a.Cin / 16;
    switch (K) {
        case 1:
            if (g_voc_narrow_k1 && MW == 3)
                return c16 % 3 == 0 ? launch_conv_split_t<1, 3, MW, 1>(s, a, B) : c16 % 2 == 0 ? launch_conv_split_t<1, 2, MW, 1>(s, a, B) : launch_conv_split_t<1, 1, MW, 1>(s, a, B);
            return c16 % 3 == 0 ? launch_conv_split_t<1, 3, MW, 2>(s, a, B) : c16 % 2 == 0 ? launch_conv_split_t<1, 2, MW, 2>(s, a, B) : launch_conv_split_t<1, 1, MW, 2>(s, a, B);
        case 2:
            if (g_voc_narrow_k1 && MW == 3 && a.Cin <= 384)   // the HBM-bound transposed convs (measured: 1536 -> 768 loses)
                return c16 % 2 == 0 ? launch_conv_split_t<2, 2, MW, 1>(s, a, B) : launch_conv_split_t<2, 1, MW, 1>(s, a, B);
            return c16 % 2 == 0 ? launch_conv_split_t<2, 2, MW, 2>(s, a, B) : launch_conv_split_t<2, 1, MW, 2>(s, a, B);
        case 3: return c16 % 2 == 0 && MW == 2 ? launch_conv_split_t<3, 2, MW, 2>(s, a, B) : launch_conv_split_t<3, 1, MW, 2>(s, a, B);
        case 7:
            if (g_voc_narrow_k1 && a.Cin <= 192) return launch_conv_split_t<7, 1, MW, 1>(s, a, B);   // the HBM-bound blocks: -4..9 %
            return launch_conv_split_t<7, 1, MW, 2>(s, a, B);
        default: return -1;
    }
}

static int launch_conv_split(hipStream_t s, const SplitArgs& a, int K, int B) {
    // 96-row tiles where they tile the rows exactly (every channel count of the decoder blocks) and still
    // give the chip enough workgroups: the input tile is read by Cout/96 workgroups instead of Cout/64
    const long tiles96 = (long)((a.Lc + 255) / 256) * (a.M / 96) * B;
    const bool fits96 = a.M % 96 == 0 && a.Mp % 96 == 0;
    const bool use96 = fits96 && (a.Mp % 64 != 0 || tiles96 >= 512);
    return use96 ? launch_conv_split_m<3>(s, a, K, B) : launch_conv_split_m<2>(s, a, K, B);
}

// ---------------------------------------------------------------------------
// The small f32 ops of the published decoder's transformer / ConvNeXt stages (activations [B][C][L], L <= a few
// hundred columns: latency-sized kernels, one thread per output or per column).
// ---------------------------------------------------------------------------
// causal depthwise conv: y[c][l] = bias[c] + sum_k w[c][k] * x[c][l - (K-1-k)]
__global__ void __launch_bounds__(256) dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int C, int L, int ld, int K) {
    const int l = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const float* xr = x + ((size_t)b * C + c) * ld;
    float acc = bias ? bias[c] : 0.f;
    for (int k = 0; k < K; k++) {
        const int ls = l - (K - 1 - k);
        if (ls >= 0) acc += w[c * K + k] * xr[ls];
    }
    y[((size_t)b * C + c) * ld + l] = acc;
}

// RMSNorm (kind 0) / LayerNorm (kind 1) over the channels of every column.  Workgroup = 64 columns x 16 channel
// lanes: a wave reads 64 consecutive columns of one channel (coalesced), the 16 partial sums of a column meet in LDS.
// Two-pass variance (mean first), like the reference implementation.
__global__ void __launch_bounds__(1024) chan_norm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int C, int L,
                                                         int ld, int kind, float eps) {
    __shared__ float part[16][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int l = blockIdx.x * 64 + col, b = blockIdx.y;
    const bool ok = l < L;
    const float* xc = x + (size_t)b * C * ld + (ok ? l : 0);
    auto column_sum = [&](float v) -> float {     // sum over the 16 channel lanes of a column, identical in all of them
        part[g][col] = v;
        __syncthreads();
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) s_ += part[i][col];
        __syncthreads();
        return s_;
    };
    float mu = 0.f;
    if (kind == 1) {
        float s_ = 0.f;
        for (int c = g; c < C; c += 16) s_ += xc[(size_t)c * ld];
        mu = column_sum(s_) / (float)C;
    }
    float ss = 0.f;
    for (int c = g; c < C; c += 16) {
        const float d = xc[(size_t)c * ld] - mu;
        ss += d * d;
    }
    const float inv = 1.0f / sqrtf(column_sum(ss) / (float)C + eps);
    if (!ok) return;
    float* yc = y + (size_t)b * C * ld + l;
    for (int c = g; c < C; c += 16) {
        float v = (xc[(size_t)c * ld] - mu) * inv * w[c];
        if (bias) v += bias[c];
        yc[(size_t)c * ld] = v;
    }
}

// x = [q | k | v] (head-major channels, [3*H*D][L]) -> causal sliding-window attention with rotate-half RoPE
// (positions = columns of the chunk).  One wave per (query column, head); lane j owns the pair (j, j + D/2).
__global__ void __launch_bounds__(64) voc_attn_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int D, int Lv,
                                                      int L, int window, float theta) {
    // L: row pitch (every row index below is scaled by it); Lv valid columns = the grid's x extent
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, j = threadIdx.x;
    const int half = D / 2, HD = H * D;
    const bool on = j < half;
    const float* xb = x + (size_t)b * 3 * HD * L;
    const float inv_freq = on ? __powf(theta, -2.0f * (float)j / (float)D) : 0.f;
    auto rope = [&](const float* base, int pos, float& a, float& c) {   // rows (j, j+half) of a head at column pos
        float x0 = 0.f, x1 = 0.f;
        if (on) {
            x0 = base[(size_t)j * L + pos];
            x1 = base[(size_t)(j + half) * L + pos];
        }
        float sn, cs;
        __sincosf((float)pos * inv_freq, &sn, &cs);
        a = x0 * cs - x1 * sn;
        c = x1 * cs + x0 * sn;
    };
    float q0, q1;
    rope(xb + (size_t)(h * D) * L, i, q0, q1);
    const float scale = 1.0f / sqrtf((float)D);
    float m = -INFINITY, lsum = 0.f, o0 = 0.f, o1 = 0.f;
    const int t0 = i - window + 1 > 0 ? i - window + 1 : 0;
    for (int t = t0; t <= i; t++) {
        float k0, k1;
        rope(xb + (size_t)(HD + h * D) * L, t, k0, k1);
        float sc = q0 * k0 + q1 * k1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o, 64);
        sc *= scale;
        const float mn = fmaxf(m, sc), corr = __expf(m - mn), p = __expf(sc - mn);
        float v0 = 0.f, v1 = 0.f;
        if (on) {
            const float* vb = xb + (size_t)(2 * HD + h * D) * L;
            v0 = vb[(size_t)j * L + t];
            v1 = vb[(size_t)(j + half) * L + t];
        }
        lsum = lsum * corr + p;
        o0 = o0 * corr + p * v0;
        o1 = o1 * corr + p * v1;
        m = mn;
    }
    if (on) {
        float* yb = y + ((size_t)b * HD + h * D) * L;
        yb[(size_t)j * L + i] = o0 / lsum;
        yb[(size_t)(j + half) * L + i] = o1 / lsum;
    }
}

// The same attention for a chunk whose q, k, v of one head fit in LDS (3 * L * D floats <= 64 KiB: the 64-column
// chunks of the pre-transformer): one workgroup per (head, chunk) applies RoPE once per element while staging, then
// every query is owned by 4 threads that split its keys 4 ways (online softmax each, merged by shuffles).
__global__ void __launch_bounds__(256) voc_attn_tile_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int D,
                                                            int L, int ld, int window, float theta) {
    extern __shared__ float sm[];            // q[L][D+1] | k[L][D+1] | v[L][D+1]  (+1: conflict-free row walks)
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int half = D / 2, HD = H * D, DP = D + 1;
    float *qs = sm, *ks = sm + (size_t)L * DP, *vs = sm + (size_t)2 * L * DP;
    const float* xb = x + (size_t)b * 3 * HD * ld;
    // stage: element (d, l) of q / k rotated with its partner (d +- half, l); consecutive threads = consecutive l
    for (int idx = tid; idx < half * L; idx += 256) {
        const int j = idx / L, l = idx - j * L;
        float sn, cs;
        __sincosf((float)l * __powf(theta, -2.0f * (float)j / (float)D), &sn, &cs);
#pragma unroll
        for (int which = 0; which < 2; which++) {
            const float* base = xb + (size_t)(which * HD + h * D) * ld;
            const float x0 = base[(size_t)j * ld + l], x1 = base[(size_t)(j + half) * ld + l];
            float* dst = which == 0 ? qs : ks;
            dst[l * DP + j] = x0 * cs - x1 * sn;
            dst[l * DP + j + half] = x1 * cs + x0 * sn;
        }
    }
    for (int idx = tid; idx < D * L; idx += 256) {
        const int d = idx / L, l = idx - d * L;
        vs[l * DP + d] = xb[(size_t)(2 * HD + h * D + d) * ld + l];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)D);
    for (int i0 = 0; i0 < L; i0 += 64) {       // 64 queries per round: thread = (query, key lane)
        const int i = i0 + (tid >> 2), kl = tid & 3;
        float m = -INFINITY, lsum = 0.f;
        float o[64];                            // D <= 64 on this path
#pragma unroll
        for (int d = 0; d < 64; d++) o[d] = 0.f;
        if (i < L) {
            const int t0 = i - window + 1 > 0 ? i - window + 1 : 0;
            for (int t = t0 + kl; t <= i; t += 4) {
                float sc = 0.f;
                for (int d = 0; d < D; d++) sc += qs[i * DP + d] * ks[t * DP + d];
                sc *= scale;
                const float mn = fmaxf(m, sc), corr = __expf(m - mn), pw = __expf(sc - mn);
                lsum = lsum * corr + pw;
#pragma unroll
                for (int d = 0; d < 64; d++)
                    if (d < D) o[d] = o[d] * corr + pw * vs[t * DP + d];
                m = mn;
            }
        }
        // merge the 4 key lanes of a query (lanes xor 1, 2); a lane that saw no key has m = -inf, l = 0
#pragma unroll
        for (int sft = 1; sft <= 2; sft <<= 1) {
            const float om = __shfl_xor(m, sft, 64), ol = __shfl_xor(lsum, sft, 64);
            const float mn = fmaxf(m, om);
            const float c0 = m == -INFINITY ? 0.f : __expf(m - mn), c1 = om == -INFINITY ? 0.f : __expf(om - mn);
            lsum = lsum * c0 + ol * c1;
#pragma unroll
            for (int d = 0; d < 64; d++) {
                const float od = __shfl_xor(o[d], sft, 64);
                o[d] = o[d] * c0 + od * c1;
            }
            m = mn;
        }
        if (i < L) {
            float* yb = y + ((size_t)b * HD + h * D) * ld + i;
            // (static register indices: `o[d]` with d starting at the lane's kl put the 64 accumulators in scratch --
            // 272 B per thread, 0.3 GB of scratch writes per launch by PMC)
            const float inv = 1.0f / lsum;
#pragma unroll
            for (int d = 0; d < 64; d++)
                if (d < D && (d & 3) == kl) yb[(size_t)d * ld] = o[d] * inv;
        }
    }
}

// y[c][l] = act(x[c][l]) * x[C + c][l]; act 0 SiLU, 1 GELU
__global__ void __launch_bounds__(256) glu_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int L, int ld, int act) {
    const int l = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const float g = x[((size_t)b * 2 * C + c) * ld + l], u = x[((size_t)b * 2 * C + C + c) * ld + l];
    y[((size_t)b * C + c) * ld + l] = (act == 0 ? g / (1.0f + __expf(-g)) : gelu_erf(g)) * u;
}

// Split residual VQ de-quantisation: codes i64 [B][T][NQ] -> y [B][OUT][T].
// Quantiser 0 (semantic) and 1..NQ-1 (acoustic) each sum their codebook rows ([NQ][CB][DIM]) and go
// through their own DIM->OUT projection (1x1 conv without bias); the two results add.
__global__ void __launch_bounds__(256) rvq_kernel(const int64_t* __restrict__ codes, const float* __restrict__ cb,
                                                  const float* __restrict__ p_sem, const float* __restrict__ p_ac,
                                                  float* __restrict__ y, int T, int ld, int NQ, int CB, int DIM, int OUT) {
    extern __shared__ float e[];  // [2][DIM]
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t* c = codes + ((size_t)b * T + t) * 16;   // 16 ids per frame in a request (vocoder_server.py:78), the first NQ are used
    for (int d = tid; d < DIM; d += blockDim.x) {
        float s0 = 0.f, s1 = 0.f;
        const int64_t c0 = c[0];
        if (c0 >= 0 && c0 < CB) s0 = cb[((size_t)0 * CB + c0) * DIM + d];
        for (int q = 1; q < NQ; q++) {
            const int64_t cq = c[q];
            if (cq >= 0 && cq < CB) s1 += cb[((size_t)q * CB + cq) * DIM + d];
        }
        e[d] = s0;
        e[DIM + d] = s1;
    }
    __syncthreads();
    for (int o = tid; o < OUT; o += blockDim.x) {
        float acc = 0.f;
        for (int d = 0; d < DIM; d++) acc += p_sem[(size_t)o * DIM + d] * e[d];
        for (int d = 0; d < DIM; d++) acc += p_ac[(size_t)o * DIM + d] * e[DIM + d];
        y[((size_t)b * OUT + o) * ld + t] = acc;
    }
}

// The embedding-mean front of the decoder family's Omni form (Qwen3OmniMoeCode2Wav.forward):
// y[b][c][t] = mean_q table[q * CB + codes[b][t][q]][c]; an id outside [0, CB) contributes zero.
__global__ void __launch_bounds__(256) embmean_kernel(const int64_t* __restrict__ codes, const float* __restrict__ tab,
                                                      float* __restrict__ y, int T, int ld, int NQ, int NQS, int CB, int DIM) {
    const int t = blockIdx.x, b = blockIdx.y;
    const int64_t* c = codes + ((size_t)b * T + t) * NQS;    // NQS ids per frame in the request, the first NQ are used
    for (int d = threadIdx.x; d < DIM; d += blockDim.x) {
        float s_ = 0.f;
        for (int q = 0; q < NQ; q++) {
            const int64_t cq = c[q];
            if (cq >= 0 && cq < CB) s_ += tab[((size_t)q * CB + cq) * DIM + d];
        }
        y[((size_t)b * DIM + d) * ld + t] = s_ / (float)NQ;
    }
}

// ---------------------------------------------------------------------------
// Batched chunk walk (voc_synthesize_batch): the reference assembles an utterance from its 64-frame chunks on the host
// (vocoder_server.py:84-117: first chunk kept, every next one either cross-faded over 16 frames with the tail of what
// is there, or -- shorter than the overlap -- appended).  Here the chunks of MANY utterances are decoded max_batch at a
// time and placed by two launches per batch: every chunk copies its samples behind the blended head to its position,
// then every blended chunk folds its head into the 30 720 samples already there (written by its predecessor's copy,
// this batch or an earlier one).  A chunk shorter than twice the overlap has no successor (the walk steps by
// chunk - 16 frames), so no sample is blended twice and the two-pass order reproduces the sequential result bit for bit:
// float32 products and one float32 add, never fused (numpy: result[-OV:] * fade_out + chunk[:OV] * fade_in), the fade
// np.linspace(1, 0, OV, dtype=float32) evaluated in double exactly as numpy does.
// ---------------------------------------------------------------------------
struct ChunkPlace {
    int row;            // row of the decode batch's output
    int len;            // samples of the chunk after the reference's slice (min(frames * 1920, chunk_samples))
    int head;           // 0: plain append; OV: the first OV samples are cross-faded into what is already there
    long long dst;      // sample index in the batch output buffer where the chunk's first sample lands
};

__global__ void __launch_bounds__(256) voc_place_copy_kernel(const float* __restrict__ dec, int pitch, const ChunkPlace* __restrict__ pl,
                                                             float* __restrict__ out) {
    const ChunkPlace p = pl[blockIdx.y];
    const float* src = dec + (size_t)p.row * pitch;
    for (int i = p.head + blockIdx.x * 256 + threadIdx.x; i < p.len; i += gridDim.x * 256) out[p.dst + i] = src[i];
}

__global__ void __launch_bounds__(256) voc_place_blend_kernel(const float* __restrict__ dec, int pitch, const ChunkPlace* __restrict__ pl,
                                                              float* __restrict__ out, int OV) {
    const ChunkPlace p = pl[blockIdx.y];
    if (p.head == 0) return;
    const float* src = dec + (size_t)p.row * pitch;
    const double step = -1.0 / (double)(OV - 1);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < OV; i += gridDim.x * 256) {
        const float fo = (i == OV - 1) ? 0.0f : (float)(1.0 + (double)i * step);
        const float fi = __fsub_rn(1.0f, fo);
        out[p.dst + i] = __fadd_rn(__fmul_rn(out[p.dst + i], fo), __fmul_rn(src[i], fi));
    }
}

// np.clip(audio * 32767, -32768, 32767).astype(np.int16) (vocoder_server.py:175): float32 product, truncation toward zero
__global__ void __launch_bounds__(256) voc_to_int16_kernel(const float* __restrict__ x, int16_t* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = __fmul_rn(x[i], 32767.0f);
    v = v < -32768.0f ? -32768.0f : (v > 32767.0f ? 32767.0f : v);
    y[i] = (int16_t)v;
}

struct VocOp {
    int op = 0, cin = 0, cout = 0, k = 0, p0 = 0, flags = 0, nq = 0, cb = 0;
    float *w = nullptr, *bias = nullptr, *alpha = nullptr, *inv_beta = nullptr;  // device
    int kind = 0, heads = 0, head_dim = 0, window = 0;   // NORM kind; ATTN geometry
    float eps = 0.f, theta = 10000.f;
    _Float16 *w_hi = nullptr, *w_lo = nullptr;  // split-precision weights (null: exact path only)
    int Mp128 = 0;
    float *p_sem = nullptr, *p_ac = nullptr;
    float* w1p = nullptr;   // 1x1 conv closing a residual unit: A operands in resunit_kernel's K order
    int lt = 0, rt = 0;     // transposed conv: samples trimmed from the (L - 1) * stride + k outputs, left / right
};

// row pitch of an activation: L rounded up to 32 floats = one 128-byte line, so that rows (and the 32-column runs a wave
// stores) start on a line whatever L is -- with a 16-byte pitch the fused units' stores straddled two lines and WRITE_SIZE
// counted 4.5-4.6 B per element instead of 4.00 (profiles/r03_pmc_vocoder.md); the kernels need 4 | pitch only
static inline long pitch4(long L) { return (L + 31) & ~31L; }
// kept outputs of a transposed conv over L input columns
static inline long convt_out(const VocOp& op, long L) { return (L - 1) * op.p0 + op.k - op.lt - op.rt; }
// columns of its polyphase GEMM that reach a kept output (virtual row p of column l lands at l * s + p - lt)
static inline long convt_cols(const VocOp& op, long L) {
    const long lc = (convt_out(op, L) + op.lt + op.p0 - 1) / op.p0;
    return lc < L ? L : lc;
}

struct Voc {
    int device = 0;           // the HIP device the handle was loaded on (q3_set_device before voc_load); entry points bind their thread to it
    int chunk = 64, max_batch = 1, upsample = 1;
    long chunk_samples = 0;   // what one decode of `chunk` frames yields (<= chunk * upsample: the transposed convs trim)
    std::vector<VocOp> ops;
    std::vector<void*> allocs;
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int64_t* d_codes = nullptr;
    int* d_ovf = nullptr;          // split path: an activation left the fp16 range (the call is redone exactly)
    bool warned_ovf = false;
    float *buf[3] = {nullptr, nullptr, nullptr};
    _Float16 *plane[4] = {nullptr, nullptr, nullptr, nullptr};   // two {hi, lo} plane sets (split path): a conv's input and output
    size_t buf_elems = 0;
    float last_ms = 0.f;
    double flops_per_chunk = 0.0;
    std::vector<float> h_chunk;
    // batched chunk walk: assembled waveforms of a request (grown on demand) and the per-batch placement table
    float* d_wave = nullptr;
    int16_t* d_wave16 = nullptr;
    size_t wave_cap = 0, wave16_cap = 0;
    ChunkPlace* d_place = nullptr;   // [max_batch]
    float batch_ms = 0.f;            // GPU time of the last voc_synthesize_batch*
    int batch_chunks = 0;            // chunks it decoded
};

static float* voc_up(Voc* v, const PackTensor* t) {
    size_t ne = t->numel();
    std::vector<float> tmp;
    const float* src = (const float*)t->data;
    if (t->dtype == F16) {
        tmp.resize(ne);
        for (size_t i = 0; i < ne; i++) tmp[i] = h2f(((const uint16_t*)t->data)[i]);
        src = tmp.data();
    } else if (t->dtype != F32) {
        return nullptr;
    }
    float* d = nullptr;
    if (hipMalloc((void**)&d, ne * 4) != hipSuccess) return nullptr;
    v->allocs.push_back(d);
    if (hipMemcpy(d, src, ne * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

static float* voc_up_host(Voc* v, const std::vector<float>& h) {
    float* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * 4) != hipSuccess) return nullptr;
    v->allocs.push_back(d);
    if (hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

static void voc_destroy(Voc* v) {
    if (!v) return;
    if (v->s) hipStreamSynchronize(v->s);
    for (void* p : v->allocs) hipFree(p);
    for (float* b : v->buf)
        if (b) hipFree(b);
    for (_Float16* b : v->plane)
        if (b) hipFree(b);
    if (v->d_codes) hipFree(v->d_codes);
    if (v->d_wave) hipFree(v->d_wave);
    if (v->d_wave16) hipFree(v->d_wave16);
    if (v->d_place) hipFree(v->d_place);
    if (v->d_ovf) hipFree(v->d_ovf);
    if (v->e0) hipEventDestroy(v->e0);
    if (v->e1) hipEventDestroy(v->e1);
    if (v->s) hipStreamDestroy(v->s);
    delete v;
}

}  // namespace q3

using namespace q3;

// A handle's buffers, stream and events live on the device it was loaded on.  The HIP current device is a per-THREAD setting that
// starts at 0: a worker thread of a process that drives GPU k (one rank of a multi-GPU job, all GPUs visible) would otherwise launch
// the decode's kernels with the wrong device current.  Every entry point that touches the GPU binds its thread first.
static inline void voc_bind(const Voc* v) {
    int d = -1;
    if (v && (hipGetDevice(&d) != hipSuccess || d != v->device)) hipSetDevice(v->device);
}

extern "C" {

void voc_free(void* vv) {
    voc_bind((Voc*)vv);
    voc_destroy((Voc*)vv);
}

void* voc_load(const char* weights, int chunk_tokens, int max_batch) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        Q3_LOG("no HIP device available -- this library has no CPU path");
        return nullptr;
    }
    if (!weights) return nullptr;
    Pack p;
    if (!p.open(weights)) return nullptr;
    const PackTensor* prog = p.find("voc.program");
    if (!prog || prog->dtype != I32 || prog->ndim != 2 || prog->shape[1] != 8) {
        Q3_LOG("%s holds no vocoder program (tensor voc.program int32 [n][8])", weights);
        return nullptr;
    }
    // (any chunk length decodes; the chunk walk of voc_synthesize needs chunk > 32 and says so itself)
    if (const char* ex = getenv("Q3_VOC_EXACT")) g_voc_split = atoi(ex) ? 0 : 1;
    Voc* v = new Voc();
    hipGetDevice(&v->device);
    v->chunk = chunk_tokens > 0 ? chunk_tokens : 64;
    v->max_batch = max_batch > 0 ? max_batch : 1;
    bool ok = true;
    {
        // lowest queue priority: the vocoder is throughput work that runs beside the latency-bound frame
        // loop (highest priority, q3_engine.hip); Q3_STREAM_PRIO=0 creates both at the default priority
        int lo = 0, hi = 0;
        const bool prio = !(getenv("Q3_STREAM_PRIO") && atoi(getenv("Q3_STREAM_PRIO")) == 0);
        if (prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
            ok = hipStreamCreateWithPriority(&v->s, hipStreamNonBlocking, lo) == hipSuccess;
        else
            ok = hipStreamCreateWithFlags(&v->s, hipStreamNonBlocking) == hipSuccess;
    }
    ok = ok && hipEventCreate(&v->e0) == hipSuccess && hipEventCreate(&v->e1) == hipSuccess;
    const int32_t* pr = (const int32_t*)prog->data;
    const int n_ops = (int)prog->shape[0];
    int C = 0;
    long L = v->chunk;
    size_t max_elems = 0;
    double flops = 0.0;
    for (int i = 0; i < n_ops && ok; i++) {
        const int32_t* r = pr + i * 8;
        VocOp op;
        op.op = r[0];
        const std::string base = "voc.op" + std::to_string(i) + ".";
        auto need = [&](const char* n) -> const PackTensor* {
            const PackTensor* t = p.find(base + n);
            if (!t) {
                Q3_LOG("vocoder program op %d needs tensor %s%s", i, base.c_str(), n);
                ok = false;
            }
            return t;
        };
        if (op.op == VOP_EMBMEAN) {
            op.nq = r[1];
            op.cb = r[2];
            op.cout = r[3];  // embedding width
            const PackTensor* tb = need("embedding");
            if (!ok) break;
            if (op.nq < 1 || op.nq > 16 || tb->numel() != (uint64_t)op.nq * op.cb * op.cout) {
                Q3_LOG("vocoder op %d: embedding table size does not match the program (1..16 quantisers)", i);
                ok = false;
                break;
            }
            op.w = voc_up(v, tb);
            ok = op.w != nullptr;
            C = op.cout;
        } else if (op.op == VOP_RVQ) {
            op.nq = r[1];
            op.cb = r[2];
            op.cin = r[3];   // codebook dim
            op.cout = r[4];  // output channels
            const PackTensor *cb = need("codebook"), *ps = need("proj_sem"), *pa = need("proj_ac");
            if (!ok) break;
            if (op.nq < 1 || op.nq > 16) {
                Q3_LOG("vocoder op %d: %d quantisers (a request carries 16 ids per frame)", i, op.nq);
                ok = false;
                break;
            }
            if (cb->numel() != (uint64_t)op.nq * op.cb * op.cin || ps->numel() != (uint64_t)op.cout * op.cin ||
                pa->numel() != (uint64_t)op.cout * op.cin) {
                Q3_LOG("vocoder op %d: RVQ tensor sizes do not match the program", i);
                ok = false;
                break;
            }
            op.w = voc_up(v, cb);
            op.p_sem = voc_up(v, ps);
            op.p_ac = voc_up(v, pa);
            ok = op.w && op.p_sem && op.p_ac;
            C = op.cout;
            flops += 2.0 * 2 * op.cin * op.cout * L;
        } else if (op.op == VOP_CONV || op.op == VOP_CONVT) {
            op.cin = r[1];
            op.cout = r[2];
            op.k = r[3];
            op.p0 = r[4];  // dilation (conv) or stride (convT)
            op.flags = r[5];
            if (op.op == VOP_CONVT) {
                op.lt = r[6];
                op.rt = r[7];
                // every kept output must come from the polyphase GEMM over the input's own columns (+ k/s - 1 more)
                if (op.lt < 0 || op.rt < 0 || op.p0 <= 0 || op.lt + op.rt > op.k || convt_out(op, L) <= 0) {
                    Q3_LOG("vocoder op %d: transposed conv k=%d s=%d cannot be trimmed by %d + %d", i, op.k, op.p0, op.lt, op.rt);
                    ok = false;
                    break;
                }
            }
            if (op.cin != C) {
                Q3_LOG("vocoder op %d: expects %d input channels, previous op produced %d", i, op.cin, C);
                ok = false;
                break;
            }
            const PackTensor *wt = need("weight"), *bs = p.find(base + "bias");
            if (!ok) break;
            if (wt->numel() != (uint64_t)op.cin * op.cout * op.k || wt->dtype != F32) {
                Q3_LOG("vocoder op %d: weight size/dtype does not match the program", i);
                ok = false;
                break;
            }
            // repack to [tap][row][cin]
            const float* src = (const float*)wt->data;
            std::vector<float> wk;
            if (op.op == VOP_CONV) {  // torch Conv1d weight [cout][cin][k]
                wk.resize((size_t)op.k * op.cout * op.cin);
                for (int co = 0; co < op.cout; co++)
                    for (int ci = 0; ci < op.cin; ci++)
                        for (int k = 0; k < op.k; k++)
                            wk[((size_t)k * op.cout + co) * op.cin + ci] = src[((size_t)co * op.cin + ci) * op.k + k];
            } else {  // torch ConvTranspose1d weight [cin][cout][k], k = J*stride: polyphase rows m = co*s + p,
                      // tap j (input offset -j) reads w[ci][co][p + j*s]; conv tap index kk = J-1-j.  Row m of input
                      // column l is output sample l*s + p of the untrimmed result; op.lt / op.rt samples are cut
                      // at the ends (both k - s in the decoder family's CausalTransConvNet; 0 / k - s = strictly causal).
                const int s = op.p0;
                const int J = s > 0 ? op.k / s : 0;
                if (s <= 0 || J < 1 || op.k != J * s) {
                    Q3_LOG("vocoder op %d: transposed conv needs kernel = J*stride (got k=%d s=%d)", i, op.k, s);
                    ok = false;
                    break;
                }
                wk.resize((size_t)J * op.cout * s * op.cin);
                for (int j = 0; j < J; j++)
                    for (int co = 0; co < op.cout; co++)
                        for (int ph = 0; ph < s; ph++)
                            for (int ci = 0; ci < op.cin; ci++)
                                wk[((size_t)(J - 1 - j) * op.cout * s + (size_t)co * s + ph) * op.cin + ci] =
                                    src[((size_t)ci * op.cout + co) * op.k + ph + j * s];
            }
            if (op.cin % 16 == 0) {   // split-precision copy: [cin/16][tap][Mp128][16] hi / lo fp16
                const int KTAPS = op.op == VOP_CONV ? op.k : op.k / op.p0;
                const int Mrows = op.op == VOP_CONV ? op.cout : op.cout * op.p0;
                const int Mp = Mrows % 96 == 0 ? Mrows : (Mrows + 127) / 128 * 128;   // 96- or 64-row tiles, in bounds
                std::vector<uint16_t> hi((size_t)(op.cin / 16) * KTAPS * Mp * 16, 0), lo(hi.size(), 0);
                bool in_range = true;   // a weight beyond the fp16 range keeps this op on the exact path
                for (int k = 0; k < KTAPS; k++)
                    for (int m = 0; m < Mrows; m++)
                        for (int ci = 0; ci < op.cin; ci++) {
                            const float wv = wk[((size_t)k * Mrows + m) * op.cin + ci];
                            in_range = in_range && fabsf(wv) <= 65504.f;
                            const uint16_t h = f2h_sat(wv);
                            const size_t d = ((((size_t)(ci >> 4) * KTAPS + k) * Mp) + m) * 16 + (ci & 15);
                            hi[d] = h;
                            lo[d] = f2h_sat((wv - h2f(h)) * 2048.0f);
                        }
                void *dh = nullptr, *dl = nullptr;
                if (in_range) {
                    if (hipMalloc(&dh, hi.size() * 2) != hipSuccess || hipMalloc(&dl, lo.size() * 2) != hipSuccess) {
                        ok = false;
                        break;
                    }
                    v->allocs.push_back(dh);
                    v->allocs.push_back(dl);
                    ok = hipMemcpy(dh, hi.data(), hi.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
                         hipMemcpy(dl, lo.data(), lo.size() * 2, hipMemcpyHostToDevice) == hipSuccess;
                    op.w_hi = (_Float16*)dh;
                    op.w_lo = (_Float16*)dl;
                    op.Mp128 = Mp;
                    if (!ok) break;
                }
            }
            if (op.op == VOP_CONV && op.k == 1 && op.cin == op.cout && resunit_channels(op.cin) &&
                (op.flags & VF_RES_ADD) && (op.flags & VF_SNAKE)) {
                // resunit_kernel's order: w1p[row tile][t = (mt, q)][lane = (h, m)] = W[32 tile + m][32 mt + (q&3) + 8 (q>>2) + 4 h]
                const int Cc = op.cin;
                std::vector<float> w1((size_t)Cc * Cc);
                for (int t2 = 0; t2 < Cc / 32; t2++)
                    for (int t = 0; t < Cc / 2; t++)
                        for (int ln = 0; ln < 64; ln++) {
                            const int mt = t / 16, q = t % 16, h = ln >> 5, m = ln & 31;
                            const int c = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * h;
                            w1[((size_t)t2 * (Cc / 2) + t) * 64 + ln] = wk[(size_t)(32 * t2 + m) * Cc + c];
                        }
                op.w1p = voc_up_host(v, w1);
                if (!op.w1p) {
                    ok = false;
                    break;
                }
            }
            {   // [tap][row][cin] -> stage-major [cin/8][tap][cin%8][Mp]
                const int KTAPS = op.op == VOP_CONV ? op.k : op.k / op.p0;
                const int Mrows = op.op == VOP_CONV ? op.cout : op.cout * op.p0;
                const int Mp = (Mrows + 3) / 4 * 4;
                std::vector<float> wp((size_t)(op.cin / 8) * KTAPS * 8 * Mp, 0.f);
                for (int k = 0; k < KTAPS; k++)
                    for (int m = 0; m < Mrows; m++)
                        for (int ci = 0; ci < op.cin; ci++)
                            wp[((((size_t)(ci >> 3) * KTAPS + k) * 8 + (ci & 7)) * Mp) + m] =
                                wk[((size_t)k * Mrows + m) * op.cin + ci];
                wk.swap(wp);
            }
            op.w = voc_up_host(v, wk);
            op.bias = bs ? voc_up(v, bs) : nullptr;
            if (op.flags & VF_SNAKE) {
                const PackTensor *al = need("alpha"), *be = need("beta");
                if (!ok) break;
                // SnakeBeta with log-scale parameters: x + sin^2(exp(a) x) / (exp(b) + 1e-9)
                std::vector<float> ha(op.cin), hb(op.cin);
                for (int c = 0; c < op.cin; c++) {
                    const float av = al->dtype == F32 ? ((const float*)al->data)[c] : h2f(((const uint16_t*)al->data)[c]);
                    const float bv = be->dtype == F32 ? ((const float*)be->data)[c] : h2f(((const uint16_t*)be->data)[c]);
                    ha[c] = expf(av);
                    hb[c] = 1.0f / (expf(bv) + 1e-9f);
                }
                op.alpha = voc_up_host(v, ha);
                op.inv_beta = voc_up_host(v, hb);
            }
            if (op.cin % 8) {
                Q3_LOG("vocoder op %d: Cin=%d is not a multiple of 8", i, op.cin);
                ok = false;
                break;
            }
            ok = ok && op.w;
            flops += 2.0 * op.cin * op.cout * op.k * L;  // per input column; convT: k taps spread over s outputs
            C = op.cout;
            if (op.op == VOP_CONVT) L = convt_out(op, L);
        } else if (op.op == VOP_DWCONV || op.op == VOP_NORM || op.op == VOP_ATTN || op.op == VOP_GLU) {
            op.cin = r[1];
            op.cout = r[2];
            op.flags = r[5];
            if (op.cin != C) {
                Q3_LOG("vocoder op %d: expects %d input channels, previous op produced %d", i, op.cin, C);
                ok = false;
                break;
            }
            auto vec = [&](const char* n, uint64_t ne, bool required) -> float* {
                const PackTensor* t = p.find(base + n);
                if (!t) {
                    if (required) {
                        Q3_LOG("vocoder program op %d needs tensor %s%s", i, base.c_str(), n);
                        ok = false;
                    }
                    return nullptr;
                }
                if (t->numel() != ne) {
                    Q3_LOG("vocoder op %d: tensor %s has %llu elements, the program needs %llu", i, n,
                           (unsigned long long)t->numel(), (unsigned long long)ne);
                    ok = false;
                    return nullptr;
                }
                float* d = voc_up(v, t);
                if (!d) ok = false;
                return d;
            };
            if (op.op == VOP_DWCONV) {          // torch depthwise Conv1d weight [C][1][k]
                op.k = r[3];
                if (op.cout != op.cin || op.k < 1 || op.k > 64) ok = false;
                op.w = vec("weight", (uint64_t)op.cin * op.k, true);
                op.bias = vec("bias", (uint64_t)op.cin, false);
                flops += 2.0 * op.cin * op.k * L;
            } else if (op.op == VOP_NORM) {
                op.kind = r[3];
                op.eps = (float)((double)r[4] * 1e-9);
                if (op.cout != op.cin || (op.kind != 0 && op.kind != 1)) ok = false;
                op.w = vec("weight", (uint64_t)op.cin, true);
                op.bias = vec("bias", (uint64_t)op.cin, false);
            } else if (op.op == VOP_ATTN) {
                op.heads = r[3];
                op.head_dim = r[4];
                op.window = r[6];
                op.theta = (float)r[7];
                if (op.heads <= 0 || op.head_dim <= 0 || op.head_dim > 128 || (op.head_dim & 1) || op.window <= 0 ||
                    op.cin != 3 * op.heads * op.head_dim || op.cout != op.heads * op.head_dim)
                    ok = false;
                flops += 4.0 * op.heads * op.head_dim * (double)(op.window < L ? op.window : L) * L;
            } else {
                op.kind = r[3];   // 0 SiLU, 1 GELU
                if (op.cin != 2 * op.cout || (op.kind != 0 && op.kind != 1)) ok = false;
            }
            if (!ok) {
                Q3_LOG("vocoder op %d: malformed program row / tensors", i);
                break;
            }
            C = op.cout;
            if ((size_t)op.cin * pitch4(L) > max_elems) max_elems = (size_t)op.cin * pitch4(L);
        } else {
            Q3_LOG("vocoder program op %d: unknown opcode %d", i, op.op);
            ok = false;
            break;
        }
        if ((size_t)C * pitch4(L) > max_elems) max_elems = (size_t)C * pitch4(L);
        v->ops.push_back(op);
    }
    if (ok && C != 1) {
        Q3_LOG("vocoder program must end with 1 channel (got %d)", C);
        ok = false;
    }
    if (ok) {
        // nominal samples per frame = the product of the strides (decoder.total_upsample,
        // scripts/export_vocoder_traced.py:46: what the callers' SAMPLES_PER_TOKEN is); a decode returns chunk_samples
        long up = 1;
        for (const VocOp& o : v->ops)
            if (o.op == VOP_CONVT) up *= o.p0;
        v->upsample = (int)up;
        v->chunk_samples = L;
        if (L > (long)v->chunk * up) {
            Q3_LOG("vocoder program yields %ld samples for %d frames, more than %ld per frame", L, v->chunk, up);
            ok = false;
        }
    }
    if (ok) {
        v->flops_per_chunk = flops;
        v->buf_elems = max_elems * v->max_batch;
        for (int i = 0; i < 3 && ok; i++)     // (zeroed once: pad columns start finite)
            ok = hipMalloc((void**)&v->buf[i], v->buf_elems * 4) == hipSuccess && hipMemset(v->buf[i], 0, v->buf_elems * 4) == hipSuccess;
        for (int i = 0; i < 4 && ok; i++) ok = hipMalloc((void**)&v->plane[i], v->buf_elems * 2) == hipSuccess;
        ok = ok && hipMalloc((void**)&v->d_codes, sizeof(int64_t) * 16 * v->chunk * v->max_batch) == hipSuccess;
        ok = ok && hipMalloc((void**)&v->d_ovf, 16) == hipSuccess && hipMemset(v->d_ovf, 0, 16) == hipSuccess;
        ok = ok && hipMalloc((void**)&v->d_place, sizeof(ChunkPlace) * v->max_batch) == hipSuccess;
    }
    if (!ok) {
        Q3_LOG("voc_load failed");
        voc_destroy(v);
        return nullptr;
    }
    v->h_chunk.resize((size_t)v->chunk_samples);
    return v;
}

// Cap the number of workgroups every vocoder launch may occupy (0 = no cap).  Process-wide.
int voc_set_exact_fp32(int on) {
    g_voc_split = on ? 0 : 1;
    return 0;
}

int voc_set_narrow_k1(int on) {   // test hook: 128-column tiles for the 1-tap convs (default on)
    g_voc_narrow_k1 = on ? 1 : 0;
    return 0;
}

int voc_set_fused_units(int on) {   // 1 (default): residual units at 96 / 192 channels run as one launch (exact path)
    g_voc_fuse = on ? 1 : 0;
    return 0;
}

int voc_set_max_workgroups(int n) {   // -> the cap in effect (0 = none)
    if (n < 0) {   // one persistent workgroup per compute unit: the co-run setting (qwen3tts_voc.h)
        int dev = 0;
        hipDeviceProp_t p;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 0;
    }
    g_voc_max_wgs = n;
    return n;
}

int voc_chunk_tokens(void* vv) { return vv ? ((Voc*)vv)->chunk : 0; }
int voc_samples_per_token(void* vv) { return vv ? ((Voc*)vv)->upsample : 0; }
int voc_chunk_samples(void* vv) { return vv ? (int)((Voc*)vv)->chunk_samples : 0; }
float voc_last_decode_ms(void* vv) { return vv ? ((Voc*)vv)->last_ms : -1.f; }
double voc_decode_flops(void* vv, int B) { return vv ? ((Voc*)vv)->flops_per_chunk * B : 0.0; }

// Split path of one conv op.  State carried between ops: which f32 buffer holds the newest f32 activation
// (and whether it is the current one), and which plane set (if any) already holds the current activation in
// the next conv's input form.
struct SplitState {
    int f32_idx = 0;
    bool f32_cur = true;
    int planes = -1;       // plane set holding the current activation (Snake of the consuming op applied), or -1
    float* res = nullptr;  // residual-unit input (f32)
};

static bool split_capable(const VocOp& op) { return (op.op == VOP_CONV || op.op == VOP_CONVT) && op.w_hi != nullptr; }

static int voc_conv_split(Voc* v, const VocOp& op, const VocOp* next, bool last, int B, int C, long L, SplitState& st) {
    const int KT = op.op == VOP_CONV ? op.k : op.k / op.p0;
    int in_set = st.planes;
    if (in_set < 0) {   // materialise the input planes from the f32 activation (this op's Snake applied)
        if (!st.f32_cur) return -1;
        in_set = 0;
        hipLaunchKernelGGL(snake_split_kernel, dim3((unsigned)((L + 255) / 256), op.cin / 8, B), dim3(256), 0, v->s,
                           v->buf[st.f32_idx], op.alpha, op.inv_beta, v->plane[0], v->plane[1], op.cin, (int)L, (int)pitch4(L),
                           (op.flags & VF_GELU) ? 1 : 0, v->d_ovf);
        Q3_HIP(hipGetLastError(), -1);
    }
    if (op.flags & VF_RES_SAVE) {
        if (!st.f32_cur) return -1;   // the producer keeps an f32 copy whenever its consumer saves a residual
        st.res = v->buf[st.f32_idx];
    }
    // (a GELU consumer takes its planes from the separate pass: erf in this epilogue costs every conv registers)
    const bool want_planes = next && split_capable(*next) && op.op == VOP_CONV && op.cout % 16 == 0 && !(next->flags & VF_GELU);
    const bool want_f32 = !want_planes || last || (next->flags & VF_RES_SAVE);
    SplitArgs sa;
    sa.xh = v->plane[2 * in_set];
    sa.xl = v->plane[2 * in_set + 1];
    sa.w_hi = op.w_hi;
    sa.w_lo = op.w_lo;
    sa.bias = op.bias;
    sa.ovf = v->d_ovf;
    sa.res = (op.flags & VF_RES_ADD) ? st.res : nullptr;
    sa.Cin = op.cin;
    sa.Cout = op.cout;
    sa.Mp = op.Mp128;
    sa.Lin = (int)L;
    sa.clamp = (op.flags & VF_CLAMP) ? 1 : 0;
    if (op.op == VOP_CONV) {
        sa.dil = op.p0;
        sa.stride = 1;
        sa.M = op.cout;
        sa.Lout = sa.Lc = (int)L;
    } else {
        sa.dil = 1;
        sa.stride = op.p0;
        sa.M = op.cout * op.p0;
        sa.lt = op.lt;
        sa.Lout = (int)convt_out(op, L);
        sa.Lc = (int)convt_cols(op, L);
    }
    sa.ldy = (int)pitch4(sa.Lout);
    int out_f32 = st.f32_idx;
    if (want_f32) {
        // never the buffer the residual (or a still-current f32 input) lives in
        out_f32 = st.f32_idx ^ 1;
        if (sa.res == v->buf[out_f32]) return -1;
        sa.y = v->buf[out_f32];
    }
    if (want_planes) {
        sa.oh = v->plane[2 * (in_set ^ 1)];
        sa.ol = v->plane[2 * (in_set ^ 1) + 1];
        if (next->flags & VF_SNAKE) {
            sa.oalpha = next->alpha;
            sa.oinv_beta = next->inv_beta;
        }
    }
    if (launch_conv_split(v->s, sa, KT, B)) return -1;
    if (want_f32) {
        st.f32_idx = out_f32;
        st.f32_cur = true;
    } else {
        st.f32_cur = false;
    }
    st.planes = want_planes ? (in_set ^ 1) : -1;
    return 0;
}

// T: frames per chunk of THIS decode (0 = the model's chunk length).  Every op is causal per column apart from the
// transposed convs' look-ahead of one input column (< 1 frame in total), so the first n frames' samples of a decode of
// T > n frames are the same bits whatever T is: the chunk walk decodes a short tail chunk at its own length + 1 pad frame
// instead of the reference's zero-padded 64 (d_codes then holds [B][T][16]).
static int voc_run(Voc* v, int B, float** out_dev, int n_ops = -1, int* outC = nullptr, long* outL = nullptr,
                   float* op_ms = nullptr, bool force_exact = false, int T = 0) {
    // ping-pong between buf[0]/buf[1]; buf[2] keeps the residual-unit input (exact path)
    int cur = 0;
    int C = 0;
    if (T <= 0 || T > v->chunk) T = v->chunk;
    long L = T;
    long Lf = v->chunk;   // the same op's length in a full-length decode (variant rules look at it)
    float* res = nullptr;
    SplitState st;
    const size_t nrun = n_ops < 0 ? v->ops.size() : (size_t)n_ops < v->ops.size() ? (size_t)n_ops : v->ops.size();
    for (size_t i = 0; i < nrun; i++) {
        const VocOp& op = v->ops[i];
        if (op_ms) hipEventRecord(v->e0, v->s);
        float* in = v->buf[cur];
        float* out = v->buf[cur ^ 1];
        const int ld = (int)pitch4(L);
        if (op.op == VOP_RVQ || op.op == VOP_EMBMEAN) {
            if (op.op == VOP_RVQ)
                hipLaunchKernelGGL(rvq_kernel, dim3(T, B), dim3(256), 2 * op.cin * sizeof(float), v->s, v->d_codes, op.w,
                                   op.p_sem, op.p_ac, out, T, ld, op.nq, op.cb, op.cin, op.cout);
            else
                hipLaunchKernelGGL(embmean_kernel, dim3(T, B), dim3(256), 0, v->s, v->d_codes, op.w, out, T, ld,
                                   op.nq, 16, op.cb, op.cout);
            Q3_HIP(hipGetLastError(), -1);
            C = op.cout;
            cur ^= 1;
            st.f32_idx = cur;
            st.f32_cur = true;
            st.planes = -1;
        } else if (op.op == VOP_DWCONV || op.op == VOP_NORM || op.op == VOP_ATTN || op.op == VOP_GLU) {
            if (!st.f32_cur) return -1;   // these ops read the f32 activation (their producer wrote one: they are no conv)
            cur = st.f32_idx;
            in = v->buf[cur];
            out = v->buf[cur ^ 1];
            if (op.flags & VF_RES_SAVE) {
                // the unit's input is needed again two ops later: it becomes buf[2] (a pointer swap, no copy), where
                // the ping-pong of the ops in between does not write
                std::swap(v->buf[2], v->buf[cur]);
                in = res = st.res = v->buf[2];
            }
            const unsigned lb = (unsigned)((L + 255) / 256);
            if (op.op == VOP_DWCONV)
                hipLaunchKernelGGL(dwconv_kernel, dim3(lb, op.cin, B), dim3(256), 0, v->s, in, op.w, op.bias, out, op.cin, (int)L, ld, op.k);
            else if (op.op == VOP_NORM)
                hipLaunchKernelGGL(chan_norm_kernel, dim3((unsigned)((L + 63) / 64), B), dim3(1024), 0, v->s, in, op.w, op.bias, out, op.cin, (int)L,
                                   ld, op.kind, op.eps);
            else if (op.op == VOP_ATTN) {
                const size_t tile_lds = (size_t)3 * L * (op.head_dim + 1) * sizeof(float);
                if (op.head_dim <= 64 && op.head_dim % 2 == 0 && tile_lds <= 64 * 1024) {
                    static bool attr = false;
                    if (!attr) {
                        Q3_HIP(hipFuncSetAttribute((const void*)voc_attn_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   64 * 1024), -1);
                        attr = true;
                    }
                    hipLaunchKernelGGL(voc_attn_tile_kernel, dim3(op.heads, B), dim3(256), tile_lds, v->s, in, out, op.heads,
                                       op.head_dim, (int)L, ld, op.window, op.theta);
                } else {
                    hipLaunchKernelGGL(voc_attn_kernel, dim3((unsigned)L, op.heads, B), dim3(64), 0, v->s, in, out, op.heads,
                                       op.head_dim, (int)L, ld, op.window, op.theta);
                }
            }
            else
                hipLaunchKernelGGL(glu_kernel, dim3(lb, op.cout, B), dim3(256), 0, v->s, in, out, op.cout, (int)L, ld, op.kind);
            Q3_HIP(hipGetLastError(), -1);
            C = op.cout;
            cur ^= 1;
            st.f32_idx = cur;
            st.f32_cur = true;
            st.planes = -1;
        } else if (g_voc_split && !force_exact && op.w_hi) {
            const bool last = i + 1 == nrun;
            const VocOp* next = (i + 1 < v->ops.size()) ? &v->ops[i + 1] : nullptr;
            if (voc_conv_split(v, op, last ? nullptr : next, last, B, C, L, st)) {
                Q3_LOG("vocoder op %zu: split path could not be scheduled", i);
                return -1;
            }
            cur = st.f32_idx;
            C = op.cout;
            if (op.op == VOP_CONVT) L = convt_out(op, L), Lf = convt_out(op, Lf);
        } else if (g_voc_fuse && op.op == VOP_CONV && op.k == 7 && (op.flags & VF_RES_SAVE) && (op.flags & VF_SNAKE) &&
                   op.cin == op.cout && resunit_channels(op.cin) && i + 1 < nrun && v->ops[i + 1].w1p && st.f32_cur) {
            // a whole residual unit (this 7-tap conv + the 1x1 conv that closes it) in one launch
            const VocOp& op1 = v->ops[i + 1];
            cur = st.f32_idx;
            ResUnitArgs ra;
            ra.x = v->buf[cur];
            ra.y = v->buf[cur ^ 1];
            ra.w7 = op.w;
            ra.w1p = op1.w1p;
            ra.b7 = op.bias;
            ra.b1 = op1.bias;
            ra.al7 = op.alpha;
            ra.ib7 = op.inv_beta;
            ra.al1 = op1.alpha;
            ra.ib1 = op1.inv_beta;
            ra.Lin = (int)L;
            ra.ld = ld;
            ra.dil = op.p0;
            if (launch_resunit(v->s, ra, op.cin, B)) return -1;
            cur ^= 1;
            st.f32_idx = cur;
            st.f32_cur = true;
            st.planes = -1;
            if (op_ms) {
                hipEventRecord(v->e1, v->s);
                hipStreamSynchronize(v->s);
                hipEventElapsedTime(&op_ms[i], v->e0, v->e1);
                op_ms[i + 1] = 0.f;
            }
            i++;   // the 1x1 conv is done
            continue;
        } else {
            if (!st.f32_cur) return -1;
            cur = st.f32_idx;
            in = v->buf[cur];
            out = v->buf[cur ^ 1];
            ConvArgs a;
            a.x = in;
            a.y = out;
            a.wk = op.w;
            a.bias = op.bias;
            a.alpha = op.alpha;
            a.inv_beta = op.inv_beta;
            a.Cin = op.cin;
            a.Cout = op.cout;
            a.Lin = (int)L;
            a.clamp = (op.flags & VF_CLAMP) ? 1 : 0;
            a.gelu = (op.flags & VF_GELU) ? 1 : 0;
            a.ldx = ld;
            if (op.op == VOP_CONV) {
                a.K = op.k;
                a.dil = op.p0;
                a.stride = 1;
                a.M = op.cout;
                a.Lout = a.Lc = (int)L;
                a.Lrule = (int)Lf;
            } else {
                a.K = op.k / op.p0;
                a.dil = 1;
                a.stride = op.p0;
                a.M = op.cout * op.p0;
                a.lt = op.lt;
                a.Lout = (int)convt_out(op, L);
                a.Lc = (int)convt_cols(op, L);
                a.Lrule = (int)convt_cols(op, Lf);
            }
            a.ldy = (int)pitch4(a.Lout);
            if (op.flags & VF_RES_SAVE) {
                // the unit's input is needed again after two convs: it becomes buf[2] (pointer swap), out of the ping-pong
                std::swap(v->buf[2], v->buf[cur]);
                res = st.res = v->buf[2];
                a.x = res;
            }
            if (op.flags & VF_RES_ADD) a.res = res ? res : st.res;
            if (launch_conv(v->s, a, B)) return -1;
            C = op.cout;
            if (op.op == VOP_CONVT) L = convt_out(op, L), Lf = convt_out(op, Lf);
            cur ^= 1;
            st.f32_idx = cur;
            st.f32_cur = true;
            st.planes = -1;
        }
        if (op_ms) {
            hipEventRecord(v->e1, v->s);
            hipStreamSynchronize(v->s);
            hipEventElapsedTime(&op_ms[i], v->e0, v->e1);
        }
    }
    if (!st.f32_cur) return -1;
    *out_dev = v->buf[st.f32_idx];
    if (outC) *outC = C;
    if (outL) *outL = L;
    return 0;
}

int voc_decode(void* vv, const int64_t* codes, int B, float* out) {
    Voc* v = (Voc*)vv;
    voc_bind(v);
    if (!v || !codes || !out || B <= 0 || B > v->max_batch) return -1;
    Q3_HIP(hipMemcpyAsync(v->d_codes, codes, sizeof(int64_t) * 16 * (size_t)v->chunk * B, hipMemcpyHostToDevice, v->s), -1);
    Q3_HIP(hipEventRecord(v->e0, v->s), -1);
    float* res = nullptr;
    if (voc_run(v, B, &res)) return -1;
    Q3_HIP(hipEventRecord(v->e1, v->s), -1);
    int ovf = 0;
    if (g_voc_split) Q3_HIP(hipMemcpyAsync(&ovf, v->d_ovf, sizeof(int), hipMemcpyDeviceToHost, v->s), -1);
    // rows of chunk_samples floats at the device pitch -> dense out[B][chunk_samples]
    const size_t row = sizeof(float) * (size_t)v->chunk_samples, dpitch = sizeof(float) * (size_t)pitch4(v->chunk_samples);
    Q3_HIP(hipMemcpy2DAsync(out, row, res, dpitch, row, (size_t)B, hipMemcpyDeviceToHost, v->s), -1);
    Q3_HIP(hipStreamSynchronize(v->s), -1);
    if (ovf) {
        // an activation beyond +-65504 (or a NaN): two fp16 terms cannot carry it -- this call is redone on the
        // exact-fp32 MFMA path, so the split arithmetic never degrades a result silently
        if (!v->warned_ovf) Q3_LOG("vocoder: activation outside the fp16 range, decoding this call with the exact-fp32 path");
        v->warned_ovf = true;
        Q3_HIP(hipMemsetAsync(v->d_ovf, 0, sizeof(int), v->s), -1);
        Q3_HIP(hipEventRecord(v->e0, v->s), -1);
        if (voc_run(v, B, &res, -1, nullptr, nullptr, nullptr, true)) return -1;
        Q3_HIP(hipEventRecord(v->e1, v->s), -1);
        Q3_HIP(hipMemcpy2DAsync(out, row, res, dpitch, row, (size_t)B, hipMemcpyDeviceToHost, v->s), -1);
        Q3_HIP(hipStreamSynchronize(v->s), -1);
    }
    hipEventElapsedTime(&v->last_ms, v->e0, v->e1);
    return 0;
}

// test hook: per-op GPU milliseconds of one decode of B chunks (codes already uploaded by a previous voc_decode)
int voc_debug_profile(void* vv, int B, float* op_ms, int max_ops) {
    Voc* v = (Voc*)vv;
    voc_bind(v);
    if (!v || B <= 0 || B > v->max_batch || (int)v->ops.size() > max_ops) return -1;
    float* res = nullptr;
    if (voc_run(v, B, &res, -1, nullptr, nullptr, op_ms)) return -1;
    return (int)v->ops.size();
}

// test hook: run only the first n_ops ops, return the activation [B][C][L]
int voc_debug_run(void* vv, const int64_t* codes, int B, int n_ops, float* out, int* C, int* L) {
    Voc* v = (Voc*)vv;
    voc_bind(v);
    if (!v || B <= 0 || B > v->max_batch) return -1;
    Q3_HIP(hipMemcpyAsync(v->d_codes, codes, sizeof(int64_t) * 16 * (size_t)v->chunk * B, hipMemcpyHostToDevice, v->s), -1);
    float* res = nullptr;
    long LL = 0;
    if (voc_run(v, B, &res, n_ops, C, &LL)) return -1;
    *L = (int)LL;
    Q3_HIP(hipMemcpy2DAsync(out, sizeof(float) * (size_t)LL, res, sizeof(float) * (size_t)pitch4(LL), sizeof(float) * (size_t)LL,
                            (size_t)B * (*C), hipMemcpyDeviceToHost, v->s), -1);
    Q3_HIP(hipStreamSynchronize(v->s), -1);
    return 0;
}

// Frames a chunk of `len` real frames is decoded at: its own length plus the one pad frame the transposed convs' look-ahead
// reaches into (voc_run's note), rounded up to 8 so that a request's tail chunks fall into few groups; the full chunk for a
// full chunk, under the split-f16 arithmetic (its overflow redo is per call) and with Q3_VOC_FULL_CHUNKS=1 (A/B knob).
static int voc_decode_frames(const Voc* v, int len) {
    static const int full = getenv("Q3_VOC_FULL_CHUNKS") ? atoi(getenv("Q3_VOC_FULL_CHUNKS")) : 0;
    static const int rnd = getenv("Q3_VOC_FRAME_ROUND") ? atoi(getenv("Q3_VOC_FRAME_ROUND")) : 8;
    if (full || g_voc_split || len >= v->chunk) return v->chunk;
    const int r = rnd > 0 ? rnd : 1;
    const int t = (len + 1 + r - 1) / r * r;
    return t < v->chunk ? t : v->chunk;
}

int voc_synthesize_max_samples(void* vv, int n) {
    Voc* v = (Voc*)vv;
    if (!v || n <= 0) return 0;
    return (n + v->chunk) * v->upsample;  // the reference's redundant tail chunk adds < chunk frames
}

// VocoderServer.synthesize (vocoder_server.py:73-121), bug-compatible chunk walk, float output.
int voc_synthesize_f32(void* vv, const int64_t* codes, int n, float* out, int32_t* n_samples) {
    Voc* v = (Voc*)vv;
    voc_bind(v);
    if (!v || !codes || !out || !n_samples || n <= 0) return -1;
    const int CH = v->chunk, SPT = v->upsample;
    // numpy slicing, as the reference writes it: `audio[:len * SAMPLES_PER_TOKEN]` of what the model returned --
    // a decode yields chunk_samples <= CH * SPT samples (the decoder family's transposed convs trim), so a slice is
    // min(len * SPT, chunk_samples) long (vocoder_server.py:81,98-99)
    const size_t CS = (size_t)v->chunk_samples;
    auto sliced = [&](int len) -> size_t { return (size_t)len * SPT < CS ? (size_t)len * SPT : CS; };
    if (n > CH && CH <= 32) {
        // the chunk walk steps by chunk - 16 and its output bound (n + chunk) frames needs chunk > 32; the
        // reference's models are traced at 64 or 256 (scripts/export_vocoder_traced.py)
        Q3_LOG("voc_synthesize: chunk_tokens=%d is too short for the 16-frame overlap walk (need > 32)", CH);
        return -1;
    }
    std::vector<int64_t> padded((size_t)CH * 16);
    std::vector<float>& chunk = v->h_chunk;
    auto run_chunk = [&](int start, int len) -> int {
        std::fill(padded.begin(), padded.end(), 0);
        memcpy(padded.data(), codes + (size_t)start * 16, sizeof(int64_t) * 16 * len);
        const int T = voc_decode_frames(v, len);
        if (T == CH) return voc_decode(v, padded.data(), 1, chunk.data());
        // a short chunk: decoded at T frames (same bits for the samples the walk keeps), only those samples come back
        Q3_HIP(hipMemcpyAsync(v->d_codes, padded.data(), sizeof(int64_t) * 16 * (size_t)T, hipMemcpyHostToDevice, v->s), -1);
        Q3_HIP(hipEventRecord(v->e0, v->s), -1);
        float* res = nullptr;
        long LL = 0;
        if (voc_run(v, 1, &res, -1, nullptr, &LL, nullptr, false, T)) return -1;
        Q3_HIP(hipEventRecord(v->e1, v->s), -1);
        if ((size_t)LL < sliced(len)) {
            Q3_LOG("voc_synthesize: a decode of %d frames yields %ld samples, fewer than the %zu kept", T, LL, sliced(len));
            return -1;
        }
        Q3_HIP(hipMemcpyAsync(chunk.data(), res, sizeof(float) * sliced(len), hipMemcpyDeviceToHost, v->s), -1);
        Q3_HIP(hipStreamSynchronize(v->s), -1);
        hipEventElapsedTime(&v->last_ms, v->e0, v->e1);
        return 0;
    };
    if (n <= CH) {
        if (run_chunk(0, n)) return -1;
        memcpy(out, chunk.data(), sizeof(float) * sliced(n));
        *n_samples = (int32_t)sliced(n);
        return 0;
    }
    const int OVERLAP = 16, OV = OVERLAP * SPT, step = CH - OVERLAP;
    const size_t capacity = (size_t)voc_synthesize_max_samples(v, n);   // what callers size `out` with
    size_t have = 0;
    for (int start = 0; start < n; start += step) {
        const int len = (start + CH <= n) ? CH : n - start;
        if (run_chunk(start, len)) return -1;
        const size_t cl = sliced(len);
        if (have + cl > capacity) {
            Q3_LOG("voc_synthesize: chunk walk would pass the output bound (%zu + %zu > %zu)", have, cl, capacity);
            return -1;
        }
        if (start == 0) {
            memcpy(out, chunk.data(), sizeof(float) * cl);
            have = cl;
        } else if (have >= (size_t)OV && cl >= (size_t)OV) {
            // np.linspace(1, 0, OV, dtype=float32) fade-out, fade-in = 1 - fade-out, blended in float32
            float* tail = out + have - OV;
            for (int i = 0; i < OV; i++) {
                const double stepv = -1.0 / (double)(OV - 1);
                const float fo = (i == OV - 1) ? 0.0f : (float)(1.0 + (double)i * stepv);
                const float fi = 1.0f - fo;
                tail[i] = tail[i] * fo + chunk[i] * fi;
            }
            memcpy(out + have, chunk.data() + OV, sizeof(float) * (cl - OV));
            have += cl - OV;
        } else {
            memcpy(out + have, chunk.data(), sizeof(float) * cl);
            have += cl;
        }
    }
    *n_samples = (int32_t)have;
    return 0;
}

// ---- batched VocoderServer.synthesize ----
namespace {
struct WalkChunk { int utt, start, len; size_t cl; int head; long long dst; };

// the reference's walk for one utterance of n frames whose output starts at sample `base` -> its chunks, returns its length
size_t plan_walk(const Voc* v, int u, int n, long long base, std::vector<WalkChunk>& out) {
    const int CH = v->chunk, SPT = v->upsample;
    const size_t CS = (size_t)v->chunk_samples, OV = (size_t)16 * SPT;
    auto sliced = [&](int len) -> size_t { return (size_t)len * SPT < CS ? (size_t)len * SPT : CS; };
    if (n <= CH) {
        out.push_back({u, 0, n, sliced(n), 0, base});
        return sliced(n);
    }
    size_t have = 0;
    for (int start = 0; start < n; start += CH - 16) {
        const int len = (start + CH <= n) ? CH : n - start;
        const size_t cl = sliced(len);
        if (start == 0) {
            out.push_back({u, start, len, cl, 0, base});
            have = cl;
        } else if (have >= OV && cl >= OV) {
            out.push_back({u, start, len, cl, (int)OV, base + (long long)(have - OV)});
            have += cl - OV;
        } else {
            out.push_back({u, start, len, cl, 0, base + (long long)have});
            have += cl;
        }
    }
    return have;
}

int synth_batch(Voc* v, const int64_t* codes, const int32_t* n_tokens, int U, int64_t* offsets, bool want16, void* out, int64_t cap) {
    voc_bind(v);
    if (!v || !codes || !n_tokens || !offsets || !out || U <= 0) return -1;
    const int CH = v->chunk;
    std::vector<WalkChunk> walk;
    std::vector<size_t> code_off(U);
    long long total = 0;
    size_t coff = 0;
    for (int u = 0; u < U; u++) {
        if (n_tokens[u] <= 0) {
            Q3_LOG("voc_synthesize_batch: utterance %d has %d frames", u, n_tokens[u]);
            return -1;
        }
        if (n_tokens[u] > CH && CH <= 32) {
            Q3_LOG("voc_synthesize_batch: chunk_tokens=%d is too short for the 16-frame overlap walk (need > 32)", CH);
            return -1;
        }
        offsets[u] = total;
        code_off[u] = coff;
        coff += (size_t)n_tokens[u] * 16;
        total += (long long)plan_walk(v, u, n_tokens[u], total, walk);
    }
    offsets[U] = total;
    if (total > cap) {
        Q3_LOG("voc_synthesize_batch: %lld samples do not fit the caller's buffer of %lld", total, (long long)cap);
        return -1;
    }
    if ((size_t)total > v->wave_cap) {
        if (v->d_wave) hipFree(v->d_wave);
        v->d_wave = nullptr;
        v->wave_cap = 0;
        Q3_HIP(hipMalloc((void**)&v->d_wave, sizeof(float) * (size_t)total), -1);
        v->wave_cap = (size_t)total;
    }
    const int OV = 16 * v->upsample;
    // chunks of one decode length run together: full chunks first, then the tail chunks by length (voc_decode_frames)
    std::vector<int> order(walk.size()), frames(walk.size());
    for (size_t i = 0; i < walk.size(); i++) order[i] = (int)i, frames[i] = voc_decode_frames(v, walk[i].len);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return frames[x] > frames[y]; });
    std::vector<int64_t> padded((size_t)v->max_batch * CH * 16);
    std::vector<ChunkPlace> place(v->max_batch);
    bool redo_exact = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        Q3_HIP(hipEventRecord(v->e0, v->s), -1);
        for (size_t c0 = 0; c0 < walk.size();) {
            const int T = frames[order[c0]];
            int B = 0;
            while (c0 + B < walk.size() && B < v->max_batch && frames[order[c0 + B]] == T) B++;
            std::fill(padded.begin(), padded.begin() + (size_t)B * T * 16, 0);
            for (int b = 0; b < B; b++) {
                const WalkChunk& w = walk[order[c0 + b]];
                memcpy(padded.data() + (size_t)b * T * 16, codes + code_off[w.utt] + (size_t)w.start * 16, sizeof(int64_t) * 16 * w.len);
                place[b] = {b, (int)w.cl, w.head, w.dst};
            }
            Q3_HIP(hipMemcpyAsync(v->d_codes, padded.data(), sizeof(int64_t) * 16 * (size_t)T * B, hipMemcpyHostToDevice, v->s), -1);
            Q3_HIP(hipMemcpyAsync(v->d_place, place.data(), sizeof(ChunkPlace) * B, hipMemcpyHostToDevice, v->s), -1);
            float* res = nullptr;
            long LL = 0;
            if (voc_run(v, B, &res, -1, nullptr, &LL, nullptr, redo_exact, T)) return -1;
            for (int b = 0; b < B; b++)
                if ((long)place[b].len > LL) {
                    Q3_LOG("voc_synthesize_batch: a decode of %d frames yields %ld samples, fewer than the %d kept", T, LL, place[b].len);
                    return -1;
                }
            const int pitch = (int)pitch4(LL);
            hipLaunchKernelGGL(voc_place_copy_kernel, dim3(64, B), dim3(256), 0, v->s, res, pitch, v->d_place, v->d_wave);
            hipLaunchKernelGGL(voc_place_blend_kernel, dim3(32, B), dim3(256), 0, v->s, res, pitch, v->d_place, v->d_wave, OV);
            Q3_HIP(hipGetLastError(), -1);
            Q3_HIP(hipStreamSynchronize(v->s), -1);   // the staging vectors are reused by the next batch
            c0 += B;
        }
        Q3_HIP(hipEventRecord(v->e1, v->s), -1);
        int ovf = 0;
        if (g_voc_split && !redo_exact) Q3_HIP(hipMemcpyAsync(&ovf, v->d_ovf, sizeof(int), hipMemcpyDeviceToHost, v->s), -1);
        Q3_HIP(hipStreamSynchronize(v->s), -1);
        if (!ovf) break;
        // an activation beyond the fp16 range: the whole request is redone on the exact-fp32 path (voc_decode's rule)
        if (!v->warned_ovf) Q3_LOG("vocoder: activation outside the fp16 range, decoding this request with the exact-fp32 path");
        v->warned_ovf = true;
        Q3_HIP(hipMemsetAsync(v->d_ovf, 0, sizeof(int), v->s), -1);
        redo_exact = true;
    }
    hipEventElapsedTime(&v->batch_ms, v->e0, v->e1);
    v->batch_chunks = (int)walk.size();
    if (!want16) {
        Q3_HIP(hipMemcpyAsync(out, v->d_wave, sizeof(float) * (size_t)total, hipMemcpyDeviceToHost, v->s), -1);
    } else {
        if ((size_t)total > v->wave16_cap) {
            if (v->d_wave16) hipFree(v->d_wave16);
            v->d_wave16 = nullptr;
            v->wave16_cap = 0;
            Q3_HIP(hipMalloc((void**)&v->d_wave16, sizeof(int16_t) * (size_t)total), -1);
            v->wave16_cap = (size_t)total;
        }
        hipLaunchKernelGGL(voc_to_int16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, v->s, v->d_wave, v->d_wave16, total);
        Q3_HIP(hipGetLastError(), -1);
        Q3_HIP(hipMemcpyAsync(out, v->d_wave16, sizeof(int16_t) * (size_t)total, hipMemcpyDeviceToHost, v->s), -1);
    }
    Q3_HIP(hipStreamSynchronize(v->s), -1);
    return 0;
}
}  // namespace

int64_t voc_synthesize_batch_max_samples(void* vv, const int32_t* n_tokens, int U) {
    Voc* v = (Voc*)vv;
    if (!v || !n_tokens || U <= 0) return 0;
    int64_t t = 0;
    for (int u = 0; u < U; u++) t += n_tokens[u] > 0 ? (int64_t)voc_synthesize_max_samples(v, n_tokens[u]) : 0;
    return t;
}

int voc_synthesize_batch_f32(void* vv, const int64_t* codes, const int32_t* n_tokens, int U, float* out, int64_t out_capacity,
                             int64_t* offsets) {
    return synth_batch((Voc*)vv, codes, n_tokens, U, offsets, false, out, out_capacity);
}

int voc_synthesize_batch(void* vv, const int64_t* codes, const int32_t* n_tokens, int U, int16_t* out, int64_t out_capacity,
                         int64_t* offsets) {
    return synth_batch((Voc*)vv, codes, n_tokens, U, offsets, true, out, out_capacity);
}

float voc_last_batch_ms(void* vv) { return vv ? ((Voc*)vv)->batch_ms : -1.f; }
int voc_last_batch_chunks(void* vv) { return vv ? ((Voc*)vv)->batch_chunks : 0; }

int voc_synthesize(void* vv, const int64_t* codes, int n, int16_t* out, int32_t* n_samples) {
    Voc* v = (Voc*)vv;
    if (!v || !out) return -1;
    std::vector<float> f((size_t)voc_synthesize_max_samples(v, n));
    if (voc_synthesize_f32(v, codes, n, f.data(), n_samples)) return -1;
    for (int32_t i = 0; i < *n_samples; i++) {
        // np.clip(audio * 32767, -32768, 32767).astype(np.int16): float32 product, truncation toward zero
        float x = f[i] * 32767.0f;
        x = x < -32768.0f ? -32768.0f : (x > 32767.0f ? 32767.0f : x);
        out[i] = (int16_t)x;
    }
    return 0;
}

}  // extern "C"
