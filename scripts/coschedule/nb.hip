// Synthetic neighbour kernels for scripts/coschedule_probe.py and scripts/coschedule/*.py (diagnostics; not part of
// the product library).  nb_kernel<NA>: per iteration `nld` 16-byte global loads (+ stores), `nlds` LDS write/read pairs,
// `nmfma` x NA dependent fp32 MFMAs (32x32x2, 64 cycles each) -- the MFMA duty cycle, the grid (all resident or
// refilling), the LDS request and the register count (NA accumulator tiles) of a neighbour are the knobs; nb_set_gap puts
// a pause (s_nop / s_sleep / one VALU op / one LDS read) after every n-th MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC nb.hip -o libnb.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ int g_gap_every = 0, g_gap_kind = 0;
template <int NA>
__global__ void __launch_bounds__(256) nb_kernel(int iters, int nmfma, int nlds, int nld, const float4* __restrict__ buf,
                                                 unsigned nbuf4, int do_store, float4* __restrict__ wbuf, float* sink) {
    extern __shared__ float lds[];
    f16v acc[NA];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[a][i] = 0.f;
    float av = threadIdx.x * 1e-3f, bv = blockIdx.x * 1e-4f;
    float4 g = make_float4(0, 0, 0, 0);
    unsigned pos = (blockIdx.x * 256u + threadIdx.x) % nbuf4;
    const unsigned stride = gridDim.x * 256u;
    for (int it = 0; it < iters; it++) {
        for (int l = 0; l < nld; l++) {
            const float4 v = buf[pos];
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
            if (do_store) wbuf[pos] = g;
            pos += stride;
            if (pos >= nbuf4) pos -= nbuf4;
        }
        for (int l = 0; l < nlds; l++) {
            lds[(threadIdx.x + l * 257) & 4095] = av;
            av += lds[(threadIdx.x * 3 + l) & 4095];
        }
        const int ge = g_gap_every, gk = g_gap_kind;
        int since = 0;
        for (int m = 0; m < nmfma; m++) {
#pragma unroll
            for (int a = 0; a < NA; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            if (ge > 0 && ++since >= ge) {
                since = 0;
                if (gk == 1) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory"); }
                else if (gk == 2) { __builtin_amdgcn_s_sleep(1); }
                else if (gk == 3) { __builtin_amdgcn_s_sleep(4); }
                else if (gk == 4) { asm volatile("v_mov_b32 %0, %0" : "+v"(av)); }
                else if (gk == 5) { av += lds[threadIdx.x]; }
                else if (gk == 6) { asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory"); }
            }
        }
    }
    float s = g.x + g.y + g.z + g.w + av;
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[a][i];
    if (s == 12345.678f) sink[0] = s;
}

static hipStream_t g_s = nullptr;
static float4 *g_buf = nullptr, *g_wbuf = nullptr;
static float* g_sink = nullptr;
static unsigned g_nbuf4 = 0;
static hipEvent_t g_e0, g_e1;

extern "C" int nb_init(size_t buf_bytes, int prio) {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    // NB_DUMMY=n: n streams created (and used once) before the neighbour's, to move it to another hardware queue / pipe
    int nd = getenv("NB_DUMMY") ? atoi(getenv("NB_DUMMY")) : 0;
    for (int i = 0; i < nd; i++) {
        hipStream_t d;
        hipStreamCreateWithPriority(&d, hipStreamNonBlocking, prio ? hi : lo);
        void* p = nullptr;
        hipMalloc(&p, 256);
        hipMemsetAsync(p, 0, 256, d);
        hipStreamSynchronize(d);
    }
    if (hipStreamCreateWithPriority(&g_s, hipStreamNonBlocking, prio ? hi : lo) != hipSuccess) return -1;
    if (hipMalloc(&g_buf, buf_bytes) != hipSuccess) return -2;
    if (hipMalloc(&g_wbuf, buf_bytes) != hipSuccess) return -2;
    hipMemset(g_buf, 0, buf_bytes);
    hipMalloc(&g_sink, 64);
    g_nbuf4 = (unsigned)(buf_bytes / 16);
    hipEventCreate(&g_e0);
    hipEventCreate(&g_e1);
    return 0;
}

// launches `launches` kernels back to back; returns elapsed ms (blocks until done)
extern "C" float nb_run(int na, int grid, int lds_bytes, int launches, int iters, int nmfma, int nlds, int nld, int do_store,
                        unsigned footprint4) {
    unsigned n4 = footprint4 && footprint4 < g_nbuf4 ? footprint4 : g_nbuf4;
    if (lds_bytes < 16384) lds_bytes = 16384;
    hipEventRecord(g_e0, g_s);
    for (int i = 0; i < launches; i++) {
#define NB_CASE(N)                                                                                                         \
    if (na == N) {                                                                                                         \
        hipFuncSetAttribute((const void*)nb_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);            \
        hipLaunchKernelGGL(nb_kernel<N>, dim3(grid), dim3(256), lds_bytes, g_s, iters, nmfma, nlds, nld, g_buf, n4, do_store, \
                           g_wbuf, g_sink);                                                                                \
    }
        NB_CASE(1) NB_CASE(4) NB_CASE(7)
    }
    hipEventRecord(g_e1, g_s);
    if (hipEventSynchronize(g_e1) != hipSuccess) return -1.f;
    float ms = 0;
    hipEventElapsedTime(&ms, g_e0, g_e1);
    return ms;
}

// asynchronous form: enqueue and return; nb_wait blocks (or polls with sleeps when poll_us > 0) and returns elapsed ms
extern "C" int nb_launch(int na, int grid, int lds_bytes, int launches, int iters, int nmfma, int nlds, int nld, int do_store,
                         unsigned footprint4) {
    unsigned n4 = footprint4 && footprint4 < g_nbuf4 ? footprint4 : g_nbuf4;
    if (lds_bytes < 16384) lds_bytes = 16384;
    hipEventRecord(g_e0, g_s);
    for (int i = 0; i < launches; i++) {
        NB_CASE(1) NB_CASE(4) NB_CASE(7)
    }
    hipEventRecord(g_e1, g_s);
    return 0;
}
#include <unistd.h>
extern "C" float nb_wait(int poll_us) {
    if (poll_us > 0) {
        while (hipEventQuery(g_e1) == hipErrorNotReady) usleep(poll_us);
    } else if (hipEventSynchronize(g_e1) != hipSuccess) return -1.f;
    float ms = 0;
    hipEventElapsedTime(&ms, g_e0, g_e1);
    return ms;
}

extern "C" int nb_set_gap(int every, int kind) {
    hipMemcpyToSymbol(HIP_SYMBOL(g_gap_every), &every, sizeof(int));
    hipMemcpyToSymbol(HIP_SYMBOL(g_gap_kind), &kind, sizeof(int));
    return 0;
}
