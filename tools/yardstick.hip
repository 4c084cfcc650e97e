// Developer tool, not part of libbvq.so: what this chip sustains for the three access mixes of the headline step
// when the kernel does NO arithmetic -- one read stream (the statistic), one read + one write (the forward), two reads +
// one write (the backward).  Same access shape as the product kernels: 16 bytes per lane, 1 KiB per wave instruction,
// every load of a wave issued before anything is consumed, non-temporal or default cache policy.
//
//   unit form:        one short-lived wave per CH KiB of every stream (the product's decomposition)
//   persistent form:  `blocks` workgroups walk the streams with a grid stride, CH wave-chunks in flight per wave
//
// Built by tools/yardstick.py with hipcc --offload-arch=gfx950; C ABI, raw device pointers.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: read a; 1: o = a; 2: o = a ^ b
template <int MODE, bool NT, int CH>
__device__ __forceinline__ void walk(const u32x4* a, const u32x4* b, u32x4* o, int64_t first, int64_t nchunks, u32x4& acc) {
    u32x4 va[CH], vb[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) {
        int64_t c = first + (int64_t)i * 64;
        bool in = c < nchunks;
        va[i] = in ? ld<NT>(a + c) : u32x4{0, 0, 0, 0};
        if (MODE == 2) vb[i] = in ? ld<NT>(b + c) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < CH; i++) {
        int64_t c = first + (int64_t)i * 64;
        if (MODE == 0) acc ^= va[i];
        if (MODE == 1 && c < nchunks) st<NT>(o + c, va[i]);
        if (MODE == 2 && c < nchunks) st<NT>(o + c, va[i] ^ vb[i]);
    }
}

template <int MODE, bool NT, int CH>
__global__ __launch_bounds__(256) void unit_kernel(const u32x4* a, const u32x4* b, u32x4* o, uint32_t* sink, int64_t nchunks) {
    int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    u32x4 acc = {0, 0, 0, 0};
    walk<MODE, NT, CH>(a, b, o, wave * (64 * CH) + (threadIdx.x & 63), nchunks, acc);
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1;  // never true for random data: keeps the loads
}

template <int MODE, bool NT, int CH>
__global__ __launch_bounds__(256) void persistent_kernel(const u32x4* a, const u32x4* b, u32x4* o, uint32_t* sink, int64_t nchunks) {
    int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int64_t waves = (int64_t)gridDim.x * 4;
    u32x4 acc = {0, 0, 0, 0};
    for (int64_t first = wave * (64 * CH); first < nchunks; first += waves * (64 * CH))
        walk<MODE, NT, CH>(a, b, o, first + (threadIdx.x & 63), nchunks, acc);
    if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1;
}

template <int MODE, bool NT, int CH>
static int launch(int form, int blocks, const void* a, const void* b, void* o, void* sink, int64_t nchunks, hipStream_t s) {
    if (form == 0) {
        int64_t waves = (nchunks + 64 * CH - 1) / (64 * CH);
        unit_kernel<MODE, NT, CH><<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s>>>(
            (const u32x4*)a, (const u32x4*)b, (u32x4*)o, (uint32_t*)sink, nchunks);
    } else {
        persistent_kernel<MODE, NT, CH><<<dim3(blocks), dim3(256), 0, s>>>(
            (const u32x4*)a, (const u32x4*)b, (u32x4*)o, (uint32_t*)sink, nchunks);
    }
    return (int)hipGetLastError();
}

template <int MODE, bool NT>
static int by_ch(int ch, int form, int blocks, const void* a, const void* b, void* o, void* sink, int64_t n, hipStream_t s) {
    switch (ch) {
        case 2: return launch<MODE, NT, 2>(form, blocks, a, b, o, sink, n, s);
        case 4: return launch<MODE, NT, 4>(form, blocks, a, b, o, sink, n, s);
        case 8: return launch<MODE, NT, 8>(form, blocks, a, b, o, sink, n, s);
        default: return -1;
    }
}

// bytes must be a multiple of 16; returns 0 or a hipError / -1 for an unknown option
extern "C" int yardstick(int mode, int nt, int ch, int form, int blocks, const void* a, const void* b, void* o, void* sink,
                         int64_t bytes, void* stream) {
    int64_t n = bytes / 16;
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) return nt ? by_ch<0, true>(ch, form, blocks, a, b, o, sink, n, s) : by_ch<0, false>(ch, form, blocks, a, b, o, sink, n, s);
    if (mode == 1) return nt ? by_ch<1, true>(ch, form, blocks, a, b, o, sink, n, s) : by_ch<1, false>(ch, form, blocks, a, b, o, sink, n, s);
    if (mode == 2) return nt ? by_ch<2, true>(ch, form, blocks, a, b, o, sink, n, s) : by_ch<2, false>(ch, form, blocks, a, b, o, sink, n, s);
    return -1;
}
