// bvq_select.hip -- exact k-th value per channel (percentile statistics) by MSD radix select.
//
// Replaces torch.kthvalue on |x| or x (AbsPercentile / NegativePercentileOrZero / PercentileInterval,
// B/core/stats/stats_op.py:41-126), which the default activation quantizer
// (Int8ActPerTensorFloat: AbsPercentile(99.999), B/quant/base.py:68-75) runs on every one of its
// first 300 training steps -- a sort-class kernel over the whole activation in the reference.
//
// Values are mapped to order-preserving unsigned keys (16 bits for bf16/f16, 32 for f32; every NaN
// sorts last, as torch.kthvalue orders them) and the k-th key is found digit by digit from the top:
// each pass streams x once, histograms one 11-bit digit of the keys that match the prefix found so
// far (LDS, one private 2048-bin histogram per wave, flushed with global atomics), and a tiny
// kernel picks the bin that holds the k-th element.  16-bit types need 2 passes (11 + 5 bits), f32
// 3 (11 + 11 + 10); after the first pass almost no key matches the prefix, so later passes are pure
// streaming reads.  Algorithmic bytes per element: passes * sizeof(x).  The result is exact.
// One-channel (per-tensor) calls of bvq_kth_value take a wider first digit instead -- 15 bits, all 32768 bins in
// the LDS of one 1024-thread workgroup per CU -- and need one read (|x| of a 16-bit type) or two: see below.
#include "bvq_common.h"
#include "bvq_ties.h"

namespace bvq {

// Bin counters are uint32 end to end (per-wave LDS bins, the global histograms, the all-reduced shard sums): a bin
// may hold every element of a channel (a constant tensor, post-ReLU zeros), so a channel -- over ALL shards of a
// batch-sharded tensor -- must have fewer than 2^32 elements.  Enforced here per call and, for the shard count,
// by brevitas_amd.distributed.sharded_kth_value.
constexpr int64_t kSelMaxCount = 0xFFFFFFFFll;


constexpr int kDigitBits = 11;
constexpr int kBins = 1 << kDigitBits;
constexpr int kSelUnroll = 4;
constexpr int64_t kSelUnitCap = 4096;  // units per channel

// order-preserving key of a value
template <typename T, bool ABS>
__device__ __forceinline__ uint32_t sel_key(T v);

template <>
__device__ __forceinline__ uint32_t sel_key<float, true>(float v) {
  return __builtin_bit_cast(uint32_t, v) & 0x7fffffffu;  // NaN patterns exceed +inf
}
template <>
__device__ __forceinline__ uint32_t sel_key<float, false>(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  if (v != v) return 0xffffffffu;
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
template <typename T, bool ABS>
__device__ __forceinline__ uint32_t sel_key16(T v) {
  const uint32_t b = __builtin_bit_cast(uint16_t, v);
  if (ABS) return b & 0x7fffu;
  if (to_f<T>(v) != to_f<T>(v)) return 0xffffu;
  return b ^ ((b & 0x8000u) ? 0xffffu : 0x8000u);
}
template <>
__device__ __forceinline__ uint32_t sel_key<bf16_t, true>(bf16_t v) {
  return sel_key16<bf16_t, true>(v);
}
template <>
__device__ __forceinline__ uint32_t sel_key<bf16_t, false>(bf16_t v) {
  return sel_key16<bf16_t, false>(v);
}
template <>
__device__ __forceinline__ uint32_t sel_key<f16_t, true>(f16_t v) {
  return sel_key16<f16_t, true>(v);
}
template <>
__device__ __forceinline__ uint32_t sel_key<f16_t, false>(f16_t v) {
  return sel_key16<f16_t, false>(v);
}

struct SelArgs {
  Tiling t;
  const void* x;
  const uint32_t* prefix;  // [channels] key bits fixed by earlier passes (already shifted down)
  uint32_t* hist;          // [channels][kBins] of this pass
  int32_t shift;           // this pass looks at (key >> shift) & (kBins - 1) ...
  int32_t bits;            // ... of which `bits` low bits are significant (last pass may be narrower)
  int32_t first_pass;      // no prefix to match yet
};

template <typename T, int VEC, bool ABS, bool NT>
__global__ __launch_bounds__(kBlock) void kth_hist_kernel(SelArgs a) {
  __shared__ uint32_t lds[kWavesPerBlock][kBins];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  uint32_t* h = lds[wave];
  for (int b = lane; b < kBins; b += kWave) h[b] = 0;
  __syncthreads();  // every wave reaches this (the unit check comes after)
  const Unit u = locate_unit(a.t);
  if (!u.valid) return;
  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + u.base;
  const uint32_t want = a.first_pass ? 0u : a.prefix[u.channel];
  const int hi_shift = a.shift + a.bits;  // bits above this pass's digit
  const uint32_t mask = (1u << a.bits) - 1u;
  ChunkWalker cur;
  cur.init(u, VEC, lane);
  const int64_t total = (int64_t)u.nrows * cur.cpr;
  for (int64_t done = 0; done < total; done += (int64_t)kWave * kSelUnroll) {
    vec_t<T, VEC> xv[kSelUnroll];
    bool ok[kSelUnroll];
#pragma unroll
    for (int j = 0; j < kSelUnroll; ++j) {
      ok[j] = cur.valid();
      if (ok[j]) xv[j] = load_vec<T, VEC, NT>(xp + cur.offset(u.row_stride, VEC));
      cur.next();
    }
#pragma unroll
    for (int j = 0; j < kSelUnroll; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t key = sel_key<T, ABS>(xv[j].v[k]);
          const bool match = a.first_pass || (hi_shift >= 32 ? true : (key >> hi_shift) == want);
          if (match) atomicAdd(&h[(key >> a.shift) & mask], 1u);
        }
      }
    }
  }
  const int64_t i = (int64_t)cur.cpr * VEC + lane;
  if (u.nrows == 1 && i < u.len) {
    const uint32_t key = sel_key<T, ABS>(xp[i]);
    const bool match = a.first_pass || (hi_shift >= 32 ? true : (key >> hi_shift) == want);
    if (match) atomicAdd(&h[(key >> a.shift) & mask], 1u);
  }
  // flush the non-empty bins of this wave's histogram
  uint32_t* gh = a.hist + (int64_t)u.channel * kBins;
  for (int b = lane; b < kBins; b += kWave) {
    const uint32_t c = h[b];
    if (c) atomicAdd(&gh[b], c);
  }
}

// one workgroup per channel: find the bin that holds the k-th (1-indexed) matching key, append its
// digit to the prefix and make k relative to that bin
// rank rule of the percentile statistics (B/core/stats/stats_op.py:56,84,114-116), evaluated on the
// device from the number of elements the first histogram counted -- for a tensor sharded over several
// devices that is the GLOBAL count once the histograms have been summed, with no host round trip.
// python evaluates `.01 * q * n` left to right in double; IEEE doubles give the same bits here.
__device__ __forceinline__ int64_t rank_from_rule(int rule, double q, int64_t n) {
  double v = 0.01 * q;
  v = v * (double)n;
  int64_t k = rule == BVQ_KTH_HIGH ? (int64_t)floor(v + 0.5) : (int64_t)ceil(v);
  if (k < 1) k = 1;  // torch.kthvalue raises for k = 0; a device kernel cannot -- documented in bvq.h
  if (k > n) k = n;
  return k;
}

__global__ __launch_bounds__(kBlock) void kth_pick_kernel(const uint32_t* __restrict__ hist,
                                                          uint32_t* __restrict__ prefix,
                                                          int64_t* __restrict__ krem, int32_t bits,
                                                          int32_t rule, double q) {
  __shared__ int64_t part[kBlock];
  const int c = blockIdx.x;
  const uint32_t* h = hist + (int64_t)c * kBins;
  constexpr int kPer = kBins / kBlock;  // consecutive bins per thread
  int64_t mine = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j) mine += h[threadIdx.x * kPer + j];
  part[threadIdx.x] = mine;
  __syncthreads();
  // exclusive prefix over the 256 partial sums (serial on one thread: 256 adds)
  __shared__ int64_t before_me[kBlock];
  __shared__ int64_t k_sh;
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int t = 0; t < kBlock; ++t) {
      before_me[t] = run;
      run += part[t];
    }
    k_sh = rule != BVQ_KTH_EXPLICIT ? rank_from_rule(rule, q, run) : krem[c];
  }
  __syncthreads();
  const int64_t k = k_sh;
  int64_t run = before_me[threadIdx.x];
  if (k > run && k <= run + mine) {  // exactly one thread owns the k-th element
    for (int j = 0; j < kPer; ++j) {
      const int64_t cnt = h[threadIdx.x * kPer + j];
      if (k <= run + cnt) {
        prefix[c] = (prefix[c] << bits) | (uint32_t)(threadIdx.x * kPer + j);
        krem[c] = k - run;
        break;
      }
      run += cnt;
    }
  }
}

template <typename T, bool ABS>
__global__ void kth_store_kernel(const uint32_t* __restrict__ prefix, void* out, int32_t channels) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= channels) return;
  const uint32_t key = prefix[c];
  T* o = reinterpret_cast<T*>(out);
  if constexpr (sizeof(T) == 4) {
    const uint32_t b = ABS ? key : ((key >> 31) ? (key ^ 0x80000000u) : ~key);
    o[c] = __builtin_bit_cast(float, b);
  } else {
    const uint16_t b = (uint16_t)(ABS ? key : ((key & 0x8000u) ? (key ^ 0x8000u) : (~key & 0xffffu)));
    o[c] = __builtin_bit_cast(T, b);
  }
}

__global__ void kth_init_kernel(uint32_t* prefix, int64_t* krem, int64_t k, int32_t channels) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < channels) {
    prefix[c] = 0;
    krem[c] = k;
  }
}

template <typename T, bool ABS>
static void launch_hist(const SelArgs& a, int vec, bool nt, hipStream_t st) {
  constexpr int V = elem<T>::vec;
  const dim3 grid(grid_for_units(a.t.units)), block(kBlock);
  if (vec == V && nt)
    kth_hist_kernel<T, V, ABS, true><<<grid, block, 0, st>>>(a);
  else if (vec == V)
    kth_hist_kernel<T, V, ABS, false><<<grid, block, 0, st>>>(a);
  else
    kth_hist_kernel<T, 1, ABS, false><<<grid, block, 0, st>>>(a);
}


// ---- per-tensor route of bvq_kth_value: 15-bit first digit in LDS, the rest through global counters ----------
// A workgroup of 1024 threads keeps ALL 32768 bins of the key's top 15 bits in LDS (128 of gfx950's 160 KB).
//  * |x| of a 16-bit type: that IS the whole key -- one streaming read decides the k-th value, where the digit
//    passes above read twice; and the keys spread over 16x more bins than an 11-bit digit, which is what the
//    LDS atomics of that first pass choke on (a bf16 activation puts ~5 % of its elements into the hottest
//    11-bit bin, 0.3 % into the hottest 15-bit one).
//  * every other key (16 signed bits, 31 or 32 bits of float32): a second read histograms the remaining
//    1 / 16 / 17 low bits of the few elements that fell into the chosen bin, straight into global counters
//    (a wave whose matching lanes all hold the same key adds their count once: constant tensors stay cheap).
// Two reads instead of three for float32, exact like the digit passes.
constexpr int kBins15 = 1 << 15;
constexpr int kBlock15 = 1024;
constexpr int kLowBitsMax = 17;

__global__ void kth_zero_kernel(uint32_t* p, int32_t n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0;
}

// head: elements in front of x (x - head .. x) that precede the first 16-byte boundary of the caller's buffer
template <typename T, bool ABS, bool NT>
__global__ __launch_bounds__(kBlock15) void kth_hist15_kernel(const T* __restrict__ x, int64_t n, int32_t shift,
                                                             uint32_t* __restrict__ hist, int32_t head = 0) {
  constexpr int VEC = elem<T>::vec;
  __shared__ uint32_t h[kBins15];
  for (int b = threadIdx.x; b < kBins15; b += kBlock15) h[b] = 0;
  __syncthreads();
  if (blockIdx.x == 0 && (int32_t)threadIdx.x < head)
    atomicAdd(&h[sel_key<T, ABS>(x[(int64_t)threadIdx.x - head]) >> shift], 1u);
  const int64_t chunks = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock15;
  for (int64_t c = (int64_t)blockIdx.x * kBlock15 + threadIdx.x; c < chunks; c += stride * kSelUnroll) {
    vec_t<T, VEC> xv[kSelUnroll];
    bool ok[kSelUnroll];
#pragma unroll
    for (int j = 0; j < kSelUnroll; ++j) {
      const int64_t cj = c + (int64_t)j * stride;
      ok[j] = cj < chunks;
      if (ok[j]) xv[j] = load_vec<T, VEC, NT>(x + cj * VEC);
    }
#pragma unroll
    for (int j = 0; j < kSelUnroll; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) atomicAdd(&h[sel_key<T, ABS>(xv[j].v[k]) >> shift], 1u);
      }
    }
  }
  if (blockIdx.x == 0 && chunks * VEC + threadIdx.x < n)  // the < VEC elements after the last chunk
    atomicAdd(&h[sel_key<T, ABS>(x[chunks * VEC + threadIdx.x]) >> shift], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < kBins15; b += kBlock15) {
    const uint32_t c = h[b];
    if (c) atomicAdd(&hist[b], c);
  }
}

// second read: low `shift` bits of the keys whose top 15 bits equal sel[0]
__device__ __forceinline__ void low_count(uint32_t key, uint32_t want, int32_t shift, uint32_t lowmask,
                                          uint32_t* __restrict__ ghist) {
  if ((key >> shift) == want) {
    const uint32_t low = key & lowmask;
    const uint32_t first = __builtin_amdgcn_readfirstlane(low);
    const uint64_t active = __builtin_amdgcn_ballot_w64(true);
    if (__builtin_amdgcn_ballot_w64(low != first) == 0) {
      if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(active)) atomicAdd(&ghist[first], (uint32_t)__builtin_popcountll(active));
    } else {
      atomicAdd(&ghist[low], 1u);
    }
  }
}

// SMALL: at most 11 low bits (the last bit of a signed 16-bit key): the matching elements of a workgroup meet in
// an LDS histogram first -- with 2 bins, one global atomic per element would be ~10^4 serialized updates of
// the same two addresses
template <typename T, bool ABS, bool NT, bool SMALL>
__global__ __launch_bounds__(kBlock) void kth_low_kernel(const T* __restrict__ x, int64_t n, int32_t shift,
                                                        const uint32_t* __restrict__ sel, uint32_t* __restrict__ gh,
                                                        int32_t nk, int32_t head = 0) {
  // nk (1 or 2) ranks are served by the same read: rank i has its own chosen bin sel[i] and its own counters
  constexpr int VEC = elem<T>::vec;
  __shared__ uint32_t lh[SMALL ? 2 * kBins : 1];
  if constexpr (SMALL) {
    for (int b = threadIdx.x; b < 2 * kBins; b += kBlock) lh[b] = 0;
    __syncthreads();
  }
  uint32_t* ghist = SMALL ? lh : gh;
  const int64_t hstride = SMALL ? kBins : ((int64_t)1 << kLowBitsMax);
  const uint32_t want = sel[0], want2 = nk > 1 ? sel[1] : 0u;
  const uint32_t lowmask = (1u << shift) - 1u;
  const int64_t chunks = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x; c < chunks; c += stride * kSelUnroll) {
    vec_t<T, VEC> xv[kSelUnroll];
    bool ok[kSelUnroll];
#pragma unroll
    for (int j = 0; j < kSelUnroll; ++j) {
      const int64_t cj = c + (int64_t)j * stride;
      ok[j] = cj < chunks;
      if (ok[j]) xv[j] = load_vec<T, VEC, NT>(x + cj * VEC);
    }
#pragma unroll
    for (int j = 0; j < kSelUnroll; ++j) {
      if (ok[j]) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
          const uint32_t key = sel_key<T, ABS>(xv[j].v[k]);
          low_count(key, want, shift, lowmask, ghist);
          if (nk > 1) low_count(key, want2, shift, lowmask, ghist + hstride);
        }
      }
    }
  }
  if (blockIdx.x == 0 && chunks * VEC + threadIdx.x < n) {
    const uint32_t key = sel_key<T, ABS>(x[chunks * VEC + threadIdx.x]);
    low_count(key, want, shift, lowmask, ghist);
    if (nk > 1) low_count(key, want2, shift, lowmask, ghist + hstride);
  }
  if (blockIdx.x == 0 && (int32_t)threadIdx.x < head) {  // (a separate branch: low_count's ballots see whole waves)
    const uint32_t key = sel_key<T, ABS>(x[(int64_t)threadIdx.x - head]);
    low_count(key, want, shift, lowmask, ghist);
    if (nk > 1) low_count(key, want2, shift, lowmask, ghist + hstride);
  }
  if constexpr (SMALL) {
    __syncthreads();
    for (int i = 0; i < nk; ++i)
      for (int b = threadIdx.x; b < (1 << shift); b += kBlock) {
        const uint32_t c = lh[i * kBins + b];
        if (c) atomicAdd(&gh[(int64_t)i * ((int64_t)1 << kLowBitsMax) + b], c);
      }
  }
}

// one workgroup: the bin of hist[0, nbins) that holds the k-th (1-indexed) element; sel[0] <- sel[0] << bits | bin
// (sel[0] starts at 0), krem[0] <- rank inside that bin.  first: k comes as an argument, else from krem[0].
// rule / q (first pick only): the rank from the total count of the histogram -- the GLOBAL element count once the
// shards' histograms have been summed -- instead of k_arg
__global__ __launch_bounds__(kBlock15) void kth_pick_wide_kernel(const uint32_t* __restrict__ hist, int32_t nbins,
                                                                int32_t bits, int32_t first, int64_t k_arg,
                                                                uint32_t* __restrict__ sel, int64_t* __restrict__ krem,
                                                                int32_t rule = BVQ_KTH_EXPLICIT, double q = 0.0) {
  __shared__ int64_t part[kBlock15];
  __shared__ int64_t before_me[kBlock15];
  __shared__ int64_t total;
  const int per = (nbins + kBlock15 - 1) / kBlock15;
  const int lo = threadIdx.x * per, hi = lo + per < nbins ? lo + per : nbins;
  int64_t mine = 0;
  for (int b = lo; b < hi; ++b) mine += hist[b];
  part[threadIdx.x] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int t = 0; t < kBlock15; ++t) {
      before_me[t] = run;
      run += part[t];
    }
    total = run;
  }
  __syncthreads();
  const int64_t k = first ? (rule != BVQ_KTH_EXPLICIT ? rank_from_rule(rule, q, total) : k_arg) : krem[0];
  const uint32_t prev = first ? 0u : sel[0];
  __syncthreads();  // everyone has read krem / sel before the owner overwrites them
  int64_t run = before_me[threadIdx.x];
  if (k > run && k <= run + mine) {
    for (int b = lo; b < hi; ++b) {
      const int64_t cnt = hist[b];
      if (k <= run + cnt) {
        sel[0] = (prev << bits) | (uint32_t)b;
        krem[0] = k - run;
        break;
      }
      run += cnt;
    }
  }
}

}  // namespace bvq

using namespace bvq;

// ---- channel-last layouts (short `inner`): select on a transposed copy -----------------------------------------
// A per-channel radix select needs a channel's elements together: one private 2048-bin LDS histogram per wave is what
// makes the first digit pass cheap, and `channels x 2048` counters do not fit on chip for a wave that sees every
// channel in each row (channel-last [tokens, hidden], NHWC).  Round 2 sent such layouts down the row-mapped route with
// one-element rows (27 ms for AbsPercentile on [802816,512] bf16).  Round 3: the tensor, seen as a [rows][L = channels *
// inner] matrix, is transposed once into the caller's workspace -- [L][rows]: a channel's inner * rows elements are
// contiguous, in another order, which a k-th VALUE does not care about -- and every digit pass reads the copy as
// (outer = 1, channels, inner * rows).  64 x 64 tiles through LDS, 16-byte loads and stores: one read + one write.
constexpr int kTile = 64;

template <typename T>
__global__ __launch_bounds__(kBlock) void transpose_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t R,
                                                           int64_t L) {
  constexpr int VEC = 16 / sizeof(T);              // elements per 16-byte access
  constexpr int kTPR = kTile / VEC;                // threads per tile row
  constexpr int kRPI = kBlock / kTPR;              // tile rows per load iteration
  __shared__ T tile[kTile][kTile + 16 / sizeof(T) / 4 * 2 + 2];  // padded pitch: the column reads below spread over the banks
  const int64_t r0 = (int64_t)blockIdx.x * kTile, c0 = (int64_t)blockIdx.y * kTile;
  const int tr = threadIdx.x / kTPR, tc = (threadIdx.x % kTPR) * VEC;
#pragma unroll
  for (int i = 0; i < kTile / kRPI; ++i) {
    const int r = tr + i * kRPI;
    if (r0 + r < R && c0 + tc < L) {  // (L is a multiple of VEC: a chunk is whole or absent)
      const vec_t<T, VEC> v = load_vec<T, VEC>(x + (r0 + r) * L + c0 + tc);
#pragma unroll
      for (int k = 0; k < VEC; ++k) tile[r][tc + k] = v.v[k];
    }
  }
  __syncthreads();
  // output row (c0 + oc) = tile column oc; a thread writes VEC consecutive original rows of it
  const bool aligned = (R * (int64_t)sizeof(T)) % 16 == 0;
#pragma unroll
  for (int i = 0; i < kTile / kRPI; ++i) {
    const int oc = tr + i * kRPI, orow = tc;
    if (c0 + oc >= L || r0 + orow >= R) continue;
    T* dst = out + (c0 + oc) * R + r0 + orow;
    if (aligned && r0 + orow + VEC <= R) {
      vec_t<T, VEC> v;
#pragma unroll
      for (int k = 0; k < VEC; ++k) v.v[k] = tile[orow + k][oc];
      store_vec<T, VEC>(dst, v);
    } else {
      for (int k = 0; k < VEC && r0 + orow + k < R; ++k) dst[k] = tile[orow + k][oc];
    }
  }
}

// the layouts that take the transposed copy, and where it lives in the workspace
static bool cols_select_applies(int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner) {
  return x && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && cols_plan(dtype, outer, channels, inner).ok;
}

static int passes_for(int dtype) { return dtype == BVQ_F32 ? 3 : 2; }

// workspace layout (depends on dtype and channels only):
//   [passes][channels][kBins] uint32 histograms | [channels] int64 remaining rank | [channels] uint32 prefix
struct SelWorkspace {
  uint32_t* hist;
  int64_t* krem;
  uint32_t* prefix;
  int64_t hist_words;
};

static int64_t sel_steps_bytes(int dtype, int64_t channels) {
  return (int64_t)passes_for(dtype) * channels * kBins * (int64_t)sizeof(uint32_t) +
         channels * (int64_t)(sizeof(uint32_t) + sizeof(int64_t)) + 256;
}

// (+ the per-tensor route of bvq_kth_value behind the stepwise layout: 32768 + 2^17 counters, key, rank)
static int64_t sel_wide_bytes() {
  return ((int64_t)kBins15 + 2 * ((int64_t)1 << kLowBitsMax)) * (int64_t)sizeof(uint32_t) + 64;
}

static int64_t sel_workspace_bytes(int dtype, int64_t channels) {
  const int64_t base = (sel_steps_bytes(dtype, channels) + 255) / 256 * 256;
  return base + (channels == 1 ? sel_wide_bytes() : 0);
}

static SelWorkspace sel_workspace(int dtype, int64_t channels, void* workspace) {
  SelWorkspace w;
  w.hist = reinterpret_cast<uint32_t*>(workspace);
  w.hist_words = (int64_t)passes_for(dtype) * channels * kBins;
  w.krem = reinterpret_cast<int64_t*>(w.hist + ((w.hist_words + 1) / 2) * 2);  // 8-byte aligned
  w.prefix = reinterpret_cast<uint32_t*>(w.krem + channels);
  return w;
}

static void pass_bits(int dtype, int pass, int& bits, int& shift) {
  const int key_bits = dtype == BVQ_F32 ? 32 : 16;
  int done = 0;
  for (int p = 0;; ++p) {
    bits = (key_bits - done) < kDigitBits ? (key_bits - done) : kDigitBits;
    shift = key_bits - done - bits;
    if (p == pass) return;
    done += bits;
  }
}

static int sel_check(const char* who, int dtype, int64_t channels, const void* workspace,
                     int64_t workspace_bytes) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16 || channels < 1) {
    set_error("%s: bad argument", who);
    return BVQ_ERR_INVALID;
  }
  if (!workspace || workspace_bytes < sel_workspace_bytes(dtype, channels)) {
    set_error("%s: workspace %lld < %lld bytes", who, (long long)workspace_bytes,
              (long long)sel_workspace_bytes(dtype, channels));
    return BVQ_ERR_WORKSPACE;
  }
  return BVQ_OK;
}

// (the transposed copy of a channel-last tensor sits behind the select's own state; sized whether or not x will turn out
//  16-byte aligned)
static int64_t sel_scratch_offset(int dtype, int64_t channels) { return (sel_workspace_bytes(dtype, channels) + 255) / 256 * 256; }

extern "C" int64_t bvq_kth_workspace_bytes(int dtype, int64_t outer, int64_t channels, int64_t inner) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16 || outer < 0 || channels < 1 || inner < 0) return -1;
  if (cols_plan(dtype, outer, channels, inner).ok)
    return sel_scratch_offset(dtype, channels) + outer * channels * inner * (int64_t)dtype_size(dtype) + 256;
  return sel_workspace_bytes(dtype, channels);
}

extern "C" int bvq_kth_passes(int dtype) {
  return (dtype < BVQ_F32 || dtype > BVQ_F16) ? -1 : passes_for(dtype);
}

extern "C" int64_t bvq_kth_hist_offset(int dtype, int64_t channels, int pass) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16 || channels < 1 || pass < 0 || pass >= passes_for(dtype)) return -1;
  return (int64_t)pass * channels * kBins * (int64_t)sizeof(uint32_t);
}

extern "C" int bvq_kth_begin(int dtype, int64_t channels, int rule, int64_t k, double q, void* workspace,
                             int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = sel_check("bvq_kth_begin", dtype, channels, workspace, workspace_bytes);
  if (rc) return rc;
  if (rule < BVQ_KTH_EXPLICIT || rule > BVQ_KTH_LOW || (rule == BVQ_KTH_EXPLICIT && k < 1) ||
      (rule != BVQ_KTH_EXPLICIT && !(q >= 0.0 && q <= 100.0))) {
    set_error("bvq_kth_begin: bad rank (rule %d, k %lld, q %g)", rule, (long long)k, q);
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const SelWorkspace w = sel_workspace(dtype, channels, workspace);
  // (a kernel, not hipMemsetAsync: every step of the select is then an ordinary launch under graph capture)
  kth_zero_kernel<<<dim3((unsigned)((w.hist_words + 255) / 256)), dim3(256), 0, st>>>(w.hist, (int32_t)w.hist_words);
  kth_init_kernel<<<dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, st>>>(
      w.prefix, w.krem, rule == BVQ_KTH_EXPLICIT ? k : 0, (int32_t)channels);
  return check_launch("bvq_kth_begin");
}

extern "C" int bvq_kth_hist(int abs_key, int dtype, const void* x, int64_t outer, int64_t channels,
                            int64_t inner, int pass, void* workspace, int64_t workspace_bytes,
                            bvq_stream_t stream) {
  int rc = sel_check("bvq_kth_hist", dtype, channels, workspace, workspace_bytes);
  if (rc) return rc;
  if (outer < 0 || inner < 0 || pass < 0 || pass >= passes_for(dtype)) {
    set_error("bvq_kth_hist: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (outer * inner == 0) return BVQ_OK;  // an empty shard adds nothing to the histogram
  if (outer * inner > kSelMaxCount) {
    set_error("bvq_kth_hist: %lld elements per channel; the digit counters are 32-bit (limit 2^32 - 1 over all shards)",
              (long long)(outer * inner));
    return BVQ_ERR_UNSUPPORTED;
  }
  if (!x) {
    set_error("bvq_kth_hist: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const SelWorkspace w = sel_workspace(dtype, channels, workspace);
  if (cols_select_applies(dtype, x, outer, channels, inner)) {
    // channel-last: pass 0 transposes x into the workspace, every pass reads the copy (outer = 1, inner * outer per channel)
    const int64_t off = sel_scratch_offset(dtype, channels);
    const int64_t n_bytes = outer * channels * inner * (int64_t)dtype_size(dtype);
    if (workspace_bytes < off + n_bytes) {
      set_error("bvq_kth_hist: workspace %lld < %lld bytes (channel-last layout: bvq_kth_workspace_bytes with the shape)",
                (long long)workspace_bytes, (long long)(off + n_bytes));
      return BVQ_ERR_WORKSPACE;
    }
    void* copy = reinterpret_cast<char*>(workspace) + off;
    if (pass == 0) {
      const int64_t L = channels * inner;
      const dim3 tg((unsigned)((outer + kTile - 1) / kTile), (unsigned)((L + kTile - 1) / kTile));
      if (tg.y > 65535) {
        set_error("bvq_kth_hist: %lld columns exceed the transpose grid", (long long)L);
        return BVQ_ERR_UNSUPPORTED;
      }
      if (dtype == BVQ_F32)
        transpose_kernel<float><<<tg, dim3(kBlock), 0, st>>>(reinterpret_cast<const float*>(x),
                                                            reinterpret_cast<float*>(copy), outer, L);
      else  // bf16 / f16: 2-byte elements move as bit patterns
        transpose_kernel<uint16_t><<<tg, dim3(kBlock), 0, st>>>(reinterpret_cast<const uint16_t*>(x),
                                                               reinterpret_cast<uint16_t*>(copy), outer, L);
      rc = check_launch("bvq_kth_hist/transpose");
      if (rc) return rc;
    }
    x = copy;
    inner = outer * inner;
    outer = 1;
  }
  const int64_t t_outer = channels > 1 ? outer : 1;
  const int64_t row_len = channels > 1 ? inner : outer * inner;
  const int full = 16 / dtype_size(dtype);
  const void* ptrs[1] = {x};
  const int els[1] = {dtype_size(dtype)};
  int vec = pick_vec(full, t_outer * channels, row_len, ptrs, els, 1);
  vec = vec == full ? full : 1;
  SelArgs a;
  // every wave flushes a 2048-bin histogram: keep the waves few and their pieces long
  a.t = make_tiling(t_outer, (int32_t)channels, row_len, vec, kSelUnitCap);
  a.x = x;
  a.prefix = w.prefix;
  a.hist = w.hist + (int64_t)pass * channels * kBins;
  int bits, shift;
  pass_bits(dtype, pass, bits, shift);
  a.bits = bits;
  a.shift = shift;
  a.first_pass = pass == 0;
  const bool nt = outer * channels * inner * (int64_t)dtype_size(dtype) >= nt_threshold_bytes();
#define BVQ_HIST(T)                             \
  do {                                          \
    if (abs_key)                                \
      launch_hist<T, true>(a, vec, nt, st);     \
    else                                        \
      launch_hist<T, false>(a, vec, nt, st);    \
  } while (0)
  if (dtype == BVQ_F32)
    BVQ_HIST(float);
  else if (dtype == BVQ_BF16)
    BVQ_HIST(bf16_t);
  else
    BVQ_HIST(f16_t);
#undef BVQ_HIST
  return check_launch("bvq_kth_hist");
}

extern "C" int bvq_kth_pick(int dtype, int64_t channels, int pass, int rule, double q, void* workspace,
                            int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = sel_check("bvq_kth_pick", dtype, channels, workspace, workspace_bytes);
  if (rc) return rc;
  if (pass < 0 || pass >= passes_for(dtype) || rule < BVQ_KTH_EXPLICIT || rule > BVQ_KTH_LOW) {
    set_error("bvq_kth_pick: bad argument");
    return BVQ_ERR_INVALID;
  }
  const SelWorkspace w = sel_workspace(dtype, channels, workspace);
  int bits, shift;
  pass_bits(dtype, pass, bits, shift);
  // the rank rule applies once, to the count of the first pass; later passes continue from krem
  kth_pick_kernel<<<dim3((unsigned)channels), dim3(kBlock), 0, (hipStream_t)stream>>>(
      w.hist + (int64_t)pass * channels * kBins, w.prefix, w.krem, bits, pass == 0 ? rule : BVQ_KTH_EXPLICIT, q);
  return check_launch("bvq_kth_pick");
}

extern "C" int bvq_kth_finish(int abs_key, int dtype, int64_t channels, void* out, void* workspace,
                              int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = sel_check("bvq_kth_finish", dtype, channels, workspace, workspace_bytes);
  if (rc) return rc;
  if (!out) {
    set_error("bvq_kth_finish: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const SelWorkspace w = sel_workspace(dtype, channels, workspace);
  const unsigned nb = (unsigned)((channels + 255) / 256);
#define BVQ_STORE(T)                                                                       \
  do {                                                                                     \
    if (abs_key)                                                                           \
      kth_store_kernel<T, true><<<dim3(nb), dim3(256), 0, st>>>(w.prefix, out, (int32_t)channels);  \
    else                                                                                   \
      kth_store_kernel<T, false><<<dim3(nb), dim3(256), 0, st>>>(w.prefix, out, (int32_t)channels); \
  } while (0)
  if (dtype == BVQ_F32)
    BVQ_STORE(float);
  else if (dtype == BVQ_BF16)
    BVQ_STORE(bf16_t);
  else
    BVQ_STORE(f16_t);
#undef BVQ_STORE
  return check_launch("bvq_kth_finish");
}

// per-tensor route (see kth_hist15_kernel): nk = 1 or 2 ranks from ONE histogram read (+ one read for the low bits
// of both); out[nk]
static bool wide_select_applies(int64_t channels, int64_t n, const void* x) {
  static const int wide = env_flag("BVQ_KTH_WIDE", 1);
  return wide && channels == 1 && n >= ((int64_t)1 << 22) && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

static int wide_select(int abs_key, int dtype, const void* x, int64_t n, const int64_t* ks, int nk, void* out,
                       void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  if (workspace_bytes < sel_workspace_bytes(dtype, 1)) {
    set_error("bvq_kth_value: workspace too small");
    return BVQ_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* base = reinterpret_cast<char*>(workspace) + (sel_steps_bytes(dtype, 1) + 255) / 256 * 256;
  uint32_t* hist = reinterpret_cast<uint32_t*>(base);  // [32768]
  uint32_t* low = hist + kBins15;                      // [2][1 << 17]
  const int64_t lstride = (int64_t)1 << kLowBitsMax;
  int64_t* krem = reinterpret_cast<int64_t*>(low + 2 * lstride);  // [2]
  uint32_t* sel = reinterpret_cast<uint32_t*>(krem + 2);          // [2]
  const int key_bits = (dtype == BVQ_F32 ? 32 : 16) - (abs_key ? 1 : 0);
  const int shift = key_bits - 15;  // 0, 1, 16 or 17 low bits left for the second read
  const int32_t words = kBins15 + (shift ? (int32_t)(2 * lstride) : 0);
  kth_zero_kernel<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st>>>(hist, words);
  const bool nt = n * (int64_t)dtype_size(dtype) >= nt_threshold_bytes();
  // one workgroup per CU is all the LDS allows; 256 of them cover the chip
#define BVQ_WIDE(T, ABS)                                                                                          \
  do {                                                                                                            \
    const T* xp = reinterpret_cast<const T*>(x);                                                                  \
    if (nt)                                                                                                       \
      kth_hist15_kernel<T, ABS, true><<<dim3(256), dim3(kBlock15), 0, st>>>(xp, n, shift, hist);                  \
    else                                                                                                          \
      kth_hist15_kernel<T, ABS, false><<<dim3(256), dim3(kBlock15), 0, st>>>(xp, n, shift, hist);                 \
    for (int i = 0; i < nk; ++i)                                                                                  \
      kth_pick_wide_kernel<<<dim3(1), dim3(kBlock15), 0, st>>>(hist, kBins15, 15, 1, ks[i], sel + i, krem + i);   \
    if (shift) {                                                                                                  \
      if (shift <= kDigitBits)                                                                                    \
        kth_low_kernel<T, ABS, false, true><<<dim3(1024), dim3(kBlock), 0, st>>>(xp, n, shift, sel, low, nk);     \
      else if (nt)                                                                                                \
        kth_low_kernel<T, ABS, true, false><<<dim3(2048), dim3(kBlock), 0, st>>>(xp, n, shift, sel, low, nk);     \
      else                                                                                                        \
        kth_low_kernel<T, ABS, false, false><<<dim3(2048), dim3(kBlock), 0, st>>>(xp, n, shift, sel, low, nk);    \
      for (int i = 0; i < nk; ++i)                                                                                \
        kth_pick_wide_kernel<<<dim3(1), dim3(kBlock15), 0, st>>>(low + i * lstride, 1 << shift, shift, 0, 0,      \
                                                                 sel + i, krem + i);                              \
    }                                                                                                             \
    kth_store_kernel<T, ABS><<<dim3(1), dim3(64), 0, st>>>(sel, out, nk);                                         \
  } while (0)
#define BVQ_WIDE_DT(ABS)        \
  do {                          \
    if (dtype == BVQ_F32)       \
      BVQ_WIDE(float, ABS);     \
    else if (dtype == BVQ_BF16) \
      BVQ_WIDE(bf16_t, ABS);    \
    else                        \
      BVQ_WIDE(f16_t, ABS);     \
  } while (0)
  if (abs_key)
    BVQ_WIDE_DT(true);
  else
    BVQ_WIDE_DT(false);
#undef BVQ_WIDE_DT
#undef BVQ_WIDE
  return check_launch("bvq_kth_value/wide");
}

// ---- the per-tensor route in steps, for a batch-sharded tensor (include/bvq.h, bvq_kthw_*) --------------------
struct WideWs {
  uint32_t* hist;  // [32768]
  uint32_t* low;   // [2][1 << 17]
  int64_t* krem;   // [2]
  uint32_t* sel;   // [2]
};
static WideWs wide_ws(int dtype, void* workspace) {
  WideWs w;
  char* base = reinterpret_cast<char*>(workspace) + (sel_steps_bytes(dtype, 1) + 255) / 256 * 256;
  w.hist = reinterpret_cast<uint32_t*>(base);
  w.low = w.hist + kBins15;
  w.krem = reinterpret_cast<int64_t*>(w.low + 2 * ((int64_t)1 << kLowBitsMax));
  w.sel = reinterpret_cast<uint32_t*>(w.krem + 2);
  return w;
}
static int wide_shift(int abs_key, int dtype) { return (dtype == BVQ_F32 ? 32 : 16) - (abs_key ? 1 : 0) - 15; }

static int kthw_check(const char* who, int dtype, const void* workspace, int64_t workspace_bytes) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16) {
    set_error("%s: bad dtype", who);
    return BVQ_ERR_INVALID;
  }
  if (!workspace || workspace_bytes < sel_workspace_bytes(dtype, 1)) {
    set_error("%s: workspace %lld < %lld bytes", who, (long long)workspace_bytes, (long long)sel_workspace_bytes(dtype, 1));
    return BVQ_ERR_WORKSPACE;
  }
  return BVQ_OK;
}

extern "C" int bvq_kthw_plan(int abs_key, int dtype, int pass, int64_t* hist_offset_bytes, int64_t* hist_words) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16) return -1;
  const int shift = wide_shift(abs_key, dtype);
  const int passes = shift ? 2 : 1;
  if (pass < 0 || pass >= passes) return passes;  // (a query for the pass count alone)
  const int64_t base = (sel_steps_bytes(dtype, 1) + 255) / 256 * 256;
  if (hist_offset_bytes) *hist_offset_bytes = base + (pass ? (int64_t)kBins15 * (int64_t)sizeof(uint32_t) : 0);
  if (hist_words) *hist_words = pass ? ((int64_t)1 << shift) : kBins15;
  return passes;
}

extern "C" int bvq_kthw_begin(int abs_key, int dtype, void* workspace, int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = kthw_check("bvq_kthw_begin", dtype, workspace, workspace_bytes);
  if (rc) return rc;
  const WideWs w = wide_ws(dtype, workspace);
  const int shift = wide_shift(abs_key, dtype);
  const int32_t words = kBins15 + (shift ? (int32_t)((int64_t)1 << shift) : 0);
  kth_zero_kernel<<<dim3((unsigned)((words + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(w.hist, words);
  return check_launch("bvq_kthw_begin");
}

extern "C" int bvq_kthw_hist(int abs_key, int dtype, const void* x, int64_t n, int pass, void* workspace,
                             int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = kthw_check("bvq_kthw_hist", dtype, workspace, workspace_bytes);
  if (rc) return rc;
  const int shift = wide_shift(abs_key, dtype);
  if (n < 0 || pass < 0 || pass >= (shift ? 2 : 1)) {
    set_error("bvq_kthw_hist: bad argument");
    return BVQ_ERR_INVALID;
  }
  if (n == 0) return BVQ_OK;  // an empty shard adds nothing
  if (n > kSelMaxCount) {
    set_error("bvq_kthw_hist: %lld elements; the counters are 32-bit (limit 2^32 - 1 over all shards)", (long long)n);
    return BVQ_ERR_UNSUPPORTED;
  }
  if (!x) {
    set_error("bvq_kthw_hist: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const WideWs w = wide_ws(dtype, workspace);
  // the kernels take 16-byte chunks from an aligned address: the elements in front of it are counted one by one
  const int es = dtype_size(dtype);
  int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15) / (uintptr_t)es);
  if (head > n) head = n;
  const char* xa = reinterpret_cast<const char*>(x) + head * es;
  const int64_t na = n - head;
  const bool nt = n * (int64_t)es >= nt_threshold_bytes();
  // few elements: few workgroups (each one flushes its own 32768-bin histogram)
  const int64_t chunks = na / (16 / es);
  unsigned g15 = (unsigned)((chunks + kBlock15 * kSelUnroll - 1) / (kBlock15 * kSelUnroll));
  g15 = g15 < 1 ? 1 : (g15 > 256 ? 256 : g15);
  unsigned glow = (unsigned)((chunks + kBlock * kSelUnroll - 1) / (kBlock * kSelUnroll));
  glow = glow < 1 ? 1 : (glow > 2048 ? 2048 : glow);
#define BVQ_KTHW(T, ABS)                                                                                              \
  do {                                                                                                                \
    const T* xp = reinterpret_cast<const T*>(xa);                                                                     \
    if (pass == 0) {                                                                                                  \
      if (nt)                                                                                                         \
        kth_hist15_kernel<T, ABS, true><<<dim3(g15), dim3(kBlock15), 0, st>>>(xp, na, shift, w.hist, (int32_t)head);  \
      else                                                                                                            \
        kth_hist15_kernel<T, ABS, false><<<dim3(g15), dim3(kBlock15), 0, st>>>(xp, na, shift, w.hist, (int32_t)head); \
    } else if (shift <= kDigitBits) {                                                                                 \
      kth_low_kernel<T, ABS, false, true><<<dim3(glow > 1024 ? 1024 : glow), dim3(kBlock), 0, st>>>(                  \
          xp, na, shift, w.sel, w.low, 1, (int32_t)head);                                                             \
    } else if (nt) {                                                                                                  \
      kth_low_kernel<T, ABS, true, false><<<dim3(glow), dim3(kBlock), 0, st>>>(xp, na, shift, w.sel, w.low, 1,        \
                                                                             (int32_t)head);                         \
    } else {                                                                                                          \
      kth_low_kernel<T, ABS, false, false><<<dim3(glow), dim3(kBlock), 0, st>>>(xp, na, shift, w.sel, w.low, 1,       \
                                                                              (int32_t)head);                        \
    }                                                                                                                 \
  } while (0)
#define BVQ_KTHW_DT(ABS)        \
  do {                          \
    if (dtype == BVQ_F32)       \
      BVQ_KTHW(float, ABS);     \
    else if (dtype == BVQ_BF16) \
      BVQ_KTHW(bf16_t, ABS);    \
    else                        \
      BVQ_KTHW(f16_t, ABS);     \
  } while (0)
  if (abs_key)
    BVQ_KTHW_DT(true);
  else
    BVQ_KTHW_DT(false);
#undef BVQ_KTHW_DT
#undef BVQ_KTHW
  return check_launch("bvq_kthw_hist");
}

extern "C" int bvq_kthw_pick(int abs_key, int dtype, int pass, int rule, int64_t k, double q, void* workspace,
                             int64_t workspace_bytes, bvq_stream_t stream) {
  int rc = kthw_check("bvq_kthw_pick", dtype, workspace, workspace_bytes);
  if (rc) return rc;
  const int shift = wide_shift(abs_key, dtype);
  if (pass < 0 || pass >= (shift ? 2 : 1) || rule < BVQ_KTH_EXPLICIT || rule > BVQ_KTH_LOW ||
      (pass == 0 && rule == BVQ_KTH_EXPLICIT && k < 1) || (pass == 0 && rule != BVQ_KTH_EXPLICIT && !(q >= 0.0 && q <= 100.0))) {
    set_error("bvq_kthw_pick: bad argument");
    return BVQ_ERR_INVALID;
  }
  const WideWs w = wide_ws(dtype, workspace);
  if (pass == 0)
    kth_pick_wide_kernel<<<dim3(1), dim3(kBlock15), 0, (hipStream_t)stream>>>(w.hist, kBins15, 15, 1, k, w.sel, w.krem,
                                                                             rule, q);
  else
    kth_pick_wide_kernel<<<dim3(1), dim3(kBlock15), 0, (hipStream_t)stream>>>(w.low, 1 << shift, shift, 0, 0, w.sel,
                                                                             w.krem);
  return check_launch("bvq_kthw_pick");
}

extern "C" int bvq_kthw_finish(int abs_key, int dtype, void* out, void* workspace, int64_t workspace_bytes,
                               bvq_stream_t stream) {
  int rc = kthw_check("bvq_kthw_finish", dtype, workspace, workspace_bytes);
  if (rc) return rc;
  if (!out) {
    set_error("bvq_kthw_finish: null pointer");
    return BVQ_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const WideWs w = wide_ws(dtype, workspace);
#define BVQ_KTHW_STORE(T)                                             \
  do {                                                                \
    if (abs_key)                                                      \
      kth_store_kernel<T, true><<<dim3(1), dim3(64), 0, st>>>(w.sel, out, 1);  \
    else                                                              \
      kth_store_kernel<T, false><<<dim3(1), dim3(64), 0, st>>>(w.sel, out, 1); \
  } while (0)
  if (dtype == BVQ_F32)
    BVQ_KTHW_STORE(float);
  else if (dtype == BVQ_BF16)
    BVQ_KTHW_STORE(bf16_t);
  else
    BVQ_KTHW_STORE(f16_t);
#undef BVQ_KTHW_STORE
  return check_launch("bvq_kthw_finish");
}

extern "C" int bvq_kth_value(int abs_key, int dtype, const void* x, int64_t outer, int64_t channels,
                             int64_t inner, int64_t k, void* out, void* workspace,
                             int64_t workspace_bytes, bvq_stream_t stream) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16 || outer < 0 || channels < 1 || inner < 0) {
    set_error("bvq_kth_value: bad argument");
    return BVQ_ERR_INVALID;
  }
  const int64_t per_channel = outer * inner;
  if (per_channel > kSelMaxCount) {
    set_error("bvq_kth_value: %lld elements per channel; the digit counters are 32-bit (limit 2^32 - 1)",
              (long long)per_channel);
    return BVQ_ERR_UNSUPPORTED;
  }
  if (k < 1 || k > per_channel) {  // torch.kthvalue: "selected index k out of range"
    set_error("bvq_kth_value: k = %lld out of range [1, %lld]", (long long)k, (long long)per_channel);
    return BVQ_ERR_INVALID;
  }
  if (!x || !out || !workspace) {
    set_error("bvq_kth_value: null pointer");
    return BVQ_ERR_INVALID;
  }
  if (wide_select_applies(channels, per_channel, x))
    return wide_select(abs_key, dtype, x, per_channel, &k, 1, out, workspace, workspace_bytes, stream);
  int rc = bvq_kth_begin(dtype, channels, BVQ_KTH_EXPLICIT, k, 0.0, workspace, workspace_bytes, stream);
  for (int p = 0; !rc && p < passes_for(dtype); ++p) {
    rc = bvq_kth_hist(abs_key, dtype, x, outer, channels, inner, p, workspace, workspace_bytes, stream);
    if (!rc) rc = bvq_kth_pick(dtype, channels, p, BVQ_KTH_EXPLICIT, 0.0, workspace, workspace_bytes, stream);
  }
  if (!rc) rc = bvq_kth_finish(abs_key, dtype, channels, out, workspace, workspace_bytes, stream);
  return rc;
}

extern "C" int bvq_kth_pair(int abs_key, int dtype, const void* x, int64_t outer, int64_t channels, int64_t inner,
                            int64_t k_first, int64_t k_second, void* out, void* workspace, int64_t workspace_bytes,
                            bvq_stream_t stream) {
  if (dtype < BVQ_F32 || dtype > BVQ_F16 || outer < 0 || channels < 1 || inner < 0) {
    set_error("bvq_kth_pair: bad argument");
    return BVQ_ERR_INVALID;
  }
  const int64_t per_channel = outer * inner;
  if (per_channel > kSelMaxCount) {
    set_error("bvq_kth_pair: %lld elements per channel; the digit counters are 32-bit (limit 2^32 - 1)",
              (long long)per_channel);
    return BVQ_ERR_UNSUPPORTED;
  }
  if (k_first < 1 || k_first > per_channel || k_second < 1 || k_second > per_channel) {
    set_error("bvq_kth_pair: rank out of range [1, %lld]", (long long)per_channel);
    return BVQ_ERR_INVALID;
  }
  if (!x || !out || !workspace) {
    set_error("bvq_kth_pair: null pointer");
    return BVQ_ERR_INVALID;
  }
  if (wide_select_applies(channels, per_channel, x)) {
    const int64_t ks[2] = {k_first, k_second};
    return wide_select(abs_key, dtype, x, per_channel, ks, 2, out, workspace, workspace_bytes, stream);
  }
  int rc = bvq_kth_value(abs_key, dtype, x, outer, channels, inner, k_first, out, workspace, workspace_bytes, stream);
  if (!rc)
    rc = bvq_kth_value(abs_key, dtype, x, outer, channels, inner, k_second,
                       reinterpret_cast<char*>(out) + channels * (int64_t)dtype_size(dtype), workspace, workspace_bytes,
                       stream);
  return rc;
}
