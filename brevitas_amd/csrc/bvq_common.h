// bvq_common.h -- shared device/host helpers of libbvq (gfx950 only).
//
// Execution model used by every streaming kernel in this library
// ---------------------------------------------------------------
// A tensor is [rows, row_len] where a row is one (outer, channel) slice, contiguous in HBM
// (per-tensor quantizers: one row holding every element).  Rows are cut into PIECES of at most
// piece_len elements; one (row, piece) pair is a UNIT and is processed by exactly ONE 64-lane wave:
//   - the unit's channel, hence its scale / zero-point, is wave-uniform: it is loaded once into
//     scalar registers, so per-channel quantization costs no per-lane gather and no per-lane
//     integer division;
//   - the wave streams its unit with 16-byte-per-lane loads (1 KiB per wave instruction),
//     UNROLL of them in flight before any arithmetic;
//   - reductions (abs-max, scale gradient) finish with a cross-lane DPP/shuffle reduce and ONE
//     store of a per-unit partial; a tiny second kernel combines partials in a fixed order, so
//     results are bit-reproducible and no atomics touch HBM.
// 256-thread workgroups carry 4 independent units (no LDS, no barrier).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bvq.h"

namespace bvq {

constexpr int kWave = 64;
#ifndef BVQ_BLOCK
#define BVQ_BLOCK 256
#endif
constexpr int kBlock = BVQ_BLOCK;
constexpr int kWavesPerBlock = kBlock / kWave;

// ---- error reporting ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- element types -----------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef _Float16 f16_t;

template <typename T>
struct elem;
template <>
struct elem<float> {
  static constexpr int id = BVQ_F32;
  static constexpr int vec = 4;  // elements per 16 B
};
template <>
struct elem<bf16_t> {
  static constexpr int id = BVQ_BF16;
  static constexpr int vec = 8;
};
template <>
struct elem<f16_t> {
  static constexpr int id = BVQ_F16;
  static constexpr int vec = 8;
};

static inline int dtype_size(int dt) { return dt == BVQ_F32 ? 4 : 2; }

// value -> float (exact widening)
template <typename T>
__device__ __forceinline__ float to_f(T v) {
  return (float)v;
}
// float -> T, round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32 on gfx950; NaN stays NaN)
template <typename T>
__device__ __forceinline__ T from_f(float v) {
  return (T)v;
}
// round a float32 intermediate to the compute dtype T and widen it again: the rounding point the
// reference's op chain has after EVERY torch op on a T tensor.
template <typename T>
__device__ __forceinline__ float rnd(float v) {
  return (float)(T)v;
}
template <>
__device__ __forceinline__ float rnd<float>(float v) {
  return v;
}
// float16: hipcc folds  (half)(a * b)  of half-extended operands into v_fma_mixlo_f16 a, b, +0, and
// the +0 addend turns a -0 product into +0.  The empty asm hides the producer of v from that
// pattern match (no instruction is emitted), keeping IEEE signed zeros.
template <>
__device__ __forceinline__ f16_t from_f<f16_t>(float v) {
  asm volatile("" : "+v"(v));
  return (f16_t)v;
}
template <>
__device__ __forceinline__ float rnd<f16_t>(float v) {
  asm volatile("" : "+v"(v));
  return (float)(f16_t)v;
}

// N-element vector of T that is loaded / stored with a single instruction
template <typename T, int N>
struct alignas(sizeof(T) * N > 16 ? 16 : sizeof(T) * N) vec_t {
  T v[N];
};

// NT = non-temporal cache policy (the `nt` bit of global_load/store): for tensors far larger than the
// 256 MiB Infinity Cache every byte is touched once per kernel, and streaming loads AND stores
// measured -9 % on the headline step (profiles/r01_variants_*.txt).  Small tensors (weights) keep the
// default policy so that their consumer finds them in L2 / Infinity Cache.
template <typename T, int N, bool NT = false>
__device__ __forceinline__ vec_t<T, N> load_vec(const T* p) {
  if constexpr (NT && sizeof(T) * N == 16) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return __builtin_bit_cast(vec_t<T, N>, r);
  } else if constexpr (NT && sizeof(T) * N == 32) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct pair_t {
      u32x4 a, b;
    } r;
    r.a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    r.b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    return __builtin_bit_cast(vec_t<T, N>, r);
  } else {
    return *reinterpret_cast<const vec_t<T, N>*>(p);
  }
}
template <typename T, int N, bool NT = false>
__device__ __forceinline__ void store_vec(T* p, const vec_t<T, N>& v) {
  if constexpr (NT && sizeof(T) * N == 16) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p));
  } else if constexpr (NT && sizeof(T) * N == 32) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct pair_t {
      u32x4 a, b;
    };
    const pair_t r = __builtin_bit_cast(pair_t, v);
    __builtin_nontemporal_store(r.a, reinterpret_cast<u32x4*>(p));
    __builtin_nontemporal_store(r.b, reinterpret_cast<u32x4*>(p) + 1);
  } else {
    *reinterpret_cast<vec_t<T, N>*>(p) = v;
  }
}

// ---- buffer addressing (raw buffer loads / stores with hardware bounds checking) -------------------------
// A unit's memory seen through a 128-bit buffer descriptor (base = the unit's first element, num_records = its
// byte extent): a lane that has nothing to load passes kBufSkip as its byte offset, the load returns zeros
// WITHOUT a memory access and the store is dropped.  That keeps a software-pipelined loop free of branches
// and execution-mask regions -- with a predicated global_load inside, the compiler's wait-count insertion
// falls back to vmcnt(0) at every join and the pipeline drains on every step (tools/hot_loop_isa.py shows
// the waits).  Offsets are 32 bits: the host keeps every unit's extent below 2^31 bytes (cap_unit_extent).
constexpr uint32_t kBufSkip = 0x80000000u;
constexpr int64_t kMaxUnitBytes = 0x7fffffff;
#ifdef __HIPCC__
typedef __amdgpu_buffer_rsrc_t buf_t;
template <typename T>
__device__ __forceinline__ buf_t make_buf(const T* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, bytes, 0x00020000);
}
template <typename T, int N, bool NT = false>
__device__ __forceinline__ vec_t<T, N> buf_load(buf_t b, uint32_t byte_off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  constexpr int kAux = NT ? 2 : 0;  // the `nt` bit
  if constexpr (sizeof(T) * N == 16) {
    return __builtin_bit_cast(vec_t<T, N>, __builtin_amdgcn_raw_buffer_load_b128(b, byte_off, 0, kAux));
  } else if constexpr (sizeof(T) * N == 32) {
    struct pair_t {
      u32x4 a, b;
    } r;
    r.a = __builtin_amdgcn_raw_buffer_load_b128(b, byte_off, 0, kAux);
    r.b = __builtin_amdgcn_raw_buffer_load_b128(b, byte_off + 16u, 0, kAux);
    return __builtin_bit_cast(vec_t<T, N>, r);
  } else if constexpr (sizeof(T) * N == 8) {
    return __builtin_bit_cast(vec_t<T, N>, __builtin_amdgcn_raw_buffer_load_b64(b, byte_off, 0, kAux));
  } else if constexpr (sizeof(T) * N == 4) {
    return __builtin_bit_cast(vec_t<T, N>, __builtin_amdgcn_raw_buffer_load_b32(b, byte_off, 0, kAux));
  } else {
    static_assert(sizeof(T) * N == 2, "buf_load: 2, 4, 8, 16 or 32 bytes");
    return __builtin_bit_cast(vec_t<T, N>, __builtin_amdgcn_raw_buffer_load_b16(b, byte_off, 0, kAux));
  }
}
// SC1: agent-scope write-through (the `sc1` bit, aux bit 4 on gfx940+): the bytes are in the memory every XCD reads
// once the wave's vmcnt has counted the store down, and the line does not stay in this XCD's L2
template <typename T, int N, bool NT = false, bool SC1 = false>
__device__ __forceinline__ void buf_store(buf_t b, uint32_t byte_off, const vec_t<T, N>& v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  constexpr int kAux = (NT ? 2 : 0) | (SC1 ? 16 : 0);
  if constexpr (sizeof(T) * N == 16) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, byte_off, 0, kAux);
  } else if constexpr (sizeof(T) * N == 32) {
    struct pair_t {
      u32x4 a, b;
    };
    const pair_t r = __builtin_bit_cast(pair_t, v);
    __builtin_amdgcn_raw_buffer_store_b128(r.a, b, byte_off, 0, kAux);
    __builtin_amdgcn_raw_buffer_store_b128(r.b, b, byte_off + 16u, 0, kAux);
  } else if constexpr (sizeof(T) * N == 8) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), b, byte_off, 0, kAux);
  } else if constexpr (sizeof(T) * N == 4) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), b, byte_off, 0, kAux);
  } else {
    static_assert(sizeof(T) * N == 2, "buf_store: 2, 4, 8, 16 or 32 bytes");
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, v), b, byte_off, 0, kAux);
  }
}
#endif

// bytes a call must stream before its kernels switch to the non-temporal policy
int64_t nt_threshold_bytes();
// host-side float -> dtype -> float rounding (python scalars that torch converts to the tensor dtype)
float round_host(float f, int dt);
int env_flag(const char* name, int dflt);

// read element 0 / element c of a scale-like buffer of runtime dtype as float (wave-uniform use)
__device__ __forceinline__ float load_scalar_as_f(const void* p, int dt, int64_t idx) {
  if (dt == BVQ_F32) return reinterpret_cast<const float*>(p)[idx];
  if (dt == BVQ_BF16) return (float)reinterpret_cast<const bf16_t*>(p)[idx];
  return (float)reinterpret_cast<const f16_t*>(p)[idx];
}

// ---- wave reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, kWave));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, off, kWave);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v |= (uint32_t)__shfl_xor((int)v, off, kWave);
  return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, kWave);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, kWave);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- tiling ------------------------------------------------------------------------------------
// x is [outer, channels, row_len] (per-tensor: outer = channels = 1, row_len = everything).
// A UNIT -- the work of one wave -- is  `rpu` consecutive outer indices of ONE channel  x  one
// piece of the row:
//   long rows  (per-tensor, big weights): rpu = 1, the row is cut into ppr pieces of piece_len;
//   short rows (NCHW activations: H*W elements): ppr = 1 and rpu rows of the same channel are
//     walked back to back, chosen so that rpu * (chunks per row) fills the 64 lanes of every load
//     instruction (a 56x56 bf16 row is 392 chunks = 6.125 wave-loads; 8 rows are exactly 49).
// unit index = (outer_block * channels + channel) * ppr + piece: neighbouring waves touch
// neighbouring memory.
struct Tiling {
  int64_t outer;      // outer extent (1 for per-tensor)
  int64_t row_len;    // elements per row
  int64_t piece_len;  // elements per piece (multiple of the vector width unless it is the row)
  int64_t ppr;        // pieces per row
  int64_t nob;        // outer blocks = ceil(outer / rpu)
  int64_t units;      // nob * channels * ppr
  int32_t channels;
  int32_t rpu;        // rows (outer indices) per unit
  int32_t reverse;    // walk the units from the tensor's end to its start (cache-reuse experiments; 0 by default)
};

// elements one wave handles per piece, in 16-byte chunks per lane.  8 chunks = 8 KiB (2-byte types).
int default_piece_chunks();

// Largest power-of-two vector width (in elements, <= max_vec) usable for rows of row_len elements:
// every row start of every buffer ptrs[i] (element size elsizes[i]) must stay aligned to
// min(16, vec * elsize) bytes.
int pick_vec(int max_vec, int64_t rows, int64_t row_len, const void* const* ptrs, const int* elsizes,
             int nptr, bool ragged_ok = false);

Tiling make_tiling(int64_t outer, int32_t channels, int64_t row_len, int vec, int64_t unit_cap = 0,
                   bool few_rows = false);
// fewer rows per unit until every unit's byte extent (first to last element, elsize bytes each) stays below
// kMaxUnitBytes: the kernels that address a unit through a buffer descriptor use 32-bit offsets.  False if a
// single piece is already larger (no tensor that fits the device gets there).
bool cap_unit_extent(Tiling& t, int elsize);

// ---- column-mapped decomposition -------------------------------------------------------------------
// For layouts whose channel axis is last or nearly last (x[outer, channels, inner] with a short `inner`:
// NHWC activations, [tokens, hidden], 7x7 maps) a "row of one channel" is a few bytes, and the row-mapped
// units above degenerate into strided single-element accesses.  Here the tensor is rows of
// L = channels * inner contiguous elements instead; a wave owns a STRIP of columns (one 16-byte chunk
// per lane) for a block of rows and walks down the rows, so every access is a contiguous run of up to
// 1 KiB and a lane's channels never change: their scales, reciprocals, running maxima and partial sums
// live in that lane's registers.  Per-(row block, column) partials are laid out [partial row][L]; a
// column-parallel kernel folds the partial rows into one row of L entries (coalesced), and that row is the
// (nob = 1, channels, ppr = inner) layout the finishing kernels already understand.
struct ColsPlan {
  bool ok;
  int64_t rows;     // outer
  int64_t L;        // channels * inner
  int32_t vec;
  int32_t cps;      // chunks per row
  int32_t lpr;      // lanes per row (min(cps, 64))
  int32_t rpp;      // rows one wave pass covers (64 / lpr)
  int32_t strips;   // column strips of 64 chunks
  int32_t rb;       // rows per row block (a multiple of rpp)
  int64_t nrb;      // row blocks
  int64_t prows;    // partial rows = nrb * rpp
  int64_t units;    // nrb * strips
};
// no_partials: the caller's kernel writes no per-unit partial rows (more, shorter units pay then)
// columns of a 16-bit type a lane of the column-mapped BACKWARD holds (8 = one 16-byte load per row, 4 = an 8-byte
// load: half the per-column state -- scale, reciprocal, partial sum, running maximum -- in registers)
#ifndef BVQ_COLS_TEAM_VEC16
#define BVQ_COLS_TEAM_VEC16 4
#endif
constexpr int kColsTeamVec16 = BVQ_COLS_TEAM_VEC16;
// team: the caller's kernel gives a unit to a whole workgroup (four waves sharing the rows, partials combined on chip)
// vec16: columns of a 16-bit type per lane (0: 8, one 16-byte load per row -- or kColsTeamVec16 for team units)
ColsPlan cols_plan(int dtype, int64_t outer, int64_t channels, int64_t inner, bool no_partials = false,
                   bool team = false, int vec16 = 0);
// columns of a 16-bit type a lane of the column-mapped FORWARD holds
#ifndef BVQ_COLS_FWD_VEC16
#define BVQ_COLS_FWD_VEC16 4
#endif
constexpr int kColsFwdVec16 = BVQ_COLS_FWD_VEC16;

static inline unsigned grid_for_units(int64_t units) {
  return (unsigned)((units + kWavesPerBlock - 1) / kWavesPerBlock);
}

#ifdef __HIPCC__
// One wave's unit (wave-uniform values: scalar registers).
struct Unit {
  int64_t id;          // unit index (slot of its partial results)
  int64_t base;        // element offset of the unit's first element
  int64_t row_stride;  // elements between consecutive rows of the unit (channels * row_len)
  int64_t len;         // elements of each row covered by this unit
  int64_t pos0;        // position of the first element in the reference's per-channel reduction
                       // order: (outer index) * row_len + offset in row  (flat index if per-tensor)
  int32_t nrows;       // rows in this unit
  int32_t channel;
  bool valid;
};

// the unit in dispatch slot `slot` (wave-uniform)
__device__ __forceinline__ Unit locate_unit_slot(const Tiling& t, int64_t slot) {
  Unit u;
  u.valid = slot < t.units;
  u.id = t.reverse ? t.units - 1 - slot : slot;  // the unit's id (slot of its partials) stays its position in memory
  if (!u.valid) {
    u.base = u.row_stride = u.len = u.pos0 = 0;
    u.nrows = u.channel = 0;
    return u;
  }
  // (outer_block, channel, piece) from the unit id.  A wave-uniform 64-bit division is ~100 VALU instructions on
  // gfx9 (no scalar divide): short rows have one piece per row (no division), and ids below 2^31 -- every tensor but
  // the very largest -- divide in 32 bits.  The two divisions were most of the ~190 VALU instructions a unit of the
  // headline backward spent outside its hot loop (profiles/r02/pmc_final_build.md).
  int64_t rc, piece, ob;
#ifndef BVQ_LOCATE64
  if (u.id < ((int64_t)1 << 31) && t.ppr < ((int64_t)1 << 31)) {
    const uint32_t id32 = (uint32_t)u.id;
    uint32_t rc32 = id32, piece32 = 0;
    if (t.ppr != 1) {
      rc32 = id32 / (uint32_t)t.ppr;
      piece32 = id32 - rc32 * (uint32_t)t.ppr;
    }
    const uint32_t ob32 = rc32 / (uint32_t)t.channels;
    rc = rc32;
    piece = piece32;
    ob = ob32;
  } else
#endif
  {
    rc = u.id / t.ppr;
    piece = u.id - rc * t.ppr;
    ob = rc / t.channels;
  }
  u.channel = (int32_t)(rc - ob * t.channels);
  const int64_t o0 = ob * t.rpu;
  const int64_t left = t.outer - o0;
  u.nrows = (int32_t)(left < t.rpu ? left : t.rpu);
  const int64_t off = piece * t.piece_len;
  const int64_t rest = t.row_len - off;
  u.len = rest < t.piece_len ? rest : t.piece_len;
  u.row_stride = (int64_t)t.channels * t.row_len;
  u.base = (o0 * t.channels + u.channel) * t.row_len + off;
  u.pos0 = o0 * t.row_len + off;
  return u;
}

// One wave's unit, decoded from blockIdx / wave id
__device__ __forceinline__ Unit locate_unit(const Tiling& t) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  return locate_unit_slot(t, (int64_t)blockIdx.x * kWavesPerBlock + wave);
}

// Per-lane walk over the VEC-element chunks of a unit: chunk k of the unit (k = lane, lane+64, ...)
// lives in row k / cpr at chunk k % cpr.  Advancing by 64 chunks is "dq rows and dr chunks, plus a
// carry" with the wave-uniform dq = 64 / cpr, dr = 64 % cpr: one division per unit, none and no
// branch per step (a single-row unit is the special case dq = 0, dr = 64).
struct ChunkCursor {
  int32_t row;    // row inside the unit
  int32_t chunk;  // chunk inside the row
  int32_t cpr;    // full chunks per row
  int32_t nrows;
  int32_t dq, dr;
  __device__ __forceinline__ void init(const Unit& u, int vec, int lane) {
    cpr = (int32_t)(u.len / vec);
    nrows = u.nrows;
    const int32_t c = cpr > 0 ? cpr : 1;
    int32_t r0 = 0;  // 0 for every lane when a row has >= 64 chunks
    if (c >= kWave) {  // (wave-uniform; spares two integer divisions, ~50 VALU instructions per unit)
      dq = 0;
      dr = kWave;
    } else {
      dq = kWave / c;
      dr = kWave - dq * c;
      r0 = lane / c;
    }
    chunk = lane - r0 * c;
    // rows shorter than one chunk: nothing to walk (the ragged-end code takes them)
    row = cpr > 0 ? r0 : nrows;
  }
  __device__ __forceinline__ bool valid() const { return row < nrows; }
  __device__ __forceinline__ void next() {
    chunk += dr;
    row += dq;
    const bool carry = chunk >= cpr;
    chunk -= carry ? cpr : 0;
    row += carry ? 1 : 0;
  }
  // element offset from the unit's base / from pos0
  __device__ __forceinline__ int64_t offset(int64_t row_stride, int vec) const {
    return (int64_t)row * row_stride + (int64_t)chunk * vec;
  }
  __device__ __forceinline__ int64_t pos(int64_t row_len, int vec) const {
    return (int64_t)row * row_len + (int64_t)chunk * vec;
  }
  // the offset in 32 bits, for units addressed through a buffer descriptor (extent < 2^31 bytes)
  __device__ __forceinline__ uint32_t offset32(uint32_t row_stride, int vec) const {
    return (uint32_t)row * row_stride + (uint32_t)chunk * (uint32_t)vec;
  }
};

// The same walk with a loop for the row carry: the form the read-only kernels (statistics, select, tie
// scan) keep -- measured on MI355X they stream 5-10 % faster with it (their loads are predicated, and a
// single-row unit, the common case, advances with one add), while the quantizer kernels, which issue
// their loads unconditionally, are faster with the branch-free ChunkCursor above.
struct ChunkWalker {
  int32_t row;    // row inside the unit
  int32_t chunk;  // chunk inside the row
  int32_t cpr;    // full chunks per row
  int32_t nrows;
  __device__ __forceinline__ void init(const Unit& u, int vec, int lane) {
    cpr = (int32_t)(u.len / vec);
    nrows = u.nrows;
    if (nrows == 1 || cpr == 0) {
      row = 0;
      chunk = lane;
    } else {
      row = lane / cpr;
      chunk = lane - row * cpr;
    }
  }
  __device__ __forceinline__ bool valid() const { return nrows == 1 ? chunk < cpr : row < nrows; }
  __device__ __forceinline__ void next() {
    chunk += kWave;
    if (nrows != 1) {
      while (chunk >= cpr && row < nrows) {
        chunk -= cpr;
        ++row;
      }
    }
  }
  // element offset from the unit's base / from pos0
  __device__ __forceinline__ int64_t offset(int64_t row_stride, int vec) const {
    return (int64_t)row * row_stride + (int64_t)chunk * vec;
  }
  __device__ __forceinline__ int64_t pos(int64_t row_len, int vec) const {
    return (int64_t)row * row_len + (int64_t)chunk * vec;
  }
};

#endif

}  // namespace bvq
