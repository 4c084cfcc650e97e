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
constexpr int kBlock = 256;
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

template <typename T, int N>
__device__ __forceinline__ vec_t<T, N> load_vec(const T* p) {
  return *reinterpret_cast<const vec_t<T, N>*>(p);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const vec_t<T, N>& v) {
  *reinterpret_cast<vec_t<T, N>*>(p) = v;
}

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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- tiling ------------------------------------------------------------------------------------
struct Tiling {
  int64_t rows;       // number of rows ((outer, channel) slices; 1 for per-tensor)
  int64_t row_len;    // elements per row
  int64_t piece_len;  // elements per piece (multiple of the vector width)
  int64_t ppr;        // pieces per row
  int64_t units;      // rows * ppr
  int32_t channels;   // row r belongs to channel r % channels
};

// elements one wave handles per unit, in 16-byte chunks per lane.  8 chunks = 8 KiB (2-byte types).
int default_piece_chunks();

// Largest power-of-two vector width (in elements, <= max_vec) usable for rows of row_len elements:
// every row start of every buffer ptrs[i] (element size elsizes[i]) must stay aligned to
// min(16, vec * elsize) bytes.
int pick_vec(int max_vec, int64_t rows, int64_t row_len, const void* const* ptrs, const int* elsizes,
             int nptr);

Tiling make_tiling(int64_t rows, int64_t row_len, int32_t channels, int vec);

static inline unsigned grid_for_units(int64_t units) {
  return (unsigned)((units + kWavesPerBlock - 1) / kWavesPerBlock);
}

}  // namespace bvq
