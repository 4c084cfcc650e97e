// bvq_elementwise.hip -- forward math of the 12 straight-through ops (seam 1) and the plain
// tensor_clamp with its mask backward.
//
// Reference: B/ops/autograd_ste_ops.py:37-382 (Python backend) and B/csrc/autograd_ste_ops.cpp:14-194
// (C++ backend): each forward is 1-4 ATen elementwise passes, each backward is the identity (or
// binary_sign(x)*g).  Here each forward is one read + one write with 16-byte accesses.  On the hot
// path these ops only ever see scale-shaped tensors (1..C elements); they exist so that the whole
// `autograd_ste_ops` namespace is served by the same library.
#include <math.h>
#include <string.h>

#include "bvq_quant_math.h"

namespace bvq {

template <typename T, int OP>
__device__ __forceinline__ float unary_op(float v) {
  if constexpr (OP == BVQ_OP_ROUND) return round_op<T, BVQ_ROUND>(v);
  if constexpr (OP == BVQ_OP_FLOOR) return round_op<T, BVQ_FLOOR>(v);
  if constexpr (OP == BVQ_OP_CEIL) return round_op<T, BVQ_CEIL>(v);
  if constexpr (OP == BVQ_OP_ROUND_TO_ZERO) return round_op<T, BVQ_ROUND_TO_ZERO>(v);
  if constexpr (OP == BVQ_OP_DPU_ROUND) return round_op<T, BVQ_DPU_ROUND>(v);
  if constexpr (OP == BVQ_OP_BINARY_SIGN) {
    // positive_mask.to(dtype) - negative_mask.to(dtype)   (B/function/ops.py:31-33)
    return (float)(v >= 0.f) - (float)(v < 0.f);
  }
  if constexpr (OP == BVQ_OP_TERNARY_SIGN) return (float)(0.f < v) - (float)(v < 0.f);
  if constexpr (OP == BVQ_OP_ABS) return __builtin_fabsf(v);
  return v;
}

// generic grid-stride elementwise driver: F maps (index, up to three loaded values) -> value
template <typename T, int VEC, int NIN, typename F>
__global__ __launch_bounds__(kBlock) void map_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                     const T* __restrict__ c, T* __restrict__ y,
                                                     int64_t n, F f) {
  const int64_t nvec = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    vec_t<T, VEC> av = load_vec<T, VEC>(a + i * VEC), bv, cv, yv;
    if (NIN > 1) bv = load_vec<T, VEC>(b + i * VEC);
    if (NIN > 2) cv = load_vec<T, VEC>(c + i * VEC);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      yv.v[k] = from_f<T>(f(to_f<T>(av.v[k]), NIN > 1 ? to_f<T>(bv.v[k]) : 0.f,
                           NIN > 2 ? to_f<T>(cv.v[k]) : 0.f));
    }
    store_vec<T, VEC>(y + i * VEC, yv);
  }
  // ragged end
  const int64_t t = nvec * VEC + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t < n) {
    y[t] = from_f<T>(f(to_f<T>(a[t]), NIN > 1 ? to_f<T>(b[t]) : 0.f, NIN > 2 ? to_f<T>(c[t]) : 0.f));
  }
}

static unsigned map_grid(int64_t n, int vec) {
  int64_t nb = (n / vec + kBlock - 1) / kBlock;
  if (nb < 1) nb = 1;
  if (nb > 4096) nb = 4096;
  return (unsigned)nb;
}

template <typename T, int NIN, typename F>
static void launch_map(const void* a, const void* b, const void* c, void* y, int64_t n, F f,
                       hipStream_t st) {
  constexpr int V = elem<T>::vec;
  const void* ptrs[4] = {a, b, c, y};
  const int els[4] = {(int)sizeof(T), (int)sizeof(T), (int)sizeof(T), (int)sizeof(T)};
  const int vec = pick_vec(V, 1, n, ptrs, els, 4);
  const T* ap = reinterpret_cast<const T*>(a);
  const T* bp = reinterpret_cast<const T*>(b);
  const T* cp = reinterpret_cast<const T*>(c);
  T* yp = reinterpret_cast<T*>(y);
  if (vec == V)
    map_kernel<T, V, NIN, F><<<dim3(map_grid(n, V)), dim3(kBlock), 0, st>>>(ap, bp, cp, yp, n, f);
  else
    map_kernel<T, 1, NIN, F><<<dim3(map_grid(n, 1)), dim3(kBlock), 0, st>>>(ap, bp, cp, yp, n, f);
}

template <typename T, int OP>
struct UnaryF {
  __device__ float operator()(float v, float, float) const { return unary_op<T, OP>(v); }
};

template <typename T>
static int dispatch_unary(int op, const void* x, void* y, int64_t n, hipStream_t st) {
  switch (op) {
#define BVQ_CASE(OP)                                                  \
  case OP:                                                            \
    launch_map<T, 1>(x, nullptr, nullptr, y, n, UnaryF<T, OP>(), st); \
    return BVQ_OK;
    BVQ_CASE(BVQ_OP_ROUND)
    BVQ_CASE(BVQ_OP_FLOOR)
    BVQ_CASE(BVQ_OP_CEIL)
    BVQ_CASE(BVQ_OP_ROUND_TO_ZERO)
    BVQ_CASE(BVQ_OP_DPU_ROUND)
    BVQ_CASE(BVQ_OP_BINARY_SIGN)
    BVQ_CASE(BVQ_OP_TERNARY_SIGN)
    BVQ_CASE(BVQ_OP_ABS)
#undef BVQ_CASE
    default:
      set_error("bvq_unary: bad op %d", op);
      return BVQ_ERR_INVALID;
  }
}

// torch.clamp / clamp_min with scalar bounds already rounded to T; NaN propagates
struct ScalarClampF {
  float lo, hi;
  int use_lo, use_hi;
  __device__ float operator()(float v, float, float) const {
    if (use_lo) v = v < lo ? lo : v;
    if (use_hi) v = v > hi ? hi : v;
    return v;
  }
};

// tensor_clamp with one-element bounds read on the device (no host sync)
template <typename T>
struct TensorClampScalarF {
  const T* lo;
  const T* hi;
  __device__ float operator()(float v, float, float) const {
    return clamp_where(v, to_f<T>(*lo), to_f<T>(*hi));
  }
};
struct TensorClampFullF {
  __device__ float operator()(float v, float lo, float hi) const { return clamp_where(v, lo, hi); }
};

// backward of where(x>hi,hi,x) / where(out<lo,lo,out) w.r.t. x; inputs (g, x[, lo, hi])
template <typename T>
struct ClampBwdScalarF {
  const T* lo;
  const T* hi;
  __device__ float operator()(float g, float x, float) const {
    const bool pass = !(x > to_f<T>(*hi)) && !(x < to_f<T>(*lo));
    return pass ? g : 0.f;
  }
};

template <typename T>
__global__ __launch_bounds__(kBlock) void clamp_bwd_full_kernel(const T* __restrict__ g,
                                                                const T* __restrict__ x,
                                                                const T* __restrict__ lo,
                                                                const T* __restrict__ hi,
                                                                T* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float xv = to_f<T>(x[i]);
    const bool pass = !(xv > to_f<T>(hi[i])) && !(xv < to_f<T>(lo[i]));
    dx[i] = pass ? g[i] : from_f<T>(0.f);
  }
}

// dx = binary_sign(x) * g      inputs (g, x)
struct SignMulF {
  __device__ float operator()(float g, float x, float) const {
    const float s = (float)(x >= 0.f) - (float)(x < 0.f);
    return s * g;
  }
};

// host-side rounding of a python-scalar bound to the tensor's dtype (torch: Scalar::to<scalar_t>())
static float host_round_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return f;  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}
static float host_round_f16(float f) {
  // round to 11 significant bits inside the half range; overflow -> inf, tiny -> half subnormal grid
  if (f != f || f == 0.f) return f;
  const float a = fabsf(f);
  if (a >= 65520.f) return copysignf(INFINITY, f);
  int e;
  frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
  int qexp = e - 11;
  if (qexp < -24) qexp = -24;  // subnormal spacing 2^-24
  const float q = ldexpf(1.f, qexp);
  return copysignf(nearbyintf(a / q) * q, f);
}
template <typename T>
static float round_bound(double v) {
  const float f = (float)v;
  if (sizeof(T) == 4) return f;
  return elem<T>::id == BVQ_BF16 ? host_round_bf16(f) : host_round_f16(f);
}

}  // namespace bvq

using namespace bvq;

#define BVQ_DISPATCH_DTYPE(dt, ...)     \
  do {                                  \
    if ((dt) == BVQ_F32) {              \
      typedef float T;                  \
      __VA_ARGS__;                      \
    } else if ((dt) == BVQ_BF16) {      \
      typedef bf16_t T;                 \
      __VA_ARGS__;                      \
    } else if ((dt) == BVQ_F16) {       \
      typedef f16_t T;                  \
      __VA_ARGS__;                      \
    } else {                            \
      set_error("bad dtype %d", (dt));  \
      return BVQ_ERR_INVALID;           \
    }                                   \
  } while (0)

static int check_n(const char* fn, int64_t n) {
  if (n < 0) {
    set_error("%s: negative size", fn);
    return BVQ_ERR_INVALID;
  }
  return BVQ_OK;
}

extern "C" int bvq_unary(int op, int dtype, const void* x, void* y, int64_t n, bvq_stream_t stream) {
  int rc = check_n("bvq_unary", n);
  if (rc || n == 0) return rc;
  if (!x || !y) {
    set_error("bvq_unary: null pointer");
    return BVQ_ERR_INVALID;
  }
  BVQ_DISPATCH_DTYPE(dtype, rc = dispatch_unary<T>(op, x, y, n, (hipStream_t)stream));
  return rc ? rc : check_launch("bvq_unary");
}

extern "C" int bvq_scalar_clamp(int dtype, const void* x, void* y, int64_t n, double lo, int use_lo,
                                double hi, int use_hi, bvq_stream_t stream) {
  int rc = check_n("bvq_scalar_clamp", n);
  if (rc || n == 0) return rc;
  if (!x || !y) {
    set_error("bvq_scalar_clamp: null pointer");
    return BVQ_ERR_INVALID;
  }
  BVQ_DISPATCH_DTYPE(dtype, {
    ScalarClampF f;
    f.lo = round_bound<T>(lo);
    f.hi = round_bound<T>(hi);
    f.use_lo = use_lo;
    f.use_hi = use_hi;
    launch_map<T, 1>(x, nullptr, nullptr, y, n, f, (hipStream_t)stream);
  });
  return check_launch("bvq_scalar_clamp");
}

extern "C" int bvq_tensor_clamp(int dtype, const void* x, const void* lo, const void* hi,
                                int bounds_full, void* y, int64_t n, bvq_stream_t stream) {
  int rc = check_n("bvq_tensor_clamp", n);
  if (rc || n == 0) return rc;
  if (!x || !lo || !hi || !y) {
    set_error("bvq_tensor_clamp: null pointer");
    return BVQ_ERR_INVALID;
  }
  BVQ_DISPATCH_DTYPE(dtype, {
    if (bounds_full) {
      launch_map<T, 3>(x, lo, hi, y, n, TensorClampFullF(), (hipStream_t)stream);
    } else {
      TensorClampScalarF<T> f;
      f.lo = reinterpret_cast<const T*>(lo);
      f.hi = reinterpret_cast<const T*>(hi);
      launch_map<T, 1>(x, nullptr, nullptr, y, n, f, (hipStream_t)stream);
    }
  });
  return check_launch("bvq_tensor_clamp");
}

extern "C" int bvq_tensor_clamp_bwd(int dtype, const void* g, const void* x, const void* lo,
                                    const void* hi, int bounds_full, void* dx, int64_t n,
                                    bvq_stream_t stream) {
  int rc = check_n("bvq_tensor_clamp_bwd", n);
  if (rc || n == 0) return rc;
  if (!g || !x || !lo || !hi || !dx) {
    set_error("bvq_tensor_clamp_bwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  BVQ_DISPATCH_DTYPE(dtype, {
    if (bounds_full) {
      int64_t nb = (n + kBlock - 1) / kBlock;
      if (nb > 4096) nb = 4096;
      clamp_bwd_full_kernel<T><<<dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream>>>(
          reinterpret_cast<const T*>(g), reinterpret_cast<const T*>(x), reinterpret_cast<const T*>(lo),
          reinterpret_cast<const T*>(hi), reinterpret_cast<T*>(dx), n);
    } else {
      ClampBwdScalarF<T> f;
      f.lo = reinterpret_cast<const T*>(lo);
      f.hi = reinterpret_cast<const T*>(hi);
      launch_map<T, 2>(g, x, nullptr, dx, n, f, (hipStream_t)stream);
    }
  });
  return check_launch("bvq_tensor_clamp_bwd");
}

extern "C" int bvq_abs_binary_sign_grad_bwd(int dtype, const void* g, const void* x, void* dx, int64_t n,
                                            bvq_stream_t stream) {
  int rc = check_n("bvq_abs_binary_sign_grad_bwd", n);
  if (rc || n == 0) return rc;
  if (!g || !x || !dx) {
    set_error("bvq_abs_binary_sign_grad_bwd: null pointer");
    return BVQ_ERR_INVALID;
  }
  BVQ_DISPATCH_DTYPE(dtype, launch_map<T, 2>(g, x, nullptr, dx, n, SignMulF(), (hipStream_t)stream));
  return check_launch("bvq_abs_binary_sign_grad_bwd");
}
