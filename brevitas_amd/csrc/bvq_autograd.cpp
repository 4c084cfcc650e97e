// bvq_autograd.cpp -- host glue, not a kernel: the stats-scaled weight quantizer's autograd node in C++.
//
// A weight-sized quantizer step is three launches (~30 us of GPU time); through the Python
// torch.autograd.Function + ctypes route the host spends ~85 us on it (profiles/r02_host_cost.txt), most of it in
// the Function machinery and the wrappers' allocations.  This node makes ONE call into libbvq.so each way:
//   forward   bvq_stats_fakequant_fwd   (statistic + scale + quantize, one launch)
//   backward  bvq_fakequant_bwd_stats   (dx with the statistic's gradient deposited, two launches)
// -- the same C-ABI entries the Python route calls (include/bvq.h), resolved with dlsym from the library the
// package has already loaded; no HIP header is needed here.  Anything this node does not cover (a gradient
// arriving through `scale`, an unaligned or non-contiguous gradient) goes back to the Python implementation
// through the fallback registered at start-up.  Reference boundary: proxy.tensor_quant(x) of a weight proxy,
// B/proxy/parameter_quant.py:83-89 -> RescalingIntQuant.forward, B/core/quant/int.py:155-163.
#include <dlfcn.h>
#include <torch/extension.h>

#include <cstdint>
#include <stdexcept>
#include <string>

namespace {

// bvq_quant_desc of include/bvq.h (layout checked against the ctypes mirror by tests/test_cabi_symbols.py)
struct QuantDesc {
  int64_t outer, channels, inner;
  int32_t x_dtype, ct_dtype, scale_dtype, zp_dtype, scale_per_channel, zp_per_channel;
  float qmin, qmax;
  int32_t round_mode, scalar_mode, clamp_ste, out_kind, pre_op, codes_dtype;
};

using fwd_ws_fn = int64_t (*)(const QuantDesc*, const void*, const void*);
using fwd_fn = int (*)(const QuantDesc*, const void*, double, int, double, void*, void*, void*, void*, int64_t, void*);
using bwd_ws_fn = int64_t (*)(const QuantDesc*);
using bwd_fn = int (*)(const QuantDesc*, const void*, const void*, const void*, const void*, const void*, void*, float*,
                       int, double, int, void*, int64_t, void*);
using err_fn = const char* (*)();

fwd_ws_fn p_fwd_ws = nullptr;
fwd_fn p_fwd = nullptr;
bwd_ws_fn p_bwd_ws = nullptr;
bwd_fn p_bwd = nullptr;
err_fn p_err = nullptr;
py::object* g_fallback = nullptr;  // python: (x, scale, zp, stat, int_threshold, desc, qrange, shape, gy, gscale) -> dx
                                   // (leaked on purpose: destroying it after the interpreter has gone would crash)

void check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (p_err ? p_err() : "error"));
}

at::ScalarType dtype_of(int code) {
  return code == 0 ? at::kFloat : (code == 1 ? at::kBFloat16 : at::kHalf);  // BVQ_F32 / BVQ_BF16 / BVQ_F16
}

struct Params {
  QuantDesc d;
  double min_val, thr_fwd, thr_bwd, thr_raw;
  int has_min, scale_dtype;
  int64_t stream;
  std::vector<int64_t> shape;  // scaling shape
};

class StatsFakeQuant : public torch::autograd::Function<StatsFakeQuant> {
 public:
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x,
                                                const at::Tensor& zp, const at::Tensor& int_threshold, int64_t wsb,
                                                const Params& p) {
    ctx->set_materialize_grads(false);
    const int64_t ch = p.d.channels;
    at::Tensor y = at::empty_like(x);
    at::Tensor stat = at::empty({ch}, x.options());
    at::Tensor scale = at::empty({ch}, x.options().dtype(dtype_of(p.scale_dtype)));
    at::Tensor ws = at::empty({wsb}, x.options().dtype(at::kByte));
    check(p_fwd(&p.d, x.data_ptr(), p.min_val, p.has_min, p.thr_fwd, stat.data_ptr(), scale.data_ptr(), y.data_ptr(),
                ws.data_ptr(), wsb, reinterpret_cast<void*>(p.stream)),
          "bvq_stats_fakequant_fwd");
    ctx->save_for_backward({x, scale, zp, stat, int_threshold});
    ctx->saved_data["desc"] = std::vector<int64_t>{p.d.outer, p.d.channels, p.d.inner, p.d.x_dtype, p.d.ct_dtype,
                                                   p.d.scale_dtype, p.d.zp_dtype, p.d.scale_per_channel,
                                                   p.d.zp_per_channel, p.d.round_mode, p.d.scalar_mode, p.d.clamp_ste,
                                                   p.d.out_kind, p.d.pre_op, p.d.codes_dtype, p.scale_dtype, p.stream};
    ctx->saved_data["qrange"] = std::vector<double>{p.d.qmin, p.d.qmax, p.thr_bwd, p.thr_raw};
    ctx->saved_data["shape"] = p.shape;
    at::Tensor scale_out = scale.view(p.shape), stat_out = stat.view(p.shape);
    ctx->mark_non_differentiable({stat_out});
    return {y, scale_out, stat_out};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &scale = saved[1], &zp = saved[2], &stat = saved[3], &int_threshold = saved[4];
    const auto dv = ctx->saved_data["desc"].toIntVector();
    const auto qr = ctx->saved_data["qrange"].toDoubleVector();
    const at::Tensor& gy = grads[0];
    const at::Tensor& gscale = grads[1];
    QuantDesc d{dv[0], dv[1], dv[2], (int32_t)dv[3], (int32_t)dv[4], (int32_t)dv[5], (int32_t)dv[6], (int32_t)dv[7],
                (int32_t)dv[8], (float)qr[0], (float)qr[1], (int32_t)dv[9], (int32_t)dv[10], (int32_t)dv[11],
                (int32_t)dv[12], (int32_t)dv[13], (int32_t)dv[14]};
    const bool direct = gy.defined() && !gscale.defined() && gy.is_contiguous() && gy.scalar_type() == x.scalar_type() &&
                        ((reinterpret_cast<uintptr_t>(gy.data_ptr()) | reinterpret_cast<uintptr_t>(x.data_ptr())) & 15) == 0;
    if (!direct) {
      if (!gy.defined() && !gscale.defined()) return {at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
      py::gil_scoped_acquire gil;
      py::object dx = (*g_fallback)(x, scale, zp, stat, int_threshold, py::cast(dv), py::cast(qr),
                                 py::cast(ctx->saved_data["shape"].toIntVector()),
                                 gy.defined() ? py::cast(gy) : py::none(), gscale.defined() ? py::cast(gscale) : py::none());
      return {dx.cast<at::Tensor>(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
    const int64_t wsb = p_bwd_ws(&d);
    if (wsb <= 0) throw std::runtime_error("bvq_fakequant_bwd_stats: layout not covered (checked at forward)");
    at::Tensor dx = at::empty_like(x);
    at::Tensor ds = at::empty({d.channels}, x.options().dtype(at::kFloat));
    at::Tensor ws = at::empty({wsb}, x.options().dtype(at::kByte));
    const int sdt = (int)dv[15];
    check(p_bwd(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(), stat.data_ptr(), dx.data_ptr(),
                ds.data_ptr<float>(), sdt, qr[2], sdt, ws.data_ptr(), wsb, reinterpret_cast<void*>(dv[16])),
          "bvq_fakequant_bwd_stats");
    return {dx, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

}  // namespace

// resolve the C-ABI entries from the library the package loaded (path: brevitas_amd/libbvq.so)
void init(const std::string& lib_path, py::object fallback) {
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) throw std::runtime_error(std::string("dlopen ") + lib_path + ": " + dlerror());
  p_fwd_ws = reinterpret_cast<fwd_ws_fn>(dlsym(h, "bvq_stats_fakequant_fwd_workspace_bytes"));
  p_fwd = reinterpret_cast<fwd_fn>(dlsym(h, "bvq_stats_fakequant_fwd"));
  p_bwd_ws = reinterpret_cast<bwd_ws_fn>(dlsym(h, "bvq_fakequant_bwd_stats_workspace_bytes"));
  p_bwd = reinterpret_cast<bwd_fn>(dlsym(h, "bvq_fakequant_bwd_stats"));
  p_err = reinterpret_cast<err_fn>(dlsym(h, "bvq_last_error"));
  if (!p_fwd_ws || !p_fwd || !p_bwd_ws || !p_bwd || !p_err) throw std::runtime_error("libbvq.so: missing entry points");
  g_fallback = new py::object(std::move(fallback));
}

// -> (y, scale, stat), or None when the one-launch forward / two-launch backward do not cover this layout
// desc: the 17 fields of bvq_quant_desc in order (qmin / qmax as floats)
py::object stats_fakequant(const at::Tensor& x, const at::Tensor& zp, const at::Tensor& int_threshold,
                           const std::vector<int64_t>& di, double qmin, double qmax, double min_val, bool has_min,
                           double thr_fwd, double thr_bwd, double thr_raw, int64_t scale_dtype,
                           std::vector<int64_t> shape, int64_t stream) {
  Params p;
  p.d = QuantDesc{di[0], di[1], di[2], (int32_t)di[3], (int32_t)di[4], (int32_t)di[5], (int32_t)di[6], (int32_t)di[7],
                  (int32_t)di[8], (float)qmin, (float)qmax, (int32_t)di[9], (int32_t)di[10], (int32_t)di[11],
                  (int32_t)di[12], (int32_t)di[13], (int32_t)di[14]};
  p.min_val = min_val;
  p.has_min = has_min ? 1 : 0;
  p.thr_fwd = thr_fwd;
  p.thr_bwd = thr_bwd;
  p.thr_raw = thr_raw;
  p.scale_dtype = (int)scale_dtype;
  p.stream = stream;
  p.shape = std::move(shape);
  if (!x.is_contiguous() || (reinterpret_cast<uintptr_t>(x.data_ptr()) & 15) != 0) return py::none();
  // coverage: both workspace queries are host-side and cheap (y's address only matters for its alignment: x's stands in)
  const int64_t wsb = p_fwd_ws(&p.d, x.data_ptr(), x.data_ptr());
  if (wsb <= 0 || p_bwd_ws(&p.d) <= 0) return py::none();
  auto out = StatsFakeQuant::apply(x, zp, int_threshold, wsb, p);
  return py::make_tuple(out[0], out[1], out[2]);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("init", &init, "resolve libbvq.so and register the python fallback of the backward");
  m.def("stats_fakequant", &stats_fakequant, "AbsMax -> scale -> IntQuant on the fused kernels, autograd node in C++");
}
