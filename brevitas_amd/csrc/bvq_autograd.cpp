// bvq_autograd.cpp -- host glue, not a kernel: the autograd nodes of the stats-scaled quantizers in C++.
//
// A weight-sized quantizer step is a few launches (~30 us of GPU time); through the Python
// torch.autograd.Function + ctypes route the host spends ~85 us on it (profiles/r02_host_cost.txt), most of it in
// the Function machinery and the wrappers' allocations -- and one rank's shard of a strong-scaled activation
// ([32,512,56,56]: 120 us of kernels) took 270 us of host time per step (profiles/r03_strong_scaling.md).  These
// nodes make a handful of calls into libbvq.so each way and nothing else:
//
//   StatsFakeQuant (weights)       forward   bvq_stats_fakequant_fwd      statistic + scale + quantize, one launch
//                                  backward  bvq_fakequant_bwd_stats      dx with the statistic's gradient deposited
//   ActStatsFakeQuant (activations whose channels do not fit one workgroup; RuntimeStatsScaling in training mode)
//                                  forward   bvq_absmax_scale_onepass | bvq_absmax_scale[_running]   statistic (+ scale,
//                                            running average), bvq_fakequant_fwd
//                                  backward  bvq_fakequant_bwd_stats[_onepass] (per-channel) |
//                                            bvq_fakequant_bwd + bvq_stat_tie_apply_dscale (per-tensor)
//     batch-sharded (brevitas_amd.distributed.shard_over_batch; a whole-tensor statistic: bvq_fakequant_bwd, bvq_shard_pack,
//     all-gather of [dscale sum | tie count], bvq_shard_unpack, bvq_stat_tie_apply_dscale over the ties of all shards):
//                                  forward   float32 statistic, all-reduce(MAX), bvq_scale_from_stat_running,
//                                            bvq_fakequant_fwd
//                                  backward  bvq_fakequant_bwd_shard, all-gather, bvq_shard_unpack_deposit
//     -- the two collectives are issued from here, so the step never returns to Python between its launches: through
//     RCCL's C API on the COMPUTE stream when the group has a native communicator (rccl_comm_init below: one
//     ncclAllReduce / ncclAllGather call each, no work object, no hop to another stream and back -- c10d costs ~17-20 us
//     of host time per call and two cross-stream event waits, profiles/r03_strong_scaling.md), else through c10d's C++
//     ProcessGroup (what torch.distributed.all_reduce / all_gather_into_tensor call underneath).
//
// The C-ABI entries are the ones the Python route calls (include/bvq.h, included here: the descriptor layout and every
// prototype come from that header), resolved with dlsym from the library the package has already loaded and checked
// against BVQ_ABI_VERSION; no HIP header is needed.  Anything a node does not cover (a gradient arriving through
// `scale`, an unaligned or non-contiguous gradient) goes back to the Python
// implementation through the fallback registered at start-up.  Reference boundary: proxy.tensor_quant(x),
// B/proxy/parameter_quant.py:83-89 and B/proxy/runtime_quant.py:80-84 -> RescalingIntQuant.forward,
// B/core/quant/int.py:155-163.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the entry points are resolved from the RCCL torch has loaded
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>
#include <torch/extension.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>

#include "bvq.h"

namespace {

// one pointer per C-ABI entry, typed by the header's own prototype
#define BVQ_ENTRIES(X)                       \
  X(bvq_abi_version)                         \
  X(bvq_last_error)                          \
  X(bvq_stats_fakequant_fwd_workspace_bytes) \
  X(bvq_stats_fakequant_fwd)                 \
  X(bvq_fakequant_bwd_stats_workspace_bytes) \
  X(bvq_fakequant_bwd_stats)                 \
  X(bvq_fakequant_bwd_stats_onepass_supported) \
  X(bvq_fakequant_bwd_stats_onepass)         \
  X(bvq_absmax_onepass_supported)            \
  X(bvq_absmax_scale_onepass)                \
  X(bvq_stats_workspace_bytes)               \
  X(bvq_stats_pre)                           \
  X(bvq_absmax_scale)                        \
  X(bvq_absmax_scale_running)                \
  X(bvq_scale_from_stat_running)             \
  X(bvq_fakequant_fwd)                       \
  X(bvq_fakequant_bwd_workspace_bytes)       \
  X(bvq_fakequant_bwd)                       \
  X(bvq_tie_info_bytes)                      \
  X(bvq_stat_tie_apply_dscale)               \
  X(bvq_fakequant_bwd_shard)                 \
  X(bvq_shard_unpack_deposit)                \
  X(bvq_shard_pack)                          \
  X(bvq_shard_unpack)
#define BVQ_DECLARE(name) decltype(&name) p_##name = nullptr;
BVQ_ENTRIES(BVQ_DECLARE)
#undef BVQ_DECLARE

py::object* g_fallback = nullptr;  // python: (x, scale, zp, stat, int_threshold, desc, qrange, shape, gy, gscale, group) -> dx
                                   // (leaked on purpose: destroying it after the interpreter has gone would crash)

void check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + (p_bvq_last_error ? p_bvq_last_error() : "error"));
}

at::ScalarType dtype_of(int code) {
  return code == BVQ_F32 ? at::kFloat : (code == BVQ_BF16 ? at::kBFloat16 : at::kHalf);
}
int code_of(at::ScalarType t) { return t == at::kFloat ? BVQ_F32 : (t == at::kBFloat16 ? BVQ_BF16 : BVQ_F16); }

bool aligned16(const at::Tensor& t) { return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0; }

struct Params {
  bvq_quant_desc d;
  double min_val, thr_fwd, thr_bwd, thr_raw;
  int has_min, scale_dtype;
  int64_t stream;
  std::vector<int64_t> shape;  // scaling shape
  // activation node only
  int quot_dtype = BVQ_F32;  // dtype torch computes dscale / int_threshold in
  double momentum = 0.0;
  int first_batch = 0;
  c10::intrusive_ptr<c10d::ProcessGroup> group;  // null: not sharded
  py::object py_group;                           // the same group as the Python object (for the fallback)
};

// the process groups seen so far, by name: the C++ group for the collectives, the Python object for the fallback
// (leaked on purpose, like g_fallback)
struct GroupRef {
  c10::intrusive_ptr<c10d::ProcessGroup> pg;
  py::object* py;
};
std::unordered_map<std::string, GroupRef>& groups() {
  static auto* m = new std::unordered_map<std::string, GroupRef>();
  return *m;
}

// ---- RCCL's C API, from the library torch.distributed already runs on ------------------------------------------------
struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl t;
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy that is already in the process (torch's), never a second one
      if (h) break;
    }
    if (!h) return t;
    t.get_unique_id = reinterpret_cast<decltype(t.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    t.comm_init_rank = reinterpret_cast<decltype(t.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    t.comm_destroy = reinterpret_cast<decltype(t.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    t.all_reduce = reinterpret_cast<decltype(t.all_reduce)>(dlsym(h, "ncclAllReduce"));
    t.all_gather = reinterpret_cast<decltype(t.all_gather)>(dlsym(h, "ncclAllGather"));
    t.error_string = reinterpret_cast<decltype(t.error_string)>(dlsym(h, "ncclGetErrorString"));
    t.ok = t.get_unique_id && t.comm_init_rank && t.comm_destroy && t.all_reduce && t.all_gather && t.error_string;
    return t;
  }();
  return r;
}
void rccl_check(ncclResult_t rc, const char* what) {
  if (rc != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + rccl().error_string(rc));
}
// native communicators by process-group name (brevitas_amd.distributed.enable_native_collectives)
struct NativeComm {
  ncclComm_t comm;
  int world, rank;
};
std::unordered_map<std::string, NativeComm>& native_comms() {
  static auto* m = new std::unordered_map<std::string, NativeComm>();
  return *m;
}
const NativeComm* native_comm(const std::string& name) {
  auto it = native_comms().find(name);
  return it == native_comms().end() ? nullptr : &it->second;
}

std::vector<int64_t> desc_ints(const Params& p) {
  return {p.d.outer, p.d.channels, p.d.inner, p.d.x_dtype, p.d.ct_dtype, p.d.scale_dtype, p.d.zp_dtype,
          p.d.scale_per_channel, p.d.zp_per_channel, p.d.round_mode, p.d.scalar_mode, p.d.clamp_ste, p.d.out_kind,
          p.d.pre_op, p.d.codes_dtype, p.scale_dtype, p.stream, p.quot_dtype};
}
bvq_quant_desc desc_from(const std::vector<int64_t>& dv, const std::vector<double>& qr) {
  bvq_quant_desc d;
  d.outer = dv[0];
  d.channels = dv[1];
  d.inner = dv[2];
  d.x_dtype = (int32_t)dv[3];
  d.ct_dtype = (int32_t)dv[4];
  d.scale_dtype = (int32_t)dv[5];
  d.zp_dtype = (int32_t)dv[6];
  d.scale_per_channel = (int32_t)dv[7];
  d.zp_per_channel = (int32_t)dv[8];
  d.qmin = (float)qr[0];
  d.qmax = (float)qr[1];
  d.round_mode = (int32_t)dv[9];
  d.scalar_mode = (int32_t)dv[10];
  d.clamp_ste = (int32_t)dv[11];
  d.out_kind = (int32_t)dv[12];
  d.pre_op = (int32_t)dv[13];
  d.codes_dtype = (int32_t)dv[14];
  return d;
}

bool direct_gradient(const at::Tensor& gy, const at::Tensor& gscale, const at::Tensor& x) {
  return gy.defined() && !gscale.defined() && gy.is_contiguous() && gy.scalar_type() == x.scalar_type() && aligned16(gy) &&
         aligned16(x);
}

// group: the Python process group of a sharded activation, or null
torch::autograd::variable_list python_backward(torch::autograd::AutogradContext* ctx, const at::Tensor& gy,
                                               const at::Tensor& gscale, const py::object* group, size_t n_inputs) {
  const auto saved = ctx->get_saved_variables();
  torch::autograd::variable_list out(n_inputs);
  if (!gy.defined() && !gscale.defined()) return out;
  py::gil_scoped_acquire gil;
  py::object dx = (*g_fallback)(saved[0], saved[1], saved[2], saved[3], saved[4],
                                py::cast(ctx->saved_data["desc"].toIntVector()),
                                py::cast(ctx->saved_data["qrange"].toDoubleVector()),
                                py::cast(ctx->saved_data["shape"].toIntVector()), gy.defined() ? py::cast(gy) : py::none(),
                                gscale.defined() ? py::cast(gscale) : py::none(), group ? *group : py::none());
  out[0] = dx.cast<at::Tensor>();
  return out;
}

// ---- weights: one-launch forward ------------------------------------------------------------------------------------
class StatsFakeQuant : public torch::autograd::Function<StatsFakeQuant> {
 public:
  // arrive: the stream's arrival buffer (an EMPTY tensor when absent)
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x,
                                                const at::Tensor& zp, const at::Tensor& int_threshold,
                                                const at::Tensor& arrive, int64_t wsb, const Params& p) {
    ctx->set_materialize_grads(false);
    const int64_t ch = p.d.channels;
    at::Tensor y = at::empty_like(x);
    at::Tensor stat = at::empty({ch}, x.options());
    at::Tensor scale = at::empty({ch}, x.options().dtype(dtype_of(p.scale_dtype)));
    at::Tensor ws = at::empty({wsb}, x.options().dtype(at::kByte));
    check(p_bvq_stats_fakequant_fwd(&p.d, x.data_ptr(), p.min_val, p.has_min, p.thr_fwd, stat.data_ptr(),
                                    scale.data_ptr(), y.data_ptr(), ws.data_ptr(), wsb,
                                    reinterpret_cast<void*>(p.stream)),
          "bvq_stats_fakequant_fwd");
    ctx->save_for_backward({x, scale, zp, stat, int_threshold, arrive});
    ctx->saved_data["desc"] = desc_ints(p);
    ctx->saved_data["qrange"] = std::vector<double>{p.d.qmin, p.d.qmax, p.thr_bwd, p.thr_raw};
    ctx->saved_data["shape"] = p.shape;
    at::Tensor scale_out = scale.view(p.shape), stat_out = stat.view(p.shape);
    ctx->mark_non_differentiable({stat_out});
    return {y, scale_out, stat_out};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &scale = saved[1], &zp = saved[2], &stat = saved[3], &arrive = saved[5];
    const at::Tensor& gy = grads[0];
    const at::Tensor& gscale = grads[1];
    if (!direct_gradient(gy, gscale, x)) return python_backward(ctx, gy, gscale, nullptr, 6);
    const auto dv = ctx->saved_data["desc"].toIntVector();
    const auto qr = ctx->saved_data["qrange"].toDoubleVector();
    const bvq_quant_desc d = desc_from(dv, qr);
    const int64_t wsb = p_bvq_fakequant_bwd_stats_workspace_bytes(&d);
    if (wsb <= 0) throw std::runtime_error("bvq_fakequant_bwd_stats: layout not covered (checked at forward)");
    at::Tensor dx = at::empty_like(x);
    at::Tensor ds = at::empty({d.channels}, x.options().dtype(at::kFloat));
    at::Tensor ws = at::empty({wsb}, x.options().dtype(at::kByte));
    const int sdt = (int)dv[15];
    void* st = reinterpret_cast<void*>(dv[16]);
    if (arrive.numel() >= d.channels && p_bvq_fakequant_bwd_stats_onepass_supported(&d)) {
      // one launch: the backward kernel's last-arriving wave per channel sums, converts and deposits
      check(p_bvq_fakequant_bwd_stats_onepass(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(),
                                              stat.data_ptr(), dx.data_ptr(), ds.data_ptr<float>(), sdt, qr[2], sdt,
                                              ws.data_ptr(), wsb, reinterpret_cast<uint32_t*>(arrive.data_ptr()),
                                              arrive.numel(), st),
            "bvq_fakequant_bwd_stats_onepass");
    } else {
      check(p_bvq_fakequant_bwd_stats(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(), stat.data_ptr(),
                                      dx.data_ptr(), ds.data_ptr<float>(), sdt, qr[2], sdt, ws.data_ptr(), wsb, st),
            "bvq_fakequant_bwd_stats");
    }
    torch::autograd::variable_list out(6);
    out[0] = dx;
    return out;
  }
};

// ---- activations: statistic kernel + quantizer kernel, optionally batch-sharded --------------------------------------
class ActStatsFakeQuant : public torch::autograd::Function<ActStatsFakeQuant> {
 public:
  // running / arrive: EMPTY tensors when absent (an undefined tensor cannot be an input of a C++ autograd Function)
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& x,
                                                const at::Tensor& zp, const at::Tensor& int_threshold,
                                                const at::Tensor& running, const at::Tensor& arrive, const Params& p) {
    ctx->set_materialize_grads(false);
    const bvq_quant_desc& d = p.d;
    const int64_t ch = d.channels;
    void* st = reinterpret_cast<void*>(p.stream);
    const bool sharded = (bool)p.group;
    const auto xdt = x.scalar_type();
    at::Tensor stat = at::empty({ch}, x.options());
    at::Tensor scale = at::empty({ch}, x.options().dtype(dtype_of(p.scale_dtype)));
    at::Tensor stat32;
    if (sharded) stat32 = at::empty({ch}, x.options().dtype(at::kFloat));
    const bool has_running = running.numel() > 0, has_arrive = arrive.numel() > 0;
    void* run_ptr = has_running ? running.data_ptr() : nullptr;
    const int run_dt = has_running ? code_of(running.scalar_type()) : BVQ_F32;
    const bool onepass = has_arrive &&
                         p_bvq_absmax_onepass_supported(d.x_dtype, x.data_ptr(), d.outer, ch, d.inner);
    if (onepass) {
      check(p_bvq_absmax_scale_onepass(d.pre_op, d.x_dtype, x.data_ptr(), d.outer, ch, d.inner,
                                       sharded ? BVQ_F32 : d.x_dtype, sharded ? stat32.data_ptr() : stat.data_ptr(),
                                       p.min_val, p.has_min, p.thr_fwd, p.scale_dtype,
                                       sharded ? nullptr : scale.data_ptr(), run_dt, sharded ? nullptr : run_ptr, p.momentum,
                                       p.first_batch, reinterpret_cast<uint32_t*>(arrive.data_ptr()), arrive.numel(), st),
            "bvq_absmax_scale_onepass");
    } else {
      const int64_t wsb = p_bvq_stats_workspace_bytes(BVQ_STAT_ABSMAX, d.x_dtype, d.outer, ch, d.inner);
      if (wsb < 0) throw std::runtime_error("bvq_stats_workspace_bytes: bad arguments");
      at::Tensor ws = at::empty({wsb > 8 ? wsb : 8}, x.options().dtype(at::kByte));
      if (sharded) {
        check(p_bvq_stats_pre(BVQ_STAT_ABSMAX, d.pre_op, d.x_dtype, x.data_ptr(), d.outer, ch, d.inner, BVQ_F32,
                              stat32.data_ptr(), ws.data_ptr(), ws.numel(), st),
              "bvq_stats_pre");
      } else if (has_running) {
        check(p_bvq_absmax_scale_running(d.pre_op, d.x_dtype, x.data_ptr(), d.outer, ch, d.inner, stat.data_ptr(),
                                         p.min_val, p.has_min, p.thr_fwd, p.scale_dtype, scale.data_ptr(), run_dt, run_ptr,
                                         p.momentum, p.first_batch, ws.data_ptr(), ws.numel(), st),
              "bvq_absmax_scale_running");
      } else {
        check(p_bvq_absmax_scale(d.pre_op, d.x_dtype, x.data_ptr(), d.outer, ch, d.inner, stat.data_ptr(), p.min_val,
                                 p.has_min, p.thr_fwd, p.scale_dtype, scale.data_ptr(), ws.data_ptr(), ws.numel(), st),
              "bvq_absmax_scale");
      }
    }
    if (sharded) {
      // the statistic of the whole batch is the max over the shards (exact, order-independent)
      // (issued with one rank too: a one-rank group then exercises the same c10d / RCCL calls as N ranks)
      if (const NativeComm* nc = native_comm(p.group->getGroupName())) {
        rccl_check(rccl().all_reduce(stat32.data_ptr<float>(), stat32.data_ptr<float>(), (size_t)ch, ncclFloat32, ncclMax,
                                     nc->comm, reinterpret_cast<hipStream_t>(st)),
                   "ncclAllReduce");
      } else {
        std::vector<at::Tensor> ts{stat32};
        c10d::AllreduceOptions opts;
        opts.reduceOp = c10d::ReduceOp::MAX;
        p.group->allreduce(ts, opts)->wait();
      }
      check(p_bvq_scale_from_stat_running(stat32.data_ptr<float>(), ch, d.x_dtype, stat.data_ptr(), p.min_val, p.has_min,
                                          p.thr_fwd, p.scale_dtype, scale.data_ptr(), run_dt, run_ptr, p.momentum,
                                          p.first_batch, st),
            "bvq_scale_from_stat_running");
    }
    at::Tensor y = at::empty(x.sizes(), x.options().dtype(dtype_of(d.ct_dtype)));
    check(p_bvq_fakequant_fwd(&d, x.data_ptr(), scale.data_ptr(), zp.data_ptr(), y.data_ptr(), nullptr, st),
          "bvq_fakequant_fwd");
    ctx->save_for_backward({x, scale, zp, stat, int_threshold, arrive});
    ctx->saved_data["desc"] = desc_ints(p);
    ctx->saved_data["qrange"] = std::vector<double>{d.qmin, d.qmax, p.thr_bwd, p.thr_raw};
    ctx->saved_data["shape"] = p.shape;
    if (sharded) {
      const std::string name = p.group->getGroupName();
      // names come back after destroy_process_group / init_process_group: the entry follows the live group
      auto it = groups().find(name);
      if (it == groups().end()) {
        groups()[name] = GroupRef{p.group, new py::object(p.py_group)};
      } else if (it->second.pg != p.group) {
        it->second.pg = p.group;
        *it->second.py = p.py_group;
      }
      ctx->saved_data["group"] = name;
    }
    at::Tensor scale_out = scale.view(p.shape), stat_out = stat.view(p.shape);
    ctx->mark_non_differentiable({stat_out});
    return {y, scale_out, stat_out};
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &scale = saved[1], &zp = saved[2], &stat = saved[3], &arrive = saved[5];
    const at::Tensor& gy = grads[0];
    const at::Tensor& gscale = grads[1];
    const auto dv = ctx->saved_data["desc"].toIntVector();
    const auto qr = ctx->saved_data["qrange"].toDoubleVector();
    const bvq_quant_desc d = desc_from(dv, qr);
    c10::intrusive_ptr<c10d::ProcessGroup> group;
    py::object* py_group = nullptr;
    if (ctx->saved_data.count("group")) {
      const GroupRef& ref = groups().at(ctx->saved_data["group"].toStringRef());
      group = ref.pg;
      py_group = ref.py;
    }
    const bool per_channel = d.channels > 1;
    if (!direct_gradient(gy, gscale, x)) {
      return python_backward(ctx, gy, gscale, py_group, 6);
    }
    void* st = reinterpret_cast<void*>(dv[16]);
    const int sdt = (int)dv[15];
    at::Tensor dx = at::empty_like(x);
    torch::autograd::variable_list out(6);
    if (per_channel) {
      const int64_t wsb = p_bvq_fakequant_bwd_stats_workspace_bytes(&d);
      if (wsb <= 0) throw std::runtime_error("bvq_fakequant_bwd_stats: layout not covered (checked at forward)");
      at::Tensor ws = at::empty({wsb}, x.options().dtype(at::kByte));
      uint32_t* arr = arrive.numel() > 0 ? reinterpret_cast<uint32_t*>(arrive.data_ptr()) : nullptr;
      const int64_t arr_n = arrive.numel();
      if (group) {
        const int rank = group->getRank(), world = group->getSize();
        at::Tensor msg = at::empty({2 * d.channels}, x.options().dtype(at::kDouble));
        at::Tensor pos = at::empty({d.channels}, x.options().dtype(at::kLong));
        check(p_bvq_fakequant_bwd_shard(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(), stat.data_ptr(),
                                        dx.data_ptr(), msg.data_ptr<double>(), pos.data_ptr<int64_t>(), rank, ws.data_ptr(),
                                        wsb, arr, arr_n, st),
              "bvq_fakequant_bwd_shard");
        at::Tensor gathered = at::empty({world * msg.numel()}, msg.options());
        if (const NativeComm* nc = native_comm(group->getGroupName())) {
          rccl_check(rccl().all_gather(msg.data_ptr<double>(), gathered.data_ptr<double>(), (size_t)msg.numel(),
                                       ncclFloat64, nc->comm, reinterpret_cast<hipStream_t>(st)),
                     "ncclAllGather");
        } else {
          group->_allgather_base(gathered, msg)->wait();
        }
        check(p_bvq_shard_unpack_deposit(d.x_dtype, x.data_ptr(), dx.data_ptr(), gathered.data_ptr<double>(), world,
                                         d.channels, rank, pos.data_ptr<int64_t>(), d.inner, sdt, qr[2], sdt, d.pre_op,
                                         nullptr, st),
              "bvq_shard_unpack_deposit");
      } else {
        at::Tensor ds = at::empty({d.channels}, x.options().dtype(at::kFloat));
        if (arr && arr_n >= d.channels && p_bvq_fakequant_bwd_stats_onepass_supported(&d)) {
          check(p_bvq_fakequant_bwd_stats_onepass(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(),
                                                  stat.data_ptr(), dx.data_ptr(), ds.data_ptr<float>(), sdt, qr[2], sdt,
                                                  ws.data_ptr(), wsb, arr, arr_n, st),
                "bvq_fakequant_bwd_stats_onepass");
        } else {
          check(p_bvq_fakequant_bwd_stats(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(),
                                          stat.data_ptr(), dx.data_ptr(), ds.data_ptr<float>(), sdt, qr[2], sdt,
                                          ws.data_ptr(), wsb, st),
                "bvq_fakequant_bwd_stats");
        }
      }
    } else {
      // one whole-tensor statistic: dx + dscale sums + the ties of the maximum, then dscale -> statistic's gradient,
      // spread evenly over the ties (torch.max backward), deposited in place
      const int64_t wsb = p_bvq_fakequant_bwd_workspace_bytes(&d);
      if (wsb < 0) throw std::runtime_error("bvq_fakequant_bwd_workspace_bytes failed");
      at::Tensor ws = at::empty({wsb > 8 ? wsb : 8}, x.options().dtype(at::kByte));
      at::Tensor ds = at::empty({1}, x.options().dtype(at::kFloat));
      at::Tensor info = at::empty({p_bvq_tie_info_bytes(1) / 8}, x.options().dtype(at::kLong));
      check(p_bvq_fakequant_bwd(&d, gy.data_ptr(), x.data_ptr(), scale.data_ptr(), zp.data_ptr(), dx.data_ptr(),
                                ds.data_ptr<float>(), nullptr, stat.data_ptr(), info.data_ptr<int64_t>(), ws.data_ptr(), ws.numel(),
                                st),
            "bvq_fakequant_bwd");
      const int64_t* total_ties = nullptr;
      at::Tensor total;
      if (group) {
        // batch shard: the shards' dscale sums are added (double, rank order: the same bits on every rank) and the ties
        // of the whole batch share the statistic's gradient evenly -- one all-gather of [dscale sum | tie count]
        const int rank = group->getRank(), world = group->getSize();
        at::Tensor msg = at::empty({2}, x.options().dtype(at::kDouble));
        check(p_bvq_shard_pack(ds.data_ptr<float>(), info.data_ptr<int64_t>(), 1, rank, 0, msg.data_ptr<double>(), st),
              "bvq_shard_pack");
        at::Tensor gathered = at::empty({world * msg.numel()}, msg.options());
        if (const NativeComm* nc = native_comm(group->getGroupName())) {
          rccl_check(rccl().all_gather(msg.data_ptr<double>(), gathered.data_ptr<double>(), (size_t)msg.numel(),
                                       ncclFloat64, nc->comm, reinterpret_cast<hipStream_t>(st)),
                     "ncclAllGather");
        } else {
          group->_allgather_base(gathered, msg)->wait();
        }
        total = at::empty({1}, x.options().dtype(at::kLong));
        check(p_bvq_shard_unpack(gathered.data_ptr<double>(), world, 1, rank, 0, ds.data_ptr<float>(),
                                 info.data_ptr<int64_t>(), total.data_ptr<int64_t>(), st),
              "bvq_shard_unpack");
        total_ties = total.data_ptr<int64_t>();
      }
      // (qr[3]: the threshold the quotient dscale / int_threshold is divided by; dv[17]: its dtype)
      check(p_bvq_stat_tie_apply_dscale(d.pre_op, d.x_dtype, x.data_ptr(), stat.data_ptr(), ds.data_ptr<float>(), sdt,
                                        qr[2], (int)dv[17], info.data_ptr<int64_t>(), total_ties, dx.data_ptr(), d.outer,
                                        d.channels, d.inner, st),
            "bvq_stat_tie_apply_dscale");
    }
    out[0] = dx;
    return out;
  }
};

Params make_params(const std::vector<int64_t>& di, double qmin, double qmax, double min_val, bool has_min, double thr_fwd,
                   double thr_bwd, double thr_raw, int64_t scale_dtype, std::vector<int64_t> shape, int64_t stream) {
  Params p;
  std::vector<double> qr{qmin, qmax};
  p.d = desc_from(di, qr);
  p.min_val = min_val;
  p.has_min = has_min ? 1 : 0;
  p.thr_fwd = thr_fwd;
  p.thr_bwd = thr_bwd;
  p.thr_raw = thr_raw;
  p.scale_dtype = (int)scale_dtype;
  p.stream = stream;
  p.shape = std::move(shape);
  return p;
}

}  // namespace

// resolve the C-ABI entries from the library the package loaded (path: brevitas_amd/libbvq.so)
void init(const std::string& lib_path, py::object fallback) {
  void* h = dlopen(lib_path.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) throw std::runtime_error(std::string("dlopen ") + lib_path + ": " + dlerror());
#define BVQ_RESOLVE(name)                                                                      \
  p_##name = reinterpret_cast<decltype(&name)>(dlsym(h, #name));                               \
  if (!p_##name) throw std::runtime_error(std::string("libbvq.so: missing entry point ") + #name);
  BVQ_ENTRIES(BVQ_RESOLVE)
#undef BVQ_RESOLVE
  const int ver = p_bvq_abi_version();
  if (ver != BVQ_ABI_VERSION)
    throw std::runtime_error("libbvq.so has ABI " + std::to_string(ver) + ", this node was built against ABI " +
                             std::to_string(BVQ_ABI_VERSION) + " (include/bvq.h): rebuild brevitas_amd/_bvq_autograd.so");
  g_fallback = new py::object(std::move(fallback));
}

int abi_version() { return BVQ_ABI_VERSION; }

// -> (y, scale, stat), or None when the one-launch forward / two-launch backward do not cover this layout
// desc: the 15 integer fields of bvq_quant_desc in order (qmin / qmax as floats)
py::object stats_fakequant(const at::Tensor& x, const at::Tensor& zp, const at::Tensor& int_threshold,
                           const py::object& arrive, const std::vector<int64_t>& di, double qmin, double qmax,
                           double min_val, bool has_min,
                           double thr_fwd, double thr_bwd, double thr_raw, int64_t scale_dtype,
                           std::vector<int64_t> shape, int64_t stream) {
  Params p = make_params(di, qmin, qmax, min_val, has_min, thr_fwd, thr_bwd, thr_raw, scale_dtype, std::move(shape), stream);
  if (!x.is_contiguous() || !aligned16(x)) return py::none();
  // coverage: both workspace queries are host-side and cheap (y's address only matters for its alignment: x's stands in)
  const int64_t wsb = p_bvq_stats_fakequant_fwd_workspace_bytes(&p.d, x.data_ptr(), x.data_ptr());
  if (wsb <= 0 || p_bvq_fakequant_bwd_stats_workspace_bytes(&p.d) <= 0) return py::none();
  at::Tensor arr_t = arrive.is_none() ? at::empty({0}, x.options().dtype(at::kInt)) : arrive.cast<at::Tensor>();
  auto out = StatsFakeQuant::apply(x, zp, int_threshold, arr_t, wsb, p);
  return py::make_tuple(out[0], out[1], out[2]);
}

// The activation route -> (y, scale, stat) or None (layout not covered).  running: the _RuntimeStats buffer to fold the
// statistic into, or None; arrive: the stream's arrival buffer (brevitas_amd._native.arrival_buffer) or None; group: the
// torch.distributed process group of a batch-sharded activation, or None; quot_dtype: dtype code of dscale / int_threshold
py::object act_stats_fakequant(const at::Tensor& x, const at::Tensor& zp, const at::Tensor& int_threshold,
                               const py::object& running, const py::object& arrive, const std::vector<int64_t>& di,
                               double qmin, double qmax, double min_val, bool has_min, double thr_fwd, double thr_bwd,
                               double thr_raw, int64_t scale_dtype, int64_t quot_dtype, std::vector<int64_t> shape,
                               int64_t stream, double momentum, bool first_batch, const py::object& group) {
  Params p = make_params(di, qmin, qmax, min_val, has_min, thr_fwd, thr_bwd, thr_raw, scale_dtype, std::move(shape), stream);
  p.momentum = momentum;
  p.first_batch = first_batch ? 1 : 0;
  if (!x.is_contiguous() || !aligned16(x)) return py::none();
  const bool per_channel = p.d.channels > 1;
  if (per_channel && p_bvq_fakequant_bwd_stats_workspace_bytes(&p.d) <= 0) return py::none();
  if (!group.is_none()) {
    p.group = group.cast<c10::intrusive_ptr<c10d::ProcessGroup>>();
    p.py_group = group;
  }
  at::Tensor run_t = running.is_none() ? at::empty({0}, x.options()) : running.cast<at::Tensor>();
  at::Tensor arr_t = arrive.is_none() ? at::empty({0}, x.options().dtype(at::kInt)) : arrive.cast<at::Tensor>();
  p.quot_dtype = (int)quot_dtype;
  auto out = ActStatsFakeQuant::apply(x, zp, int_threshold, run_t, arr_t, p);
  return py::make_tuple(out[0], out[1], out[2]);
}

// ---- native communicators: set-up and a probe, called from brevitas_amd/distributed.py ---------------------------------
py::object rccl_unique_id() {
  if (!rccl().ok) return py::none();
  ncclUniqueId id;
  rccl_check(rccl().get_unique_id(&id), "ncclGetUniqueId");
  return py::bytes(reinterpret_cast<const char*>(&id), sizeof(id));
}
// every rank of the group, with rank 0's id; the caller has made the rank's device current
void rccl_comm_init(const std::string& name, const std::string& id_bytes, int world, int rank) {
  if (!rccl().ok) throw std::runtime_error("RCCL's C API is not available in this process");
  if (id_bytes.size() != sizeof(ncclUniqueId)) throw std::runtime_error("rccl_comm_init: bad unique id");
  ncclUniqueId id;
  std::memcpy(&id, id_bytes.data(), sizeof(id));
  auto it = native_comms().find(name);
  if (it != native_comms().end()) {
    rccl().comm_destroy(it->second.comm);
    native_comms().erase(it);
  }
  ncclComm_t comm = nullptr;
  {
    py::gil_scoped_release nogil;
    rccl_check(rccl().comm_init_rank(&comm, world, id, rank), "ncclCommInitRank");
  }
  native_comms()[name] = NativeComm{comm, world, rank};
}
void rccl_comm_drop(const std::string& name) {
  auto it = native_comms().find(name);
  if (it == native_comms().end()) return;
  rccl().comm_destroy(it->second.comm);
  native_comms().erase(it);
}
bool rccl_comm_active(const std::string& name) { return native_comm(name) != nullptr; }
// the two calls the node makes, on the caller's current stream (probe and tests)
void rccl_all_reduce_max(const std::string& name, at::Tensor t, int64_t stream) {
  const NativeComm* nc = native_comm(name);
  if (!nc || t.scalar_type() != at::kFloat || !t.is_contiguous()) throw std::runtime_error("rccl_all_reduce_max: bad call");
  rccl_check(rccl().all_reduce(t.data_ptr<float>(), t.data_ptr<float>(), (size_t)t.numel(), ncclFloat32, ncclMax, nc->comm,
                               reinterpret_cast<hipStream_t>(stream)),
             "ncclAllReduce");
}
void rccl_all_gather_f64(const std::string& name, at::Tensor msg, at::Tensor out, int64_t stream) {
  const NativeComm* nc = native_comm(name);
  if (!nc || msg.scalar_type() != at::kDouble || out.scalar_type() != at::kDouble || !msg.is_contiguous() ||
      !out.is_contiguous() || out.numel() != msg.numel() * nc->world)
    throw std::runtime_error("rccl_all_gather_f64: bad call");
  rccl_check(rccl().all_gather(msg.data_ptr<double>(), out.data_ptr<double>(), (size_t)msg.numel(), ncclFloat64, nc->comm,
                               reinterpret_cast<hipStream_t>(stream)),
             "ncclAllGather");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rccl_unique_id", &rccl_unique_id, "ncclGetUniqueId as bytes (None: RCCL's C API is not in this process)");
  m.def("rccl_comm_init", &rccl_comm_init, "ncclCommInitRank for the process group of this name");
  m.def("rccl_comm_drop", &rccl_comm_drop, "destroy the native communicator of this group, if any");
  m.def("rccl_comm_active", &rccl_comm_active, "the group of this name has a native communicator");
  m.def("rccl_all_reduce_max", &rccl_all_reduce_max, "float32 all-reduce(MAX) in place on the given stream");
  m.def("rccl_all_gather_f64", &rccl_all_gather_f64, "float64 all-gather on the given stream");
  m.def("init", &init, "resolve libbvq.so, check its ABI version, register the python fallback of the backward");
  m.def("abi_version", &abi_version, "BVQ_ABI_VERSION of the include/bvq.h this module was built against");
  m.def("stats_fakequant", &stats_fakequant, "AbsMax -> scale -> IntQuant in one launch (weights), autograd node in C++");
  m.def("act_stats_fakequant", &act_stats_fakequant,
        "AbsMax -> scale -> IntQuant as statistic + quantizer kernels (activations, optionally batch-sharded), "
        "autograd node in C++");
}
