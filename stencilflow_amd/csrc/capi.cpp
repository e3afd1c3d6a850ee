// capi.cpp — the plan half of the C ABI of include/sf_hip.h (sf_plan_*): argument
// checks, status codes, nothing throws across the boundary.
#include "sf_internal.hpp"

#include <algorithm>

#include <cstdlib>
#include <memory>

using namespace sf;

extern "C" {

int sf_version(void) { return 1002; }  // 1.1: sf_halo_*, sf_plan_execute_decomposed, sf_plan_stream, sf_plan_num_buffers, sf_code_cache_stats; 1.2: the RCCL rung (sf_halo_rccl_id / _use_rccl / _transport / _configure)

const char* sf_last_error(void) { return sf::g_last_error.c_str(); }

int sf_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    sf::g_last_error = hipGetErrorString(e);
    return SF_ERR_DEVICE;
  }
  return n;
}

int sf_plan_create(const char* sfir_text, int device, const char* options, sf_plan** out_plan) {
  SF_API_BEGIN
  if (!sfir_text || !out_plan) throw Error(SF_ERR_INVALID, "null argument");
  std::unique_ptr<sf_plan> pl(new sf_plan);
  pl->P = parse_sfir(sfir_text);
  // SF_HIP_OPTIONS (same syntax) supplies site-wide defaults; the caller's options
  // take precedence key by key
  {
    const char* env = std::getenv("SF_HIP_OPTIONS");
    pl->opt = Options(env && *env ? env : nullptr);
    const Options own(options);
    for (auto& kv : own.kv) pl->opt.kv[kv.first] = kv.second;
  }
  pl->device = device;
  build_plan(*pl);
  *out_plan = pl.release();
  return SF_OK;
  SF_API_END
}

int sf_plan_destroy(sf_plan* plan) {
  SF_API_BEGIN
  if (!plan) return SF_OK;
  if (plan->device_ready) {
    (void)hipSetDevice(plan->device);
    (void)hipStreamSynchronize(plan->stream);
    collect_profile(*plan);
    for (auto& b : plan->buffers)
      if (b.d) (void)hipFree(b.d);
    for (auto& k : plan->kernels)
      if (k.mod) (void)hipModuleUnload(k.mod);
    for (auto& k : plan->check_kernels)
      if (k.mod) (void)hipModuleUnload(k.mod);
    if (plan->chain_graph) (void)hipGraphExecDestroy(plan->chain_graph);
    (void)hipEventDestroy(plan->ev_begin);
    (void)hipEventDestroy(plan->ev_end);
    (void)hipStreamDestroy(plan->stream);
  }
  delete plan;
  return SF_OK;
  SF_API_END
}

int sf_code_cache_stats(long* disk_hits, long* compiled, long* rebuilt, int drop_process_level) {
  sf::code_cache_stats(disk_hits, compiled, rebuilt, drop_process_level != 0);
  return SF_OK;
}

long sf_self_checks_run(void) { return sf::self_checks_run(); }

int sf_plan_kernel_verdict(const sf_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->kernels.size()) ? p->kernels[i].verdict : SF_ERR_INVALID;
}

int sf_plan_num_inputs(const sf_plan* p) { return p ? p->P.num_inputs : SF_ERR_INVALID; }
int sf_plan_num_scalars(const sf_plan* p) { return p ? p->P.num_scalar_inputs : SF_ERR_INVALID; }
int sf_plan_num_outputs(const sf_plan* p) { return p ? p->P.num_outputs : SF_ERR_INVALID; }

static const char* field_name_by_io(const sf_plan* p, Role role, int index) {
  if (!p) return nullptr;
  for (auto& f : p->P.fields)
    if (f.role == role && f.io_index == index) return f.name.c_str();
  return nullptr;
}
const char* sf_plan_input_name(const sf_plan* p, int i) { return field_name_by_io(p, Role::Input, i); }
const char* sf_plan_output_name(const sf_plan* p, int i) { return field_name_by_io(p, Role::Output, i); }
const char* sf_plan_scalar_name(const sf_plan* p, int index) {
  if (!p) return nullptr;
  for (auto& s : p->P.scalars)
    if (!s.is_const && s.input_index == index) return s.name.c_str();
  return nullptr;
}
size_t sf_plan_input_bytes(const sf_plan* p, int i) {
  if (!p || i < 0 || i >= p->P.num_inputs) return 0;
  const Buffer& b = p->buffers[p->input_buf[i]];
  return b.slabbed ? b.plane_bytes * (size_t)p->n_local : b.bytes();
}
size_t sf_plan_output_bytes(const sf_plan* p, int i) {
  if (!p || i < 0 || i >= p->P.num_outputs) return 0;
  const Buffer& b = p->buffers[p->output_buf[i]];
  return b.slabbed ? b.plane_bytes * (size_t)p->n_local : b.bytes();
}

int sf_plan_set_scalars(sf_plan* plan, const double* values, int count) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  if (count != plan->P.num_scalar_inputs) throw Error(SF_ERR_INVALID, "wrong number of scalar values");
  for (int i = 0; i < count; ++i) plan->scalar_values[i] = values[i];
  plan->scalars_set = true;
  return SF_OK;
  SF_API_END
}

int sf_plan_run(sf_plan* plan, const void* const* host_inputs, void* const* host_outputs, int repetitions) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  upload(*plan, host_inputs);
  execute(*plan, repetitions);
  SF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  collect_profile(*plan);
  download(*plan, host_outputs);
  return SF_OK;
  SF_API_END
}

int sf_plan_upload(sf_plan* plan, const void* const* host_inputs) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  upload(*plan, host_inputs);
  return SF_OK;
  SF_API_END
}

int sf_plan_execute(sf_plan* plan, int repetitions) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  execute(*plan, repetitions);
  return SF_OK;
  SF_API_END
}

int sf_plan_synchronize(sf_plan* plan) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  ensure_device(*plan);
  SF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  collect_profile(*plan);
  return SF_OK;
  SF_API_END
}

int sf_plan_download(sf_plan* plan, void* const* host_outputs) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  download(*plan, host_outputs);
  return SF_OK;
  SF_API_END
}

int sf_plan_elapsed_ms(sf_plan* plan, double* ms) {
  SF_API_BEGIN
  if (!plan || !ms) throw Error(SF_ERR_INVALID, "null argument");
  if (!plan->timed) throw Error(SF_ERR_STATE, "nothing has been executed yet");
  float f = 0;
  SF_HIP_CHECK(hipEventElapsedTime(&f, plan->ev_begin, plan->ev_end));
  *ms = f;
  return SF_OK;
  SF_API_END
}

int sf_plan_num_launches(const sf_plan* p) { return p ? (int)p->steps.size() : SF_ERR_INVALID; }
int sf_plan_num_kernels(const sf_plan* p) { return p ? (int)p->kernels.size() : SF_ERR_INVALID; }
const char* sf_plan_kernel_name(const sf_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->kernels.size()) ? p->kernels[i].name.c_str() : nullptr;
}
const char* sf_plan_kernel_source(const sf_plan* p, int i) {
  return (p && i >= 0 && i < (int)p->kernels.size()) ? p->kernels[i].source.c_str() : nullptr;
}
int sf_plan_kernel_stats(sf_plan* p, int i, int* launches, double* total_ms, double* updates,
                         double* alg_bytes) {
  if (!p || i < 0 || i >= (int)p->kernels.size()) return SF_ERR_INVALID;
  const CompiledKernel& k = p->kernels[i];
  if (launches) *launches = k.launches;
  if (total_ms) *total_ms = k.total_ms;
  if (updates) *updates = k.updates_per_launch;
  if (alg_bytes) *alg_bytes = k.alg_bytes_per_launch;
  return SF_OK;
}
const char* sf_compiler_id(void) {
  static thread_local std::string id;
  try {
    id = sf::compiler_id();
  } catch (...) {
    id = "?";
  }
  return id.c_str();
}
int sf_plan_step_kernel(const sf_plan* p, int step) {
  if (!p || step < 0 || step >= (int)p->steps.size()) return SF_ERR_INVALID;
  return p->steps[step].ck;
}
int sf_plan_kernel_object(const sf_plan* p, int i, const void** data, size_t* bytes, const char** flags) {
  if (!p || i < 0 || i >= (int)p->kernels.size()) return SF_ERR_INVALID;
  const CompiledKernel& k = p->kernels[i];
  if (data) *data = k.code.data();
  if (bytes) *bytes = k.code.size();
  if (flags) *flags = k.flags.c_str();
  return SF_OK;
}
int sf_plan_kernel_launch_times(sf_plan* plan, int i, double* min_ms, double* median_ms, double* max_ms) {
  SF_API_BEGIN
  if (!plan || i < 0 || i >= (int)plan->kernels.size()) throw Error(SF_ERR_INVALID, "sf_plan_kernel_launch_times: bad argument");
  if (plan->device_ready && plan->profile) {
    SF_HIP_CHECK(hipSetDevice(plan->device));
    SF_HIP_CHECK(hipStreamSynchronize(plan->stream));
    collect_profile(*plan);
  }
  std::vector<float> v = plan->kernels[i].launch_ms;
  if (v.empty()) throw Error(SF_ERR_STATE, "sf_plan_kernel_launch_times: no profiled launch of this kernel (sf_plan_set_profile)");
  std::sort(v.begin(), v.end());
  if (min_ms) *min_ms = v.front();
  if (max_ms) *max_ms = v.back();
  if (median_ms) *median_ms = v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
  return SF_OK;
  SF_API_END
}
int sf_plan_set_profile(sf_plan* plan, int on) {
  SF_API_BEGIN
  if (!plan) throw Error(SF_ERR_INVALID, "null plan");
  if (plan->device_ready) {
    SF_HIP_CHECK(hipSetDevice(plan->device));
    SF_HIP_CHECK(hipStreamSynchronize(plan->stream));
    collect_profile(*plan);
  }
  for (auto& k : plan->kernels) {
    k.launches = 0;
    k.total_ms = 0;
    k.planes_launched = 0;
    k.launch_ms.clear();
  }
  plan->profile = on != 0;
  return SF_OK;
  SF_API_END
}
int sf_plan_kernel_planes(sf_plan* p, int i, double* planes) {
  if (!p || !planes || i < 0 || i >= (int)p->kernels.size()) return SF_ERR_INVALID;
  *planes = p->kernels[i].planes_launched;
  return SF_OK;
}
int sf_plan_kernel_resources(const sf_plan* p, int i, int* vgprs, int* agprs, int* spills, int* scratch,
                             int* lds) {
  if (!p || i < 0 || i >= (int)p->kernels.size()) return SF_ERR_INVALID;
  const CompiledKernel& k = p->kernels[i];
  if (vgprs) *vgprs = k.vgprs;
  if (agprs) *agprs = k.agprs;
  if (spills) *spills = k.spills;
  if (scratch) *scratch = k.scratch;
  if (lds) *lds = k.lds;
  // (diagnostics: SF_HIP_REPORT_SGPR_SPILLS=1 reports SGPR spills in place of the scratch size)
  // (... plus 1000 x the EXEC restores found behind allocator code, count_late_exec_restores)
  if (scratch && std::getenv("SF_HIP_REPORT_SGPR_SPILLS")) *scratch = k.sgpr_spills + 1000 * k.late_exec_restores;
  return SF_OK;
}
const char* sf_describe_options(void) {
  static const std::string text = describe_options();
  return text.c_str();
}
const char* sf_plan_describe(const sf_plan* p) { return p ? p->description.c_str() : nullptr; }

int sf_plan_num_steps(const sf_plan* p) { return p ? (int)p->steps.size() : SF_ERR_INVALID; }
int sf_plan_step_halo(const sf_plan* p, int step, int* buffer_id, int* depth) {
  if (!p || step < 0 || step >= (int)p->steps.size()) return SF_ERR_INVALID;
  if (buffer_id) *buffer_id = p->steps[step].halo_buf;
  if (depth) *depth = p->steps[step].halo_buf >= 0 ? p->steps[step].halo_depth : 0;
  return SF_OK;
}
int sf_plan_step_inputs(const sf_plan* p, int step, int* buffer_ids, int capacity) {
  if (!p || step < 0 || step >= (int)p->steps.size()) return SF_ERR_INVALID;
  const auto& in = p->steps[step].in_bufs;
  for (int i = 0; buffer_ids && i < (int)in.size() && i < capacity; ++i) buffer_ids[i] = in[i];
  return (int)in.size();
}
int sf_plan_step_outputs(const sf_plan* p, int step, int* buffer_ids, int capacity) {
  if (!p || step < 0 || step >= (int)p->steps.size()) return SF_ERR_INVALID;
  const std::vector<int>& out = p->steps[step].out_bufs;
  for (int i = 0; buffer_ids && i < (int)out.size() && i < capacity; ++i) buffer_ids[i] = out[i];
  return (int)out.size();
}
int sf_plan_step_output(const sf_plan* p, int step) {
  if (!p || step < 0 || step >= (int)p->steps.size()) return SF_ERR_INVALID;
  return p->steps[step].out_buf;
}
int sf_plan_execute_step(sf_plan* plan, int step, int part, void* stream) {
  SF_API_BEGIN
  if (!plan || step < 0 || step >= (int)plan->steps.size() || part < 0 || part > 3)
    throw Error(SF_ERR_INVALID, "bad step or part");
  ensure_device(*plan);
  autotune(*plan);
  launch_step(*plan, plan->steps[step], part, stream ? (hipStream_t)stream : plan->stream);
  return SF_OK;
  SF_API_END
}
int sf_plan_execute_step_ranges(sf_plan* plan, int step, int i_begin, int i_end, int i_begin2, int i_end2,
                                void* stream) {
  SF_API_BEGIN
  if (!plan || step < 0 || step >= (int)plan->steps.size()) throw Error(SF_ERR_INVALID, "bad step");
  ensure_device(*plan);
  autotune(*plan);
  launch_ranges(*plan, plan->steps[step], i_begin, i_end, i_begin2, i_end2,
                stream ? (hipStream_t)stream : plan->stream);
  return SF_OK;
  SF_API_END
}
int sf_plan_set_reserved_cus(sf_plan* plan, int cus) {
  SF_API_BEGIN
  if (!plan || cus < 0 || cus >= 256) throw sf::Error(SF_ERR_INVALID, "reserved compute units must be in [0, 256)");
  plan->reserved_cus = cus;
  return SF_OK;
  SF_API_END
}

int sf_plan_stream(sf_plan* plan, void** stream) {
  SF_API_BEGIN
  if (!plan || !stream) throw Error(SF_ERR_INVALID, "null argument");
  ensure_device(*plan);
  *stream = (void*)plan->stream;
  return SF_OK;
  SF_API_END
}

int sf_plan_num_buffers(const sf_plan* p) { return p ? (int)p->buffers.size() : SF_ERR_INVALID; }

int sf_plan_buffer_info(const sf_plan* p, int id, void** device_ptr, size_t* plane_bytes, int* planes) {
  SF_API_BEGIN
  if (!p || id < 0 || id >= (int)p->buffers.size()) throw Error(SF_ERR_INVALID, "bad buffer id");
  ensure_device(*const_cast<sf_plan*>(p));
  if (device_ptr) *device_ptr = p->buffers[id].d;
  if (plane_bytes) *plane_bytes = p->buffers[id].plane_bytes;
  if (planes) *planes = p->buffers[id].planes;
  return SF_OK;
  SF_API_END
}
int sf_plan_input_buffer(const sf_plan* p, int i) {
  return (p && i >= 0 && i < p->P.num_inputs) ? p->input_buf[i] : SF_ERR_INVALID;
}
int sf_plan_output_buffer(const sf_plan* p, int i) {
  return (p && i >= 0 && i < p->P.num_outputs) ? p->output_buf[i] : SF_ERR_INVALID;
}

}  // extern "C"
