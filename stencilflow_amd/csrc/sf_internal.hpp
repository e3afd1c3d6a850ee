// sf_internal.hpp — declarations shared by the translation units of libsf_hip.so
// (codecache.cpp, planner.cpp, exec.cpp, capi.cpp, halo.cpp).  Host code only; every
// device kernel of a plan is generated per program (codegen.hpp, kernels/*.h) and
// compiled for gfx950 at plan creation, the way the reference compiles each program's
// SDFG before calling it (stencilflow/run_program.py:118-128).
#pragma once
#include "../../include/sf_hip.h"

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

#include "codegen.hpp"

namespace sf {

extern thread_local std::string g_last_error;

#define SF_HIP_CHECK(expr)                                                            \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      throw ::sf::Error(SF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---------------------------------------------------------------- options
struct Options {
  std::map<std::string, std::string> kv;
  explicit Options(const char* text) {
    if (!text) return;
    std::string s(text), item;
    std::istringstream is(s);
    while (std::getline(is, item, ';')) {
      if (item.empty()) continue;
      size_t eq = item.find('=');
      if (eq == std::string::npos) throw Error(SF_ERR_INVALID, "option without '=': " + item);
      kv[item.substr(0, eq)] = item.substr(eq + 1);
    }
  }
  long long get(const std::string& k, long long dflt) const {
    auto it = kv.find(k);
    return it == kv.end() ? dflt : std::stoll(it->second);
  }
  std::string gets(const std::string& k, const std::string& dflt) const {
    auto it = kv.find(k);
    return it == kv.end() ? dflt : it->second;
  }
};

// ---------------------------------------------------------------- plan pieces
struct CompiledKernel {
  std::string name, source;
  std::string flags;  // extra compiler flags (space-separated), part of the cache key
  std::vector<char> code;
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  int launches = 0;
  double total_ms = 0;
  double planes_launched = 0;  // planes written by the profiled launches
  std::vector<float> launch_ms;  // durations of the profiled launches (the first 16384), for min / median / max
  double updates_per_launch = 0, alg_bytes_per_launch = 0;
  // from the code object's amdhsa metadata (msgpack note)
  int vgprs = -1, agprs = -1, sgprs = -1, spills = -1, scratch = -1, lds = -1, sgpr_spills = -1;
  int late_exec_restores = 0;  // see count_late_exec_restores()
  bool from_disk = false;  // the code object came from the on-disk cache
  // verdict of the plan-time self-check (exec.cpp: self_check): 0 not checked yet, 1 equal to the generic
  // operator kernels bit for bit on a seeded tile, 2 differed -- never launched again; travels with the
  // code object through both cache levels
  int verdict = 0;
  std::string cache_key;   // name + flags + source: the key of both cache levels
  bool foreign = false;    // diagnostics: a hand-assembled object from $SF_HIP_OBJECT_DIR took the compiler's place
  bool env_flags = false;  // diagnostics: compiled with $SF_HIP_EXTRA_FLAGS
};

struct Buffer {
  DT dt = DT::F32;
  bool slabbed = true;     // has the stream dimension (I0)
  size_t plane_bytes = 0;  // bytes of one I0 plane (whole array if !slabbed)
  int planes = 1;          // local planes incl. halos
  void* d = nullptr;
  size_t bytes() const { return plane_bytes * (size_t)planes; }
};

struct Step {
  bool star = false;     // plane-streaming launch (star3d.h or, with `compact`, compact3d.h)
  bool compact = false;
  bool wide = false;  // radius-2 star launch (kernels/wstar3d.h)
  bool dense = false;  // dense neighbourhood of radius 2, tiles staged in LDS (kernels/dense3d.h)
  std::vector<int> kernels;    // program kernel indices fused in this launch
  int ck = -1;                 // compiled kernel
  std::vector<int> in_bufs;    // argument order
  int out_buf = -1;            // the (first) field the launch materialises
  std::vector<int> out_bufs;   // all of them, output-pointer order (DAG groups: several); out_bufs[0] == out_buf
  std::vector<std::string> out_names;
  std::vector<int> scalars;    // run-time scalars, argument / struct order
  std::vector<size_t> scalar_offsets;
  size_t scalars_bytes = 4;
  StarCfg cfg;
  int num_aux = 0;                    // centre-only auxiliary fields of a star step
  int generic_vk = 1;                 // points per thread of a generic step
  int generic_ppt = 1;                // planes per thread of a generic step
  int halo_buf = -1, halo_depth = 0;  // what must be exchanged before the step
  std::string note;
  std::vector<std::pair<StarCfg, int>> alts;  // autotune candidates (tile shape, compiled kernel)
  std::string sig;                            // steps with the same signature share the choice
  std::vector<std::string> read_names;        // the fields behind in_bufs, same order
};

}  // namespace sf

struct sf_plan {
  sf::Program P;
  sf::Options opt{nullptr};
  int device = 0;
  bool device_ready = false;
  bool self_checked = false;  // the plan-time self-check has run to its end (a verdict for every fused kernel)
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  bool timed = false, profile = false;
  std::vector<sf::CompiledKernel> kernels;
  std::map<std::string, int> kernel_by_source;
  std::vector<sf::Buffer> buffers;
  std::vector<sf::Step> steps;
  std::vector<int> input_buf, output_buf;  // by io_index
  std::vector<double> scalar_values;       // by Scalar::input_index
  // one repetition of the whole chain as an instantiated hipGraph (launch-bound
  // plans only, see execute()); rebuilt when the scalars it captured change
  hipGraphExec_t chain_graph = nullptr;
  std::vector<double> chain_graph_scalars;
  double max_updates_per_launch = 0;
  int reserved_cus = 0;  // compute units the star launches leave free (sf_plan_set_reserved_cus)
  bool scalars_set = false;
  // slab decomposition of I0
  long long n_local = 0, goff = 0, plan_extent = 0;
  int halo = 0;
  std::string description;
  bool autotuned = false;
  std::string poisoned;  // a failed plan-time self-check: every later use of the plan reports it
  // per-launch profiling events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<int> prof_kernel;
  std::vector<int> prof_planes;
  std::vector<sf::CompiledKernel> check_kernels;  // reference operators of the self-check (loaded modules)
};

namespace sf {

// ---- codecache.cpp: hipRTC, code-object metadata, the EXEC-restore detector, caches
int intern_kernel(sf_plan& pl, const std::string& prefix, const std::string& source,
                  const std::string& flags_in = "");
// compile `k` again (a cached object the loader rejected), refresh metadata and both cache levels
void recompile_kernel(CompiledKernel& k);
int count_late_exec_restores(const std::vector<char>& code);
bool kernel_unsafe(const CompiledKernel& k);
bool kernel_slow(const CompiledKernel& k);
void code_cache_stats(long* disk_hits, long* compiled, long* rebuilt, bool drop_process_level);
// a kernel that belongs to no plan's kernel list (the self-check's reference operators): both cache levels
CompiledKernel compile_cached(const std::string& prefix, const std::string& source, const std::string& flags);
// store the self-check's verdict with the code object (process and disk level)
void record_verdict(CompiledKernel& k, int verdict, bool persist = true);

// the device compiler of this process (hipRTC + the comgr it binds); $SF_HIP_COMGR pins the latter
std::string compiler_id();
void check_pinned_compiler();

// ---- planner.cpp: launch groups, tile search, buffers
void build_plan(sf_plan& pl);
std::string describe_plan(const sf_plan& pl);
std::string describe_options();  // the table of plan options (planner.cpp: kOptions)
size_t star_lds_bytes(const StarCfg& c, DT dt);
long long star_chunk_length(const sf_plan& pl, const StarCfg& c, DT dt, int range, int ranges = 1);

// ---- exec.cpp: device state, launches, profiling, autotuning
void ensure_device(sf_plan& pl);
void launch_step(sf_plan& pl, const Step& st, int part, hipStream_t stream);
void launch_ranges(sf_plan& pl, const Step& st, int i_begin, int i_end, int i_begin2, int i_end2,
                   hipStream_t stream);
void collect_profile(sf_plan& pl);
void upload(sf_plan& pl, const void* const* host_inputs);
void download(sf_plan& pl, void* const* host_outputs);
void autotune(sf_plan& pl);
void execute(sf_plan& pl, int repetitions);
void self_check(sf_plan& pl);
long self_checks_run();

}  // namespace sf

// ---- the C ABI never lets an exception out
#define SF_API_BEGIN try {
#define SF_API_END                                       \
  }                                                      \
  catch (const sf::Error& e) {                           \
    sf::g_last_error = e.what();                         \
    return e.status;                                     \
  }                                                      \
  catch (const std::exception& e) {                      \
    sf::g_last_error = e.what();                         \
    return SF_ERR_INVALID;                               \
  }                                                      \
  catch (...) {                                          \
    sf::g_last_error = "unknown failure";                \
    return SF_ERR_INVALID;                               \
  }
