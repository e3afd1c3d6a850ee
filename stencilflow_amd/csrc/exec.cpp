// exec.cpp — run time of a plan: device state, kernel launches over plane ranges,
// per-launch profiling, plan-time autotuning, the chain as a hipGraph.
// (Role in the reference: program(**dace_args), stencilflow/run_program.py:164-178.)
#include "sf_internal.hpp"

#include <map>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <set>
#include <sstream>

namespace sf {

// ---------------------------------------------------------------- runtime
void ensure_device(sf_plan& pl) {
  if (!pl.poisoned.empty()) throw Error(SF_ERR_UNSUPPORTED, pl.poisoned);
  if (pl.device_ready) {
    // every entry point runs on the plan's device, whatever device the calling
    // thread used last (one thread may drive plans on several GPUs)
    SF_HIP_CHECK(hipSetDevice(pl.device));
    // (a self-check that ended in an error other than a verdict -- scratch memory, a reference kernel that
    // did not compile, a launch error -- must not leave the fused kernels unverified: it runs again)
    if (!pl.self_checked) {
      self_check(pl);
      pl.self_checked = true;
    }
    return;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    throw Error(SF_ERR_DEVICE, "no HIP device available: the HIP backend cannot run without a GPU");
  if (pl.device < 0 || pl.device >= count) throw Error(SF_ERR_DEVICE, "device index out of range");
  SF_HIP_CHECK(hipSetDevice(pl.device));
  SF_HIP_CHECK(hipStreamCreateWithFlags(&pl.stream, hipStreamNonBlocking));
  SF_HIP_CHECK(hipEventCreate(&pl.ev_begin));
  SF_HIP_CHECK(hipEventCreate(&pl.ev_end));
  for (auto& k : pl.kernels) {
    hipError_t e = hipModuleLoadData(&k.mod, k.code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&k.fn, k.mod, k.name.c_str());
    if (e != hipSuccess && k.from_disk) {
      // a cached object the loader rejects (built by another compiler patch level
      // ...): drop it from both cache levels, recompile once and try again
      (void)hipGetLastError();
      if (k.mod) (void)hipModuleUnload(k.mod);
      k.mod = nullptr;
      k.fn = nullptr;
      recompile_kernel(k);
      // the fresh object is judged like any other: one that shows the compiler fault
      // of DESIGN.md 5.1 is never launched (the planner accepted the cached one)
      if (kernel_unsafe(k))
        throw Error(SF_ERR_UNSUPPORTED, "recompiled code object of " + k.name +
                                            " has register-allocator code ahead of an EXEC restore; drop the "
                                            "code cache ($SF_HIP_CACHE_DIR) and create the plan again");
      e = hipModuleLoadData(&k.mod, k.code.data());
      if (e == hipSuccess) e = hipModuleGetFunction(&k.fn, k.mod, k.name.c_str());
    }
    if (e != hipSuccess)
      throw Error(SF_ERR_DEVICE, "loading code object of " + k.name + ": " + hipGetErrorString(e));
  }
  for (auto& b : pl.buffers) {
    SF_HIP_CHECK(hipMalloc(&b.d, b.bytes()));
    SF_HIP_CHECK(hipMemsetAsync(b.d, 0, b.bytes(), pl.stream));
  }
  SF_HIP_CHECK(hipStreamSynchronize(pl.stream));
  pl.device_ready = true;
  self_check(pl);
  pl.self_checked = true;
}

static void store_scalar(char* dst, DT dt, double v) {
  switch (dt) {
    case DT::F32: { float x = (float)v; std::memcpy(dst, &x, 4); break; }
    case DT::F64: { std::memcpy(dst, &v, 8); break; }
    case DT::I32: { int x = (int)v; std::memcpy(dst, &x, 4); break; }
    default: { long long x = (long long)v; std::memcpy(dst, &x, 8); break; }
  }
}

void launch_step(sf_plan& pl, const Step& st, int part, hipStream_t stream) {
  int i_begin = 0, i_end = (int)pl.n_local;
  if (part != 0) {
    const int h = std::max(pl.halo, 1);
    if (2 * h > pl.n_local) throw Error(SF_ERR_STATE, "slab too thin to split into boundary and interior");
    if (part == 1) i_end = h;
    else if (part == 2) i_begin = (int)pl.n_local - h;
    else { i_begin = h; i_end = (int)pl.n_local - h; }
  }
  launch_ranges(pl, st, i_begin, i_end, 0, 0, stream);
}

// Launch `st` over planes [i_begin, i_end) and, in the same launch where the
// kernel supports it, [i_begin2, i_end2) (owned-plane coordinates; negative /
// beyond-n_local values address halo planes).
void launch_ranges(sf_plan& pl, const Step& st, int i_begin, int i_end, int i_begin2, int i_end2,
                          hipStream_t stream) {
  const Program& P = pl.P;
  const int lo_limit = -pl.halo, hi_limit = (int)pl.n_local + pl.halo;
  if (i_begin < lo_limit || i_end > hi_limit || (i_begin2 < i_end2 && (i_begin2 < lo_limit || i_end2 > hi_limit)))
    throw Error(SF_ERR_INVALID, "plane range outside the slab and its halo");
  const bool second = i_begin2 < i_end2;
  if (i_begin >= i_end && !second) return;
  // the planes a range READS must exist too: on a side with a neighbouring slab
  // the launch reaches `halo_depth` planes beyond the range it writes
  if (st.halo_buf >= 0 && st.halo_depth > 0) {
    const int reach = st.halo_depth;
    const bool lower_neighbour = pl.goff > 0, upper_neighbour = pl.goff + pl.n_local < P.n[0];
    const int firsts[2] = {i_begin, i_begin2}, lasts[2] = {i_end, i_end2};
    for (int r = 0; r < 2; ++r) {
      if (firsts[r] >= lasts[r]) continue;
      if ((lower_neighbour && firsts[r] - reach < lo_limit) || (upper_neighbour && lasts[r] + reach > hi_limit))
        throw Error(SF_ERR_INVALID, "plane range reads beyond the halo of a slab with a neighbour");
    }
  }
  if (P.num_scalar_inputs > 0 && !pl.scalars_set)
    throw Error(SF_ERR_STATE, "the program has 0-D inputs: call sf_plan_set_scalars first");
  CompiledKernel& ck = pl.kernels[st.ck];
  int halo = pl.halo, goff = (int)pl.goff, n_local = (int)pl.n_local;
  std::vector<void*> args;
  std::vector<void*> ptrs;
  ptrs.reserve(st.in_bufs.size() + 1);
  alignas(16) char scalar_store[512];
  if (st.scalars_bytes > sizeof scalar_store || st.scalars.size() * 8 > sizeof scalar_store)
    throw Error(SF_ERR_UNSUPPORTED, "too many scalar inputs for one launch");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (pl.profile) {
    SF_HIP_CHECK(hipEventCreate(&e0));
    SF_HIP_CHECK(hipEventCreate(&e1));
    SF_HIP_CHECK(hipEventRecord(e0, stream));
  }
  if (st.star) {
    const StarCfg& c = st.cfg;
    ptrs.push_back(pl.buffers[st.in_bufs[0]].d);
    ptrs.push_back(pl.buffers[st.out_buf].d);
    if ((int)st.in_bufs.size() - 1 != st.num_aux)
      throw Error(SF_ERR_STATE, "star launch: auxiliary buffers do not match the generated kernel");
    std::memset(scalar_store, 0, sizeof scalar_store);
    for (size_t s = 0; s < st.scalars.size(); ++s) {
      const Scalar& sc = P.scalars[st.scalars[s]];
      store_scalar(scalar_store + st.scalar_offsets[s], sc.dt, pl.scalar_values[sc.input_index]);
    }
    // chunking of the stream axis: whole block waves (star_chunk_planes)
    const int range1 = std::max(0, i_end - i_begin), range2 = second ? i_end2 - i_begin2 : 0;
    const int tiles = c.NJT * c.NKT;
    const long long li = star_chunk_length(pl, c, P.kernels[st.kernels[0]].dt, std::max(range1, range2),
                                           (range1 > 0 && range2 > 0) ? 2 : 1);
    int nch1 = (int)((range1 + li - 1) / li);
    const int nch2 = (int)((range2 + li - 1) / li);
    int li_i = (int)li;
    // auxiliary field pointers (argument order = in_bufs[1..], as gen_star numbers them)
    void* aux_ptrs[kMaxStarAux] = {nullptr, nullptr, nullptr, nullptr};
    for (size_t a = 1; a < st.in_bufs.size() && a <= (size_t)kMaxStarAux; ++a)
      aux_ptrs[a - 1] = pl.buffers[st.in_bufs[a]].d;
    args = {&ptrs[0], &ptrs[1], scalar_store, aux_ptrs, &halo, &goff, &i_begin, &i_end, &li_i, &nch1, &i_begin2, &i_end2};
    // a DAG group that materialises several fields: the further output pointers (kernels/star3d.h: sf_more_outs)
    void* more_outs[4] = {nullptr, nullptr, nullptr, nullptr};
    if (st.out_bufs.size() > 1) {
      if (st.out_bufs.size() > 5) throw Error(SF_ERR_STATE, "star launch: too many outputs");
      for (size_t o = 1; o < st.out_bufs.size(); ++o) more_outs[o - 1] = pl.buffers[st.out_bufs[o]].d;
      args.push_back(more_outs);
    }
    SF_HIP_CHECK(hipModuleLaunchKernel(ck.fn, (unsigned)(tiles * (nch1 + nch2)), 1, 1, c.BX, c.BY, 1, 0,
                                       stream, args.data(), nullptr));
  } else {
    for (int b : st.in_bufs) ptrs.push_back(pl.buffers[b].d);
    ptrs.push_back(pl.buffers[st.out_buf].d);
    for (auto& p : ptrs) args.push_back(&p);
    size_t off = 0;
    for (size_t s = 0; s < st.scalars.size(); ++s) {
      const Scalar& sc = P.scalars[st.scalars[s]];
      store_scalar(scalar_store + off, sc.dt, pl.scalar_values[sc.input_index]);
      args.push_back(scalar_store + off);
      off += 8;
    }
    args.push_back(&n_local);
    args.push_back(&halo);
    args.push_back(&goff);
    args.push_back(&i_begin);
    args.push_back(&i_end);
    const long long plane = P.n[1] * (P.n[2] / st.generic_vk);
    const unsigned gx = (unsigned)((plane + 255) / 256);
    const int ranges[2][2] = {{i_begin, i_end}, {i_begin2, i_end2}};
    for (int ri = 0; ri < 2; ++ri) {
      int done = ranges[ri][0];
      const int stop = ranges[ri][1];
      while (done < stop) {  // gridDim.y is limited to 65535
        int chunk_end = std::min(stop, done + 65535);
        int cb = done, ce = chunk_end;
        args[args.size() - 2] = &cb;
        args[args.size() - 1] = &ce;
        const unsigned gy = (unsigned)((ce - cb + st.generic_ppt - 1) / st.generic_ppt);
        SF_HIP_CHECK(hipModuleLaunchKernel(ck.fn, gx, gy, 1, 256, 1, 1, 0, stream,
                                           args.data(), nullptr));
        done = chunk_end;
      }
    }
  }
  if (pl.profile) {
    SF_HIP_CHECK(hipEventRecord(e1, stream));
    pl.prof_events.push_back({e0, e1});
    pl.prof_kernel.push_back(st.ck);
    pl.prof_planes.push_back(std::max(0, i_end - i_begin) + (second ? i_end2 - i_begin2 : 0));
  }
}

void collect_profile(sf_plan& pl) {
  for (size_t i = 0; i < pl.prof_events.size(); ++i) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, pl.prof_events[i].first, pl.prof_events[i].second);
    pl.kernels[pl.prof_kernel[i]].launches += 1;
    pl.kernels[pl.prof_kernel[i]].total_ms += ms;
    pl.kernels[pl.prof_kernel[i]].planes_launched += pl.prof_planes[i];
    if (pl.kernels[pl.prof_kernel[i]].launch_ms.size() < 16384) pl.kernels[pl.prof_kernel[i]].launch_ms.push_back(ms);
    (void)hipEventDestroy(pl.prof_events[i].first);
    (void)hipEventDestroy(pl.prof_events[i].second);
  }
  pl.prof_events.clear();
  pl.prof_kernel.clear();
  pl.prof_planes.clear();
}

void upload(sf_plan& pl, const void* const* host_inputs) {
  ensure_device(pl);
  for (int i = 0; i < pl.P.num_inputs; ++i) {
    if (!host_inputs || !host_inputs[i]) throw Error(SF_ERR_INVALID, "null input array");
    Buffer& b = pl.buffers[pl.input_buf[i]];
    if (b.slabbed) {
      SF_HIP_CHECK(hipMemcpyAsync((char*)b.d + (size_t)pl.halo * b.plane_bytes, host_inputs[i],
                                  b.plane_bytes * (size_t)pl.n_local, hipMemcpyHostToDevice, pl.stream));
    } else {
      SF_HIP_CHECK(hipMemcpyAsync(b.d, host_inputs[i], b.bytes(), hipMemcpyHostToDevice, pl.stream));
    }
  }
  SF_HIP_CHECK(hipStreamSynchronize(pl.stream));
}

void download(sf_plan& pl, void* const* host_outputs) {
  ensure_device(pl);
  for (int i = 0; i < pl.P.num_outputs; ++i) {
    if (!host_outputs || !host_outputs[i]) throw Error(SF_ERR_INVALID, "null output array");
    Buffer& b = pl.buffers[pl.output_buf[i]];
    if (b.slabbed) {
      SF_HIP_CHECK(hipMemcpyAsync(host_outputs[i], (char*)b.d + (size_t)pl.halo * b.plane_bytes,
                                  b.plane_bytes * (size_t)pl.n_local, hipMemcpyDeviceToHost, pl.stream));
    } else {
      SF_HIP_CHECK(hipMemcpyAsync(host_outputs[i], b.d, b.bytes(), hipMemcpyDeviceToHost, pl.stream));
    }
  }
  SF_HIP_CHECK(hipStreamSynchronize(pl.stream));
}

// Option autotune=<k>: before the first launch, time the first k clean tile shapes
// of every fused group on the device (one warm-up and two timed launches each, on
// the plan's own buffers: inputs are only read, everything written is written
// again by the execution that follows) and keep the fastest.  Results do not
// depend on the tile shape (tools/config_fuzz.py), only the time does.
void autotune(sf_plan& pl) {
  if (pl.autotuned) return;
  if (pl.opt.get("autotune", 0) <= 1) {
    pl.autotuned = true;
    return;
  }
  if (pl.P.num_scalar_inputs > 0 && !pl.scalars_set) return;  // not yet launchable
  pl.autotuned = true;
  const bool profile = pl.profile;
  pl.profile = false;
  std::map<std::string, std::pair<StarCfg, int>> best;
  std::ostringstream note;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  SF_HIP_CHECK(hipEventCreate(&e0));
  SF_HIP_CHECK(hipEventCreate(&e1));
  try {
    for (auto& st : pl.steps) {
      if (!st.star || st.alts.size() < 2) continue;
      if (!best.count(st.sig)) {
        double best_ms = 1e30;
        note << "  autotune";
        for (auto& alt : st.alts) {
          Step probe = st;
          probe.cfg = alt.first;
          probe.ck = alt.second;
          // warm-up launch, timed to size the measurement: about 3 ms of launches,
          // at least 3 (short launches are noisy), best of two rounds
          launch_step(pl, probe, 0, pl.stream);
          SF_HIP_CHECK(hipEventRecord(e0, pl.stream));
          launch_step(pl, probe, 0, pl.stream);
          SF_HIP_CHECK(hipEventRecord(e1, pl.stream));
          SF_HIP_CHECK(hipEventSynchronize(e1));
          float one = 0;
          SF_HIP_CHECK(hipEventElapsedTime(&one, e0, e1));
          const int reps = (int)std::min(100.0, std::max(3.0, 3.0 / std::max(1e-3, (double)one)));
          float ms = 1e30f;
          for (int round = 0; round < 2; ++round) {
            SF_HIP_CHECK(hipEventRecord(e0, pl.stream));
            for (int i = 0; i < reps; ++i) launch_step(pl, probe, 0, pl.stream);
            SF_HIP_CHECK(hipEventRecord(e1, pl.stream));
            SF_HIP_CHECK(hipEventSynchronize(e1));
            float t = 0;
            SF_HIP_CHECK(hipEventElapsedTime(&t, e0, e1));
            ms = std::min(ms, t / (float)reps);
          }
          note << " " << pl.kernels[alt.second].name << " [" << alt.first.BX << "x" << alt.first.BY << " rows "
               << alt.first.RJ << "] " << ms << " ms;";
          if (ms < best_ms) {
            best_ms = ms;
            best[st.sig] = alt;
          }
        }
        note << " -> " << pl.kernels[best[st.sig].second].name << "\n";
      }
      const auto& pick = best[st.sig];
      if (pick.second != st.ck) {
        pl.kernels[pick.second].updates_per_launch = pl.kernels[st.ck].updates_per_launch;
        pl.kernels[pick.second].alg_bytes_per_launch = pl.kernels[st.ck].alg_bytes_per_launch;
      }
      st.cfg = pick.first;
      st.ck = pick.second;
    }
  } catch (...) {
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    pl.profile = profile;
    throw;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  pl.profile = profile;
  pl.description = describe_plan(pl) + note.str();  // the launches as they run now, then the timings
}

void execute(sf_plan& pl, int repetitions) {
  ensure_device(pl);
  autotune(pl);
  if (repetitions < 0) throw Error(SF_ERR_INVALID, "negative repetition count");
  for (auto& k : pl.kernels) {
    k.launches = 0;
    k.total_ms = 0;
    k.planes_launched = 0;
  }
  // Launch-bound chains (many launches of a few microseconds each: small grids)
  // are replayed as one hipGraph, which removes the per-launch host cost; big
  // grids keep plain stream launches (nothing to gain, 213 us per launch on C3).
  // graph=0|1 forces the choice.
  const bool small = pl.max_updates_per_launch > 0 && pl.max_updates_per_launch < 8.0e6;
  const bool want_graph = pl.opt.get("graph", (small && pl.steps.size() >= 4) ? 1 : 0) != 0;
  if (want_graph && !pl.profile && repetitions > 0) {
    if (!pl.chain_graph || pl.chain_graph_scalars != pl.scalar_values) {
      if (pl.chain_graph) {
        (void)hipGraphExecDestroy(pl.chain_graph);
        pl.chain_graph = nullptr;
      }
      hipGraph_t graph = nullptr;
      SF_HIP_CHECK(hipStreamBeginCapture(pl.stream, hipStreamCaptureModeThreadLocal));
      try {
        for (auto& st : pl.steps) launch_step(pl, st, 0, pl.stream);
      } catch (...) {
        (void)hipStreamEndCapture(pl.stream, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
      }
      SF_HIP_CHECK(hipStreamEndCapture(pl.stream, &graph));
      const hipError_t inst = hipGraphInstantiate(&pl.chain_graph, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (inst != hipSuccess) {
        pl.chain_graph = nullptr;
        throw Error(SF_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(inst));
      }
      pl.chain_graph_scalars = pl.scalar_values;
    }
    SF_HIP_CHECK(hipEventRecord(pl.ev_begin, pl.stream));
    for (int r = 0; r < repetitions; ++r) SF_HIP_CHECK(hipGraphLaunch(pl.chain_graph, pl.stream));
    SF_HIP_CHECK(hipEventRecord(pl.ev_end, pl.stream));
    pl.timed = true;
    return;
  }
  SF_HIP_CHECK(hipEventRecord(pl.ev_begin, pl.stream));
  for (int r = 0; r < repetitions; ++r)
    for (auto& st : pl.steps) launch_step(pl, st, 0, pl.stream);
  SF_HIP_CHECK(hipEventRecord(pl.ev_end, pl.stream));
  pl.timed = true;
}

// ---------------------------------------------------------------- plan-time self-check
// The second guard against a wrong code object (the first is the EXEC-restore detector of
// codecache.cpp, a heuristic over the machine code): every FUSED kernel (star3d.h /
// compact3d.h) that has no verdict yet is run once, before the plan's first use, on seeded
// data over the planes next to both ends of the slab, and its result is compared bit for
// bit with the same operators evaluated one by one by the plain generic kernel
// (gen_generic: one point per thread, every access guarded as the reference's tasklet
// guards it, stencilflow/stencil/cpu.py:71-102) -- two independent product kernels, no
// oracle involved.  The verdict travels with the code object through both cache levels
// (cache files "SFCO0003"), so only the first plan for a shape pays (one hipRTC
// compilation per distinct operator, microseconds of launches).  A kernel that differs is
// never launched: the plan's first use fails with SF_ERR_UNSUPPORTED, and the planner
// skips the object from then on (kernel_unsafe), taking the next tile shape.
// $SF_HIP_SELF_CHECK=0 turns the check off.
namespace {

__device__ __forceinline__ unsigned sf_check_hash(unsigned long long i, unsigned seed) {
  unsigned long long x = (i + 0x9E3779B97F4A7C15ull) * (0xBF58476D1CE4E5B9ull + 2ull * seed);
  x ^= x >> 29;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 32;
  return (unsigned)x;
}
// dt: 0 f32, 1 f64, 2 i32, 3 i64 -- values in [-1, 1) (integers: -8 .. 7)
__global__ void sf_check_fill(void* p, unsigned long long n, unsigned seed, int dt) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned h = sf_check_hash(i, seed);
    if (dt == 0) static_cast<float*>(p)[i] = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    else if (dt == 1) static_cast<double*>(p)[i] = (double)h * (1.0 / 2147483648.0) - 1.0;
    else if (dt == 2) static_cast<int*>(p)[i] = (int)(h & 15u) - 8;
    else static_cast<long long*>(p)[i] = (long long)(h & 15u) - 8;
  }
}
// words that differ (tol == 0: bit for bit; else floats within tol relative, for groups with
// device math calls whose last bits are not promised to agree between two kernels)
__global__ void sf_check_diff(const void* a, const void* b, unsigned long long n, int dt, double tol, unsigned* count) {
  unsigned bad = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    if (dt == 0 || dt == 2) {
      const unsigned x = static_cast<const unsigned*>(a)[i], y = static_cast<const unsigned*>(b)[i];
      if (x == y) continue;
      if (tol > 0 && dt == 0) {
        const float fx = __uint_as_float(x), fy = __uint_as_float(y);
        if (fabsf(fx - fy) <= (float)tol * fmaxf(fabsf(fx), fabsf(fy))) continue;
      }
      ++bad;
    } else {
      const unsigned long long x = static_cast<const unsigned long long*>(a)[i],
                               y = static_cast<const unsigned long long*>(b)[i];
      if (x == y) continue;
      if (tol > 0 && dt == 1) {
        const double fx = __longlong_as_double((long long)x), fy = __longlong_as_double((long long)y);
        if (fabs(fx - fy) <= tol * 1e-6 * fmax(fabs(fx), fabs(fy))) continue;
      }
      ++bad;
    }
  }
  if (bad) atomicAdd(count, bad);
}

std::atomic<long> g_self_checks{0};

int dt_code(DT dt) { return dt == DT::F32 ? 0 : dt == DT::F64 ? 1 : dt == DT::I32 ? 2 : 3; }

bool calls_device_math(const Kernel& K) {
  static const char* const names[] = {"sin(", "cos(", "tan(", "sinh(", "cosh(", "tanh(", "exp(", "log(", "pow(", "sqrt(",
                                      "sinf(", "cosf(", "tanf(", "sinhf(", "coshf(", "tanhf(", "expf(", "logf(", "powf(", "sqrtf("};
  std::string text = K.ret;
  for (auto& l : K.lets) text += " " + l.expr;
  for (const char* n : names)
    if (text.find(n) != std::string::npos) return true;
  return false;
}

}  // namespace

long self_checks_run() { return g_self_checks.load(); }

// Compare the fused launch `st` with its operators run one by one, over planes [b, e).  The operators of the group
// are a DAG in stage order (a chain: each reads the one before): an operator reads the group's input buffers or the
// scratch results of earlier operators, and every field the launch materialises is compared.
static unsigned long long self_check_range(sf_plan& pl, const Step& st, std::vector<CompiledKernel*>& refs,
                                           const std::vector<GenericKernelSource>& gens, int b, int e) {
  const Program& P = pl.P;
  const int T = (int)st.kernels.size();
  const DT dt = P.kernels[st.kernels[0]].dt;
  const int n = (int)pl.n_local, halo = pl.halo, goff = (int)pl.goff;
  const Buffer& primary = pl.buffers[st.in_bufs[0]];
  const size_t plane_bytes = primary.plane_bytes;
  // planes a launch may touch: inside the global domain and inside the buffers
  const int q_lo = std::max(-halo, -goff), q_hi = std::min(n + halo, (int)(P.n[0] - goff));
  // producer (position in the group) of every field made inside the group
  std::map<std::string, int> made;
  for (int s = 0; s < T; ++s) made[P.kernels[st.kernels[s]].name] = s;
  // need[s]: how many planes beyond [b, e) operator s must produce for the operators that read it
  std::vector<int> reach(T, 0), need(T, 0);
  for (int s = 0; s < T; ++s)
    for (auto& a : P.kernels[st.kernels[s]].acc) reach[s] = std::max(reach[s], std::abs(a.off[0]));
  int input_need = 0;
  for (int s = T - 1; s >= 0; --s) {
    bool reads_memory = false;
    for (auto& a : P.kernels[st.kernels[s]].acc) {
      auto it = made.find(a.field);
      if (it != made.end() && it->second < s) need[it->second] = std::max(need[it->second], need[s] + reach[s]);
      else reads_memory = true;
    }
    if (reads_memory) input_need = std::max(input_need, need[s] + reach[s]);
  }
  const int lo_s = std::max(q_lo, b - input_need), hi_s = std::min(q_hi, e + input_need);
  std::vector<void*> scratch(T, nullptr);
  unsigned* d_count = nullptr;
  unsigned long long bad = 0;
  try {
    for (int s = 0; s < T; ++s) SF_HIP_CHECK(hipMalloc(&scratch[s], (size_t)(hi_s - lo_s) * plane_bytes));
    SF_HIP_CHECK(hipMalloc((void**)&d_count, sizeof(unsigned)));
    SF_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(unsigned), pl.stream));
    // the fused launch
    launch_ranges(pl, st, b, e, 0, 0, pl.stream);
    // the operators one by one, each over the planes the later ones need of it
    for (int s = 0; s < T; ++s) {
      const Kernel& K = P.kernels[st.kernels[s]];
      const GenericKernelSource& g = gens[s];
      const int rb = std::max(q_lo, b - need[s]), re = std::min(q_hi, e + need[s]);
      // (a scratch buffer holds planes [lo_s, hi_s): the kernels index planes from the start of a slab
      // buffer, so they are handed the address plane -halo would have)
      auto virtual_base = [&](void* p) { return (void*)((char*)p - (long long)(lo_s + halo) * (long long)plane_bytes); };
      std::vector<void*> ptrs;
      for (auto& name : g.reads) {
        void* p = nullptr;
        auto it = made.find(name);
        if (it != made.end() && it->second < s) {
          p = virtual_base(scratch[it->second]);
        } else {
          for (size_t r = 0; r < st.read_names.size(); ++r)
            if (st.read_names[r] == name) p = pl.buffers[st.in_bufs[r]].d;
        }
        if (!p) throw Error(SF_ERR_STATE, "self-check: operator '" + K.name + "' reads '" + name + "', which the launch does not");
        ptrs.push_back(p);
      }
      ptrs.push_back(virtual_base(scratch[s]));
      std::vector<void*> args;
      for (auto& p : ptrs) args.push_back(&p);
      alignas(16) char scalar_store[512];
      if (g.scalars.size() * 8 > sizeof scalar_store) throw Error(SF_ERR_UNSUPPORTED, "too many scalar inputs for one launch");
      size_t off = 0;
      for (int sc_ix : g.scalars) {
        const Scalar& sc = P.scalars[sc_ix];
        const double v = pl.scalar_values[sc.input_index];
        switch (sc.dt) {
          case DT::F32: { float x = (float)v; std::memcpy(scalar_store + off, &x, 4); break; }
          case DT::F64: { std::memcpy(scalar_store + off, &v, 8); break; }
          case DT::I32: { int x = (int)v; std::memcpy(scalar_store + off, &x, 4); break; }
          default: { long long x = (long long)v; std::memcpy(scalar_store + off, &x, 8); break; }
        }
        args.push_back(scalar_store + off);
        off += 8;
      }
      int a_n = n, a_halo = halo, a_goff = goff, a_b = rb, a_e = re;
      args.push_back(&a_n);
      args.push_back(&a_halo);
      args.push_back(&a_goff);
      args.push_back(&a_b);
      args.push_back(&a_e);
      const long long plane = P.n[1] * P.n[2];
      if (re > rb)
        SF_HIP_CHECK(hipModuleLaunchKernel(refs[s]->fn, (unsigned)((plane + 255) / 256), (unsigned)(re - rb), 1, 256, 1, 1, 0,
                                           pl.stream, args.data(), nullptr));
    }
    const unsigned long long words = (unsigned long long)(e - b) * plane_bytes / size_of(dt);
    bool math = false;
    for (int k : st.kernels) math = math || calls_device_math(P.kernels[k]);
    // every materialised field against the scratch result of the operator that makes it
    for (size_t o = 0; o < st.out_bufs.size(); ++o) {
      const Buffer& out = pl.buffers[st.out_bufs[o]];
      const int maker = o < st.out_names.size() && made.count(st.out_names[o]) ? made[st.out_names[o]] : T - 1;
      hipLaunchKernelGGL(sf_check_diff, dim3(1024), dim3(256), 0, pl.stream,
                         (const void*)((char*)out.d + (size_t)(b + halo) * plane_bytes),
                         (const void*)((char*)scratch[maker] + (size_t)(b - lo_s) * plane_bytes), words, dt_code(dt),
                         math ? 1e-6 : 0.0, d_count);
      SF_HIP_CHECK(hipGetLastError());
    }
    unsigned h_count = 0;
    SF_HIP_CHECK(hipMemcpyAsync(&h_count, d_count, sizeof h_count, hipMemcpyDeviceToHost, pl.stream));
    SF_HIP_CHECK(hipStreamSynchronize(pl.stream));
    bad = h_count;
  } catch (...) {
    (void)hipStreamSynchronize(pl.stream);
    for (void* p : scratch)
      if (p) (void)hipFree(p);
    if (d_count) (void)hipFree(d_count);
    throw;
  }
  for (void* p : scratch) (void)hipFree(p);
  (void)hipFree(d_count);
  return bad;
}

void self_check(sf_plan& pl) {
  if (const char* env = std::getenv("SF_HIP_SELF_CHECK"))
    if (std::string(env) == "0") return;
  // fused launches whose code object carries no verdict yet (one per distinct object)
  std::vector<Step> todo;
  std::set<int> seen;
  for (const Step& st : pl.steps) {
    if (!st.star) continue;
    auto consider = [&](const StarCfg& cfg, int ck) {
      if (pl.kernels[ck].verdict != 0 || !seen.insert(ck).second) return;
      Step probe = st;
      probe.cfg = cfg;
      probe.ck = ck;
      todo.push_back(probe);
    };
    consider(st.cfg, st.ck);
    for (auto& alt : st.alts) consider(alt.first, alt.second);
  }
  if (todo.empty()) return;
  const Program& P = pl.P;
  const int n = (int)pl.n_local;
  // the state the check borrows: profiling off, seeded scalars, seeded fields
  const bool profile = pl.profile, scalars_set = pl.scalars_set;
  const std::vector<double> scalars = pl.scalar_values;
  const int reserved = pl.reserved_cus;
  pl.profile = false;
  pl.reserved_cus = 0;
  for (size_t i = 0; i < pl.scalar_values.size(); ++i) pl.scalar_values[i] = 0.0625 * (double)(i + 1) - 0.28125;
  pl.scalars_set = true;
  std::set<int> dirty;
  std::string failed;
  try {
    for (Step& st : todo) {
      // reference operators: compiled on demand, loaded beside the plan's kernels
      std::vector<GenericKernelSource> gens;
      std::vector<CompiledKernel*> refs;
      const size_t first = pl.check_kernels.size();
      for (int k : st.kernels) {
        gens.push_back(gen_generic(P, k, false, false));
        pl.check_kernels.push_back(compile_cached(std::string("sf_check_") + short_of(P.kernels[k].dt), gens.back().source, ""));
      }
      for (size_t i = first; i < pl.check_kernels.size(); ++i) {
        CompiledKernel& k = pl.check_kernels[i];
        if (kernel_unsafe(k)) throw Error(SF_ERR_UNSUPPORTED, "self-check: the reference kernel itself is flagged");
        SF_HIP_CHECK(hipModuleLoadData(&k.mod, k.code.data()));
        SF_HIP_CHECK(hipModuleGetFunction(&k.fn, k.mod, k.name.c_str()));
      }
      for (size_t i = first; i < pl.check_kernels.size(); ++i) refs.push_back(&pl.check_kernels[i]);
      // seeded data in everything the launch reads (whole buffers: ghost planes included)
      for (size_t r = 0; r < st.in_bufs.size(); ++r) {
        Buffer& buf = pl.buffers[st.in_bufs[r]];
        const unsigned long long elems = buf.bytes() / size_of(buf.dt);
        hipLaunchKernelGGL(sf_check_fill, dim3(1024), dim3(256), 0, pl.stream, buf.d, elems, (unsigned)(17 + st.in_bufs[r]),
                           dt_code(buf.dt));
        SF_HIP_CHECK(hipGetLastError());
        dirty.insert(st.in_bufs[r]);
      }
      for (int ob : st.out_bufs) dirty.insert(ob);
      // planes next to both ends of the slab (all of it when it is thin)
      const int span = 6 + 2 * st.halo_depth;
      unsigned long long bad = 0;
      if (n <= 2 * span) {
        bad = self_check_range(pl, st, refs, gens, 0, n);
      } else {
        bad = self_check_range(pl, st, refs, gens, 0, span);
        bad += self_check_range(pl, st, refs, gens, n - span, n);
      }
      if (bad != 0) {
        // a mismatch is recorded for good (the tile shape is skipped from then on): only if a second run of the
        // comparison reproduces it
        unsigned long long again = 0;
        if (n <= 2 * span) {
          again = self_check_range(pl, st, refs, gens, 0, n);
        } else {
          again = self_check_range(pl, st, refs, gens, 0, span);
          again += self_check_range(pl, st, refs, gens, n - span, n);
        }
        if (again == 0) {
          //  a mismatch that does not reproduce is the signature of a race inside the fused kernel -- an LDS
          // ring or a counted wait one short -- at least as much as of a disturbed box: the code object is refused for
          // the rest of THIS process (verdict kept in memory only: the next process checks it afresh, and a pass can
          // then be recorded only by a run that sees no mismatch at all), and the plan is poisoned like after a
          // reproducing mismatch.
          record_verdict(pl.kernels[st.ck], 2, /*persist=*/false);
          if (failed.empty())
            failed = pl.kernels[st.ck].name + " (" + std::to_string(bad) + " results differed from the operators run one by one in the planes [0, " +
                     std::to_string(std::min(n, span)) + ") and [" + std::to_string(std::max(0, n - span)) + ", " + std::to_string(n) +
                     ") of seeded data, and none when the comparison was repeated: a race is suspected, nothing written to the disk cache)";
          ++g_self_checks;
          continue;
        }
        bad = again;
      }
      ++g_self_checks;
      record_verdict(pl.kernels[st.ck], bad == 0 ? 1 : 2);
      if (bad != 0 && failed.empty())
        failed = pl.kernels[st.ck].name + " (" + std::to_string(bad) + " results differ from the operators run one by one)";
    }
  } catch (...) {
    pl.profile = profile;
    pl.reserved_cus = reserved;
    pl.scalar_values = scalars;
    pl.scalars_set = scalars_set;
    for (int b : dirty) (void)hipMemsetAsync(pl.buffers[b].d, 0, pl.buffers[b].bytes(), pl.stream);
    (void)hipStreamSynchronize(pl.stream);
    throw;
  }
  pl.profile = profile;
  pl.reserved_cus = reserved;
  pl.scalar_values = scalars;
  pl.scalars_set = scalars_set;
  // the buffers as ensure_device left them
  for (int b : dirty) SF_HIP_CHECK(hipMemsetAsync(pl.buffers[b].d, 0, pl.buffers[b].bytes(), pl.stream));
  SF_HIP_CHECK(hipStreamSynchronize(pl.stream));
  if (!failed.empty()) {
    pl.poisoned = "plan-time self-check: the fused kernel " + failed +
                  "; the code object is marked and will not be used again -- create the plan once more (the planner "
                  "takes the next tile shape)";
    throw Error(SF_ERR_UNSUPPORTED, pl.poisoned);
  }
}

}  // namespace sf
