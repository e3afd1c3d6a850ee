// exec.cpp — run time of a plan: device state, kernel launches over plane ranges,
// per-launch profiling, plan-time autotuning, the chain as a hipGraph.
// (Role in the reference: program(**dace_args), stencilflow/run_program.py:164-178.)
#include "sf_internal.hpp"

#include <algorithm>
#include <cstring>
#include <sstream>

namespace sf {

// ---------------------------------------------------------------- runtime
void ensure_device(sf_plan& pl) {
  if (pl.device_ready) {
    // every entry point runs on the plan's device, whatever device the calling
    // thread used last (one thread may drive plans on several GPUs)
    SF_HIP_CHECK(hipSetDevice(pl.device));
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
  SF_HIP_CHECK(hipMalloc(&pl.debug_buffer, 64));
  SF_HIP_CHECK(hipMemsetAsync(pl.debug_buffer, 0, 64, pl.stream));
  SF_HIP_CHECK(hipStreamSynchronize(pl.stream));
  pl.device_ready = true;
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
    if (c.stamp) args.push_back(&pl.debug_buffer);
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
  bool stamp = false;
  for (auto& st : pl.steps) stamp = stamp || (st.star && st.cfg.stamp);
  if (want_graph && !pl.profile && !stamp && repetitions > 0) {
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

}  // namespace sf
