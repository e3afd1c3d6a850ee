// planner.cpp — from a parsed program to launches: grouping of operators into fused
// launches, tile-shape search for the plane-streaming kernels (compile candidates in
// order of modelled cost, read the code objects' metadata), liveness-based device
// buffers.  (The reference's counterpart is generate_sdfg / generate_reference,
// stencilflow/sdfg_generator.py:219-677.)
#include "sf_internal.hpp"

#include <algorithm>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

namespace sf {

// ---------------------------------------------------------------- planner
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// A candidate the compiler rejects is skipped and the planner moves on -- to the next tile shape, in the end to another
// kernel family or the generic kernel.  That is the right thing for a shape that does not fit (registers, LDS); for an
// ERROR in the generated source it would hide a broken kernel family behind a slower plan (round 4: one wrong identifier
// in compact3d.h and the 27-point box ran on the dense kernel, bit-exact and 30 % slower).  So anything that does not
// read like a resource limit is reported on stderr, once per kernel family and process, whatever `debug` says.
static void report_rejected_candidate(const sf_plan& pl, const char* family, const std::string& what, const Error& e) {
  const std::string msg = e.what();
  const bool resources = msg.find("ran out of registers") != std::string::npos || msg.find("local memory") != std::string::npos ||
                         msg.find("LDS size") != std::string::npos || msg.find("exceeds limit") != std::string::npos;
  if (pl.opt.get("debug", 0) != 0)
    std::fprintf(stderr, "[sf_hip] %s candidate %s rejected by the compiler: %.400s\n", family, what.c_str(), msg.c_str());
  if (resources) return;
  static std::mutex told_mutex;  // (plans are created from several threads)
  static std::set<std::string> told;
  {
    std::lock_guard<std::mutex> lock(told_mutex);
    if (!told.insert(family).second) return;
  }
  const size_t at = msg.find("error:");
  std::fprintf(stderr, "[sf_hip] warning: a %s kernel failed to COMPILE and the planner falls back to other kernels: %.300s\n",
               family, at == std::string::npos ? msg.c_str() : msg.c_str() + at);
}

size_t star_lds_bytes(const StarCfg& c, DT dt) {
  if (c.dense) return c.lds_bytes;  // kernels/dense3d.h (select_dense computed it)
  if (c.R == 2) {
    // kernels/wstar3d.h: two images, each with the first / last two rows of every thread row per
    // window and four edge words per row and wave (one virtual wave at either end)
    const size_t rows = c.noj ? 0 : (size_t)c.T * c.BY * 4 * c.BX * c.VK;
    const size_t edge = (size_t)c.T * c.BY * c.RJ * (c.BX / 64 + 2) * 4;
    return 2 * std::max<size_t>(1, rows + edge) * size_of(dt);
  }
  if (c.compact) {
    // kernels/compact3d.h: per window a ring of images (first / last row of every
    // thread row + the wave-edge columns of every row incl. two virtual waves)
    const size_t win = (size_t)c.BY * 2 * (c.BX * c.VK) + (size_t)(c.BY + 2) * c.RJ * (c.BX / 64 + 2) * 2;
    return std::max<size_t>(1, (size_t)c.lds_images * win) * size_of(dt);
  }
  const size_t windows = c.dag.on ? (size_t)c.dag.nwin : (size_t)c.T;  // (kernels/star3d.h: SF_NW)
  const size_t rows = c.noj ? 0 : windows * c.BY * 2 * c.BX * c.VK;
  const size_t edge = windows * c.BY * c.RJ * (c.BX / 64 + (c.BX > 64 ? 2 : 0)) * 2;  // (two virtual waves beside a row)
  return (rows + edge) * size_of(dt) * 2;  // (two images alternate from step to step)
}

// ---- launch-geometry model ------------------------------------------------------
// Measured on MI355X (profiles/r01_sweep_*.log): for a given fused depth the raw
// update rate of the star kernel is nearly independent of the tile shape; what
// separates configurations is (a) redundant halo work in j / k / along the
// stream axis and (b) how evenly the blocks fill the 256 CUs.  The planner
// therefore minimises
//   cost = (tile rows / interior rows) * (tile cols / interior cols)
//          * (chunk planes + 2T) / chunk planes * (block slots used / blocks)
// over the tile shapes whose register footprint fits without spilling.
static int star_regs_estimate(const StarCfg& c, DT dt) {
  // window + staging registers, plus what the compiler needs around them: fitted
  // to the code objects of this round (f32 T=2 P=20: 226; f64 T=3 P=6/8/10:
  // 142/188/232) as 20 + 3.3 P.  It only has to be roughly right -- a shape that
  // spills after all is rejected by select_star from its metadata.
  const int words = (dt == DT::F64) ? 2 : 1;
  const int P = c.RJ * c.VK;
  if (c.R == 2)  // five planes per window; fitted to the code objects of round 3 (f32 T = 2: P = 16 -> 190, P = 20 -> 232)
    return 5 * c.T * P * words + 22 + P / 2;
  if (c.compact)  // three live planes per window, one more in flight per loaded window; fitted to
                  // the code objects of round 2 (box, T = 2: P = 16 -> 180, P = 20 -> 212)
    return (3 * c.nwin + 1 + c.nwin - c.T) * P * words + 52 + P;
  // (a DAG group: one window per field held, and an evaluation's temporaries per stage beyond the chain's)
  const int windows = c.dag.on ? c.dag.nwin : c.T;
  return 3 * windows * P * words + 20 + (33 * P) / 10 + (c.ring4 ? P * words : 0);
}

static int star_blocks_per_cu(const StarCfg& c, DT dt) {
  if (c.dense)  // LDS and wave slots decide (about 120 registers per thread)
    return std::max(1, std::min({(int)(160 * 1024 / std::max<size_t>(1, c.lds_bytes)), 32 / ((c.BX * c.BY + 63) / 64), 4}));
  const int threads = c.BX * c.BY;
  const int waves_per_simd = (threads + 255) / 256;  // a block's waves on one SIMD
  const int regs = star_regs_estimate(c, dt);
  const int alloc = (regs + 7) / 8 * 8;
  if (alloc > 256) return 0;  // would lean on AGPR / scratch spills
  const int by_regs = (512 / alloc) / waves_per_simd;
  const size_t lds = std::max<size_t>(star_lds_bytes(c, dt), 1);
  const int by_lds = (int)(160 * 1024 / lds);
  const int by_waves = 32 / ((threads + 63) / 64);
  return std::max(0, std::min(std::min(by_regs, by_lds), std::min(by_waves, 8)));
}

static void star_finish_cfg(StarCfg& c, const Program& P, int T) {
  const long long tkh = (long long)c.BX * c.VK;
  const int hk = round_up(T * c.R, c.VK);
  c.ktiled = (tkh != P.n[2]);
  c.HK = c.ktiled ? hk : 0;
  c.NKT = c.ktiled ? (int)((P.n[2] + (tkh - 2 * c.HK) - 1) / (tkh - 2 * c.HK)) : 1;
  // wide stars: the k-tiles share the row evenly (kernels/wstar3d.h: SF_TKI)
  c.TKI = (c.R == 2 && c.ktiled) ? std::min<int>((int)(tkh - 2 * c.HK), round_up((int)((P.n[2] + c.NKT - 1) / c.NKT), c.VK)) : 0;
  if (c.noj) {
    c.NJT = 1;
  } else {
    const int tji = c.BY * c.RJ - 2 * T * c.R;
    if (tji < 1) throw Error(SF_ERR_INVALID, "star kernel: tile has no interior rows (raise k1.by / k1.rj)");
    c.NJT = (int)((P.n[1] + tji - 1) / tji);
  }
}

// chunk length along the stream axis for `range` planes: whole block waves
static int star_chunk_planes(const StarCfg& c, DT dt, int range, double* cost_out = nullptr,
                             int reserved_cus = 0, int ranges = 1) {
  // (`ranges`: plane ranges of this length served by the one launch -- the two
  // slab boundaries of a split step -- whose blocks share the block slots)
  const int tiles = c.NJT * c.NKT * std::max(1, ranges);
  const int slots = std::max(1, 256 - reserved_cus) * std::max(1, star_blocks_per_cu(c, dt));
  double best = 1e30;
  int best_li = range;
  const int max_nch = std::max(1, range / std::max(1, 2 * c.T * c.R));
  for (int nch = 1; nch <= std::min(max_nch, 4096); ++nch) {
    const int li = (range + nch - 1) / nch;
    const int real_nch = (range + li - 1) / li;
    const long long blocks = (long long)tiles * real_nch;
    if (c.noj) {
      // 2-D programs: a block is one (or a few) self-contained waves with a
      // short dependent step, so the sweep is latency-bound until about three
      // waves share a SIMD (profiles/r01_sweep_10_c2_chunks.log: 4096^2 is
      // fastest at 24-row chunks = 2.8 waves per SIMD, 1.8x the rate of
      // 92-row chunks).  Below that, time follows the chunk length; above it,
      // the warm-up redundancy.
      const double waves_per_simd = (double)blocks * (double)(c.BX / 64) / 1024.0;
      const double warm = (double)(li + 2 * c.T * c.R + (c.compact ? c.T - 1 : 0)) / (double)li;
      const double cost = warm * std::max(1.0, 3.0 / waves_per_simd);
      if (cost < best - 1e-12) {
        best = cost;
        best_li = li;
      }
      if (waves_per_simd > 8.0) break;
      continue;
    }
    const long long rounds = (blocks + slots - 1) / slots;
    const double quant = (double)(rounds * slots) / (double)blocks;
    // (the compact kernel's stages read planes finished in earlier steps only: T - 1 more steps to drain)
    const double warm = (double)(li + 2 * c.T * c.R + (c.compact ? c.T - 1 : 0)) / (double)li;
    // more rounds amortise the tail when block times differ
    const double cost = warm * quant * (1.0 + 0.02 / (double)rounds);
    if (cost < best - 1e-12) {
      best = cost;
      best_li = li;
    }
    if (blocks > 64LL * slots) break;
  }
  if (cost_out) *cost_out = best;
  return best_li;
}

// Extra compiler flags of a kernel family: the SLP vectoriser is off everywhere but for the 3-D wide stars.  On gfx950 a
// v_pk_add_f32 holds the vector pipe as long as the two v_add_f32 it replaces, its operand pairs are assembled with
// moves and two dependent ones need wait states between them (an `s_nop` each): measured slower wherever it was
// tried -- 27-point box (round 2), 125-point box 505 -> 450 us per launch, stars with float-typed sums 0.5-8 %
// (round 4, profiles/r04_dense_slp.log, r04_slp_families.log); 3-D wide stars 8.8e5 against 8.0e5 without it.
static std::string slp_flags(bool slp) { return slp ? "" : "-fno-slp-vectorize"; }

// chunk length used for a launch over `range` planes (options k1.li / k2.li pin it)
long long star_chunk_length(const sf_plan& pl, const StarCfg& c, DT dt, int range, int ranges) {
  long long li = pl.opt.get(c.noj ? "k2.li" : "k1.li", 0);
  if (li <= 0) li = star_chunk_planes(c, dt, range, nullptr, pl.reserved_cus, ranges);
  if (li > range) li = range;
  return std::max<long long>(li, 1);
}

static std::vector<StarCfg> rank_star_cfgs(const sf_plan& pl, int T, DT dt, const StarCfg* proto = nullptr) {
  const Program& P = pl.P;
  StarCfg base;
  if (proto) base = *proto;  // (compact kernels: windows and LDS images of the group)
  base.T = T;
  // one 16-byte vector per row and lane: 4 floats or 2 doubles
  base.VK = (int)pl.opt.get("k1.vk", dt == DT::F64 ? 2 : 4);
  if (base.VK != 1 && base.VK != 2 && base.VK != 4) throw Error(SF_ERR_INVALID, "k1.vk must be 1, 2 or 4");
  // rows that do not hold whole 16-byte vectors: 8-byte vectors (rows of 4m+2 floats)
  // or single elements (odd rows) keep the program on the star kernel -- slower per
  // point than 16-byte vectors, several times faster than the generic kernel
  if (!pl.opt.kv.count("k1.vk"))
    while (base.VK > 1 && P.n[2] % base.VK != 0) base.VK /= 2;
  if (P.n[2] % base.VK != 0) throw Error(SF_ERR_INVALID, "innermost extent must be a multiple of k1.vk");
  if (base.R == 2 && base.VK < 2) throw Error(SF_ERR_INVALID, "wide stars need vectors of two or more elements");
  base.n0g = P.n[0];
  base.n1 = P.n[1];
  base.n2 = P.n[2];
  base.noj = (P.n[1] == 1);
  base.row_fence = 1;
  base.opaque = base.noj ? 0 : 1;  // 2-D: registers are plentiful
  // Input planes: the four-slot ring (two steps to land, no copy; the step loop is unrolled by 4) for 2-D and f32 3-D
  // -- C2 +10 %, C3 +1.7 %, hotspot chains +1 % --; f64 3-D loads straight into the window slot stage 1 has just freed
  // (the ring needs a smaller tile there; staging registers cost 20-30 registers and made C5's five-row tile spill
  // under ROCm 7.2's hipRTC, profiles/r04_compilers_c5_c3.log).  The compact kernels have their own ring.
  base.ring4 = !base.compact && (base.noj || dt == DT::F32);
  // Planes go through buffer instructions (out-of-range offsets instead of branches around loads and stores: the
  // compiler counts the memory operations in flight instead of draining them once per step; 2-D +10 %, f32 3-D chains
  // with auxiliary fields +38 %, profiles/r01_sweep_17_buffer_io.log): a plane must stay below the 2 GiB offset range
  const double plane_bytes = (double)P.n[1] * (double)P.n[2] * (double)size_of(dt);
  if (plane_bytes > 1024.0 * 1024 * 1024) throw Error(SF_ERR_INVALID, "planes of more than 1 GiB take the generic kernel");
  // auxiliary (centre-only) fields: 1 = a stage requests all its rows before its first row is evaluated (3-D hotspot
  // chains +21 %); 2 = rows are requested a whole step ahead into per-stage slots (2-D, where a thread has one row and
  // registers to spare, +11 %)
  base.aux_ahead = base.noj ? 2 : 1;
  base.aux_pass = 1;
  // Non-temporal output stores when a field is larger than the 256 MiB Infinity Cache: nothing of it would survive
  // until the next launch reads it, and not allocating the written lines leaves the cache to the input stream (C3 +3 %,
  // C5 +1.4 %; the cache-resident 64 MiB field of C2 loses 13 % with them).  Bit 4 (round 4): the rows of a tile no other
  // block reads -- neither a neighbouring tile's halo nor its source of halo rows -- are LOADED non-temporally as well,
  // the shared rows keep the default policy and meet their second reader in the XCD's L2 (C3 199.9 -> 198.1-198.6 us,
  // profiles/r04_c3_prio_nt_fork.log; all-rows non-temporal loads had lost 4-8 % in round 2)
  const double field_bytes = (double)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local) * (double)P.n[1] *
                             (double)P.n[2] * (double)size_of(dt);
  base.nt = (int)pl.opt.get("k1.nt", field_bytes >= 256.0 * 1024 * 1024 ? 5 : 0);
  if (base.nt & ~5) throw Error(SF_ERR_INVALID, "k1.nt: 1 (non-temporal stores), 4 (non-temporal loads of unshared rows) or 5");
  const std::string pfx = base.noj ? "k2." : "k1.";
  const long long pin_bx = pl.opt.get(pfx + "bx", 0);
  const long long pin_by = base.noj ? 1 : pl.opt.get("k1.by", 0);
  const long long pin_rj = base.noj ? 1 : pl.opt.get("k1.rj", 0);
  const int range = (int)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local);

  std::vector<std::pair<double, StarCfg>> ranked;
  for (int bx : {64, 128, 256}) {
    if (pin_bx && bx != pin_bx) continue;
    for (int rj = 1; rj <= 8; ++rj) {
      if (pin_rj && rj != pin_rj) continue;
      for (int by = 1; by <= 16; ++by) {
        if (pin_by && by != pin_by) continue;
        if (bx * by > 1024) continue;
        StarCfg c = base;
        c.BX = bx;
        c.RJ = rj;
        c.BY = by;
        if (!c.noj && by * rj - 2 * T * c.R < 1) continue;
        if (c.R == 2 && !c.noj && rj < 2) continue;  // a thread row publishes its first and last two rows
        star_finish_cfg(c, P, T);
        if (star_lds_bytes(c, dt) > 160 * 1024) continue;
        const bool pinned = pin_bx && pin_by && pin_rj;
        if (!pinned && star_blocks_per_cu(c, dt) < 1) continue;  // would spill
        double chunk_cost = 1.0;
        star_chunk_planes(c, dt, range, &chunk_cost);
        const double jcost = c.noj ? 1.0 : (double)c.NJT * c.BY * c.RJ / (double)P.n[1];
        const double kcost = (double)c.NKT * c.BX * c.VK / (double)P.n[2];
        // a block whose waves do not divide evenly over the 4 SIMDs of its unit
        // leaves SIMDs idle (one block per unit); every thread pays two LDS edge
        // rows per stage whatever its row count (profiles/r01_sweep_13_c5_tiles.log:
        // 64x7 / 64x11 threads lose 10 % / 5 % to 64x8, 3 rows per thread 5-10 % to 4-5)
        const int waves = (c.BX * c.BY + 63) / 64;
        const double simd_balance =
            (!c.noj && star_blocks_per_cu(c, dt) == 1) ? (double)((waves + 3) / 4 * 4) / (double)waves : 1.0;
        // 2-D: a one-wave block needs neither LDS nor a barrier; wider blocks exchange
        // their edge columns through LDS every step (measured 10-30 % slower on C2)
        const double edge_rows = c.noj ? (c.BX > 64 ? 1.2 : 1.0) : 1.0 + 0.8 * c.R / (double)c.RJ;
        // ties go to the larger block (fewer barriers per point)
        const double cost = jcost * kcost * chunk_cost * simd_balance * edge_rows *
                            (1.0 + 1e-4 / (double)(c.BX * c.BY * c.RJ));
        ranked.push_back({cost, c});
      }
    }
  }
  if (ranked.empty()) throw Error(SF_ERR_INVALID, "star kernel: no tile shape satisfies the given k1.* options");
  std::stable_sort(ranked.begin(), ranked.end(),
                   [](const std::pair<double, StarCfg>& a, const std::pair<double, StarCfg>& b) {
                     return a.first < b.first;
                   });
  std::vector<StarCfg> out;
  for (auto& rc : ranked) out.push_back(rc.second);
  if (pl.opt.get("debug", 0) != 0)
    for (size_t i = 0; i < std::min<size_t>(ranked.size(), 8); ++i)
      std::fprintf(stderr, "[sf_hip] rank %zu cost %.4f: T=%d block %dx%d rows/thread %d tiles %dx%d\n", i + 1,
                   ranked[i].first, T, ranked[i].second.BX, ranked[i].second.BY, ranked[i].second.RJ,
                   ranked[i].second.NJT, ranked[i].second.NKT);
  return out;
}

// Pick a tile shape for a fused group by compiling candidates in order of
// modelled cost and reading the code object's metadata.  A kernel that spills,
// uses scratch or overflows into AGPRs is rejected: besides being slow, such
// kernels were observed to produce wrong results on gfx950 / ROCm 7 for
// programs with device math calls (profiles/r01_config_fuzz.log).  Returns
// false if no clean shape exists (caller shortens the group or goes generic).
struct StarChoice {
  bool ok = false;
  StarCfg cfg;
  int ck = -1;
  // option autotune=<k>: the first k clean candidates in ranked order (the first is
  // cfg / ck); they are timed on the device before the first execution
  std::vector<std::pair<StarCfg, int>> alts;
  std::string sig;
};

static StarChoice select_star(sf_plan& pl, std::map<std::string, StarChoice>& memo,
                              const std::vector<int>& kernels, DT dt, const StarDag* dag = nullptr) {
  const Program& P = pl.P;
  StarCfg probe;
  probe.T = dag ? dag->depth : (int)kernels.size();
  if (dag) probe.dag = *dag;
  const std::string sig = std::to_string(fnv1a(gen_star(P, kernels, probe).source));
  auto it = memo.find(sig);
  if (it != memo.end()) return it->second;
  const std::string prefix = std::string("sf_star") + (P.n[1] == 1 ? "2d_" : "3d_") + short_of(dt) + "_t" +
                             std::to_string(probe.T) + (dag ? "g" + std::to_string(kernels.size()) : std::string());
  StarChoice out;
  std::vector<StarCfg> ranked;
  try {
    ranked = rank_star_cfgs(pl, probe.T, dt, dag ? &probe : nullptr);
  } catch (const Error&) {
    memo[sig] = out;
    return out;
  }
  const bool pinned = pl.opt.kv.count(P.n[1] == 1 ? "k2.bx" : "k1.bx") &&
                      (P.n[1] == 1 || (pl.opt.kv.count("k1.by") && pl.opt.kv.count("k1.rj")));
  const size_t tries = std::min<size_t>(ranked.size(), (size_t)8);
  int rejected = 0, sgpr_rejects = 0;
  for (size_t ci = 0; ci < tries && rejected < 2; ++ci) {  // compile errors rarely depend on the shape
    ranked[ci].lds_bytes = star_lds_bytes(ranked[ci], dt);
    StarKernelSource g = gen_star(P, kernels, ranked[ci]);
    int ck = -1;
    try {
      // (off since round 4: float-typed sums -- integer boundary literals, the generator's programs -- gain 0.5-8 %
      // without the packed adds, nothing loses: profiles/r04_slp_families.log)
      ck = intern_kernel(pl, prefix, g.source, slp_flags(false));
    } catch (const Error& e) {
      // a shape the compiler rejects is no candidate (a pinned shape reports it);
      // the group is shortened and in the end the generic kernel takes over
      if (pinned || e.status != SF_ERR_COMPILE) throw;
      report_rejected_candidate(pl, "star", std::to_string(ci + 1) + "/" + std::to_string(ranked.size()), e);
      ++rejected;
      continue;
    }
    const CompiledKernel& k = pl.kernels[ck];
    if (pl.opt.get("debug", 0) != 0)
      std::fprintf(stderr, "[sf_hip] candidate %zu/%zu T=%d block %dx%d rows/thread %d: vgpr %d agpr %d spill %d scratch %d sgpr spill %d late exec restores %d\n",
                   ci + 1, ranked.size(), ranked[ci].T, ranked[ci].BX, ranked[ci].BY, ranked[ci].RJ, k.vgprs,
                   k.agprs, k.spills, k.scratch, k.sgpr_spills, k.late_exec_restores);
    (void)sgpr_rejects;  // (star kernels: SGPR spills do depend on the shape -- hotspot 512^3: shapes 1-3 spill, 4 does not)
    if (!kernel_unsafe(k) && (!kernel_slow(k) || (pinned && pl.opt.get("allow_spills", 0) != 0))) {
      if (!out.ok) {
        out.ok = true;
        out.cfg = ranked[ci];
        out.ck = ck;
      }
      out.alts.push_back({ranked[ci], ck});
      if (pinned || (long long)out.alts.size() >= std::max<long long>(1, pl.opt.get("autotune", 0))) break;
    }
  }
  out.sig = sig;
  memo[sig] = out;
  return out;
}

// The same for a group of radius-2 star operators (kernels/wstar3d.h).
static StarChoice select_wide(sf_plan& pl, std::map<std::string, StarChoice>& memo, const std::vector<int>& kernels, DT dt) {
  const Program& P = pl.P;
  StarCfg probe;
  probe.T = (int)kernels.size();
  probe.R = 2;
  probe.noj = (P.n[1] == 1);
  const std::string sig = "wide" + std::to_string(fnv1a(gen_wide(P, kernels, probe).source));
  auto it = memo.find(sig);
  if (it != memo.end()) return it->second;
  const std::string prefix = std::string(P.n[1] == 1 ? "sf_wstar2d_" : "sf_wstar3d_") + short_of(dt) + "_t" +
                             std::to_string(kernels.size());
  StarChoice out;
  std::vector<StarCfg> ranked;
  try {
    ranked = rank_star_cfgs(pl, (int)kernels.size(), dt, &probe);
  } catch (const Error&) {
    memo[sig] = out;
    return out;
  }
  const bool pinned = pl.opt.kv.count(P.n[1] == 1 ? "k2.bx" : "k1.bx") &&
                      (P.n[1] == 1 || (pl.opt.kv.count("k1.by") && pl.opt.kv.count("k1.rj")));
  const size_t tries = std::min<size_t>(ranked.size(), (size_t)8);
  int rejected = 0;
  for (size_t ci = 0; ci < tries && rejected < 2; ++ci) {
    StarKernelSource g = gen_wide(P, kernels, ranked[ci]);
    int ck = -1;
    try {
      // (3-D wide stars are the one family that gains from the packed adds: 8.8e5 against 8.0e5 without; 2-D: off, +2 %)
      ck = intern_kernel(pl, prefix, g.source, slp_flags(P.n[1] != 1));
    } catch (const Error& e) {
      if (pinned || e.status != SF_ERR_COMPILE) throw;
      report_rejected_candidate(pl, "wide-star", std::to_string(ci + 1) + "/" + std::to_string(ranked.size()), e);
      ++rejected;
      continue;
    }
    const CompiledKernel& k = pl.kernels[ck];
    if (pl.opt.get("debug", 0) != 0)
      std::fprintf(stderr,
                   "[sf_hip] wide candidate %zu/%zu T=%d block %dx%d rows/thread %d: vgpr %d agpr %d spill %d scratch %d "
                   "sgpr spill %d late exec restores %d lds %d\n",
                   ci + 1, ranked.size(), ranked[ci].T, ranked[ci].BX, ranked[ci].BY, ranked[ci].RJ, k.vgprs, k.agprs, k.spills,
                   k.scratch, k.sgpr_spills, k.late_exec_restores, k.lds);
    if (!kernel_unsafe(k) && (!kernel_slow(k) || (pinned && pl.opt.get("allow_spills", 0) != 0))) {
      if (!out.ok) {
        out.ok = true;
        out.cfg = ranked[ci];
        out.ck = ck;
      }
      out.alts.push_back({ranked[ci], ck});
      if (pinned || (long long)out.alts.size() >= std::max<long long>(1, pl.opt.get("autotune", 0))) break;
    }
  }
  out.sig = sig;
  memo[sig] = out;
  return out;
}

// A dense-neighbourhood operator (kernels/dense3d.h): a few block shapes in order of preference -- the
// tile is staged in LDS, so what matters is the halo it re-reads and two blocks per unit (LDS, waves).
static StarChoice select_dense(sf_plan& pl, std::map<std::string, StarChoice>& memo, int kidx, DT dt, int radius = 2) {
  const Program& P = pl.P;
  const bool noj = P.n[1] == 1;
  struct Shape {
    int bx, by, rj;
  };
  // (measured on the 125-point box 512^3, profiles/r03_dense_shapes.log: 2.53 / 2.52 / 2.47 / 2.38 / 2.18e5 Mcells/s
  //  in this order; four rows per thread 1.8e5)
  static const Shape shapes3d[] = {{64, 8, 1}, {32, 16, 1}, {32, 8, 1}, {32, 8, 2}, {16, 16, 2}};
  static const Shape shapes2d[] = {{64, 1, 1}, {128, 1, 1}};
  StarChoice out;
  const std::string prefix = std::string(noj ? "sf_dense2d_" : "sf_dense3d_") + short_of(dt);
  const long long pin_bx = pl.opt.get(noj ? "k2.bx" : "k1.bx", 0), pin_by = pl.opt.get("k1.by", 0), pin_rj = pl.opt.get("k1.rj", 0);
  std::vector<Shape> todo;
  if (pin_bx && (noj || (pin_by && pin_rj))) todo.push_back({(int)pin_bx, noj ? 1 : (int)pin_by, noj ? 1 : (int)pin_rj});
  else todo.assign(noj ? std::begin(shapes2d) : std::begin(shapes3d), noj ? std::end(shapes2d) : std::end(shapes3d));
  // a plain sum streams (the planes by LDS-DMA, every plane read from LDS once); anything else: every shape with row
  // segments held in registers first, then, for the first shapes, the form that reads each operand from LDS where the
  // text uses it (operators whose typing doubles what a segment takes)
  struct Variant {
    Shape first;
    bool second;  // operands read from LDS where the text uses them
    bool stream;  // ONE left-associated sum: the streaming form (SF_DENSE_STREAM)
  };
  std::vector<Variant> variants;
  // Streaming form: more rows per thread pay, the accumulators of the open output planes being the only state
  // (125-point box 512^3, profiles/r04_dense_stream.log: 64x2x4 0.397 ms per operator, 32x4x4 0.39, 64x2x2 0.41,
  //  64x4x2 0.43, 64x4x4 0.47)
  // (one row per thread last: operators whose sums are typed double -- a float boundary literal -- hold two registers per
  //  accumulator, seven sets of them at radius 3)
  static const Shape streams3d[] = {{64, 2, 4}, {32, 4, 4}, {64, 2, 2}, {64, 4, 2}, {64, 4, 1}, {64, 8, 1}};
  // (a float32 sum typed double -- a float boundary literal -- at radius 3: seven sets of two-register accumulators fit one
  //  row per thread only; the other shapes would each cost seconds of compilation to find that out)
  static const Shape streams3d_wide_acc[] = {{64, 4, 1}, {64, 8, 1}};
  // (few terms -- a radius-3 cross, 18 of them: the launch is paced by memory, two rows per thread leave room for twice
  //  the waves; radius-3 cross 512^3: 64x4x2 200.7 us per operator, 64x2x4 213.5, 128x4x2 210.4, 64x4x4 219.0, 64x2x2 267.6,
  //  against 392 on the generic kernel, profiles/r05_cross3.log)
  static const Shape streams3d_sparse[] = {{64, 4, 2}, {64, 2, 4}, {128, 4, 2}, {64, 4, 4}, {64, 4, 1}};
  DenseSum sum_form;
  if (dense_sum_form(P, P.kernels[kidx], &sum_form)) {
    const bool wide_acc = dt == DT::F32 && radius == 3 &&
                          (sum_form.ttype[0] == DT::F64 || sum_form.ttype[1] == DT::F64);
    if (pin_bx && (noj || (pin_by && pin_rj))) variants.push_back({todo[0], false, true});
    else if (noj) for (const Shape& sh : shapes2d) variants.push_back({sh, false, true});
    else if (wide_acc) for (const Shape& sh : streams3d_wide_acc) variants.push_back({sh, false, true});
    else if (sum_form.terms.size() <= 32) for (const Shape& sh : streams3d_sparse) variants.push_back({sh, false, true});
    else for (const Shape& sh : streams3d) variants.push_back({sh, false, true});
  }
  if (radius != 3) {  // (radius 3: the streaming form or nothing)
    for (const Shape& sh : todo) variants.push_back({sh, false, false});
    for (size_t i = 0; i < todo.size() && i < 2; ++i) variants.push_back({todo[i], true, false});
  }
  for (const auto& variant : variants) {
    const Shape& sh = variant.first;
    StarCfg c;
    c.T = 1;
    c.R = radius;
    c.dense = true;
    c.dense_scalar = variant.second;
    c.dense_stream = variant.stream ? 1 : 0;
    c.VK = dt == DT::F64 ? 2 : 4;  // 16 bytes of output per lane and row
    c.BX = sh.bx;
    c.BY = sh.by;
    c.RJ = sh.rj;
    c.noj = noj;
    c.n0g = P.n[0];
    c.n1 = P.n[1];
    c.n2 = P.n[2];
    if (sh.bx * sh.by > 1024 || sh.bx * sh.by < 64 || (sh.bx * sh.by) % 64 != 0) continue;
    const long long tk = (long long)c.BX * c.VK, tj = noj ? 1 : (long long)c.BY * c.RJ;
    c.ktiled = true;
    c.HK = 0;
    c.NKT = (int)((P.n[2] + tk - 1) / tk);
    c.NJT = noj ? 1 : (int)((P.n[1] + tj - 1) / tj);
    const int rc = (radius + 1) / 2 * 2;  // halo columns of a row segment (dense3d.h: SF_RC)
    size_t lds = 6 * (size_t)(tj + (noj ? 0 : 2 * radius)) * (size_t)(tk + 2 * rc) * size_of(dt);
    if (variant.stream) {
      // planes by LDS-DMA: rows start a whole 16-byte chunk left of the tile, slots are whole 1-KiB pieces (SF_RCL, SF_SLOT_STRIDE)
      const int ce = (int)(16 / size_of(dt)), rcl = (rc + ce - 1) / ce * ce;
      // (two slots = requested one plane ahead, measured equal to three on the 125-point box, 355.8 us sustained either
      //  way -- the launch is paced by vector issue at the clock the chip holds, profiles/r05_dense_whatif.log)
      c.dense_in_slots = 2;
      // (a sum whose terms are not ordered by plane -- a cross -- keeps the planes its late terms read: stream_schedule)
      c.dense_lag = stream_schedule(P.kernels[kidx], sum_form).max_lag;
      const size_t slot = ((size_t)(tj + (noj ? 0 : 2 * radius)) * (size_t)(tk + 2 * rcl) * size_of(dt) + 1023) / 1024 * 1024;
      lds = (size_t)(c.dense_in_slots + c.dense_lag) * slot;
    }
    if (lds > 160 * 1024) continue;
    c.lds_bytes = lds;
    const double field_bytes = (double)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local) * (double)P.n[1] * (double)P.n[2] *
                               (double)size_of(dt);
    c.nt = (int)pl.opt.get("k1.nt", field_bytes >= 256.0 * 1024 * 1024 ? 1 : 0);
    StarKernelSource g;
    g = gen_dense(P, kidx, c);
    const std::string sig = "dense" + std::to_string(fnv1a(g.source));
    auto it = memo.find(sig);
    if (it != memo.end()) return it->second;
    int ck = -1;
    try {
      ck = intern_kernel(pl, prefix, g.source, slp_flags(false));
    } catch (const Error& e) {
      if (e.status != SF_ERR_COMPILE) throw;
      report_rejected_candidate(pl, "dense", std::to_string(sh.bx) + "x" + std::to_string(sh.by) + "x" + std::to_string(sh.rj), e);
      continue;
    }
    const CompiledKernel& k = pl.kernels[ck];
    if (pl.opt.get("debug", 0) != 0)
      std::fprintf(stderr, "[sf_hip] dense candidate block %dx%d rows/thread %d%s: vgpr %d agpr %d spill %d scratch %d lds %d\n", sh.bx,
                   sh.by, sh.rj, variant.stream ? " (plain sum, planes streamed)" : variant.second ? " (operands read where used)" : "", k.vgprs, k.agprs, k.spills, k.scratch, k.lds);
    if (!kernel_unsafe(k) && (!kernel_slow(k) || (pin_bx != 0 && pl.opt.get("allow_spills", 0) != 0))) {
      out.ok = true;
      out.cfg = c;
      out.ck = ck;
      out.alts.push_back({c, ck});
      out.sig = sig;
      memo[sig] = out;
      return out;
    }
  }
  return out;
}

// Two plain sums in one streaming dense launch (kernels/dense3d.h: SF_DENSE_T2, codegen.hpp: gen_dense_t2): radius-1
// boxes, or (round 5) radius-2 sums of few terms -- the generator's crosses --, or THREE radius-1 sums (the benchmark's
// chain), with a factor per term where the caller allows it.  Tiles overlap by what all operators but the first reach in
// rows (and, when a row is cut, four columns) on either side.
static StarChoice select_dense_t2(sf_plan& pl, std::map<std::string, StarChoice>& memo, const std::vector<int>& kernels, DT dt) {
  const Program& P = pl.P;
  const bool noj = P.n[1] == 1;
  struct Shape {
    int bx, by, rj;
  };
  StarChoice out;
  const int nst = (int)kernels.size();  // operators per launch: two, or three (radius 1)
  if (nst != 2 && nst != 3) return out;
  int reach = 1;
  std::vector<int> lags;
  for (int s = 0; s < nst; ++s) {
    int rs = 1;
    if (s + 1 < nst && !dense_t2_eligible(P, P.kernels[kernels[s]], P.kernels[kernels[s + 1]], &rs, true, true)) return out;
    reach = std::max(reach, rs);
    DenseSum sum;
    dense_sum_form(P, P.kernels[kernels[s]], &sum);
    lags.push_back(stream_schedule(P.kernels[kernels[s]], sum).max_lag);
  }
  if (nst == 3 && reach != 1) return out;
  const int k2 = kernels[1];
  const int lag1 = lags[0], lag2 = lags[1], lag3 = nst == 3 ? lags[2] : 0;
  // (27-point box 512^3 f32, profiles/r04_dense_t2.log: 128x8x2 9.3e5 Mcells/s, 128x4x4 8.7e5, 128x6x2 8.5-8.7e5,
  //  128x4x3 8.4e5, 64x8x2 7.7e5, 64x4x4 7.4e5 against 8.5e5 on the compact kernel; 9-point box 4096^2: 64 lanes 1.18e6,
  //  128 lanes 1.11e6, 256 lanes 0.97e6 against 1.0e6)
  //  with ONE slot for the input planes (a second barrier per step, 124 KB) 18-row tiles fit: 128x6x3, 32 row tiles x 8
  //  chunks = one block per unit: 9.7-9.8e5)
  static const Shape shapes3d_f32[] = {{128, 6, 3}, {128, 8, 2}, {128, 4, 4}, {128, 6, 2}, {64, 8, 2}, {64, 4, 4}};
  // (float64, 27-point box 512^3, profiles/r04_box_f64.log: 256x3x3 4.0e5 Mcells/s, 256x4x2 3.8e5, against 3.2e5 on the compact
  //  kernel two deep and 1.6e5 three deep)
  static const Shape shapes3d_f64[] = {{256, 3, 3}, {256, 4, 2}, {128, 8, 2}, {128, 4, 4}, {64, 8, 2}};
  static const Shape shapes2d[] = {{64, 1, 1}, {128, 1, 1}, {256, 1, 1}};
  // reach two: rows of 34 threads -- 136 columns, 128 of them kept: a row of 512 is four tiles -- or of 32; the shape that
  // keeps most of what it computes on this grid is tried first, among equals one row per thread (radius-2 cross 512^3,
  // 30-row tiles: 34x30 threads x 1 row 226.6 us per launch, 34x15 x 2 rows 236.6 on the same box, profiles/r05_cross2_fused.log)
  static const Shape shapes3d_reach2[] = {{34, 30, 1}, {34, 15, 2}, {34, 16, 2}, {32, 16, 2}, {34, 12, 2}, {32, 14, 2}, {32, 12, 2}, {32, 8, 2}, {32, 6, 2}};
  // radius-1 sums whose terms are not ordered by plane, and three operators per launch
  static const Shape shapes3d_lagged[] = {{34, 15, 3}, {34, 30, 1}, {34, 15, 2}, {34, 16, 2}, {34, 12, 3}, {34, 12, 2}, {32, 16, 2}, {32, 12, 2}, {32, 8, 2}};
  // dense.t2: 0 never, 1 (default) where a tile shape wastes at most a quarter of its lanes and rows on this grid,
  // 2 wherever a shape compiles (tests, fuzz campaigns on small grids)
  const bool force = pl.opt.get("dense.t2", 1) >= 2;
  const std::string prefix = std::string(noj ? "sf_dense2d_" : "sf_dense3d_") + short_of(dt) + (nst == 3 ? "_t3" : "_t2");
  const long long pin_bx = pl.opt.get(noj ? "k2.bx" : "k1.bx", 0), pin_by = pl.opt.get("k1.by", 0), pin_rj = pl.opt.get("k1.rj", 0);
  std::vector<Shape> todo;
  if (pin_bx && (noj || (pin_by && pin_rj))) todo.push_back({(int)pin_bx, noj ? 1 : (int)pin_by, noj ? 1 : (int)pin_rj});
  else if (noj) todo.assign(std::begin(shapes2d), std::end(shapes2d));
  else if (reach == 2) todo.assign(std::begin(shapes3d_reach2), std::end(shapes3d_reach2));
  else if (dt == DT::F64) todo.assign(std::begin(shapes3d_f64), std::end(shapes3d_f64));
  else todo.assign(std::begin(shapes3d_f32), std::end(shapes3d_f32));
  const int edge = (nst - 1) * reach;  // rows on either side of a tile whose results are not stored
  const bool lagged = lag1 != 0 || lag2 != 0 || lag3 != 0;
  auto kept = [&](const Shape& sh) {  // what the tiles of this shape cover against what the grid holds
    const long long tk = (long long)sh.bx * (dt == DT::F64 ? 2 : 4), tj = noj ? 1 : (long long)sh.by * sh.rj;
    if (tk <= 8 || (!noj && tj <= 2 * edge)) return 0.0;
    const long long nkt = tk != P.n[2] ? (P.n[2] + (tk - 8) - 1) / (tk - 8) : 1, njt = noj ? 1 : (P.n[1] + (tj - 2 * edge) - 1) / (tj - 2 * edge);
    return ((double)P.n[2] / ((double)nkt * (double)tk)) * (noj ? 1.0 : (double)P.n[1] / ((double)njt * (double)tj));
  };
  if (reach == 1 && (lagged || nst == 3) && !noj && !pin_bx && dt == DT::F32) todo.assign(std::begin(shapes3d_lagged), std::end(shapes3d_lagged));
  if ((reach == 2 || lagged || nst == 3) && !pin_bx)
    std::stable_sort(todo.begin(), todo.end(), [&](const Shape& a, const Shape& b) { return kept(a) > kept(b) + 0.02; });
  for (const Shape& sh : todo) {
    StarCfg c;
    c.T = 1;
    c.R = nst * reach;  // (what the group reaches, `reach` planes per operator: what the slab halo and the chunking see)
    c.dense = true;
    c.dense_stream = 1;
    c.dense_t2 = 1;
    c.dense_k2 = k2;
    c.VK = dt == DT::F64 ? 2 : 4;
    c.BX = sh.bx;
    c.BY = sh.by;
    c.RJ = noj ? 1 : sh.rj;
    c.noj = noj;
    c.n0g = P.n[0];
    c.n1 = P.n[1];
    c.n2 = P.n[2];
    // (a block need not be whole waves -- its last wave runs with lanes off and requests no pieces)
    if (sh.bx * sh.by > 1024 || sh.bx * sh.by < 64) continue;
    const long long tk = (long long)c.BX * c.VK, tj = noj ? 1 : (long long)c.BY * c.RJ;
    if (!noj && tj < 2 * edge + 1) continue;
    c.ktiled = tk != P.n[2];
    if (c.ktiled && tk <= 8) continue;
    c.HK = 0;
    c.NKT = c.ktiled ? (int)((P.n[2] + (tk - 8) - 1) / (tk - 8)) : 1;
    c.NJT = noj ? 1 : (int)((P.n[1] + (tj - 2 * edge) - 1) / (tj - 2 * edge));
    // what the tiles cover against what the grid holds (rows recomputed by the neighbouring tile, lanes beyond the row)
    const double used = ((double)P.n[2] / ((double)c.NKT * (double)tk)) * (noj ? 1.0 : (double)P.n[1] / ((double)c.NJT * (double)tj));
    if (!force && !pin_bx && used < ((reach == 2 || nst == 3) ? 0.6 : 0.75)) continue;
    // (a row cut into tiles has no halo columns in LDS -- codegen.hpp: gen_dense_t2 -- and one 1-KiB piece in front of slot 0)
    const size_t row_bytes = (size_t)(tk + (c.ktiled ? 0 : 2 * (16 / size_of(dt)))) * size_of(dt);
    const size_t slot = ((size_t)(tj + (noj ? 0 : 2 * reach)) * row_bytes + 1023) / 1024 * 1024;
    c.dense_in_slots = 2;  // (the plane requested a whole step ahead)
    c.dense_lag = lag1;
    size_t lds = 0;
    if (reach == 1 && nst == 2 && !lagged) {
      // two input slots and two between the operators where four slots fit; three slots: ONE between the operators,
      // written at the very end of a step behind a second barrier
      const long long fit = (long long)((160 * 1024 - (c.ktiled ? 1024 : 0)) / slot);
      if (fit < 3) continue;
      c.dense_mid_slots = fit >= 4 ? 2 : 1;
      lds = (size_t)(c.dense_in_slots + c.dense_mid_slots) * slot + (c.ktiled ? 1024 : 0);
    } else {
      // the input ring keeps the planes the first operator's late terms read; between the operators: the slot being
      // written, the plane published a step ago and the ones the second operator reads late -- TJ rows each, no halo
      // rows (dense3d.h: SF_MID_HALO 0)
      c.dense_mid_slots = 2 + lag2;
      lds = (size_t)(c.dense_in_slots + lag1) * slot + (size_t)(c.dense_mid_slots + (nst == 3 ? 2 + lag3 : 0)) * ((size_t)tj * row_bytes) +
            (size_t)(noj ? 0 : reach) * row_bytes + 32 + (c.ktiled ? 1024 : 0);
      if (lds > 160 * 1024) continue;
    }
    c.lds_bytes = lds;
    const double field_bytes = (double)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local) * (double)P.n[1] * (double)P.n[2] *
                               (double)size_of(dt);
    c.nt = (int)pl.opt.get("k1.nt", field_bytes >= 256.0 * 1024 * 1024 ? 1 : 0);
    StarKernelSource g;
    try {
      g = gen_dense_t2(P, kernels, c);
    } catch (const Error&) {
      return out;  // (not a pair this form takes)
    }
    const std::string sig = "dense_t2" + std::to_string(fnv1a(g.source));
    auto it = memo.find(sig);
    if (it != memo.end()) return it->second;
    int ck = -1;
    try {
      ck = intern_kernel(pl, prefix, g.source, slp_flags(false));
    } catch (const Error& e) {
      if (e.status != SF_ERR_COMPILE) throw;
      report_rejected_candidate(pl, nst == 3 ? "dense (three fused)" : "dense (two fused)", std::to_string(sh.bx) + "x" + std::to_string(sh.by) + "x" + std::to_string(sh.rj), e);
      continue;
    }
    const CompiledKernel& k = pl.kernels[ck];
    if (pl.opt.get("debug", 0) != 0)
      std::fprintf(stderr, "[sf_hip] dense candidate block %dx%d rows/thread %d (%d plain sums fused, planes streamed): vgpr %d agpr %d spill %d scratch %d lds %d\n",
                   sh.bx, sh.by, sh.rj, nst, k.vgprs, k.agprs, k.spills, k.scratch, k.lds);
    if (!kernel_unsafe(k) && (!kernel_slow(k) || (pin_bx != 0 && pl.opt.get("allow_spills", 0) != 0))) {
      out.ok = true;
      out.cfg = c;
      out.ck = ck;
      out.alts.push_back({c, ck});
      out.sig = sig;
      memo[sig] = out;
      return out;
    }
  }
  return out;
}

// Extra compiler flags of the compact kernels: without the SLP vectoriser hipcc
// keeps the 26 adds of a box stencil scalar instead of pairing them into
// v_pk_add_f32, whose operand pairs it has to assemble with moves (option compact.slp).
static std::string compact_flags(const sf_plan& pl) {
  (void)pl;
  return "-fno-slp-vectorize";
}

// The same for a group of compact operators (kernels/compact3d.h).
static StarChoice select_compact(sf_plan& pl, std::map<std::string, StarChoice>& memo, const std::vector<int>& kernels,
                                 DT dt) {
  const Program& P = pl.P;
  StarCfg probe;
  probe.T = (int)kernels.size();
  const std::string sig = "compact" + std::to_string(fnv1a(gen_compact(P, kernels, probe).source));
  auto it = memo.find(sig);
  if (it != memo.end()) return it->second;
  const std::string prefix = std::string(P.n[1] == 1 ? "sf_compact2d_" : "sf_compact3d_") + short_of(dt) + "_t" +
                             std::to_string(kernels.size());
  StarChoice out;
  std::vector<StarCfg> ranked;
  try {
    ranked = rank_star_cfgs(pl, (int)kernels.size(), dt, &probe);
  } catch (const Error&) {
    memo[sig] = out;
    return out;
  }
  const bool pinned = pl.opt.kv.count(P.n[1] == 1 ? "k2.bx" : "k1.bx") &&
                      (P.n[1] == 1 || (pl.opt.kv.count("k1.by") && pl.opt.kv.count("k1.rj")));
  // (a single operator that finds no clean tile falls to the generic kernel, ten times slower or worse: where registers
  //  are scarce -- float64, or a second field's window -- it gets more shapes to try; only the failures cost compile time)
  bool second_field = false;
  for (int kk : kernels) {
    CompactShape sh;
    if (compact_eligible(P, P.kernels[kk], &sh) && !sh.extra.empty()) second_field = true;
  }
  const bool more = kernels.size() == 1 && (dt == DT::F64 || second_field);
  const size_t tries = std::min<size_t>(ranked.size(), (size_t)(more ? 24 : 8));
  int rejected = 0, sgpr_rejects = 0;
  // Dense 3-D groups (box-like: every operator reads 18 or more of the 27 offsets of one field)
  // are bound by vector-instruction issue; letting the scheduler mix the rows of a step is worth
  // +4 % there (27-point box 512^3: 7.49 -> 7.75e5, profiles/r02_synth_box.log) and costs 12
  // registers, so the unfenced form is tried first and the fenced one if it does not come out clean.
  bool dense = P.n[1] != 1 && !pl.opt.kv.count("k1.fence");
  for (size_t si = 0; dense && si < kernels.size(); ++si) {
    CompactShape sh;
    const std::string want = si == 0 ? std::string() : P.kernels[kernels[si - 1]].name;
    dense = compact_eligible(P, P.kernels[kernels[si]], &sh, want) && sh.extra.empty() &&
            __builtin_popcount(sh.need) >= 18;
  }
  for (size_t ci2 = 0; ci2 < 2 * tries && rejected < 2; ++ci2) {
    const size_t ci = ci2 / 2;
    if (!dense && (ci2 & 1)) continue;
    if (dense) ranked[ci].row_fence = (int)(ci2 & 1);
    StarKernelSource g = gen_compact(P, kernels, ranked[ci]);
    int ck = -1;
    try {
      ck = intern_kernel(pl, prefix, g.source, compact_flags(pl));
    } catch (const Error& e) {
      if (pinned || e.status != SF_ERR_COMPILE) throw;
      report_rejected_candidate(pl, "compact", std::to_string(ci + 1) + "/" + std::to_string(ranked.size()), e);
      ++rejected;
      continue;
    }
    const CompiledKernel& k = pl.kernels[ck];
    if (pl.opt.get("debug", 0) != 0)
      std::fprintf(stderr,
                   "[sf_hip] compact candidate %zu/%zu T=%d block %dx%d rows/thread %d: vgpr %d agpr %d spill %d "
                   "scratch %d sgpr spill %d late exec restores %d lds %d\n",
                   ci + 1, ranked.size(), ranked[ci].T, ranked[ci].BX, ranked[ci].BY, ranked[ci].RJ, k.vgprs, k.agprs,
                   k.spills, k.scratch, k.sgpr_spills, k.late_exec_restores, k.lds);
    // scalar registers are spent on the group's windows and descriptors more than on the
    // tile shape: four shapes that spill them settle it for this group length
    if (kernel_unsafe(k) && ++sgpr_rejects >= 4) break;
    if (!kernel_unsafe(k) && (!kernel_slow(k) || (pinned && pl.opt.get("allow_spills", 0) != 0))) {
      if (!out.ok) {
        out.ok = true;
        out.cfg = ranked[ci];
        out.ck = ck;
      }
      out.alts.push_back({ranked[ci], ck});
      if (pinned || (long long)out.alts.size() >= std::max<long long>(1, pl.opt.get("autotune", 0))) break;
    }
  }
  out.sig = sig;
  memo[sig] = out;
  return out;
}

// One line of sf_plan_describe per launch.
static void describe_step(std::ostringstream& desc, const sf_plan& pl, const Step& st) {
  const Program& P = pl.P;
  const DT dt = P.kernels[st.kernels[0]].dt;
  const CompiledKernel& ck = pl.kernels[st.ck];
  desc << "  launch " << ck.name << ": ";
  for (int k : st.kernels) desc << P.kernels[k].name << " ";
  if (st.star)
    desc << (st.compact ? "[compact windows " + std::to_string(st.cfg.nwin) + " T=" : st.dense ? std::string("[dense T=") : st.wide ? std::string("[wide star T=") : std::string("[star T=")) << ((st.dense && st.cfg.dense_t2) ? (int)st.kernels.size() : st.cfg.T) << " block " << st.cfg.BX << "x" << st.cfg.BY << " rows/thread "
         << st.cfg.RJ << " tiles " << st.cfg.NJT << "x" << st.cfg.NKT << " chunk "
         << star_chunk_length(pl, st.cfg, dt, (int)pl.n_local) << " lds " << star_lds_bytes(st.cfg, dt)
         << " B]";
  else
    desc << "[point]";
  desc << " in";
  for (int b : st.in_bufs) desc << " b" << b;
  desc << " out";
  for (int b : st.out_bufs) desc << " b" << b;
  if (st.star && st.cfg.dag.on)
    desc << " [dag: " << st.cfg.dag.stages.size() << " stages, " << st.cfg.dag.nwin << " windows]";
  desc << " {vgpr " << ck.vgprs << " agpr " << ck.agprs << " spill " << ck.spills
       << " scratch " << ck.scratch << "}";
  // results produced under the diagnostic environment switches must not pass for normal runs
  if (ck.foreign) desc << " [foreign object: $SF_HIP_OBJECT_DIR]";
  if (ck.env_flags) desc << " [$SF_HIP_EXTRA_FLAGS: " << ck.flags << "]";
  if (ck.late_exec_restores != 0 && std::getenv("SF_HIP_UNSAFE_SGPR_SPILLS"))
    desc << " [UNSAFE: EXEC-restore fault tolerated, $SF_HIP_UNSAFE_SGPR_SPILLS]";
  desc << "\n";
}

// The whole description: header line, one line per launch (long chains: the first ones).
std::string describe_plan(const sf_plan& pl) {
  const Program& P = pl.P;
  std::ostringstream desc;
  desc << "program " << P.name << ": dims " << P.n[0] << "x" << P.n[1] << "x" << P.n[2] << ", "
       << P.kernels.size() << " operators, " << pl.steps.size() << " launches, " << pl.buffers.size()
       << " device buffers\n";
  for (auto& st : pl.steps) {
    if (desc.tellp() > 16384) break;  // long chains: describe the first launches only
    describe_step(desc, pl, st);
  }
  desc << "  compiler: " << compiler_id() << "\n";
  return desc.str();
}

// THE plan options (sf_plan_create's `options`, $SF_HIP_OPTIONS): a key that is not in this table is refused, and so is
// a value outside its range -- the caller's mistake, reported (a shape no kernel can serve is not: that group falls
// back to the generic kernel).  `slab` is the one option with a text value.  Round 5 froze this list: the switches of
// rounds 1-4's measurements (kernel variants that were never planned, timing-only builds with wrong results) are gone
// from the library; what each measured is in NOTES.md, and timing-only code objects are built outside the library
// (tools/whatif_objects.py).
struct OptionSpec {
  const char* key;
  long long lo, hi;
  const char* what;
};
static const OptionSpec kOptions[] = {
    {"fuse", 1, 8, "operators per fused launch (default: 2 for 3-D float32, 3 for float64, 4 for 2-D)"},
    {"reorder", 0, 1, "plan the operators depth first, the operators of one branch next to each other (default 1)"},
    {"dag", 0, 1, "fuse forks and joins into DAG groups (default 1)"},
    {"dag.windows", 1, 8, "register windows a DAG group may hold (default: 6 in 2-D, fuse + 1 in 3-D)"},
    {"star", 0, 1, "kernel family switch: fused radius-1 star groups, kernels/star3d.h (default 1)"},
    {"compact", 0, 1, "kernel family switch: fused groups over {-1,0,1}^3, kernels/compact3d.h (default 1)"},
    {"wide", 0, 1, "kernel family switch: fused radius-2 star groups, kernels/wstar3d.h (default 1)"},
    {"dense", 0, 1, "kernel family switch: dense neighbourhoods and plain sums, kernels/dense3d.h (default 1)"},
    {"dense.t2", 0, 3, "two plain sums per dense launch (radius-1 boxes; float32 3-D sums of few terms within two points): 0 never, 1 where a tile fits the grid (default; chains of radius-1 star sums three per launch), 2 wherever one compiles, 3 as 2 and radius-1 sums in any order of their terms, stars too, two or three per launch"},
    {"generic_only", 0, 1, "every operator on the generic kernel, one per launch"},
    {"k1.bx", 0, 1024, "pin the tile shape, 3-D: lanes per row (with k1.by and k1.rj)"},
    {"k1.by", 0, 64, "pin the tile shape, 3-D: thread rows"},
    {"k1.rj", 0, 16, "pin the tile shape, 3-D: rows per thread"},
    {"k2.bx", 0, 1024, "pin the tile shape, 2-D: lanes per block"},
    {"k1.vk", 1, 4, "elements per lane and row: 1, 2 or 4 (default: 16 bytes, less where rows do not hold whole vectors)"},
    {"k1.li", 0, 1 << 30, "planes per chunk of a 3-D launch (default: the launch-geometry model)"},
    {"k2.li", 0, 1 << 30, "rows per chunk of a 2-D launch"},
    {"k1.nt", 0, 5, "cache policy bits: 1 non-temporal stores, 4 non-temporal loads of unshared rows (default: 5 beyond 256 MiB)"},
    {"allow_spills", 0, 1, "accept a PINNED tile shape whose code object spills (experiments)"},
    {"autotune", 0, 8, "time the first k clean tile shapes on the device and keep the fastest"},
    {"graph", 0, 1, "replay the chain as one hipGraph (default: launch-bound chains only)"},
    {"profile", 0, 1, "HIP events around every launch (sf_plan_kernel_launch_times)"},
    {"debug", 0, 1, "planner trace on stderr"},
    {"slab", 0, 0, "<lo>:<hi>:<halo>[:<extent>] -- this plan computes planes [lo, hi) of a decomposed run"},
};

std::string describe_options() {
  std::ostringstream o;
  for (const OptionSpec& sp : kOptions) o << sp.key << (std::strcmp(sp.key, "slab") == 0 ? "=" : "=<int>  ") << sp.what << "\n";
  return o.str();
}

static void validate_options(const sf_plan& pl) {
  for (auto& kv : pl.opt.kv) {
    const OptionSpec* spec = nullptr;
    for (const OptionSpec& sp : kOptions)
      if (kv.first == sp.key) spec = &sp;
    if (!spec)
      throw Error(SF_ERR_INVALID, "unknown plan option '" + kv.first + "' (sf_describe_options lists the " +
                                      std::to_string(sizeof(kOptions) / sizeof(kOptions[0])) + " there are)");
    if (kv.first == "slab") continue;
    long long v = 0;
    try {
      size_t used = 0;
      v = std::stoll(kv.second, &used);
      if (used != kv.second.size()) throw std::invalid_argument("trailing text");
    } catch (const std::exception&) {
      throw Error(SF_ERR_INVALID, "option " + kv.first + " takes an integer, not '" + kv.second + "'");
    }
    if (v < spec->lo || v > spec->hi)
      throw Error(SF_ERR_INVALID, std::string("option ") + spec->key + " must lie in [" + std::to_string(spec->lo) + ", " +
                                      std::to_string(spec->hi) + "]");
  }
  if (pl.opt.kv.count("k1.vk") && pl.opt.get("k1.vk", 4) == 3) throw Error(SF_ERR_INVALID, "k1.vk must be 1, 2 or 4");
  if (pl.opt.kv.count("k1.nt") && (pl.opt.get("k1.nt", 0) & ~5) != 0)
    throw Error(SF_ERR_INVALID, "k1.nt: 1 (non-temporal stores), 4 (non-temporal loads of unshared rows) or 5");
}

// ---- DAG groups (kernels/star3d.h: stages over register windows) ---------------------------------------------
// What a kernel reads as a stage of a group whose input field is F and whose earlier stages produced `inside`:
// up to two of those fields through radius-1 stars (`primary`, `second`), every other field at the point itself
// from memory.  False if the kernel cannot be such a stage.
struct DagShape {
  std::string primary, second;
  std::vector<std::string> aux;
  std::map<std::string, std::string> bc;  // source field -> the boundary constant this kernel declares for it ("" none)
};

static bool dag_stage_shape(const Program& P, const Kernel& K, const std::string& F, const std::map<std::string, int>& inside,
                            DagShape* out) {
  if (K.acc.empty() || (K.dt != DT::F32 && K.dt != DT::F64)) return false;
  DagShape sh;
  std::vector<std::string> sources;
  for (auto& a : K.acc) {
    const Field& f = P.field(a.field);
    const bool source = a.field == F || inside.count(a.field) != 0;
    int nz = 0;
    for (int d = 0; d < 3; ++d) {
      if (a.off[d] != 0) ++nz;
      if (a.off[d] < -1 || a.off[d] > 1) return false;
    }
    if (source) {
      if (f.dt != K.dt || !f.full() || nz > 1) return false;
      if (P.n[1] == 1 && a.off[1] != 0) return false;
      if (nz == 1 && a.bckind != "copy") {
        if (a.bckind != "constant" && a.bckind != "shrink") return false;
        if (!literal_exact_in(a.bcval, K.dt)) return false;
        auto it = sh.bc.find(a.field);
        if (it != sh.bc.end() && it->second != a.bcval) return false;
        sh.bc[a.field] = a.bcval;
      }
      if (std::find(sources.begin(), sources.end(), a.field) == sources.end()) sources.push_back(a.field);
    } else {
      if (nz != 0) return false;
      if (f.dt != DT::F32 && f.dt != DT::F64) return false;
      if (!f.full() && f.has[0] + f.has[1] + f.has[2] == 0) return false;
      if (std::find(sh.aux.begin(), sh.aux.end(), a.field) == sh.aux.end()) sh.aux.push_back(a.field);
    }
  }
  if (sources.empty() || sources.size() > 2) return false;
  sh.primary = sources[0];
  if (sources.size() == 2) sh.second = sources[1];
  if (out) *out = sh;
  return true;
}

// The largest DAG group that starts at kernel k0 and takes kernels k0, k0 + 1, ... in program order: every stage
// reads fields of the level before its own (the window of a field holds three planes: a reader lags its source by
// exactly one step), depth <= max_depth, windows <= max_windows, auxiliary fields <= kMaxStarAux.  Returns the kernels in
// stage order (level by level) with the description codegen needs; `count` = how many program kernels were taken.
struct DagGroup {
  std::vector<int> kernels;  // stage order
  StarDag dag;
  std::vector<std::string> aux;
  int count = 0;
};

static bool build_dag_group(const Program& P, int k0, int count, const std::map<std::string, std::vector<int>>& readers,
                            DagGroup* out) {
  const int K = (int)P.kernels.size();
  if (k0 + count > K) return false;
  StarShape first;
  if (!star_eligible(P, P.kernels[k0], &first)) return false;
  const std::string F = first.primary;
  const DT dt = P.kernels[k0].dt;
  std::map<std::string, int> inside;  // field -> index in `taken`
  std::vector<int> taken;
  std::vector<DagShape> shapes;
  std::vector<int> depth;
  std::map<std::string, std::string> window_bc;  // source field -> the constant its readers declare
  std::set<std::string> aux;
  for (int n = k0; n < k0 + count; ++n) {
    const Kernel& Kn = P.kernels[n];
    if (Kn.dt != dt) return false;
    DagShape sh;
    if (!dag_stage_shape(P, Kn, F, inside, &sh)) return false;
    auto level_of = [&](const std::string& f) { return f == F ? 0 : depth[inside.at(f)]; };
    const int d = level_of(sh.primary) + 1;
    if (!sh.second.empty() && level_of(sh.second) + 1 != d) return false;  // (both sources one level below)
    for (auto& kv : sh.bc) {
      auto it = window_bc.find(kv.first);
      if (it != window_bc.end() && it->second != kv.second) return false;  // readers of one window disagree
      window_bc[kv.first] = kv.second;
    }
    for (auto& f : sh.aux) {
      if (inside.count(f)) return false;
      aux.insert(f);
    }
    if ((int)aux.size() > kMaxStarAux) return false;
    inside[Kn.name] = (int)taken.size();
    taken.push_back(n);
    shapes.push_back(sh);
    depth.push_back(d);
  }
  // stage order: level by level (the readers of the input window first), program order within a level
  std::vector<int> order(taken.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return depth[a] < depth[b]; });
  DagGroup g;
  g.count = count;
  g.dag.on = true;
  std::map<std::string, int> window_of;  // field -> register window
  window_of[F] = 0;
  int nwin = 1;
  // a stage fills a window if a later stage of the group reads its field
  std::set<std::string> read_inside;
  for (auto& sh : shapes) {
    if (sh.primary != F) read_inside.insert(sh.primary);
    if (!sh.second.empty() && sh.second != F) read_inside.insert(sh.second);
  }
  for (int i : order)
    if (read_inside.count(P.kernels[taken[i]].name)) window_of[P.kernels[taken[i]].name] = nwin++;
  int last_input_reader = -1;
  for (size_t pos = 0; pos < order.size(); ++pos) {
    const int i = order[pos];
    const Kernel& Kn = P.kernels[taken[i]];
    StarDag::Stage st;
    st.primary = shapes[i].primary;
    st.second = shapes[i].second;
    st.src = window_of.at(st.primary);
    st.src2 = st.second.empty() ? -1 : window_of.at(st.second);
    st.depth = depth[i];
    auto w = window_of.find(Kn.name);
    st.dst = w == window_of.end() ? -1 : w->second;
    // materialised: a program output, or read by a kernel outside the group
    bool outside = P.field(Kn.name).role == Role::Output;
    auto rd = readers.find(Kn.name);
    if (rd != readers.end())
      for (int r : rd->second)
        if (r < k0 || r >= k0 + count) outside = true;
    if (outside) {
      st.out = (int)g.dag.outputs.size();
      g.dag.outputs.push_back(Kn.name);
    }
    if (st.dst < 0 && st.out < 0) return false;  // (a dead operator: left to the plain paths)
    if (st.src == 0 || st.src2 == 0) last_input_reader = (int)pos;
    g.dag.depth = std::max(g.dag.depth, st.depth);
    g.dag.stages.push_back(st);
    g.kernels.push_back(taken[i]);
  }
  if (last_input_reader < 0 || g.dag.outputs.empty() || g.dag.outputs.size() > 4) return false;
  g.dag.stages[last_input_reader].refill = true;
  g.dag.nwin = nwin;
  g.dag.nout = (int)g.dag.outputs.size();
  g.aux.assign(aux.begin(), aux.end());
  if (out) *out = g;
  return true;
}

// is the group a plain chain (what the star path plans without this machinery)?
static bool dag_is_chain(const DagGroup& g) {
  for (size_t s = 0; s < g.dag.stages.size(); ++s) {
    const StarDag::Stage& st = g.dag.stages[s];
    if (st.src != (int)s || st.src2 >= 0 || st.depth != (int)s + 1) return false;
    if (s + 1 < g.dag.stages.size() ? (st.dst != (int)s + 1 || st.out >= 0) : (st.dst >= 0 || st.out != 0)) return false;
  }
  return true;
}

// Depth-first topological order of the operators: an operator is followed by a consumer that has become ready,
// so the operators of one branch of a fork (bin/synthesize.py:228-253 emits b3a0 b3b0 b3a1 b3b1 -- the order of
// networkx' topological sort, reference sdfg_generator.py:638) sit next to each other and fuse like any chain.
// Any topological order gives the same results (every operator is a pure function of complete fields); a chain
// keeps its order.  Option reorder=0 keeps the order of the program record.
static void reorder_depth_first(Program& P) {
  const int K = (int)P.kernels.size();
  std::map<std::string, int> producer;
  for (int k = 0; k < K; ++k) producer[P.kernels[k].name] = k;
  std::vector<std::vector<int>> consumers(K);
  std::vector<int> waiting(K, 0);
  for (int k = 0; k < K; ++k) {
    std::set<int> deps;
    for (auto& a : P.kernels[k].acc) {
      auto it = producer.find(a.field);
      if (it != producer.end() && it->second != k) deps.insert(it->second);
    }
    waiting[k] = (int)deps.size();
    for (int d : deps) consumers[d].push_back(k);
  }
  std::vector<int> order;
  std::vector<char> done(K, 0);
  std::vector<int> stack;
  for (int seed = 0; seed < K; ++seed) {
    if (done[seed] || waiting[seed] != 0) continue;
    stack.push_back(seed);
    while (!stack.empty()) {
      const int k = stack.back();
      stack.pop_back();
      if (done[k] || waiting[k] != 0) continue;
      done[k] = 1;
      order.push_back(k);
      // consumers in program order: the first one is taken next (pushed last)
      for (auto it = consumers[k].rbegin(); it != consumers[k].rend(); ++it) {
        if (--waiting[*it] == 0) stack.push_back(*it);
      }
    }
  }
  if ((int)order.size() != K) return;  // (a cycle: left to the checks that report it)
  bool same = true;
  for (int i = 0; i < K; ++i) same = same && order[i] == i;
  if (same) return;
  std::vector<Kernel> sorted;
  sorted.reserve(K);
  for (int k : order) sorted.push_back(P.kernels[k]);
  P.kernels.swap(sorted);
}

void build_plan(sf_plan& pl) {
  check_pinned_compiler();
  if (pl.opt.get("reorder", 1) != 0) reorder_depth_first(pl.P);
  const Program& P = pl.P;
  const int K = (int)P.kernels.size();
  pl.profile = pl.opt.get("profile", 0) != 0;
  validate_options(pl);

  // slab of the stream dimension owned by this plan
  pl.n_local = P.n[0];
  pl.goff = 0;
  pl.halo = 0;
  {
    const std::string slab = pl.opt.gets("slab", "");
    if (!slab.empty()) {
      long long lo, hi, extent = 0;
      int h;
      const int got = std::sscanf(slab.c_str(), "%lld:%lld:%d:%lld", &lo, &hi, &h, &extent);
      if (got < 3 || lo < 0 || hi > P.n[0] || lo >= hi || h < 0 || (got == 4 && extent < 1))
        throw Error(SF_ERR_INVALID, "option slab=<lo>:<hi>:<halo>[:<planning extent>] out of range");
      pl.n_local = hi - lo;
      pl.goff = lo;
      pl.halo = h;
      // The ranks of a decomposed run must plan ALIKE (same launch groups, same reach:
      // they exchange the same planes): everything the planner derives from the slab's
      // height uses this common extent -- the thinnest slab, given by the caller -- not
      // the rank's own height.
      pl.plan_extent = got == 4 ? extent : pl.n_local;
    }
  }

  // consumers per field
  std::map<std::string, int> consumers;
  for (auto& k : P.kernels) {
    std::set<std::string> seen;
    for (auto& a : k.acc)
      if (seen.insert(a.field).second) consumers[a.field]++;
  }

  // default fusion depth (measured, profiles/r01_sweep_*): 2 for 3-D f32 (register
  // budget), 4 for 2-D (3 values of state per stage), 3 for f64 chains
  long long fuse_default = 2;
  if (P.n[1] == 1) fuse_default = 4;
  else if (P.kernels[0].dt == DT::F64) fuse_default = 3;
  const int fuse = (int)std::max<long long>(1, pl.opt.get("fuse", fuse_default));
  // Two (or three) consecutive plain sums in one launch of the dense kernel's fused streaming form (select_dense_t2):
  // operators k0, k0 + 1 (, k0 + 2) qualify pairwise, every field between them is a temporary with one reader.
  // `need_reach`: 0, or what the group must reach per operator.
  const long long t2mode = pl.opt.get("dense", 1) != 0 ? pl.opt.get("dense.t2", 1) : 0;
  std::function<bool(Step&, int, int, int, int, int)> stream_group_n;
  // (`how`: 1 = the terms in any order, 2 = a factor per term allowed too -- what the star and wide-star branches pass)
  auto stream_group = [&](Step& st, int k0, int how, int max_ops, int need_reach, int min_ops = 2) {
    return stream_group_n(st, k0, how, max_ops, need_reach, min_ops);
  };
  // (dense.t2=3: up to three per launch unless fuse= says two)
  const int stream_depth = t2mode >= 3 ? (int)std::max<long long>(2, std::min<long long>(3, pl.opt.get("fuse", 3))) : 2;
  const bool generic_only = pl.opt.get("generic_only", 0) != 0;
  // (any row length: the vector width follows it, rank_star_cfgs)
  const bool star_ok_dims = (P.nd >= 2) && P.n[0] > 1;

  // ---- DAG groups (round 4): forks, joins and intermediates with several readers inside one launch
  // readers of every field, by kernel index
  std::map<std::string, std::vector<int>> readers;
  for (int k = 0; k < K; ++k) {
    std::set<std::string> seen;
    for (auto& a : P.kernels[k].acc)
      if (seen.insert(a.field).second) readers[a.field].push_back(k);
  }
  // register windows a DAG group may hold (option dag.windows; a chain of depth T holds T).  Measured on the
  // generator's fork / join programs (profiles/r04_dag_fork_perf.log): 2-D kernels have registers to spare (one
  // row per thread) -- six windows hold a fork, both branches, the join and the operator after it: 1.40 ->
  // 2.11e6 Mcells/s; 3-D f32 with a third window evaluates both branches of a fork from one read (12-row tiles:
  // 1.19 -> 1.28e6); 3-D f64 with a fourth 5.4 -> 7.4e5.
  const bool dag_on = pl.opt.get("dag", 1) != 0 && !generic_only && star_ok_dims && pl.opt.get("star", 1) != 0;
  const int dag_windows = (int)pl.opt.get("dag.windows", P.n[1] == 1 ? 6 : fuse + 1);
  // the chain the star path would form at k (its conditions, no compilation), or 1
  auto star_chain_len = [&](int k) {
    StarShape sh;
    if (!star_eligible(P, P.kernels[k], &sh)) return 1;
    int len = 1;
    std::set<std::string> aux(sh.aux.begin(), sh.aux.end());
    while (len < fuse && k + len < K) {
      const Kernel& kc = P.kernels[k + len - 1];
      StarShape ns;
      if (!star_eligible(P, P.kernels[k + len], &ns) || ns.primary != kc.name) break;
      if (P.field(kc.name).role != Role::Temp || consumers[kc.name] != 1 || P.kernels[k + len].dt != kc.dt) break;
      bool ok = true;
      for (auto& f : ns.aux)
        for (int g = k; g < k + len; ++g)
          if (P.kernels[g].name == f) ok = false;
      aux.insert(ns.aux.begin(), ns.aux.end());
      if (!ok || (int)aux.size() > kMaxStarAux) break;
      ++len;
    }
    return len;
  };
  // the group the compact path would form at k (its conditions, no compilation), or 0 where it does not apply
  auto compact_chain_len = [&](int k) {
    const bool compact_dims = ((P.nd == 3 && P.n[1] > 1) || P.nd == 2) && P.n[0] > 1 && pl.opt.get("compact", 1) != 0;
    CompactShape cs;
    if (generic_only || !compact_dims || !compact_eligible(P, P.kernels[k], &cs)) return 0;
    auto diagonal = [](const CompactShape& sh) {
      return compact_lateral(sh.need, 0) || compact_lateral(sh.need, 2) || compact_lateral(sh.xneed, 0) ||
             compact_lateral(sh.xneed, 2);
    };
    bool group_diagonal = diagonal(cs);
    std::set<std::string> extras;
    if (!cs.extra.empty()) extras.insert(cs.extra);
    int len = 1;
    const int cfuse = (P.kernels[k].dt == DT::F64 && P.nd == 3 && !pl.opt.kv.count("fuse")) ? std::min(fuse, 2) : fuse;
    while (len < cfuse && k + len < K) {
      const Kernel& kc = P.kernels[k + len - 1];
      CompactShape ns;
      if (!compact_eligible(P, P.kernels[k + len], &ns, kc.name)) break;
      if (P.nd == 2 && !pl.opt.kv.count("fuse") && len >= 2 && (group_diagonal || diagonal(ns))) break;
      group_diagonal = group_diagonal || diagonal(ns);
      if (P.field(kc.name).role != Role::Temp || consumers[kc.name] != 1 || P.kernels[k + len].dt != kc.dt) break;
      if (!ns.extra.empty()) {
        bool produced_inside = false;
        for (int g = k; g < k + len; ++g)
          if (P.kernels[g].name == ns.extra) produced_inside = true;
        if (produced_inside) break;
        extras.insert(ns.extra);
        if ((int)extras.size() > kMaxStarAux) break;
      }
      ++len;
    }
    return len;
  };
  // field passes (reads + writes of whole fields) of the chain-only plan from kernel k until it has covered `until`
  auto chain_plan_cost = [&](int k, int until, int* end) {
    double cost = 0;
    while (k < until && k < K) {
      StarShape sh;
      if (star_eligible(P, P.kernels[k], &sh)) {
        const int len = star_chain_len(k);
        std::set<std::string> aux;
        for (int g = k; g < k + len; ++g) {
          StarShape gs;
          star_eligible(P, P.kernels[g], &gs);
          aux.insert(gs.aux.begin(), gs.aux.end());
        }
        cost += 2.0 + (double)aux.size();
        k += len;
      } else {
        std::set<std::string> fields;
        for (auto& a : P.kernels[k].acc) fields.insert(a.field);
        cost += 1.0 + (double)fields.size();
        k += 1;
      }
    }
    if (end) *end = k;
    return cost;
  };

  // ---- group kernels into launches
  std::map<std::string, StarChoice> star_memo;
  std::set<int> star_first;  // operators whose longer compact group did not come out: the star path after all
  stream_group_n = [&](Step& st, int k0, int how, int max_ops, int need_reach, int min_ops) {
    const bool any_order = (how & 1) != 0, weighted = (how & 2) != 0;
    if (t2mode == 0 || generic_only) return false;
    int n = 1;
    while (n < max_ops && k0 + n < K) {
      int reach = 1;
      const Kernel& prev = P.kernels[k0 + n - 1];
      if (!dense_t2_eligible(P, prev, P.kernels[k0 + n], &reach, any_order, weighted) || (need_reach != 0 && reach != need_reach)) break;
      if (P.field(prev.name).role != Role::Temp || consumers[prev.name] != 1) break;
      ++n;
    }
    for (; n >= min_ops; --n) {
      std::vector<int> group;
      for (int i = 0; i < n; ++i) group.push_back(k0 + i);
      StarChoice choice = select_dense_t2(pl, star_memo, group, P.kernels[k0].dt);
      if (!choice.ok) continue;
      st.star = true;
      st.dense = true;
      st.kernels = group;
      st.cfg = choice.cfg;
      st.ck = choice.ck;
      st.alts = choice.alts;
      st.sig = choice.sig;
      return true;
    }
    return false;
  };
  for (int k = 0; k < K;) {
    Step st;
    StarShape shape;
    // (star=0: diagnostics -- star chains then run on the compact kernel)
    bool star = !generic_only && star_ok_dims && star_eligible(P, P.kernels[k], &shape) && pl.opt.get("star", 1) != 0;
    if (star && P.n[1] == 1) {
      for (auto& a : P.kernels[k].acc)
        if (a.off[1] != 0) star = false;
    }
    // A star chain that ends early because the NEXT operator is not a star (the generator's operators with a second
    // spatial field, a box after a cross) while the compact kernel -- whose 27 offsets include every star -- could take
    // both: the longer group wins, it saves a write and a read of the field between them (round 4; the generator's
    // `num_fields_spatial 0.5` chains: 5 launches -> 4).  The estimate is compile-free; if the compact group then
    // does not come out -- no clean tile, or cut back to no more operators than the star chain -- the operator is planned
    // again on the star path instead of dropping to the generic kernel.
    int demoted_from = 0;
    if (star && !star_first.count(k)) {
      const int ls = star_chain_len(k);
      if (ls < fuse && compact_chain_len(k) > ls) {
        star = false;
        demoted_from = ls;
      }
    }
    // radius-2 stars (bin/synthesize.py with an extent of 2): kernels/wstar3d.h, two fused by default
    // (ten planes of register window per thread: deeper groups shrink the tile too far)
    const bool wide = !generic_only && star_ok_dims && pl.opt.get("wide", 1) != 0 && wide_eligible(P, P.kernels[k]);
    // ... unless the operator and the next one are plain sums of few terms (the generator's crosses): the dense kernel's
    // fused streaming form (round 5: no register windows, no lanes recomputed beyond the tile's rim; select_dense_t2)
    // (float32, three dimensions: the radius-2 cross 512^3 runs 249 us per launch of two against 303 on the wide-star
    //  kernel; float64 524-598 against 424 -- profiles/r05_cross2_fused.log)
    bool wide_pair = wide && P.n[1] > 1 && P.kernels[k].dt == DT::F32 && pl.opt.get("fuse", 2) >= 2 && stream_group(st, k, 2, 2, 0);
    // Chains of radius-1 star sums (the benchmark's jacobi3d), float32, three dimensions: THREE per launch of the same
    // form where a tile shape fits the grid -- 91.0 against 97.9 us per operator at 512^3 on one box (1.47 against
    // 1.37e6 Mcells/s), ahead on every grid tried from 128^3 to 512x256x1024 (profiles/r05_c3_streaming.log); pairs, and
    // plans whose depth or tile shape the caller chose (fuse=, k1.bx ...), stay on the star kernel.  dense.t2=3: pairs
    // too, on any grid.
    const bool caller_tuned = pl.opt.kv.count("fuse") || pl.opt.kv.count("k1.bx") || pl.opt.kv.count("k1.by") || pl.opt.kv.count("k1.rj") ||
                              pl.opt.kv.count("k1.vk");
    if (!wide && star && P.n[1] > 1 && P.kernels[k].dt == DT::F32 && !star_first.count(k)) {
      if (t2mode >= 3) {
        wide_pair = stream_group(st, k, 3, stream_depth, 0);
      } else if (t2mode >= 1 && !caller_tuned) {
        // (a launch of three costs about 1.5 launches of two, a lone operator as much as two: three now unless that leaves
        //  one operator over -- a chain of four is two pairs)
        int left = 1;
        while (left < 5 && k + left < K && dense_t2_eligible(P, P.kernels[k + left - 1], P.kernels[k + left], nullptr, true, true) &&
               P.field(P.kernels[k + left - 1].name).role == Role::Temp && consumers[P.kernels[k + left - 1].name] == 1)
          ++left;
        if (left >= 3 && left != 4) wide_pair = stream_group(st, k, 3, 3, 0, 3);
      }
    }
    if (wide_pair) {
      // (planned above)
    } else if (wide) {
      std::vector<int> group{k};
      // (f64 too since round 4: 64x8 threads x 5 rows, 238 registers -- 3.4 -> 6.3e5 Mcells/s on the generator's
      // radius-2 cross 512^3, profiles/r04_c5_tiles_wide_f64.log)
      const int wfuse = (int)std::max<long long>(1, pl.opt.get("fuse", 2));
      while ((int)group.size() < wfuse && k + (int)group.size() < K) {
        const int cur = group.back(), nxt = cur + 1;
        const Kernel& kc = P.kernels[cur];
        std::string nprimary;
        if (!wide_eligible(P, P.kernels[nxt], &nprimary) || nprimary != kc.name) break;
        if (P.field(kc.name).role != Role::Temp || consumers[kc.name] != 1 || P.kernels[nxt].dt != kc.dt) break;
        group.push_back(nxt);
      }
      StarChoice choice;
      while (!group.empty()) {
        choice = select_wide(pl, star_memo, group, P.kernels[k].dt);
        if (choice.ok) break;
        group.pop_back();
      }
      if (choice.ok) {
        st.star = true;
        st.wide = true;
        st.kernels = group;
        st.cfg = choice.cfg;
        st.ck = choice.ck;
        st.alts = choice.alts;
        st.sig = choice.sig;
      } else {
        st.kernels.push_back(k);
      }
    } else if (star) {
      // a DAG group first: the largest one that moves fewer field passes per operator than the chains it replaces
      // (the comparison runs to the end of the last chain either plan would form, so that cutting a branch in
      // two to fill a group does not pass for a gain)
      bool dagged = false;
      if (dag_on) {
        const int chain_len = star_chain_len(k);
        for (int count = std::min(K - k, 12); count > chain_len && !dagged; --count) {
          DagGroup g;
          if (!build_dag_group(P, k, count, readers, &g) || dag_is_chain(g)) continue;
          if (g.dag.depth > fuse || g.dag.nwin > dag_windows) continue;
          int end_chain = k, end_dag = k + count;
          const double cost_chain = chain_plan_cost(k, k + count, &end_chain);
          const double cost_dag = 1.0 + (double)g.aux.size() + (double)g.dag.nout +
                                  (end_dag < end_chain ? chain_plan_cost(end_dag, end_chain, &end_dag) : 0.0);
          if (cost_dag / (double)(end_dag - k) >= cost_chain / (double)(end_chain - k) - 1e-9) continue;
          StarChoice choice = select_star(pl, star_memo, g.kernels, P.kernels[k].dt, &g.dag);
          if (!choice.ok) continue;
          st.star = true;
          st.kernels = g.kernels;
          st.cfg = choice.cfg;
          st.ck = choice.ck;
          st.alts = choice.alts;
          st.sig = choice.sig;
          // (groups of the same structure share the compiled kernel and the memoised choice; the field names in
          // the description are this group's)
          st.cfg.dag = g.dag;
          for (auto& alt : st.alts) alt.first.dag = g.dag;
          st.out_names = g.dag.outputs;
          dagged = true;
          if (pl.opt.get("debug", 0) != 0) {
            std::fprintf(stderr, "[sf_hip] DAG group at %s: %d operators, depth %d, %d windows, outputs", P.kernels[k].name.c_str(),
                         count, g.dag.depth, g.dag.nwin);
            for (auto& o : g.dag.outputs) std::fprintf(stderr, " %s", o.c_str());
            std::fprintf(stderr, " (passes %.0f over %d operators against %.0f over %d for chains)\n", cost_dag, end_dag - k,
                         cost_chain, end_chain - k);
          }
        }
      }
      std::vector<int> group{k};
      std::set<std::string> group_aux(shape.aux.begin(), shape.aux.end());
      if (dagged) group.clear();
      while (!dagged && (int)group.size() < fuse && k + (int)group.size() < K) {
        const int cur = group.back(), nxt = cur + 1;
        const Kernel& kc = P.kernels[cur];
        StarShape nshape;
        if (!star_eligible(P, P.kernels[nxt], &nshape)) break;
        if (nshape.primary != kc.name) break;
        if (P.field(kc.name).role != Role::Temp) break;
        if (consumers[kc.name] != 1) break;
        if (P.kernels[nxt].dt != kc.dt) break;
        // auxiliary fields must exist in memory: not produced inside this group
        bool aux_ok = true;
        for (auto& f : nshape.aux)
          for (int g : group)
            if (P.kernels[g].name == f) aux_ok = false;
        if (!aux_ok) break;
        // one launch takes kMaxStarAux auxiliary pointers
        std::set<std::string> all_aux = group_aux;
        all_aux.insert(nshape.aux.begin(), nshape.aux.end());
        if ((int)all_aux.size() > kMaxStarAux) break;
        group_aux.swap(all_aux);
        group.push_back(nxt);
      }
      // longest prefix of the group for which a clean kernel exists
      StarChoice choice;
      while (!group.empty()) {
        choice = select_star(pl, star_memo, group, P.kernels[k].dt);
        if (choice.ok) break;
        group.pop_back();
      }
      if (dagged) {
        // (planned above)
      } else if (choice.ok) {
        st.star = true;
        st.kernels = group;
        st.cfg = choice.cfg;
        st.ck = choice.ck;
        st.alts = choice.alts;
        st.sig = choice.sig;
      } else {
        st.kernels.push_back(k);
      }
    } else {
      // compact operators (27-point neighbourhoods, one extra streamed field).  In a slab
      // run an extra field is one more slab-split field the launch reads across planes:
      // the runner exchanges every such field at the launch's reach (SlabRunner's rule for
      // launches that are not a pure chain; the extra field reaches at most T planes)
      const bool whole_domain = true;
      const bool compact_dims = ((P.nd == 3 && P.n[1] > 1) || P.nd == 2) && P.n[0] > 1 && pl.opt.get("compact", 1) != 0;
      CompactShape cshape;
      bool compact = !generic_only && compact_dims && compact_eligible(P, P.kernels[k], &cshape) &&
                     (cshape.extra.empty() || whole_domain);
      // two plain radius-1 sums (the generator's 27-point boxes): the dense kernel's fused streaming form where the pair
      // qualifies and a tile shape fits the grid (dense.t2, select_dense_t2); else the compact kernel
      const bool dense_pair = compact && cshape.extra.empty() && fuse >= 2 &&
                              stream_group(st, k, t2mode >= 3 ? 1 : 0, (P.kernels[k].dt == DT::F32 && P.n[1] > 1) ? stream_depth : 2, 0);
      if (dense_pair) {
        // (planned above)
      } else if (compact) {
        std::vector<int> group{k};
        std::set<std::string> extras;
        if (!cshape.extra.empty()) extras.insert(cshape.extra);
        // 2-D groups with diagonal accesses (the 9-point box) are fastest two deep
        // (profiles/r02_synth_perf.log: 8.7e5 / 8.6e5 / 6.8e5 Mcells/s at depth 2 / 3 / 4;
        // star-like 2-D groups with extra fields keep the 2-D default of 4); an explicit
        // fuse= option is followed as given
        auto diagonal = [](const CompactShape& sh) {
          return compact_lateral(sh.need, 0) || compact_lateral(sh.need, 2) || compact_lateral(sh.xneed, 0) ||
                 compact_lateral(sh.xneed, 2);
        };
        bool group_diagonal = diagonal(cshape);
        // (float64 groups two deep unless fuse= says otherwise: the 27-point box 512^3 runs 3.2e5 Mcells/s two deep and
        //  1.6e5 three deep -- 247 registers and 9-row tiles --, profiles/r04_box_f64.log; the default of 3 is the star kernel's)
        const int cfuse = (P.kernels[k].dt == DT::F64 && P.nd == 3 && !pl.opt.kv.count("fuse")) ? std::min(fuse, 2) : fuse;
        while ((int)group.size() < cfuse && k + (int)group.size() < K) {
          const int cur = group.back(), nxt = cur + 1;
          const Kernel& kc = P.kernels[cur];
          CompactShape nshape;
          if (!compact_eligible(P, P.kernels[nxt], &nshape, kc.name)) break;
          if (P.nd == 2 && !pl.opt.kv.count("fuse") && (int)group.size() >= 2 && (group_diagonal || diagonal(nshape))) break;
          group_diagonal = group_diagonal || diagonal(nshape);
          if (P.field(kc.name).role != Role::Temp) break;
          if (consumers[kc.name] != 1) break;
          if (P.kernels[nxt].dt != kc.dt) break;
          if (!nshape.extra.empty()) {
            if (!whole_domain) break;
            bool produced_inside = false;
            for (int g : group)
              if (P.kernels[g].name == nshape.extra) produced_inside = true;
            if (produced_inside) break;
            std::set<std::string> all = extras;
            all.insert(nshape.extra);
            if ((int)all.size() > kMaxStarAux) break;
            extras.swap(all);
          }
          group.push_back(nxt);
        }
        StarChoice choice;
        while (!group.empty()) {
          choice = select_compact(pl, star_memo, group, P.kernels[k].dt);
          if (choice.ok) break;
          group.pop_back();
        }
        // A group whose only clean tile recomputes more than it keeps is worse than its operators one by one: boxes
        // with a second spatial field two deep fit 64x4 threads x 3 rows only -- 8 of 12 rows and 512 of 768 columns
        // useful; the next shape 5 of 9 rows -- and run 2.2-2.8e5 Mcells/s where one operator per launch runs 4.3e5
        // (profiles/r04_box_extra.log).
        if (choice.ok && group.size() > 1 && !pl.opt.kv.count("fuse")) {
          auto kept = [&](const StarCfg& c) {  // share of a tile's rows and lanes whose results are stored
            const double rows = c.noj ? 1.0 : (double)(c.BY * c.RJ - 2 * c.T) / (double)(c.BY * c.RJ);
            const double cols = (double)P.n[2] / ((double)std::max(1, c.NKT) * (double)c.BX * (double)c.VK);
            return rows * cols;
          };
          // (grids narrower than a tile waste lanes at any depth and are no benchmark: the comparison -- one more
          //  group to compile -- is for rows at least a tile wide)
          if (P.n[2] >= (long long)choice.cfg.BX * choice.cfg.VK && kept(choice.cfg) < 0.6) {
            // (against what ONE operator's tile keeps on this grid: a small grid wastes lanes at any depth)
            std::vector<int> one{k};
            StarChoice single = select_compact(pl, star_memo, one, P.kernels[k].dt);
            if (single.ok && kept(choice.cfg) < 0.65 * kept(single.cfg)) {
              group = one;
              choice = single;
            }
          }
        }
        if (choice.ok) {
          st.star = true;
          st.compact = true;
          st.kernels = group;
          st.cfg = choice.cfg;
          st.ck = choice.ck;
          st.alts = choice.alts;
          st.sig = choice.sig;
        } else {
          st.kernels.push_back(k);
        }
      } else {
        st.kernels.push_back(k);
      }
      // plain sums of few terms within two points (not stars: those are planned above), float32, three dimensions: two
      // per launch in the dense kernel's fused streaming form
      if (!st.star && st.kernels.size() == 1 && !generic_only && star_ok_dims && P.n[1] > 1 && P.kernels[k].dt == DT::F32 && fuse >= 2)
        stream_group(st, k, 0, 2, 2);
      // dense neighbourhoods of radius 2 (the generator's box of extent 2): one operator per launch, LDS tiles
      const bool dense_r3 = dense_r3_eligible(P, P.kernels[k]);
      if (!st.star && st.kernels.size() == 1 && !generic_only && star_ok_dims && pl.opt.get("dense", 1) != 0 &&
          (dense_eligible(P, P.kernels[k]) || dense_r3)) {
        StarChoice choice = select_dense(pl, star_memo, k, P.kernels[k].dt, dense_r3 ? 3 : 2);
        if (choice.ok) {
          st.star = true;
          st.dense = true;
          st.cfg = choice.cfg;
          st.ck = choice.ck;
          st.alts = choice.alts;
          st.sig = choice.sig;
        }
      }
    }
    if (demoted_from > 0 && !(st.compact && (int)st.kernels.size() > demoted_from)) {
      star_first.insert(k);
      continue;  // (the same operator again, star path; everything compiled on the way is cached)
    }
    k += (int)st.kernels.size();
    pl.steps.push_back(st);
  }

  // ---- buffers with liveness-based reuse (the reference keeps one full-size
  // transient per intermediate, sdfg_generator.py:626-630; a 1000-stage chain
  // needs two)
  auto make_buffer = [&](const Field& f) {
    Buffer b;
    b.dt = f.dt;
    b.slabbed = f.has[0] && P.n[0] > 1;
    size_t plane = size_of(f.dt);
    if (f.has[1]) plane *= (size_t)P.n[1];
    if (f.has[2]) plane *= (size_t)P.n[2];
    if (b.slabbed) {
      b.plane_bytes = plane;
      b.planes = (int)(pl.n_local + 2 * pl.halo);
    } else {
      b.plane_bytes = plane * (f.has[0] ? (size_t)P.n[0] : 1);
      b.planes = 1;
    }
    pl.buffers.push_back(b);
    return (int)pl.buffers.size() - 1;
  };
  std::map<std::string, int> buf_of;   // live field -> buffer
  std::map<std::string, int> last_use; // field -> last step reading it
  auto step_reads = [&](const Step& st) {
    std::vector<std::string> r;
    if (st.wide || st.dense) {
      r.push_back(P.kernels[st.kernels[0]].acc[0].field);  // one field, streamed
    } else if (st.compact) {
      // argument 0: the streamed field of the first stage; then the extra fields in
      // first-use order (as gen_compact numbers them)
      for (size_t si = 0; si < st.kernels.size(); ++si) {
        CompactShape sh;
        compact_eligible(P, P.kernels[st.kernels[si]], &sh, si == 0 ? std::string() : P.kernels[st.kernels[si - 1]].name);
        if (si == 0) r.push_back(sh.primary);
        if (!sh.extra.empty() && std::find(r.begin() + 1, r.end(), sh.extra) == r.end()) r.push_back(sh.extra);
      }
    } else if (st.star && st.cfg.dag.on) {
      // argument 0: the group's input field (window 0); then, in stage order and order of first use, the fields
      // read at the point itself from memory (as gen_star numbers them)
      const StarDag& dag = st.cfg.dag;
      for (auto& ds : dag.stages)
        if (ds.src == 0 && r.empty()) r.push_back(ds.primary);
      for (size_t si = 0; si < st.kernels.size(); ++si)
        for (auto& a : P.kernels[st.kernels[si]].acc)
          if (a.field != dag.stages[si].primary && a.field != dag.stages[si].second &&
              std::find(r.begin() + 1, r.end(), a.field) == r.end())
            r.push_back(a.field);
    } else if (st.star) {
      StarShape sh0;
      star_eligible(P, P.kernels[st.kernels[0]], &sh0);
      r.push_back(sh0.primary);  // argument 0: the streamed field
      // then the auxiliary fields in first-use order; a later stage may name the
      // streamed field itself (centre read at its own plane), which then appears twice
      for (int k : st.kernels) {
        StarShape sh;
        star_eligible(P, P.kernels[k], &sh);
        for (auto& f : sh.aux)
          if (std::find(r.begin() + 1, r.end(), f) == r.end()) r.push_back(f);
      }
    } else {
      for (auto& a : P.kernels[st.kernels[0]].acc)
        if (std::find(r.begin(), r.end(), a.field) == r.end()) r.push_back(a.field);
    }
    return r;
  };
  for (size_t s = 0; s < pl.steps.size(); ++s)
    for (auto& f : step_reads(pl.steps[s])) last_use[f] = (int)s;

  pl.input_buf.assign(P.num_inputs, -1);
  pl.output_buf.assign(P.num_outputs, -1);
  for (auto& f : P.fields)
    if (f.role == Role::Input) {
      const int b = make_buffer(f);
      buf_of[f.name] = b;
      pl.input_buf[f.io_index] = b;
    }
  std::multimap<std::pair<size_t, int>, int> free_pool;  // (bytes, dt) -> buffer
  for (size_t s = 0; s < pl.steps.size(); ++s) {
    Step& st = pl.steps[s];
    for (auto& f : step_reads(st)) {
      auto it = buf_of.find(f);
      if (it == buf_of.end()) throw Error(SF_ERR_INVALID, "field '" + f + "' is read before it is produced");
      st.in_bufs.push_back(it->second);
      st.read_names.push_back(f);
    }
    // the fields the launch materialises (one: the last operator's, unless a DAG group names several)
    if (st.out_names.empty()) st.out_names.push_back(P.kernels[st.kernels.back()].name);
    for (auto& oname : st.out_names) {
      const Field& of = P.field(oname);
      int ob = -1;
      if (of.role == Role::Output) {
        ob = make_buffer(of);
        pl.output_buf[of.io_index] = ob;
      } else {
        Buffer probe;
        {
          // size the candidate without registering it
          const size_t before = pl.buffers.size();
          const int tmp = make_buffer(of);
          probe = pl.buffers[tmp];
          pl.buffers.resize(before);
        }
        auto key = std::make_pair(probe.bytes(), (int)probe.dt);
        auto it = free_pool.find(key);
        if (it != free_pool.end()) {
          ob = it->second;
          free_pool.erase(it);
        } else {
          ob = make_buffer(of);
        }
      }
      st.out_bufs.push_back(ob);
      buf_of[of.name] = ob;
    }
    st.out_buf = st.out_bufs[0];
    // release temporaries whose last reader was this step
    std::set<std::string> released;
    for (auto& f : step_reads(st)) {
      const Field& rf = P.field(f);
      if (rf.role == Role::Temp && last_use[f] == (int)s && released.insert(f).second) {
        const int b = buf_of[f];
        free_pool.insert({{pl.buffers[b].bytes(), (int)pl.buffers[b].dt}, b});
        buf_of.erase(f);
      }
    }
  }

  // ---- generate + compile kernels
  const double cells = (double)pl.n_local * (double)P.n[1] * (double)P.n[2];
  for (auto& st : pl.steps) {
    const DT dt = P.kernels[st.kernels[0]].dt;
    if (st.star) {
      if (star_lds_bytes(st.cfg, dt) > 160 * 1024)
        throw Error(SF_ERR_INVALID, "star kernel: tile needs more than 160 KiB of LDS");
      StarKernelSource g = (st.dense && st.cfg.dense_t2) ? gen_dense_t2(P, st.kernels, st.cfg)
                           : st.dense  ? gen_dense(P, st.kernels[0], st.cfg)
                           : st.wide ? gen_wide(P, st.kernels, st.cfg)
                           : st.compact ? gen_compact(P, st.kernels, st.cfg) : gen_star(P, st.kernels, st.cfg);
      st.scalars = g.scalars;
      st.scalar_offsets = g.scalar_offsets;
      st.scalars_bytes = g.scalars_bytes;
      st.num_aux = (int)g.aux.size();
      // the generator numbers auxiliary pointers in first-use order over the
      // fused stages, which is the order step_reads() lists them after the primary
      {
        const std::vector<std::string> reads = step_reads(st);
        if (reads.size() != g.aux.size() + 1) throw Error(SF_ERR_STATE, "star step: auxiliary count mismatch");
        for (size_t a = 0; a < g.aux.size(); ++a)
          if (reads[a + 1] != g.aux[a]) throw Error(SF_ERR_STATE, "star step: auxiliary order mismatch");
      }
      // (dense3d.h streams planes cb - R .. ce + R whatever the operator reaches along the stream axis, and its
      // plane test knows the global domain only: the slab buffer must hold R ghost planes for it -- ADVICE r03)
      st.halo_depth = st.cfg.T * st.cfg.R;
      st.halo_buf = st.in_bufs[0];
    } else {
      // 4 points per thread with aligned vector loads when rows allow it; the
      // one-point form is kept for short rows and for operators whose vector
      // form would spill (same acceptance rule as for the star kernels)
      const bool vec = P.n[2] % 4 == 0;
      const bool xcd = true;  // (XCD-aware block order)
      // non-temporal output stores for fields beyond the Infinity Cache (see rank_star_cfgs)
      const double out_bytes = (double)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local) * (double)P.n[1] *
                               (double)P.n[2] * (double)size_of(dt);  // (alike on all ranks of a slab run)
      const bool nts = out_bytes >= 256.0 * 1024 * 1024;
      // marching form (a thread walks 8 planes with a register window) for 3-D programs; the one-plane form where
      // the marching one does not compile cleanly
      const bool march = vec && P.n[0] > 1;
      const int ppt = march ? 8 : 1;
      auto make = [&](bool marching) {
        return marching ? gen_generic_march(P, st.kernels[0], xcd, nts, ppt)
               : vec    ? gen_generic_vec(P, st.kernels[0], xcd, nts, 1, false, false)
                        : gen_generic(P, st.kernels[0], xcd, nts);
      };
      GenericKernelSource g = make(march);
      st.ck = intern_kernel(pl, std::string("sf_point_") + short_of(dt), g.source);
      auto unclean = [&](int ck) { return kernel_unsafe(pl.kernels[ck]) || kernel_slow(pl.kernels[ck]); };
      if (march && unclean(st.ck)) {
        g = make(false);
        st.ck = intern_kernel(pl, std::string("sf_point_") + short_of(dt), g.source);
      }
      if (vec && unclean(st.ck)) {
        g = gen_generic(P, st.kernels[0], xcd, nts);
        st.ck = intern_kernel(pl, std::string("sf_point_") + short_of(dt), g.source);
      }
      // the one-point form is the last resort: VGPR spills there are slow but correct,
      // a code object with allocator code ahead of an EXEC restore is not acceptable anywhere
      if (kernel_unsafe(pl.kernels[st.ck]))
        throw Error(SF_ERR_UNSUPPORTED, "operator '" + P.kernels[st.kernels[0]].name +
                                            "': the compiler placed register-allocator code ahead of an EXEC "
                                            "restore in the generated kernel, which gives wrong results on gfx950 "
                                            "(DESIGN.md 5.1); simplify the operator");
      st.generic_vk = g.vk;
      st.generic_ppt = g.planes_per_thread;
      st.scalars = g.scalars;
      int depth = 0, hb = -1;
      for (size_t ai = 0; ai < P.kernels[st.kernels[0]].acc.size(); ++ai) {
        const Access& a = P.kernels[st.kernels[0]].acc[ai];
        if (std::abs(a.off[0]) > depth) {
          depth = std::abs(a.off[0]);
        }
      }
      // exchange descriptor names the first slab-split field read across planes
      for (size_t r = 0; r < g.reads.size(); ++r)
        for (auto& a : P.kernels[st.kernels[0]].acc)
          if (a.field == g.reads[r] && a.off[0] != 0 && hb < 0) hb = st.in_bufs[r];
      st.halo_depth = depth;
      st.halo_buf = hb;
    }
    // a rank with a neighbour reads `halo_depth` planes of that neighbour's slab:
    // they must exist in the local buffers (halo = 0 is only valid for a slab that
    // touches both ends of the global domain)
    const bool has_neighbour = pl.goff > 0 || pl.goff + pl.n_local < P.n[0];
    if (has_neighbour && st.halo_buf >= 0 && st.halo_depth > pl.halo)
      throw Error(SF_ERR_INVALID, "slab halo is shallower than a launch's reach; raise the halo");
    CompiledKernel& ck = pl.kernels[st.ck];
    ck.updates_per_launch = cells * (double)st.kernels.size();
    pl.max_updates_per_launch = std::max(pl.max_updates_per_launch, ck.updates_per_launch);
    ck.alg_bytes_per_launch = 0;
    for (int k : st.kernels) ck.alg_bytes_per_launch += cells * 2.0 * (double)size_of(P.kernels[k].dt);
  }
  pl.description = describe_plan(pl);
  pl.scalar_values.assign(P.num_scalar_inputs, 0.0);
}

}  // namespace sf
