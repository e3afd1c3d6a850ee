// sf_hip.cpp — libsf_hip.so: planner, hipRTC driver and runtime behind the C ABI
// of include/sf_hip.h.  Host code only; every device kernel is generated per
// program (codegen.hpp, kernels/star3d.h) and compiled for gfx950 at plan
// creation, the way the reference compiles each program's SDFG before calling
// it (stencilflow/run_program.py:118-128).
#include "../../include/sf_hip.h"

#include <hip/hip_runtime.h>
#include <amd_comgr/amd_comgr.h>
#include <hip/hiprtc.h>

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <set>

#include "codegen.hpp"

namespace sf {

static thread_local std::string g_last_error;

#define SF_HIP_CHECK(expr)                                                            \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      throw Error(SF_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
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
  double updates_per_launch = 0, alg_bytes_per_launch = 0;
  // from the code object's amdhsa metadata (msgpack note)
  int vgprs = -1, agprs = -1, sgprs = -1, spills = -1, scratch = -1, lds = -1, sgpr_spills = -1;
  int late_exec_restores = 0;  // see count_late_exec_restores()
  bool from_disk = false;  // the code object came from the on-disk cache
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
  std::vector<int> kernels;    // program kernel indices fused in this launch
  int ck = -1;                 // compiled kernel
  std::vector<int> in_bufs;    // argument order
  int out_buf = -1;
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
};

}  // namespace sf

using namespace sf;

struct sf_plan {
  Program P;
  Options opt{nullptr};
  int device = 0;
  bool device_ready = false;
  hipStream_t stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  bool timed = false, profile = false;
  std::vector<CompiledKernel> kernels;
  std::map<std::string, int> kernel_by_source;
  std::vector<Buffer> buffers;
  std::vector<Step> steps;
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
  void* debug_buffer = nullptr;  // diagnostic builds (option stamp=1): 8 x uint64
  // per-launch profiling events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<int> prof_kernel;
};

namespace sf {

// ---------------------------------------------------------------- hipRTC
static void compile_kernel(CompiledKernel& k) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, k.source.c_str(), (k.name + ".hip").c_str(), 0, nullptr, nullptr) !=
      HIPRTC_SUCCESS)
    throw Error(SF_ERR_COMPILE, "hiprtcCreateProgram failed");
  const std::string def = "-DSF_KERNEL_NAME=" + k.name;
  std::vector<std::string> extra = split_ws(k.flags);
  std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", def.c_str()};
  for (auto& f : extra) opts.push_back(f.c_str());
  hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    throw Error(SF_ERR_COMPILE, "hipRTC failed for " + k.name + ":\n" + log);
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  k.code.resize(n);
  hiprtcGetCode(prog, k.code.data());
  hiprtcDestroyProgram(&prog);
}

// Value of an unsigned msgpack integer stored right after the string key `key`
// inside the code object's NT_AMDGPU_METADATA note (-1 if absent).
static int metadata_uint(const std::vector<char>& code, const char* key) {
  const size_t klen = std::strlen(key);
  for (size_t i = 0; i + klen + 1 < code.size(); ++i) {
    if (std::memcmp(&code[i], key, klen) != 0) continue;
    const unsigned char* p = (const unsigned char*)&code[i + klen];
    const unsigned char t = p[0];
    if (t <= 0x7f) return t;
    if (t == 0xcc && i + klen + 1 < code.size()) return p[1];
    if (t == 0xcd && i + klen + 2 < code.size()) return (p[1] << 8) | p[2];
    if (t == 0xce && i + klen + 4 < code.size())
      return (int)(((unsigned)p[1] << 24) | (p[2] << 16) | (p[3] << 8) | p[4]);
  }
  return -1;
}

// The toolchain fault behind the wrong results of "spilling" code objects (ROCm 7.2
// LLVM for gfx950; found and proven in round 2, DESIGN.md §5.1, tools/asm_objects.py):
// after a divergent `if` the compiler restores EXEC at the top of the join block
// (`s_or_b64 exec, exec, s[a:b]`).  Under scalar-register pressure the greedy SGPR
// allocator splits live ranges and puts its split copies (s_mov_b32/b64) or spill
// code (v_readlane / v_writelane) at the top of that block, AHEAD of the restore --
// harmless by themselves.  The VGPR allocator then no longer recognises the restore
// as part of the block's prologue and places ITS copies and spill code ahead of it as
// well, where they run under the narrowed EXEC of the `if` body: lanes (here: whole
// waves, the condition being a thread row) that did not take the branch keep stale
// registers.  Moving the restore back to the top of the block, and nothing else,
// makes every failing object correct; padding every instruction with s_nop changes
// nothing; the basic / fast SGPR allocators (which never split) do not produce it.
// The enabling condition can be read off the machine code: an EXEC restore preceded by
// a run of copy / spill instructions that contains an SGPR copy or an SGPR spill-lane
// access.  A code object that contains it is never run.  (The instructions are told apart with comgr's
// single-instruction disassembler; without one every SGPR-spilling object is refused, the
// proxy that held in all measurements: -1.)
namespace {
struct DisasmCursor {
  const char* base;
  uint64_t size;
  std::string text;
};
uint64_t disasm_read(uint64_t from, char* to, uint64_t size, void* user) {
  auto* c = static_cast<DisasmCursor*>(user);
  if (from >= c->size) return 0;
  const uint64_t n = std::min<uint64_t>(size, c->size - from);
  std::memcpy(to, c->base + from, n);
  return n;
}
void disasm_print(const char* instruction, void* user) { static_cast<DisasmCursor*>(user)->text = instruction; }
void disasm_address(uint64_t, void*) {}
}  // namespace

static int count_late_exec_restores(const std::vector<char>& code) {
  if (code.size() < 64 || std::memcmp(code.data(), "\177ELF", 4) != 0 || code[4] != 2) return 0;
  auto rd = [&](size_t off, int bytes) -> unsigned long long {
    unsigned long long v = 0;
    if (off + bytes > code.size()) return 0;
    std::memcpy(&v, &code[off], bytes);
    return v;
  };
  amd_comgr_disassembly_info_t info;
  if (amd_comgr_create_disassembly_info("amdgcn-amd-amdhsa--gfx950", disasm_read, disasm_print, disasm_address, &info) !=
      AMD_COMGR_STATUS_SUCCESS)
    return -1;
  const size_t shoff = rd(0x28, 8), shentsize = rd(0x3A, 2), shnum = rd(0x3C, 2);
  int hits = 0;
  for (size_t sidx = 0; sidx < shnum; ++sidx) {
    const size_t sh = shoff + sidx * shentsize;
    if (sh + 64 > code.size()) break;
    const unsigned long long type = rd(sh + 4, 4), flags = rd(sh + 8, 8), off = rd(sh + 0x18, 8), size = rd(sh + 0x20, 8);
    if (type != 1 /*SHT_PROGBITS*/ || !(flags & 4 /*SHF_EXECINSTR*/) || off + size > code.size()) continue;
    // classes: R = EXEC restore, S = scalar allocator code (SGPR copy, spill-lane access),
    // V = vector copy / spill code, N = padding, X = anything else
    std::string classes;
    DisasmCursor cur{code.data() + off, size, ""};
    for (uint64_t at = 0; at < size;) {
      uint64_t len = 0;
      cur.text.clear();
      if (amd_comgr_disassemble_instruction(info, at, &cur, &len) != AMD_COMGR_STATUS_SUCCESS || len == 0) {
        classes += 'X';
        at += 4;
        continue;
      }
      at += len;
      const size_t b = cur.text.find_first_not_of(" \t");
      const std::string t = b == std::string::npos ? "" : cur.text.substr(b);
      auto starts = [&](const char* p) { return t.compare(0, std::strlen(p), p) == 0; };
      char c = 'X';
      if (starts("s_or_b64 exec, exec, s[") || starts("s_xor_b64 exec, exec, s[") || starts("s_andn2_b64 exec, exec, s[") ||
          starts("s_or_saveexec_b64 ") || starts("s_andn2_saveexec_b64 "))
        c = 'R';  // end of an `if`, `else` entry (two forms), loop exit: the EXEC updates that open a block
      else if ((starts("s_mov_b32 s") || starts("s_mov_b64 s[") || starts("s_mov_b32 vcc") || starts("s_mov_b64 vcc")) &&
               t.find("exec") == std::string::npos)
        c = 'S';  // a split copy, or a constant: the allocator rematerialises values the same way
      else if (starts("v_readlane_b32 ") || starts("v_writelane_b32 "))
        c = 'S';
      else if (starts("v_mov_b32_e32 ") || starts("v_mov_b64_e32 ") || starts("v_accvgpr_") || starts("scratch_load_") ||
               starts("scratch_store_"))
        c = 'V';
      else if (starts("s_nop") || starts("s_waitcnt"))
        c = 'N';
      classes += c;
    }
    for (size_t i = 0; i < classes.size(); ++i) {
      if (classes[i] != 'R') continue;
      // (no exemption for constants, nor for runs that reach back to where EXEC was narrowed: an object
      // with `s_or_saveexec; s_mov vcc_lo, <constant>; v_mov_b64 copies; s_xor_b64 exec` -- allocator code
      // inside an `else` prologue -- gave wrong results, profiles/r02_config_fuzz_detector.log)
      bool scalar_code = false;
      for (size_t j = i; j-- > 0 && (classes[j] == 'S' || classes[j] == 'V' || classes[j] == 'N');)
        scalar_code = scalar_code || classes[j] == 'S';
      if (scalar_code) ++hits;
    }
  }
  amd_comgr_destroy_disassembly_info(info);
  return hits;
}

static void read_metadata(CompiledKernel& k) {
  k.vgprs = metadata_uint(k.code, ".vgpr_count");
  k.agprs = metadata_uint(k.code, ".agpr_count");
  k.sgprs = metadata_uint(k.code, ".sgpr_count");
  k.spills = metadata_uint(k.code, ".vgpr_spill_count");
  k.sgpr_spills = metadata_uint(k.code, ".sgpr_spill_count");
  k.scratch = metadata_uint(k.code, ".private_segment_fixed_size");
  k.lds = metadata_uint(k.code, ".group_segment_fixed_size");
  // The SGPR allocator splits and spills only once it has run out of registers, and then the
  // object reports all of them in use (106 on gfx950: every object of the probes that shows the
  // fault; the kernels of the benchmarks report 46-89 and show nothing).  Far below that no
  // allocator code exists, and scalar moves next to a restore are what the program says.
  k.late_exec_restores = k.sgprs >= 64 ? count_late_exec_restores(k.code) : 0;
}

// Code objects are cached per process: plans of the same program (slab ranks,
// repeated runs, the tile search of another chain) do not recompile.
static std::mutex g_code_cache_mutex;
static std::map<std::string, std::vector<char>> g_code_cache;  // name + source -> code object

// <dir>/<hash>.co, or "" when the disk cache is off.  Directory: $SF_HIP_CACHE_DIR
// ("off" disables), default $XDG_CACHE_HOME or ~/.cache + /stencilflow_amd.
static std::string disk_cache_path(const std::string& key) {
  const char* env = std::getenv("SF_HIP_CACHE_DIR");
  std::string dir;
  if (env && *env) {
    if (std::string(env) == "off" || std::string(env) == "0") return "";
    dir = env;
  } else {
    const char* xdg = std::getenv("XDG_CACHE_HOME");
    const char* home = std::getenv("HOME");
    if (xdg && *xdg) dir = std::string(xdg) + "/stencilflow_amd";
    else if (home && *home) dir = std::string(home) + "/.cache/stencilflow_amd";
    else return "";
  }
  // mkdir -p (two levels are enough for the defaults)
  const size_t slash = dir.rfind('/');
  if (slash != std::string::npos && slash > 0) ::mkdir(dir.substr(0, slash).c_str(), 0755);
  if (::mkdir(dir.c_str(), 0755) != 0 && errno != EEXIST) return "";
  // the compiler that would produce this object: hipRTC major.minor, the HIP
  // runtime's full version number (patch level included) and the build id of the
  // ROCm headers this library was compiled against
  int major = 0, minor = 0, runtime = 0;
  hiprtcVersion(&major, &minor);
  (void)hipRuntimeGetVersion(&runtime);
  const std::string salted = key + "\nhiprtc " + std::to_string(major) + "." + std::to_string(minor) +
                             " runtime " + std::to_string(runtime) + " build " + HIP_VERSION_GITHASH +
                             "\ngfx950 -O3 -std=c++17 -ffp-contract=off";
  char name[40];
  std::snprintf(name, sizeof name, "%016llx%08x", (unsigned long long)fnv1a(salted), (unsigned)salted.size());
  return dir + "/" + name + ".co";
}

// Cache file = 24-byte header {magic "SFCO0002", payload bytes, FNV-1a of the
// payload} + the code object.  A file that is truncated, damaged or of another
// format is deleted and the kernel recompiled.
static const char kCacheMagic[9] = "SFCO0002";

static bool read_cache_file(const std::string& path, std::vector<char>& code) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  f.close();
  bool ok = blob.size() > 24 && std::memcmp(blob.data(), kCacheMagic, 8) == 0;
  if (ok) {
    uint64_t size = 0, hash = 0;
    std::memcpy(&size, blob.data() + 8, 8);
    std::memcpy(&hash, blob.data() + 16, 8);
    ok = size == blob.size() - 24 && size > 4 && std::memcmp(blob.data() + 24, "\177ELF", 4) == 0 &&
         hash == fnv1a(std::string(blob.data() + 24, blob.size() - 24));
  }
  if (!ok) {
    std::remove(path.c_str());  // stale or corrupt: never hand it to the loader
    return false;
  }
  code.assign(blob.begin() + 24, blob.end());
  return true;
}

static void write_cache_file(const std::string& path, const std::vector<char>& code) {
  const std::string tmp = path + "." + std::to_string((long)getpid());
  std::ofstream f(tmp, std::ios::binary);
  if (!f) return;
  const uint64_t size = code.size(), hash = fnv1a(std::string(code.data(), code.size()));
  f.write(kCacheMagic, 8);
  f.write(reinterpret_cast<const char*>(&size), 8);
  f.write(reinterpret_cast<const char*>(&hash), 8);
  f.write(code.data(), (std::streamsize)code.size());
  f.close();
  if (!f || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

static std::atomic<long> g_cache_hits{0}, g_cache_misses{0}, g_cache_recompiles{0};

static int intern_kernel(sf_plan& pl, const std::string& prefix, const std::string& source,
                         const std::string& flags_in = "") {
  // (diagnostics: $SF_HIP_EXTRA_FLAGS adds compiler flags to every kernel, e.g.
  // "-mllvm -amdgpu-spill-sgpr-to-vgpr=0"; they become part of name and cache key)
  std::string flags = flags_in;
  if (const char* extra = std::getenv("SF_HIP_EXTRA_FLAGS"))
    if (*extra) flags += (flags.empty() ? "" : " ") + std::string(extra);
  // (kernels without extra flags keep the names and cache keys they always had)
  const std::string keyed = flags.empty() ? source : flags + "\n" + source;
  auto it = pl.kernel_by_source.find(keyed);
  if (it != pl.kernel_by_source.end()) return it->second;
  CompiledKernel k;
  k.name = prefix + "_" + hex8(fnv1a(keyed));
  k.source = source;
  k.flags = flags;
  const std::string key = k.name + "\n" + keyed;
  bool cached = false;
  // (diagnostics: $SF_HIP_OBJECT_DIR/<kernel name>.co, a code object assembled by hand --
  // e.g. the compiler's own output with instructions padded or moved, tools/asm_objects.py --
  // takes the place of the compiler's; nothing is cached)
  if (const char* dir = std::getenv("SF_HIP_OBJECT_DIR")) {
    std::ifstream f(std::string(dir) + "/" + k.name + ".co", std::ios::binary);
    if (f) {
      k.code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      read_metadata(k);
      pl.kernels.push_back(std::move(k));
      pl.kernel_by_source[keyed] = (int)pl.kernels.size() - 1;
      return (int)pl.kernels.size() - 1;
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_code_cache_mutex);
    auto c = g_code_cache.find(key);
    if (c != g_code_cache.end()) {
      k.code = c->second;
      cached = true;
    }
  }
  if (!cached) {
    // second level: code objects on disk, keyed by source, name and hipRTC version
    const std::string path = disk_cache_path(key);
    if (!path.empty()) cached = read_cache_file(path, k.code);
    if (cached) {
      k.from_disk = true;
      ++g_cache_hits;
    } else {
      compile_kernel(k);
      ++g_cache_misses;
      if (!path.empty()) write_cache_file(path, k.code);
    }
    std::lock_guard<std::mutex> lock(g_code_cache_mutex);
    g_code_cache[key] = k.code;
  }
  read_metadata(k);
  pl.kernels.push_back(std::move(k));
  pl.kernel_by_source[keyed] = (int)pl.kernels.size() - 1;
  return (int)pl.kernels.size() - 1;
}

// ---------------------------------------------------------------- planner
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static size_t star_lds_bytes(const StarCfg& c, DT dt) {
  if (c.compact) {
    // kernels/compact3d.h: per window a ring of images (first / last row of every
    // thread row + the wave-edge columns of every row incl. two virtual waves)
    const size_t win = (size_t)c.BY * 2 * c.BX * c.VK + (size_t)(c.BY + 2) * c.RJ * (c.BX / 64 + 2) * 2;
    return std::max<size_t>(1, (size_t)c.lds_images * win) * size_of(dt);
  }
  const size_t rows = c.noj ? 0 : (size_t)c.T * c.BY * 2 * c.BX * c.VK;
  const size_t edge = (size_t)c.T * c.BY * c.RJ * (c.BX / 64 + ((c.dpp == 4 && c.BX > 64) ? 2 : 0)) * 2;
  return (rows + edge) * size_of(dt) * (c.lds_db ? 2 : 1);
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
  if (c.compact)  // three live planes per window, one more in flight per loaded window; fitted to
                  // the code objects of round 2 (box, T = 2: P = 16 -> 180, P = 20 -> 212)
    return (3 * c.nwin + 1 + c.nwin - c.T) * P * words + 52 + P;
  return 3 * c.T * P * words + 20 + (33 * P) / 10 +
         ((c.prefetch2 || c.reverse == 2) ? P * words * c.pfd : 0);
}

static int star_blocks_per_cu(const StarCfg& c, DT dt) {
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
  const int hk = round_up(T, c.VK);
  c.ktiled = (tkh != P.n[2]);
  c.HK = c.ktiled ? hk : 0;
  c.NKT = c.ktiled ? (int)((P.n[2] + (tkh - 2 * c.HK) - 1) / (tkh - 2 * c.HK)) : 1;
  if (c.noj) {
    c.NJT = 1;
  } else {
    const int tji = c.BY * c.RJ - 2 * T;
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
  const int max_nch = std::max(1, range / std::max(1, 2 * c.T));
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
      const double warm = (double)(li + 2 * c.T + (c.reverse ? c.T - 1 : 0)) / (double)li;
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
    const double warm = (double)(li + 2 * c.T + (c.reverse ? c.T - 1 : 0)) / (double)li;
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

// chunk length used for a launch over `range` planes (options k1.li / k2.li pin it)
static long long star_chunk_length(const sf_plan& pl, const StarCfg& c, DT dt, int range, int ranges = 1) {
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
  base.n0g = P.n[0];
  base.n1 = P.n[1];
  base.n2 = P.n[2];
  base.noj = (P.n[1] == 1);
  base.row_fence = (int)pl.opt.get("k1.fence", 1);
  base.lds_db = (int)pl.opt.get("k1.db", 1);
  base.opaque = (int)pl.opt.get("k1.opaque", base.noj ? 0 : 1);  // 2-D: registers are plentiful
  base.stamp = (int)pl.opt.get("stamp", 0);
  base.spread = (int)pl.opt.get("k1.spread", 1);
  // step order: 3-D kernels run stage 1 first (k1.rev=0); 2-D kernels run the
  // storing stage first, which makes the stages of one step independent of each
  // other (more instruction-level parallelism for the lone wave) -- measured
  // +30 % on C2, no change on C3 (profiles/r01_sweep_9_step_order.log)
  // -- with branch-free buffer loads and stores (k1.bio, below) the compiler counts
  // the memory operations in flight, a wave no longer drains them once per step,
  // and stage-1-first with the four-slot input ring overtakes it: C2 +10 %
  // (profiles/r01_sweep_17_buffer_io.log); k1.rev=1 remains available
  base.reverse = base.compact ? 1 : (int)pl.opt.get("k1.rev", 0);
  // input planes: 0 = loaded into the window slot stage 1 has just freed, 1 = into
  // staging registers a step earlier and copied, 2 = four-slot input ring (two
  // steps to land, no copy; the step loop is unrolled by 4)
  // (defaults: the ring for 2-D and f32 3-D -- C2 +10 %, C3 +1.7 %, hotspot chains
  // +1 % with k1.bio; staging registers for f64, whose ring needs a smaller tile)
  base.prefetch2 = base.reverse ? 0 : (int)pl.opt.get("k1.pf2", (base.noj || dt == DT::F32) ? 2 : 1);
  if (base.prefetch2 < 0 || base.prefetch2 > 2) throw Error(SF_ERR_INVALID, "k1.pf2 must be 0, 1 or 2");
  base.uniform_loads = (int)pl.opt.get("k1.ul", 0);
  // planes through buffer instructions (out-of-range offsets instead of branches
  // around loads and stores); a plane must stay well below the 2 GiB offset range
  const double plane_bytes = (double)P.n[1] * (double)P.n[2] * (double)size_of(dt);
  // Measured (profiles/r01_sweep_17_buffer_io.log): 2-D +10 % (with the step order
  // and input ring above), f64 3-D +1..4 %, f32 3-D jacobi +1.7 % with the ring,
  // f32 3-D chains with auxiliary fields (hotspot) +38 %
  base.buffer_io = plane_bytes <= 1024.0 * 1024 * 1024 ? (int)pl.opt.get("k1.bio", 1) : 0;
  base.pfd = (int)pl.opt.get("k1.pfd", 1);
  if (base.pfd != 1 && base.pfd != 3) throw Error(SF_ERR_INVALID, "k1.pfd must be 1 or 3");
  if ((base.prefetch2 != 1 && base.reverse != 2) || base.prefetch2 == 2) base.pfd = 1;
  base.experiment = (int)pl.opt.get("experiment", 0);
  // lane exchange: 0 = __shfl, 1 = DPP, 2 = DPP with bound_ctrl (no copy before the
  // move), 3 = as 2 and the wave's edge lane gets its value (boundary constant or
  // the neighbouring wave's edge column) from the move's starting destination
  // instead of a select, 4 = as 3 with the neighbour test removed: virtual waves
  // beside every row hold the boundary constant in LDS.  Measured: C3 2 -> 4 +1 %
  // (3 costs 7 % there: scalar branches per row), C5 2 -> 3/4 +4.7 %, C2 +8 %.
  const long long dpp_opt = pl.opt.get("k1.dpp", -1);
  base.dpp = dpp_opt >= 0 ? (int)dpp_opt : 4;
  base.uniform = (int)pl.opt.get("k1.uni", 0);
  // auxiliary (centre-only) fields: 1 = a stage requests all its rows before its
  // first row is evaluated (3-D hotspot chains +21 %); 2 = rows are requested a
  // whole step ahead into per-stage slots (2-D, where a thread has one row and
  // registers to spare, +11 %)
  base.aux_ahead = (int)pl.opt.get("k1.auxpre", base.noj ? 2 : 1);
  base.aux_pass = (int)pl.opt.get("k1.auxpass", 1);
  // Non-temporal output stores when a field is larger than the 256 MiB Infinity
  // Cache: nothing of it would survive until the next launch reads it, and not
  // allocating the written lines leaves the cache to the input stream (C3 +3 %,
  // C5 +1.4 %; the cache-resident 64 MiB field of C2 loses 13 % with them).
  const double field_bytes = (double)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local) * (double)P.n[1] *
                             (double)P.n[2] * (double)size_of(dt);
  base.nt = (int)pl.opt.get("k1.nt", field_bytes >= 256.0 * 1024 * 1024 ? 1 : 0);
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
        if (!c.noj && by * rj - 2 * T < 1) continue;
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
        const double edge_rows = c.noj ? (c.BX > 64 ? 1.2 : 1.0) : 1.0 + 0.8 / (double)c.RJ;
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

// What a code object's metadata says about its fitness.
//  * A code object in which register-allocator code sits ahead of an EXEC restore
//    (count_late_exec_restores) is WRONG -- the toolchain fault behind every wrong result of
//    "spilling" code objects seen in rounds 1 and 2 (tools/spill_probe.py: all failing shapes
//    spill SGPRs; tools/asm_objects.py: why).  Never accepted, unless the diagnostic
//    environment variable SF_HIP_UNSAFE_SGPR_SPILLS=1 is set (the probes).  SGPR spills as
//    such (lane moves into a VGPR) are correct.
//  * VGPR spills, scratch and AGPR copies are correct but slow: rejected by the
//    planner's search, accepted for a pinned shape with allow_spills=1 (experiments).
static bool kernel_unsafe(const CompiledKernel& k) {
  static const bool tolerate = std::getenv("SF_HIP_UNSAFE_SGPR_SPILLS") != nullptr;
  // (SF_HIP_STRICT_SGPR_SPILLS=1: round 2's first criterion, any SGPR spill, on top)
  static const bool strict = std::getenv("SF_HIP_STRICT_SGPR_SPILLS") != nullptr;
  return (k.late_exec_restores > 0 || ((strict || k.late_exec_restores < 0) && k.sgpr_spills > 0)) && !tolerate;
}
static bool kernel_slow(const CompiledKernel& k) {
  return std::max(0, k.spills) + std::max(0, k.scratch) + std::max(0, k.agprs) > 0;
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
                              const std::vector<int>& kernels, DT dt) {
  const Program& P = pl.P;
  StarCfg probe;
  probe.T = (int)kernels.size();
  const std::string sig = std::to_string(fnv1a(gen_star(P, kernels, probe).source));
  auto it = memo.find(sig);
  if (it != memo.end()) return it->second;
  const std::string prefix = std::string("sf_star") + (P.n[1] == 1 ? "2d_" : "3d_") + short_of(dt) + "_t" +
                             std::to_string(kernels.size());
  StarChoice out;
  std::vector<StarCfg> ranked;
  try {
    ranked = rank_star_cfgs(pl, (int)kernels.size(), dt);
  } catch (const Error&) {
    memo[sig] = out;
    return out;
  }
  const bool pinned = pl.opt.kv.count(P.n[1] == 1 ? "k2.bx" : "k1.bx") &&
                      (P.n[1] == 1 || (pl.opt.kv.count("k1.by") && pl.opt.kv.count("k1.rj")));
  const size_t tries = std::min<size_t>(ranked.size(), (size_t)std::max<long long>(1, pl.opt.get("k1.tries", 8)));
  int rejected = 0, sgpr_rejects = 0;
  for (size_t ci = 0; ci < tries && rejected < 2; ++ci) {  // compile errors rarely depend on the shape
    ranked[ci].lds_bytes = star_lds_bytes(ranked[ci], dt);
    StarKernelSource g = gen_star(P, kernels, ranked[ci]);
    int ck = -1;
    try {
      ck = intern_kernel(pl, prefix, g.source);
    } catch (const Error& e) {
      // a shape the compiler rejects is no candidate (a pinned shape reports it);
      // the group is shortened and in the end the generic kernel takes over
      if (pinned || e.status != SF_ERR_COMPILE) throw;
      if (pl.opt.get("debug", 0) != 0)
        std::fprintf(stderr, "[sf_hip] candidate %zu/%zu rejected by the compiler: %.200s\n", ci + 1,
                     ranked.size(), e.what());
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

// Extra compiler flags of the compact kernels: without the SLP vectoriser hipcc
// keeps the 26 adds of a box stencil scalar instead of pairing them into
// v_pk_add_f32, whose operand pairs it has to assemble with moves (option compact.slp).
static std::string compact_flags(const sf_plan& pl) {
  return pl.opt.get("compact.slp", 0) != 0 ? "" : "-fno-slp-vectorize";
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
  const size_t tries = std::min<size_t>(ranked.size(), (size_t)std::max<long long>(1, pl.opt.get("k1.tries", 8)));
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
      if (pl.opt.get("debug", 0) != 0)
        std::fprintf(stderr, "[sf_hip] compact candidate %zu/%zu rejected by the compiler: %.400s\n", ci + 1,
                     ranked.size(), e.what());
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
    desc << (st.compact ? "[compact windows " + std::to_string(st.cfg.nwin) + " T=" : std::string("[star T=")) << st.cfg.T << " block " << st.cfg.BX << "x" << st.cfg.BY << " rows/thread "
         << st.cfg.RJ << " tiles " << st.cfg.NJT << "x" << st.cfg.NKT << " chunk "
         << star_chunk_length(pl, st.cfg, dt, (int)pl.n_local) << " lds " << star_lds_bytes(st.cfg, dt)
         << " B]";
  else
    desc << "[point]";
  desc << " in";
  for (int b : st.in_bufs) desc << " b" << b;
  desc << " out b" << st.out_buf << " {vgpr " << ck.vgprs << " agpr " << ck.agprs << " spill " << ck.spills
       << " scratch " << ck.scratch << "}\n";
}

// The whole description: header line, one line per launch (long chains: the first ones).
static std::string describe_plan(const sf_plan& pl) {
  const Program& P = pl.P;
  std::ostringstream desc;
  desc << "program " << P.name << ": dims " << P.n[0] << "x" << P.n[1] << "x" << P.n[2] << ", "
       << P.kernels.size() << " operators, " << pl.steps.size() << " launches, " << pl.buffers.size()
       << " device buffers\n";
  for (auto& st : pl.steps) {
    if (desc.tellp() > 16384) break;  // long chains: describe the first launches only
    describe_step(desc, pl, st);
  }
  return desc.str();
}

// Values outside an option's range are the caller's mistake and are reported;
// (a shape no kernel can serve is not: that group falls back to the generic kernel)
static void validate_options(const sf_plan& pl) {
  struct Range {
    const char* key;
    long long lo, hi;
  };
  static const Range ranges[] = {{"k1.pf2", 0, 2}, {"k1.rev", 0, 2},  {"k1.dpp", 0, 4},   {"k1.bio", 0, 3},
                                 {"k1.ul", 0, 1},  {"k1.db", 0, 1},   {"k1.nt", 0, 3},    {"k1.auxpre", 0, 2},
                                 {"graph", 0, 1},  {"autotune", 0, 8}};
  for (const Range& r : ranges) {
    if (!pl.opt.kv.count(r.key)) continue;
    const long long v = pl.opt.get(r.key, r.lo);
    if (v < r.lo || v > r.hi)
      throw Error(SF_ERR_INVALID, std::string("option ") + r.key + " must lie in [" + std::to_string(r.lo) + ", " +
                                      std::to_string(r.hi) + "]");
  }
  if (pl.opt.kv.count("k1.vk") && pl.opt.get("k1.vk", 4) != 1 && pl.opt.get("k1.vk", 4) != 2 &&
      pl.opt.get("k1.vk", 4) != 4)
    throw Error(SF_ERR_INVALID, "k1.vk must be 1, 2 or 4");
  if (pl.opt.kv.count("k1.pfd") && pl.opt.get("k1.pfd", 1) != 1 && pl.opt.get("k1.pfd", 1) != 3)
    throw Error(SF_ERR_INVALID, "k1.pfd must be 1 or 3");
}

static void build_plan(sf_plan& pl) {
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
  const bool generic_only = pl.opt.get("generic_only", 0) != 0;
  // (any row length: the vector width follows it, rank_star_cfgs)
  const bool star_ok_dims = (P.nd >= 2) && P.n[0] > 1;

  // ---- group kernels into launches
  std::map<std::string, StarChoice> star_memo;
  for (int k = 0; k < K;) {
    Step st;
    StarShape shape;
    // (star=0: diagnostics -- star chains then run on the compact kernel)
    bool star = !generic_only && star_ok_dims && star_eligible(P, P.kernels[k], &shape) && pl.opt.get("star", 1) != 0;
    if (star && P.n[1] == 1) {
      for (auto& a : P.kernels[k].acc)
        if (a.off[1] != 0) star = false;
    }
    if (star) {
      std::vector<int> group{k};
      std::set<std::string> group_aux(shape.aux.begin(), shape.aux.end());
      while ((int)group.size() < fuse && k + (int)group.size() < K) {
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
      if (choice.ok) {
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
      if (compact) {
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
        while ((int)group.size() < fuse && k + (int)group.size() < K) {
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
    if (st.compact) {
      // argument 0: the streamed field of the first stage; then the extra fields in
      // first-use order (as gen_compact numbers them)
      for (size_t si = 0; si < st.kernels.size(); ++si) {
        CompactShape sh;
        compact_eligible(P, P.kernels[st.kernels[si]], &sh, si == 0 ? std::string() : P.kernels[st.kernels[si - 1]].name);
        if (si == 0) r.push_back(sh.primary);
        if (!sh.extra.empty() && std::find(r.begin() + 1, r.end(), sh.extra) == r.end()) r.push_back(sh.extra);
      }
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
    }
    const Field& of = P.field(P.kernels[st.kernels.back()].name);
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
    st.out_buf = ob;
    buf_of[of.name] = ob;
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
      StarKernelSource g = st.compact ? gen_compact(P, st.kernels, st.cfg) : gen_star(P, st.kernels, st.cfg);
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
      st.halo_depth = st.cfg.T;
      st.halo_buf = st.in_bufs[0];
    } else {
      // 4 points per thread with aligned vector loads when rows allow it; the
      // one-point form is kept for short rows and for operators whose vector
      // form would spill (same acceptance rule as for the star kernels)
      const bool vec = (P.n[2] % 4 == 0) && pl.opt.get("generic.vec", 1) != 0;
      const bool xcd = pl.opt.get("generic.xcd", 1) != 0;
      // non-temporal output stores for fields beyond the Infinity Cache (see rank_star_cfgs)
      const double out_bytes = (double)(pl.plan_extent > 0 ? pl.plan_extent : pl.n_local) * (double)P.n[1] *
                               (double)P.n[2] * (double)size_of(dt);  // (alike on all ranks of a slab run)
      const bool nts = pl.opt.get("generic.nt", out_bytes >= 256.0 * 1024 * 1024 ? 1 : 0) != 0;
      // marching form (a thread walks `generic.ppt` planes with a register window,
      // default 8) for 3-D programs; generic.march=0 restores the one-plane form
      const bool march = vec && P.n[0] > 1 && pl.opt.get("generic.march", 1) != 0 && pl.opt.get("generic.bio", 0) == 0 &&
                         pl.opt.get("generic.fast", 0) == 0;  // (those two are variants of the one-plane form)
      const int ppt = (int)std::max<long long>(1, std::min<long long>(march ? 256 : 8, pl.opt.get("generic.ppt", march ? 8 : 1)));
      auto make = [&](bool marching) {
        return marching ? gen_generic_march(P, st.kernels[0], xcd, nts, ppt)
               : vec    ? gen_generic_vec(P, st.kernels[0], xcd, nts, march ? 1 : ppt, pl.opt.get("generic.fast", 0) != 0,
                                          pl.opt.get("generic.bio", 0) != 0)
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

// ---------------------------------------------------------------- runtime
static void ensure_device(sf_plan& pl) {
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
      const std::string key = k.name + "\n" + (k.flags.empty() ? k.source : k.flags + "\n" + k.source);
      const std::string path = disk_cache_path(key);
      if (!path.empty()) std::remove(path.c_str());
      compile_kernel(k);
      ++g_cache_recompiles;
      k.from_disk = false;
      read_metadata(k);
      {
        std::lock_guard<std::mutex> lock(g_code_cache_mutex);
        g_code_cache[key] = k.code;
      }
      if (!path.empty()) write_cache_file(path, k.code);
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

static void launch_ranges(sf_plan& pl, const Step& st, int i_begin, int i_end, int i_begin2, int i_end2,
                          hipStream_t stream);

static void launch_step(sf_plan& pl, const Step& st, int part, hipStream_t stream) {
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
static void launch_ranges(sf_plan& pl, const Step& st, int i_begin, int i_end, int i_begin2, int i_end2,
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
  }
}

static void collect_profile(sf_plan& pl) {
  for (size_t i = 0; i < pl.prof_events.size(); ++i) {
    float ms = 0;
    (void)hipEventElapsedTime(&ms, pl.prof_events[i].first, pl.prof_events[i].second);
    pl.kernels[pl.prof_kernel[i]].launches += 1;
    pl.kernels[pl.prof_kernel[i]].total_ms += ms;
    (void)hipEventDestroy(pl.prof_events[i].first);
    (void)hipEventDestroy(pl.prof_events[i].second);
  }
  pl.prof_events.clear();
  pl.prof_kernel.clear();
}

static void upload(sf_plan& pl, const void* const* host_inputs) {
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

static void download(sf_plan& pl, void* const* host_outputs) {
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
static void autotune(sf_plan& pl) {
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

static void execute(sf_plan& pl, int repetitions) {
  ensure_device(pl);
  autotune(pl);
  if (repetitions < 0) throw Error(SF_ERR_INVALID, "negative repetition count");
  for (auto& k : pl.kernels) {
    k.launches = 0;
    k.total_ms = 0;
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

// ---------------------------------------------------------------- C ABI
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

extern "C" {

int sf_version(void) { return 1001; }  // 1.1: sf_halo_*, sf_plan_execute_decomposed, sf_plan_stream, sf_plan_num_buffers, sf_code_cache_stats

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
    if (plan->debug_buffer) (void)hipFree(plan->debug_buffer);
    for (auto& k : plan->kernels)
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
  if (disk_hits) *disk_hits = sf::g_cache_hits.load();
  if (compiled) *compiled = sf::g_cache_misses.load();
  if (rebuilt) *rebuilt = sf::g_cache_recompiles.load();
  if (drop_process_level) {
    std::lock_guard<std::mutex> lock(sf::g_code_cache_mutex);
    sf::g_code_cache.clear();
  }
  return SF_OK;
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
int sf_plan_debug_counters(sf_plan* plan, unsigned long long* out, int count) {
  SF_API_BEGIN
  if (!plan || !out || count < 0 || count > 8) throw Error(SF_ERR_INVALID, "bad argument");
  ensure_device(*plan);
  SF_HIP_CHECK(hipStreamSynchronize(plan->stream));
  SF_HIP_CHECK(hipMemcpy(out, plan->debug_buffer, sizeof(unsigned long long) * count, hipMemcpyDeviceToHost));
  SF_HIP_CHECK(hipMemset(plan->debug_buffer, 0, 64));
  return SF_OK;
  SF_API_END
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

// ---------------------------------------------------------------- flag kernels
// One lane each.  The flag lives in pinned host memory that several processes
// have mapped: system-scope atomics, so that neither side's caches hold it.
static __global__ void sf_flag_set_kernel(unsigned int* flag, unsigned int value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static __global__ void sf_flag_wait_kernel(const unsigned int* flag, unsigned int value,
                                           unsigned long long timeout_ticks, unsigned int* status) {
  const unsigned long long t0 = wall_clock64();  // constant-rate counter (100 MHz)
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {  // never spin forever
      if (status) __hip_atomic_store(status, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

extern "C" {

int sf_host_register(void* ptr, size_t bytes, void** device_ptr) {
  SF_API_BEGIN
  if (!ptr || bytes == 0) throw sf::Error(SF_ERR_INVALID, "sf_host_register: null range");
  SF_HIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  if (device_ptr) {
    void* dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess) {
      (void)hipHostUnregister(ptr);
      throw sf::Error(SF_ERR_DEVICE, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    *device_ptr = dev;
  }
  return SF_OK;
  SF_API_END
}

int sf_host_unregister(void* ptr) {
  SF_API_BEGIN
  if (!ptr) return SF_OK;
  SF_HIP_CHECK(hipHostUnregister(ptr));
  return SF_OK;
  SF_API_END
}

int sf_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  SF_API_BEGIN
  if ((!dst || !src) && bytes) throw sf::Error(SF_ERR_INVALID, "sf_copy_async: null pointer");
  if (bytes) SF_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream));
  return SF_OK;
  SF_API_END
}

int sf_flag_set(void* stream, unsigned int* flag, unsigned int value) {
  SF_API_BEGIN
  if (!flag) throw sf::Error(SF_ERR_INVALID, "sf_flag_set: null flag");
  hipLaunchKernelGGL(sf_flag_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
  SF_API_END
}

int sf_flag_wait(void* stream, const unsigned int* flag, unsigned int value, unsigned int timeout_ms,
                 unsigned int* status) {
  SF_API_BEGIN
  if (!flag) throw sf::Error(SF_ERR_INVALID, "sf_flag_wait: null flag");
  const unsigned long long ticks = (unsigned long long)std::max(1u, timeout_ms) * 100000ull;
  hipLaunchKernelGGL(sf_flag_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value, ticks, status);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
  SF_API_END
}

}  // extern "C"

// ---------------------------------------------------------------- sf_halo
// Peer-to-peer halo transport of the slab decomposition, owned by the library: a
// rank PUSHES the planes next to a slab boundary straight into its neighbour's
// ghost planes -- device memory of the neighbour's plan, mapped here through a HIP
// IPC handle -- with DMA copies (no compute units, over xGMI between the GPUs of a
// node), ordered by the flag words of a small page of host memory that the ranks of
// the node share (POSIX shared memory, pinned; sf_flag_set / sf_flag_wait above).
// Everything is enqueued on two streams of the transport; the caller's compute
// stream only waits for their events.
//
// Flag page of rank r (unsigned words), per registered buffer `key` (< 60):
//   [16 key + d]      ready[d]  : r may receive exchange n from neighbour d (0 lower, 1 upper)
//   [16 key + 2 + d]  arrived[d]: neighbour d has delivered exchange n (written by that neighbour)
//   [1023]            status    : a wait of r timed out
#include <fcntl.h>
#include <sys/mman.h>

namespace sf {

struct HaloBlob {  // what a rank tells its neighbours about one buffer (plain bytes)
  char magic[8];
  hipIpcMemHandle_t mem;
  unsigned long long plane_bytes;
  int n_local, halo, rank, device;
  char flags_name[96];
};
static_assert(sizeof(HaloBlob) <= SF_HALO_BLOB_BYTES, "SF_HALO_BLOB_BYTES too small");

struct FlagPage {
  std::string name;
  void* host = nullptr;
  unsigned* dev = nullptr;  // address kernels dereference
  bool owner = false;
};

struct HaloBuffer {
  char* base = nullptr;
  size_t plane_bytes = 0;
  int n_local = 0, halo = 0;
  unsigned count = 0;
  // neighbours (0 lower, 1 upper): their buffer mapped here and its geometry
  char* peer[2] = {nullptr, nullptr};
  int peer_n_local[2] = {0, 0}, peer_halo[2] = {0, 0};
  hipEvent_t sent = nullptr, received = nullptr;
  bool pending = false;
};

}  // namespace sf

struct sf_halo {
  int rank = 0, world = 1, device = 0;
  unsigned timeout_ms = 20000;
  std::string session;
  sf::FlagPage own, nb[2];
  std::map<int, sf::HaloBuffer> bufs;
  hipStream_t send = nullptr, recv = nullptr;
  hipEvent_t now = nullptr;
};

namespace sf {

static const size_t kFlagPageBytes = 4096;

static void map_flag_page(FlagPage& page, const std::string& name, bool create) {
  const int fd = ::shm_open(name.c_str(), O_RDWR | (create ? (O_CREAT | O_EXCL) : 0), 0600);
  if (fd < 0) throw Error(SF_ERR_DEVICE, "sf_halo: shm_open(" + name + ") failed: " + std::strerror(errno));
  if (create && ::ftruncate(fd, (off_t)kFlagPageBytes) != 0) {
    ::close(fd);
    ::shm_unlink(name.c_str());
    throw Error(SF_ERR_DEVICE, "sf_halo: ftruncate failed");
  }
  void* p = ::mmap(nullptr, kFlagPageBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  ::close(fd);
  if (p == MAP_FAILED) throw Error(SF_ERR_DEVICE, "sf_halo: mmap of the flag page failed");
  if (create) std::memset(p, 0, kFlagPageBytes);
  hipError_t e = hipHostRegister(p, kFlagPageBytes, hipHostRegisterPortable | hipHostRegisterMapped);
  void* dev = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, p, 0);
  if (e != hipSuccess) {
    ::munmap(p, kFlagPageBytes);
    if (create) ::shm_unlink(name.c_str());
    throw Error(SF_ERR_DEVICE, std::string("sf_halo: pinning the flag page: ") + hipGetErrorString(e));
  }
  page.name = name;
  page.host = p;
  page.dev = static_cast<unsigned*>(dev);
  page.owner = create;
}

static void unmap_flag_page(FlagPage& page) {
  if (!page.host) return;
  (void)hipHostUnregister(page.host);
  ::munmap(page.host, kFlagPageBytes);
  if (page.owner) ::shm_unlink(page.name.c_str());
  page = FlagPage();
}

static void halo_flag_set(hipStream_t s, unsigned* flag, unsigned value) {
  hipLaunchKernelGGL(sf_flag_set_kernel, dim3(1), dim3(1), 0, s, flag, value);
  SF_HIP_CHECK(hipGetLastError());
}
static void halo_flag_wait(sf_halo& h, hipStream_t s, const unsigned* flag, unsigned value) {
  const unsigned long long ticks = (unsigned long long)std::max(1u, h.timeout_ms) * 100000ull;
  hipLaunchKernelGGL(sf_flag_wait_kernel, dim3(1), dim3(1), 0, s, flag, value, ticks, h.own.dev + 1023);
  SF_HIP_CHECK(hipGetLastError());
}

}  // namespace sf

extern "C" {

int sf_halo_create(int rank, int world, const char* session, int device, unsigned int timeout_ms, sf_halo** out) {
  SF_API_BEGIN
  if (!out || !session || rank < 0 || world < 1 || rank >= world) throw Error(SF_ERR_INVALID, "sf_halo_create: bad argument");
  std::unique_ptr<sf_halo> h(new sf_halo);
  h->rank = rank;
  h->world = world;
  h->device = device;
  h->timeout_ms = timeout_ms ? timeout_ms : 20000;
  h->session = session;
  SF_HIP_CHECK(hipSetDevice(device));
  SF_HIP_CHECK(hipStreamCreateWithFlags(&h->send, hipStreamNonBlocking));
  SF_HIP_CHECK(hipStreamCreateWithFlags(&h->recv, hipStreamNonBlocking));
  SF_HIP_CHECK(hipEventCreateWithFlags(&h->now, hipEventDisableTiming));
  sf::map_flag_page(h->own, "/sf_halo_" + h->session + "_" + std::to_string(rank), true);
  *out = h.release();
  return SF_OK;
  SF_API_END
}

int sf_halo_destroy(sf_halo* h) {
  SF_API_BEGIN
  if (!h) return SF_OK;
  (void)hipSetDevice(h->device);
  if (h->send) (void)hipStreamSynchronize(h->send);
  if (h->recv) (void)hipStreamSynchronize(h->recv);
  for (auto& kv : h->bufs) {
    for (int d = 0; d < 2; ++d)
      if (kv.second.peer[d]) (void)hipIpcCloseMemHandle(kv.second.peer[d]);
    if (kv.second.sent) (void)hipEventDestroy(kv.second.sent);
    if (kv.second.received) (void)hipEventDestroy(kv.second.received);
  }
  sf::unmap_flag_page(h->nb[0]);
  sf::unmap_flag_page(h->nb[1]);
  sf::unmap_flag_page(h->own);
  if (h->now) (void)hipEventDestroy(h->now);
  if (h->send) (void)hipStreamDestroy(h->send);
  if (h->recv) (void)hipStreamDestroy(h->recv);
  delete h;
  return SF_OK;
  SF_API_END
}

int sf_halo_export(sf_halo* h, int key, void* device_base, size_t plane_bytes, int n_local, int halo, void* blob) {
  SF_API_BEGIN
  if (!h || !device_base || !blob || key < 0 || key >= 60 || plane_bytes == 0 || n_local < 1 || halo < 1)
    throw Error(SF_ERR_INVALID, "sf_halo_export: bad argument");
  if (h->bufs.count(key)) throw Error(SF_ERR_STATE, "sf_halo_export: buffer key already registered");
  SF_HIP_CHECK(hipSetDevice(h->device));
  sf::HaloBuffer b;
  b.base = static_cast<char*>(device_base);
  b.plane_bytes = plane_bytes;
  b.n_local = n_local;
  b.halo = halo;
  SF_HIP_CHECK(hipEventCreateWithFlags(&b.sent, hipEventDisableTiming));
  SF_HIP_CHECK(hipEventCreateWithFlags(&b.received, hipEventDisableTiming));
  sf::HaloBlob out;
  std::memset(&out, 0, sizeof out);
  std::memcpy(out.magic, "SFHALO1", 8);
  if (h->world > 1) SF_HIP_CHECK(hipIpcGetMemHandle(&out.mem, device_base));
  out.plane_bytes = plane_bytes;
  out.n_local = n_local;
  out.halo = halo;
  out.rank = h->rank;
  out.device = h->device;
  std::snprintf(out.flags_name, sizeof out.flags_name, "%s", h->own.name.c_str());
  std::memset(blob, 0, SF_HALO_BLOB_BYTES);
  std::memcpy(blob, &out, sizeof out);
  h->bufs[key] = b;
  return SF_OK;
  SF_API_END
}

int sf_halo_connect(sf_halo* h, int key, const void* lower_blob, const void* upper_blob) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_connect: unknown buffer key");
  SF_HIP_CHECK(hipSetDevice(h->device));
  sf::HaloBuffer& b = h->bufs[key];
  const void* blobs[2] = {h->rank > 0 ? lower_blob : nullptr, h->rank < h->world - 1 ? upper_blob : nullptr};
  for (int d = 0; d < 2; ++d) {
    const bool expected = d == 0 ? h->rank > 0 : h->rank < h->world - 1;
    if (!expected) continue;
    if (!blobs[d]) throw Error(SF_ERR_INVALID, "sf_halo_connect: missing neighbour description");
    sf::HaloBlob in;
    std::memcpy(&in, blobs[d], sizeof in);
    if (std::memcmp(in.magic, "SFHALO1", 8) != 0 || in.rank != h->rank + (d == 0 ? -1 : 1))
      throw Error(SF_ERR_INVALID, "sf_halo_connect: not the description of the neighbouring rank");
    if (in.plane_bytes != b.plane_bytes || in.halo != b.halo)
      throw Error(SF_ERR_INVALID, "sf_halo_connect: the neighbour's buffer has another plane size or halo");
    void* mapped = nullptr;
    SF_HIP_CHECK(hipIpcOpenMemHandle(&mapped, in.mem, hipIpcMemLazyEnablePeerAccess));
    b.peer[d] = static_cast<char*>(mapped);
    b.peer_n_local[d] = in.n_local;
    b.peer_halo[d] = in.halo;
    if (!h->nb[d].host) sf::map_flag_page(h->nb[d], in.flags_name, false);
  }
  return SF_OK;
  SF_API_END
}

int sf_halo_start(sf_halo* h, int key, int depth, void* compute_stream) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_start: unknown buffer key");
  sf::HaloBuffer& b = h->bufs[key];
  if (depth < 1 || depth > b.halo || depth > b.n_local) throw Error(SF_ERR_INVALID, "sf_halo_start: bad depth");
  if (b.pending) throw Error(SF_ERR_STATE, "sf_halo_start: the previous exchange of this buffer was not finished");
  if (h->world == 1) return SF_OK;
  SF_HIP_CHECK(hipSetDevice(h->device));
  const unsigned n = ++b.count;
  const size_t bytes = (size_t)depth * b.plane_bytes;
  // the planes to send are final and the ghost planes no longer read once
  // everything queued on the compute stream so far has completed
  SF_HIP_CHECK(hipEventRecord(h->now, (hipStream_t)compute_stream));
  SF_HIP_CHECK(hipStreamWaitEvent(h->send, h->now, 0));
  SF_HIP_CHECK(hipStreamWaitEvent(h->recv, h->now, 0));
  for (int d = 0; d < 2; ++d)
    if (b.peer[d]) sf::halo_flag_set(h->recv, h->own.dev + 16 * key + d, n);
  for (int d = 0; d < 2; ++d) {
    if (!b.peer[d]) continue;
    const int their = 1 - d;  // which of the neighbour's sides we are on
    sf::halo_flag_wait(*h, h->send, h->nb[d].dev + 16 * key + their, n);
    const char* src = b.base + (size_t)(d == 0 ? b.halo : b.halo + b.n_local - depth) * b.plane_bytes;
    char* dst = b.peer[d] + (size_t)(d == 0 ? b.peer_halo[d] + b.peer_n_local[d] : b.peer_halo[d] - depth) * b.plane_bytes;
    SF_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->send));
    sf::halo_flag_set(h->send, h->nb[d].dev + 16 * key + 2 + their, n);
  }
  for (int d = 0; d < 2; ++d)
    if (b.peer[d]) sf::halo_flag_wait(*h, h->recv, h->own.dev + 16 * key + 2 + d, n);
  SF_HIP_CHECK(hipEventRecord(b.sent, h->send));
  SF_HIP_CHECK(hipEventRecord(b.received, h->recv));
  b.pending = true;
  return SF_OK;
  SF_API_END
}

int sf_halo_finish(sf_halo* h, int key, void* compute_stream) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_finish: unknown buffer key");
  sf::HaloBuffer& b = h->bufs[key];
  if (!b.pending) return SF_OK;
  SF_HIP_CHECK(hipSetDevice(h->device));
  SF_HIP_CHECK(hipStreamWaitEvent((hipStream_t)compute_stream, b.sent, 0));
  SF_HIP_CHECK(hipStreamWaitEvent((hipStream_t)compute_stream, b.received, 0));
  b.pending = false;
  return SF_OK;
  SF_API_END
}

int sf_halo_check(sf_halo* h) {
  SF_API_BEGIN
  if (!h) throw Error(SF_ERR_INVALID, "sf_halo_check: null transport");
  const unsigned status = __atomic_load_n(static_cast<unsigned*>(h->own.host) + 1023, __ATOMIC_ACQUIRE);
  if (status != 0)
    throw Error(SF_ERR_DEVICE, "sf_halo: a neighbour of rank " + std::to_string(h->rank) + " did not answer within " +
                                   std::to_string(h->timeout_ms) + " ms");
  return SF_OK;
  SF_API_END
}

}  // extern "C"

// ---------------------------------------------------------------- native slab schedule
extern "C" int sf_plan_execute_decomposed(sf_plan* plan, sf_halo* halo, int repetitions) {
  SF_API_BEGIN
  if (!plan || !halo || repetitions < 0) throw Error(SF_ERR_INVALID, "sf_plan_execute_decomposed: bad argument");
  sf_plan& pl = *plan;
  ensure_device(pl);
  autotune(pl);
  const Program& P = pl.P;
  const int n = (int)pl.n_local, H = pl.halo;
  const bool has_lower = pl.goff > 0, has_upper = pl.goff + pl.n_local < P.n[0];
  const bool alone = !has_lower && !has_upper;
  if (!alone && H < 1) throw Error(SF_ERR_STATE, "sf_plan_execute_decomposed: the plan has no halo (option slab=lo:hi:halo)");
  // a chain: every launch reads exactly the slab buffer the previous one wrote
  bool chain = true;
  for (size_t s = 0; s < pl.steps.size(); ++s) {
    const Step& st = pl.steps[s];
    if (st.in_bufs.size() != 1 || (s > 0 && st.in_bufs[0] != pl.steps[s - 1].out_buf)) chain = false;
  }
  auto exchange_all = [&](const std::vector<int>& bufs, int depth) {
    for (int b : bufs) {
      const int rc = sf_halo_start(halo, b, depth, (void*)pl.stream);
      if (rc != SF_OK) throw Error(rc, sf_last_error());
    }
  };
  auto finish_all = [&](const std::vector<int>& bufs) {
    for (int b : bufs) {
      const int rc = sf_halo_finish(halo, b, (void*)pl.stream);
      if (rc != SF_OK) throw Error(rc, sf_last_error());
    }
  };
  // program inputs no launch writes (extra fields, auxiliary fields): their ghost planes are
  // filled once per call, to the full halo depth, at the first launch that reads them
  std::set<int> fixed, fresh;
  for (int i = 0; i < P.num_inputs; ++i) fixed.insert(pl.input_buf[i]);
  for (const Step& st : pl.steps) fixed.erase(st.out_buf);
  for (int rep = 0; rep < repetitions; ++rep) {
    int valid = 0;  // ghost planes of the chain's current field that are still good
    for (size_t s = 0; s < pl.steps.size(); ++s) {
      const Step& st = pl.steps[s];
      const int d = st.halo_buf >= 0 ? st.halo_depth : 0;
      if (alone || d == 0) {
        launch_ranges(pl, st, 0, n, 0, 0, pl.stream);
        continue;
      }
      if (2 * std::max(d, chain ? H : d) > n) throw Error(SF_ERR_STATE, "slab too thin for its halo");
      if (chain && d <= valid) {
        const int ext = valid - d;
        launch_ranges(pl, st, has_lower ? -ext : 0, n + (has_upper ? ext : 0), 0, 0, pl.stream);
        valid = ext;
        continue;
      }
      std::vector<int> bufs;
      if (chain) {
        bufs.push_back(st.in_bufs[0]);
      } else {
        for (int b : st.in_bufs)
          if (pl.buffers[b].slabbed && pl.buffers[b].planes > 1 && std::find(bufs.begin(), bufs.end(), b) == bufs.end())
            bufs.push_back(b);
      }
      const int depth = chain ? H : d;
      if (!chain) {
        std::vector<int> now, once;
        for (int b : bufs) {
          if (fresh.count(b)) continue;
          if (fixed.count(b)) {
            once.push_back(b);
            fresh.insert(b);
          } else {
            now.push_back(b);
          }
        }
        exchange_all(once, H);
        bufs = now;
        bufs.insert(bufs.end(), once.begin(), once.end());  // (finish_all below waits for both)
        exchange_all(now, depth);
      } else
      exchange_all(bufs, depth);
      launch_ranges(pl, st, has_lower ? d : 0, n - (has_upper ? d : 0), 0, 0, pl.stream);  // beside the transfer
      finish_all(bufs);
      const int ext = depth - d;
      launch_ranges(pl, st, has_lower ? -ext : 0, has_lower ? d : 0, has_upper ? n - d : 0, has_upper ? n + ext : 0,
                    pl.stream);
      valid = chain ? ext : 0;
    }
  }
  return SF_OK;
  SF_API_END
}
