// codecache.cpp — hipRTC driver, code-object metadata, the EXEC-restore detector and the
// two cache levels (process, disk) of libsf_hip.so.
#include "sf_internal.hpp"

#include <amd_comgr/amd_comgr.h>
#include <climits>
#include <cstdlib>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <link.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>

namespace sf {

thread_local std::string g_last_error;

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

int count_late_exec_restores(const std::vector<char>& code) {
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
      if (starts("s_or_b64 exec, exec, ") || starts("s_xor_b64 exec, exec, ") || starts("s_andn2_b64 exec, exec, ") ||
          starts("s_or_saveexec_b64 ") || starts("s_andn2_saveexec_b64 "))
        c = 'R';  // end of an `if`, `else` entry (two forms), loop exit: the EXEC updates that open a block
                  // (the saved mask in an SGPR pair or in vcc)
      else if ((starts("s_mov_b32 s") || starts("s_mov_b64 s[") || starts("s_mov_b32 vcc") || starts("s_mov_b64 vcc")) &&
               t.find("exec") == std::string::npos)
        c = 'S';  // a split copy, or a constant: the allocator rematerialises values the same way
      else if (starts("v_readlane_b32 ") || starts("v_writelane_b32 "))
        c = 'S';
      else if (starts("v_mov_b32") || starts("v_mov_b64") || starts("v_pk_mov_b32") || starts("v_accvgpr_") ||
               starts("scratch_load_") || starts("scratch_store_"))
        c = 'V';  // every encoding of a vector copy (e32 / e64 / dpp / sdwa, packed) and the spill code
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
  // (metadata that cannot be read says nothing about the register count: such an object is
  // disassembled like a large one)
  k.late_exec_restores = (k.sgprs >= 64 || k.sgprs < 0) ? count_late_exec_restores(k.code) : 0;
}

// Code objects are cached per process: plans of the same program (slab ranks,
// repeated runs, the tile search of another chain) do not recompile.
static std::mutex g_code_cache_mutex;
static std::map<std::string, std::vector<char>> g_code_cache;  // name + source -> code object

// <dir>/<hash>.co, or "" when the disk cache is off.  Directory: $SF_HIP_CACHE_DIR
// ("off" disables), default $XDG_CACHE_HOME or ~/.cache + /stencilflow_amd.
// The device compiler of this process: hipRTC major.minor, the HIP runtime's full version number (patch level
// included), the build id of the ROCm headers this library was compiled against -- and the libamd_comgr the
// process really compiles through: hipRTC binds libamd_comgr by soname, so whichever copy was loaded first does
// the work (PyTorch's bundled copy after `import torch`, ROCm's in a process without torch or when a profiler
// preloads it): same versions reported, different compilers (round 3: C5's five-row tile spills under one of
// them).  $SF_HIP_COMGR pins it: stencilflow_amd.backend loads that file first, and a process whose comgr is
// another file is refused at plan creation (check_pinned_compiler) instead of compiling other code silently.
// Every libamd_comgr resident in the process (`dladdr` of the symbol THIS library resolved names the copy this
// library bound, which need not be the one libhiprtc bound -- a wheel with renamed sonames or RPATH-local copies can hold
// two).  More than one: the id says so, and a pinned compiler is refused, because which copy compiles can then not be
// told from outside.
static std::vector<std::string> resident_comgrs() {
  std::vector<std::string> found;
  ::dl_iterate_phdr(
      [](struct dl_phdr_info* info, size_t, void* data) {
        auto* out = static_cast<std::vector<std::string>*>(data);
        if (info->dlpi_name && std::strstr(info->dlpi_name, "libamd_comgr")) {
          char real[PATH_MAX];
          const char* r = ::realpath(info->dlpi_name, real);
          const std::string path = r ? r : info->dlpi_name;
          if (std::find(out->begin(), out->end(), path) == out->end()) out->push_back(path);
        }
        return 0;
      },
      &found);
  return found;
}

std::string compiler_id() {
  int major = 0, minor = 0, runtime = 0;
  hiprtcVersion(&major, &minor);
  (void)hipRuntimeGetVersion(&runtime);
  size_t cmaj = 0, cmin = 0;
  amd_comgr_get_version(&cmaj, &cmin);
  Dl_info comgr_lib;
  const char* comgr_path = (::dladdr(reinterpret_cast<void*>(&amd_comgr_get_version), &comgr_lib) && comgr_lib.dli_fname)
                               ? comgr_lib.dli_fname
                               : "?";
  std::string also;
  const std::vector<std::string> all = resident_comgrs();
  if (all.size() > 1) {
    also = " [" + std::to_string(all.size()) + " copies of libamd_comgr resident:";
    for (auto& p : all) also += " " + p;
    also += "]";
  }
  return "hiprtc " + std::to_string(major) + "." + std::to_string(minor) + " runtime " + std::to_string(runtime) +
         " build " + HIP_VERSION_GITHASH + " comgr " + comgr_path + " (" + std::to_string(cmaj) + "." +
         std::to_string(cmin) + ")" + also;
}

void check_pinned_compiler() {
  const char* want = std::getenv("SF_HIP_COMGR");
  if (!want || !*want) return;
  Dl_info comgr_lib;
  const char* have = (::dladdr(reinterpret_cast<void*>(&amd_comgr_get_version), &comgr_lib) && comgr_lib.dli_fname)
                         ? comgr_lib.dli_fname
                         : "";
  char a[PATH_MAX], b[PATH_MAX];
  const char* ra = ::realpath(want, a);
  const char* rb = ::realpath(have, b);
  if (!ra) throw Error(SF_ERR_INVALID, std::string("SF_HIP_COMGR names '") + want + "', which does not exist");
  const std::vector<std::string> all = resident_comgrs();
  if (all.size() > 1) {
    std::string list;
    for (auto& p : all) list += " " + p;
    throw Error(SF_ERR_STATE, std::string("SF_HIP_COMGR pins the device compiler to ") + ra + " but " + std::to_string(all.size()) +
                                  " copies of libamd_comgr are resident (" + list.substr(1) + "): which one hipRTC compiles through cannot be told");
  }
  if (!rb || std::strcmp(ra, rb) != 0)
    throw Error(SF_ERR_STATE, std::string("SF_HIP_COMGR pins the device compiler to ") + ra + " but this process compiles through " +
                                  (rb ? rb : have) + ": that library must be loaded before libhiprtc (stencilflow_amd.backend does so; "
                                  "a C program: LD_PRELOAD it or link it first)");
}

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
  // the compiler that would produce this object (compiler_id(): hipRTC, runtime, build of the ROCm headers, and
  // the libamd_comgr the process really compiles through)
  const std::string salted = key + "\n" + compiler_id() + "\ngfx950 -O3 -std=c++17 -ffp-contract=off";
  char name[40];
  std::snprintf(name, sizeof name, "%016llx%08x", (unsigned long long)fnv1a(salted), (unsigned)salted.size());
  return dir + "/" + name + ".co";
}

// Cache file = 32-byte header {magic "SFCO0003", payload bytes, FNV-1a of the payload,
// verdict of the plan-time self-check (0 not checked, 1 passed, 2 failed)} + the code
// object.  A file that is truncated, damaged or of another format is deleted and the
// kernel recompiled.
static const char kCacheMagic[9] = "SFCO0003";
static const size_t kCacheHeader = 32;

static bool read_cache_file(const std::string& path, std::vector<char>& code, int* verdict = nullptr) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  f.close();
  bool ok = blob.size() > kCacheHeader && std::memcmp(blob.data(), kCacheMagic, 8) == 0;
  uint64_t v = 0;
  if (ok) {
    uint64_t size = 0, hash = 0;
    std::memcpy(&size, blob.data() + 8, 8);
    std::memcpy(&hash, blob.data() + 16, 8);
    std::memcpy(&v, blob.data() + 24, 8);
    ok = size == blob.size() - kCacheHeader && size > 4 && v <= 2 &&
         std::memcmp(blob.data() + kCacheHeader, "\177ELF", 4) == 0 &&
         hash == fnv1a(std::string(blob.data() + kCacheHeader, blob.size() - kCacheHeader));
  }
  if (!ok) {
    std::remove(path.c_str());  // stale or corrupt: never hand it to the loader
    return false;
  }
  code.assign(blob.begin() + kCacheHeader, blob.end());
  if (verdict) *verdict = (int)v;
  return true;
}

static void write_cache_file(const std::string& path, const std::vector<char>& code, int verdict = 0) {
  const std::string tmp = path + "." + std::to_string((long)getpid());
  std::ofstream f(tmp, std::ios::binary);
  if (!f) return;
  const uint64_t size = code.size(), hash = fnv1a(std::string(code.data(), code.size())), v = (uint64_t)verdict;
  f.write(kCacheMagic, 8);
  f.write(reinterpret_cast<const char*>(&size), 8);
  f.write(reinterpret_cast<const char*>(&hash), 8);
  f.write(reinterpret_cast<const char*>(&v), 8);
  f.write(code.data(), (std::streamsize)code.size());
  f.close();
  if (!f || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}

static std::map<std::string, int> g_verdicts;  // process level, beside g_code_cache (same mutex)

static std::atomic<long> g_cache_hits{0}, g_cache_misses{0}, g_cache_recompiles{0};

// Code object for `source` (+ flags) through the process level and the disk level of the
// cache, compiled when neither has it.  The kernel belongs to no plan yet.
// An operator with hundreds of terms (the generator's 343-point box) is one left-associated sum, 342 parentheses deep,
// where the text is evaluated as it stands (generic kernel, self-check references): clang stops at 256 unless told
// otherwise.  Only sources that need it get the flag (the others keep their names and cache keys).
std::string with_bracket_depth(const std::string& source, const std::string& flags) {
  if (flags.find("-fbracket-depth") != std::string::npos) return flags;
  int depth = 0, deepest = 0;
  for (const char ch : source) {
    if (ch == '(') deepest = std::max(deepest, ++depth);
    else if (ch == ')') --depth;
    else if (ch == '\n') depth = 0;
  }
  return deepest > 200 ? flags + (flags.empty() ? "" : " ") + "-fbracket-depth=4096" : flags;
}

CompiledKernel compile_cached(const std::string& prefix, const std::string& source, const std::string& flags_in) {
  const std::string flags = with_bracket_depth(source, flags_in);
  const std::string keyed = flags.empty() ? source : flags + "\n" + source;
  CompiledKernel k;
  k.name = prefix + "_" + hex8(fnv1a(keyed));
  k.source = source;
  k.flags = flags;
  const std::string key = k.name + "\n" + keyed;
  k.cache_key = key;
  bool cached = false;
  {
    std::lock_guard<std::mutex> lock(g_code_cache_mutex);
    auto c = g_code_cache.find(key);
    if (c != g_code_cache.end()) {
      k.code = c->second;
      k.verdict = g_verdicts.count(key) ? g_verdicts[key] : 0;
      cached = true;
    }
  }
  if (!cached) {
    // second level: code objects on disk, keyed by source, name and hipRTC version
    const std::string path = disk_cache_path(key);
    if (!path.empty()) cached = read_cache_file(path, k.code, &k.verdict);
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
    g_verdicts[key] = k.verdict;
  }
  read_metadata(k);
  return k;
}

void record_verdict(CompiledKernel& k, int verdict, bool persist) {
  k.verdict = verdict;
  if (k.foreign || k.cache_key.empty()) return;  // (hand-assembled diagnostics objects are never cached)
  {
    std::lock_guard<std::mutex> lock(g_code_cache_mutex);
    g_verdicts[k.cache_key] = verdict;
  }
  if (!persist) return;  // (this process only)
  const std::string path = disk_cache_path(k.cache_key);
  if (!path.empty()) write_cache_file(path, k.code, verdict);
}

int intern_kernel(sf_plan& pl, const std::string& prefix, const std::string& source, const std::string& flags_in) {
  // (diagnostics: $SF_HIP_EXTRA_FLAGS adds compiler flags to every kernel, e.g.
  // "-mllvm -amdgpu-spill-sgpr-to-vgpr=0"; they become part of name and cache key)
  std::string flags = flags_in;
  bool env_flags = false;
  if (const char* extra = std::getenv("SF_HIP_EXTRA_FLAGS"))
    if (*extra) {
      flags += (flags.empty() ? "" : " ") + std::string(extra);
      env_flags = true;
    }
  flags = with_bracket_depth(source, flags);
  // (kernels without extra flags keep the names and cache keys they always had)
  const std::string keyed = flags.empty() ? source : flags + "\n" + source;
  auto it = pl.kernel_by_source.find(keyed);
  if (it != pl.kernel_by_source.end()) return it->second;
  CompiledKernel k;
  bool have = false;
  // (diagnostics: $SF_HIP_OBJECT_DIR/<kernel name>.co, a code object assembled by hand --
  // e.g. the compiler's own output with instructions padded or moved, tools/asm_objects.py --
  // takes the place of the compiler's; nothing is cached)
  if (const char* dir = std::getenv("SF_HIP_OBJECT_DIR")) {
    const std::string name = prefix + "_" + hex8(fnv1a(keyed));
    std::ifstream f(std::string(dir) + "/" + name + ".co", std::ios::binary);
    if (f) {
      k.name = name;
      k.source = source;
      k.flags = flags;
      k.code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      k.foreign = true;
      read_metadata(k);
      have = true;
    }
  }
  if (!have) k = compile_cached(prefix, source, flags);
  k.env_flags = env_flags;
  pl.kernels.push_back(std::move(k));
  pl.kernel_by_source[keyed] = (int)pl.kernels.size() - 1;
  return (int)pl.kernels.size() - 1;
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
bool kernel_unsafe(const CompiledKernel& k) {
  static const bool tolerate = std::getenv("SF_HIP_UNSAFE_SGPR_SPILLS") != nullptr;
  // (SF_HIP_STRICT_SGPR_SPILLS=1: round 2's first criterion, any SGPR spill, on top)
  static const bool strict = std::getenv("SF_HIP_STRICT_SGPR_SPILLS") != nullptr;
  // (an object the plan-time self-check has seen differ from the generic operator kernels -- verdict 2,
  // remembered in both cache levels -- is wrong whatever the detector says)
  return k.verdict == 2 || ((k.late_exec_restores > 0 || ((strict || k.late_exec_restores < 0) && k.sgpr_spills > 0)) && !tolerate);
}
bool kernel_slow(const CompiledKernel& k) {
  return std::max(0, k.spills) + std::max(0, k.scratch) + std::max(0, k.agprs) > 0;
}

void recompile_kernel(CompiledKernel& k) {
  const std::string key = k.name + "\n" + (k.flags.empty() ? k.source : k.flags + "\n" + k.source);
  const std::string path = disk_cache_path(key);
  if (!path.empty()) std::remove(path.c_str());
  compile_kernel(k);
  ++g_cache_recompiles;
  k.from_disk = false;
  k.verdict = 0;  // another object: to be checked again
  k.cache_key = key;
  read_metadata(k);
  {
    std::lock_guard<std::mutex> lock(g_code_cache_mutex);
    g_code_cache[key] = k.code;
    g_verdicts[key] = 0;
  }
  if (!path.empty()) write_cache_file(path, k.code);
}

void code_cache_stats(long* disk_hits, long* compiled, long* rebuilt, bool drop_process_level) {
  if (disk_hits) *disk_hits = g_cache_hits.load();
  if (compiled) *compiled = g_cache_misses.load();
  if (rebuilt) *rebuilt = g_cache_recompiles.load();
  if (drop_process_level) {
    std::lock_guard<std::mutex> lock(g_code_cache_mutex);
    g_code_cache.clear();
    g_verdicts.clear();
  }
}

}  // namespace sf
