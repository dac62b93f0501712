// Run-time compilation of ONE instance of the ionic step kernel (see beat_ode_jit.h): source written to the cache directory,
// hipcc --genco for gfx950 as a child process, the code object kept on disk under a name that hashes the instance AND the kernel
// sources it was built from, loaded with hipModuleLoadData.  Nothing here runs on the host in place of a kernel: when an
// instance cannot be had the caller launches the run-time-index kernel of the library.
#include "beat_ode_jit.h"

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <vector>

extern char** environ;

namespace {
#define BEAT_STR2(x) #x
#define BEAT_STR(x) BEAT_STR2(x)
// the flags of the library's own build of beat_ode.hip (csrc/Makefile: CXXFLAGS + FLAGS_beat_ode), device side only
#ifndef BEAT_ARCH
#define BEAT_ARCH gfx950  // csrc/Makefile passes -DBEAT_ARCH=$(ARCH): an instance is compiled for the architecture the library was
#endif
const char* const kFlags[] = {"--genco", "--offload-arch=" BEAT_STR(BEAT_ARCH), "-O3", "-std=c++17", "-ffp-contract=off",
                              "-DBEAT_ODE_WAVES=" BEAT_STR(BEAT_ODE_WAVES), "-DBEAT_ODE_WAVES_PER_NODE=" BEAT_STR(BEAT_ODE_WAVES_PER_NODE),
                              "-mllvm", "-disable-machine-licm", "-w"};

struct JitState {
  std::mutex m;
  bool init = false, usable = false, verbose = false;
  std::string srcdir, cachedir, hipcc, srchash, why, extra;
  std::map<std::string, hipFunction_t> fn;  // key@device
  std::set<std::string> failed;
  std::vector<hipModule_t> modules;
  long long loaded = 0, compiled = 0, disk_hits = 0, failures = 0;
};
JitState& state() {
  static JitState s;
  return s;
}

}  // namespace
#include "beat_jit_hash.h"  // fnv, beat_jit_source_hash: shared with the build-time tool that writes beat_build_hash.h
#if __has_include("beat_build_hash.h")
#include "beat_build_hash.h"  // BEAT_BUILD_SRC_HASH (csrc/Makefile)
#endif
namespace {
using beat_jit_hash::fnv;
using beat_jit_hash::beat_jit_source_hash;

bool is_file(const std::string& p) {
  struct stat st;
  return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// The cache directory holds code objects this process will LOAD AND RUN: it has to be this user's own and closed to everybody
// else (a directory another local user created in /tmp under the predictable name could hold planted objects).
// Created with mode 0700; an existing one is accepted only if lstat says: a directory (not a link), owned by us, no group /
// other write permission.
bool make_dirs(const std::string& p) {
  std::string cur;
  for (size_t i = 0; i <= p.size(); ++i) {
    if (i == p.size() || p[i] == '/') {
      if (!cur.empty() && ::mkdir(cur.c_str(), i == p.size() ? 0700 : 0755) != 0 && errno != EEXIST) return false;
    }
    if (i < p.size()) cur += p[i];
  }
  struct stat st;
  if (::lstat(p.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  if (st.st_uid != ::getuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) return false;
  return ::access(p.c_str(), W_OK) == 0;
}

std::string read_file(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  std::ostringstream o;
  o << f.rdbuf();
  return o.str();
}

void init_locked(JitState& s) {
  s.init = true;
  s.verbose = std::getenv("BEAT_JIT_VERBOSE") != nullptr;
  // the kernel sources: BEAT_JIT_SRC, or csrc/ beside the package the library was loaded from (<pkg>/beat/lib/libbeat_hip.so)
  if (const char* d = std::getenv("BEAT_JIT_SRC")) {
    s.srcdir = d;
  } else {
    Dl_info info;
    if (dladdr((const void*)&beat_jit_enabled, &info) && info.dli_fname) {
      std::string lib = info.dli_fname;
      char real[4096];
      if (::realpath(lib.c_str(), real)) lib = real;
      const size_t cut = lib.rfind("/beat/lib/");
      if (cut != std::string::npos) s.srcdir = lib.substr(0, cut) + "/csrc";
    }
  }
  if (s.srcdir.empty() || !is_file(s.srcdir + "/beat_ode_kernel.h") || !is_file(s.srcdir + "/../../include/beat_hip.h")) {
    s.why = "kernel sources not found (BEAT_JIT_SRC, or csrc/ and include/ beside the package)";
    return;
  }
  if (const char* h = std::getenv("BEAT_HIPCC")) s.hipcc = h;
  else if (::access("/opt/rocm/bin/hipcc", X_OK) == 0) s.hipcc = "/opt/rocm/bin/hipcc";
  else s.hipcc = "hipcc";  // looked up on PATH by posix_spawnp
  std::vector<std::string> cands;
  if (const char* c = std::getenv("BEAT_JIT_CACHE")) cands.push_back(c);
  if (const char* x = std::getenv("XDG_CACHE_HOME")) cands.push_back(std::string(x) + "/beat_hip");
  if (const char* h = std::getenv("HOME")) cands.push_back(std::string(h) + "/.cache/beat_hip");
  cands.push_back("/tmp/beat_hip_" + std::to_string((long long)::getuid()));
  for (const std::string& c : cands) {
    if (make_dirs(c)) {
      s.cachedir = c;
      break;
    }
  }
  if (s.cachedir.empty()) {
    s.why = "no writable cache directory (BEAT_JIT_CACHE)";
    return;
  }
  // what a cached code object was built from: every header of csrc/, include/beat_hip.h, the flags, the compiler
  // The sources an instance would be compiled from must be the sources THIS library was compiled from (an edited header without a
  // rebuilt library, another BEAT_ODE_WAVES or ARCH: the instance's kernel-argument layout or architecture would differ from what
  // the host code passes -- silently).  csrc/Makefile stores the hash of its headers and flags in the library (beat_build_hash.h,
  // written by tools/jit_hash.cpp with the function below); a mismatch switches run-time compilation off.
  const unsigned long long hsrc = beat_jit_source_hash(s.srcdir);
#ifdef BEAT_BUILD_SRC_HASH
  if (hsrc != BEAT_BUILD_SRC_HASH && std::getenv("BEAT_JIT_ANY_SOURCES") == nullptr) {
    char msg[256];
    std::snprintf(msg, sizeof msg, "the kernel sources in %s (hash %016llx) are not the ones this library was built from (%016llx): rebuild",
                  s.srcdir.c_str(), hsrc, (unsigned long long)BEAT_BUILD_SRC_HASH);
    s.why = msg;
    return;
  }
#endif
  unsigned long long h = fnv("beat-jit-2", hsrc);
  for (const char* f : kFlags) h = fnv(f, h);
  {  // the compiler: another hipcc (another ROCm) gets its own objects -- path, size and modification time of the resolved binary
    std::string cc = s.hipcc;
    char real[4096];
    if (cc.find('/') != std::string::npos && ::realpath(cc.c_str(), real)) cc = real;
    struct stat st;
    h = fnv(cc, h);
    if (::stat(cc.c_str(), &st) == 0) h = fnv(std::to_string((long long)st.st_size) + ":" + std::to_string((long long)st.st_mtime), h);
  }
#ifdef BEAT_BUILD_EXTRA_FLAGS
  // the EXTRA flags this library was built with (csrc/Makefile: variant libraries for A/B runs, -DBEAT_ODE_NT=3 ...): an instance
  // compiled at run time is built with them too and cached under its own key
  s.extra = BEAT_BUILD_EXTRA_FLAGS;
  h = fnv(s.extra, h);
#endif
  if (const char* x = std::getenv("BEAT_JIT_EXTRA_FLAGS")) {  // experiments with the compiler (tools/jit_flags_ab.sh): part of the cache key
    s.extra += std::string(s.extra.empty() ? "" : " ") + x;
    h = fnv(x, h);
  }
  char buf[32];
  std::snprintf(buf, sizeof buf, "%016llx", h);
  s.srchash = buf;
  s.usable = true;
}

// hipcc as a child process, output to `log`; true when it exited with 0
bool run_hipcc(const JitState& s, const std::string& src, const std::string& out, const std::string& log) {
  std::vector<std::string> a{s.hipcc};
  for (const char* f : kFlags) a.push_back(f);
  {
    std::istringstream extra(s.extra);
    for (std::string f; extra >> f;) a.push_back(f);
  }
  a.push_back("-I" + s.srcdir);
  a.push_back(src);
  a.push_back("-o");
  a.push_back(out);
  std::vector<char*> argv;
  for (std::string& x : a) argv.push_back(&x[0]);
  argv.push_back(nullptr);
  // The child gets the parent's environment MINUS whatever makes a tool library load into it: under rocprofv3 (LD_PRELOAD,
  // ROCP_TOOL_LIBRARIES, HSA_TOOLS_LIB ...) the preloaded tool initialises the GPU inside hipcc, which then exec's clang and lld --
  // the exec of a process that has touched the GPU, which this pool forbids.  The compiler needs none of them.
  std::vector<std::string> envs;
  for (char** e = environ; e != nullptr && *e != nullptr; ++e) {
    static const char* const drop[] = {"LD_PRELOAD=", "LD_AUDIT=", "ROCP", "ROCPROF", "ROCTRACER", "ROCTX", "HSA_TOOLS", "RPD_", "OMNITRACE",
                                       "ROCM_SYSTEMS", "AMD_LOG_LEVEL=", "HIP_TRACE", "HSA_ENABLE_DEBUG"};
    bool skip = false;
    for (const char* d : drop) skip = skip || std::strncmp(*e, d, std::strlen(d)) == 0;
    if (!skip) envs.push_back(*e);
  }
  std::vector<char*> envp;
  for (std::string& x : envs) envp.push_back(&x[0]);
  envp.push_back(nullptr);
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  ::unlink(log.c_str());  // (a link somebody left under the log's name is removed, not followed)
  posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
  posix_spawn_file_actions_adddup2(&fa, 1, 2);
  pid_t pid = 0;
  const int rc = posix_spawnp(&pid, s.hipcc.c_str(), &fa, nullptr, argv.data(), envp.data());
  posix_spawn_file_actions_destroy(&fa);
  if (rc != 0) return false;
  // a compile takes seconds; one that has not finished after BEAT_JIT_TIMEOUT_S (default 300) is killed: a step must not hang on it
  const char* te = std::getenv("BEAT_JIT_TIMEOUT_S");
  const double limit = te ? std::max(1.0, std::atof(te)) : 300.0;
  int status = 0;
  for (double waited = 0.0;; waited += 0.02) {
    const pid_t w = ::waitpid(pid, &status, WNOHANG);
    if (w == pid) break;
    if (w < 0 && errno != EINTR) return false;
    if (waited > limit) {
      ::kill(pid, SIGKILL);
      while (::waitpid(pid, &status, 0) < 0 && errno == EINTR) {
      }
      return false;
    }
    ::usleep(20000);
  }
  return WIFEXITED(status) && WEXITSTATUS(status) == 0;
}

// the kernel's symbol in a code object (an offload bundle or a bare ELF): the NUL-terminated string that starts with _Z, names
// ode_step_kernel (or ode_run_kernel: a translation unit instantiates ONE kernel) and carries no suffix (.kd, .private_seg_size ...
// are the descriptor and its metadata)
std::string kernel_symbol(const std::string& blob, const char* stem);
std::string kernel_symbol(const std::string& blob) {
  const std::string s = kernel_symbol(blob, "ode_step_kernel");
  return s.empty() ? kernel_symbol(blob, "ode_run_kernel") : s;
}
std::string kernel_symbol(const std::string& blob, const char* stem) {
  size_t pos = 0;
  while ((pos = blob.find(stem, pos)) != std::string::npos) {
    size_t b = pos;
    while (b > 0 && blob[b - 1] != '\0' && (pos - b) < 8) --b;
    size_t e = pos;
    while (e < blob.size() && blob[e] != '\0') ++e;
    const std::string name = blob.substr(b, e - b);
    if (name.size() > 2 && name[0] == '_' && name[1] == 'Z' && name.find('.') == std::string::npos &&
        std::all_of(name.begin(), name.end(), [](unsigned char c) { return c > 32 && c < 127; }))
      return name;
    pos = e;
  }
  return "";
}

void fail_locked(JitState& s, const std::string& full_key, const std::string& msg) {
  s.failed.insert(full_key);
  s.failures += 1;
  s.why = msg;
  std::fprintf(stderr, "[libbeat_hip] run-time compilation unavailable for %s: %s -- using the run-time-index kernel\n", full_key.c_str(),
               msg.c_str());
}
}  // namespace

bool beat_jit_enabled() {
  const char* e = std::getenv("BEAT_JIT");  // read per call: a caller may switch routes between steps (tests, A/B runs)
  if (e && e[0] == '0') return false;
  JitState& s = state();
  std::lock_guard<std::mutex> lock(s.m);
  if (!s.init) init_locked(s);
  return s.usable;
}

hipFunction_t beat_jit_lookup(beat_ctx* ctx, const std::string& key, bool* known) {
  JitState& s = state();
  std::lock_guard<std::mutex> lock(s.m);
  const std::string full = key + "@" + std::to_string(ctx->device);
  auto it = s.fn.find(full);
  *known = it != s.fn.end() || s.failed.count(full) != 0;
  return it != s.fn.end() ? it->second : nullptr;
}

void beat_jit_reject(beat_ctx* ctx, const std::string& key, const std::string& why) {
  JitState& s = state();
  std::lock_guard<std::mutex> lock(s.m);
  const std::string full = key + "@" + std::to_string(ctx->device);
  s.fn.erase(full);
  fail_locked(s, full, why);
}

hipFunction_t beat_jit_get(beat_ctx* ctx, const std::string& key, const std::string& source) {
  JitState& s = state();
  std::lock_guard<std::mutex> lock(s.m);
  if (!s.init) init_locked(s);
  if (!s.usable) return nullptr;
  const std::string full = key + "@" + std::to_string(ctx->device);
  auto it = s.fn.find(full);
  if (it != s.fn.end()) return it->second;
  if (s.failed.count(full)) return nullptr;
  char hx[32];
  std::snprintf(hx, sizeof hx, "%016llx", fnv(key, fnv(s.srchash)));
  const std::string base = s.cachedir + "/beatjit_" + hx;
  const std::string obj = base + ".hsaco";
  if (is_file(obj)) {
    s.disk_hits += 1;
  } else {
    const std::string tag = "." + std::to_string((long long)::getpid());
    const std::string src = base + tag + ".hip", tmp = base + tag + ".tmp", log = base + tag + ".log";
    {
      std::ofstream f(src);
      f << "// " << key << "\n" << source;
    }
    if (s.verbose) std::fprintf(stderr, "[libbeat_hip] compiling %s -> %s\n", key.c_str(), obj.c_str());
    const bool ok = run_hipcc(s, src, tmp, log);
    ::unlink(src.c_str());
    if (!ok || !is_file(tmp)) {
      ::unlink(tmp.c_str());
      fail_locked(s, full, "hipcc failed (" + s.hipcc + ", log " + log + ")");
      return nullptr;
    }
    s.compiled += 1;
    if (::rename(tmp.c_str(), obj.c_str()) != 0) {  // (another process may have put the same object there: either will do)
      ::unlink(tmp.c_str());
      if (!is_file(obj)) {
        fail_locked(s, full, "cannot write " + obj);
        return nullptr;
      }
    }
  }
  const std::string blob = read_file(obj);
  const std::string sym = kernel_symbol(blob);
  hipModule_t mod = nullptr;
  hipFunction_t f = nullptr;
  if (sym.empty() || hipSetDevice(ctx->device) != hipSuccess || hipModuleLoadData(&mod, blob.data()) != hipSuccess ||
      hipModuleGetFunction(&f, mod, sym.c_str()) != hipSuccess || f == nullptr) {
    (void)hipGetLastError();
    if (mod) (void)hipModuleUnload(mod);
    ::unlink(obj.c_str());  // a stale or truncated object: compiled again next time
    fail_locked(s, full, "the code object " + obj + " did not load");
    return nullptr;
  }
  s.modules.push_back(mod);
  s.fn[full] = f;
  s.loaded += 1;
  return f;
}

// out[4]: kernels loaded in this process, hipcc runs, code objects taken from the cache directory, failures.  Returns 1 when
// run-time compilation is usable here (sources, compiler and cache directory found), 0 otherwise.
extern "C" int beat_ode_jit_stats(long long* host_out) {
  JitState& s = state();
  std::lock_guard<std::mutex> lock(s.m);
  if (!s.init) init_locked(s);
  if (host_out) {
    host_out[0] = s.loaded;
    host_out[1] = s.compiled;
    host_out[2] = s.disk_hits;
    host_out[3] = s.failures;
  }
  if (!s.usable) beat_set_error("run-time compilation unavailable: %s", s.why.c_str());  // (beat_last_error says why)
  return s.usable ? 1 : 0;
}

// ---- cell models that are not shipped: registered as source, compiled at first use (beat.models.from_ode) -----------------------------
namespace {
struct CustomModel {
  std::string name, source;
  int ns, np, v_index;
};
std::vector<CustomModel>& customs() {
  static std::vector<CustomModel> v;
  return v;
}
std::mutex& customs_mutex() {
  static std::mutex m;
  return m;
}
}  // namespace

extern "C" int beat_ode_model_register(const char* name, const char* source, int num_states, int num_params, int v_index, int* model_id_out) {
  BEAT_REQUIRE(name != nullptr && source != nullptr && model_id_out != nullptr, "null argument");
  BEAT_REQUIRE(num_states >= 1 && num_states <= 512 && num_params >= 1 && num_params <= 1024, "bad state / parameter count");
  BEAT_REQUIRE(v_index >= 0 && v_index < num_states, "v_index %d out of range", v_index);
  for (const char* c = name; *c; ++c)
    BEAT_REQUIRE((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '_', "model name: [A-Za-z0-9_]+");
  BEAT_REQUIRE(std::string(source).find(std::string("struct ") + name) != std::string::npos, "the source does not define struct %s", name);
  BEAT_REQUIRE(beat_jit_enabled(), "a model given as source needs run-time compilation, which is not available here (beat_ode_jit_stats)");
  std::lock_guard<std::mutex> lock(customs_mutex());
  std::vector<CustomModel>& v = customs();
  for (size_t k = 0; k < v.size(); ++k)
    if (v[k].name == name && v[k].source == source) {
      *model_id_out = BEAT_MODEL_CUSTOM_BASE + (int)k;
      return BEAT_OK;
    }
  v.push_back(CustomModel{name, source, num_states, num_params, v_index});
  *model_id_out = BEAT_MODEL_CUSTOM_BASE + (int)v.size() - 1;
  return BEAT_OK;
}

int beat_custom_model_info(int model_id, int* ns, int* np, int* v_index) {
  std::lock_guard<std::mutex> lock(customs_mutex());
  const int k = model_id - BEAT_MODEL_CUSTOM_BASE;
  if (k < 0 || k >= (int)customs().size()) return BEAT_EINVAL;
  if (ns) *ns = customs()[k].ns;
  if (np) *np = customs()[k].np;
  if (v_index) *v_index = customs()[k].v_index;
  return BEAT_OK;
}

// A kernel of a registered model: the translation unit is the registered source behind beat_ode_kernel.h plus ONE explicit
// instantiation; `what` names it in the cache key.
namespace {
int custom_model(int model_id, CustomModel& m) {
  std::lock_guard<std::mutex> lock(customs_mutex());
  const int k = model_id - BEAT_MODEL_CUSTOM_BASE;
  BEAT_REQUIRE(k >= 0 && k < (int)customs().size(), "unknown model id %d", model_id);
  m = customs()[k];
  return BEAT_OK;
}
hipFunction_t custom_instance(beat_ctx* ctx, const CustomModel& m, const std::string& what, const std::string& instantiation) {
  char hx[32];
  std::snprintf(hx, sizeof hx, "%016llx", fnv(m.source));
  const std::string key = "custom_" + m.name + "_" + what + "_" + hx;
  bool known = false;
  hipFunction_t f = beat_jit_lookup(ctx, key, &known);
  if (!known) {
    std::string src = "// written by libbeat_hip (beat_ode_model_register): a cell model given as source\n#include \"beat_ode_kernel.h\"\n";
    src += m.source;
    src += "\n" + instantiation + "\n";
    f = beat_jit_get(ctx, key, src);
  }
  if (f == nullptr)
    beat_set_error("the kernel of model %s could not be compiled (see the log in the cache directory; BEAT_JIT_VERBOSE=1)", m.name.c_str());
  return f;
}
}  // namespace

namespace {
std::set<std::string>& custom_checked() {
  static std::set<std::string> v;
  return v;
}

// A variant instance of a registered model (per-node rows, pending update, classes) against the model's PLAIN instance, once per
// instance and process: one step of the caller's first nodes on two scratch copies, the variant given what makes it compute the
// plain step (every node the parameters of node 0 / class 0, nothing pending) -- same arithmetic, same values.  Why: the kernel of
// a big generated model is heavily spilled (the reference's ToR-ORd files: ~400 SGPRs and ~500 VGPRs), and with ROCm 7.2 one of two
// instances of such a kernel has been seen to reload registers under another lane mask than it spilled them under: wrong values on
// the nodes that took the other arm of a branch, in ONE instance, the other one right (tools/diag_spill.py,
// profiles/r05_generated_spills.md; the generator has emitted branch-free code since, which removes the trigger that was found).
// The plain instance itself is held against the NumPy evaluation of the same expressions when the model is registered from
// Python (beat/models/ode_file.py).  A variant that fails is rejected: the call fails, nothing falls back.
int custom_cross_check(beat_ctx* ctx, const CustomModel& m, hipFunction_t f, const std::string& what, bool per_node, bool marked,
                       const double* states, int64_t n, int64_t ld, const double* host_params, const double* ppn, int64_t pld,
                       const MarkedArgs& mk_in, double t, double dt, int v_index) {
  if (const char* e = std::getenv("BEAT_JIT_SELF_CHECK"))
    if (e[0] == '0') return BEAT_OK;
  const std::string key = m.name + "/" + what + "@" + std::to_string(ctx->device);
  {
    std::lock_guard<std::mutex> lock(customs_mutex());
    if (custom_checked().count(key)) return BEAT_OK;
  }
  hipFunction_t f0 = custom_instance(ctx, m, "step_n0p0m0",
      "template __global__ void ode_step_kernel<" + m.name + ", false, false, false>(\n    double*, int64_t, int64_t, ParamPack<" + m.name +
      "::NP>, typename " + m.name + "::Derived, const double*, int64_t, double, double, int, double*, PendingV, MarkedArgs, SparseRows);");
  if (f0 == nullptr) return BEAT_EINVAL;
  int64_t nc = std::min<int64_t>(n, 1024);
  if (nc < 1) return BEAT_OK;
  // the parameters of node 0 / class 0
  std::vector<double> p0(m.np, 1.0);
  if (host_params != nullptr) {
    p0.assign(host_params, host_params + m.np);
  } else if (per_node) {
    BEAT_HIP_CHECK(hipMemcpy2DAsync(p0.data(), sizeof(double), ppn, sizeof(double) * (size_t)pld, sizeof(double), (size_t)m.np,
                                    hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  } else if (marked) {
    BEAT_HIP_CHECK(hipMemcpyAsync(p0.data(), mk_in.table, sizeof(double) * (size_t)m.np, hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  const size_t ns = (size_t)m.ns, np = (size_t)m.np;
  const size_t doubles = 2 * ns * (size_t)nc + np * (size_t)nc + (np + 1) + ((size_t)nc + 7) / 8;
  double* scratch = nullptr;
  BEAT_HIP_CHECK(hipMalloc(&scratch, sizeof(double) * doubles));
  struct Free {
    double* p;
    ~Free() { (void)hipFree(p); }
  } guard{scratch};
  double* sa = scratch;
  double* sb = sa + ns * (size_t)nc;
  double* rows = sb + ns * (size_t)nc;
  double* table = rows + np * (size_t)nc;
  unsigned char* marks = (unsigned char*)(table + np + 1);
  for (size_t k = 0; k < ns; ++k) {
    BEAT_HIP_CHECK(hipMemcpyAsync(sa + k * nc, states + (int64_t)k * ld, sizeof(double) * nc, hipMemcpyDeviceToDevice, ctx->stream));
    BEAT_HIP_CHECK(hipMemcpyAsync(sb + k * nc, states + (int64_t)k * ld, sizeof(double) * nc, hipMemcpyDeviceToDevice, ctx->stream));
  }
  std::vector<double> hrows(np * (size_t)nc), htab(np + 1, 0.0);
  for (size_t k = 0; k < np; ++k) {
    htab[k] = p0[k];
    for (int64_t i = 0; i < nc; ++i) hrows[k * nc + i] = p0[k];
  }
  BEAT_HIP_CHECK(hipMemcpyAsync(rows, hrows.data(), sizeof(double) * hrows.size(), hipMemcpyHostToDevice, ctx->stream));
  BEAT_HIP_CHECK(hipMemcpyAsync(table, htab.data(), sizeof(double) * htab.size(), hipMemcpyHostToDevice, ctx->stream));
  BEAT_HIP_CHECK(hipMemsetAsync(marks, 0, (size_t)nc, ctx->stream));
  PendingV none{nullptr, 0, nullptr, 0, {}, nullptr, 0};
  MarkedArgs mk0{nullptr, nullptr, 0, nullptr, nullptr};
  MarkedArgs mkv = marked ? MarkedArgs{marks, table, m.np + 1, nullptr, nullptr} : mk0;
  SparseRows sp{{0}, 0};
  double drv = 0.0;
  double* vc = nullptr;
  int64_t ldc = nc;
  const double* ppn_v = per_node ? rows : nullptr;
  const double* ppn_0 = nullptr;
  int64_t pld_v = per_node ? nc : 0, pld_0 = 0;
  const unsigned grid = (unsigned)((nc + BEAT_BLOCK - 1) / BEAT_BLOCK);
  {
    void* args[] = {&sa, &nc, &ldc, p0.data(), &drv, &ppn_v, &pld_v, &t, &dt, &v_index, &vc, &none, &mkv, &sp};
    BEAT_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, BEAT_BLOCK, 1, 1, 0, ctx->stream, args, nullptr));
  }
  {
    void* args[] = {&sb, &nc, &ldc, p0.data(), &drv, &ppn_0, &pld_0, &t, &dt, &v_index, &vc, &none, &mk0, &sp};
    BEAT_HIP_CHECK(hipModuleLaunchKernel(f0, grid, 1, 1, BEAT_BLOCK, 1, 1, 0, ctx->stream, args, nullptr));
  }
  std::vector<double> h(2 * ns * (size_t)nc);
  BEAT_HIP_CHECK(hipMemcpyAsync(h.data(), scratch, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  const double* a = h.data();
  const double* b = h.data() + ns * (size_t)nc;
  for (size_t k = 0; k < ns; ++k) {
    double scale = 0.0;
    for (int64_t i = 0; i < nc; ++i) {
      const double v = std::fabs(b[k * nc + i]);
      if (v == v && v > scale && v < 1e300) scale = v;
    }
    for (int64_t i = 0; i < nc; ++i) {
      const double x = a[k * nc + i], y = b[k * nc + i];
      if (x != x && y != y) continue;  // both NaN (a caller's garbage in, the same garbage out)
      if (!(std::fabs(x - y) <= 1e-10 * std::fabs(y) + 1e-13 * scale)) {
        char msg[320];
        std::snprintf(msg, sizeof msg, "instance %s of model %s differs from the model's plain instance (state %d, node %lld: %.17g against %.17g): "
                      "miscompiled (heavily spilled) kernel; try other BEAT_JIT_EXTRA_FLAGS", what.c_str(), m.name.c_str(), (int)k, (long long)i, x, y);
        beat_set_error("%s", msg);
        return BEAT_EINVAL;
      }
    }
  }
  std::lock_guard<std::mutex> lock(customs_mutex());
  custom_checked().insert(key);
  return BEAT_OK;
}
}  // namespace

// One step of a registered model, every form the shipped models' step takes except the compiled sparse rows: uniform parameters,
// all per-node rows, or parameter classes (markers + table, optionally the compact layout's node map), each with or without a pending
// update.  The kernel is ode_step_kernel<Name, PER_NODE, PEND, MARKED> of csrc/beat_ode_kernel.h.
int beat_custom_step(beat_ctx* ctx, int model_id, unsigned grid, double* states, int64_t n, int64_t ld, const double* host_params,
                     int num_params, const double* ppn, int64_t pld, double t, double dt, int v_index, double* v_copy,
                     const PendingV& pend_in, const MarkedArgs& mk_in) {
  CustomModel m;
  if (int rc = custom_model(model_id, m)) return rc;
  const bool marked = mk_in.markers != nullptr, per_node = ppn != nullptr;
  BEAT_REQUIRE(!(marked && per_node) && (!marked || mk_in.table != nullptr), "parameter classes come with a table, not with per-node rows");
  BEAT_REQUIRE(marked || ((host_params != nullptr || per_node) && num_params == m.np),
               "model %s expects %d parameters (a host vector, per-node rows or classes), got %d", m.name.c_str(), m.np, num_params);
  BEAT_REQUIRE(!per_node || pld >= n, "params_ld %lld < n %lld", (long long)pld, (long long)n);
  const bool have_pend = pend_in.count > 0 || pend_in.gt.d != nullptr || pend_in.dev_st != nullptr;
  BEAT_REQUIRE(!have_pend || v_index == m.v_index, "a pending update needs v_index = %d (the model's membrane potential), got %d", m.v_index, v_index);
  BEAT_REQUIRE(v_copy == nullptr || (v_index >= 0 && v_index < m.ns), "v_index %d out of range", v_index);
  BEAT_REQUIRE(!marked || v_copy == nullptr || v_index == m.v_index, "the class kernel mirrors the model's potential (row %d), not row %d",
               m.v_index, v_index);
  const char* tf[2] = {"false", "true"};
  const std::string what = std::string("step_n") + (per_node ? "1" : "0") + "p" + (have_pend ? "1" : "0") + "m" + (marked ? "1" : "0");
  hipFunction_t f = custom_instance(ctx, m, what,
      "template __global__ void ode_step_kernel<" + m.name + ", " + tf[per_node] + ", " + tf[have_pend] + ", " + tf[marked] + ">(\n    double*, int64_t, int64_t, ParamPack<" +
      m.name + "::NP>, typename " + m.name + "::Derived, const double*, int64_t, double, double, int, double*, PendingV, MarkedArgs, SparseRows);");
  if (f == nullptr) return BEAT_EINVAL;
  if (per_node || have_pend || marked) {
    if (int rc = custom_cross_check(ctx, m, f, what, per_node, marked, states, n, ld, host_params, ppn, pld, mk_in, t, dt, v_index)) return rc;
  }
  std::vector<double> prm(m.np, 1.0);
  if (host_params != nullptr) prm.assign(host_params, host_params + m.np);
  double drv = 0.0;
  PendingV pend = pend_in;
  MarkedArgs mk = mk_in;
  SparseRows sp{{0}, 0};
  void* args[] = {&states, &n, &ld, prm.data(), &drv, &ppn, &pld, &t, &dt, &v_index, &v_copy, &pend, &mk, &sp};
  BEAT_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, BEAT_BLOCK, 1, 1, 0, ctx->stream, args, nullptr));
  return BEAT_OK;
}

// nbeats x nsteps steps of a registered model inside one launch (beat_ode_run): ode_run_kernel<Name, PER_NODE>.
int beat_custom_run(beat_ctx* ctx, int model_id, double* states, int64_t n, int64_t ld, const double* host_params, int num_params,
                    const double* ppn, int64_t pld, double t0, double dt, int64_t nsteps, int nbeats, int save_freq, const int* track_idx,
                    int ntrack, double* trace) {
  CustomModel m;
  if (int rc = custom_model(model_id, m)) return rc;
  const bool per_node = ppn != nullptr;
  BEAT_REQUIRE((host_params != nullptr || per_node) && num_params == m.np, "model %s expects %d parameters (a host vector or per-node rows), got %d",
               m.name.c_str(), m.np, num_params);
  BEAT_REQUIRE(!per_node || pld >= n, "params_ld %lld < n %lld", (long long)pld, (long long)n);
  BEAT_REQUIRE(ntrack >= 0 && ntrack <= 8, "at most 8 tracked states");
  BEAT_REQUIRE(ntrack == 0 || (trace != nullptr && save_freq >= 1 && track_idx != nullptr), "tracking needs a trace buffer and save_freq >= 1");
  TrackSpec tr{};
  tr.n = ntrack;
  for (int a = 0; a < ntrack; ++a) {
    BEAT_REQUIRE(track_idx[a] >= 0 && track_idx[a] < m.ns, "tracked state %d out of range", track_idx[a]);
    tr.idx[a] = track_idx[a];
  }
  hipFunction_t f = custom_instance(ctx, m, std::string("run_n") + (per_node ? "1" : "0"),
      "template __global__ void ode_run_kernel<" + m.name + ", " + (per_node ? "true" : "false") + ">(\n    double*, int64_t, int64_t, ParamPack<" + m.name +
      "::NP>, typename " + m.name + "::Derived, const double*, int64_t, double, double, int64_t, int, int, TrackSpec, double*);");
  if (f == nullptr) return BEAT_EINVAL;
  std::vector<double> prm(m.np, 1.0);
  if (host_params != nullptr) prm.assign(host_params, host_params + m.np);
  double drv = 0.0;
  const unsigned grid = (unsigned)((n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  void* args[] = {&states, &n, &ld, prm.data(), &drv, &ppn, &pld, &t0, &dt, &nsteps, &nbeats, &save_freq, &tr, &trace};
  BEAT_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, BEAT_BLOCK, 1, 1, 0, ctx->stream, args, nullptr));
  return BEAT_OK;
}
