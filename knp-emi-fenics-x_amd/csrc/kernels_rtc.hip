// Membrane models supplied by the user as HIP source, compiled at bind time with hipRTC.
//
// The reference accepts any module that exposes `rhs_numba.address`, a numba cfunc with numbalsoda's signature
// `rhs(t, states, values, parameters)` (src/knpemi/odeSolver.py:96; e.g. examples/benchmark/mm_glial.py:127-215, which
// has another parameter order and other constants than the glial model of the astrocyte example).  A GPU cannot call
// a host function pointer; the MI355X-native counterpart is a plug-in that brings the same function as a
// `__device__` function in HIP source (module attribute RHS_HIP).  It is pasted under the sweep kernel
// (ode_kernel.h: the very code of the shipped models) and compiled for gfx950 when the model is bound; the result is
// launched through the module API.  Parameters are an in/out row, exactly as for the cfunc: whatever the last
// right-hand-side call stored there (the currents I_ch_*) is what the PDEs receive.
#include <hip/hiprtc.h>

#include <cstring>
#include <map>
#include <mutex>

#include "knpemi_internal.h"
#include "ode_kernel.h"

namespace {

// the two headers the generated translation unit includes, embedded at build time (csrc/Makefile)
const char* const SRC_LSODA_CORE =
#include "rtc_lsoda_core.inc"
    ;
const char* const SRC_ODE_KERNEL =
#include "rtc_ode_kernel.inc"
    ;

std::string wrapper_source(int ns, int np, int lanes, const std::string& user) {
  std::string s;
  s += "#include \"ode_kernel.h\"\n";
  s += "// ---- plug-in source -------------------------------------------------------------\n";
  s += user;
  s += "\n// ---- adapter: the model functor the integrator expects ---------------------------\n";
  s += "struct ModelUser {\n";
  s += "  static constexpr int NS = " + std::to_string(ns) + ", NP = " + std::to_string(np) + ", CURRENT_LANE = 0;\n";
  s += "  double par[NP];\n";
  s += "  template <class Row> __device__ void prepare(const Row& p) {\n";
  s += "    _Pragma(\"unroll\") for (int j = 0; j < NP; ++j) par[j] = p[j];\n  }\n";
  s += "  __device__ void rhs(double t, const double* y, double* dy) { ::rhs(t, y, dy, par); }\n";
  s += "  // every lane of a system evaluates the whole right-hand side and keeps its own component\n";
  s += "  __device__ double rhs_lane(int c, double t, const double* y) {\n";
  s += "    double dy[NS];\n    ::rhs(t, y, dy, par);\n    double out = dy[0];\n";
  s += "    _Pragma(\"unroll\") for (int k = 1; k < NS; ++k) out = (k == c) ? dy[k] : out;\n    return out;\n  }\n";
  s += "  template <class Row> __device__ void finish(const Row& p) const {\n";
  s += "    _Pragma(\"unroll\") for (int j = 0; j < NP; ++j) p[j] = par[j];\n  }\n};\n";
  s += "extern \"C\" __global__ __launch_bounds__(ODE_BLOCK, 1) void ode_user_kernel(OdeDev D, OdeArgs a, const LsodaCoef* cf) {\n";
  s += "  ode_step_body<ModelUser, " + std::to_string(lanes) + ", 1, false>(D, a, cf);\n}\n";
  return s;
}

int lanes_for(int ns) { return (ns == 1 || ns == 2 || ns == 4 || ns == 8) ? ns : 1; }

// compile for gfx950; `code` receives the code object, `log` the compiler's messages
int compile(int ns, int np, const std::string& user, std::vector<char>* code, std::string* log) {
  const std::string src = wrapper_source(ns, np, lanes_for(ns), user);
  const char* headers[2] = {SRC_LSODA_CORE, SRC_ODE_KERNEL};
  const char* names[2] = {"lsoda_core.h", "ode_kernel.h"};
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "knpemi_user_model.hip", 2, headers, names) != HIPRTC_SUCCESS) {
    *log = "hiprtcCreateProgram failed";
    return KNPEMI_EHIP;
  }
  // the integrator keeps its state in registers: same flag as csrc/Makefile uses for kernels_ode.hip
  // (the same code generation options as kernels_ode.o, Makefile: no common-code sinking, instruction scheduling for ILP)
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-simplifycfg-sink-common=false",
                        "-mllvm", "-amdgpu-sched-strategy=max-ilp"};
  const hiprtcResult res = hiprtcCompileProgram(prog, 7, opts);
  size_t n = 0;
  if (hiprtcGetProgramLogSize(prog, &n) == HIPRTC_SUCCESS && n > 1) {
    log->assign(n, '\0');
    (void)hiprtcGetProgramLog(prog, &(*log)[0]);
  }
  if (res != HIPRTC_SUCCESS) {
    *log = std::string("hipRTC: ") + hiprtcGetErrorString(res) + "\n" + *log;
    (void)hiprtcDestroyProgram(&prog);
    return KNPEMI_EINVAL;
  }
  if (code) {
    size_t sz = 0;
    if (hiprtcGetCodeSize(prog, &sz) != HIPRTC_SUCCESS) { (void)hiprtcDestroyProgram(&prog); *log = "hiprtcGetCodeSize failed"; return KNPEMI_EHIP; }
    code->resize(sz);
    if (hiprtcGetCode(prog, code->data()) != HIPRTC_SUCCESS) { (void)hiprtcDestroyProgram(&prog); *log = "hiprtcGetCode failed"; return KNPEMI_EHIP; }
  }
  (void)hiprtcDestroyProgram(&prog);
  return KNPEMI_OK;
}

std::mutex g_cache_mutex;
std::map<std::string, std::vector<char>> g_code_cache;    // (ns, np, source) -> code object, per process

}  // namespace

// Compile only (no device needed): the CPU test suite checks plug-in sources with this, and a driver can validate a
// model before the first launch.  log_len bytes of the compiler's messages go to `log` (may be NULL).
extern "C" int knpemi_ode_compile_source(int n_states, int n_params, const char* rhs_source, char* log, size_t log_len) {
  if (!rhs_source || n_states < 1 || n_states > 16 || n_params < 1 || n_params > 128) {
    kn_set_error("knpemi_ode_compile_source: bad argument (1..16 states, 1..128 parameters)");
    return KNPEMI_EINVAL;
  }
  std::string msg;
  std::vector<char> code;
  const int rc = compile(n_states, n_params, rhs_source, &code, &msg);
  if (log && log_len) {
    const size_t n = std::min(log_len - 1, msg.size());
    std::memcpy(log, msg.data(), n);
    log[n] = '\0';
  }
  if (rc) { kn_set_error("membrane model source did not compile:\n" + msg); return rc; }
  std::lock_guard<std::mutex> lock(g_cache_mutex);
  g_code_cache[std::to_string(n_states) + "/" + std::to_string(n_params) + "/" + rhs_source] = std::move(code);
  return KNPEMI_OK;
}

int kn_rtc_bind(knpemi_handle* h, KnOdeModel& m, int n_states, int n_params, const char* rhs_source) {
  const std::string key = std::to_string(n_states) + "/" + std::to_string(n_params) + "/" + rhs_source;
  std::vector<char> code;
  {
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    auto it = g_code_cache.find(key);
    if (it != g_code_cache.end()) code = it->second;
  }
  if (code.empty()) {
    std::string msg;
    const int rc = compile(n_states, n_params, rhs_source, &code, &msg);
    if (rc) { kn_set_error("membrane model source did not compile:\n" + msg); return rc; }
    std::lock_guard<std::mutex> lock(g_cache_mutex);
    g_code_cache[key] = code;
  }
  hipModule_t mod = nullptr;
  KN_HIP(hipModuleLoadData(&mod, code.data()));
  hipFunction_t fn = nullptr;
  if (hipModuleGetFunction(&fn, mod, "ode_user_kernel") != hipSuccess) {
    (void)hipModuleUnload(mod);
    kn_set_error("hipModuleGetFunction(ode_user_kernel) failed");
    return KNPEMI_EHIP;
  }
  m.rtc_module = mod;
  m.rtc_function = fn;
  m.rtc_lanes = lanes_for(n_states);
  h->rtc_modules.push_back(mod);
  return KNPEMI_OK;
}

int kn_rtc_launch(knpemi_handle* h, const KnOdeModel& m, const void* dev_view, size_t dev_bytes, const void* args,
                  size_t args_bytes, const void* coef) {
  // kernel parameters (OdeDev, OdeArgs, const LsodaCoef*) laid out as the compiler lays out the parameter list
  struct Params { OdeDev D; OdeArgs a; const LsodaCoef* cf; } p;
  if (dev_bytes != sizeof(OdeDev) || args_bytes != sizeof(OdeArgs)) { kn_set_error("rtc launch: argument size mismatch"); return KNPEMI_EINVAL; }
  std::memcpy(&p.D, dev_view, sizeof(OdeDev));
  std::memcpy(&p.a, args, sizeof(OdeArgs));
  p.cf = static_cast<const LsodaCoef*>(coef);
  size_t size = sizeof(p);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  const unsigned grid = (unsigned)(((size_t)p.a.nq * m.rtc_lanes + ODE_BLOCK - 1) / ODE_BLOCK);
  KN_HIP(hipModuleLaunchKernel(static_cast<hipFunction_t>(m.rtc_function), grid, 1, 1, ODE_BLOCK, 1, 1, 0, h->cur, nullptr, config));
  return KNPEMI_OK;
}
