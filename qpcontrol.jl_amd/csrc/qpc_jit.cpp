// qpc_jit.cpp -- specialises qpc_tick.h on one controller's QpcDims with hiprtc: every loop bound
// and every LDS offset of the per-QP program becomes a compile-time constant (full unrolling of
// the small dot products, immediate-offset ds_read/ds_write, no kernarg/SGPR traffic for the
// layout).  This is the device-side analogue of Parametron compiling its expression graphs once
// at addtask!/initialize! time (momentum.jl:99-160) instead of interpreting them every tick.
#include "qpc_jit.h"

#include <hip/hiprtc.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unistd.h>
#include <sstream>
#include <string>
#include <vector>

#include "qpc_embedded.inc"

static void put_ints(std::ostringstream &o, const int *a, int n) {
  o << "{";
  for (int i = 0; i < n; i++) o << a[i] << (i + 1 < n ? "," : "");
  o << "}";
}
static std::string dbl(double x) {
  char b[64];
  snprintf(b, sizeof b, "%.17g", x);
  std::string s(b);
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  return s;
}

bool qpc_jit_dynamic_lds() {
  const char *e = QPC_EXP_ENV("QPC_EXP_DYN_LDS");
  return e && e[0] == '1';
}

int qpc_jit_sim_rpw(const QpcDims &D) {
  if (const char *e = QPC_EXP_ENV("QPC_SIM_RPW")) { if (e[0] == '1') return 1; }
  const int slab = qpc_sim_layout(D).total + (qpc_sim_layout(D).total & 1);
  return (D.nb <= 32 && D.nc <= 32 && (size_t)2 * slab * sizeof(double) <= 64 * 1024) ? 2 : 1;
}

static void put_dims(std::ostringstream &o, const char *name, const QpcDims &D) {
  o << "static constexpr QpcDims " << name << " = {" << D.nb << "," << D.nq << "," << D.nv << "," << D.nf << "," << D.npin << ","
    << D.nsj << "," << D.ntask << "," << D.nc << "," << D.N << "," << D.nrho << "," << D.ne << "," << D.m << ","
    << D.m1 << "," << D.nw << "," << D.floating << "," << D.fv0 << "," << D.n_desired << "," << D.n_tw << ","
    << D.n_surf << "," << D.maxdepth << "," << D.n_items << "," << D.ve_exact_ok << "," << D.n_des_inputs << "," << D.n_des_late << "," << D.law_mask << "," << D.maxd << "," << D.crb_full_body << "," << D.nfs << "," << D.ms << "," << D.m1s << "," << D.wsize << "," << D.h_static << "," << D.k_tab << "," << D.k_classes << ",";
  put_ints(o, D.k_class, 16); o << ",";
  put_ints(o, D.it, IT_COUNT); o << ",";
  put_ints(o, D.dt, DT_COUNT); o << ",";
  put_ints(o, D.lds, L_COUNT); o << ",";
  o << D.lds_doubles << "," << D.n_int << "," << D.n_dbl << "," << D.dfs << "," << D.presolve << "," << D.pre_nf << "," << D.pre_m << "," << D.pre_c0[0] << "," << D.pre_c0[1] << "," << D.pre_c0[2] << "," << D.pre_c0[3] << "," << D.pre_b0 << "," << D.pre_fold << ",";
  put_ints(o, D.pre_wd, 6); o << ",";
  o  << D.max_as_iters << "," << D.max_refine << ","
    << D.max_pdas << "," << dbl(D.eps_primal) << "," << dbl(D.tau_pivot) << "," << dbl(D.tol_bound) << ","
    << dbl(D.tol_accept) << "," << dbl(D.h_static_max) << "," << dbl(D.h_static_min) << "," << dbl(D.k_mu2) << "};\n";
}

int qpc_jit_tick_waves(bool f32, bool pack2) {
  // fp64: 256 registers (two waves); two fp64 QPs per wave need two 20 KB slabs: four workgroups per CU, one wave per SIMD with 512
  // registers.  fp32: three waves (see below); packed: two 10 KB slabs per workgroup, eight per CU = two waves per SIMD
  if (pack2) return f32 ? 2 : 1;
  return f32 ? QPC_F32_WAVES_PER_SIMD : 2;
}

std::string qpc_jit_source(const QpcDims &D, const QpcDims &DS, bool warm, bool f32, bool pack2) {
  std::ostringstream o;
  o << "typedef int int32_t;\ntypedef unsigned int uint32_t;\ntypedef long long int64_t;\ntypedef unsigned long long uint64_t;\n";
  if (f32) o << "#define QPC_F32 1\n#define QPC_DTAB_F32 1\n"; // (the runtime hands the fp32 kernel a float copy of the double tables)
  if (pack2) o << "#define QPC_PACK2 1\n#define QPC_NO_MFMA 1\n";
  o << "#define QPC_JIT 1\n#define QPC_M1_MAX " << (D.m1 > 0 ? D.m1 : 1) << "\n#define QPC_MAXD " << D.maxd << "\n#define QPC_NTILE_MAX " << (((D.m + 15) / 16) * (((D.m + 15) / 16) + 1) / 2 > 0 ? ((D.m + 15) / 16) * (((D.m + 15) / 16) + 1) / 2 : 1) << "\n#define QPC_XTILE_MAX " << (((D.m1 + 15) / 16) * (((D.m1 + 15) / 16) + 1) / 2 + (D.m1 + 15) / 16 > 0 ? ((D.m1 + 15) / 16) * (((D.m1 + 15) / 16) + 1) / 2 + (D.m1 + 15) / 16 : 1)
    << "\n#include \"qpc_device.h\"\n";
  const char *st = getenv("QPC_JIT_STAMPS"); // diagnostic: in-kernel section timers
  const bool stamps = st && st[0] == '1';
  if (stamps)
    o << "#define QPC_STAMPS 1\n"
         "#define QPC_PROF_T0() do { *PT = clock64(); } while (0)\n"
         "#define QPC_PROF_ACC(i) do { unsigned long long n__ = clock64(); PA[i] += n__ - *PT; *PT = n__; } while (0)\n";
  if (D.N >= 1 && D.N <= 8) { // friction-cone basis directions (contacts.jl:16-23) as literals
    o << "#define QPC_BASIS_TABLE {";
    for (int i = 0; i < 8; i++) {
      const double a = i * (6.283185307179586 / D.N);
      o << dbl(i < D.N ? std::cos(a) : 1.0) << "," << dbl(i < D.N ? std::sin(a) : 0.0) << (i < 7 ? "," : "");
    }
    o << "}\n";
  }
  o << "#include \"qpc_tick.h\"\n";
  if (!f32) o << "#include \"qpc_sim.h\"\n"; // (the simulation step is fp64 code)
  // field order == struct QpcDims (qpc_program.h)
  o << "static_assert(sizeof(QpcDims) == " << sizeof(QpcDims) << ", \"QpcDims layout differs between host and JIT source\");\n";
  put_dims(o, "kD", D);
  put_dims(o, "kDS", DS);
  o << "struct QpcBatchArgs { long long B; const double *q, *v, *desired, *cw, *cmax, *tw, *surf; double *tau, *vdot, "
       "*cwrench; int32_t *status, *iters; unsigned long long *prof; unsigned long long *ticket; unsigned long long *wsets; const int32_t *order; };\n";
  // the tick kernel (cold start) and, on request, its warm-started twin: same loop, qpc_tick<P, true>
  // experiments (tools/ + DESIGN 8.1): QPC_EXP_DYN_LDS=1 declares the slab as DYNAMIC shared memory (the compiler then does not
  // derive an occupancy bound from it), QPC_EXP_WAVES=n is the second __launch_bounds__ argument (waves per SIMD to allocate for)
  const bool dyn_lds = qpc_jit_dynamic_lds();
  // waves per SIMD the register allocation aims at: the fp64 tick needs its 256 registers (two waves).  The fp32 tick's slab is half
  // the size (16 fit a CU), but measured on MI355X (config 5, 32 768 QPs): 2 waves / 256 registers 36.7 M solves/s, 3 waves / 168
  // registers (272 B of scratch per lane) 37.1 M, 4 waves / 128 registers (368 B of scratch) 33.9 M -- three it is
  int waves = qpc_jit_tick_waves(f32, pack2);
  if (const char *e = QPC_EXP_ENV("QPC_EXP_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 8) waves = v; }
  if (dyn_lds) o << "extern __shared__ __attribute__((aligned(16))) double qpc_lds_dyn[];\n";
  o << "#ifdef QPC_KERNARG_RELOAD\n"
       "template <class T> __device__ __forceinline__ T qpc_karg(const __attribute__((address_space(4))) T *p) { asm volatile(\"\" : \"+s\"(p)); return *p; }\n"
       "#define KA(f) qpc_karg(&Ak->f)\n"
       "#else\n"
       "#define KA(f) A.f\n"
       "#endif\n";
  for (int variant = 0; variant < (warm ? 2 : 1); variant++) {
    o << "extern \"C\" __global__ void __launch_bounds__(64, " << waves << ") " << (variant ? "qpc_tick_kernel_jit_warm" : "qpc_tick_kernel_jit")
      << "(const int *__restrict__ IT, const " << (f32 ? "float" : "double") << " *__restrict__ DT, const QpcBatchArgs A) {\n";
    if (dyn_lds && !f32) o << "  double *const qpc_lds_raw = qpc_lds_dyn;\n";
    else o << "  __shared__ __attribute__((aligned(16))) " << (f32 ? "float" : "double") << " qpc_lds_raw[" << (pack2 ? 2 : 1) * (D.lds_doubles > 0 ? D.lds_doubles : 1) << "];\n";
    // (fp32: `real` is a class holding one float -- qpc_real.h -- so the slab of floats is addressed as an array of reals)
    o <<
         "  real *const qpc_lds = (real *)qpc_lds_raw" << (pack2 ? " + (threadIdx.x >> 5) * " + std::to_string(D.lds_doubles) : std::string()) << "; // (packed: one slab per half-wave)\n"
         "  constexpr int kW = " << (pack2 ? 32 : 64) << ", kQ = " << (pack2 ? 2 : 1) << "; // lanes per QP, QPs per wavefront\n"
         "  typedef real *LdsP;\n"
         "  qpc::qpc_tick_prologue<LdsP>(kD, kDS, IT, qpc_lds);\n"
         "  while (true) {\n"
         "#ifdef QPC_KERNARG_RELOAD\n"
         "    // the launch arguments are re-read from the kernarg segment (scalar loads, scalar cache) where they are used: kept in\n"
         "    // scalar registers across the persistent loop the 16-dword block is spilled to VGPR lanes and restored with 16\n"
         "    // v_readlane per use (96 VALU instructions in the input phase alone)\n"
         "    const __attribute__((address_space(4))) QpcBatchArgs *Ak = (const __attribute__((address_space(4))) QpcBatchArgs *)\n"
         "        ((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + 16);\n"
         "    asm volatile(\"\" : \"+s\"(Ak));\n"
         "#endif\n"
         "    unsigned long long t = 0;\n"
         "    if ((threadIdx.x & (kW - 1)) == 0) t = atomicAdd(KA(ticket), 1ull);\n"
         "    const long long tk = (long long)__shfl(t, threadIdx.x & (64 - kW), 64);\n"
         "    if (tk >= KA(B)) {\n"
         "      // a launch draws exactly B + kQ gridDim.x tickets; whoever draws the last one re-arms the counter\n"
         "      if ((threadIdx.x & (kW - 1)) == 0 && tk == KA(B) + (long long)kQ * gridDim.x - 1) atomicExch(KA(ticket), 0ull);\n"
         "      break;\n"
         "    }\n"
         "    // ticket -> QP: in batch order, or through the caller's cost order (expensive QPs first: qpc_ticket_order_device)\n"
         "    const long long b = KA(order) ? (long long)KA(order)[tk] : tk;\n"
         "    qpc::TickIO io;\n"
         "    io.q = KA(q) + b * kD.nq; io.v = KA(v) + b * kD.nv; io.desired = KA(desired) + b * kD.n_desired;\n"
         "    io.cw = KA(cw) + b * kD.nc; io.cmax = KA(cmax) + b * kD.nc; io.tw = KA(tw) + b * kD.n_tw; io.surf = KA(surf) + b * kD.n_surf;\n"
         "    io.tau = KA(tau) + b * kD.nv; io.vdot = KA(vdot) + b * kD.nv;\n"
         "    io.cwrench = KA(cwrench) ? KA(cwrench) + b * 6 * kD.nc : nullptr;\n"
         "    io.status = KA(status) + b; io.iters = KA(iters) ? KA(iters) + b : nullptr; io.prof = KA(prof) ? KA(prof) + b * PS_COUNT : nullptr;\n"
         "    io.wset = " << (variant ? "KA(wsets) ? KA(wsets) + 2 * b : nullptr" : "nullptr") << ";\n"
         "    // opaque copies of the table pointers: keeps the compiler from hoisting dozens of per-lane table\n"
         "    // addresses out of this loop (they would stay live across the whole tick and spill).  The copies are made in\n"
         "    // the GLOBAL address space and cast back, so that address-space inference still sees where they point: table\n"
         "    // reads stay global_load (vmcnt only) instead of flat_load, which also counts on lgkmcnt and serialises with LDS\n"
         "    const __attribute__((address_space(1))) int *ITg = (const __attribute__((address_space(1))) int *)IT;\n"
         "    const __attribute__((address_space(1))) " << (f32 ? "float" : "double") << " *DTg = (const __attribute__((address_space(1))) " << (f32 ? "float" : "double") << " *)DT;\n"
         "    asm volatile(\"\" : \"+s\"(ITg), \"+s\"(DTg));\n"
         "    const int *ITq = (const int *)ITg; const " << (f32 ? "float" : "double") << " *DTq = (const " << (f32 ? "float" : "double") << " *)DTg;\n"
         "    qpc::qpc_tick<LdsP, " << (variant ? "true" : "false") << ">(kD, kDS, ITq, qpc_dtab{DTq}, qpc_lds, io);\n"
         "    __syncthreads();\n"

         "  }\n"
         "}\n";
  }
  if (f32) return o.str();
  // f-2: the simulation step specialised on the same dims (qpc_sim.h)
  const int rpw = qpc_jit_sim_rpw(D);
  const int slab = qpc_sim_layout(D).total + (qpc_sim_layout(D).total & 1);
  o << "struct QpcSimArgs { long long B; double *q, *v; const double *tau; const int32_t *status; int32_t *bad; qpc_sim_params sp; };\n";
  o << "extern \"C\" __global__ void __launch_bounds__(64, 2) qpc_sim_kernel_jit(const int *__restrict__ IT, "
       "const double *__restrict__ DT, const QpcSimArgs A) {\n"
       "  __shared__ __attribute__((aligned(16))) double qpc_lds[" << rpw * slab << "];\n"
       "  for (long long b0 = (long long)blockIdx.x * " << rpw << "; b0 < A.B; b0 += (long long)gridDim.x * " << rpw << ") {\n"
       "    qpc::SimIO io[" << rpw << "];\n"
       "    for (int r = 0; r < " << rpw << "; r++) {\n"
       "      const long long b = b0 + r < A.B ? b0 + r : A.B - 1; // an odd batch: the last robot is stepped by both halves (same result twice)\n"
       "      io[r].q = A.q + b * kD.nq; io[r].v = A.v + b * kD.nv; io[r].tau = A.tau + b * kD.nv;\n"
       "      io[r].status = A.status ? A.status + b : nullptr;\n"
       "      io[r].q_out = A.q + b * kD.nq; io[r].v_out = A.v + b * kD.nv;\n"
       "      io[r].bad = (A.bad && b0 + r < A.B) ? A.bad + b : nullptr;\n"
       "    }\n"
       "    const __attribute__((address_space(1))) int *ITg = (const __attribute__((address_space(1))) int *)IT;\n"
       "    const __attribute__((address_space(1))) double *DTg = (const __attribute__((address_space(1))) double *)DT;\n"
       "    asm volatile(\"\" : \"+s\"(ITg), \"+s\"(DTg));\n"
       "    const int *ITq = (const int *)ITg; const double *DTq = (const double *)DTg;\n"
       "    qpc::qpc_sim_step<double *, " << rpw << ">(kD, ITq, DTq, qpc_lds, io, A.sp, " << slab << ");\n"
       "    __syncthreads();\n"
       "  }\n"
       "}\n";
  return o.str();
}

// code objects of this process by (translation unit, architecture, options): controllers with the same task program (a test-suite,
// one controller per device thread, ...) compile once.  QPC_JIT_CACHE=0 disables it.
extern "C" const char *qpc_kernel_fingerprint(void);
static std::mutex g_jit_mu;
static std::map<std::string, std::vector<char>> g_jit_cache;

void qpc_jit_forget(const std::string &disk_path) {
  if (!disk_path.empty()) (void)remove(disk_path.c_str());
  std::lock_guard<std::mutex> lk(g_jit_mu);
  g_jit_cache.clear();
}

int qpc_jit_compile(const QpcDims &D, const QpcDims &DS, const std::string &arch, bool warm, bool f32, bool pack2, std::vector<char> &code, std::string &log,
                    std::string *disk_hit) {
  if (disk_hit) disk_hit->clear();
  std::string src = qpc_jit_source(D, DS, warm, f32, pack2);
  std::string key;
  {
    const char *ce = getenv("QPC_JIT_CACHE"), *xe = QPC_EXP_ENV("QPC_JIT_EXTRA_FLAGS");
    if (!(ce && ce[0] == '0')) {
      const char *oe_ = QPC_EXP_ENV("QPC_JIT_OPT"), *se_ = QPC_EXP_ENV("QPC_JIT_SCHED");
      key = arch + "\n" + (xe ? xe : "") + "\n" + (oe_ ? oe_ : "") + "\n" + (se_ ? se_ : "") + "\n" + src;
      std::lock_guard<std::mutex> lk(g_jit_mu);
      auto it = g_jit_cache.find(key);
      if (it != g_jit_cache.end()) { code = it->second; log.clear(); return 0; }
    }
  }
  // on-disk cache (opt-in: QPC_JIT_CACHE_DIR=<directory>): qpc_finalize spends ~5 s in hiprtc per distinct controller; a process that
  // comes back with the same controller on the same library reads the code object instead.  The file name carries the fingerprint of the
  // embedded device headers and a hash of everything else that goes into the compile (specialised source, architecture, options).
  std::string disk_path;
  if (const char *dir = getenv("QPC_JIT_CACHE_DIR")) {
    if (dir[0] && !key.empty()) {
      // (the compiler's version is part of the name: a code object of another ROCm is another file -- ADVICE r4)
      int vmaj = 0, vmin = 0;
      (void)hiprtcVersion(&vmaj, &vmin);
      const std::string vkey = "hiprtc " + std::to_string(vmaj) + "." + std::to_string(vmin) + "\n" + key;
      unsigned long long h = 1469598103934665603ull;
      for (unsigned char ch : vkey) { h ^= ch; h *= 1099511628211ull; }
      char name[96];
      snprintf(name, sizeof name, "/qpc_%s_%016llx.co", qpc_kernel_fingerprint(), h);
      disk_path = std::string(dir) + name;
      if (FILE *f = fopen(disk_path.c_str(), "rb")) {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        std::vector<char> buf(sz > 0 ? (size_t)sz : 0);
        bool ok = sz > 64 && fread(buf.data(), 1, (size_t)sz, f) == (size_t)sz;
        fclose(f);
        ok = ok && buf[0] == 0x7f && buf[1] == 'E' && buf[2] == 'L' && buf[3] == 'F'; // (a code object is an ELF file; anything else is recompiled)
        if (ok) {
          code.swap(buf);
          log.clear();
          if (disk_hit) *disk_hit = disk_path;
          std::lock_guard<std::mutex> lk(g_jit_mu);
          if (g_jit_cache.size() >= 64) g_jit_cache.clear();
          g_jit_cache[key] = code;
          return 0;
        }
      }
    }
  }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "qpc_tick_jit.hip", qpc_src_count, qpc_src_texts, qpc_src_names) != HIPRTC_SUCCESS) {
    log = "hiprtcCreateProgram failed";
    return -1;
  }
  std::string archopt = "--offload-arch=" + arch;
  std::vector<std::string> extra; // QPC_JIT_EXTRA_FLAGS="-mllvm -foo ...": compiler experiments without a rebuild
  if (const char *e = QPC_EXP_ENV("QPC_JIT_EXTRA_FLAGS")) {
    std::istringstream is(e);
    for (std::string tok; is >> tok;) extra.push_back(tok);
  }
  // -O2 + the iterative ILP scheduler: measured on MI355X against -O3 and the default scheduler (25.65 M solves/s):
  // iterative-ilp 26.4-26.5 M, with -O2 26.7 M; max-ilp 25.5, iterative-maxocc 25.4, iterative-minreg 23.2, -ffp-contract=on 24.8.
  // The phases are long straight-line blocks of dependent LDS -> FMA chains: a scheduler that looks for ILP inside them
  // pays.  Both kernels of the module pass the whole GPU test-suite with it.  QPC_JIT_EXTRA_FLAGS is appended and wins.
  // (experiments: QPC_JIT_OPT replaces -O2, QPC_JIT_SCHED the scheduler strategy, "default" = none given)
  const char *oe = QPC_EXP_ENV("QPC_JIT_OPT"), *se = QPC_EXP_ENV("QPC_JIT_SCHED");
  const std::string optlevel = oe && oe[0] ? oe : "-O2";
  const std::string sched = std::string("-amdgpu-sched-strategy=") + (se && se[0] ? se : "iterative-ilp");
  std::vector<const char *> opts = {archopt.c_str(), optlevel.c_str(), "-std=c++17"};
  if (!(se && std::string(se) == "default")) { opts.push_back("-mllvm"); opts.push_back(sched.c_str()); }
  for (auto &x : extra) opts.push_back(x.c_str());
  hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  size_t ls = 0;
  hiprtcGetProgramLogSize(prog, &ls);
  log.assign(ls, ' ');
  if (ls) hiprtcGetProgramLog(prog, &log[0]);
  if (r != HIPRTC_SUCCESS) { hiprtcDestroyProgram(&prog); return -2; }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code.resize(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (!key.empty()) {
    std::lock_guard<std::mutex> lk(g_jit_mu);
    if (g_jit_cache.size() >= 64) g_jit_cache.clear(); // (bounded: a code object is ~1 MB)
    g_jit_cache[key] = code;
  }
  if (!disk_path.empty()) { // written under another name and renamed: a reader never sees half a file
    const std::string tmp = disk_path + ".tmp" + std::to_string((long long)getpid());
    if (FILE *f = fopen(tmp.c_str(), "wb")) {
      const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
      fclose(f);
      if (!ok || rename(tmp.c_str(), disk_path.c_str()) != 0) remove(tmp.c_str());
    }
  }
  return 0;
}

// identity of the device code this library was built from (FNV-1a over the embedded device headers, code only): profiles under
// profiles/ record it, and bench.py only quotes a measured-traffic figure whose fingerprint equals the built kernel's
extern "C" const char *qpc_kernel_fingerprint(void) {
  static char buf[24];
  if (!buf[0]) {
    // comments and white space are skipped, so that a reworded comment in a header does not orphan the profiles
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < qpc_src_count; i++) {
      const char *p = qpc_src_texts[i];
      while (*p) {
        if (p[0] == '/' && p[1] == '/') { while (*p && *p != '\n') p++; continue; }
        if (p[0] == '/' && p[1] == '*') { p += 2; while (*p && !(p[0] == '*' && p[1] == '/')) p++; if (*p) p += 2; continue; }
        if (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') { p++; continue; }
        h ^= (unsigned char)*p; h *= 1099511628211ull;
        p++;
      }
    }
    snprintf(buf, sizeof buf, "%016llx", h);
  }
  return buf;
}

// offline check / experiments: compile the specialised kernel of a controller without a device
#include "qpc_compile.h"
extern "C" int32_t qpc_debug_jit_compile(qpc_controller *c, const char *arch, const char *out_path, char *logbuf, int32_t logn) {
  if (!c) return QPC_ERR_INVALID_ARGUMENT;
  if (!c->finalized) { int rc = qpc_compile_program(c); if (rc != QPC_OK) return rc; }
  std::vector<char> code;
  std::string log;
  int rc = qpc_jit_compile(c->prog.dims, c->prog.dims_s, arch ? arch : "gfx950", c->warm_start != 0, c->precision == QPC_PRECISION_F32, c->qps_per_wave == 2, code, log);
  if (logbuf && logn > 0) snprintf(logbuf, logn, "%s", log.c_str());
  if (rc != 0) { qpc_set_error("JIT compile failed: " + log); return QPC_ERR_HIP; }
  if (out_path) { FILE *f = fopen(out_path, "wb"); if (f) { fwrite(code.data(), 1, code.size(), f); fclose(f); } }
  return QPC_OK;
}

// diagnostic: write the specialised translation unit to a file (for offline ISA experiments)
extern "C" int32_t qpc_debug_jit_source(qpc_controller *c, const char *out_path) {
  if (!c || !out_path) return QPC_ERR_INVALID_ARGUMENT;
  if (!c->finalized) { int rc = qpc_compile_program(c); if (rc != QPC_OK) return rc; }
  std::string src = qpc_jit_source(c->prog.dims, c->prog.dims_s, c->warm_start != 0, c->precision == QPC_PRECISION_F32, c->qps_per_wave == 2);
  FILE *f = fopen(out_path, "wb");
  if (!f) return QPC_ERR_INVALID_ARGUMENT;
  fwrite(src.data(), 1, src.size(), f);
  fclose(f);
  return QPC_OK;
}
