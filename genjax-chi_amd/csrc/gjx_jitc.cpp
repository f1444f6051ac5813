// gjx_jitc — the plan compiler's child process (gjx_plan_jit.hpp: compile_to_code).
//
// hiprtc runs the whole AMDGPU backend in the caller's process; a backend crash on a generated kernel (seen once: "LLVM
// ERROR: SmallVector unable to grow" on a kernel with a constant-folded NaN log-weight) is an abort() of that process.
// The library therefore compiles generated kernels HERE: a fresh process that never touches the GPU (hiprtc needs no
// device), started with posix_spawn, source and header in files, the code object back in a file.  If this process dies,
// the caller gets GJX_ERR_JIT and a log line, never an abort.
//
//   gjx_jitc <source file> <device header file> <output code object> <log file> [compiler options...]
// exit status: 0 = code object written, 1 = compilation failed (log written), 2 = usage / IO error.
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static bool slurp(const char* path, std::string* out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  std::fclose(f);
  return true;
}
static bool spill(const char* path, const std::string& s) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return false;
  const bool ok = std::fwrite(s.data(), 1, s.size(), f) == s.size();
  return std::fclose(f) == 0 && ok;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  std::string src, hdr;
  if (!slurp(argv[1], &src) || !slurp(argv[2], &hdr)) return 2;
  hiprtcProgram prog;
  const char* hn[] = {"gjx_device.hpp"};
  const char* hs[] = {hdr.c_str()};
  if (hiprtcCreateProgram(&prog, src.c_str(), "gjx_plan.hip", 1, hs, hn) != HIPRTC_SUCCESS) return 2;
  std::vector<const char*> opts(argv + 5, argv + argc);
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, 0);
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    spill(argv[4], std::string(hiprtcGetErrorString(r)) + "\n" + log);
    return 1;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::string code(cs, 0);
  hiprtcGetCode(prog, &code[0]);
  hiprtcDestroyProgram(&prog);
  return spill(argv[3], code) ? 0 : 2;
}
