// TEST / TOOLING ONLY -- a null HIP runtime for dry runs of libopenscene_amd's HOST logic on a machine without a GPU.
// Linked (with -Bsymbolic-functions) into a private copy of the library, it turns every kernel launch, memset, copy
// and event call the library makes into a log line instead of device work, so that the launch SEQUENCE of a code path
// (kernel symbol, grid, block) can be compared between the per-module Python path and the network executor, and the
// host cost of a pass can be measured.  Never part of the product library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <mutex>

namespace {
struct Cfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<Cfg> g_cfg;
std::mutex g_mu;
std::string g_log;
long long g_launches = 0;
int g_logging = 1;
void add(const char* line) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_logging) { g_log += line; g_log += '\n'; }
}
}  // namespace

extern "C" {
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_cfg.push_back(Cfg{grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    Cfg c = g_cfg.back();
    g_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void* fn, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t stream) {
    Dl_info info;
    const char* name = "?";
    if (dladdr(fn, &info) && info.dli_sname) name = info.dli_sname;
    char buf[512];
    snprintf(buf, sizeof(buf), "K %s grid=%u,%u,%u block=%u", name, grid.x, grid.y, grid.z, block.x);
    ++g_launches;
    add(buf);
    snprintf(buf, sizeof(buf), "S %llx", (unsigned long long)reinterpret_cast<uintptr_t>(stream));     // the stream of the line above
    add(buf);
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    char buf[128]; snprintf(buf, sizeof(buf), "MEMSET %zu", n); ++g_launches; add(buf); return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
    char buf[128]; snprintf(buf, sizeof(buf), "MEMCPY %zu", n); ++g_launches; add(buf); return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// events are numbered (their handle IS the number): the log says which event a stream records / waits for
static uintptr_t g_next_event = 0x1000;
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(g_next_event++); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(g_next_event++); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t e, unsigned) {
    char buf[128];
    snprintf(buf, sizeof(buf), "WAIT e%llx s%llx", (unsigned long long)reinterpret_cast<uintptr_t>(e), (unsigned long long)reinterpret_cast<uintptr_t>(st));
    add(buf);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) {
    char buf[128];
    snprintf(buf, sizeof(buf), "EVENT e%llx s%llx", (unsigned long long)reinterpret_cast<uintptr_t>(e), (unsigned long long)reinterpret_cast<uintptr_t>(st));
    add(buf);
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }

// ---- tool interface
const char* osn_dry_log(void) { std::lock_guard<std::mutex> lk(g_mu); static std::string copy; copy = g_log; return copy.c_str(); }
void osn_dry_reset(void) { std::lock_guard<std::mutex> lk(g_mu); g_log.clear(); g_launches = 0; }
void osn_dry_logging(int on) { std::lock_guard<std::mutex> lk(g_mu); g_logging = on; }
long long osn_dry_launches(void) { return g_launches; }
}
