// LD_PRELOAD shim: logs every event record / stream wait / launch / async memset+memcpy issued between hipStreamBeginCapture and
// hipStreamEndCapture (any thread), with stream and event handles, so that the dependency shape a capture recorded can be
// reconstructed offline (tools/probe/capture_trace_report.py).  Built for the round-3 root-cause hunt of the hipStreamEndCapture
// crash.   build: g++ -O2 -fPIC -shared -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/probe/hip_capture_trace.cpp -o tools/probe/bin/libhipcaptrace.so -ldl
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <sys/syscall.h>

static FILE* out() {
    static FILE* f = nullptr;
    if (!f) {
        const char* p = getenv("HIPCAPTRACE_OUT");
        f = p ? fopen(p, "w") : stderr;
        if (!f) f = stderr;
        setvbuf(f, nullptr, _IOLBF, 0);
    }
    return f;
}
static volatile int g_active = 0;
static long tid() { return syscall(SYS_gettid); }
// torch dlopen()s its bundled libamdhip64 (SONAME libamdhip64.so.7) long after this shim was preloaded, so RTLD_NEXT does not see it:
// take the handle of the copy that is already mapped when the first intercepted call arrives
template <class F> static F real(const char* name) {
    static void* h = nullptr;
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    void* p = h ? dlsym(h, name) : dlsym(RTLD_NEXT, name);
    if (!p) { fprintf(stderr, "hipcaptrace: no %s\n", name); abort(); }
    return reinterpret_cast<F>(p);
}
#define LOG(...) do { if (g_active) fprintf(out(), __VA_ARGS__); } while (0)

// HIPCAPTRACE_DEFER_DESTROY=1: events destroyed while a capture is in progress are destroyed after hipStreamEndCapture instead
// (test of the hypothesis that EndCapture touches events the application has already destroyed)
#include <mutex>
#include <vector>
static std::mutex g_mu;
static std::vector<hipEvent_t> g_deferred;
static bool defer_destroy() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("HIPCAPTRACE_DEFER_DESTROY"); v = e && e[0] == '1'; }
    return v;
}

extern "C" {
hipError_t hipEventDestroy(hipEvent_t ev) {
    static auto fn = real<hipError_t (*)(hipEvent_t)>("hipEventDestroy");
    LOG("D %p tid %ld\n", (void*)ev, tid());
    if (g_active && defer_destroy()) {
        std::lock_guard<std::mutex> l(g_mu);
        g_deferred.push_back(ev);
        return hipSuccess;
    }
    return fn(ev);
}
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned flags) {
    static auto fn = real<hipError_t (*)(hipEvent_t*, unsigned)>("hipEventCreateWithFlags");
    hipError_t e = fn(ev, flags);
    LOG("C %p tid %ld flags %u\n", (void*)*ev, tid(), flags);
    return e;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode) {
    static auto fn = real<hipError_t (*)(hipStream_t, hipStreamCaptureMode)>("hipStreamBeginCapture");
    g_active = 1;
    LOG("B %p mode %d tid %ld\n", (void*)s, (int)mode, tid());
    return fn(s, mode);
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
    static auto fn = real<hipError_t (*)(hipStream_t, hipGraph_t*)>("hipStreamEndCapture");
    LOG("E %p tid %ld\n", (void*)s, tid());
    fflush(out());
    hipError_t e = fn(s, g);
    LOG("E-done %d\n", (int)e);
    g_active = 0;
    {
        static auto destroy = real<hipError_t (*)(hipEvent_t)>("hipEventDestroy");
        std::lock_guard<std::mutex> l(g_mu);
        for (hipEvent_t ev : g_deferred) destroy(ev);
        g_deferred.clear();
    }
    return e;
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s) {
    static auto fn = real<hipError_t (*)(hipEvent_t, hipStream_t)>("hipEventRecord");
    LOG("R %p %p tid %ld\n", (void*)ev, (void*)s, tid());
    return fn(ev, s);
}
hipError_t hipEventRecordWithFlags(hipEvent_t ev, hipStream_t s, unsigned flags) {
    static auto fn = real<hipError_t (*)(hipEvent_t, hipStream_t, unsigned)>("hipEventRecordWithFlags");
    LOG("R %p %p tid %ld flags %u\n", (void*)ev, (void*)s, tid(), flags);
    return fn(ev, s, flags);
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t ev, unsigned flags) {
    static auto fn = real<hipError_t (*)(hipStream_t, hipEvent_t, unsigned)>("hipStreamWaitEvent");
    LOG("W %p %p tid %ld\n", (void*)s, (void*)ev, tid());
    return fn(s, ev, flags);
}
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t shm, hipStream_t s) {
    static auto fn = real<hipError_t (*)(const void*, dim3, dim3, void**, size_t, hipStream_t)>("hipLaunchKernel");
    LOG("K %p tid %ld\n", (void*)s, tid());
    return fn(f, g, b, args, shm, s);
}
hipError_t hipExtLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t shm, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int flags) {
    static auto fn = real<hipError_t (*)(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int)>("hipExtLaunchKernel");
    LOG("K %p tid %ld ext\n", (void*)s, tid());
    return fn(f, g, b, args, shm, s, e0, e1, flags);
}
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shm,
                                 hipStream_t s, void** params, void** extra) {
    static auto fn = real<hipError_t (*)(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void**, void**)>(
        "hipModuleLaunchKernel");
    LOG("K %p tid %ld module\n", (void*)s, tid());
    return fn(f, gx, gy, gz, bx, by, bz, shm, s, params, extra);
}
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s) {
    static auto fn = real<hipError_t (*)(void*, int, size_t, hipStream_t)>("hipMemsetAsync");
    LOG("M %p tid %ld memset %zu\n", (void*)s, tid(), n);
    return fn(p, v, n, s);
}
hipError_t hipMemcpyAsync(void* d, const void* src, size_t n, hipMemcpyKind k, hipStream_t s) {
    static auto fn = real<hipError_t (*)(void*, const void*, size_t, hipMemcpyKind, hipStream_t)>("hipMemcpyAsync");
    LOG("M %p tid %ld memcpy %zu kind %d\n", (void*)s, tid(), n, (int)k);
    return fn(d, src, n, k, s);
}
}
