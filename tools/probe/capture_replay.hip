// Replays a capture trace recorded by hip_capture_trace.cpp (B / K / M / R / W / E lines with stream and event handles) with dummy
// kernels on streams and events of its own, single-threaded, inside a forked child; `ddmin` shrinks a trace whose replay kills the
// child to a (1-)minimal one.  Built to find what in the captured two-chain train step makes hipStreamEndCapture crash.
//   capture_replay replay <trace>         exit status of the child
//   capture_replay ddmin <trace> <out> [seconds]
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/capture_replay.hip -o tools/probe/bin/capture_replay
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include <atomic>
#include <thread>

__global__ void bump(float* p) { p[threadIdx.x] += 1.f; }

struct Op { char k; std::string a, b; int th; };      // th: 0 = the thread that began the capture, 1 = any other (the autograd engine's)

static std::vector<Op> load(const char* path) {
    std::vector<Op> ops;
    std::ifstream f(path);
    std::string line;
    std::string main_tid;
    while (std::getline(f, line)) {
        std::istringstream is(line);
        std::vector<std::string> w;
        for (std::string t; is >> t;) w.push_back(t);
        if (w.empty()) continue;
        std::string tid;
        for (size_t i = 0; i + 1 < w.size(); ++i)
            if (w[i] == "tid") tid = w[i + 1];
        const std::string& k = w[0];
        if (k == "B") main_tid = tid;
        const int th = (tid.empty() || tid == main_tid) ? 0 : 1;
        if (k == "B" || k == "E") ops.push_back({k[0], w[1], "", 0});
        else if (k == "K" || k == "M") ops.push_back({'K', w[1], "", th});
        else if (k == "R" || k == "W") ops.push_back({k[0], w[1], w[2], th});      // R event stream / W stream event
        else if (k == "C" || k == "D") ops.push_back({k[0], w[1], "", th});        // event created / destroyed (same handle = the allocator reused the block)
    }
    return ops;
}

// runs in the child: 0 ok, 2 = EndCapture returned an error, 10 = another HIP error
static bool threaded = true;
static int replay(const std::vector<Op>& ops, bool verbose) {
    std::map<std::string, hipStream_t> st;
    std::map<std::string, hipEvent_t> ev;
    float* buf = nullptr;
    if (hipMalloc(&buf, 4096) != hipSuccess) return 10;
    auto S = [&](const std::string& h) {
        auto it = st.find(h);
        if (it != st.end()) return it->second;
        hipStream_t s;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) _exit(10);
        st[h] = s;
        return s;
    };
    auto E = [&](const std::string& h) {
        auto it = ev.find(h);
        if (it != ev.end()) return it->second;
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) _exit(10);
        ev[h] = e;
        return e;
    };
    hipStream_t origin = nullptr;
    // the ops of thread 1 run on a second host thread, in trace order (a turn counter hands over)
    std::atomic<size_t> turn{0};
    std::atomic<int> failed{0};
    auto exec = [&](const Op& o) -> int {
        hipError_t e = hipSuccess;
        switch (o.k) {
            case 'B': origin = S(o.a); e = hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal); break;
            case 'K': hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, S(o.a), buf); e = hipGetLastError(); break;
            case 'C': {
                hipEvent_t n;
                e = hipEventCreateWithFlags(&n, hipEventDisableTiming);
                ev[o.a] = n;                                  // (an event created before the trace began keeps its lazily made stand-in)
                break;
            }
            case 'D': {
                auto it = ev.find(o.a);
                if (it != ev.end()) { e = hipEventDestroy(it->second); ev.erase(it); }
                break;
            }
            case 'R': e = hipEventRecord(E(o.a), S(o.b)); break;
            case 'W': e = hipStreamWaitEvent(S(o.a), E(o.b), 0); break;
            case 'E': {
                hipGraph_t g = nullptr;
                e = hipStreamEndCapture(origin, &g);
                if (verbose) printf("EndCapture -> %s\n", hipGetErrorString(e));
                return e == hipSuccess ? 100 : 2;
            }
        }
        if (e != hipSuccess) {
            if (verbose) printf("op %c %s %s -> %s\n", o.k, o.a.c_str(), o.b.c_str(), hipGetErrorString(e));
            return 10;
        }
        return 0;
    };
    int result = 0;
    auto worker = [&](int me) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if ((threaded ? ops[i].th : 0) != me) continue;
            while (turn.load(std::memory_order_acquire) != i)
                if (failed.load()) return;
            const int rc = exec(ops[i]);
            if (rc) { result = rc == 100 ? 0 : rc; failed.store(1); turn.store(ops.size()); return; }
            turn.store(i + 1, std::memory_order_release);
        }
    };
    if (threaded) {
        std::thread t(worker, 1);
        worker(0);
        t.join();
    } else {
        worker(0);
    }
    return result;
}

static int run_child(const std::vector<Op>& ops, bool verbose) {        // >= 0 exit code, < 0 -signal
    fflush(stdout);
    pid_t pid = fork();
    if (pid == 0) _exit(replay(ops, verbose));
    int stt = 0;
    waitpid(pid, &stt, 0);
    return WIFSIGNALED(stt) ? -WTERMSIG(stt) : WEXITSTATUS(stt);
}

static void dump(const std::vector<Op>& ops, const char* path) {
    std::map<std::string, std::string> sn, en;
    auto nm = [](std::map<std::string, std::string>& m, const std::string& h, const char* pre) {
        auto it = m.find(h);
        if (it == m.end()) it = m.emplace(h, pre + std::to_string(m.size())).first;
        return it->second;
    };
    FILE* f = fopen(path, "w");
    for (const Op& o : ops) {
        if (o.k == 'R') fprintf(f, "R %s %s tid %d\n", nm(en, o.a, "e").c_str(), nm(sn, o.b, "s").c_str(), o.th);
        else if (o.k == 'W') fprintf(f, "W %s %s tid %d\n", nm(sn, o.a, "s").c_str(), nm(en, o.b, "e").c_str(), o.th);
        else if (o.k == 'C' || o.k == 'D') fprintf(f, "%c %s tid %d\n", o.k, nm(en, o.a, "e").c_str(), o.th);
        else fprintf(f, "%c %s tid %d\n", o.k, nm(sn, o.a, "s").c_str(), o.th);
    }
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    std::vector<Op> ops = load(argv[2]);
    printf("%zu ops\n", ops.size());
    if (std::string(argv[1]) == "replay") {
        for (int th = 0; th < 2; ++th) {
            threaded = th;
            const int rc = run_child(ops, true);
            printf("%s replay: child %s %d\n", th ? "two-thread" : "single-thread", rc < 0 ? "killed by signal" : "exit code", rc < 0 ? -rc : rc);
        }
        return 0;
    }
    const double budget = argc > 4 ? atof(argv[4]) : 400.0;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    // 1. runs of launches on one stream with nothing in between are one node as far as dependencies go
    {
        std::vector<Op> c;
        for (const Op& o : ops) {
            if (o.k == 'K' && !c.empty() && c.back().k == 'K' && c.back().a == o.a && c.back().th == o.th) continue;
            c.push_back(o);
        }
        if (run_child(c, false) < 0) { ops = c; printf("launch runs merged: %zu ops, still crashes\n", ops.size()); }
        else printf("merging launch runs hides the crash: keeping all %zu ops\n", ops.size());
    }
    if (run_child(ops, false) >= 0) { printf("the replay does not crash: nothing to minimise (thread-dependent?)\n"); return 0; }
    // 2. ddmin over the ops between B and E
    std::vector<Op> body(ops.begin() + 1, ops.end() - 1);
    const Op first = ops.front(), last = ops.back();
    auto test = [&](const std::vector<Op>& b) {
        std::vector<Op> t;
        t.push_back(first);
        t.insert(t.end(), b.begin(), b.end());
        t.push_back(last);
        return run_child(t, false) < 0;
    };
    size_t n = 2;
    int tests = 0;
    while (body.size() >= 2 && elapsed() < budget) {
        const size_t chunk = (body.size() + n - 1) / n;
        bool reduced = false;
        for (size_t i = 0; i < body.size() && elapsed() < budget; i += chunk) {
            std::vector<Op> t(body.begin(), body.begin() + i);
            t.insert(t.end(), body.begin() + std::min(body.size(), i + chunk), body.end());
            ++tests;
            if (!t.empty() && test(t)) {
                body = t;
                n = n > 2 ? n - 1 : 2;
                reduced = true;
                break;
            }
        }
        if (!reduced) {
            if (chunk == 1) break;
            n = std::min(body.size(), n * 2);
        }
    }
    printf("ddmin: %d tests, %.0f s, %zu ops left\n", tests, elapsed(), body.size() + 2);
    std::vector<Op> t;
    t.push_back(first);
    t.insert(t.end(), body.begin(), body.end());
    t.push_back(last);
    dump(t, argv[3]);
    return 0;
}
