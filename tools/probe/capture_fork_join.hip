// What does ending a hipGraph stream capture do with (a) a stream waiting on its OWN event and (b) the fork / cross-wait / join
// shape of TrainStep._generators_two_chains?  Round 2 saw hipStreamEndCapture take the process down when the captured step held
// either; no log was kept.  Every pattern runs in a child process forked BEFORE the parent touches HIP, so a crash in one is
// reported (signal number) and the others still run.
// build: hipcc -O2 --offload-arch=gfx950 tools/probe/capture_fork_join.hip -o tools/probe/bin/capture_fork_join
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); _exit(10); } } while (0)

__global__ void bump(float* p, float v) { p[threadIdx.x] += v; }

struct Ctx {
    hipStream_t s[6];
    float* buf;
    std::vector<hipEvent_t> evs;
    hipEvent_t rec(int i) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); CK(hipEventRecord(e, s[i])); evs.push_back(e); return e; }
    void wait(int i, hipEvent_t e) { CK(hipStreamWaitEvent(s[i], e, 0)); }
    void k(int i, float v = 1.f) { hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s[i], buf + 64 * i, v); }
    void join(int into, int from) { wait(into, rec(from)); }
};

static int run_pattern(int id, hipStreamCaptureMode mode) {
    Ctx c;
    for (auto& s : c.s) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMalloc(&c.buf, 6 * 64 * sizeof(float)));
    CK(hipMemset(c.buf, 0, 6 * 64 * sizeof(float)));
    CK(hipDeviceSynchronize());
    const int M = 0, X = 1, I = 2, S = 3, A = 4, B = 5;      // main, chain A, identity, weight gradients, critic A, critic B
    CK(hipStreamBeginCapture(c.s[M], mode));
    c.k(M);
    switch (id) {
        case 0:                                              // plain fork / join
            c.join(X, M); c.k(X); c.join(M, X);
            break;
        case 1: {                                            // a stream waits on the event it has just recorded (nothing in between)
            hipEvent_t e = c.rec(M);
            c.wait(M, e);
            c.k(M);
            break;
        }
        case 2: {                                            // ... with a kernel between record and wait (a redundant edge, not a duplicate)
            hipEvent_t e = c.rec(M);
            c.k(M);
            c.wait(M, e);
            c.k(M);
            break;
        }
        case 3: {                                            // self-wait on a FORKED stream (the layout "001212" case: chain A and critic B share a stream)
            c.join(X, M);
            c.k(X);
            hipEvent_t e = c.rec(X);
            c.k(X);
            printf("    before the self-wait\n"); fflush(stdout);
            c.wait(X, e);
            printf("    after the self-wait\n"); fflush(stdout);
            c.k(X);
            printf("    after the next kernel\n"); fflush(stdout);
            c.join(M, X);
            printf("    joined\n"); fflush(stdout);
            break;
        }
        case 8: {                                            // the same with nothing between record and wait
            c.join(X, M);
            c.k(X);
            hipEvent_t e = c.rec(X);
            printf("    before the self-wait\n"); fflush(stdout);
            c.wait(X, e);
            printf("    after the self-wait\n"); fflush(stdout);
            c.k(X);
            c.join(M, X);
            printf("    joined\n"); fflush(stdout);
            break;
        }
        case 9: {                                            // forked stream waits on an event of the ORIGIN recorded before the fork (redundant edge across streams)
            hipEvent_t e0 = c.rec(M);
            c.k(M);
            c.join(X, M);
            c.k(X);
            c.wait(X, e0);
            c.k(X);
            c.join(M, X);
            break;
        }
        case 4: {                                            // the two-chain shape, no self-waits
            hipEvent_t ev_in = c.rec(M);
            c.wait(I, ev_in); c.k(I);                        // identity A forward
            hipEvent_t ev_idt_A = c.rec(I);
            c.k(I); c.join(S, I); c.k(S); c.k(I);            // its loss + backward, weight gradients on S
            c.wait(X, ev_in); c.k(X);                        // chain A: A2B pass 1
            hipEvent_t ev_a2b_1 = c.rec(X);
            c.join(B, X); c.k(B);                            // critic B
            hipEvent_t ev_pred_B = c.rec(B);
            c.k(X);
            c.wait(X, ev_idt_A); c.k(X);                     // B2A pass 2
            hipEvent_t ev_b2a_2 = c.rec(X);
            c.wait(X, ev_pred_B); c.k(X);                    // chain A's losses
            c.k(B); c.join(X, B);                            // backward through the critic on its stream, back to X
            c.join(S, X); c.k(S); c.k(X);                    // chain A backward + weight gradients
            c.wait(M, ev_b2a_2); c.k(M);                     // chain B: B2A pass 3
            c.join(A, M); c.k(A);                            // critic A
            hipEvent_t ev_pred_A = c.rec(A);
            c.wait(I, ev_in); c.wait(I, ev_a2b_1); c.k(I);   // identity B
            hipEvent_t ev_idt_B = c.rec(I);
            c.k(I); c.join(S, I); c.k(S);
            c.wait(M, ev_idt_B); c.k(M);                     // A2B pass 3
            c.wait(M, ev_pred_A); c.k(M);                    // chain B's losses, backward
            c.k(A); c.join(M, A);
            c.join(S, M); c.k(S);
            c.join(M, I); c.join(M, X); c.join(M, A); c.join(M, B); c.join(M, S);
            c.k(M);
            break;
        }
        case 5: {                                            // unjoined fork at EndCapture (expected: an error code, not a crash)
            c.join(X, M); c.k(X);
            break;
        }
        case 6: {                                            // the same event object waited on twice by one stream, and by two streams
            c.join(X, M); c.k(X);
            hipEvent_t e = c.rec(X);
            c.wait(M, e); c.wait(M, e); c.k(M);
            c.join(I, M); c.wait(I, e); c.k(I);
            c.join(M, I); c.join(M, X);
            break;
        }
        case 7: {                                            // join twice (two events of the same forked stream, no work in between)
            c.join(X, M); c.k(X);
            c.join(M, X); c.join(M, X);
            c.k(M);
            break;
        }
        case 10: {                                           // an event recorded in the capture is DESTROYED before the capture ends,
            c.join(X, M);                                    // and its memory is handed to something else
            for (int r = 0; r < 64; ++r) {
                c.k(X);
                hipEvent_t e;
                CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                CK(hipEventRecord(e, c.s[X]));
                CK(hipStreamWaitEvent(c.s[I], e, 0));
                c.k(I);
                CK(hipEventDestroy(e));
                void* junk[8];
                for (auto& j : junk) { j = malloc(64 + 16 * (r & 7)); memset(j, 0xff, 64); }
                for (auto& j : junk) if ((reinterpret_cast<size_t>(j) >> 4) & 1) free(j);       // (half of them stay: the freed event's block gets reused)
            }
            c.join(M, X); c.join(M, I);
            break;
        }
        case 11: {                                           // two forked streams wait on each other's events (what the two-chain schedule does)
            hipEvent_t e0 = c.rec(M);
            c.wait(X, e0); c.wait(I, e0);
            c.k(X);
            hipEvent_t e1 = c.rec(X);
            c.wait(I, e1);                                   // I after X
            c.k(I);
            hipEvent_t e2 = c.rec(I);
            c.wait(X, e2);                                   // X after I
            c.k(X);
            c.join(M, X); c.join(M, I);
            break;
        }
        default: break;
    }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(c.s[M], &g);
    printf("    EndCapture -> %s, graph %p\n", hipGetErrorString(e), (void*)g);
    fflush(stdout);
    if (e != hipSuccess) return 2;
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    hipGraphExec_t ex;
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ex, c.s[M]));
    CK(hipStreamSynchronize(c.s[M]));
    float h[6 * 64];
    CK(hipMemcpy(h, c.buf, sizeof(h), hipMemcpyDeviceToHost));
    printf("    nodes %zu; after 3 launches buf[stream] =", nn);
    for (int i = 0; i < 6; ++i) printf(" %g", h[64 * i]);
    printf("\n");
    fflush(stdout);
    return 0;
}

int main() {
    const char* names[] = {"fork/join", "self-wait right after record", "self-wait after a kernel", "self-wait on a forked stream",
                           "two-chain shape (no self-waits)", "unjoined fork", "one event waited on repeatedly", "double join",
                           "self-wait on a forked stream, right after record", "forked stream waits on an older event of the origin",
                           "events recorded in the capture and destroyed before it ends", "two forked streams wait on each other"};
    for (int mode = 1; mode < 2; ++mode)        // (Global and ThreadLocal behaved identically: profiles/r03_capture_probe.log)
        for (int id = 0; id < 12; ++id) {
            printf("[%s] pattern %d: %s\n", mode ? "ThreadLocal" : "Global", id, names[id]);
            fflush(stdout);
            pid_t pid = fork();
            if (pid == 0) _exit(run_pattern(id, mode ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeGlobal));
            int st = 0;
            waitpid(pid, &st, 0);
            if (WIFSIGNALED(st)) printf("    ==> child killed by signal %d\n", WTERMSIG(st));
            else printf("    ==> exit code %d\n", WEXITSTATUS(st));
            fflush(stdout);
        }
    return 0;
}
