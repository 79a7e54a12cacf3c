// stream_lifetime_probe.hip — what does this HIP runtime do with a stream handle after hipStreamDestroy?
//
// Round 4's workspace guard asked hipStreamQuery about the stream of the previous counted call; a host segmentation fault
// in c2d_sat_rect_pairs_verts right after a host-batch call (gpurun_out/r4e_pytest.log) pointed at that query on a
// destroyed stream.  This probe runs each case in its own forked child (forked BEFORE anything touches HIP, so every child
// initialises the runtime itself) and reports how the child ended: the return code of the query, or the signal.
// Output kept in profiles/r05_stream_lifetime_probe.txt; the conclusion in profiles/notes_r05_workspace_guard.md.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void spin_kernel(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *out = t - t0;
}

static const char* name_of(hipError_t e) { return hipGetErrorName(e); }

// the cases; each returns the text it wants printed
static int case_query_after_destroy(int n_between, bool work_in_flight, bool sync_api)
{
    hipStream_t s = nullptr;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 8) != hipSuccess) return 90;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 91;
    if (work_in_flight) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, 200000000ull);  // ~0.1 s
    hipError_t e = hipStreamDestroy(s);
    std::printf("    hipStreamDestroy -> %s\n", name_of(e));
    hipStream_t others[64] = {};
    int same = 0;
    for (int i = 0; i < n_between && i < 64; i++) {
        if (hipStreamCreateWithFlags(&others[i], hipStreamNonBlocking) != hipSuccess) return 92;
        if (others[i] == s) same++;
    }
    if (n_between) std::printf("    %d streams created afterwards, %d of them at the destroyed stream's address\n", n_between, same);
    std::fflush(stdout);
    e = sync_api ? hipStreamSynchronize(s) : hipStreamQuery(s);
    std::printf("    %s(destroyed handle) -> %s\n", sync_api ? "hipStreamSynchronize" : "hipStreamQuery", name_of(e));
    (void)hipGetLastError();
    std::fflush(stdout);
    return 0;
}

static int case_query_garbage(int kind)
{
    if (hipFree(nullptr) != hipSuccess) return 90;  // initialise the runtime
    void* junk = nullptr;
    if (kind == 0) {
        junk = std::malloc(4096);
        std::memset(junk, 0x5a, 4096);
    } else if (kind == 1) {
        junk = std::malloc(4096);
        std::free(junk);  // freed heap block, as a destroyed stream's object would be
    } else {
        junk = (void*)(uintptr_t)0x10;  // unmapped
    }
    hipError_t e = hipStreamQuery((hipStream_t)junk);
    std::printf("    hipStreamQuery(%s) -> %s\n", kind == 0 ? "live heap bytes" : kind == 1 ? "freed heap block" : "unmapped address", name_of(e));
    std::fflush(stdout);
    return 0;
}

template <class F>
static void in_child(const char* title, F f)
{
    std::printf("%s\n", title);
    std::fflush(stdout);
    const pid_t pid = fork();
    if (pid == 0) {
        const int rc = f();
        std::fflush(stdout);
        _exit(rc);
    }
    int st = 0;
    waitpid(pid, &st, 0);
    if (WIFSIGNALED(st)) std::printf("    child ended by signal %d (%s)\n", WTERMSIG(st), strsignal(WTERMSIG(st)));
    else std::printf("    child exited with %d\n", WEXITSTATUS(st));
    std::fflush(stdout);
}

int main()
{
    in_child("1. query an idle stream after hipStreamDestroy", [] { return case_query_after_destroy(0, false, false); });
    in_child("2. the same, 64 new streams created in between", [] { return case_query_after_destroy(64, false, false); });
    in_child("3. destroy with 0.1 s of work in flight, then query", [] { return case_query_after_destroy(0, true, false); });
    in_child("4. destroy with work in flight, 64 new streams, then query", [] { return case_query_after_destroy(64, true, false); });
    in_child("5. hipStreamSynchronize on a destroyed handle", [] { return case_query_after_destroy(0, false, true); });
    in_child("6. query a pointer to live heap bytes that never were a stream", [] { return case_query_garbage(0); });
    in_child("7. query a pointer to a freed heap block", [] { return case_query_garbage(1); });
    in_child("8. query an unmapped address", [] { return case_query_garbage(2); });
    return 0;
}
