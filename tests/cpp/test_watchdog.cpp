// test_watchdog.cpp — the deadline mechanics of csrc/c2d_watchdog.hpp under -fsanitize=thread (tests/test_sanitizers.py).
//
// The operations here are the shape of c2d_dist.hip's watched calls with the blocking RCCL / HIP call replaced by a sleep:
// a helper that fills its job block and returns, a caller that reads the block only when the helper was in time, and — after a
// time-out — a caller that writes its error text, marks the communicator, frees its context and goes on while the helper is
// still blocked and writes its block later.  ThreadSanitizer reports any access the two sides share without ordering.
#include <atomic>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "c2d_watchdog.hpp"

using c2d::watchdog::Job;
using c2d::watchdog::run_job;
using c2d::watchdog::run_with_deadline;

namespace {

struct FakeCtx { std::string last_error; };                       // written by the CALLER only, as c2d_ctx::last_error
struct FakeComm { int handle = 42; bool timed_out = false; };     // c2d_dist: `timed_out` is the caller's, `handle` is read by helpers
struct InitJob : Job { int comm = 0; };

std::atomic<int> helpers_finished{0};

void nap(int ms) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); }

int fails = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); fails++; } \
    } while (0)

// c2d_dist_barrier's shape
int watched_barrier(FakeCtx* ctx, FakeComm* comm, int block_ms, double timeout_s, bool fail)
{
    if (comm->timed_out) { ctx->last_error = "abandoned"; return -6; }
    Job res;
#ifdef WATCHDOG_TEST_PLANT_RACE
    // what the header forbids (and what round 3's barrier did): the helper writes the caller's ctx when it finally returns.
    // tests/test_sanitizers.py builds this variant to see that the sanitizer in use WOULD report the defect.
    const bool in_time = run_job<Job>([ctx, comm, block_ms, fail](Job& job) {
        const int h = comm->handle;
        nap(block_ms);
        ctx->last_error = "written by the helper";
#else
    const bool in_time = run_job<Job>([comm, block_ms, fail](Job& job) {
        const int h = comm->handle;            // the helper reads the communicator it was given (kept alive after a time-out)
        nap(block_ms);                         // ncclAllReduce + hipStreamSynchronize
#endif
        if (fail || h != 42) { job.st = -6; job.error = "ncclAllReduce failed: unhandled system error"; }
        helpers_finished++;
    }, timeout_s, &res);
    if (!in_time) {
        comm->timed_out = true;                // the caller's word; no helper reads it
        ctx->last_error = "barrier did not complete in time";
        return -6;
    }
    if (res.st != 0 && !res.error.empty()) ctx->last_error = res.error;
    return res.st;
}

// dist_init_impl's shape: the communicator comes back inside the job block
int watched_init(FakeCtx* ctx, int block_ms, double timeout_s, int* comm_out)
{
    InitJob res;
    const bool in_time = run_job<InitJob>([block_ms](InitJob& job) {
        nap(block_ms);                         // ncclCommInitRank
        job.comm = 7;
        helpers_finished++;
    }, timeout_s, &res);
    if (!in_time) { ctx->last_error = "ncclCommInitRank did not complete in time"; return -6; }
    *comm_out = res.comm;
    return res.st;
}

}  // namespace

int main()
{
    // 1. in-time completion, many times: results and error texts cross from the helper to the caller
    {
        FakeCtx ctx;
        FakeComm comm;
        for (int i = 0; i < 300; i++) {
            const bool fail = i % 3 == 0;
            const int st = watched_barrier(&ctx, &comm, 0, 5.0, fail);
            CHECK(st == (fail ? -6 : 0));
            if (fail) CHECK(ctx.last_error == "ncclAllReduce failed: unhandled system error");
            int c = 0;
            CHECK(watched_init(&ctx, 0, 5.0, &c) == 0 && c == 7);
        }
        CHECK(!comm.timed_out);
    }
    // 2. a time-out whose helper finishes later: the caller reports and goes on, the helper writes its own block afterwards
    {
        FakeCtx ctx;
        FakeComm* comm = new FakeComm();       // (left allocated after a time-out, as c2d_dist_destroy leaves a timed-out c2d_dist)
        const int before = helpers_finished.load();
        const int st = watched_barrier(&ctx, comm, 300, 0.03, false);
        CHECK(st == -6 && comm->timed_out && ctx.last_error == "barrier did not complete in time");
        CHECK(watched_barrier(&ctx, comm, 0, 1.0, false) == -6 && ctx.last_error == "abandoned");   // refuses further use
        CHECK(helpers_finished.load() == before);                                                    // still blocked
        int c = 0;
        CHECK(watched_init(&ctx, 300, 0.03, &c) == -6 && c == 0);
        for (int i = 0; i < 100 && helpers_finished.load() < before + 2; i++) nap(20);
        CHECK(helpers_finished.load() == before + 2);                                                // both finished on their own
    }
    // 3. the caller tears its state down right after the time-out (the drivers report and _exit; a library user may destroy the
    //    ctx): the helper must touch nothing of it when it finally returns
    {
        const int before = helpers_finished.load();
        std::vector<std::thread> callers;
        for (int t = 0; t < 4; t++)
            callers.emplace_back([t] {
                FakeCtx* ctx = new FakeCtx();
                FakeComm* comm = new FakeComm();
                const int st = watched_barrier(ctx, comm, 150 + 20 * t, 0.02, t % 2 == 0);
                if (st != -6 || !comm->timed_out) { std::printf("FAILED: caller %d\n", t); std::fflush(stdout); }
                ctx->last_error = "reported";
                delete ctx;                    // gone while the helper is still inside its call
                // (comm stays: c2d_dist_destroy returns early for a timed-out communicator)
            });
        for (auto& c : callers) c.join();
        for (int i = 0; i < 100 && helpers_finished.load() < before + 4; i++) nap(20);
        CHECK(helpers_finished.load() == before + 4);
    }
    // 4. the bare primitive with a deadline shorter than thread start-up, and one that is generous
    {
        std::atomic<int> ran{0};
        for (int i = 0; i < 50; i++) (void)run_with_deadline([&ran] { ran++; }, 1e-6);   // may or may not be in time: no hang, no race
        for (int i = 0; i < 100 && ran.load() < 50; i++) nap(10);
        CHECK(ran.load() == 50);
        CHECK(run_with_deadline([] { nap(5); }, 5.0));
    }
    if (fails) return 1;
    std::printf("watchdog ok\n");
    return 0;
}
