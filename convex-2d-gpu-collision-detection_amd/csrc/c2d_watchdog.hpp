// c2d_watchdog.hpp — blocking calls under a deadline (the watchdog of c2d_dist.hip).
//
// ncclCommInitRank and every collective block until all ranks have arrived; a rank whose peers never show up, or die after
// the link was built, must get an error instead of hanging for ever.  The blocking call therefore runs on a helper thread
// and the caller waits for it with a deadline.  After a time-out the helper is STILL inside the call — it cannot be
// cancelled — so the rule of this header is: the helper owns everything it touches.  Its result travels in a block that
// caller and helper co-own (shared_ptr), the caller looks at that block only when the helper finished in time, and whatever
// else the operation needs it captures by value.  The caller may then report, destroy its context and leave while the helper
// is still running.
//
// Plain C++17, no HIP, no RCCL: tests/cpp/test_watchdog.cpp runs it under -fsanitize=thread (tests/test_sanitizers.py) —
// in-time completion, a time-out whose helper finishes later, and the caller tearing its state down right after a time-out.
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>

namespace c2d {
namespace watchdog {

// Runs `fn` on a helper thread and waits for it for at most timeout_s seconds.  false: the deadline passed; the thread
// is detached and stays wherever it blocks, so everything it touches must be kept alive by `fn` itself.
template <class F>
bool run_with_deadline(F fn, double timeout_s)
{
    struct Shared { std::mutex m; std::condition_variable cv; bool done = false; };
    auto sh = std::make_shared<Shared>();
    std::thread t([sh, fn]() mutable {
        fn();
        { std::lock_guard<std::mutex> lk(sh->m); sh->done = true; }
        sh->cv.notify_all();
    });
    std::unique_lock<std::mutex> lk(sh->m);
    const bool ok = sh->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return sh->done; });
    lk.unlock();
    if (ok) t.join();
    else t.detach();
    return ok;
}

// result of a watched operation; operations with more to hand back derive from it
struct Job {
    int st = 0;          // a C2D_* status
    std::string error;   // text for c2d_last_error, "" if none
};

// Runs op(J&) under the deadline on a job block that caller and helper co-own.  In time: true, and *out is the job as the
// operation left it (the helper has been joined: no concurrent access).  Late: false, *out is untouched, and the helper
// keeps the only remaining reference to its block, which it may write whenever the blocked call returns.
template <class J, class Op>
bool run_job(Op op, double timeout_s, J* out)
{
    auto job = std::make_shared<J>();
    const bool in_time = run_with_deadline([job, op]() mutable { op(*job); }, timeout_s);
    if (in_time) *out = std::move(*job);
    return in_time;
}

// timeout_s <= 0 of the entry points: $C2D_DIST_TIMEOUT_S, or 300 s
inline double default_timeout_s()
{
    const char* t = std::getenv("C2D_DIST_TIMEOUT_S");
    const double v = t ? std::atof(t) : 0.0;
    return v > 0.0 ? v : 300.0;
}

}  // namespace watchdog
}  // namespace c2d
