// host_pool.h — a few parked host threads for the per-chain scalar work of the KPM set-up (Arnoldi bounds,
// Hessenberg eigenvalues, Chebyshev coefficients; kpm_host.cpp).  setup!(P) runs before every force evaluation
// (HMC.jl:817-845), so with 64 chains resident the cost of STARTING 16 threads each time (~0.3 ms) was as
// large as the work they did; parked threads are woken in ~20-40 us.  The calling thread works too.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

class ElphHostPool {
  public:
    explicit ElphHostPool(int nworkers) {
        for (int t = 0; t < nworkers; ++t) thr_.emplace_back([this]() { worker(); });
    }
    ~ElphHostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_go_.notify_all();
        for (auto &t : thr_) t.join();
    }
    int workers() const { return (int)thr_.size(); }

    // fn(i) for i = 0..nitems-1, items claimed one at a time; returns when all are done.  Not re-entrant.
    void run(int nitems, const std::function<void(int)> &fn) {
        if (nitems <= 0) return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            nitems_ = nitems;
            next_.store(0, std::memory_order_relaxed);
            pending_ = (int)thr_.size();
            ++gen_;
        }
        cv_go_.notify_all();
        for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < nitems;) fn(i);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }

  private:
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *fn;
            int n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_go_.wait(lk, [&]() { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
                n = nitems_;
            }
            for (int i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n;) (*fn)(i);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) cv_done_.notify_one();
            }
        }
    }
    std::vector<std::thread> thr_;
    std::mutex mu_;
    std::condition_variable cv_go_, cv_done_;
    const std::function<void(int)> *fn_ = nullptr;
    int nitems_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    std::atomic<int> next_{0};
    bool stop_ = false;
};
