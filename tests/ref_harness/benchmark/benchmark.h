// benchmark.h -- a small google-benchmark-compatible harness: enough for the reference's benchmark/*.cpp
// (benchmark::Fixture, BENCHMARK_F, `for (auto st : state)`, Initialize, RunSpecifiedBenchmarks) to compile
// UNMODIFIED against libhexl-fpga.so in an image without google-benchmark. Each benchmark runs its timed loop for
// --benchmark_min_time seconds (default 0.5) after one untimed iteration and prints time per iteration.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace benchmark {

class State {
public:
    explicit State(double min_s) : min_s_(min_s) {}
    struct Iter {
        State* s; bool end;
        bool operator!=(const Iter&) { return s->keep_running(); }
        void operator++() {}
        int operator*() const { return 0; }
    };
    Iter begin() { start_ = std::chrono::steady_clock::now(); iters_ = 0; return {this, false}; }
    Iter end() { return {this, true}; }
    bool keep_running() {
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count();
        if (iters_ > 0 && el >= min_s_) { elapsed_ = el; return false; }
        ++iters_;
        return true;
    }
    long iterations() const { return iters_; }
    double elapsed() const { return elapsed_; }
    void SetItemsProcessed(long) {}
    void SetBytesProcessed(long) {}
private:
    double min_s_, elapsed_ = 0;
    long iters_ = 0;
    std::chrono::steady_clock::time_point start_;
};

class Fixture {
public:
    virtual ~Fixture() {}
    virtual void SetUp(const State&) {}
    virtual void TearDown(const State&) {}
    virtual void BenchmarkCase(State&) = 0;
};

namespace internal {
struct Entry { std::string name; std::function<Fixture*()> make; };
inline std::vector<Entry>& entries() { static std::vector<Entry> e; return e; }
inline double& min_time() { static double t = 0.5; return t; }
inline std::string& filter() { static std::string f; return f; }
struct Registrar { Registrar(const char* n, std::function<Fixture*()> m) { entries().push_back({n, m}); } };
}  // namespace internal

inline void Initialize(int* argc, char** argv) {
    for (int i = 1; i < *argc; ++i) {
        if (!std::strncmp(argv[i], "--benchmark_min_time=", 21)) internal::min_time() = std::atof(argv[i] + 21);
        if (!std::strncmp(argv[i], "--benchmark_filter=", 19)) internal::filter() = argv[i] + 19;
    }
}
inline size_t RunSpecifiedBenchmarks() {
    std::printf("%-56s %14s %12s\n", "Benchmark", "Time/iter", "Iterations");
    size_t n = 0;
    for (auto& e : internal::entries()) {
        if (!internal::filter().empty() && e.name.find(internal::filter()) == std::string::npos) continue;
        Fixture* f = e.make();
        State st(internal::min_time());
        f->SetUp(st);
        f->BenchmarkCase(st);
        f->TearDown(st);
        delete f;
        std::printf("%-56s %11.3f ms %12ld\n", e.name.c_str(), 1e3 * st.elapsed() / st.iterations(), st.iterations());
        std::fflush(stdout);
        ++n;
    }
    return n;
}
}  // namespace benchmark

#define BENCHMARK_F(fixture, name)                                                                          \
    class fixture##_##name##_Benchmark : public fixture { public: void BenchmarkCase(::benchmark::State&) override; }; \
    static ::benchmark::internal::Registrar bm_reg_##fixture##_##name(#fixture "/" #name,                  \
        []() -> ::benchmark::Fixture* { return new fixture##_##name##_Benchmark(); });                    \
    void fixture##_##name##_Benchmark::BenchmarkCase
