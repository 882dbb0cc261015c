// gtest.h -- a small googletest-compatible harness, just enough of the API for the reference's own test
// sources (intel/hexl-fpga tests/main.cpp, test_fwd_ntt.cpp, test_inv_ntt.cpp, test_dyadic_multiply.cpp) to
// compile UNMODIFIED against this repository's libhexl-fpga.so in an image that has no googletest.
// Supported: TEST / TEST_F, ::testing::Test (SetUp/TearDown), ::testing::Environment +
// AddGlobalTestEnvironment, InitGoogleTest (--gtest_filter=substring), RUN_ALL_TESTS, ASSERT_* / EXPECT_*
// (EQ NE LT LE GT GE TRUE FALSE), EXPECT_DEATH (compiled, reported as skipped).
#pragma once
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace testing {

class Environment {
public:
    virtual ~Environment() {}
    virtual void SetUp() {}
    virtual void TearDown() {}
};

class Test {
public:
    virtual ~Test() {}
    virtual void SetUp() {}
    virtual void TearDown() {}
    virtual void TestBody() = 0;
};

namespace internal {
struct Case { std::string name; std::function<Test*()> make; };
inline std::vector<Case>& cases() { static std::vector<Case> c; return c; }
inline std::vector<Environment*>& envs() { static std::vector<Environment*> e; return e; }
inline std::string& filter() { static std::string f; return f; }
inline bool& failed() { static bool f = false; return f; }
inline bool& fatal() { static bool f = false; return f; }
struct Registrar { Registrar(const char* n, std::function<Test*()> m) { cases().push_back({n, m}); } };
template <class A, class B>
bool report(bool ok, const char* op, const char* ea, const char* eb, const A&, const B&, const char* file, int line) {
    if (!ok) { failed() = true; std::printf("%s:%d: Failure\n  Expected: (%s) %s (%s)\n", file, line, ea, op, eb); }
    return ok;
}
}  // namespace internal

inline void InitGoogleTest(int* argc, char** argv) {
    for (int i = 1; i < *argc; ++i)
        if (!std::strncmp(argv[i], "--gtest_filter=", 15)) internal::filter() = argv[i] + 15;
}
inline Environment* AddGlobalTestEnvironment(Environment* e) { internal::envs().push_back(e); return e; }

inline int RunAllTests() {
    for (auto* e : internal::envs()) e->SetUp();
    int run = 0, bad = 0;
    for (auto& c : internal::cases()) {
        std::string f = internal::filter();
        if (!f.empty() && f != "*") {
            if (f.back() == '*') f.pop_back();
            if (!f.empty() && f.front() == '*') f.erase(0, 1);
            if (c.name.find(f) == std::string::npos) continue;
        }
        std::printf("[ RUN      ] %s\n", c.name.c_str());
        internal::failed() = false; internal::fatal() = false;
        Test* t = c.make();
        t->SetUp();
        if (!internal::fatal()) t->TestBody();
        t->TearDown();
        delete t;
        ++run;
        if (internal::failed()) { ++bad; std::printf("[  FAILED  ] %s\n", c.name.c_str()); }
        else std::printf("[       OK ] %s\n", c.name.c_str());
        std::fflush(stdout);
    }
    for (auto it = internal::envs().rbegin(); it != internal::envs().rend(); ++it) { (*it)->TearDown(); delete *it; }
    std::printf("[==========] %d test(s) ran, %d failed.\n", run, bad);
    std::printf(bad ? "[  FAILED  ]\n" : "[  PASSED  ] %d test(s).\n", run);
    return bad ? 1 : 0;
}
}  // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()

#define GT_CLASS_(suite, name) suite##_##name##_Test
#define GT_TEST_(suite, name, parent)                                                                    \
    class GT_CLASS_(suite, name) : public parent { public: void TestBody() override; };                  \
    static ::testing::internal::Registrar gt_reg_##suite##_##name(#suite "." #name,                      \
        []() -> ::testing::Test* { return new GT_CLASS_(suite, name); });                                \
    void GT_CLASS_(suite, name)::TestBody()
#define TEST(suite, name) GT_TEST_(suite, name, ::testing::Test)
#define TEST_F(fixture, name) GT_TEST_(fixture, name, fixture)

#define GT_CMP_(a, b, op, fatal_)                                                                        \
    do { if (!::testing::internal::report((a) op (b), #op, #a, #b, (a), (b), __FILE__, __LINE__)) {      \
             if (fatal_) { ::testing::internal::fatal() = true; return; } } } while (0)
#define ASSERT_EQ(a, b) GT_CMP_(a, b, ==, true)
#define ASSERT_NE(a, b) GT_CMP_(a, b, !=, true)
#define ASSERT_LT(a, b) GT_CMP_(a, b, <, true)
#define ASSERT_LE(a, b) GT_CMP_(a, b, <=, true)
#define ASSERT_GT(a, b) GT_CMP_(a, b, >, true)
#define ASSERT_GE(a, b) GT_CMP_(a, b, >=, true)
#define EXPECT_EQ(a, b) GT_CMP_(a, b, ==, false)
#define EXPECT_NE(a, b) GT_CMP_(a, b, !=, false)
#define EXPECT_LT(a, b) GT_CMP_(a, b, <, false)
#define EXPECT_LE(a, b) GT_CMP_(a, b, <=, false)
#define EXPECT_GT(a, b) GT_CMP_(a, b, >, false)
#define EXPECT_GE(a, b) GT_CMP_(a, b, >=, false)
#define ASSERT_TRUE(c) GT_CMP_(static_cast<bool>(c), true, ==, true)
#define ASSERT_FALSE(c) GT_CMP_(static_cast<bool>(c), false, ==, true)
#define EXPECT_TRUE(c) GT_CMP_(static_cast<bool>(c), true, ==, false)
#define EXPECT_FALSE(c) GT_CMP_(static_cast<bool>(c), false, ==, false)
#define EXPECT_DEATH(stmt, re) do { std::printf("  (EXPECT_DEATH skipped by the shim: %s)\n", #stmt); } while (0)
#define ASSERT_DEATH(stmt, re) EXPECT_DEATH(stmt, re)
