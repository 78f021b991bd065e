// assoc_host.hpp -- host-only (HIP-free) declarations shared by lsap.cpp, assoc_host.cpp and tracker.cpp: error plumbing of the
// C ABI and the integer association logic.  Everything here compiles with plain g++ (tools/asan_host.sh).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/aicam.h"

namespace aic {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

#define AIC_REQUIRE(cond, code, msg)                                              \
    do {                                                                          \
        if (!(cond)) throw ::aic::Error((code), std::string(msg));                \
    } while (0)

// Wraps the body of every extern "C" entry point.
template <class F>
static inline int guarded(F&& f) {
    try {
        f();
        return AIC_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return AIC_ERR_RUNTIME;
    }
}

// lsap.cpp
int lsap_solve(const double* cost, int nr, int nc, int64_t* rows, int64_t* cols);
void min_cost_matching(const float* cost, int nr, int nc, double max_distance, std::vector<int>& mrow, std::vector<int>& mcol);
// assoc_host.cpp: matches as (track index, detection index); unmatched tracks in the reference's order
void cascade_match(int T, int N, const int* state, const int* tsu, const float* app, const float* maha, const float* iou,
                   double max_cosine_distance, double max_iou_distance, int max_age,
                   std::vector<std::pair<int, int>>& matches, std::vector<int>& unmatched_t, std::vector<int>& unmatched_d);

}  // namespace aic
