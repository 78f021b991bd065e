// lsap.cpp -- rectangular linear sum assignment + thresholded matching (HOST integer logic).
//
// The reference calls scipy.optimize.linear_sum_assignment (SciPy 1.15.3, requirements.txt:8)
// at src/tracker/core/linear_assignment.py:62.  SciPy is not part of /root/reference; its
// algorithm is restated here from the published method (D. F. Crouse, "On implementing 2D
// rectangular assignment algorithms", IEEE T-AES 2016 -- the shortest-augmenting-path variant
// SciPy documents): dual variables u/v, one Dijkstra-like augmentation per row, tall matrices
// solved transposed.  Which optimum is returned when several exist is decided by three
// details that are kept exactly, because track ids depend on them:
//   (1) unscanned columns are visited in a list initialised in DESCENDING column order and
//       a scanned column is removed by moving the list's last entry into its place;
//   (2) among equal reduced path costs the LAST visited unassigned column wins, otherwise the
//       FIRST visited column;
//   (3) the dual update and the back-tracking along `pred` follow the augmentation order.
// Pinned against SciPy itself in tests/test_host_logic.py::test_lsap_matches_scipy (random, tied, quantised, constant
// and partly infeasible matrices) and against the reference through tests/golden/assign.npz
// (test_min_cost_matching_matches_reference_fixtures); tools/asan_host.sh runs both under ASan/UBSan.
#include <cmath>
#include <limits>
#include <numeric>
#include <algorithm>

#include "assoc_host.hpp"

namespace aic {

namespace {

struct SolverBufs {       // one set per thread: a solve allocates nothing once the vectors have grown
    std::vector<double> u, v, dist;
    std::vector<int> pred, col_of_row, row_of_col, todo;
    std::vector<char> row_seen, col_seen;
};

struct Solver {
    int nr, nc;            // nr <= nc
    const double* c;       // row-major nr x nc
    std::vector<double> &u, &v, &dist;
    std::vector<int> &pred, &col_of_row, &row_of_col, &todo;
    std::vector<char> &row_seen, &col_seen;

    static SolverBufs& bufs() { static thread_local SolverBufs b; return b; }
    Solver(int r, int cc, const double* cost)
        : nr(r), nc(cc), c(cost), u(bufs().u), v(bufs().v), dist(bufs().dist), pred(bufs().pred), col_of_row(bufs().col_of_row),
          row_of_col(bufs().row_of_col), todo(bufs().todo), row_seen(bufs().row_seen), col_seen(bufs().col_seen) {
        u.assign(r, 0.0), v.assign(cc, 0.0), dist.assign(cc, 0.0), pred.assign(cc, -1), col_of_row.assign(r, -1);
        row_of_col.assign(cc, -1), todo.assign(cc, 0), row_seen.assign(r, 0), col_seen.assign(cc, 0);
    }

    // Grow the alternating tree from free row `root` until it reaches an unassigned column.
    int augment_from(int root, double& reached) {
        const double inf = std::numeric_limits<double>::infinity();
        double base = 0.0;
        int live = nc;
        for (int k = 0; k < nc; ++k) todo[k] = nc - 1 - k;   // detail (1)
        std::fill(row_seen.begin(), row_seen.end(), 0);
        std::fill(col_seen.begin(), col_seen.end(), 0);
        std::fill(dist.begin(), dist.end(), inf);
        int i = root, sink = -1;
        while (sink < 0) {
            row_seen[i] = 1;
            int pick = -1;
            double best = inf;
            const double* row = c + (size_t)i * nc;
            for (int k = 0; k < live; ++k) {
                const int j = todo[k];
                const double red = base + row[j] - u[i] - v[j];
                if (red < dist[j]) {
                    dist[j] = red;
                    pred[j] = i;
                }
                // detail (2)
                if (dist[j] < best || (dist[j] == best && row_of_col[j] < 0)) {
                    best = dist[j];
                    pick = k;
                }
            }
            base = best;
            if (base == inf) return -1;   // no feasible completion
            const int j = todo[pick];
            if (row_of_col[j] < 0) sink = j; else i = row_of_col[j];
            col_seen[j] = 1;
            todo[pick] = todo[--live];
        }
        reached = base;
        return sink;
    }

    bool run() {
        for (int r = 0; r < nr; ++r) {
            double m = 0.0;
            const int sink = augment_from(r, m);
            if (sink < 0) return false;
            u[r] += m;
            for (int i = 0; i < nr; ++i)
                if (row_seen[i] && i != r) u[i] += m - dist[col_of_row[i]];
            for (int j = 0; j < nc; ++j)
                if (col_seen[j]) v[j] -= m - dist[j];
            int j = sink;
            for (;;) {   // flip the path back to the root
                const int i = pred[j];
                row_of_col[j] = i;
                std::swap(col_of_row[i], j);
                if (i == r) break;
            }
        }
        return true;
    }
};

}  // namespace

// Returns 0 on success, -1 invalid entries (NaN / -inf), -2 infeasible.
int lsap_solve(const double* cost, int nr, int nc, int64_t* rows, int64_t* cols) {
    if (nr == 0 || nc == 0) return 0;
    for (size_t k = 0, n = (size_t)nr * nc; k < n; ++k)
        if (cost[k] != cost[k] || cost[k] == -std::numeric_limits<double>::infinity()) return -1;
    if (nc >= nr) {
        Solver s(nr, nc, cost);
        if (!s.run()) return -2;
        for (int i = 0; i < nr; ++i) rows[i] = i, cols[i] = s.col_of_row[i];
        return 0;
    }
    // tall: solve the transpose, then report pairs ordered by original row
    std::vector<double> t((size_t)nr * nc);
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) t[(size_t)j * nr + i] = cost[(size_t)i * nc + j];
    Solver s(nc, nr, t.data());
    if (!s.run()) return -2;
    std::vector<int> order(nc);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return s.col_of_row[a] < s.col_of_row[b]; });
    for (int k = 0; k < nc; ++k) rows[k] = s.col_of_row[order[k]], cols[k] = order[k];
    return 0;
}

// linear_assignment.py:55-88 on a dense fp32 block. The clamp value and both comparisons are
// done in fp32 exactly as NumPy does them on a float32 array with a Python-float threshold.
void min_cost_matching(const float* cost, int nr, int nc, double max_distance_f64, std::vector<int>& mrow,
                       std::vector<int>& mcol) {
    mrow.clear();
    mcol.clear();
    if (nr == 0 || nc == 0) return;
    const float max_distance = (float)max_distance_f64;        // weak Python float vs float32 array
    const float clamp = (float)(max_distance_f64 + 1e-5);      // fp64 sum stored into a float32 array
    // (scratch kept per thread: the cascade calls this once per level, 10-25 times per frame, on blocks of a few hundred entries)
    static thread_local std::vector<double> c;
    static thread_local std::vector<float> cf;
    static thread_local std::vector<int64_t> ri, ci;
    c.resize((size_t)nr * nc);
    cf.resize((size_t)nr * nc);
    bool admissible = false, odd = false;
    for (size_t k = 0, tot = (size_t)nr * nc; k < tot; ++k) {
        float x = cost[k];
        if (x > max_distance) x = clamp;
        else if (x <= max_distance) admissible = true;
        else odd = true;                                       // NaN: the solver below reports it
        if (x == -std::numeric_limits<float>::infinity()) odd = true;
        cf[k] = x;
        c[k] = (double)x;
    }
    // Every entry above the threshold: whatever assignment the solver returned, linear_assignment.py:76 would keep none of its
    // pairs (each costs `clamp` > max_distance).  The answer is the empty matching; the solve is skipped.
    if (!admissible && !odd) return;
    // Unique optimum read off the rows (nr <= nc, the cascade's usual shape): if every row that has an entry <= max_distance has a
    // STRICT row minimum and those minima sit in pairwise different columns, the sum of the row minima is attained (rows without an
    // admissible entry cost `clamp` in any column, and there are columns enough), so EVERY optimal assignment -- SciPy's included --
    // gives each such row exactly that column, and :76 keeps exactly those pairs, in row order.  A tie anywhere: the solver decides.
    if (!odd && nr <= nc) {
        static thread_local std::vector<int> arg, owner;
        arg.assign(nr, -1);
        owner.assign(nc, -1);
        bool unique = true;
        for (int r = 0; r < nr && unique; ++r) {
            const float* row = cf.data() + (size_t)r * nc;
            float best = row[0];
            int at = 0, ties = 0;
            for (int k = 1; k < nc; ++k) {
                if (row[k] < best) { best = row[k]; at = k; ties = 0; }
                else if (row[k] == best) ++ties;
            }
            if (best > max_distance) continue;               // no admissible entry: this row ends up unmatched either way
            if (ties || owner[at] >= 0) unique = false;
            else { owner[at] = r; arg[r] = at; }
        }
        if (unique) {
            for (int r = 0; r < nr; ++r)
                if (arg[r] >= 0) { mrow.push_back(r); mcol.push_back(arg[r]); }
            return;
        }
    }
    const int n = std::min(nr, nc);
    ri.resize(n), ci.resize(n);
    const int rc = lsap_solve(c.data(), nr, nc, ri.data(), ci.data());
    AIC_REQUIRE(rc == 0, AIC_ERR_INVALID, "cost matrix contains NaN/-inf or is infeasible");
    for (int k = 0; k < n; ++k) {
        if (cf[(size_t)ri[k] * nc + ci[k]] <= max_distance) {
            mrow.push_back((int)ri[k]);
            mcol.push_back((int)ci[k]);
        }
    }
}

}  // namespace aic

using namespace aic;

extern "C" {

int aic_lsap(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind) {
    return guarded([&] {
        AIC_REQUIRE(nr >= 0 && nc >= 0, AIC_ERR_INVALID, "negative shape");
        AIC_REQUIRE((nr == 0 || nc == 0) || (cost && row_ind && col_ind), AIC_ERR_INVALID, "NULL argument");
        const int rc = lsap_solve(cost, nr, nc, row_ind, col_ind);
        AIC_REQUIRE(rc != -1, AIC_ERR_INVALID, "matrix contains invalid numeric entries");
        AIC_REQUIRE(rc != -2, AIC_ERR_INVALID, "cost matrix is infeasible");
    });
}

int aic_min_cost_matching(const float* cost, int nr, int nc, double max_distance, int32_t* match_row,
                          int32_t* match_col, int32_t* n_match) {
    return guarded([&] {
        AIC_REQUIRE(nr >= 0 && nc >= 0 && n_match, AIC_ERR_INVALID, "bad argument");
        std::vector<int> r, c;
        min_cost_matching(cost, nr, nc, max_distance, r, c);
        *n_match = (int32_t)r.size();
        for (size_t k = 0; k < r.size(); ++k) match_row[k] = r[k], match_col[k] = c[k];
    });
}

}  // extern "C"
