// assoc_host.cpp -- the age-ordered matching cascade + IoU stage of DeepSORT on precomputed cost matrices (HOST integer logic).
//
// Mirrors src/tracker/core/linear_assignment.py:91-157 (matching_cascade), :160-212 (Mahalanobis gate) and
// src/tracker/core/tracker_core.py:83-177 (_match).  The reference recomputes cost sub-blocks per cascade level; the values
// are those of the full [T,N] matrices the GPU produced once per frame, so this file only selects, thresholds and assigns.
// No HIP type or call in here: tools/asan_host.sh builds it together with lsap.cpp under ASan + UBSan.
#include <algorithm>
#include <utility>
#include <vector>

#include "assoc_host.hpp"

namespace aic {

static const float kInfty = 1e5f;                              // linear_assignment.py:9
static const float kChi2_4 = (float)9.487729036781154;          // kalman_filter.py:16, compared in fp32

void cascade_match(int T, int N, const int* state, const int* tsu, const float* app, const float* maha, const float* iou,
                   double max_cosine_distance, double max_iou_distance, int max_age,
                   std::vector<std::pair<int, int>>& matches, std::vector<int>& unmatched_t, std::vector<int>& unmatched_d) {
    matches.clear();
    unmatched_t.clear();
    unmatched_d.clear();
    std::vector<int> confirmed, tentative;
    for (int i = 0; i < T; ++i) {
        if (state[i] == 2) confirmed.push_back(i);           // TrackState.Confirmed, track.py:10-14
        else if (state[i] == 1) tentative.push_back(i);      // TrackState.Tentative
    }
    for (int j = 0; j < N; ++j) unmatched_d.push_back(j);
    std::vector<char> got(T, 0);
    static thread_local std::vector<float> sub;
    static thread_local std::vector<int> mr, mc, rows, keep, first, next;
    static thread_local std::vector<char> dead;
    // the confirmed tracks of every level, in track order (one pass instead of one scan of the table per level: max_age = 70 levels
    // x 160 tracks was a fifth of this function on a crowded scene)
    first.assign(max_age + 2, -1);
    next.assign(T, -1);
    for (int k = (int)confirmed.size() - 1; k >= 0; --k) {
        const int i = confirmed[k], lv = tsu[i];
        if (lv >= 1 && lv <= max_age) { next[i] = first[lv]; first[lv] = i; }
    }
    // stage 1: cascade over time_since_update = 1 .. max_age, gated appearance cost
    for (int level = 0; level < max_age; ++level) {
        if (unmatched_d.empty()) break;
        rows.clear();
        for (int i = first[level + 1]; i >= 0; i = next[i]) rows.push_back(i);
        if (rows.empty()) continue;
        const int nr = (int)rows.size(), nc = (int)unmatched_d.size();
        sub.resize((size_t)nr * nc);
        for (int r = 0; r < nr; ++r)
            for (int c = 0; c < nc; ++c) {
                const size_t k = (size_t)rows[r] * N + unmatched_d[c];
                sub[(size_t)r * nc + c] = (maha[k] > kChi2_4) ? kInfty : app[k];   // linear_assignment.py:187-210
            }
        min_cost_matching(sub.data(), nr, nc, max_cosine_distance, mr, mc);
        if (mr.empty()) continue;                        // (nothing accepted: the unmatched list stays as it is)
        dead.assign(nc, 0);
        for (size_t k = 0; k < mr.size(); ++k) {
            matches.emplace_back(rows[mr[k]], unmatched_d[mc[k]]);
            got[rows[mr[k]]] = 1;
            dead[mc[k]] = 1;
        }
        keep.clear();
        for (int c = 0; c < nc; ++c)
            if (!dead[c]) keep.push_back(unmatched_d[c]);
        unmatched_d.swap(keep);
    }
    // stage 2: IoU on tentative + confirmed tracks that missed exactly one frame
    std::vector<int> cand = tentative, stale;
    for (int i : confirmed) {
        if (got[i]) continue;
        if (tsu[i] == 1) cand.push_back(i);
        else stale.push_back(i);   // tsu > 1 (tsu == 0 cannot occur after predict(); kept unmatched like the reference)
    }
    // NB reference order: tentative first, then confirmed (tracker_core.py:138-141)
    std::vector<int> un_cand = cand;
    if (!cand.empty() && !unmatched_d.empty()) {
        const int nr = (int)cand.size(), nc = (int)unmatched_d.size();
        sub.resize((size_t)nr * nc);
        for (int r = 0; r < nr; ++r)
            for (int c = 0; c < nc; ++c) sub[(size_t)r * nc + c] = iou[(size_t)cand[r] * N + unmatched_d[c]];
        min_cost_matching(sub.data(), nr, nc, max_iou_distance, mr, mc);
        std::vector<char> dead(nc, 0), rdead(nr, 0);
        for (size_t k = 0; k < mr.size(); ++k) {
            matches.emplace_back(cand[mr[k]], unmatched_d[mc[k]]);
            dead[mc[k]] = 1;
            rdead[mr[k]] = 1;
        }
        std::vector<int> keep;
        for (int c = 0; c < nc; ++c)
            if (!dead[c]) keep.push_back(unmatched_d[c]);
        unmatched_d.swap(keep);
        un_cand.clear();
        for (int r = 0; r < nr; ++r)
            if (!rdead[r]) un_cand.push_back(cand[r]);
    }
    unmatched_t = stale;
    unmatched_t.insert(unmatched_t.end(), un_cand.begin(), un_cand.end());
}

}  // namespace aic

using namespace aic;

extern "C" int aic_match_cascade(const float* app, const float* maha, const float* iou, int t, int n, const int32_t* state,
                                 const int32_t* tsu, double max_cosine_distance, double max_iou_distance, int max_age,
                                 int32_t* match_track, int32_t* match_det, int32_t* n_match, int32_t* unmatched_tracks,
                                 int32_t* n_unmatched_tracks, int32_t* unmatched_dets, int32_t* n_unmatched_dets) {
    return guarded([&] {
        AIC_REQUIRE(t >= 0 && n >= 0 && n_match && n_unmatched_tracks && n_unmatched_dets, AIC_ERR_INVALID, "bad argument");
        AIC_REQUIRE(t == 0 || (state && tsu), AIC_ERR_INVALID, "NULL track arrays");
        AIC_REQUIRE(t == 0 || n == 0 || (app && maha && iou), AIC_ERR_INVALID, "NULL cost matrices");
        std::vector<std::pair<int, int>> m;
        std::vector<int> ut, ud, st(state, state + t), ts(tsu, tsu + t);
        cascade_match(t, n, st.data(), ts.data(), app, maha, iou, max_cosine_distance, max_iou_distance, max_age, m, ut, ud);
        *n_match = (int32_t)m.size(), *n_unmatched_tracks = (int32_t)ut.size(), *n_unmatched_dets = (int32_t)ud.size();
        for (size_t k = 0; k < m.size(); ++k) match_track[k] = m[k].first, match_det[k] = m[k].second;
        for (size_t k = 0; k < ut.size(); ++k) unmatched_tracks[k] = ut[k];
        for (size_t k = 0; k < ud.size(); ++k) unmatched_dets[k] = ud[k];
    });
}
