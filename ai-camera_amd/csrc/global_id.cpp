// global_id.cpp -- cross-camera global-ID policy of configs[4] (BASELINE.json; the reference lists "smarter gallery management
// in ReID" as future work, README.md:209).  HOST code, HIP-free (built under ASan/UBSan by
// tools/asan_host.sh): integer bookkeeping on top of the nearest-neighbour table the HIP pass (aic_gallery_annotate,
// kernels_trk_dev.hip::gallery_nearest_kernel) computes from the all-gathered gallery shards.
//
// Policy.  A track is known by (rank, track id).  Its global id is the (rank, track id) of the FIRST sighting of the identity: a
// track seen for the first time gets its own key, and when two tracks of different cameras are each other's nearest neighbour
// within the cosine threshold, both adopt the smaller of their two global ids (and so does everything that adopted either
// earlier: union-find with the smaller id as the root).  Per-stream association never reads this table (per-stream rows are
// those of configs[3]).
//
// Determinism across ranks.  Every rank holds the table for ALL ranks' tracks and updates it from the SAME all-gathered bytes:
// the distances are bit-symmetric (d(i, j) == d(j, i): same products, same summation order), ties go to the lowest row, links are
// applied in ascending row order.  No rank-dependent input enters, so after exchange e every rank has the same table without
// any further communication.
#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "assoc_host.hpp"

namespace aic {

struct GidTable {
    int world;
    std::map<uint64_t, uint64_t> first;     // (rank << 32 | track id) -> global id given at its first sighting
    std::map<uint64_t, uint64_t> parent;    // global id -> smaller global id it was merged into
    long links = 0, updates = 0;
    static uint64_t key(int rank, int id) { return ((uint64_t)(uint32_t)rank << 32) | (uint32_t)id; }
    uint64_t find(uint64_t g) {
        uint64_t r = g;
        for (auto it = parent.find(r); it != parent.end(); it = parent.find(r)) r = it->second;
        while (g != r) {                      // path compression
            auto it = parent.find(g);
            const uint64_t nx = it->second;
            it->second = r;
            g = nx;
        }
        return r;
    }
    bool unite(uint64_t a, uint64_t b) {
        a = find(a), b = find(b);
        if (a == b) return false;
        if (a < b) parent[b] = a; else parent[a] = b;
        return true;
    }
};

}  // namespace aic

using namespace aic;

struct aic_gid { GidTable t; };

extern "C" {

int aic_gid_create(int world, aic_gid** out) {
    return guarded([&] {
        AIC_REQUIRE(out && world >= 1 && world <= 4096, AIC_ERR_INVALID, "bad argument");
        *out = new aic_gid();
        (*out)->t.world = world;
    });
}

int aic_gid_destroy(aic_gid* g) {
    return guarded([&] { delete g; });
}

int aic_gid_update(aic_gid* g, int world, int t_max, const int32_t* track_id, const int32_t* near_row, const float* near_dist,
                   double max_cosine_distance, int32_t* n_links) {
    return guarded([&] {
        AIC_REQUIRE(g && track_id && near_row && near_dist && t_max > 0, AIC_ERR_INVALID, "bad argument");
        AIC_REQUIRE(world == g->t.world, AIC_ERR_INVALID, "world size differs from the table's");
        GidTable& t = g->t;
        const int n = world * t_max;
        const float thr = (float)max_cosine_distance;
        for (int i = 0; i < n; ++i) {
            if (track_id[i] < 0) continue;
            const uint64_t k = GidTable::key(i / t_max, track_id[i]);
            t.first.emplace(k, k);             // first sighting: its own (rank, track id)
        }
        int links = 0;
        for (int i = 0; i < n; ++i) {          // mutual nearest neighbours within the threshold, ascending row order
            const int j = near_row[i];
            if (track_id[i] < 0 || j <= i || j >= n || track_id[j] < 0) continue;
            if (near_row[j] != i || !(near_dist[i] <= thr)) continue;
            AIC_REQUIRE(i / t_max != j / t_max, AIC_ERR_INVALID, "nearest-neighbour table links two tracks of one camera");
            if (t.unite(t.first[GidTable::key(i / t_max, track_id[i])], t.first[GidTable::key(j / t_max, track_id[j])])) ++links;
        }
        t.links += links, t.updates += 1;
        if (n_links) *n_links = links;
    });
}

int aic_gid_lookup(aic_gid* g, int rank, int track_id, int64_t* global_id) {
    return guarded([&] {
        AIC_REQUIRE(g && global_id, AIC_ERR_INVALID, "NULL argument");
        auto it = g->t.first.find(GidTable::key(rank, track_id));
        *global_id = it == g->t.first.end() ? -1 : (int64_t)g->t.find(it->second);
    });
}

int aic_gid_size(aic_gid* g, int64_t* n_tracks, int64_t* n_identities, int64_t* n_links) {
    return guarded([&] {
        AIC_REQUIRE(g, AIC_ERR_INVALID, "NULL argument");
        if (n_tracks) *n_tracks = (int64_t)g->t.first.size();
        if (n_identities) *n_identities = (int64_t)g->t.first.size() - (int64_t)g->t.parent.size();
        if (n_links) *n_links = g->t.links;
    });
}

}  // extern "C"
