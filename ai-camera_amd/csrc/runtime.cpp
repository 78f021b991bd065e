// runtime.cpp -- device contexts, error state, HIP-event profiling, library-level C ABI.
#include "common.hpp"

#include <algorithm>

#include <cstdlib>

namespace aic {

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }

int device_count() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return 0;
    return n;
}

static std::mutex g_dev_mu;
static std::vector<std::unique_ptr<Device>> g_devs;

Device& device(int id) {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    int n = device_count();
    AIC_REQUIRE(n > 0, AIC_ERR_NO_DEVICE,
                "no HIP device visible: libaicam.so has no CPU path (gfx950 / MI355X required)");
    AIC_REQUIRE(id >= 0 && id < n, AIC_ERR_INVALID, "device id out of range");
    if ((int)g_devs.size() < n) g_devs.resize(n);
    if (!g_devs[id]) {
        auto d = std::make_unique<Device>();
        d->id = id;
        HIP_CHECK(hipSetDevice(id));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, id));
        AIC_REQUIRE(std::strncmp(prop.gcnArchName, "gfx950", 6) == 0, AIC_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
        d->n_cu = prop.multiProcessorCount;
        HIP_CHECK(hipStreamCreateWithFlags(&d->s_main, hipStreamNonBlocking));
        {   // the tracker chain is latency-critical: its small launches should win CU slots as soon as they free up
            int lo = 0, hi = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            const bool prio = getenv("AICAM_NO_TRK_PRIO") == nullptr;
            HIP_CHECK(hipStreamCreateWithPriority(&d->s_trk, hipStreamNonBlocking, prio ? hi : lo));
        }
        HIP_CHECK(hipStreamCreateWithFlags(&d->s_det, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&d->s_reid, hipStreamNonBlocking));
        g_devs[id] = std::move(d);
    }
    HIP_CHECK(hipSetDevice(id));
    return *g_devs[id];
}

void Device::prof_begin(int cls, hipStream_t s, double fl, double by) {
    std::lock_guard<std::mutex> lk(prof_mu);
    Pair p;
    if (!pool.empty()) {
        p = pool.back();
        pool.pop_back();
    } else {
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
    }
    (void)hipEventRecord(p.a, s);
    pending[cls].push_back(p);
    if (fl != 0 || by != 0 || cls != PROF_CONV) {   // conv launches are accounted one by one (prof_account)
        launches[cls] += 1;
        flops[cls] += fl;
        bytes[cls] += by;
    }
}

void Device::prof_account(int cls, double fl, double by) {
    std::lock_guard<std::mutex> lk(prof_mu);
    launches[cls] += 1;
    flops[cls] += fl;
    bytes[cls] += by;
}

void Device::prof_end(int cls, hipStream_t s) {
    std::lock_guard<std::mutex> lk(prof_mu);
    if (pending[cls].empty()) return;
    (void)hipEventRecord(pending[cls].back().b, s);
}

void Device::prof_collect() {
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(prof_mu);
    std::vector<std::pair<double, double>> iv;
    for (int c = 0; c < AIC_PROF_CLASSES; ++c) {
        iv.clear();
        bool stamps = prof_ref != nullptr;
        const bool any = !pending[c].empty();
        for (auto& p : pending[c]) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) ms[c] += t;
            float ta = 0.f, tb = 0.f;
            if (stamps && hipEventElapsedTime(&ta, prof_ref, p.a) == hipSuccess && hipEventElapsedTime(&tb, prof_ref, p.b) == hipSuccess) iv.emplace_back(ta, tb);
            else stamps = false;
            pool.push_back(p);
        }
        pending[c].clear();
        (void)hipGetLastError();
        if (!stamps) { if (any) ms_union[c] = -1e30; continue; }   // (stays negative: reported as unavailable)
        // intervals of different collects cannot overlap (each collect follows a device synchronisation): union per collect, summed
        std::sort(iv.begin(), iv.end());
        double lo = 0.0, hi = -1.0;
        for (auto& q : iv) {
            if (hi < lo || q.first > hi) { if (hi >= lo) ms_union[c] += hi - lo; lo = q.first, hi = q.second; }
            else if (q.second > hi) hi = q.second;
        }
        if (hi >= lo) ms_union[c] += hi - lo;
    }
    // the stamps are fp32 milliseconds since prof_ref: re-record it after every collect (the device is idle here), so that their
    // resolution does not decay with the time since the last reset (0.5 us at 4 s, 30 us after five minutes)
    if (prof_ref) { (void)hipEventRecord(prof_ref, s_main); (void)hipEventSynchronize(prof_ref); }
}

void Device::prof_reset() {
    prof_collect();
    for (int c = 0; c < AIC_PROF_CLASSES; ++c) ms[c] = 0, ms_union[c] = 0, launches[c] = 0, flops[c] = 0, bytes[c] = 0;
    if (!prof_ref && hipEventCreate(&prof_ref) != hipSuccess) prof_ref = nullptr;
    if (prof_ref) { (void)hipEventRecord(prof_ref, s_main); (void)hipEventSynchronize(prof_ref); }
}

}  // namespace aic

using namespace aic;

extern "C" {

const char* aic_last_error(void) { return g_last_error.c_str(); }
int aic_abi_version(void) { return AIC_ABI_VERSION; }

int aic_device_count(int* count) {
    return guarded([&] {
        AIC_REQUIRE(count, AIC_ERR_INVALID, "count is NULL");
        *count = device_count();
    });
}

int aic_device_sync(int dev) {
    return guarded([&] {
        device(dev);
        HIP_CHECK(hipDeviceSynchronize());
    });
}

int aic_prof_enable(int dev, int on) {
    return guarded([&] {
        Device& d = device(dev);
        if (!on && d.prof_mask) d.prof_collect();
        d.prof_mask = (unsigned)on;   // bit mask of classes; 0 = off, -1 = all
    });
}

int aic_prof_reset(int dev) {
    return guarded([&] { device(dev).prof_reset(); });
}

int aic_prof_read_union(int dev, int cls, double* ms_union) {
    return guarded([&] {
        AIC_REQUIRE(cls >= 0 && cls < AIC_PROF_CLASSES && ms_union, AIC_ERR_INVALID, "bad argument");
        Device& d = device(dev);
        d.prof_collect();
        *ms_union = d.ms_union[cls] < 0 ? -1.0 : d.ms_union[cls];
    });
}

int aic_prof_read(int dev, int cls, double* ms, int64_t* launches, double* flops, double* bytes) {
    return guarded([&] {
        AIC_REQUIRE(cls >= 0 && cls < AIC_PROF_CLASSES, AIC_ERR_INVALID, "profile class out of range");
        Device& d = device(dev);
        d.prof_collect();
        if (ms) *ms = d.ms[cls];
        if (launches) *launches = d.launches[cls];
        if (flops) *flops = d.flops[cls];
        if (bytes) *bytes = d.bytes[cls];
    });
}

}  // extern "C"
