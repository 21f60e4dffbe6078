// CPU check of the host layer's device-object cache (adapter/mp2p_hip_host.hpp, Runtime::global_layer / local_layer /
// evict / release_layer / invalidate_layers) against a MOCK of the few C-ABI calls it makes: every handle the mock hands
// out must be freed at most once and never used after its free, the cache never holds more than its bounds, a layer that
// is seen again is not uploaded again, an edited one is.  No GPU, no libmp2p_hip: the mock below IS the library here.
#include "mp2p_hip_host.hpp"
#include <cstdio>
#include <random>
#include <set>
using namespace mp2p_hip_host;

static std::set<void*> g_live;
static size_t          g_next = 16, g_double_free = 0, g_uploads = 0;
static void*           fresh()
{
    void* h = (void*)(g_next += 16);
    g_live.insert(h);
    g_uploads++;
    return h;
}
static void drop(void* h)
{
    if (!g_live.erase(h)) g_double_free++;
}
extern "C" {
const char* mp2p_hip_last_error(const mp2p_hip_ctx*) { return "mock"; }
int mp2p_hip_map_upload(mp2p_hip_ctx*, const float*, const float*, const float*, size_t, const mp2p_hip_map_params*, mp2p_hip_map** out)
{
    *out = (mp2p_hip_map*)fresh();
    return 0;
}
void mp2p_hip_map_free(mp2p_hip_ctx*, mp2p_hip_map* m) { drop(m); }
int  mp2p_hip_map_get_info(mp2p_hip_ctx*, const mp2p_hip_map*, mp2p_hip_map_info* info)
{
    memset(info, 0, sizeof(*info));
    info->device_bytes = 1000000;
    return 0;
}
int mp2p_hip_cloud_upload(mp2p_hip_ctx*, const float*, const float*, const float*, size_t, mp2p_hip_cloud** out)
{
    *out = (mp2p_hip_cloud*)fresh();
    return 0;
}
void mp2p_hip_cloud_free(mp2p_hip_ctx*, mp2p_hip_cloud* c) { drop(c); }
}

struct Buf
{
    std::vector<float> x, y, z;
    explicit Buf(size_t n, unsigned seed) : x(n), y(n), z(n)
    {
        std::mt19937 r(seed);
        std::uniform_real_distribution<float> u(-50.f, 50.f);
        for (size_t i = 0; i < n; i++) x[i] = u(r), y[i] = u(r), z[i] = u(r);
    }
};

int main()
{
    int     bad = 0;
    Runtime rt;
    rt.ctx = (mp2p_hip_ctx*)0x10;
    auto expect = [&](bool c, const char* what) {
        if (!c) printf("FAILED: %s\n", what), bad++;
    };
    const size_t     N = 5000;
    std::vector<Buf> bufs;
    for (unsigned k = 0; k < 12; k++) bufs.emplace_back(N, 100 + k);
    auto cloud = [&](int k, bool full) { return rt.local_layer(&bufs[k], bufs[k].x.data(), bufs[k].y.data(), bufs[k].z.data(), N, full); };
    auto map   = [&](int k, bool full) { return rt.global_layer(&bufs[k], bufs[k].x.data(), bufs[k].y.data(), bufs[k].z.data(), N, full); };

    // ---- a layer seen again is not uploaded again; the second iteration-0 visit is the strided re-check
    rt.max_layers = 3;
    map(0, true), cloud(1, true);
    expect(g_uploads == 2, "two uploads for two new layers");
    map(0, false), cloud(1, false);
    expect(g_uploads == 2, "sampled check: nothing uploaded");
    size_t full0 = rt.n_full_checks, re0 = rt.n_reseen_checks;
    map(0, true), cloud(1, true);
    expect(g_uploads == 2 && rt.n_full_checks == full0 + 2 && rt.n_reseen_checks == re0, "re-seen layers (default): hashed in full again, no upload");
    // a ONE-point edit off every sample and stride is found at the next iteration 0 (the live layer decides)
    bufs[1].y[1238] += 0.25f;
    cloud(1, true);
    expect(g_uploads == 3, "default: a single edited point is found at iteration 0");
    // a host that vouches for its layers (trust_reseen): the strided re-check, no full hash
    rt.trust_reseen = true;
    map(0, true), cloud(1, true);  // (the stride's print of the re-uploaded layer is taken with this full check)
    full0 = rt.n_full_checks, re0 = rt.n_reseen_checks;
    map(0, true), cloud(1, true);
    expect(g_uploads == 3 && rt.n_full_checks == full0 && rt.n_reseen_checks == re0 + 2, "re-seen layers (trusted): strided re-check, no full hash, no upload");
    // ---- an in-place edit of 61 consecutive points (unseen by the 1024-point sample) is found at iteration 0 only
    for (size_t i = 1237; i < 1237 + 61; i++) bufs[1].x[i] += 50.f;
    void* before = (void*)rt.local_layer(&bufs[1], bufs[1].x.data(), bufs[1].y.data(), bufs[1].z.data(), N, false);
    (void)before;  // (whether the 1024-point sample of a later iteration sees an interior run depends on the layer's size)
    const size_t up_before = g_uploads;
    void* after = cloud(1, true);
    expect(g_uploads >= up_before && g_uploads == 4, "the edit is found at iteration 0 at the latest: exactly one re-upload");
    expect(g_live.count(after) == 1 && g_live.size() == 2, "the old copy was freed, the new one is live");
    // ---- bounded: six more clouds with max_layers = 3
    for (int k = 2; k < 8; k++) cloud(k, true);
    expect(rt.n_evictions >= 3 && g_live.size() <= 3 + 1 + 1, "LRU keeps the cloud cache at its bound");
    expect(g_live.size() == rt.cached_layers(), "every live device object is a cache entry");
    const size_t up = g_uploads;
    cloud(1, true);  // evicted long ago
    expect(g_uploads == up + 1, "an evicted layer is uploaded again when it returns");
    // ---- byte budget: maps report 1 MB each
    rt.max_layers  = 100;
    rt.byte_budget = 2500000;
    for (int k = 8; k < 12; k++) map(k, true);
    expect(rt.cached_bytes() <= 2500000, "byte budget respected, the entry just filled included");
    expect(g_live.size() == rt.cached_layers(), "live == cached after byte evictions");
    // ---- explicit hooks
    const size_t live = g_live.size();
    rt.release_layer(&bufs[11]);
    expect(g_live.size() == live - 1, "release_layer frees the device copy");
    rt.release_layer(&bufs[11]);
    expect(g_live.size() == live - 1, "releasing twice is harmless");
    rt.byte_budget = (size_t)1 << 40;
    const size_t up2 = g_uploads;
    map(10, false);
    rt.invalidate_layers();
    map(10, false);
    expect(g_uploads == up2 + 1, "invalidate_layers forces a re-verification (content unchanged here: re-upload by the flipped sample)");
    // ---- random traffic: no handle is freed twice, live == cached at every step
    std::mt19937 r(7);
    rt.max_layers = 4, rt.byte_budget = 6000000;
    for (int it = 0; it < 20000 && !bad; it++)
    {
        const int  k = r() % 12;
        const bool full = (r() % 4) == 0;
        if (r() % 97 == 0)
            for (size_t i = 0; i < N; i++) bufs[k].y[i] += 0.5f;  // a bulk edit
        if (r() % 2) map(k, full);
        else cloud(k, full);
        if (r() % 211 == 0) rt.release_layer(&bufs[r() % 12]);
        if (r() % 503 == 0) rt.invalidate_layers();
        if (g_live.size() != rt.cached_layers()) printf("step %d: live %zu != cached %zu\n", it, g_live.size(), rt.cached_layers()), bad++;
        if (rt.cached_layers() > 2 * 4 + 2) printf("step %d: %zu cached layers\n", it, rt.cached_layers()), bad++;
    }
    expect(g_double_free == 0, "no handle was freed twice");
    printf(bad ? "FAIL\n" : "ok\n");
    return bad;
}
