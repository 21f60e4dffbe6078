// mp2p_hip_host.hpp -- the reference-side binding of libmp2p_hip.so WITHOUT the MRPT types.
//
// Everything the plugin (mp2p_hip_plugin.cpp) does per matcher / solver call, written against plain
// host containers: SoA float buffers (CPointsMap::getPointsBufferRef_{x,y,z}), packed bit-fields
// (the storage of the std::vector<bool> inside pointcloud_bitfield_t::DenseOrSparseBitField) and
// vectors of the byte-compatible pair records.  The plugin only converts MRPT containers to these
// views; tests/test_gpu_boundary_hostpath.py and bench.py drive the SAME code through
// adapter/hostpath_capi.cpp, so the host path that is measured is the host path that ships.
//
// Per matcher call on the host (N_g = global points, N_l = local points, P = pairs emitted):
//   * MatchState in : one pass over the PACKED words (N/64 words; 0.17 M words for 10 M + 1 M points)
//                     to see whether anything is marked; nothing marked -> device reset, no transfer;
//                     otherwise the packed words are uploaded (1.4 MB instead of 11 MB of bytes)
//   * MatchState out: the marks a matcher leaves ARE the localIdx / globalIdx of the pairs it emitted
//                     (Matcher_Points_DistanceThreshold.cpp:116-120), so the host bit-fields are
//                     updated from the downloaded pair list: O(P), no per-point transfer
//   * Pairings      : only the P entries this call appended are downloaded (36 B / 72 B each)
// Nothing is O(N_g) per call except the read of the packed global bit-field.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "mp2p_hip.h"

#include <dlfcn.h>

namespace mp2p_hip_host
{
struct Error : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

// roctx ranges named after the reference's profiler sections (ICP.cpp:141 "align.3.1_matchers", :162 "align.3.2_solvers";
// SURVEY.md section 5): a rocprofv3 --marker-trace of an application that runs the plugin shows the matcher and the solver
// calls under the names `icp-run --profiler` prints.  libroctx64 is looked up at run time (no link dependency) and only
// when asked for: MP2P_HIP_ROCTX=1 opens it, otherwise it is used only if the process has loaded it already (RTLD_NOLOAD:
// a profiler that attaches brings it) -- a plain run of the plugin opens no library as a side effect, and a
// range then costs one null test.
struct RoctxRange
{
    using push_fn = int (*)(const char*);
    using pop_fn  = int (*)();
    static void resolve(push_fn& push, pop_fn& pop)
    {
        static push_fn s_push = nullptr;
        static pop_fn  s_pop  = nullptr;
        static const bool once  = [] {
            const char* want = std::getenv("MP2P_HIP_ROCTX");
            const int   mode = RTLD_LAZY | RTLD_LOCAL | ((want && want[0] == '1') ? 0 : RTLD_NOLOAD);
            void* h = dlopen("libroctx64.so", mode);
            if (!h) h = dlopen("libroctx64.so.4", mode);
            if (h)
            {
                s_push = reinterpret_cast<push_fn>(dlsym(h, "roctxRangePushA"));
                s_pop  = reinterpret_cast<pop_fn>(dlsym(h, "roctxRangePop"));
            }
            return true;
        }();
        (void)once;
        push = s_push, pop = s_pop;
    }
    pop_fn pop_ = nullptr;
    explicit RoctxRange(const char* name)
    {
        push_fn push;
        resolve(push, pop_);
        if (push && pop_) push(name);
        else pop_ = nullptr;
    }
    ~RoctxRange()
    {
        if (pop_) pop_();
    }
    RoctxRange(const RoctxRange&)            = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// ---- packed bit-field view (bit i = bit (i & 63) of word i / 64) ---------------------------------
struct BitView
{
    uint64_t* words = nullptr;
    size_t    nbits = 0;
    size_t    nwords() const { return (nbits + 63) / 64; }
    bool      valid() const { return words != nullptr || nbits == 0; }
    bool      any() const
    {
        const size_t full = nbits / 64;
        uint64_t     acc  = 0;
        for (size_t i = 0; i < full; i++) acc |= words[i];
        if (nbits & 63) acc |= words[full] & ((1ull << (nbits & 63)) - 1ull);
        return acc != 0;
    }
    void set(size_t i) const { words[i >> 6] |= 1ull << (i & 63); }
    bool test(size_t i) const { return (words[i >> 6] >> (i & 63)) & 1ull; }
};

// the marks of a matcher call: bit idx[i] of a packed field for every new pair (a software prefetch of the word 24 entries
// ahead was measured in round 4: 0.31 -> 0.43 ms for 250 k pairs -- the scattered words already overlap in the core's
// own miss queue -- and dropped)
inline void set_marks(const BitView& bits, const uint32_t* idx, size_t n)
{
    uint64_t* const w = bits.words;
    for (size_t i = 0; i < n; i++) w[idx[i] >> 6] |= 1ull << (idx[i] & 63);
}

// ---- content fingerprints of a point layer -------------------------------------------------------
// full: every byte (threads share the work); sampled: 1024 evenly strided points.  ICP::align holds
// its maps const for the whole call, so the plugin verifies a layer in full at ICP iteration 0 and
// with the sampled fingerprint (+ size + buffer addresses) afterwards.
inline uint64_t mix64(uint64_t h, uint64_t v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    return h ^ (h >> 32);
}
inline uint64_t hash_span(const float* p, size_t n)
{
    uint64_t     h = 0x243F6A8885A308D3ull;
    const size_t n2 = n / 2;
    for (size_t i = 0; i < n2; i++)
    {
        uint64_t v;
        std::memcpy(&v, p + 2 * i, 8);
        h = mix64(h, v);
    }
    if (n & 1)
    {
        uint32_t v;
        std::memcpy(&v, p + n - 1, 4);
        h = mix64(h, v);
    }
    return h;
}
inline uint64_t full_fingerprint(const float* x, const float* y, const float* z, size_t n, unsigned threads = 0)
{
    if (n == 0) return 0;
    // memory-bound: 120 MB for a 10 M-point layer; the host's memory system wants more than a handful of readers
    if (threads == 0) threads = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    threads = std::max(1u, std::min<unsigned>(threads, (unsigned)(n / 65536 + 1)));
    std::vector<uint64_t>    part(3 * threads, 0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back(
            [&, t]()
            {
                const size_t b = n * t / threads, e = n * (t + 1) / threads;
                part[3 * t] = hash_span(x + b, e - b), part[3 * t + 1] = hash_span(y + b, e - b),
                         part[3 * t + 2] = hash_span(z + b, e - b);
            });
    for (auto& t : th) t.join();
    uint64_t h = n;
    for (uint64_t v : part) h = mix64(h, v);
    return h;
}
inline uint64_t sampled_fingerprint(const float* x, const float* y, const float* z, size_t n)
{
    uint64_t     h = mix64(n, (uint64_t)(uintptr_t)x ^ ((uint64_t)(uintptr_t)y << 1) ^ ((uint64_t)(uintptr_t)z << 2));
    const size_t S = std::min<size_t>(n, 1024);
    for (size_t k = 0; k < S; k++)
    {
        const size_t i = S > 1 ? (n - 1) * k / (S - 1) : 0;
        uint32_t     a, b, c;
        std::memcpy(&a, x + i, 4), std::memcpy(&b, y + i, 4), std::memcpy(&c, z + i, 4);
        h = mix64(h, ((uint64_t)a << 32) | b), h = mix64(h, c);
    }
    return h;
}

// Fingerprint of a Pairings' point / plane list: its length, its last record, and the records at the
// sampled positions (the first 32 and every 64th).  It decides whether the device-resident list the
// matchers of this plugin left behind is the list a solver was handed (run_matchers copies Pairings by
// value, Matcher.cpp:74-77, so identity cannot be carried by the container; nothing in ICP::align edits
// a Pairings between run_matchers and run_solvers, ICP.cpp:148-260).  What it catches: any change of
// length (Pairings::push_back of another matcher's list, erasures), any reordering, truncation, or a
// foreign list of the same length -- with the certainty of a 64-bit hash over ~n/64 records.  What it
// does not: an in-place edit confined to unsampled records.  A caller that does such edits sets
// MP2P_HIP_HOST_STRICT=1 (Runtime::strict): every solver call then uploads the host list (0.09 ms per
// 10^5 point pairs) and nothing is assumed.  Reading every record instead costs the same 0.1 ms as the
// upload (the list has just come off the link and is in no cache), which is why it is not the default.
struct ListPrint
{
    uint64_t h = 0x13198A2E03707344ull, last = 0;
    size_t   n = 0;
    bool operator==(const ListPrint& o) const { return h == o.h && last == o.last && n == o.n; }
};
constexpr size_t PRINT_STRIDE = 64;  // a power of two >= 32
inline bool      sampled_pos(size_t i) { return i < 32 || (i & (PRINT_STRIDE - 1)) == 0; }
inline uint64_t pair_word(const mp2p_hip_pair_pt2pt& r)
{
    uint32_t e;
    std::memcpy(&e, &r.errorSquareAfterTransformation, 4);
    return (((uint64_t)r.globalIdx << 32) | r.localIdx) + ((uint64_t)e << 17);
}
inline uint64_t pair_word(const mp2p_hip_pair_pt2pl& r)
{
    uint64_t a, b;
    std::memcpy(&a, &r.plane[3], 8), std::memcpy(&b, &r.pt_local[0], 8);
    return a + (b << 1);
}
// continue fingerprint `f` (covering f.n records) with the n records at p
template <class Rec>
inline void print_feed(ListPrint& f, const Rec* p, size_t n)
{
    if (!n) return;
    size_t i = 0;
    for (; f.n + i < 32 && i < n; i++) f.h = mix64(f.h, pair_word(p[i]));
    i = ((f.n + i + PRINT_STRIDE - 1) & ~(PRINT_STRIDE - 1)) - f.n;  // next absolute multiple of the stride
    for (; i < n; i += PRINT_STRIDE) f.h = mix64(f.h, pair_word(p[i]));
    f.last = pair_word(p[n - 1]), f.n += n;
}
template <class Rec>
inline ListPrint list_print(const Rec* p, size_t n)
{
    ListPrint f;
    print_feed(f, p, n);
    return f;
}

// ---- one context + handle caches per thread (ICP::align is single-threaded per object) ----------
class Runtime
{
   public:
    mp2p_hip_ctx* ctx = nullptr;

    // the device of this thread's context: device_id() before the first get() (a host application that runs one
    // ICP object per GPU calls Runtime::device_id() = k in the thread that owns GPU k); MP2P_HIP_DEVICE otherwise
    static int& device_id()
    {
        static thread_local int id = std::getenv("MP2P_HIP_DEVICE") ? std::atoi(std::getenv("MP2P_HIP_DEVICE")) : 0;
        return id;
    }
    static Runtime& get()
    {
        static thread_local Runtime r;
        if (!r.ctx)
        {
            // this translation unit's view of mp2p_hip.h against the library it was loaded with (ABI version + struct sizes):
            // a plugin built against an older header stops here, not inside a solver that reads its weights from moved fields
            if (MP2P_HIP_ABI_CHECK() != MP2P_HIP_OK) throw Error(std::string("libmp2p_hip ABI mismatch: ") + mp2p_hip_last_error(nullptr));
            const int rc = mp2p_hip_ctx_create(device_id(), nullptr, &r.ctx);
            if (rc) throw Error(std::string("mp2p_hip_ctx_create: ") + mp2p_hip_last_error(nullptr));
        }
        return r;
    }
    void check(int rc) const
    {
        if (rc) throw Error(std::string("libmp2p_hip: ") + mp2p_hip_last_error(ctx));
    }

    // ---- point layers, keyed by the layer object's address.  full_check: verify every byte (the
    //      plugin asks for it at ICP iteration 0 and for layers it has not seen); otherwise size,
    //      buffer addresses and 1024 sampled points decide.  A changed layer is uploaded again --
    //      the role of nn_prepare_for_3d_queries() after mark_as_modified().
    struct Layer
    {
        void*    handle  = nullptr;
        size_t   n       = 0;
        uint64_t sampled = 0, full = 0;
        bool     full_known = false;
        uint64_t strided = 0;    // every 61st point (re-seen layers at ICP iteration 0)
        bool     strided_known = false;
        uint64_t last_use = 0;   // LRU stamp
        size_t   bytes    = 0;   // device bytes behind the handle
    };
    // Device copies are cached by the layer object's address.  A SLAM loop builds fresh layers per scan, so the
    // cache is bounded: at most `max_layers` entries per kind and `byte_budget` device bytes in total
    // (MP2P_HIP_HOST_CACHE_MB, default 16 GB of the 288); the least recently used entries go first -- never the two
    // of the running call -- and release_layer() is the explicit hook (a layer's destructor, an ICP-end callback).
    size_t max_layers  = 8;
    size_t byte_budget = (std::getenv("MP2P_HIP_HOST_CACHE_MB") ? (size_t)std::atoll(std::getenv("MP2P_HIP_HOST_CACHE_MB")) : 16384) << 20;
    mp2p_hip_map* global_layer(const void* key, const float* x, const float* y, const float* z, size_t n,
                               bool full_check)
    {
        Layer& e = maps_[key];
        e.last_use = ++clock_;
        if (!current(e, x, y, z, n, full_check))
        {
            if (e.handle) mp2p_hip_map_free(ctx, (mp2p_hip_map*)e.handle);
            e.handle = nullptr, e.bytes = 0;
            evict(key);
            mp2p_hip_map* h = nullptr;
            check(mp2p_hip_map_upload(ctx, x, y, z, n, nullptr, &h));
            e.handle = h;
            mp2p_hip_map_info info;
            if (mp2p_hip_map_get_info(ctx, h, &info) == 0) e.bytes = (size_t)info.device_bytes;
            n_map_uploads++;
            evict(key);  // the byte budget again, now that this layer's bytes count
        }
        return (mp2p_hip_map*)e.handle;
    }
    mp2p_hip_cloud* local_layer(const void* key, const float* x, const float* y, const float* z, size_t n,
                                bool full_check)
    {
        Layer& e = clouds_[key];
        e.last_use = ++clock_;
        if (!current(e, x, y, z, n, full_check))
        {
            if (e.handle) mp2p_hip_cloud_free(ctx, (mp2p_hip_cloud*)e.handle);
            e.handle = nullptr, e.bytes = 0;
            evict(key);
            mp2p_hip_cloud* h = nullptr;
            check(mp2p_hip_cloud_upload(ctx, x, y, z, n, &h));
            e.handle = h, e.bytes = n * 44;  // sorted copy + SoA + two index arrays
            n_cloud_uploads++;
            evict(key);
        }
        return (mp2p_hip_cloud*)e.handle;
    }
    // drop the device copy of a layer (either kind) now
    void release_layer(const void* key)
    {
        auto m = maps_.find(key);
        if (m != maps_.end())
        {
            if (m->second.handle) mp2p_hip_map_free(ctx, (mp2p_hip_map*)m->second.handle);
            maps_.erase(m);
        }
        auto c = clouds_.find(key);
        if (c != clouds_.end())
        {
            if (c->second.handle) mp2p_hip_cloud_free(ctx, (mp2p_hip_cloud*)c->second.handle);
            clouds_.erase(c);
        }
        token = Token();
    }
    size_t cached_layers() const { return maps_.size() + clouds_.size(); }
    size_t cached_bytes() const
    {
        size_t b = 0;
        for (auto& kv : maps_) b += kv.second.bytes;
        for (auto& kv : clouds_) b += kv.second.bytes;
        return b;
    }
    // a caller that edited a layer in place between two ICP iterations of its own loop
    void invalidate_layers()
    {
        for (auto& kv : maps_) kv.second.sampled = ~kv.second.sampled, kv.second.full_known = false;
        for (auto& kv : clouds_) kv.second.sampled = ~kv.second.sampled, kv.second.full_known = false;
    }

    // ---- MatchState: one device object per (N_g, N_l), brought to the host fields' content --------
    mp2p_hip_mstate* match_state(BitView g, BitView l, bool anyG, bool anyL)
    {
        auto& ms = mstates_[std::make_pair(g.nbits, l.nbits)];
        if (!ms) check(mp2p_hip_mstate_create(ctx, g.nbits, l.nbits, &ms));  // created clear
        if (!anyG && !anyL) check(mp2p_hip_mstate_reset(ctx, ms));
        else
        {
            if (!anyG || !anyL) check(mp2p_hip_mstate_reset(ctx, ms));
            check(mp2p_hip_mstate_upload_bits(ctx, ms, anyG ? g.words : nullptr, anyL ? l.words : nullptr));
            n_mstate_uploads++;
        }
        return ms;
    }

    // ---- device-resident Pairings shared by the matchers and solvers of one ICP iteration ----------
    mp2p_hip_pairs* pairs(size_t cap_pt, size_t cap_pl)
    {
        cap_pt = std::max<size_t>(cap_pt, 1);
        if (!dev_pairs_) check(mp2p_hip_pairs_create(ctx, cap_pt, cap_pl, &dev_pairs_));
        else check(mp2p_hip_pairs_reserve(ctx, dev_pairs_, cap_pt, cap_pl));
        cap_pt_ = std::max(cap_pt_, cap_pt), cap_pl_ = std::max(cap_pl_, cap_pl);
        return dev_pairs_;
    }
    mp2p_hip_pairs* conv_pairs(size_t cap)
    {
        if (!conv_pairs_) check(mp2p_hip_pairs_create(ctx, std::max<size_t>(cap, 1), 0, &conv_pairs_));
        else check(mp2p_hip_pairs_reserve(ctx, conv_pairs_, cap, 0));
        check(mp2p_hip_pairs_clear(ctx, conv_pairs_));
        return conv_pairs_;
    }
    size_t cap_pl() const { return cap_pl_; }
    // the list of a quality evaluator's private matcher (count_pt2pt_layer): its own device object, so that the
    // resident list of the running ICP iteration -- and the token that vouches for it -- stay as they are
    mp2p_hip_pairs* quality_pairs(size_t cap)
    {
        if (!quality_pairs_) check(mp2p_hip_pairs_create(ctx, std::max<size_t>(cap, 1), 0, &quality_pairs_));
        else check(mp2p_hip_pairs_reserve(ctx, quality_pairs_, cap, 0));
        check(mp2p_hip_pairs_clear(ctx, quality_pairs_));
        return quality_pairs_;
    }

    // what the device list holds, as the matchers of this plugin built it
    struct Token
    {
        bool      valid = false;
        size_t    n_pt = 0, n_pl = 0;
        ListPrint print_pt, print_pl;
        // the run_matchers call the list belongs to: (MatchState address, ICP iteration)
        const void* ms_key = nullptr;
        uint32_t    iteration = 0;
    } token;
    bool has_lines_planes = false;

    // counters for tests / the bench's host_boundary block
    size_t n_map_uploads = 0, n_cloud_uploads = 0, n_mstate_uploads = 0, n_pair_uploads = 0;
    // every solver call uploads the host Pairings (see ListPrint)
    bool strict = std::getenv("MP2P_HIP_HOST_STRICT") && std::getenv("MP2P_HIP_HOST_STRICT")[0] == '1';
    // opt-in: a layer verified in full before is re-verified on every 61st point only at later ICP
    // iteration-0 calls.  Off by default: an in-place edit between two align() calls that misses the stride would be
    // matched against the stale device copy, which the reference (it queries the live map) never does.
    bool trust_reseen = std::getenv("MP2P_HIP_HOST_TRUST_RESEEN") && std::getenv("MP2P_HIP_HOST_TRUST_RESEEN")[0] == '1';
    // page-locked scratch for the index arrays of the pairs a matcher call appended
    uint32_t* idx_scratch(size_t n)
    {
        if (n > idx_cap_)
        {
            if (idx_) mp2p_hip_host_free(ctx, idx_);
            idx_cap_ = std::max<size_t>(2 * n, 1 << 16);
            idx_     = static_cast<uint32_t*>(mp2p_hip_host_alloc(ctx, 2 * idx_cap_ * sizeof(uint32_t)));
            if (!idx_) idx_cap_ = 0, throw Error("mp2p_hip_host_alloc failed");
        }
        return idx_;
    }
    size_t idx_stride() const { return idx_cap_; }
    size_t predicted_pairs = 0;  // pairs the next point-matcher call is expected to add (see match_pt2pt_layer)
    // wall time of the stages of the last matcher call [ms]: {layers + MatchState in, device work until
    // the list length is known, container resize, pair copy-out (whole window), marks (inside that
    // window: they run on the index arrays while the records are on the link), list fingerprint}
    double stage_ms[6] = {0, 0, 0, 0, 0, 0};
    static double now_ms()
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    ~Runtime()
    {  // process exit: the HIP runtime may already be gone, nothing is freed explicitly
    }

   private:
    // least recently used entries out, until both bounds hold; `keep` (the entry being filled) and whatever was
    // used by the two most recent look-ups (the other layer of the running call) stay
    void evict(const void* keep)
    {
        auto oldest = [&](std::map<const void*, Layer>& m) {
            auto best = m.end();
            for (auto it = m.begin(); it != m.end(); ++it)
                if (it->first != keep && it->second.last_use + 2 <= clock_ && (best == m.end() || it->second.last_use < best->second.last_use)) best = it;
            return best;
        };
        for (;;)
        {
            const bool over_m = maps_.size() > max_layers, over_c = clouds_.size() > max_layers, over_b = cached_bytes() > byte_budget;
            if (!over_m && !over_c && !over_b) return;
            auto im = oldest(maps_), ic = oldest(clouds_);
            // the container that is over ITS count pays (freeing entries of the other kind would not help it); the byte
            // budget takes the least recently used entry of either kind; nothing evictable (everything left belongs
            // to the running call) ends the loop
            int take = 0;  // 1 = a map, 2 = a cloud
            if (over_m && im != maps_.end()) take = 1;
            else if (over_c && ic != clouds_.end()) take = 2;
            else if (over_b)
            {
                if (im != maps_.end() && (ic == clouds_.end() || im->second.last_use < ic->second.last_use)) take = 1;
                else if (ic != clouds_.end()) take = 2;
            }
            if (take == 1)
            {
                if (im->second.handle) mp2p_hip_map_free(ctx, (mp2p_hip_map*)im->second.handle);
                maps_.erase(im);
            }
            else if (take == 2)
            {
                if (ic->second.handle) mp2p_hip_cloud_free(ctx, (mp2p_hip_cloud*)ic->second.handle);
                clouds_.erase(ic);
            }
            else
                return;
            token = Token();
            n_evictions++;
        }
    }
    uint64_t clock_ = 0;

   public:
    size_t n_evictions = 0, n_full_checks = 0, n_reseen_checks = 0;

   private:
    // full_check (ICP iteration 0): the layer is hashed in full (120 MB for the bench map, spread over up to 32 threads)
    // -- what the reference's nn_prepare_for_3d_queries() after mark_as_modified() amounts to here, since the layer's
    // modification flag is not visible from outside MRPT.  With trust_reseen (MP2P_HIP_HOST_TRUST_RESEEN=1) a layer that
    // was verified in full before and still has the same address, size and 1024-point print is re-verified on every
    // 61st point only (~0.1 ms): for hosts that call release_layer() / invalidate_layers() when they edit a layer.
    static uint64_t strided_fingerprint(const float* x, const float* y, const float* z, size_t n)
    {
        uint64_t h = n;
        for (size_t i = 0; i < n; i += 61)
        {
            uint32_t a, b, c;
            std::memcpy(&a, x + i, 4), std::memcpy(&b, y + i, 4), std::memcpy(&c, z + i, 4);
            h = mix64(h, ((uint64_t)a << 32) | b), h = mix64(h, c);
        }
        return h;
    }
    bool current(Layer& e, const float* x, const float* y, const float* z, size_t n, bool full_check)
    {
        const uint64_t s = sampled_fingerprint(x, y, z, n);
        bool           ok = e.handle && e.n == n && e.sampled == s;
        if (ok && full_check && e.full_known && trust_reseen && !strict)
        {
            const uint64_t m = strided_fingerprint(x, y, z, n);
            if (e.strided_known && e.strided == m)
            {
                n_reseen_checks++;
                return true;
            }
            if (e.strided_known) ok = false;  // the content changed under the same address
            e.strided = m, e.strided_known = true;
        }
        uint64_t f = 0;
        if (full_check || !ok)
        {
            f = full_fingerprint(x, y, z, n);
            n_full_checks++;
            if (ok && e.full_known && e.full != f) ok = false;
            if (!e.strided_known || !ok) e.strided = strided_fingerprint(x, y, z, n), e.strided_known = true;
        }
        if (!ok || full_check) e.full = f, e.full_known = true;
        e.n = n, e.sampled = s;
        return ok;
    }
    std::map<const void*, Layer>                             maps_, clouds_;
    std::map<std::pair<size_t, size_t>, mp2p_hip_mstate*>    mstates_;
    uint32_t*       idx_     = nullptr;
    size_t          idx_cap_ = 0;
    mp2p_hip_pairs *dev_pairs_ = nullptr, *conv_pairs_ = nullptr, *quality_pairs_ = nullptr;
    size_t          cap_pt_ = 0, cap_pl_ = 0;
};

// ---- per-call glue shared by the matchers ---------------------------------------------------------
struct MatchCall
{
    const void* ms_key    = nullptr;  // address of the MatchState (one run_matchers call = one state)
    uint32_t    iteration = 0;        // MatchContext::icpIteration
    BitView     gbits, lbits;         // packed host bit-fields of this (global layer, local layer)
    // the local layer's own coordinate arrays (what its cloud handle was uploaded from): with them -- and MP2P_HIP_HOST_COPY_SOA=1 --
    // the point pairings come back as 24 instead of 44 bytes per pair and their `local` member is filled from here
    // (mp2p_hip_pairs_copy_pt2pt_begin_soa); else the record form
    const float* lx = nullptr;
    const float* ly = nullptr;
    const float* lz = nullptr;
    size_t       n_local = 0;
};

// the three-step copy of new point pairings: SoA form when the call carries the local layer's arrays
inline int copy_pt2pt_begin(Runtime& rt, const MatchCall& c, mp2p_hip_pairs* dp, size_t first, size_t n, mp2p_hip_pair_pt2pt* dst,
                            uint32_t* li, uint32_t* gi);

// start of a matcher call: the device list is continued when it belongs to the same run_matchers call
// (same MatchState object, same ICP iteration, and that state carries marks: a state without any is
// what run_matchers starts from -- Matcher.cpp:57-66 -- and clearing needlessly only costs the solver
// an upload, whereas continuing a list of an earlier call would grow it for nothing) and cleared
// otherwise.  Whatever is decided here, the solver trusts the device list only on its fingerprint.
inline mp2p_hip_pairs* begin_match(Runtime& rt, const MatchCall& c, bool fresh_state, size_t add_pt, size_t add_pl)
{
    auto&      tk   = rt.token;
    const bool same = tk.valid && !fresh_state && tk.ms_key == c.ms_key && tk.iteration == c.iteration;
    if (!same) tk = Runtime::Token();
    mp2p_hip_pairs* dp = rt.pairs(tk.n_pt + add_pt, std::max(rt.cap_pl(), tk.n_pl + add_pl));
    if (!same) rt.check(mp2p_hip_pairs_clear(rt.ctx, dp));
    tk.ms_key = c.ms_key, tk.iteration = c.iteration;
    return dp;
}

inline int copy_pt2pt_begin(Runtime& rt, const MatchCall& c, mp2p_hip_pairs* dp, size_t first, size_t n, mp2p_hip_pair_pt2pt* dst,
                            uint32_t* li, uint32_t* gi)
{
    const bool soa = [] {
        // opt-in: measured on two boxes (bench.py host_boundary, 1 M x 10 M, 238 k pairs per step, same call) the copy-out window was
        // 1.36-1.43 ms in this form against 0.66-1.26 ms in the record form -- the host side of the copy (two threads writing the
        // caller's vector) bounds it there, not the link, and assembling a record costs more than copying one
        const char* e = std::getenv("MP2P_HIP_HOST_COPY_SOA");
        return e && e[0] == '1';
    }();
    if (soa && c.lx && c.ly && c.lz && c.n_local)
        return mp2p_hip_pairs_copy_pt2pt_begin_soa(rt.ctx, dp, first, n, dst, li, gi, c.lx, c.ly, c.lz, c.n_local, 0);
    return mp2p_hip_pairs_copy_pt2pt_begin(rt.ctx, dp, first, n, dst, li, gi);
}

// ends an open split copy on every path out of a matcher call (an exception between begin and end would leave the
// context with a page-locked destination and every later begin refused: ADVICE r2)
struct CopyGuard
{
    Runtime& rt;
    bool     open = false;
    ~CopyGuard()
    {
        if (open) (void)mp2p_hip_pairs_copy_end(rt.ctx);
    }
};

// the new point pairings of a matcher call (everything beyond token.n_pt) into the caller's vector; the marks the
// matcher leaves ARE the indices of the new pairs: set from the two index arrays (8 bytes per pair, first on the
// link) while the 36-byte records are still arriving
template <class PairVec>
size_t fetch_new_pt2pt(Runtime& rt, const MatchCall& c, mp2p_hip_pairs* dp, PairVec& out, bool mark_local, bool mark_global,
                       double* t_marks_ms = nullptr)
{
    static_assert(sizeof(typename PairVec::value_type) == sizeof(mp2p_hip_pair_pt2pt), "pair record layout");
    auto&    tk   = rt.token;
    uint64_t n_pt = 0;
    rt.check(mp2p_hip_pairs_counts(rt.ctx, dp, &n_pt, nullptr, nullptr));
    const size_t n0 = out.size(), n = (size_t)n_pt - tk.n_pt;
    out.resize(n0 + n);
    auto* dst = reinterpret_cast<mp2p_hip_pair_pt2pt*>(out.data()) + n0;
    if (n)
    {
        uint32_t* li = rt.idx_scratch(n);
        uint32_t* gi = li + rt.idx_stride();
        CopyGuard guard{rt};
        rt.check(copy_pt2pt_begin(rt, c, dp, tk.n_pt, n, dst, li, gi));
        guard.open = true;
        rt.check(mp2p_hip_pairs_copy_wait_idx(rt.ctx));
        const double tm = Runtime::now_ms();
        if (mark_local && c.lbits.words) set_marks(c.lbits, li, n);
        if (mark_global && c.gbits.words) set_marks(c.gbits, gi, n);
        if (t_marks_ms) *t_marks_ms = Runtime::now_ms() - tm;
        guard.open = false;
        rt.check(mp2p_hip_pairs_copy_end(rt.ctx));
    }
    print_feed(tk.print_pt, dst, n);
    tk.n_pt += n, tk.valid = true;
    return n;
}
// ... and the new plane pairings (records handed to `emit`), local marks from their indices
template <class Emit>
size_t fetch_new_pt2pl(Runtime& rt, const MatchCall& c, mp2p_hip_pairs* dp, Emit&& emit)
{
    auto&    tk   = rt.token;
    uint64_t n_pl = 0;
    rt.check(mp2p_hip_pairs_counts(rt.ctx, dp, nullptr, &n_pl, nullptr));
    const size_t n = (size_t)n_pl - tk.n_pl;
    static thread_local std::vector<mp2p_hip_pair_pt2pl> rec;
    static thread_local std::vector<uint32_t>            idx;
    rec.resize(n), idx.resize(n);
    if (n) rt.check(mp2p_hip_pairs_copy_pt2pl(rt.ctx, dp, tk.n_pl, n, rec.data(), idx.data()));
    for (size_t i = 0; i < n; i++)
    {
        emit(rec[i]);
        if (c.lbits.words) c.lbits.set(idx[i]);
    }
    print_feed(tk.print_pl, rec.data(), n);
    tk.n_pl += n, tk.valid = true;
    return n;
}

// Matcher_Points_InlierRatio::implMatchOneLayer (Matcher_Points_InlierRatio.cpp:41-143) on host containers: the
// nearest neighbour of every visited point, the `inliersRatio` best kept.  Both marks are left for every emitted
// pair (:127-131).
template <class PairVec>
size_t match_inlier_ratio_layer(Runtime& rt, const MatchCall& c, mp2p_hip_map* map, mp2p_hip_cloud* cloud,
                                const double pose[12], const mp2p_hip_inlier_ratio_params& prm, const uint32_t* visit,
                                size_t n_visit, PairVec& out)
{
    const size_t    n_l  = mp2p_hip_cloud_size(cloud);
    const bool      anyG = c.gbits.words && c.gbits.any(), anyL = c.lbits.words && c.lbits.any();
    mp2p_hip_pairs* dp   = begin_match(rt, c, !anyG && !anyL, n_l, 0);
    rt.check(mp2p_hip_cloud_set_visit_order(rt.ctx, cloud, n_visit ? visit : nullptr, n_visit));
    mp2p_hip_mstate* ms = rt.match_state(c.gbits, c.lbits, anyG, anyL);
    rt.check(mp2p_hip_match_inlier_ratio(rt.ctx, map, cloud, pose, &prm, ms, dp));
    return fetch_new_pt2pt(rt, c, dp, out, true, true);
}

// Matcher_Adaptive::implMatchOneLayer (Matcher_Adaptive.cpp:59-314) on host containers, in the three steps of the
// boundary: neighbour search + histogram on the device, the THRESHOLD on the host by the caller's function (the
// plugin hands the 50 bins to MRPT's own CHistogram / confidenceIntervalsFromHistogram, :191-205, which removes the
// one step this repository cannot pin; tests pass mp2p_hip_adaptive_ci_high), selection on the device.  Point
// pairings into `out_pt`, plane pairings to `emit`; local marks for both (:260, 289-293), global marks are read only.
// Returns {point pairs, plane pairs} added.
template <class PairVec, class Threshold, class Emit>
std::pair<size_t, size_t> match_adaptive_layer(Runtime& rt, const MatchCall& c, mp2p_hip_map* map, mp2p_hip_cloud* cloud,
                                               const double pose[12], const mp2p_hip_adaptive_params& prm,
                                               Threshold&& ci_high_of, PairVec& out_pt, Emit&& emit, double* ci_high_out = nullptr)
{
    const size_t    n_l  = mp2p_hip_cloud_size(cloud);
    const bool      anyG = c.gbits.words && c.gbits.any(), anyL = c.lbits.words && c.lbits.any();
    mp2p_hip_pairs* dp   = begin_match(rt, c, !anyG && !anyL, n_l * std::max<uint32_t>(1u, prm.maxPt2PtCorrespondences), n_l);
    rt.check(mp2p_hip_cloud_set_visit_order(rt.ctx, cloud, nullptr, 0));  // the reference throws for a subset here
    mp2p_hip_mstate* ms = rt.match_state(c.gbits, c.lbits, anyG, anyL);
    mp2p_hip_adaptive_hist hist;
    rt.check(mp2p_hip_adaptive_search(rt.ctx, map, cloud, pose, &prm, ms, &hist));
    if (!hist.valid) return {0, 0};  // no local point has a neighbour within absoluteMaxSearchDistance
    const double ci = ci_high_of(hist);
    if (ci_high_out) *ci_high_out = ci;
    rt.check(mp2p_hip_adaptive_select(rt.ctx, map, cloud, &prm, ci, ms, dp));
    const size_t n_pl = fetch_new_pt2pl(rt, c, dp, emit);
    const size_t n_pt = fetch_new_pt2pt(rt, c, dp, out_pt, true, false);
    return {n_pt, n_pl};
}

// FilterDecimateVoxels (mp2p_icp_filters/src/FilterDecimateVoxels.cpp:107-381) on host arrays: the decimated points
// (and, per output point, the index of the input point it is -- NONE for an average) -- one input layer, or several
// concatenated by the caller in layer order.  RandomPoint is not offered (mrpt::random stream): the plugin keeps the
// reference's own code for it.
inline size_t filter_decimate(Runtime& rt, const float* x, const float* y, const float* z, size_t n,
                              const mp2p_hip_decimate_params& prm, std::vector<float>& ox, std::vector<float>& oy,
                              std::vector<float>& oz, std::vector<uint32_t>& src)
{
    ox.resize(n), oy.resize(n), oz.resize(n), src.resize(n);
    size_t m = 0;
    if (n) rt.check(mp2p_hip_filter_decimate_voxels(rt.ctx, x, y, z, n, &prm, ox.data(), oy.data(), oz.data(), src.data(), &m));
    ox.resize(m), oy.resize(m), oz.resize(m), src.resize(m);
    return m;
}

// Matcher_Points_DistanceThreshold::implMatchOneLayer on host containers.  `out` (a vector of records
// layout-compatible with mp2p_hip_pair_pt2pt, e.g. mrpt::tfest::TMatchingPairList) is appended to.
// Returns the number of pairs added; *potential_add = what the reference adds to potential_pairings.
template <class PairVec>
size_t match_pt2pt_layer(Runtime& rt, const MatchCall& c, mp2p_hip_map* map, mp2p_hip_cloud* cloud,
                         const double pose[12], const mp2p_hip_pt2pt_params& prm, const uint32_t* visit,
                         size_t n_visit, PairVec& out)
{
    static_assert(sizeof(typename PairVec::value_type) == sizeof(mp2p_hip_pair_pt2pt), "pair record layout");
    const size_t    n_l  = mp2p_hip_cloud_size(cloud);
    double          t0   = Runtime::now_ms(), t1;
    const bool      anyG = c.gbits.words && c.gbits.any(), anyL = c.lbits.words && c.lbits.any();
    mp2p_hip_pairs* dp   = begin_match(rt, c, !anyG && !anyL, n_l * prm.pairingsPerPoint, 0);
    auto&           tk   = rt.token;
    rt.check(mp2p_hip_cloud_set_visit_order(rt.ctx, cloud, n_visit ? visit : nullptr, n_visit));
    mp2p_hip_mstate* ms = rt.match_state(c.gbits, c.lbits, anyG, anyL);
    rt.stage_ms[0] = (t1 = Runtime::now_ms()) - t0, t0 = t1;
    rt.check(mp2p_hip_match_pt2pt(rt.ctx, map, cloud, pose, &prm, ms, dp));
    // std::vector::resize value-initialises what it adds (0.09 ms for 10^5 records): the container grows to the
    // PREDICTED length (last call's + 10 %) while the device is still searching; the exact resize afterwards only
    // trims, or fills the few records the prediction missed
    const size_t n0 = out.size();
    if (rt.predicted_pairs) out.resize(n0 + std::min(rt.predicted_pairs, n_l * (size_t)prm.pairingsPerPoint));
    // the new list length first (24 bytes, one wait), then exactly the new entries into the caller's vector
    uint64_t n_pt = 0;
    rt.check(mp2p_hip_pairs_counts(rt.ctx, dp, &n_pt, nullptr, nullptr));
    rt.stage_ms[1] = (t1 = Runtime::now_ms()) - t0, t0 = t1;
    const size_t n = (size_t)n_pt - tk.n_pt;
    out.resize(n0 + n);
    rt.predicted_pairs = n + n / 10 + 64;
    rt.stage_ms[2] = (t1 = Runtime::now_ms()) - t0, t0 = t1;
    auto* dst = reinterpret_cast<mp2p_hip_pair_pt2pt*>(out.data()) + n0;
    // the marks this matcher leaves (only when global re-use is forbidden, :116-120) ARE the indices of
    // the new pairs: set from the two index arrays (8 bytes per pair, first on the link) while the
    // 36-byte records are still arriving in the caller's vector
    const bool marks = !prm.allowMatchAlreadyMatchedGlobalPoints;
    double     t_marks = 0.0;
    if (n)
    {
        uint32_t* li = rt.idx_scratch(n);
        uint32_t* gi = li + rt.idx_stride();
        CopyGuard guard{rt};
        rt.check(copy_pt2pt_begin(rt, c, dp, tk.n_pt, n, dst, li, gi));
        guard.open = true;
        rt.check(mp2p_hip_pairs_copy_wait_idx(rt.ctx));
        const double tm = Runtime::now_ms();
        if (marks && c.lbits.words) set_marks(c.lbits, li, n);
        if (marks && c.gbits.words) set_marks(c.gbits, gi, n);
        t_marks = Runtime::now_ms() - tm;
        guard.open = false;
        rt.check(mp2p_hip_pairs_copy_end(rt.ctx));
    }
    rt.stage_ms[3] = (t1 = Runtime::now_ms()) - t0, t0 = t1;
    rt.stage_ms[4] = t_marks;
    print_feed(tk.print_pt, dst, n);
    tk.n_pt += n, tk.valid = true;
    rt.stage_ms[5] = Runtime::now_ms() - t0;
    return n;
}

// Matcher_Point2Plane::implMatchOneLayer on host containers: planes into `out` (records layout-
// compatible with mp2p_hip_pair_pt2pl are produced into a scratch vector and handed to `emit`)
template <class Emit>
size_t match_pt2pl_layer(Runtime& rt, const MatchCall& c, mp2p_hip_map* map, mp2p_hip_cloud* cloud,
                         const double pose[12], const mp2p_hip_pt2pl_params& prm, const uint32_t* visit,
                         size_t n_visit, Emit&& emit)
{
    const size_t    n_l  = mp2p_hip_cloud_size(cloud);
    const bool      anyG = c.gbits.words && c.gbits.any(), anyL = c.lbits.words && c.lbits.any();
    mp2p_hip_pairs* dp   = begin_match(rt, c, !anyG && !anyL, 0, n_l);
    auto&           tk   = rt.token;
    rt.check(mp2p_hip_cloud_set_visit_order(rt.ctx, cloud, n_visit ? visit : nullptr, n_visit));
    // global marks are neither read nor set by this matcher (:87-90)
    mp2p_hip_mstate* ms = rt.match_state(BitView{nullptr, c.gbits.nbits}, c.lbits, false, anyL);
    rt.check(mp2p_hip_match_pt2pl(rt.ctx, map, cloud, pose, &prm, ms, dp));
    uint64_t n_pl = 0;
    rt.check(mp2p_hip_pairs_counts(rt.ctx, dp, nullptr, &n_pl, nullptr));
    const size_t n = (size_t)n_pl - tk.n_pl;
    static thread_local std::vector<mp2p_hip_pair_pt2pl> rec;
    static thread_local std::vector<uint32_t>            idx;
    rec.resize(n), idx.resize(n);
    if (n) rt.check(mp2p_hip_pairs_copy_pt2pl(rt.ctx, dp, tk.n_pl, n, rec.data(), idx.data()));
    for (size_t i = 0; i < n; i++)
    {
        emit(rec[i]);
        if (c.lbits.words) c.lbits.set(idx[i]);  // Matcher_Point2Plane.cpp:109
    }
    print_feed(tk.print_pl, rec.data(), n);
    tk.n_pl += n, tk.valid = true;
    return n;
}

// QualityEvaluator_PairedRatio::evaluate with reuse_icp_pairings = false (QualityEvaluator_PairedRatio.cpp:57-75) runs its
// private Matcher_Points_DistanceThreshold on a FRESH MatchState (:59) for the sake of two numbers: Pairings::size() and
// potential_pairings (:67-70).  So nothing else comes back from the device: 24 bytes instead of 36 per pair, no marks
// (the state is discarded), and the list lives in a device object of its own.  Called at every quality checkpoint and
// at the end of ICP::align (ICP.cpp:259-283, 322-324) -- on the map whose index is already resident.
inline size_t count_pt2pt_layer(Runtime& rt, mp2p_hip_map* map, mp2p_hip_cloud* cloud, const double pose[12],
                                const mp2p_hip_pt2pt_params& prm, const uint32_t* visit, size_t n_visit)
{
    const size_t    n_l = mp2p_hip_cloud_size(cloud);
    mp2p_hip_pairs* dp  = rt.quality_pairs(n_l * std::max<uint32_t>(1u, prm.pairingsPerPoint));
    rt.check(mp2p_hip_cloud_set_visit_order(rt.ctx, cloud, n_visit ? visit : nullptr, n_visit));
    rt.check(mp2p_hip_match_pt2pt(rt.ctx, map, cloud, pose, &prm, nullptr /* a fresh MatchState: nothing marked */, dp));
    uint64_t n_pt = 0;
    rt.check(mp2p_hip_pairs_counts(rt.ctx, dp, &n_pt, nullptr, nullptr));
    return (size_t)n_pt;
}

// Pairings -> the device handle a solver reads.  The lists the matchers of this plugin produced in the
// same ICP iteration are still in HBM: recognised by their fingerprint (ListPrint); anything else is uploaded.  The
// token is consumed: the next solver call without a matcher call in between uploads again.
inline mp2p_hip_pairs* pairings_to_device(Runtime& rt, const mp2p_hip_pair_pt2pt* pt, size_t n_pt,
                                          const mp2p_hip_pair_pt2pl* pl, size_t n_pl,
                                          const mp2p_hip_pair_pt2ln* ln, size_t n_ln,
                                          const mp2p_hip_pair_pl2pl* pp, size_t n_pp)
{
    auto& tk = rt.token;
    bool  resident = !rt.strict && tk.valid && tk.n_pt == n_pt && tk.n_pl == n_pl && (n_pt + n_pl) > 0;
    if (resident && n_pt) resident = list_print(pt, n_pt) == tk.print_pt;
    if (resident && n_pl) resident = list_print(pl, n_pl) == tk.print_pl;
    mp2p_hip_pairs* dp = rt.pairs(n_pt, std::max(rt.cap_pl(), n_pl));
    if (!resident)
    {
        rt.check(mp2p_hip_pairs_upload(rt.ctx, dp, pt, n_pt, pl, n_pl));
        rt.n_pair_uploads++;
    }
    tk = Runtime::Token();  // consumed
    if (n_ln || n_pp || rt.has_lines_planes)
        rt.check(mp2p_hip_pairs_upload_lines_planes(rt.ctx, dp, ln, n_ln, pp, n_pp));
    rt.has_lines_planes = n_ln || n_pp;
    return dp;
}

}  // namespace mp2p_hip_host
