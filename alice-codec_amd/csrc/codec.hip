// Host side of libalice_codec.so: device/stream/memory plumbing, the chunk object and
// `.alc` (de)serialiser, the FrameEncoder / FrameDecoder orchestration and the C ABI
// declared in include/alice_codec.h.
//
// Reference behaviour restated (reference checkout, file:line):
//   FrameEncoder::encode   src/pipeline.rs:377-507     FrameDecoder::decode  src/pipeline.rs:537-624
//   EncodedChunk::to_bytes src/pipeline.rs:200-226     from_bytes            src/pipeline.rs:235-313
//   C ABI                  src/ffi.rs:12-315
//
// Every compute step is a HIP kernel; nothing here falls back to the CPU.  If no HIP
// device is usable the entry points fail (NULL / error code) -- loudly via
// alice_codec_last_error().
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/alice_codec.h"
#include "../../include/alice_codec_test.h"
#include "common.h"
#include "kernels.h"

using namespace alice;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------

namespace {

thread_local int tl_err = kOk;
thread_local std::string tl_msg;

int fail(int code, const std::string& msg) {
    tl_err = code;
    tl_msg = msg;
    return code;
}
void clear_error() { tl_err = kOk; tl_msg.clear(); }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            int code__ = (e__ == hipErrorOutOfMemory) ? (int)kOutOfMemory : (int)kDeviceError; \
            return fail(code__, std::string(#expr) + ": " + hipGetErrorString(e__));         \
        }                                                                                      \
    } while (0)

// ------------------------------------------------------------------------------------------
// device, stream, memory pool
// ------------------------------------------------------------------------------------------

thread_local int tl_device = -1;         // -1: not chosen yet (device 0 on first use)
// stream the work of the current entry point runs on: buffers allocated meanwhile remember it and drain it before they
// go back to the pool (an error return may leave kernels in flight on them)
thread_local hipStream_t tl_scope_stream = nullptr;

// HIP maps streams onto hardware queues, 4 by default, and kernels of streams that share a queue run one after the other.
// A chain kernel runs for seconds, so with the default only four host threads' calls make progress at a time (measured:
// 64 threads through alice_codec_encode64 ran 4 chunks at a time).  Ask for more before the runtime creates its queues --
// 8 is what the device honours (32 behaved like 8); a host that has initialised HIP already, or set the variable itself,
// keeps what it has.  ALICE_CODEC_KEEP_HW_QUEUES=1 leaves the environment alone.
void widen_hw_queues_once() {
    static const bool done = [] {
        if (!getenv("ALICE_CODEC_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
        return true;
    }();
    (void)done;
}

int ensure_device() {
    widen_hw_queues_once();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(kDeviceError, "no usable HIP device: this library has no CPU fallback (" +
                                      std::string(e != hipSuccess ? hipGetErrorString(e) : "device count 0") + ")");
    if (tl_device < 0) {
        int cur = 0;
        if (hipGetDevice(&cur) == hipSuccess) tl_device = cur; else tl_device = 0;
    }
    if (tl_device >= count) return fail(kDeviceError, "device index out of range");
    HIP_TRY(hipSetDevice(tl_device));
    return kOk;
}

// the stream this thread's host-side work goes to: one of the chain hub's shared short streams (defined below ChainHub)
int get_stream(hipStream_t* out);

// Size-bucketed cache of device allocations.  hipFree waits for EVERY kernel running on the device -- with other threads'
// chains in flight that is seconds (64 host threads through alice_codec_encode64 took 49 s instead of 6 while the cache
// was capped at 8 GB and every call's buffers were freed behind it) -- so blocks are kept up to half of the device's
// memory and given back only by alice_codec_trim() or when an allocation fails.
class DevicePool {
public:
    int alloc(size_t bytes, void** out) {
        const size_t b = bucket(bytes);
        int dev = tl_device;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = free_.find({dev, b});
            if (it != free_.end() && !it->second.empty()) {
                *out = it->second.back();
                it->second.pop_back();
                cached_ -= b;
                return kOk;
            }
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, b);
        if (e != hipSuccess) {
            trim();
            e = hipMalloc(&p, b);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(kOutOfMemory, "hipMalloc of " + std::to_string(b) + " bytes failed: " + hipGetErrorString(e));
        }
        *out = p;
        return kOk;
    }
    // dev = the device the block was allocated on (the releasing thread may have moved on to another one)
    void release(void* p, size_t bytes, int dev) {
        if (!p) return;
        const size_t b = bucket(bytes);
        std::lock_guard<std::mutex> g(mu_);
        // (very large blocks -- a batch's symbol volume -- only ever fit the batch that made them: not worth keeping)
        if (b > (size_t(8) << 30) || cached_ + b > max_cached()) { (void)hipFree(p); return; }
        free_[{dev, b}].push_back(p);
        cached_ += b;
    }
    void trim() {
        std::lock_guard<std::mutex> g(mu_);
        for (auto& kv : free_) for (void* p : kv.second) (void)hipFree(p);
        free_.clear();
        cached_ = 0;
    }
    size_t cached() { std::lock_guard<std::mutex> g(mu_); return cached_; }
private:
    static size_t bucket(size_t bytes) {
        if (bytes < 256) bytes = 256;
        if (bytes <= (1u << 20)) {  // next power of two
            size_t b = 256;
            while (b < bytes) b <<= 1;
            return b;
        }
        const size_t g = size_t(2) << 20;  // 2 MiB granules
        return (bytes + g - 1) / g * g;
    }
    // half of the device's memory (of the first device asked about: the devices of a node are alike), at least 8 GB
    static size_t max_cached() {
        static const size_t cap = [] {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return size_t(8) << 30; }
            return std::max(total_b / 2, size_t(8) << 30);
        }();
        return cap;
    }
    std::mutex mu_;
    std::map<std::pair<int, size_t>, std::vector<void*>> free_;
    size_t cached_ = 0;
};

DevicePool& pool() {
    static DevicePool* p = new DevicePool();  // leaked on purpose: no HIP calls during static destruction
    return *p;
}

struct DevBuf {
    void* p = nullptr;
    size_t n = 0;
    int dev = 0;                // device the block lives on
    hipStream_t st = nullptr;   // stream whose work may still touch it
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { reset(); }
    int alloc(size_t bytes) {
        reset();
        if (bytes == 0) bytes = 16;
        int rc = pool().alloc(bytes, &p);
        if (rc == kOk) { n = bytes; dev = tl_device; st = tl_scope_stream; }
        return rc;
    }
    // The pool is shared by all threads, each with a stream of its own: a block must be idle before another thread can
    // get it.  On the normal paths the stream has been synchronised already and this returns at once.
    void reset() {
        if (p) {
            if (st) (void)hipStreamSynchronize(st);
            pool().release(p, n, dev);
        }
        p = nullptr; n = 0; st = nullptr;
    }
    template <typename T> T* as() const { return (T*)p; }
};

}  // namespace
// RansDecoder (src/rans.rs:321-326: state, input, pos).  The input is uploaded once, at the first decode, and stays on the
// device the object was first used on until the object is destroyed (decode() one symbol at a time must not re-send it).
struct AliceRansDecoder {
    std::vector<uint8_t> input;
    uint32_t state = 0;
    uint64_t pos = 0;
    bool started = false;
    DevBuf d_input;
    int device = -1;
};
namespace {

// Entry points that run on a CALLER's stream name it here for the duration of the call: buffers allocated meanwhile
// remember it (DevBuf::alloc) and drain it before they go back to the pool.  Cleared on return, so that a later call of
// the same thread never tags its buffers with a stream the caller may have destroyed since.
struct ScopeStream {
    explicit ScopeStream(hipStream_t s) { tl_scope_stream = s; }
    ~ScopeStream() { tl_scope_stream = nullptr; }
    ScopeStream(const ScopeStream&) = delete;
    ScopeStream& operator=(const ScopeStream&) = delete;
};

// Entry points that work on an object tied to a device (a batch) switch this thread to it for the call.
struct DeviceScope {
    int saved;
    bool ok;
    explicit DeviceScope(int device) : saved(tl_device), ok(true) {
        tl_device = device;
        if (ensure_device() != kOk) ok = false;
    }
    ~DeviceScope() {
        tl_device = saved;
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};

#define TRY(expr) do { int rc__ = (expr); if (rc__ != kOk) return rc__; } while (0)

// ------------------------------------------------------------------------------------------
// ChainHub: the chain launches of concurrent host calls, merged
//
// A chain kernel runs for seconds and kernels of streams that share a hardware queue run one after the other; the device
// grants eight queues.  With a stream per calling thread, 64 threads in alice_codec_encode64 therefore had eight chunks'
// chains running at a time (398 Mpix/s; profiles/r03_host_api_1080p64_8_hw_queues.json) on a GPU that holds 341 chunks'
// chains.  The whole-chunk host entry points (encode / decode of one chunk, of many chunks) now run like this:
//   * everything short -- copies, transforms, table builds, stream compaction -- goes to one of kShort streams shared by
//     all calling threads; nothing on them ever waits for a chain, so they stay short;
//   * a call hands its chains (descriptors + an event that says their inputs are complete) to the hub and sleeps; one of
//     the waiting threads, the leader, merges everything that is pending into ONE launch on one of kLanes lane streams and
//     every member then waits on the host for that launch's event before it queues its own tail on its short stream;
//   * kShort + kLanes = 7 streams in all, so every one of them owns a hardware queue as long as the host process does not
//     crowd the eight with streams of its own.
// Gathering: a call announces itself when it enters (HubTicket) and the leader waits for the announced calls to arrive, but
// never longer than a tenth of its own chains' run time (a call with a small chunk never waits for a large one's upload);
// a lone caller launches at once.  When all lanes are busy the pending calls pile up and leave together with the next
// free lane.  (Round 3 first tried a combiner that kept the callers' own streams: their short work then sat in hardware
// queues behind other callers' merged chains, and it was slower than no combiner at all:
// profiles/r03_host_api_1080p64_chain_combiner_rejected.json.)
// ------------------------------------------------------------------------------------------
constexpr int kHubShort = 3, kHubLanes = 4, kHubMaxMerged = 1023;   // <= 1023 chains: the one-chain-per-SIMD instances

struct HubLaunch {
    hipEvent_t done = nullptr;
    DevBuf descs;
    std::vector<RansEncodeDesc> enc;      // host copies: alive until the launch is over
    std::vector<RansDecodeDesc> dec;
    int rc = kOk;
    std::string msg;
    ~HubLaunch() { if (done) (void)hipEventDestroy(done); }
};

struct HubJob {
    bool encode = true;
    std::vector<RansEncodeDesc> enc;
    std::vector<RansDecodeDesc> dec;
    hipEvent_t ready = nullptr;           // recorded by the caller on its short stream: the chains' inputs are complete
    double seconds = 0.0;                 // rough run time of the chains (bounds the gather wait)
    std::shared_ptr<HubLaunch> launch;    // set by the leader
    size_t chains() const { return encode ? enc.size() : dec.size(); }
};

class ChainHub {
public:
    static ChainHub* of_device(int device) {
        static std::mutex mu;
        static std::map<int, ChainHub*> hubs;   // leaked on purpose: no HIP calls during static destruction
        std::lock_guard<std::mutex> g(mu);
        auto it = hubs.find(device);
        if (it != hubs.end()) return it->second;
        ChainHub* h = new ChainHub();
        for (auto& s : h->short_) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) h->ok_ = false;
        for (auto& l : h->lanes_) if (hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking) != hipSuccess) h->ok_ = false;
        hubs[device] = h;
        return h;
    }
    bool ok() const { return ok_; }
    hipStream_t short_stream() {
        static std::atomic<unsigned> next{0};
        thread_local unsigned mine = next.fetch_add(1u);
        return short_[mine % kHubShort];
    }
    // Admission: the whole-chunk calls in flight together must fit the device.  A call states what it is about to allocate
    // and waits here while the calls already inside hold too much (a call alone always enters); without this a thread pool
    // larger than the memory allows would turn the surplus calls into out-of-memory errors instead of a queue.
    void admit(size_t bytes) {
        std::unique_lock<std::mutex> lk(adm_mu_);
        const uint64_t my_turn = next_turn_++;   // first come, first admitted: a large call is not starved by a stream of small ones
        adm_cv_.wait(lk, [&] {
            if (my_turn != serving_) return false;
            if (in_flight_ == 0) {
                // nobody inside: what the device has free now plus what the pool would hand back is the budget (90 % of it)
                if (!budget_fixed_) {
                    size_t free_b = 0, total_b = 0;
                    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = size_t(16) << 30; }
                    budget_ = (free_b + pool().cached()) / 10 * 9;
                }
                return true;
            }
            return in_flight_ + bytes <= budget_;
        });
        in_flight_ += bytes;
        ++serving_;
        adm_cv_.notify_all();
        if (getenv("ALICE_CODEC_DEBUG"))
            fprintf(stderr, "[alice] hub: admitted %.2f GB, %.2f GB in flight of a budget of %.2f GB\n", bytes / 1e9, in_flight_ / 1e9, budget_ / 1e9);
    }
    void leave(size_t bytes) { { std::lock_guard<std::mutex> g(adm_mu_); in_flight_ -= std::min(in_flight_, bytes); } adm_cv_.notify_all(); }
    void set_budget(size_t bytes) { { std::lock_guard<std::mutex> g(adm_mu_); budget_ = bytes; budget_fixed_ = bytes != 0; } adm_cv_.notify_all(); }
    void announce() { std::lock_guard<std::mutex> g(mu_); ++expected_; }
    void arrived_or_gone() { { std::lock_guard<std::mutex> g(mu_); if (expected_ > 0) --expected_; } cv_.notify_all(); }

    // Hands the job's chains to the next merged launch and returns when they have run.  The caller has recorded job.ready.
    int run(HubJob& job) {
        std::unique_lock<std::mutex> lk(mu_);
        pending_.push_back(&job);
        cv_.notify_all();
        while (!job.launch) {
            if (leader_active_) { cv_.wait(lk); continue; }
            leader_active_ = true;
            // gather the announced calls, for at most a tenth of this job's chain time (half a second at most)
            const auto deadline = std::chrono::steady_clock::now() +
                                  std::chrono::microseconds((long long)(std::min(0.5, 0.1 * job.seconds) * 1e6));
            cv_.wait_until(lk, deadline, [&] { return expected_ == 0; });
            // a free lane (the pending list keeps growing meanwhile)
            int lane = -1;
            for (;;) {
                for (int i = 0; i < kHubLanes && lane < 0; ++i) {
                    Lane& l = lanes_[(next_lane_ + i) % kHubLanes];
                    if (!l.last || !l.last->done || l.last->rc != kOk || hipEventQuery(l.last->done) == hipSuccess)
                        lane = (next_lane_ + i) % kHubLanes;
                }
                if (lane >= 0) break;
                (void)hipGetLastError();   // hipErrorNotReady is not an error
                cv_.wait_for(lk, std::chrono::milliseconds(1));
            }
            next_lane_ = (lane + 1) % kHubLanes;
            // everything pending of this job's kind, this job first, up to the merged-launch limit
            std::vector<HubJob*> take{&job};
            size_t total = job.chains();
            for (HubJob* j : pending_)
                if (j != &job && j->encode == job.encode && total + j->chains() <= (size_t)kHubMaxMerged) { take.push_back(j); total += j->chains(); }
            for (HubJob* j : take) pending_.erase(std::find(pending_.begin(), pending_.end(), j));
            auto L = std::make_shared<HubLaunch>();
            lanes_[lane].last = L;
            lk.unlock();
            try {
                launch(*L, take, lanes_[lane].st, job.encode, total);
            } catch (const std::exception& e) {   // host allocation failure while merging the descriptors
                L->rc = kOutOfMemory; L->msg = std::string("merged chain launch: ") + e.what();
            }
            if (getenv("ALICE_CODEC_DEBUG")) {
                static const auto t0 = std::chrono::steady_clock::now();
                fprintf(stderr, "[alice] hub: t=%.3f s, %s launch of %zu calls, %zu chains, lane %d\n",
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), job.encode ? "encode" : "decode",
                        take.size(), total, lane);
            }
            lk.lock();
            for (HubJob* j : take) j->launch = L;
            leader_active_ = false;
            cv_.notify_all();
        }
        std::shared_ptr<HubLaunch> L = job.launch;
        lk.unlock();
        if (L->rc != kOk) return fail(L->rc, L->msg);
        HIP_TRY(hipEventSynchronize(L->done));
        cv_.notify_all();   // a lane has come free
        return kOk;
    }

private:
    struct Lane { hipStream_t st = nullptr; std::shared_ptr<HubLaunch> last; };
    static void launch(HubLaunch& L, const std::vector<HubJob*>& take, hipStream_t st, bool encode, size_t total) {
        auto bad = [&](hipError_t e, const char* what) {
            if (e == hipSuccess || L.rc != kOk) return;
            L.rc = e == hipErrorOutOfMemory ? (int)kOutOfMemory : (int)kDeviceError;
            L.msg = std::string(what) + ": " + hipGetErrorString(e);
        };
        bad(hipEventCreateWithFlags(&L.done, hipEventDisableTiming), "hipEventCreate");
        size_t bytes = 0;
        const void* host = nullptr;
        if (encode) {
            L.enc.reserve(total);
            for (HubJob* j : take) L.enc.insert(L.enc.end(), j->enc.begin(), j->enc.end());
            bytes = L.enc.size() * sizeof(RansEncodeDesc); host = L.enc.data();
        } else {
            L.dec.reserve(total);
            for (HubJob* j : take) L.dec.insert(L.dec.end(), j->dec.begin(), j->dec.end());
            bytes = L.dec.size() * sizeof(RansDecodeDesc); host = L.dec.data();
        }
        if (L.rc == kOk && L.descs.alloc(bytes) != kOk) { L.rc = kOutOfMemory; L.msg = "descriptor buffer of a merged chain launch"; }
        L.descs.st = nullptr;   // every member waits for `done` before the launch object dies: nothing to drain then
        if (L.rc != kOk) return;
        for (HubJob* j : take) bad(hipStreamWaitEvent(st, j->ready, 0), "hipStreamWaitEvent");
        bad(hipMemcpyAsync(L.descs.p, host, bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync(descriptors)");
        if (L.rc != kOk) return;
        if (encode) launch_rans_encode_descs(L.descs.as<RansEncodeDesc>(), (int)total, st);
        else launch_rans_decode(L.descs.as<RansDecodeDesc>(), nullptr, (int)total, st);
        bad(hipGetLastError(), "chain launch");
        bad(hipEventRecord(L.done, st), "hipEventRecord");
    }

    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<HubJob*> pending_;
    int expected_ = 0;
    bool leader_active_ = false;
    bool ok_ = true;
    std::mutex adm_mu_;
    std::condition_variable adm_cv_;
    size_t in_flight_ = 0, budget_ = 0;   // bytes; the budget is re-measured whenever a call enters an idle hub
    bool budget_fixed_ = false;           // set by the test hook
    uint64_t next_turn_ = 0, serving_ = 0;
    int next_lane_ = 0;
    hipStream_t short_[kHubShort] = {nullptr};
    Lane lanes_[kHubLanes];
};

// A whole-chunk host call between its entry and the moment its chains are handed over: the hub's leader waits for it.
struct HubTicket {
    ChainHub* hub = nullptr;
    bool counted = false;
    hipStream_t st = nullptr;
    size_t admitted = 0;
    // device_bytes: what the call is about to allocate on the device, roughly (see ChainHub::admit)
    int open(size_t device_bytes) {
        TRY(ensure_device());
        hub = ChainHub::of_device(tl_device);
        if (!hub->ok()) return fail(kDeviceError, "the streams of the chain hub could not be created");
        hub->admit(device_bytes);
        admitted = device_bytes;
        hub->announce();
        counted = true;
        st = hub->short_stream();
        tl_scope_stream = st;
        return kOk;
    }
    void arrived() { if (counted) { counted = false; hub->arrived_or_gone(); } }
    ~HubTicket() { arrived(); if (hub) hub->leave(admitted); tl_scope_stream = nullptr; }
};
// device memory of an encode / a decode of n chunks of shape d (inputs or pixels, symbols, .alc at a byte per symbol, the
// transform scratch): the figure admission works with
size_t encode_device_bytes(const ChunkDims& d, uint64_t n) { return (size_t)(n * (d.n_pixels * 3 + d.padded * 6) + forward_scratch_bytes(d)); }
size_t decode_device_bytes(const ChunkDims& d, uint64_t n, uint64_t payload) {
    return (size_t)(n * (d.n_pixels * 3 + d.padded * 3) + payload + inverse_scratch_bytes(d, false));
}

// Large device-to-host copies into the caller's pageable memory.  hipMemcpyAsync to pageable memory goes through the
// runtime's own staging at about 3 GB/s and keeps the stream busy meanwhile (110 MB of .alc: 37 ms; 64 threads' decoded
// chunks, 25 GB, over three shared streams: most of the call).  Here the copy lands in two pinned pieces of this thread,
// alternately, at the rate of the link, and the calling thread moves each piece on while the next one is in flight -- so
// the threads' CPU copies run side by side and the stream carries only the DMA.  Synchronous: returns when dst is filled.
constexpr size_t kStagePiece = size_t(32) << 20;
struct HostStage {
    void* piece[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int device = -1;
    bool ready() {
        if (piece[0] && device == tl_device) return true;
        release();
        for (int i = 0; i < 2; ++i)
            if (hipHostMalloc(&piece[i], kStagePiece, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); release(); return false; }
        device = tl_device;
        return true;
    }
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (piece[i]) (void)hipHostFree(piece[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            piece[i] = nullptr; ev[i] = nullptr;
        }
    }
    ~HostStage() { release(); }
};
thread_local HostStage tl_stage;

int copy_to_host(void* dst, const void* d_src, size_t bytes, hipStream_t st) {
    if (bytes < (size_t(4) << 20) || !tl_stage.ready()) {
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return kOk;
    }
    const size_t n = (bytes + kStagePiece - 1) / kStagePiece;
    auto len = [&](size_t k) { return std::min(kStagePiece, bytes - k * kStagePiece); };
    for (size_t k = 0; k <= n; ++k) {
        if (k < n) {
            HIP_TRY(hipMemcpyAsync(tl_stage.piece[k & 1], (const uint8_t*)d_src + k * kStagePiece, len(k), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipEventRecord(tl_stage.ev[k & 1], st));
        }
        if (k >= 1) {
            HIP_TRY(hipEventSynchronize(tl_stage.ev[(k - 1) & 1]));
            memcpy((uint8_t*)dst + (k - 1) * kStagePiece, tl_stage.piece[(k - 1) & 1], len(k - 1));
        }
    }
    return kOk;
}

// fn(i) for i in [0, n) on up to `width` helper threads bound to the calling thread's device (the copies of a many-chunk
// call: every helper moves its chunks through pinned pieces of its own, so the CPU side of the copies runs side by side).
// Returns the first failure, with its message, as this thread's error.
template <typename Fn>
int parallel_chunks(uint32_t n, uint32_t width, Fn fn) {
    if (n <= 1 || width <= 1) {
        for (uint32_t i = 0; i < n; ++i) TRY(fn(i));
        return kOk;
    }
    const uint32_t T = std::min(width, n);
    const int device = tl_device;
    std::vector<int> code(T, kOk);
    std::vector<std::string> msg(T);
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> th;
    th.reserve(T);
    for (uint32_t t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            clear_error();
            tl_device = device;
            int rc = ensure_device();
            for (uint32_t i; rc == kOk && (i = next.fetch_add(1u)) < n;) rc = fn(i);
            code[t] = rc;
            if (rc != kOk) msg[t] = tl_msg;
        });
    for (auto& t : th) t.join();
    for (uint32_t t = 0; t < T; ++t)
        if (code[t] != kOk) return fail(code[t], msg[t]);
    return kOk;
}
constexpr uint32_t kCopyThreads = 8;

// records `ready` on st, runs the job through the hub, destroys the event
int hub_run(ChainHub* hub, HubTicket* t, HubJob& job, hipStream_t st) {
    HIP_TRY(hipEventCreateWithFlags(&job.ready, hipEventDisableTiming));
    int rc = kOk;
    if (hipEventRecord(job.ready, st) != hipSuccess) rc = fail(kDeviceError, "hipEventRecord failed");
    if (t) t->arrived();
    if (rc == kOk) rc = hub->run(job);
    (void)hipEventDestroy(job.ready);
    job.ready = nullptr;
    return rc;
}
int hub_run(HubTicket& t, HubJob& job, hipStream_t st) { return hub_run(t.hub, &t, job, st); }

// Every host entry point works on a short stream of its device's hub: the library owns kHubShort + kHubLanes streams per
// device and none per calling thread.
int get_stream(hipStream_t* out) {
    TRY(ensure_device());
    ChainHub* hub = ChainHub::of_device(tl_device);
    if (!hub->ok()) return fail(kDeviceError, "the streams of the chain hub could not be created");
    *out = hub->short_stream();
    tl_scope_stream = *out;
    return kOk;
}

// The chains of a stage-level call (RansEncoder / RansDecoder objects, the one-shot and the interleaved calls): they may run
// for seconds, so they too leave with the hub's merged launches instead of sitting on a short stream.  `st`: the short
// stream the call's other work is on (the chains' inputs are complete at this point of it).
int stage_chains(hipStream_t st, std::vector<RansEncodeDesc>&& enc, std::vector<RansDecodeDesc>&& dec, uint64_t symbols_per_chain) {
    HubJob job;
    job.encode = !enc.empty();
    job.enc = std::move(enc);
    job.dec = std::move(dec);
    job.seconds = (double)symbols_per_chain * (job.encode ? 21e-9 : 39e-9);
    return hub_run(ChainHub::of_device(tl_device), nullptr, job, st);
}

// ------------------------------------------------------------------------------------------
// chunk object and .alc (de)serialisation
// ------------------------------------------------------------------------------------------

struct ChannelHeader {            // reference src/pipeline.rs:123-134
    uint32_t compressed_len = 0;
    int32_t quant_step = 1;
    int32_t quant_dead_zone = 1;
    uint32_t num_symbols = 0;
    uint32_t histogram[256] = {0};
};

}  // namespace

struct EncodedChunk {             // reference src/pipeline.rs:172-185
    uint32_t width = 0, height = 0, frames = 0;
    uint8_t wavelet = kCdf53;
    ChannelHeader ch[3];
    std::vector<uint8_t> data;    // Y || Co || Cg streams
};
struct FrameEncoder { uint8_t quality; uint8_t wavelet; };
struct Wavelet1D { int kind; };
struct FastQuantizer { uint64_t reciprocal; uint32_t shift; int32_t step; int32_t dead_zone; };
// RansEncoder (src/rans.rs:238-242: state + output vector).  Every call emits its bytes back to front, so a call's
// segment reads front to back in the order finish() wants; finish() = state bytes, then the segments newest first.
struct AliceRansEncoder { uint32_t state = alice::kRansL; std::vector<std::vector<uint8_t>> segments; uint64_t bytes = 0; };
// RansDecoder (src/rans.rs:321-326: state, input, pos)
struct AliceRansDecoder;   // (defined below DevBuf: the object keeps its device copy of the input between calls)

namespace {

inline void put_u32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
inline uint32_t get_u32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// src/pipeline.rs:200-226
void chunk_to_bytes(const EncodedChunk& c, std::vector<uint8_t>& out) {
    out.resize((size_t)kAlcHeaderBytes + c.data.size());
    uint8_t* p = out.data();
    memcpy(p, "ALCC", 4);
    p[4] = 1;
    p[5] = c.wavelet;
    put_u32(p + 6, c.width); put_u32(p + 10, c.height); put_u32(p + 14, c.frames);
    size_t off = kFixedHeaderBytes;
    for (int k = 0; k < 3; ++k) {
        put_u32(p + off, c.ch[k].compressed_len); off += 4;
        put_u32(p + off, (uint32_t)c.ch[k].quant_step); off += 4;
        put_u32(p + off, (uint32_t)c.ch[k].quant_dead_zone); off += 4;
        put_u32(p + off, c.ch[k].num_symbols); off += 4;
        for (int i = 0; i < 256; ++i) { put_u32(p + off, c.ch[k].histogram[i]); off += 4; }
    }
    if (!c.data.empty()) memcpy(p + off, c.data.data(), c.data.size());
}

// header part of src/pipeline.rs:235-313; *payload_len = sum of compressed_len
int parse_alc_header(const uint8_t* data, uint64_t len, EncodedChunk& c, uint64_t* payload_len) {
    if (len < kAlcHeaderBytes)
        return fail(kInvalidBitstream, "data too short: " + std::to_string(len) + " bytes (minimum 3138)");
    if (memcmp(data, "ALCC", 4) != 0) return fail(kInvalidBitstream, "bad magic (expected ALCC)");
    if (data[4] != 1) return fail(kInvalidBitstream, "unsupported version: " + std::to_string((int)data[4]) + " (expected 1)");
    if (data[5] > 2) return fail(kInvalidBitstream, "unknown wavelet type byte: " + std::to_string((int)data[5]));
    c.wavelet = data[5];
    c.width = get_u32(data + 6); c.height = get_u32(data + 10); c.frames = get_u32(data + 14);
    size_t off = kFixedHeaderBytes;
    uint64_t total = 0;
    for (int k = 0; k < 3; ++k) {
        c.ch[k].compressed_len = get_u32(data + off); off += 4;
        c.ch[k].quant_step = (int32_t)get_u32(data + off); off += 4;
        c.ch[k].quant_dead_zone = (int32_t)get_u32(data + off); off += 4;
        c.ch[k].num_symbols = get_u32(data + off); off += 4;
        for (int i = 0; i < 256; ++i) { c.ch[k].histogram[i] = get_u32(data + off); off += 4; }
        total += c.ch[k].compressed_len;
    }
    *payload_len = total;
    return kOk;
}

int chunk_from_bytes(const uint8_t* data, uint64_t len, EncodedChunk& c) {
    uint64_t total = 0;
    TRY(parse_alc_header(data, len, c, &total));
    if (len < (uint64_t)kAlcHeaderBytes + total)
        return fail(kInvalidBitstream, "truncated payload: need " + std::to_string(kAlcHeaderBytes + total - len) + " more bytes");
    c.data.assign(data + kAlcHeaderBytes, data + kAlcHeaderBytes + total);
    return kOk;
}

// src/pipeline.rs:67-71
int checked_pixel_count(uint64_t w, uint64_t h, uint64_t f, uint64_t* out) {
    unsigned __int128 v = (unsigned __int128)w * h;
    if (v > UINT64_MAX) return fail(kDimensionOverflow, "dimensions overflow usize");
    v *= f;
    if (v > UINT64_MAX) return fail(kDimensionOverflow, "dimensions overflow usize");
    *out = (uint64_t)v;
    return kOk;
}

// ------------------------------------------------------------------------------------------
// encode / decode on device buffers
// ------------------------------------------------------------------------------------------

inline uint64_t round_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// Rigorous magnitude bound through the inverse 3-D lifting.  Symbols are u8, so |q| <= 128 and every
// dequantised sample is at most 128 * |step|; each lifting step adds at most (2 * other * |c| + 4096) / 8192 + 1.
// fast: 32-bit (24 x 24-bit) products are exact everywhere.  mid16: additionally every value after the
// temporal pass fits i16, so the intermediate can be stored in 16 bits.
struct InverseBounds { bool fast; bool mid16; bool lds16; };
InverseBounds inverse_bounds(int wavelet, const int32_t step[3]) {
    const LiftSteps ls = lift_steps(wavelet);
    long double worst = 0;
    for (int c = 0; c < 3; ++c) {
        long double a = 128.0L * fabsl((long double)step[c]);
        if (a > worst) worst = a;
    }
    InverseBounds r{true, false, false};
    long double m = worst;  // bound on every sample
    for (int pass = 0; pass < 3; ++pass) {
        long double me = m, mo = m;
        for (int k = ls.n - 1; k >= 0; --k) {
            const long double cabs = fabsl((long double)ls.coeff[k]);
            long double& target = (k & 1) == 0 ? mo : me;
            const long double other = (k & 1) == 0 ? me : mo;
            // v_mul_i24 operands: |a + b| < 2^23; product + rounding below 2^31
            if (2 * other >= 8388607.0L || 2 * other * cabs + 4096 >= 2147483647.0L) { r.fast = false; return r; }
            target = target + (2 * other * cabs + 4096) / 8192 + 1;
            if (target >= 1073741824.0L) { r.fast = false; return r; }
        }
        m = me > mo ? me : mo;
        if (pass == 0) r.mid16 = m <= 32767.0L;
        if (pass == 1) r.lds16 = r.mid16 && m <= 32767.0L;
    }
    return r;
}

// (re)allocates only when the buffer is too small; the caller guarantees nothing in flight still uses it
int ensure_bytes(DevBuf& b, size_t bytes) {
    if (b.p && b.n >= bytes) return kOk;
    return b.alloc(bytes);
}

struct EncodeWork {
    ChunkDims d{};
    int n_chunks = 0;
    uint64_t cap[3] = {0, 0, 0}, alc_stride = 0;   // stream regions of the Y, Co, Cg chains of a chunk
    DevBuf scratch;   // band slots of the tile path (forward_scratch_bytes; a batch sizes it for its decode side too)
    DevBuf gen, tmp, sym, hist, tables, results, alc, sizes, planes;   // gen / tmp / planes: generic path only
};

int encode_work_alloc(EncodeWork& w, const ChunkDims& d, int n_chunks, size_t scratch_bytes = 0) {
    w.d = d; w.n_chunks = n_chunks; w.cap[0] = w.cap[1] = w.cap[2] = 0; w.alc_stride = 0;
    if (transform_tiles_eligible(d)) TRY(w.scratch.alloc(std::max(forward_scratch_bytes(d), scratch_bytes)));
    TRY(w.sym.alloc((size_t)n_chunks * 3 * d.padded));
    TRY(w.hist.alloc((size_t)n_chunks * 3 * 256 * sizeof(uint32_t)));
    TRY(w.tables.alloc((size_t)n_chunks * 3 * sizeof(RansTable)));
    TRY(w.results.alloc((size_t)n_chunks * 3 * sizeof(RansResult)));
    TRY(w.sizes.alloc((size_t)n_chunks * sizeof(unsigned long long)));
    return kOk;
}

// .alc buffers for the capacities of the three chains of a chunk (re-allocated only when one grows).  A chunk's buffer is
// [kStreamHead bytes][region Y][region Co][region Cg]: the chains write their streams at the tails of their regions and
// the compaction moves them, in place, behind the header at the front.
int encode_work_set_cap(EncodeWork& w, const uint64_t cap[3]) {
    if (w.alc.p && cap[0] <= w.cap[0] && cap[1] <= w.cap[1] && cap[2] <= w.cap[2]) return kOk;
    w.alc.reset();
    for (int c = 0; c < 3; ++c) w.cap[c] = std::max(w.cap[c], cap[c]);
    w.alc_stride = round_up(kStreamHead + w.cap[0] + w.cap[1] + w.cap[2], 256);
    TRY(w.alc.alloc((size_t)w.n_chunks * w.alc_stride + 256));  // slack: the compaction copy reads whole dwords
    return kOk;
}

// Upper bound of a chain's stream length from its histogram.  The table is the reference's
// (src/rans.rs:102-150); a symbol of frequency f costs log2(4096 / f) bits, the floor in x / f loses less
// than log2(1 + 2^-11) bits per symbol (the state is at least f * 2^11 when it is divided), and the final
// state adds 4 bytes.  A margin on top keeps this a capacity, not a prediction; the kernel still checks.
uint64_t estimate_stream_cap(const uint32_t* hist, uint64_t n) {
    unsigned long long total = 0;
    for (int i = 0; i < 256; ++i) total += hist[i];
    if (total == 0) return 4096;
    long double bits = 0;
    unsigned nt = 0;
    unsigned freq[256];
    for (int i = 0; i < 256; ++i) {
        unsigned f = hist[i] == 0 ? 1u : (unsigned)std::max<unsigned long long>((unsigned long long)hist[i] * kProbScale / total, 1ull);
        freq[i] = f;
        nt += f;
    }
    if (nt != kProbScale) freq[255] = (unsigned)((int)freq[255] + ((int)kProbScale - (int)nt)) & 0xFFFFu;
    for (int i = 0; i < 256; ++i) {
        if (!hist[i]) continue;
        const unsigned f = freq[i];
        if (f == 0) return 0;                                   // reported by the kernel as a divergence
        if (f < kProbScale) bits += (long double)hist[i] * log2l((long double)kProbScale / (long double)f);
    }
    const long double bytes = bits / 8.0L;
    return (uint64_t)(bytes * 1.002L) + n / 4096 + 4096 + 64;
}

uint64_t worst_cap(const ChunkDims& d) { return round_up(2 * d.padded + 4 + 64 + 64, 256); }  // +64: dummy-store guard band

// Exact reference arithmetic on caller-shaped data for chunks of more than 64 padded frames.
int forward_generic(const uint8_t* d_rgb, const ChunkDims& d, int wavelet, int32_t step, EncodeWork& w,
                    uint8_t* d_sym, uint32_t* d_hist, hipStream_t st) {
    if (!w.planes.p) TRY(w.planes.alloc(3 * d.n_pixels * sizeof(int16_t)));
    if (!w.tmp.p) TRY(w.tmp.alloc(d.padded * sizeof(int32_t)));
    if (!w.gen.p) TRY(w.gen.alloc(2 * d.padded * sizeof(int32_t)));
    int16_t* pl = w.planes.as<int16_t>();
    launch_rgb_to_ycocg(d_rgb, d.n_pixels, pl, pl + d.n_pixels, pl + 2 * d.n_pixels, st);
    int32_t* vol = w.gen.as<int32_t>();
    int32_t* qb = vol + d.padded;
    const uint64_t W = d.pw, H = d.ph, D = d.pf;
    for (int c = 0; c < 3; ++c) {
        launch_pad_channel(pl + (size_t)c * d.n_pixels, d, vol, st);
        launch_wavelet_axis(vol, w.tmp.as<int32_t>(), W, 1, D * H, W, 1, 0, wavelet, false, st);
        launch_wavelet_axis(vol, w.tmp.as<int32_t>(), H, W, D, W * H, W, 1, wavelet, false, st);
        launch_wavelet_axis(vol, w.tmp.as<int32_t>(), D, W * H, 1, 0, W * H, 1, wavelet, false, st);
        launch_quantize(vol, qb, d.padded, step, step, st);
        launch_to_symbols(qb, d_sym + (size_t)c * d.padded, d.padded, st);
        launch_histogram(d_sym + (size_t)c * d.padded, d.padded, d_hist + c * 256, st);
    }
    return kOk;
}

thread_local uint64_t tl_test_first_cap = 0;   // alice_codec_test_force_first_cap
thread_local uint32_t tl_dec_stats[4] = {0, 0, 0, 0};   // last single-chain decode of this thread: fast tiles, exact tiles, path mask, bytes consumed
void note_decode_stats(const RansResult& r) {
    tl_dec_stats[0] = r.fast_tiles; tl_dec_stats[1] = r.slow_tiles; tl_dec_stats[2] = r.paths;
    tl_dec_stats[3] = (uint32_t)(r.len > 0xFFFFFFFFull ? 0xFFFFFFFFull : r.len);
}

struct StageEvents {
    hipEvent_t ev[8] = {nullptr};
    bool ready = false;
    int init() {
        if (ready) return kOk;
        for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
        ready = true;
        return kOk;
    }
    ~StageEvents() { if (ready) for (auto& e : ev) (void)hipEventDestroy(e); }
};

// Encode of n_chunks chunks on `st`; results stay on the device.  How the stream regions are sized:
//   kCapReuse    the work area already owns .alc buffers (a batch after its first encode): keep their capacities and
//                queue everything without touching the host -- the chains check their regions, and a chunk that outgrew
//                them (new content) comes back as an overflow, which the caller answers with kCapEstimate;
//   kCapEstimate one host round trip between the transforms and the chains reads the histograms and derives a capacity
//                per channel (estimate_stream_cap); also what kCapReuse falls back to when there are no buffers yet;
//   kCapWorst    2 bytes per symbol, the bound of the format (the last resort: 3 x 2 x padded bytes per chunk).
enum CapMode { kCapReuse = 0, kCapEstimate = 1, kCapWorst = 2 };
int encode_launch(const uint8_t* d_rgb, EncodeWork& w, uint8_t quality, int wavelet, hipStream_t st,
                  StageEvents* evs, CapMode mode = kCapReuse, HubTicket* hub = nullptr) {
    const ChunkDims& d = w.d;
    const int32_t step = quality_to_step(quality);
    const int B = w.n_chunks;
    HIP_TRY(hipMemsetAsync(w.hist.p, 0, (size_t)B * 3 * 256 * sizeof(uint32_t), st));
    if (evs) HIP_TRY(hipEventRecord(evs->ev[0], st));
    for (int b = 0; b < B; ++b) {
        const uint8_t* rgb = d_rgb + (size_t)b * d.n_pixels * 3;
        uint8_t* sym = w.sym.as<uint8_t>() + (size_t)b * 3 * d.padded;
        uint32_t* hist = w.hist.as<uint32_t>() + (size_t)b * 3 * 256;
        if (!w.scratch.p || !launch_forward_transform(rgb, d, wavelet, step, w.scratch.p, sym, hist, st))
            TRY(forward_generic(rgb, d, wavelet, step, w, sym, hist, st));
    }
    if (evs) HIP_TRY(hipEventRecord(evs->ev[1], st));
    const bool had_alc = w.alc.p != nullptr;
    uint64_t cap[3] = {w.cap[0], w.cap[1], w.cap[2]};
    if (mode == kCapWorst) {
        cap[0] = cap[1] = cap[2] = worst_cap(d);
    } else if (mode == kCapEstimate || !had_alc) {
        cap[0] = cap[1] = cap[2] = 0;
        std::vector<uint32_t> hist((size_t)B * 3 * 256);
        HIP_TRY(hipMemcpyAsync(hist.data(), w.hist.p, hist.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // one capacity per channel (the largest over the chunks): Y streams are about twice as long as Co / Cg streams
        for (int c = 0; c < 3 * B; ++c) cap[c % 3] = std::max(cap[c % 3], estimate_stream_cap(&hist[(size_t)c * 256], d.padded));
        for (int c = 0; c < 3; ++c) cap[c] = std::min(round_up(cap[c], 256), worst_cap(d));
        // test-only override (alice_codec_test_force_first_cap): pretend the first estimate was far too small, to
        // exercise the overflow-and-retry path
        if (mode == kCapReuse && !had_alc)
            if (const uint64_t forced = tl_test_first_cap) cap[0] = cap[1] = cap[2] = forced;
    }
    TRY(encode_work_set_cap(w, cap));
    launch_rans_table(w.hist.as<uint32_t>(), w.tables.as<RansTable>(), 3 * B, st);
    if (evs) HIP_TRY(hipEventRecord(evs->ev[2], st));
    if (hub) {
        // host call: the chains leave with the hub's next merged launch (same regions as the grouped layout below); this
        // thread sleeps until they have run and then queues the tail on its short stream
        HubJob job;
        job.encode = true;
        job.enc.resize((size_t)3 * B);
        for (int c = 0; c < 3 * B; ++c) {
            RansEncodeDesc& e = job.enc[(size_t)c];
            const int g = c % 3;
            e.sym = w.sym.as<uint8_t>() + (size_t)c * d.padded;
            e.n = d.padded;
            e.table = w.tables.as<RansTable>() + c;
            e.region = w.alc.as<uint8_t>() + (size_t)(c / 3) * w.alc_stride + kStreamHead + (g == 0 ? 0ull : (g == 1 ? w.cap[0] : w.cap[0] + w.cap[1]));
            e.cap = w.cap[g];
            e.result = w.results.as<RansResult>() + c;
            e.x_init = kRansL;
            e.keep_open = 0u;
        }
        job.seconds = (double)d.padded * 21e-9;
        TRY(hub_run(*hub, job, st));
    } else {
        launch_rans_encode(w.sym.as<uint8_t>(), d.padded, d.padded, w.tables.as<RansTable>(), w.alc.as<uint8_t>(),
                           w.cap[0], w.results.as<RansResult>(), 3 * B, st, w.alc_stride, kStreamHead, 0xFFFFFFFFu, w.cap[1], w.cap[2]);
    }
    if (evs) HIP_TRY(hipEventRecord(evs->ev[3], st));
    launch_write_headers(w.alc.as<uint8_t>(), w.alc_stride, d, wavelet, step, w.hist.as<uint32_t>(),
                         w.results.as<RansResult>(), w.sizes.as<unsigned long long>(), B, st);
    launch_compact_streams(w.alc.as<uint8_t>(), w.alc_stride, kStreamHead, w.cap, w.results.as<RansResult>(), B, st);
    if (evs) HIP_TRY(hipEventRecord(evs->ev[4], st));
    HIP_TRY(hipGetLastError());
    return kOk;
}

// After the stream has drained: fetch per-chain results, map flags to errors.
int encode_collect(EncodeWork& w, hipStream_t st, std::vector<RansResult>& res) {
    res.resize((size_t)w.n_chunks * 3);
    HIP_TRY(hipMemcpyAsync(res.data(), w.results.p, res.size() * sizeof(RansResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (auto& r : res) {
        if (r.flags & kTableDiverges)
            return fail(kReferenceDiverges, "a symbol whose table frequency wrapped to 0 is present: the reference encoder does not terminate on this input");
        if (r.flags & kRansInternal) return fail(kInternal, "rANS kernel invariant violated");
    }
    if (getenv("ALICE_CODEC_DEBUG"))
        for (size_t i = 0; i < res.size(); ++i)
            fprintf(stderr, "[alice] encode chain %zu: %llu bytes, %.1f Mcycles, %.1f ms of 100 MHz ticks => %.2f GHz, xcc %u se %u cu %u simd %u\n", i, res[i].len,
                    res[i].cycles_k * 1024.0 / 1e6, res[i].ticks_k * 1024.0 / 1e5,
                    res[i].ticks_k ? (res[i].cycles_k / (double)res[i].ticks_k) * 0.1 : 0.0,
                    res[i].xcc_id & 15u, (res[i].hw_id >> 13) & 7u, (res[i].hw_id >> 8) & 15u, (res[i].hw_id >> 4) & 3u);
    for (auto& r : res)
        if (r.flags & kRansOverflow) return -1;  // caller retries with the worst-case capacity
    return kOk;
}

struct DecodeWork {
    ChunkDims d{};
    int n_chunks = 0;
    DevBuf scratch_own, gen, tmp, sym, hist, tables, descs, results, planes;
    uint8_t* sym_ptr = nullptr;   // decoded symbols: own buffer, or one lent by the caller
    DevBuf* scratch = nullptr;    // band slots of the tile path: own buffer, or the one a batch shares with its encode side
};

// sym_ext / scratch_ext: buffers the caller lends (a batch reuses its encode-side symbol and scratch buffers,
// which are dead once the encode has finished)
int decode_work_alloc(DecodeWork& w, const ChunkDims& d, int n_chunks, uint8_t* sym_ext = nullptr, DevBuf* scratch_ext = nullptr) {
    w.d = d; w.n_chunks = n_chunks;
    w.scratch = scratch_ext ? scratch_ext : &w.scratch_own;
    if (sym_ext) w.sym_ptr = sym_ext;
    else { TRY(w.sym.alloc((size_t)n_chunks * 3 * d.padded)); w.sym_ptr = w.sym.as<uint8_t>(); }
    TRY(w.hist.alloc((size_t)n_chunks * 3 * 256 * sizeof(uint32_t)));
    TRY(w.tables.alloc((size_t)n_chunks * 3 * sizeof(RansTable)));
    TRY(w.descs.alloc((size_t)n_chunks * 3 * sizeof(RansDecodeDesc)));
    TRY(w.results.alloc((size_t)n_chunks * 3 * sizeof(RansResult)));
    return kOk;
}

int inverse_generic(const uint8_t* d_sym, const ChunkDims& d, int wavelet, const int32_t step[3], DecodeWork& w,
                    uint8_t* d_rgb, hipStream_t st) {
    if (!w.planes.p) TRY(w.planes.alloc(3 * d.n_pixels * sizeof(int16_t)));
    if (!w.tmp.p) TRY(w.tmp.alloc(d.padded * sizeof(int32_t)));
    if (!w.gen.p) TRY(w.gen.alloc(2 * d.padded * sizeof(int32_t)));
    int16_t* pl = w.planes.as<int16_t>();
    int32_t* qb = w.gen.as<int32_t>();
    int32_t* vol = qb + d.padded;
    const uint64_t W = d.pw, H = d.ph, D = d.pf;
    for (int c = 0; c < 3; ++c) {
        launch_from_symbols(d_sym + (size_t)c * d.padded, qb, d.padded, st);
        launch_dequantize(qb, vol, d.padded, step[c], st);
        launch_wavelet_axis(vol, w.tmp.as<int32_t>(), D, W * H, 1, 0, W * H, 1, wavelet, true, st);
        launch_wavelet_axis(vol, w.tmp.as<int32_t>(), H, W, D, W * H, W, 1, wavelet, true, st);
        launch_wavelet_axis(vol, w.tmp.as<int32_t>(), W, 1, D * H, W, 1, 0, wavelet, true, st);
        launch_strip_channel(vol, d, pl + (size_t)c * d.n_pixels, st);
    }
    launch_ycocg_to_rgb(pl, pl + d.n_pixels, pl + 2 * d.n_pixels, d.n_pixels, d_rgb, st);
    return kOk;
}

// headers[b]: parsed chunk headers (validated); d_payload[b]: device pointer to chunk b's payload;
// d_rgb[b]: where chunk b's pixels go
int decode_launch(const std::vector<EncodedChunk>& headers, const std::vector<const uint8_t*>& d_payload,
                  DecodeWork& w, const std::vector<uint8_t*>& d_rgb, hipStream_t st, StageEvents* evs, HubTicket* hub = nullptr) {
    const ChunkDims& d = w.d;
    const int B = w.n_chunks;
    if (transform_tiles_eligible(d)) {
        // band slots: i16 when every chunk's bound allows it, else i32 (sized before anything is queued: the chunks of
        // a batch share the ring, and growing it must not wait for the chains)
        bool all16 = true;
        for (int b = 0; b < B; ++b) {
            int32_t step[3] = {headers[b].ch[0].quant_step, headers[b].ch[1].quant_step, headers[b].ch[2].quant_step};
            const InverseBounds ib = inverse_bounds(headers[b].wavelet, step);
            all16 = all16 && ib.fast && ib.mid16;
        }
        const size_t need = inverse_scratch_bytes(d, all16);
        if (!w.scratch->p || w.scratch->n < need) {
            HIP_TRY(hipStreamSynchronize(st));   // an earlier call's tail may still use the old ring
            TRY(ensure_bytes(*w.scratch, need));
        }
    }
    std::vector<uint32_t> hist((size_t)B * 3 * 256);
    std::vector<RansDecodeDesc> descs((size_t)B * 3);
    for (int b = 0; b < B; ++b) {
        uint64_t off = 0;
        for (int c = 0; c < 3; ++c) {
            const ChannelHeader& h = headers[b].ch[c];
            memcpy(&hist[((size_t)b * 3 + c) * 256], h.histogram, 256 * sizeof(uint32_t));
            RansDecodeDesc& ds = descs[(size_t)b * 3 + c];
            ds.in = d_payload[b] + off;
            ds.in_len = h.compressed_len;
            ds.out = w.sym_ptr + ((size_t)b * 3 + c) * d.padded;
            ds.n = d.padded;
            ds.table = w.tables.as<RansTable>() + ((size_t)b * 3 + c);
            off += h.compressed_len;
        }
    }
    HIP_TRY(hipMemcpyAsync(w.hist.p, hist.data(), hist.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (!hub) HIP_TRY(hipMemcpyAsync(w.descs.p, descs.data(), descs.size() * sizeof(RansDecodeDesc), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // host vectors go out of scope
    if (evs) HIP_TRY(hipEventRecord(evs->ev[5], st));
    launch_rans_table(w.hist.as<uint32_t>(), w.tables.as<RansTable>(), 3 * B, st);
    if (hub) {   // host call: see encode_launch
        HubJob job;
        job.encode = false;
        job.dec = std::move(descs);
        for (size_t c = 0; c < job.dec.size(); ++c) job.dec[c].result = w.results.as<RansResult>() + c;
        job.seconds = (double)d.padded * 39e-9;
        TRY(hub_run(*hub, job, st));
    } else {
        launch_rans_decode(w.descs.as<RansDecodeDesc>(), w.results.as<RansResult>(), 3 * B, st);
    }
    if (evs) HIP_TRY(hipEventRecord(evs->ev[6], st));
    {
        const bool tiles = transform_tiles_eligible(d) && w.scratch->p;
        for (int b = 0; b < B; ++b) {
            int32_t step[3] = {headers[b].ch[0].quant_step, headers[b].ch[1].quant_step, headers[b].ch[2].quant_step};
            const InverseBounds ib = inverse_bounds(headers[b].wavelet, step);
            const uint8_t* sym = w.sym_ptr + (size_t)b * 3 * d.padded;
            if (!tiles || !launch_inverse_transform(sym, d, headers[b].wavelet, step, !ib.fast, ib.fast && ib.mid16, ib.fast && ib.lds16, w.scratch->p, d_rgb[b], st))
                TRY(inverse_generic(sym, d, headers[b].wavelet, step, w, d_rgb[b], st));
        }
    }
    if (evs) HIP_TRY(hipEventRecord(evs->ev[7], st));
    HIP_TRY(hipGetLastError());
    return kOk;
}

int decode_collect(DecodeWork& w, hipStream_t st) {
    std::vector<RansResult> res((size_t)w.n_chunks * 3);
    HIP_TRY(hipMemcpyAsync(res.data(), w.results.p, res.size() * sizeof(RansResult), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (auto& r : res)
        if (r.flags & kRansInternal) return fail(kInternal, "rANS decode table invariant violated");
    if (getenv("ALICE_CODEC_DEBUG"))
        for (size_t i = 0; i < res.size(); ++i)
            fprintf(stderr, "[alice] decode chain %zu: consumed %llu bytes, fast tiles %u, slow tiles %u, %.1f Mcycles, xcc %u se %u cu %u simd %u\n", i, res[i].len,
                    res[i].fast_tiles, res[i].slow_tiles, res[i].cycles_k * 1024.0 / 1e6,
                    res[i].xcc_id & 15u, (res[i].hw_id >> 13) & 7u, (res[i].hw_id >> 8) & 15u, (res[i].hw_id >> 4) & 3u);
    return kOk;
}

// src/pipeline.rs:537-579 validation, in the reference's order
int validate_for_decode(const EncodedChunk& c, ChunkDims* dims, uint64_t payload_len) {
    uint64_t n_pixels = 0;
    TRY(checked_pixel_count(c.width, c.height, c.frames, &n_pixels));
    *dims = make_dims(c.width, c.height, c.frames);
    if (n_pixels == 0) return kOk;
    uint64_t off = 0;
    for (int k = 0; k < 3; ++k) {
        if ((uint64_t)c.ch[k].num_symbols != dims->padded)
            return fail(kInvalidBitstream, "channel " + std::to_string(k) + ": num_symbols " + std::to_string(c.ch[k].num_symbols) +
                                               " != padded_pixels " + std::to_string(dims->padded));
        if (off + c.ch[k].compressed_len > payload_len)
            return fail(kInvalidBitstream, "channel " + std::to_string(k) + ": compressed data overrun");
        off += c.ch[k].compressed_len;
    }
    return kOk;
}

// FrameEncoder::encode on host buffers (src/pipeline.rs:377-507)
int encode_host(const FrameEncoder& enc, const uint8_t* rgb, uint64_t rgb_len, uint32_t width, uint32_t height,
                uint32_t frames, EncodedChunk& out) {
    uint64_t n_pixels = 0;
    TRY(checked_pixel_count(width, height, frames, &n_pixels));                 // :388
    out.width = width; out.height = height; out.frames = frames; out.wavelet = enc.wavelet;
    for (auto& h : out.ch) h = ChannelHeader();
    out.data.clear();
    if (n_pixels == 0) {                                                         // :391-412
        if (rgb_len != 0) return fail(kInvalidBufferSize, "buffer size mismatch: expected 0, got " + std::to_string(rgb_len));
        return kOk;
    }
    if (width == 0 || height == 0) return fail(kInvalidDimensions, "invalid dimensions");  // :415-417
    if (n_pixels > UINT64_MAX / 3) return fail(kDimensionOverflow, "dimensions overflow usize");
    if (rgb_len != n_pixels * 3)                                                // :422-427
        return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n_pixels * 3) + ", got " + std::to_string(rgb_len));
    const ChunkDims d = make_dims(width, height, frames);
    if (d.padded > 0xFFFFFFFFull) return fail(kDimensionOverflow, "padded pixel count does not fit the header's u32 num_symbols");

    HubTicket ticket;
    TRY(ticket.open(encode_device_bytes(d, 1)));
    const hipStream_t st = ticket.st;
    DevBuf d_rgb;
    TRY(d_rgb.alloc(rgb_len));
    HIP_TRY(hipMemcpyAsync(d_rgb.p, rgb, rgb_len, hipMemcpyHostToDevice, st));
    EncodeWork w;
    std::vector<RansResult> res;
    TRY(encode_work_alloc(w, d, 1));
    for (int attempt = 0;; ++attempt) {
        TRY(encode_launch(d_rgb.as<uint8_t>(), w, enc.quality, enc.wavelet, st, nullptr, (CapMode)attempt, &ticket));
        int rc = encode_collect(w, st, res);
        if (rc == kOk) break;
        if (rc != -1 || attempt >= 2) return rc == -1 ? fail(kInternal, "rANS output exceeded the worst-case bound") : rc;
    }
    uint64_t payload = res[0].len + res[1].len + res[2].len;
    // the header into a small buffer, the payload straight into the chunk object (one allocation, one pass over it)
    std::vector<uint8_t> hdr((size_t)kAlcHeaderBytes);
    HIP_TRY(hipMemcpyAsync(hdr.data(), w.alc.p, hdr.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint64_t tot = 0;
    TRY(parse_alc_header(hdr.data(), hdr.size() + payload, out, &tot));
    if (tot != payload) return fail(kInternal, "device header/payload length mismatch");
    out.data.resize((size_t)payload);
    TRY(copy_to_host(out.data.data(), w.alc.as<uint8_t>() + kAlcHeaderBytes, (size_t)payload, st));
    return kOk;
}

// FrameDecoder::decode on a host chunk (src/pipeline.rs:537-624)
// A buffer the C ABI hands to the caller (released with free(): alice_codec_data_free64).  Large ones are 2 MiB aligned and
// marked for huge pages: 64 threads that each fault a fresh 398 MB result in 4 KiB pages spend more time in the kernel's
// page-fault path than the GPU spends on their chains.
uint8_t* host_result_alloc(uint64_t bytes) {
    constexpr size_t kHuge = size_t(2) << 20;
    if (bytes < 2 * kHuge) return (uint8_t*)malloc(bytes ? (size_t)bytes : 1);
    const size_t len = round_up(bytes, kHuge);
    void* p = aligned_alloc(kHuge, len);
    if (p) (void)madvise(p, len, MADV_HUGEPAGE);
    return (uint8_t*)p;
}

// *out: malloc'ed pixels (nullptr for an empty chunk), *out_len their count
int decode_host(const EncodedChunk& c, uint8_t** out, uint64_t* out_len) {
    ChunkDims d;
    TRY(validate_for_decode(c, &d, c.data.size()));
    *out = nullptr; *out_len = 0;
    if (d.n_pixels == 0) { *out = host_result_alloc(0); return *out ? kOk : fail(kOutOfMemory, "out of host memory"); }
    HubTicket ticket;
    TRY(ticket.open(decode_device_bytes(d, 1, c.data.size())));
    const hipStream_t st = ticket.st;
    DevBuf d_payload, d_rgb;
    TRY(d_payload.alloc(c.data.size() + 16));
    TRY(d_rgb.alloc(d.n_pixels * 3));
    if (!c.data.empty())
        HIP_TRY(hipMemcpyAsync(d_payload.p, c.data.data(), c.data.size(), hipMemcpyHostToDevice, st));
    DecodeWork w;
    TRY(decode_work_alloc(w, d, 1));
    std::vector<EncodedChunk> hdrs(1);
    hdrs[0].width = c.width; hdrs[0].height = c.height; hdrs[0].frames = c.frames; hdrs[0].wavelet = c.wavelet;
    for (int k = 0; k < 3; ++k) hdrs[0].ch[k] = c.ch[k];
    std::vector<const uint8_t*> pay(1, d_payload.as<uint8_t>());
    TRY(decode_launch(hdrs, pay, w, std::vector<uint8_t*>(1, d_rgb.as<uint8_t>()), st, nullptr, &ticket));
    TRY(decode_collect(w, st));
    uint8_t* rgb = host_result_alloc(d.n_pixels * 3);
    if (!rgb) return fail(kOutOfMemory, "out of host memory");
    const int rc = copy_to_host(rgb, d_rgb.p, d.n_pixels * 3, st);
    if (rc != kOk) { free(rgb); return rc; }
    *out = rgb; *out_len = d.n_pixels * 3;
    return kOk;
}

uint8_t* to_c_buffer(const std::vector<uint8_t>& v) {
    uint8_t* p = (uint8_t*)malloc(v.size() ? v.size() : 1);
    if (p && !v.empty()) memcpy(p, v.data(), v.size());
    return p;
}

// generic staged run: copy in, run fn on device pointers, copy out
template <typename Tin, typename Tout, typename Fn>
int staged(const Tin* in, uint64_t n_in, Tout* out, uint64_t n_out, Fn fn) {
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf a, b;
    TRY(a.alloc(n_in * sizeof(Tin)));
    TRY(b.alloc(n_out * sizeof(Tout)));
    if (n_in) HIP_TRY(hipMemcpyAsync(a.p, in, n_in * sizeof(Tin), hipMemcpyHostToDevice, st));
    fn(a.as<Tin>(), b.as<Tout>(), st);
    HIP_TRY(hipGetLastError());
    if (n_out) HIP_TRY(hipMemcpyAsync(out, b.p, n_out * sizeof(Tout), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}

// Wavelet1D / 2D / 3D of a device-resident volume, result in `v` (tmp: same size).  Volumes the tile kernels cover run
// their exact instances (two passes over the data); the rest -- odd lengths, tiny sizes, 1-D signals -- runs the
// per-axis kernels of generic.hip.
void wavelet_on_device(int kind, int32_t* v, int32_t* tmp, uint64_t W, uint64_t H, uint64_t D, int ndim, bool inverse, hipStream_t st) {
    if (stage_tiles_eligible(W, H, D, ndim)) { launch_stage_wavelet(v, tmp, W, H, D, ndim, kind, inverse, st); return; }
    if (!inverse) {
        launch_wavelet_axis(v, tmp, W, 1, D * H, W, 1, 0, kind, false, st);
        if (ndim >= 2) launch_wavelet_axis(v, tmp, H, W, D, W * H, W, 1, kind, false, st);
        if (ndim >= 3) launch_wavelet_axis(v, tmp, D, W * H, 1, 0, W * H, 1, kind, false, st);
    } else {
        if (ndim >= 3) launch_wavelet_axis(v, tmp, D, W * H, 1, 0, W * H, 1, kind, true, st);
        if (ndim >= 2) launch_wavelet_axis(v, tmp, H, W, D, W * H, W, 1, kind, true, st);
        launch_wavelet_axis(v, tmp, W, 1, D * H, W, 1, 0, kind, true, st);
    }
}

int wavelet_nd(int kind, int32_t* data, uint64_t W, uint64_t H, uint64_t D, int ndim, bool inverse) {
    if (!data) return fail(kNullArgument, "null data");
    if (kind < 0 || kind > 2) return fail(kInvalidBitstream, "unknown wavelet type");
    unsigned __int128 tot = (unsigned __int128)W * H * D;
    if (tot > ((unsigned __int128)1 << 40)) return fail(kDimensionOverflow, "volume too large");
    const uint64_t n = (uint64_t)tot;
    if (n == 0) return kOk;
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf a, t;
    TRY(a.alloc(n * 4));
    TRY(t.alloc(n * 4));
    HIP_TRY(hipMemcpyAsync(a.p, data, n * 4, hipMemcpyHostToDevice, st));
    wavelet_on_device(kind, a.as<int32_t>(), t.as<int32_t>(), W, H, D, ndim, inverse, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(data, a.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------------

struct AliceBatch {
    ChunkDims d{};
    uint32_t n_chunks = 0;
    uint8_t quality = 0, wavelet = 0;
    int device = 0;
    EncodeWork enc;
    DecodeWork dec;
    DevBuf spare;             // in-place decode: the pixels of chunk 0 (chunk i > 0 lands on the symbols of chunk i - 1)
    std::vector<uint8_t*> rgb_dst;   // where the last decode put each chunk's pixels
    bool dec_ready = false;
    StageEvents evs;
    hipStream_t enc_stream = nullptr, dec_stream = nullptr;
    const uint8_t* last_rgb = nullptr;
    bool enc_timed = false, dec_timed = false;
    float stage_ms[6] = {0, 0, 0, 0, 0, 0};
};

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------

extern "C" {

// ---- PART 1: reference ABI ----

Wavelet1D* alice_codec_wavelet1d_haar(void) { return new (std::nothrow) Wavelet1D{kHaar}; }
Wavelet1D* alice_codec_wavelet1d_cdf53(void) { return new (std::nothrow) Wavelet1D{kCdf53}; }
Wavelet1D* alice_codec_wavelet1d_cdf97(void) { return new (std::nothrow) Wavelet1D{kCdf97}; }
void alice_codec_wavelet1d_destroy(Wavelet1D* ptr) { delete ptr; }

void alice_codec_wavelet1d_forward(const Wavelet1D* w, int32_t* data, uint32_t len) {
    clear_error();
    if (!w || !data || len < 2) return;
    (void)wavelet_nd(w->kind, data, len, 1, 1, 1, false);
}
void alice_codec_wavelet1d_inverse(const Wavelet1D* w, int32_t* data, uint32_t len) {
    clear_error();
    if (!w || !data || len < 2) return;
    (void)wavelet_nd(w->kind, data, len, 1, 1, 1, true);
}

FrameEncoder* alice_codec_encoder_create(uint8_t quality) { return new (std::nothrow) FrameEncoder{quality, (uint8_t)kCdf53}; }
void alice_codec_encoder_destroy(FrameEncoder* ptr) { delete ptr; }

EncodedChunk* alice_codec_encode64(const FrameEncoder* encoder, const uint8_t* rgb, uint64_t rgb_len, uint32_t width,
                                   uint32_t height, uint32_t frames) {
    clear_error();
    if (!encoder || !rgb) { fail(kNullArgument, "null argument"); return nullptr; }
    EncodedChunk* c = new (std::nothrow) EncodedChunk();
    if (!c) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    if (encode_host(*encoder, rgb, rgb_len, width, height, frames, *c) != kOk) { delete c; return nullptr; }
    return c;
}
EncodedChunk* alice_codec_encode(const FrameEncoder* encoder, const uint8_t* rgb, uint32_t rgb_len, uint32_t width,
                                 uint32_t height, uint32_t frames) {
    return alice_codec_encode64(encoder, rgb, rgb_len, width, height, frames);
}

uint8_t* alice_codec_decode64(const EncodedChunk* chunk, uint64_t* out_len) {
    clear_error();
    if (!chunk || !out_len) { fail(kNullArgument, "null argument"); return nullptr; }
    uint8_t* p = nullptr;
    if (decode_host(*chunk, &p, out_len) != kOk) return nullptr;
    return p;
}
uint8_t* alice_codec_decode(const EncodedChunk* chunk, uint32_t* out_len) {
    if (!chunk || !out_len) { clear_error(); fail(kNullArgument, "null argument"); return nullptr; }
    uint64_t n = 0;
    uint8_t* p = alice_codec_decode64(chunk, &n);
    if (p) *out_len = (uint32_t)n;  // `rgb.len() as u32`, src/ffi.rs:157
    return p;
}

void alice_codec_chunk_destroy(EncodedChunk* ptr) { delete ptr; }

uint8_t* alice_codec_chunk_to_bytes64(const EncodedChunk* chunk, uint64_t* out_len) {
    clear_error();
    if (!chunk || !out_len) { fail(kNullArgument, "null argument"); return nullptr; }
    std::vector<uint8_t> v;
    chunk_to_bytes(*chunk, v);
    uint8_t* p = to_c_buffer(v);
    if (!p) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    *out_len = v.size();
    return p;
}
uint8_t* alice_codec_chunk_to_bytes(const EncodedChunk* chunk, uint32_t* out_len) {
    if (!chunk || !out_len) { clear_error(); fail(kNullArgument, "null argument"); return nullptr; }
    uint64_t n = 0;
    uint8_t* p = alice_codec_chunk_to_bytes64(chunk, &n);
    if (p) *out_len = (uint32_t)n;
    return p;
}
EncodedChunk* alice_codec_chunk_from_bytes64(const uint8_t* data, uint64_t len) {
    clear_error();
    if (!data) { fail(kNullArgument, "null argument"); return nullptr; }
    EncodedChunk* c = new (std::nothrow) EncodedChunk();
    if (!c) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    if (chunk_from_bytes(data, len, *c) != kOk) { delete c; return nullptr; }
    return c;
}
EncodedChunk* alice_codec_chunk_from_bytes(const uint8_t* data, uint32_t len) { return alice_codec_chunk_from_bytes64(data, len); }

uint32_t alice_codec_chunk_width(const EncodedChunk* c) { return c ? c->width : 0; }
uint32_t alice_codec_chunk_height(const EncodedChunk* c) { return c ? c->height : 0; }
uint32_t alice_codec_chunk_frames(const EncodedChunk* c) { return c ? c->frames : 0; }

double alice_codec_psnr(const uint8_t* a, const uint8_t* b, uint32_t len) {
    clear_error();
    if (!a || !b) return -1.0;
    if (len == 0) return INFINITY;  // mse 0.0 for empty buffers (src/metrics.rs:23-25)
    hipStream_t st;
    if (get_stream(&st) != kOk) return -1.0;
    DevBuf da, db, ds;
    if (da.alloc(len) || db.alloc(len) || ds.alloc(8)) return -1.0;
    unsigned long long sum = 0;
    if (hipMemcpyAsync(da.p, a, len, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(db.p, b, len, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(ds.p, 0, 8, st) != hipSuccess) { fail(kDeviceError, "copy failed"); return -1.0; }
    launch_sq_diff_sum(da.as<uint8_t>(), db.as<uint8_t>(), len, ds.as<unsigned long long>(), st);
    if (hipMemcpyAsync(&sum, ds.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { fail(kDeviceError, "psnr kernel failed"); return -1.0; }
    // the reference sums squared differences in f64; every partial sum is an integer < 2^53, so the
    // integer total converts to the same f64 (src/metrics.rs:26-34)
    const double mse = (double)sum / (double)len;
    if (mse == 0.0) return INFINITY;
    return 10.0 * log10(255.0 * 255.0 / mse);
}

void alice_codec_data_free64(uint8_t* ptr, uint64_t len) { (void)len; if (ptr) free(ptr); }
void alice_codec_data_free(uint8_t* ptr, uint32_t len) {
    // reference: no-op when ptr is NULL or len == 0 (src/ffi.rs:289); our zero-length buffers are
    // 1-byte mallocs, released here too so nothing leaks
    (void)len;
    if (ptr) free(ptr);
}
void alice_codec_string_free(char* s) { free(s); }
char* alice_codec_version(void) {
    const char* v = "0.1.2";  // CARGO_PKG_VERSION of the reference (Cargo.toml)
    char* p = (char*)malloc(strlen(v) + 1);
    if (p) strcpy(p, v);
    return p;
}

// ---- PART 2: extensions ----

int alice_codec_last_error(void) { return tl_err; }
const char* alice_codec_last_error_message(void) { return tl_msg.c_str(); }
int alice_codec_device_count(void) {
    widen_hw_queues_once();
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
int alice_codec_set_device(int device) {
    clear_error();
    int n = alice_codec_device_count();
    if (device < 0 || device >= n) return fail(kDeviceError, "device index out of range");
    tl_device = device;
    return ensure_device();
}
void alice_codec_trim(void) { pool().trim(); }
void alice_codec_test_force_first_cap(uint64_t cap) { tl_test_first_cap = cap; }

FrameEncoder* alice_codec_encoder_create_ex(uint8_t quality, uint8_t wavelet_type) {
    clear_error();
    if (wavelet_type > 2) { fail(kInvalidBitstream, "unknown wavelet type"); return nullptr; }
    return new (std::nothrow) FrameEncoder{quality, wavelet_type};
}
uint8_t alice_codec_encoder_quality(const FrameEncoder* e) { return e ? e->quality : 0; }
uint8_t alice_codec_encoder_wavelet(const FrameEncoder* e) { return e ? e->wavelet : 0; }
uint8_t alice_codec_chunk_wavelet(const EncodedChunk* c) { return c ? c->wavelet : 0; }
uint64_t alice_codec_chunk_compressed_size(const EncodedChunk* c) { return c ? c->data.size() : 0; }

AliceBatch* alice_codec_batch_create(uint32_t width, uint32_t height, uint32_t frames, uint32_t n_chunks, uint8_t quality,
                                     uint8_t wavelet_type) {
    clear_error();
    if (wavelet_type > 2) { fail(kInvalidBitstream, "unknown wavelet type"); return nullptr; }
    uint64_t n_pixels = 0;
    if (checked_pixel_count(width, height, frames, &n_pixels)) return nullptr;
    if (n_pixels == 0 || n_chunks == 0) { fail(kInvalidDimensions, "empty batch"); return nullptr; }
    const ChunkDims d = make_dims(width, height, frames);
    if (d.padded > 0xFFFFFFFFull) { fail(kDimensionOverflow, "padded pixel count exceeds u32"); return nullptr; }
    if (ensure_device()) return nullptr;
    tl_scope_stream = nullptr;   // the batch's buffers outlive every call: no stream of this thread's past belongs on them
    AliceBatch* b = new (std::nothrow) AliceBatch();
    if (!b) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    b->d = d; b->n_chunks = n_chunks; b->quality = quality; b->wavelet = wavelet_type; b->device = tl_device;
    // one ring of band slots serves the encode and the decode side (i16 both ways for every quality the bound covers)
    if (encode_work_alloc(b->enc, d, (int)n_chunks, inverse_scratch_bytes(d, true)) != kOk || b->evs.init() != kOk) { delete b; return nullptr; }
    return b;
}
void alice_codec_batch_destroy(AliceBatch* b) {
    if (!b) return;
    {
        // the caller's streams may be gone by now; drain the device the batch lives on before its buffers return to the pool
        DeviceScope ds(b->device);
        if (ds.ok) (void)hipDeviceSynchronize();
        b->enc_stream = b->dec_stream = nullptr;
        tl_scope_stream = nullptr;
    }
    delete b;
}

int alice_codec_batch_encode(AliceBatch* b, const void* d_rgb, void* hip_stream) {
    clear_error();
    if (!b || !d_rgb) return fail(kNullArgument, "null argument");
    DeviceScope ds(b->device);
    if (!ds.ok) return tl_err;
    tl_scope_stream = nullptr;   // batch buffers outlive the call; alice_codec_batch_destroy drains the device
    b->enc_stream = (hipStream_t)hip_stream;
    b->enc_timed = false;
    b->last_rgb = (const uint8_t*)d_rgb;
    return encode_launch((const uint8_t*)d_rgb, b->enc, b->quality, b->wavelet, b->enc_stream, &b->evs);
}
int alice_codec_batch_encode_finish(AliceBatch* b, uint64_t* sizes) {
    clear_error();
    if (!b) return fail(kNullArgument, "null argument");
    DeviceScope ds(b->device);
    if (!ds.ok) return tl_err;
    tl_scope_stream = nullptr;
    std::vector<RansResult> res;
    int rc = encode_collect(b->enc, b->enc_stream, res);
    // a chain outgrew its region: the capacities came from an earlier encode of other content (kCapReuse) -- size them
    // from this content's histograms and run again; should even that fall short (never observed), take the format's bound
    for (int mode = kCapEstimate; rc == -1 && b->last_rgb && mode <= kCapWorst; ++mode) {
        TRY(encode_launch(b->last_rgb, b->enc, b->quality, b->wavelet, b->enc_stream, &b->evs, (CapMode)mode));
        rc = encode_collect(b->enc, b->enc_stream, res);
    }
    if (rc == -1) return fail(kInternal, "rANS output exceeded the worst-case capacity");
    if (rc) return rc;
    for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&b->stage_ms[i], b->evs.ev[i], b->evs.ev[i + 1]);
    b->enc_timed = true;
    if (sizes)
        for (uint32_t i = 0; i < b->n_chunks; ++i)
            sizes[i] = (uint64_t)kAlcHeaderBytes + res[3 * i].len + res[3 * i + 1].len + res[3 * i + 2].len;
    return kOk;
}
const void* alice_codec_batch_alc_ptr(const AliceBatch* b, uint32_t chunk) {
    if (!b || chunk >= b->n_chunks) return nullptr;
    return b->enc.alc.as<uint8_t>() + (size_t)chunk * b->enc.alc_stride;
}
uint64_t alice_codec_batch_alc_stride(const AliceBatch* b) { return b ? b->enc.alc_stride : 0; }
const void* alice_codec_batch_symbols_ptr(const AliceBatch* b) { return b ? b->enc.sym.p : nullptr; }
// device memory the batch holds per chunk (symbols = decoded pixels, the .alc buffer at its current capacities, tables,
// histograms, results of both directions) and independent of the chunk count (transform scratch, the spare pixel buffer
// of a banded in-place decode): what a caller needs to size a batch to the free HBM
uint64_t alice_codec_batch_bytes_per_chunk(const AliceBatch* b) {
    if (!b) return 0;
    const uint64_t tables = 3ull * (sizeof(RansTable) + 256 * sizeof(uint32_t) + sizeof(RansResult));
    return 3 * b->d.padded + b->enc.alc_stride + 2 * tables + 3 * sizeof(RansDecodeDesc) + sizeof(unsigned long long);
}
uint64_t alice_codec_batch_fixed_bytes(const AliceBatch* b) {
    if (!b) return 0;
    uint64_t n = (uint64_t)b->enc.scratch.n + (16u << 20);   // + allocation granules of the small buffers
    if (transform_tiles_eligible(b->d) && inverse_cuts_chunk(b->d)) n += b->d.n_pixels * 3;
    return n;
}
uint64_t alice_codec_batch_padded_pixels(const AliceBatch* b) { return b ? b->d.padded : 0; }

int alice_codec_batch_pack_alc(AliceBatch* b, const uint64_t* sizes, void* d_dst, uint64_t dst_capacity, void* hip_stream) {
    clear_error();
    if (!b || !sizes || !d_dst) return fail(kNullArgument, "null argument");
    DeviceScope ds(b->device);
    if (!ds.ok) return tl_err;
    uint64_t off = 0;
    for (uint32_t i = 0; i < b->n_chunks; ++i) {
        if (sizes[i] > b->enc.alc_stride || off + sizes[i] > dst_capacity) return fail(kInvalidBufferSize, "pack buffer too small");
        HIP_TRY(hipMemcpyAsync((uint8_t*)d_dst + off, b->enc.alc.as<uint8_t>() + (size_t)i * b->enc.alc_stride, sizes[i],
                               hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
        off += sizes[i];
    }
    return kOk;
}

int alice_codec_batch_decode(AliceBatch* b, const void* d_alc, uint64_t alc_stride, void* d_rgb_out, void* hip_stream) {
    clear_error();
    if (!b || !d_alc) return fail(kNullArgument, "null argument");
    DeviceScope ds(b->device);
    if (!ds.ok) return tl_err;
    tl_scope_stream = nullptr;
    hipStream_t st = (hipStream_t)hip_stream;
    b->dec_stream = st;
    b->dec_timed = false;
    if (!b->dec_ready) {
        TRY(decode_work_alloc(b->dec, b->d, (int)b->n_chunks, b->enc.sym.as<uint8_t>(), &b->enc.scratch));
        b->dec_ready = true;
    }
    // headers: one strided device-to-host copy, then validation on the host
    std::vector<uint8_t> hdr((size_t)b->n_chunks * kAlcHeaderBytes);
    HIP_TRY(hipMemcpy2DAsync(hdr.data(), kAlcHeaderBytes, d_alc, alc_stride, kAlcHeaderBytes, b->n_chunks, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<EncodedChunk> headers(b->n_chunks);
    std::vector<const uint8_t*> pay(b->n_chunks);
    for (uint32_t i = 0; i < b->n_chunks; ++i) {
        uint64_t payload = 0;
        TRY(parse_alc_header(hdr.data() + (size_t)i * kAlcHeaderBytes, kAlcHeaderBytes, headers[i], &payload));
        if (headers[i].width != b->d.w || headers[i].height != b->d.h || headers[i].frames != b->d.f)
            return fail(kInvalidDimensions, "chunk dimensions differ from the batch shape");
        if ((uint64_t)kAlcHeaderBytes + payload > alc_stride) return fail(kInvalidBitstream, "truncated payload");
        ChunkDims dd;
        TRY(validate_for_decode(headers[i], &dd, payload));
        pay[i] = (const uint8_t*)d_alc + (size_t)i * alc_stride + kAlcHeaderBytes;
    }
    // d_rgb_out == NULL: the batch keeps the pixels in its own storage (alice_codec_batch_rgb_ptr).  An uncut chunk is
    // reconstructed over its own (by then consumed) symbols: its temporal pass is the last reader of those symbols and
    // runs before the tile pass that writes the pixels.  A chunk that is cut into bands writes the pixels of its first
    // band while the temporal pass of its later bands still reads symbols: the pixels of chunk i then land on the symbols
    // of chunk i - 1, whose last reader finished launches ago, and chunk 0 gets a buffer of its own.
    b->rgb_dst.assign(b->n_chunks, nullptr);
    if (!d_rgb_out) {
        const bool cut = inverse_cuts_chunk(b->d);
        if (cut && !b->spare.p) TRY(b->spare.alloc(b->d.n_pixels * 3));
        for (uint32_t i = 0; i < b->n_chunks; ++i)
            b->rgb_dst[i] = !cut ? b->dec.sym_ptr + (size_t)i * 3 * b->d.padded
                                 : (i == 0 ? b->spare.as<uint8_t>() : b->dec.sym_ptr + (size_t)(i - 1) * 3 * b->d.padded);
    } else {
        for (uint32_t i = 0; i < b->n_chunks; ++i) b->rgb_dst[i] = (uint8_t*)d_rgb_out + (size_t)i * b->d.n_pixels * 3;
    }
    return decode_launch(headers, pay, b->dec, b->rgb_dst, st, &b->evs);
}
const void* alice_codec_batch_rgb_ptr(const AliceBatch* b, uint32_t chunk) {
    if (!b || chunk >= b->n_chunks || chunk >= b->rgb_dst.size()) return nullptr;
    return b->rgb_dst[chunk];
}
int alice_codec_batch_decode_finish(AliceBatch* b) {
    clear_error();
    if (!b) return fail(kNullArgument, "null argument");
    DeviceScope ds(b->device);
    if (!ds.ok) return tl_err;
    TRY(decode_collect(b->dec, b->dec_stream));
    (void)hipEventElapsedTime(&b->stage_ms[4], b->evs.ev[5], b->evs.ev[6]);
    (void)hipEventElapsedTime(&b->stage_ms[5], b->evs.ev[6], b->evs.ev[7]);
    b->dec_timed = true;
    return kOk;
}
int alice_codec_batch_stage_ms(const AliceBatch* b, float out[6]) {
    if (!b || !out) return kNullArgument;
    for (int i = 0; i < 6; ++i) out[i] = b->stage_ms[i];
    return kOk;
}

// ---- stage level ----

int alice_codec_wavelet2d_forward(uint8_t k, int32_t* img, uint64_t w, uint64_t h) { clear_error(); return wavelet_nd(k, img, w, h, 1, 2, false); }
int alice_codec_wavelet2d_inverse(uint8_t k, int32_t* img, uint64_t w, uint64_t h) { clear_error(); return wavelet_nd(k, img, w, h, 1, 2, true); }
int alice_codec_wavelet3d_forward(uint8_t k, int32_t* v, uint64_t w, uint64_t h, uint64_t d) { clear_error(); return wavelet_nd(k, v, w, h, d, 3, false); }
int alice_codec_wavelet3d_inverse(uint8_t k, int32_t* v, uint64_t w, uint64_t h, uint64_t d) { clear_error(); return wavelet_nd(k, v, w, h, d, 3, true); }

int alice_codec_quantize_buffer(int32_t step, int32_t dead_zone, const int32_t* in, uint64_t n_in, int32_t* out, uint64_t n_out) {
    clear_error();
    if ((!in || !out) && n_in) return fail(kNullArgument, "null argument");
    if (n_out < n_in) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n_in) + ", got " + std::to_string(n_out));
    if (step == 0) return fail(kInvalidQuantStep, "step 0: the reference divides by zero");
    if (!n_in) return kOk;
    return staged<int32_t, int32_t>(in, n_in, out, n_in, [&](const int32_t* a, int32_t* b, hipStream_t st) { launch_quantize(a, b, n_in, step, dead_zone, st); });
}
int alice_codec_dequantize_buffer(int32_t step, const int32_t* in, uint64_t n_in, int32_t* out, uint64_t n_out) {
    clear_error();
    if ((!in || !out) && n_in) return fail(kNullArgument, "null argument");
    if (n_out < n_in) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n_in) + ", got " + std::to_string(n_out));
    if (!n_in) return kOk;
    return staged<int32_t, int32_t>(in, n_in, out, n_in, [&](const int32_t* a, int32_t* b, hipStream_t st) { launch_dequantize(a, b, n_in, step, st); });
}

FastQuantizer* alice_codec_fastquant_new(int32_t step) {
    clear_error();
    if (step <= 0) { fail(kInvalidQuantStep, "quantization step must be positive, got " + std::to_string(step)); return nullptr; }
    // src/quant.rs:200-216
    const uint32_t step_u = (uint32_t)step;
    const uint32_t extra = 32u - (uint32_t)__builtin_clz(step_u);
    const uint32_t shift = 32u + extra;
    const unsigned __int128 power = (unsigned __int128)1 << shift;
    const uint64_t rec = (uint64_t)((power + step_u - 1) / step_u);
    return new (std::nothrow) FastQuantizer{rec, shift, step, step};
}
FastQuantizer* alice_codec_fastquant_with_dead_zone(int32_t step, int32_t dead_zone) {
    FastQuantizer* q = alice_codec_fastquant_new(step);
    if (q) q->dead_zone = dead_zone;
    return q;
}
void alice_codec_fastquant_destroy(FastQuantizer* q) { delete q; }
int32_t alice_codec_fastquant_step(const FastQuantizer* q) { return q ? q->step : 0; }
int32_t alice_codec_fastquant_dead_zone(const FastQuantizer* q) { return q ? q->dead_zone : 0; }
int alice_codec_fastquant_quantize_buffer(const FastQuantizer* q, const int32_t* in, uint64_t n_in, int32_t* out, uint64_t n_out) {
    clear_error();
    if (!q || ((!in || !out) && n_in)) return fail(kNullArgument, "null argument");
    if (n_out < n_in) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n_in) + ", got " + std::to_string(n_out));
    if (!n_in) return kOk;
    return staged<int32_t, int32_t>(in, n_in, out, n_in, [&](const int32_t* a, int32_t* b, hipStream_t st) {
        launch_fast_quantize(a, b, n_in, q->reciprocal, q->shift, q->dead_zone, st);
    });
}
int alice_codec_fastquant_dequantize_buffer(const FastQuantizer* q, const int32_t* in, uint64_t n_in, int32_t* out, uint64_t n_out) {
    if (!q) { clear_error(); return fail(kNullArgument, "null argument"); }
    return alice_codec_dequantize_buffer(q->step, in, n_in, out, n_out);
}

int alice_codec_to_symbols(const int32_t* coeffs, uint64_t n, uint8_t* symbols, uint64_t n_out) {
    clear_error();
    if ((!coeffs || !symbols) && n) return fail(kNullArgument, "null argument");
    if (n_out < n) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n) + ", got " + std::to_string(n_out));
    if (!n) return kOk;
    return staged<int32_t, uint8_t>(coeffs, n, symbols, n, [&](const int32_t* a, uint8_t* b, hipStream_t st) { launch_to_symbols(a, b, n, st); });
}
int alice_codec_from_symbols(const uint8_t* symbols, uint64_t n, int32_t* coeffs, uint64_t n_out) {
    clear_error();
    if ((!coeffs || !symbols) && n) return fail(kNullArgument, "null argument");
    if (n_out < n) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n) + ", got " + std::to_string(n_out));
    if (!n) return kOk;
    return staged<uint8_t, int32_t>(symbols, n, coeffs, n, [&](const uint8_t* a, int32_t* b, hipStream_t st) { launch_from_symbols(a, b, n, st); });
}
int alice_codec_build_histogram(const uint8_t* symbols, uint64_t n, uint32_t hist[256]) {
    clear_error();
    if (!hist || (!symbols && n)) return fail(kNullArgument, "null argument");
    return staged<uint8_t, uint32_t>(symbols, n, hist, 256, [&](const uint8_t* a, uint32_t* b, hipStream_t st) {
        (void)hipMemsetAsync(b, 0, 256 * sizeof(uint32_t), st);
        launch_histogram(a, n, b, st);
    });
}

int alice_codec_freq_table_from_histogram(const uint32_t hist[256], uint16_t cum_freq[256], uint16_t freq[256]) {
    return alice_codec_freq_table_from_histogram_n(hist, 256, cum_freq, freq);
}
// FrequencyTable::from_histogram(&[u32]) for a slice of any length the u8 symbol API can address (src/rans.rs:102-150;
// the reference's own test_uniform_table_small uses 2, :934-944).  Entries from n_symbols on come back as (0, 0).
int alice_codec_freq_table_from_histogram_n(const uint32_t* hist, uint32_t n_symbols, uint16_t cum_freq[256], uint16_t freq[256]) {
    clear_error();
    if (!hist || !cum_freq || !freq) return fail(kNullArgument, "null argument");
    if (n_symbols == 0) return fail(kReferenceDiverges, "empty histogram: the reference divides by zero (uniform(0), src/rans.rs:159)");
    if (n_symbols > 256) return fail(kInvalidDimensions, "more than 256 symbols: the coders address symbols as u8");
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf h, t;
    TRY(h.alloc(256 * 4));
    TRY(t.alloc(sizeof(RansTable)));
    HIP_TRY(hipMemsetAsync(h.p, 0, 256 * 4, st));
    HIP_TRY(hipMemcpyAsync(h.p, hist, (size_t)n_symbols * 4, hipMemcpyHostToDevice, st));
    launch_rans_table(h.as<uint32_t>(), t.as<RansTable>(), 1, st, n_symbols);
    std::vector<RansTable> host(1);
    HIP_TRY(hipMemcpyAsync(host.data(), t.p, sizeof(RansTable), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < 256; ++i) { cum_freq[i] = (uint16_t)host[0].enc[i].cum; freq[i] = (uint16_t)host[0].enc[i].freq; }
    return kOk;
}

uint8_t* alice_codec_rans_encode(const uint8_t* symbols, uint64_t n, const uint16_t cum_freq[256], const uint16_t freq[256],
                                 uint64_t* out_len) {
    clear_error();
    if ((!symbols && n) || !cum_freq || !freq || !out_len) { fail(kNullArgument, "null argument"); return nullptr; }
    hipStream_t st;
    if (get_stream(&st)) return nullptr;
    const uint64_t cap = round_up(2 * n + 4 + 64 + 64, 256);
    DevBuf ds, dc, df, dt, dout, dres;
    if (ds.alloc(n) || dc.alloc(512) || df.alloc(512) || dt.alloc(sizeof(RansTable)) || dout.alloc(cap) || dres.alloc(sizeof(RansResult))) return nullptr;
    RansResult res{};
    auto ok = [&](hipError_t e) { if (e != hipSuccess) { fail(kDeviceError, hipGetErrorString(e)); return false; } return true; };
    if (n && !ok(hipMemcpyAsync(ds.p, symbols, n, hipMemcpyHostToDevice, st))) return nullptr;
    if (!ok(hipMemcpyAsync(dc.p, cum_freq, 512, hipMemcpyHostToDevice, st)) || !ok(hipMemcpyAsync(df.p, freq, 512, hipMemcpyHostToDevice, st))) return nullptr;
    launch_rans_table_from_arrays(dc.as<uint16_t>(), df.as<uint16_t>(), dt.as<RansTable>(), st);
    if (stage_chains(st, {RansEncodeDesc{ds.as<uint8_t>(), n, dt.as<RansTable>(), dout.as<uint8_t>(), cap, dres.as<RansResult>(), kRansL, 0u}}, {}, n) != kOk)
        return nullptr;
    if (!ok(hipMemcpyAsync(&res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) return nullptr;
    if (res.flags & kTableDiverges) { fail(kReferenceDiverges, "symbol with table frequency 0 encoded"); return nullptr; }
    if (res.flags & (kRansOverflow | kRansInternal)) { fail(kInternal, "rANS encode failed"); return nullptr; }
    uint8_t* p = (uint8_t*)malloc(res.len ? res.len : 1);
    if (!p) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    if (!ok(hipMemcpyAsync(p, dout.as<uint8_t>() + (cap - res.len), res.len, hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) { free(p); return nullptr; }
    *out_len = res.len;
    return p;
}

int alice_codec_rans_decode(const uint8_t* bytes, uint64_t len, const uint16_t cum_freq[256], const uint16_t freq[256],
                            uint64_t n, uint8_t* symbols) {
    clear_error();
    if ((!bytes && len) || !cum_freq || !freq || (!symbols && n)) return fail(kNullArgument, "null argument");
    if (!n) return kOk;
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf din, dc, df, dt, dout, ddesc, dres;
    TRY(din.alloc(len + 16)); TRY(dc.alloc(512)); TRY(df.alloc(512)); TRY(dt.alloc(sizeof(RansTable)));
    TRY(dout.alloc(n)); TRY(ddesc.alloc(sizeof(RansDecodeDesc))); TRY(dres.alloc(sizeof(RansResult)));
    if (len) HIP_TRY(hipMemcpyAsync(din.p, bytes, len, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dc.p, cum_freq, 512, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(df.p, freq, 512, hipMemcpyHostToDevice, st));
    RansDecodeDesc desc{din.as<uint8_t>(), len, dout.as<uint8_t>(), n, dt.as<RansTable>()};
    desc.result = dres.as<RansResult>();
    launch_rans_table_from_arrays(dc.as<uint16_t>(), df.as<uint16_t>(), dt.as<RansTable>(), st);
    TRY(stage_chains(st, {}, {desc}, n));
    RansResult res{};
    HIP_TRY(hipMemcpyAsync(&res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(symbols, dout.p, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    note_decode_stats(res);
    if (res.flags & kRansInternal) return fail(kInternal, "rANS decode kernel invariant violated");
    return kOk;
}
void alice_codec_test_last_decode_stats(uint32_t out[4]) { if (out) for (int i = 0; i < 4; ++i) out[i] = tl_dec_stats[i]; }

// ---- RansEncoder / RansDecoder as objects that live across calls (src/rans.rs:238-309, 321-389) ----

AliceRansEncoder* alice_codec_rans_encoder_new(void) { clear_error(); return new (std::nothrow) AliceRansEncoder(); }
void alice_codec_rans_encoder_destroy(AliceRansEncoder* e) { delete e; }
uint32_t alice_codec_rans_encoder_state(const AliceRansEncoder* e) { return e ? e->state : 0u; }

// encode_symbols(&mut self, symbols, table), :288-294: the symbols in reverse order, from the object's current state.
// Two calls s1 then s2 leave the same stream as one call on s2 || s1.
int alice_codec_rans_encoder_encode_symbols(AliceRansEncoder* e, const uint8_t* symbols, uint64_t n, const uint16_t cum_freq[256],
                                            const uint16_t freq[256]) {
    clear_error();
    if (!e || (!symbols && n) || !cum_freq || !freq) return fail(kNullArgument, "null argument");
    if (!n) return kOk;
    hipStream_t st;
    TRY(get_stream(&st));
    const uint64_t cap = round_up(2 * n + 4 + 64 + 64, 256);
    DevBuf ds, dc, df, dt, dout, dres;
    TRY(ds.alloc(n)); TRY(dc.alloc(512)); TRY(df.alloc(512)); TRY(dt.alloc(sizeof(RansTable))); TRY(dout.alloc(cap)); TRY(dres.alloc(sizeof(RansResult)));
    HIP_TRY(hipMemcpyAsync(ds.p, symbols, n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dc.p, cum_freq, 512, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(df.p, freq, 512, hipMemcpyHostToDevice, st));
    launch_rans_table_from_arrays(dc.as<uint16_t>(), df.as<uint16_t>(), dt.as<RansTable>(), st);
    TRY(stage_chains(st, {RansEncodeDesc{ds.as<uint8_t>(), n, dt.as<RansTable>(), dout.as<uint8_t>(), cap, dres.as<RansResult>(), e->state, 1u}}, {}, n));
    RansResult res{};
    HIP_TRY(hipMemcpyAsync(&res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (res.flags & kTableDiverges) return fail(kReferenceDiverges, "symbol with table frequency 0 encoded (the reference does not terminate / indexes out of bounds)");
    if (res.flags & (kRansOverflow | kRansInternal)) return fail(kInternal, "rANS encode failed");
    std::vector<uint8_t> seg((size_t)res.len);
    if (res.len) {
        HIP_TRY(hipMemcpyAsync(seg.data(), dout.as<uint8_t>() + (cap - res.len), res.len, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    e->state = res.final_state;
    e->bytes += res.len;
    e->segments.push_back(std::move(seg));
    return kOk;
}
// encode(&mut self, &RansSymbol), :269-285: one symbol given by its (cum_freq, freq) pair
int alice_codec_rans_encoder_encode(AliceRansEncoder* e, uint16_t cum_freq, uint16_t freq) {
    if (!e) { clear_error(); return fail(kNullArgument, "null argument"); }
    uint16_t c[256] = {0}, f[256] = {0};
    c[0] = cum_freq; f[0] = freq;
    const uint8_t sym = 0;
    return alice_codec_rans_encoder_encode_symbols(e, &sym, 1, c, f);
}
// finish(self), :298-308: consumes the encoder
uint8_t* alice_codec_rans_encoder_finish(AliceRansEncoder* e, uint64_t* out_len) {
    clear_error();
    if (!e || !out_len) { fail(kNullArgument, "null argument"); return nullptr; }
    const uint64_t total = e->bytes + 4;
    uint8_t* p = (uint8_t*)malloc(total);
    if (!p) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    p[0] = (uint8_t)(e->state >> 24); p[1] = (uint8_t)(e->state >> 16); p[2] = (uint8_t)(e->state >> 8); p[3] = (uint8_t)e->state;
    uint64_t off = 4;
    for (size_t i = e->segments.size(); i-- > 0;) {
        if (!e->segments[i].empty()) memcpy(p + off, e->segments[i].data(), e->segments[i].size());
        off += e->segments[i].size();
    }
    *out_len = total;
    delete e;
    return p;
}

AliceRansDecoder* alice_codec_rans_decoder_new(const uint8_t* data, uint64_t len) {   // RansDecoder::new, :330-347
    clear_error();
    if (!data && len) { fail(kNullArgument, "null argument"); return nullptr; }
    AliceRansDecoder* d = new (std::nothrow) AliceRansDecoder();
    if (!d) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    d->input.assign(data, data + len);
    if (len >= 4) { d->state = ((uint32_t)data[0] << 24) | ((uint32_t)data[1] << 16) | ((uint32_t)data[2] << 8) | data[3]; d->pos = 4; }
    return d;
}
void alice_codec_rans_decoder_destroy(AliceRansDecoder* d) { delete d; }
int alice_codec_rans_decoder_is_empty(const AliceRansDecoder* d) {   // :385-389
    return d ? (d->pos >= d->input.size() && d->state < kRansL) : 1;
}
uint32_t alice_codec_rans_decoder_state(const AliceRansDecoder* d) { return d ? d->state : 0u; }
uint64_t alice_codec_rans_decoder_position(const AliceRansDecoder* d) { return d ? d->pos : 0u; }
// decode_n(&mut self, n, table), :375-381: the next n symbols, continuing from the current state and position
int alice_codec_rans_decoder_decode_n(AliceRansDecoder* d, uint64_t n, const uint16_t cum_freq[256], const uint16_t freq[256],
                                      uint8_t* symbols) {
    clear_error();
    if (!d || !cum_freq || !freq || (!symbols && n)) return fail(kNullArgument, "null argument");
    if (!n) return kOk;
    hipStream_t st;
    TRY(get_stream(&st));
    const uint64_t len = d->input.size();
    if (d->d_input.p && d->device != tl_device) d->d_input.reset();   // the calling thread moved to another device
    if (!d->d_input.p) {
        hipStream_t keep = tl_scope_stream;
        tl_scope_stream = nullptr;   // the copy outlives this call
        const int rc = d->d_input.alloc(len + 16);
        tl_scope_stream = keep;
        TRY(rc);
        d->device = tl_device;
        if (len) HIP_TRY(hipMemcpyAsync(d->d_input.p, d->input.data(), len, hipMemcpyHostToDevice, st));
    }
    DevBuf dc, df, dt, dout, ddesc, dres;
    TRY(dc.alloc(512)); TRY(df.alloc(512)); TRY(dt.alloc(sizeof(RansTable)));
    TRY(dout.alloc(n)); TRY(ddesc.alloc(sizeof(RansDecodeDesc))); TRY(dres.alloc(sizeof(RansResult)));
    HIP_TRY(hipMemcpyAsync(dc.p, cum_freq, 512, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(df.p, freq, 512, hipMemcpyHostToDevice, st));
    RansDecodeDesc desc{d->d_input.as<uint8_t>(), len, dout.as<uint8_t>(), n, dt.as<RansTable>(), 1u, d->state, d->pos};
    desc.result = dres.as<RansResult>();
    launch_rans_table_from_arrays(dc.as<uint16_t>(), df.as<uint16_t>(), dt.as<RansTable>(), st);
    TRY(stage_chains(st, {}, {desc}, n));
    RansResult res{};
    HIP_TRY(hipMemcpyAsync(&res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(symbols, dout.p, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    note_decode_stats(res);
    if (res.flags & kRansInternal) return fail(kInternal, "rANS decode kernel invariant violated");
    d->state = res.final_state;
    d->pos = res.len;
    d->started = true;
    return kOk;
}

// quantize_subband / dequantize_subband (src/quant.rs:518-545): a sub-band's coefficients through a Quantizer
int alice_codec_quantize_subband(int32_t step, int32_t dead_zone, const int32_t* coeffs, uint64_t n, int32_t* out, uint64_t n_out) {
    return alice_codec_quantize_buffer(step, dead_zone, coeffs, n, out, n_out);
}
int alice_codec_dequantize_subband(int32_t step, const int32_t* coeffs, uint64_t n, int32_t* out, uint64_t n_out) {
    return alice_codec_dequantize_buffer(step, coeffs, n, out, n_out);
}

// ssim / ms_ssim (src/ssim.rs:63-176).  Returns the value, or -1.0 with the thread's error set (the Result::Err cases).
static int ssim_device(const uint8_t* d_a, const uint8_t* d_b, uint64_t w, uint64_t h, double* d_blocks, double* d_acc,
                       hipStream_t st, double* out) {
    const uint64_t bw = w / 8, bh = h / 8, nb = bw * bh;
    if (nb == 0) { *out = 1.0; return kOk; }                              // block_count == 0, :111-113
    launch_ssim_blocks(d_a, d_b, w, bw, nb, d_blocks, st);
    launch_ordered_sum_f64(d_blocks, nb, d_acc, st);
    double total = 0.0;
    HIP_TRY(hipMemcpyAsync(&total, d_acc, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *out = total / (double)nb;
    return kOk;
}
static int ssim_impl(const uint8_t* a, uint64_t a_len, const uint8_t* b, uint64_t b_len, uint64_t w, uint64_t h, bool multi, double* out) {
    if (a_len != b_len) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(a_len) + ", got " + std::to_string(b_len));
    unsigned __int128 wh = (unsigned __int128)w * h;
    if (wh != a_len) return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string((uint64_t)wh) + ", got " + std::to_string(a_len));
    if (a_len == 0) { *out = 1.0; return kOk; }
    if (!a || !b) return fail(kNullArgument, "null argument");
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf da, db, da2, db2, dblk, dacc;
    TRY(da.alloc(a_len)); TRY(db.alloc(a_len)); TRY(dblk.alloc(((w / 8) * (h / 8) + 1) * sizeof(double))); TRY(dacc.alloc(8));
    HIP_TRY(hipMemcpyAsync(da.p, a, a_len, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(db.p, b, a_len, hipMemcpyHostToDevice, st));
    if (!multi) return ssim_device(da.as<uint8_t>(), db.as<uint8_t>(), w, h, dblk.as<double>(), dacc.as<double>(), st, out);
    TRY(da2.alloc(a_len / 4 + 1)); TRY(db2.alloc(a_len / 4 + 1));
    const double weights[3] = {0.3333, 0.3333, 0.3334};
    uint8_t *ca = da.as<uint8_t>(), *cb = db.as<uint8_t>(), *na = da2.as<uint8_t>(), *nb = db2.as<uint8_t>();
    uint64_t cw = w, ch = h;
    double result = 0.0;
    for (int wi = 0; wi < 3; ++wi) {
        const double weight = weights[wi];
        double s = 0.0;
        TRY(ssim_device(ca, cb, cw, ch, dblk.as<double>(), dacc.as<double>(), st, &s));
        double l = log(s > 0.0 ? s : 0.0);                                   // s.max(0.0).ln().max(-10.0), :146
        if (!(l > -10.0)) l = -10.0;
        result += weight * l;
        const uint64_t nw = cw / 2, nh = ch / 2;
        if (nw < 8 || nh < 8) {
            // the reference finds the current weight by value (first match), :153-163: the second scale maps to
            // index 0 again and counts weights[1] once more
            int pos = 0;
            for (int k = 0; k < 3; ++k) if (fabs(weights[k] - weight) < 1e-10) { pos = k; break; }
            for (int k = pos + 1; k < 3; ++k) result += weights[k] * l;
            break;
        }
        launch_downsample2(ca, cw, ch, na, st);
        launch_downsample2(cb, cw, ch, nb, st);
        std::swap(ca, na); std::swap(cb, nb);
        cw = nw; ch = nh;
    }
    *out = exp(result);
    return kOk;
}
double alice_codec_ssim(const uint8_t* a, uint64_t a_len, const uint8_t* b, uint64_t b_len, uint64_t width, uint64_t height) {
    clear_error();
    double v = -1.0;
    return ssim_impl(a, a_len, b, b_len, width, height, false, &v) == kOk ? v : -1.0;
}
double alice_codec_ms_ssim(const uint8_t* a, uint64_t a_len, const uint8_t* b, uint64_t b_len, uint64_t width, uint64_t height) {
    clear_error();
    double v = -1.0;
    return ssim_impl(a, a_len, b, b_len, width, height, true, &v) == kOk ? v : -1.0;
}

// AnalyticalRDO (src/quant.rs:377-505) + SubBand3D::quant_strength (src/lib.rs:149-158)
double alice_codec_rdo_target_bpp(uint8_t quality) {   // with_quality, :398-411
    const double RCP_100 = 1.0 / 100.0;
    const unsigned qq = quality > 100 ? 100u : quality;
    const double q = (double)qq * RCP_100;
    return fma(q * q, 23.9, 0.1);
}
uint8_t alice_codec_subband_quant_strength(uint8_t subband) {
    switch (subband) { case 0: return 1; case 1: case 2: case 4: return 2; case 3: case 5: case 6: return 4; default: return 8; }
}
int alice_codec_rdo_compute_quantizer(double target_bpp, const int32_t* coeffs, uint64_t n, uint8_t subband, int32_t* step,
                                      int32_t* dead_zone) {
    clear_error();
    if ((!coeffs && n) || !step || !dead_zone) return fail(kNullArgument, "null argument");
    if (subband > 7) return fail(kInvalidDimensions, "unknown sub-band");
    double variance = 1.0;                                           // estimate_variance of an empty slice, :415-417
    if (n) {
        hipStream_t st;
        TRY(get_stream(&st));
        DevBuf dx, dsum, dacc;
        TRY(dx.alloc(n * sizeof(int32_t))); TRY(dsum.alloc(8)); TRY(dacc.alloc(8));
        HIP_TRY(hipMemcpyAsync(dx.p, coeffs, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(dsum.p, 0, 8, st));
        launch_sum_i32(dx.as<int32_t>(), n, dsum.as<unsigned long long>(), st);
        long long sum = 0;
        HIP_TRY(hipMemcpyAsync(&sum, dsum.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const double inv_n = 1.0 / (double)n;
        const double mean = (double)sum * inv_n;                      // :421-424
        launch_ordered_sqdev_sum(dx.as<int32_t>(), n, mean, dacc.as<double>(), st);
        double acc = 0.0;
        HIP_TRY(hipMemcpyAsync(&acc, dacc.p, 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        variance = acc * inv_n;                                       // :426-433
        if (!(variance > 1.0)) variance = 1.0;                        // f64::max(1.0)
    }
    const double lambda = (6.0 * M_LN2 * variance) / target_bpp;      // :440-443
    const double r = round(sqrt(12.0 * lambda));                      // :448-451
    int32_t base = r != r ? 0 : (r >= 2147483647.0 ? INT32_MAX : (r <= -2147483648.0 ? INT32_MIN : (int32_t)r));
    if (base < 1) base = 1;
    int32_t s = (int32_t)((uint32_t)base * (uint32_t)alice_codec_subband_quant_strength(subband));   // :461-462
    if (s < 1) s = 1;
    *step = s;
    *dead_zone = (int32_t)((uint32_t)s + (uint32_t)(s / 2));          // :465
    return kOk;
}

// InterleavedRansEncoder::{encode, finish} (src/rans.rs:393-456): four independent single-stream coders over the
// sub-sequences i = j mod 4, behind a 32-byte header (4 stream lengths, 4 symbol counts, u32 LE).  On the GPU
// that is four chains of the same encode kernel running side by side.
uint8_t* alice_codec_rans_encode_interleaved(const uint8_t* symbols, uint64_t n, const uint16_t cum_freq[256],
                                             const uint16_t freq[256], uint64_t* out_len) {
    clear_error();
    if ((!symbols && n) || !cum_freq || !freq || !out_len) { fail(kNullArgument, "null argument"); return nullptr; }
    hipStream_t st;
    if (get_stream(&st)) return nullptr;
    uint64_t cnt[4];
    for (int j = 0; j < 4; ++j) cnt[j] = (n + 3 - j) / 4;        // :423-425
    if (cnt[0] > 0xFFFFFFFFull) { fail(kDimensionOverflow, "symbol count does not fit the u32 header field"); return nullptr; }
    const uint64_t stride = round_up(cnt[0] + 16, 256);
    const uint64_t cap = round_up(2 * cnt[0] + 4 + 64 + 64, 256);
    DevBuf ds, d4, dc, df, dt, dout, dres;
    if (ds.alloc(n) || d4.alloc(4 * stride) || dc.alloc(512) || df.alloc(512) || dt.alloc(4 * sizeof(RansTable)) ||
        dout.alloc(4 * cap) || dres.alloc(4 * sizeof(RansResult))) return nullptr;
    auto ok = [&](hipError_t e) { if (e != hipSuccess) { fail(kDeviceError, hipGetErrorString(e)); return false; } return true; };
    if (n && !ok(hipMemcpyAsync(ds.p, symbols, n, hipMemcpyHostToDevice, st))) return nullptr;
    if (!ok(hipMemcpyAsync(dc.p, cum_freq, 512, hipMemcpyHostToDevice, st)) || !ok(hipMemcpyAsync(df.p, freq, 512, hipMemcpyHostToDevice, st))) return nullptr;
    launch_split4(ds.as<uint8_t>(), n, d4.as<uint8_t>(), stride, st);
    for (int j = 0; j < 4; ++j) launch_rans_table_from_arrays(dc.as<uint16_t>(), df.as<uint16_t>(), dt.as<RansTable>() + j, st);
    {
        std::vector<RansEncodeDesc> enc(4);
        for (int j = 0; j < 4; ++j)
            enc[j] = RansEncodeDesc{d4.as<uint8_t>() + (size_t)j * stride, cnt[j], dt.as<RansTable>() + j, dout.as<uint8_t>() + (size_t)j * cap, cap,
                                    dres.as<RansResult>() + j, kRansL, 0u};
        if (stage_chains(st, std::move(enc), {}, cnt[0]) != kOk) return nullptr;
    }
    RansResult res[4];
    if (!ok(hipMemcpyAsync(res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st)) || !ok(hipStreamSynchronize(st))) return nullptr;
    uint64_t total = 32;
    for (int j = 0; j < 4; ++j) {
        if (res[j].flags & kTableDiverges) { fail(kReferenceDiverges, "symbol with table frequency 0 encoded"); return nullptr; }
        if (res[j].flags & (kRansOverflow | kRansInternal)) { fail(kInternal, "rANS encode failed"); return nullptr; }
        if (res[j].len > 0xFFFFFFFFull) { fail(kDimensionOverflow, "stream length does not fit the u32 header field"); return nullptr; }
        total += res[j].len;
    }
    uint8_t* p = (uint8_t*)malloc(total);
    if (!p) { fail(kOutOfMemory, "out of host memory"); return nullptr; }
    uint64_t off = 32;
    for (int j = 0; j < 4; ++j) {
        const uint32_t l = (uint32_t)res[j].len, c = (uint32_t)cnt[j];
        memcpy(p + 4 * j, &l, 4);
        memcpy(p + 16 + 4 * j, &c, 4);
        if (!ok(hipMemcpyAsync(p + off, dout.as<uint8_t>() + (size_t)j * cap + (cap - res[j].len), res[j].len, hipMemcpyDeviceToHost, st))) { free(p); return nullptr; }
        off += res[j].len;
    }
    if (!ok(hipStreamSynchronize(st))) { free(p); return nullptr; }
    *out_len = total;
    return p;
}

// InterleavedRansDecoder::{new, decode_n} (src/rans.rs:468-519; SimdRansDecoder reads the same format): the four
// streams are decoded by four chains, then merged in the reference's round-robin order (streams that run out
// are skipped).  Where the reference would index out of bounds or spin forever this returns an error.
int alice_codec_rans_decode_interleaved(const uint8_t* bytes, uint64_t len, const uint16_t cum_freq[256],
                                        const uint16_t freq[256], uint64_t n, uint8_t* symbols) {
    clear_error();
    if ((!bytes && len) || !cum_freq || !freq || (!symbols && n)) return fail(kNullArgument, "null argument");
    if (len < 32) return fail(kInvalidBitstream, "interleaved stream shorter than its 32-byte header");
    uint64_t slen[4], cnt[4], total = 0, end = 32;
    for (int j = 0; j < 4; ++j) {
        uint32_t a, b;
        memcpy(&a, bytes + 4 * j, 4); memcpy(&b, bytes + 16 + 4 * j, 4);
        slen[j] = a; cnt[j] = b; total += b; end += a;
    }
    if (end > len) return fail(kInvalidBitstream, "stream lengths exceed the input");
    if (n > total) return fail(kReferenceDiverges, "more symbols requested than the header counts hold: the reference decoder does not terminate");
    if (!n) return kOk;
    // symbols of stream j that land below position n: pos(j, k) = sum_i min(cnt_i, k) + #{i < j : cnt_i > k}
    auto pos = [&](int j, uint64_t k) {
        uint64_t p = 0;
        for (int i = 0; i < 4; ++i) { p += std::min(cnt[i], k); if (i < j && cnt[i] > k) ++p; }
        return p;
    };
    uint64_t need[4], mx = 0;
    for (int j = 0; j < 4; ++j) {
        uint64_t lo = 0, hi = cnt[j];      // first k with pos(j, k) >= n
        while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (pos(j, mid) >= n) hi = mid; else lo = mid + 1; }
        need[j] = lo; mx = std::max(mx, lo);
    }
    hipStream_t st;
    TRY(get_stream(&st));
    const uint64_t stride = round_up(mx + 16, 256);
    DevBuf din, dc, df, dt, d4, dout, ddesc, dres;
    TRY(din.alloc(len + 16)); TRY(dc.alloc(512)); TRY(df.alloc(512)); TRY(dt.alloc(sizeof(RansTable)));
    TRY(d4.alloc(4 * stride)); TRY(dout.alloc(n)); TRY(ddesc.alloc(4 * sizeof(RansDecodeDesc))); TRY(dres.alloc(4 * sizeof(RansResult)));
    HIP_TRY(hipMemcpyAsync(din.p, bytes, len, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dc.p, cum_freq, 512, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(df.p, freq, 512, hipMemcpyHostToDevice, st));
    std::vector<RansDecodeDesc> desc(4);
    uint64_t off = 32;
    for (int j = 0; j < 4; ++j) {
        desc[j] = RansDecodeDesc{din.as<uint8_t>() + off, slen[j], d4.as<uint8_t>() + (size_t)j * stride, need[j], dt.as<RansTable>()};
        desc[j].result = dres.as<RansResult>() + j;
        off += slen[j];
    }
    launch_rans_table_from_arrays(dc.as<uint16_t>(), df.as<uint16_t>(), dt.as<RansTable>(), st);
    TRY(stage_chains(st, {}, std::move(desc), mx));
    launch_merge4(d4.as<uint8_t>(), stride, need, cnt, dout.as<uint8_t>(), n, st);
    RansResult res[4];
    HIP_TRY(hipMemcpyAsync(res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(symbols, dout.p, n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (auto& r : res)
        if (r.flags & kRansInternal) return fail(kInternal, "decode table outside the packed range (freq > 4096 with live slots)");
    return kOk;
}

int alice_codec_rgb_to_ycocg_r(const uint8_t* rgb, uint64_t rgb_len, int16_t* y, int16_t* co, int16_t* cg, uint64_t n_out) {
    clear_error();
    if (rgb_len % 3 != 0) return fail(kInvalidBufferSize, "rgb length is not a multiple of 3");  // src/color.rs:205-210
    const uint64_t n = rgb_len / 3;
    if (n_out < n) return fail(kInvalidBufferSize, "output planes too small");                   // :212-218
    if (!n) return kOk;
    if (!rgb || !y || !co || !cg) return fail(kNullArgument, "null argument");
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf a, b;
    TRY(a.alloc(rgb_len)); TRY(b.alloc(3 * n * 2));
    HIP_TRY(hipMemcpyAsync(a.p, rgb, rgb_len, hipMemcpyHostToDevice, st));
    int16_t* p = b.as<int16_t>();
    launch_rgb_to_ycocg(a.as<uint8_t>(), n, p, p + n, p + 2 * n, st);
    HIP_TRY(hipMemcpyAsync(y, p, n * 2, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(co, p + n, n * 2, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(cg, p + 2 * n, n * 2, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}
int alice_codec_ycocg_r_to_rgb(const int16_t* y, const int16_t* co, const int16_t* cg, uint64_t n, uint8_t* rgb, uint64_t rgb_len) {
    clear_error();
    if (rgb_len < n * 3) return fail(kInvalidBufferSize, "rgb buffer too small");  // src/color.rs:258-263
    if (!n) return kOk;
    if (!rgb || !y || !co || !cg) return fail(kNullArgument, "null argument");
    hipStream_t st;
    TRY(get_stream(&st));
    DevBuf a, b;
    TRY(a.alloc(3 * n * 2)); TRY(b.alloc(n * 3));
    int16_t* p = a.as<int16_t>();
    HIP_TRY(hipMemcpyAsync(p, y, n * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p + n, co, n * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(p + 2 * n, cg, n * 2, hipMemcpyHostToDevice, st));
    launch_ycocg_to_rgb(p, p + n, p + 2 * n, n, b.as<uint8_t>(), st);
    HIP_TRY(hipMemcpyAsync(rgb, b.p, n * 3, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}

// ---- many chunks from host memory in one call: the chunk driver's fast path (the chains of all chunks run side
// by side; a single chunk is bound by its three serial chains) ----

// How many chunks of this shape the calling thread's device can hold at once (inputs or outputs, symbols, .alc
// buffers with the worst observed payload of 1 byte per pixel sample, tables), leaving a fifth of the free memory alone.
static uint32_t chunks_that_fit(const ChunkDims& d, uint32_t want) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return want; }
    const long double per_chunk = (long double)d.n_pixels * 3 + (long double)d.padded * 3 * 2 + 3.0L * sizeof(RansTable) * 2 + 65536;
    const long double fixed = (long double)std::max(forward_scratch_bytes(d), inverse_scratch_bytes(d, false)) + (64u << 20);
    const long double room = (long double)free_b * 0.8L - fixed;
    if (room < per_chunk) return 1;
    const long double n = room / per_chunk;
    return n >= (long double)want ? want : (uint32_t)n;
}

// Encodes the chunks at rgb[i] (host pointers, n_pixels * 3 bytes each) on the calling thread's device, as many at a
// time as its memory holds; out[i] receives the chunk objects.  On error nothing is left in out.
static int encode_chunks_on_device(const FrameEncoder& enc, const std::vector<const uint8_t*>& rgb, const ChunkDims& d,
                                   const std::vector<EncodedChunk**>& out) {
    const uint32_t n = (uint32_t)rgb.size();
    for (uint32_t i = 0; i < n; ++i) *out[i] = nullptr;
    if (!n) return kOk;
    TRY(ensure_device());
    const uint32_t per_pass = chunks_that_fit(d, n);
    HubTicket ticket;
    TRY(ticket.open(encode_device_bytes(d, per_pass)));
    const hipStream_t st = ticket.st;
    auto undo = [&](int rc) { for (uint32_t k = 0; k < n; ++k) { delete *out[k]; *out[k] = nullptr; } return rc; };
    const uint64_t chunk_bytes = d.n_pixels * 3;
    for (uint32_t first = 0; first < n; first += per_pass) {
        const uint32_t B = std::min(per_pass, n - first);
        DevBuf d_rgb;
        EncodeWork w;
        std::vector<RansResult> res;
        int rc = d_rgb.alloc(chunk_bytes * B);
        if (rc == kOk) rc = encode_work_alloc(w, d, (int)B);
        if (rc != kOk) return undo(rc);
        rc = parallel_chunks(B, chunk_bytes >= (4u << 20) ? kCopyThreads : 1u, [&](uint32_t i) {
            HIP_TRY(hipMemcpyAsync(d_rgb.as<uint8_t>() + (size_t)i * chunk_bytes, rgb[first + i], chunk_bytes, hipMemcpyHostToDevice, st));
            return (int)kOk;
        });
        if (rc != kOk) return undo(rc);
        for (int attempt = 0;; ++attempt) {
            rc = encode_launch(d_rgb.as<uint8_t>(), w, enc.quality, enc.wavelet, st, nullptr, (CapMode)attempt, &ticket);
            if (rc != kOk) return undo(rc);
            rc = encode_collect(w, st, res);
            if (rc == kOk) break;
            if (rc != -1 || attempt >= 2) return undo(rc == -1 ? fail(kInternal, "rANS output exceeded the worst-case bound") : rc);
        }
        // all headers with one strided copy, then the payloads straight into the chunk objects, several at a time
        std::vector<uint8_t> hdr((size_t)B * kAlcHeaderBytes);
        if (hipMemcpy2DAsync(hdr.data(), kAlcHeaderBytes, w.alc.p, w.alc_stride, kAlcHeaderBytes, B, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            return undo(fail(kDeviceError, "device to host copy failed"));
        for (uint32_t i = 0; i < B; ++i) {
            const uint64_t payload = res[3 * i].len + res[3 * i + 1].len + res[3 * i + 2].len;
            EncodedChunk* c = new (std::nothrow) EncodedChunk();
            uint64_t tot = 0;
            if (!c || parse_alc_header(hdr.data() + (size_t)i * kAlcHeaderBytes, kAlcHeaderBytes + payload, *c, &tot) != kOk || tot != payload) {
                delete c;
                return undo(fail(kInternal, "device header/payload length mismatch"));
            }
            *out[first + i] = c;   // (undo() deletes it from here on)
        }
        rc = parallel_chunks(B, w.alc_stride >= (4u << 20) ? kCopyThreads : 1u, [&](uint32_t i) {
            EncodedChunk* c = *out[first + i];
            const uint64_t payload = res[3 * i].len + res[3 * i + 1].len + res[3 * i + 2].len;
            c->data.resize((size_t)payload);
            return copy_to_host(c->data.data(), w.alc.as<uint8_t>() + (size_t)i * w.alc_stride + kAlcHeaderBytes, (size_t)payload, st);
        });
        if (rc != kOk) return undo(rc);
    }
    return kOk;
}

// Decodes chunks (all of one shape, validated by the caller) on the calling thread's device into rgb_out[i].
static int decode_chunks_on_device(const std::vector<const EncodedChunk*>& chunks, const ChunkDims& d, const std::vector<uint8_t*>& rgb_out) {
    const uint32_t n = (uint32_t)chunks.size();
    if (!n || d.n_pixels == 0) return kOk;
    TRY(ensure_device());
    const uint32_t per_pass = chunks_that_fit(d, n);
    uint64_t max_payload = 0;
    for (const EncodedChunk* c : chunks) max_payload = std::max<uint64_t>(max_payload, c->data.size());
    HubTicket ticket;
    TRY(ticket.open(decode_device_bytes(d, per_pass, (uint64_t)per_pass * (max_payload + 512))));
    const hipStream_t st = ticket.st;
    const uint64_t chunk_bytes = d.n_pixels * 3;
    for (uint32_t first = 0; first < n; first += per_pass) {
        const uint32_t B = std::min(per_pass, n - first);
        std::vector<EncodedChunk> hdrs(B);
        uint64_t total_payload = 0;
        for (uint32_t i = 0; i < B; ++i) {
            const EncodedChunk& c = *chunks[first + i];
            hdrs[i].width = c.width; hdrs[i].height = c.height; hdrs[i].frames = c.frames; hdrs[i].wavelet = c.wavelet;
            for (int k = 0; k < 3; ++k) hdrs[i].ch[k] = c.ch[k];
            total_payload += round_up(c.data.size() + 16, 256);
        }
        DevBuf d_payload, d_rgb;
        TRY(d_payload.alloc(total_payload + 256));
        TRY(d_rgb.alloc(chunk_bytes * B));
        std::vector<const uint8_t*> pay(B);
        std::vector<uint8_t*> dst(B);
        uint64_t off = 0;
        for (uint32_t i = 0; i < B; ++i) {
            const EncodedChunk& c = *chunks[first + i];
            pay[i] = d_payload.as<uint8_t>() + off;
            dst[i] = d_rgb.as<uint8_t>() + (size_t)i * chunk_bytes;
            if (!c.data.empty())
                HIP_TRY(hipMemcpyAsync(d_payload.as<uint8_t>() + off, c.data.data(), c.data.size(), hipMemcpyHostToDevice, st));
            off += round_up(c.data.size() + 16, 256);
        }
        DecodeWork w;
        TRY(decode_work_alloc(w, d, (int)B));
        TRY(decode_launch(hdrs, pay, w, dst, st, nullptr, &ticket));
        TRY(decode_collect(w, st));
        TRY(parallel_chunks(B, chunk_bytes >= (4u << 20) ? kCopyThreads : 1u,
                            [&](uint32_t i) { return copy_to_host(rgb_out[first + i], dst[i], chunk_bytes, st); }));
    }
    return kOk;
}

// validation shared by the many-chunk encode entry points (FrameEncoder::encode's, src/pipeline.rs:388-427, per chunk)
static int validate_encode_many(const FrameEncoder* encoder, const uint8_t* rgb, uint64_t rgb_len, uint32_t width, uint32_t height,
                                uint32_t frames, uint32_t n_chunks, EncodedChunk** out_chunks, ChunkDims* d) {
    if (!encoder || !out_chunks || (!rgb && rgb_len)) return fail(kNullArgument, "null argument");
    for (uint32_t i = 0; i < n_chunks; ++i) out_chunks[i] = nullptr;
    if (n_chunks == 0) return kOk;
    uint64_t n_pixels = 0;
    TRY(checked_pixel_count(width, height, frames, &n_pixels));
    if (n_pixels == 0 || width == 0 || height == 0) return fail(kInvalidDimensions, "invalid dimensions");
    if (n_pixels > UINT64_MAX / 3 / n_chunks) return fail(kDimensionOverflow, "dimensions overflow usize");
    if (rgb_len != n_pixels * 3 * n_chunks)
        return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(n_pixels * 3 * n_chunks) + ", got " + std::to_string(rgb_len));
    *d = make_dims(width, height, frames);
    if (d->padded > 0xFFFFFFFFull) return fail(kDimensionOverflow, "padded pixel count does not fit the header's u32 num_symbols");
    return kOk;
}
static int validate_decode_many(const EncodedChunk* const* chunks, uint32_t n_chunks, uint8_t* rgb_out, uint64_t rgb_out_len, ChunkDims* d) {
    if (!chunks || (!rgb_out && rgb_out_len)) return fail(kNullArgument, "null argument");
    if (n_chunks == 0) return kOk;
    for (uint32_t i = 0; i < n_chunks; ++i) {
        if (!chunks[i]) return fail(kNullArgument, "null chunk");
        if (chunks[i]->width != chunks[0]->width || chunks[i]->height != chunks[0]->height || chunks[i]->frames != chunks[0]->frames)
            return fail(kInvalidDimensions, "chunks of one call must have the same shape");
    }
    for (uint32_t i = 0; i < n_chunks; ++i) TRY(validate_for_decode(*chunks[i], d, chunks[i]->data.size()));
    if (d->n_pixels == 0) return rgb_out_len == 0 ? kOk : fail(kInvalidBufferSize, "output buffer size mismatch");
    if (rgb_out_len != d->n_pixels * 3 * n_chunks)
        return fail(kInvalidBufferSize, "buffer size mismatch: expected " + std::to_string(d->n_pixels * 3 * n_chunks) + ", got " + std::to_string(rgb_out_len));
    return kOk;
}

int alice_codec_encode_many(const FrameEncoder* encoder, const uint8_t* rgb, uint64_t rgb_len, uint32_t width, uint32_t height,
                            uint32_t frames, uint32_t n_chunks, EncodedChunk** out_chunks) {
    clear_error();
    ChunkDims d{};
    TRY(validate_encode_many(encoder, rgb, rgb_len, width, height, frames, n_chunks, out_chunks, &d));
    if (n_chunks == 0) return kOk;
    std::vector<const uint8_t*> src(n_chunks);
    std::vector<EncodedChunk**> dst(n_chunks);
    for (uint32_t i = 0; i < n_chunks; ++i) { src[i] = rgb + (size_t)i * d.n_pixels * 3; dst[i] = &out_chunks[i]; }
    return encode_chunks_on_device(*encoder, src, d, dst);
}

int alice_codec_decode_many(const EncodedChunk* const* chunks, uint32_t n_chunks, uint8_t* rgb_out, uint64_t rgb_out_len) {
    clear_error();
    ChunkDims d{};
    TRY(validate_decode_many(chunks, n_chunks, rgb_out, rgb_out_len, &d));
    if (n_chunks == 0 || d.n_pixels == 0) return kOk;
    std::vector<const EncodedChunk*> src(chunks, chunks + n_chunks);
    std::vector<uint8_t*> dst(n_chunks);
    for (uint32_t i = 0; i < n_chunks; ++i) dst[i] = rgb_out + (size_t)i * d.n_pixels * 3;
    return decode_chunks_on_device(src, d, dst);
}

// ---- the same over several GPUs of the node: chunk k goes to devices[k mod n_devices] (64-frame chunks are independent
// bitstreams, src/pipeline.rs:461-497: no cross-chunk state), one host thread per listed device, every
// device moves its own chunks over its own PCIe link, results land in the caller's arrays in chunk order ----

int alice_codec_many_devices_plan(uint32_t n_chunks, const int* devices, uint32_t n_devices, int* device_of_chunk) {
    clear_error();
    if ((!devices && n_devices) || (!device_of_chunk && n_chunks)) return fail(kNullArgument, "null argument");
    if (n_devices == 0) return fail(kDeviceError, "empty device list");
    for (uint32_t i = 0; i < n_devices; ++i)
        if (devices[i] < 0) return fail(kDeviceError, "negative device index");
    for (uint32_t k = 0; k < n_chunks; ++k) device_of_chunk[k] = devices[k % n_devices];
    return kOk;
}

}  // extern "C"
namespace {
struct WorkerResult { int code = kOk; std::string msg; };
// runs fn(slot) on one thread per slot of the device list, each bound to its device; returns the first failure
template <typename Fn>
int run_on_devices(const int* devices, uint32_t n_devices, Fn fn) {
    const int count = alice_codec_device_count();
    for (uint32_t i = 0; i < n_devices; ++i)
        if (devices[i] >= count) return fail(kDeviceError, "device index " + std::to_string(devices[i]) + " out of range (" + std::to_string(count) + " visible)");
    std::vector<WorkerResult> res(n_devices);
    std::vector<std::thread> th;
    th.reserve(n_devices);
    for (uint32_t s = 0; s < n_devices; ++s)
        th.emplace_back([&, s] {
            clear_error();
            tl_device = devices[s];
            int rc = ensure_device();
            if (rc == kOk) rc = fn(s);
            res[s].code = rc;
            if (rc != kOk) res[s].msg = tl_msg;
        });
    for (auto& t : th) t.join();
    for (uint32_t s = 0; s < n_devices; ++s)
        if (res[s].code != kOk) return fail(res[s].code, "device " + std::to_string(devices[s]) + ": " + res[s].msg);
    return kOk;
}
}  // namespace
extern "C" {

int alice_codec_encode_many_devices(const FrameEncoder* encoder, const uint8_t* rgb, uint64_t rgb_len, uint32_t width, uint32_t height,
                                    uint32_t frames, uint32_t n_chunks, const int* devices, uint32_t n_devices, EncodedChunk** out_chunks) {
    clear_error();
    ChunkDims d{};
    TRY(validate_encode_many(encoder, rgb, rgb_len, width, height, frames, n_chunks, out_chunks, &d));
    std::vector<int> plan(n_chunks);
    TRY(alice_codec_many_devices_plan(n_chunks, devices, n_devices, plan.data()));
    if (n_chunks == 0) return kOk;
    const int rc = run_on_devices(devices, n_devices, [&](uint32_t slot) {
        std::vector<const uint8_t*> src;
        std::vector<EncodedChunk**> dst;
        for (uint32_t k = slot; k < n_chunks; k += n_devices) { src.push_back(rgb + (size_t)k * d.n_pixels * 3); dst.push_back(&out_chunks[k]); }
        return encode_chunks_on_device(*encoder, src, d, dst);
    });
    if (rc != kOk)
        for (uint32_t k = 0; k < n_chunks; ++k) { delete out_chunks[k]; out_chunks[k] = nullptr; }
    return rc;
}

int alice_codec_decode_many_devices(const EncodedChunk* const* chunks, uint32_t n_chunks, const int* devices, uint32_t n_devices,
                                    uint8_t* rgb_out, uint64_t rgb_out_len) {
    clear_error();
    ChunkDims d{};
    TRY(validate_decode_many(chunks, n_chunks, rgb_out, rgb_out_len, &d));
    std::vector<int> plan(n_chunks);
    TRY(alice_codec_many_devices_plan(n_chunks, devices, n_devices, plan.data()));
    if (n_chunks == 0 || d.n_pixels == 0) return kOk;
    return run_on_devices(devices, n_devices, [&](uint32_t slot) {
        std::vector<const EncodedChunk*> src;
        std::vector<uint8_t*> dst;
        for (uint32_t k = slot; k < n_chunks; k += n_devices) { src.push_back(chunks[k]); dst.push_back(rgb_out + (size_t)k * d.n_pixels * 3); }
        return decode_chunks_on_device(src, d, dst);
    });
}

// ---- PART 3: device-resident stage calls (building blocks of the row-slab sharded path) ----
// Every pointer named d_* is a device pointer; launches go on `hip_stream` and the call returns after the
// stream has drained (the rANS calls need their result on the host anyway).

int alice_codec_dev_forward_symbols(const void* d_rgb, uint32_t width, uint32_t height, uint32_t frames, uint8_t wavelet_type,
                                    uint8_t quality, void* d_symbols, void* d_hist, void* hip_stream) {
    clear_error();
    if (!d_rgb || !d_symbols) return fail(kNullArgument, "null argument");
    if (wavelet_type > 2) return fail(kInvalidBitstream, "unknown wavelet type");
    uint64_t n_pixels = 0;
    TRY(checked_pixel_count(width, height, frames, &n_pixels));
    if (n_pixels == 0) return fail(kInvalidDimensions, "invalid dimensions");
    const ChunkDims d = make_dims(width, height, frames);
    if (d.padded > 0xFFFFFFFFull) return fail(kDimensionOverflow, "padded pixel count exceeds u32");
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    ScopeStream scope(st);   // temporaries drain the caller's stream before they return to the pool
    EncodeWork w;
    w.d = d; w.n_chunks = 1;
    const bool tiles = transform_tiles_eligible(d);
    if (tiles) TRY(w.scratch.alloc(forward_scratch_bytes(d)));
    TRY(w.hist.alloc(3 * 256 * sizeof(uint32_t)));
    uint32_t* hist = d_hist ? (uint32_t*)d_hist : w.hist.as<uint32_t>();
    HIP_TRY(hipMemsetAsync(hist, 0, 3 * 256 * sizeof(uint32_t), st));
    const int32_t step = quality_to_step(quality);
    if (!tiles || !launch_forward_transform((const uint8_t*)d_rgb, d, wavelet_type, step, w.scratch.p, (uint8_t*)d_symbols, hist, st))
        TRY(forward_generic((const uint8_t*)d_rgb, d, wavelet_type, step, w, (uint8_t*)d_symbols, hist, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}

int alice_codec_dev_inverse_symbols(const void* d_symbols, uint32_t width, uint32_t height, uint32_t frames, uint8_t wavelet_type,
                                    const int32_t step[3], void* d_rgb, void* hip_stream) {
    clear_error();
    if (!d_rgb || !d_symbols || !step) return fail(kNullArgument, "null argument");
    if (wavelet_type > 2) return fail(kInvalidBitstream, "unknown wavelet type");
    uint64_t n_pixels = 0;
    TRY(checked_pixel_count(width, height, frames, &n_pixels));
    if (n_pixels == 0) return fail(kInvalidDimensions, "invalid dimensions");
    const ChunkDims d = make_dims(width, height, frames);
    if (d.padded > 0xFFFFFFFFull) return fail(kDimensionOverflow, "padded pixel count exceeds u32");
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    ScopeStream scope(st);   // temporaries drain the caller's stream before they return to the pool
    DecodeWork w;
    w.d = d; w.n_chunks = 1;
    const InverseBounds ib = inverse_bounds(wavelet_type, step);
    const bool tiles = transform_tiles_eligible(d);
    if (tiles) TRY(w.scratch_own.alloc(inverse_scratch_bytes(d, ib.fast && ib.mid16)));
    if (!tiles || !launch_inverse_transform((const uint8_t*)d_symbols, d, wavelet_type, step, !ib.fast, ib.fast && ib.mid16, ib.fast && ib.lds16, w.scratch_own.p, (uint8_t*)d_rgb, st))
        TRY(inverse_generic((const uint8_t*)d_symbols, d, wavelet_type, step, w, (uint8_t*)d_rgb, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}

static int dev_wavelet3d(uint8_t k, void* d_volume, void* d_tmp, uint64_t w, uint64_t h, uint64_t d, bool inverse, void* hip_stream) {
    clear_error();
    if (!d_volume || !d_tmp) return fail(kNullArgument, "null argument");
    if (k > 2) return fail(kInvalidBitstream, "unknown wavelet type");
    unsigned __int128 tot = (unsigned __int128)w * h * d;
    if (tot > ((unsigned __int128)1 << 40)) return fail(kDimensionOverflow, "volume too large");
    if (tot == 0) return kOk;
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    wavelet_on_device(k, (int32_t*)d_volume, (int32_t*)d_tmp, w, h, d, 3, inverse, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}
int alice_codec_dev_wavelet3d_forward(uint8_t k, void* d_volume, void* d_tmp, uint64_t w, uint64_t h, uint64_t d, void* hip_stream) {
    return dev_wavelet3d(k, d_volume, d_tmp, w, h, d, false, hip_stream);
}
int alice_codec_dev_wavelet3d_inverse(uint8_t k, void* d_volume, void* d_tmp, uint64_t w, uint64_t h, uint64_t d, void* hip_stream) {
    return dev_wavelet3d(k, d_volume, d_tmp, w, h, d, true, hip_stream);
}

int alice_codec_dev_histogram(const void* d_symbols, uint64_t n, void* d_hist, void* hip_stream) {
    clear_error();
    if (!d_hist || (!d_symbols && n)) return fail(kNullArgument, "null argument");
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    ScopeStream scope(st);   // temporaries drain the caller's stream before they return to the pool
    HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(uint32_t), st));
    if (n) launch_histogram((const uint8_t*)d_symbols, n, (uint32_t*)d_hist, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return kOk;
}

uint64_t alice_codec_rans_stream_bound(const uint32_t hist[256], uint64_t n) {
    if (!hist) return round_up(2 * n + 4 + 64 + 64, 256);
    const uint64_t worst = round_up(2 * n + 4 + 64 + 64, 256);
    const uint64_t est = round_up(estimate_stream_cap(hist, n) + 64, 256);
    return est < worst ? est : worst;
}

int alice_codec_dev_rans_encode(const void* d_symbols, uint64_t n, const uint32_t hist[256], void* d_out, uint64_t cap,
                                uint64_t* out_offset, uint64_t* out_len, void* hip_stream) {
    clear_error();
    if ((!d_symbols && n) || !hist || !d_out || !out_offset || !out_len) return fail(kNullArgument, "null argument");
    if (cap < 4 + 64 + 64) return fail(kInvalidBufferSize, "stream region too small");
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    ScopeStream scope(st);   // temporaries drain the caller's stream before they return to the pool
    DevBuf dh, dt, dres;
    TRY(dh.alloc(256 * 4)); TRY(dt.alloc(sizeof(RansTable))); TRY(dres.alloc(sizeof(RansResult)));
    HIP_TRY(hipMemcpyAsync(dh.p, hist, 256 * 4, hipMemcpyHostToDevice, st));
    launch_rans_table(dh.as<uint32_t>(), dt.as<RansTable>(), 1, st);
    launch_rans_encode((const uint8_t*)d_symbols, n, n, dt.as<RansTable>(), (uint8_t*)d_out, cap, dres.as<RansResult>(), 1, st);
    RansResult res{};
    HIP_TRY(hipMemcpyAsync(&res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (res.flags & kTableDiverges)
        return fail(kReferenceDiverges, "a symbol whose table frequency wrapped to 0 is present: the reference encoder does not terminate on this input");
    if (res.flags & kRansInternal) return fail(kInternal, "rANS kernel invariant violated");
    if (res.flags & kRansOverflow) return fail(kInvalidBufferSize, "stream region too small for this chain (use alice_codec_rans_stream_bound(NULL, n))");
    *out_len = res.len;
    *out_offset = cap - res.len;
    return kOk;
}

int alice_codec_dev_rans_decode(const void* d_stream, uint64_t len, const uint32_t hist[256], void* d_symbols, uint64_t n,
                                void* hip_stream) {
    clear_error();
    if ((!d_stream && len) || !hist || (!d_symbols && n)) return fail(kNullArgument, "null argument");
    if (!n) return kOk;
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    ScopeStream scope(st);   // temporaries drain the caller's stream before they return to the pool
    DevBuf dh, dt, ddesc, dres;
    TRY(dh.alloc(256 * 4)); TRY(dt.alloc(sizeof(RansTable))); TRY(ddesc.alloc(sizeof(RansDecodeDesc))); TRY(dres.alloc(sizeof(RansResult)));
    HIP_TRY(hipMemcpyAsync(dh.p, hist, 256 * 4, hipMemcpyHostToDevice, st));
    RansDecodeDesc desc{(const uint8_t*)d_stream, len, (uint8_t*)d_symbols, n, dt.as<RansTable>()};
    HIP_TRY(hipMemcpyAsync(ddesc.p, &desc, sizeof(desc), hipMemcpyHostToDevice, st));
    launch_rans_table(dh.as<uint32_t>(), dt.as<RansTable>(), 1, st);
    launch_rans_decode(ddesc.as<RansDecodeDesc>(), dres.as<RansResult>(), 1, st);
    RansResult res{};
    HIP_TRY(hipMemcpyAsync(&res, dres.p, sizeof(res), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    note_decode_stats(res);
    if (res.flags & kRansInternal) return fail(kInternal, "rANS decode table invariant violated");
    return kOk;
}


// ---- test and measurement hooks (include/alice_codec_test.h) ----

void alice_codec_test_set_tuning(long band_kb) { set_transform_tuning(band_kb); }
void alice_codec_test_set_value_table_radius(int r) { set_value_table_radius(r); }
int alice_codec_test_set_admission_budget(uint64_t bytes) {
    clear_error();
    TRY(ensure_device());
    ChainHub::of_device(tl_device)->set_budget((size_t)bytes);
    return kOk;
}

int alice_codec_test_chain_occupancy(uint32_t out[6]) {
    clear_error();
    if (!out) return fail(kNullArgument, "null argument");
    TRY(ensure_device());
    if (!chain_kernel_occupancy(out)) return fail(kDeviceError, "hipFuncGetAttributes / occupancy query failed");
    return kOk;
}

int alice_codec_test_transform_ms(const void* d_rgb, void* d_sym, void* d_rgb_out, uint32_t n_buffers, uint32_t width,
                                  uint32_t height, uint32_t frames, uint8_t wavelet_type, uint8_t quality, uint32_t n_chunks,
                                  uint32_t reps, int probe, float out_ms[2], void* hip_stream) {
    clear_error();
    if (!d_rgb || !d_sym || !d_rgb_out || !out_ms || !n_buffers || !n_chunks || !reps) return fail(kNullArgument, "null argument");
    if (wavelet_type > 2) return fail(kInvalidBitstream, "unknown wavelet type");
    uint64_t n_pixels = 0;
    TRY(checked_pixel_count(width, height, frames, &n_pixels));
    if (n_pixels == 0) return fail(kInvalidDimensions, "invalid dimensions");
    const ChunkDims d = make_dims(width, height, frames);
    if (!transform_tiles_eligible(d)) return fail(kInvalidDimensions, "shape runs the generic path: nothing to time");
    TRY(ensure_device());
    hipStream_t st = (hipStream_t)hip_stream;
    ScopeStream scope(st);
    const int32_t step = quality_to_step(quality);
    const int32_t steps[3] = {step, step, step};
    const InverseBounds ib = inverse_bounds(wavelet_type, steps);
    DevBuf scratch, hist;
    TRY(scratch.alloc(std::max(forward_scratch_bytes(d), inverse_scratch_bytes(d, ib.fast && ib.mid16))));
    TRY(hist.alloc(3 * 256 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(hist.p, 0, 3 * 256 * sizeof(uint32_t), st));
    StageEvents ev;
    TRY(ev.init());
    set_transform_probe(probe);
    auto fwd = [&]() {
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t k = c % n_buffers;
            launch_forward_transform((const uint8_t*)d_rgb + (size_t)k * d.n_pixels * 3, d, wavelet_type, step, scratch.p,
                                     (uint8_t*)d_sym + (size_t)k * 3 * d.padded, hist.as<uint32_t>(), st);
        }
    };
    auto inv = [&]() {
        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t k = c % n_buffers;
            launch_inverse_transform((const uint8_t*)d_sym + (size_t)k * 3 * d.padded, d, wavelet_type, steps, !ib.fast, ib.fast && ib.mid16,
                                     ib.fast && ib.lds16, scratch.p, (uint8_t*)d_rgb_out + (size_t)k * d.n_pixels * 3, st);
        }
    };
    fwd();   // warm-up; also leaves real symbols for the inverse
    hipError_t e = hipEventRecord(ev.ev[0], st);
    for (uint32_t r = 0; r < reps && e == hipSuccess; ++r) fwd();
    if (e == hipSuccess) e = hipEventRecord(ev.ev[1], st);
    inv();
    if (e == hipSuccess) e = hipEventRecord(ev.ev[2], st);
    for (uint32_t r = 0; r < reps && e == hipSuccess; ++r) inv();
    if (e == hipSuccess) e = hipEventRecord(ev.ev[3], st);
    set_transform_probe(0);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(kDeviceError, hipGetErrorString(e));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, ev.ev[0], ev.ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, ev.ev[2], ev.ev[3]));
    out_ms[0] = a / (float)(reps * n_chunks);
    out_ms[1] = b / (float)(reps * n_chunks);
    return kOk;
}

}  // extern "C"
