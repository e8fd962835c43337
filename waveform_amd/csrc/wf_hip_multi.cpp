// wf_hip_multi.cpp -- one batch over the devices of a node (include/wf_hip.h, "one batch over several devices").
//
// Shape: the reference's sources share nothing, so the streams shard contiguously; every shard is a
// plain wf_hip handle on its own device and is driven by its own host thread, which keeps that device current for its whole
// life (hipSetDevice is per thread) -- the API thread only hands jobs to the workers and collects their status, so the n
// devices' launches are issued concurrently instead of one after the other.  No collective on the data path.  The one
// exchange is the all-gather of the bar heights: ncclAllGather (RCCL over xGMI; librccl.so is dlopen()ed, the library does
// not link it) on a side stream per device, or direct peer copies where RCCL is unavailable.  Host code only: gfx950 kernels
// live in wf_hip.hip.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h> // types and prototypes only: every RCCL entry point is resolved with dlsym

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <barrier>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "wf_hip.h"

namespace {

thread_local std::string g_multi_create_error;

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *path = std::getenv("WF_HIP_RCCL_LIBRARY");
        for(const char *name : {path, "librccl.so.1", "librccl.so"}) {
            if(name == nullptr)
                continue;
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if(r.lib)
                break;
        }
        if(r.lib == nullptr) {
            const char *e = dlerror();
            r.why = std::string("librccl.so not loadable: ") + (e ? e : "?");
            return;
        }
#define WF_NCCL_SYM(name)                                                          \
    r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.lib, "nccl" #name));      \
    if(r.name == nullptr) {                                                        \
        r.why = "librccl.so lacks nccl" #name;                                     \
        return;                                                                    \
    }
        WF_NCCL_SYM(CommInitAll)
        WF_NCCL_SYM(CommDestroy)
        WF_NCCL_SYM(CommAbort)
        WF_NCCL_SYM(AllGather)
        WF_NCCL_SYM(GetErrorString)
#undef WF_NCCL_SYM
        r.ok = true;
    });
    return r;
}

// A device's host thread: runs the jobs it is handed, one at a time, with its device current.
struct Worker {
    std::thread th;
    std::mutex mtx;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = 0;

    void start(int device)
    {
        th = std::thread([this, device] {
            (void)hipSetDevice(device);
            std::unique_lock lock(mtx);
            for(;;) {
                cv.wait(lock, [this] { return has_job || quit; });
                if(quit)
                    return;
                auto fn = std::move(job);
                has_job = false;
                lock.unlock();
                const int r = fn();
                lock.lock();
                rc = r;
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> fn)
    {
        std::lock_guard lock(mtx);
        job = std::move(fn);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock lock(mtx);
        cv.wait(lock, [this] { return done; });
        return rc;
    }
    void stop()
    {
        {
            std::lock_guard lock(mtx);
            quit = true;
            cv.notify_all();
        }
        if(th.joinable())
            th.join();
    }
};

enum class Transport { LOCAL, RCCL, PEER };

struct Shard {
    int device = 0;
    wf_hip *h = nullptr;
    uint32_t first = 0, count = 0;
    hipStream_t gstream = nullptr;       // the gather's side stream on this device
    float *send[2] = {nullptr, nullptr}; // this shard's bars, [largest][disp_ch][num_bars] (padded for ragged RCCL gathers)
    float *recv_pad[2] = {nullptr, nullptr}; // RCCL with shards of unequal size: [n][largest][...] before compaction
    float *gathered[2] = {nullptr, nullptr}; // [streams_total][disp_ch][num_bars]
    hipEvent_t ev_sent[2] = {nullptr, nullptr}; // peer transport: this shard's copies into every device's buffer have run
    hipEvent_t ev_done[2] = {nullptr, nullptr}; // the gathered result of the slot is complete on this device
    bool slot_used[2] = {false, false};
    // zero-copy gathers (wf_hip_set_bars_mirror): the handle's tick kernel writes its bars into the slot's send buffer itself
    // (LOCAL: into the result), alternating with the ticks -- the slot of a gather is then the buffer the newest tick wrote
    bool mirror = false;
    bool direct = false; // peer transport with peer access everywhere: the tick kernel stores this shard's slice into every device's result itself
    uint32_t cur_slot = 0; // the slot of the gather in flight / issued last on this shard
    bool peer_ok[64] = {}; // [j]: this device may address device j's memory (peer access enabled, or the same device)
    ncclComm_t comm = nullptr;
    Worker worker;
    std::string err;
};

} // namespace

struct wf_hip_multi {
    wf_config cfg{};
    uint32_t n = 0, total = 0, largest = 0;
    bool ragged = false;
    size_t per = 0; // floats per stream in the bars buffer: display_channels * num_bars
    std::vector<std::unique_ptr<Shard>> shard;
    Transport transport = Transport::LOCAL;
    std::string transport_note;
    uint32_t gathers = 0; // gathers issued so far; without the mirror the slot of the next one = gathers & 1
    int last_slot = -1;   // the slot that holds the newest complete result; -1: none (no gather yet, or the last timed run failed half way)
    // A gather that failed on one shard leaves the others with a collective nobody answers (RCCL) or with copies that never
    // come (peer): the group then aborts its communicators (ncclCommAbort ends the kernels already enqueued, so the gather
    // streams drain), refuses every later gather and goes on ticking, reading and destroying normally.
    bool gather_failed = false;
    std::string gather_failure;
    int debug_fail_shard = -1; // test aid (wf_hip_multi_debug_fail_next_gather): that shard's next gather_issue reports an error
    std::string last_error;
};

namespace {

int mfail(wf_hip_multi *m, int code, const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if(m)
        m->last_error = buf;
    else
        g_multi_create_error = buf;
    return code;
}

// fn(i) on every device's own thread, concurrently; the first failure wins (its text is kept)
int run_all(wf_hip_multi *m, const std::function<int(uint32_t)> &fn)
{
    for(uint32_t i = 0; i < m->n; ++i)
        m->shard[i]->worker.post([&fn, i] { return fn(i); });
    int rc = WF_HIP_OK;
    for(uint32_t i = 0; i < m->n; ++i) {
        const int r = m->shard[i]->worker.wait();
        if(r != WF_HIP_OK && rc == WF_HIP_OK) {
            rc = r;
            Shard &s = *m->shard[i];
            m->last_error = "device " + std::to_string(s.device) + " (shard " + std::to_string(i) + "): " +
                            (s.err.empty() ? std::string(wf_hip_last_error(s.h)) : s.err);
        }
        m->shard[i]->err.clear();
    }
    return rc;
}

int hip_rc(Shard &s, hipError_t e, const char *what)
{
    if(e == hipSuccess)
        return WF_HIP_OK;
    s.err = std::string(what) + " failed: " + hipGetErrorString(e);
    return WF_HIP_ERR_RUNTIME;
}
#define WF_MHIP(s, expr)                          \
    do {                                          \
        const int rc_ = hip_rc((s), (expr), #expr); \
        if(rc_)                                   \
            return rc_;                           \
    } while(0)

// the part of the global range [first, first+count) that falls into shard s: local first / count, offset into the range
bool overlap(const Shard &s, uint32_t first, uint32_t count, uint32_t *lfirst, uint32_t *lcount, uint32_t *off)
{
    const uint64_t lo = std::max<uint64_t>(first, s.first), hi = std::min<uint64_t>((uint64_t)first + count, (uint64_t)s.first + s.count);
    if(lo >= hi)
        return false;
    *lfirst = (uint32_t)(lo - s.first);
    *lcount = (uint32_t)(hi - lo);
    *off = (uint32_t)(lo - first);
    return true;
}

int check_range(wf_hip_multi *m, uint32_t first, uint32_t count)
{
    if(m == nullptr)
        return WF_HIP_ERR_INVALID;
    if(count == 0 || first >= m->total || count > m->total - first)
        return mfail(m, WF_HIP_ERR_INVALID, "stream range [%u, %u+%u) outside 0..%u", first, first, count, m->total);
    return WF_HIP_OK;
}

// The gather of shard i, first half: its bars into the slot's send buffer behind the ticks issued so far, then -- on the
// device's gather stream, which waits only for that copy -- the collective (RCCL) or this shard's copies into every
// device's result (peer).  Nothing here waits on the host.
int gather_issue(wf_hip_multi *m, uint32_t i, uint32_t k)
{
    Shard &s = *m->shard[i];
#ifdef WF_DEV_BUILD
    if(m->debug_fail_shard == (int)i) { // (test aid: fails before anything is enqueued, as a failed wait or copy would)
        s.err = "injected failure (wf_hip_multi_debug_fail_next_gather)";
        return WF_HIP_ERR_RUNTIME;
    }
#endif
    int rc;
    if(s.mirror) {
        // the handle's ticks have written the bars into one of the two sets themselves (wf_hip_bars_mirror_ready: the gather stream
        // waits for the newest tick, the other set becomes the ticks' target; a set no tick has written is filled from the
        // handle's own bars): nothing is copied here
        void *buf = nullptr;
        rc = wf_hip_bars_mirror_ready(s.h, s.gstream, &buf);
        if(rc)
            return rc;
        // buffer 0 of a set: the send buffer (RCCL, peer copies), the result itself (one device), or -- direct peer stores -- this
        // shard's slice of device 0's result
        const float *b1 = s.direct ? m->shard[0]->gathered[1] + (size_t)s.first * m->per : (m->transport == Transport::LOCAL) ? s.gathered[1] : s.send[1];
        k = buf == (const void *)b1 ? 1u : 0u;
    } else {
        if(s.slot_used[k]) { // the gather that read this send buffer two gathers ago must have run before the buffer is rewritten
            rc = wf_hip_wait_event(s.h, s.ev_done[k]);
            if(rc)
                return rc;
        }
        float *dst = (m->transport == Transport::LOCAL) ? s.gathered[k] : s.send[k];
        rc = wf_hip_copy_bars_device_async(s.h, 0, s.count, dst, s.gstream);
        if(rc)
            return rc;
    }
    s.cur_slot = k;
    const size_t per = m->per;
    switch(m->transport) {
    case Transport::LOCAL: break;
    case Transport::RCCL: {
        float *recv = m->ragged ? s.recv_pad[k] : s.gathered[k];
        const ncclResult_t r = rccl().AllGather(s.send[k], recv, (size_t)m->largest * per, ncclFloat, s.comm, s.gstream);
        if(r != ncclSuccess) {
            s.err = std::string("ncclAllGather failed: ") + rccl().GetErrorString(r);
            return WF_HIP_ERR_RUNTIME;
        }
        if(m->ragged) // rank r's block of `largest` streams holds count_r valid ones: compact into global stream order
            for(uint32_t r2 = 0; r2 < m->n; ++r2) {
                const Shard &o = *m->shard[r2];
                WF_MHIP(s, hipMemcpyAsync(s.gathered[k] + (size_t)o.first * per, recv + (size_t)r2 * m->largest * per,
                                          (size_t)o.count * per * sizeof(float), hipMemcpyDeviceToDevice, s.gstream));
            }
        break;
    }
    case Transport::PEER:
        for(uint32_t j = 0; j < m->n && !s.direct; ++j) { // (direct: the tick kernel has stored the slice everywhere already)
            Shard &o = *m->shard[j];
            WF_MHIP(s, hipMemcpyPeerAsync(o.gathered[k] + (size_t)s.first * per, o.device, s.send[k], s.device,
                                          (size_t)s.count * per * sizeof(float), s.gstream));
        }
        WF_MHIP(s, hipEventRecord(s.ev_sent[k], s.gstream));
        break;
    }
    return WF_HIP_OK;
}

// Second half: the result on device i is complete when every shard's copies into it have run (peer transport: events of the
// other devices' gather streams, all recorded by now -- the caller put a host barrier between the halves).
int gather_complete(wf_hip_multi *m, uint32_t i, uint32_t k)
{
    Shard &s = *m->shard[i];
    if(s.mirror)
        k = s.cur_slot; // (the group has handed over every shard's set the same number of times: the same slot on all of them -- checked by the callers)
    if(m->transport == Transport::PEER)
        for(uint32_t j = 0; j < m->n; ++j)
            if(j != i)
                WF_MHIP(s, hipStreamWaitEvent(s.gstream, m->shard[j]->ev_sent[k], 0));
    WF_MHIP(s, hipEventRecord(s.ev_done[k], s.gstream));
    s.slot_used[k] = true;
    // The ticks issued from here on write the other slot's buffers (the hand-over in gather_issue made them the write set).  Where
    // that is a SEND buffer (RCCL; peer copies), the exchange that read it -- a gather old -- must have run: a host wait that returns
    // at once (a device-side wait in front of every tick cost 4 % of the tick rate).  Where the kernels store into the results
    // themselves (local, direct peer stores) nothing inside the group reads the buffer: it is the result of the gather before this
    // one, which the header's contract gives up with the first tick after this gather.
    if(s.mirror && !s.direct && m->transport != Transport::LOCAL && s.slot_used[k ^ 1u])
        WF_MHIP(s, hipEventSynchronize(s.ev_done[k ^ 1u]));
    return WF_HIP_OK;
}

// Zero-copy gathers: the slot is the set each shard's handle has just handed over.  The group hands over all of them together, so
// they agree -- unless somebody called wf_hip_bars_mirror_ready / wf_hip_set_bars_mirrors on a shard handle behind the group's back
// (the header forbids it): the pieces of the result would then lie in different buffers.
int slots_agree(wf_hip_multi *m)
{
    for(uint32_t i = 1; i < m->n; ++i)
        if(m->shard[i]->mirror && m->shard[0]->mirror && m->shard[i]->cur_slot != m->shard[0]->cur_slot)
            return mfail(m, WF_HIP_ERR_RUNTIME, "shard %u handed over bars buffer %u, shard 0 buffer %u: the shards' mirror buffers were handed over or replaced outside the group",
                         i, m->shard[i]->cur_slot, m->shard[0]->cur_slot);
    return WF_HIP_OK;
}

int gather_check(wf_hip_multi *m)
{
    if(m == nullptr)
        return WF_HIP_ERR_INVALID;
    if(m->per == 0)
        return mfail(m, WF_HIP_ERR_INVALID, "the configuration has no bars or curve (cfg.bars == 0 and cfg.curve == 0): nothing to gather");
    if(m->gather_failed)
        return mfail(m, WF_HIP_ERR_RUNTIME, "the group's gather is out of service after an earlier failure (%s); ticks and reads go on", m->gather_failure.c_str());
    return WF_HIP_OK;
}

// A shard failed inside a gather while others had already enqueued their half of it.  From the API thread, no worker running:
// the communicators are aborted (every one of them, so that no rank keeps waiting in a kernel for a peer that never launched),
// the events the handles' streams may be waiting for are recorded afresh on the (now draining) gather streams, and the group
// is marked so that sync / destroy / later gathers do not count on a collective any more.  m->last_error keeps the cause.
void gather_fail(wf_hip_multi *m)
{
    if(m->gather_failed)
        return;
    m->gather_failed = true;
    m->gather_failure = m->last_error;
    m->debug_fail_shard = -1;
    for(auto &sp : m->shard)
        if(sp->comm) {
            (void)hipSetDevice(sp->device);
            (void)rccl().CommAbort(sp->comm);
            sp->comm = nullptr;
        }
    // whatever a handle's streams were told to wait for (ev_done of a slot whose collective is gone) completes now
    for(auto &sp : m->shard) {
        Shard &s = *sp;
        if(!s.worker.th.joinable() || s.gstream == nullptr)
            continue;
        s.worker.post([&s] {
            for(int k = 0; k < 2; ++k) {
                if(s.ev_sent[k]) (void)hipEventRecord(s.ev_sent[k], s.gstream);
                if(s.ev_done[k]) (void)hipEventRecord(s.ev_done[k], s.gstream);
                s.slot_used[k] = false;
            }
            return 0;
        });
        (void)s.worker.wait();
    }
}

void destroy_impl(wf_hip_multi *m)
{
    if(m == nullptr)
        return;
    // the workers' last job: drain and free what lives on their device
    for(auto &sp : m->shard) {
        Shard &s = *sp;
        if(!s.worker.th.joinable())
            continue;
        s.worker.post([&s] {
            if(s.gstream)
                (void)hipStreamSynchronize(s.gstream);
            if(s.h)
                (void)wf_hip_sync(s.h);
            return 0;
        });
        (void)s.worker.wait();
    }
    for(auto &sp : m->shard) // communicators go first, all of them, from one thread (ncclCommDestroy may synchronise with peers)
        if(sp->comm) {
            (void)hipSetDevice(sp->device);
            (void)rccl().CommDestroy(sp->comm);
            sp->comm = nullptr;
        }
    for(auto &sp : m->shard) {
        Shard &s = *sp;
        if(s.worker.th.joinable()) {
            s.worker.post([&s] {
                for(int k = 0; k < 2; ++k) {
                    if(s.send[k]) (void)hipFree(s.send[k]);
                    if(s.recv_pad[k]) (void)hipFree(s.recv_pad[k]);
                    if(s.gathered[k]) (void)hipFree(s.gathered[k]);
                    if(s.ev_sent[k]) (void)hipEventDestroy(s.ev_sent[k]);
                    if(s.ev_done[k]) (void)hipEventDestroy(s.ev_done[k]);
                }
                if(s.gstream)
                    (void)hipStreamDestroy(s.gstream);
                if(s.h)
                    wf_hip_destroy(s.h);
                return 0;
            });
            (void)s.worker.wait();
            s.worker.stop();
        }
    }
    delete m;
}

} // namespace

extern "C" {

const char *wf_hip_multi_last_error(const wf_hip_multi *m) { return m ? m->last_error.c_str() : g_multi_create_error.c_str(); }
uint32_t wf_hip_multi_num_devices(const wf_hip_multi *m) { return m ? m->n : 0; }
uint32_t wf_hip_multi_num_streams(const wf_hip_multi *m) { return m ? m->total : 0; }

const char *wf_hip_multi_transport(const wf_hip_multi *m)
{
    if(m == nullptr)
        return "";
    return m->transport == Transport::RCCL ? "rccl" : m->transport == Transport::PEER ? "peer" : "local";
}

int wf_hip_multi_create(const wf_config *cfg, const int *devices, uint32_t n_devices, uint32_t streams_total, uint32_t ring_frames,
                        wf_hip_multi **out)
{
    if(out == nullptr)
        return WF_HIP_ERR_INVALID;
    *out = nullptr;
    if(cfg == nullptr || devices == nullptr || n_devices == 0 || n_devices > 64)
        return mfail(nullptr, WF_HIP_ERR_INVALID, "cfg / devices NULL, or n_devices %u outside 1..64", n_devices);
    if(streams_total < n_devices)
        return mfail(nullptr, WF_HIP_ERR_INVALID, "%u streams cannot be spread over %u devices (every shard needs one)", streams_total, n_devices);
    const int have = wf_hip_device_count();
    if(have <= 0)
        return mfail(nullptr, WF_HIP_ERR_NO_DEVICE, "no usable HIP device");
    bool duplicates = false;
    for(uint32_t i = 0; i < n_devices; ++i) {
        if(devices[i] < 0 || devices[i] >= have)
            return mfail(nullptr, WF_HIP_ERR_INVALID, "devices[%u] = %d, the box has %d", i, devices[i], have);
        for(uint32_t j = 0; j < i; ++j)
            duplicates = duplicates || devices[j] == devices[i];
    }
    auto *m = new(std::nothrow) wf_hip_multi;
    if(m == nullptr)
        return mfail(nullptr, WF_HIP_ERR_NOMEM, "out of host memory");
    m->cfg = *cfg;
    m->n = n_devices;
    m->total = streams_total;
    const uint32_t base = streams_total / n_devices, extra = streams_total % n_devices;
    m->largest = base + (extra ? 1u : 0u);
    m->ragged = extra != 0;
    for(uint32_t i = 0; i < n_devices; ++i) {
        auto s = std::make_unique<Shard>();
        s->device = devices[i];
        s->count = base + (i < extra ? 1u : 0u);
        s->first = i * base + std::min(i, extra);
        s->worker.start(s->device);
        m->shard.push_back(std::move(s));
    }
    // the shards' handles, every device building its own concurrently
    int rc = run_all(m, [m, cfg, ring_frames](uint32_t i) {
        Shard &s = *m->shard[i];
        const int r = wf_hip_create(cfg, s.device, s.count, ring_frames, &s.h);
        if(r != WF_HIP_OK)
            s.err = wf_hip_last_error(nullptr);
        return r;
    });
    if(rc) {
        g_multi_create_error = m->last_error;
        destroy_impl(m);
        return rc;
    }
    m->per = (size_t)wf_hip_display_channels(m->shard[0]->h) * wf_hip_num_bars(m->shard[0]->h);
    // transport of the gather
    const char *force = std::getenv("WF_HIP_MULTI_TRANSPORT");
    const bool want_rccl = force ? std::strcmp(force, "rccl") == 0 : (n_devices > 1 && !duplicates);
    const bool want_peer = force ? std::strcmp(force, "peer") == 0 : false;
    m->transport = n_devices == 1 ? Transport::LOCAL : Transport::PEER;
    if(force && !want_rccl && !want_peer && std::strcmp(force, "local") != 0) {
        destroy_impl(m);
        return mfail(nullptr, WF_HIP_ERR_INVALID, "WF_HIP_MULTI_TRANSPORT=%s: expected rccl or peer", force);
    }
    if(want_peer)
        m->transport = Transport::PEER;
    if(m->per != 0 && want_rccl) {
        Rccl &r = rccl();
        if(!r.ok)
            m->transport_note = r.why;
        else {
            std::vector<ncclComm_t> comms(n_devices, nullptr);
            const ncclResult_t e = r.CommInitAll(comms.data(), (int)n_devices, devices);
            if(e == ncclSuccess) {
                for(uint32_t i = 0; i < n_devices; ++i)
                    m->shard[i]->comm = comms[i];
                m->transport = Transport::RCCL;
            } else
                m->transport_note = std::string("ncclCommInitAll failed: ") + r.GetErrorString(e);
        }
        if(m->transport != Transport::RCCL && force) { // asked for by name: do not quietly use something else
            const std::string why = m->transport_note;
            destroy_impl(m);
            return mfail(nullptr, WF_HIP_ERR_RUNTIME, "WF_HIP_MULTI_TRANSPORT=rccl: %s", why.c_str());
        }
    }
    if(m->per != 0) {
        rc = run_all(m, [m](uint32_t i) {
            Shard &s = *m->shard[i];
            // HIP multiplexes a process's streams of one priority onto four hardware queues per device, which a handle's lanes fill: a
            // gather stream that shares an in-order hardware queue with a lane puts that lane's next tick behind whatever the exchange
            // of the last one enqueued (two shards on one device, peer copies: +36 us per tick), and one with a queue of its own
            // (another priority) runs its kernels INTO the tick, which fills the CUs in whole rounds (+77 us; both traced:
            // profiles/r05h_peer_trace*.txt).  The exchange therefore enqueues no kernel where it can avoid it (mirrors above); RCCL's
            // own kernels remain -- WF_HIP_MULTI_GATHER_PRIORITY=high lets a node check try them on a queue of their own.
            {
                const char *pe = std::getenv("WF_HIP_MULTI_GATHER_PRIORITY");
                if(pe && std::strcmp(pe, "high") == 0) {
                    int least = 0, greatest = 0;
                    WF_MHIP(s, hipDeviceGetStreamPriorityRange(&least, &greatest));
                    WF_MHIP(s, hipStreamCreateWithPriority(&s.gstream, hipStreamNonBlocking, greatest));
                } else
                    WF_MHIP(s, hipStreamCreateWithFlags(&s.gstream, hipStreamNonBlocking));
            }
            const size_t per = m->per;
            for(int k = 0; k < 2; ++k) {
                WF_MHIP(s, hipEventCreateWithFlags(&s.ev_sent[k], hipEventDisableTiming));
                WF_MHIP(s, hipEventCreateWithFlags(&s.ev_done[k], hipEventDisableTiming));
                WF_MHIP(s, hipMalloc((void **)&s.gathered[k], (size_t)m->total * per * sizeof(float)));
                if(m->transport != Transport::LOCAL) {
                    WF_MHIP(s, hipMalloc((void **)&s.send[k], (size_t)m->largest * per * sizeof(float)));
                    WF_MHIP(s, hipMemset(s.send[k], 0, (size_t)m->largest * per * sizeof(float)));
                }
                if(m->transport == Transport::RCCL && m->ragged)
                    WF_MHIP(s, hipMalloc((void **)&s.recv_pad[k], (size_t)m->n * m->largest * per * sizeof(float)));
            }
            if(m->transport == Transport::PEER) // direct xGMI stores where the link allows; hipMemcpyPeerAsync stages otherwise
                for(uint32_t j = 0; j < m->n; ++j) {
                    const int other = m->shard[j]->device;
                    int can = 0;
                    bool ok = other == s.device;
                    if(!ok && hipDeviceCanAccessPeer(&can, s.device, other) == hipSuccess && can) {
                        // (refused: not fatal for the copies, which are then staged -- but kernel stores to that device's memory would
                        // fault: direct peer stores need the mapping to exist, for every ordered pair)
                        const hipError_t pe = hipDeviceEnablePeerAccess(other, 0);
                        ok = pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled;
                        (void)hipGetLastError();
                    }
                    s.peer_ok[j] = ok;
                }
            return (int)WF_HIP_OK;
        });
        // zero-copy gathers, once every device's buffers exist: the tick kernel writes the slot's send buffer (LOCAL: the result)
        // itself -- or, peer transport on devices that can all address each other, this shard's slice of EVERY device's result:
        // no copy, no kernel on the gather streams at all.  The batches whose display comes from a kernel of its own keep the copy
        // behind the tick (WF_HIP_MULTI_MIRROR=0: the copy for everybody, =send: no direct peer stores; A/B aids)
        if(rc == WF_HIP_OK) {
            // Which shards' tick kernels write the exchange's buffers themselves (wf_hip_set_bars_mirrors) instead of a device copy
            // behind the tick: by default only where that saves the PEER copies -- every device addresses every other, the slices
            // go straight into every device's result.  Into a send buffer (RCCL; peer copies) or the one device's own result the
            // kernel-side stores bought nothing once the kernels came in display-specific instantiations: the copy behind the tick
            // 0.651 against 0.618 through this group, 0.649 against 0.650 from a torch process (profiles/r06o_gather_mirror_one_ab.txt).
            // WF_HIP_MULTI_MIRROR: 1 = the kernels write wherever they can (round 5's default), send = that without the direct peer
            // stores, 0 = the copy for everybody (A/B aids).
            const char *e = std::getenv("WF_HIP_MULTI_MIRROR");
            bool all_peer = m->transport == Transport::PEER && m->n <= 8 && !(e && std::strcmp(e, "send") == 0);
            for(uint32_t i = 0; i < m->n && all_peer; ++i) // hipDeviceEnablePeerAccess succeeded (or had before) on device i for device j
                for(uint32_t j = 0; j < m->n && all_peer; ++j)
                    all_peer = m->shard[i]->peer_ok[j];
            if(m->transport == Transport::PEER && !all_peer && m->transport_note.empty())
                m->transport_note = "peer access is not enabled between every pair of devices: the bars travel by hipMemcpyPeerAsync";
            rc = run_all(m, [m, e, all_peer](uint32_t i) {
                Shard &s = *m->shard[i];
                if(e ? e[0] == '0' : !all_peer)
                    return (int)WF_HIP_OK;
                int mrc;
                if(all_peer) {
                    void *set0[8], *set1[8];
                    for(uint32_t j = 0; j < m->n; ++j) {
                        set0[j] = m->shard[j]->gathered[0] + (size_t)s.first * m->per;
                        set1[j] = m->shard[j]->gathered[1] + (size_t)s.first * m->per;
                    }
                    mrc = wf_hip_set_bars_mirrors(s.h, m->n, set0, set1);
                } else {
                    float *const *tgt = (m->transport == Transport::LOCAL) ? s.gathered : s.send;
                    void *b0 = tgt[0], *b1 = tgt[1];
                    mrc = wf_hip_set_bars_mirrors(s.h, 1, &b0, &b1);
                }
                if(mrc == WF_HIP_OK) {
                    s.mirror = true;
                    s.direct = all_peer;
                } else if(mrc != WF_HIP_ERR_UNSUPPORTED) {
                    s.err = wf_hip_last_error(s.h);
                    return mrc;
                }
                return (int)WF_HIP_OK;
            });
        }
        if(rc) {
            g_multi_create_error = m->last_error;
            destroy_impl(m);
            return rc;
        }
    }
    // not an error, but what a node check wants to see: why the group did not get the transport it would have picked
    // (wf_hip_multi_last_error(m) right after create; the next failing call overwrites it)
    m->last_error = m->transport_note;
    *out = m;
    return WF_HIP_OK;
}

void wf_hip_multi_destroy(wf_hip_multi *m) { destroy_impl(m); }

wf_hip *wf_hip_multi_shard(wf_hip_multi *m, uint32_t i, int *device, uint32_t *first, uint32_t *count)
{
    if(m == nullptr || i >= m->n)
        return nullptr;
    const Shard &s = *m->shard[i];
    if(device) *device = s.device;
    if(first) *first = s.first;
    if(count) *count = s.count;
    return s.h;
}

int wf_hip_multi_push_audio(wf_hip_multi *m, uint32_t first, uint32_t count, const float *samples, uint32_t frames)
{
    int rc = check_range(m, first, count);
    if(rc)
        return rc;
    if(samples == nullptr)
        return mfail(m, WF_HIP_ERR_INVALID, "samples is NULL");
    const size_t per_stream = (size_t)wf_hip_capture_channels(m->shard[0]->h) * frames;
    return run_all(m, [=](uint32_t i) {
        Shard &s = *m->shard[i];
        uint32_t lf, lc, off;
        if(!overlap(s, first, count, &lf, &lc, &off))
            return (int)WF_HIP_OK;
        return wf_hip_push_audio(s.h, lf, lc, samples + (size_t)off * per_stream, frames);
    });
}

int wf_hip_multi_push_synth(wf_hip_multi *m, uint32_t first, uint32_t count, uint64_t seed, uint32_t stream_id0, uint64_t index0, uint32_t frames)
{
    int rc = check_range(m, first, count);
    if(rc)
        return rc;
    return run_all(m, [=](uint32_t i) {
        Shard &s = *m->shard[i];
        uint32_t lf, lc, off;
        if(!overlap(s, first, count, &lf, &lc, &off))
            return (int)WF_HIP_OK;
        return wf_hip_push_synth(s.h, lf, lc, seed, stream_id0 + off, index0, frames);
    });
}

int wf_hip_multi_set_hidden(wf_hip_multi *m, uint32_t first, uint32_t count, const uint8_t *mask)
{
    int rc = check_range(m, first, count);
    if(rc)
        return rc;
    if(mask == nullptr)
        return mfail(m, WF_HIP_ERR_INVALID, "mask is NULL");
    return run_all(m, [=](uint32_t i) {
        Shard &s = *m->shard[i];
        uint32_t lf, lc, off;
        if(!overlap(s, first, count, &lf, &lc, &off))
            return (int)WF_HIP_OK;
        return wf_hip_set_hidden(s.h, lf, lc, mask + off);
    });
}

int wf_hip_multi_reset(wf_hip_multi *m, uint32_t first, uint32_t count)
{
    int rc = check_range(m, first, count);
    if(rc)
        return rc;
    return run_all(m, [=](uint32_t i) {
        Shard &s = *m->shard[i];
        uint32_t lf, lc, off;
        if(!overlap(s, first, count, &lf, &lc, &off))
            return (int)WF_HIP_OK;
        return wf_hip_reset(s.h, lf, lc);
    });
}

int wf_hip_multi_tick(wf_hip_multi *m, const wf_hip_tick_params *p)
{
    if(m == nullptr || p == nullptr)
        return WF_HIP_ERR_INVALID;
    const wf_hip_tick_params q = *p;
    return run_all(m, [m, q](uint32_t i) { return wf_hip_tick(m->shard[i]->h, &q); });
}

int wf_hip_multi_sync(wf_hip_multi *m)
{
    if(m == nullptr)
        return WF_HIP_ERR_INVALID;
    return run_all(m, [m](uint32_t i) {
        Shard &s = *m->shard[i];
        const int rc = wf_hip_sync(s.h);
        if(rc)
            return rc;
        if(s.gstream)
            WF_MHIP(s, hipStreamSynchronize(s.gstream));
        return (int)WF_HIP_OK;
    });
}

int wf_hip_multi_read(wf_hip_multi *m, wf_hip_output what, uint32_t first, uint32_t count, void *out)
{
    int rc = check_range(m, first, count);
    if(rc)
        return rc;
    if(out == nullptr)
        return mfail(m, WF_HIP_ERR_INVALID, "out is NULL");
    const size_t per_stream = wf_hip_output_bytes(m->shard[0]->h, what);
    if(per_stream == 0) // (the shard's own error text: why the batch has no such output)
        return run_all(m, [=](uint32_t i) { return i == 0 ? wf_hip_read(m->shard[0]->h, what, 0, 1, out) : (int)WF_HIP_OK; });
    return run_all(m, [=](uint32_t i) {
        Shard &s = *m->shard[i];
        uint32_t lf, lc, off;
        if(!overlap(s, first, count, &lf, &lc, &off))
            return (int)WF_HIP_OK;
        return wf_hip_read(s.h, what, lf, lc, static_cast<char *>(out) + (size_t)off * per_stream);
    });
}

int wf_hip_multi_allgather_bars(wf_hip_multi *m)
{
    int rc = gather_check(m);
    if(rc)
        return rc;
    const uint32_t k = m->gathers & 1u;
    rc = run_all(m, [m, k](uint32_t i) { return gather_issue(m, i, k); });
    // (run_all returning is the host barrier between the halves: every ev_sent of this slot has been recorded)
    if(rc == WF_HIP_OK)
        rc = slots_agree(m);
    if(rc == WF_HIP_OK)
        rc = run_all(m, [m, k](uint32_t i) { return gather_complete(m, i, k); });
    if(rc) { // some shards have enqueued their half: see gather_fail
        gather_fail(m);
        return rc;
    }
    ++m->gathers;
    m->last_slot = (int)m->shard[0]->cur_slot;
    return WF_HIP_OK;
}

const float *wf_hip_multi_gathered_device(wf_hip_multi *m, uint32_t i)
{
    if(m == nullptr || i >= m->n || m->last_slot < 0)
        return nullptr;
    return m->shard[i]->gathered[m->last_slot];
}

#ifdef WF_DEV_BUILD
// Test aid: shard `shard`'s next gather reports a failure before it enqueues anything -- what a failed wait or copy on one
// device looks like to the others, which have their half of the exchange in flight by then.
int wf_hip_multi_debug_fail_next_gather(wf_hip_multi *m, uint32_t shard)
{
    if(m == nullptr || shard >= m->n)
        return WF_HIP_ERR_INVALID;
    m->debug_fail_shard = (int)shard;
    return WF_HIP_OK;
}
#endif

void *wf_hip_multi_gather_stream(wf_hip_multi *m, uint32_t i) { return (m && i < m->n) ? m->shard[i]->gstream : nullptr; }

int wf_hip_multi_read_gathered(wf_hip_multi *m, uint32_t i, float *out)
{
    int rc = gather_check(m);
    if(rc)
        return rc;
    if(i >= m->n || out == nullptr)
        return mfail(m, WF_HIP_ERR_INVALID, "device index %u outside 0..%u, or out is NULL", i, m->n);
    if(m->last_slot < 0)
        return mfail(m, WF_HIP_ERR_INVALID, "no complete gather to read (none issued yet, or the last timed run failed half way)");
    const uint32_t k = (uint32_t)m->last_slot;
    Shard &s = *m->shard[i];
    s.worker.post([m, &s, k, out] {
        WF_MHIP(s, hipMemcpyAsync(out, s.gathered[k], (size_t)m->total * m->per * sizeof(float), hipMemcpyDeviceToHost, s.gstream));
        WF_MHIP(s, hipStreamSynchronize(s.gstream));
        return (int)WF_HIP_OK;
    });
    rc = s.worker.wait();
    if(rc) {
        m->last_error = "device " + std::to_string(s.device) + ": " + s.err;
        s.err.clear();
    }
    return rc;
}

int wf_hip_multi_time_ticks(wf_hip_multi *m, const wf_hip_tick_params *p, uint32_t ticks, uint32_t hop, int gather, float *avg_ms,
                            float *per_device_ms)
{
    if(m == nullptr || p == nullptr || ticks == 0 || avg_ms == nullptr)
        return WF_HIP_ERR_INVALID;
    if(gather) {
        const int rc = gather_check(m);
        if(rc)
            return rc;
    }
    const uint32_t period = hop ? p->delay_frames / hop + 1 : ticks;
    const wf_hip_tick_params p0 = *p;
    std::vector<float> ms(m->n, 0.0f);
    std::vector<int> status(m->n, WF_HIP_OK);
    // the halves of a peer gather need every device's events recorded in between: a barrier among the workers.  A worker that
    // fails keeps arriving at the barriers (doing nothing) so that the others do not wait for it for ever.
    std::barrier sync((std::ptrdiff_t)m->n);
    std::atomic<bool> any_failed{false};
    const uint32_t k0 = m->gathers;
    int rc = run_all(m, [&, m](uint32_t i) {
        Shard &s = *m->shard[i];
        int &st = status[i];
        st = wf_hip_time_begin(s.h);
        wf_hip_tick_params q = p0;
        for(uint32_t t = 0; t < ticks; ++t) {
            q.delay_frames = p0.delay_frames - (t % period) * hop;
            if(st == WF_HIP_OK)
                st = wf_hip_tick(s.h, &q);
            if(gather) {
                const uint32_t k = (k0 + t) & 1u;
                if(st == WF_HIP_OK)
                    st = gather_issue(m, i, k);
                if(m->transport == Transport::PEER)
                    sync.arrive_and_wait();
                if(st == WF_HIP_OK)
                    st = gather_complete(m, i, k);
            }
            if(st != WF_HIP_OK)
                any_failed.store(true, std::memory_order_relaxed);
        }
        // Nothing above waits on the host.  Below it does -- for streams that, after a failure anywhere, may be waiting for a
        // collective one rank never joined: every worker first learns whether all of them got through (gather_fail, called by
        // the API thread once the workers are back, is what releases those streams).
        if(gather) {
            sync.arrive_and_wait();
            if(any_failed.load(std::memory_order_relaxed))
                return st; // (a bystander returns OK: the failing shard's own status and text are the call's)
        }
        if(st == WF_HIP_OK)
            st = wf_hip_time_end(s.h, &ms[i]);
        if(st == WF_HIP_OK && s.gstream && hipStreamSynchronize(s.gstream) != hipSuccess)
            st = WF_HIP_ERR_RUNTIME;
        return st;
    });
    if(gather) {
        if(rc == WF_HIP_OK && !any_failed.load()) {
            m->gathers = k0 + ticks;
            m->last_slot = (int)m->shard[0]->cur_slot;
        } else {
            // some gathers of the run completed on some shards and overwrote the slots: no slot holds a result every device agrees
            // on -- wf_hip_multi_gathered_device / _read_gathered say so until the next complete gather
            m->last_slot = -1;
        }
    }
    if(rc) {
        // A shard that stopped in a tick (nothing to do with the exchange) left the others with a collective it never joined as
        // well -- from that tick on; the communicators are aborted in both cases, but only because peers are stuck, and the text
        // kept is the failing call's
        if(gather && any_failed.load())
            gather_fail(m);
        return rc;
    }
    float worst = 0.0f;
    for(uint32_t i = 0; i < m->n; ++i) {
        ms[i] /= (float)ticks;
        worst = std::max(worst, ms[i]);
        if(per_device_ms)
            per_device_ms[i] = ms[i];
    }
    *avg_ms = worst;
    return WF_HIP_OK;
}

} // extern "C"
