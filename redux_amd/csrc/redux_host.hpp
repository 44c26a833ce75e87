// redux_host.hpp -- the host-pointer side of the C ABI (redux_encode_blocks, redux_decode_blocks,
// redux_compress, redux_decompress): what a Rust / C / ctypes caller holding plain host memory binds.
//
// A call is a pipeline over CHUNKS of whole blocks:
//
//   caller memory --(N CPU threads)--> pinned ring --(H2D DMA)--> chunk slot in HBM
//        --> coder kernels of that chunk (its own HIP stream) --> dense chunk output
//        --(D2H DMA)--> caller memory
//
// Why it looks like this (numbers: tools/ubench/pcie.hip on the MI355X box, profiles/r02_host_abi/):
//   * hipMalloc / hipFree / hipHostMalloc cost milliseconds to hundreds of milliseconds: everything is
//     allocated once, kept in a per-device context and only ever grown (redux_host_release() frees it).
//     The context is guarded by one mutex: concurrent calls are safe and serialise.
//   * H2D from pageable memory through the runtime's own staging runs at 29 GB/s, from pinned memory at
//     57 GB/s, and 4 CPU threads fill a pinned buffer at 76 GB/s: the input is staged by a small pool of
//     copy threads (alive for the duration of the call) through a ring of pinned pieces.
//   * a block is a serial chain: the coder kernel of ANY number of 64 KiB blocks takes ~12 ms (decode
//     ~28 ms).  Eight chunks are therefore in flight on eight streams -- a chunk of 128 MiB is 32
//     workgroups, the chip holds 1024 -- so that the PCIe transfers of later chunks hide under the
//     kernels of earlier ones and only ONE kernel latency is exposed at the end of the call.
//   * the way back is one DMA per chunk straight into the caller's (pageable) memory, issued by a drain
//     thread when the chunk's event fires, so that draining chunk k never delays staging and launching
//     chunk k+8.  Nothing that depends on a running kernel is ever put into a copy queue: the SDMA
//     queues are in order, and a 16 KiB result copy waiting for its kernel blocks every copy behind it.
//
// Included by redux_hip.hip (one translation unit) after the _dev entry points it drives.
#pragma once

#include "../../include/redux_hip.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>

namespace redux {
namespace host {

// The HIP runtime multiplexes the streams of a process onto FOUR hardware queues per priority level
// (GPU_MAX_HW_QUEUES), the null stream holds one of the normal-priority ones, and streams that share a
// queue run one after the other (tools/ubench/queues.hip: 3 default-priority streams run at once, the
// 4th waits; 4 high- + 4 low-priority streams all run at once).  Measured with 16 default-priority
// compute streams + a copy stream + a drain stream: 2.6 chunk kernels in flight, 13 GB/s.
// So: eight streams, four created at the highest and four at the lowest priority (none at the
// application's own level), chunk k does EVERYTHING (its H2D, its kernels, its D2H) on stream k % 8,
// and chunks are large enough that eight kernels in flight outrun the PCIe link:
//   chunk bytes / PCIe rate >= kernel latency / 8   ->  >= 78 MB (encode, 12.5 ms), >= 175 MB (decode, 28 ms).
constexpr int      kSlots      = 8;             // chunk slots in HBM = streams (a slot is reused once its chunk has been drained)
constexpr int      kStreams    = kSlots;
constexpr int      kPieces     = 8;             // pinned staging ring
constexpr uint64_t kPieceBytes = 16ull << 20;
constexpr uint64_t kEncChunkMax = 128ull << 20; // input bytes per chunk
constexpr uint64_t kDecChunkMax = 256ull << 20; // output bytes per chunk
constexpr uint64_t kChunkMin    = 16ull << 20;
constexpr int      kCopyThreads = 4;            // incl. the calling thread

// ---- N threads that copy one buffer together -------------------------------------------------
class CopyPool {
    int                      n_;
    std::vector<std::thread> th_;
    std::mutex               m_;
    std::condition_variable  go_, done_;
    uint64_t                 gen_ = 0;
    int                      pending_ = 0;
    bool                     stop_ = false;
    char                    *d_ = nullptr;
    const char              *s_ = nullptr;
    size_t                   len_ = 0;

    void worker(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> l(m_);
            go_.wait(l, [&] { return stop_ || gen_ != seen; });
            if (stop_)
                return;
            seen = gen_;
            char *d = d_; const char *s = s_; const size_t len = len_;
            l.unlock();
            const size_t a = len * (size_t)(id + 1) / (size_t)(n_ + 1), b = len * (size_t)(id + 2) / (size_t)(n_ + 1);
            memcpy(d + a, s + a, b - a);
            l.lock();
            if (--pending_ == 0)
                done_.notify_one();
        }
    }

public:
    explicit CopyPool(int helpers) : n_(helpers)
    {
        for (int i = 0; i < n_; i++)
            th_.emplace_back([this, i] { worker(i); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        go_.notify_all();
        for (auto &t : th_)
            t.join();
    }
    void copy(void *dst, const void *src, size_t len)
    {
        if (len < (1u << 20) || n_ == 0) {
            memcpy(dst, src, len);
            return;
        }
        {
            std::lock_guard<std::mutex> l(m_);
            d_ = (char *)dst; s_ = (const char *)src; len_ = len; pending_ = n_; gen_++;
        }
        go_.notify_all();
        memcpy(dst, src, len / (size_t)(n_ + 1)); // slice 0 on the calling thread
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return pending_ == 0; });
    }
};

// ---- persistent per-device context -----------------------------------------------------------
struct Buf {
    void  *p = nullptr;
    size_t cap = 0;
};

struct Slot {          // one chunk in flight
    Buf d_in, d_ws, d_out, d_off, d_sz, d_st, d_sum, d_used, d_tab; // device
    Buf h_off, h_sz, h_st, h_sum, h_used, h_tab;                    // pinned mirrors of the small arrays
    hipEvent_t done = nullptr;                               // recorded after the chunk's kernels and small D2H copies
};

struct Ctx {
    std::mutex  mu;
    // Both only ever written with mu held: `want` by the call that has just locked the context (the device it is to run on),
    // `device` by ctx_init_locked / ctx_teardown_locked (the device the streams, events and buffers below were created on).
    int         want = -1;
    int         device = -1;
    bool        ready = false;
    hipStream_t stream[kStreams] = {};
    hipStream_t drain = nullptr; // bulk D2H, issued by the drain thread once a chunk's event has fired
    Slot        slot[kSlots];
    void       *piece[kPieces] = {};      // input staging ring
    hipEvent_t  piece_free[kPieces] = {};
    uint64_t    allocs = 0; // hipMalloc / hipHostMalloc calls so far (redux_host_allocations)
    // timeline of the last call, seconds since its start: per chunk {staging begins, device work enqueued,
    // kernels done (drain thread saw the event), results in caller memory}; trace[0..3] of chunk 0 etc.
    std::vector<double> trace;
    double              t0 = 0;
};

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static Ctx g_ctx[16];

#define HOST_TRY(expr)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "redux_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_),       \
                    __FILE__, __LINE__);                                                               \
            return REDUX_IO_ERROR;                                                                     \
        }                                                                                              \
    } while (0)

static int grow_dev(Ctx &c, Buf &b, size_t need)
{
    if (b.cap >= need)
        return REDUX_OK;
    if (b.p)
        HOST_TRY(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    const size_t cap = (need + need / 8 + 4095) & ~(size_t)4095;
    HOST_TRY(hipMalloc(&b.p, cap));
    c.allocs++;
    b.cap = cap;
    return REDUX_OK;
}

static int grow_pinned(Ctx &c, Buf &b, size_t need)
{
    if (b.cap >= need)
        return REDUX_OK;
    if (b.p)
        HOST_TRY(hipHostFree(b.p));
    b.p = nullptr; b.cap = 0;
    const size_t cap = (need + need / 8 + 4095) & ~(size_t)4095;
    HOST_TRY(hipHostMalloc(&b.p, cap, hipHostMallocDefault));
    c.allocs++;
    b.cap = cap;
    return REDUX_OK;
}

// the context of HIP's current device; its streams, events and staging ring are created by the first call that
// holds its mutex (ctx_init_locked) and torn down by redux_host_release under the same mutex
static int ctx_of_current_device(Ctx **out, int *dev_out)
{
    int dev = 0;
    HOST_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16)
        return REDUX_UNSUPPORTED;
    *out     = &g_ctx[dev];
    *dev_out = dev;
    return REDUX_OK;
}

static void ctx_teardown_locked(Ctx &c);

// c.mu is held and c.want says which device the call runs on.  A context that was built for another device -- the fleet
// was reconfigured, or fleet entry i and default-mode device d share g_ctx[] -- is torn down first: its streams, events,
// HBM and pinned buffers belong to the device they were created on.
static int ctx_init_locked(Ctx &c)
{
    if (c.ready && c.device != c.want)
        ctx_teardown_locked(c); // (leaves HIP's current device at c.device: set again below)
    HOST_TRY(hipSetDevice(c.want));
    if (c.ready)
        return REDUX_OK;
    int pri_lo = 0, pri_hi = 0; // numerically: lo = least urgent, hi = most urgent
    HOST_TRY(hipDeviceGetStreamPriorityRange(&pri_lo, &pri_hi));
    for (int i = 0; i < kStreams; i++)
        HOST_TRY(hipStreamCreateWithPriority(&c.stream[i], hipStreamNonBlocking, (i & 1) ? pri_hi : pri_lo));
    HOST_TRY(hipStreamCreateWithFlags(&c.drain, hipStreamNonBlocking));
    for (int i = 0; i < kSlots; i++)
        HOST_TRY(hipEventCreateWithFlags(&c.slot[i].done, hipEventDisableTiming));
    for (int i = 0; i < kPieces; i++) {
        HOST_TRY(hipHostMalloc(&c.piece[i], kPieceBytes, hipHostMallocDefault));
        HOST_TRY(hipEventCreateWithFlags(&c.piece_free[i], hipEventDisableTiming));
        c.allocs++;
    }
    c.device = c.want;
    c.ready  = true;
    return REDUX_OK;
}

static void free_buf_dev(Buf &b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
static void free_buf_pin(Buf &b) { if (b.p) (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }

// frees everything the context holds on the device it was built for (c.mu is held)
static void ctx_teardown_locked(Ctx &c)
{
    if (!c.ready)
        return;
    (void)hipSetDevice(c.device);
    for (int i = 0; i < kStreams; i++) // (a call in flight holds c.mu, so these are idle: belt and braces)
        if (c.stream[i]) (void)hipStreamSynchronize(c.stream[i]);
    for (Slot &s : c.slot) {
        free_buf_dev(s.d_in); free_buf_dev(s.d_ws); free_buf_dev(s.d_out); free_buf_dev(s.d_off); free_buf_dev(s.d_sz);
        free_buf_dev(s.d_st); free_buf_dev(s.d_sum); free_buf_dev(s.d_used); free_buf_dev(s.d_tab);
        free_buf_pin(s.h_off); free_buf_pin(s.h_sz); free_buf_pin(s.h_st); free_buf_pin(s.h_sum); free_buf_pin(s.h_used);
        free_buf_pin(s.h_tab);
        if (s.done) (void)hipEventDestroy(s.done);
        s.done = nullptr;
    }
    for (int i = 0; i < kPieces; i++) {
        if (c.piece[i]) (void)hipHostFree(c.piece[i]);
        if (c.piece_free[i]) (void)hipEventDestroy(c.piece_free[i]);
        c.piece[i] = nullptr; c.piece_free[i] = nullptr;
    }
    for (int i = 0; i < kStreams; i++) {
        if (c.stream[i]) (void)hipStreamDestroy(c.stream[i]);
        c.stream[i] = nullptr;
    }
    if (c.drain) (void)hipStreamDestroy(c.drain);
    c.drain = nullptr;
    c.ready = false;
}

// A call on an unusually large shape (one block of hundreds of MiB, a generous decode capacity) leaves slot buffers behind
// that no ordinary call needs: the context keeps what the chunk pipeline itself can ask for (256 MiB of payload + workspace)
// and gives back anything above kTrimBytes when such a call ends.
constexpr size_t kTrimBytes = 1ull << 30;
static void ctx_trim_locked(Ctx &c)
{
    if (!c.ready)
        return;
    for (Slot &s : c.slot)
        for (Buf *b : {&s.d_in, &s.d_ws, &s.d_out})
            if (b->cap > kTrimBytes)
                free_buf_dev(*b);
}

static int ctx_release_all()
{
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev); // restored below: freeing another device's context must not move the caller
    struct Restore {
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{caller_dev};
    for (Ctx &c : g_ctx) {
        std::lock_guard<std::mutex> lc(c.mu); // waits for a call in flight on that device
        ctx_teardown_locked(c);
    }
    return REDUX_OK;
}

// ---- hand-over between the issuing thread and the drain thread -----------------------------------
struct Handover {
    std::mutex              m;
    std::condition_variable cv;
    uint64_t                issued = 0;  // chunks whose device work has been enqueued
    uint64_t                drained = 0; // chunks whose results are in the caller's memory
    int                     error = REDUX_OK;
    bool                    abort = false;
};

// stage `len` host bytes into the slot's device input at byte offset 0, through the pinned ring, on `s`
static int stage_h2d(Ctx &c, CopyPool &pool, uint64_t &piece_no, void *d_dst, const uint8_t *src, uint64_t len, hipStream_t s)
{
    for (uint64_t o = 0; o < len; o += kPieceBytes) {
        const uint64_t n = len - o < kPieceBytes ? len - o : kPieceBytes;
        const int      k = (int)(piece_no % kPieces);
        if (piece_no >= (uint64_t)kPieces)
            HOST_TRY(hipEventSynchronize(c.piece_free[k])); // the H2D that last read this piece has finished
        pool.copy(c.piece[k], src + o, n);
        HOST_TRY(hipMemcpyAsync((uint8_t *)d_dst + o, c.piece[k], n, hipMemcpyHostToDevice, s));
        HOST_TRY(hipEventRecord(c.piece_free[k], s));
        piece_no++;
    }
    return REDUX_OK;
}

// `len` device bytes -> caller memory on the drain stream.  The destination is pageable: the runtime
// pins it on the fly and DMAs straight into it at ~55 GB/s -- no second CPU pass over host DRAM, whose
// bandwidth the staging threads and both DMA directions already share.  (An own pinned ring + copy
// threads on this side measured 15 % slower on 4 GiB for that reason.)
static int drain_d2h(Ctx &c, uint8_t *dst, const void *d_src, uint64_t len)
{
    HOST_TRY(hipMemcpyAsync(dst, d_src, len, hipMemcpyDeviceToHost, c.drain));
    HOST_TRY(hipStreamSynchronize(c.drain));
    return REDUX_OK;
}

// ---- which contexts a host-pointer call runs on ------------------------------------------------
// Default: the context of HIP's current device (g_ctx[device id]).  redux_host_set_devices() installs a FLEET instead:
// context i serves device fleet[i] -- ids may repeat, each entry is its own context with its own streams and buffers --
// and every redux_encode_blocks / redux_decode_blocks call deals its chunks round-robin over all of them.  The data starts
// and ends in host memory, so nothing is exchanged between devices: each context is fed over its own PCIe link.
static std::mutex       g_fleet_mu;
static std::vector<int> g_fleet;                                    // empty: current device only
static std::atomic<uint64_t> g_chunk_min{0}, g_chunk_max{0};        // test hook (redux_host_set_chunk_bytes): 0 = the defaults

// Which contexts a call runs on, and the device each is to run on.  Nothing of a context is touched here: the caller locks
// every context's mutex (always in this order) and only then records the device in it (take_contexts).
static int contexts_for_call(std::vector<Ctx *> &out, std::vector<int> &want)
{
    std::lock_guard<std::mutex> l(g_fleet_mu);
    if (g_fleet.empty()) {
        Ctx *cp  = nullptr;
        int  dev = 0;
        int  rc  = ctx_of_current_device(&cp, &dev);
        if (rc != REDUX_OK)
            return rc;
        out.push_back(cp);
        want.push_back(dev);
        return REDUX_OK;
    }
    for (size_t i = 0; i < g_fleet.size(); i++) {
        out.push_back(&g_ctx[i]);
        want.push_back(g_fleet[i]);
    }
    return REDUX_OK;
}

// contexts_for_call + their mutexes, taken in context order (two calls cannot deadlock); each context then knows its device
static int take_contexts(std::vector<Ctx *> &ctx, std::vector<std::unique_lock<std::mutex>> &locks)
{
    std::vector<int> want;
    int rc = contexts_for_call(ctx, want);
    if (rc != REDUX_OK)
        return rc;
    for (size_t i = 0; i < ctx.size(); i++) {
        locks.emplace_back(ctx[i]->mu);
        ctx[i]->want = want[i];
    }
    return REDUX_OK;
}

// The fleet changes under g_fleet_mu, held across the assignment AND the release of the old contexts: a call that asks for
// its contexts meanwhile waits, then sees the new fleet.  (A call that already holds contexts of the old fleet finishes
// first -- the release waits for its mutexes -- and whatever it leaves built for an old device is rebuilt by the next
// call that finds the device changed: ctx_init_locked.)
static int set_devices(const int32_t *ids, uint32_t n)
{
    if (n > 16 || (n && !ids))
        return REDUX_INVALID_INPUT;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess)
        return REDUX_IO_ERROR;
    for (uint32_t i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= count)
            return REDUX_INVALID_INPUT;
    std::lock_guard<std::mutex> l(g_fleet_mu);
    g_fleet.assign(ids, ids + n);
    ctx_release_all(); // contexts are bound to a device when they are built: start from none
    return REDUX_OK;
}

// blocks per chunk: whole 64-block waves; an eighth of a context's share of the call (eight chunks in flight per
// context), within [kChunkMin, chunk_max] bytes of payload
static uint64_t chunk_blocks_for(uint64_t nblocks, uint32_t block_size, uint64_t chunk_max, size_t nctx)
{
    const uint64_t lo = g_chunk_min.load() ? g_chunk_min.load() : kChunkMin;
    const uint64_t hi = g_chunk_max.load() ? g_chunk_max.load() : chunk_max;
    uint64_t bytes = (nblocks * (uint64_t)block_size + kSlots * nctx - 1) / (kSlots * nctx);
    bytes = bytes < lo ? lo : bytes > hi ? hi : bytes;
    uint64_t cb = (bytes + block_size - 1) / block_size;
    cb = (cb + 63) / 64 * 64;
    return cb < nblocks ? cb : nblocks;
}

// what the contexts of one call share
struct Job {
    std::mutex              m;
    std::condition_variable cv;
    bool                    abort = false;
    int                     error = REDUX_OK;
    uint64_t                bad_chunk = ~0ull; // first chunk (in block order) with a non-OK block, and that status
    int                     bad_status = REDUX_OK;
    // encode only: where a chunk's streams go in the dense output is known once every earlier chunk's size is
    std::vector<uint64_t> total, prefix; // prefix[k] = sum of total[0..k): valid for k <= prefix_n
    std::vector<char>     known;
    uint64_t              prefix_n = 0;

    void fail(int rc)
    {
        std::lock_guard<std::mutex> l(m);
        if (error == REDUX_OK)
            error = rc;
        abort = true;
        cv.notify_all();
    }
    void note_bad(uint64_t chunk, int st)
    {
        std::lock_guard<std::mutex> l(m);
        if (chunk < bad_chunk) {
            bad_chunk  = chunk;
            bad_status = st;
        }
    }
};

// ================================================================================================
// encode
// ================================================================================================
struct EncCall {
    const redux_params *p;
    const uint8_t      *in;
    uint64_t            in_len;
    uint32_t            block_size;
    uint8_t            *out;
    uint64_t            out_cap;
    uint64_t           *out_offsets;
    int32_t            *block_status;
    uint64_t            nblocks, cb, nchunks, ws_bytes, bound, chunk_in;
};

// the chunks first, first + stride, ... of the call on context c (the calling thread of the call holds c.mu)
static void encode_on_ctx(Ctx &c, const EncCall &E, Job &J, uint64_t first, uint64_t stride)
{
    int rc = ctx_init_locked(c); // (makes c.want HIP's current device on this thread)
    if (rc != REDUX_OK)
        return J.fail(rc);
    const uint64_t mine = first < E.nchunks ? (E.nchunks - first + stride - 1) / stride : 0; // chunks of this context
    const uint64_t first_len = E.chunk_in < E.in_len ? E.chunk_in : E.in_len;
    const int      nslots = (int)(mine < (uint64_t)kSlots ? mine : (uint64_t)kSlots);
    for (int i = 0; i < nslots; i++) {
        Slot &s = c.slot[i];
        if ((rc = grow_dev(c, s.d_in, first_len + 16)) || (rc = grow_dev(c, s.d_ws, E.ws_bytes + 256)) || (rc = grow_dev(c, s.d_out, E.bound + 16)) ||
            (rc = grow_dev(c, s.d_off, (E.cb + 1) * 8)) || (rc = grow_dev(c, s.d_st, E.cb * 4)) || (rc = grow_dev(c, s.d_sum, 8)) ||
            (rc = grow_pinned(c, s.h_off, (E.cb + 1) * 8)) || (rc = grow_pinned(c, s.h_st, E.cb * 4)) || (rc = grow_pinned(c, s.h_sum, 8)))
            return J.fail(rc);
    }
    c.trace.assign(mine * 4, 0.0);
    c.t0 = now_s();
    Handover H; // between this context's issuing thread and its drain thread; j = ordinal of a chunk within this context
    // ---- drain thread: results of chunk k -> caller memory -------------------------------------
    std::thread drain([&] {
        (void)hipSetDevice(c.device);
        for (uint64_t j = 0; j < mine; j++) {
            const uint64_t k = first + j * stride;
            {
                std::unique_lock<std::mutex> l(H.m);
                H.cv.wait(l, [&] { return H.issued > j || H.abort; });
                if (H.abort)
                    return;
            }
            Slot          &s  = c.slot[j % kSlots];
            const uint64_t b0 = k * E.cb, nb = (b0 + E.cb <= E.nblocks ? E.cb : E.nblocks - b0);
            int            err = REDUX_OK;
            hipStream_t st = c.stream[j % kStreams]; // idle once the chunk's event has fired
            if (hipEventSynchronize(s.done) != hipSuccess ||
                hipMemcpyAsync(s.h_off.p, s.d_off.p, (nb + 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(s.h_st.p, s.d_st.p, nb * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(s.h_sum.p, s.d_sum.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                err = REDUX_IO_ERROR;
            c.trace[j * 4 + 2] = now_s() - c.t0;
            const uint64_t *ho    = (const uint64_t *)s.h_off.p;
            const uint64_t  total = err ? 0 : ho[nb];
            uint64_t        base  = 0;
            if (!err) {
                if (((const int32_t *)s.h_sum.p)[0] != REDUX_OK)
                    J.note_bad(k, ((const int32_t *)s.h_sum.p)[0]);
                // publish this chunk's size, then wait for the sizes of all earlier chunks (other contexts' included)
                std::unique_lock<std::mutex> l(J.m);
                J.total[k] = total;
                J.known[k] = 1;
                while (J.prefix_n < E.nchunks && J.known[J.prefix_n]) {
                    J.prefix[J.prefix_n + 1] = J.prefix[J.prefix_n] + J.total[J.prefix_n];
                    J.prefix_n++;
                }
                J.cv.notify_all();
                J.cv.wait(l, [&] { return J.abort || J.prefix_n >= k; });
                if (J.abort)
                    err = -1; // (another context failed: its error is the call's)
                else
                    base = J.prefix[k];
            }
            if (!err && base + total > E.out_cap)
                err = REDUX_OUTPUT_TOO_SMALL;
            if (!err && total)
                err = drain_d2h(c, E.out + base, s.d_out.p, total);
            if (!err) {
                for (uint64_t i = 0; i <= nb; i++)
                    E.out_offsets[b0 + i] = base + ho[i]; // (entry b0 + nb is written again, with the same value, by the next chunk)
                if (E.block_status)
                    memcpy(E.block_status + b0, s.h_st.p, nb * 4);
            }
            c.trace[j * 4 + 3] = now_s() - c.t0;
            if (err > 0)
                J.fail(err);
            std::lock_guard<std::mutex> l(H.m);
            H.drained = j + 1;
            if (err)
                H.abort = true;
            H.cv.notify_all();
            if (err)
                return;
        }
    });

    // ---- issuing side (this thread): stage, H2D, kernels ----------------------------------------
    {
        CopyPool pool(kCopyThreads - 1);
        uint64_t piece_no = 0;
        for (uint64_t j = 0; j < mine; j++) {
            const uint64_t k = first + j * stride;
            {
                std::unique_lock<std::mutex> l(H.m); // the slot's previous chunk must be in the caller's memory
                H.cv.wait(l, [&] { return H.abort || j < (uint64_t)kSlots || H.drained + kSlots > j; });
                if (H.abort)
                    break;
            }
            {
                std::lock_guard<std::mutex> l(J.m);
                if (J.abort)
                    break;
            }
            c.trace[j * 4 + 0] = now_s() - c.t0;
            Slot          &s  = c.slot[j % kSlots];
            hipStream_t    st = c.stream[j % kStreams];
            const uint64_t b0 = k * E.cb, nb = (b0 + E.cb <= E.nblocks ? E.cb : E.nblocks - b0);
            const uint64_t o0 = b0 * (uint64_t)E.block_size;
            const uint64_t len = (o0 + nb * (uint64_t)E.block_size <= E.in_len) ? nb * (uint64_t)E.block_size : E.in_len - o0;
            auto issue = [&]() -> int {
                int r = stage_h2d(c, pool, piece_no, s.d_in.p, E.in + o0, len, st);
                if (r != REDUX_OK)
                    return r;
                HOST_TRY(hipMemsetAsync(s.d_sum.p, 0, 8, st));
                uint8_t *ws = (uint8_t *)(((uintptr_t)s.d_ws.p + 255) & ~(uintptr_t)255);
                r = redux_encode_blocks_dev(E.p, s.d_in.p, len, E.block_size, s.d_out.p, E.bound, s.d_off.p, s.d_st.p, s.d_sum.p, ws,
                                            E.ws_bytes, st);
                if (r != REDUX_OK)
                    return r;
                // (the small result arrays are fetched by the drain thread once the event has fired: a D2H
                // enqueued here would sit in the in-order SDMA queue until this chunk's kernels end, with
                // every other chunk's drain copies stuck behind it -- measured: 15-20 ms stalls)
                HOST_TRY(hipEventRecord(s.done, st));
                return REDUX_OK;
            };
            rc = issue();
            c.trace[j * 4 + 1] = now_s() - c.t0;
            if (rc != REDUX_OK)
                J.fail(rc);
            std::lock_guard<std::mutex> l(H.m);
            if (rc != REDUX_OK)
                H.abort = true;
            else
                H.issued = j + 1;
            H.cv.notify_all();
            if (rc != REDUX_OK)
                break;
        }
        {
            std::lock_guard<std::mutex> l(H.m); // (left early because another context failed: release the drain thread)
            if (H.issued < mine)
                H.abort = true;
            H.cv.notify_all();
        }
    }
    drain.join();
    for (int i = 0; i < kStreams; i++) // nothing of this call stays in flight
        (void)hipStreamSynchronize(c.stream[i]);
}

static int encode_blocks(const redux_params *p, const uint8_t *in, uint64_t in_len, uint32_t block_size, uint8_t *out,
                         uint64_t out_cap, uint64_t *out_offsets, int32_t *block_status)
{
    std::vector<Ctx *> ctx;
    std::vector<std::unique_lock<std::mutex>> locks;
    int rc = take_contexts(ctx, locks);
    if (rc != REDUX_OK)
        return rc;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);

    EncCall E;
    E.p = p; E.in = in; E.in_len = in_len; E.block_size = block_size; E.out = out; E.out_cap = out_cap;
    E.out_offsets = out_offsets; E.block_status = block_status;
    E.nblocks  = redux_block_count(in_len, block_size);
    E.cb       = chunk_blocks_for(E.nblocks, block_size, kEncChunkMax, ctx.size());
    E.nchunks  = (E.nblocks + E.cb - 1) / E.cb;
    E.chunk_in = E.cb * (uint64_t)block_size; // bytes of a full chunk
    E.ws_bytes = redux_encode_workspace_bytes(p, E.chunk_in < in_len ? E.chunk_in : in_len, block_size);
    if (E.nchunks > 1) // several chunks in flight keep the chip busy: no pairs area, so the chunks run on the pair kernel (encode_slots_impl)
        E.ws_bytes = geometry(p, E.chunk_in, block_size, false, false).total;
    E.bound    = redux_encode_bound(p, E.chunk_in < in_len ? E.chunk_in : in_len, block_size);
    Job J;
    J.total.assign(E.nchunks, 0);
    J.known.assign(E.nchunks, 0);
    J.prefix.assign(E.nchunks + 1, 0);
    const uint64_t nctx = ctx.size() < E.nchunks ? ctx.size() : E.nchunks; // (a call of one chunk uses one context)
    std::vector<std::thread> th;
    for (uint64_t d = 1; d < nctx; d++)
        th.emplace_back([&, d] { encode_on_ctx(*ctx[d], E, J, d, nctx); });
    encode_on_ctx(*ctx[0], E, J, 0, nctx);
    for (auto &t : th)
        t.join();
    for (Ctx *c : ctx) {
        if (c->ready)
            (void)hipSetDevice(c->device);
        ctx_trim_locked(*c);
    }
    if (caller_dev >= 0)
        (void)hipSetDevice(caller_dev);
    if (J.error != REDUX_OK)
        return J.error;
    return J.bad_status;
}

// ================================================================================================
// decode
// ================================================================================================
// decode_blocks_dev_impl of redux_hip.hip (defined after this header is included)
typedef int (*DecodeDevCall)(const redux_params *, const void *, const void *, uint64_t, uint32_t, void *, uint64_t, void *, void *,
                             void *, void *, uint64_t, void *, void *, const redux_block *, bool, uint64_t);

struct DecCall {
    const redux_params *p;
    const uint8_t      *in;
    const uint64_t     *in_offsets;
    uint64_t            nblocks;
    uint32_t            block_size;
    uint8_t            *out;
    uint32_t           *out_sizes;
    int32_t            *block_status;
    uint64_t           *in_used;
    DecodeDevCall       dev_call;
    uint64_t            cb, nchunks, wsb, max_in;
};

static void decode_on_ctx(Ctx &c, const DecCall &D, Job &J, uint64_t first, uint64_t stride)
{
    int rc = ctx_init_locked(c); // (makes c.want HIP's current device on this thread)
    if (rc != REDUX_OK)
        return J.fail(rc);
    const uint64_t mine   = first < D.nchunks ? (D.nchunks - first + stride - 1) / stride : 0;
    const int      nslots = (int)(mine < (uint64_t)kSlots ? mine : (uint64_t)kSlots);
    for (int i = 0; i < nslots; i++) {
        Slot &s = c.slot[i];
        if ((rc = grow_dev(c, s.d_in, D.max_in + 32)) || (rc = grow_dev(c, s.d_ws, D.wsb + 256)) ||
            (rc = grow_dev(c, s.d_out, D.cb * (uint64_t)D.block_size + 16)) || (rc = grow_dev(c, s.d_off, (D.cb + 1) * 8)) ||
            (rc = grow_dev(c, s.d_sz, D.cb * 4)) || (rc = grow_dev(c, s.d_st, D.cb * 4)) || (rc = grow_dev(c, s.d_sum, 8)) ||
            (rc = grow_dev(c, s.d_used, D.in_used ? D.cb * 8 : 8)) || (rc = grow_pinned(c, s.h_off, (D.cb + 1) * 8)) ||
            (rc = grow_pinned(c, s.h_sz, D.cb * 4)) || (rc = grow_pinned(c, s.h_st, D.cb * 4)) || (rc = grow_pinned(c, s.h_sum, 8)) ||
            (rc = grow_pinned(c, s.h_used, D.in_used ? D.cb * 8 : 8)))
            return J.fail(rc);
    }
    c.trace.assign(mine * 4, 0.0);
    c.t0 = now_s();
    Handover H;
    std::thread drain([&] {
        (void)hipSetDevice(c.device);
        for (uint64_t j = 0; j < mine; j++) {
            const uint64_t k = first + j * stride;
            {
                std::unique_lock<std::mutex> l(H.m);
                H.cv.wait(l, [&] { return H.issued > j || H.abort; });
                if (H.abort)
                    return;
            }
            Slot          &s  = c.slot[j % kSlots];
            const uint64_t b0 = k * D.cb, nb = (b0 + D.cb <= D.nblocks ? D.cb : D.nblocks - b0);
            int            err = REDUX_OK;
            hipStream_t st = c.stream[j % kStreams];
            if (hipEventSynchronize(s.done) != hipSuccess ||
                hipMemcpyAsync(s.h_sz.p, s.d_sz.p, nb * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(s.h_st.p, s.d_st.p, nb * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(s.h_sum.p, s.d_sum.p, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                (D.in_used && hipMemcpyAsync(s.h_used.p, s.d_used.p, nb * 8, hipMemcpyDeviceToHost, st) != hipSuccess) ||
                hipStreamSynchronize(st) != hipSuccess)
                err = REDUX_IO_ERROR;
            c.trace[j * 4 + 2] = now_s() - c.t0;
            if (!err) {
                if (((const int32_t *)s.h_sum.p)[0] != REDUX_OK)
                    J.note_bad(k, ((const int32_t *)s.h_sum.p)[0]);
                err = drain_d2h(c, D.out + b0 * (uint64_t)D.block_size, s.d_out.p, nb * (uint64_t)D.block_size);
            }
            if (!err) {
                memcpy(D.out_sizes + b0, s.h_sz.p, nb * 4);
                if (D.block_status)
                    memcpy(D.block_status + b0, s.h_st.p, nb * 4);
                if (D.in_used)
                    memcpy(D.in_used + b0, s.h_used.p, nb * 8);
            }
            c.trace[j * 4 + 3] = now_s() - c.t0;
            if (err)
                J.fail(err);
            std::lock_guard<std::mutex> l(H.m);
            H.drained = j + 1;
            if (err)
                H.abort = true;
            H.cv.notify_all();
            if (err)
                return;
        }
    });

    {
        CopyPool pool(kCopyThreads - 1);
        uint64_t piece_no = 0;
        for (uint64_t j = 0; j < mine; j++) {
            const uint64_t k = first + j * stride;
            {
                std::unique_lock<std::mutex> l(H.m);
                H.cv.wait(l, [&] { return H.abort || j < (uint64_t)kSlots || H.drained + kSlots > j; });
                if (H.abort)
                    break;
            }
            {
                std::lock_guard<std::mutex> l(J.m);
                if (J.abort)
                    break;
            }
            c.trace[j * 4 + 0] = now_s() - c.t0;
            Slot          &s  = c.slot[j % kSlots];
            hipStream_t    st = c.stream[j % kStreams];
            const uint64_t b0 = k * D.cb, nb = (b0 + D.cb <= D.nblocks ? D.cb : D.nblocks - b0);
            const uint64_t i0 = D.in_offsets[b0], len = D.in_offsets[b0 + nb] - i0;
            auto issue = [&]() -> int {
                // the chunk's offsets, rebased to the chunk's first byte (the pinned mirror of the previous
                // chunk in this slot has been consumed: that chunk is drained)
                uint64_t *ho = (uint64_t *)s.h_off.p;
                for (uint64_t i = 0; i <= nb; i++) {
                    if (D.in_offsets[b0 + i] < i0 || (i && D.in_offsets[b0 + i] < D.in_offsets[b0 + i - 1]))
                        return REDUX_INVALID_INPUT;
                    ho[i] = D.in_offsets[b0 + i] - i0;
                }
                HOST_TRY(hipMemcpyAsync(s.d_off.p, ho, (nb + 1) * 8, hipMemcpyHostToDevice, st));
                int r = stage_h2d(c, pool, piece_no, s.d_in.p, D.in + i0, len, st);
                if (r != REDUX_OK)
                    return r;
                HOST_TRY(hipMemsetAsync(s.d_sum.p, 0, 8, st));
                uint8_t *ws = (uint8_t *)(((uintptr_t)s.d_ws.p + 255) & ~(uintptr_t)255);
                r = D.dev_call(D.p, s.d_in.p, s.d_off.p, nb, D.block_size, s.d_out.p, nb * (uint64_t)D.block_size, s.d_sz.p, s.d_st.p,
                               s.d_sum.p, ws, D.wsb, st, D.in_used ? s.d_used.p : nullptr, nullptr, false, 0);
                if (r != REDUX_OK)
                    return r;
                HOST_TRY(hipEventRecord(s.done, st)); // (small result arrays: fetched by the drain thread, see encode_on_ctx)
                return REDUX_OK;
            };
            rc = issue();
            c.trace[j * 4 + 1] = now_s() - c.t0;
            if (rc != REDUX_OK)
                J.fail(rc);
            std::lock_guard<std::mutex> l(H.m);
            if (rc != REDUX_OK)
                H.abort = true;
            else
                H.issued = j + 1;
            H.cv.notify_all();
            if (rc != REDUX_OK)
                break;
        }
        {
            std::lock_guard<std::mutex> l(H.m);
            if (H.issued < mine)
                H.abort = true;
            H.cv.notify_all();
        }
    }
    drain.join();
    for (int i = 0; i < kStreams; i++)
        (void)hipStreamSynchronize(c.stream[i]);
}

static int decode_blocks(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint64_t nblocks,
                         uint32_t block_size, uint8_t *out, uint64_t out_cap, uint32_t *out_sizes, int32_t *block_status,
                         uint64_t *in_used, DecodeDevCall dev_call)
{
    (void)out_cap;
    std::vector<Ctx *> ctx;
    std::vector<std::unique_lock<std::mutex>> locks;
    int rc = take_contexts(ctx, locks);
    if (rc != REDUX_OK)
        return rc;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);

    DecCall D;
    D.p = p; D.in = in; D.in_offsets = in_offsets; D.nblocks = nblocks; D.block_size = block_size; D.out = out;
    D.out_sizes = out_sizes; D.block_status = block_status; D.in_used = in_used; D.dev_call = dev_call;
    D.cb      = chunk_blocks_for(nblocks, block_size, kDecChunkMax, ctx.size());
    D.nchunks = (nblocks + D.cb - 1) / D.cb;
    D.wsb     = redux_decode_workspace_bytes(p, D.cb, block_size);
    D.max_in  = 0;
    for (uint64_t k = 0; k < D.nchunks; k++) {
        const uint64_t b0 = k * D.cb, b1 = (b0 + D.cb <= nblocks ? b0 + D.cb : nblocks);
        if (in_offsets[b1] < in_offsets[b0])
            return REDUX_INVALID_INPUT;
        const uint64_t n = in_offsets[b1] - in_offsets[b0];
        D.max_in = n > D.max_in ? n : D.max_in;
    }
    Job J;
    const uint64_t nctx = ctx.size() < D.nchunks ? ctx.size() : D.nchunks;
    std::vector<std::thread> th;
    for (uint64_t d = 1; d < nctx; d++)
        th.emplace_back([&, d] { decode_on_ctx(*ctx[d], D, J, d, nctx); });
    decode_on_ctx(*ctx[0], D, J, 0, nctx);
    for (auto &t : th)
        t.join();
    for (Ctx *c : ctx) {
        if (c->ready)
            (void)hipSetDevice(c->device);
        ctx_trim_locked(*c);
    }
    if (caller_dev >= 0)
        (void)hipSetDevice(caller_dev);
    if (J.error != REDUX_OK)
        return J.error;
    return J.bad_status;
}

// ================================================================================================
// many independent inputs in one call (redux_encode_blocks_v / redux_decode_blocks_v)
//
// Consecutive inputs are packed into GROUPS of at most kVGroupBytes; a group is staged into HBM with every input at a
// 16-byte boundary (so that the fast kernels apply), coded by ONE launch over its block table and copied back.  The point
// of these calls is many small inputs -- the reference's corpus harness, 36 files of 4 KB - 4 MB, is one group -- where
// what counts is that all blocks share a launch.  A batch of several groups is dealt over the fleet (redux_host_set_devices):
// group k runs on context k mod n, each context on its first slot and its own thread; the encoder's dense output keeps
// block order through a ledger of group sizes.
// ================================================================================================
constexpr uint64_t kVGroupBytes = 512ull << 20;

struct DeviceScope { // makes `dev` HIP's current device for a scope and puts the caller's back
    int prev = -1;
    explicit DeviceScope(int dev)
    {
        (void)hipGetDevice(&prev);
        if (dev != prev)
            (void)hipSetDevice(dev);
    }
    ~DeviceScope()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// One group of consecutive inputs = one launch.  first_block = number of its first block; nb its blocks.
struct VGroup {
    uint64_t i0, i1, first_block, nb;
};

// inputs [0, ninputs) -> groups of at most kVGroupBytes of (16-byte padded) payload each
static std::vector<VGroup> v_groups(const uint64_t *len, uint64_t ninputs, uint32_t block_size)
{
    std::vector<VGroup> g;
    uint64_t            blk = 0;
    const uint64_t      cap = g_chunk_max.load() ? g_chunk_max.load() : kVGroupBytes; // (redux_host_set_chunk_bytes: a harness drives many groups through a small batch)
    for (uint64_t i0 = 0; i0 < ninputs;) {
        uint64_t i1 = i0, pos = 0;
        do {
            pos += (len[i1] + 15) & ~15ull;
            i1++;
        } while (i1 < ninputs && pos + len[i1] <= cap);
        const uint64_t nb = redux_block_count_v(len + i0, i1 - i0, block_size);
        g.push_back({i0, i1, blk, nb});
        blk += nb;
        i0 = i1;
    }
    return g;
}

// What the contexts of one `_v` call share: the groups are dealt round-robin (group k on context k mod n, each context
// taking its groups in order), and -- encode only -- a group's place in the dense output is known once every earlier
// group's size is.
struct VJob {
    std::mutex              m;
    std::condition_variable cv;
    std::vector<uint64_t>   total;
    std::vector<char>       known;
    std::vector<int>        bad;   // first non-OK block status of each group
    int                     error = REDUX_OK;
    bool                    abort = false;

    explicit VJob(size_t n) : total(n, 0), known(n, 0), bad(n, REDUX_OK) {}
    void fail(int rc)
    {
        std::lock_guard<std::mutex> l(m);
        if (error == REDUX_OK)
            error = rc;
        abort = true;
        cv.notify_all();
    }
    void publish(size_t g, uint64_t t)
    {
        std::lock_guard<std::mutex> l(m);
        total[g] = t;
        known[g] = 1;
        cv.notify_all();
    }
    // bytes of all groups before g, once they are all known; false if the call was aborted meanwhile
    bool base_of(size_t g, uint64_t &base)
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] {
            if (abort)
                return true;
            for (size_t h = 0; h < g; h++)
                if (!known[h])
                    return false;
            return true;
        });
        if (abort)
            return false;
        base = 0;
        for (size_t h = 0; h < g; h++)
            base += total[h];
        return true;
    }
};

struct EncVCall {
    const redux_params *p;
    const uint8_t      *in;
    const uint64_t     *in_off, *in_len;
    uint32_t            block_size;
    uint8_t            *out;
    uint64_t            out_cap;
    uint64_t           *out_offsets;
    int32_t            *block_status;
};

static int encode_v_group(Ctx &c, const EncVCall &E, const VGroup &G, size_t gi, VJob &J, CopyPool &pool, uint64_t &piece_no)
{
    int         rc;
    Slot       &s  = c.slot[0];
    hipStream_t st = c.stream[0];
    std::vector<uint64_t> doff;
    uint64_t              pos = 0;
    for (uint64_t i = G.i0; i < G.i1; i++) {
        doff.push_back(pos);
        pos += (E.in_len[i] + 15) & ~15ull;
    }
    if (pos > 0xFFFFFFFFull) // (one input of 4 GiB or more: lane offsets are 32-bit; redux_encode_blocks takes it)
        return REDUX_UNSUPPORTED;
    const uint64_t nb = G.nb;
    const uint64_t ne = redux_block_table_v(doff.data(), E.in_len + G.i0, G.i1 - G.i0, E.block_size, nullptr); // entries: blocks + idle lanes
    std::vector<redux_block> tbl(ne);
    redux_block_table_v(doff.data(), E.in_len + G.i0, G.i1 - G.i0, E.block_size, tbl.data());
    const uint64_t ws_bytes = redux_encode_workspace_bytes(E.p, ne * (uint64_t)E.block_size, E.block_size);
    const uint64_t bound    = nb * redux_encode_slot_bytes(E.p, E.block_size);
    if ((rc = grow_dev(c, s.d_in, pos + 16)) || (rc = grow_dev(c, s.d_ws, ws_bytes + 256)) || (rc = grow_dev(c, s.d_out, bound + 16)) ||
        (rc = grow_dev(c, s.d_off, (nb + 1) * 8)) || (rc = grow_dev(c, s.d_st, nb * 4)) || (rc = grow_dev(c, s.d_sum, 8)) ||
        (rc = grow_dev(c, s.d_tab, ne * sizeof(redux_block))) || (rc = grow_pinned(c, s.h_off, (nb + 1) * 8)) ||
        (rc = grow_pinned(c, s.h_st, nb * 4)) || (rc = grow_pinned(c, s.h_sum, 8)) || (rc = grow_pinned(c, s.h_tab, ne * sizeof(redux_block))))
        return rc;
    memcpy(s.h_tab.p, tbl.data(), ne * sizeof(redux_block));
    HOST_TRY(hipMemcpyAsync(s.d_tab.p, s.h_tab.p, ne * sizeof(redux_block), hipMemcpyHostToDevice, st));
    for (uint64_t k = 0; k < G.i1 - G.i0; k++)
        if (E.in_len[G.i0 + k] &&
            (rc = stage_h2d(c, pool, piece_no, (uint8_t *)s.d_in.p + doff[k], E.in + E.in_off[G.i0 + k], E.in_len[G.i0 + k], st)))
            return rc;
    HOST_TRY(hipMemsetAsync(s.d_sum.p, 0, 8, st));
    uint8_t *ws = (uint8_t *)(((uintptr_t)s.d_ws.p + 255) & ~(uintptr_t)255);
    if ((rc = redux_encode_blocks_v_dev(E.p, s.d_in.p, pos, s.d_tab.p, ne, nb, E.block_size, REDUX_V_ALIGNED16, s.d_out.p, bound, s.d_off.p,
                                        s.d_st.p, s.d_sum.p, ws, ws_bytes, st)))
        return rc;
    HOST_TRY(hipMemcpyAsync(s.h_off.p, s.d_off.p, (nb + 1) * 8, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipMemcpyAsync(s.h_st.p, s.d_st.p, nb * 4, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipMemcpyAsync(s.h_sum.p, s.d_sum.p, 8, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipStreamSynchronize(st));
    const uint64_t *ho    = (const uint64_t *)s.h_off.p;
    const uint64_t  total = ho[nb];
    J.publish(gi, total);
    uint64_t out_base = 0;
    if (!J.base_of(gi, out_base)) // (another group failed)
        return REDUX_OK;
    if (out_base + total > E.out_cap)
        return REDUX_OUTPUT_TOO_SMALL;
    if (total && (rc = drain_d2h(c, E.out + out_base, s.d_out.p, total)))
        return rc;
    for (uint64_t i = 0; i <= nb; i++) // (entry nb is also the next group's entry 0: the same value from either side)
        E.out_offsets[G.first_block + i] = out_base + ho[i];
    if (E.block_status)
        memcpy(E.block_status + G.first_block, s.h_st.p, nb * 4);
    J.bad[gi] = ((const int32_t *)s.h_sum.p)[0];
    return REDUX_OK;
}

// runs fn(group index) for the groups first, first + stride, ... on context c (whose mutex the call holds)
template <typename F>
static void v_on_ctx(Ctx &c, size_t ngroups, size_t first, size_t stride, VJob &J, F &&fn)
{
    int rc = ctx_init_locked(c); // (makes c.want HIP's current device on this thread)
    if (rc != REDUX_OK)
        return J.fail(rc);
    CopyPool pool(kCopyThreads - 1);
    uint64_t piece_no = 0;
    for (size_t g = first; g < ngroups; g += stride) {
        {
            std::lock_guard<std::mutex> l(J.m);
            if (J.abort)
                break;
        }
        if ((rc = fn(c, g, pool, piece_no)) != REDUX_OK)
            return J.fail(rc);
    }
    (void)hipStreamSynchronize(c.stream[0]);
}

template <typename F>
static int v_deal(size_t ngroups, VJob &J, F &&fn)
{
    std::vector<Ctx *> ctx;
    std::vector<std::unique_lock<std::mutex>> locks;
    int rc = take_contexts(ctx, locks);
    if (rc != REDUX_OK)
        return rc;
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);
    const size_t nctx = ctx.size() < ngroups ? ctx.size() : ngroups; // (the reference's corpus is one group: one context)
    std::vector<std::thread> th;
    for (size_t d = 1; d < nctx; d++)
        th.emplace_back([&, d] { v_on_ctx(*ctx[d], ngroups, d, nctx, J, fn); });
    v_on_ctx(*ctx[0], ngroups, 0, nctx, J, fn);
    for (auto &t : th)
        t.join();
    for (Ctx *c : ctx) {
        if (c->ready)
            (void)hipSetDevice(c->device);
        ctx_trim_locked(*c);
    }
    if (caller_dev >= 0)
        (void)hipSetDevice(caller_dev);
    if (J.error != REDUX_OK)
        return J.error;
    for (int b : J.bad) // the first group (in block order) with a non-OK block decides
        if (b != REDUX_OK)
            return b;
    return REDUX_OK;
}

static int encode_blocks_v(const redux_params *p, const uint8_t *in, const uint64_t *in_off, const uint64_t *in_len, uint64_t ninputs,
                           uint32_t block_size, uint8_t *out, uint64_t out_cap, uint64_t *out_offsets, int32_t *block_status)
{
    const std::vector<VGroup> groups = v_groups(in_len, ninputs, block_size);
    EncVCall E{p, in, in_off, in_len, block_size, out, out_cap, out_offsets, block_status};
    VJob     J(groups.size());
    return v_deal(groups.size(), J, [&](Ctx &c, size_t g, CopyPool &pool, uint64_t &piece_no) {
        return encode_v_group(c, E, groups[g], g, J, pool, piece_no);
    });
}

struct DecVCall {
    const redux_params *p;
    const uint8_t      *in;
    const uint64_t     *in_offsets;
    uint8_t            *out;
    const uint64_t     *out_off, *out_len;
    uint32_t            block_size;
    uint32_t           *out_sizes;
    int32_t            *block_status;
    DecodeDevCall       dev_call;
};

static int decode_v_group(Ctx &c, const DecVCall &D, const VGroup &G, size_t gi, VJob &J, CopyPool &pool, uint64_t &piece_no)
{
    int         rc;
    Slot       &s  = c.slot[0];
    hipStream_t st = c.stream[0];
    std::vector<uint64_t> doff;
    uint64_t              pos = 0;
    for (uint64_t i = G.i0; i < G.i1; i++) {
        doff.push_back(pos);
        pos += (D.out_len[i] + 15) & ~15ull;
    }
    const uint64_t nb = G.nb, blk_base = G.first_block;
    const uint64_t ne = redux_block_table_v(doff.data(), D.out_len + G.i0, G.i1 - G.i0, D.block_size, nullptr);
    std::vector<redux_block> tbl(ne);
    redux_block_table_v(doff.data(), D.out_len + G.i0, G.i1 - G.i0, D.block_size, tbl.data()); // offset = where the block goes, length = its room
    const uint64_t sb0 = D.in_offsets[blk_base];
    for (uint64_t i = 0; i < nb; i++)
        if (D.in_offsets[blk_base + i + 1] < D.in_offsets[blk_base + i])
            return REDUX_INVALID_INPUT;
    const uint64_t len_in = D.in_offsets[blk_base + nb] - sb0;
    if (len_in && !D.in)
        return REDUX_INVALID_INPUT;
    const uint64_t wsb = redux_decode_workspace_bytes(D.p, ne, D.block_size);
    if ((rc = grow_dev(c, s.d_in, len_in + 32)) || (rc = grow_dev(c, s.d_ws, wsb + 256)) || (rc = grow_dev(c, s.d_out, pos + 16)) ||
        (rc = grow_dev(c, s.d_off, (nb + 1) * 8)) || (rc = grow_dev(c, s.d_sz, nb * 4)) || (rc = grow_dev(c, s.d_st, nb * 4)) ||
        (rc = grow_dev(c, s.d_sum, 8)) || (rc = grow_dev(c, s.d_tab, ne * sizeof(redux_block))) ||
        (rc = grow_pinned(c, s.h_off, (nb + 1) * 8)) || (rc = grow_pinned(c, s.h_sz, nb * 4)) || (rc = grow_pinned(c, s.h_st, nb * 4)) ||
        (rc = grow_pinned(c, s.h_sum, 8)) || (rc = grow_pinned(c, s.h_tab, ne * sizeof(redux_block))))
        return rc;
    uint64_t *ho = (uint64_t *)s.h_off.p;
    for (uint64_t i = 0; i <= nb; i++)
        ho[i] = D.in_offsets[blk_base + i] - sb0;
    memcpy(s.h_tab.p, tbl.data(), ne * sizeof(redux_block));
    HOST_TRY(hipMemcpyAsync(s.d_off.p, ho, (nb + 1) * 8, hipMemcpyHostToDevice, st));
    HOST_TRY(hipMemcpyAsync(s.d_tab.p, s.h_tab.p, ne * sizeof(redux_block), hipMemcpyHostToDevice, st));
    if (len_in && (rc = stage_h2d(c, pool, piece_no, s.d_in.p, D.in + sb0, len_in, st)))
        return rc;
    HOST_TRY(hipMemsetAsync(s.d_sum.p, 0, 8, st));
    uint8_t *ws = (uint8_t *)(((uintptr_t)s.d_ws.p + 255) & ~(uintptr_t)255);
    if ((rc = D.dev_call(D.p, s.d_in.p, s.d_off.p, ne, D.block_size, s.d_out.p, pos, s.d_sz.p, s.d_st.p, s.d_sum.p, ws, wsb, st, nullptr,
                         (const redux_block *)s.d_tab.p, true, nb)))
        return rc;
    HOST_TRY(hipMemcpyAsync(s.h_sz.p, s.d_sz.p, nb * 4, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipMemcpyAsync(s.h_st.p, s.d_st.p, nb * 4, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipMemcpyAsync(s.h_sum.p, s.d_sum.p, 8, hipMemcpyDeviceToHost, st));
    HOST_TRY(hipStreamSynchronize(st));
    const uint32_t *hs = (const uint32_t *)s.h_sz.p;
    // an input's blocks are back to back in both buffers: whole runs of full blocks leave as one copy
    uint64_t b = 0;
    for (uint64_t k = 0; k < G.i1 - G.i0; k++) {
        const uint64_t cnt = redux_block_count(D.out_len[G.i0 + k], D.block_size);
        for (uint64_t j = 0; j < cnt;) {
            uint64_t run = 0, j1 = j;
            while (j1 < cnt) { // extend over blocks that decoded to a whole block_size; the first shorter one ends the run
                const uint32_t sz = hs[b + j1];
                run += sz;
                j1++;
                if (sz != D.block_size)
                    break;
            }
            if (run && (rc = drain_d2h(c, D.out + D.out_off[G.i0 + k] + j * (uint64_t)D.block_size,
                                       (const uint8_t *)s.d_out.p + doff[k] + j * (uint64_t)D.block_size, run)))
                return rc;
            j = j1;
        }
        b += cnt;
    }
    memcpy(D.out_sizes + blk_base, hs, nb * 4);
    if (D.block_status)
        memcpy(D.block_status + blk_base, s.h_st.p, nb * 4);
    J.bad[gi] = ((const int32_t *)s.h_sum.p)[0];
    return REDUX_OK;
}

static int decode_blocks_v(const redux_params *p, const uint8_t *in, const uint64_t *in_offsets, uint8_t *out, const uint64_t *out_off,
                           const uint64_t *out_len, uint64_t ninputs, uint32_t block_size, uint32_t *out_sizes, int32_t *block_status,
                           DecodeDevCall dev_call)
{
    const std::vector<VGroup> groups = v_groups(out_len, ninputs, block_size);
    DecVCall D{p, in, in_offsets, out, out_off, out_len, block_size, out_sizes, block_status, dev_call};
    VJob     J(groups.size());
    return v_deal(groups.size(), J, [&](Ctx &c, size_t g, CopyPool &pool, uint64_t &piece_no) {
        return decode_v_group(c, D, groups[g], g, J, pool, piece_no);
    });
}

} // namespace host
} // namespace redux
