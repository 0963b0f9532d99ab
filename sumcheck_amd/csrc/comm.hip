// comm.hip -- the communicators of a sharded proof: RCCL (bound at run time), a host transport (two caller functions),
// and peer-to-peer between the thread ranks of one process (fine-grained inboxes, k_p2p_allreduce).
#include "prover_internal.hpp"


// ---------------------------------------------------------------------------------------------------
// Sharded rounds inside the library: one RCCL all-reduce per round on the handle's stream.
// RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that resolves to the copy torch already
// mapped), so the library itself has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------------
NcclApi g_nccl;
int nccl_load() {
    if (g_nccl.lib) return SC_OK;
    // A copy the process has already mapped (PyTorch-ROCm ships its own librccl) is the one to use: two RCCLs in one process is one too
    // many.  Nothing is promoted to the global namespace (RTLD_LOCAL): the entry points are taken with dlsym from this handle, and the
    // host process's own symbol resolution is left alone.
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return sc_internal_fail(SC_ERR_HIP, "cannot load librccl: %s", dlerror());
    g_nccl.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_nccl.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_nccl.AllReduce = reinterpret_cast<decltype(&ncclAllReduce)>(dlsym(h, "ncclAllReduce"));
    g_nccl.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(dlsym(h, "ncclAllGather"));
    g_nccl.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_nccl.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.AllReduce || !g_nccl.AllGather || !g_nccl.CommDestroy)
        return sc_internal_fail(SC_ERR_HIP, "librccl lacks the NCCL entry points");
    g_nccl.lib = h;
    return SC_OK;
}
static std::mutex g_p2p_mu;
static std::map<uint64_t, std::shared_ptr<P2PGroup>> g_p2p_groups;


extern "C" int sc_comm_unique_id(uint8_t *out128) {
    if (!out128) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    int rc = nccl_load();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    NCCL_TRY(g_nccl.GetUniqueId(reinterpret_cast<ncclUniqueId *>(out128)));
    return SC_OK;
}
// Direct publication, probed once per RCCL communicator (collective: every rank of the communicator runs it inside sc_comm_init).
// A sharded round's exchange is ncclAllReduce followed by a kernel that copies the totals to a host-mapped page and raises a flag --
// a second launch of latency per round.  Instead the all-reduce's receive buffer IS the host-mapped page, and every rank ORs a tag
// into the top bits of its lanes (kernels.h: wide_tag_of): a word that reads nranks * tag is this round's total, the host's poll is
// the fetch.  Whether RCCL's kernels deliver into host-mapped memory on this system, and whether the host sees the words without a
// stream synchronisation, is tested here with two tagged all-reduces of known lanes.
// THE DECISION IS COLLECTIVE: a rank that tags its lanes and waits for nranks * tag cannot work with a peer that does neither, so after
// the two probes the ranks take the minimum of their verdicts (a third all-reduce, ncclMin) -- one rank's failed probe, failed
// allocation or sc_set_policy("rccl_direct", 0) leaves EVERY rank on the publish kernel.  Every rank issues all three collectives
// whatever it observed or failed to allocate (the peers are inside them): without the host-mapped page the probes run in place in
// device memory and the rank votes no.
static bool rccl_direct_probe(sc_comm *c) {
    constexpr int kWords = 40;
    hipStream_t s = nullptr;
    uint64_t *d = nullptr, *h = nullptr, *h_dev = nullptr;
    // what the collectives themselves need: a stream and 41 device words (failing these, this rank cannot take part in any collective at all)
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d), (kWords + 1) * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (s) (void)hipStreamDestroy(s);
        return false;
    }
    bool ok = scd::policy(scd::kPolRcclDirect) != 0 &&
              hipHostMalloc(reinterpret_cast<void **>(&h), kWords * 8, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void **>(&h_dev), h, 0) == hipSuccess;
    if (!ok) h_dev = nullptr;
    (void)hipGetLastError();
    if (h) std::memset(h, 0, kWords * 8);
    const uint64_t tri = (uint64_t)c->nranks * (uint64_t)(c->nranks + 1) / 2;
    uint64_t mine[kWords];
    for (uint32_t it = 0; it < 2; ++it) {
        const uint32_t tag = scd::wide_tag_of(0x7ffeu + it); // (the second one wraps the tag)
        for (int w = 0; w < kWords; ++w) mine[w] = ((uint64_t)(c->rank + 1) * (uint64_t)(w + 1 + it)) | ((uint64_t)tag << scd::kWideTagShift);
        ok = (hipMemcpyAsync(d, mine, sizeof(mine), hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) && ok;
        const bool issued = (int)g_nccl.AllReduce(d, h_dev ? h_dev : d, (size_t)kWords, ncclUint64, ncclSum, c->comm, s) == 0; // (always issued)
        ok = ok && issued && h_dev;
        if (ok) { // the words must show up by themselves: the rounds of a proof never synchronise the stream
            const auto t0 = std::chrono::steady_clock::now();
            const uint64_t expect = (uint64_t)c->nranks * tag;
            bool seen = false;
            while (!seen && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) {
                seen = true;
                for (int w = kWords - 1; w >= 0 && seen; --w) seen = (__atomic_load_n(h + w, __ATOMIC_ACQUIRE) >> scd::kWideTagShift) == expect;
            }
            ok = seen;
            for (int w = 0; w < kWords && ok; ++w) ok = (h[w] & ((1ULL << scd::kWideTagShift) - 1)) == tri * (uint64_t)(w + 1 + it);
        }
        (void)hipStreamSynchronize(s);
    }
    // the agreement: min over the ranks of "my probe passed"
    uint64_t vote = ok ? 1 : 0, agreed = 0;
    bool v_ok = hipMemcpyAsync(d + kWords, &vote, 8, hipMemcpyHostToDevice, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    v_ok = ((int)g_nccl.AllReduce(d + kWords, d + kWords, 1, ncclUint64, ncclMin, c->comm, s) == 0) && v_ok; // (always issued)
    v_ok = v_ok && hipMemcpyAsync(&agreed, d + kWords, 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    (void)hipStreamSynchronize(s);
    (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    (void)hipStreamDestroy(s);
    (void)hipGetLastError();
    // (a rank whose vote could not be cast or read has no way of knowing what the others agreed on: it reports no -- and so, by the
    // minimum, do all the others unless the failure was in reading the result back, which a broken device does not survive anyway)
    return v_ok && agreed == 1;
}

extern "C" int sc_comm_init(const uint8_t *id128, int rank, int nranks, sc_comm **out) {
    if (!id128 || !out || rank < 0 || rank >= nranks) return sc_internal_fail(SC_ERR_BAD_ARG, "bad argument");
    int rc = nccl_load();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(sc_internal_device_ref()));
    sc_comm *c = new (std::nothrow) sc_comm();
    if (!c) return sc_internal_fail(SC_ERR_OOM, "host allocation failed");
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    int r = (int)g_nccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
        delete c;
        return sc_internal_fail(SC_ERR_HIP, "ncclCommInitRank failed: %s", g_nccl.GetErrorString ? g_nccl.GetErrorString((ncclResult_t)r) : "?");
    }
    c->rank = rank;
    c->nranks = nranks;
    c->direct_publish = nranks <= 16 && rccl_direct_probe(c);
    *out = c;
    return SC_OK;
}
extern "C" int sc_comm_init_host(int rank, int nranks, sc_allreduce_u64_fn allreduce, sc_allgather_fn allgather, void *ctx, sc_comm **out) {
    if (!out || rank < 0 || rank >= nranks || (nranks > 1 && (!allreduce || !allgather))) return sc_internal_fail(SC_ERR_BAD_ARG, "bad argument");
    sc_comm *c = new (std::nothrow) sc_comm();
    if (!c) return sc_internal_fail(SC_ERR_OOM, "host allocation failed");
    c->rank = rank;
    c->nranks = nranks;
    c->h_allreduce = allreduce;
    c->h_allgather = allgather;
    c->ctx = ctx;
    *out = c;
    return SC_OK;
}
// Peer-to-peer communicator for thread ranks (one host thread and one GPU per rank inside ONE process): no collective library.  Every
// rank allocates a fine-grained inbox on its own device, the ranks meet in the process-wide registry under `group_id` (any number
// not in use by another live group; the call blocks until all `nranks` threads have made it, 60 s at most), peer access is enabled
// between the devices, and from then on a round's all-reduce is one small kernel per rank (kernels.h: P2PArgs) and the tail's
// gather is peer copies.  Ranks may share a GPU (functional tests on a one-GPU box): the exchange kernel then gives up quickly when a
// peer's kernel has not run yet -- it may be queued behind this one -- and the host launches it again.
extern "C" int sc_comm_init_p2p(uint64_t group_id, int rank, int nranks, sc_comm **out) {
    if (!out || rank < 0 || rank >= nranks || nranks > scd::kP2PMaxRanks) return sc_internal_fail(SC_ERR_BAD_ARG, "bad argument");
    *out = nullptr;
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    HIP_TRY(hipSetDevice(sc_internal_device_ref()));
    std::shared_ptr<P2PGroup> g;
    {
        std::lock_guard<std::mutex> lk(g_p2p_mu);
        auto &slot = g_p2p_groups[group_id];
        if (!slot) {
            slot = std::make_shared<P2PGroup>();
            slot->nranks = nranks;
        }
        g = slot;
    }
    uint64_t *inbox = nullptr;
    {
        DeviceGate gate_(sc_internal_device_ref());
        // fine-grained: peers' stores become visible to this device's polls without a kernel boundary
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&inbox), scd::kP2PInboxWords * 8, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&inbox), scd::kP2PInboxWords * 8));
        }
        HIP_TRY(hipMemset(inbox, 0, scd::kP2PInboxWords * 8));
        HIP_TRY(hipDeviceSynchronize());
    }
    auto give_up = [&](int code, const char *msg) {
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->broken = true;
            g->cv.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(g_p2p_mu);
            auto it = g_p2p_groups.find(group_id);
            if (it != g_p2p_groups.end() && it->second == g) g_p2p_groups.erase(it);
        }
        (void)hipFree(inbox);
        return sc_internal_fail(code, "%s", msg);
    };
    {
        std::unique_lock<std::mutex> lk(g->mu);
        if (g->nranks != nranks || g->inbox[rank] || g->broken) {
            lk.unlock();
            (void)hipFree(inbox);
            return sc_internal_fail(SC_ERR_BAD_ARG, "p2p group %llu: rank %d joined twice, or the ranks disagree on the group's size", (unsigned long long)group_id, rank);
        }
        g->inbox[rank] = inbox;
        g->device[rank] = sc_internal_device_ref();
        ++g->joined;
        g->cv.notify_all();
        if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->joined == g->nranks || g->broken; }) || g->broken) {
            lk.unlock();
            return give_up(SC_ERR_HIP, "p2p group: not every rank joined within 60 s");
        }
    }
    sc_comm *c = new (std::nothrow) sc_comm();
    if (!c) return give_up(SC_ERR_OOM, "host allocation failed");
    c->rank = rank;
    c->nranks = nranks;
    c->p2p = g;
    c->p2p_id = group_id;
    c->device = sc_internal_device_ref();
    for (int q = 0; q < nranks; ++q) {
        if (q == rank) continue;
        if (g->device[q] == sc_internal_device_ref()) {
            c->p2p_shared_device = true;
            continue;
        }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, sc_internal_device_ref(), g->device[q]) != hipSuccess || !can) {
            delete c;
            return give_up(SC_ERR_HIP, "p2p group: a peer device is not accessible from this one (no xGMI / PCIe peer access)");
        }
        const hipError_t e = hipDeviceEnablePeerAccess(g->device[q], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            delete c;
            return give_up(SC_ERR_HIP, "hipDeviceEnablePeerAccess failed");
        }
        (void)hipGetLastError();
    }
    // (a rank that shares its GPU makes the whole group cautious: every rank must agree on whether rounds are pipelined)
    void *all[scd::kP2PMaxRanks];
    if (!g->exchange(rank, c->p2p_shared_device ? (void *)1 : nullptr, all)) {
        delete c;
        return give_up(SC_ERR_HIP, "p2p group: a rank dropped out during set-up");
    }
    for (int q = 0; q < nranks; ++q) c->p2p_shared_device |= all[q] != nullptr;
    *out = c;
    return SC_OK;
}

// p2p: every rank's `bytes` from d_send into every rank's d_recv (rank order), by peer copies; returns when all pieces are in place
int p2p_allgather(sc_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t s) {
    void *recv[scd::kP2PMaxRanks];
    HIP_TRY(hipStreamSynchronize(s)); // d_send is complete (and d_recv no longer read by this rank's earlier work)
    {
        GateYield yield_(c->device, true);
        if (!c->p2p->exchange(c->rank, d_recv, recv)) return sc_internal_fail(SC_ERR_HIP, "p2p group: a rank did not reach the gather");
    }
    for (int q = 0; q < c->nranks; ++q) {
        char *dst = static_cast<char *>(recv[q]) + (size_t)c->rank * bytes;
        if (c->p2p->device[q] == c->device) HIP_TRY(hipMemcpyAsync(dst, d_send, bytes, hipMemcpyDeviceToDevice, s));
        else HIP_TRY(hipMemcpyPeerAsync(dst, c->p2p->device[q], d_send, c->device, bytes, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(c->device, true);
        if (!c->p2p->exchange(c->rank, nullptr, nullptr)) return sc_internal_fail(SC_ERR_HIP, "p2p group: a rank did not finish the gather");
    }
    return SC_OK;
}
// p2p: any number of lanes summed in place over the group (the ranks read each other's buffers directly)
int p2p_allreduce_table(sc_comm *c, uint64_t *d_lanes, size_t n_words, hipStream_t s) {
    void *all[scd::kP2PMaxRanks];
    uint64_t *tmp = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&tmp), n_words * 8));
    struct Free {
        void *p;
        ~Free() { (void)hipFree(p); }
    } free_tmp{tmp};
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(c->device, true);
        if (!c->p2p->exchange(c->rank, d_lanes, all)) return sc_internal_fail(SC_ERR_HIP, "p2p group: a rank did not reach the all-reduce");
    }
    scd::PeerLanes pl;
    std::memset(&pl, 0, sizeof(pl));
    pl.n = c->nranks;
    for (int q = 0; q < c->nranks; ++q) pl.p[q] = static_cast<const uint64_t *>(all[q]);
    HIP_TRY(scd::launch_sum_peer_lanes(pl, n_words, tmp, s));
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(c->device, true); // nobody overwrites its lanes while a peer still reads them
        if (!c->p2p->exchange(c->rank, nullptr, nullptr)) return sc_internal_fail(SC_ERR_HIP, "p2p group: a rank did not finish the all-reduce");
    }
    HIP_TRY(hipMemcpyAsync(d_lanes, tmp, n_words * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return SC_OK;
}

// Diagnostic: one all-reduce and one all-gather of known patterns over the communicator, checked on every rank.
extern "C" int sc_comm_selftest(sc_comm *c) {
    if (!c) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    const int G = c->nranks, n = 40;
    std::vector<uint64_t> lanes(n), gathered((size_t)n * G);
    for (int i = 0; i < n; ++i) lanes[i] = (uint64_t)(c->rank + 1) * (uint64_t)(i + 1) + ((uint64_t)(c->rank + 1) << 40);
    const std::vector<uint64_t> mine = lanes;
    if (c->comm) {
        HIP_TRY(hipSetDevice(sc_internal_device_ref()));
        uint64_t *d = nullptr, *dg = nullptr;
        HIP_TRY(hipMalloc(&d, n * 8));
        HIP_TRY(hipMalloc(&dg, (size_t)n * 8 * G));
        HIP_TRY(hipMemcpy(d, lanes.data(), n * 8, hipMemcpyHostToDevice));
        NCCL_TRY(g_nccl.AllGather(d, dg, (size_t)n, ncclUint64, c->comm, nullptr));
        NCCL_TRY(g_nccl.AllReduce(d, d, (size_t)n, ncclUint64, ncclSum, c->comm, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        HIP_TRY(hipMemcpy(lanes.data(), d, n * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(gathered.data(), dg, (size_t)n * 8 * G, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        (void)hipFree(dg);
    } else if (c->p2p) {
        HIP_TRY(hipSetDevice(c->device));
        DeviceGate gate_(c->device);
        uint64_t *d = nullptr, *dg = nullptr;
        HIP_TRY(hipMalloc(&d, n * 8));
        HIP_TRY(hipMalloc(&dg, (size_t)n * 8 * G));
        HIP_TRY(hipMemcpy(d, lanes.data(), n * 8, hipMemcpyHostToDevice));
        int rc = p2p_allgather(c, d, dg, (size_t)n * 8, nullptr);
        if (!rc) rc = p2p_allreduce_table(c, d, (size_t)n, nullptr);
        if (!rc) {
            HIP_TRY(hipMemcpy(lanes.data(), d, n * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(gathered.data(), dg, (size_t)n * 8 * G, hipMemcpyDeviceToHost));
        }
        {
            GateYield yield_(c->device, true); // (a peer may still be reading d: leave together)
            (void)c->p2p->exchange(c->rank, nullptr, nullptr);
        }
        (void)hipFree(d);
        (void)hipFree(dg);
        if (rc) return rc;
    } else if (G > 1) {
        GateYield yield_(sc_internal_device_ref(), true);
        if (c->h_allgather(c->ctx, mine.data(), gathered.data(), (size_t)n * 8) != 0) return sc_internal_fail(SC_ERR_HIP, "the host transport's all-gather failed");
        if (c->h_allreduce(c->ctx, lanes.data(), (size_t)n) != 0) return sc_internal_fail(SC_ERR_HIP, "the host transport's all-reduce failed");
    } else {
        gathered = mine;
    }
    const uint64_t tri = (uint64_t)G * (uint64_t)(G + 1) / 2;
    for (int i = 0; i < n; ++i) {
        if (lanes[i] != tri * (uint64_t)(i + 1) + (tri << 40)) return sc_internal_fail(SC_ERR_HIP, "all-reduce returned a wrong sum in word %d", i);
        for (int g = 0; g < G; ++g)
            if (gathered[(size_t)g * n + i] != (uint64_t)(g + 1) * (uint64_t)(i + 1) + ((uint64_t)(g + 1) << 40))
                return sc_internal_fail(SC_ERR_HIP, "all-gather returned a wrong word (rank %d, word %d)", g, i);
    }
    return SC_OK;
}
// gkr.hip: sum `n_words` uint64 lanes in device memory over the ranks of `comm`, in place; returns with the result visible on `s`
int sc_internal_allreduce_lanes(sc_comm *c, uint64_t *d_lanes, size_t n_words, hipStream_t s) {
    if (!c || c->nranks == 1) return SC_OK;
    if (c->comm) {
        NCCL_TRY(g_nccl.AllReduce(d_lanes, d_lanes, n_words, ncclUint64, ncclSum, c->comm, s));
        return SC_OK;
    }
    if (c->p2p) return p2p_allreduce_table(c, d_lanes, n_words, s);
    std::vector<uint64_t> h(n_words);
    HIP_TRY(hipMemcpyAsync(h.data(), d_lanes, n_words * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    {
        GateYield yield_(sc_internal_device_ref(), true);
        if (c->h_allreduce(c->ctx, h.data(), n_words) != 0) return sc_internal_fail(SC_ERR_HIP, "the host transport's all-reduce failed");
    }
    HIP_TRY(hipMemcpyAsync(d_lanes, h.data(), n_words * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // `h` goes out of scope
    return SC_OK;
}
int sc_internal_comm_ranks(sc_comm *c) { return c ? c->nranks : 1; }

extern "C" int sc_comm_info(sc_comm *c, int *rank, int *nranks, int *kind) {
    if (!c) return sc_internal_fail(SC_ERR_BAD_ARG, "null argument");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (kind) *kind = c->comm ? (SC_COMM_RCCL | (c->direct_publish ? SC_COMM_DIRECT_PUBLISH : 0)) : c->p2p ? SC_COMM_P2P : SC_COMM_HOST;
    return SC_OK;
}
// Measurement, collective: `iters` back-to-back exchanges of n_words uint64 lanes in exactly the form a sharded round uses on this
// communicator -- RCCL: ncclAllReduce on a stream, the publishing kernel, the host's poll of the flag; peer-to-peer: the one exchange
// kernel and the poll; host transport: the publishing kernel, the poll, the caller's all-reduce function -- each waited for before
// the next is issued, as the rounds of a proof are.  The sums are checked.  *us_mean_out = wall time per exchange on this rank.
extern "C" int sc_comm_exchange_bench(sc_comm *c, uint32_t n_words, uint32_t iters, double *us_mean_out, double *us_min_out) {
    if (!c || !us_mean_out || n_words == 0 || n_words > (uint32_t)scd::kP2PWords || iters == 0) return sc_internal_fail(SC_ERR_BAD_ARG, "bad argument");
    if (sc_device_count() <= 0) return sc_internal_fail(SC_ERR_HIP, "no HIP device visible: libsumcheck_hip has no CPU fallback");
    const int device = c->p2p ? c->device : sc_internal_device_ref();
    HIP_TRY(hipSetDevice(device));
    struct Res {
        hipStream_t s = nullptr;
        uint64_t *d = nullptr, *h = nullptr, *h_dev = nullptr;
        uint32_t *flag = nullptr, *flag_dev = nullptr;
        ~Res() {
            if (s) (void)hipStreamSynchronize(s);
            if (d) (void)hipFree(d);
            if (h) (void)hipHostFree(h);
            if (flag) (void)hipHostFree(flag);
            if (s) (void)hipStreamDestroy(s);
        }
    } R;
    {
        DeviceGate gate_(device);
        HIP_TRY(hipStreamCreateWithFlags(&R.s, hipStreamNonBlocking));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&R.d), (size_t)n_words * 8));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&R.h), (size_t)n_words * 8, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&R.h_dev), R.h, 0));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&R.flag), 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&R.flag_dev), R.flag, 0));
        *R.flag = 0;
    }
    const bool p2p = c->p2p != nullptr && c->nranks > 1;
    const bool direct = c->comm != nullptr && c->direct_publish; // RCCL delivering tagged lanes into the host-mapped page (no publish kernel)
    if (direct) std::memset(R.h, 0, (size_t)n_words * 8);
    std::vector<uint64_t> mine(n_words), lanes(n_words);
    double total_us = 0.0, min_us = 1e30;
    const uint64_t tri = (uint64_t)c->nranks * (uint64_t)(c->nranks + 1) / 2;
    for (uint32_t it = 0; it <= iters; ++it) { // (iteration 0 warms up and is not counted)
        const uint32_t tag = scd::wide_tag_of(it);
        for (uint32_t w = 0; w < n_words; ++w) mine[w] = ((uint64_t)(c->rank + 1) * (uint64_t)(w + 1 + it)) | (direct ? (uint64_t)tag << scd::kWideTagShift : 0);
        {
            DeviceGate gate_(device);
            HIP_TRY(hipMemcpyAsync(R.d, mine.data(), (size_t)n_words * 8, hipMemcpyHostToDevice, R.s));
            HIP_TRY(hipStreamSynchronize(R.s));
        }
        const uint32_t want = it + 1;
        scd::P2PArgs xa;
        const auto t0 = std::chrono::steady_clock::now();
        {
            DeviceGate gate_(device);
            if (c->comm) NCCL_TRY(g_nccl.AllReduce(R.d, direct ? R.h_dev : R.d, (size_t)n_words, ncclUint64, ncclSum, c->comm, R.s));
            if (p2p) {
                std::memset(&xa, 0, sizeof(xa));
                for (int q = 0; q < c->nranks; ++q) xa.inbox[q] = c->p2p->inbox[q];
                xa.nranks = c->nranks;
                xa.rank = c->rank;
                xa.n_words = (int)n_words;
                xa.gen = ++c->p2p_gen;
                xa.max_spins = c->p2p_shared_device ? 2048u : scd::wait_spins_default();
                HIP_TRY(scd::launch_p2p_allreduce(xa, R.d, R.h_dev, R.flag_dev, want, R.s));
            } else if (!direct) {
                HIP_TRY(scd::launch_publish_words(R.d, R.h_dev, (int)n_words, R.flag_dev, want, R.s));
            }
        }
        uint64_t spins = 0;
        for (;;) {
            if (direct) { // the poll is the fetch: every word tagged with nranks * this exchange's tag
                bool seen = true;
                for (int w = (int)n_words - 1; w >= 0 && seen; --w) seen = (__atomic_load_n(R.h + w, __ATOMIC_ACQUIRE) >> scd::kWideTagShift) == (uint64_t)c->nranks * tag;
                if (seen) break;
                if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > publish_timeout()) return sc_internal_fail(SC_ERR_HIP, "exchange %u did not arrive", it);
                continue;
            }
            const uint32_t f = __atomic_load_n(R.flag, __ATOMIC_ACQUIRE);
            if (f == want) break;
            if (p2p && f == (want | scd::kP2PRetryBit)) { // (ranks sharing a GPU: a peer's kernel was queued behind this one)
                if (!c->p2p_shared_device) return sc_internal_fail(SC_ERR_HIP, "p2p all-reduce: a peer's lanes did not arrive");
                __atomic_store_n(R.flag, 0u, __ATOMIC_RELEASE);
                std::this_thread::yield();
                DeviceGate gate_(device);
                HIP_TRY(scd::launch_p2p_allreduce(xa, R.d, R.h_dev, R.flag_dev, want, R.s));
                continue;
            }
            if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > publish_timeout()) return sc_internal_fail(SC_ERR_HIP, "exchange %u did not publish", it);
        }
        std::copy(R.h, R.h + n_words, lanes.begin());
        if (direct)
            for (uint64_t &w : lanes) w &= (1ULL << scd::kWideTagShift) - 1;
        if (!c->comm && !p2p && c->nranks > 1) {
            if (c->h_allreduce(c->ctx, lanes.data(), (size_t)n_words) != 0) return sc_internal_fail(SC_ERR_HIP, "the host transport's all-reduce failed");
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        for (uint32_t w = 0; w < n_words; ++w)
            if (lanes[w] != tri * (uint64_t)(w + 1 + it)) return sc_internal_fail(SC_ERR_HIP, "exchange %u returned a wrong sum in word %u", it, w);
        if (it > 0) {
            total_us += us;
            min_us = std::min(min_us, us);
        }
    }
    *us_mean_out = total_us / iters;
    if (us_min_out) *us_min_out = min_us;
    return SC_OK;
}

extern "C" void sc_comm_free(sc_comm *c) {
    if (!c) return;
    if (c->comm && g_nccl.CommDestroy) (void)g_nccl.CommDestroy(c->comm);
    if (c->p2p) { // the inboxes go when the LAST rank leaves: a peer's kernel may still be pushing into this one
        std::shared_ptr<P2PGroup> g = c->p2p;
        bool last = false;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            last = ++g->left == g->nranks;
        }
        if (last) {
            for (int q = 0; q < g->nranks; ++q)
                if (g->inbox[q]) {
                    (void)hipSetDevice(g->device[q]);
                    (void)hipDeviceSynchronize();
                    (void)hipFree(g->inbox[q]);
                }
            (void)hipSetDevice(sc_internal_device_ref());
            std::lock_guard<std::mutex> lk(g_p2p_mu);
            auto it = g_p2p_groups.find(c->p2p_id);
            if (it != g_p2p_groups.end() && it->second == g) g_p2p_groups.erase(it);
        }
    }
    delete c;
}

