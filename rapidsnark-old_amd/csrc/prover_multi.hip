// One proof on several GPUs (the reference has no multi-GPU path, src/fullprover.cpp:96-97):
// (a) zk_shard_*: one process per GPU (torch.distributed / RCCL): the caller owns the exchange of the
//     chain's blocks — four all_to_all per proof on buffers it registered — and drives the phases.
// (b) zk_multi_prover: all GPUs of the node in ONE process (what the reference's CLI and server are):
//     one shard prover per device, phases enqueued device by device, blocks exchanged by peer writes
//     over xGMI, cross-device ordering by events (hipStreamWaitEvent across devices).
#include "prover_internal.hpp"

namespace {

// stream 1 of the prover <-> the caller's stream (on which its collectives are ordered)
// (a NULL handle is the default stream, which is what torch.cuda.current_stream() is unless the caller
// switched streams: it must be ordered like any other)
void ext_in(zk_prover *p, void *stream) {
    HIP_TRY(hipEventRecord(p->ev_ext_in, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_ext_in, 0));
}
void ext_out(zk_prover *p, void *stream) {
    HIP_TRY(hipEventRecord(p->ev_ext_out, p->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->ev_ext_out, 0));
}

}   // namespace

struct zk_multi_prover {
    std::vector<zk_prover *> shard;
    std::vector<hipEvent_t> ev[4];       // per shard: chunks pushed (DIF), cross DIF done, chunks pushed (DIT), cross DIT done
    uint8_t *stage_pin[ZK_MAX_IN_FLIGHT] = {nullptr};
    hipEvent_t ev_staged[ZK_MAX_IN_FLIGHT] = {nullptr};
    StageJob stage[ZK_MAX_IN_FLIGHT];
    uint64_t submitted = 0;
    bool part = false;
    std::mutex mtx, cmtx, sync_mtx;      // submit / collect / one synchronous call at a time (as in zk_prover)
    ~zk_multi_prover() {
        for (zk_prover *q : shard) zk_prover_destroy(q);        // drains every stream first
        for (auto &v : ev) for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e);
        for (auto &b : stage_pin) if (b) (void)hipHostFree(b);
        for (auto &e : ev_staged) if (e) (void)hipEventDestroy(e);
    }
};

namespace {

void multi_create(zk_multi_prover **out, const zk_zkey_view *z, const int32_t *devices, uint32_t nd, const zk_opts *o) {
    if (!out || !z || !devices || nd == 0) throw std::invalid_argument("null argument");
    if (nd > 8) throw std::invalid_argument("at most 8 devices");
    need_device_count();
    std::unique_ptr<zk_multi_prover> mp(new zk_multi_prover());
    uint32_t lg = 0;
    while ((1u << lg) < nd) lg++;
    uint32_t logn = 0;
    while ((1ull << logn) < z->domainSize) logn++;
    bool can_part = nd > 1 && (1u << lg) == nd && logn >= 2 * lg && !getenv("ZKHIP_REPLICATED_CHAIN");
    // the partitioned chain writes into its peers' buffers: every pair of distinct devices must be able to map each other
    // (xGMI inside a node).  Where one cannot, the chain stays replicated (partial sums only travel through the host).
    bool peers_ok = true;
    for (uint32_t a = 0; a < nd && peers_ok; a++)
        for (uint32_t b = 0; b < nd && peers_ok; b++) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
            if (!can) peers_ok = false;
        }
    if (!peers_ok) can_part = false;
    mp->part = can_part;
    for (uint32_t g = 0; g < nd; g++) {
        zk_opts so;
        memset(&so, 0, sizeof so);
        so.device = devices[g];
        so.shard_index = g;
        so.shard_count = nd;
        so.window_bits = o ? o->window_bits : 0;
        so.flags = (o ? o->flags : 0) & ~ZK_FLAG_PARTITIONED_CHAIN;
        if (can_part) so.flags |= ZK_FLAG_PARTITIONED_CHAIN;
        zk_prover *q = nullptr;
        prover_create(&q, z, &so);
        mp->shard.push_back(q);
    }
    // peer access between every pair of distinct devices (xGMI inside a node); only the partitioned chain needs it
    for (uint32_t a = 0; a < nd && can_part; a++)
        for (uint32_t b = 0; b < nd; b++) {
            if (devices[a] == devices[b]) continue;
            DeviceGuard g(devices[a]);
            hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else HIP_TRY(e);
        }
    for (uint32_t a = 0; a < nd; a++) {
        zk_prover *q = mp->shard[a];
        for (uint32_t b = 0; b < nd; b++) {
            q->peer_abc[b] = mp->shard[b]->abc_use;
            q->peer_xb[b] = mp->shard[b]->xb_use;
        }
        q->have_peers = can_part;
        DeviceGuard g(q->device);
        for (auto &v : mp->ev) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            v.push_back(e);
        }
    }
    {
        DeviceGuard g(mp->shard[0]->device);
        for (auto &e : mp->ev_staged) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = mp.release();
}

// every stream-1 waits for the given event of EVERY shard (the all-to-all dependency of a cross step)
void wait_all(zk_multi_prover *mp, int which) {
    for (zk_prover *q : mp->shard) {
        DeviceGuard g(q->device);
        for (size_t b = 0; b < mp->shard.size(); b++) HIP_TRY(hipStreamWaitEvent(q->stream, mp->ev[which][b], 0));
    }
}

void multi_submit(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32) {
    const size_t G = mp->shard.size();
    std::vector<std::unique_lock<std::mutex>> locks;
    for (zk_prover *q : mp->shard) locks.emplace_back(q->mtx);
    zk_prover *p0 = mp->shard[0];
    if (p0->in_flight >= ZK_MAX_IN_FLIGHT) throw std::invalid_argument("too many proofs in flight (ZK_MAX_IN_FLIGHT): collect one first");
    // the witness goes to every GPU (each needs all of it for its rows of A.w / B.w): staged ONCE into
    // pinned memory (host function on shard 0's upload stream), then G DMA copies
    const uint8_t *src = wtns;
    hipEvent_t ready = nullptr;
    {
        hipPointerAttribute_t attr;
        const bool pinned = hipPointerGetAttributes(&attr, wtns) == hipSuccess && attr.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (!pinned && G > 1) {
            const int k = (int)(mp->submitted % ZK_MAX_IN_FLIGHT);
            const size_t bytes = (size_t)p0->nVars * 32;
            DeviceGuard g(p0->device);
            if (!mp->stage_pin[k]) HIP_TRY(hipHostMalloc((void **)&mp->stage_pin[k], bytes, hipHostMallocPortable));
            mp->stage[k] = StageJob{mp->stage_pin[k], wtns, bytes};
            HIP_TRY(hipLaunchHostFunc(p0->stream_h2d, stage_job_run, &mp->stage[k]));
            HIP_TRY(hipEventRecord(mp->ev_staged[k], p0->stream_h2d));
            src = mp->stage_pin[k];
            ready = mp->ev_staged[k];
        }
    }
    std::vector<PhaseAbort> guards;
    guards.reserve(G);
    for (zk_prover *q : mp->shard) guards.push_back(PhaseAbort{q});
    if (!mp->part) {
        for (zk_prover *q : mp->shard) {
            phase_front(q, nullptr, src, r32, s32, ready);
            phase_local(q);
            phase_back(q);
        }
    } else {
        for (size_t a = 0; a < G; a++) {
            phase_front(mp->shard[a], nullptr, src, r32, s32, ready);
            phase_cross_push(mp->shard[a], mp->ev[0][a]);
        }
        wait_all(mp, 0);
        for (size_t a = 0; a < G; a++) phase_cross_run(mp->shard[a], true, mp->ev[1][a]);
        wait_all(mp, 1);
        for (size_t a = 0; a < G; a++) {
            phase_local(mp->shard[a]);
            phase_cross_push(mp->shard[a], mp->ev[2][a]);
        }
        wait_all(mp, 2);
        for (size_t a = 0; a < G; a++) phase_cross_run(mp->shard[a], false, mp->ev[3][a]);
        wait_all(mp, 3);
        for (size_t a = 0; a < G; a++) phase_back(mp->shard[a]);
    }
    for (auto &gd : guards) gd.armed = false;
    mp->submitted++;
}

void multi_collect(zk_multi_prover *mp, zk_proof *out) {
    const size_t G = mp->shard.size();
    std::vector<zk_msm_sums> sums(G);
    SubmittedRS rs;
    // EVERY shard's oldest submission is retired, whatever one of them throws: a shard left behind would pair its partial
    // sums with the next proof's on every later collect
    std::exception_ptr first;
    for (size_t a = 0; a < G; a++) {
        try {
            collect_sums(mp->shard[a], &sums[a], a == 0 ? &rs : nullptr);
        } catch (...) {
            if (!first) first = std::current_exception();
        }
    }
    if (first) std::rethrow_exception(first);
    prove_finish(mp->shard[0], sums.data(), (uint32_t)G, rs.have_r ? rs.r32 : nullptr, rs.have_s ? rs.s32 : nullptr, out);
}

}   // namespace

extern "C" {

int zk_multi_prover_create(zk_multi_prover **out, const zk_zkey_view *zkey, const int32_t *devices, uint32_t n_devices, const zk_opts *opts) {
    return guarded([&] { multi_create(out, zkey, devices, n_devices, opts); });
}

void zk_multi_prover_destroy(zk_multi_prover *mp) { delete mp; }

int zk_multi_prove_submit(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32) {
    return guarded([&] {
        if (!mp || !wtns) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(mp->mtx);
        multi_submit(mp, wtns, r32, s32);
    });
}

int zk_multi_prove_collect(zk_multi_prover *mp, zk_proof *out) {
    return guarded([&] {
        if (!mp || !out) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(mp->cmtx);          // submissions (mp->mtx) go on meanwhile
        multi_collect(mp, out);
    });
}

int zk_multi_prove(zk_multi_prover *mp, const uint8_t *wtns, const uint8_t *r32, const uint8_t *s32, zk_proof *out) {
    return guarded([&] {
        if (!mp || !wtns || !out) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> one(mp->sync_mtx);
        {
            std::lock_guard<std::mutex> lk(mp->mtx);
            if (mp->shard[0]->in_flight) throw std::invalid_argument("asynchronous proofs in flight: collect them first");
            multi_submit(mp, wtns, r32, s32);
        }
        std::lock_guard<std::mutex> lk(mp->cmtx);
        multi_collect(mp, out);
    });
}

int zk_multi_prover_info(zk_multi_prover *mp, uint32_t *n_shards, uint32_t *chain_partitioned) {
    return guarded([&] {
        if (!mp) throw std::invalid_argument("null argument");
        if (n_shards) *n_shards = (uint32_t)mp->shard.size();
        if (chain_partitioned) *chain_partitioned = mp->part ? 1u : 0u;
    });
}

int zk_shard_info(zk_prover *p, uint64_t *block_elems, uint32_t *chain_partitioned) {
    return guarded([&] {
        if (!p) throw std::invalid_argument("null argument");
        if (block_elems) *block_elems = p->nloc;
        if (chain_partitioned) *chain_partitioned = p->part ? 1u : 0u;
    });
}

int zk_shard_set_exchange(zk_prover *p, void *d_send, void *d_recv) {
    return guarded([&] {
        if (!p || !d_send || !d_recv) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(p->mtx);
        if (!p->part) throw std::invalid_argument("prover was not created with ZK_FLAG_PARTITIONED_CHAIN");
        if (p->in_flight || p->phase_open >= 0) throw std::invalid_argument("proofs in flight");
        p->pk_use = (Fr *)d_send;
        p->xb_use = (Fr *)d_recv;
        p->xb.release();
    });
}

int zk_shard_begin(zk_prover *p, const uint8_t *wtns, const void *d_wtns, const uint8_t *r32, const uint8_t *s32, void *stream) {
    return guarded([&] {
        if (!p || (!wtns == !d_wtns)) throw std::invalid_argument("exactly one of wtns / d_wtns");
        std::lock_guard<std::mutex> lk(p->mtx);
        if (!p->part) throw std::invalid_argument("prover was not created with ZK_FLAG_PARTITIONED_CHAIN");
        DeviceGuard g(p->device);
        PhaseAbort guard{p};
        ext_in(p, stream);
        phase_front(p, (const Fr *)d_wtns, wtns, r32, s32);
        ext_out(p, stream);
        guard.armed = false;
    });
}

int zk_shard_step(zk_prover *p, int step, void *stream) {
    return guarded([&] {
        if (!p) throw std::invalid_argument("null argument");
        std::lock_guard<std::mutex> lk(p->mtx);
        DeviceGuard g(p->device);
        PhaseAbort guard{p};
        ext_in(p, stream);
        switch (step) {
        case ZK_STEP_CROSS_INVERSE: phase_cross_run(p, true, nullptr); break;
        case ZK_STEP_LOCAL: phase_local(p); break;
        case ZK_STEP_CROSS_FORWARD: phase_cross_run(p, false, nullptr); break;
        case ZK_STEP_FINISH: phase_back(p); break;
        default: throw std::invalid_argument("unknown step");
        }
        ext_out(p, stream);
        guard.armed = false;
    });
}

}   // extern "C"
