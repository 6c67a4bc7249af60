#include "fullprover.hpp"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <spawn.h>
#include <stdexcept>
#include <sys/wait.h>
#include <unistd.h>

#include "json_min.hpp"

extern char **environ;

namespace {

// BN254 scalar field order, little-endian (the reference compares decimal strings, src/fullprover.cpp:31-35)
const uint8_t kBn254Order[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                 0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

// ZKHIP_FIXED_R / ZKHIP_FIXED_S (64 hex digits, LE): deterministic proofs for parity tests
bool scalarFromEnv(const char *name, uint8_t out[32]) {
    const char *v = getenv(name);
    if (!v || strlen(v) != 64) return false;
    for (int i = 0; i < 32; i++) {
        unsigned x;
        if (sscanf(v + 2 * i, "%2x", &x) != 1) return false;
        out[i] = (uint8_t)x;
    }
    return true;
}

// The witness generator hand-off of the reference (src/fullprover.cpp:112-135): run
//   ./build/<circuit> <input.json> <out.wtns>
// with its standard output captured, echo the output and the raw wait status like the reference does,
// and (unlike it, quirk Q11) fail the job on a non-zero exit.  posix_spawn instead of popen: no shell
// between the server and the generator.
int runGenerator(const std::string &exe, const std::string &inputFile, const std::string &witnessFile, std::string &captured) {
    int fds[2];
    if (pipe(fds) != 0) throw std::runtime_error("Couldn't start command.");
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, fds[1], STDOUT_FILENO);
    posix_spawn_file_actions_addclose(&fa, fds[0]);
    posix_spawn_file_actions_addclose(&fa, fds[1]);
    std::string a0 = exe, a1 = inputFile, a2 = witnessFile;
    char *argv[] = {a0.data(), a1.data(), a2.data(), nullptr};
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, exe.c_str(), &fa, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    close(fds[1]);
    if (rc != 0) {
        close(fds[0]);
        throw std::runtime_error("Couldn't start command.");
    }
    char chunk[4096];
    for (ssize_t k; (k = read(fds[0], chunk, sizeof chunk)) > 0;) captured.append(chunk, (size_t)k);
    close(fds[0]);
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
    }
    return status;
}

std::vector<int> workerDevicesFromEnv() {
    const char *w = getenv("ZKHIP_WORKERS");
    if (!w || !*w) return {-1};                       // one replica on ZKHIP_DEVICE / ZKHIP_DEVICES / the current device
    if (std::string(w) == "all") {
        int n = 0;
        if (zk_device_count(&n) != 0 || n <= 0) throw std::runtime_error(zk_last_error());
        std::vector<int> v;
        for (int i = 0; i < n; i++) v.push_back(i);
        return v;
    }
    std::vector<int> v;
    for (int32_t d : Groth16::parseDeviceList(w)) v.push_back(d);
    return v;
}

}   // namespace

// (depth per replica: what the library plans for the prover — zk_prover_info through makeProver(..., kLibraryDepth): three proofs in
// flight saturate the GPU on large circuits, and each costs GiBs of workspace; small circuits are latency-bound and want the maximum)

FullProver::FullProver(std::string zkeyFileNames[], int size) {
    workerDevices = workerDevicesFromEnv();
    if (const char *q = getenv("ZKHIP_QUEUE")) queueCap = (size_t)strtoul(q, nullptr, 10);
    keepResults = std::max<size_t>(4096, 4 * queueCap);
    if (const char *kr = getenv("ZKHIP_KEEP_RESULTS")) keepResults = std::max<size_t>(1, (size_t)strtoul(kr, nullptr, 10));
    for (int i = 0; i < size; i++) {
        const std::string circuit = std::filesystem::path(zkeyFileNames[i]).stem().string();   // circuit name = file stem (fullprover.cpp:14-19,25)
        auto zkey = BinFileUtils::openExisting(zkeyFileNames[i], "zkey", 1);
        auto hdr = ZKeyUtils::loadHeader(zkey.get());
        if (memcmp(hdr->rPrime.data(), kBn254Order, 32) != 0) throw std::invalid_argument("zkey curve not supported");
        const uint64_t sizes[6] = {zkey->getSectionSize(4), zkey->getSectionSize(5), zkey->getSectionSize(6),
                                   zkey->getSectionSize(7), zkey->getSectionSize(8), zkey->getSectionSize(9)};
        Circuit &c = circuits[circuit];
        // throughput mode, small circuits: a submission carries up to `batch` witnesses (one set of kernel launches:
        // a 2^14 proof is ~60 latency-bound kernels, four proofs in them cost little more than one).  ZKHIP_BATCH=n
        // (0/1 = off, at most ZK_MAX_BATCH) overrides the default: 8 up to 2^16 constraints, 4 at 2^17 (round 4, with the host
        // tails of a submission running side by side: 0.39 -> 0.35 / 0.54 -> 0.48 / 0.81 -> 0.68 ms per proof at 2^14 / 2^15 /
        // 2^16 for eight instead of four, profiles/r04af_batch_sweep.txt).
        uint32_t batch = 0;
        if (queueMode()) {
            const char *be = getenv("ZKHIP_BATCH");
            batch = be ? (uint32_t)strtoul(be, nullptr, 10) : (hdr->domainSize <= (1u << 16) ? 8u : hdr->domainSize <= (1u << 17) ? 4u : 0u);
        }
        // every slot and lane the pipeline will walk is allocated at start-up (an out-of-memory there falls back to the tables
        // as in the zkey, or ends the start — never a proof later)
        const uint32_t reserve = queueMode() ? Groth16::kLibraryDepth : 1u;
        for (int dev : workerDevices)
            c.replica.push_back(Groth16::makeProver(hdr->nVars, hdr->nPublic, hdr->domainSize, hdr->nCoefs, hdr->vk_alpha1, hdr->vk_beta1,
                                                    hdr->vk_beta2, hdr->vk_delta1, hdr->vk_delta2, zkey->getSectionData(4),
                                                    zkey->getSectionData(5), zkey->getSectionData(6), zkey->getSectionData(7),
                                                    zkey->getSectionData(8), zkey->getSectionData(9), sizes, /*precompDefault=*/true, dev, batch, reserve));
        // libzkhip copied everything it needs to the GPU: only the scalar header fields are kept
        // (the vk pointers into the mapping die with `zkey` and are never used again here)
        hdr->vk_alpha1 = hdr->vk_beta1 = hdr->vk_beta2 = hdr->vk_gamma2 = hdr->vk_delta1 = hdr->vk_delta2 = nullptr;
        c.header = std::move(hdr);
        std::cerr << "circuit: " << circuit << '\n';
    }
    if (queueMode()) {
        size_t nw = 4;
        if (const char *t = getenv("ZKHIP_WITNESS_THREADS")) nw = (size_t)strtoul(t, nullptr, 10);
        if (nw == 0) nw = 1;
        for (size_t i = 0; i < nw; i++) threads.emplace_back(&FullProver::witnessLoop, this);
        for (size_t w = 0; w < workerDevices.size(); w++) threads.emplace_back(&FullProver::deviceLoop, this, w);
        std::cerr << "throughput mode: queue " << queueCap << ", " << workerDevices.size() << " GPU worker(s), " << nw << " witness thread(s)\n";
    }
}

FullProver::~FullProver() {
    {
        std::lock_guard<std::mutex> guard(mtx);
        stopping = true;
    }
    cvIncoming.notify_all();
    cvReady.notify_all();
    for (auto &t : threads) t.join();
}

// Everything between the request body and the call of prove(): input file, generator, witness image,
// public signals (src/fullprover.cpp:104-152).  `tag` distinguishes the files of concurrent jobs; it is
// empty in single-slot mode, where the paths are exactly the reference's.
void FullProver::generateWitness(Job &job, const std::string &tag) {
    if (!JsonMin::isValid(job.input)) throw std::runtime_error("input is not valid JSON");
    auto known = circuits.find(job.circuit);
    if (known == circuits.end()) throw std::runtime_error("unknown circuit: " + job.circuit);
    const std::string base = "./build/";
    const std::string inputFile = base + "input_" + job.circuit + tag + ".json";
    const std::string witnessFile = base + job.circuit + tag + ".wtns";
    // concurrent jobs (tag non-empty): the per-job files go away on EVERY path out of here — the mapping keeps a good
    // witness alive, a failing generator / bad header / nVars mismatch must not leave files in ./build
    struct Cleanup {
        const std::string &a, &b;
        bool on;
        ~Cleanup() {
            if (on) {
                std::remove(a.c_str());
                std::remove(b.c_str());
            }
        }
    } cleanup{inputFile, witnessFile, !tag.empty()};
    {
        std::ofstream f(inputFile);
        f << job.input;
    }
    // the body (up to 128 MB) is on disk now: a job stays in the `jobs` map long after it is done, its request body must not
    std::string().swap(job.input);
    std::string output;
    const int waitStatus = runGenerator(base + job.circuit, inputFile, witnessFile, output);
    std::cout << output << std::endl;
    std::cout << waitStatus << std::endl;
    if (waitStatus != 0)
        throw std::runtime_error("witness generator failed with code " + std::to_string(WIFEXITED(waitStatus) ? WEXITSTATUS(waitStatus) : waitStatus));

    job.wtns = BinFileUtils::openExisting(witnessFile, "wtns", 2);
    adoptWitness(job, known->second.header.get());
}

void FullProver::adoptWitness(Job &job, const ZKeyUtils::Header *zh) {
    auto wh = WtnsUtils::loadHeader(job.wtns.get());
    if (memcmp(wh->prime.data(), kBn254Order, 32) != 0) throw std::invalid_argument("different wtns curve");
    if (wh->nVars != zh->nVars || job.wtns->getSectionSize(2) < (uint64_t)zh->nVars * 32)
        throw std::invalid_argument("witness does not match the zkey (nVars)");
    job.wtnsData = static_cast<const uint8_t *>(job.wtns->getSectionData(2));
    std::string pub(zk_public_to_json(job.wtnsData, zh->nPublic, nullptr, 0), '\0');
    zk_public_to_json(job.wtnsData, zh->nPublic, pub.data(), pub.size() + 1);
    job.pubData = pub;
}

// ------------------------------------------------------------------ single-slot mode (the reference's)
void FullProver::startProve(std::string input, std::string circuit) {
    std::lock_guard<std::mutex> guard(mtx);
    pending = std::make_shared<Job>();
    pending->input = std::move(input);
    pending->circuit = std::move(circuit);
    if (status == busy && executing) executing->canceled = true;   // reference: abort() here re-locks mtx and deadlocks (Q2)
    checkPending();
}

void FullProver::checkPending() {
    if (status == busy || !pending) return;
    if (pending->input.empty() || pending->circuit.empty()) return;
    status = busy;
    executing = pending;
    pending.reset();
    std::thread(&FullProver::runSingle, this, executing).detach();
}

void FullProver::runSingle(JobPtr job) {
    std::string proofJson = "null", error;
    try {
        generateWitness(*job, "");
        bool run;
        {
            std::lock_guard<std::mutex> guard(mtx);
            run = !job->canceled;
        }
        uint8_t r[32], s[32];
        const bool haveR = scalarFromEnv("ZKHIP_FIXED_R", r), haveS = scalarFromEnv("ZKHIP_FIXED_S", s);
        if (run) proofJson = circuits[job->circuit].replica[0]->prove(job->wtnsData, haveR ? r : nullptr, haveS ? s : nullptr)->toJson();   // HOT PATH (fullprover.cpp:155)
    } catch (std::exception &e) {   // reference catches runtime_error only: a JSON error kills it (Q3)
        error = e.what();
    }
    std::lock_guard<std::mutex> guard(mtx);
    job->wtns.reset();
    job->proof = proofJson;
    job->error = job->canceled ? "" : error;
    job->status = job->canceled ? aborted : (!error.empty() ? failed : success);
    status = job->status;
    last = job;
    executing.reset();
    checkPending();
}

void FullProver::abort() {
    std::lock_guard<std::mutex> guard(mtx);
    if (queueMode()) {           // cancels everything that has not reached a GPU yet: waiting for a generator, or ready for a dispatcher
        for (auto *q : {&incoming, &readyJobs}) {
            for (auto &j : *q) {
                j->status = aborted;
                j->canceled = true;
                j->wtns.reset();
                std::string().swap(j->input);      // the job stays in `jobs` for status polls, its request body need not
            }
            q->clear();
        }
        abortEpoch++;            // a job inside its generator right now finishes it and is dropped in witnessLoop
        return;
    }
    if (status == busy && executing) executing->canceled = true;
}

// Same documents as nlohmann's dump() of FullProver::getStatus (fullprover.cpp:216-240): keys in
// alphabetical order, compact; proof and pubData are STRINGS containing JSON.
std::string FullProver::statusDocument(const Job &job) {
    switch (job.status) {
        case ready: return "{\"status\":\"ready\"}";
        case aborted: return "{\"status\":\"aborted\"}";
        case failed: return "{\"error\":" + JsonMin::quote(job.error) + ",\"status\":\"failed\"}";
        case success: return "{\"proof\":" + JsonMin::quote(job.proof) + ",\"pubData\":" + JsonMin::quote(job.pubData) + ",\"status\":\"success\"}";
        case busy: return "{\"status\":\"busy\"}";
    }
    return "{}";
}

std::string FullProver::getStatus() {
    std::lock_guard<std::mutex> guard(mtx);
    if (queueMode()) {
        if (jobs.empty()) return "{\"status\":\"ready\"}";
        return statusDocument(*jobs.rbegin()->second);
    }
    if (status == busy) return "{\"status\":\"busy\"}";
    if (status == ready || !last) return "{\"status\":\"ready\"}";
    return statusDocument(*last);
}

// ------------------------------------------------------------------ throughput mode
bool FullProver::enqueue(std::string input, std::string circuit, uint64_t &id) {
    std::lock_guard<std::mutex> guard(mtx);
    // everything that has not reached a GPU counts: waiting for a generator, inside one, or ready for a dispatcher (each
    // ready job holds a mapped witness image) — the 503 must fire when the GPUs are the slow side too
    if (incoming.size() + inWitness + readyJobs.size() >= queueCap) return false;
    JobPtr j = std::make_shared<Job>();
    j->id = id = nextId++;
    j->epoch = abortEpoch;
    j->input = std::move(input);
    j->circuit = std::move(circuit);
    incoming.push_back(j);
    remember(j);
    cvIncoming.notify_one();
    return true;
}

bool FullProver::enqueueWitness(std::string wtnsImage, std::string circuit, uint64_t &id) {
    {   // a place in the queue FIRST (counted like a job inside a generator): a request that is going to get a 503 costs
        // nothing — the image is attacker-sized (up to 128 MB, millions of directory entries) and parsed on the HTTP thread
        std::lock_guard<std::mutex> guard(mtx);
        if (incoming.size() + inWitness + readyJobs.size() >= queueCap) return false;
        inWitness++;
    }
    // the reserved place is given back on EVERY way out of the parsing below — a bad_alloc beside a 128 MB body or any other
    // exception that is not a std::exception used to leave it taken for good (the queue answered 503 forever after enough of them)
    struct Place {
        FullProver *fp;
        bool held = true;
        void release_locked() {
            if (held) fp->inWitness--;
            held = false;
        }
        ~Place() {
            if (!held) return;
            std::lock_guard<std::mutex> guard(fp->mtx);
            fp->inWitness--;
        }
    } place{this};
    JobPtr j;
    std::string error;
    try {        // nothing here needs the prover's lock
        j = std::make_shared<Job>();
        j->circuit = std::move(circuit);
        auto known = circuits.find(j->circuit);
        if (known == circuits.end()) throw std::runtime_error("unknown circuit: " + j->circuit);
        j->wtns = BinFileUtils::fromMemory(std::move(wtnsImage), "wtns", 2);
        adoptWitness(*j, known->second.header.get());
    } catch (std::exception &e) {
        error = e.what();
    } catch (...) {
        error = "witness could not be read";
    }
    if (!j) throw std::runtime_error("out of memory for a witness job");        // (the place is released by ~Place; the HTTP layer answers 500)
    std::lock_guard<std::mutex> guard(mtx);
    place.release_locked();                      // the reserved place becomes a ready job (or is given back)
    j->id = id = nextId++;
    j->epoch = abortEpoch;
    remember(j);
    if (!error.empty()) {
        j->wtns.reset();
        j->error = error;
        j->status = failed;
        return true;
    }
    readyJobs.push_back(j);
    cvReady.notify_one();
    return true;
}

void FullProver::remember(const JobPtr &job) {
    jobs[job->id] = job;
    // Results wait here for their owner's GET /status/<id>.  Only FINISHED jobs are ever dropped, oldest first, once more than
    // keepResults jobs are known: the first version dropped the 4096th-oldest job whatever its state — a client that submitted
    // 8192 requests before polling found half of them "unknown" while they were still queued (tools/soak.py's server run).
    for (auto it = jobs.begin(); jobs.size() > keepResults && it != jobs.end();) {
        if (it->second->status != busy) it = jobs.erase(it);
        else ++it;
    }
}

std::string FullProver::getStatus(uint64_t id) {
    std::lock_guard<std::mutex> guard(mtx);
    auto it = jobs.find(id);
    if (it == jobs.end()) return "{\"error\":\"unknown job\",\"status\":\"failed\"}";
    return statusDocument(*it->second);
}

// host side of the pipeline: witness generators of later jobs run while earlier proofs are on the GPUs
void FullProver::witnessLoop() {
    for (;;) {
        JobPtr job;
        {
            std::unique_lock<std::mutex> lk(mtx);
            cvIncoming.wait(lk, [&] { return stopping || !incoming.empty(); });
            if (stopping) return;
            job = incoming.front();
            incoming.pop_front();
            inWitness++;
        }
        std::string error;
        try {
            generateWitness(*job, "." + std::to_string(job->id));
        } catch (std::exception &e) {
            error = e.what();
        }
        std::lock_guard<std::mutex> guard(mtx);
        inWitness--;
        if (job->canceled || job->epoch < abortEpoch) {             // /cancel arrived while the generator ran
            job->wtns.reset();
            job->status = aborted;
            continue;
        }
        if (!error.empty()) {
            job->wtns.reset();
            job->error = error;
            job->status = failed;
            continue;
        }
        readyJobs.push_back(job);
        cvReady.notify_one();
    }
}

// Two threads per GPU.  The SUBMITTER takes ready jobs and enqueues their proofs (up to `depth` in flight per
// circuit replica); the COLLECTOR retires them in submission order (a collect returns the oldest proof of
// THAT prover, and the global FIFO preserves every prover's order): it waits for the GPU and runs the host
// tail of a proof (Horner over the window sums + final assembly, ≈ 1 ms) while the submitter is already
// enqueueing the next ones — on Semaphore-sized circuits the host time of a one-thread loop, not the GPU,
// was the bound.
void FullProver::deviceLoop(size_t worker) {
    std::mutex wm;
    std::condition_variable cv;
    std::deque<std::vector<JobPtr>> fifo;          // submissions (one or, on a batch replica, several jobs) not yet collected
    std::map<std::string, size_t> perCircuit;
    bool submitterDone = false;
    uint8_t r[32], s[32];
    const bool haveR = scalarFromEnv("ZKHIP_FIXED_R", r), haveS = scalarFromEnv("ZKHIP_FIXED_S", s);

    std::thread collector([&] {
        for (;;) {
            std::vector<JobPtr> jobs;
            {
                std::unique_lock<std::mutex> lk(wm);
                cv.wait(lk, [&] { return submitterDone || !fifo.empty(); });
                if (fifo.empty()) return;
                jobs = fifo.front();
            }
            std::vector<std::string> proofJson(jobs.size());
            std::string error;
            try {
                Groth16::Prover &pr = *circuits[jobs[0]->circuit].replica[worker];
                if (jobs.size() == 1 && pr.batch() == 1) {
                    proofJson[0] = pr.collect()->toJson();
                } else {
                    auto proofs = pr.collectBatch((uint32_t)jobs.size());
                    for (size_t k = 0; k < jobs.size(); k++) proofJson[k] = proofs[k]->toJson();
                }
            } catch (std::exception &e) {
                error = e.what();
            }
            {
                std::lock_guard<std::mutex> guard(mtx);
                for (size_t k = 0; k < jobs.size(); k++) {
                    jobs[k]->wtns.reset();
                    jobs[k]->proof = error.empty() ? proofJson[k] : "null";
                    jobs[k]->error = error;
                    jobs[k]->status = error.empty() ? success : failed;
                }
            }
            {
                std::lock_guard<std::mutex> lk(wm);
                fifo.pop_front();
                perCircuit[jobs[0]->circuit]--;
            }
            cv.notify_all();
        }
    });

    for (;;) {
        JobPtr job;
        std::vector<JobPtr> jobs;
        {
            std::unique_lock<std::mutex> lk(mtx);
            cvReady.wait(lk, [&] { return stopping || !readyJobs.empty(); });
            if (stopping) break;
            job = readyJobs.front();
            readyJobs.pop_front();
            jobs.push_back(job);
        }
        size_t depth = ZK_MAX_IN_FLIGHT;            // (a replica always reports what start-up reserved: the library's depth, or what fitted)
        if (const uint32_t fits = circuits[job->circuit].replica[worker]->reservedInFlight()) depth = std::min(depth, (size_t)fits);
        {
            std::unique_lock<std::mutex> lk(wm);
            cv.wait(lk, [&] { return perCircuit[job->circuit] < depth; });
        }
        {
            // a batch replica: whatever else is READY NOW for the same circuit rides along (it never waits for more)
            std::lock_guard<std::mutex> lk(mtx);
            const uint32_t cap = circuits[job->circuit].replica[worker]->batch();
            for (auto it = readyJobs.begin(); it != readyJobs.end() && jobs.size() < cap;) {
                if ((*it)->circuit == job->circuit) {
                    jobs.push_back(*it);
                    it = readyJobs.erase(it);
                } else {
                    ++it;
                }
            }
        }
        try {
            Groth16::Prover &pr = *circuits[job->circuit].replica[worker];
            if (pr.batch() == 1) {
                pr.submit(job->wtnsData, haveR ? r : nullptr, haveS ? s : nullptr);
            } else {
                std::vector<const void *> ws;
                for (auto &j : jobs) ws.push_back(j->wtnsData);
                pr.submitBatch(ws, haveR ? r : nullptr, haveS ? s : nullptr);
            }
            {
                std::lock_guard<std::mutex> lk(wm);
                fifo.push_back(jobs);
                perCircuit[job->circuit]++;
            }
            cv.notify_all();
        } catch (std::exception &e) {
            std::lock_guard<std::mutex> guard(mtx);
            for (auto &j : jobs) {
                j->wtns.reset();
                j->error = e.what();
                j->status = failed;
            }
        }
    }
    {
        std::lock_guard<std::mutex> lk(wm);
        submitterDone = true;
    }
    cv.notify_all();
    collector.join();
}
