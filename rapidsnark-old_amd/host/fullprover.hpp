// FullProver — job orchestration of the reference server (src/fullprover.hpp:13-50,
// src/fullprover.cpp:21-240) over libzkhip: one prover per zkey keyed by file stem, witness
// generation by an external circom binary, prove on the MI355X.  Same public surface
// (startProve / abort / getStatus) and, by default, the same single-slot state machine: one job
// executes, one waits, the latest request wins.
//
// Throughput mode (BASELINE configs[4]; not in the reference, opt-in through the environment so the
// argv and the default behaviour stay the reference's):
//   ZKHIP_WORKERS=0,1,...|all   one resident replica of every circuit per listed GPU, one dispatcher
//                               thread per GPU ("replicas only": no collective)
//   ZKHIP_QUEUE=n               n > 0: requests are QUEUED (at most n waiting) instead of replacing
//                               each other; POST /input/:circuit answers {"job":id}, GET /status/<id>
//                               reports that job, GET /status the most recent one.  Every dispatcher
//                               keeps up to ZK_MAX_IN_FLIGHT proofs in flight on its GPU
//                               (zk_prove_submit / zk_prove_collect) while witness generators of later
//                               jobs run on the host (ZKHIP_WITNESS_THREADS of them, default 4).
// Deliberate deviations from reference bugs: Q2 no self-deadlock when a request
// arrives while busy; Q3 a malformed body fails the job instead of killing the process;
// Q5 getStatus takes the lock; Q11 a failing witness generator fails the job; an unknown
// circuit name fails the job (the reference dereferences a null map entry).
#pragma once
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "groth16.hpp"
#include "zkfile.hpp"

class FullProver {
public:
    FullProver(std::string zkeyFileNames[], int size);
    ~FullProver();
    void startProve(std::string input, std::string circuit);   // POST /input/:circuit (single-slot mode)
    void abort();                                              // POST /cancel
    std::string getStatus();                                   // the JSON document of GET /status

    // throughput mode
    bool queueMode() const { return queueCap > 0; }
    bool enqueue(std::string input, std::string circuit, uint64_t &id);   // false: queue full
    // POST /witness/:circuit — the witness arrives as a .wtns image in the request body (a generator running inside the
    // caller, or another host): no generator process, no files; everything else is the same job
    bool enqueueWitness(std::string wtnsImage, std::string circuit, uint64_t &id);
    std::string getStatus(uint64_t id);                                    // GET /status/<id>

private:
    enum Status { aborted = -2, busy = -1, failed = 0, success = 1, ready = 6 };

    struct Circuit {
        std::vector<std::unique_ptr<Groth16::Prover>> replica;   // one per worker GPU
        std::unique_ptr<ZKeyUtils::Header> header;               // scalar fields only (vk pointers are cleared after create)
    };
    struct Job {
        uint64_t id = 0;
        std::string input, circuit;
        Status status = busy;
        std::string proof = "null", pubData, error;
        std::unique_ptr<BinFileUtils::BinFile> wtns;   // the witness image stays mapped until the proof is collected
        const uint8_t *wtnsData = nullptr;
        bool canceled = false;
        uint64_t epoch = 0;                            // value of abortEpoch when the job was accepted
    };
    typedef std::shared_ptr<Job> JobPtr;

    std::mutex mtx;
    std::condition_variable cvIncoming, cvReady;
    std::map<std::string, Circuit> circuits;   // keyed by zkey file stem
    std::vector<int> workerDevices;            // GPU of every replica
    size_t queueCap = 0;
    bool stopping = false;
    uint64_t nextId = 1;

    // single-slot mode (the reference's state machine)
    Status status = ready;
    JobPtr pending, executing, last;

    // throughput mode
    std::deque<JobPtr> incoming, readyJobs;
    size_t inWitness = 0;                      // jobs inside a witness generator right now (count against queueCap)
    uint64_t abortEpoch = 0;                   // POST /cancel: jobs accepted before it that have not reached a GPU are dropped
    std::map<uint64_t, JobPtr> jobs;           // jobs by id: every unfinished one + the most recent `keepResults` finished ones
    size_t keepResults = 4096;                 // ZKHIP_KEEP_RESULTS (default: 4096, or four queues' worth)
    std::vector<std::thread> threads;

    void generateWitness(Job &job, const std::string &tag);   // throws; fills job.wtns / wtnsData / pubData
    void adoptWitness(Job &job, const ZKeyUtils::Header *zh); // the checks + public signals once job.wtns is open
    static std::string statusDocument(const Job &job);
    void remember(const JobPtr &job);          // caller holds mtx
    void checkPending();                       // caller holds mtx
    void runSingle(JobPtr job);
    void witnessLoop();
    void deviceLoop(size_t worker);
};
