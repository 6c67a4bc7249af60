// FullProver — single-slot job orchestration of the reference server
// (src/fullprover.hpp:13-50, src/fullprover.cpp:21-240): one prover per zkey keyed by file stem,
// pending/executing state machine, witness generation by an external circom binary, prove on
// the MI355X through libzkhip.  Same public surface: startProve / abort / getStatus.
// Deliberate deviations from reference bugs: Q2 no self-deadlock when a request
// arrives while busy; Q3 a malformed body fails the job instead of killing the process;
// Q5 getStatus takes the lock; Q11 a failing witness generator fails the job; an unknown
// circuit name fails the job (the reference dereferences a null map entry).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "groth16.hpp"
#include "zkfile.hpp"

class FullProver {
public:
    FullProver(std::string zkeyFileNames[], int size);
    void startProve(std::string input, std::string circuit);   // POST /input/:circuit
    void abort();                                              // POST /cancel
    std::string getStatus();                                   // the JSON document of GET /status

private:
    enum Status { aborted = -2, busy = -1, failed = 0, success = 1, ready = 6 };

    struct Circuit {
        std::unique_ptr<Groth16::Prover> prover;
        std::unique_ptr<ZKeyUtils::Header> header;   // scalar fields only (vk pointers are cleared after create)
    };
    struct Job {
        std::string input, circuit;
        bool empty() const { return input.empty() || circuit.empty(); }
        void clear() { input.clear(); circuit.clear(); }
    };

    std::mutex mtx;
    Status status = ready;
    std::map<std::string, Circuit> circuits;   // keyed by zkey file stem
    Job pending, executing;                    // one waiting slot: the latest request wins
    std::string proof;                         // compact proof JSON of the last successful job
    std::string pubData;                       // compact JSON array of decimal strings
    std::string errString;
    bool canceled = false;

    bool isCanceled();
    void calcFinished();
    void thread_calculateProve();
    void checkPending();   // caller holds mtx
};
