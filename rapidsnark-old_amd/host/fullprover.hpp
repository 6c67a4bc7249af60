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

#include "binfile_utils.hpp"
#include "groth16.hpp"
#include "zkey_utils.hpp"

class FullProver {
    enum Status { aborted = -2, busy = -1, failed = 0, success = 1, ready = 6 };
    Status status = ready;
    std::mutex mtx;

    std::string pendingInput, executingInput, pendingCircuit, executingCircuit;
    std::map<std::string, std::unique_ptr<Groth16::Prover>> provers;
    std::map<std::string, std::unique_ptr<ZKeyUtils::Header>> zkHeaders;

    std::string proof;     // compact proof JSON
    std::string pubData;   // compact JSON array of decimal strings
    std::string errString;
    bool canceled = false;

    bool isCanceled();
    void calcFinished();
    void thread_calculateProve();
    void checkPending();     // caller holds mtx

public:
    FullProver(std::string zkeyFileNames[], int size);
    void startProve(std::string input, std::string circuit);
    void abort();
    std::string getStatus();   // the JSON document of GET /status
};
