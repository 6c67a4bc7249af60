#include "fullprover.hpp"

#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <sys/wait.h>
#include <thread>

#include "json_min.hpp"

static const uint8_t kAltBn128r[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                       0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

static bool env_scalar(const char *name, uint8_t out[32]) {
    const char *v = getenv(name);
    if (!v || strlen(v) != 64) return false;
    for (int i = 0; i < 32; i++) {
        unsigned x;
        if (sscanf(v + 2 * i, "%2x", &x) != 1) return false;
        out[i] = (uint8_t)x;
    }
    return true;
}

static std::string getfilename(std::string path) {   // file stem = circuit name (fullprover.cpp:14-19)
    path = path.substr(path.find_last_of("/\\") + 1);
    return path.substr(0, path.find_last_of('.'));
}

FullProver::FullProver(std::string zkeyFileNames[], int size) {
    for (int i = 0; i < size; i++) {
        std::string circuit = getfilename(zkeyFileNames[i]);
        auto zkey = BinFileUtils::openExisting(zkeyFileNames[i], "zkey", 1);
        auto hdr = ZKeyUtils::loadHeader(zkey.get());
        if (memcmp(hdr->rPrime.data(), kAltBn128r, 32) != 0) throw std::invalid_argument("zkey curve not supported");
        const uint64_t sizes[6] = {zkey->getSectionSize(4), zkey->getSectionSize(5), zkey->getSectionSize(6),
                                   zkey->getSectionSize(7), zkey->getSectionSize(8), zkey->getSectionSize(9)};
        circuits[circuit].prover = Groth16::makeProver(hdr->nVars, hdr->nPublic, hdr->domainSize, hdr->nCoefs, hdr->vk_alpha1, hdr->vk_beta1,
                                               hdr->vk_beta2, hdr->vk_delta1, hdr->vk_delta2, zkey->getSectionData(4),
                                               zkey->getSectionData(5), zkey->getSectionData(6), zkey->getSectionData(7),
                                               zkey->getSectionData(8), zkey->getSectionData(9), sizes, /*precompDefault=*/true);
        // libzkhip copied everything it needs to the GPU: only the scalar header fields are kept
        // (the vk pointers into the mapping die with `zkey` and are never used again here)
        hdr->vk_alpha1 = hdr->vk_beta1 = hdr->vk_beta2 = hdr->vk_gamma2 = hdr->vk_delta1 = hdr->vk_delta2 = nullptr;
        circuits[circuit].header = std::move(hdr);
        std::cerr << "circuit: " << circuit << '\n';
    }
    status = ready;
}

void FullProver::startProve(std::string input, std::string circuit) {
    std::lock_guard<std::mutex> guard(mtx);
    pending = Job{input, circuit};
    if (status == busy) canceled = true;   // reference: abort() here re-locks mtx and deadlocks (Q2)
    checkPending();
}

void FullProver::checkPending() {
    if (status == busy) return;
    if (pending.empty()) return;
    status = busy;
    executing = pending;
    pending.clear();
    errString.clear();
    canceled = false;
    proof = "null";
    std::thread th(&FullProver::thread_calculateProve, this);
    th.detach();
}

void FullProver::thread_calculateProve() {
    try {
        std::string input, circuit;
        {
            std::lock_guard<std::mutex> guard(mtx);
            input = executing.input;
            circuit = executing.circuit;
        }
        if (!JsonMin::isValid(input)) throw std::runtime_error("input is not valid JSON");
        auto pit = circuits.find(circuit);
        if (pit == circuits.end()) throw std::runtime_error("unknown circuit: " + circuit);

        // witness generation: the exact hand-off of fullprover.cpp:112-135 (same paths, same argv)
        {
            std::ofstream file("./build/input_" + circuit + ".json");
            file << input;
        }
        std::string witnessFile("./build/" + circuit + ".wtns");
        std::string command("./build/" + circuit + " ./build/input_" + circuit + ".json " + witnessFile);
        std::array<char, 128> buffer;
        std::string result;
        FILE *pipe = popen(command.c_str(), "r");
        if (!pipe) throw std::runtime_error("Couldn't start command.");
        while (fgets(buffer.data(), 128, pipe) != NULL) result += buffer.data();
        int returnCode = pclose(pipe);
        std::cout << result << std::endl;
        std::cout << returnCode << std::endl;
        if (returnCode != 0) throw std::runtime_error("witness generator failed with code " + std::to_string(WEXITSTATUS(returnCode)));

        auto wtns = BinFileUtils::openExisting(witnessFile, "wtns", 2);
        auto wtnsHeader = WtnsUtils::loadHeader(wtns.get());
        if (memcmp(wtnsHeader->prime.data(), kAltBn128r, 32) != 0) throw std::invalid_argument("different wtns curve");
        const ZKeyUtils::Header *zh = pit->second.header.get();
        if (wtnsHeader->nVars != zh->nVars || wtns->getSectionSize(2) < (uint64_t)zh->nVars * 32)
            throw std::invalid_argument("witness does not match the zkey (nVars)");
        const uint8_t *wtnsData = static_cast<const uint8_t *>(wtns->getSectionData(2));

        size_t n = zk_public_to_json(wtnsData, zh->nPublic, nullptr, 0);
        std::string pub(n + 1, '\0');
        zk_public_to_json(wtnsData, zh->nPublic, &pub[0], n + 1);
        pub.resize(n);

        // ZKHIP_FIXED_R / ZKHIP_FIXED_S (64 hex digits, LE): deterministic proofs for parity tests
        uint8_t r[32], s[32];
        bool fr = env_scalar("ZKHIP_FIXED_R", r), fs = env_scalar("ZKHIP_FIXED_S", s);
        std::string pr = "null";
        if (!isCanceled()) pr = pit->second.prover->prove(wtnsData, fr ? r : nullptr, fs ? s : nullptr)->toJson();   // HOT PATH (fullprover.cpp:155)
        {
            std::lock_guard<std::mutex> guard(mtx);
            pubData = pub;
            proof = pr;
        }
        calcFinished();
    } catch (std::exception &e) {   // reference catches runtime_error only: a JSON error kills it (Q3)
        if (!isCanceled()) {
            std::lock_guard<std::mutex> guard(mtx);
            errString = e.what();
        }
        calcFinished();
    }
}

void FullProver::calcFinished() {
    std::lock_guard<std::mutex> guard(mtx);
    if (canceled) status = aborted;
    else if (!errString.empty()) status = failed;
    else status = success;
    canceled = false;
    executing.clear();
    checkPending();
}

bool FullProver::isCanceled() {
    std::lock_guard<std::mutex> guard(mtx);
    return canceled;
}

void FullProver::abort() {
    std::lock_guard<std::mutex> guard(mtx);
    if (status != busy) return;
    canceled = true;
}

// Same documents as nlohmann's dump() of FullProver::getStatus (fullprover.cpp:216-240): keys in
// alphabetical order, compact; proof and pubData are STRINGS containing JSON.
std::string FullProver::getStatus() {
    std::lock_guard<std::mutex> guard(mtx);
    switch (status) {
        case ready: return "{\"status\":\"ready\"}";
        case aborted: return "{\"status\":\"aborted\"}";
        case failed: return "{\"error\":" + JsonMin::quote(errString) + ",\"status\":\"failed\"}";
        case success: return "{\"proof\":" + JsonMin::quote(proof) + ",\"pubData\":" + JsonMin::quote(pubData) + ",\"status\":\"success\"}";
        case busy: return "{\"status\":\"busy\"}";
    }
    return "{}";
}
