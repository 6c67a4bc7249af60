// prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>
//
// Drop-in for the reference CLI (src/main_prover.cpp:23-103): same argv, same messages, same
// exit codes (-1 on usage / any error, 0 on success), same output bytes (compact JSON, no
// trailing newline).  The hot path runs on the MI355X through libzkhip.so.
// Extras via the environment, so argv stays identical: ZKHIP_FIXED_R / ZKHIP_FIXED_S = 64 hex
// digits (32-byte little-endian scalars) make the proof deterministic for parity tests.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <fstream>
#include <iostream>
#include <optional>
#include <stdexcept>
#include <string>
#include <unistd.h>

#include "groth16.hpp"
#include "zkfile.hpp"

namespace {

// BN254 scalar field order r, little-endian (the reference compares against the decimal
// string 21888242871839275222246405745257275088548364400416034343698204186575808495617)
constexpr uint8_t kBn254R[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                 0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

bool is_bn254_r(const std::array<uint8_t, 32> &p) { return memcmp(p.data(), kBn254R, sizeof kBn254R) == 0; }

struct Scalar32 {
    uint8_t b[32];
};

std::optional<Scalar32> scalar_from_env(const char *name) {
    const char *hex = getenv(name);
    if (!hex) return std::nullopt;
    if (strlen(hex) != 64) throw std::invalid_argument(std::string(name) + " must be 64 hex digits");
    Scalar32 s;
    for (int i = 0; i < 32; i++) {
        unsigned byte;
        if (sscanf(hex + 2 * i, "%2x", &byte) != 1) throw std::invalid_argument(std::string(name) + " is not hex");
        s.b[i] = static_cast<uint8_t>(byte);
    }
    return s;
}

void write_text(const std::string &path, const std::string &text) {
    std::ofstream out(path);
    out << text;
    out.close();                 // before the state is read: a full disk shows up at the flush
    if (!out) throw std::runtime_error("could not write " + path);      // (the program leaves through _exit below: nothing later would notice)
}

std::string public_signals_json(const uint8_t *witness, uint32_t nPublic) {
    std::string s(zk_public_to_json(witness, nPublic, nullptr, 0), '\0');
    zk_public_to_json(witness, nPublic, s.data(), s.size() + 1);
    return s;
}

// ZKHIP_VERBOSE=1: phase times on stderr (with ZKHIP_T0 = the launcher's time.time() also the time from exec to main
// and the wall-clock stamp of the last lap: what tools/cli_timing.py needs to price process start-up and exit)
double since_t0() {
    const char *t0 = getenv("ZKHIP_T0");
    if (!t0) return -1.0;
    const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    return (now - atof(t0)) * 1e3;
}
struct Lap {
    bool on = getenv("ZKHIP_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    Lap() {
        if (on && since_t0() >= 0) std::cerr << "[prover] spawn to main: " << since_t0() << " ms\n";
    }
    void operator()(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        std::cerr << "[prover] " << what << ": " << std::chrono::duration<double, std::milli>(now - t).count() << " ms\n";
        t = now;
    }
};

int run(const std::string &zkeyPath, const std::string &wtnsPath, const std::string &proofPath, const std::string &publicPath) {
    Lap lap;
    auto zkey = BinFileUtils::openExisting(zkeyPath, "zkey", 1);
    auto zh = ZKeyUtils::loadHeader(zkey.get());
    if (!is_bn254_r(zh->rPrime)) throw std::invalid_argument("zkey curve not supported");

    auto wtns = BinFileUtils::openExisting(wtnsPath, "wtns", 2);
    auto wh = WtnsUtils::loadHeader(wtns.get());
    if (!is_bn254_r(wh->prime)) throw std::invalid_argument("different wtns curve");
    // the reference indexes the witness blindly (out-of-bounds read on a mismatch, quirk Q8)
    if (wh->nVars != zh->nVars || wtns->getSectionSize(2) < uint64_t(zh->nVars) * 32)
        throw std::invalid_argument("witness does not match the zkey (nVars)");

    lap("open zkey + wtns");
    uint64_t bytes[6];
    for (uint32_t sec = 4; sec <= 9; sec++) bytes[sec - 4] = zkey->getSectionSize(sec);
    auto prover = Groth16::makeProver(zh->nVars, zh->nPublic, zh->domainSize, zh->nCoefs, zh->vk_alpha1, zh->vk_beta1, zh->vk_beta2,
                                      zh->vk_delta1, zh->vk_delta2,
                                      zkey->getSectionData(4),   // coefficient records
                                      zkey->getSectionData(5),   // A
                                      zkey->getSectionData(6),   // B1
                                      zkey->getSectionData(7),   // B2
                                      zkey->getSectionData(8),   // C
                                      zkey->getSectionData(9),   // H
                                      bytes);

    lap("makeProver");
    const auto *witness = static_cast<const uint8_t *>(wtns->getSectionData(2));
    const auto r = scalar_from_env("ZKHIP_FIXED_R"), s = scalar_from_env("ZKHIP_FIXED_S");
    auto proof = prover->prove(witness, r ? r->b : nullptr, s ? s->b : nullptr);

    lap("prove");
    write_text(proofPath, proof->toJson());
    write_text(publicPath, public_signals_json(witness, zh->nPublic));
    lap("write json");
    if (lap.on && since_t0() >= 0) std::cerr << "[prover] spawn to proof on disk: " << since_t0() << " ms\n";
    // One-shot process: both files are written and closed.  Destroying the prover (47-60 ms at 2^22) and the HIP runtime's
    // exit handlers only hand back memory and queues that the kernel driver reclaims with the process anyway: median wall
    // of the whole program 0.41 -> 0.27 s at 2^16, 0.68 -> 0.64 s at 2^22 (tools/cli_exit_ab.py, same box).
    // ZKHIP_CLEAN_EXIT=1 tears everything down in order instead (leak checkers).
    if (!getenv("ZKHIP_CLEAN_EXIT")) {
        std::cerr.flush();
        fflush(nullptr);
        _exit(EXIT_SUCCESS);
    }
    prover.reset();
    lap("destroy prover");
    return 0;
}

}   // namespace

int main(int argc, char **argv) {
    if (argc != 5) {
        std::cerr << "Invalid number of parameters:\n";
        std::cerr << "Usage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>\n";
        return -1;
    }
    try {
        run(argv[1], argv[2], argv[3], argv[4]);
    } catch (std::exception &e) {
        std::cerr << e.what() << '\n';
        return -1;
    }
    exit(EXIT_SUCCESS);
}
