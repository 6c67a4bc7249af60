// prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>
// Drop-in for the reference CLI (src/main_prover.cpp:23-103): same argv, same messages, same
// exit codes (-1 on usage / any error, 0 on success), same output bytes (compact JSON, no
// trailing newline).  The hot path runs on the MI355X through libzkhip.so.
// Extras (env, so argv stays identical): ZKHIP_FIXED_R / ZKHIP_FIXED_S = 64 hex digits
// (32-byte little-endian scalars) make the proof deterministic for parity tests.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>

#include "binfile_utils.hpp"
#include "groth16.hpp"
#include "wtns_utils.hpp"
#include "zkey_utils.hpp"

// BN254 scalar field order, little-endian (src/main_prover.cpp:34)
static const uint8_t kAltBn128r[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                       0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};

static bool env_scalar(const char *name, uint8_t out[32]) {
    const char *v = getenv(name);
    if (!v) return false;
    if (strlen(v) != 64) throw std::invalid_argument(std::string(name) + " must be 64 hex digits");
    for (int i = 0; i < 32; i++) {
        unsigned x;
        if (sscanf(v + 2 * i, "%2x", &x) != 1) throw std::invalid_argument(std::string(name) + " is not hex");
        out[i] = (uint8_t)x;
    }
    return true;
}

int main(int argc, char **argv) {
    if (argc != 5) {
        std::cerr << "Invalid number of parameters:\n";
        std::cerr << "Usage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>\n";
        return -1;
    }
    try {
        std::string zkeyFilename = argv[1];
        std::string wtnsFilename = argv[2];
        std::string proofFilename = argv[3];
        std::string publicFilename = argv[4];

        auto zkey = BinFileUtils::openExisting(zkeyFilename, "zkey", 1);
        auto zkeyHeader = ZKeyUtils::loadHeader(zkey.get());
        if (memcmp(zkeyHeader->rPrime.data(), kAltBn128r, 32) != 0) throw std::invalid_argument("zkey curve not supported");

        auto wtns = BinFileUtils::openExisting(wtnsFilename, "wtns", 2);
        auto wtnsHeader = WtnsUtils::loadHeader(wtns.get());
        if (memcmp(wtnsHeader->prime.data(), kAltBn128r, 32) != 0) throw std::invalid_argument("different wtns curve");
        // quirk Q8: the reference reads out of bounds on a size mismatch; here it is an error
        if (wtnsHeader->nVars != zkeyHeader->nVars || wtns->getSectionSize(2) < (uint64_t)zkeyHeader->nVars * 32)
            throw std::invalid_argument("witness does not match the zkey (nVars)");

        const uint64_t sizes[6] = {zkey->getSectionSize(4), zkey->getSectionSize(5), zkey->getSectionSize(6),
                                   zkey->getSectionSize(7), zkey->getSectionSize(8), zkey->getSectionSize(9)};
        auto prover = Groth16::makeProver(zkeyHeader->nVars, zkeyHeader->nPublic, zkeyHeader->domainSize, zkeyHeader->nCoefs,
                                          zkeyHeader->vk_alpha1, zkeyHeader->vk_beta1, zkeyHeader->vk_beta2,
                                          zkeyHeader->vk_delta1, zkeyHeader->vk_delta2,
                                          zkey->getSectionData(4),    // Coefs
                                          zkey->getSectionData(5),    // pointsA
                                          zkey->getSectionData(6),    // pointsB1
                                          zkey->getSectionData(7),    // pointsB2
                                          zkey->getSectionData(8),    // pointsC
                                          zkey->getSectionData(9),    // pointsH1
                                          sizes);
        const uint8_t *wtnsData = static_cast<const uint8_t *>(wtns->getSectionData(2));
        uint8_t r[32], s[32];
        bool fr = env_scalar("ZKHIP_FIXED_R", r), fs = env_scalar("ZKHIP_FIXED_S", s);
        auto proof = prover->prove(wtnsData, fr ? r : nullptr, fs ? s : nullptr);

        std::ofstream proofFile(proofFilename);
        proofFile << proof->toJson();
        proofFile.close();

        size_t n = zk_public_to_json(wtnsData, zkeyHeader->nPublic, nullptr, 0);
        std::string pub(n + 1, '\0');
        zk_public_to_json(wtnsData, zkeyHeader->nPublic, &pub[0], n + 1);
        pub.resize(n);
        std::ofstream publicFile(publicFilename);
        publicFile << pub;
        publicFile.close();
    } catch (std::exception &e) {
        std::cerr << e.what() << '\n';
        return -1;
    }
    exit(EXIT_SUCCESS);
}
