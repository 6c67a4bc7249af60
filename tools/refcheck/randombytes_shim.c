/* randombytes_shim.c — parity kit, NOT part of the product.
 *
 * The reference's only entropy source is libsodium's randombytes_buf(), called exactly twice per
 * proof: 31 bytes for r, then 31 bytes for s (reference src/groth16.cpp:216-217).  Preloading this
 * shim into a real `rapidsnark` prover binary (dynamically linked against libsodium, as
 * tasksfile.js:58,83 builds it: `-lsodium`) makes its proof.json a pure function of
 * (zkey, wtns, r, s), so it can be compared byte for byte with tests/golden/<circuit>/proof.json —
 * the cross-check SURVEY.md §8(c) describes and which cannot be run in the build container
 * (the reference's arithmetic submodule `depends/ffiasm` is absent there).
 *
 *   gcc -shared -fPIC -O2 -o librandshim.so randombytes_shim.c
 *   ZKREF_R=<64 hex chars, little-endian> ZKREF_S=<64 hex chars> LD_PRELOAD=./librandshim.so \
 *       prover circuit.zkey witness.wtns proof.json public.json
 *
 * Call 1 is served from ZKREF_R, call 2 from ZKREF_S, call 3 from ZKREF_R again, ... (a server
 * proving several times stays deterministic).  Only the first `size` (= 31) bytes are used, exactly
 * as the reference fills them; the golden (r, s) values all fit 31 bytes.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

void randombytes_buf(void *const buf, const size_t size) {
    static unsigned calls = 0;
    const char *name = (calls++ & 1u) ? "ZKREF_S" : "ZKREF_R";
    const char *hex = getenv(name);
    unsigned char *out = (unsigned char *)buf;
    if (!hex || strlen(hex) < 2 * size) {
        fprintf(stderr, "randombytes_shim: %s must hold at least %zu hex bytes\n", name, size);
        abort();
    }
    for (size_t i = 0; i < size; i++) {
        int hi = hexval(hex[2 * i]), lo = hexval(hex[2 * i + 1]);
        if (hi < 0 || lo < 0) {
            fprintf(stderr, "randombytes_shim: %s is not hex\n", name);
            abort();
        }
        out[i] = (unsigned char)(hi * 16 + lo);
    }
}

/* libsodium entry points a binary may also reference; harmless no-ops here */
int sodium_init(void) { return 0; }
