// main_prover.cpp -- `prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>`
// Command line, exit status and stderr texts of the reference CLIs (src/main_prover.cpp:17-85 and
// src/main_prover_ultra_groth.cpp:17-85), on top of include/prover.h. Built twice: -DUG_ULTRA selects the UltraGroth calls.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include "host_util.hpp"
#include "../../include/prover.h"

namespace {

struct Api {                         // the three one-shot calls of one protocol
    int (*publicSize)(const void*, unsigned long long, unsigned long long*, char*, unsigned long long);
    void (*proofSize)(unsigned long long*);
    int (*prove)(const void*, unsigned long long, const void*, unsigned long long, char*, unsigned long long*, char*,
                 unsigned long long*, char*, unsigned long long);
};
#ifdef UG_ULTRA
const Api kApi = {ultra_groth_public_size_for_zkey_buf, ultra_groth_proof_size, ultra_groth_prover};
#else
const Api kApi = {groth16_public_size_for_zkey_buf, groth16_proof_size, groth16_prover};
#endif

// the library pads its text buffers with NULs (strncpy): a file gets the text up to the first one
void writeText(const char* path, const std::string& buffer) {
    FILE* f = fopen(path, "wb");
    if (!f) throw std::runtime_error(std::string("cannot write ") + path);
    fwrite(buffer.data(), 1, strnlen(buffer.data(), buffer.size()), f);
    fclose(f);
}

void run(char** argv) {
    ughost::FileMap zkey(argv[1]), wtns(argv[2]);
    char message[1024] = {0};
    unsigned long long publicBytes = 0, proofBytes = 0;
    if (kApi.publicSize(zkey.data(), zkey.size(), &publicBytes, message, sizeof(message) - 1) != PROVER_OK)
        throw std::runtime_error(message);
    kApi.proofSize(&proofBytes);
    std::string proof(proofBytes, '\0'), pub(publicBytes, '\0');
    if (kApi.prove(zkey.data(), zkey.size(), wtns.data(), wtns.size(), &proof[0], &proofBytes, &pub[0], &publicBytes, message,
                   sizeof(message) - 1) != PROVER_OK)
        throw std::runtime_error(message);
    writeText(argv[3], proof);
    writeText(argv[4], pub);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 5) {
        fputs("Invalid number of parameters\nUsage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>\n", stderr);
        return EXIT_FAILURE;
    }
    try {
        run(argv);
    } catch (const std::exception& e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
