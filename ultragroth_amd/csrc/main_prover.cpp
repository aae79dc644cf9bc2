// main_prover.cpp -- `prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>`
// Same command line, exit codes and messages as the reference CLIs (src/main_prover.cpp:17-85,
// src/main_prover_ultra_groth.cpp:17-85), on top of include/prover.h. Built twice:
// -DUG_ULTRA selects the UltraGroth entry points.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>
#include "host_util.hpp"
#include "../../include/prover.h"

#ifdef UG_ULTRA
#define PUBLIC_SIZE ultra_groth_public_size_for_zkey_buf
#define PROOF_SIZE ultra_groth_proof_size
#define PROVE ultra_groth_prover
#else
#define PUBLIC_SIZE groth16_public_size_for_zkey_buf
#define PROOF_SIZE groth16_proof_size
#define PROVE groth16_prover
#endif

// the buffers are NUL-padded by strncpy: the file is the text before the first NUL
static void truncateAtNul(std::vector<char>& v) {
    for (size_t i = 0; i < v.size(); i++)
        if (v[i] == 0) { v.resize(i); break; }
}

int main(int argc, char** argv) {
    if (argc != 5) {
        std::cerr << "Invalid number of parameters" << std::endl;
        std::cerr << "Usage: prover <circuit.zkey> <witness.wtns> <proof.json> <public.json>" << std::endl;
        return EXIT_FAILURE;
    }
    try {
        ughost::FileMap zkey(argv[1]);
        ughost::FileMap wtns(argv[2]);
        unsigned long long publicSize = 0, proofSize = 0;
        char errorMsg[1024] = {0};
        if (PUBLIC_SIZE(zkey.data(), zkey.size(), &publicSize, errorMsg, sizeof(errorMsg) - 1) != PROVER_OK)
            throw std::runtime_error(errorMsg);
        PROOF_SIZE(&proofSize);
        std::vector<char> publicBuffer(publicSize), proofBuffer(proofSize);
        if (PROVE(zkey.data(), zkey.size(), wtns.data(), wtns.size(), proofBuffer.data(), &proofSize, publicBuffer.data(),
                  &publicSize, errorMsg, sizeof(errorMsg) - 1) != PROVER_OK)
            throw std::runtime_error(errorMsg);
        truncateAtNul(proofBuffer);
        truncateAtNul(publicBuffer);
        std::ofstream(argv[3]).write(proofBuffer.data(), (std::streamsize)proofBuffer.size());
        std::ofstream(argv[4]).write(publicBuffer.data(), (std::streamsize)publicBuffer.size());
    } catch (std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
