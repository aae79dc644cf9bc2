// main_verifier.cpp -- `verifier <verification_key.json> <inputs.json> <proof.json>`
// Same command line, messages and exit codes as the reference CLIs (src/main_verifier.cpp:8-60,
// src/main_verifier_ultra_groth.cpp), on top of include/verifier.h. Built twice: -DUG_ULTRA selects ultra_groth_verify.
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include "host_util.hpp"
#include "../../include/verifier.h"

#ifdef UG_ULTRA
#define VERIFY ultra_groth_verify
#else
#define VERIFY groth16_verify
#endif

static std::string fileAsString(const char* path) {
    ughost::FileMap m(path);
    return std::string(reinterpret_cast<const char*>(m.data()), m.size());
}

int main(int argc, char** argv) {
    if (argc != 4) {
        std::cerr << "Invalid number of parameters:\n";
        std::cerr << "Usage: verifier <verification_key.json> <inputs.json> <proof.json>\n";
        return EXIT_FAILURE;
    }
    try {
        const std::string proof = fileAsString(argv[3]);
        const std::string inputs = fileAsString(argv[2]);
        const std::string key = fileAsString(argv[1]);
        char errorMessage[256] = {0};
        const int error = VERIFY(proof.c_str(), inputs.c_str(), key.c_str(), errorMessage, sizeof(errorMessage) - 1);
        if (error == VERIFIER_VALID_PROOF) {
            std::cerr << "Result: Valid proof" << std::endl;
            return EXIT_SUCCESS;
        } else if (error == VERIFIER_INVALID_PROOF) {
            std::cerr << "Result: Invalid proof" << std::endl;
            return EXIT_FAILURE;
        } else {
            std::cerr << "Error: " << errorMessage << '\n';
            return EXIT_FAILURE;
        }
    } catch (std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_FAILURE;
}
