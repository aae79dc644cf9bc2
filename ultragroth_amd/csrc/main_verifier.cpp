// main_verifier.cpp -- `verifier <verification_key.json> <inputs.json> <proof.json>`
// Command line, stderr messages and exit codes of the reference CLIs (src/main_verifier.cpp:8-60 and
// src/main_verifier_ultra_groth.cpp), on top of include/verifier.h. Built twice: -DUG_ULTRA selects ultra_groth_verify.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>
#include "host_util.hpp"
#include "../../include/verifier.h"

namespace {

typedef int (*VerifyFn)(const char*, const char*, const char*, char*, unsigned long);
#ifdef UG_ULTRA
const VerifyFn kVerify = ultra_groth_verify;
#else
const VerifyFn kVerify = groth16_verify;
#endif

std::string slurp(const char* path) {
    ughost::FileMap m(path);
    return std::string(reinterpret_cast<const char*>(m.data()), m.size());
}

// what the reference prints for each outcome (all on stderr), and the exit status that goes with it
int report(int outcome, const char* detail) {
    switch (outcome) {
        case VERIFIER_VALID_PROOF:   fputs("Result: Valid proof\n", stderr);   return EXIT_SUCCESS;
        case VERIFIER_INVALID_PROOF: fputs("Result: Invalid proof\n", stderr); return EXIT_FAILURE;
        default:                     fprintf(stderr, "Error: %s\n", detail);   return EXIT_FAILURE;
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 4) {
        fputs("Invalid number of parameters:\nUsage: verifier <verification_key.json> <inputs.json> <proof.json>\n", stderr);
        return EXIT_FAILURE;
    }
    char detail[256] = {0};
    try {
        const std::string key = slurp(argv[1]), inputs = slurp(argv[2]), proof = slurp(argv[3]);
        return report(kVerify(proof.c_str(), inputs.c_str(), key.c_str(), detail, sizeof(detail) - 1), detail);
    } catch (const std::exception& e) {
        return report(VERIFIER_ERROR, e.what());
    }
}
