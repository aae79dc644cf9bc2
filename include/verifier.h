/* verifier.h -- proof verification, same contract as the reference's src/verifier.h:1-46 (names, argument order,
 * return codes). Implemented in ultragroth_amd/csrc/verifier_api.cpp (host code inside libultragroth_hip.so). */
#ifndef ULTRAGROTH_AMD_VERIFIER_H
#define ULTRAGROTH_AMD_VERIFIER_H

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes returned by the functions (src/verifier.h:9-11). */
#define VERIFIER_VALID_PROOF        0x0
#define VERIFIER_INVALID_PROOF      0x1
#define VERIFIER_ERROR              0x2

/* 'proof', 'inputs' and 'verification_key' are null-terminated json strings in the snarkjs layout
 * (proof.json, public.json, verification_key.json).
 * Returns VERIFIER_VALID_PROOF, VERIFIER_INVALID_PROOF, or VERIFIER_ERROR with a message in error_msg:
 * "invalid proof data", "invalid inputs data", "invalid verification key data", "len(inputs)+1 != len(vk.IC)"
 * (src/verifier.cpp:16-146, src/groth16.cpp:318-320). Differences from the reference: points that are not on their
 * curve give VERIFIER_INVALID_PROOF (the reference evaluates the pairing on them regardless), and
 * ultra_groth_verify does not print "inputs.size(): N" on stdout (src/ultra_groth.cpp:591). */
int groth16_verify(const char *proof, const char *inputs, const char *verification_key,
                   char *error_msg, unsigned long error_msg_maxsize);

/* UltraGroth (protocol "ultragroth": pi_a, pi_b, pi_f, pi_r; key fields vk_delta_c1_2, vk_delta_c2_2, IC_rand);
 * length error text "len(inputs) != len(vk.IC)" (src/ultra_groth.cpp:585-587). */
int ultra_groth_verify(const char *proof, const char *inputs, const char *verification_key,
                       char *error_msg, unsigned long error_msg_maxsize);

#ifdef __cplusplus
}
#endif

#endif
