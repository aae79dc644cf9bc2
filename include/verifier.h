/* verifier.h -- proof verification with the contract of the reference's src/verifier.h:1-46: the same two entry points,
 * argument order and result codes. Implemented as host code inside libultragroth_hip.so
 * (ultragroth_amd/csrc/verifier_api.cpp). */
#ifndef ULTRAGROTH_AMD_VERIFIER_H
#define ULTRAGROTH_AMD_VERIFIER_H

#ifdef __cplusplus
extern "C" {
#endif

/* result codes (src/verifier.h:9-11) */
enum {
    VERIFIER_VALID_PROOF = 0,      /* the pairing equation holds                           */
    VERIFIER_INVALID_PROOF = 1,    /* it does not (or a point is not on its curve)         */
    VERIFIER_ERROR = 2             /* malformed input; error_msg says which part           */
};

/* All three texts are NUL-terminated JSON in the snarkjs layout: proof.json, public.json, verification_key.json.
 * Messages on VERIFIER_ERROR: "invalid proof data", "invalid inputs data", "invalid verification key data",
 * "len(inputs)+1 != len(vk.IC)" (src/verifier.cpp:16-146, src/groth16.cpp:318-320).
 * Differences from the reference: points that are not on their curve give VERIFIER_INVALID_PROOF (the reference
 * evaluates the pairing on them regardless), and ultra_groth_verify does not print "inputs.size(): N" on stdout
 * (src/ultra_groth.cpp:591). error_msg may be NULL. */
int groth16_verify(const char *proof, const char *inputs, const char *verification_key, char *error_msg, unsigned long error_msg_maxsize);

/* UltraGroth: proof fields pi_a, pi_b, pi_f, pi_r with protocol "ultragroth"; key fields vk_delta_c1_2, vk_delta_c2_2 and
 * IC_rand; the length error reads "len(inputs) != len(vk.IC)" (src/ultra_groth.cpp:585-587). */
int ultra_groth_verify(const char *proof, const char *inputs, const char *verification_key, char *error_msg, unsigned long error_msg_maxsize);

#ifdef __cplusplus
}
#endif
#endif
