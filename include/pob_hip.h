/* pob_hip.h -- C ABI of the MI355X-native witness generator for the proof_of_burn / spend circuits.
 *
 * Drop-in boundary.  In the reference, the path "input.json -> witness -> .wtns" is the
 * circom-emitted calculator invoked as a process (reference Makefile:4-5:
 *     ./main_proof_of_burn input.json witness.wtns ; ./main_spend input.json witness.wtns
 * and tests/test.py:60-64 for gadget mains).  There is no in-process FFI in the reference; the functions
 * below are what a binding for that path would call: one handle per (GPU, circuit instantiation), inputs
 * as flat arrays in the declaration order of the circuit's `signal input`s, public outputs / pass-fail /
 * .wtns out.  Plain pointers and sizes only; every buffer is caller-allocated; no callbacks; a handle is
 * bound to one GPU and is not thread-safe (one process per GPU).  All functions return 0 on success or a
 * negative code; pob_strerror() gives the text.
 */
#ifndef POB_HIP_H
#define POB_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pob_ctx* pob_handle;

enum { POB_CIRCUIT_PROOF_OF_BURN = 0, POB_CIRCUIT_SPEND = 1 };
enum { POB_OK = 0, POB_E_ARG = -1, POB_E_HIP = -2, POB_E_NOMEM = -3, POB_E_STATE = -4, POB_E_IO = -5 };

typedef struct {
    uint64_t n_witness;          /* W: O0 wires incl. the constant-1 wire (= nWitness of the .wtns)            */
    uint64_t n_bit, n_sm, n_fr;  /* wires per storage class (policy.hpp)                                        */
    uint32_t n_fr_inputs, n_sm_inputs, n_outputs;
    uint32_t n_units, n_sponges, n_perms, n_stages, max_batch;
    uint64_t group_bytes;        /* HBM-resident bytes of the compact witness vector per 64 witnesses           */
    uint64_t keccak_bit_wires;   /* wires handled by the bit-sliced Keccak kernels                              */
    uint64_t n_sb;               /* wires of the int8 class (operands of the Keccak output selectors' IsEqual gadgets) */
} pob_info_t;

/* Replaces `component main = ProofOfBurn(...)` / `Spend(...)` + circom -c + make (reference
 * circuits/main_proof_of_burn.circom:27, circuits/main_spend.circom:6, Makefile:2-3).
 * params: template parameters as canonical 4x64-bit LE limbs each:
 *   ProofOfBurn: maxNumLayers, maxNodeBlocks, maxHeaderBlocks, minLeafAddressNibbles, amountBytes,
 *                powMinimumZeroBytes, maxIntendedBalance, maxActualBalance   (proof_of_burn.circom:34)
 *   Spend:       maxAmountBytes                                              (spend.circom:32)        */
int pob_open(int device, int circuit, const uint64_t* params, int nparams, uint32_t max_batch, pob_handle* out);
void pob_close(pob_handle h);
int pob_get_info(pob_handle h, pob_info_t* info);
/* Layout planner only (no GPU touched): wire / class counts of an instantiation, e.g. to size buffers. */
int pob_plan_info(int circuit, const uint64_t* params, int nparams, pob_info_t* info);
const char* pob_strerror(pob_handle h);

/* Replaces the emitted loader (loadJson; reference tests/test.py:57-59 writes input.json).  Witness-major:
 *   fr_inputs[n][n_fr_inputs][32]  canonical LE field elements, circuit declaration order of the FR-class inputs
 *     ProofOfBurn: burnKey, actualBalance, intendedBalance, revealAmount, burnExtraCommitment, _proofExtraCommitment
 *     Spend:       burnKey, balance, withdrawnBalance, extraCommitment
 *   sm_inputs[n][n_sm_inputs]      int32, declaration order of the small inputs
 *     ProofOfBurn: numLeafAddressNibbles, layers[L][136*NB], layerLens[L], numLayers, blockHeader[136*HB],
 *                  blockHeaderLen, byteSecurityRelax      (values >= 2^31 must be rejected by the caller:
 *                  every one of them is range-constrained to <= 16 bits in-circuit)
 * Host pointers; the copy to HBM is synchronous.                                                                */
int pob_upload_inputs(pob_handle h, const uint8_t* fr_inputs, const int32_t* sm_inputs, uint32_t n);

/* Replaces the run of the calculator (all `<==` / `<--` / `===`; reference Makefile:4-5): enqueues every stage on
 * `stream` (a hipStream_t, NULL = the handle's own stream) for the n uploaded inputs.  Asynchronous.              */
int pob_generate(pob_handle h, void* stream);
/* Per-gate constraint evaluator over the resident witness vector (north star; the reference checks inline,
 * e.g. assert.circom:46,62,78, divide.circom:32): re-reads every wire, asynchronous.                             */
int pob_constraint_check(pob_handle h, void* stream);
int pob_sync(pob_handle h);
/* Two-batch pipeline (no counterpart in the reference, whose calculator runs one witness per process, Makefile:4-5): links two
 * handles of one device that work on consecutive batches.  While linked, a handle's pob_generate starts once the partner's
 * generation is complete -- so its latency-bound stages run beside the partner's (read-streaming) evaluation -- and its Keccak
 * round expansion, the HBM-write-saturating kernel, follows the end of that evaluation; pob_constraint_check then runs on streams
 * the generation does not use.  Call order per batch: pob_constraint_check(previous), pob_generate(next).  The link is
 * symmetric (one call links both handles and drops their previous links); partner = NULL unlinks; pob_close unlinks. */
int pob_set_partner(pob_handle h, pob_handle partner);

/* Replaces "stderr non-empty => failure" + the output dump patched in by tests/test.py:36-54.
 * status[i] = 0 ok, else (template id << 12 | source line) of the first failing assert; outputs[i][32] = public
 * output (commitment) canonical LE.  check_status / bad_wire (may be NULL): result of pob_constraint_check:
 * first failing === site and lowest wire whose stored value contradicts its definition (0xFFFFFFFF = none).      */
int pob_results(pob_handle h, uint32_t* status, uint8_t* outputs, uint32_t* check_status, uint32_t* bad_wire);
/* Device-resident results for the RCCL gather: status u32[max_batch_padded], outputs u8[max_batch_padded][32]. */
int pob_results_device(pob_handle h, void** d_status, void** d_outputs);
/* The same results as ONE packed record per witness, {u32 status, u8 commitment[32]} = 36 bytes, device-resident
 * (u8[max_batch_padded][36]): what a rank contributes to the single all-gather of the multi-GPU path.          */
#define POB_RECORD_BYTES 36
int pob_results_records_device(pob_handle h, void** d_records);

/* Replaces writeBinWitness (patch point `fclose(write_ptr)` at reference tests/test.py:36): expands witness `idx`
 * of the batch to canonical 32-byte LE values.  pob_emit_witness: payload only (32*W bytes) into host memory;
 * pob_write_wtns: full iden3 .wtns file.                                                                         */
int pob_emit_witness(pob_handle h, uint32_t idx, uint8_t* dst, uint64_t cap);
int pob_write_wtns(pob_handle h, uint32_t idx, const char* path);
/* The streaming form underneath both: the canonical payload is expanded from the compact resident vector in WINDOWS of
 * `window_wires` wires (0 = 8 Mi wires = 256 MiB), double-buffered on the device and in pinned host memory, so that expanding window
 * k+1, copying it D2H and the caller's consumption of window k overlap, and a whole 32*W-byte device buffer never exists.
 * pob_emit_next hands out the next window (pinned host memory owned by the handle, valid until the following call) with its first
 * wire and wire count; n_wires = 0 ends the witness.  The buffers are kept for the next witness.  A witness whose status is non-zero
 * is refused (POB_E_STATE): like the reference binary, a failed input produces no witness (reference tests/test.py:65-68).         */
int pob_emit_begin(pob_handle h, uint32_t idx, uint64_t window_wires);
int pob_emit_next(pob_handle h, const uint8_t** data, uint64_t* first_wire, uint64_t* n_wires);
/* Measurement: `count` witnesses from `first_idx` on, back to back through the window pipeline into pinned host memory.          */
int pob_emit_measure(pob_handle h, uint32_t first_idx, uint32_t count, uint64_t window_wires, double* seconds, uint64_t* bytes);

/* Measurement: average duration (ms, HIP events on `stream`) of `iters` back-to-back launches of one kernel over
 * the current batch.  which: 0 = Keccak round expansion (generate), 1 = Keccak round constraint evaluation,
 * 2 = G-unit constraint evaluation (every family, back to back), 3 = sponge chain (generate); 100 + k / 200 + k = evaluation /
 * generation of all units of kind k (circuits.hpp UnitKind) alone on the device; 300 + f = the evaluation kernel of family f
 * (circuits.hpp Fam) alone (tools/unit_times.py).                                                                */
int pob_time_kernel(pob_handle h, int which, int iters, void* stream, float* avg_ms);

/* Test hook: XOR `mask` into the stored word of BIT-class storage index `bit_index` of witness group `group`. */
int pob_debug_xor_bits(pob_handle h, uint32_t group, uint64_t bit_index, uint64_t mask);
/* Test hook for the constraint evaluator: corrupt ONE stored value of ONE witness (lane `lane` of group `group`) of storage class
 * `cls` at storage index `index` (the wire's rank within its class): BIT: flips the bit if xor_mask & 1; SM: int32 ^= xor_mask;
 * FR: 32-bit limb `sub` (Montgomery form) ^= xor_mask; SB: int8 ^= xor_mask.  IsZero.inv wires live in the SM / SB slabs as their
 * operand code, so poking those indices pokes the hint.                                                                          */
enum { POB_CLASS_BIT = 0, POB_CLASS_SM = 1, POB_CLASS_FR = 2, POB_CLASS_SB = 3 };
int pob_debug_poke(pob_handle h, int cls, uint32_t group, uint64_t index, uint32_t sub, uint32_t lane, uint32_t xor_mask);
/* Test hook: storage class, rank within the class and wire index of a few named wires: "commitment"; "poseidon" (k-th wire of the
 * first Poseidon block); "pad.div.out" / "pad.div.rem" / "pad.iseq.inv" of KeccakBytes instance k (the Divide hint of
 * divide.circom:23-24 and an IsZero.inv hint); ProofOfBurn only: "sc.M" / "sc.exists" / "sc.isz.inv" [k] of layer 1's SubstringCheck. */
int pob_debug_ref(pob_handle h, const char* name, uint32_t k, int* cls, uint64_t* index, uint64_t* wire);

/* Host helper used by the input producers (next row f1): Keccak-256 of a byte string.                           */
void pob_keccak256(const uint8_t* msg, uint64_t len, uint8_t out[32]);
/* Input producer's proof-of-work (reference tests/main.py:47-56): first key >= start_key (256-bit big-endian counter)
 * with keccak256(key | postfix)[0:zero_bytes] == 0; returns the number of increments or -1.                      */
int64_t pob_pow_search(const uint8_t start_key[32], const uint8_t* postfix, uint32_t postfix_len, uint32_t zero_bytes,
                       uint64_t max_tries, uint8_t out_key[32]);
/* The same search as a HIP kernel on `device` (one candidate key per thread, Keccak-f on 64-bit lanes, windows of 2^24 keys,
 * smallest offset of the first window with a hit): identical result, -2 on a HIP error.                          */
int64_t pob_pow_search_gpu(int device, const uint8_t start_key[32], const uint8_t* postfix, uint32_t postfix_len,
                           uint32_t zero_bytes, uint64_t max_tries, uint8_t out_key[32]);

#ifdef __cplusplus
}
#endif
#endif
