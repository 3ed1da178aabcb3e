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

enum { POB_CIRCUIT_PROOF_OF_BURN = 0, POB_CIRCUIT_SPEND = 1, POB_CIRCUIT_GADGET = 2 };
enum { POB_OK = 0, POB_E_ARG = -1, POB_E_HIP = -2, POB_E_NOMEM = -3, POB_E_STATE = -4, POB_E_IO = -5, POB_E_RANGE = -6 };

typedef struct {
    uint64_t n_witness;          /* W: O0 wires incl. the constant-1 wire (= nWitness of the .wtns)            */
    uint64_t n_bit, n_sm, n_fr;  /* STORED wires per storage class (policy.hpp); n_bit + n_sm + n_fr + n_derived + n_alias + 1 = n_witness */
    uint32_t n_fr_inputs, n_sm_inputs, n_outputs;
    uint32_t n_units, n_sponges, n_perms, n_stages, max_batch;
    uint64_t group_bytes;        /* HBM-resident bytes of the compact witness vector per 64 witnesses           */
    uint64_t keccak_bit_wires;   /* wires handled by the bit-sliced Keccak kernels                              */
    uint64_t n_derived;          /* wires that are not stored because they are FUNCTIONS of stored wires / inputs / constants inside their unit and
                                    no other unit reads them (policy.hpp DV): the operand wires in[0], in[1], IsZero.in, IsZero.inv of every IsZero /
                                    IsEqual (over small operands and, in SubstringCheck, over field elements), copies (Selector.vals[],
                                    SelectorArray1D.arrays / arraysT, Pad's and AssertByteString's byte copies, SubstringCheck.mainInput[]) and running
                                    sums (Selector.sum[], SubstringCheck.M[]); round 4: BIT-valued copies of stored bits (the padded bytes' bits, the
                                    hash bits, the Keccak output selectors' vals[] / isEq[] / sum[] and their children, AssertByteString's bits) and the
                                    products of ShiftRight / ShiftLeft / CompConstant.  Generation and evaluation skip them; the emitter rebuilds them from
                                    the same expressions; pob_emit_selfcheck evaluates their relations on the values it writes                         */
    uint64_t n_alias;            /* wires of the Keccak round blocks that are not stored because they ARE another wire: copies of a stored gate
                                    output / of the round's input or output state, possibly at a rotated bit position (ShL / ShR / RhoPi) or negated
                                    (NotArray), or constants (shifted-out positions, round constants).  76 of the 1 604 arrays of a KeccakfRound
                                    block are stored (keccak_kernels.hpp); the emitter expands the others through one table                          */
    uint32_t kchk_rounds;        /* consecutive rounds of a permutation that ONE wavefront of the round evaluation covers (k_rounds_check): it fetches
                                    midRound[r0] once and then 101 arrays per round, the verified midRound[r+1] staying in registers as the next round's
                                    input -- (101 k + 25) / k arrays of 512 B per (64 witnesses, round)                                               */
    uint32_t kgc_rounds;         /* ... that one wavefront of the launch which expands AND evaluates the round blocks covers (k_rounds_gc, pob_set_inorder bit 2): per round
                                    76 arrays stored and loaded back + the 25 of the stored midRound[r+1], midRound[r0] once                             */
} pob_info_t;

/* Replaces `component main = ProofOfBurn(...)` / `Spend(...)` + circom -c + make (reference
 * circuits/main_proof_of_burn.circom:27, circuits/main_spend.circom:6, Makefile:2-3).
 * params: template parameters as canonical 4x64-bit LE limbs each:
 *   ProofOfBurn: maxNumLayers, maxNodeBlocks, maxHeaderBlocks, minLeafAddressNibbles, amountBytes,
 *                powMinimumZeroBytes, maxIntendedBalance, maxActualBalance   (proof_of_burn.circom:34)
 *   Spend:       maxAmountBytes                                              (spend.circom:32)
 *   POB_CIRCUIT_GADGET: a gadget-level main -- what the reference's test harness builds with `component main = T(params);`
 *                around ONE template (tests/test.py:24-33, the 54 non-circuit entries of tests/test.py:146-201):
 *                params[0] = pob_gadget_template("T"), params[1..] = T's template parameters.  Inputs: the main's input signals in
 *                declaration order, field-valued ones in fr_inputs, the others (bytes, lengths, selectors: int32) in sm_inputs --
 *                pob_plan_info gives the two counts; outputs: wires 1 .. n_outputs of the witness (pob_emit_begin / pob_write_wtns),
 *                which is where the reference's harness reads them (tests/test.py:40-47); pob_results' commitment is zero.          */
int pob_open(int device, int circuit, const uint64_t* params, int nparams, uint32_t max_batch, pob_handle* out);
void pob_close(pob_handle h);
int pob_get_info(pob_handle h, pob_info_t* info);
/* Layout planner only (no GPU touched): wire / class counts of an instantiation, e.g. to size buffers. */
int pob_plan_info(int circuit, const uint64_t* params, int nparams, pob_info_t* info);
const char* pob_strerror(pob_handle h);
/* template name (as written after `component main =`, e.g. "SubstringCheck") -> id for params[0] of POB_CIRCUIT_GADGET; *nparams = the
 * number of template parameters it takes.  -1: not a template of the reference's circuits/utils. */
int pob_gadget_template(const char* name, int* nparams);

/* Replaces the emitted loader (loadJson; reference tests/test.py:57-59 writes input.json).  Witness-major:
 *   fr_inputs[n][n_fr_inputs][32]  canonical LE field elements, circuit declaration order of the FR-class inputs
 *     ProofOfBurn: burnKey, actualBalance, intendedBalance, revealAmount, burnExtraCommitment, _proofExtraCommitment
 *     Spend:       burnKey, balance, withdrawnBalance, extraCommitment
 *   sm_inputs[n][n_sm_inputs]      int32, declaration order of the small inputs
 *     ProofOfBurn: numLeafAddressNibbles, layers[L][136*NB], layerLens[L], numLayers, blockHeader[136*HB],
 *                  blockHeaderLen, byteSecurityRelax      (values >= 2^31 must be rejected by the caller:
 *                  every one of them is range-constrained to <= 16 bits in-circuit)
 * Host pointers; the copy to HBM is synchronous.  The inputs are double-buffered on the device (generation AND evaluation read them): the
 * batch becomes the current one with the next pob_generate, so it may be uploaded while the previous batch is still being worked on. */
int pob_upload_inputs(pob_handle h, const uint8_t* fr_inputs, const int32_t* sm_inputs, uint32_t n);
/* The same for a service loop: the copies are enqueued on `stream` (NULL = the handle's upload stream) into the input buffer the current
 * batch does not use and the following pob_generate waits for them, so batch k+2 is uploaded while batches k and k+1 are in flight.
 * The host buffers must be pinned (pob_host_alloc) for the copy to be asynchronous and must stay
 * untouched until that pob_generate has been enqueued and its stream has passed the copy (e.g. until the batch's results are in). */
int pob_upload_inputs_async(pob_handle h, const uint8_t* fr_inputs, const int32_t* sm_inputs, uint32_t n, void* stream);
/* The emitted loader itself, natively (reference: loadJson of the emitted calculator; tests/test.py:57-59 and tests/main.py:160-178 write
 * the input.json it reads): one input.json TEXT -> the packed row of pob_upload_inputs (fr_row[n_fr_inputs][32], sm_row[n_sm_inputs],
 * *forced = 0 or the failure code of an input that cannot fit its row: such a witness is reported as failed whatever the device computes).
 * Accepts what the host mirror (witness.py WitnessCalculator.pack) accepts: exact key set, scalars possibly wrapped in one-element arrays,
 * nested arrays flattened, JSON integers of any size / booleans / decimal or 0x strings, reduced mod p; anything else is POB_E_ARG with
 * the reason in err[errcap].  params as for pob_open.  No GPU is touched.
 * pob_pack_json_batch: n texts on `threads` host threads (0 = all) into witness-major arrays, e.g. pinned memory from pob_host_alloc. */
int pob_pack_json(int circuit, const uint64_t* params, int nparams, const char* json, uint64_t len, uint8_t* fr_row, int32_t* sm_row,
                  uint32_t* forced, char* err, uint32_t errcap);
int pob_pack_json_batch(int circuit, const uint64_t* params, int nparams, const char* const* json, const uint64_t* len, uint32_t n, int threads,
                        uint8_t* fr, int32_t* sm, uint32_t* forced, char* err, uint32_t errcap);
/* The same inputs with the byte-class values as BYTES on the wire (round 5: every one of ProofOfBurn's small inputs is a byte, a nibble count or a length --
 * proof_of_burn.circom:43-72 -- and 10 900 of them per witness as int32 made the upload 44.8 MB per 1 024 witnesses, 0.79 ms of PCIe per 1.9 ms step):
 *   sm8[n][n_sm_inputs]        the small inputs that are in 0..255, as they are; 0 where the value is not
 *   exc[n][POB_EXC_CAP]        per witness the (few) small inputs outside 0..255 -- layerLens[] and blockHeaderLen above 255, and whatever a caller feeds
 *                              out of range on purpose (the reference does: tests/testcases/rlp/integer.py:51-53) -- as {index in the row, int32 value};
 *                              unused slots: k = POB_EXC_NONE
 * 11.4 KB per production witness instead of 43.8.  ProofOfBurn: the upload is three copies and nothing else -- the generation's first kernel (and the evaluation's
 * input check) read the byte form and write / compare the inputs' wires in one pass (round 6); the other circuits' kernels read int32 rows, for them the device
 * widens the bytes on the upload stream, behind the copy.  The two forms are interchangeable batch by batch, results identical.  A witness with more than POB_EXC_CAP values
 * outside 0..255 does not fit: pob_narrow_inputs / pob_pack_json_batch8 return POB_E_RANGE and the batch goes through the int32 entry points. */
#define POB_EXC_CAP 32
#define POB_EXC_NONE 0xFFFFFFFFu
typedef struct { uint32_t k; int32_t v; } pob_sm_exc_t;
int pob_upload_inputs8(pob_handle h, const uint8_t* fr_inputs, const uint8_t* sm8, const pob_sm_exc_t* exc, uint32_t n);
int pob_upload_inputs8_async(pob_handle h, const uint8_t* fr_inputs, const uint8_t* sm8, const pob_sm_exc_t* exc, uint32_t n, void* stream);
/* int32 rows (pob_pack_json*, the Python packer) -> the byte form; no GPU is touched */
int pob_narrow_inputs(const int32_t* sm, uint32_t n, uint32_t n_sm_inputs, uint8_t* sm8, pob_sm_exc_t* exc);
/* the loader straight into the byte form (same acceptance, same forced codes as pob_pack_json_batch) */
int pob_pack_json_batch8(int circuit, const uint64_t* params, int nparams, const char* const* json, const uint64_t* len, uint32_t n, int threads,
                         uint8_t* fr, uint8_t* sm8, pob_sm_exc_t* exc, uint32_t* forced, char* err, uint32_t errcap);
/* Pinned host memory for the above (hipHostMalloc), so that a ctypes / cgo caller need not link the HIP runtime itself. */
int pob_host_alloc(void** p, uint64_t bytes);
void pob_host_free(void* p);

/* Replaces the run of the calculator (all `<==` / `<--` / `===`; reference Makefile:4-5): enqueues every stage on
 * `stream` (a hipStream_t, NULL = the handle's own stream) for the n uploaded inputs.  Asynchronous.  pob_generate and
 * pob_constraint_check of one handle may be given different streams: each is ordered behind the other's last call by the handle's own events
 * (the same stream needs none).                                                                                    */
int pob_generate(pob_handle h, void* stream);
/* Per-gate constraint evaluator over the resident witness vector (north star; the reference checks inline,
 * e.g. assert.circom:46,62,78, divide.circom:32): re-reads every STORED wire and checks it against its defining expression evaluated on
 * STORED operands, and every `===`; asynchronous.  Wires without storage are not read: a derived wire's (n_derived) relations hold by
 * construction in whatever the emitter writes for it, an alias wire's (n_alias) copy constraint holds structurally, and every non-copy
 * gate that consumes one is evaluated on the stored wire it stands for.  pob_emit_begin*'s self-check (POB_EMIT_SELFCHECK) evaluates the
 * derived / alias wires' own relations on the values as written.                                                  */
int pob_constraint_check(pob_handle h, void* stream);
int pob_sync(pob_handle h);
/* Two-batch pipeline (no counterpart in the reference, whose calculator runs one witness per process, Makefile:4-5): links two
 * handles of one device that work on consecutive batches.  While linked, a handle's pob_generate starts once the partner's
 * generation is complete -- so its latency-bound stages run beside the partner's (read-streaming) evaluation -- and its Keccak
 * round expansion, the HBM-write-saturating kernel, follows the end of that evaluation; pob_constraint_check then runs on streams
 * the generation does not use.  Call order per batch: pob_constraint_check(previous), pob_generate(next).  The link is
 * symmetric (one call links both handles and drops their previous links); partner = NULL unlinks; pob_close unlinks. */
int pob_set_partner(pob_handle h, pob_handle partner);

/* Schedule of a calculator (no counterpart in the reference, whose calculator runs one witness per process, Makefile:4-5).
 * Default -- TRACKS: pob_generate / pob_constraint_check spread the circuit's independent parts over side streams of the device (lowest latency for ONE batch:
 * what a lone calculator wants; pob_set_partner pipelines two of them).
 * on = 1 -- IN ORDER: every launch of the calculator goes to the caller's stream, in dependency order, with no side stream and no event (a cross-stream hand-over
 * costs 0.1-0.3 ms on this runtime, a dependent kernel boundary on one stream 1.5 us): the units of every track that are ready at the same depth of the stage graph
 * leave in ONE launch.  A batch takes longer by itself; a job that keeps several calculators in flight, each on a stream of its own, fills the machine with them.
 * on = 3 -- IN ORDER with the FUSED launch (what bench.py runs): the lane-spread Poseidon blocks share a launch with the sponge chain that does not depend on them
 * (header and layers); the units that continue from the blocks' outputs follow one level later.  A lone batch's launches: 3.03 -> 2.61 ms; same wires, same records.
 * on | 4 -- the Keccak ROUND BLOCKS ARE EVALUATED BY THE LAUNCH THAT WRITES THEM (k_rounds_gc): the wavefront that has stored the 76 gate-output arrays of a round loads them
 * back -- from L2 / the Infinity Cache instead of HBM -- together with the stored midRound[r+1] and checks every XOR / AND gate of the round on the loaded operands, as
 * pob_constraint_check's round kernel does; pob_constraint_check then skips that kernel (1.77 of the 2.27 GB it would read per 1 024 production witnesses); the input rows are compared with the
 * inputs by the launch that writes them (k_inputs MODE 2), and -- for the two main circuits -- every G unit's stores are loaded back and compared by the unit itself
 * (policy.hpp GenPT<true>, poseidon_wide.hpp PosWideT<true>; the RLP units excepted): of the evaluation pass pob_constraint_check then runs the sponge chains' kernel and the
 * RLP family's and collects the records -- unless a debug poke touched the vector in between, in which case it runs everything.  Same records, same failing sites. */
int pob_set_inorder(pob_handle h, int on);

/* Replaces "stderr non-empty => failure" + the output dump patched in by tests/test.py:36-54.
 * status[i] = 0 ok, else (template id << 12 | source line) of the first failing assert; outputs[i][32] = public
 * output (commitment) canonical LE.  check_status / bad_wire (may be NULL): result of pob_constraint_check:
 * first failing === site and lowest wire whose stored value contradicts its definition (0xFFFFFFFF = none;
 * POB_NOT_EVALUATED = no pob_constraint_check has run on this batch).                                             */
int pob_results(pob_handle h, uint32_t* status, uint8_t* outputs, uint32_t* check_status, uint32_t* bad_wire);
/* Device-resident results for the RCCL gather: status u32[max_batch_padded], outputs u8[max_batch_padded][32]. */
int pob_results_device(pob_handle h, void** d_status, void** d_outputs);
/* The same results as ONE packed record per witness (reference: one pass/fail + outputs per run, tests/test.py:40-47,65-68),
 *   {u32 status, u32 check_status, u32 bad_wire, u8 commitment[32]} = 44 bytes,
 * device-resident (u8[max_batch_padded][44]): what a rank contributes to the single all-gather of the multi-GPU path.
 * Written at the end of pob_generate with check_status = bad_wire = POB_NOT_EVALUATED and re-written at the end of
 * pob_constraint_check with the evaluator's verdict (0xFFFFFFFF there = clean), i.e. AFTER the evaluation of the batch.         */
#define POB_RECORD_BYTES 44
#define POB_NOT_EVALUATED 0xFFFFFFFEu
int pob_results_records_device(pob_handle h, void** d_records);
/* The multi-GPU path's ONE collective for a host that is not Python (bench.py does the same through torch.distributed): all-gather of this handle's device records
 * over RCCL.  comm = the caller's ncclComm_t (one rank per GPU, created by the caller: ncclCommInitRank), stream = the stream the collective runs on (ordered behind
 * the handle's last evaluation -- or generation -- by an event), d_out = device memory for nranks x n_per_rank x POB_RECORD_BYTES bytes; every rank passes the same
 * n_per_rank (>= its batch: uneven slices pad, distributed.py gather_records shows the trimming).  librccl.so is loaded on first use (dlopen: the library has no
 * link-time dependency on it); POB_E_STATE if it cannot be loaded, POB_E_HIP with the RCCL error text in pob_last_error otherwise.
 * Replaces: nothing in the reference (one witness per process, Makefile:4-5); SURVEY.md 8e.                                                                     */
int pob_gather_records(pob_handle h, void* comm, void* stream, void* d_out, uint32_t n_per_rank);
/* Per-batch, host-visible verdicts without stalling the device: the kernel that packs the records also writes them straight into pinned
 * host memory owned by the handle (two buffers alternating per batch: no copy, no copy stream).  pob_results_fetch marks the CURRENT
 * batch's buffer as the one to read; pob_results_wait blocks until that buffer is written -- an event of this handle, recorded behind its
 * last pob_constraint_check (or pob_generate, if no evaluation was enqueued since): a partner handle's work is not waited for -- and hands
 * out the records (n x POB_RECORD_BYTES).  The pointer stays valid until the handle's pob_generate after the next one.            */
int pob_results_fetch(pob_handle h);
int pob_results_wait(pob_handle h, const uint8_t** records, uint32_t* n);

/* Replaces writeBinWitness (patch point `fclose(write_ptr)` at reference tests/test.py:36): expands witness `idx`
 * of the batch to canonical 32-byte LE values.  pob_emit_witness: payload only (32*W bytes) into host memory;
 * pob_write_wtns: full iden3 .wtns file.                                                                         */
int pob_emit_witness(pob_handle h, uint32_t idx, uint8_t* dst, uint64_t cap);
int pob_write_wtns(pob_handle h, uint32_t idx, const char* path);
/* .wtns of the reduced witness (pob_emit_begin_reduced): nWitness = n_keep. */
int pob_write_wtns_reduced(pob_handle h, uint32_t idx, const uint32_t* keep, uint64_t n_keep, const char* path);
/* The streaming form underneath both: the canonical payload is expanded from the compact resident vector in WINDOWS of
 * `window_wires` wires (0 = 8 Mi wires = 256 MiB), double-buffered on the device and in pinned host memory, so that expanding window
 * k+1, copying it D2H and the caller's consumption of window k overlap, and a whole 32*W-byte device buffer never exists.
 * pob_emit_next hands out the next window (pinned host memory owned by the handle, valid until the following call) with its first
 * wire and wire count; n_wires = 0 ends the witness.  The buffers are kept for the next witness.  A witness whose status is non-zero
 * is refused (POB_E_STATE): like the reference binary, a failed input produces no witness (reference tests/test.py:65-68).         */
int pob_emit_begin(pob_handle h, uint32_t idx, uint64_t window_wires);
/* Reduced witness (reference: the circuits are compiled with --O0 today, Makefile:2-3; the deployed keys use the compiler's default
 * simplification, .github/workflows/circuitscan.yml:29,36): only the wires keep[0..n_keep) (strictly increasing O0 wire indices,
 * keep[0] == 0; which representative circom keeps is data of the caller, see circuit_model/o1.py) are expanded and copied, in windows of
 * `window_wires` KEPT wires; a Keccak round block whose wires are all dropped costs nothing.  pob_emit_next then hands out windows
 * of the reduced payload (first_wire / n_wires count kept wires).  The map is uploaded once per (handle, keep pointer contents).   */
int pob_emit_begin_reduced(pob_handle h, uint32_t idx, const uint32_t* keep, uint64_t n_keep, uint64_t window_wires);
/* A caller that emits many witnesses through ONE map pins it: the map at this address and length is hashed now and recognised by address from then
 * on (the unpinned calls above hash the caller's whole map every time to recognise it: 9 ms for the production map's 86 MB).  The caller promises
 * that keep[0..n_keep) stays allocated and unchanged until it is unpinned (keep = NULL) or the handle is closed.  One pinned map per handle. */
int pob_reduced_map_pin(pob_handle h, const uint32_t* keep, uint64_t n_keep);
int pob_emit_next(pob_handle h, const uint8_t** data, uint64_t* first_wire, uint64_t* n_wires);
/* Announce the witness that will be emitted AFTER the current (or the next) one, with the same payload kind and window size: its first
 * window -- the one that holds most gadget wires and takes longest to expand -- is then expanded while the current witness' last windows are
 * still being copied, and the pob_emit_begin* / pob_write_wtns* call for it continues from there (three window slots per handle).  Without
 * it every witness pays its first expansion un-overlapped (6 of 134 ms for the production circuit).  POB_E_STATE: next_idx failed an assert. */
int pob_emit_queue(pob_handle h, uint32_t next_idx);
/* Emit-time self-check of what is written (the reference checks every `===` on the values it writes: inline asserts of the emitted calculator, e.g.
 * circomlib comparators.circom IsZero, utils/substring_check.circom:45-49).  The constraint evaluator reads STORED wires; the derived wires (n_derived)
 * exist only in the emitter's output, so their own relations are evaluated THERE, on the canonical values as written into each emission window by a
 * kernel that re-reads the window before it is copied to the host:
 *     every IsZero [out | in | inv]:                 in * inv === 1 - out,   in * out === 0
 *     every IsEqual [out | in[2]] over an IsZero:    IsZero.in === in[1] - in[0],   out === IsZero.out
 *     SubstringCheck's M[]:                          M[i+1] === M[i] + mainInput[i] * 256^i
 *     a derived copy of a STORED wire:               Pad.isEq[i] / isLast[i] / Selector.isEq[i] (stored) === the IsEqual child's out, its IsZero's out (derived)
 * enable = 1: every following emission (pob_emit_begin / pob_emit_witness / pob_write_wtns, and -- round 5 -- the reduced form, pob_emit_begin_reduced /
 * pob_write_wtns_reduced: what a prover built at circom's default level consumes, .github/workflows/circuitscan.yml:29,36) is checked; the first one also
 * runs a recording pass that finds the sites.  In the reduced form a site is evaluated where it lies among the KEPT wires (each wire through the keep
 * bitmap, at its rank); a site with a dropped wire is skipped -- its relation holds between class representatives, which the map alone does not name.
 * pob_emit_selfcheck_result, called when the emission is complete: relations checked, relations skipped (wires straddling two windows, dropped wires), and
 * the lowest wire whose relation does not hold (0xFFFFFFFF = none): of THIS emission -- while the check is on, pob_emit_queue does not pre-make the next
 * witness' window.  A witness emitted from a corrupted resident vector (pob_debug_poke of an operand) violates the relations of the derived wires that
 * consume the operand.                                                                                                                          */
int pob_emit_selfcheck(pob_handle h, int enable);
/* Reduced emissions: the class representative of every O0 wire under the map that will be emitted (alias[w] = the kept wire that stands for w, w itself if
 * it is kept; a wire pinned to the constant c < 2^30: -1 - c, to a larger constant: INT32_MIN; from circuit_model/o1.py O1Map), n_wires = n_witness.  With
 * it a site whose wires were dropped is evaluated on their representatives, without it only the sites whose own wires are all kept are.  The library keeps
 * the POINTER and reads the array whenever a reduced emission begins with a map whose site lists it has not built yet: the array must stay valid and
 * unchanged for as long as it is set -- until it is replaced, cleared (NULL) or the handle is closed.                                              */
int pob_emit_selfcheck_alias(pob_handle h, const int32_t* alias, uint64_t n_wires);
int pob_emit_selfcheck_result(pob_handle h, uint64_t* checked, uint64_t* skipped, uint32_t* first_bad_wire);
/* Measurement: `count` witnesses from `first_idx` on, back to back through the window pipeline into pinned host memory.          */
int pob_emit_measure(pob_handle h, uint32_t first_idx, uint32_t count, uint64_t window_wires, double* seconds, uint64_t* bytes);
/* The same with an optional reduced map (keep != NULL: pob_emit_begin_reduced).  The first witness a handle emits at a window size
 * also allocates the window buffers (2 x device, 2 x pinned host) and runs the probe pass: call twice to separate that from the steady state. */
int pob_emit_measure_ex(pob_handle h, uint32_t first_idx, uint32_t count, uint64_t window_wires, const uint32_t* keep, uint64_t n_keep, double* seconds, uint64_t* bytes);

/* Measurement: average duration (ms, HIP events on `stream`) of `iters` back-to-back launches of one kernel over
 * the current batch.  which: 0 = Keccak round expansion (generate), 1 = Keccak round constraint evaluation,
 * 2 = G-unit constraint evaluation (every family, back to back), 3 = sponge chain (generate); 100 + k / 200 + k = evaluation /
 * generation of all units of kind k (circuits.hpp UnitKind) alone on the device; 300 + f = the evaluation kernel of family f
 * (circuits.hpp Fam) alone (tools/unit_times.py); 6 = round expansion + evaluation in one launch (k_rounds_gc).  */
int pob_time_kernel(pob_handle h, int which, int iters, void* stream, float* avg_ms);
/* Measurement INSIDE a running job: enable = 1 makes every following pob_constraint_check record HIP events (on the stream the kernel is
 * launched on) around its Keccak round evaluation kernel -- the dominant kernel as it runs in the step, beside the other batch's
 * generation; *ms (may be NULL) = duration of the last such kernel, whose batch must be complete.  enable = 0 stops recording.    */
int pob_probe_check_kernel(pob_handle h, int enable, float* ms);

/* Test hook: XOR `mask` into the stored word of BIT-class storage index `bit_index` of witness group `group`. */
int pob_debug_xor_bits(pob_handle h, uint32_t group, uint64_t bit_index, uint64_t mask);
/* Test hook for the evaluation that rides with the generation (pob_set_inorder bit 2): in the NEXT pob_generate of this in-order calculator ONE store reaches memory corrupted
 * while the generating wavefront goes on with the right value: the word of storage class `cls` at rank `index` of group `group`, for the witnesses of `mask` (bit l = witness l of
 * the group): a BIT word XORed with `mask`, an SM value / the lowest limb of an FR element with bit 0 flipped.  The launch's own evaluation, which works on what it LOADS, must
 * flag exactly those witnesses.  *wire (may be NULL) = the wire it must report where the hook knows it -- a gate output of a KeccakfRound block: the block's first wire; an input
 * row of a ProofOfBurn main: the input's wire -- else 0xFFFFFFFF: a store of a G unit (reported at the wire itself) if a riding unit stores the word at all (the sponge chains'
 * words and the RLP units' wires are stored by launches that do not evaluate).  POB_E_ARG: no such word.  One generation, then disarmed. */
int pob_debug_store_fault(pob_handle h, int cls, uint32_t group, uint64_t index, uint64_t mask, uint32_t* wire);
/* Experiment hook (profiles/round6_experiments.txt 9): a non-blocking stream of `device` restricted to the compute units of cu_mask (hipExtStreamCreateWithCUMask; words x 32 bits,
 * bit i = CU i), for callers that want to partition the device between calculators.  pob_debug_stream_destroy frees it. */
int pob_debug_stream_create(int device, const uint32_t* cu_mask, uint32_t words, void** stream);
void pob_debug_stream_destroy(int device, void* stream);
/* Test hook for the constraint evaluator: corrupt ONE stored value of ONE witness (lane `lane` of group `group`) of storage class
 * `cls` at storage index `index` (the wire's rank within its class): BIT: flips the bit if xor_mask & 1; SM: int32 ^= xor_mask;
 * FR: 32-bit limb `sub` (Montgomery form) ^= xor_mask.  (Derived and alias wires -- pob_info_t.n_derived / n_alias -- have no storage
 * to corrupt: corrupting the stored wire they are a function / a copy of changes them with it.)                                    */
enum { POB_CLASS_BIT = 0, POB_CLASS_SM = 1, POB_CLASS_FR = 2 };
int pob_debug_poke(pob_handle h, int cls, uint32_t group, uint64_t index, uint32_t sub, uint32_t lane, uint32_t xor_mask);
/* Test hook: how many IsZero.inv wires the emitter has written since the last reset through each of its paths: out[0] from the table of small
 * inverses (|operand| <= 4096), out[1] by Fermat exponentiation (larger small operands), out[2] / out[3] field-element operands, non-zero
 * (Kaliski inversion) / zero.  Call when an emission is complete.                                                                     */
int pob_debug_emit_counters(pob_handle h, uint64_t out[4], int reset);
/* Test hook: the device code's two field inversions (the generator's / emitter's Kaliski almost-inverse and the emitter's Fermat fall-back) on n
 * canonical 32-byte LE inputs < p (0 -> 0), canonical outputs.                                                                          */
int pob_debug_fr_inv(int device, const uint8_t* in, uint32_t n, uint8_t* out_kaliski, uint8_t* out_fermat);
/* Test hook: storage class, rank within the class and wire index of a few named wires: "commitment"; "poseidon" (k-th wire of the
 * first Poseidon block); "pad.div.out" / "pad.div.rem" of KeccakBytes instance k (the Divide hint of divide.circom:23-24); ProofOfBurn only: "sc.exists" [k] of layer 1's SubstringCheck. */
int pob_debug_ref(pob_handle h, const char* name, uint32_t k, int* cls, uint64_t* index, uint64_t* wire);      /* also "kb.inLen": inLen of KeccakBytes instance k */

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
