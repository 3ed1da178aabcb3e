// Top-level circuits (ProofOfBurn, Spend) and the Keccak wrapper, cut into UNITS.
//
// A unit = a piece of the component tree that one wavefront (64 witnesses, lane = witness) executes start to finish;
// its UnitDesc carries the O0 cursor where its first wire lives.  The same `unit_run<P>` body serves the host planner
// (P = CountP), the generator (GenP), the constraint evaluator (CheckP) and the .wtns emitter (EmitP).  Units of one STAGE are
// independent; a unit only reads wires written in earlier stages (or by itself).  Keccak sponges run between G stages on
// the bit-sliced kernels (keccak_kernels.hpp).
//
// Large templates are cut into RANGE units (byte ranges of KeccakBytes/AssertByteString, selector ranges, ShiftLeft rows,
// SubstringCheck position ranges) so that a stage consists of thousands of short wavefronts instead of a few long ones.
// Range units jump to their wires with cursor arithmetic (fixed footprint per array element); the host planner validates
// every such jump against the monolithic template walked by CountP (`expect_cursor`), so the wire order keeps ONE definition.
#pragma once
#include <string.h>
#include "gadgets.hpp"
#include "poseidon_consts.h"

#define KECCAKF_ROUND_WIRES 102656u
#define KECCAKF_WIRES (43200u + 24u * KECCAKF_ROUND_WIRES)        // 2 506 944
#define ABSORB_OWN 5888u                                            // out, s, block[17], aux
#define ABSORB_WIRES (ABSORB_OWN + 17u * 384u + KECCAKF_WIRES)     // 2 519 360
// Storage of an Absorb block (round 4).  Its own wires, its 17 XorArrays and Keccakf's own wires (out, in, midRound[25]) are stored 1:1
// (AB_DIRECT BIT ranks in wire order).  Of the 102 656 wires of a KeccakfRound block only the KR_STORED = 76 arrays of 64 that are OUTPUTS
// OF A GATE and not the round's output state are stored (20 Xor5 partials, 5 D, 25 theta, 25 chi-AND, chi.out[0] before iota); every other
// wire of the block is an ALIAS: a copy of a stored wire / of midRound[r] / of midRound[r+1], possibly at a rotated bit position or
// negated (NotArray), or a constant (shifted-out positions, round constants).  keccak_kernels.hpp states which (one walker), the emitter
// expands through that table.
#define KR_STORED 76u
#define KR_BITS (KR_STORED * 64u)                                   // 4 864 BIT ranks per round block
#define AB_DIRECT (ABSORB_OWN + 17u * 384u + 43200u)               // 55 616
#define ABSORB_BITS (AB_DIRECT + 24u * KR_BITS)                    // 172 352
#define ABSORB_ALIAS (ABSORB_WIRES - ABSORB_BITS)                  // 2 347 008 alias wires per permutation

enum UnitKind : uint32_t {
    U_POB_INPUT = 1, U_POB_RANGE, U_POB_LAYER_ASSERT, U_POB_HDR_ASSERT, U_POB_POSEIDONS, U_BAH_PRE, U_BAH_POST,
    U_KB_HEAD, U_KB_RANGE, U_KB_SELROW, U_KB_POST, U_POB_N2B, U_PC_PRE, U_PC_POST, U_POB_LASTLAYER, U_POB_LASTLAYER_RANGE,
    U_POB_LASTLEN, U_POB_LEAF, U_POB_LAYER_POST, U_SC_M, U_SC_RANGE, U_SC_SUMS, U_POB_LASTLEAF,
    U_RL_A, U_RL_SLROW, U_RL_ACC, U_RL_B, U_POW_PRE, U_POW_POST, U_POB_FINAL, U_ABS_RANGE, U_LD_HEAD, U_LD_SELR, U_LD_TAIL, U_POB_INPUT_FR, U_RL_ACC_B, U_RL_ACC_C,
    U_SP_INPUT, U_SP_HEAD,
    // evaluator-only units (UNIT_CHECK): sub-blocks of composite units run as wavefronts of their own, from STORED wires
    U_SC_MI,             // SubstringCheck(a0).mainInput / mainLen copies (inputs only: runs on the pre-work track)
    CK_POS_SEG,          // a0 = T, a1 = segment (>= 1) of the Poseidon block at cur (gadgets.hpp gPoseidonSegStored)
    CK_SR_COLS,          // (rounds 2-3: columns of ShiftRight's stored temps[][]; they are derived wires now -- the kind is never planned, the number stays)
    CK_SL_ROWS,          // rows [a1, a2) of the ShiftLeft(a0) block at cur, IsEqual children from cursor (a3, a4, a5) on (gShiftLeftRows)
    CK_N2BE,             // Num2BigEndianBytes(a0) at cur; a1,a2 = source FR wire; a3,a4,a5 = caller's copy of out[] (w, i, present);
                         // also a GENERATION unit (same stage as its composite): a6,a7 = a wire of an earlier stage with the source's value
    CK_CAT,              // child a0 (0 Mask(A), 1 Mask(B), 2 ShiftRight, 3 the sums) of the Concat(a1, a2) block whose first wire / SM rank are a3 / a4; cur = the child (gConcatPart)
    CK_RL_B2, CK_RL_B3,  // U_RL_B's evaluation cut at stored wires (keyLen / accLen; pkLen / valLen): the value / key prefixes, Concat(3 + 33, 2 + 4 + amountBytes + 66) + the leaf copies
    // generation-only unit (UNIT_GEN): the Poseidon(a0 - 1) block at cur with the state spread over lanes (poseidon_wide.hpp);
    // a1 = prefix index, a2..a4 = FR ranks of inputs 1.., a5 = FR rank subtracted from the last input, a6 = FR rank of the caller's copy
    U_POS_WIDE,
    // gadget-level mains (gadget_mains.hpp): a0 = template, a1..a4 = its parameters | the packed inputs of a main planned from split units
    U_GM, U_GM_INPUT,
    U_PC_COPY,           // PublicCommitment: the 32 bytes of input a0 through every copy the template makes of them (in[][], AssertByteString, Flatten in / out, Fit in / out, the Keccak block); a0 = N: the zero padding
    U_LD_COPY,           // LeafDetector a0: layer[a1 .. a2) <== the source bytes (the head's 544-byte copy as range units: in the head it was 68 dependent batches of loads)
    U_KIND_COUNT
};
enum : uint32_t { UNIT_GEN = 1, UNIT_CHECK = 2, UNIT_EMIT = 4 };      // UnitDesc.flags: generation / constraint evaluation / .wtns emission
// Kernel FAMILIES: every family is compiled as a kernel of its own (own register allocation -- one kernel over every unit kind
// spilled ~1.9 k VGPRs in the evaluator); generation merges families into the classes of its stage scheduler.
enum Fam : uint32_t { F_MISC = 0, F_RANGE, F_SELROW, F_LD, F_RL, F_SC, F_POS, F_N2B, F_GM, F_COUNT };      // F_GM: gadget-level mains (kernels of their own, g_*_gm.hip)
#define FAM_BIT(f) (1u << (f))
#define FAM_LIGHT (FAM_BIT(F_MISC) | FAM_BIT(F_RANGE) | FAM_BIT(F_SELROW) | FAM_BIT(F_LD) | FAM_BIT(F_RL))
#define FAM_HEAVY (FAM_BIT(F_POS) | FAM_BIT(F_N2B))
#define FAM_ALL ((1u << F_COUNT) - 1)
HD constexpr uint32_t fam_of(uint32_t k) {
    return (k == U_KB_RANGE || k == U_ABS_RANGE) ? F_RANGE
         : k == U_KB_SELROW ? F_SELROW
         : (k == U_LD_HEAD || k == U_LD_SELR || k == U_LD_TAIL || k == U_LD_COPY || k == U_POB_LASTLAYER_RANGE || k == U_SC_SUMS) ? F_LD
         : (k == U_RL_A || k == U_RL_SLROW || k == U_RL_ACC_B || k == U_RL_ACC_C || k == U_RL_B || k == CK_SR_COLS || k == CK_SL_ROWS || k == CK_RL_B2 || k == CK_RL_B3 || k == CK_CAT) ? F_RL
         : (k == U_POB_LAYER_POST || k == U_SC_M || k == U_SC_RANGE) ? F_SC
         : (k == U_POB_POSEIDONS || k == U_BAH_PRE || k == U_SP_HEAD || k == CK_POS_SEG || k == U_POS_WIDE) ? F_POS
         : (k == U_GM || k == U_GM_INPUT) ? F_GM
         : (k == U_POB_INPUT_FR || k == U_POB_RANGE || k == U_POB_N2B || k == U_PC_POST || k == U_RL_ACC || k == U_POW_PRE || k == U_SP_INPUT || k == CK_N2BE) ? F_N2B
         : F_MISC;
}
// the kernel a unit's GENERATION is compiled into may differ from its evaluation / emission family: PublicCommitment's head is a long BIT/SM unit whose
// generation needs more than the light kernel's 64 VGPRs (it spilled 12) -- it is generated by the <= 128-VGPR kernel, evaluated and emitted where it was
HD constexpr uint32_t gen_fam_of(uint32_t k) { return k == U_PC_PRE ? (uint32_t)F_N2B : fam_of(k); }
// (a riding generation policy does not instantiate the unit kinds that run on the plain policy inside its kernel: ride_keeps_evaluation, below)
HD constexpr bool ride_keeps_evaluation(uint32_t kind) { return fam_of(kind) == F_RL; }
#define UCASE(K) case K: if constexpr ((((MASK) >> (P::is_gen ? gen_fam_of(K) : fam_of(K))) & 1u) && !(P::ride && ride_keeps_evaluation(K)))

struct PobParams { int L, NB, HB, minNib, amountBytes, powZero; Fr maxIntended, maxActual; };   // Montgomery
struct SpendParams { int maxAmountBytes; };

// footprints {wires, BIT, SM, FR} of fixed-size components
#define FP_ISEQ_S (Cur{6, 2, 0, 0, 4})       // IsEqual [out | in[2]] + IsZero [out | in | inv] over small operands: two BIT outputs, four DERIVED operand wires (policy.hpp)
#define FP_ISEQ_F (Cur{6, 2, 0, 0, 4})       // ... over field elements with derived operand wires (gIsEqualFd)
#define FP_N2B8 (Cur{9, 0, 0, 0, 9})         // Num2Bits(8) [out[8] | in] of KeccakBytes' byte loop: derived (the byte's bits are stored once, as Keccak's inBlocks)
#define FP_ISEQ_D (Cur{6, 0, 0, 0, 6})       // IsEqual + IsZero whose two outputs are copies of a stored bit of the parent (Pad's isEq[] / isLast[]): all six wires derived

// references to main's own wires (proof_of_burn.circom:41-72 in/out, :113-200 intermediates)
struct PobMain {
    FrRef commitment, burnKey, actualBalance, intendedBalance, revealAmount, burnExtraCommitment;
    SmRef numLeafAddressNibbles, layers, layerLens, numLayers, blockHeader, blockHeaderLen, byteSecurityRelax;
    FrRef proofExtraCommitment, remainingCoin, nullifier;
    SmRef addressHashNibbles, blockRoot, stateRoot, nullifierBytes, remainingCoinBytes, revealAmountBytes, burnExtraCommitmentBytes,
        extraCommitmentBytes, lastLayer, lastLayerLen;
    BitRef layerExists, substringCheckers;
    SmRef layerKeccaks, reducedLayerKeccaks;
    BitRef isLeaf, isLastLayerLeaf;
    SmRef leaf, leafLen;
};
struct SpendMain {   // spend.circom:33-38, :43-49
    FrRef commitment, burnKey, balance, withdrawnBalance, extraCommitment, coin, remainingCoin;
    SmRef coinBytes, withdrawnBalanceBytes, remainingCoinBytes, extraCommitmentBytes;
};
// KeccakBytes own wires, Pad's own wires, and the Keccak/Final/SelectorArray2D wires its G units touch
struct KBRefs {
    SmRef out, in, inLen, numBlocks; BitRef inBitsArray, inBits, inBlocks, outBits, outBytes;       // (round 4: of these only inBlocks is stored; a DERIVED BIT array has .i = NO_RANK)
    uint32_t padded_w, pad_o_w, pad_in_w;          // padded[m], Pad.out[m], Pad.in[m]: derived wires (functions of the byte, inLen, numBlocks)
    SmRef pad_nb, pad_il, pad_dv, pad_rm; BitRef pad_flt, pad_isEq, pad_isLast;
    Cur c_loop;                       // first IsEqual of Pad's isEq loop; then m isLast IsEquals, m Num2Bits(8), Flatten(m,8)
    Cur c_div;                        // Pad's Divide(16) block: [out, rem | a, b] ...  (divide.circom:17-33)
    BitRef k_out, k_in; SmRef k_blocks; BitRef k_finalState, f_out, f_in; SmRef f_blocks; BitRef f_s;
    uint32_t abs_w, abs_b;
    BitRef sel_out, sel_arrays; SmRef sel_select; BitRef sel_T;
    Cur c_post;                       // Reshape(32,8) [out | in] then Bits2Num(8) x 32 [out | in[8]]
    SmRef dst; uint32_t has_dst;      // the parent's copy of out[32]
    uint32_t mb, index;
};
// SubstringCheck instance (substring_check.circom:24-100)
struct ScRefs {
    BitRef out; uint32_t mi_w; SmRef ml, si; FrRef num; uint32_t M_w; BitRef ex, isl, alw; SmRef sums; BitRef dne;     // mainInput[] / M[]: derived wires
    Cur c_abs_sub, c_abs_main, c_after_abs, c_loop, c_tail;
    SmRef abs_main_in;
};
// RlpMerklePatriciaTrieLeaf(32, amountBytes) of proof_of_burn.circom:198 and the nested templates whose wires are split over units
struct RlRefs {
    SmRef o, ol, in, inl; FrRef bal; SmRef key, keyLen, acc, accLen, pk, pkLen, val, valLen;      // own (:102-189)
    SmRef t_o, t_ol, t_in, t_il, t_dv, t_rm, t_shf, t_on, t_tmp;                                     // TruncatedAddressHash own (:50-90)
    SmRef s_o, s_in, s_cn; BitRef s_isEq; SmRef s_temp;                                              // ShiftLeft(64) own (shift.circom:17-37)
    Cur c_sl_iseq, c_mux, c_age, c_acc, c_concat;
};
// LeafDetector(N) instance (rlp/merkle_patricia_trie_leaf.circom:247-294), cut into head / selector ranges / tail
struct LdRefs {
    BitRef isLeaf; SmRef layer, ll;
    BitRef leafPrefixIsF8; SmRef totalLength; BitRef isConsistentWithLayerLen; SmRef keyPrefix; BitRef keyPrefixIsValid, keyIsMultiByte;
    SmRef keyExtraLen, keyLen, valueWrapperPrefix; BitRef valueWrapperPrefixIsB8; SmRef valueWrapperLen, valuePrefix; BitRef valuePrefixIsF8;
    SmRef valueLen; BitRef isValueWrapperLenConsistent, isKeyValueLenEqualWithLayerLen;
    Cur c_sel[4], c_eq[4], c_mand, c_end;
    SmRef src, len_src; BitRef dst;
};
// RlpEmptyAccount(mb) (rlp/empty_account.circom:20-134) and its RlpInteger(mb) (rlp/integer.circom:67-110), cut in three units
struct RaRefs {
    SmRef ea_o, ea_ol; FrRef ea_ib; SmRef pn, pnl, br, brl, nbl, sc;                          // RlpEmptyAccount own
    SmRef ri_o, ri_ol; FrRef ri_i; SmRef by, len, be; BitRef isb, isz; SmRef frb;             // RlpInteger own
    Cur c_cb, c_sl, c_lt, c_concat, c_end;
};
struct SpongeDesc { uint32_t n, stage, src_b, kin_b, fin_b, fs_b, abs_b, kin_w, fin_w, fs_w, abs_w, src_w; };
struct UnitDesc { uint32_t kind, stage; Cur cur; uint32_t a[8]; uint32_t cost, flags; };

#define SC_RANGE_POS 32          // positions of SubstringCheck's existence loop per unit (one batch inversion each; <= 64: the zero flags are one 64-bit word).  64 measured: generation 1.07 -> 1.00 ms alone, evaluation 0.40 -> 0.57 ms
#define MAX_KB 72
#define MAX_SC 64
// everything a unit body needs besides the policy; lives in device memory, read-only on the device
struct CircuitLayout {
    int circuit;                  // 0 = ProofOfBurn, 1 = Spend, 2 = a gadget-level main (gm)
    struct { uint32_t tid, nfr_in, nsm_in, nout; } gm;
    uint32_t decl_order;          // policy.hpp POB_DECL_ORDER: the numbering variant this layout was planned with (the device policies follow it)
    PobParams pob; SpendParams spend;
    PobMain pm; SpendMain sm;
    Fr prefix[3];                 // POSEIDON_PREFIX + 0/1/2 (constants.circom:3-14), Montgomery
    struct { SmRef nibbles; FrRef in; SmRef addressBytes, block, hash; uint32_t kb; } bah;
    struct { FrRef out; SmRef in, flat, block, hash, reduced; uint32_t kb; int N, nb; SmRef fl_o, fl_i, fit_o, fit_i; Cur c_abs; } pc;      // (fl_* / fit_*: Flatten's and Fit's own [out | in]; c_abs: the first AssertByteString(32) block)
    struct { FrRef in; SmRef mzb, keyBytes, raBytes, becBytes, eip, hin, block, keccak; BitRef sbz; uint32_t kb; } pw;
    struct { SmRef out; uint32_t arr_w, sel_w, T_w; Cur c_sel0; } ll;    // SelectorArray1D(L, 136*NB) of :142 (arrays / select / arraysT: derived wires)
    RlRefs rl;
    RaRefs ra;
    uint32_t kb_hdr, kb_layer0, nkb, nsc;
    Cur fp_n2be32, fp_n2beN;       // footprints of Num2BigEndianBytes(32) / (amountBytes): the evaluator's composite units step over these blocks
    KBRefs kbs[MAX_KB];
    ScRefs scs[MAX_SC];
    LdRefs lds[MAX_SC + 1];
};

HD PosOff pos_off(int t) {
    PosOff k;
    if (t == 3) { k.C = POS_OFF_C_3; k.S = POS_OFF_S_3; k.M = POS_OFF_M_3; k.Pm = POS_OFF_P_3; k.rp = POS_RP_3; }
    else if (t == 4) { k.C = POS_OFF_C_4; k.S = POS_OFF_S_4; k.M = POS_OFF_M_4; k.Pm = POS_OFF_P_4; k.rp = POS_RP_4; }
    else { k.C = POS_OFF_C_5; k.S = POS_OFF_S_5; k.M = POS_OFF_M_5; k.Pm = POS_OFF_P_5; k.rp = POS_RP_5; }
    return k;
}

// ---------------------------------------------------------------------------- keccak.circom: Pad / KeccakBytes
// KeccakBytes(mb) :454-489 up to Pad's loops:
// [out[32] | in[m], inLen | padded[m], numBlocks, inBitsArray[m][8], inBits[8m], inBlocks[mb][17][64], outBits[256], outBytes[32][8]]
// || AssertLessThan(16)(inLen, m), Pad(mb,136) :412-446 [out[m], numBlocks | in[m], inLen | div, rem, filter[m+1], isEq[m], isLast[m]]
//    || Divide(16)(inLen,136), AssertLessEqThan(16)(numBlocks, mb), IsEqual([i,inLen]) x m, IsEqual([i,numBlocks*136-1]) x m
// || Num2Bits(8) x m, Flatten(m,8), Keccak(mb), Reshape(32,8), Bits2Num(8) x 32
#define NO_RANK 0xFFFFFFFFu
template <class P> HD BitRef dbits(P& p, uint32_t n) { BitRef r = {p.dvs(n), NO_RANK}; return r; }      // n DERIVED BIT wires: wire indices only
template <class P> GD void kb_head(P& p, int mb, S inLen, KBRefs& r) {
    const uint32_t m = 136 * mb;
    r.mb = mb;
    r.out = p.sms(32); r.in = p.sms(m); r.inLen = p.sms(1); r.padded_w = p.dvs(m); r.numBlocks = p.sms(1);
    // the padded bytes' bits exist six times in the circuit (Num2Bits(8).out, inBitsArray, Flatten in / out, inBits, inBlocks): stored ONCE, as inBlocks (the sponge's
    // source); the hash bits ten times (Selector.out, SelectorArray2D.out, Final.out, Keccak.finalState / out, outBits, Reshape in / out, outBytes, Bits2Num.in): stored
    // once, as Selector.out.  The copies are derived wires (policy.hpp run_derived).
    r.inBitsArray = dbits(p, 8 * m); r.inBits = dbits(p, 8 * m); r.inBlocks = p.bits(8 * m); r.outBits = dbits(p, 256); r.outBytes = dbits(p, 256);
    inLen = p.put(r.inLen, inLen);
    gAssertLessThanS(p, 16, inLen, (S)m);
    r.pad_o_w = p.dvs(m); r.pad_nb = p.sms(1); r.pad_in_w = p.dvs(m); r.pad_il = p.sms(1); r.pad_dv = p.sms(1); r.pad_rm = p.sms(1);
    r.pad_flt = p.bits(m + 1); r.pad_isEq = p.bits(m); r.pad_isLast = p.bits(m);
    inLen = p.put(r.pad_il, inLen);
    S q, rem;
    r.c_div = p.cur;
    gDivide(p, 16, inLen, (S)136, q, rem);
    q = p.put(r.pad_dv, q); p.put(r.pad_rm, rem);
    S nb = p.put(r.pad_nb, q + 1);
    gAssertLessEqThanS(p, 16, nb, (S)mb);
    p.put(r.numBlocks, nb);
    p.put(r.pad_flt, ~(B)0);
    r.c_loop = p.cur;
    p.cur = cur_add(cur_add(cur_add(p.cur, FP_ISEQ_D, 2 * m), FP_N2B8, m), Cur{16u * m, 0, 0, 0, 16u * m}, 1);   // -> Keccak(mb)   (Flatten(m, 8) [out | in]: derived)
}
// the four derived operand wires of an IsEqual([a, b]) child at cursor c (its two BIT wires are written by the caller as part of a run)
template <class P> HD void iseq_derived(P& p, Cur c, S a, S b) {
    p.derived(c.w + 1, a); p.derived(c.w + 2, b);
    const S x = (S)((uint32_t)b - (uint32_t)a);
    p.derived(c.w + 4, x); p.derived_inv(c.w + 5, x, true);
}
template <class P> GD void kb_range(P& p, const KBRefs& r, SmRef src, uint32_t lo, uint32_t hi, bool assert_src = false, SmRef also = SmRef{0, 0}) {
    const uint32_t m = 136 * r.mb, cnt = hi - lo, ln = p.lane_id();     // cnt <= 16 bytes
    const S inLen = p.get(r.inLen), nb = p.get(r.numBlocks);
    const S last = (S)((uint32_t)nb * 136u - 1u);
    const Cur cE = r.c_loop, cL = cur_add(cE, FP_ISEQ_D, m), cN = cur_add(cL, FP_ISEQ_D, m), cF = cur_add(cN, FP_N2B8, m);
    const uint32_t flat_o_w = cF.w, flat_i_w = cF.w + 8 * m;
    // filter[i] = prod_{j<i}(1 - isEq[j]) = [inLen >= i] (unsigned: an out-of-range inLen never hits); the evaluator re-reads it.
    // Every bit of the range is first THIS witness' bit of a per-lane word (isEq / isLast / filter: bit t = byte lo + t; the byte's own bits: eight per byte), and the words
    // become the lane-distributed runs of the wires by bit-matrix transpositions across the wavefront (policy.hpp xpose64) -- rounds 1-5 built every run from one ballot and
    // two selects under a `lane == k` mask per wire: 40 SGPR pairs per byte.
    bool fl = P::is_gen ? ((uint32_t)inLen >= lo) : p.bit(p.get(r.pad_flt + lo));
    uint32_t wE = 0, wL = 0, wF = 0, wPE = 0, wPL = 0;
    B wb0 = 0, wb1 = 0;
    // the range's loads first, all of them in flight at once (round 5: as `put(in + i, get(src + i))` inside the loop every byte paid its own memory round trip -- two in
    // the evaluator --, 17 us per wavefront for 32 dependent loads; the unit kind is a sixth of the G side's wave cycles in generation and in evaluation)
    S vs[16];
    if constexpr (P::is_count) { for (uint32_t t = 0; t < cnt; t++) p.put(r.in + (lo + t), (S)0); for (int q = 0; q < 16; q++) vs[q] = 0; }
    else {
        SmRef rr[16]; S sv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) rr[q] = r.in + (lo + ((uint32_t)q < cnt ? (uint32_t)q : cnt - 1));      // (a short last range repeats its last byte: same wire, same value)
        const SmLoaded<16> h = sm_load(p, rr);
#pragma unroll
        for (int q = 0; q < 16; q++) sv[q] = p.get(src + (lo + ((uint32_t)q < cnt ? (uint32_t)q : cnt - 1)));
        sm_commit(p, rr, h, sv);
#pragma unroll
        for (int q = 0; q < 16; q++) vs[q] = P::is_check ? h.s[q] : sv[q];        // the evaluator's later expressions see the STORED in[i]
        // assert_src: the source array has an AssertByteString of its own in the circuit (the layers, the header: proof_of_burn.circom:102,106 and SubstringCheck's
        // assert on the same layer bytes, substring_check.circom:37) whose wires are all derived: generation and evaluation hold its assert HERE, on the bytes this
        // range has just loaded -- as range units of their own (rounds 1-5: 595 of them) they read every layer row a second and a third time
        if (assert_src) {
#pragma unroll
            for (int q = 0; q < 16; q++) p.require_lane((uint32_t)sv[q] < 256u, FAILCODE(T_NUM2BITS, 38));
        }
        // also: another template's copy of the same source bytes (LeafDetector.layer[] <== the layer, merkle_patricia_trie_leaf.circom:248), written / compared from the
        // values this range holds -- one read of the layer row serves KeccakBytes.in, both AssertByteStrings and the leaf detector's copy
        if (also.w) {
            SmRef ar[16];
#pragma unroll
            for (int q = 0; q < 16; q++) ar[q] = also + (lo + ((uint32_t)q < cnt ? (uint32_t)q : cnt - 1));
            const SmLoaded<16> ha = sm_load(p, ar);
            sm_commit(p, ar, ha, sv);
        }
    }
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) if (t < cnt) {
        const uint32_t i = lo + t;
        const Cur ce = cur_add(cE, FP_ISEQ_D, i), cl = cur_add(cL, FP_ISEQ_D, i);
        const S v = vs[t];
        iseq_derived(p, ce, (S)i, inLen); iseq_derived(p, cl, (S)i, last);      // IsEqual([i, inLen]), IsEqual([i, numBlocks*136 - 1]): operand wires derived
        const bool e = (uint32_t)inLen == i, l = (uint32_t)last == i;
        fl = fl && !e;
        const S pv = (fl ? v : 0) + (S)e + (l ? 0x80 : 0);
        p.derived(r.pad_in_w + i, v); p.derived(r.pad_o_w + i, pv); p.derived(r.padded_w + i, pv); p.derived(cN.w + 9 * i + 8, pv);   // (derived copies)
        p.require_lane((uint32_t)pv < 256u, FAILCODE(T_NUM2BITS, 38));
        wE |= (uint32_t)e << t; wL |= (uint32_t)l << t; wF |= (uint32_t)fl << t;
        wPE |= (e ? 3u : 0u) << (2 * t); wPL |= (l ? 3u : 0u) << (2 * t);
        if (t < 8) wb0 |= (B)((uint32_t)pv & 0xffu) << (8 * t); else wb1 |= (B)((uint32_t)pv & 0xffu) << (8 * (t - 8));
    }
    const B runE = p.xpose64(wE, 0, cnt), runL = p.xpose64(wL, 0, cnt), runF = p.xpose64(wF, 0, cnt);
    const B runPairE = p.xpose64(wPE, 0, 2 * cnt), runPairL = p.xpose64(wPL, 0, 2 * cnt);
    const B bits0 = p.xpose(wb0, 64), bits1 = cnt > 8 ? p.xpose(wb1, 64) : 0;
    p.run_put(cnt, r.pad_isEq.w + lo + ln, r.pad_isEq.i + lo + ln, runE);
    p.run_put(cnt, r.pad_flt.w + lo + 1 + ln, r.pad_flt.i + lo + 1 + ln, runF);
    p.run_put(cnt, r.pad_isLast.w + lo + ln, r.pad_isLast.i + lo + ln, runL);
    {   // the IsEqual children's [IsEqual.out, IsZero.out] pairs: copies of isEq[i] / isLast[i]
        const uint32_t i = lo + (ln >> 1), wh = ln & 1;
        p.run_derived(2 * cnt, cE.w + 6 * i + 3 * wh, runPairE);
        p.run_derived(2 * cnt, cL.w + 6 * i + 3 * wh, runPairL);
        if constexpr (P::is_emit) { if (ln < 2 * cnt) { p.site_c(cE.w + 6 * i + 3 * wh, r.pad_isEq.w + i); p.site_c(cL.w + 6 * i + 3 * wh, r.pad_isLast.w + i); } }      // (self-check: isEq[i] <== IsEqual.out)
    }
    for (uint32_t h2 = 0; h2 < 2 && 8 * h2 < cnt; h2++) {      // Keccak's inBlocks (stored) and its five copies: Num2Bits(8).out, inBitsArray, Flatten in/out, inBits
        const uint32_t n = (cnt - 8 * h2 < 8 ? cnt - 8 * h2 : 8) * 8;
        const B x = h2 ? bits1 : bits0;
        const uint32_t j = 8 * (lo + 8 * h2) + ln, i = lo + 8 * h2 + (ln >> 3);
        p.run_put(n, r.inBlocks.w + j, r.inBlocks.i + j, x);
        p.run_derived(n, cN.w + 9 * i + (ln & 7), x);
        p.run_derived(n, r.inBitsArray.w + j, x);
        p.run_derived(n, flat_i_w + j, x);
        p.run_derived(n, flat_o_w + j, x);
        p.run_derived(n, r.inBits.w + j, x);
    }
}
// Keccak(n) :374-385 / Final(n) :330-349 own wires + the n Absorb blocks (K kernels) + SelectorArray2D own wires.
template <class P> GD void kb_declare_keccak(P& p, KBRefs& r) {
    const uint32_t n = r.mb;
    r.k_out = dbits(p, 256); r.k_in = p.bits(n * 1088); r.k_blocks = p.sms(1); r.k_finalState = dbits(p, 1600);
    r.f_out = dbits(p, 1600); r.f_in = p.bits(n * 1088); r.f_blocks = p.sms(1); r.f_s = p.bits((n + 1) * 1600);
    r.abs_w = p.cur.w; r.abs_b = p.cur.b;
    p.skip_alias(n * ABSORB_WIRES, n * ABSORB_BITS);
    r.sel_out = dbits(p, 1600); r.sel_arrays = dbits(p, (n + 1) * 1600); r.sel_select = p.sms(1); r.sel_T = dbits(p, 1600 * (n + 1));      // (copies of Selector.out / of Final.s: derived)
}
// selectors [j0, j1) of row `row` of SelectorArray2D(n+1, 25, 64) (selector.circom:91-111) + the copies of its outputs;
// the hash output (first 256 selector outputs) also flows through outBits, Reshape, outBytes, Bits2Num, out[] and the parent's copy.
// BIT wires are handled lane-distributed: lane k = selector j0 + k, value = that wire's 64-witness mask, so the selection itself is
// plain bit-sliced logic (out = OR_k isEq_k & state_k) and a run of wires is one coalesced access.  The SM wires of the IsEqual
// children (the same four per-witness rows for every selector) stay lane = witness.
// Selector(n1) block on BIT data (selector.circom:21-46):  [out | vals[n1], select | isEq[n1], sum[n1+1]] || IsEqual x n1,
// IsEqual (comparators.circom): [out | in[2]] || IsZero [out | in | inv]
template <class P, int N1> GD void kb_selrow_n(P& p, const KBRefs& r, uint32_t row, uint32_t j0, uint32_t j1) {
    const uint32_t n1 = N1 ? (uint32_t)N1 : r.mb + 1, n = j1 - j0, ln = p.lane_id();
    const uint32_t fw = 9 * n1 + 3, fb = 1, fq = 9 * n1 + 2;                // footprint of one selector: wires, stored BIT (Selector.out), derived (no SM)
    const Cur c0 = p.cur;
    const uint32_t idx = row * 64 + j0 + ln;
    const S blocks = p.get(r.numBlocks);
    p.require(p.ballot((uint32_t)blocks < n1), FAILCODE(T_SELECTOR, 43));
    const uint32_t bw = c0.w + ln * fw, bb = c0.b + ln * fb;                // this lane's selector block
    // Of a selector's 9 n1 + 3 wires ONE is stored: its output (round 4).  vals[] / SelectorArray2D.arrays / arraysT are copies of Final.s, isEq[] and the
    // IsEqual children's outputs functions of numBlocks, sum[] the running OR of isEq & vals: derived wires, rebuilt by the emitter from the same expressions
    // (rounds 1-3 stored all 5 n1 + 2 bits: 1.36 M of the G side's 2.59 M stored BIT wires and a fifth of its wave cycles in generation and evaluation).
    B acc = 0;
    p.run_derived(n, bw + 2 * n1 + 2, 0);                                   // sum[0]
#pragma unroll
    for (uint32_t k = 0; k < n1; k++) {
        const B v = p.run_get(n, r.f_s.i + k * 1600 + idx);
        const B e = p.ballot((uint32_t)blocks == k);
        acc |= e & v;
        if constexpr (P::is_emit) {
            p.run_derived(n, r.sel_arrays.w + k * 1600 + idx, v);
            p.run_derived(n, r.sel_T.w + idx * n1 + k, v);
            p.run_derived(n, bw + 1 + k, v);                                // vals[k]
            p.run_derived(n, bw + n1 + 2 + k, e);                           // isEq[k]
            p.run_derived(n, bw + 2 * n1 + 3 + k, acc);                     // sum[k+1]
            const uint32_t cw = bw + 3 * n1 + 3 + 6 * k;
            p.run_derived(n, cw, e);                                        // IsEqual.out
            p.run_derived(n, cw + 3, e);                                    // IsZero.out
        }
    }
    p.run_put(n, bw, bb, acc);                                              // Selector.out: the stored copy (the evaluator checks it against OR_k isEq_k & Final.s_k)
    p.run_derived(n, r.sel_out.w + idx, acc);
    p.run_derived(n, r.f_out.w + idx, acc);
    p.run_derived(n, r.k_finalState.w + idx, acc);
    if (row < 4) {           // the 256 hash bits: Keccak.out, KeccakBytes.outBits, Reshape in/out, outBytes, Bits2Num(8).in -- derived copies
        const uint32_t pw = r.c_post.w;
        p.run_derived(n, r.k_out.w + idx, acc);
        p.run_derived(n, r.outBits.w + idx, acc);
        p.run_derived(n, pw + 256 + idx, acc);
        p.run_derived(n, pw + idx, acc);
        p.run_derived(n, r.outBytes.w + idx, acc);
        p.run_derived(n, pw + 512 + 9 * (idx >> 3) + 1 + (idx & 7), acc);
        for (uint32_t jj = 0; jj < n; jj += 8) {                            // Bits2Num(8) outputs, per witness
            S by = 0;
#pragma unroll
            for (uint32_t b = 0; b < 8; b++) by |= (S)p.bit(p.run_bcast(acc, jj + b)) << b;
            const uint32_t i = (row * 64 + j0 + jj) >> 3;
            const SmRef b2n_o = {pw + 512 + 9 * i, r.c_post.s + i};
            const S o = p.put(r.out + i, p.put(b2n_o, by));
            if (r.has_dst) p.put(r.dst + i, o);
        }
    }
    // The non-BIT side of the n selectors -- select, and per IsEqual child in[0] = select, in[1] = k, IsZero.in = k - select, IsZero.inv: the same four
    // per-witness values for every selector, all functions of numBlocks -- are DERIVED wires (policy.hpp): only the emitter writes them.
    derived_rows_same(p, c0.w + n1 + 1, fw, n, blocks);
    for (uint32_t k = 0; k < n1; k++) {
        const uint32_t cw = c0.w + 3 * n1 + 3 + 6 * k;
        const S x = (S)(k - (uint32_t)blocks);
        derived_rows_same(p, cw + 1, fw, n, blocks);
        derived_rows_same(p, cw + 2, fw, n, (S)k);
        derived_rows_same(p, cw + 4, fw, n, x);
        derived_rows_same(p, cw + 5, fw, n, x, true);
    }
    p.cur = cur_add(c0, Cur{fw, fb, 0, 0, fq}, n);
}
template <class P> GD void kb_selrow(P& p, const KBRefs& r, uint32_t row, uint32_t j0, uint32_t j1) { kb_selrow_n<P, 0>(p, r, row, j0, j1); }
// after the selectors: Keccak/Final/selector `blocks` inputs; the Reshape(32,8) + Bits2Num(8) x 32 blocks that follow in wire
// order (512 + 32*9 wires) are written by the selector-row units of rows 0..3 (kb_out_bit)
template <class P> GD void kb_post(P& p, const KBRefs& r) {
    S nb = p.get(r.numBlocks);
    p.put(r.k_blocks, nb); p.put(r.f_blocks, nb); p.put(r.sel_select, nb);
    p.cur = cur_add(r.c_post, Cur{512 + 32 * 9, 0, 32, 0, 512 + 32 * 8}, 1);       // (Reshape(32,8) [out | in] and the Bits2Num(8) inputs: derived copies of the hash bits)
}

#include "gadget_mains.hpp"

// U_RL_B's second and third part (merkle_patricia_trie_leaf.circom:166-188): the value / key prefixes and their copies | Concat + the leaf copies.  Generation and emission
// run them inside U_RL_B; the evaluator as units of their own from the stored lengths.
template <class P> GD void rl_b_prefixes(P& p, const RlRefs& R, const PobParams& prm, S kl, S& pl, S& vl) {
    const int ab = 32, bb = prm.amountBytes, maxAcc = 4 + bb + 66, maxKey = 1 + ab;
    const S al = p.get(R.accLen);
    p.put(R.val, 0xb8); p.put(R.val + 1, al);
    copy_n(p, R.val + 2, R.acc, maxAcc);
    vl = p.put(R.valLen, 2 + al);
    p.put(R.pk, 0xf8); p.put(R.pk + 1, (kl + 1) + vl); p.put(R.pk + 2, 0x80 + kl);
    copy_n(p, R.pk + 3, R.key, maxKey);
    pl = p.put(R.pkLen, 3 + kl);
}
template <class P, class MR> GD void rl_b_concat(P& p, const RlRefs& R, const MR& M, const PobParams& prm, S pl, S vl) {
    const int ab = 32, bb = prm.amountBytes, maxAcc = 4 + bb + 66, maxVal = 2 + maxAcc, maxKey = 1 + ab, maxPK = 2 + 1 + maxKey, maxOut = maxPK + maxVal;
    p.cur = R.c_concat;
    S cl;
    SmRef c;
    if constexpr (P::is_check) c = gConcatHead(p, maxPK, maxVal, R.pk, pl, R.val, vl, cl);          // (its children: CK_CAT units)
    else c = gConcat(p, maxPK, maxVal, R.pk, pl, R.val, vl, cl, true);
    { copy_n(p, R.o, c, (int)(maxOut)); copy_n(p, M.leaf, c, (int)(maxOut)); }
    p.put(M.leafLen, p.put(R.ol, cl));
}

// PublicCommitment(N) between its own wires (declared by the planner) and KeccakBytes: AssertByteString(32) x N, Flatten(N, 32) [out | in], Fit(32 N, 136 nb) [out | in] -- the
// places of their wires (the bytes are written by U_PC_COPY units)
template <class P> GD void L_pc_declare(P& p, CircuitLayout& L, int N) {
    const Cur c_abs = p.cur;
    for (int j = 0; j < N; j++) { p.dvs(32); p.cur = cur_add(p.cur, FP_ABITS8, 32); }
    const SmRef fl_o = p.sms(32 * N), fl_i = p.sms(32 * N);
    const SmRef fit_o = p.sms(136 * L.pc.nb), fit_i = p.sms(32 * N);
    if (P::is_count) { L.pc.c_abs = c_abs; L.pc.fl_o = fl_o; L.pc.fl_i = fl_i; L.pc.fit_o = fit_o; L.pc.fit_i = fit_i; }
}

// ---------------------------------------------------------------------------- unit bodies
// ONE switch over every unit kind; a kernel instantiates it with the MASK of the families it serves and the other cases
// compile to nothing.  LIGHT families touch only BIT/SM wires (few VGPRs -> 8 waves/SIMD, which is what hides the load latency of
// this lane-per-witness code); F_SC / F_POS / F_N2B do BN254 arithmetic.
HD bool unit_is_heavy(uint32_t k) { return fam_of(k) >= F_SC; }
HD bool unit_gen_is_heavy(uint32_t k) { return gen_fam_of(k) >= F_SC; }

template <class P, uint32_t MASK> GD void unit_run(P& p, const UnitDesc& d, CircuitLayout& L);
#ifdef __HIPCC__
// a riding generation kernel (policy.hpp GenPT<true>) runs the RLP family's units on the plain policy, and their evaluation stays a launch of pob_constraint_check
// (ride_keeps_evaluation: what the host asks).  One of them -- U_RL_A, the head of RlpMerklePatriciaTrieLeaf -- spills 179 VGPRs with the pending compares in registers, the only unit
// kind that does; with the other four riding and only U_RL_A's evaluation left the kernel has no spill either, but the RLP units are the generation's longest serial chains: a lone
// generation 2.10 -> 2.58 ms, the loop with 8 / 16 in flight 1.06 -> 1.10 / 1.06 -> 1.10 ms, with 12 the same (profiles/round6_experiments.txt 22): the family stays plain.
template <uint32_t MASK, bool FAULT> GD void unit_run_ride(GenPT<true, FAULT>& p, const UnitDesc& d, CircuitLayout& L) {
    if constexpr ((MASK >> F_RL) & 1u) { if (ride_keeps_evaluation(d.kind)) { GenPT<false> q = plain_of(p); unit_run<GenPT<false>, FAM_BIT(F_RL)>(q, d, L); p.status = q.status; return; } }
    unit_run<GenPT<true, FAULT>, MASK>(p, d, L);
}
#endif
template <class P, uint32_t MASK> GD void unit_run(P& p, const UnitDesc& d, CircuitLayout& L) {

    const PobMain& M = L.pm;
    const PobParams& prm = L.pob;
    const int LB = 136 * prm.NB, HBy = 136 * prm.HB;
    (void)M; (void)LB; (void)HBy;
    p.cur = d.cur;
    switch (d.kind) {
    UCASE(U_GM) { gm_run(p, d, L); } break;
    UCASE(U_GM_INPUT) { gm_input(p, d); } break;
    UCASE(U_POB_INPUT) {   // SM main inputs [a0, a1) from the packed batch buffer (declaration order = contiguous SM ranks)
        for (uint32_t k = d.a[0]; k < d.a[1]; k++) { SmRef r = {M.numLeafAddressNibbles.w + k, M.numLeafAddressNibbles.i + k}; p.put(r, p.input_sm(k)); }
    } break;
    UCASE(U_POB_LAYER_ASSERT) {   // :101 (the AssertByteString of :102 runs as U_ABS_RANGE units)
        gAssertLessThanS(p, 16, p.get(M.layerLens + d.a[0]), (S)(LB * 8));
    } break;
    UCASE(U_POB_HDR_ASSERT) {     // :105, stateRoot copy :125-129
        gAssertLessThanS(p, 16, p.get(M.blockHeaderLen), (S)(HBy * 8));
        for (int i = 0; i < 32; i++) p.put(M.stateRoot + i, p.get(M.blockHeader + 91 + i));
    } break;
    UCASE(U_ABS_RANGE) {          // cur = first AssertBits(8) child; a = first wire of the own in[], -, src (w,i), lo, hi
        SmRef src = {d.a[2], d.a[3]};
        abs_range(p, d.cur, d.a[0], src, d.a[4], d.a[5]);
    } break;
    UCASE(U_BAH_POST) {           // :82 Bytes2Nibbles(32) + main.addressHashNibbles (:119)
        SmRef nb = gBytes2Nibbles(p, 32, L.bah.hash);
        { copy_n(p, L.bah.nibbles, nb, (int)(64)); copy_n(p, M.addressHashNibbles, nb, (int)(64)); }
    } break;
    UCASE(U_KB_HEAD) {            // a[0] = kb index, a[1..2] = inLen ref
        SmRef len = {d.a[1], d.a[2]};
        KBRefs r = L.kbs[d.a[0]];
        // a[3] = 1 + index of the length among the packed SM inputs: generation then reads the input buffer, so the head can run in
        // stage 0 beside the input units that write the wire (the evaluator reads the wire)
        const S inLen = (P::is_gen && d.a[3]) ? p.input_sm(d.a[3] - 1) : p.get(len);
        kb_head(p, r.mb, inLen, r);
        if (P::is_count) L.kbs[d.a[0]] = r;
    } break;
    UCASE(U_KB_RANGE) {           // a[0] = kb index, a[1..2] = src ref, a[3..4] = byte range
        SmRef src = {d.a[1], d.a[2]};
        kb_range(p, L.kbs[d.a[0]], src, d.a[3], d.a[4], d.a[5] != 0, SmRef{d.a[6], d.a[7]});
    } break;
    UCASE(U_KB_SELROW) { kb_selrow(p, L.kbs[d.a[0]], d.a[1], d.a[2], d.a[3]); } break;
    UCASE(U_KB_POST) { kb_post(p, L.kbs[d.a[0]]); } break;
    UCASE(U_PC_PRE) {             // PublicCommitment(N) public_commitment.circom:18-36 up to the sponge: the wires' places and KeccakBytes' head; the BYTES travel in U_PC_COPY units
        const int N = L.pc.N;
        L_pc_declare(p, L, N);
        KBRefs r = L.kbs[L.pc.kb];
        kb_head(p, L.pc.nb, (S)(32 * N), r);
        if (P::is_count) L.kbs[L.pc.kb] = r;
    } break;
    UCASE(U_PC_COPY) {            // the 32 bytes of input a0: in[a0][] <== the caller's bytes, AssertByteString(32), Flatten [out | in], flat, Fit [out | in], block -- seven copies of the same
                                 // values, requested together (as one serial unit the head was 370 dependent memory round trips: 0.17 ms, a generation level of its own)
        const uint32_t N = (uint32_t)L.pc.N, j = d.a[0], nblk = 136u * (uint32_t)L.pc.nb;
        if (j < N) {
            SmRef src;
            if (L.circuit == 2) src = L.pc.in + 32 * j;      // PublicCommitment(N) as the main: in[][] holds the packed inputs already (U_GM_INPUT)
            else if (L.circuit == 0) src = j == 0 ? M.blockRoot : j == 1 ? M.nullifierBytes : j == 2 ? M.remainingCoinBytes : j == 3 ? M.revealAmountBytes : j == 4 ? M.burnExtraCommitmentBytes : M.extraCommitmentBytes;
            else src = j == 0 ? L.sm.coinBytes : j == 1 ? L.sm.withdrawnBalanceBytes : j == 2 ? L.sm.remainingCoinBytes : L.sm.extraCommitmentBytes;
            auto dst_of = [&](uint32_t a) -> SmRef {       // (a uniform select chain: an array of references indexed by the loop counter would live in scratch)
                const SmRef b = a == 0 ? L.pc.in : a == 1 ? L.pc.fl_i : a == 2 ? L.pc.fl_o : a == 3 ? L.pc.flat : a == 4 ? L.pc.fit_i : a == 5 ? L.pc.fit_o : L.pc.block;
                return b + 32 * j;
            };
#pragma unroll 1
            for (uint32_t h2 = 0; h2 < 2; h2++) {
                S v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = p.get(src + (16 * h2 + q));
#pragma unroll 1
                for (uint32_t a = 0; a < 7; a++) {
                    const SmRef dst = dst_of(a);
                    SmRef rr[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) rr[q] = dst + (16 * h2 + q);
                    const SmLoaded<16> h = sm_load(p, rr);
                    sm_commit(p, rr, h, v);
                }
            }
            // AssertByteString(32)(in[j]): [ | in[32]] || AssertBits(8) x 32 -- derived wires, functions of the bytes
            const Cur fp = {32u + 32u * 18u, 0, 0, 0, 32u + 32u * 18u};
            const Cur cj = cur_add(L.pc.c_abs, fp, j);
            abs_range(p, Cur{cj.w + 32, cj.b, cj.s, cj.f, cj.q + 32}, cj.w, L.pc.in + 32 * j, 0, 32);
        } else {
            for (uint32_t i = 32 * N; i < nblk; i++) { p.put(L.pc.fit_o + i, 0); p.put(L.pc.block + i, 0); }      // Fit(32 N, 136 nb) pads with zeros
        }
    } break;
    UCASE(U_POB_LASTLAYER) {      // :142-143 SelectorArray1D(L, LB): the select input; selectors run as range units
        p.derived(L.ll.sel_w, p.get(M.numLayers) - 1);
    } break;
    UCASE(U_POB_LASTLAYER_RANGE) {   // selectors [a0, a1) of SelectorArray1D (selector.circom:62-77): selector j reads column j of layers[][] directly
        const uint32_t n = prm.L, q = LB;
        const Cur fp = sel_fp(n);
        S select = p.get(M.numLayers) - 1;
        for (uint32_t j = d.a[0]; j < d.a[1]; j++) {
            if constexpr (P::is_emit) {               // arrays[i][j] and arraysT[j][i]: derived copies of layers[i][j]
                for (uint32_t i = 0; i < n; i++) { const S v = p.get(M.layers + (i * q + j)); p.derived(L.ll.arr_w + i * q + j, v); p.derived(L.ll.T_w + j * n + i, v); }
            }
            p.cur = cur_add(L.ll.c_sel0, fp, j);
            S v = p.put(L.ll.out + j, gSelectorS(p, n, M.layers + j, select, q));
            p.put(M.lastLayer + j, v);
        }
    } break;
    UCASE(U_POB_LASTLEN) {        // :146, :150
        S nl = p.get(M.numLayers);
        p.put(M.lastLayerLen, gSelectorS(p, prm.L, M.layerLens, nl - 1));
        BitRef f = gFilter(p, prm.L, nl);
        copy_n(p, M.layerExists, f, (int)(prm.L));
    } break;
    UCASE(U_SC_SUMS) {            // sums[a1+1 .. a2] (:94); the unit of the last range also doesNotExist, out (:98-99), substringCheckers[i-1]
                                 // and the constraint of proof_of_burn.circom:179
        const uint32_t i = d.a[0], lo = d.a[1], hi = d.a[2];
        const ScRefs& sc = L.scs[i];
        const uint32_t kk = LB - 31 + 1, ln = p.lane_id();
        // allowed[] / exists[] come as lane-distributed runs (64 wires per load).  One unit per 64 positions: generation rebuilds the sum entering
        // its range from the runs below it (register work only, no wire is read back), the evaluator starts from the STORED sums[a1] (the
        // relation is local).  As one unit per layer this was 515 dependent stores behind 1 030 loads: 0.12 ms alone, 1-2 ms beside the other
        // batch's round evaluation, at the very end of the generation.
        S sum = 0;
        if (lo == 0) { p.put(sc.alw, ~(B)0); sum = p.put(sc.sums, 0); }
        else if constexpr (P::is_gen) {
            for (uint32_t c0 = 0; c0 * 64 < lo; c0 += 8) {
                B rr[8];
#pragma unroll
                for (uint32_t c = 0; c < 8; c++) {
                    const uint32_t k0 = 64 * (c0 + c), n = k0 < lo ? (lo - k0 < 64 ? lo - k0 : 64) : 0;
                    rr[c] = n ? (p.run_get(n, sc.alw.i + k0 + 1 + ln) & p.run_get(n, sc.ex.i + k0 + ln)) : 0;
                }
#pragma unroll
                for (uint32_t c = 0; c < 8; c++) {
                    const uint32_t k0 = 64 * (c0 + c), n = k0 < lo ? (lo - k0 < 64 ? lo - k0 : 64) : 0;
                    for (uint32_t j = 0; j < n; j++) sum += (S)p.bit(p.run_bcast(rr[c], j));
                }
            }
        } else sum = p.get(sc.sums + lo);
        for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
            const uint32_t n = hi - k0 < 64 ? hi - k0 : 64;
            const B rr = p.run_get(n, sc.alw.i + k0 + 1 + ln) & p.run_get(n, sc.ex.i + k0 + ln);
            for (uint32_t j = 0; j < n; j++) sum = p.put(sc.sums + (k0 + j + 1), sum + (S)p.bit(p.run_bcast(rr, j)));
        }
        if (hi == kk) {
            p.cur = sc.c_tail;
            B none = p.put(sc.dne, gIsZeroS(p, sum));
            B out = p.put(M.substringCheckers + (i - 1), p.put(sc.out, ~none));
            p.require(out | ~p.get(M.layerExists + i), FAILCODE(T_POB, 179));
        }
    } break;
    UCASE(U_LD_HEAD) {            // LeafDetector(N) :247-278 up to keyLen; selector heads (select input, sum[0], range check)
        LdRefs R = L.lds[d.a[0]];
        const uint32_t N = LB;
        R.isLeaf = p.bits(1); R.layer = p.sms(N); R.ll = p.sms(1);
        R.leafPrefixIsF8 = p.bits(1); R.totalLength = p.sms(1); R.isConsistentWithLayerLen = p.bits(1); R.keyPrefix = p.sms(1);
        R.keyPrefixIsValid = p.bits(1); R.keyIsMultiByte = p.bits(1); R.keyExtraLen = p.sms(1); R.keyLen = p.sms(1); R.valueWrapperPrefix = p.sms(1);
        R.valueWrapperPrefixIsB8 = p.bits(1); R.valueWrapperLen = p.sms(1); R.valuePrefix = p.sms(1); R.valuePrefixIsF8 = p.bits(1);
        R.valueLen = p.sms(1); R.isValueWrapperLenConsistent = p.bits(1); R.isKeyValueLenEqualWithLayerLen = p.bits(1);
        // (layer[] <== the source bytes: U_LD_COPY range units)
        const S layerLen = p.put(R.ll, p.get(R.len_src));
        gAssertLessEqThanS(p, 16, layerLen, (S)N);
        p.put(R.leafPrefixIsF8, gIsEqualS(p, p.get(R.src), (S)0xf8));
        const S tl = p.put(R.totalLength, p.get(R.src + 1));
        p.put(R.isConsistentWithLayerLen, gIsEqualS(p, tl + 2, layerLen));
        const S kp = p.put(R.keyPrefix, p.get(R.src + 2));
        p.put(R.keyPrefixIsValid, gLessEqThanS(p, 16, kp, (S)0xb7));
        const B multi = p.put(R.keyIsMultiByte, gIsInRange(p, 16, (S)0x81, kp, (S)0xb7));
        const S kel = p.put(R.keyExtraLen, p.bit(multi) ? kp - 0x80 : 0);
        const S kl = p.put(R.keyLen, 1 + kel);
        const Cur fp = sel_fp(N);
        R.c_sel[0] = p.cur; R.c_eq[0] = cur_add(R.c_sel[0], fp, 1); R.c_sel[1] = cur_add(R.c_eq[0], FP_ISEQ_S, 1); R.c_sel[2] = cur_add(R.c_sel[1], fp, 1);
        R.c_eq[1] = cur_add(R.c_sel[2], fp, 1); R.c_sel[3] = cur_add(R.c_eq[1], FP_ISEQ_S, 1); R.c_eq[2] = cur_add(R.c_sel[3], fp, 1);
        R.c_eq[3] = cur_add(R.c_eq[2], FP_ISEQ_S, 1); R.c_mand = cur_add(R.c_eq[3], FP_ISEQ_S, 1);
        for (uint32_t k = 0; k < 4; k++) {
            const SelBlk sb = sel_blk(R.c_sel[k], N);
            const S select = 2 + kl + (S)k;
            p.derived(sb.sel_w, select); p.derived(sb.sum_w, 0);
            p.require(p.ballot((uint32_t)select < N), FAILCODE(T_SELECTOR, 43));      // sum isEq === 1  <=>  0 <= select < N
        }
        { B m[7] = {0, 0, 0, 0, 0, 0, 0}; p.cur = R.c_mand; CountP q; q.cur = p.cur; q.decl_order = p.decl_order; MultiANDg<CountP, 7>::run(q, m); R.c_end = q.cur; }
        p.cur = R.c_end;
        if (P::is_count) L.lds[d.a[0]] = R;
    } break;
    UCASE(U_LD_COPY) {            // LeafDetector.layer[a1 .. a2) <== layer bytes (merkle_patricia_trie_leaf.circom:248: the template's own input array)
        const LdRefs& R = L.lds[d.a[0]];
        copy_n(p, R.layer + d.a[1], R.src + d.a[1], (int)(d.a[2] - d.a[1]));
    } break;
    UCASE(U_LD_SELR) {            // entries [a2, a3) of selector a1 of LeafDetector a0: bits only (gadgets.hpp sel_range); the last range also writes out
        const LdRefs& R = L.lds[d.a[0]];
        const uint32_t N = LB, which = d.a[1], lo = d.a[2], hi = d.a[3];
        const SelBlk sb = sel_blk(R.c_sel[which], N);
        const S select = 2 + p.get(R.keyLen) + (S)which;
        const uint32_t us = (uint32_t)select;
        S out = 0;
        if (P::is_emit || hi == N) out = us < N ? p.get_lane(R.src, us) : 0;
        sel_range(p, sb, R.src, 1, select, out, lo, hi);
        if (hi == N) {
            const SmRef dst = which == 0 ? R.valueWrapperPrefix : which == 1 ? R.valueWrapperLen : which == 2 ? R.valuePrefix : R.valueLen;
            p.put(dst, p.put(sb.o, out));
        }
    } break;
    UCASE(U_LD_TAIL) {            // :280-293 the four IsEquals after the selectors, MultiAND(7), isLeaf
        const LdRefs& R = L.lds[d.a[0]];
        const S kl = p.get(R.keyLen), ll = p.get(R.ll), vwp = p.get(R.valueWrapperPrefix), vwl = p.get(R.valueWrapperLen), vp = p.get(R.valuePrefix), vl = p.get(R.valueLen);
        B m[7];
        m[0] = p.get(R.leafPrefixIsF8); m[1] = p.get(R.isConsistentWithLayerLen); m[2] = p.get(R.keyPrefixIsValid);
        p.cur = R.c_eq[0]; m[3] = p.put(R.valueWrapperPrefixIsB8, gIsEqualS(p, vwp, (S)0xb8));
        p.cur = R.c_eq[1]; m[5] = p.put(R.valuePrefixIsF8, gIsEqualS(p, vp, (S)0xf8));
        p.cur = R.c_eq[2]; m[4] = p.put(R.isValueWrapperLenConsistent, gIsEqualS(p, vwl, vl + 2));
        p.cur = R.c_eq[3]; m[6] = p.put(R.isKeyValueLenEqualWithLayerLen, gIsEqualS(p, kl + vl + 6, ll));
        p.cur = R.c_mand;
        p.put(R.dst, p.put(R.isLeaf, MultiANDg<P, 7>::run(p, m)));
    } break;
    UCASE(U_RL_A) {               // RlpMerklePatriciaTrieLeaf(32, AB) :102-189, part 1: own inputs, TruncatedAddressHash(32) :50-90 head
                                 // (AssertLessEqThan(7), Divide(7)) and ShiftLeft(64) head (shift.circom:17-24)
        RlRefs R = L.rl;
        const int ab = 32, bb = prm.amountBytes, maxAcc = 4 + bb + 66, maxVal = 2 + maxAcc, maxKey = 1 + ab, maxPK = 2 + 1 + maxKey, maxOut = maxPK + maxVal;
        R.o = p.sms(maxOut); R.ol = p.sms(1); R.in = p.sms(2 * ab); R.inl = p.sms(1); R.bal = p.frs(1);
        R.key = p.sms(maxKey); R.keyLen = p.sms(1); R.acc = p.sms(maxAcc); R.accLen = p.sms(1); R.pk = p.sms(maxPK); R.pkLen = p.sms(1); R.val = p.sms(maxVal); R.valLen = p.sms(1);
        copy_n(p, R.in, M.addressHashNibbles, (int)(2 * ab));
        S nibLen = p.put(R.inl, p.get(M.numLeafAddressNibbles));
        p.put(R.bal, p.get(M.actualBalance));
        const int n2 = 2 * ab;
        R.t_o = p.sms(ab + 1); R.t_ol = p.sms(1); R.t_in = p.sms(n2); R.t_il = p.sms(1); R.t_dv = p.sms(1); R.t_rm = p.sms(1); R.t_shf = p.sms(n2); R.t_on = p.sms(n2 + 2); R.t_tmp = p.sms(n2 - 1);
        copy_n(p, R.t_in, M.addressHashNibbles, (int)(n2));
        nibLen = p.put(R.t_il, nibLen);
        for (int i = 0; i < n2 - 1; i++) p.put(R.t_tmp + i, 0);
        gAssertLessEqThanS(p, 7, nibLen, (S)n2);
        S q, r;
        gDivide(p, 7, nibLen, (S)2, q, r);
        p.put(R.t_dv, q); p.put(R.t_rm, r);
        R.s_o = p.sms(n2); R.s_in = p.sms(n2); R.s_cn = p.sms(1); R.s_isEq = p.bits(n2 * n2); R.s_temp = SmRef{p.dvs(n2 * n2), NO_RANK};      // (temp[][]: derived)
        copy_n(p, R.s_in, M.addressHashNibbles, (int)(n2));
        S count = p.put(R.s_cn, n2 - nibLen);
        gAssertLessEqThanS(p, 16, count, (S)n2);
        R.c_sl_iseq = p.cur;
        p.cur = cur_add(p.cur, FP_ISEQ_S, n2 * n2);
        R.c_mux = p.cur;
        if (P::is_count) { Cur a = L.rl.c_age, b = L.rl.c_acc, c = L.rl.c_concat; L.rl = R; L.rl.c_age = a; L.rl.c_acc = b; L.rl.c_concat = c; }
    } break;
    UCASE(U_RL_SLROW) {           // rows [a0, a1) of ShiftLeft(64) (shift.circom:27-36): out[i] = sum_j in[j]*(i == j - count)
        const RlRefs& R = L.rl;
        const uint32_t n = 64;
        const S count = (S)n - p.get(M.numLeafAddressNibbles);
        for (uint32_t i = d.a[0]; i < d.a[1]; i++) {
            S acc = 0;
            p.cur = cur_add(R.c_sl_iseq, FP_ISEQ_S, i * n);
            for (uint32_t j = 0; j < n; j++) {
                B e = p.put(R.s_isEq + (i * n + j), gIsEqualS(p, (S)i, (S)((S)j - count)));
                const S tv = p.bit(e) ? p.get(M.addressHashNibbles + j) : 0;
                p.derived(R.s_temp.w + i * n + j, tv);
                acc += tv;
            }
            p.put(R.s_o + i, acc);
        }
    } break;
    UCASE(U_RL_ACC_B) {           // CountBytes(N) (integer.circom:16-49) and ShiftLeft(N) (:85) of RlpInteger
        const RaRefs& A = L.ra;
        const int N = prm.amountBytes;
        p.cur = A.c_cb;
        const S length = p.put(A.len, gCountBytes(p, N, A.by));
        SmRef r = gShiftLeft(p, N, A.by, N - length, true);
        copy_n(p, A.be, r, (int)(N));
    } break;
    UCASE(U_RL_ACC_C) {           // RlpInteger outputs (:96-109), RlpEmptyAccount prefixes + Concat(4+N, 66) (empty_account.circom:40-133)
        const RaRefs& A = L.ra;
        const RlRefs& R = L.rl;
        const int N = prm.amountBytes, maxAcc = 4 + N + 66;
        const bool sb = p.bit(p.get(A.isb)), zb = p.bit(p.get(A.isz));
        const S first = p.get(A.frb), length = p.get(A.len);
        p.put(A.pn + 2, 0x80);
        p.put(A.pn + 3, p.put(A.br, p.put(A.ri_o, first + (zb ? 0x80 : 0))));
        for (int j = 1; j < N + 1; j++) p.put(A.pn + 3 + j, p.put(A.br + j, p.put(A.ri_o + j, sb ? 0 : p.get(A.be + (j - 1)))));
        const S blen = p.put(A.brl, p.put(A.ri_ol, (sb ? 0 : 1) + length + (zb ? 1 : 0)));
        const S nb = p.put(A.nbl, 1 + blen);
        const S pl = p.put(A.pnl, 2 + nb);
        for (int j = 0; j < 66; j++) p.put(A.sc + j, (S)empty_account_tail(j));
        p.put(A.pn, 0xf8);
        p.put(A.pn + 1, nb + 66);
        p.cur = A.c_concat;
        S clen;
        SmRef cc;
        if constexpr (P::is_check) cc = gConcatHead(p, 4 + N, 66, A.pn, pl, A.sc, (S)66, clen);       // (its children: CK_CAT units)
        else cc = gConcat(p, 4 + N, 66, A.pn, pl, A.sc, (S)66, clen, true);
        { copy_n(p, A.ea_o, cc, (int)(maxAcc)); copy_n(p, R.acc, cc, (int)(maxAcc)); }
        p.put(R.accLen, p.put(A.ea_ol, clen));
    } break;
    UCASE(U_RL_B) {               // rest of TruncatedAddressHash (:62-90), AssertGreaterEqThan (:151), prefixes (:166-181), Concat (:183-188)
        const RlRefs& R = L.rl;
        const int ab = 32, n2 = 64, bb = prm.amountBytes, maxAcc = 4 + bb + 66, maxVal = 2 + maxAcc, maxKey = 1 + ab, maxPK = 2 + 1 + maxKey, maxOut = maxPK + maxVal;
        copy_n(p, R.t_shf, R.s_o, (int)(n2));
        const S r = p.get(R.t_rm), q = p.get(R.t_dv);
        p.put(R.t_on, 2 + r);
        p.put(R.t_on + 1, r * p.get(R.t_shf));
        const B rbit = p.ballot(r & 1);
        p.cur = R.c_mux;
        for (int i0 = 0; i0 < n2; i0 += 8) {              // (9 operands requested together per 8 multiplexers)
            S v[9];
#pragma unroll
            for (int q = 0; q < 9; q++) v[q] = p.get(R.t_shf + (i0 + q < n2 ? i0 + q : n2 - 1));
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = i0 + q;
                if (i < n2 - 1) p.put(R.t_on + i + 2, gMux1S(p, v[q], v[q + 1], rbit));
                else p.put(R.t_on + i + 2, (1 - r) * v[q]);
            }
        }
        SmRef by = gNibbles2Bytes(p, ab + 1, R.t_on);
        { copy_n(p, R.t_o, by, (int)(ab + 1)); copy_n(p, R.key, by, (int)(ab + 1)); }
        const S kl = p.put(R.keyLen, p.put(R.t_ol, 1 + q));
        gAssertGreaterEqThanS(p, 16, kl, (S)2);
        // the evaluator runs the rest as two wavefronts of their own (CK_RL_B2, CK_RL_B3): every relation below is between stored wires, and this unit was the longest
        // of the narrow evaluation kernel (0.24 ms alone)
        if constexpr (!P::is_check) { S pl, vl; rl_b_prefixes(p, R, prm, kl, pl, vl); rl_b_concat(p, R, M, prm, pl, vl); }
    } break;
    UCASE(CK_CAT) { gConcatPart(p, (int)d.a[0], (int)d.a[1], (int)d.a[2], d.a[3], d.a[4]); } break;
    UCASE(CK_RL_B2) { S pl, vl; rl_b_prefixes(p, L.rl, prm, p.get(L.rl.keyLen), pl, vl); } break;
    UCASE(CK_RL_B3) { rl_b_concat(p, L.rl, M, prm, p.get(L.rl.pkLen), p.get(L.rl.valLen)); } break;
    UCASE(U_POW_POST) {           // :73-79
        B fr[4] = {0, 0, 0, 0};
        const BitRef f = gFilter(p, 32, p.get(L.pw.mzb), fr);
        for (int i = 0; i < 32; i++) {
            // shouldBeZero[i] <== Filter.out[i]: generation / emission from the values (not read back), the evaluator from the stored wire
            B z = p.put(L.pw.sbz + i, P::is_check ? p.get(f + (uint32_t)i) : p.run_bcast(fr[0], (uint32_t)i));
            p.require(p.ballot(p.get(L.pw.keccak + i) == 0) | ~z, FAILCODE(T_POW, 79));
        }
    } break;
    UCASE(U_POB_FINAL) {          // :186, :188, :191-193, :203-206
        S cnt = 0;
        for (int i = 0; i < prm.L; i++) cnt += (S)p.bit(p.get(M.isLeaf + i));
        p.require(p.ballot(cnt == 1), FAILCODE(T_POB, 186));
        p.require(p.get(M.isLastLayerLeaf), FAILCODE(T_POB, 188));
        auto same = [&](SmRef a, SmRef b, int n) {       // a[i] == b[i] for every i < n, 16 loads requested together
            bool ok = true;
            for (int i0 = 0; i0 < n; i0 += 8) {
                S x[8], y[8];
#pragma unroll
                for (int q = 0; q < 8; q++) { const int i = i0 + q < n ? i0 + q : n - 1; x[q] = p.get(a + i); y[q] = p.get(b + i); }
#pragma unroll
                for (int q = 0; q < 8; q++) ok = ok && x[q] == y[q];
            }
            return ok;
        };
        p.require(p.ballot(same(M.layerKeccaks, M.stateRoot, 32)), FAILCODE(T_POB, 192));
        p.require(p.ballot(same(M.leaf, M.lastLayer, 139)), FAILCODE(T_POB, 204));
        p.require(p.ballot(p.get(M.leafLen) == p.get(M.lastLayerLen)), FAILCODE(T_POB, 206));
    } break;
    UCASE(U_POB_INPUT_FR) {  // FR main inputs (canonical LE -> Montgomery)
        p.put(M.burnKey, p.input_fr(0)); p.put(M.actualBalance, p.input_fr(1)); p.put(M.intendedBalance, p.input_fr(2));
        p.put(M.revealAmount, p.input_fr(3)); p.put(M.burnExtraCommitment, p.input_fr(4)); p.put(M.proofExtraCommitment, p.input_fr(5));
    } break;
    UCASE(U_POB_RANGE) {   // proof_of_burn.circom:84-97 in five parallel parts (a[0]): each comparison's bit decompositions are one wavefront's work
        const int AB8 = prm.amountBytes * 8;
        F intended = p.get(M.intendedBalance);
        if (d.a[0] == 0) gAssertLessEqThanF(p, AB8, intended, prm.maxIntended);
        else if (d.a[0] == 1) gAssertLessEqThanF(p, AB8, p.get(M.actualBalance), prm.maxActual);
        else if (d.a[0] == 2) gAssertLessEqThanF(p, AB8, intended, p.get(M.actualBalance));
        else if (d.a[0] == 3) {
            S relax = p.get(M.byteSecurityRelax);
            gAssertLessEqThanS(p, 16, (S)((uint32_t)relax * 2u), (S)prm.minNib);
            gAssertGreaterEqThanS(p, 16, p.get(M.numLeafAddressNibbles), (S)((uint32_t)prm.minNib - (uint32_t)relax * 2u));
            gAssertBitsF(p, AB8, p.get(M.revealAmount));
        } else gAssertLessEqThanF(p, AB8, p.get(M.revealAmount), intended);
    } break;
    UCASE(U_POB_POSEIDONS) {      // :113 (a[0] = 0) remainingCoin = Poseidon3, :116 (a[0] = 1) nullifier = Poseidon2 -- two parallel units
        // (evaluation / emission only: generation runs the two blocks as U_POS_WIDE units, which also write remainingCoin / nullifier)
        F bk = p.get(M.burnKey);
        if (d.a[0] == 0) {
            F in3[3] = {L.prefix[2], bk, fr_sub(p.get(M.intendedBalance), p.get(M.revealAmount))};
            const PosSrc ps = {2, {M.burnKey.i, M.intendedBalance.i, POS_NONE}, M.revealAmount.i, M.remainingCoin.i};
            p.put(M.remainingCoin, gPoseidonU<P, 4>(p, pos_off(4), in3, ps));
        } else {
            F in2[2] = {L.prefix[1], bk};
            const PosSrc ps = {1, {M.burnKey.i, POS_NONE, POS_NONE}, POS_NONE, M.nullifier.i};
            p.put(M.nullifier, gPoseidonU<P, 3>(p, pos_off(3), in2, ps));
        }
    } break;
    UCASE(U_BAH_PRE) {            // BurnAddressHash burn_address.circom:67-79 up to the sponge
        F bk = p.put(L.bah.in, p.get(M.burnKey)), ra = p.put(L.bah.in + 1, p.get(M.revealAmount)), bec = p.put(L.bah.in + 2, p.get(M.burnExtraCommitment));
        F hc;
        const PosSrc ps = {0, {M.burnKey.i, M.revealAmount.i, M.burnExtraCommitment.i}, POS_NONE, POS_NONE};
        gBurnAddress(p, pos_off(5), L.prefix[0], bk, ra, bec, L.fp_n2be32, ps, &hc);
        // addressBytes, Fit(20, 136) [out[136] | in[20]] and the Keccak block: the 20 bytes again, written per witness (nothing read back)
        SmRef fo = p.sms(136), fi = p.sms(20);
        auto ab_byte = [&](int i) { return i < 20 ? canon_byte(hc, 31 - i) : (S)0; };
        sm_puts<P, 8>(p, L.bah.addressBytes, 20, ab_byte); sm_puts<P, 8>(p, fi, 20, ab_byte);
        sm_puts<P, 8>(p, fo, 136, ab_byte); sm_puts<P, 8>(p, L.bah.block, 136, ab_byte);
        KBRefs r = L.kbs[L.bah.kb];
        kb_head(p, 1, (S)20, r);
        if (P::is_count) L.kbs[L.bah.kb] = r;
    } break;
    UCASE(U_POB_N2B) {            // :132-136  Num2BigEndianBytes(32) of nullifier, remainingCoin, revealAmount, burnExtraCommitment, _proofExtraCommitment
        const int j = d.a[0];
        FrRef src = j == 0 ? M.nullifier : j == 1 ? M.remainingCoin : j == 2 ? M.revealAmount : j == 3 ? M.burnExtraCommitment : M.proofExtraCommitment;
        SmRef dst = j == 0 ? M.nullifierBytes : j == 1 ? M.remainingCoinBytes : j == 2 ? M.revealAmountBytes : j == 3 ? M.burnExtraCommitmentBytes : M.extraCommitmentBytes;
        gNum2BigEndianBytesF(p, 32, p.get(src), &dst);
    } break;
    UCASE(U_PC_POST) {            // :40-41 Fit(32,31), BigEndianBytes2Num(31); commitment (proof_of_burn.circom:137 / spend.circom:50)
        SmRef f = gFitS(p, 32, 31, L.pc.hash);
        copy_n(p, L.pc.reduced, f, (int)(31));
        F c = p.put(L.pc.out, gBigEndianBytes2NumF(p, 31, L.pc.reduced));
        if (L.circuit != 2) p.put(L.circuit == 0 ? M.commitment : L.sm.commitment, c);
    } break;
    UCASE(U_RL_ACC) {             // RlpEmptyAccount/RlpInteger, field-element part: Num2BigEndianBytes(N)(balance), LessThan(8N), IsZero, Mux1
                                 // (integer.circom:83,88-90); CountBytes/ShiftLeft and the byte assembly run as light units B and C
        RaRefs A = L.ra;
        const int N = prm.amountBytes;
        A.ea_o = p.sms(70 + N); A.ea_ol = p.sms(1); A.ea_ib = p.frs(1);
        A.pn = p.sms(4 + N); A.pnl = p.sms(1); A.br = p.sms(N + 1); A.brl = p.sms(1); A.nbl = p.sms(1); A.sc = p.sms(66);
        F bal = p.put(A.ea_ib, p.get(M.actualBalance));
        A.ri_o = p.sms(N + 1); A.ri_ol = p.sms(1); A.ri_i = p.frs(1); A.by = p.sms(N); A.len = p.sms(1); A.be = p.sms(N);
        A.isb = p.bits(1); A.isz = p.bits(1); A.frb = p.sms(1);
        F x = p.put(A.ri_i, bal);
        const SmRef r = {p.cur.w, p.cur.s};                 // Num2BigEndianBytes.out[N] = the block's first wires
        F xc;
        gNum2BigEndianBytesFU(p, N, A.ri_i, x, L.fp_n2beN, M.actualBalance, nullptr, &xc);
        S lead = 0; bool still = true;
        for (int j = 0; j < N; j++) {           // bytes[] <== Num2BigEndianBytes.out (generation: from the canonical value, the block is written beside this unit)
            S b = p.put(A.by + j, P::is_gen ? canon_byte(xc, N - 1 - j) : p.get(r + j));
            still = still && b == 0; lead += still;
        }
        A.c_cb = p.cur;
        { CountP q; q.cur = p.cur; q.decl_order = p.decl_order; gCountBytes(q, N, A.by); A.c_sl = q.cur; gShiftLeft(q, N, A.by, 0); A.c_lt = q.cur; }
        p.cur = A.c_lt;
        B single = p.put(A.isb, gLessThanF(p, 8 * N, x, fr_from_i64(128)));
        p.put(A.isz, gIsZeroFd(p, x));
        const S length = P::is_gen ? (S)N - lead : p.get(A.len);
        p.put(A.frb, gMux1SF(p, 0x80 + length, x, single));
        A.c_concat = p.cur;
        { CountP q; q.cur = p.cur; q.decl_order = p.decl_order; S cl; gConcat(q, 4 + N, 66, A.pn, 0, A.sc, 0, cl); A.c_end = q.cur; }
        p.cur = A.c_end;
        if (P::is_count) L.ra = A;
    } break;
    UCASE(U_POW_PRE) {            // ProofOfWorkChecker proof_of_work.circom:54-71 up to the sponge
        F bk = p.put(L.pw.in, p.get(M.burnKey)), ra = p.put(L.pw.in + 1, p.get(M.revealAmount)), bec = p.put(L.pw.in + 2, p.get(M.burnExtraCommitment));
        p.put(L.pw.mzb, (S)((uint32_t)prm.powZero + (uint32_t)p.get(M.byteSecurityRelax)));
        F cb, cr, ce;
        gNum2BigEndianBytesFU(p, 32, L.pw.in, bk, L.fp_n2be32, M.burnKey, &L.pw.keyBytes, &cb);
        gNum2BigEndianBytesFU(p, 32, L.pw.in + 1, ra, L.fp_n2be32, M.revealAmount, &L.pw.raBytes, &cr);
        gNum2BigEndianBytesFU(p, 32, L.pw.in + 2, bec, L.fp_n2be32, M.burnExtraCommitment, &L.pw.becBytes, &ce);
        SmRef e = p.sms(8);                              // EIP7503 :11-21  [out[8]]
        const uint64_t tag = 0x333035372D504945ULL;      // "EIP-7503", first character in the low byte
        for (int i = 0; i < 8; i++) p.put(L.pw.eip + i, p.put(e + i, (S)((tag >> (8 * i)) & 0xff)));
        SmRef co = p.sms(104), ci = p.sms(104);          // ConcatFixed4(32,32,32,8) :28-48  [out | a,b,c,d]
        auto cat_byte = [&](int i) -> S {     // generation: the three byte strings are written beside this unit (CK_N2BE), so the bytes come from the canonical values
            if constexpr (P::is_gen) return i < 32 ? canon_byte(cb, 31 - i) : i < 64 ? canon_byte(cr, 63 - i) : i < 96 ? canon_byte(ce, 95 - i) : (S)((tag >> (8 * (i - 96))) & 0xff);
            else return p.get(i < 32 ? L.pw.keyBytes + i : i < 64 ? L.pw.raBytes + (i - 32) : i < 96 ? L.pw.becBytes + (i - 64) : L.pw.eip + (i - 96));
        };
        sm_puts<P, 8>(p, ci, 104, cat_byte); sm_puts<P, 8>(p, co, 104, cat_byte); sm_puts<P, 8>(p, L.pw.hin, 104, cat_byte);
        SmRef f = gFitS(p, 104, 136, L.pw.hin);
        copy_n(p, L.pw.block, f, (int)(136));
        KBRefs kr = L.kbs[L.pw.kb];
        kb_head(p, 1, (S)104, kr);
        if (P::is_count) L.kbs[L.pw.kb] = kr;
    } break;
    UCASE(U_SP_INPUT) {
        p.put(L.sm.burnKey, p.input_fr(0)); p.put(L.sm.balance, p.input_fr(1));
        p.put(L.sm.withdrawnBalance, p.input_fr(2)); p.put(L.sm.extraCommitment, p.input_fr(3));
    } break;
    UCASE(U_SP_HEAD) {            // spend.circom:41-49
        F bk = p.get(L.sm.burnKey), bal = p.get(L.sm.balance), wd = p.get(L.sm.withdrawnBalance), ec = p.get(L.sm.extraCommitment);
        gAssertGreaterEqThanF(p, L.spend.maxAmountBytes * 8, bal, wd);
        F in3[3] = {L.prefix[2], bk, bal};
        const FrRef pos1 = {p.cur.w, p.cur.f};                 // Poseidon.out of the first block
        const PosSrc ps1 = {2, {L.sm.burnKey.i, L.sm.balance.i, POS_NONE}, POS_NONE, POS_NONE};
        F coin = p.put(L.sm.coin, gPoseidonU<P, 4>(p, pos_off(4), in3, ps1));
        in3[2] = fr_sub(bal, wd);
        const FrRef pos2 = {p.cur.w, p.cur.f};
        const PosSrc ps2 = {2, {L.sm.burnKey.i, L.sm.balance.i, POS_NONE}, L.sm.withdrawnBalance.i, POS_NONE};
        F rc = p.put(L.sm.remainingCoin, gPoseidonU<P, 4>(p, pos_off(4), in3, ps2));
        gNum2BigEndianBytesFU(p, 32, L.sm.coin, coin, L.fp_n2be32, pos1, &L.sm.coinBytes);
        gNum2BigEndianBytesFU(p, 32, L.sm.withdrawnBalance, wd, L.fp_n2be32, L.sm.withdrawnBalance, &L.sm.withdrawnBalanceBytes);
        gNum2BigEndianBytesFU(p, 32, L.sm.remainingCoin, rc, L.fp_n2be32, pos2, &L.sm.remainingCoinBytes);
        gNum2BigEndianBytesFU(p, 32, L.sm.extraCommitment, ec, L.fp_n2be32, L.sm.extraCommitment, &L.sm.extraCommitmentBytes);
    } break;
    UCASE(U_POB_LAYER_POST) {     // :166-170 Fit(32,31) + the head of SubstringCheck (:24-41): own inputs, AssertByteString(sl),
                                 // AssertLessEqThan x2, LittleEndianBytes2Num(sl); AssertByteString(mm) runs as U_ABS_RANGE units
        const int i = d.a[0];
        SmRef f = gFitS(p, 32, 31, M.layerKeccaks + 32 * i);
        for (int k = 0; k < 31; k++) p.put(M.reducedLayerKeccaks + (31 * i + k), p.get(f + k));
        if (i > 0) {
            ScRefs sc = L.scs[i];
            const int mm = LB, sl = 31, kk = mm - sl + 1;
            sc.out = p.bits(1); sc.mi_w = p.dvs(mm); sc.ml = p.sms(1); sc.si = p.sms(sl);
            sc.num = p.frs(1); sc.M_w = p.dvs(mm + 1); sc.ex = p.bits(kk); sc.isl = p.bits(kk); sc.alw = p.bits(kk + 1); sc.sums = p.sms(kk + 1); sc.dne = p.bits(1);
            const S mainLen = p.get(sc.ml);                  // mainInput[] / mainLen are written by U_SC_MI (inputs only: pre-work track)
            copy_n(p, sc.si, M.reducedLayerKeccaks + (uint32_t)(31 * i), sl);      // (batched: the evaluator's compares are resolved per 8 wires -- as single puts LLVM sank all 31 compares and spilled 311 VGPRs)
            sc.c_abs_sub = p.cur;
            gAssertByteString(p, sl, sc.si);
            sc.abs_main_in = SmRef{p.dvs(mm), 0};
            sc.c_abs_main = p.cur;
            p.cur = cur_add(p.cur, FP_ABITS8, mm);
            sc.c_after_abs = p.cur;
            gAssertLessEqThanS(p, 16, mainLen, (S)mm);
            gAssertLessEqThanS(p, 16, (S)sl, mainLen);
            p.put(sc.num, gLittleEndianBytes2NumF(p, sl, sc.si));
            sc.c_loop = p.cur;
            p.cur = cur_add(cur_add(p.cur, FP_ISEQ_S, kk), FP_ISEQ_F, kk);
            sc.c_tail = p.cur;
            p.cur = cur_add(p.cur, Cur{3, 1, 0, 0, 2}, 1);         // the final IsZero (in, inv: derived)
            if (P::is_count) L.scs[i] = sc;
        }
    } break;
    UCASE(U_SC_MI) {              // SubstringCheck's own copies of its inputs: mainInput[mm] <== layers[i-1], mainLen <== layerLens[i-1] (:25-26)
        const ScRefs& sc = L.scs[d.a[0]];
        if constexpr (P::is_emit) { for (int k = 0; k < LB; k++) p.derived(sc.mi_w + k, p.get(M.layers + ((d.a[0] - 1) * LB + k))); }      // (derived copies)
        p.put(sc.ml, p.get(M.layerLens + (d.a[0] - 1)));
    } break;
    UCASE(U_SC_M) {               // EMISSION ONLY (M[] are derived wires): M[k+1] <== mainInput[k]*256^k + M[k]  (substring_check.circom:45-49) for k in [a1, a2); reads the source
                                 // bytes; 256^k comes from a table in "double Montgomery" form so that byte * 256^k is ONE Montgomery
                                 // product.  The prefix M[a1] is rebuilt from the bytes below a1, 31 at a time (31 bytes packed into limbs
                                 // are one canonical value: one product per 31 bytes), so the 17 ranges of a layer run side by side.
        const ScRefs& sc = L.scs[d.a[0]];
        const uint32_t lo = d.a[1], hi = d.a[2];
        const SmRef src = M.layers + ((d.a[0] - 1) * LB);
        F acc = fr_zero();
        for (uint32_t c0 = 0; c0 < lo; c0 += 31) {
            const uint32_t n = lo - c0 < 31 ? lo - c0 : 31;
            S by[31];
#pragma unroll
            for (uint32_t q = 0; q < 31; q++) by[q] = p.get(src + (c0 + (q < n ? q : 0)));
            bool small = true;
            Fr v = fr_zero();
#pragma unroll
            for (uint32_t q = 0; q < 31; q++) if (q < n) { small = small && (uint32_t)by[q] < 256u; v.l[q >> 2] |= ((uint32_t)by[q] & 0xffu) << (8 * (q & 3)); }
            if (p.ballot(!small) == 0) acc = fr_add(acc, fr_mul(v, p.k256r(c0)));
            else {               // some witness carries a non-byte here (it fails AssertByteString): keep the per-byte definition exact
#pragma unroll 1
                for (uint32_t q = 0; q < n; q++) {
                    Fr b1 = {{(uint32_t)p.get(src + (c0 + q)), 0, 0, 0, 0, 0, 0, 0}};
                    acc = fr_add(fr_mul(b1, p.k256r(c0 + q)), acc);
                }
            }
        }
        if (lo == 0) p.derived_fr(sc.M_w, fr_zero());
        for (uint32_t k = lo; k < hi; k++) {
            Fr b1 = {{(uint32_t)p.get(src + k), 0, 0, 0, 0, 0, 0, 0}};
            acc = fr_add(fr_mul(b1, p.k256r(k)), acc);
            p.derived_fr(sc.M_w + k + 1, acc);
            p.site_m(sc.M_w + k + 1, sc.mi_w + k, k);
        }
    } break;
    UCASE(U_SC_RANGE) {           // positions [a1, a2) of the existence loop (:83-95): IsEqual(isLastIndex), IsEqual(exists) per position
        const ScRefs& sc = L.scs[d.a[0]];
        const uint32_t lo = d.a[1], hi = d.a[2], sl = 31;
        const S mainLen = p.get(sc.ml);
        const F subNum = p.get(sc.num);
        const Fr subC = fr_from_mont(subNum);
        const SmRef src = M.layers + ((d.a[0] - 1) * LB);
        // allowed[i] = prod_{j<i}(1 - isLastIndex[j]) = [mainLen - sl + 1 >= i] (unsigned); the evaluator re-reads it
        const uint32_t lastIdx = (uint32_t)(mainLen - (S)sl + 1);
        const uint32_t ln = p.lane_id(), cnt = hi - lo;                      // cnt <= 32
        bool allowed = P::is_gen ? (lastIdx >= lo) : p.bit(p.get(sc.alw + lo));
        // Per position: IsEqual([i, lastIndex]) and IsEqual([subNum * 256^i, M[i+sl] - M[i]]) -- 12 wires: 4 BIT outputs (IsEqual.out, IsZero.out
        // twice) and 8 DERIVED operand wires (policy.hpp), i.e. nothing but bits is stored, and exists[i] needs no field arithmetic: it is
        // [subNum == the 31 bytes from position i] on a sliding window (gadgets.hpp sc_window; rounds 1-2 stored M[] and the four field-element
        // operands and ran a Montgomery batch inversion per unit through the witness' own slots: 2.3 GB per batch).
        // The bits leave as lane-distributed runs: isLastIndex[], allowed[], exists[] (one wire per position) and the children's outputs
        // (four per position, consecutive BIT ranks: 16 positions per run).
        uint32_t wIsl = 0, wAlw = 0, wEx = 0;                                 // this witness' bits of the range (bit t = position lo + t): transposed into runs below
        B wC0 = 0, wC1 = 0;
        F ddL = fr_zero();                                                   // (emitter: lane t <- position lo + t's IsZero operand of the emitted witness)
        Fr win = sc_window(p, src, lo, sl);
        for (uint32_t t = 0; t < cnt; t++) {
            const uint32_t i = lo + t;
            const S nxt = i + sl < (uint32_t)LB ? p.get(src + (i + sl)) : 0;    // (the byte that enters the window next is requested before this position's compares)
            const bool e = fr_eq(win, subC), last = i == lastIdx;
            allowed = allowed && !last;
            wIsl |= (uint32_t)last << t; wAlw |= (uint32_t)allowed << t; wEx |= (uint32_t)e << t;
            const B four = (B)((last ? 3u : 0u) | (e ? 12u : 0u)) << (4 * (t & 15));
            if (t < 16) wC0 |= four; else wC1 |= four;
            if constexpr (P::is_emit) {
                const Cur c = cur_add(cur_add(sc.c_loop, FP_ISEQ_S, i), FP_ISEQ_F, i);
                iseq_derived(p, c, (S)i, (S)lastIdx);
                const F t1 = fr_mul(subNum, p.k256(i)), t2 = fr_mul(fr_to_mont(win), p.k256(i)), dd = fr_sub(t2, t1);
                p.derived_fr(c.w + 7, t1); p.derived_fr(c.w + 8, t2); p.derived_fr(c.w + 10, dd);
                const F ds = p.bcast_sel(dd);
                if (ln == t) ddL = ds;
            }
            sc_window_step(win, sl, (uint32_t)nxt);
        }
        if constexpr (P::is_emit) {      // the cnt inverses of the emitted witness: one per lane, ONE inversion time per unit (a lane at a time they were 10 ms per unit)
            const F inv = fr_is_zero(ddL) ? fr_zero() : fr_inv(ddL);
            if (ln < cnt) p.w32(cur_add(cur_add(sc.c_loop, FP_ISEQ_S, lo + ln), FP_ISEQ_F, lo + ln).w + 11, fr_from_mont(inv), (fr_is_zero(ddL) ? 3 : 2) | 4);      // (counted: pob_debug_emit_counters)
        }
        const B runIsl = p.xpose64(wIsl, 0, cnt), runAlw = p.xpose64(wAlw, 0, cnt), runEx = p.xpose64(wEx, 0, cnt);
        const B runC0 = p.xpose(wC0, 64), runC1 = cnt > 16 ? p.xpose(wC1, 64) : 0;
        p.run_put(cnt, sc.isl.w + lo + ln, sc.isl.i + lo + ln, runIsl);
        p.run_put(cnt, sc.alw.w + lo + 1 + ln, sc.alw.i + lo + 1 + ln, runAlw);
        p.run_put(cnt, sc.ex.w + lo + ln, sc.ex.i + lo + ln, runEx);
        for (uint32_t h2 = 0; h2 < 2 && 16 * h2 < cnt; h2++) {               // children: position lo + 16*h2 + lane/4, output lane%4 (wire offsets 0, 3, 6, 9)
            const uint32_t n = (cnt - 16 * h2 < 16 ? cnt - 16 * h2 : 16) * 4, i = lo + 16 * h2 + (ln >> 2), wh = ln & 3;
            p.run_put(n, sc.c_loop.w + 12 * i + 3 * wh, sc.c_loop.b + 4 * i + wh, h2 ? runC1 : runC0);
        }
    } break;
    UCASE(CK_POS_SEG) {          // segment a1 of the Poseidon(a0 - 1) block at cur, from the stored state wires before it
        if constexpr (!(P::is_check || P::is_count)) { (void)d; }
        else if (d.a[0] == 3) gPoseidonSegStored<P, 3>(p, pos_off(3), d.cur, d.a[1]);
        else if (d.a[0] == 4) gPoseidonSegStored<P, 4>(p, pos_off(4), d.cur, d.a[1]);
        else gPoseidonSegStored<P, 5>(p, pos_off(5), d.cur, d.a[1]);
    } break;
    UCASE(CK_SL_ROWS) {
        if constexpr (P::is_check || P::is_count) gShiftLeftRows(p, (int)d.a[0], d.cur, Cur{d.a[3], d.a[4], d.a[5], d.cur.f, d.cur.q}, d.a[1], d.a[2]);
    } break;
    UCASE(CK_N2BE) {             // Num2BigEndianBytes(a0) at cur of the stored FR wire (a1, a2), caller's copy of out[] at (a3, a4) if a5
        if constexpr (P::is_check || P::is_count || P::is_gen) {
            const FrRef src = P::is_gen ? FrRef{d.a[6], d.a[7]} : FrRef{d.a[1], d.a[2]}; const SmRef also = {d.a[3], d.a[4]};
            gNum2BigEndianBytesFv(p, (int)d.a[0], p.get(src), also, d.a[5] != 0);
        }
    } break;
    default: break;
    }
}
// (host planner: every family)
template <class P> GD void unit_run_all(P& p, const UnitDesc& d, CircuitLayout& L) { unit_run<P, FAM_ALL>(p, d, L); }

// ---------------------------------------------------------------------------- host planner
#include <stdlib.h>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>
struct Plan {
    CircuitLayout L;
    std::vector<UnitDesc> units;
    std::vector<SpongeDesc> sponges;
    Cur total;            // = counts (total.w = nWitness)
    uint32_t nfr_in, nsm_in, max_stage;
    // Side tracks: stage ids TRACK_STRIDE*t + s belong to track t.  Track 0 is the main sequence; track t > 0 starts once stage
    // track_fork[t] has completed and must have completed before stage track_join[t] starts.  A track may only be joined by a
    // lower-numbered track (the host enqueues a stage's forked tracks highest first, each one completely).
    enum { TRACK_STRIDE = 32, MAX_TRACKS = 7 };
    uint32_t ntracks, track_fork[MAX_TRACKS], track_join[MAX_TRACKS];
    uint32_t wide_tracks = 0;                // tracks made of chip-filling launches (bit t); the others are narrow serial chains
    CountP p;

    static bool same(Cur a, Cur b) { return a.w == b.w && a.b == b.b && a.s == b.s && a.f == b.f && a.q == b.q; }
    // the monolithic template (gadgets.hpp) walked by CountP must end where the split units say it ends
    void expect_cursor(const char* what, Cur got, Cur want) {
        if (!same(got, want)) throw std::runtime_error(std::string("layout planner: split units disagree with the monolithic template: ") + what);
    }
    void record(uint32_t kind, uint32_t stage, Cur cur, uint32_t a0 = 0, uint32_t a1 = 0, uint32_t a2 = 0, uint32_t a3 = 0, uint32_t a4 = 0, uint32_t a5 = 0, uint32_t a6 = 0, uint32_t a7 = 0) {
        UnitDesc d; d.kind = kind; d.stage = stage; d.cur = cur; d.cost = 0; d.a[0] = a0; d.a[1] = a1; d.a[2] = a2; d.a[3] = a3; d.a[4] = a4; d.a[5] = a5; d.a[6] = a6; d.a[7] = a7;
        d.flags = (kind == CK_POS_SEG || kind == CK_SR_COLS || kind == CK_SL_ROWS || kind == CK_RL_B2 || kind == CK_RL_B3 || kind == CK_CAT) ? UNIT_CHECK      // sub-blocks the evaluator runs on their own
                : kind == CK_N2BE ? (UNIT_GEN | UNIT_CHECK)                                            // ... and the generator too
                : kind == U_POS_WIDE ? UNIT_GEN
                : kind == U_SC_M ? UNIT_EMIT                                                           // (M[]: derived wires, rebuilt by the emitter only)
                : kind == U_POB_POSEIDONS ? (UNIT_CHECK | UNIT_EMIT)                                   // generation: nothing but the two U_POS_WIDE blocks
                : kind == U_GM_INPUT ? (UNIT_GEN | UNIT_CHECK)                                           // (the wires belong to the units of the template and are emitted there)
                : kind == U_POB_INPUT ? UNIT_EMIT                                                      // generation / evaluation: the tile-transposing k_inputs kernel (pob_host.hip)
                : (UNIT_GEN | UNIT_CHECK | UNIT_EMIT);
        units.push_back(d);
        if (stage > max_stage) max_stage = stage;
    }
    void unit(uint32_t kind, uint32_t stage, uint32_t a0 = 0, uint32_t a1 = 0, uint32_t a2 = 0, uint32_t a3 = 0, uint32_t a4 = 0) {
        record(kind, stage, p.cur, a0, a1, a2, a3, a4);
        const UnitDesc d = units.back();
        p.nnotes = 0;
        unit_run_all(p, d, L);             // CountP: advances p.cur over the unit's wires, fills the reference tables
        take_notes(p, stage);
    }
    // sub-blocks the evaluator runs as wavefronts of their own (they start from stored wires): Poseidon segments 1.., byte conversions,
    // ShiftRight columns, ShiftLeft rows -- from the notes the counting policy took while it walked a composite unit
    void take_notes(CountP& q, uint32_t stage) {
        if (q.notes_overflow) throw std::runtime_error("layout planner: a composite unit has more split sub-blocks than PLAN_MAX_NOTES");
        for (uint32_t k = 0; k < q.nnotes; k++) {
            const PlanNote& nt = q.notes[k];
            if (nt.what == NOTE_POSEIDON) {
                const uint32_t ns = pos_nseg(pos_off((int)nt.n).rp);
                for (uint32_t seg = 1; seg < ns; seg++) record(CK_POS_SEG, stage, nt.cur, nt.n, seg);
                // generation: the block one stage ahead of the composite that continues from its output
                if (stage % TRACK_STRIDE == 0) throw std::runtime_error("layout planner: a Poseidon composite needs a stage before it");
                record(U_POS_WIDE, stage - 1, nt.cur, nt.n, nt.a[0], nt.a[1], nt.a[2], nt.a[3], nt.a[4], nt.a[5]);
            } else if (nt.what == NOTE_N2BE) record(CK_N2BE, stage, nt.cur, nt.n, nt.a[0], nt.a[1], nt.a[2], nt.a[3], nt.a[4], nt.a[5], nt.a[6]);
            else if (nt.what == NOTE_SHIFTLEFT) { for (uint32_t i = 0; i < nt.n; i += 2) record(CK_SL_ROWS, stage, nt.cur, nt.n, i, std::min(i + 2, nt.n), nt.a[0], nt.a[1], nt.a[2]); }
        }
        q.nnotes = 0;
    }
    // a unit recorded at an explicit cursor (not walked by `unit`): walk it on a scratch counting policy for its notes
    void record_composite(uint32_t kind, uint32_t stage, Cur cur) {
        record(kind, stage, cur);
        const UnitDesc d = units.back();
        CountP q; q.nnotes = 0;
        unit_run_all(q, d, L);
        take_notes(q, stage);
    }
    // the children of the Concat(La, Lb) block at c0 as evaluator units (gadgets.hpp gConcatPart)
    void cat_units(uint32_t stage, Cur c0, int La, int Lb) {
        Cur cc[3]; concat_cursors(p, c0, La, Lb, cc);
        for (uint32_t part = 0; part < 4; part++) record(CK_CAT, stage, cc[part < 3 ? part : 2], part, (uint32_t)La, (uint32_t)Lb, c0.w, c0.s);
    }
    // AssertByteString(N)(src) as range units; p.cur = start of the AssertByteString block
    // (emit_only: the assert itself is held by the KeccakBytes range units that load the same bytes -- kb_range(assert_src) --, the units only write the derived wires)
    void abs_units(uint32_t stage, uint32_t N, SmRef src, uint32_t chunk = 32, bool emit_only = false) {
        CountP chk; chk.cur = p.cur; gAssertByteString(chk, (int)N, src);
        const uint32_t own_w = p.dvs(N);
        const Cur c0 = p.cur;
        for (uint32_t lo = 0; lo < N; lo += chunk) { record(U_ABS_RANGE, stage, c0, own_w, 0, src.w, src.i, lo, std::min(lo + chunk, N)); if (emit_only) units.back().flags = UNIT_EMIT; }
        p.cur = cur_add(c0, FP_ABITS8, N);
        expect_cursor("AssertByteString", p.cur, chk.cur);
    }
    // sponge + selector rows + post of a KeccakBytes whose head/ranges are planned; p.cur = start of the Keccak(mb) block
    void keccak_tail(uint32_t kb, uint32_t range_stage, SmRef dst, bool has_dst) {
        KBRefs& r = L.kbs[kb];
        r.index = kb;
        kb_declare_keccak(p, r);
        SpongeDesc s; s.n = r.mb; s.stage = range_stage + 1; s.src_b = r.inBlocks.i; s.src_w = r.inBlocks.w;
        s.kin_b = r.k_in.i; s.kin_w = r.k_in.w; s.fin_b = r.f_in.i; s.fin_w = r.f_in.w; s.fs_b = r.f_s.i; s.fs_w = r.f_s.w; s.abs_b = r.abs_b; s.abs_w = r.abs_w;
        sponges.push_back(s);
        r.dst = dst; r.has_dst = has_dst ? 1 : 0;
        {   // Reshape/Bits2Num blocks start after the 1600 selectors (all of equal footprint)
            CountP q; q.cur = p.cur; r.c_post = Cur{0, 0, 0, 0}; r.has_dst = 0; kb_selrow(q, r, 24, 0, 1);
            const Cur fp1 = {q.cur.w - p.cur.w, q.cur.b - p.cur.b, q.cur.s - p.cur.s, 0, q.cur.q - p.cur.q};
            r.c_post = cur_add(p.cur, fp1, 1600); r.has_dst = has_dst ? 1 : 0;
        }
        const uint32_t split = r.mb >= 8 ? 4 : r.mb >= 3 ? 2 : 1;            // a row's 64 selectors are split over several wavefronts
        for (uint32_t row = 0; row < 25; row++) {
            const Cur c0 = p.cur;
            CountP q; q.cur = c0; kb_selrow(q, r, row, 0, 64);
            const Cur fp = {(q.cur.w - c0.w) / 64, (q.cur.b - c0.b) / 64, (q.cur.s - c0.s) / 64, 0, (q.cur.q - c0.q) / 64};
            for (uint32_t k = 0; k < split; k++) record(U_KB_SELROW, range_stage + 2, cur_add(c0, fp, k * (64 / split)), kb, row, k * (64 / split), (k + 1) * (64 / split));
            p.cur = q.cur;
        }
        if (!same(p.cur, r.c_post)) throw std::runtime_error("layout planner: selector footprint");
        unit(U_KB_POST, range_stage + 2, kb);
        if (range_stage + 1 > max_stage) max_stage = range_stage + 1;
    }
    void keccak_bytes(uint32_t kb, int mb, uint32_t stage, SmRef src, SmRef len, SmRef dst, uint32_t len_input = 0, bool has_dst = true, bool assert_src = false, SmRef also = SmRef{0, 0}) {
        L.kbs[kb].mb = mb;
        const Cur start = p.cur;
        unit(U_KB_HEAD, stage, kb, len.w, len.i, len_input);
        const uint32_t m = 136 * mb;
        for (uint32_t lo = 0; lo < m; lo += 16) record(U_KB_RANGE, stage + 1, start, kb, src.w, src.i, lo, std::min(lo + 16, m), assert_src ? 1u : 0u, also.w, also.i);
        keccak_tail(kb, stage + 1, dst, has_dst);
    }
    // byte-range units of a KeccakBytes whose head ran inside another unit at `head_stage`
    void kb_ranges(uint32_t kb, uint32_t stage, SmRef src) {
        const uint32_t m = 136 * L.kbs[kb].mb;
        for (uint32_t lo = 0; lo < m; lo += 16) record(U_KB_RANGE, stage, p.cur, kb, src.w, src.i, lo, std::min(lo + 16, m));
    }
    // LeafDetector(N)(src, len) -> dst, as head (stage), 4 selectors x ranges (stage+1), tail (stage+2)
    void leaf_detector(uint32_t inst, uint32_t stage, SmRef src, SmRef len, BitRef dst, bool copy_units = true) {      // copy_units false: the layer's KeccakBytes ranges write layer[] too (kb_range `also`)
        const uint32_t N = 136 * L.pob.NB;
        CountP chk; chk.cur = p.cur; gLeafDetector(chk, (int)N, src, 0);
        L.lds[inst].src = src; L.lds[inst].len_src = len; L.lds[inst].dst = dst;
        unit(U_LD_HEAD, stage, inst);
        expect_cursor("LeafDetector", p.cur, chk.cur);
        if (copy_units) for (uint32_t lo = 0; lo < N; lo += 32) record(U_LD_COPY, stage, p.cur, inst, lo, std::min(lo + 32, N));
        for (uint32_t k = 0; k < 4; k++) for (uint32_t lo = 0; lo < N; lo += 34) record(U_LD_SELR, stage + 1, p.cur, inst, k, lo, std::min(lo + 34, N));
        record(U_LD_TAIL, stage + 2, p.cur, inst);
    }
    // BurnAddressHash (burn_address.circom:67-84): pre | byte ranges +1 | sponge +2 | selector rows, post +3 | Bytes2Nibbles +4.  As a MAIN
    // (circuit 2) the template's own input wires play the part of the parent's burnKey / revealAmount / burnExtraCommitment.
    void burn_address_hash(uint32_t pre_stage) {
        L.bah.nibbles = p.sms(64); L.bah.in = p.frs(3); L.bah.addressBytes = p.sms(20); L.bah.block = p.sms(136); L.bah.hash = p.sms(32);
        L.bah.kb = L.nkb++;
        if (L.circuit == 2) { L.pm.burnKey = L.bah.in; L.pm.revealAmount = L.bah.in + 1; L.pm.burnExtraCommitment = L.bah.in + 2; L.pm.addressHashNibbles = L.bah.nibbles; }
        unit(U_BAH_PRE, pre_stage);
        kb_ranges(L.bah.kb, pre_stage + 1, L.bah.block);
        keccak_tail(L.bah.kb, pre_stage + 1, L.bah.hash, true);
        unit(U_BAH_POST, pre_stage + 4);
    }
    // ProofOfWorkChecker (proof_of_work.circom:54-81), same stages.  As a main: minimumZeroBytes is an input (powZero = 0 + the stored input wire)
    void proof_of_work_checker(uint32_t pre_stage) {
        L.pw.in = p.frs(3); L.pw.mzb = p.sms(1); L.pw.keyBytes = p.sms(32); L.pw.raBytes = p.sms(32); L.pw.becBytes = p.sms(32); L.pw.eip = p.sms(8);
        L.pw.hin = p.sms(104); L.pw.block = p.sms(136); L.pw.keccak = p.sms(32); L.pw.sbz = p.bits(32);
        L.pw.kb = L.nkb++;
        if (L.circuit == 2) { L.pm.burnKey = L.pw.in; L.pm.revealAmount = L.pw.in + 1; L.pm.burnExtraCommitment = L.pw.in + 2; L.pm.byteSecurityRelax = L.pw.mzb; }
        unit(U_POW_PRE, pre_stage);
        kb_ranges(L.pw.kb, pre_stage + 1, L.pw.block);
        keccak_tail(L.pw.kb, pre_stage + 1, L.pw.keccak, true);
        unit(U_POW_POST, pre_stage + 4);
    }
    void public_commitment(int N, uint32_t pre_stage) {   // public_commitment.circom:18-42
        L.pc.N = N; L.pc.nb = N * 32 / 136 + ((N * 32) % 136 != 0);
        L.pc.out = p.frs(1); L.pc.in = p.sms(32 * N); L.pc.flat = p.sms(32 * N); L.pc.block = p.sms(136 * L.pc.nb); L.pc.hash = p.sms(32); L.pc.reduced = p.sms(31);
        L.pc.kb = L.nkb++; L.kbs[L.pc.kb].mb = L.pc.nb;
        {   // the monolithic statement (public_commitment.circom:22-36) must end where the split units say it ends
            CountP chk; chk.cur = p.cur; chk.decl_order = p.decl_order;
            for (int j = 0; j < N; j++) gAssertByteString(chk, 32, L.pc.in + 32 * j);
            gFlattenS(chk, 32 * N, L.pc.in); gFitS(chk, 32 * N, 136 * L.pc.nb, L.pc.flat);
            unit(U_PC_PRE, pre_stage);
            for (int j = 0; j <= N; j++) record(U_PC_COPY, pre_stage, p.cur, (uint32_t)j);
            KBRefs r2 = L.kbs[L.pc.kb]; kb_head(chk, L.pc.nb, (S)(32 * N), r2);
            expect_cursor("PublicCommitment", p.cur, chk.cur);
        }
        kb_ranges(L.pc.kb, pre_stage + 1, L.pc.block);
        keccak_tail(L.pc.kb, pre_stage + 1, L.pc.hash, true);
        unit(U_PC_POST, pre_stage + 4);
    }

    // cost estimate (wires written, FR wires x8) of every unit, by replaying it on the counting policy
    void estimate_costs() {
        for (UnitDesc& d : units) {
            if (d.kind == U_POS_WIDE) { d.cost = 8 * pos_wires((int)d.a[0], pos_off((int)d.a[0]).rp); continue; }
            CountP q; const UnitDesc dd = d; unit_run_all(q, dd, L); d.cost = q.nput * (unit_is_heavy(d.kind) ? 4 : 1);
        }
    }
    void plan_pob(const PobParams& prm) {
        memset(&L, 0, sizeof L);
        L.circuit = 0; L.pob = prm; L.nkb = 0; max_stage = 0; L.decl_order = p.decl_order;
        L.fp_n2be32 = n2be_footprint(32); L.fp_n2beN = n2be_footprint(prm.amountBytes);
        // track 1 (TB): everything that hangs off the main inputs only -- range checks, Poseidons, BurnAddressHash, ProofOfWorkChecker,
        // RlpMerklePatriciaTrieLeaf -- runs beside the layer/header Keccak sponges of the main track and is joined before main stage 5
        // (joining it only before the final stage 10 was measured: no gain, the tracks' streams just contend differently).  track 2 (TR): the RlpMerklePatriciaTrieLeaf assembly, forked once BurnAddressHash is done (TB + 5) and
        // joined before the final comparisons (main stage 10).  track 3 (TC): RlpEmptyAccount's serial chain, joined before TR + 2.
        // track 4 (TN): the five Num2BigEndianBytes of PublicCommitment's inputs, forked once the Poseidons are done (TB + 1), joined
        // before PublicCommitment's track forks (main stage 4); it runs on the main track's BN254 stream, which is idle until then.
        const uint32_t TB = TRACK_STRIDE, TR = 2 * TRACK_STRIDE, TC = 3 * TRACK_STRIDE, TN = 4 * TRACK_STRIDE;
        ntracks = 6; track_fork[1] = 0; track_join[1] = 5; track_fork[2] = TB + 6; track_join[2] = 10; track_fork[3] = 0; track_join[3] = TR + 2;
        track_fork[4] = TB + 1; track_join[4] = 4;
        // track 5 (TP): everything of the layers / header that is NOT on the way to their Keccak sponges -- byte asserts, SelectorArray1D,
        // leaf detectors -- so that the main track is only inputs + KeccakBytes heads (0), byte ranges (1), sponges A (2), rows (3)
        // and the round expansion starts 0.5 ms earlier; joined before main stage 5.
        const uint32_t TP = 5 * TRACK_STRIDE;
        track_fork[5] = 0; track_join[5] = 5; wide_tracks = 1u << 5;
        // track 6 (TQ): PublicCommitment -- KeccakBytes head, byte ranges, its 2-block sponge, selector rows, the commitment -- depends on the
        // header hash (main stage 3) and the Num2BigEndianBytes of track 4 only, not on the SubstringChecks: forked after (empty) main stage 4,
        // once track 4 is joined, it runs beside main stages 5..7 instead of extending them by four short dependent stages; joined before 10.
        const uint32_t TQ = 6 * TRACK_STRIDE;
        ntracks = 7; track_fork[6] = 4; track_join[6] = 10;
        PobMain& M = L.pm;
        const int Ln = prm.L, LB = 136 * prm.NB, HBy = 136 * prm.HB;
        p.cur = Cur{1, 0, 0, 0};                      // wire 0 = constant 1
        M.commitment = p.frs(1);
        M.burnKey = p.frs(1); M.actualBalance = p.frs(1); M.intendedBalance = p.frs(1); M.revealAmount = p.frs(1); M.burnExtraCommitment = p.frs(1);
        M.numLeafAddressNibbles = p.sms(1); M.layers = p.sms(Ln * LB); M.layerLens = p.sms(Ln); M.numLayers = p.sms(1);
        M.blockHeader = p.sms(HBy); M.blockHeaderLen = p.sms(1); M.byteSecurityRelax = p.sms(1); M.proofExtraCommitment = p.frs(1);
        M.remainingCoin = p.frs(1); M.nullifier = p.frs(1); M.addressHashNibbles = p.sms(64); M.blockRoot = p.sms(32); M.stateRoot = p.sms(32);
        M.nullifierBytes = p.sms(32); M.remainingCoinBytes = p.sms(32); M.revealAmountBytes = p.sms(32); M.burnExtraCommitmentBytes = p.sms(32);
        M.extraCommitmentBytes = p.sms(32); M.lastLayer = p.sms(LB); M.lastLayerLen = p.sms(1); M.layerExists = p.bits(Ln);
        M.substringCheckers = p.bits(Ln - 1); M.layerKeccaks = p.sms(32 * Ln); M.reducedLayerKeccaks = p.sms(31 * Ln); M.isLeaf = p.bits(Ln);
        M.isLastLayerLeaf = p.bits(1); M.leaf = p.sms(139); M.leafLen = p.sms(1);
        nfr_in = 6; nsm_in = 1 + Ln * LB + Ln + 1 + HBy + 2;

        // main track: 0 inputs + KeccakBytes heads | 1 KeccakBytes byte ranges | 2 sponges A | 3 output selector rows, posts
        //             | 5 consumers (SubstringCheck, PublicCommitment) | 6 ... | 10 final ===          (side tracks: see above)
        unit(U_POB_INPUT_FR, 0);
        for (uint32_t k = 0; k < nsm_in; k += 256) unit(U_POB_INPUT, 0, k, std::min(k + 256, nsm_in));
        for (int k = 0; k < 5; k++) unit(U_POB_RANGE, TB + 1, k);
        for (int i = 0; i < Ln; i++) { unit(U_POB_LAYER_ASSERT, TP + 1, i); abs_units(TP + 1, LB, M.layers + i * LB, 32, true); }
        unit(U_POB_HDR_ASSERT, TP + 1); abs_units(TP + 1, HBy, M.blockHeader, 32, true);
        // The three Poseidon blocks (:113, :116, burn_address.circom:55) run in TB + 1 as U_POS_WIDE units (state spread over lanes);
        // the composites that continue from their outputs follow in TB + 2, with the Num2BigEndianBytes blocks beside them (CK_N2BE).
        unit(U_POB_POSEIDONS, TB + 2, 0);
        unit(U_POB_POSEIDONS, TB + 2, 1);
        burn_address_hash(TB + 2);                             // :119
        L.kb_hdr = L.nkb++;
        keccak_bytes(L.kb_hdr, prm.HB, 0, M.blockHeader, M.blockHeaderLen, M.blockRoot, 0, true, true);       // :122  (the length: the input's WIRE -- the input kernel runs ahead of every stage; a batch in the byte form has no packed int32 rows to read)
        for (int j = 0; j < 5; j++) unit(U_POB_N2B, TN + 1, j);                                    // :132-136
        public_commitment(6, TQ + 1);                                                           // :137  (track 6: pre 1, ranges 2, sponge 3, rows/post 4, commitment 5)
        {   // SelectorArray1D(L, LB)(layers, numLayers - 1) :142-143
            CountP chk; chk.cur = p.cur; gSelectorArray1D(chk, Ln, LB, M.layers, 0);
            L.ll.out = p.sms(LB); L.ll.arr_w = p.dvs(Ln * LB); L.ll.sel_w = p.dvs(1); L.ll.T_w = p.dvs(LB * Ln);
            unit(U_POB_LASTLAYER, TP + 1);
            L.ll.c_sel0 = p.cur;
            const Cur fp = sel_fp((uint32_t)Ln);
            for (uint32_t j = 0; j < (uint32_t)LB; j += 4) record(U_POB_LASTLAYER_RANGE, TP + 1, p.cur, j, std::min<uint32_t>(j + 4, LB));
            p.cur = cur_add(p.cur, fp, LB);
            expect_cursor("SelectorArray1D", p.cur, chk.cur);
        }
        unit(U_POB_LASTLEN, TP + 1);
        L.kb_layer0 = L.nkb; L.nkb += Ln;
        L.nsc = Ln;
        for (int i = 0; i < Ln; i++) {                                                          // :157-181
            leaf_detector(i, TP + 1, M.layers + i * LB, M.layerLens + i, M.isLeaf + i, false);
            keccak_bytes(L.kb_layer0 + i, prm.NB, 0, M.layers + i * LB, M.layerLens + i, M.layerKeccaks + 32 * i, 0, true, true, L.lds[i].layer);
            const Cur start = p.cur;
            unit(U_POB_LAYER_POST, 5, i);
            if (i > 0) {
                const ScRefs sc = L.scs[i];
                CountP chk; chk.cur = start; { gFitS(chk, 32, 31, M.layerKeccaks); gSubstringCheck(chk, LB, 31, M.layers, 0, M.reducedLayerKeccaks); }
                expect_cursor("SubstringCheck", p.cur, chk.cur);
                const SmRef src = M.layers + (i - 1) * LB;
                for (uint32_t lo = 0; lo < (uint32_t)LB; lo += 32) {       // (SubstringCheck's assert on layer i - 1's bytes: held by that layer's KeccakBytes ranges too)
                    record(U_ABS_RANGE, TP + 1, sc.c_abs_main, sc.abs_main_in.w, sc.abs_main_in.i, src.w, src.i, lo, std::min<uint32_t>(lo + 32, LB));
                    units.back().flags = UNIT_EMIT;
                }
                // M[] depends on the layer's INPUT bytes only (not on any hash): with the byte asserts it runs on track 5, beside the expansion
                for (uint32_t lo = 0; lo < (uint32_t)LB; lo += 32) record(U_SC_M, TP + 1, start, i, lo, std::min<uint32_t>(lo + 32, LB));
                const uint32_t kk = LB - 31 + 1;
                record(U_SC_MI, TP + 1, start, i);
                for (uint32_t lo = 0; lo < kk; lo += SC_RANGE_POS) record(U_SC_RANGE, 6, sc.c_loop, i, lo, std::min<uint32_t>(lo + SC_RANGE_POS, kk));
                const uint32_t sums_step = 64;            // (one unit per layer: 0.12 ms alone but 1-2 ms beside the other batch's evaluation, round 2)
                for (uint32_t lo = 0; lo < kk; lo += sums_step) record(U_SC_SUMS, 7, sc.c_tail, i, lo, std::min<uint32_t>(lo + sums_step, kk));
            }
        }
        leaf_detector(Ln, TP + 2, M.lastLayer, M.lastLayerLen, M.isLastLayerLeaf);                    // :187 (lastLayer is written in stage 1)
        {   // RlpMerklePatriciaTrieLeaf :198  (needs addressHashNibbles, written in stage TB+5)
            const Cur start = p.cur;
            CountP chk; chk.cur = start; { S ll; gRlpMptLeaf(chk, 32, prm.amountBytes, M.addressHashNibbles, 0, fr_zero(), ll); }
            unit(U_RL_A, TR + 1);
            RlRefs& R = L.rl;
            for (uint32_t i = 0; i < 64; i += 4) record(U_RL_SLROW, TR + 1, R.c_sl_iseq, i, i + 4);
            // after ShiftLeft: Mux1 x 63, Nibbles2Bytes(33), AssertGreaterEqThan(16), RlpEmptyAccount, Concat
            CountP q; q.cur = R.c_mux;
            for (int i = 0; i < 63; i++) gMux1S(q, 0, 0, 0);
            gNibbles2Bytes(q, 33, R.t_on);
            R.c_age = q.cur;
            gAssertGreaterEqThanS(q, 16, 0, 2);
            R.c_acc = q.cur;
            { S al; gRlpEmptyAccount(q, prm.amountBytes, fr_zero(), al); }
            R.c_concat = q.cur;
            { S cl; gConcat(q, 2 + 1 + 33, 2 + 4 + prm.amountBytes + 66, R.pk, 0, R.val, 0, cl); }
            expect_cursor("RlpMerklePatriciaTrieLeaf", q.cur, chk.cur);
            {   // RlpEmptyAccount: heavy part A (stage 6), CountBytes/ShiftLeft B (7), assembly C (8)
                const Cur keep = p.cur;
                p.cur = R.c_acc;
                CountP qa; qa.cur = R.c_acc; { S al; gRlpEmptyAccount(qa, prm.amountBytes, fr_zero(), al); }
                unit(U_RL_ACC, TC + 1);                             // depends on the balance input only: its own track
                expect_cursor("RlpEmptyAccount", p.cur, qa.cur);
                p.cur = keep;
                record_composite(U_RL_ACC_B, TC + 2, L.ra.c_cb);
                record_composite(U_RL_ACC_C, TC + 3, L.ra.c_concat);          // long serial unit
                cat_units(TC + 3, L.ra.c_concat, 4 + prm.amountBytes, 66);
            }
            record_composite(U_RL_B, TR + 2, R.c_mux);
            record(CK_RL_B2, TR + 2, R.c_mux); record(CK_RL_B3, TR + 2, R.c_concat);
            cat_units(TR + 2, R.c_concat, 2 + 1 + 33, 2 + 4 + prm.amountBytes + 66);
            p.cur = chk.cur;
        }
        proof_of_work_checker(TB + 2);                         // :211 (inputs only; in step with BurnAddressHash so that the two sponges share their launches)
        unit(U_POB_FINAL, 10);
        total = p.cur;
        estimate_costs();
    }
    // A gadget-level main (gadget_mains.hpp; reference tests/test.py:146-201): template `tid` with parameters prm[0..nprm).  Returns null or
    // the reason the instantiation is refused.  Mains without a sponge are ONE unit (stage 1); the four Keccak mains are planned from the split
    // units of the production circuits, with the template's own input wires written from the packed inputs by a U_GM_INPUT unit first.
    const char* plan_gadget(uint32_t tid, const int* prm, int nprm) {
        memset(&L, 0, sizeof L);
        L.circuit = 2; L.nkb = 0; max_stage = 0; ntracks = 1; L.decl_order = p.decl_order;
        L.pob = PobParams{1, 1, 1, 0, 31, 0, fr_zero(), fr_zero()};
        L.fp_n2be32 = n2be_footprint(32); L.fp_n2beN = n2be_footprint(31);
        L.gm.tid = tid;
        const GmName* nm = nullptr;
        for (const GmName& g : GM_NAMES) if (g.id == tid) nm = &g;
        if (!nm) return "unknown gadget template";
        if (nprm != nm->nparams) return "wrong number of template parameters";
        const int a = nprm > 0 ? prm[0] : 0, b = nprm > 1 ? prm[1] : 0, c = nprm > 2 ? prm[2] : 0, d = nprm > 3 ? prm[3] : 0;
        auto in = [](long v, long lo, long hi) { return v >= lo && v <= hi; };
        bool ok = true;
        switch (tid) {
        case GM_CONCAT_FIXED4: ok = in(a, 0, 4096) && in(b, 0, 4096) && in(c, 0, 4096) && in(d, 0, 4096) && a + b + c + d >= 1; break;
        case GM_PUBLIC_COMMITMENT: ok = in(a, 1, 16); break;
        case GM_POSEIDON: ok = in(a, 2, 4); break;
        case GM_DIVIDE: case GM_ASSERT_LESS_EQ_THAN: case GM_ASSERT_LESS_THAN: case GM_ASSERT_GREATER_EQ_THAN: case GM_IS_IN_RANGE: ok = in(a, 1, 30); break;   // int32 signals
        case GM_SUBSTRING_CHECK: ok = in(b, 1, 31) && in(a, b, 4096); break;
        case GM_SHIFT_LEFT: ok = in(a, 1, 256); break;
        case GM_SHIFT_RIGHT: ok = in(a, 1, 4096) && in(b, 0, 4096) && in((long)a * (b + 1), 1, 1 << 20); break;
        case GM_MASK: case GM_SELECTOR: case GM_BYTES2NIBBLES: case GM_NIBBLES2BYTES: case GM_ASSERT_BYTE_STRING: case GM_FILTER: case GM_REVERSE:
        case GM_COUNT_BYTES: ok = in(a, 1, 4096); break;
        case GM_CONCAT: ok = in(a, 1, 2048) && in(b, 1, 2048); break;
        case GM_SELECTOR_ARRAY_1D: ok = in(a, 1, 4096) && in(b, 1, 4096) && in((long)a * b, 1, 1 << 16); break;
        case GM_SELECTOR_ARRAY_2D: ok = in(a, 1, 4096) && in(b, 1, 4096) && in(c, 1, 4096) && in((long)a * b * c, 1, 1 << 16); break;
        case GM_BIG_ENDIAN_BYTES2NUM: case GM_LITTLE_ENDIAN_BYTES2NUM: case GM_RLP_INTEGER: case GM_RLP_EMPTY_ACCOUNT: ok = in(a, 1, 31); break;
        case GM_NUM2BIG_ENDIAN_BYTES: case GM_NUM2LITTLE_ENDIAN_BYTES: ok = in(a, 1, 32); break;
        case GM_NUM2BITS_SAFE: ok = in(a, 1, 256); break;
        case GM_PAD: ok = in(a, 1, 4096) && in(b, 1, 4096) && in((long)a * b, 1, 4096); break;
        case GM_KECCAK_BYTES: ok = in(a, 1, 32); break;
        case GM_ASSERT_BITS: ok = in(a, 1, 253); break;
        case GM_FIT: ok = in(a, 1, 4096) && in(b, 1, 4096); break;
        case GM_FLATTEN: case GM_RESHAPE: ok = in(a, 1, 4096) && in(b, 1, 4096) && in((long)a * b, 1, 1 << 16); break;
        case GM_TRUNCATED_ADDRESS_HASH: ok = in(a, 1, 32); break;
        case GM_LEAF_DETECTOR: ok = in(a, 3, 4096); break;
        case GM_RLP_MPT_LEAF: ok = in(a, 1, 32) && in(b, 1, 31); break;
        default: break;
        }
        if (!ok) return "template parameters outside the range the gadget-main path supports";
        p.cur = Cur{1, 0, 0, 0, 0};                     // wire 0 = constant 1; the main component's block starts at wire 1
        nfr_in = nsm_in = 0;
        if (tid == GM_KECCAK_BYTES) {                  // keccak.circom:454-489  [out[32] | in[136 mb], inLen | ...]
            const uint32_t m = 136u * (uint32_t)a;
            const SmRef own_in = {1 + 32, 32}, own_len = own_in + m;
            L.nkb = 1;
            keccak_bytes(0, a, 1, own_in, own_len, SmRef{0, 0}, m + 1, false);
            const KBRefs& r = L.kbs[0];
            if (r.in.w != own_in.w || r.in.i != own_in.i || r.inLen.w != own_len.w || r.inLen.i != own_len.i) throw std::runtime_error("layout planner: KeccakBytes main inputs");
            record(U_GM_INPUT, 1, Cur{1, 0, 0, 0, 0}, 0, 0, 0, own_in.w, own_in.i, m + 1);
            L.gm.nfr_in = 0; L.gm.nsm_in = m + 1; L.gm.nout = 32;
        } else if (tid == GM_PUBLIC_COMMITMENT) {      // public_commitment.circom:18-42  [out | in[N][32] | ...]
            public_commitment(a, 2);
            record(U_GM_INPUT, 1, Cur{1, 0, 0, 0, 0}, 0, 0, 0, L.pc.in.w, L.pc.in.i, 32u * (uint32_t)a);
            L.gm.nfr_in = 0; L.gm.nsm_in = 32u * (uint32_t)a; L.gm.nout = 1;
        } else if (tid == GM_BURN_ADDRESS_HASH) {      // burn_address.circom:67-84  [addressHashNibbles[64] | burnKey, revealAmount, burnExtraCommitment | ...]
            burn_address_hash(2);                       // (the Poseidon block runs as a U_POS_WIDE unit in stage 1, behind the inputs of stage 0)
            record(U_GM_INPUT, 0, Cur{1, 0, 0, 0, 0}, L.bah.in.w, L.bah.in.i, 3, 0, 0, 0);
            L.gm.nfr_in = 3; L.gm.nsm_in = 0; L.gm.nout = 64;
        } else if (tid == GM_PROOF_OF_WORK_CHECKER) {  // proof_of_work.circom:54-81  [ | burnKey, revealAmount, burnExtraCommitment, minimumZeroBytes | ...]
            proof_of_work_checker(2);
            record(U_GM_INPUT, 0, Cur{1, 0, 0, 0, 0}, L.pw.in.w, L.pw.in.i, 3, L.pw.mzb.w, L.pw.mzb.i, 1);
            L.gm.nfr_in = 3; L.gm.nsm_in = 1; L.gm.nout = 0;
        } else unit(U_GM, 1, tid, (uint32_t)a, (uint32_t)b, (uint32_t)c, (uint32_t)d);
        nfr_in = L.gm.nfr_in; nsm_in = L.gm.nsm_in;
        total = p.cur;
        estimate_costs();
        return nullptr;
    }
    void plan_spend(const SpendParams& prm) {
        memset(&L, 0, sizeof L);
        L.circuit = 1; L.spend = prm; L.nkb = 0; max_stage = 0; ntracks = 1; L.decl_order = p.decl_order;
        L.fp_n2be32 = n2be_footprint(32); L.fp_n2beN = n2be_footprint(prm.maxAmountBytes);
        L.pob = PobParams{1, 1, 1, 0, prm.maxAmountBytes, 0, fr_zero(), fr_zero()};
        SpendMain& M = L.sm;
        p.cur = Cur{1, 0, 0, 0};
        M.commitment = p.frs(1); M.burnKey = p.frs(1); M.balance = p.frs(1); M.withdrawnBalance = p.frs(1); M.extraCommitment = p.frs(1);
        M.coin = p.frs(1); M.remainingCoin = p.frs(1); M.coinBytes = p.sms(32); M.withdrawnBalanceBytes = p.sms(32); M.remainingCoinBytes = p.sms(32);
        M.extraCommitmentBytes = p.sms(32);
        nfr_in = 4; nsm_in = 0;
        unit(U_SP_INPUT, 0);
        unit(U_SP_HEAD, 2);                                       // (the two Poseidon blocks: U_POS_WIDE units in stage 1)
        public_commitment(4, 3);
        total = p.cur;
        estimate_costs();
    }
};
