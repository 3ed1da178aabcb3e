// Top-level circuits (ProofOfBurn, Spend) and the Keccak wrapper, cut into UNITS.
//
// A unit = a run of consecutive sub-components of one template that one wavefront (64 witnesses)
// executes start to finish; its UnitDesc carries the O0 cursor where its first wire lives.  The same
// `unit_run<P>` body serves the host planner (P = CountP: advances the cursor, fills the descriptors),
// the generator (GenP), the constraint evaluator (CheckP) and the .wtns emitter (EmitP).
// Units of one STAGE are independent; a unit only reads wires written in earlier stages (or by itself).
// Keccak sponges run between G stages on the bit-sliced kernels (keccak_kernels.hpp).
#pragma once
#include "gadgets.hpp"
#include "poseidon_consts.h"

#define KECCAKF_ROUND_WIRES 102656u
#define KECCAKF_WIRES (43200u + 24u * KECCAKF_ROUND_WIRES)        // 2 506 944
#define ABSORB_OWN 5888u                                            // out, s, block[17], aux
#define ABSORB_WIRES (ABSORB_OWN + 17u * 384u + KECCAKF_WIRES)     // 2 519 360

enum UnitKind : uint32_t {
    U_POB_INPUT = 1, U_POB_RANGE, U_POB_LAYER_ASSERT, U_POB_HDR_ASSERT, U_POB_POSEIDONS, U_BAH_PRE, U_BAH_POST,
    U_KB_PRE, U_KB_SELROW, U_KB_POST, U_POB_N2B, U_PC_PRE, U_PC_POST, U_POB_LASTLAYER, U_POB_LASTLEN, U_POB_LEAF,
    U_POB_LAYER_POST, U_POB_LASTLEAF, U_POB_RLPLEAF, U_POW_PRE, U_POW_POST, U_POB_FINAL,
    U_SP_INPUT, U_SP_HEAD, U_SP_FINAL
};

struct PobParams { int L, NB, HB, minNib, amountBytes, powZero; Fr maxIntended, maxActual; };   // Montgomery
struct SpendParams { int maxAmountBytes; };

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
// KeccakBytes own wires + the Keccak/Final/SelectorArray2D wires its G units touch
struct KBRefs {
    SmRef out, in, inLen, padded, numBlocks; BitRef inBitsArray, inBits, inBlocks, outBits, outBytes;
    BitRef k_out, k_in; SmRef k_blocks; BitRef k_finalState, f_out, f_in; SmRef f_blocks; BitRef f_s;
    uint32_t abs_w, abs_b;
    BitRef sel_out, sel_arrays; SmRef sel_select; BitRef sel_T;
    uint32_t mb, index;
};
struct SpongeDesc { uint32_t n, stage, src_b, kin_b, fin_b, fs_b, abs_b, kin_w, fin_w, fs_w, abs_w, src_w; };

struct UnitDesc { uint32_t kind, stage; Cur cur; uint32_t a[6]; };

// everything a unit body needs besides the policy; lives in device memory, read-only
struct CircuitLayout {
    int circuit;                  // 0 = ProofOfBurn, 1 = Spend
    PobParams pob; SpendParams spend;
    PobMain pm; SpendMain sm;
    Fr prefix[3];                 // POSEIDON_PREFIX + 0/1/2 (constants.circom:3-14), Montgomery
    // unit-structured sub-templates (refs to their own wires)
    struct { SmRef nibbles; FrRef in; SmRef addressBytes, block, hash; uint32_t kb; } bah;
    struct { FrRef out; SmRef in, flat, block, hash, reduced; uint32_t kb; int N, nb; } pc;
    struct { FrRef in; SmRef mzb, keyBytes, raBytes, becBytes, eip, hin, block, keccak; BitRef sbz; uint32_t kb; } pw;
    uint32_t kb_hdr, kb_layer0;   // indices into kbs[]
    uint32_t nkb;
};
#define MAX_KB 24

HD PosOff pos_off(int t) {
    PosOff k;
    if (t == 3) { k.C = POS_OFF_C_3; k.S = POS_OFF_S_3; k.M = POS_OFF_M_3; k.Pm = POS_OFF_P_3; k.rp = POS_RP_3; }
    else if (t == 4) { k.C = POS_OFF_C_4; k.S = POS_OFF_S_4; k.M = POS_OFF_M_4; k.Pm = POS_OFF_P_4; k.rp = POS_RP_4; }
    else { k.C = POS_OFF_C_5; k.S = POS_OFF_S_5; k.M = POS_OFF_M_5; k.Pm = POS_OFF_P_5; k.rp = POS_RP_5; }
    return k;
}

// ---------------------------------------------------------------------------- keccak.circom: Pad / KeccakBytes
// Pad(mb, 136) :412-446  [out[m], numBlocks | in[m], inLen | div, rem, filter[m+1], isEq[m], isLast[m]]
// || Divide(16)(inLen, 136), AssertLessEqThan(16)(numBlocks, mb), IsEqual([i, inLen]) x m, IsEqual([i, numBlocks*136-1]) x m
template <class P> GD SmRef gPad(P& p, int mb, SmRef src, S inLen, S& numBlocks) {
    const int m = 136 * mb;
    SmRef o = p.sms(m), nbr = p.sms(1), in = p.sms(m), il = p.sms(1), dv = p.sms(1), rm = p.sms(1);
    BitRef flt = p.bits(m + 1), isEq = p.bits(m), isLast = p.bits(m);
    for (int i = 0; i < m; i++) p.put(in + i, p.get(src + i));
    inLen = p.put(il, inLen);
    S q, r;
    gDivide(p, 16, inLen, (S)136, q, r);
    q = p.put(dv, q); p.put(rm, r);
    S nb = p.put(nbr, q + 1);
    gAssertLessEqThanS(p, 16, nb, (S)mb);
    B f = p.put(flt, ~(B)0);
    for (int i = 0; i < m; i++) {
        B e = p.put(isEq + i, gIsEqualS(p, (S)i, inLen));
        f = p.put(flt + i + 1, f & ~e);
    }
    for (int i = 0; i < m; i++) {
        B l = p.put(isLast + i, gIsEqualS(p, (S)i, (S)(nb * 136 - 1)));
        S v = (p.bit(p.get(flt + i + 1)) ? p.get(in + i) : 0) + (S)p.bit(p.get(isEq + i)) + (p.bit(l) ? 0x80 : 0);
        p.put(o + i, v);
    }
    numBlocks = nb;
    return o;
}
// KeccakBytes(mb) :454-489, part before the sponge:
// [out[32] | in[m], inLen | padded[m], numBlocks, inBitsArray[m][8], inBits[8m], inBlocks[mb][17][64], outBits[256], outBytes[32][8]]
// || AssertLessThan(16)(inLen, m), Pad, Num2Bits(8) x m, Flatten(m,8), [Keccak(mb)], ...
template <class P> GD void kb_pre(P& p, int mb, SmRef src, S inLen, KBRefs& r) {
    const int m = 136 * mb;
    r.mb = mb;
    r.out = p.sms(32); r.in = p.sms(m); r.inLen = p.sms(1); r.padded = p.sms(m); r.numBlocks = p.sms(1);
    r.inBitsArray = p.bits(8 * m); r.inBits = p.bits(8 * m); r.inBlocks = p.bits(8 * m); r.outBits = p.bits(256); r.outBytes = p.bits(256);
    for (int i = 0; i < m; i++) p.put(r.in + i, p.get(src + i));
    inLen = p.put(r.inLen, inLen);
    gAssertLessThanS(p, 16, inLen, (S)m);
    S nb;
    SmRef po = gPad(p, mb, r.in, inLen, nb);
    p.put(r.numBlocks, nb);
    for (int i = 0; i < m; i++) {
        BitRef b = gNum2BitsS(p, 8, p.put(r.padded + i, p.get(po + i)));
        for (int k = 0; k < 8; k++) p.put(r.inBitsArray + (8 * i + k), p.get(b + k));
    }
    BitRef f = gFlattenB(p, 8 * m, r.inBitsArray);
    for (int j = 0; j < 8 * m; j++) p.put(r.inBlocks + j, p.put(r.inBits + j, p.get(f + j)));
}
template <class P> GD void kb_pre_at(P& p, int mb, SmRef src, S inLen, KBRefs* kbs, uint32_t idx) {
    KBRefs r = kbs[idx];
    kb_pre(p, mb, src, inLen, r);
    if (P::is_count) kbs[idx] = r;      // only the host planner records the references
}
// Keccak(n) :374-385 / Final(n) :330-349 own wires + the n Absorb blocks (K kernels) + SelectorArray2D own wires.
template <class P> GD void kb_declare_keccak(P& p, KBRefs& r) {
    const uint32_t n = r.mb;
    r.k_out = p.bits(256); r.k_in = p.bits(n * 1088); r.k_blocks = p.sms(1); r.k_finalState = p.bits(1600);
    r.f_out = p.bits(1600); r.f_in = p.bits(n * 1088); r.f_blocks = p.sms(1); r.f_s = p.bits((n + 1) * 1600);
    r.abs_w = p.cur.w; r.abs_b = p.cur.b;
    p.skip_bits(n * ABSORB_WIRES);
    r.sel_out = p.bits(1600); r.sel_arrays = p.bits((n + 1) * 1600); r.sel_select = p.sms(1); r.sel_T = p.bits(1600 * (n + 1));
}
// one row (64 selectors) of SelectorArray2D(n+1, 25, 64) (selector.circom:91-111) + the copies of its outputs
template <class P> GD void kb_selrow(P& p, const KBRefs& r, int row) {
    const int n1 = r.mb + 1;
    S blocks = p.get(r.numBlocks);
    for (int j = 0; j < 64; j++) {
        const uint32_t idx = row * 64 + j;
        for (int k = 0; k < n1; k++) p.put(r.sel_T + (idx * n1 + k), p.put(r.sel_arrays + (k * 1600 + idx), p.get(r.f_s + (k * 1600 + idx))));
        B o = gSelectorB(p, n1, r.sel_T + idx * n1, blocks);
        o = p.put(r.sel_out + idx, o);
        o = p.put(r.f_out + idx, o);
        o = p.put(r.k_finalState + idx, o);
        if (idx < 256) p.put(r.k_out + idx, o);
    }
}
// part after the sponge: Reshape(32,8), Bits2Num(8) x 32 and the copy into the parent's array
template <class P> GD void kb_post(P& p, const KBRefs& r, SmRef dst, bool has_dst) {
    const int n1 = r.mb + 1;
    S nb = p.get(r.numBlocks);
    p.put(r.k_blocks, nb); p.put(r.f_blocks, nb); p.put(r.sel_select, nb);
    for (int j = 0; j < 256; j++) {
        B v;
        if (P::is_gen) { v = 0; for (int k = 0; k < n1; k++) v |= p.ballot(nb == k) & p.get(r.f_s + (k * 1600 + j)); }   // s[blocks] (:348)
        else v = p.get(r.k_out + j);
        p.put(r.outBits + j, v);
    }
    BitRef rs = gFlattenB(p, 256, r.outBits);
    for (int j = 0; j < 256; j++) p.put(r.outBytes + j, p.get(rs + j));
    for (int i = 0; i < 32; i++) {
        S by = p.put(r.out + i, gBits2Num8(p, r.outBytes + 8 * i));
        if (has_dst) p.put(dst + i, by);
    }
}

// ---------------------------------------------------------------------------- unit bodies
template <class P> GD void unit_run(P& p, const UnitDesc& d, const CircuitLayout& L, KBRefs* kbs) {
    const PobMain& M = L.pm;
    const PobParams& prm = L.pob;
    const int LB = 136 * prm.NB, HBy = 136 * prm.HB;
    p.cur = d.cur;
    switch (d.kind) {
    case U_POB_INPUT: {   // main inputs from the packed batch buffer (FR inputs 0..5, SM inputs in declaration order)
        p.put(M.burnKey, p.input_fr(0)); p.put(M.actualBalance, p.input_fr(1)); p.put(M.intendedBalance, p.input_fr(2));
        p.put(M.revealAmount, p.input_fr(3)); p.put(M.burnExtraCommitment, p.input_fr(4)); p.put(M.proofExtraCommitment, p.input_fr(5));
        uint32_t k = 0;
        p.put(M.numLeafAddressNibbles, p.input_sm(k++));
        for (int i = 0; i < prm.L * LB; i++) p.put(M.layers + i, p.input_sm(k++));
        for (int i = 0; i < prm.L; i++) p.put(M.layerLens + i, p.input_sm(k++));
        p.put(M.numLayers, p.input_sm(k++));
        for (int i = 0; i < HBy; i++) p.put(M.blockHeader + i, p.input_sm(k++));
        p.put(M.blockHeaderLen, p.input_sm(k++));
        p.put(M.byteSecurityRelax, p.input_sm(k++));
    } break;
    case U_POB_RANGE: {   // proof_of_burn.circom:84-97
        const int AB8 = prm.amountBytes * 8;
        F intended = p.get(M.intendedBalance), actual = p.get(M.actualBalance), reveal = p.get(M.revealAmount);
        gAssertLessEqThanF(p, AB8, intended, prm.maxIntended);
        gAssertLessEqThanF(p, AB8, actual, prm.maxActual);
        gAssertLessEqThanF(p, AB8, intended, actual);
        S relax = p.get(M.byteSecurityRelax);
        gAssertLessEqThanS(p, 16, (S)((uint32_t)relax * 2u), (S)prm.minNib);
        gAssertGreaterEqThanS(p, 16, p.get(M.numLeafAddressNibbles), (S)((uint32_t)prm.minNib - (uint32_t)relax * 2u));
        gAssertBitsF(p, AB8, reveal);
        gAssertLessEqThanF(p, AB8, reveal, intended);
    } break;
    case U_POB_LAYER_ASSERT: {   // :99-103
        const int i = d.a[0];
        gAssertLessThanS(p, 16, p.get(M.layerLens + i), (S)(LB * 8));
        gAssertByteString(p, LB, M.layers + i * LB);
    } break;
    case U_POB_HDR_ASSERT: {     // :105-106, stateRoot copy :125-129
        gAssertLessThanS(p, 16, p.get(M.blockHeaderLen), (S)(HBy * 8));
        gAssertByteString(p, HBy, M.blockHeader);
        for (int i = 0; i < 32; i++) p.put(M.stateRoot + i, p.get(M.blockHeader + 91 + i));
    } break;
    case U_POB_POSEIDONS: {      // :113, :116
        F bk = p.get(M.burnKey);
        F in3[3] = {L.prefix[2], bk, fr_sub(p.get(M.intendedBalance), p.get(M.revealAmount))};
        p.put(M.remainingCoin, gPoseidon<P, 4>(p, pos_off(4), in3));
        F in2[2] = {L.prefix[1], bk};
        p.put(M.nullifier, gPoseidon<P, 3>(p, pos_off(3), in2));
    } break;
    case U_BAH_PRE: {            // BurnAddressHash burn_address.circom:67-79 up to the sponge
        F bk = p.put(L.bah.in, p.get(M.burnKey)), ra = p.put(L.bah.in + 1, p.get(M.revealAmount)), bec = p.put(L.bah.in + 2, p.get(M.burnExtraCommitment));
        SmRef ab = gBurnAddress(p, pos_off(5), L.prefix[0], bk, ra, bec);
        for (int i = 0; i < 20; i++) p.put(L.bah.addressBytes + i, p.get(ab + i));
        SmRef f = gFitS(p, 20, 136, L.bah.addressBytes);
        for (int i = 0; i < 136; i++) p.put(L.bah.block + i, p.get(f + i));
        kb_pre_at(p, 1, L.bah.block, (S)20, kbs, L.bah.kb);
    } break;
    case U_BAH_POST: {           // :82 Bytes2Nibbles(32) + main.addressHashNibbles (:119)
        SmRef nb = gBytes2Nibbles(p, 32, L.bah.hash);
        for (int i = 0; i < 64; i++) p.put(M.addressHashNibbles + i, p.put(L.bah.nibbles + i, p.get(nb + i)));
    } break;
    case U_KB_PRE: {             // a[0] = kb index, a[1] = src SM ref (wire, idx), a[3] = len ref
        SmRef src = {d.a[1], d.a[2]}, len = {d.a[3], d.a[4]};
        kb_pre_at(p, kbs[d.a[0]].mb, src, p.get(len), kbs, d.a[0]);
    } break;
    case U_KB_SELROW: kb_selrow(p, kbs[d.a[0]], d.a[1]); break;
    case U_KB_POST: {
        SmRef dst = {d.a[1], d.a[2]};
        kb_post(p, kbs[d.a[0]], dst, d.a[3] != 0);
    } break;
    case U_POB_N2B: {            // :132-136  Num2BigEndianBytes(32) of nullifier, remainingCoin, revealAmount, burnExtraCommitment, _proofExtraCommitment
        const int j = d.a[0];
        FrRef src = j == 0 ? M.nullifier : j == 1 ? M.remainingCoin : j == 2 ? M.revealAmount : j == 3 ? M.burnExtraCommitment : M.proofExtraCommitment;
        SmRef dst = j == 0 ? M.nullifierBytes : j == 1 ? M.remainingCoinBytes : j == 2 ? M.revealAmountBytes : j == 3 ? M.burnExtraCommitmentBytes : M.extraCommitmentBytes;
        SmRef r = gNum2BigEndianBytesF(p, 32, p.get(src));
        for (int i = 0; i < 32; i++) p.put(dst + i, p.get(r + i));
    } break;
    case U_PC_PRE: {             // PublicCommitment(N) public_commitment.circom:18-36 up to the sponge
        const int N = L.pc.N;
        for (int j = 0; j < N; j++) {
            SmRef src;
            if (L.circuit == 0) src = j == 0 ? M.blockRoot : j == 1 ? M.nullifierBytes : j == 2 ? M.remainingCoinBytes : j == 3 ? M.revealAmountBytes : j == 4 ? M.burnExtraCommitmentBytes : M.extraCommitmentBytes;
            else src = j == 0 ? L.sm.coinBytes : j == 1 ? L.sm.withdrawnBalanceBytes : j == 2 ? L.sm.remainingCoinBytes : L.sm.extraCommitmentBytes;
            for (int i = 0; i < 32; i++) p.put(L.pc.in + (32 * j + i), p.get(src + i));
        }
        for (int j = 0; j < N; j++) gAssertByteString(p, 32, L.pc.in + 32 * j);
        SmRef f = gFlattenS(p, 32 * N, L.pc.in);
        for (int i = 0; i < 32 * N; i++) p.put(L.pc.flat + i, p.get(f + i));
        f = gFitS(p, 32 * N, 136 * L.pc.nb, L.pc.flat);
        for (int i = 0; i < 136 * L.pc.nb; i++) p.put(L.pc.block + i, p.get(f + i));
        kb_pre_at(p, L.pc.nb, L.pc.block, (S)(32 * N), kbs, L.pc.kb);
    } break;
    case U_PC_POST: {            // :40-41 Fit(32,31), BigEndianBytes2Num(31); commitment (proof_of_burn.circom:137 / spend.circom:50)
        SmRef f = gFitS(p, 32, 31, L.pc.hash);
        for (int i = 0; i < 31; i++) p.put(L.pc.reduced + i, p.get(f + i));
        F c = p.put(L.pc.out, gBigEndianBytes2NumF(p, 31, L.pc.reduced));
        p.put(L.circuit == 0 ? M.commitment : L.sm.commitment, c);
    } break;
    case U_POB_LASTLAYER: {      // :142-143
        SmRef r = gSelectorArray1D(p, prm.L, LB, M.layers, p.get(M.numLayers) - 1);
        for (int i = 0; i < LB; i++) p.put(M.lastLayer + i, p.get(r + i));
    } break;
    case U_POB_LASTLEN: {        // :146, :150
        S nl = p.get(M.numLayers);
        p.put(M.lastLayerLen, gSelectorS(p, prm.L, M.layerLens, nl - 1));
        BitRef f = gFilter(p, prm.L, nl);
        for (int i = 0; i < prm.L; i++) p.put(M.layerExists + i, p.get(f + i));
    } break;
    case U_POB_LEAF: {           // :159
        const int i = d.a[0];
        p.put(M.isLeaf + i, gLeafDetector(p, LB, M.layers + i * LB, p.get(M.layerLens + i)));
    } break;
    case U_POB_LAYER_POST: {     // :166-180  Fit(32,31), SubstringCheck, (1-sc)*exists === 0
        const int i = d.a[0];
        SmRef f = gFitS(p, 32, 31, M.layerKeccaks + 32 * i);
        for (int k = 0; k < 31; k++) p.put(M.reducedLayerKeccaks + (31 * i + k), p.get(f + k));
        if (i > 0) {
            B sc = p.put(M.substringCheckers + (i - 1), gSubstringCheck(p, LB, 31, M.layers + (i - 1) * LB, p.get(M.layerLens + (i - 1)), M.reducedLayerKeccaks + 31 * i));
            p.require(sc | ~p.get(M.layerExists + i), FAILCODE(T_POB, 179));
        }
    } break;
    case U_POB_LASTLEAF:         // :187
        p.put(M.isLastLayerLeaf, gLeafDetector(p, LB, M.lastLayer, p.get(M.lastLayerLen)));
        break;
    case U_POB_RLPLEAF: {        // :198-200
        S ll;
        SmRef r = gRlpMptLeaf(p, 32, prm.amountBytes, M.addressHashNibbles, p.get(M.numLeafAddressNibbles), p.get(M.actualBalance), ll);
        for (int i = 0; i < 139; i++) p.put(M.leaf + i, p.get(r + i));
        p.put(M.leafLen, ll);
    } break;
    case U_POW_PRE: {            // ProofOfWorkChecker proof_of_work.circom:54-71 up to the sponge
        F bk, ra, bec;
        if (L.circuit == 0) { bk = p.get(M.burnKey); ra = p.get(M.revealAmount); bec = p.get(M.burnExtraCommitment); }
        else { bk = ra = bec = fr_zero(); }
        bk = p.put(L.pw.in, bk); ra = p.put(L.pw.in + 1, ra); bec = p.put(L.pw.in + 2, bec);
        p.put(L.pw.mzb, (S)((uint32_t)prm.powZero + (uint32_t)p.get(M.byteSecurityRelax)));
        SmRef r = gNum2BigEndianBytesF(p, 32, bk);
        for (int i = 0; i < 32; i++) p.put(L.pw.keyBytes + i, p.get(r + i));
        r = gNum2BigEndianBytesF(p, 32, ra);
        for (int i = 0; i < 32; i++) p.put(L.pw.raBytes + i, p.get(r + i));
        r = gNum2BigEndianBytesF(p, 32, bec);
        for (int i = 0; i < 32; i++) p.put(L.pw.becBytes + i, p.get(r + i));
        SmRef e = p.sms(8);                              // EIP7503 :11-21  [out[8]]
        const char tag[9] = "EIP-7503";
        for (int i = 0; i < 8; i++) p.put(L.pw.eip + i, p.put(e + i, (S)tag[i]));
        SmRef co = p.sms(104), ci = p.sms(104);          // ConcatFixed4(32,32,32,8) :28-48  [out | a,b,c,d]
        for (int i = 0; i < 104; i++) {
            SmRef s = i < 32 ? L.pw.keyBytes + i : i < 64 ? L.pw.raBytes + (i - 32) : i < 96 ? L.pw.becBytes + (i - 64) : L.pw.eip + (i - 96);
            p.put(L.pw.hin + i, p.put(co + i, p.put(ci + i, p.get(s))));
        }
        SmRef f = gFitS(p, 104, 136, L.pw.hin);
        for (int i = 0; i < 136; i++) p.put(L.pw.block + i, p.get(f + i));
        kb_pre_at(p, 1, L.pw.block, (S)104, kbs, L.pw.kb);
    } break;
    case U_POW_POST: {           // :73-79
        BitRef f = gFilter(p, 32, p.get(L.pw.mzb));
        for (int i = 0; i < 32; i++) {
            B z = p.put(L.pw.sbz + i, p.get(f + i));
            p.require(p.ballot(p.get(L.pw.keccak + i) == 0) | ~z, FAILCODE(T_POW, 79));
        }
    } break;
    case U_POB_FINAL: {          // :186, :188, :191-193, :203-206
        S cnt = 0;
        for (int i = 0; i < prm.L; i++) cnt += (S)p.bit(p.get(M.isLeaf + i));
        p.require(p.ballot(cnt == 1), FAILCODE(T_POB, 186));
        p.require(p.get(M.isLastLayerLeaf), FAILCODE(T_POB, 188));
        bool ok = true;
        for (int i = 0; i < 32; i++) ok = ok && (p.get(M.layerKeccaks + i) == p.get(M.stateRoot + i));
        p.require(p.ballot(ok), FAILCODE(T_POB, 192));
        ok = true;
        for (int i = 0; i < 139; i++) ok = ok && (p.get(M.leaf + i) == p.get(M.lastLayer + i));
        p.require(p.ballot(ok), FAILCODE(T_POB, 204));
        p.require(p.ballot(p.get(M.leafLen) == p.get(M.lastLayerLen)), FAILCODE(T_POB, 206));
    } break;
    case U_SP_INPUT: {
        p.put(L.sm.burnKey, p.input_fr(0)); p.put(L.sm.balance, p.input_fr(1));
        p.put(L.sm.withdrawnBalance, p.input_fr(2)); p.put(L.sm.extraCommitment, p.input_fr(3));
    } break;
    case U_SP_HEAD: {            // spend.circom:41-49
        F bk = p.get(L.sm.burnKey), bal = p.get(L.sm.balance), wd = p.get(L.sm.withdrawnBalance), ec = p.get(L.sm.extraCommitment);
        gAssertGreaterEqThanF(p, L.spend.maxAmountBytes * 8, bal, wd);
        F in3[3] = {L.prefix[2], bk, bal};
        F coin = p.put(L.sm.coin, gPoseidon<P, 4>(p, pos_off(4), in3));
        in3[2] = fr_sub(bal, wd);
        F rc = p.put(L.sm.remainingCoin, gPoseidon<P, 4>(p, pos_off(4), in3));
        SmRef r = gNum2BigEndianBytesF(p, 32, coin);
        for (int i = 0; i < 32; i++) p.put(L.sm.coinBytes + i, p.get(r + i));
        r = gNum2BigEndianBytesF(p, 32, wd);
        for (int i = 0; i < 32; i++) p.put(L.sm.withdrawnBalanceBytes + i, p.get(r + i));
        r = gNum2BigEndianBytesF(p, 32, rc);
        for (int i = 0; i < 32; i++) p.put(L.sm.remainingCoinBytes + i, p.get(r + i));
        r = gNum2BigEndianBytesF(p, 32, ec);
        for (int i = 0; i < 32; i++) p.put(L.sm.extraCommitmentBytes + i, p.get(r + i));
    } break;
    default: break;
    }
}

// ---------------------------------------------------------------------------- host planner
#include <vector>
struct Plan {
    CircuitLayout L;
    std::vector<UnitDesc> units;
    std::vector<SpongeDesc> sponges;
    KBRefs kbs[MAX_KB];
    Cur total;            // = counts (total.w = nWitness)
    uint32_t nfr_in, nsm_in, max_stage;
    CountP p;

    void unit(uint32_t kind, uint32_t stage, uint32_t a0 = 0, uint32_t a1 = 0, uint32_t a2 = 0, uint32_t a3 = 0, uint32_t a4 = 0) {
        UnitDesc d; d.kind = kind; d.stage = stage; d.cur = p.cur; d.a[0] = a0; d.a[1] = a1; d.a[2] = a2; d.a[3] = a3; d.a[4] = a4; d.a[5] = 0;
        units.push_back(d);
        unit_run(p, d, L, kbs);            // CountP: advances p.cur over the unit's wires, fills kbs[] refs
        if (stage > max_stage) max_stage = stage;
    }
    // KeccakBytes whose "pre" part already ran inside the current unit: sponge + selector rows + post
    void keccak_tail(uint32_t kb, uint32_t pre_stage, SmRef dst, bool has_dst) {
        KBRefs& r = kbs[kb];
        r.index = kb;
        kb_declare_keccak(p, r);
        SpongeDesc s; s.n = r.mb; s.stage = pre_stage + 1; s.src_b = r.inBlocks.i; s.src_w = r.inBlocks.w;
        s.kin_b = r.k_in.i; s.kin_w = r.k_in.w; s.fin_b = r.f_in.i; s.fin_w = r.f_in.w; s.fs_b = r.f_s.i; s.fs_w = r.f_s.w; s.abs_b = r.abs_b; s.abs_w = r.abs_w;
        sponges.push_back(s);
        for (uint32_t row = 0; row < 25; row++) unit(U_KB_SELROW, pre_stage + 2, kb, row);
        unit(U_KB_POST, pre_stage + 2, kb, dst.w, dst.i, has_dst ? 1 : 0);
        if (pre_stage + 1 > max_stage) max_stage = pre_stage + 1;
    }
    void keccak_bytes(uint32_t kb, int mb, uint32_t pre_stage, SmRef src, SmRef len, SmRef dst) {
        kbs[kb].mb = mb;
        unit(U_KB_PRE, pre_stage, kb, src.w, src.i, len.w, len.i);
        keccak_tail(kb, pre_stage, dst, true);
    }
    void public_commitment(int N, uint32_t pre_stage) {   // public_commitment.circom:18-42
        L.pc.N = N; L.pc.nb = N * 32 / 136 + ((N * 32) % 136 != 0);
        L.pc.out = p.frs(1); L.pc.in = p.sms(32 * N); L.pc.flat = p.sms(32 * N); L.pc.block = p.sms(136 * L.pc.nb); L.pc.hash = p.sms(32); L.pc.reduced = p.sms(31);
        L.pc.kb = L.nkb++; kbs[L.pc.kb].mb = L.pc.nb;
        unit(U_PC_PRE, pre_stage);
        keccak_tail(L.pc.kb, pre_stage, L.pc.hash, true);
        unit(U_PC_POST, pre_stage + 3);
    }

    void plan_pob(const PobParams& prm) {
        L.circuit = 0; L.pob = prm; L.nkb = 0; max_stage = 0;
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

        unit(U_POB_INPUT, 0);
        unit(U_POB_RANGE, 1);
        for (int i = 0; i < Ln; i++) unit(U_POB_LAYER_ASSERT, 1, i);
        unit(U_POB_HDR_ASSERT, 1);
        unit(U_POB_POSEIDONS, 1);
        {   // BurnAddressHash :119
            L.bah.nibbles = p.sms(64); L.bah.in = p.frs(3); L.bah.addressBytes = p.sms(20); L.bah.block = p.sms(136); L.bah.hash = p.sms(32);
            L.bah.kb = L.nkb++;
            unit(U_BAH_PRE, 1);
            keccak_tail(L.bah.kb, 1, L.bah.hash, true);
            unit(U_BAH_POST, 4);
        }
        L.kb_hdr = L.nkb++;
        keccak_bytes(L.kb_hdr, prm.HB, 1, M.blockHeader, M.blockHeaderLen, M.blockRoot);       // :122
        for (int j = 0; j < 5; j++) unit(U_POB_N2B, 3, j);                                      // :132-136
        public_commitment(6, 4);                                                                // :137
        unit(U_POB_LASTLAYER, 1);
        unit(U_POB_LASTLEN, 1);
        L.kb_layer0 = L.nkb; L.nkb += Ln;
        for (int i = 0; i < Ln; i++) {                                                          // :157-181
            unit(U_POB_LEAF, 1, i);
            keccak_bytes(L.kb_layer0 + i, prm.NB, 1, M.layers + i * LB, M.layerLens + i, M.layerKeccaks + 32 * i);
            unit(U_POB_LAYER_POST, 4, i);
        }
        unit(U_POB_LASTLEAF, 3);                                                                // :187
        unit(U_POB_RLPLEAF, 5);                                                                 // :198
        {   // ProofOfWorkChecker :211
            L.pw.in = p.frs(3); L.pw.mzb = p.sms(1); L.pw.keyBytes = p.sms(32); L.pw.raBytes = p.sms(32); L.pw.becBytes = p.sms(32); L.pw.eip = p.sms(8);
            L.pw.hin = p.sms(104); L.pw.block = p.sms(136); L.pw.keccak = p.sms(32); L.pw.sbz = p.bits(32);
            L.pw.kb = L.nkb++;
            unit(U_POW_PRE, 1);
            keccak_tail(L.pw.kb, 1, L.pw.keccak, true);
            unit(U_POW_POST, 4);
        }
        unit(U_POB_FINAL, 8);
        total = p.cur;
    }
    void plan_spend(const SpendParams& prm) {
        L.circuit = 1; L.spend = prm; L.nkb = 0; max_stage = 0;
        L.pob = PobParams{1, 1, 1, 0, prm.maxAmountBytes, 0, fr_zero(), fr_zero()};
        SpendMain& M = L.sm;
        p.cur = Cur{1, 0, 0, 0};
        M.commitment = p.frs(1); M.burnKey = p.frs(1); M.balance = p.frs(1); M.withdrawnBalance = p.frs(1); M.extraCommitment = p.frs(1);
        M.coin = p.frs(1); M.remainingCoin = p.frs(1); M.coinBytes = p.sms(32); M.withdrawnBalanceBytes = p.sms(32); M.remainingCoinBytes = p.sms(32);
        M.extraCommitmentBytes = p.sms(32);
        nfr_in = 4; nsm_in = 0;
        unit(U_SP_INPUT, 0);
        unit(U_SP_HEAD, 1);
        public_commitment(4, 2);
        total = p.cur;
    }
};
