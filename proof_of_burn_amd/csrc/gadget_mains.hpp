// Gadget-level mains: the 54 entries of the reference's test list that instantiate ONE template as `component main`
// (tests/test.py:146-201; wrappers written by tests/test.py:24-33).  The main component IS the template, so wire 1 is its first
// output and its inputs are the packed inputs of the batch: SM input arrays are handed to the gadget functions (gadgets.hpp) as
// input references (policy.hpp GM_INPUT_W), scalar / field inputs as values.  One wavefront = one main for 64 test cases; the same
// body serves the planner (CountP), generation, constraint evaluation and .wtns emission like every other unit.
//
// Mains that contain a Keccak sponge (KeccakBytes, PublicCommitment, BurnAddressHash, ProofOfWorkChecker) are not here: they are
// planned from the split units of circuits.hpp (Plan::plan_gadget) so that the sponge runs on the bit-sliced kernels.
//
// SM inputs are int32 (the class the gadget functions compute these signals in): the host loader refuses larger values for them
// (witness.py); field-valued inputs (Poseidon, Num2Bits*, Num2*Bytes, AssertBits, RlpInteger / RlpEmptyAccount balances, BurnAddress) are FR.
#pragma once

enum GmTemplate : uint32_t {
    GM_NONE = 0, GM_EIP7503, GM_CONCAT_FIXED4, GM_PROOF_OF_WORK_CHECKER, GM_PUBLIC_COMMITMENT, GM_POSEIDON, GM_DIVIDE, GM_SUBSTRING_CHECK,
    GM_SHIFT_LEFT, GM_SHIFT_RIGHT, GM_MASK, GM_CONCAT, GM_SELECTOR, GM_SELECTOR_ARRAY_1D, GM_SELECTOR_ARRAY_2D, GM_BIG_ENDIAN_BYTES2NUM,
    GM_BYTES2NIBBLES, GM_LITTLE_ENDIAN_BYTES2NUM, GM_NUM2BIG_ENDIAN_BYTES, GM_NUM2LITTLE_ENDIAN_BYTES, GM_NIBBLES2BYTES, GM_NUM2BITS_SAFE,
    GM_PAD, GM_KECCAK_BYTES, GM_BURN_ADDRESS, GM_BURN_ADDRESS_HASH, GM_ASSERT_BITS, GM_ASSERT_BYTE_STRING, GM_ASSERT_LESS_EQ_THAN,
    GM_ASSERT_LESS_THAN, GM_ASSERT_GREATER_EQ_THAN, GM_FILTER, GM_FIT, GM_REVERSE, GM_FLATTEN, GM_RESHAPE, GM_RLP_INTEGER, GM_COUNT_BYTES,
    GM_RLP_EMPTY_ACCOUNT, GM_TRUNCATED_ADDRESS_HASH, GM_IS_IN_RANGE, GM_LEAF_DETECTOR, GM_RLP_MPT_LEAF, GM_TEMPLATE_COUNT
};
struct GmName { const char* name; uint32_t id; int nparams; };
// template name (what the reference writes after `component main =`) -> id, number of template parameters
static const GmName GM_NAMES[] = {
    {"EIP7503", GM_EIP7503, 0}, {"ConcatFixed4", GM_CONCAT_FIXED4, 4}, {"ProofOfWorkChecker", GM_PROOF_OF_WORK_CHECKER, 0},
    {"PublicCommitment", GM_PUBLIC_COMMITMENT, 1}, {"Poseidon", GM_POSEIDON, 1}, {"Divide", GM_DIVIDE, 1}, {"SubstringCheck", GM_SUBSTRING_CHECK, 2},
    {"ShiftLeft", GM_SHIFT_LEFT, 1}, {"ShiftRight", GM_SHIFT_RIGHT, 2}, {"Mask", GM_MASK, 1}, {"Concat", GM_CONCAT, 2}, {"Selector", GM_SELECTOR, 1},
    {"SelectorArray1D", GM_SELECTOR_ARRAY_1D, 2}, {"SelectorArray2D", GM_SELECTOR_ARRAY_2D, 3}, {"BigEndianBytes2Num", GM_BIG_ENDIAN_BYTES2NUM, 1},
    {"Bytes2Nibbles", GM_BYTES2NIBBLES, 1}, {"LittleEndianBytes2Num", GM_LITTLE_ENDIAN_BYTES2NUM, 1}, {"Num2BigEndianBytes", GM_NUM2BIG_ENDIAN_BYTES, 1},
    {"Num2LittleEndianBytes", GM_NUM2LITTLE_ENDIAN_BYTES, 1}, {"Nibbles2Bytes", GM_NIBBLES2BYTES, 1}, {"Num2BitsSafe", GM_NUM2BITS_SAFE, 1},
    {"Pad", GM_PAD, 2}, {"KeccakBytes", GM_KECCAK_BYTES, 1}, {"BurnAddress", GM_BURN_ADDRESS, 0}, {"BurnAddressHash", GM_BURN_ADDRESS_HASH, 0},
    {"AssertBits", GM_ASSERT_BITS, 1}, {"AssertByteString", GM_ASSERT_BYTE_STRING, 1}, {"AssertLessEqThan", GM_ASSERT_LESS_EQ_THAN, 1},
    {"AssertLessThan", GM_ASSERT_LESS_THAN, 1}, {"AssertGreaterEqThan", GM_ASSERT_GREATER_EQ_THAN, 1}, {"Filter", GM_FILTER, 1}, {"Fit", GM_FIT, 2},
    {"Reverse", GM_REVERSE, 1}, {"Flatten", GM_FLATTEN, 2}, {"Reshape", GM_RESHAPE, 2}, {"RlpInteger", GM_RLP_INTEGER, 1}, {"CountBytes", GM_COUNT_BYTES, 1},
    {"RlpEmptyAccount", GM_RLP_EMPTY_ACCOUNT, 1}, {"TruncatedAddressHash", GM_TRUNCATED_ADDRESS_HASH, 1}, {"IsInRange", GM_IS_IN_RANGE, 1},
    {"LeafDetector", GM_LEAF_DETECTOR, 1}, {"RlpMerklePatriciaTrieLeaf", GM_RLP_MPT_LEAF, 2},
};

// the packed inputs of the main, in declaration order per class
template <class P> struct GmIn {
    P& p; uint32_t nf, ns;
    HD GmIn(P& q) : p(q), nf(0), ns(0) {}
    HD F f() { return p.input_fr(nf++); }
    HD S s() { return p.input_sm(ns++); }
    HD SmRef sm(uint32_t n) { SmRef r = {GM_INPUT_W, ns}; ns += n; return r; }
};

// BurnAddress (burn_address.circom:47-58) in one piece: [addressBytes[20] | burnKey, revealAmount, burnExtraCommitment | hash, hashBytes[32]]
// || Poseidon(4), Num2BigEndianBytes(32), Fit(32, 20)   (the production circuit runs it as U_BAH_PRE with the Poseidon block spread over lanes)
template <class P> GD SmRef gBurnAddressMain(P& p, const F& prefix0, const F& bk, const F& ra, const F& bec) {
    SmRef o = p.sms(20); FrRef in = p.frs(3), h = p.frs(1); SmRef hb = p.sms(32);
    F pin[4]; pin[0] = prefix0; pin[1] = p.put(in, bk); pin[2] = p.put(in + 1, ra); pin[3] = p.put(in + 2, bec);
    const F hash = p.put(h, gPoseidon<P, 5>(p, pos_off(5), pin));
    F c;
    gNum2BigEndianBytesF(p, 32, hash, &hb, &c);
    SmRef fo = p.sms(20), fi = p.sms(32);                // Fit(32, 20)  [out[20] | in[32]]
    for (int i = 0; i < 32; i++) { const S by = canon_byte(c, 31 - i); p.put(fi + i, by); if (i < 20) { p.put(fo + i, by); p.put(o + i, by); } }
    return o;
}

// a[0] = template, a[1..4] = its parameters (validated by the planner, Plan::plan_gadget).  CountP: the input / output counts go to L.gm.
template <class P> GD void gm_run(P& p, const UnitDesc& d, CircuitLayout& L) {
    GmIn<P> in(p);
    const int p0 = (int)d.a[1], p1 = (int)d.a[2], p2 = (int)d.a[3], p3 = (int)d.a[4];
    uint32_t nout = 0;
    (void)p1; (void)p2; (void)p3;
    switch (d.a[0]) {
    case GM_EIP7503: {             // proof_of_work.circom:11-21  [out[8]] = "EIP-7503"
        SmRef o = p.sms(8);
        const uint64_t tag = 0x333035372D504945ULL;
        for (int i = 0; i < 8; i++) p.put(o + i, (S)((tag >> (8 * i)) & 0xff));
        nout = 8;
    } break;
    case GM_CONCAT_FIXED4: {       // proof_of_work.circom:28-48  [out[n] | a[], b[], c[], d[]]
        const int n = p0 + p1 + p2 + p3;
        SmRef src = in.sm(n), o = p.sms(n), own = p.sms(n);
        for (int i = 0; i < n; i++) p.put(o + i, p.put(own + i, p.get(src + i)));
        nout = n;
    } break;
    case GM_POSEIDON: {            // circomlib poseidon.circom  [out | inputs[n]]
        F x[4];
        for (int i = 0; i < 4; i++) x[i] = i < p0 ? in.f() : fr_zero();
        if (p0 == 2) gPoseidon<P, 3>(p, pos_off(3), x);
        else if (p0 == 3) gPoseidon<P, 4>(p, pos_off(4), x);
        else gPoseidon<P, 5>(p, pos_off(5), x);
        nout = 1;
    } break;
    case GM_DIVIDE: { S a = in.s(), b = in.s(), q, r; gDivide(p, p0, a, b, q, r); nout = 2; } break;
    case GM_SUBSTRING_CHECK: { SmRef a = in.sm(p0); S len = in.s(); SmRef b = in.sm(p1); gSubstringCheck(p, p0, p1, a, len, b); nout = 1; } break;
    case GM_SHIFT_LEFT: { SmRef a = in.sm(p0); gShiftLeft(p, p0, a, in.s()); nout = p0; } break;
    case GM_SHIFT_RIGHT: { SmRef a = in.sm(p0); gShiftRight(p, p0, p1, a, in.s()); nout = p0 + p1; } break;
    case GM_MASK: { SmRef a = in.sm(p0); gMask(p, p0, a, in.s()); nout = p0; } break;
    case GM_CONCAT: { SmRef a = in.sm(p0); S al = in.s(); SmRef b = in.sm(p1); S bl = in.s(), ol; gConcat(p, p0, p1, a, al, b, bl, ol); nout = p0 + p1 + 1; } break;
    case GM_SELECTOR: { SmRef v = in.sm(p0); gSelectorS(p, p0, v, in.s()); nout = 1; } break;
    case GM_SELECTOR_ARRAY_1D: { SmRef v = in.sm(p0 * p1); gSelectorArray1D(p, p0, p1, v, in.s()); nout = p1; } break;
    // SelectorArray2D(n, p, q) selector.circom:91-111 has the wire layout of SelectorArray1D(n, p*q): [out[p][q] | arrays[n][p][q], select | arraysT[p][q][n]] || Selector(n) x pq
    case GM_SELECTOR_ARRAY_2D: { SmRef v = in.sm(p0 * p1 * p2); gSelectorArray1D(p, p0, p1 * p2, v, in.s()); nout = p1 * p2; } break;
    case GM_BIG_ENDIAN_BYTES2NUM: { gBigEndianBytes2NumF(p, p0, in.sm(p0)); nout = 1; } break;
    case GM_LITTLE_ENDIAN_BYTES2NUM: { gLittleEndianBytes2NumF(p, p0, in.sm(p0)); nout = 1; } break;
    case GM_BYTES2NIBBLES: { gBytes2Nibbles(p, p0, in.sm(p0)); nout = 2 * p0; } break;
    case GM_NIBBLES2BYTES: { gNibbles2Bytes(p, p0, in.sm(2 * p0)); nout = p0; } break;
    case GM_NUM2BIG_ENDIAN_BYTES: { gNum2BigEndianBytesF(p, p0, in.f()); nout = p0; } break;
    case GM_NUM2LITTLE_ENDIAN_BYTES: { gNum2LittleEndianBytesF(p, p0, in.f()); nout = p0; } break;
    case GM_NUM2BITS_SAFE: { gNum2BitsSafeF(p, p0, in.f()); nout = p0; } break;
    case GM_PAD: { SmRef a = in.sm(p0 * p1); S nb; gPad(p, p0, p1, a, in.s(), nb); nout = p0 * p1 + 1; } break;
    case GM_BURN_ADDRESS: { F bk = in.f(), ra = in.f(), bec = in.f(); gBurnAddressMain(p, L.prefix[0], bk, ra, bec); nout = 20; } break;
    case GM_ASSERT_BITS: { gAssertBitsF(p, p0, in.f()); } break;
    case GM_ASSERT_BYTE_STRING: { gAssertByteString(p, p0, in.sm(p0)); } break;
    case GM_ASSERT_LESS_EQ_THAN: { S a = in.s(), b = in.s(); gAssertLessEqThanS(p, p0, a, b); } break;
    case GM_ASSERT_LESS_THAN: { S a = in.s(), b = in.s(); gAssertLessThanS(p, p0, a, b); } break;
    case GM_ASSERT_GREATER_EQ_THAN: { S a = in.s(), b = in.s(); gAssertGreaterEqThanS(p, p0, a, b); } break;
    case GM_FILTER: { gFilter(p, p0, in.s()); nout = p0; } break;
    case GM_FIT: { gFitS(p, p0, p1, in.sm(p0)); nout = p1; } break;
    case GM_REVERSE: { gReverseS(p, p0, in.sm(p0)); nout = p0; } break;
    case GM_FLATTEN: case GM_RESHAPE: { gFlattenS(p, p0 * p1, in.sm(p0 * p1)); nout = p0 * p1; } break;     // array.circom:64-87: the identity on row-major data
    case GM_RLP_INTEGER: { S ol; gRlpInteger(p, p0, in.f(), ol); nout = p0 + 2; } break;
    case GM_COUNT_BYTES: { gCountBytes(p, p0, in.sm(p0)); nout = 1; } break;
    case GM_RLP_EMPTY_ACCOUNT: { S ol; gRlpEmptyAccount(p, p0, in.f(), ol); nout = 71 + p0; } break;
    case GM_TRUNCATED_ADDRESS_HASH: { SmRef a = in.sm(2 * p0); S ol; gTruncatedAddressHash(p, p0, a, in.s(), ol); nout = p0 + 2; } break;
    case GM_IS_IN_RANGE: { S lo = in.s(), v = in.s(), hi = in.s(); gIsInRange(p, p0, lo, v, hi); nout = 1; } break;
    case GM_LEAF_DETECTOR: { SmRef a = in.sm(p0); gLeafDetector(p, p0, a, in.s()); nout = 1; } break;
    case GM_RLP_MPT_LEAF: {        // [out[maxOut], outLen | addressHashNibbles[2ab], addressHashNibblesLen, balance]
        SmRef a = in.sm(2 * p0); S nl = in.s(), ol;
        gRlpMptLeaf(p, p0, p1, a, nl, in.f(), ol);
        nout = (2 + 1 + 1 + p0) + (2 + 4 + p1 + 66) + 1;
    } break;
    default: break;
    }
    if (P::is_count) { L.gm.nfr_in = in.nf; L.gm.nsm_in = in.ns; L.gm.nout = nout; }
}
// the packed inputs of a main that is planned from split units (the Keccak mains): FR inputs 0.. -> FR wires from (a0, a1) on (a2 of them),
// SM inputs 0.. -> SM wires from (a3, a4) on (a5 of them): the template's own input signals, contiguous in declaration order
template <class P> GD void gm_input(P& p, const UnitDesc& d) {
    for (uint32_t k = 0; k < d.a[2]; k++) p.put(FrRef{d.a[0] + k, d.a[1] + k}, p.input_fr(k));
    for (uint32_t k = 0; k < d.a[5]; k++) p.put(SmRef{d.a[3] + k, d.a[4] + k}, p.input_sm(k));
}
