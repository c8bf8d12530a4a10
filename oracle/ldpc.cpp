// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// Scalar restatement of the reference's layered offset-min-sum LDPC decoder.
// Pinned bit-exact against the compiled reference (oracle/_ref, see oracle/Makefile;
// tests/test_oracle_golden.py::test_ldpc_oracle_equals_reference_on_random_frames) and against the fixtures in tests/golden/.
//
// Follows, function by function:
//   LDPCDecoder::init        xdsopl-ldpc-pabr/layered_decoder.hh:79-120   (pos/cnc build + layer-major permutation)
//   LDPCDecoder::operator()  layered_decoder.hh:121-133                    (parity permutation, bad/update loop)
//   LDPCDecoder::bad         layered_decoder.hh:28-45
//   LDPCDecoder::update      layered_decoder.hh:46-74
//   OffsetMinSumAlgorithm<SIMD<int8_t,W>,NormalUpdate,2>  algorithms.hh:206-277
//   int8 lane semantics      sse4_1.hh (adds/subs saturating, vqabs, vsign)
//   BBFrameLDPC::decode      dvbs2/codings/bbframe_ldpc.cpp:123-139        (return value mapping)
//   LDPC<TABLE> iterator     ldpc.hh:42-108                                (address expansion)
#include "oracle.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace orc {

static inline int8_t sat8(int v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
static inline int8_t ssub(int8_t a, int8_t b) { return sat8((int)a - (int)b); }   // _mm_subs_epi8
static inline int8_t sadd(int8_t a, int8_t b) { return sat8((int)a + (int)b); }   // _mm_adds_epi8
static inline int8_t qabs(int8_t a) { int v = a < -127 ? -127 : a; return (int8_t)(v < 0 ? -v : v); }  // vqabs
static inline int8_t vsign(int8_t a, int8_t b) { return b < 0 ? (int8_t)-a : (b == 0 ? 0 : a); }      // _mm_sign_epi8

LdpcCode::LdpcCode(int code_index) {
    const QcCodeDesc& d = QC_CODES[code_index];
    N = d.N; K = d.K; M = 360; R = N - K; q = d.q; CNL = d.max_deg; LT = d.edges;
    // Expand exactly like LDPCDecoder::init: walk information bits in ascending order and append
    // each to the list of every check it touches (layered_decoder.hh:99-108).  A bit 360*r+m touches
    // check (x + q*m) mod R for every address x of table row r (ldpc.hh:88-95); with x = q*s+i this is
    // check q*((s+m) mod 360) + i.
    std::vector<uint16_t> pos0((size_t)R * CNL, 0);
    std::vector<uint8_t> cnc0(R, 0);
    // invert the layer-major description into per-table-row address lists
    int groups = K / M;
    std::vector<std::vector<int>> rows(groups);
    for (int i = 0; i < q; ++i)
        for (int e = d.off[i]; e < d.off[i + 1]; ++e) {
            int r = d.ent[e] >> 16, s = d.ent[e] & 0xffff;
            rows[r].push_back(q * s + i);
        }
    for (int r = 0; r < groups; ++r)
        for (int m = 0; m < M; ++m) {
            int bit = M * r + m;
            for (int x : rows[r]) {
                int c = (x + q * m) % R;
                pos0[(size_t)CNL * c + cnc0[c]++] = (uint16_t)bit;
            }
        }
    cnc.assign(cnc0.begin(), cnc0.end());
    // layer-major permutation of the rows (layered_decoder.hh:113-119); cnc is NOT permuted
    pos.assign((size_t)R * CNL, 0);
    for (int i = 0; i < q; ++i)
        for (int j = 0; j < M; ++j)
            for (int c = 0; c < CNL; ++c)
                pos[(size_t)CNL * (M * i + j) + c] = pos0[(size_t)CNL * (q * j + i) + c];
}

static bool ldpc_bad(const LdpcCode& C, const int8_t* data, const int8_t* pty) {
    const int M = C.M, q = C.q, CNL = C.CNL;
    for (int i = 0; i < q; ++i) {
        int cnt = C.cnc[i];
        for (int j = 0; j < M; ++j) {
            int8_t cnv = vsign(1, pty[M * i + j]);
            if (i) cnv = vsign(cnv, pty[M * (i - 1) + j]);
            else if (j) cnv = vsign(cnv, pty[j + (q - 1) * M - 1]);
            for (int c = 0; c < cnt; ++c) cnv = vsign(cnv, data[C.pos[(size_t)CNL * (M * i + j) + c]]);
            if (!(cnv > 0)) return true;
        }
    }
    return false;
}

// algorithms.hh:233-256
static void finalp(int8_t* links, int cnt) {
    int8_t mags[64];
    for (int i = 0; i < cnt; ++i) {
        int m = (int)(uint8_t)qabs(links[i]) - 1;  // vqsub on unsigned with beta = 1
        mags[i] = (int8_t)(m < 0 ? 0 : m);
    }
    int8_t mins[2];
    mins[0] = std::min(mags[0], mags[1]);
    mins[1] = std::max(mags[0], mags[1]);
    for (int i = 2; i < cnt; ++i) {
        mins[1] = std::min(mins[1], std::max(mins[0], mags[i]));
        mins[0] = std::min(mins[0], mags[i]);
    }
    int8_t signs = links[0];
    for (int i = 1; i < cnt; ++i) signs = (int8_t)(signs ^ links[i]);
    for (int i = 0; i < cnt; ++i) {
        int8_t other = (mags[i] == mins[0]) ? mins[1] : mins[0];
        int8_t sg = (int8_t)((signs ^ links[i]) | 127);
        links[i] = vsign(other, sg);
    }
}

static void ldpc_update(const LdpcCode& C, int8_t* data, int8_t* pty, int8_t* bnl) {
    const int M = C.M, q = C.q, CNL = C.CNL;
    int8_t* bl = bnl;
    int8_t inp[64], out[64];
    for (int i = 0; i < q; ++i) {
        int cnt = C.cnc[i];
        for (int j = 0; j < M; ++j) {
            int deg = cnt + 2 - !(i | j);
            const uint16_t* p = &C.pos[(size_t)CNL * (M * i + j)];
            for (int c = 0; c < cnt; ++c) inp[c] = out[c] = ssub(data[p[c]], bl[c]);
            inp[cnt] = out[cnt] = ssub(pty[M * i + j], bl[cnt]);
            if (i) inp[cnt + 1] = out[cnt + 1] = ssub(pty[M * (i - 1) + j], bl[cnt + 1]);
            else if (j) inp[cnt + 1] = out[cnt + 1] = ssub(pty[j + (q - 1) * M - 1], bl[cnt + 1]);
            finalp(out, deg);
            for (int d = 0; d < deg; ++d) {  // OffsetMinSum::update -> NormalUpdate: clamp to [-32,31], overwrite
                int v = out[d];
                bl[d] = (int8_t)(v < -32 ? -32 : (v > 31 ? 31 : v));
            }
            for (int c = 0; c < cnt; ++c) data[p[c]] = sadd(inp[c], bl[c]);
            pty[M * i + j] = sadd(inp[cnt], bl[cnt]);
            if (i) pty[M * (i - 1) + j] = sadd(inp[cnt + 1], bl[cnt + 1]);
            else if (j) pty[j + (q - 1) * M - 1] = sadd(inp[cnt + 1], bl[cnt + 1]);
            bl += deg;
        }
    }
}

// frame: N int8 LLRs (negative = bit 1), decoded in place.  Returns the number of update sweeps that
// were needed (0..max_trials) or -1 (bbframe_ldpc.cpp:135-138).
// force != 0 is this repo's benchmark switch (SURVEY 8d / hard part 7): no early exit, exactly
// max_trials sweeps; returns max_trials if the result satisfies all checks, else -1.
int ldpc_decode(const LdpcCode& C, int8_t* frame, int max_trials, int force) {
    const int M = C.M, q = C.q, K = C.K, R = C.R;
    std::vector<int8_t> bnl(C.LT, 0), pty(R);
    int8_t* data = frame;
    int8_t* parity = frame + K;
    for (int i = 0; i < q; ++i)
        for (int j = 0; j < M; ++j) pty[M * i + j] = parity[q * j + i];
    int trials = max_trials;
    int ret;
    if (!force) {
        while (ldpc_bad(C, data, pty.data()) && --trials >= 0) ldpc_update(C, data, pty.data(), bnl.data());
        ret = trials < 0 ? trials : max_trials - trials;
    } else {
        for (int t = 0; t < max_trials; ++t) ldpc_update(C, data, pty.data(), bnl.data());
        ret = ldpc_bad(C, data, pty.data()) ? -1 : max_trials;
    }
    for (int i = 0; i < q; ++i)
        for (int j = 0; j < M; ++j) parity[q * j + i] = pty[M * i + j];
    return ret;
}

// Systematic IRA encoder per ETSI EN 302 307-1 5.3.2 (own code; the reference's BBFrameLDPC::encode
// produces non-codewords for some rates).  bits: N bytes of 0/1, [0,K) filled on entry.
void ldpc_encode(const LdpcCode& C, uint8_t* bits) {
    const int M = C.M, q = C.q, K = C.K, R = C.R, CNL = C.CNL;
    uint8_t* p = bits + K;
    for (int i = 0; i < q; ++i) {
        int cnt = C.cnc[i];
        for (int j = 0; j < M; ++j) {
            uint8_t acc = 0;
            for (int c = 0; c < cnt; ++c) acc ^= bits[C.pos[(size_t)CNL * (M * i + j) + c]];
            p[q * j + i] = acc;
        }
    }
    for (int c = 1; c < R; ++c) p[c] ^= p[c - 1];
}

}  // namespace orc
