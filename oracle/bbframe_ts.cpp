// ORACLE (test infrastructure, CPU only) -- not part of the shipped engine.
// Restatement of the BBFRAME -> TS / GSE parser that consumes the demodulator's output in the reference's sink handler
// (main.cpp:532-558):
//   bbts_crc8_bits           dvbs2/bbframe_ts_parser.cpp:70-83    BBHEADER check over 80 bits (reflected 0xD5 -> 0xAB, LSB-first register)
//   BbTsParser::set_frame_size   .cpp:31-42
//   BbTsParser::work         .cpp:104-390   header checks :119-152, resynchronisation :159-170, MPEG-TS packetisation :176-207,
//                                            GSE -> GRE decapsulation with three reassembly slots :211-383, CRC-32 :85-102
// PARITY UNPINNED: bbframe_ts_parser.h includes <dsp/stream.h> of SDR++ core, which is not vendored with the reference
// and absent from this image, so the reference's own translation unit cannot be compiled here (no stand-in headers are
// written); the reference holds no test vectors for it either.  The restatement is anchored on the reference's code as
// read, and on round trips through this repo's transmitter (TS packets -> BBFRAMEs per EN 302 307-1 5.1.4-5.1.6 -> parser).
//
// Where the reference's behaviour is undefined (reads beyond the caller's input buffer, writes beyond the output buffer
// or the 64 KiB reassembly buffers, a negative length passed to memcpy) this restatement -- and the engine -- do:
//   * a GSE packet whose header or data would extend beyond the END OF THE INPUT BUFFER (all `cnt` frames) ends the
//     parsing of that BBFRAME (reads that stay inside the buffer but run into the next BBFRAME are reproduced as is);
//   * a PDU that does not fit into the remaining output space, or whose reassembled length is negative, is dropped;
//   * a fragment that would overflow a reassembly buffer frees the slot and is dropped;
//   * the TS loop running out of output space with more than 187 bytes of data field left returns -1.
#include "bbframe_ts.h"

#include <cstring>

namespace orc {

unsigned bbts_crc8_bits(const uint8_t* in, int nbits) {
    unsigned crc = 0;
    for (int n = 0; n < nbits; ++n) {
        unsigned bit = (in[n >> 3] >> (7 - (n & 7))) & 1u;
        unsigned fb = bit ^ (crc & 1u);
        crc >>= 1;
        if (fb) crc ^= 0xAB;
    }
    return crc;
}

BbTsParser::BbTsParser() {
    memset(partial, 0, sizeof(partial));
    // crc32_init (.cpp:85-95): MPEG-2 CRC-32, polynomial 0x04C11DB7, MSB first
    for (unsigned i = 0; i < 256; ++i) {
        uint32_t k = i << 24;
        for (int b = 0; b < 8; ++b) k = (k & 0x80000000u) ? (k << 1) ^ 0x04c11db7u : (k << 1);
        crc32_tab[i] = k;
    }
}

uint32_t BbTsParser::crc32(const uint8_t* p, int n, uint32_t c) const {
    for (int i = 0; i < n; ++i) c = (c << 8) ^ crc32_tab[((c >> 24) ^ p[i]) & 0xff];
    return c;
}

void BbTsParser::set_frame_size(int kbch_bits) {
    kbch = kbch_bits;
    max_dfl = kbch - 80;
    count = 0;
    synched = 0;          // index, spanning: never read by the reference
}

static BbHeader parse_header(const uint8_t* b) {
    BbHeader h;
    h.ts_gs = b[0] >> 6;
    h.sis_mis = (b[0] >> 5) & 1;
    h.ccm_acm = (b[0] >> 4) & 1;
    h.issyi = (b[0] >> 3) & 1;
    h.npd = (b[0] >> 2) & 1;
    h.ro = b[0] & 3;
    h.isi = h.sis_mis == 0 ? b[1] : 0;
    h.upl = b[2] << 8 | b[3];
    h.dfl = b[4] << 8 | b[5];
    h.sync = b[6];
    h.syncd = b[7] << 8 | b[8];
    return h;
}

int BbTsParser::work(const uint8_t* bb, int cnt, uint8_t* out, int cap) {
    const long fbytes = kbch / 8;
    const long total = fbytes * cnt;
    int out_p = 0, bbproc = 0;

    auto put_gre = [&](uint16_t proto, const uint8_t* data, int len) {   // .cpp:254-263 / :338-349
        int hdr = 2 + ((proto == 0x0800 || proto == 0x86DD) ? 2 : 0);
        if (len < 0 || (long)out_p + hdr + len > cap) return;            // dropped (see header)
        out[out_p++] = 0;
        out[out_p++] = 0;
        if (hdr == 4) { out[out_p++] = proto >> 8; out[out_p++] = proto & 0xff; }
        memcpy(out + out_p, data, len);
        out_p += len;
    };

    for (int f = 0; f < cnt; ++f) {
        const long base = fbytes * f;
        if (bbts_crc8_bits(bb + base, 80) != 0) { synched = 0; continue; }
        const BbHeader hd = parse_header(bb + base);
        if ((unsigned)hd.dfl > (unsigned)max_dfl || hd.syncd >= hd.dfl - 8) { synched = 0; continue; }
        if (hd.dfl % 8 != 0) { synched = 0; continue; }
        unsigned df = hd.dfl / 8;
        long pos = base + 10;
        if (!synched) {
            pos += hd.syncd / 8 + 1;
            df -= hd.syncd / 8 + 1;
            count = 0;
            synched = 1;
        }
        last_header = hd;
        ++bbproc;
        if (hd.ts_gs == 3) {
            while (df >= 188 && cap - out_p > 188) {
                const uint8_t* cur;
                if (count > 0) {
                    int rem = 188 - count;
                    memcpy(partial + count, bb + pos, rem);
                    pos += rem;
                    df -= rem;
                    cur = partial;
                    count = 0;
                } else {
                    cur = bb + pos;
                    pos += 188;
                    df -= 188;
                }
                out[out_p] = 0x47;
                memcpy(out + out_p + 1, cur, 187);
                out_p += 188;
            }
            if (df >= 188) { synched = 0; return -1; }
            if (df > 0) {
                count = (int)df;
                memcpy(partial, bb + pos, df);
            }
            if (cap - out_p <= 188) break;
        } else if (hd.ts_gs == 1) {
            const int dfl8 = hd.dfl / 8;
            int cur = 0;
            while (cur < dfl8) {
                if (hd.issyi || hd.npd || hd.upl != 0) { cur = dfl8; break; }
                const long p = pos + cur;
                if (p + 2 > total) break;
                const uint8_t h1 = bb[p], h2 = bb[p + 1];
                const int S = h1 >> 7, E = (h1 >> 6) & 1;
                const int lt = (h1 & 0x30) >> 2;            // 0, 4, 8 or 12: only label type 00 is ever told apart (.cpp:216)
                if (!S && !E && lt == 0) break;
                uint16_t len = (uint16_t)(((h1 & 0x0f) << 8) | h2);
                if (S && E) {
                    int start = 4;
                    len -= 2;
                    if (lt == 0) { start += 6; len -= 6; }
                    if (p + start + len > total) break;
                    uint16_t proto = (uint16_t)(bb[p + 2] << 8 | bb[p + 3]);
                    put_gre(proto, bb + p + start, len);
                    cur += start + len;
                } else if (S) {
                    int start = 7;
                    len -= 5;
                    if (lt == 0) { start += 6; len -= 6; }
                    if (p + start + len > total) break;
                    const int fragid = bb[p + 2];
                    for (auto& s : slot) {
                        if (!s.active || s.id == fragid) {
                            s.active = true;
                            s.id = fragid;
                            s.proto = (uint16_t)(bb[p + 5] << 8 | bb[p + 6]);
                            s.buf.assign(bb + p + start, bb + p + start + len);
                            s.ctr = len;
                            s.crc = 0xffffffffu;
                            s.crc = crc32(bb + p + 3, 2, s.crc);        // total length
                            s.crc = crc32(bb + p + 5, 2, s.crc);        // protocol type
                            if (lt == 0) s.crc = crc32(bb + p + 7, 6, s.crc);
                            s.crc = crc32(bb + p + start, len, s.crc);
                            break;
                        }
                    }
                    cur += start + len;
                } else {
                    const int start = 3;
                    len -= 1;
                    if (p + start + len > total) break;
                    const int fragid = bb[p + 2];
                    for (auto& s : slot) {
                        if (s.active && s.id == fragid) {
                            if ((long)s.ctr + len > 65536) { s.active = false; break; }
                            s.buf.resize(s.ctr);
                            s.buf.insert(s.buf.end(), bb + p + start, bb + p + start + len);
                            if (E) {
                                s.active = false;
                                s.ctr += (int)len - 4;
                                s.crc = crc32(bb + p + start, (int)len - 4, s.crc);
                                uint32_t rx = 0;
                                for (int cb = 1; cb <= 4; ++cb) rx |= (uint32_t)bb[p + start + len - cb] << (8 * (cb - 1));
                                if (s.crc != rx) {
                                    last_gse_crc_err = 1;
                                } else {
                                    last_gse_crc_err = 0;
                                    put_gre(s.proto, s.buf.data(), s.ctr);
                                }
                            } else {
                                s.ctr += len;
                                s.crc = crc32(bb + p + start, len, s.crc);
                            }
                            break;
                        }
                    }
                    cur += start + len;
                }
            }
        }
    }
    last_bb_cnt = cnt;
    last_bb_proc = bbproc;
    last_ts_errs = 0;
    return out_p;
}

}  // namespace orc
