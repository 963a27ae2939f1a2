/*
 * host_pack.cpp -- the host side of the bitstream: stream header, block framing, Fletcher-16 and
 * the bit-serial Rice / recursive-Rice / static-Huffman pack (this part of the codec stays on
 * the host by design: header fields, Huffman-coded taps, concatenation, framing, checksum.  The GPU
 * supplies every parameter and the already Rice-coded residual bitstring of each channel,
 * so the host ORs words together instead of coding sample by sample).
 *
 * Bitstream facts restated from the reference (paths relative to the reference tree):
 *   stream header   libs/srla_encoder/src/srla_encoder.c:134-161
 *   block header    srla_encoder.c:1583-1595, 1629-1636
 *   compress data   srla_encoder.c:1368-1452
 *   raw data        srla_encoder.c:823-852
 *   residual codes  libs/srla_coder/src/srla_coder.c:165-190, 532-595
 *   bit order       libs/bit_stream/include/bit_stream.h:245-307, 400-437 (MSB first, zero pad)
 *   checksum        libs/srla_internal/src/srla_utility.c:36-60
 */
#include "host_pack.h"

#include <string.h>

#include "huffman_codes.inc"

namespace srla {

namespace {

inline uint32_t zigzag(int32_t s) { return ((uint32_t)s << 1) ^ (uint32_t)(-(int32_t)(s < 0)); }

struct BitSink {
    uint8_t *p;
    uint64_t acc;
    uint32_t cnt; /* pending bits in acc, always < 32 between calls */

    explicit BitSink(uint8_t *dst) : p(dst), acc(0), cnt(0) {}
    inline void put(uint32_t val, uint32_t n)
    {
        /* n <= 32; val may carry garbage above bit n (the reference masks, bit_stream.h:262) */
        if (n == 0) return;
        const uint64_t v = (n == 32) ? (uint64_t)val : ((uint64_t)val & ((1ull << n) - 1ull));
        acc = (acc << n) | v;
        cnt += n;
        if (cnt >= 32) {
            cnt -= 32;
            const uint32_t w = (uint32_t)(acc >> cnt);
            p[0] = (uint8_t)(w >> 24); p[1] = (uint8_t)(w >> 16); p[2] = (uint8_t)(w >> 8); p[3] = (uint8_t)w;
            p += 4;
        }
    }
    inline void zeros_then_one(uint32_t run)
    {
        while (run >= 31) { put(0, 31); run -= 31; }
        put(1, run + 1);
    }
    inline uint8_t *finish()
    {
        while (cnt >= 8) { cnt -= 8; *p++ = (uint8_t)(acc >> cnt); }
        if (cnt > 0) { *p++ = (uint8_t)((acc << (8 - cnt)) & 0xFF); cnt = 0; }
        return p;
    }
};

inline void put_u16be(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
inline void put_u32be(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

inline uint64_t load_be64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v); }
inline void store_be64(uint8_t *p, uint64_t v) { v = __builtin_bswap64(v); memcpy(p, &v, 8); }

/* Append `nbits` bits of a byte-aligned MSB-first bitstring (the device pads every bitstring to a multiple
 * of 8 bytes, so whole 64-bit words can be read).  The sink's pending bits are merged once, then the body
 * is a shifted 64-bit copy; the tail goes back through put(). */
inline void append_bits(BitSink &w, const uint8_t *src, uint32_t nbits)
{
    uint32_t done = 0;
    if (nbits >= 128) {
        const uint32_t c = w.cnt;                       /* pending bits (< 32) in the low end of acc */
        const uint64_t pend = c ? (w.acc & ((1ull << c) - 1ull)) : 0ull;
        uint64_t carry = c ? (pend << (64 - c)) : 0ull; /* pending bits, left aligned */
        uint8_t *p = w.p;
        const uint32_t words = nbits >> 6;
        if (c == 0) {
            memcpy(p, src, (size_t)words * 8);
            p += (size_t)words * 8;
        } else {
            for (uint32_t i = 0; i < words; i++) {
                const uint64_t v = load_be64(src + (size_t)i * 8);
                store_be64(p, carry | (v >> c));
                carry = v << (64 - c);
                p += 8;
            }
        }
        w.p = p;
        w.acc = c ? (carry >> (64 - c)) : 0ull;          /* the last c bits read are pending again */
        done = words << 6;
        src += (size_t)words * 8;
    }
    uint32_t rem = nbits - done;
    while (rem >= 32) {
        w.put(((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | (uint32_t)src[3], 32);
        src += 4; rem -= 32;
    }
    while (rem >= 8) { w.put(*src++, 8); rem -= 8; }
    if (rem) w.put((uint32_t)(*src) >> (8 - rem), rem);
}

}  // namespace

uint16_t fletcher16(const uint8_t *data, size_t size)
{
    /* The reference (srla_utility.c:36-60) runs c0 += b, c1 += c0 and folds both mod 255 every 5802 bytes with
     * (x + x/255) & 0xFF, which IS x mod 255 (255q + r + q = 256q + r).  Hence c0 = sum(b_i) mod 255 and
     * c1 = sum((size - i) * b_i) mod 255, which needs no running dependency: two plain sums the compiler can
     * vectorise, accumulated per 256-byte chunk in 32 bits and widened to 64. */
    uint64_t a = 0, wsum = 0;            /* a = sum b_i ; wsum = sum i * b_i (i = global index) */
    size_t base = 0;
    while (base < size) {
        const size_t len = (size - base < 256) ? (size - base) : 256;
        uint32_t ca = 0, cw = 0;
        const uint8_t *p = data + base;
        for (size_t i = 0; i < len; i++) { ca += p[i]; cw += (uint32_t)i * p[i]; }
        a += ca;
        wsum += cw + (uint64_t)base * ca;
        base += len;
    }
    const uint64_t c0 = a % 255u;
    const uint64_t c1 = ((uint64_t)(size % 255u) * c0 + 255u * 255u - (wsum % 255u)) % 255u;
    return (uint16_t)((c1 << 8) | c0);
}

void write_stream_header(const StreamInfo &s, uint8_t *p)
{
    p[0] = '1'; p[1] = '2'; p[2] = '4'; p[3] = '9';
    put_u32be(p + 4, 10);   /* SRLA_FORMAT_VERSION */
    put_u32be(p + 8, 18);   /* SRLA_CODEC_VERSION  */
    put_u16be(p + 12, s.num_channels);
    put_u32be(p + 14, s.num_samples);
    put_u32be(p + 18, s.sampling_rate);
    put_u16be(p + 22, s.bits_per_sample);
    p[24] = (uint8_t)s.offset_lshift;
    put_u32be(p + 25, s.max_block);
    p[29] = (uint8_t)s.preset;
}

uint32_t pack_block(const StreamInfo &s, const SrlaBlockRecord &br, const SrlaChanRecord *chan,
                    const uint8_t *region, uint8_t *out)
{
    const uint32_t nch = s.num_channels, bps = s.bits_per_sample, n = br.n;
    uint8_t *payload = out + 11;
    uint32_t payload_bytes = 0;
    if (br.block_type == SRLA_BLOCK_COMPRESS) {
        BitSink w(payload);
        w.put(br.ch_method, 2);
        for (uint32_t ch = 0; ch < nch; ch++) {
            w.put(zigzag(chan[ch].preemph_prev), bps + 1);
            w.put(zigzag(chan[ch].preemph_coef), 5);
        }
        for (uint32_t ch = 0; ch < nch; ch++) {
            const SrlaChanRecord &c = chan[ch];
            w.put(c.lpc_order, 8);
            w.put(c.lpc_rshift, 4);
            w.put(c.use_sum, 1);
            if (!c.use_sum) {
                for (uint32_t i = 0; i < c.lpc_order; i++) {
                    const uint32_t u = zigzag(c.lpc_coef[i]);
                    w.put(srla_huff_plain_code[u], srla_huff_plain_len[u]);
                }
            } else {
                uint32_t u = zigzag(c.lpc_coef[0]);
                w.put(srla_huff_plain_code[u], srla_huff_plain_len[u]);
                for (uint32_t i = 1; i < c.lpc_order; i++) {
                    u = zigzag((int32_t)c.lpc_coef[i] + (int32_t)c.lpc_coef[i - 1]);
                    w.put(srla_huff_summed_code[u], srla_huff_summed_len[u]);
                }
            }
        }
        for (uint32_t ch = 0; ch < nch; ch++) {
            const SrlaChanRecord &c = chan[ch];
            w.put(c.ltp_period != 0, 1);
            if (c.ltp_period > 0) {
                w.put((s.ltp_order - 1) / 2, 1);
                w.put(c.ltp_period - SRLA_LTP_MIN_PERIOD, 8);
                for (uint32_t i = 0; i < s.ltp_order; i++) w.put(zigzag(c.ltp_coef[i]), 6);
            }
        }
        /* residuals: the device coded them (srla_coder.c:532-595); OR the words in behind the header */
        const uint8_t *src = region;
        for (uint32_t ch = 0; ch < nch; ch++) {
            append_bits(w, src, chan[ch].res_bits);
            src += ((chan[ch].res_bits + 63u) >> 6) << 3;
        }
        payload_bytes = (uint32_t)(w.finish() - payload);
    } else if (br.block_type == SRLA_BLOCK_RAW) {
        payload_bytes = (bps / 8) * n * nch;
        memcpy(payload, region, payload_bytes);
    }
    put_u16be(out, 0xFFFF);
    put_u32be(out + 2, payload_bytes + 5);
    out[8] = (uint8_t)br.block_type;
    put_u16be(out + 9, n);
    put_u16be(out + 6, fletcher16(out + 8, payload_bytes + 3));
    return 11 + payload_bytes;
}

const unsigned char *huffman_plain_lengths() { return srla_huff_plain_len; }
const unsigned char *huffman_summed_lengths() { return srla_huff_summed_len; }

}  // namespace srla
