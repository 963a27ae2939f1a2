/*
 * host_pack.cpp -- the host side of the bitstream: stream header, block framing, Fletcher-16 and
 * the bit-serial Rice / recursive-Rice / static-Huffman pack (this part of the codec stays on
 * the host by design; the GPU supplies residuals and every parameter, so nothing is searched
 * again here).
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

void pack_residual(BitSink &w, const SrlaItemResult &rec, const int32_t *res, uint32_t n)
{
    w.put(rec.res_code_type, 2);
    if (rec.res_code_type == SRLA_CODE_ALLZERO) return;
    const uint32_t porder = rec.res_porder, len = n >> porder;
    w.put(porder, 10);
    uint32_t prev = 0;
    for (uint32_t part = 0; part < (1u << porder); part++) {
        const uint32_t k = rec.kparam[part];
        if (part == 0) w.put(k, 5);
        else w.zeros_then_one(zigzag((int32_t)k - (int32_t)prev));
        prev = k;
        const int32_t *r = res + (size_t)part * len;
        if (rec.res_code_type == SRLA_CODE_RICE) {
            for (uint32_t i = 0; i < len; i++) {
                const uint32_t u = zigzag(r[i]);
                w.zeros_then_one(u >> k);
                w.put(u, k);
            }
        } else {
            const uint32_t k1 = k + 1, k1pow = 1u << k1;
            for (uint32_t i = 0; i < len; i++) {
                const uint32_t u = zigzag(r[i]);
                if (u < k1pow) {
                    w.put(k1pow | u, k1 + 1);
                } else {
                    const uint32_t v = u - k1pow;
                    const uint32_t q = 1 + (v >> k);
                    if (q + 1 + k <= 32) {
                        /* q zeros, the terminating one and the k low bits in one store */
                        w.put((1u << k) | (v & ((1u << k) - 1u)), q + 1 + k);
                    } else {
                        w.zeros_then_one(q);
                        w.put(v, k);
                    }
                }
            }
        }
    }
}

}  // namespace

uint16_t fletcher16(const uint8_t *data, size_t size)
{
    uint32_t c0 = 0, c1 = 0;
    while (size > 0) {
        size_t chunk = (size < 5802) ? size : 5802;
        size -= chunk;
        while (chunk--) { c0 += *data++; c1 += c0; }
        c0 = (c0 + c0 / 255u) & 0xFFu;
        c1 = (c1 + c1 / 255u) & 0xFFu;
    }
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

uint32_t pack_block(const StreamInfo &s, const SrlaBlockRecord &br, const SrlaItemResult *chan,
                    const int32_t *const *data, uint8_t *out)
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
            const SrlaItemResult &c = chan[ch];
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
            const SrlaItemResult &c = chan[ch];
            w.put(c.ltp_period != 0, 1);
            if (c.ltp_period > 0) {
                w.put((s.ltp_order - 1) / 2, 1);
                w.put(c.ltp_period - SRLA_LTP_MIN_PERIOD, 8);
                for (uint32_t i = 0; i < s.ltp_order; i++) w.put(zigzag(c.ltp_coef[i]), 6);
            }
        }
        for (uint32_t ch = 0; ch < nch; ch++) pack_residual(w, chan[ch], data[ch], n);
        payload_bytes = (uint32_t)(w.finish() - payload);
    } else if (br.block_type == SRLA_BLOCK_RAW) {
        const uint32_t bytes = bps / 8;
        uint8_t *q = payload;
        for (uint32_t i = 0; i < n; i++)
            for (uint32_t ch = 0; ch < nch; ch++) {
                const uint32_t u = zigzag(data[ch][i]);
                for (uint32_t b = 0; b < bytes; b++) *q++ = (uint8_t)(u >> (8 * (bytes - 1 - b)));
            }
        payload_bytes = (uint32_t)(q - payload);
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
