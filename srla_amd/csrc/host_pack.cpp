/*
 * host_pack.cpp -- what is left of the bitstream on the host: the 30-byte stream header and the static
 * Huffman tables handed to the device.  Blocks (header, compress / raw payload, residual codes, Fletcher-16)
 * are assembled by srla_pack_blocks in kernels.hip and land in the output buffer complete.
 *
 *   stream header   libs/srla_encoder/src/srla_encoder.c:134-161
 */
#include "host_pack.h"

#include "huffman_codes.inc"

namespace srla {

namespace {
inline void put_u16be(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
inline void put_u32be(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}
}  // namespace

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

const unsigned char *huffman_plain_lengths() { return srla_huff_plain_len; }
const unsigned char *huffman_summed_lengths() { return srla_huff_summed_len; }
const unsigned int *huffman_plain_codes() { return srla_huff_plain_code; }
const unsigned int *huffman_summed_codes() { return srla_huff_summed_code; }

}  // namespace srla
