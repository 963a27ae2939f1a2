/* host_pack.h -- host-side bitstream writer (see host_pack.cpp). */
#ifndef SRLA_HOST_PACK_H
#define SRLA_HOST_PACK_H

#include <stddef.h>
#include <stdint.h>

#include "device_layout.h"

namespace srla {

struct StreamInfo {
    uint32_t num_channels;
    uint32_t bits_per_sample;
    uint32_t sampling_rate;
    uint32_t num_samples;
    uint32_t offset_lshift;
    uint32_t max_block;
    uint32_t preset;
    uint32_t ltp_order;
};

uint16_t fletcher16(const uint8_t *data, size_t size);
void write_stream_header(const StreamInfo &s, uint8_t *p /* 30 bytes */);
/* Writes one complete block (11-byte header + payload) and returns its size, which equals
 * br.bytes by construction.  chan: one record per channel (compress blocks); region: the block's
 * region of the device-packed buffer -- per channel the residual bitstring (MSB first, each starting on
 * an 8-byte boundary, chan[ch].res_bits long) for compress blocks, the final payload bytes for raw. */
uint32_t pack_block(const StreamInfo &s, const SrlaBlockRecord &br, const SrlaChanRecord *chan,
                    const uint8_t *region, uint8_t *out);
const unsigned char *huffman_plain_lengths();
const unsigned char *huffman_summed_lengths();

}  // namespace srla
#endif
