/* host_pack.h -- what is left of the bitstream on the host (see host_pack.cpp). */
#ifndef SRLA_HOST_PACK_H
#define SRLA_HOST_PACK_H

#include <stddef.h>
#include <stdint.h>

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

void write_stream_header(const StreamInfo &s, uint8_t *p /* 30 bytes */);
/* static Huffman tables of the tap codes (format constants), uploaded to the device */
const unsigned char *huffman_plain_lengths();
const unsigned char *huffman_summed_lengths();
const unsigned int *huffman_plain_codes();
const unsigned int *huffman_summed_codes();

}  // namespace srla
#endif
