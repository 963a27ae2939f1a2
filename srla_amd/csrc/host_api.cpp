/*
 * host_api.cpp -- the C ABI of include/srla_mi355x.h: argument checking exactly as the reference API
 * (libs/srla_encoder/src/srla_encoder.c), then the host runtime (host_impl.h).
 */
#include "host_impl.h"

#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <string.h>

static_assert(sizeof(SrlaItemResult) == SRLAMI355X_ITEM_RECORD_BYTES, "record size");
static_assert(SRLA_DBG_STRIDE == SRLAMI355X_DEBUG_DOUBLES, "debug stride");

namespace {

int32_t work_size_of(const SRLAEncoderConfig *config)
{
    /* validity rules of srla_encoder.c:468-496 */
    if (config == NULL) return -1;
    if (config->max_num_samples_per_block == 0 || config->min_num_samples_per_block == 0
        || config->max_num_lookahead_samples == 0 || config->max_num_channels == 0) return -1;
    if (config->max_num_parameters > config->max_num_samples_per_block) return -1;
    if (config->min_num_samples_per_block > config->max_num_samples_per_block) return -1;
    if (config->max_num_lookahead_samples < config->max_num_samples_per_block) return -1;
    return (int32_t)(sizeof(SRLAEncoder) + 16);
}

Impl *impl_of(SRLAEncoder *e) { return (e && e->magic == SRLA_HANDLE_MAGIC) ? e->impl : nullptr; }

}  // namespace

extern "C" {

const char *SRLAMI355X_Version(void) { return "srla-mi355x 0.1 (gfx950 HIP; SRLA codec 18 / format 10)"; }

int SRLAMI355X_SetDevice(int device_index)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_index < 0 || device_index >= count) return -1;
    g_device_index = device_index;
    return (hipSetDevice(device_index) == hipSuccess) ? 0 : -1;
}

SRLAApiResult SRLAEncoder_EncodeHeader(const struct SRLAHeader *header, uint8_t *data, uint32_t data_size)
{
    /* srla_encoder.c:85-165 */
    if (header == NULL || data == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (data_size < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (header->num_channels == 0 || header->num_samples == 0 || header->sampling_rate == 0
        || header->bits_per_sample == 0 || header->offset_lshift >= 32 || header->max_num_samples_per_block == 0
        || header->preset >= SRLA_NUM_PARAMETER_PRESETS) return SRLA_APIRESULT_INVALID_FORMAT;
    srla::StreamInfo si;
    si.num_channels = header->num_channels; si.bits_per_sample = header->bits_per_sample;
    si.sampling_rate = header->sampling_rate; si.num_samples = header->num_samples;
    si.offset_lshift = header->offset_lshift; si.max_block = header->max_num_samples_per_block;
    si.preset = header->preset; si.ltp_order = 0;
    srla::write_stream_header(si, data);
    return SRLA_APIRESULT_OK;
}

int32_t SRLAEncoder_CalculateWorkSize(const struct SRLAEncoderConfig *config) { return work_size_of(config); }

struct SRLAEncoder *SRLAEncoder_Create(const struct SRLAEncoderConfig *config, void *work, int32_t work_size)
{
    /* srla_encoder.c:549-694 */
    uint8_t own = 0;
    if (work == NULL && work_size == 0) {
        if ((work_size = work_size_of(config)) < 0) return NULL;
        work = malloc((size_t)work_size);
        own = 1;
    }
    if (config == NULL || work == NULL || work_size < work_size_of(config) || work_size_of(config) < 0) {
        if (own) free(work);
        return NULL;
    }
    SRLAEncoder *e = reinterpret_cast<SRLAEncoder *>(((uintptr_t)work + 15u) & ~(uintptr_t)15u);
    e->magic = SRLA_HANDLE_MAGIC;
    e->alloced_by_own = own;
    e->work = work;
    e->impl = new Impl();
    e->impl->cfg = *config;
    e->impl->device = g_device_index;
    return e;
}

void SRLAEncoder_Destroy(struct SRLAEncoder *encoder)
{
    if (encoder == NULL || encoder->magic != SRLA_HANDLE_MAGIC) return;
    delete encoder->impl;
    encoder->impl = nullptr;
    encoder->magic = 0;
    if (encoder->alloced_by_own) free(encoder->work);
}

SRLAApiResult SRLAEncoder_SetEncodeParameter(struct SRLAEncoder *encoder, const struct SRLAEncodeParameter *p)
{
    /* srla_encoder.c:710-763 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || p == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (p->num_channels == 0 || p->bits_per_sample == 0 || p->sampling_rate == 0
        || p->preset >= SRLA_NUM_PARAMETER_PRESETS) return SRLA_APIRESULT_INVALID_FORMAT;
    if (p->min_num_samples_per_block == 0) return SRLA_APIRESULT_INVALID_FORMAT; /* the reference divides by it */
    if (p->min_num_samples_per_block > p->max_num_samples_per_block
        || p->num_lookahead_samples < p->max_num_samples_per_block
        || (p->num_lookahead_samples % p->min_num_samples_per_block) != 0
        || (p->ltp_order > 0 && (p->ltp_order % 2) == 0) || p->ltp_order > SRLA_MAX_LTP_ORDER)
        return SRLA_APIRESULT_INVALID_FORMAT;
    if (im->cfg.max_num_samples_per_block < p->max_num_samples_per_block
        || im->cfg.min_num_samples_per_block > p->min_num_samples_per_block
        || im->cfg.max_num_lookahead_samples < p->num_lookahead_samples
        || im->cfg.max_num_channels < p->num_channels) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    /* limits of this implementation (documented in DESIGN.md) */
    if (p->num_channels > SRLA_MAX_CH) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (p->max_num_samples_per_block > 65535u) {
        /* (a block header holds its sample count in 16 bits, srla_encoder.c:1583-1595: larger blocks cannot be written by the reference either) */
        fprintf(stderr, "[srla-mi355x] max block size %u exceeds the limit of %u samples\n",
                p->max_num_samples_per_block, 65535u);
        return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    }
    if (p->num_lookahead_samples / p->min_num_samples_per_block + 1 > SRLA_MAX_NODES) {
        fprintf(stderr, "[srla-mi355x] look-ahead / min block = %u exceeds %u search nodes\n",
                p->num_lookahead_samples / p->min_num_samples_per_block, SRLA_MAX_NODES - 1);
        return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    }
    if (p->bits_per_sample != 8 && p->bits_per_sample != 16 && p->bits_per_sample != 24) return SRLA_APIRESULT_INVALID_FORMAT;
    /* SVR refinement (--svr-filter-learning-iteration, lpc.c:1036-1136): every preset and block size; orders above 64 and
     * blocks above 8192 samples take the global-memory version of the kernel */
    im->par = *p;
    im->param_generation++;
    im->offset_lshift = 0;
    im->set_parameter = true;
    /* bit-identity is the contract: parameters under which it cannot be promised are named here, loudly, and every call made
     * under them is counted (SRLAMI355XStats::num_nonidentical_calls) */
    im->warned_reasons = 0;
    if (const uint32_t r = im->nonidentical_reasons(0)) {
        im->warned_reasons = r;
        fprintf(stderr, "[srla-mi355x] WARNING: with these parameters the output is valid and lossless but NOT guaranteed bit-identical to the reference: %s\n",
                Impl::nonidentical_text(r).c_str());
    }
    return SRLA_APIRESULT_OK;
}

uint32_t SRLAMI355X_NonIdenticalReasons(struct SRLAEncoder *encoder, uint32_t num_samples)
{
    Impl *im = impl_of(encoder);
    return (im && im->set_parameter) ? im->nonidentical_reasons(num_samples) : 0u;
}

/* OR of every sample of planar host input: the one whole-stream quantity (srla_utility.c:177-203), by the handle's pool threads */
SRLAApiResult SRLAMI355X_OrMask(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples, uint32_t *mask)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || mask == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    const uint32_t nch = im->par.num_channels, chunk = 1u << 20, per_ch = (num_samples + chunk - 1) / chunk;
    /* a host-only reduction: the pool threads where the handle has them (they come with the device), else this thread alone --
     * the one call of the library that works without a GPU */
    if (im->pool == nullptr && !im->init_device()) {
        uint32_t m = 0;
        for (uint32_t ch = 0; ch < nch; ch++) m |= srla::or_reduce(input[ch], num_samples);
        *mask = m;
        return SRLA_APIRESULT_OK;
    }
    std::atomic<uint32_t> acc{ 0 };
    im->pool->parallel_for(per_ch * nch, [&](uint32_t i) {
        const uint32_t ch = i / per_ch, o = (i % per_ch) * chunk, len = std::min(chunk, num_samples - o);
        acc.fetch_or(srla::or_reduce(input[ch] + o, len), std::memory_order_relaxed);
    });
    *mask = acc.load();
    return SRLA_APIRESULT_OK;
}

/* one stream of host samples through encode_streams */
static SRLAApiResult one_stream(Impl *im, const int32_t *const *input, const int32_t *d_input, uint32_t d_stride, uint32_t num_samples,
                                uint8_t *data, uint32_t data_size, uint32_t *output_size, SRLAEncoder_EncodeBlockCallback cb,
                                bool with_header, bool search)
{
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    StreamCtx st;
    st.host_in = input; st.d_in = d_input; st.d_stride = d_stride; st.num_samples = num_samples;
    st.data = data; st.data_size = data_size; st.with_header = with_header; st.cb = cb;
    st.reference_call = true;
    im->sx.clear();
    im->sx.push_back(st);
    const SRLAApiResult rc = im->encode_streams(search);
    if (rc == SRLA_APIRESULT_OK && output_size) *output_size = im->sx[0].write_off;
    return rc;
}

SRLAApiResult SRLAEncoder_ComputeBlockSize(struct SRLAEncoder *encoder, const int32_t *const *input,
                                           uint32_t num_samples, uint32_t *output_size)
{
    /* srla_encoder.c:1477-1546: the size of the block EncodeBlock would write (no output buffer: the bytes stay in the
     * library's staging memory) */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || num_samples == 0 || output_size == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > im->par.max_num_samples_per_block) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    /* With more than two channels the reference's answer is NOT the size EncodeBlock writes: ComputeCoefficients adds up the code
     * lengths of the first two channels only (srla_encoder.c:1287-1301), and that sum is what ComputeBlockSize returns and what its
     * RAW fall-back compares (:1519-1532) -- the price the block division search works with.  The same number here. */
    im->want_block_price = im->par.num_channels > 2 && !im->no_chain;
    im->block_price = 0;
    const SRLAApiResult rc = one_stream(im, input, nullptr, 0, num_samples, nullptr, 0, output_size, nullptr, false, false);
    if (rc == SRLA_APIRESULT_OK && im->want_block_price) *output_size = im->block_price;
    im->want_block_price = false;
    return rc;
}

SRLAApiResult SRLAEncoder_EncodeBlock(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                      uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    /* srla_encoder.c:1549-1643 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || num_samples == 0 || data == NULL || data_size == 0 || output_size == NULL)
        return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > im->par.max_num_samples_per_block) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    return one_stream(im, input, nullptr, 0, num_samples, data, data_size, output_size, nullptr, false, false);
}

SRLAApiResult SRLAEncoder_EncodeOptimalPartitionedBlock(struct SRLAEncoder *encoder, const int32_t *const *input,
                                                        uint32_t num_samples, uint8_t *data, uint32_t data_size,
                                                        uint32_t *output_size)
{
    /* srla_encoder.c:1646-1698 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || data == NULL || output_size == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples == 0 || num_samples > im->par.num_lookahead_samples) return SRLA_APIRESULT_NG;
    return one_stream(im, input, nullptr, 0, num_samples, data, data_size, output_size, nullptr, false, true);
}

SRLAApiResult SRLAEncoder_EncodeWhole(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                      uint8_t *data, uint32_t data_size, uint32_t *output_size,
                                      SRLAEncoder_EncodeBlockCallback encode_callback)
{
    /* srla_encoder.c:1701-1788 */
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || data == NULL || output_size == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (data_size < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (num_samples == 0) return SRLA_APIRESULT_INVALID_FORMAT;
    return one_stream(im, input, nullptr, 0, num_samples, data, data_size, output_size, encode_callback, true, im->search_enabled());
}

SRLAApiResult SRLAMI355X_EncodeWholeDevice(struct SRLAEncoder *encoder, const int32_t *d_input, uint32_t channel_stride,
                                           uint32_t num_samples, uint8_t *data, uint32_t data_size, uint32_t *output_size,
                                           SRLAEncoder_EncodeBlockCallback encode_callback)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || d_input == NULL || data == NULL || output_size == NULL || channel_stride < num_samples)
        return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (data_size < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (num_samples == 0) return SRLA_APIRESULT_INVALID_FORMAT;
    return one_stream(im, nullptr, d_input, channel_stride, num_samples, data, data_size, output_size, encode_callback, true,
                      im->search_enabled());
}

SRLAApiResult SRLAMI355X_EncodeWindows(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                       uint32_t offset_lshift, int is_stream_end, uint8_t *data, uint32_t data_size, uint32_t *output_size)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || data == NULL || output_size == NULL || offset_lshift >= 32) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples == 0) return SRLA_APIRESULT_INVALID_FORMAT;
    const bool search = im->search_enabled();
    const uint32_t window_len = search ? im->par.num_lookahead_samples : im->par.max_num_samples_per_block;
    if (!is_stream_end && (num_samples % window_len) != 0) return SRLA_APIRESULT_INVALID_ARGUMENT;   /* only the stream's end may hold a partial window */
    if (im->history_regime(search)) {
        /* an odd minimum block, or LTP with blocks of at most 256 samples: every window's analysis depends on the windows before
         * it, so a range of windows cannot be encoded on its own */
        fprintf(stderr, "[srla-mi355x] SRLAMI355X_EncodeWindows: under these parameters the windows of a stream are not independent "
                        "(history mode); encode the stream with SRLAEncoder_EncodeWhole\n");
        return SRLA_APIRESULT_INVALID_FORMAT;
    }
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    StreamCtx st;
    st.host_in = input; st.num_samples = num_samples; st.data = data; st.data_size = data_size; st.with_header = false;
    st.lshift = offset_lshift; st.lshift_final = true;
    im->sx.clear();
    im->sx.push_back(st);
    const SRLAApiResult rc = im->encode_streams(search);
    if (rc == SRLA_APIRESULT_OK) *output_size = im->sx[0].write_off;
    return rc;
}

SRLAApiResult SRLAMI355X_EncodeBatch(struct SRLAEncoder *encoder, uint32_t num_streams, const int32_t *const *const *inputs,
                                     const uint32_t *num_samples, uint8_t *const *data, const uint32_t *data_size,
                                     uint32_t *output_size, SRLAApiResult *results)
{
    return SRLAMI355X_EncodeBatchEx(encoder, num_streams, inputs, num_samples, NULL, data, data_size, output_size, results);
}

void *SRLAMI355X_AllocHost(size_t bytes)
{
    void *p = nullptr;
    if (hipSetDevice(g_device_index) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return NULL; }
    return p;
}

void SRLAMI355X_FreeHost(void *p)
{
    if (p) (void)hipHostFree(p);
}

SRLAApiResult SRLAMI355X_EncodeBatchPcm(struct SRLAEncoder *encoder, uint32_t num_streams, const void *const *frames,
                                        const uint32_t *num_samples, uint32_t bytes_per_sample, uint8_t *const *data,
                                        const uint32_t *data_size, uint32_t *output_size, SRLAApiResult *results)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || num_streams == 0 || frames == NULL || num_samples == NULL || data == NULL || data_size == NULL || output_size == NULL)
        return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (bytes_per_sample * 8u != im->par.bits_per_sample) return SRLA_APIRESULT_INVALID_FORMAT;   /* the container is the sample format */
    for (uint32_t i = 0; i < num_streams; i++) {
        if (frames[i] == NULL || data[i] == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
        if (num_samples[i] == 0) return SRLA_APIRESULT_INVALID_FORMAT;
        if (data_size[i] < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    }
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    const uint32_t nch = im->par.num_channels;
    /* the frames must be readable by DMA: page-locked already (SRLAMI355X_AllocHost, hipHostMalloc, hipHostRegister), or locked
     * in place for the call */
    struct Pins { std::vector<const void *> held; ~Pins() { for (const void *p : held) host_pin_release(p); } } pins;
    bool dma = !im->force_staging;
    for (uint32_t i = 0; i < num_streams && dma; i++) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof(at));
        if (hipPointerGetAttributes(&at, frames[i]) == hipSuccess && at.type == hipMemoryTypeHost) {
            /* locked by another handle's call (the registry)?  keep it locked for this one too */
            if (const void *key = host_pin_addref(frames[i])) pins.held.push_back(key);
            continue;
        }
        (void)hipGetLastError();
        if (host_pin_acquire(frames[i], (size_t)num_samples[i] * nch * bytes_per_sample, nullptr)) pins.held.push_back(frames[i]);
        else dma = false;
    }
    /* otherwise (locking refused, SRLA_MI355X_STAGING): de-interleave on the host and take the ordinary path */
    std::vector<std::vector<int32_t>> planes;
    std::vector<std::vector<const int32_t *>> plane_ptrs;
    if (!dma) {
        planes.resize(num_streams); plane_ptrs.resize(num_streams);
        for (uint32_t i = 0; i < num_streams; i++) {
            planes[i].resize((size_t)nch * num_samples[i]);
            plane_ptrs[i].resize(nch);
            for (uint32_t ch = 0; ch < nch; ch++) {
                int32_t *dst = planes[i].data() + (size_t)ch * num_samples[i];
                (void)pcm_channel(static_cast<const uint8_t *>(frames[i]), bytes_per_sample, nch, ch, 0, num_samples[i], dst);
                plane_ptrs[i][ch] = dst;
            }
        }
    }
    im->sx.clear();
    im->sx.resize(num_streams);
    for (uint32_t i = 0; i < num_streams; i++) {
        StreamCtx &st = im->sx[i];
        if (dma) { st.pcm = static_cast<const uint8_t *>(frames[i]); st.pcm_bytes = bytes_per_sample; }
        else st.host_in = plane_ptrs[i].data();
        st.num_samples = num_samples[i];
        st.data = data[i]; st.data_size = data_size[i]; st.with_header = true;
    }
    const SRLAApiResult rc = im->encode_streams(im->search_enabled());
    if (rc != SRLA_APIRESULT_OK && rc != SRLA_APIRESULT_INSUFFICIENT_BUFFER) return rc;
    for (uint32_t i = 0; i < num_streams; i++) {
        output_size[i] = (im->sx[i].rc == SRLA_APIRESULT_OK) ? im->sx[i].write_off : 0u;
        if (results) results[i] = im->sx[i].rc;
    }
    return rc;
}

SRLAApiResult SRLAMI355X_EncodeBatchEx(struct SRLAEncoder *encoder, uint32_t num_streams, const int32_t *const *const *inputs,
                                       const uint32_t *num_samples, const uint32_t *sample_or, uint8_t *const *data,
                                       const uint32_t *data_size, uint32_t *output_size, SRLAApiResult *results)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || num_streams == 0 || inputs == NULL || num_samples == NULL || data == NULL || data_size == NULL || output_size == NULL)
        return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    for (uint32_t i = 0; i < num_streams; i++) {
        if (inputs[i] == NULL || data[i] == NULL) return SRLA_APIRESULT_INVALID_ARGUMENT;
        if (num_samples[i] == 0) return SRLA_APIRESULT_INVALID_FORMAT;
        if (data_size[i] < SRLA_HEADER_SIZE) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    }
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    im->sx.clear();
    im->sx.resize(num_streams);
    for (uint32_t i = 0; i < num_streams; i++) {
        StreamCtx &st = im->sx[i];
        st.host_in = inputs[i]; st.num_samples = num_samples[i];
        st.data = data[i]; st.data_size = data_size[i]; st.with_header = true;
        if (sample_or != NULL) {
            /* the caller gathered the OR of the stream's samples already (srla_utility.c:177): the offset shift is known */
            uint32_t sh = 0;
            if (sample_or[i] != 0) while (((sample_or[i] >> sh) & 1u) == 0) sh++;
            st.lshift = sh; st.lshift_final = true;
        }
    }
    const SRLAApiResult rc = im->encode_streams(im->search_enabled());
    if (rc != SRLA_APIRESULT_OK && rc != SRLA_APIRESULT_INSUFFICIENT_BUFFER) return rc;
    for (uint32_t i = 0; i < num_streams; i++) {
        output_size[i] = (im->sx[i].rc == SRLA_APIRESULT_OK) ? im->sx[i].write_off : 0u;
        if (results) results[i] = im->sx[i].rc;
    }
    return rc;
}

void SRLAMI355X_SetPackThreads(struct SRLAEncoder *encoder, uint32_t num_threads)
{
    Impl *im = impl_of(encoder);
    if (!im) return;
    im->pack_threads = num_threads;
    if (im->pool) { delete im->pool; im->pool = new Pool(num_threads ? num_threads : 1); }
}

void SRLAMI355X_GetStats(struct SRLAEncoder *encoder, struct SRLAMI355XStats *stats, int reset)
{
    (void)SRLAMI355X_GetStatsSized(encoder, stats, (uint32_t)sizeof(struct SRLAMI355XStats), reset);
}

uint32_t SRLAMI355X_GetStatsSized(struct SRLAEncoder *encoder, void *stats, uint32_t stats_bytes, int reset)
{
    Impl *im = impl_of(encoder);
    if (!im) return 0;
    const uint32_t n = std::min<uint32_t>(stats_bytes, (uint32_t)sizeof(im->stats));
    if (stats && n) memcpy(stats, &im->stats, n);
    if (reset) memset(&im->stats, 0, sizeof(im->stats));
    return (uint32_t)sizeof(im->stats);
}

SRLAApiResult SRLAMI355X_ProbeBlock(struct SRLAEncoder *encoder, const int32_t *const *input, uint32_t num_samples,
                                    void *records, int32_t *residuals, double *debug)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || input == NULL || num_samples == 0) return SRLA_APIRESULT_INVALID_ARGUMENT;
    if (!im->set_parameter) return SRLA_APIRESULT_PARAMETER_NOT_SET;
    if (num_samples > im->par.max_num_samples_per_block) return SRLA_APIRESULT_INSUFFICIENT_BUFFER;
    if (num_samples <= im->preset_order()) return SRLA_APIRESULT_INVALID_ARGUMENT; /* RAW by length: nothing to analyse */
    if (!im->init_device()) return SRLA_APIRESULT_NG;
    StreamCtx st;
    st.host_in = input; st.num_samples = num_samples; st.with_header = false;
    st.lshift = im->offset_lshift; st.lshift_final = true;
    im->sx.clear();
    im->sx.push_back(st);
    struct Pins { std::vector<const void *> held; ~Pins() { for (const void *p : held) host_pin_release(p); } } pins;
    im->classify_buffers(im->sx[0], pins.held);
    im->overrides.clear();
    Slot &s = im->slot[0];
    JobPlan plan;
    plan.segs.push_back({ 0u, 0u, num_samples, 0u });
    plan.total = (num_samples + 15u) & ~15u;
    s.emits = true; s.merge_cb = false;
    im->keep_residuals = residuals != nullptr;
    const bool ran = im->run_job_sync(s, plan, false, true, 0);
    im->keep_residuals = false;
    if (!ran) return SRLA_APIRESULT_NG;
    const uint32_t nv = im->num_variants();
    if (records && !im->d2h(records, s.d_results.p, (size_t)nv * sizeof(SrlaItemResult))) return SRLA_APIRESULT_NG;
    if (debug && !im->d2h(debug, s.d_dbg.p, (size_t)nv * SRLA_DBG_STRIDE * sizeof(double))) return SRLA_APIRESULT_NG;
    if (residuals) {
        for (uint32_t v = 0; v < nv; v++)
            if (!im->d2h(residuals + (size_t)v * num_samples, s.d_res_ws.as<int32_t>() + s.job.items[v].res_off, (size_t)num_samples * 4))
                return SRLA_APIRESULT_NG;
    }
    return SRLA_APIRESULT_OK;
}

uint32_t SRLAMI355X_TestPack16(int16_t *dst, const int32_t *src, uint32_t n, uint32_t *wide)
{
    uint32_t w = 0;
    const uint32_t m = srla::pack16_or(dst, src, n, &w);
    if (wide) *wide = w;
    return m;
}

int SRLAMI355X_TestPlanJobs(struct SRLAEncoder *encoder, uint32_t num_streams, const uint32_t *num_samples, int device_input,
                            uint32_t *out, uint32_t cap_words)
{
    Impl *im = impl_of(encoder);
    if (im == nullptr || !im->set_parameter || num_streams == 0 || num_samples == nullptr || out == nullptr) return -1;
    const bool search = im->search_enabled();
    static const int32_t dummy = 0;
    if (!im->dev_ready) im->read_environment();     /* (a handle that has not met its device yet: the sizing variables, SRLA_MI355X_SLOTS above all, as a call would see them) */
    im->sx.assign(num_streams, StreamCtx());
    for (uint32_t i = 0; i < num_streams; i++) {
        StreamCtx &st = im->sx[i];
        st.num_samples = num_samples[i];
        st.chain_n = (num_streams == 1 && im->history_regime(search)) ? 0u : im->chain_tail(num_samples[i], search);   /* as encode_streams */
        st.body = st.num_samples - st.chain_n;
        st.d_in = device_input ? &dummy : nullptr;
    }
    std::vector<JobPlan> plan;
    im->plan_jobs(plan, search);
    im->sx.clear();
    size_t w = 0;
    for (const JobPlan &jp : plan) {
        if (w + 3 + 4 * jp.segs.size() > cap_words) return -1;
        out[w++] = jp.slot; out[w++] = (uint32_t)jp.segs.size(); out[w++] = jp.total;
        for (const SegPlan &sp : jp.segs) { out[w++] = sp.stream; out[w++] = sp.s0; out[w++] = sp.ns; out[w++] = sp.base; }
    }
    return (int)w;
}

}  /* extern "C" */
