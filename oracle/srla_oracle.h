/*
 * srla_oracle.h -- CPU oracle for the SRLA encode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's algorithm
 * (aikiriao/SRLA, codec version 18 / format version 10) used as the parity checker for the
 * HIP implementation in srla_amd/csrc and as the `cpu_baseline` leg of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it; the product
 * library never links or calls anything in oracle/.
 *
 * Parity is PINNED: tests/test_oracle_vs_reference.py byte-compares this oracle with the
 * compiled reference (oracle/_ref/libsrla_ref.so, built from /root/reference by
 * oracle/Makefile) where the reference exists, and tests/golden/ holds vectors generated
 * from that compiled reference (tools/gen_golden.py) for machines where it does not.
 */
#ifndef SRLA_ORACLE_H
#define SRLA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_CHANNELS 8
#define ORACLE_MAX_NODES    1025  /* look-ahead / minimum block + 1 */
#define ORACLE_MAX_ORDER    255
#define ORACLE_LTP_TAPS     3

typedef struct OracleConfig {
    uint32_t num_channels;
    uint32_t bits_per_sample;
    uint32_t sampling_rate;
    uint32_t min_block;      /* min_num_samples_per_block */
    uint32_t max_block;      /* max_num_samples_per_block */
    uint32_t lookahead;      /* num_lookahead_samples     */
    uint32_t ltp_order;      /* 0, 1 or 3                 */
    uint32_t preset;         /* 0..6                      */
} OracleConfig;

/* Everything the bitstream carries for one analysed channel (struct SRLAEncoderCoefficient,
 * libs/srla_encoder/src/srla_encoder.c:23-31) plus the cost the search uses. */
typedef struct OracleChannelParams {
    int32_t  preemph_prev;
    int32_t  preemph_coef;
    uint32_t lpc_order;
    uint32_t lpc_rshift;
    uint32_t use_sum;
    uint32_t ltp_period;
    int32_t  ltp_coef[ORACLE_LTP_TAPS];
    uint32_t code_length;    /* bits, srla_encoder.c:1121-1187 */
    uint32_t res_code_type;  /* 0 rice, 1 recursive rice, 2 all zero */
    uint32_t res_porder;
    uint32_t res_bits;       /* SRLACoder_ComputeCodeLength */
    int32_t  lpc_coef[ORACLE_MAX_ORDER];
} OracleChannelParams;

typedef struct OracleBlockInfo {
    uint32_t block_type;     /* 0 compress, 1 silent, 2 raw (after the raw fall-back rule) */
    uint32_t ch_method;      /* 0 LR, 1 MS, 2 LS, 3 SR */
    uint32_t payload_bits;   /* rounded to bytes; 0 unless compress was evaluated */
    uint32_t block_bytes;    /* SRLAEncoder_ComputeBlockSize */
} OracleBlockInfo;

struct Oracle;

struct Oracle *oracle_create(const OracleConfig *cfg);
void oracle_destroy(struct Oracle *o);
/* SRLAEncoder_SetEncodeParameter on a used handle (srla_encoder.c:710-763): the parameters change -- within what the handle was
 * created for; this restatement allocates by the maximum block, so that has to stay --, the header starts over (offset shift 0), the
 * calculator and with it the persistent FFT buffer STAY.  0 on success. */
int oracle_set_parameter(struct Oracle *o, const OracleConfig *cfg);
void oracle_set_offset_lshift(struct Oracle *o, uint32_t lshift);
void oracle_set_svr_iterations(struct Oracle *o, uint32_t iterations);   /* --svr-filter-learning-iteration (lpc.c:1036-1136) */
int oracle_svr_refine(const double *data, uint32_t num_samples, double *coef, uint32_t order, uint32_t max_iter);

/* --- whole path ------------------------------------------------------------------------ */
int oracle_encode_whole(struct Oracle *o, const int32_t *const *input, uint32_t num_samples,
                        uint8_t *data, uint32_t data_size, uint32_t *output_size);
int oracle_encode_block(struct Oracle *o, const int32_t *const *input, uint32_t num_samples,
                        uint8_t *data, uint32_t data_size, uint32_t *output_size);
int oracle_compute_block_size(struct Oracle *o, const int32_t *const *input, uint32_t num_samples,
                              uint32_t *output_size);
int oracle_search_partitions(struct Oracle *o, const int32_t *const *input, uint32_t num_samples,
                             uint32_t *num_partitions, uint32_t *partitions);
/* Block analysis with every intermediate exposed: params[0..1] = chosen pair for channels 0/1
 * (params[ch] for ch >= 2), variants[0..3] = L, R, M, S analyses (stereo only),
 * residual_out[ch] (may be NULL) receives the chosen residuals. */
int oracle_analyze_block(struct Oracle *o, const int32_t *const *input, uint32_t num_samples,
                         OracleBlockInfo *info, OracleChannelParams *params,
                         OracleChannelParams *variants, int32_t *const *residual_out);
/* One channel variant: `buf` holds the variant's samples on entry (already >> offset_lshift,
 * already M/S-combined) and the pre-emphasised (and LTP-filtered) signal on return. */
int oracle_analyze_channel(struct Oracle *o, int32_t *buf, uint32_t num_samples,
                           int32_t *residual, OracleChannelParams *out);

/* --- decoder (verifier) ---------------------------------------------------------------- */
int oracle_decode_header(const uint8_t *data, uint32_t data_size, OracleConfig *cfg_out,
                         uint32_t *num_samples, uint32_t *offset_lshift);
int oracle_decode_whole(const uint8_t *data, uint32_t data_size, int32_t *const *buffer,
                        uint32_t buffer_channels, uint32_t buffer_samples);
/* Walk the block headers of a stream: fills up to `cap` entries of (type, num_samples, bytes). */
int oracle_list_blocks(const uint8_t *data, uint32_t data_size, uint32_t *types,
                       uint32_t *nsamples, uint32_t *nbytes, uint32_t cap, uint32_t *count);

/* --- stage-level pieces (parity tests of individual kernels) --------------------------- */
uint32_t oracle_offset_lshift(const int32_t *const *input, uint32_t num_channels, uint32_t num_samples);
uint16_t oracle_fletcher16(const uint8_t *data, uint32_t size);
void oracle_lr_to_ms(int32_t *ch0, int32_t *ch1, uint32_t n);
int32_t oracle_preemphasis_coef(const int32_t *x, uint32_t n);
void oracle_preemphasis(int32_t *x, uint32_t n, int32_t prev, int32_t coef);
void oracle_fft_real(int n, int flag, double *x, double *work);
void oracle_welch_window(const double *in, uint32_t n, double *out);
/* Windowed FFT autocorrelation of `signal` (length n); `state` supplies the persistent FFT
 * buffer of the handle (odd n keeps its middle sample, lpc.c:260-264). */
void oracle_autocorr(struct Oracle *o, const double *signal, uint32_t n, double *lags, uint32_t num_lags);
/* Levinson-Durbin for all orders: coefs is [order][order] row-major, error_vars[order+1],
 * window-compensated as LPC_CalculateCoef does (lpc.c:444-500). */
void oracle_levinson(const double *lags_ridged, uint32_t order, uint32_t num_samples,
                     double *coefs, double *error_vars);
uint32_t oracle_select_order(const double *error_vars, uint32_t max_order, uint32_t num_samples,
                             uint32_t bits_per_sample, double *lens_out);
void oracle_quantize(const double *coef, uint32_t order, int32_t *icoef, uint32_t *rshift);
void oracle_lpc_predict(const int32_t *data, uint32_t n, const int32_t *coef, uint32_t order,
                        int32_t *residual, uint32_t rshift);
void oracle_ltp_predict(const int32_t *data, uint32_t n, const int32_t *coef, uint32_t order,
                        uint32_t period, int32_t *residual, uint32_t rshift);
int oracle_detect_pitch(const double *lags, uint32_t min_period, uint32_t max_period, uint32_t *period);
int oracle_ltp_coefficients(struct Oracle *o, const double *signal, uint32_t n, uint32_t order,
                            double *coef, uint32_t *period);
void oracle_residual_code_search(const int32_t *data, uint32_t n, uint32_t *code_type,
                                 uint32_t *porder, uint32_t *bits);
uint32_t oracle_rice_k(double mean);
uint32_t oracle_recursive_rice_k2(double mean);
uint32_t oracle_coef_bits(const int32_t *coef, uint32_t order, uint32_t *use_sum);
/* Dijkstra on a dense matrix (row-major n*n, 2^24 = no edge); path[] as the reference leaves it. */
int oracle_dijkstra(const double *adj, uint32_t n, uint32_t start, uint32_t goal,
                    double *min_cost, uint32_t *path);

#ifdef __cplusplus
}
#endif
#endif /* SRLA_ORACLE_H */
