/*
 * synth.c -- deterministic, integer-only synthetic PCM generator for tests and bench.py.
 *
 * Not part of the product and not part of the oracle: it only manufactures the "synthetic
 * 48 kHz/16-bit stereo WAV" content BASELINE.json asks for.  Everything is integer arithmetic
 * (xorshift PRNG, phase-accumulator oscillators on a sine table built by an exact Q62 rotation
 * recurrence, integer AR(2) noise shaping), so the same (kind, seed, length) gives the same
 * samples on every machine and the SHA-256 fixtures in tests/golden stay valid.
 *
 * kinds:
 *   0  sine      440 Hz, 0.5 full scale, identical in every channel        (BASELINE config C1)
 *   1  music     6 vibrato voices x 5 partials + AR(2) coloured noise under a slow envelope,
 *                about -25 dBFS RMS; right = 0.78 * left-tonal + independent noise
 *   2  varied    one-second sections cycling white noise / pure tone / resonant noise /
 *                impulse train / harmonic stack / one silent channel / digital silence /
 *                very low level, so RAW, SILENT, RICE, ALLZERO, LS/SR paths all occur
 *   3  noise     full-band white noise at -12 dBFS
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TABLE_BITS 12
#define TABLE_SIZE (1 << TABLE_BITS)

static int32_t g_sine[TABLE_SIZE]; /* Q30 */
static int g_sine_ready = 0;

static void build_sine(void)
{
    /* rotate (1,0) by 2*pi/4096 per step in Q62 with 128-bit products */
    const __int128 c = (__int128)4611680592556051597LL;
    const __int128 s = (__int128)7074234977634094LL;
    __int128 x = ((__int128)1) << 62, y = 0;
    int i;
    for (i = 0; i < TABLE_SIZE; i++) {
        const __int128 nx = (x * c - y * s + (((__int128)1) << 61)) >> 62;
        const __int128 ny = (x * s + y * c + (((__int128)1) << 61)) >> 62;
        g_sine[i] = (int32_t)((y + (((__int128)1) << 31)) >> 32);
        x = nx;
        y = ny;
    }
    g_sine_ready = 1;
}

static inline int32_t osc(uint32_t phase) { return g_sine[phase >> (32 - TABLE_BITS)]; } /* Q30 */

typedef struct { uint64_t s; } Rng;
static inline uint64_t rng_next(Rng *r)
{
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 2685821657736338717ULL;
}
static inline int32_t rng_tri(Rng *r, int bits)
{
    /* triangular noise in (-2^bits, 2^bits) */
    const uint64_t v = rng_next(r);
    const int32_t a = (int32_t)(v & ((1u << bits) - 1u));
    const int32_t b = (int32_t)((v >> 32) & ((1u << bits) - 1u));
    return a - b;
}
static inline uint32_t rng_range(Rng *r, uint32_t lo, uint32_t hi) { return lo + (uint32_t)(rng_next(r) % (hi - lo + 1)); }

static inline uint32_t hz_to_inc(uint32_t millihz, uint32_t rate)
{
    return (uint32_t)((((uint64_t)millihz << 32) / 1000u) / rate);
}

static inline int32_t clip_bits(int64_t v, int bps)
{
    const int64_t hi = ((int64_t)1 << (bps - 1)) - 1, lo = -((int64_t)1 << (bps - 1));
    return (int32_t)(v > hi ? hi : (v < lo ? lo : v));
}

#ifndef TONAL_GAIN
#define TONAL_GAIN 1200
#endif
#ifndef NOISE_SHIFT
#define NOISE_SHIFT 15
#endif
#define NVOICE 6
#define NPART  5
typedef struct {
    uint32_t phase[NPART], inc, vib_phase, vib_inc, vib_depth, env_phase, env_inc;
    int32_t amp[NPART];
} Voice;

static void gen_music(Rng *rng, uint32_t rate, uint32_t nch, uint32_t n, int bps, int32_t *const *out)
{
    Voice v[NVOICE];
    int64_t ar1[2] = { 0, 0 }, ar2[2] = { 0, 0 };
    uint32_t i, k, p, slow_phase = 0;
    const uint32_t slow_inc = hz_to_inc(70, rate);
    const int shift = 30 - (bps - 1); /* Q30 -> full scale of bps */
    for (k = 0; k < NVOICE; k++) {
        const uint32_t f0 = rng_range(rng, 80000, 900000);
        v[k].inc = hz_to_inc(f0, rate);
        v[k].vib_inc = hz_to_inc(rng_range(rng, 4000, 7000), rate);
        v[k].vib_depth = rng_range(rng, 3, 10); /* per mille of inc */
        v[k].vib_phase = (uint32_t)rng_next(rng);
        v[k].env_inc = hz_to_inc(rng_range(rng, 100, 500), rate);
        v[k].env_phase = (uint32_t)rng_next(rng);
        for (p = 0; p < NPART; p++) {
            v[k].phase[p] = (uint32_t)rng_next(rng);
            v[k].amp[p] = (int32_t)(rng_range(rng, 400, 1000) / (p + 1)); /* 1/1000 units */
        }
    }
    for (i = 0; i < n; i++) {
        int64_t tonal = 0; /* Q30 * 1e-3 units */
        int64_t l, r, nl, nr;
        int32_t slow;
        for (k = 0; k < NVOICE; k++) {
            const int32_t vib = osc(v[k].vib_phase) >> 15;                    /* Q15 */
            const int64_t dinc = ((int64_t)v[k].inc * v[k].vib_depth / 1000) * vib >> 15;
            const uint32_t inc = (uint32_t)((int64_t)v[k].inc + dinc);
            const int32_t env = (osc(v[k].env_phase) >> 16) + 24576;          /* 8192..40960 */
            int64_t acc = 0;
            for (p = 0; p < NPART; p++) {
                acc += (int64_t)osc(v[k].phase[p]) * v[k].amp[p];
                v[k].phase[p] += inc * (p + 1);
            }
            tonal += (acc >> 15) * env >> 15;
            v[k].vib_phase += v[k].vib_inc;
            v[k].env_phase += v[k].env_inc;
        }
        slow = (osc(slow_phase) >> 16) + 20480; /* 4096..36864 */
        slow_phase += slow_inc;
        /* calibrated so that the tonal part sits near -26 dBFS and the noise near -46 dBFS */
        tonal = ((tonal / 1000) * slow >> 15) * TONAL_GAIN;
        /* AR(2) noise, poles 1.6 / -0.8 (Q14), white input ~ +-2^22 in Q30 */
        {
            const int64_t e0 = (int64_t)rng_tri(rng, 22), e1 = (int64_t)rng_tri(rng, 22);
            nl = ((26214 * ar1[0] - 13107 * ar2[0]) >> 14) + e0;
            ar2[0] = ar1[0]; ar1[0] = nl;
            nr = ((26214 * ar1[1] - 13107 * ar2[1]) >> 14) + e1;
            ar2[1] = ar1[1]; ar1[1] = nr;
        }
        l = tonal + ((nl * slow) >> NOISE_SHIFT);
        r = (tonal * 799 >> 10) + ((nr * slow) >> NOISE_SHIFT);
        out[0][i] = clip_bits(l >> shift, bps);
        if (nch > 1) out[1][i] = clip_bits(r >> shift, bps);
        for (k = 2; k < nch; k++) out[k][i] = clip_bits(((k & 1) ? l : r) >> (shift + 1), bps);
    }
}

static void gen_varied(Rng *rng, uint32_t rate, uint32_t nch, uint32_t n, int bps, int32_t *const *out)
{
    const int shift = 30 - (bps - 1);
    uint32_t pos = 0, section = 0, ch, i;
    while (pos < n) {
        const uint32_t len = (n - pos < rate) ? (n - pos) : rate;
        const uint32_t mode = section % 8;
        const uint32_t f = rng_range(rng, 100000, 6000000);
        const uint32_t inc = hz_to_inc(f, rate);
        const uint32_t period = rng_range(rng, 60, 700);
        section++;
        for (ch = 0; ch < nch; ch++) {
            uint32_t phase = (uint32_t)rng_next(rng);
            int64_t y1 = 0, y2 = 0;
            for (i = 0; i < len; i++) {
                int64_t v = 0;
                switch (mode) {
                case 0: v = (int64_t)rng_tri(rng, 27); break;                                  /* white */
                case 1: v = (int64_t)osc(phase) >> 2; phase += inc; break;                     /* tone  */
                case 2: {                                                                      /* resonant noise */
                    const int64_t e = rng_tri(rng, 20);
                    const int64_t y = ((31130 * y1 - 15729 * y2) >> 14) + e;
                    y2 = y1; y1 = y; v = y; break; }
                case 3: v = ((pos + i) % period == 0) ? ((int64_t)1 << 28) : 0; break;         /* impulses */
                case 4: {                                                                      /* harmonic stack */
                    uint32_t h; for (h = 1; h <= 8; h++) v += (int64_t)osc(phase * h) / (int64_t)(4 * h);
                    phase += inc >> 2; break; }
                case 5: v = (ch == 0) ? ((int64_t)osc(phase) >> 3) + rng_tri(rng, 18) : 0; phase += inc; break; /* one silent ch */
                case 6: v = 0; break;                                                          /* digital silence */
                default: v = (int64_t)rng_tri(rng, shift + 2); break;                          /* a few LSBs */
                }
                out[ch][pos + i] = clip_bits(v >> shift, bps);
            }
        }
        if (mode == 1 && nch > 1) { /* make the tone section strongly correlated: R = L */
            memcpy(out[1] + pos, out[0] + pos, sizeof(int32_t) * len);
        }
        pos += len;
    }
}

int synth_generate(uint32_t kind, uint64_t seed, uint32_t rate, uint32_t nch, uint32_t n, uint32_t bps,
                   int32_t *const *out)
{
    Rng rng;
    uint32_t i, ch;
    if (!g_sine_ready) build_sine();
    if (nch == 0 || n == 0 || bps < 8 || bps > 24) return -1;
    rng.s = seed * 0x9E3779B97F4A7C15ULL + 0x632BE59BD9B4E019ULL;
    if (rng.s == 0) rng.s = 1;
    for (i = 0; i < 8; i++) (void)rng_next(&rng);
    switch (kind) {
    case 0: {
        const uint32_t inc = hz_to_inc(440000, rate);
        uint32_t phase = 0;
        for (i = 0; i < n; i++) {
            const int32_t v = clip_bits((int64_t)osc(phase) >> (30 - (bps - 1) + 1), (int)bps);
            for (ch = 0; ch < nch; ch++) out[ch][i] = v;
            phase += inc;
        }
        return 0; }
    case 1: gen_music(&rng, rate, nch, n, (int)bps, out); return 0;
    case 2: gen_varied(&rng, rate, nch, n, (int)bps, out); return 0;
    case 3:
        for (ch = 0; ch < nch; ch++)
            for (i = 0; i < n; i++) out[ch][i] = clip_bits((int64_t)rng_tri(&rng, (int)bps - 3), (int)bps);
        return 0;
    default: return -1;
    }
}
