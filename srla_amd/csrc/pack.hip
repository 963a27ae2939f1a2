/*
 * pack.hip -- from the items' prices to the bytes of the stream: srla_price_windows (stereo decision, block sizes, shortest path: one
 * wavefront per window), srla_block_offsets (byte offset of every chosen block, per-window sizes, the job's place in its streams),
 * srla_pack_blocks (one workgroup per chosen block: the COMPLETE block assembled in LDS) and srla_stream_out (the job's finished
 * bytes -> host memory where the host does not copy them itself).  DESIGN.md 3.4, 3.5.
 */
#include "kernels_common.h"
SRLA_DIAG_PHASE_READER(pack)

/* ------------------------------------------------------------------------- pricing -------- */
/* One wave per window.  Block cost: ComputeBlockSize (srla_encoder.c:1477-1546) on top of the
 * stereo decision of ComputeCoefficients (:1275-1327), one candidate per lane; path:
 * ApplyDijkstraMethod (:249-307) with its exact tie behaviour (lowest-index minimum, strict
 * improvement), the node scan and the edge relaxation spread over the lanes; partition read-back
 * (:397-421), one block record per lane. */
/* candidates of a window whose prices and end nodes are kept in LDS (8 bytes each); a window with more keeps them in a global
 * workspace (look-ahead / minimum block above 128 with a large maximum / minimum: slow, and so is everything else about such
 * parameters -- a window of `srla -e -V 7` has 57 000 candidates) */
#define SRLA_PRICE_LDS_CANDS 14400u

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, WAVE), hi = __shfl_xor((uint32_t)(v >> 32), off, WAVE);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = (o < v) ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(WAVE) void srla_price_windows(SrlaJobParams jp, const SrlaWindowDesc *__restrict__ windows,
                                                           const SrlaCandDesc *__restrict__ cands,
                                                           const SrlaItemResult *__restrict__ results,
                                                           SrlaBlockRecord *__restrict__ blocks, uint32_t lds_nodes, uint32_t lds_cands,
                                                           uint32_t *__restrict__ price_ws)
{
    NARROW_KERNEL_PRIORITY();
    /* dynamic LDS: five words per node (+ one: s_first has nodes + 1 entries), then the candidates' prices and end nodes -- of a
     * window of at most lds_cands candidates; a larger one keeps them in price_ws (two words per candidate of the job) */
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t *s_cost = (uint32_t *)lds, *s_path = s_cost + lds_nodes, *s_via = s_path + lds_nodes, *s_used = s_via + lds_nodes;
    uint32_t *s_order = s_used + lds_nodes, *s_first = s_order + lds_nodes;
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    const SrlaWindowDesc wd = windows[w];
    const uint32_t nch = jp.num_channels, bps = jp.bits_per_sample, nodes = wd.num_nodes;
    const bool in_lds = wd.num_cands <= lds_cands;
    uint32_t *s_packed = in_lds ? (s_first + lds_nodes + 1u) : (price_ws + 2u * (size_t)wd.cand_base);
    uint32_t *s_nij = s_packed + (in_lds ? lds_cands : wd.num_cands);          /* node_i | node_j << 16 */

    for (uint32_t c = lane; c < wd.num_cands; c += WAVE) {
        const SrlaCandDesc cd = cands[wd.cand_base + c];
        const uint32_t raw_bytes = 11u + (bps * cd.n * nch) / 8u;
        uint32_t bytes = raw_bytes, type = SRLA_BLOCK_RAW, method = 0;
        if (cd.item_base != 0xFFFFFFFFu) {
            bool silent = true;
            if (cd.raw_silence != 0) silent = cd.raw_silence == 1u;
            else
            for (uint32_t ch = 0; ch < nch; ch++)
                if (!(results[cd.item_base + ch].flags & SRLA_ITEM_INPUT_ZERO)) { silent = false; break; }
            if (silent) { type = SRLA_BLOCK_SILENT; bytes = 11u; }
            else {
                uint32_t bits;
                if (nch == 1) { bits = results[cd.item_base].code_length; method = 0; }
                else {
                    const uint32_t l = results[cd.item_base + 0].code_length, r = results[cd.item_base + 1].code_length;
                    const uint32_t m = results[cd.item_base + nch].code_length, s = results[cd.item_base + nch + 1].code_length;
                    uint32_t len[4] = { l + r, m + s, l + s, r + s };
                    bits = len[0]; method = 0;
                    for (uint32_t k = 1; k < 4; k++) if (bits > len[k]) { bits = len[k]; method = k; }
                }
                bits += 2u;
                bits = ((bits + 7u) / 8u) * 8u;
                if (bits >= bps * cd.n * nch) { type = SRLA_BLOCK_RAW; bytes = raw_bytes; }
                else { type = SRLA_BLOCK_COMPRESS; bytes = 11u + bits / 8u; }
            }
        }
        s_packed[c] = bytes | (type << 28) | (method << 30);
        s_nij[c] = cd.node_i | (cd.node_j << 16);
    }
    for (uint32_t i = lane; i < nodes; i += WAVE) {
        s_cost[i] = (i == 0) ? 0u : SRLA_BIG_WEIGHT; s_path[i] = 0xFFFFFFFFu; s_via[i] = 0xFFFFFFFFu; s_used[i] = 0;
        s_first[i] = wd.num_cands;
    }
    if (lane == 0) s_first[nodes] = wd.num_cands;
    __threadfence();                 /* (the candidates' words may stand in global memory: price_ws) */
    __syncthreads();
    /* the candidates stand in the order of their start node (host_plan.cpp: i outer, j inner): s_first[i] = the first one leaving
     * node i, so that a step of the search looks at the edges of ITS node only (all of a window's candidates per step cost
     * nodes x candidates: 15 million visits for a window of 257 nodes) */
    for (uint32_t c = lane; c < wd.num_cands; c += WAVE) {
        const uint32_t ni = s_nij[c] & 0xFFFFu;
        if (c == 0 || (s_nij[c - 1] & 0xFFFFu) != ni) s_first[ni] = c;
    }
    __syncthreads();
    /* shortest path 0 -> nodes-1 over candidate edges */
    uint32_t target = 0;
    for (uint32_t guard = 0; guard <= nodes; guard++) {
        /* first unused node whose cost is below BIG and minimal (`mn > cost[i]`, ascending i) */
        uint64_t key = ~0ull;
        for (uint32_t i = lane; i < nodes; i += WAVE)
            if (!s_used[i] && s_cost[i] < SRLA_BIG_WEIGHT) { const uint64_t k = ((uint64_t)s_cost[i] << 11) | i; key = (k < key) ? k : key; }
        key = wave_min_u64(key);
        if (key != ~0ull) target = (uint32_t)(key & 0x7FFu);
        if (target == nodes - 1) break;
        const uint32_t base_cost = s_cost[target];
        /* relax every edge leaving `target`; its edges end on distinct nodes, so the lanes never collide */
        /* (every node but the last has the candidate (i, i + 1), so node i's run ends where node i + 1's starts) */
        for (uint32_t c = s_first[target] + lane; c < s_first[target + 1]; c += WAVE) {
            if ((s_nij[c] & 0xFFFFu) != target) continue;
            const uint32_t j = s_nij[c] >> 16;
            const uint32_t via = (s_packed[c] & 0x0FFFFFFFu) + base_cost;
            if (s_cost[j] > via) { s_cost[j] = via; s_path[j] = target; s_via[j] = c; }
        }
        if (lane == 0) s_used[target] = 1;
        __syncthreads();
    }
    /* read the partition back and emit block records in stream order */
    uint32_t count = 0;
    if (lane == 0) {
        for (uint32_t node = nodes - 1; node != 0 && s_path[node] != 0xFFFFFFFFu; node = s_path[node]) s_order[count++] = node;
    }
    count = __shfl(count, 0, WAVE);
    __syncthreads();
    for (uint32_t k = lane; k < nodes - 1; k += WAVE) {
        if (k >= count) { blocks[wd.block_base + k].valid = 0; continue; }
        const uint32_t node = s_order[k];
        const uint32_t c = s_via[node];
        const SrlaCandDesc cd = cands[wd.cand_base + c];
        const uint32_t packed = s_packed[c];
        uint32_t block_type = (packed >> 28) & 3u, bytes = packed & 0x0FFFFFFFu;
        const uint32_t ch_method = (packed >> 30) & 3u;
        if (block_type == SRLA_BLOCK_COMPRESS && nch > 2) {
            /* ComputeBlockSize prices only the first two channels (srla_encoder.c:1287-1301) and that
             * price drives the search; EncodeBlock then writes every channel and applies its RAW
             * fall-back to the size actually written (srla_encoder.c:1605-1611) */
            uint32_t bits = 2u;
            for (uint32_t ch = 2; ch < nch; ch++) bits += results[cd.item_base + ch].code_length;
            const uint32_t l = results[cd.item_base + 0].code_length, r = results[cd.item_base + 1].code_length;
            const uint32_t m = results[cd.item_base + nch].code_length, s2 = results[cd.item_base + nch + 1].code_length;
            bits += (ch_method == 0) ? l + r : (ch_method == 1) ? m + s2 : (ch_method == 2) ? l + s2 : r + s2;
            const uint32_t payload = (bits + 7u) / 8u;
            if (8u * payload >= bps * cd.n * nch) { block_type = SRLA_BLOCK_RAW; bytes = 11u + (bps * cd.n * nch) / 8u; }
            else bytes = 11u + payload;
        }
        SrlaBlockRecord *rec = &blocks[wd.block_base + (count - 1 - k)];
        rec->valid = 1;
        rec->sample_off = cd.sample_off;
        rec->n = cd.n;
        rec->block_type = block_type;
        rec->ch_method = ch_method;
        rec->bytes = bytes;
        for (uint32_t ch = 0; ch < SRLA_MAX_CH; ch++) {
            uint32_t it = 0xFFFFFFFFu;
            if (block_type == SRLA_BLOCK_COMPRESS && ch < nch) {
                it = cd.item_base + ch;
                if (nch >= 2) {
                    const uint32_t mi = cd.item_base + nch, si = cd.item_base + nch + 1;
                    if (ch == 0 && ch_method == 1) it = mi;
                    if (ch == 0 && ch_method == 3) it = si;
                    if (ch == 1 && (ch_method == 1 || ch_method == 2)) it = si;
                }
            }
            rec->item[ch] = it;
        }
        rec->seg = wd.seg; rec->price = packed & 0x0FFFFFFFu;
    }
}

/* ------------------------------------------------------------------------- pack ----------- */
/* srla_block_offsets (one workgroup per job): exclusive prefix sum of the chosen blocks' byte sizes in stream order;
 * per-window byte counts for the encode callback; then, per SEGMENT of the job (the run of windows that belong to one
 * stream): where its blocks go in the stream's output buffer -- at init_pos, or behind what the earlier jobs of the stream
 * wrote (a device-resident running offset per stream, so jobs are enqueued back to back without a host round trip) --, the
 * overflow check of SRLAEncoder_EncodeWhole (srla_encoder.c:1756-1783), and where the segment is assembled in the job's
 * staging buffer: with the 16-byte phase of its final address, so that srla_stream_out moves it with aligned 16-byte
 * accesses.  block_off[] ends up as offsets into the staging buffer. */
#define SRLA_SEGCTL_WORDS 8   /* device-side record per segment: bytes, pos, stage_off, skip, rel_start, - - - */
__global__ __launch_bounds__(NT) void srla_block_offsets(
    SrlaJobParams jp, const SrlaWindowDesc *__restrict__ windows, const SrlaBlockRecord *__restrict__ blocks,
    const SrlaItemResult *__restrict__ results, uint32_t num_slots, uint32_t *__restrict__ block_off,
    uint32_t *__restrict__ stream_pos /* per stream: [0] running offset, [1] sticky skip flag */,
    const SrlaSegDesc *__restrict__ segs, uint32_t *__restrict__ seg_ctl, uint64_t stage_addr,
    SrlaJobInfo *__restrict__ info, uint32_t *__restrict__ window_bytes, SrlaSegInfo *__restrict__ seg_info,
    const uint32_t *__restrict__ ties, SrlaTieGather tg)
{
    __shared__ uint32_t s_wave[NWAVES];
    __shared__ uint32_t s_cnt[6];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nch = jp.num_channels;
    if (tid < 6) s_cnt[tid] = 0;
    __syncthreads();
    uint32_t carry = 0;
    uint32_t nblk = 0, nraw = 0, nsil = 0, nodd = 0;
    for (uint32_t base = 0; base < num_slots; base += NT) {
        const uint32_t i = base + tid;
        uint32_t b = 0;
        if (i < num_slots && blocks[i].valid) {
            const SrlaBlockRecord *rec = &blocks[i];
            b = rec->bytes;
            nblk++;
            if (rec->block_type == SRLA_BLOCK_RAW) nraw++;
            else if (rec->block_type == SRLA_BLOCK_SILENT) nsil++;
            else for (uint32_t ch = 0; ch < nch; ch++) {
                const uint32_t f = results[rec->item[ch]].flags;
                if (f & SRLA_ITEM_ODD_LENGTH) nodd++;
            }
        }
        uint32_t incl = b;
        for (int off = 1; off < WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off, WAVE); if (lane >= (uint32_t)off) incl += t; }
        if (lane == WAVE - 1) s_wave[wave] = incl;
        __syncthreads();
        uint32_t pre = carry, tot = 0;
        for (uint32_t k = 0; k < NWAVES; k++) { if (k < wave) pre += s_wave[k]; tot += s_wave[k]; }
        if (i < num_slots) block_off[i] = pre + incl - b;
        carry += tot;
        __syncthreads();
    }
    const uint32_t total = carry;
    {
        const uint32_t v[4] = { nblk, nraw, nsil, nodd };
        for (int k = 0; k < 4; k++) { const uint32_t s = wave_sum_u32(v[k]); if (lane == 0 && s) atomicAdd(&s_cnt[k], s); }
    }
    __syncthreads();
    /* per-window sizes + coverage: the chosen blocks of a window must tile it exactly */
    uint32_t bad = 0;
    for (uint32_t w = tid; w < jp.num_windows; w += NT) {
        const SrlaWindowDesc wd = windows[w];
        const uint32_t b0 = wd.block_base, b1 = wd.block_base + wd.num_nodes - 1;
        const uint32_t start = block_off[b0], end = (b1 < num_slots) ? block_off[b1] : total;
        window_bytes[w] = end - start;
        uint32_t covered = 0;
        for (uint32_t k = b0; k < b1; k++) { if (!blocks[k].valid) break; covered += blocks[k].n; }
        if (covered != wd.n) bad = 1;
    }
    bad = wave_max_u32(bad);
    if (lane == 0 && bad) atomicOr(&s_cnt[4], 1u);
    __syncthreads();
    const uint32_t cover_bad = s_cnt[4];
    /* segments */
    for (uint32_t sg = tid; sg < jp.num_segs; sg += NT) {
        const SrlaSegDesc sd = segs[sg];
        const SrlaWindowDesc w0 = windows[sd.first_window], w1 = windows[sd.first_window + sd.num_windows - 1];
        const uint32_t b0 = w0.block_base, b1 = w1.block_base + w1.num_nodes - 1;
        const uint32_t rel_start = block_off[b0], rel_end = (b1 < num_slots) ? block_off[b1] : total;
        const uint32_t bytes = rel_end - rel_start;
        const uint32_t pos = sd.use_init ? sd.init_pos : stream_pos[2u * sd.stream];
        uint32_t skip = sd.use_init ? 0u : stream_pos[2u * sd.stream + 1u];
        if ((uint64_t)pos + bytes > (uint64_t)sd.limit) { skip = 1; atomicOr(&s_cnt[5], SRLA_JOBERR_OVERFLOW); }
        if (cover_bad) skip = 1;
        stream_pos[2u * sd.stream] = skip ? pos : pos + bytes;
        stream_pos[2u * sd.stream + 1u] = skip;
        /* segment k starts at rel_start + 16 k + adj (adj < 16 gives it the phase of its destination): segments never overlap */
        const uint32_t phase = sd.dst ? (uint32_t)((sd.dst + pos) & 15u) : 0u;
        const uint32_t base = rel_start + 16u * sg;
        const uint32_t stage_off = base + ((phase - (uint32_t)((stage_addr + base) & 15u)) & 15u);
        uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * sg;
        c[0] = bytes; c[1] = pos; c[2] = stage_off; c[3] = skip; c[4] = rel_start;
        seg_info[sg].bytes = bytes; seg_info[sg].pos = pos; seg_info[sg].stage_off = stage_off; seg_info[sg].skip = skip;
    }
    __syncthreads();
    __threadfence_block();
    /* block offsets: from job-relative to staging-buffer positions */
    for (uint32_t w = tid; w < jp.num_windows; w += NT) {
        const SrlaWindowDesc wd = windows[w];
        const uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * wd.seg;
        const uint32_t delta = c[2] - c[4];
        for (uint32_t k = wd.block_base; k < wd.block_base + wd.num_nodes - 1; k++) block_off[k] += delta;
    }
    if (tid == 0) {
        info->total_bytes = total;
        info->base = seg_ctl[1];
        info->num_blocks = s_cnt[0];
        info->num_raw = s_cnt[1];
        info->num_silent = s_cnt[2];
        info->num_tie_items = ties ? ties[0] : 0u;
        info->num_odd_items = s_cnt[3];
        info->error = s_cnt[5] | (cover_bad ? SRLA_JOBERR_COVER : 0u);
    }
    /* the near-ties' numbers for the host (SrlaTieGather) */
    if (tg.out != nullptr && ties != nullptr) {
        const uint32_t count = ties[0];
        if (count != 0 && count <= tg.cap) {
            const uint32_t P = jp.max_order, stride = (P + 2u > 8u) ? P + 2u : 8u;
            for (uint32_t k = 0; k < count; k++) {
                const uint32_t e = ties[1 + k], item = e & 0x3FFFFFFFu, kind = e >> 30;
                double *dst = tg.out + tg.cap + (size_t)k * stride;
                if (tid == 0) tg.out[k] = (double)e;
                if (kind == 0 && item < tg.num_items && tg.err != nullptr) {
                    for (uint32_t o = tid; o <= P; o += NT) dst[o] = tg.err[(size_t)o * tg.num_items + item];
                    if (tid == 0) dst[P + 1] = (double)results[item].lpc_order;
                } else if (kind == 1 && tg.tie_data != nullptr) {
                    if (tid < 8) dst[tid] = tg.tie_data[8 * (size_t)k + tid];
                }
            }
        }
    }
}

/* srla_pack_blocks, one workgroup per chosen block: assembles the COMPLETE block of the stream -- the
 * 11-byte block header (srla_encoder.c:1583-1595, 1629-1636), the compress payload (:1368-1452: channel
 * method, pre-emphasis state, LPC order / shift / static-Huffman coded taps, LTP fields, then per channel the
 * residual code of SRLACoder_Encode, srla_coder.c:532-595: 2-bit code type, 10-bit partition order, per
 * partition the parameter -- 5 bits, then unary zig-zag deltas -- followed by the (recursive) Rice codes) or
 * the raw payload (:823-852), and the Fletcher-16 checksum (srla_utility.c:36-60) -- MSB first in LDS (bit
 * offsets from a workgroup prefix sum over the code lengths, bits merged with LDS atomic ORs), and stores it
 * at its final byte offset of the output stream with 16-byte stores.  Blocks too large for LDS are assembled in
 * a global scratch region instead (template parameter G). */
template <bool G>
__device__ __forceinline__ void put_bits(uint32_t *w, uint32_t bitpos, uint32_t value, uint32_t nbits)
{
    /* nbits in [1,32]; word k holds stream bits 32k..32k+31, most significant first */
    if (nbits < 32) value &= (1u << nbits) - 1u;
    const uint32_t wi = bitpos >> 5, o = bitpos & 31u;
    if (o + nbits <= 32u) atomicOr(&w[wi], value << (32u - o - nbits));
    else {
        const uint32_t spill = o + nbits - 32u;
        atomicOr(&w[wi], value >> spill);
        atomicOr(&w[wi + 1], value << (32u - spill));
    }
}

template <bool G>
__device__ __forceinline__ uint32_t get_word(const uint32_t *w, uint32_t i)
{
    /* the global scratch is filled by L2 atomics: read it past the (non-coherent) vector L1 */
    if (G) return __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return w[i];
}

template <bool G>
__device__ __forceinline__ void pack_block_body(
    const SrlaJobParams &jp, const SrlaBlockRecord *__restrict__ recp, const int32_t *__restrict__ input,
    const SrlaItemDesc *__restrict__ items, const SrlaItemResult *__restrict__ results, const int32_t *__restrict__ res_ws,
    const uint32_t *__restrict__ huff_code, const uint8_t *__restrict__ huff_len, uint32_t *w, uint32_t *aux, uint32_t *kpar,
    uint8_t *__restrict__ dst, SrlaJobInfo *__restrict__ info)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    /* scalars of the record; item[] stays in memory (dynamic indexing of a by-value copy would spill) */
    const uint32_t nch = jp.num_channels, bps = jp.bits_per_sample, n = recp->n, T = recp->bytes;
    const uint32_t block_type = recp->block_type, sample_off = recp->sample_off;
    const uint32_t nwords = ((T + 3u) >> 2) + 1u;
    PHASE_INIT();
    for (uint32_t i = tid; i < nwords; i += NT) w[i] = 0;
    __syncthreads();
    PHASE(0);                                                       /* record fetched, words zeroed */
    if (tid == 0) {
        put_bits<G>(w, 0, 0xFFFFu, 16);                 /* sync code */
        put_bits<G>(w, 16, T - 11u + 5u, 32);           /* size of what follows the size field */
        put_bits<G>(w, 64, block_type, 8);
        put_bits<G>(w, 72, n, 16);
    }
    uint32_t end_bits = 88;                             /* uniform: first bit behind the payload */
    if (block_type == SRLA_BLOCK_RAW) {
        /* interleaved, zig-zag mapped, big endian */
        const uint32_t count = n * nch;
        for (uint32_t i = tid; i < count; i += NT) {
            const uint32_t s = i / nch, ch = i - s * nch;
            const int32_t v = input[(size_t)ch * jp.channel_stride + sample_off + s];
            put_bits<G>(w, 88u + i * bps, zigzag32(v), bps);
        }
        end_bits = 88u + count * bps;
    } else if (block_type == SRLA_BLOCK_COMPRESS) {
        /* header bits of the payload: every thread needs their count, wave 0 writes them */
        uint32_t hdr_bits = 2u + nch * (bps + 1u + 5u);
        for (uint32_t ch = 0; ch < nch; ch++) {
            const SrlaItemResult *ir = &results[recp->item[ch]];
            hdr_bits += 8u + 4u + 1u + ir->pad[0] + 1u;
            if (ir->ltp_period > 0) hdr_bits += 1u + 8u + jp.ltp_order * 6u;
        }
        if (wave == 0) {
            uint32_t p = 88;
            if (lane == 0) put_bits<G>(w, p, recp->ch_method, 2);
            p += 2;
            if (lane < nch) {
                const SrlaItemResult *ir = &results[recp->item[lane]];
                put_bits<G>(w, p + lane * (bps + 6u), zigzag32(ir->preemph_prev), bps + 1u);
                put_bits<G>(w, p + lane * (bps + 6u) + bps + 1u, zigzag32(ir->preemph_coef), 5);
            }
            p += nch * (bps + 6u);
            for (uint32_t ch = 0; ch < nch; ch++) {
                const SrlaItemResult *ir = &results[recp->item[ch]];
                const uint32_t order = ir->lpc_order, use_sum = ir->use_sum;
                if (lane == 0) {
                    put_bits<G>(w, p, order, 8);
                    put_bits<G>(w, p + 8, ir->lpc_rshift, 4);
                    put_bits<G>(w, p + 12, use_sum, 1);
                }
                p += 13;
                for (uint32_t b0 = 0; b0 < order; b0 += WAVE) {
                    const uint32_t i = b0 + lane;
                    uint32_t code = 0, len = 0;
                    if (i < order) {
                        const int32_t c = ir->lpc_coef[i];
                        if (!use_sum || i == 0) { const uint32_t u = zigzag32(c); code = huff_code[u]; len = huff_len[u]; }
                        else { const uint32_t u = zigzag32(c + (int32_t)ir->lpc_coef[i - 1]); code = huff_code[256 + u]; len = huff_len[256 + u]; }
                    }
                    uint32_t incl = len;
                    for (int off = 1; off < WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off, WAVE); if (lane >= (uint32_t)off) incl += t; }
                    if (len) put_bits<G>(w, p + incl - len, code, len);
                    p += __shfl(incl, WAVE - 1, WAVE);
                }
            }
            for (uint32_t ch = 0; ch < nch; ch++) {
                const SrlaItemResult *ir = &results[recp->item[ch]];
                const uint32_t period = ir->ltp_period;
                if (lane == 0) {
                    put_bits<G>(w, p, period != 0, 1);
                    if (period > 0) {
                        put_bits<G>(w, p + 1, (jp.ltp_order - 1u) / 2u, 1);
                        put_bits<G>(w, p + 2, period - SRLA_LTP_MIN_PERIOD, 8);
                        for (uint32_t i = 0; i < jp.ltp_order; i++) put_bits<G>(w, p + 10 + 6 * i, zigzag32(ir->ltp_coef[i]), 6);
                    }
                }
                p += 1u + (period > 0 ? 9u + 6u * jp.ltp_order : 0u);
            }
            if (lane == 0 && p != 88u + hdr_bits) info->error = SRLA_JOBERR_SIZE;
        }
        /* residual codes, channel after channel.  The partition parameters go to LDS first (a parameter fetched from
         * global memory at every partition boundary stalled the loops), the thread's residuals are fetched once with
         * 16-byte loads and kept in registers for both passes. */
        constexpr uint32_t CACHE = 16;                                   /* residuals a thread can keep */
        uint32_t chan_base = 88u + hdr_bits;
        PHASE(1);                                                   /* payload header (wave 0), the others wait at the first barrier below */
        for (uint32_t ch = 0; ch < nch; ch++) {
            const uint32_t item = recp->item[ch];
            const SrlaItemResult *ir = &results[item];
            const uint32_t total_bits = ir->res_bits, code_type = ir->res_code_type, porder = ir->res_porder;
            if (code_type == SRLA_CODE_ALLZERO) {
                if (tid == 0) put_bits<G>(w, chan_base, SRLA_CODE_ALLZERO, 2);
            } else {
                uint8_t *kp = reinterpret_cast<uint8_t *>(kpar) + (ch & 1u) * 1024u;
                {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(ir->kparam);
                    uint32_t *d32 = reinterpret_cast<uint32_t *>(kp);
                    for (uint32_t i = tid; i < (((1u << porder) + 3u) >> 2); i += NT) d32[i] = src[i];
                }
                /* the residual where srla_residual_cost (_big) left it */
                const int32_t *res = res_ws + items[item].res_off;
                const uint32_t plen = n >> porder;
                const uint32_t per = (n + NT - 1) / NT;                  /* contiguous samples per thread */
                const uint32_t s0 = (tid * per < n) ? tid * per : n, s1 = (s0 + per < n) ? (s0 + per) : n;
                const uint32_t part0 = (s0 < n) ? s0 / plen : 0;
                const bool cached = per <= CACHE && (per & 3u) == 0 && s0 + per <= n;   /* s0 is a multiple of 4: aligned */
                /* srla_residual_cost left the zig-zag mapped residual as uint16 where the block's values fit (see there) */
                const bool res_u16 = (ir->flags & SRLA_ITEM_RES_U16) != 0u;
                const uint16_t *res16 = reinterpret_cast<const uint16_t *>(res);
                uint32_t uc[CACHE];
                if (cached) {
#pragma unroll
                    for (uint32_t c = 0; c < CACHE / 4; c++) {
                        if (4 * c < per) {
                            if (res_u16) {
                                const uint2 q = *reinterpret_cast<const uint2 *>(res16 + s0 + 4 * c);
                                uc[4 * c] = q.x & 0xFFFFu; uc[4 * c + 1] = q.x >> 16; uc[4 * c + 2] = q.y & 0xFFFFu; uc[4 * c + 3] = q.y >> 16;
                            } else {
                                const int4 q = *reinterpret_cast<const int4 *>(res + s0 + 4 * c);
                                uc[4 * c] = zigzag32(q.x); uc[4 * c + 1] = zigzag32(q.y); uc[4 * c + 2] = zigzag32(q.z); uc[4 * c + 3] = zigzag32(q.w);
                            }
                        }
                    }
                }
                __syncthreads();                                         /* kp is complete */
                PHASE(2);                                                /* parameters to LDS, residuals fetched */
                /* pass 1: bits this thread will emit */
                uint32_t mybits = 0;
                {
                    uint32_t part = part0, next = (part0 + 1) * plen;
                    uint32_t k = kp[part];
                    auto count = [&](uint32_t s, uint32_t u) {
                        if (s == next) { part++; next += plen; k = kp[part]; }
                        if (s == part * plen) {
                            if (part == 0) mybits += 2u + 10u + 5u;
                            else mybits += zigzag32((int32_t)k - (int32_t)kp[part - 1]) + 1u;
                        }
                        if (code_type == SRLA_CODE_RICE) mybits += 1u + k + (u >> k);
                        else mybits += (k + 2u) + (__builtin_elementwise_sub_sat(u, 2u << k) >> k);
                    };
                    if (cached) {
#pragma unroll
                        for (uint32_t i = 0; i < CACHE; i++) if (i < per) count(s0 + i, uc[i]);
                    } else {
                        for (uint32_t s = s0; s < s1; s++) count(s, res_u16 ? (uint32_t)res16[s] : zigzag32(res[s]));
                    }
                }
                /* exclusive prefix sum over the workgroup */
                uint32_t incl = mybits;
                for (int off = 1; off < WAVE; off <<= 1) { const uint32_t t = __shfl_up(incl, off, WAVE); if (lane >= (uint32_t)off) incl += t; }
                __syncthreads();                                         /* aux is reused channel after channel */
                if (lane == WAVE - 1) aux[wave] = incl;
                __syncthreads();
                uint32_t pos = chan_base + incl - mybits;
                for (uint32_t k = 0; k < wave; k++) pos += aux[k];
                PHASE(3);                                                /* pass 1 + prefix sum */
                /* pass 2: emit */
                {
                    uint32_t part = part0, next = (part0 + 1) * plen;
                    uint32_t k = kp[part];
                    auto emit = [&](uint32_t s, uint32_t u) {
                        if (s == next) { part++; next += plen; k = kp[part]; }
                        if (s == part * plen) {
                            if (part == 0) {
                                put_bits<G>(w, pos, (code_type << 15) | (porder << 5) | k, 17); pos += 17;   /* 2 + 10 + 5 bits */
                            } else {
                                pos += zigzag32((int32_t)k - (int32_t)kp[part - 1]);   /* zeros */
                                put_bits<G>(w, pos, 1u, 1); pos += 1;
                            }
                        }
                        if (code_type == SRLA_CODE_RICE) {
                            pos += u >> k;                                /* quotient in unary: zeros */
                            put_bits<G>(w, pos, (1u << k) | (u & ((1u << k) - 1u)), k + 1u); pos += k + 1u;
                        } else {
                            const uint32_t k1 = k + 1u, k1pow = 1u << k1;
                            if (u < k1pow) {
                                put_bits<G>(w, pos, 1u, 1); pos += 1;                       /* (2^k1 | u) in k1 + 1 bits */
                                put_bits<G>(w, pos, u, k1); pos += k1;
                            } else {
                                const uint32_t v = u - k1pow;
                                pos += 1u + (v >> k);
                                put_bits<G>(w, pos, (1u << k) | (v & ((1u << k) - 1u)), k + 1u); pos += k + 1u;
                            }
                        }
                    };
                    if (cached) {
#pragma unroll
                        for (uint32_t i = 0; i < CACHE; i++) if (i < per) emit(s0 + i, uc[i]);
                    } else {
                        for (uint32_t s = s0; s < s1; s++) emit(s, res_u16 ? (uint32_t)res16[s] : zigzag32(res[s]));
                    }
                }
            }
            chan_base += total_bits;
            PHASE(4);                                                    /* pass 2: emit */
        }
        end_bits = chan_base;
    }
    if (tid == 0 && 11u + ((end_bits - 88u + 7u) >> 3) != T) info->error = SRLA_JOBERR_SIZE;
    __syncthreads();

    /* Fletcher-16 over bytes [8, T): the reference folds c0 += b, c1 += c0 modulo 255, i.e.
     * c0 = sum b_i, c1 = sum (L - i) b_i  (mod 255) with i counted from byte 8 and L = T - 8 */
    {
        const uint32_t L = T - 8u;
        uint64_t a = 0, ws = 0;
        for (uint32_t wi = 2u + tid; wi < nwords; wi += NT) {
            const uint32_t v = get_word<G>(w, wi);
            const uint32_t i0 = 4u * wi - 8u;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t b = (v >> (24u - 8u * k)) & 0xFFu;       /* zero beyond T */
                a += b;
                ws += (uint64_t)b * (uint64_t)(uint32_t)((L > i0 + k) ? (L - i0 - k) : 0u);
            }
        }
        uint32_t c0 = (uint32_t)(a % 255u), c1 = (uint32_t)(ws % 255u);
        c0 = wave_sum_u32(c0); c1 = wave_sum_u32(c1);
        if (lane == 0) { aux[8 + wave] = c0; aux[16 + wave] = c1; }
        __syncthreads();
        if (tid == 0) {
            uint32_t s0 = 0, s1 = 0;
            for (uint32_t k = 0; k < NWAVES; k++) { s0 += aux[8 + k]; s1 += aux[16 + k]; }
            put_bits<G>(w, 48, ((s1 % 255u) << 8) | (s0 % 255u), 16);
        }
        __syncthreads();
    }

    PHASE(5);                                                       /* Fletcher-16 */
    /* store at the block's byte offset of the stream: bytes up to the first 16-byte boundary, 16-byte body, tail */
    {
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u);
        uint32_t head = (16u - mis) & 15u;
        if (head > T) head = T;
        const uint32_t nvec = (T - head) >> 4;
        const uint32_t tail0 = head + (nvec << 4);
        auto byte_at = [&](uint32_t b) -> uint8_t { return (uint8_t)(get_word<G>(w, b >> 2) >> (24u - 8u * (b & 3u))); };
        if (tid < head) dst[tid] = byte_at(tid);
        if (tid >= 32 && tid - 32 < T - tail0) dst[tail0 + tid - 32] = byte_at(tail0 + tid - 32);
        const uint32_t r8 = 8u * (head & 3u);
        for (uint32_t v = tid; v < nvec; v += NT) {
            const uint32_t q = (head + (v << 4)) >> 2;
            uint32_t x[5];
#pragma unroll
            for (int k = 0; k < 5; k++) x[k] = get_word<G>(w, q + k);   /* q + 4 <= nwords - 1 */
            uint4 o;
            if (r8 == 0) { o.x = x[0]; o.y = x[1]; o.z = x[2]; o.w = x[3]; }
            else {
                o.x = (x[0] << r8) | (x[1] >> (32u - r8)); o.y = (x[1] << r8) | (x[2] >> (32u - r8));
                o.z = (x[2] << r8) | (x[3] >> (32u - r8)); o.w = (x[3] << r8) | (x[4] >> (32u - r8));
            }
            o.x = __builtin_bswap32(o.x); o.y = __builtin_bswap32(o.y); o.z = __builtin_bswap32(o.z); o.w = __builtin_bswap32(o.w);
            *reinterpret_cast<uint4 *>(dst + head + (v << 4)) = o;
        }
    }
    PHASE(6);                                                       /* store */
}

__global__ __launch_bounds__(NT) void srla_pack_blocks(
    SrlaJobParams jp, const int32_t *__restrict__ input, const SrlaItemDesc *__restrict__ items,
    const SrlaBlockRecord *__restrict__ blocks, const SrlaItemResult *__restrict__ results,
    const int32_t *__restrict__ res_ws, const uint32_t *__restrict__ huff_code, const uint8_t *__restrict__ huff_len,
    const uint32_t *__restrict__ block_off, const uint32_t *__restrict__ seg_ctl, uint8_t *__restrict__ out,
    uint8_t *__restrict__ scratch, SrlaJobInfo *__restrict__ info, uint32_t lds_words,
    const SrlaSegDesc *__restrict__ segs /* non-null: store the block where the stream wants it (below) */, uint8_t *__restrict__ host_stage)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint32_t *aux = (uint32_t *)lds;                         /* 32 words: wave sums */
    uint32_t *kpar = aux + 32;                               /* 2 x 1024 bytes: partition parameters of the channel in work */
    uint32_t *words = kpar + 512;                            /* lds_words entries */
    const uint32_t slot = blockIdx.x;
    const SrlaBlockRecord *recp = &blocks[slot];
    if (!recp->valid || seg_ctl[(size_t)SRLA_SEGCTL_WORDS * recp->seg + 3u]) return;
    uint8_t *dst = out + block_off[slot];
    if (segs != nullptr) {
        /* The LAST job of a call (round 6): nothing runs behind it that its workgroups could slow down by staying resident for the
         * link's stores, so every block goes straight to its place in host memory -- its segment's destination (the stream's buffer
         * where the device reaches it, else the job's pinned staging buffer) plus its distance from the segment's start in the
         * staging layout, which has the destination's 16-byte phase -- and srla_stream_out with its launch boundary (14 us of a
         * 10 s stream's 0.29 ms) is not launched. */
        const uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * recp->seg;
        uint8_t *seg_dst = segs[recp->seg].dst ? reinterpret_cast<uint8_t *>(segs[recp->seg].dst) + c[1] : host_stage + c[2];
        dst = seg_dst + (block_off[slot] - c[2]);
    }
    const uint32_t nwords = ((recp->bytes + 3u) >> 2) + 1u;
    if (nwords <= lds_words) {
        pack_block_body<false>(jp, recp, input, items, results, res_ws, huff_code, huff_len, words, aux, kpar, dst, info);
    } else {
        const size_t off = ((size_t)recp->sample_off * jp.num_channels * (jp.bits_per_sample >> 3) + (size_t)slot * SRLA_PACK_SLACK + 3u) & ~(size_t)3u;
        pack_block_body<true>(jp, recp, input, items, results, res_ws, huff_code, huff_len, (uint32_t *)(scratch + off), aux, kpar, dst, info);
    }
}

/* srla_stream_out: the job's finished bytes, device buffer -> their place in host memory (the stream's pinned buffer, or
 * the job's pinned staging buffer), segment by segment.  A handful of workgroups keep the PCIe link busy; letting the
 * 1000+ pack workgroups store to host memory themselves kept them (and their LDS) resident for the duration of the link
 * transfer and slowed the concurrently running srla_autocorr by 30 %. */
__device__ __forceinline__ void copy_same_phase(const uint8_t *__restrict__ src, uint8_t *__restrict__ d, uint32_t total,
                                                uint32_t gtid, uint32_t gsize, uint32_t pause)
{
    uint32_t head = (16u - (uint32_t)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u;    /* src has the same phase */
    if (head > total) head = total;
    const uint32_t nvec = (total - head) >> 4, tail0 = head + (nvec << 4);
    if (gtid < head) d[gtid] = src[gtid];
    if (gtid >= 32 && gtid - 32 < total - tail0) d[tail0 + gtid - 32] = src[tail0 + gtid - 32];
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + head);
    uint4 *d4 = reinterpret_cast<uint4 *>(d + head);
    uint32_t v = gtid;
    for (; v + 3 * gsize < nvec; v += 4 * gsize) {
        const uint4 a = s4[v], b = s4[v + gsize], c = s4[v + 2 * gsize], e = s4[v + 3 * gsize];
        d4[v] = a; d4[v + gsize] = b; d4[v + 2 * gsize] = c; d4[v + 3 * gsize] = e;
        for (uint32_t z = 0; z < pause; z++) __builtin_amdgcn_s_sleep(8);
    }
    for (; v < nvec; v += gsize) d4[v] = s4[v];
}

__global__ __launch_bounds__(NT) void srla_stream_out(const uint8_t *__restrict__ stage, const uint32_t *__restrict__ seg_ctl,
                                                      const SrlaSegDesc *__restrict__ segs, uint32_t num_segs,
                                                      uint8_t *__restrict__ host_stage, uint32_t pause)
{
    if (num_segs == 1) {
        /* the usual case: every workgroup of the launch works on the one segment */
        if (seg_ctl[3]) return;
        uint8_t *d = segs[0].dst ? reinterpret_cast<uint8_t *>(segs[0].dst) + seg_ctl[1] : host_stage + seg_ctl[2];
        copy_same_phase(stage + seg_ctl[2], d, seg_ctl[0], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, pause);
        return;
    }
    for (uint32_t sg = blockIdx.x; sg < num_segs; sg += gridDim.x) {
        const uint32_t *c = seg_ctl + (size_t)SRLA_SEGCTL_WORDS * sg;
        if (c[3]) continue;
        uint8_t *d = segs[sg].dst ? reinterpret_cast<uint8_t *>(segs[sg].dst) + c[1] : host_stage + c[2];
        copy_same_phase(stage + c[2], d, c[0], threadIdx.x, blockDim.x, pause);
    }
}

/* --------------------------------------------------------------------------- launchers ---- */
extern "C" uint32_t srla_price_lds_cands(void) { return SRLA_PRICE_LDS_CANDS; }

extern "C" int srla_launch_price(hipStream_t stream, const SrlaJobParams *jp, const SrlaWindowDesc *windows,
                                 const SrlaCandDesc *cands, const SrlaItemResult *results,
                                 SrlaBlockRecord *blocks, hipEvent_t ev_start, hipEvent_t ev_stop,
                                 uint32_t max_nodes, uint32_t max_window_cands, uint32_t *price_ws)
{
    if (jp->num_windows == 0) return 0;
    /* LDS for the largest window of the job: its nodes, and its candidates where they fit (else price_ws, which the caller sized
     * for two words per candidate of the job) */
    const uint32_t lds_nodes = max_nodes, lds_cands = (max_window_cands <= SRLA_PRICE_LDS_CANDS) ? max_window_cands : 0u;
    if (max_window_cands > SRLA_PRICE_LDS_CANDS && price_ws == nullptr) return -1;
    const uint32_t lds = (6u * lds_nodes + 1u + 2u * lds_cands) * 4u + 16u;
    SET_LDS_ATTR(srla_price_windows);
    hipExtLaunchKernelGGL(srla_price_windows, dim3(jp->num_windows), dim3(WAVE), lds, stream, ev_start, ev_stop, 0,
                       *jp, windows, cands, results, blocks, lds_nodes, lds_cands, price_ws);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

/* LDS words the pack kernel gets for one block; larger blocks (see srla_pack_needs_scratch) are assembled in a
 * global scratch region.  SRLA_MI355X_PACK_LDS_WORDS lowers the cap (tests use it to reach the global path). */
static uint32_t pack_lds_cap()
{
    const uint32_t cap = 24 * 1024;                         /* <= 96 KB */
    return (g_srla_tune.pack_lds_cap_words >= 8u && g_srla_tune.pack_lds_cap_words < cap) ? g_srla_tune.pack_lds_cap_words : cap;
}

extern "C" uint32_t srla_pack_lds_words(const SrlaJobParams *jp)
{
    const uint64_t bytes = 11ull + ((uint64_t)jp->bits_per_sample * jp->max_block * jp->num_channels) / 8;
    return (uint32_t)std::min<uint64_t>((bytes + 3) / 4 + 1, pack_lds_cap());
}

extern "C" int srla_pack_needs_scratch(const SrlaJobParams *jp)
{
    const uint64_t bytes = 11ull + ((uint64_t)jp->bits_per_sample * jp->max_block * jp->num_channels) / 8;
    return ((bytes + 3) / 4 + 1) > pack_lds_cap();
}

extern "C" int srla_launch_pack(hipStream_t stream, const SrlaJobParams *jp, uint32_t num_slots,
                                const int32_t *input, const SrlaItemDesc *items, const SrlaWindowDesc *windows,
                                const SrlaBlockRecord *blocks, const SrlaItemResult *results, const int32_t *res_ws,
                                const uint32_t *huff_code, const uint8_t *huff_len, uint32_t *block_off,
                                uint32_t *stream_pos, const SrlaSegDesc *segs, uint32_t *seg_ctl,
                                uint8_t *stage, uint8_t *host_stage, uint8_t *scratch, SrlaJobInfo *info,
                                uint32_t *window_bytes, SrlaSegInfo *seg_info, const uint32_t *ties,
                                hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t out_boost, hipStream_t out_stream, hipEvent_t ev_packed,
                                uint32_t no_stream_out, const SrlaTieGather *gather)
{
    if (num_slots == 0) return 0;
    SrlaTieGather tg{};
    if (gather != nullptr) tg = *gather;
    hipExtLaunchKernelGGL(srla_block_offsets, dim3(1), dim3(NT), 0, stream, ev_start, nullptr, 0, *jp, windows, blocks, results, num_slots, block_off,
                       stream_pos, segs, seg_ctl, (uint64_t)reinterpret_cast<uintptr_t>(stage), info, window_bytes, seg_info, ties, tg);
    const uint32_t lds_words = srla_pack_lds_words(jp);
    const uint32_t lds = (lds_words + 32 + 512) * 4;
    SET_LDS_ATTR(srla_pack_blocks);
    const bool direct = no_stream_out == 2u;      /* 2: no copy at all -- the assembly stores every block where the stream wants it */
    hipExtLaunchKernelGGL(srla_pack_blocks, dim3(num_slots), dim3(NT), lds, stream, nullptr, no_stream_out ? ev_stop : nullptr, 0,
                          *jp, input, items, blocks, results, res_ws, huff_code, huff_len, block_off, seg_ctl, stage, scratch, info,
                          lds_words, direct ? segs : nullptr, host_stage);
    if (no_stream_out) return (hipGetLastError() == hipSuccess) ? 0 : -2;
    /* Two workgroups (three for 24-bit streams, which carry more bytes).  One moves a 4 M-sample job's 6.5 MB in 0.45-0.57 ms
     * beside the other kernels -- longer than the job's wide kernels take (0.46 ms), so the block assembly stream, not stream W,
     * set the pace of a long stream (kernel trace of round 3); two take 0.25 ms.  More slow srla_autocorr down through the
     * PCIe write path's back-pressure (4: 0.25 -> 0.30 ms per job) and lose more than they gain.  A stream of its own for this
     * kernel (so that it runs beside the next job's assembly) was measured again and is worse by 13 %: a fifth compute
     * queue serialises with the others. */
    const uint32_t wgs = jp->bits_per_sample > 16 ? 3u : 2u;
    hipStream_t os = stream;
    if (out_stream != nullptr && ev_packed != nullptr) {
        if (hipEventRecord(ev_packed, stream) != hipSuccess || hipStreamWaitEvent(out_stream, ev_packed, 0) != hipSuccess) return -2;
        os = out_stream;
    }
    hipExtLaunchKernelGGL(srla_stream_out, dim3(wgs * (out_boost ? out_boost : 1u)), dim3(NT), 0, os, nullptr, ev_stop, 0,
                          stage, seg_ctl, segs, jp->num_segs, host_stage, 0u);
    return (hipGetLastError() == hipSuccess) ? 0 : -2;
}

