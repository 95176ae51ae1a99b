// k_png_parse.hip -- row P4 of SURVEY.md 8a, the near-optimal half: the min-cost-path parse (png_parse.h; statement: oracle/png_oracle.c deep_parse())
// over the chunks k_png_hist marked (PngChunk::deep: enough matches in the greedy parse for a choice between them to matter).
//   k_png_deep_hist  replaces the greedy counts of every marked (trial, chunk) by the final parse's -- sizes decide the winning trial;
//   k_png_deep_emit  runs the parse again over the winner's marked chunks and packs the bits (the parse is not stored: 4 bytes per position and trial).
// Both are launched with as many workgroups (one wave each) as the device holds at their LDS footprint; a workgroup owns 512 KiB of scratch in HBM
// (candidates, choices, costs: png_parse.h) and takes the marked items it finds at its stride.
#include "png_emit.h"
#include "png_parse.h"

namespace csp {

// the next item of a launch-wide queue (one counter, zeroed before the launch), the same in every lane of the workgroup
__device__ __forceinline__ static uint32_t next_item(uint32_t *counter, uint32_t *slot) {
    LFOR(l) if (CSP_WAVE0 && l == 0) *slot = atomicAdd(counter, 1u);
    CSP_WG_SYNC();
    const uint32_t v = *slot;
    CSP_WG_SYNC();
    return v;
}

__global__ void __launch_bounds__(CSP_DEEP_THREADS) k_png_deep_hist(DeflateCtx c) {
    CSH_SHARED DeepLds S;
    uint8_t *scratch = c.deep_scratch + uint64_t(blockIdx.x) * CSP_DEEP_SCRATCH;
    const uint32_t nitems = c.total_chunks * uint32_t(c.plan.ntrials);
    for (;;) {
        const uint32_t item = next_item(&c.deep_queue[0], &S.item);   // a queue, not a stride: the marked items cluster, and the workgroups past the device's residency start late
        if (item >= nitems) break;
        const uint32_t trial = item / c.total_chunks, bc = item % c.total_chunks;
        const uint32_t image = c.chunk_image[bc];
        if (c.status[image]) continue;
        const PngImg &im = c.imgs[image];
        const uint32_t ci = bc - c.chunk_first[image];
        const int slot = c.plan.trial_slot[trial];
        PngChunk &rec = chunk_rec(c, im, slot, ci);
        if (!rec.deep) continue;
        if (!c.trial_live[uint64_t(image) * CSP_MAX_STREAMS + trial]) {   // a trial too far behind to win keeps its greedy parse
            LFOR(l) if (CSP_WAVE0 && l == 0) rec.deep = 0;
            continue;
        }
        const uint8_t *data = c.streams + im.stream_off + uint64_t(slot) * im.stream_stride;
        const uint64_t start = uint64_t(ci) * CSP_CHUNK, end = start + CSP_CHUNK < im.raw_len ? start + CSP_CHUNK : im.raw_len;
        NoSink none;
        deep_chunk(data, im.raw_len, start, end, S, scratch, c.deep_iters, false, none);
        if (CSP_WAVE0) {
            // the parse's counts replace the greedy ones only where they promise a smaller block (segment ends cut long runs: a flat chunk can lose)
            const uint64_t est_greedy = deep_estimate([&](uint32_t i) { return rec.freq[i]; }), est_deep = deep_estimate([&](uint32_t i) { return i == 256 ? 1u : S.hist[i]; });
            if (est_deep < est_greedy) {
                LV<uint64_t> e;
                LFOR(l) {
                    e[l] = 0;
                    for (uint32_t i = uint32_t(l); i < CSP_NSYM; i += 64) {
                        const uint32_t f = i == 256 ? 1u : S.hist[i];
                        rec.freq[i] = f;
                        if (i > 256 && i < CSP_NLIT) e[l] += uint64_t(f) * len_extra_of(i - 257);
                        if (i >= CSP_NLIT) e[l] += uint64_t(f) * dist_extra_of(i - CSP_NLIT);
                    }
                }
                const uint64_t extra = lsum(e);
                LFOR(l) if (l == 0) { rec.extra_bits = uint32_t(extra); c.deep_list[atomicAdd(&c.deep_queue[2], 1u)] = item; }   // (its codes are made again: k_png_codes over this list)
            } else
                LFOR(l) if (l == 0) rec.deep = 0;
        }
        CSP_WG_SYNC();
    }
}

struct DeepEmitLds { DeepLds deep; uint32_t code[CSP_NSYM]; uint32_t win[160]; };
__global__ void __launch_bounds__(CSP_DEEP_THREADS) k_png_deep_emit(DeflateCtx c) {
    CSH_SHARED DeepEmitLds S;
    uint8_t *scratch = c.deep_scratch + uint64_t(blockIdx.x) * CSP_DEEP_SCRATCH;
    for (;;) {
        const uint32_t bc = next_item(&c.deep_queue[1], &S.deep.item);
        if (bc >= c.total_chunks) break;
        const uint32_t image = c.chunk_image[bc];
        if (c.status[image]) continue;
        const PngImg &im = c.imgs[image];
        const uint32_t ci = bc - c.chunk_first[image];
        const int slot = c.plan.trial_slot[c.winner[image]];
        const PngChunk &rec = chunk_rec(c, im, slot, ci);
        if (!rec.deep) continue;
        const uint8_t *data = c.streams + im.stream_off + uint64_t(slot) * im.stream_stride;
        // where the chunk goes: after the zlib header and the chunks in front of it
        uint64_t at = uint64_t(im.prefix_len) + 8 + 2;
        {
            LV<uint64_t> part;
            LFOR(l) { uint64_t s = 0; for (uint32_t k = uint32_t(l); k < ci; k += 64) s += chunk_rec(c, im, slot, k).bytes; part[l] = s; }
            at += lsum(part);
        }
        const bool last = ci + 1 == im.nchunks;
        BitOut bo;
        if (CSP_WAVE0) emit_block_begin(rec, last, S.code, S.win, c.out + im.out_off + at, bo);
        EmitSink sink; sink.code = S.code; sink.bo = &bo;
        const uint64_t start = uint64_t(ci) * CSP_CHUNK, end = start + CSP_CHUNK < im.raw_len ? start + CSP_CHUNK : im.raw_len;
        deep_chunk(data, im.raw_len, start, end, S.deep, scratch, c.deep_iters, true, sink);   // (the sink sees the tokens on the first wave)
        if (CSP_WAVE0 && !emit_block_end(rec, last, S.code, bo)) LFOR(l) if (l == 0) c.status[image] = CSP_ERR_POOL;   // the size pass and this pass disagree: never ship it
        CSP_WG_SYNC();
    }
}

// behind the greedy parse's hist / codes / choose: the parse over the live trials' marked chunks, their codes again, the winner again
void launch_png_deep(hipStream_t st, const DeflateCtx &c) {
    if (c.deep_iters <= 0 || !c.deep_slots) return;
    (void)hipMemsetAsync(c.deep_queue, 0, 4 * sizeof(uint32_t), st);
    CSH_LAUNCH(k_png_deep_hist, dim3(c.deep_slots), dim3(CSP_DEEP_THREADS), st, c);
    launch_png_codes(st, c, 1);
    launch_png_choose(st, c);
}
void launch_png_deep_emit(hipStream_t st, const DeflateCtx &c) { if (c.deep_iters > 0 && c.deep_slots) CSH_LAUNCH(k_png_deep_emit, dim3(c.deep_slots), dim3(CSP_DEEP_THREADS), st, c); }

}  // namespace csp
