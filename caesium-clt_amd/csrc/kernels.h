// kernels.h -- host-callable launchers of the device kernels (defined in k_*.hip).
#pragma once
#include "gpu_rt.h"
#include "types.h"

namespace csh {

// coefficient tile addressing (types.h): element (block b, zig-zag k) of a component
__host__ __device__ static inline int coef_off(int k) { return (k >> 3) * CSH_OCT_STRIDE + (k & 7); }  // relative to the block's base
__host__ __device__ static inline size_t coef_index(uint32_t tile_base, int b, int k) {
    return (size_t(tile_base) + size_t(b >> 6)) * CSH_TILE_I16 + size_t((b & 63) * CSH_BLK_STRIDE) + size_t(coef_off(k));
}

// the retained unquantised DCT (size targeting, the trellis quantiser) is BLOCK-major -- int16 raw[block][64 k]: one block = one 128-byte line.
// Its readers take whole blocks, and the trellis takes them in order of list length (k_trellis.hip): with the coefficient tiles' octet-major
// layout a block is eight 16-byte pieces in eight lines, and such a gather moved eight lines per block.
#define CSH_RAW_OCT 8   // int16 elements between a block's octets
__host__ __device__ static inline size_t raw_index(uint32_t tile_base, int b) { return size_t(tile_base) * CSH_TILE_I16 + size_t(b) * 64; }

// ---- phase 0: entropy decode (k_decode.hip)
void launch_decode_seq(hipStream_t st, const uint8_t *bits, ImgDesc *imgs, const DecScan *scans, const DevHuffSet *huffs,
                       int16_t *coef, int nimg, const uint32_t *need_seq);

// parallel self-synchronising decoder (k_decode_par.hip); need_seq[image] != 0 -> the sequential kernel (re)does it
void launch_unstuff_count(hipStream_t st, const uint8_t *raw, const ParScan *ps, int nps, uint32_t nchunks, uint32_t *cnt);
void launch_unstuff_copy(hipStream_t st, const uint8_t *raw, uint8_t *clean, ParScan *ps, int nps, uint32_t nchunks, const uint64_t *off);
struct DenseArgs {
    const uint8_t *clean; const ParScan *pss; const void *huffs; int compact;   // huffs: ParHuffSet4[] if compact, else ParHuffSet[]
    uint64_t *state; uint32_t *nblk; uint64_t *list_out; uint32_t *cnt_out;           // relax
    uint16_t *hyp; const uint32_t *scan_pending;                                       // label hypotheses (mode 3)
    const uint64_t *blk_off; const ImgDesc *imgs; int16_t *coef; int32_t *dcdiff; uint32_t *need_seq;  // write
    uint32_t *cut_block;   // per segment: first block that stays zero because the data ran out (0xFFFFFFFF: none), write pass -> k_dc_scatter
    uint8_t *zero_ptr; uint64_t zero_bytes;   // the store-less passes (modes 0, 1): a region their workgroups clear between them -- the coefficient tiles, which the write pass needs at zero (a multiple of 16 bytes)
    uint8_t *zero2_ptr; uint64_t zero2_bytes; // a second one: the encoder's EOBRUN array (4 GB per 2048 files, cleared for every run)
};
void launch_dec_dense(hipStream_t st, int mode /*0 speculate, 1 relax, 2 write*/, int nps, uint32_t max_sub, const DenseArgs &a);
void launch_dec_relax_list(hipStream_t st, const uint8_t *clean, const ParScan *ps, uint32_t total_sub, const void *huffs, int compact, uint64_t *state, uint32_t *nblk,
                           const uint64_t *list_in, const uint32_t *cnt_in, uint64_t *list_out, uint32_t *cnt_out, size_t nstate, uint32_t *claim, uint32_t epoch);
// progressive inputs: one wave per chain of scans (k_decode_prog.hip); images with need_seq == 4
// AC refinement chains of progressive inputs (k_decode_refine.hip): history masks, the serial parse (one wave per chain), the parallel apply
void launch_refine_chains(hipStream_t st, const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                          const int *chain_scans, int nchains, const RefineUnit *units, int nunits, uint32_t max_blocks, const ImgDesc *imgs, int16_t *coef,
                          uint32_t *need_seq, uint64_t *hist, uint32_t *posv, uint32_t *prog);
// per (work item, 256-unit chunk): SlotRec, slot -> work item, and the slot's entry in the list-coded / token-coded slot lists (k_aclist.hip)
void launch_rebind_slots(hipStream_t st, const ScanWork *works, uint32_t nworks, const NzList *nzlists, SlotRec *slots);   // after the host re-points work items' lists
void launch_make_slots(hipStream_t st, const ScanWork *works, uint32_t nworks, const EncScan *script, const NzList *nzlists, SlotRec *slots, uint32_t *slot_work, uint32_t *list_slots,
                       uint32_t *tok_slots);
void launch_decode_prog(hipStream_t st, const uint8_t *clean, const ParScan *pss, const ParHuffSet *huffs, const DecScan *scans, const ProgChain *chains,
                        const int *chain_scans, int nchains, const ImgDesc *imgs, int16_t *coef, uint32_t *need_seq);
void launch_dec_mark_pending(hipStream_t st, const ParScan *ps, uint32_t total_sub, const uint64_t *list_in, const uint32_t *cnt_in, uint32_t *scan_pending);
void launch_dec_chain(hipStream_t st, const ParScan *ps, int nps, uint64_t *state, uint32_t *nblk, const uint16_t *hyp, const uint32_t *scan_pending, uint32_t *need_seq);
void launch_dc_refine(hipStream_t st, const uint8_t *clean, const ParScan *ps, int nps, uint32_t max_blocks, const ImgDesc *imgs, int16_t *coef, const uint32_t *need_seq);   // after launch_dc_scatter
void launch_dc_scatter(hipStream_t st, const ParScan *ps, int nps, uint32_t max_blocks, const ImgDesc *imgs, const uint64_t *dc_off, int16_t *coef,
                       const uint32_t *need_seq, const uint32_t *cut_block);

// ---- phase 1: pixel-domain transcode (k_pixel.hip)
// direct: dequant -> jidctint -> range limit -> jfdctint -> quantise, one block per lane
// dct_raw != nullptr: also keep the unquantised DCT (tile index relative to raw_tile0) for launch_requant
// dering: mozjpeg's overshoot deringing on the level-shifted samples in front of every forward DCT (CSH_PROFILE=mozjpeg)
void launch_xform_direct(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                         const int16_t *coef_in, int16_t *coef_out, int16_t *dct_raw, uint32_t raw_tile0, bool dering);
void launch_requant(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant, const int16_t *dct_raw,
                    uint32_t raw_tile0, int16_t *coef_out);
// subsampled components: IDCT to a u8 plane (edges replicated), then resample + FDCT + quantise
void launch_idct_plane(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                       const int16_t *coef_in, uint8_t *planes);
void launch_resample_plane(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, uint32_t max_quads, const uint8_t *planes, uint8_t *oplanes);
void launch_plane_fdct(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                       const uint8_t *oplanes, int16_t *coef_out, int16_t *dct_raw, uint32_t raw_tile0, bool dering);
void launch_resample_fdct_420(hipStream_t st, const ImgDesc *imgs, const PlaneWork *work, int nwork, int max_tiles, const DevQuant *quant,
                              const uint8_t *planes, int16_t *coef_out, int16_t *dct_raw, uint32_t raw_tile0, bool dering);
void launch_fix_dummy(hipStream_t st, const ImgDesc *imgs, int nimg, int max_blocks, int16_t *coef_out);

// resize branch (k_resize.hip): decoded planes -> RGB -> Lanczos3 (f32, image-rs order) -> full-resolution YCbCr planes
bool resize_is_fused(uint32_t max_row_in);   // both Lanczos passes in one kernel, no f32 intermediate image (the batch's widest source row fits LDS)
void launch_resize(hipStream_t st, const ImgDesc *imgs, const ResizeWork *work, int nwork, const ResizeTap *taps, const float *weights,
                   uint8_t *planes, uint8_t *rgb, float *tmp, uint32_t max_src_px, uint64_t max_tmp, uint64_t max_dst, uint32_t max_row_in, uint32_t max_out_w, uint32_t max_nh, bool to_planes);

// ---- phases 2-5: entropy encode (k_entropy.hip)
// tokens -> runs -> tables -> chunk sizes -> (scan) -> pack.  A token is one u32 (k_entropy.hip); the tokens of one (scan, 256-unit
// chunk) are contiguous in the token pool, unit after unit.
struct EncCtx {  // device pointers + sizes every entropy kernel needs
    const ImgDesc *imgs;
    const EncScan *script;     // output scripts
    const ScanWork *work;      // [nwork]
    int nwork;
    const EChunk *echunks;     // the token kernel's grid
    uint32_t nechunks;
    const TokPlan *plans;      // what its kind-0 chunks need, per (image, component)
    const uint32_t *slot_work; // per slot (work item, 256-unit chunk): its work item -- the grid of the per-scan kernels
    const SlotRec *slots;      // the same slots, with what the packer kernels need of them
    uint32_t slot0, nslots;    // the slots this run of the kernels covers: [slot0, slot0 + nslots)
    const int16_t *coef;       // coefficients to code (tiles)
    uint64_t *sym_bits;        // per AC scan: bit b set iff block b emits >=1 Huffman symbol in this scan
    uint64_t *eob_bits;        // per AC scan: bit b set iff block b ends with a pending EOB
    uint8_t *tail;             // per unit: # of correction bits left over at the end of the block (refine scans)
    uint16_t *eobrun;          // per unit: EOBRUN value this block must emit at its EOB token (0 = none)
    uint32_t *long_runs, *long_cnt;   // runs longer than 512 blocks: (work item, first block) pairs for k_ac_runs_long
    uint64_t *corr;            // per unit (refinement scans): its correction bits in stream order, left-aligned
    uint32_t *tokens;          // token pool
    const TokRegion *regions;  // its regions (EChunk::region)
    uint32_t *tok_cursor;      // per region: tokens handed out
    uint64_t *tok_off;         // per segment (slot x 4 + wave of k_tokens): first token
    uint32_t *chunk_ntok;      // per segment: number of tokens
    uint16_t *slot_hist;       // per slot: ntables rows of 256 symbol counts (EOBRUN symbols excluded)
    uint32_t *slot_raw;        // per slot: raw bits (EOBRUN bits excluded)
    uint32_t *slot_eobh;       // per slot: [16] EOBn symbols owned by its units
    uint32_t *chunk_bits;      // per slot: size in bits of everything the chunk emits
    const uint64_t *chunk_off; // exclusive scan of chunk_bits over the whole batch
    DevEncTable *tables;       // [ntables]
    uint32_t *raw;             // unstuffed scan bytes as big-endian-logical u32 words, zero-initialised
    uint64_t raw_words;        // capacity of raw in u32 words
    uint32_t *status;          // per image
    uint32_t *overflow;        // [4]: [1] = the token pool was too small
    uint32_t debug;            // CSH_DEBUG: performance experiments (parts of k_tokens switched off; output is then garbage)
    uint32_t stats_only;       // the trellis stage's statistics scans: histograms, flags and EOB runs only -- no token is written
    const uint8_t *work_active;// null: every work item of the run is coded; else per work item 1 = coded, 0 = skipped (conditional stages of the scan search)
    // the compacted coefficient lists and the slots coded from them (k_aclist.hip, types.h NzList)
    const NzList *nzlists;
    const NzSet *nzsets;
    const NzChunk *nzchunks;   // the builder's grid for this run
    uint32_t nnzchunks;
    uint32_t *nz_pool;         // the entries
    uint32_t *nz_cursor;       // per list: entries handed out
    uint32_t *nz_chunk_off;    // per (list, chunk): first entry, relative to the list's region
    uint32_t *nz_chunk_cnt;    // per (list, chunk): entries (END entries included, padding excluded)
    uint8_t *nz_blk_cnt;       // null, or (the trellis stage's statistics lists) per block: its non-zero AC coefficients (NzSet::cnt_base)
    uint16_t *nz_blk_off;      // null, or (the same lists) per block: where its first entry stands in its chunk -- k_trellis_ac writes the levels it chose into the entries
    const uint32_t *list_slots;// the run's slots that are coded from a list (k_list_stats, k_list_pack) ...
    uint32_t nlist_slots;
    const uint32_t *tok_slots; // ... and those packed from tokens (k_pack); null: every slot of [slot0, slot0 + nslots)
    uint32_t ntok_slots;
};
void launch_tokens(hipStream_t st, const EncCtx &c);
void launch_ac_runs(hipStream_t st, const EncCtx &c);
void launch_gen_tables(hipStream_t st, DevEncTable *tables, int ntables);
void launch_chunk_sizes(hipStream_t st, const EncCtx &c);
void launch_zero_edges(hipStream_t st, const EncCtx &c);   // between the scans' placement and the pack
void launch_pack(hipStream_t st, const EncCtx &c);
// progressive AC scans from the compacted lists (k_aclist.hip): build the lists of the run's NzChunks; statistics (histograms, raw-bit
// counts, has-symbol / ends-with-EOB flags) of the run's list slots -- in launch_tokens' place --; their bits -- in launch_pack's place
void launch_nzlist(hipStream_t st, const EncCtx &c);
void launch_list_stats(hipStream_t st, const EncCtx &c);
void launch_list_pack(hipStream_t st, const EncCtx &c);
// marks every work item "in no file, no bits" at the start of a run (a conditional stage of the scan search that does not run this time
// must not leave the placement of an earlier run behind)
void launch_reset_works(hipStream_t st, ScanWork *work, int nwork);

// ---- mozjpeg's trellis quantiser (k_trellis.hip): re-quantise every block from the retained DCT with the statistics pass's code
// lengths as rates -- AC coefficients per block (k_trellis_ac), DC coefficients along each row of blocks (k_trellis_dc)
#ifndef CSH_TR_WG
#define CSH_TR_WG 256   // blocks per workgroup (and per TrellisChunk) of k_trellis_ac (64 -- no wave waits for another's densest block, but the tables are staged four times as often -- measured 6 % slower)
#endif
struct TrellisCtx {
    const ImgDesc *imgs;
    const DevQuant *quant;
    const TrellisWork *work;
    int nwork;
    const TrellisChunk *chunks;
    uint32_t nchunks;
    const DevEncTable *tables;
    const int16_t *raw;        // unquantised jfdctint output, tiles (index relative to raw_tile0)
    uint32_t raw_tile0;
    int16_t *coef;             // re-quantised coefficients (tiles)
    uint64_t *dcrec;           // per real block: lambda (float bits) << 32 | the unquantised DC & 0xFFFF (AC kernel -> DC kernel)
    uint64_t *dcbt;            // per real block: back-pointers of the DC path (9 x 4 bits) | rounded DC level << 36 | sign << 47
    uint32_t *spill;           // per workgroup of the AC kernel: entries of the block lists that do not fit LDS
    uint32_t max_rows;         // DC kernel: most iMCU rows of a component
    const uint8_t *blk_cnt;    // per real block (unit_base + u): non-zero scalar levels, the length of its list (the builder of the statistics lists counted them); null: no sorting
    uint32_t *perm;            // per work item, unit_base + slot -> block: the blocks in order of list length (k_trellis_sort), so that the 256 blocks of a chunk -- the 64 of a wave -- run equally long programmes
    const uint32_t *rows;      // DC kernel: every (work item << 16 | iMCU row) of the batch, longest rows first: one lane each
    uint32_t nrows;
    uint32_t debug;            // CSH_TR_DEBUG: timing experiments (parts of k_trellis_ac switched off; the output is then garbage)
    // progressive output: the statistics scan's level-0 NzList holds exactly the coefficients the trellis decides about, block by block in the
    // programme's own order -- k_trellis_ac writes its levels into those entries (a dropped coefficient: magnitude 0) and the coding stages'
    // lists are FILTERED from that list (k_nzfilter, NZ_COMPACT0) instead of being built from the coefficient tiles a second time.  null: no lists
    uint32_t *nz_pool;
    const NzList *nzlists;
    const NzSet *nzsets;
    const uint32_t *nz_chunk_off, *nz_chunk_cnt;
    const uint16_t *blk_off;   // EncCtx::nz_blk_off
};
void launch_trellis_sort(hipStream_t st, const TrellisCtx &c);   // fills TrellisCtx::perm from ::blk_cnt
void launch_trellis_ac(hipStream_t st, const TrellisCtx &c);
void launch_trellis_dc(hipStream_t st, const TrellisCtx &c);
size_t trellis_spill_words();   // size of TrellisCtx::spill in u32

// generic device primitive: out[i] = sum_{j<i} in[j] for i in [0, n]  (n+1 outputs; in[] has n entries)
void launch_exclusive_scan(hipStream_t st, const uint32_t *in, uint64_t *out, uint64_t n, void *tmp, size_t tmp_bytes);
size_t exclusive_scan_tmp_bytes(uint64_t n);

// ---- phase 6: byte stuffing + file assembly (k_assemble.hip)
struct AsmCtx {
    const ImgDesc *imgs;
    const EncScan *script;
    ScanWork *work;               // raw_off / raw_bytes / out_off / hdr_bytes are filled in here
    int nwork, nimg, scans_per_image;
    int work0, nwork_run;         // the work items this run of the per-scan kernels covers (a stage of the scan search; everything otherwise)
    const uint32_t *img_list;     // [nimg][CSH_LIST_MAX] the work items that make up each file, in file order
    const uint32_t *img_nlist;    // [nimg]
    uint32_t *scan_cost;          // [nwork] bytes a scan puts into a file: DHT + SOS + stuffed data (what mozjpeg's scan search compares)
    const DevEncTable *tables;
    const uint64_t *chunk_off;    // exclusive scan of the chunks' bit sizes (+ trailing total)
    uint32_t *scan_pad_bytes;     // [nwork] raw bytes per scan rounded up to 64 (+64 slack)
    uint64_t *scan_raw_off;       // [nwork+1] exclusive scan of scan_pad_bytes
    const uint32_t *raw;          // packed bits, big-endian-logical u32 words
    uint64_t raw_chunks;          // capacity of raw in 64-byte chunks
    uint32_t *chunk_ff;           // [raw_chunks] number of 0xFF bytes in the chunk (k_ff_count); after k_ff_prefix, for the scans of a file: of the chunks in front of it
    const uint8_t *hdr_pool;      // per-image frame headers (host-built: SOI .. SOFn)
    const uint32_t *hdr_off;      // [nimg+1]
    uint32_t *img_size_pad;       // [nimg] file size rounded up to 16
    uint32_t *img_size;           // [nimg] file size
    uint64_t *img_off;            // [nimg+1] exclusive scan of img_size_pad
    uint8_t *out;                 // output pool
    uint64_t out_cap;
    uint32_t *status;             // per image
    uint32_t *overflow;           // [1] set when a pool is too small
};
void launch_scan_cost(hipStream_t st, const AsmCtx &a);      // scan_cost of the run's work items (after launch_ff_count)
void launch_scan_sizes(hipStream_t st, const AsmCtx &a);     // fills scan_pad_bytes + work[].raw_bytes
void launch_scan_place(hipStream_t st, const AsmCtx &a);     // work[].raw_off from scan_raw_off
void launch_ff_count(hipStream_t st, const AsmCtx &a);
void launch_layout(hipStream_t st, const AsmCtx &a);         // per image: scan offsets, file size
void launch_emit(hipStream_t st, const AsmCtx &a);           // headers, DHT/SOS, stuffed data, EOI

}  // namespace csh
