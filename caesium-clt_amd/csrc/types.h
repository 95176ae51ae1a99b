// types.h -- descriptors shared by the host pipeline and the kernels.
//
// HBM data layout (DESIGN.md "Data layout"):
//  * coefficient TILE = 64 consecutive blocks (padded raster order of one component of one image), 8 KiB,
//    stored as 8 "octets": int16 tile[8 /*k>>3*/][64 /*block in tile*/][8 /*k&7*/].  A wave owns a tile,
//    lane = block: octet j of every block is one 16-byte vector per lane = one fully coalesced 1 KiB access per
//    wave (8 such loads/stores move a tile); a band-limited progressive scan (Ss..Se) touches only octets
//    Ss>>3..Se>>3; a block's coefficients touch at most 8 cache lines (shared with 7 neighbour blocks).
//  * per-block 64-bit planes (bit k: |coef k| >= 2^l, bit l of |coef k|, sign) exist only in registers (k_entropy.hip): the
//    progressive coder's run/EOB logic is bit algebra on them.
//  * u8 sample planes (subsampled components only), pitch = real_bw*8, rows = bh*8, edges replicated.
#pragma once
#include <cstdint>

#define CSH_TILE_BLOCKS 64
#define CSH_TILE_I16 4096  // int16 elements per tile
// position of coefficient k (zig-zag) of block `lane` inside a tile: lane*CSH_BLK_STRIDE + (k>>3)*CSH_OCT_STRIDE + (k&7)
#ifndef CSH_BLOCK_MAJOR
#define CSH_BLOCK_MAJOR 0
#endif
#if CSH_BLOCK_MAJOR      // int16 tile[64 blocks][64 k]: one block = one 128-byte line
#define CSH_OCT_STRIDE 8
#define CSH_BLK_STRIDE 64
#else                    // int16 tile[8 k>>3][64 blocks][8 k&7]: a wave's octet loads are contiguous
#define CSH_OCT_STRIDE 512
#define CSH_BLK_STRIDE 8
#endif
#define CSH_MAX_COMPS 3
// per-block 64-bit planes (bit = zig-zag index), SoA per tile u64[CSH_MASK_PLANES][64]:
// 0..2 significance |c| >= 1, 2, 4;  3, 4 = bit 0, bit 1 of |c|;  5 = sign.  Refinement scans are coded from these alone.
#define CSH_MASK_PLANES 6
#define CSH_MASK_TILE (CSH_MASK_PLANES * 64)
#define CSH_MAX_SCANS 20
#define CSH_LIST_MAX 16   // scans of one output file (the scan search ends with at most 14)

namespace csh {

struct CompGeom {
    int h, v;              // sampling factors
    int comp_w, comp_h;    // real size in samples
    int real_bw, real_bh;  // ceil(comp/8)
    int bw, bh;            // MCU-padded block grid
    int ntiles;            // ceil(bw*bh/64)
    uint32_t tile_base;    // first tile of this component in the coefficient pool
};

// decode LUT for one Huffman table (T.81 Annex C / F.2.2.3 with a 9-bit lookahead)
struct DevHuff {
    uint16_t look[512];   // (len<<8)|sym for codes of <= 9 bits, else 0
    int32_t maxcode[18];  // per length, -1 if none
    int32_t valptr[17];
    uint8_t vals[256];
};
struct DevHuffSet { DevHuff dc[4], ac[4]; };

// the same tables as the parallel decoder wants them in LDS (k_decode_par.hip): two-level look-up.
// root[t][top 9 bits]: (len<<8)|sym for a code of <= 9 bits; 0x8000 | nbits<<12 | offset for a 9-bit prefix shared by longer
// codes -- then sub[offset + next nbits bits] holds (len<<8)|sym (len 10..16); 0 = no such code.  The sub-tables of all
// eight tables share one pool; a set that does not fit (possible only with pathological tables) sends its images to the
// sequential decoder.
#define CSH_PAR_SUB 2048
struct alignas(16) ParHuffSet {
    uint16_t root[8][512];   // 0..3 DC tables, 4..7 AC tables
    uint16_t sub[CSH_PAR_SUB];   // directly behind root (the kernels address it as sizeof(root))
};
// compact form for the self-synchronising decoder: almost every file defines four tables (two DC, two AC), and 6 KB instead
// of 12 KB of LDS is what lets a fourth workgroup onto each CU.  Slots are assigned per table set in order of presence;
// ParScan::sel holds slot numbers.  A batch uses this form when ALL its table sets fit it.
#define CSH_PAR_SUB4 1024
struct alignas(16) ParHuffSet4 {
    uint16_t root[4][512];
    uint16_t sub[CSH_PAR_SUB4];
};
// where block m of an MCU goes (write pass), one entry per block-in-MCU index
// block index in the component = my*row_step + mx*col_step + first; DC difference slot = mcu*dc_per_mcu + dc_first
struct ParBlockInfo { uint32_t tile_base; int row_step, col_step, first; uint32_t dc_first, dc_per_mcu; };

// one entropy-coded scan of one input image
struct DecScan {
    uint32_t bits_off, bits_len;  // entropy segment in the bitstream pool
    int huff_set;
    int ncomp;                    // components in scan
    int comp[CSH_MAX_COMPS], td[CSH_MAX_COMPS], ta[CSH_MAX_COMPS];
    int Ss, Se, Ah, Al;
    int restart_interval;
    int par_index;                // index of this scan's ParScan (unstuffed copy of the segment), -1 if none
};

// progressive input: a chain = the scans of one image that must run in file order because they touch the same
// coefficients -- all DC scans, or all AC scans of one component.  Chains of an image are independent of each other.
// refine != 0: every scan of the chain is an AC refinement scan of component `comp` (k_decode_refine.hip: parse + apply); its nblocks history masks
// start at hist_off, its count * nblocks block positions at pos_off.  Otherwise the chain runs in one wave of k_decode_prog.hip.
struct RefineUnit { int chain, s, prev; };   // one scan of a refinement chain = one wave of k_refine_parse; prev = the unit of the chain's scan s - 1, or -1
struct ProgChain { int image, first, count, refine, comp; uint32_t nblocks, hist_off, pos_off; };   // chain_scans[first .. first+count) = DecScan indices

struct ImgDesc {
    int width, height, ncomp;
    int mcus_x, mcus_y;           // MCU grid of the INPUT frame (entropy decode)
    int omcus_x, omcus_y;         // MCU grid of the OUTPUT frame (entropy encode)
    int progressive_in;
    CompGeom in[CSH_MAX_COMPS];   // decoded coefficient geometry
    CompGeom out[CSH_MAX_COMPS];  // re-encoded geometry (== in for the lossless transcode)
    int qt_in[CSH_MAX_COMPS];     // index into the quant pool (zig-zag u16[64] + divisors)
    int qt_out[CSH_MAX_COMPS];
    uint64_t plane_off[CSH_MAX_COMPS];  // byte offset of the decoded u8 plane (components that are resampled)
    uint64_t oplane_off[CSH_MAX_COMPS]; // byte offset of the resampled u8 plane (encoder-side geometry)
    // planes that FEED the encoder-side resample: the decoded planes (src == in, enc == image size), or -- resize path --
    // full-resolution planes of the resized image (k_resize.hip)
    CompGeom src[CSH_MAX_COMPS];
    uint64_t splane_off[CSH_MAX_COMPS];
    int enc_w, enc_h;
    int first_scan, nscans_in;    // range in the DecScan array
    int comp_id[CSH_MAX_COMPS];   // component identifiers written to SOF/SOS
    int first_work, nscans_out;   // host bookkeeping: the image's first work item and the number of them (the file's scans are EncCtx-independent: AsmCtx::img_list)
    uint32_t status;              // 0 ok; set by kernels on malformed data
};

// one sequential-mode scan handled by the parallel (self-synchronising) decoder, k_decode_par.hip
#define CSH_SUBSEQ_BYTES 128
struct ParScan {
    uint32_t bits_off, bits_len;  // stuffed entropy segment in the bitstream pool (64-byte aligned)
    uint32_t clean_len;           // unstuffed length (device-computed)
    int huff_set, image, ncomp;
    int nb_mcu;                   // blocks per MCU (1 for a non-interleaved scan)
    int comp_of[10], by_of[10], bx_of[10], dct[10], act[10];
    uint32_t total_blocks;        // blocks the scan must produce
    uint32_t sub_base;            // first sub-sequence of this scan in the flat per-sub-sequence arrays
    uint32_t par_index;           // index among ParScans (state arrays hold nsub+1 entries per scan)
    uint32_t dc_base[10], dc_per_mcu[10], dc_idx[10];  // where block m's DC difference goes (scan order, per component)
    uint64_t sel;                 // table selectors, 6 bits per block-in-MCU index m: dct[m] | (4 + act[m]) << 3
    uint32_t first_mcu;           // restart interval: MCU (or block, non-interleaved) of the scan this segment starts at
    int kind;                     // what the segment is to the self-synchronising decoder (k_decode_par.hip):
                                  //   0 sequential-mode scan (whole blocks);  1 listed only to be unstuffed (progressive AC refinement: k_decode_prog.hip);
                                  //   2 progressive DC first scan (one DC symbol per block);  3 progressive AC first scan (band Ss..Se of one component,
                                  //   EOB runs);  4 progressive DC refinement (unstuffed, then one bit per block: k_dc_refine)
    int Ss, Se, Al;               // kinds 2, 3, 4
};
enum { CSH_PS_SEQ = 0, CSH_PS_UNSTUFF = 1, CSH_PS_DC_FIRST = 2, CSH_PS_AC_FIRST = 3, CSH_PS_DC_REFINE = 4 };

// quantisation table as the kernels want it: zig-zag order, with exact-division helpers
struct DevQuant {
    uint16_t q[64];      // table value, zig-zag order
    int32_t div[64];     // q*8 (jfdctint output is scaled by 8)
    float rcp[64];       // fl((1 / div) (1 + 2^-19)), host-computed in double: the quantiser's one fma (k_pixel.hip quant_one)
    float lt[64];        // 1 / q^2 as float(1.0 / double(q * q)): the distortion weight of mozjpeg's trellis (k_trellis.hip), host-computed
    // exact a / div for 0 <= a < 2^17 on the full-rate 24-bit multiplier: (a << sh) * mul >> 32  (v_mul_hi_u32_u24).  With P = 32 - sh =
    // max(25, floor(log2 div) + 18) and mul = floor(2^P / div) + 1: the estimate exceeds a / div by less than 1 / div (a * div < 2^P), so its
    // floor is the quotient's; a << sh < 2^24 (sh <= 7) and mul < 2^24 (P - log2 div < 24).  Valid for div < 2^14 (8-bit tables: div <= 2040);
    // mul = 0 marks a table the kernels must not quantise with (16-bit input tables: they only dequantise)
    uint32_t mul[64];
    uint32_t sh[64];
};

// work item of the pixel kernels: one component of one image
struct PlaneWork {
    int image, comp;
    int mode;  // 0: full-res in and out (IDCT->FDCT in one lane); 10: h2v2 kept, no resize (k_resample_fdct_420); else 1 + 3*in_kind + out_kind with kinds 0 full, 1 h2v2, 2 h2v1
};

// resize path work item (k_resize.hip): one image
struct ResizeWork {
    int image, in_kind;            // in_kind: how the decoded chroma planes relate to full resolution (0 full, 1 h2v2, 2 h2v1)
    int nw, nh;                    // new size
    uint64_t rgb_src_off, rgb_dst_off;  // byte offsets in the RGB pool (W*H*nc source, nw*nh*nc resized)
    uint64_t tmp_off;              // float offset in the f32 pool (nh*W*nc)
    uint32_t vtap_base, htap_base; // first ResizeTap of the vertical (nh entries) / horizontal (nw entries) pass
};
struct ResizeTap { int left, n; uint32_t woff; };  // output coordinate -> first source index, #taps, offset of its weights

// one scan of the OUTPUT script (same for every image of the batch with equal ncomp)
struct EncScan {
    int ncomp, comp[CSH_MAX_COMPS];
    int Ss, Se, Ah, Al;
    int dc_tbl[CSH_MAX_COMPS];  // per scan component: index of its DC table inside this scan's table group
    int ac_tbl[CSH_MAX_COMPS];  // sequential scans only: index of the component's AC table inside the group
    int sequential;              // 1: sequential-mode scan (whole blocks, Huffman DC+AC+EOB; --jpeg-baseline)
    int ntables;                 // tables this scan defines (0 for DC refine)
    int dht_id[4];               // (Tc<<4)|Th of each table of the group, in DHT emission order
    int sos_tdta[CSH_MAX_COMPS]; // (Td<<4)|Ta byte of each scan component
};

// per (image, output scan) bookkeeping
struct ScanWork {
    int image, scan;
    uint32_t nunits;       // lanes of work: blocks (AC, non-interleaved) or MCUs (interleaved DC)
    uint32_t unit_base;    // offset of this scan in the flat per-unit arrays
    uint32_t first_chunk;  // first 256-unit chunk of this scan in the flat chunk list (EncCtx::chunk_work)
    uint32_t word_base;    // offset in the u64 bitmask arrays (AC scans), nwords = ceil(nunits/64)
    uint32_t table_base;   // first of this scan's Huffman tables in the table pool
    uint32_t ff_bytes;     // 0xFF bytes among raw_bytes = bytes the stuffing adds (device-computed, k_ff_count)
    uint64_t raw_off;      // byte offset (multiple of 64) of the scan's unstuffed bytes in the raw pool (device-computed)
    uint32_t raw_bytes;    // unstuffed length in bytes (device-computed)
    uint32_t out_off;      // offset of this scan's DHT marker inside the image's output file (device-computed)
    uint32_t hdr_bytes;    // DHT + SOS bytes in front of the entropy-coded data (device-computed)
    uint32_t no_room;      // the raw pool cannot hold this scan (device-computed): the packer skips it, the batch is re-run with a larger pool
    uint32_t list;         // progressive AC scans: the NzList (component, point transform) the scan is coded from (k_aclist.hip); 0xFFFFFFFF otherwise
    uint32_t corr_base;    // refinement scans: the scan's first correction word in EncCtx::corr (one u64 per unit of a REFINEMENT scan only: 8 bytes for every unit of
                           // every candidate scan were 7.5 MB of a 1080p picture's 37 MB of pools); 0xFFFFFFFF otherwise
    uint32_t hist_row0;    // first histogram row of its first slot (slot j: hist_row0 + j * ntables)
    uint32_t ls_base;      // where its slots stand in the list of list-coded slots (list != 0xFFFFFFFF) or of token-coded slots
};

// one workgroup of the token kernel (k_tokens): 256 consecutive units, [256 j, 256 j + 256)
//   kind 0: ALL progressive AC scans of component `comp` of image `a` (the block's coefficients are read once for all of them)
//   kind 1: the one scan of work item `a` (DC scans: unit = MCU or block; sequential-mode scans)
// Whatever the kind, the tokens of (work item w, chunk j) belong to slot w.first_chunk + j of the per-chunk arrays.
struct EChunk { uint32_t a; uint16_t comp, kind; uint32_t j; uint32_t plan; uint32_t region; };
// the token pool is cut into regions with a bump cursor each -- one per (image, component) and one per DC / sequential scan: a single
// cursor for the whole batch would be one L2 address taking a million atomics, one after the other
struct TokRegion { uint64_t base; uint32_t cap, pad; };
#define CSH_TK_MAXSLOT 12  // AC scans of one component that a kind-0 chunk can carry (a stage of the scan search has 11)
// what a kind-0 chunk needs of its image, component and scans, in one piece (host-built per (image, component); the workgroup copies it
// to LDS with one coalesced load instead of chasing ImgDesc -> ScanWork -> EncScan through dependent scalar loads, scan after scan)
struct AcSlot { uint32_t unit_base, word_base, first_chunk, table_base, nunits_work; uint8_t Ss, Se, Ah, Al; uint32_t corr_base, pad; };   // 32 bytes; corr_base: the scan's first correction word (ScanWork)
struct TokPlan {
    uint32_t nslot, nunits;        // AC scans of the component; its blocks
    int32_t real_bw, bw;           // block grid: real and MCU-padded width
    uint32_t tile_base;
    uint32_t work0;                // first work item of these scans (EncCtx::work_active: a gated stage codes or skips a plan's scans together)
    uint32_t pad[2];
    AcSlot s[CSH_TK_MAXSLOT];
};
// ---- the compacted coefficient lists (k_aclist.hip): what every progressive AC scan is coded from.
// One NzList per (image, component, point transform Al): for every real block, in scan order, one entry per coefficient k = 1..63 with
// |c_k| >> Al != 0, then one END entry.  An entry is one u32:
//   [6:0] k (1..63; 64 = END; 0 = padding, codes nothing)   [7] sign (1: negative)   [22:8] |c_k| >> Al   [30:23] block inside its 256-block chunk
// The entries of a 256-block chunk are contiguous and start on a 16-byte boundary (chunk_off / chunk_cnt per (list, chunk)); chunks of a
// list are placed with a bump cursor inside the list's region of the pool, like the token regions.
#define CSH_NZ_END 64u
#define CSH_NZ_LEVELS 4   // Al 0..3 (the scan search tries luma up to Al 3)
struct NzList {
    uint64_t base;         // first entry of the list's region in the pool
    uint32_t cap;          // entries the region holds
    uint32_t chunk0;       // first of the list's (nunits + 255) / 256 records in the per-chunk arrays
};
// the lists of one component of one image, and what the builder needs to read its blocks
struct NzSet {
    uint32_t list[CSH_NZ_LEVELS];   // NzList index per Al, 0xFFFFFFFF: not kept
    uint32_t nunits;                // real blocks of the component
    int32_t real_bw, bw;            // block grid: real and MCU-padded width
    uint32_t tile_base;
    uint32_t cnt_base;              // where the builder leaves every block's number of non-zero AC coefficients (EncCtx::nz_blk_cnt: the trellis stage sorts its blocks by it); 0xFFFFFFFF: nowhere
    uint32_t pad[3];
};
// one wave of the builder: chunk j of set `set`, the levels in `levels` (bit Al).  Level 0 is made from the coefficients; the others are
// filtered from level 0 (which an earlier stage may have made)
#define CSH_NZ_COMPACT0 0x80000000u   // NzChunk::levels: level 0 exists (the trellis stage's statistics list, the trellis's levels written into it): drop its zero entries in place, filter the other levels from the result
struct NzChunk { uint32_t set, j, levels, work0; };   // work0: a work item whose gate (EncCtx::work_active) stands for the chunk in a conditional stage

// what the per-slot kernels need of (work item, chunk j), in one load (written by k_make_slots from the work items; slot = work.first_chunk + j)
struct SlotRec {
    uint32_t work, j, nch;       // work item, chunk number, chunks of the work item
    uint32_t first_chunk;        // the work item's first slot
    uint32_t unit0, nun;         // first unit (index into the per-unit arrays) and number of units of the chunk
    uint32_t table_base;
    uint16_t ntables, flags;     // flags: 1 progressive AC scan (EOB tokens), 2 refinement (correction words), 4 coded from its NzList (k_aclist.hip), not from tokens
    uint32_t hist_row;           // first of the slot's ntables rows of 256 symbol counts (EncCtx::slot_hist)
    uint32_t word_base, unit_base, nunits_work;   // of the work item (k_ac_runs)
    uint8_t Ss, Se, Ah, Al; uint32_t corr0;           // corr0: the chunk's first correction word (refinement scans)
    uint32_t nzlist, nzrec;      // list-coded slots: the NzList and the chunk's record in the per-chunk arrays (nz_chunk_off / nz_chunk_cnt)
};

// mozjpeg's trellis quantiser (k_trellis.hip; CSH_PROFILE=mozjpeg): one work item per (image, component) -- the component's statistics
// scan (a ScanWork of the trellis stage) supplies the rate tables -- and one chunk of the AC kernel per 256 of its blocks
struct TrellisWork {
    int image, comp;
    uint32_t table_ac;       // DevEncTable of the statistics pass: code lengths of the AC symbols
    int32_t table_dc;        // sequential output: the optimal DC table of the same pass; -1: the Annex K table of the component (progressive)
    uint32_t nunits;         // real blocks of the component
    uint32_t unit_base;      // first entry of the component in the per-block side arrays (lambda, DC back-pointers)
    uint32_t nzset;          // the component's NzSet (its level-0 list takes the chosen levels: TrellisCtx::nz_pool), 0xFFFFFFFF: none
    uint32_t pad[1];
};
struct TrellisChunk { uint32_t work, j; };

// encoder-side Huffman table as generated on the device
struct DevEncTable {
    uint32_t freq[257];
    uint8_t bits[17];
    uint8_t vals[256];
    uint16_t code[256];
    uint8_t size[256];
    int nsym;
    uint32_t lut[256];   // size << 16 | code, what the packer gathers
};

}  // namespace csh
